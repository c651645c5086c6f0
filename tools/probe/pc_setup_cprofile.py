"""cProfile of the THIRD pc_setup (4 levels, 256^3, multicolour, library defaults): where the host spends the set-up."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
ctx = pa.context()
r1 = pa.DebugArray([1])
ordering = sys.argv[1] if len(sys.argv) > 1 else "multicolor_spmv"
for k in range(2):
    S = pa.pc_setup(r1, 1, 4, 256, 256, 256, ordering=ordering); ctx.sync(); del S
pr = cProfile.Profile()
ctx.sync(); t = time.perf_counter()
pr.enable()
S = pa.pc_setup(r1, 1, 4, 256, 256, 256, ordering=ordering)
ctx.sync()
pr.disable()
print(f"{ordering}: {time.perf_counter() - t:.3f} s")
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
