"""Host side of an MG-PCG iteration at n^3: time to enqueue against time to finish, and where the host time goes."""
import os, sys, time, cProfile, pstats
root, tag = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
sys.path.insert(0, root)
from __graft_entry__ import load_package
pa = load_package()
S = pa.pc_setup(pa.DebugArray([1]), 1, 4, n, n, n, "multicolor_spmv")
A, b = S.A_vec[-1], S.r[-1]
pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=25, Pl=S, fuse=True)
for rep in range(2):
    x = pa.pzeros(A.col_partition)
    pa.context().sync()
    t = time.perf_counter()
    pa.opt_cg_(x, A, b, maxiter=30, Pl=S, fuse=True)
    t1 = time.perf_counter()
    pa.context().sync()
    t2 = time.perf_counter()
    print(f"[{tag}] {n}^3: returned after {(t1 - t) / 30 * 1e3:.3f} ms per iteration, finished after {(t2 - t) / 30 * 1e3:.3f} ms", flush=True)
pr = cProfile.Profile(); pr.enable()
pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=30, Pl=S, fuse=True)
pa.context().sync()
pr.disable()
import io
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats('tottime').print_stats(14)
print("\n".join(f"[{tag}] " + l for l in s.getvalue().splitlines() if l.strip())[:6000])
