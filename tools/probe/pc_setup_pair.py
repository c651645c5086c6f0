"""The two set-ups tools/hpcg_driver.py times (reference ordering with raw columns kept; the optimised solver cut from that
hierarchy), pool in hand, each under cProfile with the library's PA_SETUP_TIMING lines: where 0.40 + 0.32 s go."""
import cProfile, gc, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
ctx = pa.context()
r1 = pa.DebugArray([1])
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
mk_ref = lambda: pa.pc_setup(r1, 1, 4, n, n, n, ordering="sequential", keep_raw_columns=True)
mk_opt = lambda prev: pa.pc_setup(r1, 1, 4, n, n, n, ordering="multicolor_spmv", graph=True, reuse=prev, keep_raw_columns=True)
S = mk_ref(); T = mk_opt(S); ctx.sync(); del S, T; gc.collect()          # pool, code objects
os.environ["PA_SETUP_TIMING"] = "1"
for name, f in (("reference", mk_ref), ("optimised", None)):
    pr = cProfile.Profile()
    print(f"==== {name} starts", file=sys.stderr, flush=True)
    ctx.sync(); t = time.perf_counter(); pr.enable()
    if name == "reference": S = f()
    else: T = mk_opt(S)
    ctx.sync(); pr.disable()
    print(f"==== {name}: {time.perf_counter() - t:.3f} s", flush=True)
    print(f"==== {name} ends", file=sys.stderr, flush=True)
    pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
