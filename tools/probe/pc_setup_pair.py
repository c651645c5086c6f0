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
if os.environ.get("PA_PAIR_LINES", "1") == "1": os.environ["PA_SETUP_TIMING"] = "1"
import collections
import pa_amd._lib as L
per_call = collections.defaultdict(lambda: [0, 0.0])
_call = L.call
def timed_call(name, *a):
    t0 = time.perf_counter()
    try:
        return _call(name, *a)
    finally:
        e = per_call[name]; e[0] += 1; e[1] += time.perf_counter() - t0
        if os.environ.get("PA_PAIR_EACH") and time.perf_counter() - t0 > 2e-3: print(f"      {name} {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
L.call = timed_call
for m in list(sys.modules.values()):
    if m is not None and getattr(m, "__name__", "").startswith("pa_amd") and getattr(m, "L", None) is L: pass
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
    for k, (n_, t_) in sorted(per_call.items(), key=lambda kv: -kv[1][1])[:18]: print(f"   {t_ * 1e3:8.1f} ms  {n_:4d} x  {k}")
    per_call.clear()
