"""Which library calls pc_setup spends its time in (256^3, one part), per ordering; second pass of each (warm)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L, pa_amd.hpcg as H, pa_amd.gallery as G, pa_amd.p_sparse_matrix as P, pa_amd.p_vector as V
orig = L.call
acc = collections.defaultdict(float); cnt = collections.defaultdict(int)
def timed(name, *a):
    t = time.perf_counter(); r = orig(name, *a); acc[name] += time.perf_counter() - t; cnt[name] += 1; return r
for m in (L, H.L, G.L, P.L, V.L): m.call = timed
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for ordering in ("multicolor_spmv", "sequential", "multicolor_spmv", "sequential"):
    acc.clear(); cnt.clear()
    pa.context().sync(); t = time.perf_counter()
    S = pa.pc_setup(pa.DebugArray([1]), 1, 4, n, n, n, ordering=ordering)
    pa.context().sync(); dt = time.perf_counter() - t
    top = sorted(acc.items(), key=lambda kv: -kv[1])[:9]
    print(f"{ordering}: {dt:.3f} s, in library calls {sum(acc.values()):.3f} s: " + ", ".join(f"{k} x{cnt[k]} {v * 1e3:.0f} ms" for k, v in top), flush=True)
    del S
