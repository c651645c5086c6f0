"""Where build_split_blocks_device spends its time (one part, n^3)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
orig = L.call
acc = {}
def timed(name, *a):
    pa.context().sync() if name.startswith("pa_h") or name.startswith("pa_csr") else None
    t = time.perf_counter()
    r = orig(name, *a)
    acc[name] = acc.get(name, 0.0) + time.perf_counter() - t
    return r
for rep in range(3):
    acc.clear()
    L.call = timed
    import pa_amd.gallery as G, pa_amd.p_sparse_matrix as P, pa_amd.p_vector as V
    G.L.call = timed
    t = time.perf_counter()
    A, b = pa.build_p_matrix(pa.DebugArray([1]), n, n, n, n, n, n, 1, 1, 1)
    pa.context().sync()
    dt = time.perf_counter() - t
    print(f"rep {rep}: build_p_matrix {dt:.3f} s;", ", ".join(f"{k} {v * 1e3:.0f} ms" for k, v in sorted(acc.items(), key=lambda kv: -kv[1])[:8]), flush=True)
    del A, b
