"""A Q1 FEM mesh numbered at random, one part: pa_spmv as the mesher numbered it, and after renumber_for_locality (device RCM)."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
import bench
ctx = pa.context()
os.environ.setdefault("PA_SPMV_VALUE_DICT", "0")
for nxm, nym in ((1000, 800), (3000, 2000)):
    nn = nxm * nym
    r1 = pa.DebugArray([1])
    I, J, V, rows, cols = pa.laplacian_fem((nxm, nym), (1, 1), r1)
    perm = np.random.default_rng(29).permutation(nn) + 1
    Ip, Jp, Vp = perm[I.items[0] - 1], perm[J.items[0] - 1], V.items[0]
    A = pa.psparse_from_coo(pa.DebugArray([Ip]), pa.DebugArray([Jp]), pa.DebugArray([Vp]), pa.uniform_partition(r1, (1,), (nn,)))
    b0 = pa.local_items(A.matrix_partition)[0].own_own
    ms0 = bench.time_block(pa, ctx, L, b0, nn, nn)
    ctx.sync(); t = time.perf_counter()
    A2 = pa.renumber_for_locality(A)
    ctx.sync(); t = time.perf_counter() - t
    b1 = pa.local_items(A2.matrix_partition)[0].own_own
    ms1 = bench.time_block(pa, ctx, L, b1, nn, nn)
    alg = b0.nnz * 12 + nn * 20
    print(json.dumps({"mesh": [nxm, nym], "nnz": b0.nnz, "ms_random": round(ms0, 4), "gbps_random": round(alg / ms0 / 1e6, 1), "ms_renumbered": round(ms1, 4),
                      "gbps_renumbered": round(alg / ms1 / 1e6, 1), "renumber_s": round(t, 2), "band": pa.local_items(A2.bandwidths)[0], "xwin": b1.xwin(),
                      "encoding": b1.encoding()}), flush=True)
