"""Power-of-two grids: with nx = 256 the three lines of a plane a row gathers from start 2 KiB apart (notebook R4.13).  A permutation that ROTATES every grid
line inside itself by a line-dependent multiple of a few cache lines moves them apart without any padding.  Times the 27-point 256^3 product on the block as
generated and on pa_csr_create_permuted twins.  python tools/probe/rotated_lines.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("PA_SPMV_VALUE_DICT", "0")
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
ctx = pa.context()
n = int(os.environ.get("N", "256"))
A, _ = pa.build_p_matrix(pa.DebugArray([1]), n, n, n, n, n, n, 1, 1, 1)
blk = A.matrix_partition.items[0].own_own
N = n ** 3
xh = np.random.default_rng(1).standard_normal(N)

def rate(b, xhost, tag):
    x, y = pa.DeviceVector(N, 0).upload(xhost), pa.DeviceVector(N, 0)
    for _ in range(300): pa.spmv_(y, b, x)
    best = []
    for rep in range(3):
        e0 = ctx.event().record(L.STREAM_COMPUTE)
        for _ in range(50): pa.spmv_(y, b, x)
        e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
        best.append(e0.elapsed_ms(e1) / 50)
    print(f"{tag:44s} {min(best):.4f} ms = {2 * b.nnz / min(best) / 1e6:6.0f} GFLOP/s  {b.encoding()}", flush=True)
    return y.download()

y0 = rate(blk, xh, "as generated")
idx = np.arange(N, dtype=np.int64)
i, j, k = idx % n, (idx // n) % n, idx // (n * n)
for name, rot in (("32 * ((j + 3k) & 7)", 32 * ((j + 3 * k) & 7)), ("16 * (j & 15)", 16 * (j & 15)), ("64 * (j & 3)", 64 * (j & 3)), ("8 * ((j + 5k) & 31)", 8 * ((j + 5 * k) & 31))):
    pos = (((i + rot) % n) + n * j + n * n * k).astype(np.int32)
    q = C.c_void_p()
    L.call("pa_csr_create_permuted", blk.h, L.ptr(pos), L.ptr(pos), C.byref(q))
    Q = pa.DeviceCSR.from_handle(q, N, N, blk.nnz)
    xq = np.zeros(N); xq[pos] = xh
    yq = rate(Q, xq, "lines rotated by " + name)
    print("   same bits:", np.array_equal(yq[pos], y0), flush=True)
    del Q
