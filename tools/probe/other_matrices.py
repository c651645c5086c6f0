"""SpMV rates of K1 on matrices other than the HPCG operator (one part, one GPU)."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
ctx = pa.context()

def rate(name, blk, n_rows, n_cols):
    x = pa.DeviceVector(n_cols, 0).upload(np.random.default_rng(1).standard_normal(n_cols))
    y = pa.DeviceVector(n_rows, 0)
    for _ in range(5): pa.spmv_(y, blk, x)
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    for _ in range(30): pa.spmv_(y, blk, x)
    e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
    ms = e0.elapsed_ms(e1) / 30
    nnz = blk.nnz
    print(f"{name:46s} rows {n_rows:>10d} nnz {nnz:>11d}  {ms:8.4f} ms  {2*nnz/ms/1e6:7.1f} GFLOP/s  {(nnz*12 + n_rows*20)/ms/1e6:7.1f} GB/s alg  enc {blk.encoding()}", flush=True)

ranks = pa.DebugArray([1])
n = 160
I, J, V, rows, cols = pa.laplacian_fem((n, n, n), (1, 1, 1), ranks)
A = pa.psparse_disassembled(I, J, V, rows, cols)
rate(f"Q1 FEM Laplacian 3-D {n}^3 nodes", A.matrix_partition.items[0].own_own, n ** 3, n ** 3)
n = 3000
I, J, V, rows, cols = pa.laplacian_fem((n, n), (1, 1), ranks)
A = pa.psparse_disassembled(I, J, V, rows, cols)
rate(f"Q1 FEM Laplacian 2-D {n}^2 nodes", A.matrix_partition.items[0].own_own, n * n, n * n)
I, J, V, rows, cols = pa.laplacian_fdm((200, 200, 200), (1, 1, 1), ranks)
A = pa.psparse_from_coo(I, J, V, rows)
rate("7-point FDM Laplacian 200^3", A.matrix_partition.items[0].own_own, 200 ** 3, 200 ** 3)
rng = np.random.default_rng(0)
m = 4_000_000
for name, width in (("random columns within +-2000 of the diagonal", 2000), ("random columns anywhere", m)):
    lens = np.full(m, 16)
    rp = np.concatenate([[1], 1 + np.cumsum(lens)]).astype(np.int32)
    base = np.repeat(np.arange(m), 16)
    if width < m:
        col = np.clip(base + rng.integers(-width, width, size=m * 16), 0, m - 1)
    else:
        col = rng.integers(0, m, size=m * 16)
    col = np.sort(col.reshape(m, 16), axis=1).ravel().astype(np.int32) + 1
    H = pa.HostCSR(m, m, rp, col, rng.standard_normal(m * 16))
    rate(f"4M rows x 16: {name}", pa.DeviceCSR(H), m, m)
