"""A vector-valued problem with interleaved unknowns (2 per node of a Q1 grid: the sparsity of 2-D elasticity): rows alternate
between two patterns, which the row-pattern encoder (runs of consecutive rows with ONE pattern) does not describe."""
import os, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import scipy.sparse as sp
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
ctx = pa.context()

def rate(name, H):
    x = pa.DeviceVector(H.n, 0).upload(np.random.default_rng(1).standard_normal(H.n))
    for sw in ("0", None):
        if sw is None: os.environ.pop("PA_SPMV_XWIN", None)
        else: os.environ["PA_SPMV_XWIN"] = sw
        blk = pa.DeviceCSR(H)
        y = pa.DeviceVector(H.m, 0)
        for _ in range(60): pa.spmv_(y, blk, x)
        e0 = ctx.event().record(L.STREAM_COMPUTE)
        for _ in range(50): pa.spmv_(y, blk, x)
        e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
        ms = e0.elapsed_ms(e1) / 50
        alg = (H.nnz * 12 + H.m * 20) / 1e6
        print(f"{name:44s} XWIN={sw}: {ms:7.4f} ms {alg/ms:6.0f} GB/s alg  {2*H.nnz/ms/1e6:6.0f} GFLOP/s  {blk.encoding()} {blk.xwin()}", flush=True)
        del blk, y

for dims, dof in (((1000, 1000), 2), ((1600, 1250), 2), ((100, 100, 100), 3)):
    I, J, V, rows_, cols_ = pa.laplacian_fem(dims, (1,) * len(dims), pa.DebugArray([1]))
    n = int(np.prod(dims))
    A = sp.csr_matrix((V.items[0], (I.items[0] - 1, J.items[0] - 1)), shape=(n, n))
    B = sp.kron(A, np.arange(1.0, dof * dof + 1).reshape(dof, dof), format="csr")
    B.sort_indices()
    H = pa.HostCSR(n * dof, n * dof, (B.indptr + 1).astype(np.int32), (B.indices + 1).astype(np.int32), B.data.astype(np.float64))
    rate(f"Q1 grid {dims}, {dof} interleaved unknowns per node", H)
