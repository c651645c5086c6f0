"""SELL-C-sigma (one lane per row) against the row-split kernel: 27-point 128^3 and banded-unstructured rows."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
ctx = pa.context()

def rate(name, blk, n_rows, n_cols, nnz):
    x = pa.DeviceVector(n_cols, 0).upload(np.random.default_rng(1).standard_normal(n_cols))
    y = pa.DeviceVector(n_rows, 0)
    for _ in range(5): pa.spmv_(y, blk, x)
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    for _ in range(30): pa.spmv_(y, blk, x)
    e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
    ms = e0.elapsed_ms(e1) / 30
    print(f"{name:64s} {ms:8.4f} ms  {2*nnz/ms/1e6:7.1f} GFLOP/s  {(nnz*12 + n_rows*20)/ms/1e6:7.1f} GB/s alg", flush=True)

A, _ = pa.build_p_matrix(pa.DebugArray([1]), 128, 128, 128, 128, 128, 128, 1, 1, 1, keep_host=True)
H = pa.local_items(A.host_blocks)[0][0]
rate("27-pt 128^3: row split (row patterns)", A.matrix_partition.items[0].own_own, H.m, H.n, H.nnz)
S = pa.DeviceSELL(H)
print("   SELL padding:", S.info())
rate("27-pt 128^3: SELL-64, one lane per row", S, H.m, H.n, H.nnz)
rng = np.random.default_rng(0)
m = 2_000_000
base = np.repeat(np.arange(m), 16)
col = np.sort(np.clip(base + rng.integers(-2000, 2000, size=m * 16), 0, m - 1).reshape(m, 16), axis=1).ravel().astype(np.int32) + 1
Hb = pa.HostCSR(m, m, (1 + 16 * np.arange(m + 1)).astype(np.int32), col, rng.standard_normal(m * 16))
rate("2M rows x 16 within +-2000: row split (16-bit windows)", pa.DeviceCSR(Hb), m, m, Hb.nnz)
rate("2M rows x 16 within +-2000: SELL-64, one lane per row", pa.DeviceSELL(Hb), m, m, Hb.nnz)
lens = rng.integers(1, 40, m)
rp = np.concatenate([[1], 1 + np.cumsum(lens)]).astype(np.int32)
rows = np.repeat(np.arange(m), lens)
colr = np.clip(rows + rng.integers(-2000, 2000, size=len(rows)), 0, m - 1)
order = np.lexsort((colr, rows))
Hr = pa.HostCSR(m, m, rp, (colr[order] + 1).astype(np.int32), rng.standard_normal(len(rows)))
rate("2M ragged rows (1..39) within +-2000: row split", pa.DeviceCSR(Hr), m, m, Hr.nnz)
for sg in (1, 1024):
    S = pa.DeviceSELL(Hr, sigma=sg)
    rate(f"2M ragged rows: SELL-64 sigma={sg} (padded {S.info()['padded_entries'] / Hr.nnz:.2f}x)", S, m, m, Hr.nnz)
