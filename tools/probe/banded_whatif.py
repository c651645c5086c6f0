"""Banded-unstructured rows through the library named by PA_HIP_LIBRARY (probe builds: LDS pad, lane-contiguous x reads)."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
ctx = pa.context()
tag = sys.argv[1] if len(sys.argv) > 1 else "product"

def rate(name, H):
    blk = pa.DeviceCSR(H)
    x = pa.DeviceVector(H.n, 0).upload(np.random.default_rng(1).standard_normal(H.n))
    y = pa.DeviceVector(H.m, 0)
    for _ in range(60): pa.spmv_(y, blk, x)
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    for _ in range(50): pa.spmv_(y, blk, x)
    e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
    ms = e0.elapsed_ms(e1) / 50
    print(f"[{tag:14s}] {name:50s} {ms:8.4f} ms  {(H.nnz*12 + H.m*20)/ms/1e6:7.1f} GB/s alg  {blk.info()}", flush=True)

rng = np.random.default_rng(0)
m = 4_000_000
for band in (2000, 500):
    base = np.repeat(np.arange(m), 16)
    col = np.sort(np.clip(base + rng.integers(-band, band, size=m * 16), 0, m - 1).reshape(m, 16), axis=1).ravel().astype(np.int32) + 1
    H = pa.HostCSR(m, m, (1 + 16 * np.arange(m + 1)).astype(np.int32), col, rng.standard_normal(m * 16))
    rate(f"4M rows x 16 within +-{band}", H)
    del H, col, base
m = 2_000_000
lens = rng.integers(1, 40, m)
rp = np.concatenate([[1], 1 + np.cumsum(lens)]).astype(np.int32)
rows = np.repeat(np.arange(m), lens)
colr = np.clip(rows + rng.integers(-2000, 2000, size=len(rows)), 0, m - 1)
order = np.lexsort((colr, rows))
rate("2M ragged rows (1..39) within +-2000", pa.HostCSR(m, m, rp, (colr[order] + 1).astype(np.int32), rng.standard_normal(len(rows))))
