"""What would column-sorted chunks buy on banded-unstructured rows?  Timing only: the entries of every 1536-entry chunk
(96 rows x 16) are re-ordered (sorted by column, then interleaved so that one gather instruction of a wavefront covers 64
consecutive sorted entries) and fed to the existing kernel -- the row sums are wrong, the memory behaviour is the candidate's
minus its scattered LDS writes."""
import sys, os
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
    print(f"{name:60s} {ms:8.4f} ms  {(blk.nnz*12 + n_rows*20)/ms/1e6:7.1f} GB/s alg  enc {blk.encoding()}", flush=True)

rng = np.random.default_rng(0)
m = 4_000_000 // 96 * 96
for width in (2000, 500, 20000):
    rp = (1 + 16 * np.arange(m + 1)).astype(np.int32)
    base = np.repeat(np.arange(m), 16)
    col = np.clip(base + rng.integers(-width, width, size=m * 16), 0, m - 1)
    col_rows = np.sort(col.reshape(m, 16), axis=1).ravel().astype(np.int32)
    val = rng.standard_normal(m * 16)
    rate(f"+-{width}: rows as they are", pa.DeviceCSR(pa.HostCSR(m, m, rp, col_rows + 1, val)), m, m)
    ch = np.sort(col.reshape(m // 96, 1536), axis=1)                  # sorted inside each chunk
    rate(f"+-{width}: sorted inside chunks (pairs adjacent)", pa.DeviceCSR(pa.HostCSR(m, m, rp, (ch.ravel() + 1).astype(np.int32), val)), m, m)
    # interleave: position (k*256 + t)*2 + h  <-  sorted index (k*4 + t//64)*128 + h*64 + t%64
    k, t, h = np.meshgrid(np.arange(3), np.arange(256), np.arange(2), indexing="ij")
    pos = ((k * 256 + t) * 2 + h).ravel()
    src = ((k * 4 + t // 64) * 128 + h * 64 + t % 64).ravel()
    inter = np.empty_like(ch)
    inter[:, pos] = ch[:, src]
    rate(f"+-{width}: sorted + interleaved (64 consecutive per gather)", pa.DeviceCSR(pa.HostCSR(m, m, rp, (inter.ravel() + 1).astype(np.int32), val)), m, m)
