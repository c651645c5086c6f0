"""Banded rows without a pattern: the sliding x window (k_spmv_xring) against the window tiers and the row split, same block."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
ctx = pa.context()

MODES = (("row split", {"PA_SPMV_XWIN": "0", "PA_SPMV_COLSPLIT": "0"}), ("windows", {"PA_SPMV_XRING": "0"}), ("40K + ring", {"PA_SPMV_XRING": "1"}), ("ring only", {"PA_SPMV_XRING": "2"}))


def rate(name, H):
    x = pa.DeviceVector(H.n, 0).upload(np.random.default_rng(1).standard_normal(H.n))
    alg = (H.nnz * 12 + H.m * 20) / 1e6
    ref, line = None, f"{name:38s}"
    for label, env in MODES:
        for k in ("PA_SPMV_XWIN", "PA_SPMV_XRING", "PA_SPMV_COLSPLIT"):
            os.environ.pop(k, None)
        os.environ.update(env)
        blk = pa.DeviceCSR(H)
        y = pa.DeviceVector(H.m, 0)
        t_end = time.perf_counter() + 0.25
        while time.perf_counter() < t_end:
            for _ in range(20): pa.spmv_(y, blk, x)
            ctx.sync()
        e0 = ctx.event().record(L.STREAM_COMPUTE)
        for _ in range(50): pa.spmv_(y, blk, x)
        e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
        ms = e0.elapsed_ms(e1) / 50
        got = y.download()
        if ref is None: ref = got
        xw = blk.xwin()
        line += f" | {label} {ms:7.4f} ms {alg / ms:5.0f} GB/s [{xw['groups'] - xw['big_groups'] - xw['ring_groups']}/{xw['big_groups']}/{xw['ring_groups']}]{'' if np.array_equal(got, ref) else ' BITS DIFFER'}"
        del blk, y
    print(line, flush=True)


rng = np.random.default_rng(0)
m = 4_000_000
bands = tuple(int(b) for b in sys.argv[1].split(",")) if len(sys.argv) > 1 else (7900, 7000, 5000, 3000, 2000, 1000)
for band in bands:
    base = np.repeat(np.arange(m), 16)
    col = np.sort(np.clip(base + rng.integers(-band, band, size=m * 16), 0, m - 1).reshape(m, 16), axis=1).ravel().astype(np.int32) + 1
    rate(f"4M rows x 16 within +-{band}", pa.HostCSR(m, m, (1 + 16 * np.arange(m + 1)).astype(np.int32), col, rng.standard_normal(m * 16)))
    del col, base
m = 2_000_000
lens = rng.integers(1, 40, m)
rp = np.concatenate([[1], 1 + np.cumsum(lens)]).astype(np.int32)
rows = np.repeat(np.arange(m), lens)
for band in (2000, 6000):
    colr = np.clip(rows + rng.integers(-band, band, size=len(rows)), 0, m - 1)
    order = np.lexsort((colr, rows))
    rate(f"2M ragged rows (1..39) within +-{band}", pa.HostCSR(m, m, rp, (colr[order] + 1).astype(np.int32), rng.standard_normal(len(rows))))
