"""Bands beyond the sliding x window (column-split chains): the chain as ONE launch (k_spmv_xring_chain) against a launch per piece,
same pieces, same process.  usage: chain_rate.py 16000,12000   (PA_SPMV_XRING_GROUPS=n: workgroups of the launch)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
ctx = pa.context()


def ms_of(blk, x, y, reps=50):
    t_end = time.perf_counter() + 0.25
    while time.perf_counter() < t_end:
        for _ in range(20): pa.spmv_(y, blk, x)
        ctx.sync()
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    for _ in range(reps): pa.spmv_(y, blk, x)
    e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
    return e0.elapsed_ms(e1) / reps


rng = np.random.default_rng(0)
m = 4_000_000
bands = tuple(int(b) for b in sys.argv[1].split(",")) if len(sys.argv) > 1 else (16000, 12000)
for band in bands:
    col = np.repeat(np.arange(m, dtype=np.int32), 16).reshape(m, 16)
    col += rng.integers(-band, band, size=(m, 16), dtype=np.int32)
    np.clip(col, 0, m - 1, out=col)
    col.sort(axis=1)
    col += 1
    H = pa.HostCSR(m, m, (1 + 16 * np.arange(m + 1)).astype(np.int32), col.ravel(), rng.standard_normal(m * 16))
    del col
    blk = pa.DeviceCSR(H)
    alg = (H.nnz * 12 + (m + 1) * 4 + m * 16) / 1e6
    x = pa.DeviceVector(m, 0).upload(rng.standard_normal(m))
    y = pa.DeviceVector(m, 0)
    info = blk.chain()
    out = {}
    for fused in ("1", "0", "1", "0"):
        os.environ["PA_SPMV_CHAIN_FUSED"] = fused
        ctx.reload_env()
        out.setdefault(fused, []).append(ms_of(blk, x, y))
        got = y.download()
        out.setdefault("bits", got)
        assert np.array_equal(got, out["bits"]), "BITS DIFFER"
    os.environ.pop("PA_SPMV_CHAIN_FUSED")
    ctx.reload_env()
    f, s = min(out["1"]), min(out["0"])
    print(f"+-{band}: {info}  one launch {f:.4f} ms = {alg / f / 1e3:.2f} TB/s algorithmic | a launch per piece {s:.4f} ms = {alg / s / 1e3:.2f}"
          f" | moved {blk.stream_bytes() / 1e6 + m * 16 / 1e6:.0f} MB", flush=True)
    del blk, x, y, H
