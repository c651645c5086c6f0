"""A large banded-unstructured block (20 M rows x 16 = 320 M entries): x-window launches against the row split, same bits."""
import os, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
ctx = pa.context()
rng = np.random.default_rng(0)
m = 20_000_000
t = time.time()
col = np.repeat(np.arange(m, dtype=np.int32), 16).reshape(m, 16)
col += rng.integers(-2000, 2000, size=(m, 16), dtype=np.int32)
np.clip(col, 0, m - 1, out=col); col.sort(axis=1); col += 1
H = pa.HostCSR(m, m, (1 + 16 * np.arange(m + 1, dtype=np.int64)).astype(np.int32), col.ravel(), rng.standard_normal(m * 16))
print(f"host matrix {time.time()-t:.1f} s", flush=True)
x = pa.DeviceVector(m, 0).upload(rng.standard_normal(m))
out = []
for sw in ("0", None):
    if sw is None: os.environ.pop("PA_SPMV_XWIN", None)
    else: os.environ["PA_SPMV_XWIN"] = sw
    t = time.time()
    blk = pa.DeviceCSR(H)
    ts = time.time() - t
    y = pa.DeviceVector(m, 0)
    for _ in range(20): pa.spmv_(y, blk, x)
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    for _ in range(20): pa.spmv_(y, blk, x)
    e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
    ms = e0.elapsed_ms(e1) / 20
    out.append(y.download())
    print(f"XWIN={sw}: {ms:.4f} ms  {(H.nnz*12+m*20)/ms/1e6:.0f} GB/s alg  set-up {ts:.1f} s  {blk.xwin()} classes {blk.memory_class()} {x.memory_class()} {y.memory_class()}", flush=True)
    del blk, y
print("same bits", np.array_equal(out[0], out[1]))
