"""Does K1's time depend on where hipMalloc puts the arrays?  One process, the same host matrix, T trials: a dummy
allocation of varying size, then block + x + y created anew and timed (HIP events, 30 launches)."""
import sys, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
T = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ctx = pa.context()
A, b = pa.build_p_matrix(pa.DebugArray([1]), n, n, n, n, n, n, 1, 1, 1, keep_host=True)
h = pa.local_items(A.host_blocks)[0][0]
rows = h.m
del A
rng = np.random.default_rng(0)
xh = rng.random(rows)


def ptr(v):
    p = C.c_void_p()
    L.call("pa_vec_data", v.h, C.byref(p))
    return p.value


keep = []
for t in range(T):
    dummy = pa.DeviceVector(int(rng.integers(1, 1 << 22)) * 16 + 1, 0) if t else None
    dA = pa.DeviceCSR(h)
    x = pa.DeviceVector(rows, 0).upload(xh)
    y = pa.DeviceVector(rows, 0)
    for _ in range(5):
        pa.spmv_(y, dA, x)
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    for _ in range(30):
        pa.spmv_(y, dA, x)
    e1 = ctx.event().record(L.STREAM_COMPUTE)
    ctx.sync()
    print(f"trial {t}: {e0.elapsed_ms(e1) / 30:.4f} ms   x@{ptr(x):#x} y@{ptr(y):#x}  d(y-x)={(ptr(y) - ptr(x)) / 2**20:.2f} MiB", flush=True)
    if t % 2:
        keep.append((dA, x, y, dummy))      # odd trials keep their memory: later trials land elsewhere
    del dA, x, y, dummy
