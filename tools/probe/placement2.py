"""K1 time vs where x and y sit (sub-ranges of one arena wrapped with pa_vec_wrap), the block fixed."""
import sys, ctypes as C
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
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


class Wrapped:
    def __init__(self, address, n):
        self.h = C.c_void_p()
        L.call("pa_vec_wrap", ctx.h, C.c_void_p(address), n, 0, C.byref(self.h))


def time_it(dA, x, y, reps=30):
    for _ in range(3):
        L.call("pa_spmv", dA.h, x.h, L.SEG_OWN, y.h, L.SEG_OWN, 1.0, 0.0)
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    for _ in range(reps):
        L.call("pa_spmv", dA.h, x.h, L.SEG_OWN, y.h, L.SEG_OWN, 1.0, 0.0)
    e1 = ctx.event().record(L.STREAM_COMPUTE)
    ctx.sync()
    return e0.elapsed_ms(e1) / reps


dA = pa.DeviceCSR(h)
x0 = pa.DeviceVector(rows, 0).upload(xh)
y0 = pa.DeviceVector(rows, 0)
print(f"separate allocations: {time_it(dA, x0, y0):.4f} ms  x@{ptr(x0):#x} y@{ptr(y0):#x}")
G = 1 << 27                                    # doubles per GiB
arena = pa.DeviceVector(4 * G, 0)
base = ptr(arena)
print(f"arena @{base:#x}")
xs = Wrapped(base, rows)
L.call("pa_vec_copy", xs.h, x0.h, L.SEG_OWN)
for off in (1 << 30, (1 << 30) + 4096, (1 << 30) + (1 << 16), (1 << 30) + (1 << 21), (1 << 30) + (1 << 21) * 33, (1 << 30) + (1 << 28),
            (1 << 27), (1 << 27) + (1 << 12), 3 << 30):
    y = Wrapped(base + off, rows)
    print(f"x at arena+0, y at arena+{off / 2**20:9.3f} MiB: {time_it(dA, xs, y):.4f} ms", flush=True)
for off in (0, 4096, 1 << 16, 1 << 21, 1 << 26, (1 << 30) + (1 << 20)):
    x = Wrapped(base + (2 << 30) + off, rows)
    L.call("pa_vec_copy", x.h, x0.h, L.SEG_OWN)
    print(f"y separate, x at arena+2GiB+{off / 2**20:9.3f} MiB: {time_it(dA, x, y0):.4f} ms", flush=True)
# the block created again (new addresses for the matrix arrays), vectors unchanged
for t in range(4):
    dummy = pa.DeviceVector(int(rng.integers(1, 1 << 24)) * 16 + 1, 0)
    dB = pa.DeviceCSR(h)
    print(f"block re-created ({t}): {time_it(dB, x0, y0):.4f} ms   first block again: {time_it(dA, x0, y0):.4f} ms", flush=True)
    del dB
