"""Two-stream read rate (the dot kernel) by footprint: does streaming slow down when the operands span tens of GB?"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
ctx = pa.context()
for gib in (1, 4, 12, 24):
    n = gib * (1 << 30) // 8
    a, b = pa.DeviceVector(n, 0), pa.DeviceVector(n, 0)
    a.fill(1.0); b.fill(2.0)
    for _ in range(5): L.call("pa_vec_dot", a.h, b.h, None)
    ctx.sync()
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    for _ in range(10): L.call("pa_vec_dot", a.h, b.h, None)
    e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
    ms = e0.elapsed_ms(e1) / 10
    print(f"2 x {gib} GiB: {ms:.3f} ms  {2 * 8 * n / ms / 1e6:.0f} GB/s  classes {a.memory_class()} {b.memory_class()}", flush=True)
    del a, b
