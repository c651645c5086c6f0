"""Does a plain two-stream read (k_dot_partial) slow down with the size of its operands?  python tools/probe/stream_size.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
ctx = pa.context()
for gib in (0.25, 0.5, 1, 2, 3, 4, 6, 8):
    m = int(gib * (1 << 30) / 8)
    va, vb = pa.DeviceVector(m, 0), pa.DeviceVector(m, 0)
    va.fill(1.0); vb.fill(2.0)
    for _ in range(20): L.call("pa_vec_dot_slot", va.h, vb.h, 5, 0)
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    for _ in range(20): L.call("pa_vec_dot_slot", va.h, vb.h, 5, 0)
    e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
    ms = e0.elapsed_ms(e1) / 20
    print(f"2 x {gib:5.2f} GiB: {ms:8.4f} ms  {2 * 8 * m / ms / 1e6:7.1f} GB/s  classes {va.memory_class()} {vb.memory_class()}", flush=True)
    del va, vb
