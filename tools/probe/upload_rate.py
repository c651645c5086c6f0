"""How fast do set-up uploads go?  512 MB of float64 from a numpy array into a device vector: first and later copies."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
ctx = pa.context()
n = 64 * 1024 * 1024
h = np.random.default_rng(0).standard_normal(n)
for name, mk in (("first vector (builds the arena)", lambda: pa.DeviceVector(n, 0)), ("second vector", lambda: pa.DeviceVector(n, 0))):
    t = time.perf_counter(); v = mk(); ctx.sync(); ta = time.perf_counter() - t
    for rep in range(3):
        t = time.perf_counter(); v.upload(h); ctx.sync(); dt = time.perf_counter() - t
        print(f"{name}: alloc {ta:.3f} s; upload {rep}: {dt:.3f} s = {8 * n / dt / 1e9:.1f} GB/s", flush=True)
    t = time.perf_counter(); out = v.download(); dt = time.perf_counter() - t
    print(f"{name}: download {dt:.3f} s = {8 * n / dt / 1e9:.1f} GB/s", flush=True)
