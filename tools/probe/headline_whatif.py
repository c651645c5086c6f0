"""The headline product (27-pt 256^3) through the library named by PA_HIP_LIBRARY (probe builds)."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
ctx = pa.context()
tag = sys.argv[1] if len(sys.argv) > 1 else "product"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 256
A, _ = pa.build_p_matrix(pa.DebugArray([1]), n, n, n, n, n, n, 1, 1, 1)
blk = A.matrix_partition.items[0].own_own
x = pa.DeviceVector(blk.n, 0).upload(np.random.default_rng(1).standard_normal(blk.n))
y = pa.DeviceVector(blk.m, 0)
for _ in range(200): pa.spmv_(y, blk, x)
best = []
for rep in range(3):
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    for _ in range(50): pa.spmv_(y, blk, x)
    e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
    best.append(e0.elapsed_ms(e1) / 50)
print(f"[{tag:14s}] 27-pt {n}^3 spmv  {min(best):8.4f} ms (of {[round(b,4) for b in best]})  classes val/x/y {blk.memory_class()} {x.memory_class()} {y.memory_class()}", flush=True)
