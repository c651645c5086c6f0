"""One-bit pattern-ELL stream at n^3: ms per product (for an A/B of PA_SPMV_PELL_BITS_U27 / PA_SPMV_PELL_RUNS3 across processes).
  PA_SPMV_PELL_BITS_U27=0 python tools/probe/pell_bits_ab.py 256"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = pa.context()
A, _ = pa.build_p_matrix(pa.DebugArray([1]), n, n, n, n, n, n, 1, 1, 1)
blk = A.matrix_partition.items[0].own_own
x = pa.DeviceVector(blk.n, 0).upload(np.random.default_rng(1).standard_normal(blk.n))
y = pa.DeviceVector(blk.m, 0)
nl = max(200, int(1.0e9 / max(blk.nnz, 1)))
for _ in range(6 * nl): pa.spmv_(y, blk, x)
ctx.sync()
ts = []
for r in range(5):
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    for _ in range(nl): pa.spmv_(y, blk, x)
    e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
    ts.append(e0.elapsed_ms(e1) / nl)
print(f"n={n} U27={os.environ.get('PA_SPMV_PELL_BITS_U27', '1')} RUNS3={os.environ.get('PA_SPMV_PELL_RUNS3', '1')} mode {blk.pell()['mode']}: min {min(ts):.4f} med {sorted(ts)[2]:.4f} ms")
