"""A part with more than 2^31 stored entries (27-pt, 432^3 rows): set-up time, slabs, SpMV rate, A*1 == b."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 432
t = time.perf_counter()
A, b = pa.build_p_matrix(pa.DebugArray([1]), n, n, n, n, n, n, 1, 1, 1)
pa.context().sync()
print('set-up', round(time.perf_counter() - t, 1), 's', flush=True)
blk = A.matrix_partition.items[0].own_own
print('info', blk.info(), 'encoding', blk.encoding(), '2^31 =', 2 ** 31, 'HBM bytes', blk.device_bytes(), '= %.2f per stored entry' % (blk.device_bytes() / blk.nnz))
x = pa.pones(A.col_partition); y = pa.pzeros(A.row_partition)
pa.mul_(y, A, x)
print('A*1 == b:', all(np.array_equal(g, e) for g, e in zip(y.own_values().items, b.own_values().items)))
xv, yv = x.vector_partition.items[0], y.vector_partition.items[0]
ctx = pa.context()
e0 = ctx.event().record(L.STREAM_COMPUTE)
for _ in range(10): pa.spmv_(yv, blk, xv)
e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
ms = e0.elapsed_ms(e1) / 10
nnz = blk.nnz
print(f"{n}^3: {ms:.3f} ms per SpMV, {2*nnz/ms/1e6:.1f} GFLOP/s, {(nnz*12 + n**3*20)/ms/1e6:.1f} GB/s algorithmic")
