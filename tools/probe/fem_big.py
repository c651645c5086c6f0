"""BASELINE config 5 at size: gallery laplacian_fem on n x n nodes, 8 parts (4,2): set-up and mul! timings on one GPU."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ranks = pa.DebugArray(list(range(1, 9)))
t = time.perf_counter(); I, J, V, rows, cols = pa.laplacian_fem((n, n), (4, 2), ranks); t1 = time.perf_counter() - t
t = time.perf_counter(); A = pa.psparse_disassembled(I, J, V, rows, cols); pa.context().sync(); t2 = time.perf_counter() - t
print('generate', round(t1, 1), 's; psparse (disassembled -> assemble -> split -> upload)', round(t2, 1), 's', flush=True)
print('encodings', [b.own_own.encoding() for b in A.matrix_partition.items][:2], 'nnz', [ (b.own_own.nnz, b.own_ghost.nnz) for b in A.matrix_partition.items][:2])
x = pa.pones(A.col_partition); y = pa.pzeros(A.row_partition)
pa.mul_(y, A, x); pa.context().sync()
t = time.perf_counter()
for _ in range(20): pa.mul_(y, A, x)
pa.context().sync(); dt = (time.perf_counter() - t) / 20
nnz = sum(b.own_own.nnz + b.own_ghost.nnz for b in A.matrix_partition.items)
print('mul_ 8 parts on one GPU:', round(dt * 1e3, 3), 'ms ->', round(2 * nnz / dt / 1e9, 1), 'GFLOP/s,', round((nnz * 12 + 20 * n * n) / dt / 1e9, 1), 'GB/s algorithmic')
print('max |A*1| over own rows (row sums of a stiffness matrix are ~0):', max(float(np.abs(v).max()) for v in y.own_values().items))
