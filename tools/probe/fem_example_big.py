"""test/fem_example.jl at benchmark size: set-up, assembly and mul! timings on one GPU (8 parts as (4,2))."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ranks = pa.DebugArray(list(range(1, 9)))
t = time.perf_counter(); S = pa.fem_example.fem_example_system(ranks, (4, 2), (n, n)); t1 = time.perf_counter() - t
t = time.perf_counter(); A = pa.psparse_disassembled(S["I"], S["J"], S["V"], S["dof_partition"], S["dof_partition"]); pa.context().sync(); t2 = time.perf_counter() - t
t = time.perf_counter(); b = pa.pvector_disassembled(S["II"], S["VV"], S["dof_partition"]); pa.context().sync(); t3 = time.perf_counter() - t
print('cells', n, 'x', n, 'dofs', S["n_global_dofs"], ': generate', round(t1, 1), 's; psparse', round(t2, 1), 's; pvector', round(t3, 1), 's', flush=True)
print('encodings', A.matrix_partition.items[0].own_own.encoding(), 'ghost dofs per part', [c.n_ghost for c in A.col_partition.items])
x = pa.pones(A.col_partition); y = pa.pzeros(A.row_partition)
pa.mul_(y, A, x); pa.context().sync()
t = time.perf_counter()
for _ in range(20): pa.mul_(y, A, x)
pa.context().sync(); dt = (time.perf_counter() - t) / 20
nnz = sum(bk.own_own.nnz + bk.own_ghost.nnz for bk in A.matrix_partition.items)
print('mul_ 8 parts on one GPU:', round(dt * 1e3, 3), 'ms,', nnz, 'entries ->', round(2 * nnz / dt / 1e9, 1), 'GFLOP/s')
