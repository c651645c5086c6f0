"""BASELINE config 2: 7-point Laplacian 256^3, one part, SpMV only."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
I, J, V, rows, cols = pa.laplacian_fdm((n, n, n), (1, 1, 1), pa.DebugArray([1]))
A = pa.psparse_from_coo(I, J, V, rows)
blk = A.matrix_partition.items[0]
x = pa.pvector_from_function(lambda i: (i.get_local_to_global() % 7) - 3.0, A.col_partition); y = pa.pzeros(A.row_partition)
xv, yv = x.vector_partition.items[0], y.vector_partition.items[0]
ctx = pa.context()
for _ in range(5): pa.spmv_(yv, blk.own_own, xv)
e0 = ctx.event().record(L.STREAM_COMPUTE)
for _ in range(50): pa.spmv_(yv, blk.own_own, xv)
e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
ms = e0.elapsed_ms(e1) / 50
nnz, nr = blk.own_own.nnz, n ** 3
print(f"7-pt {n}^3: {ms:.4f} ms per SpMV, {2*nnz/ms/1e6:.1f} GFLOP/s, {(nnz*12 + nr*20 + 4)/ms/1e6:.1f} GB/s algorithmic, encoding {blk.own_own.encoding()}")
