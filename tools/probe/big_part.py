"""One large part (n^3 rows of the 27-point operator) on one GPU with the arena: where things land and how fast mul! runs."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
ctx = pa.context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
t = time.perf_counter()
A, b = pa.build_p_matrix(pa.DebugArray([1]), n, n, n, n, n, n, 1, 1, 1)
ctx.sync()
print(f"{n}^3: set-up {time.perf_counter() - t:.1f} s, arena {ctx.arena()}", flush=True)
blk = A.matrix_partition.items[0].own_own
x = pa.pones(A.col_partition)
y = pa.pzeros(A.row_partition)
pa.mul_(y, A, x)
ok = all(np.array_equal(g, e) for g, e in zip(y.own_values().items, b.own_values().items))
for _ in range(30): pa.mul_c_(y, A, x)
e0 = ctx.event().record(L.STREAM_COMPUTE)
for _ in range(20): pa.mul_c_(y, A, x)
e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
ms = e0.elapsed_ms(e1) / 20
print(f"{n}^3: nnz {blk.nnz}, A*1 == b {ok}, mul! {ms:.3f} ms = {2 * blk.nnz / ms / 1e6:.0f} GFLOP/s, moved {(blk.stream_bytes() + 16 * blk.m) / ms / 1e6:.0f} GB/s, "
      f"classes val/x/y {blk.memory_class()} {x.vector_partition.items[0].memory_class()} {y.vector_partition.items[0].memory_class()}", flush=True)
