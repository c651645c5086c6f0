"""BASELINE config 3 (27-pt 128^3 per part, 2 parts) on one GPU: per-step cost of mul! composed from Python vs one call."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
for n, P, shape in ((128, 2, (2, 1, 1)), (128, 8, (2, 2, 2)), (64, 8, (2, 2, 2))):
    A, b = pa.build_p_matrix(pa.DebugArray(list(range(1, P + 1))), n, n, n, *(n * s for s in shape), *shape)
    x = pa.pones(A.col_partition); y = pa.pzeros(A.row_partition)
    ctx = pa.context()
    def timed(f, reps=200):
        for _ in range(5): f()
        ctx.sync(); t = time.perf_counter()
        for _ in range(reps): f()
        ctx.sync(); return (time.perf_counter() - t) / reps * 1e3
    t1, t2 = timed(lambda: pa.mul_(y, A, x)), timed(lambda: pa.mul_c_(y, A, x))
    nnz = sum(bk.own_own.nnz + bk.own_ghost.nnz for bk in A.matrix_partition.items)
    print(f"{n}^3 x {P} parts on one GPU: mul_ {t1:.4f} ms  mul_c_ (one call) {t2:.4f} ms  -> {2*nnz/t2/1e6:.0f} GFLOP/s", flush=True)
