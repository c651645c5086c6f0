"""mul! over 2 and 8 parts of n^3 resident on ONE GPU, after a spin-up: per-part time of mul_c_ (one library call) vs own*own alone."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for P, shape in ((2, (2, 1, 1)), (8, (2, 2, 2))):
    ranks = pa.DebugArray(list(range(1, P + 1)))
    A, b = pa.build_p_matrix(ranks, n, n, n, *(n * s for s in shape), *shape)
    x = pa.pones(A.col_partition); y = pa.pzeros(A.row_partition)
    ctx = pa.context()
    def timed(f, reps=20):
        for _ in range(80 // P + 5): f()
        ctx.sync()
        t = time.perf_counter()
        for _ in range(reps): f()
        ctx.sync()
        return (time.perf_counter() - t) / reps * 1e3
    t_oo = timed(lambda: pa.pmap(lambda yv, blk, xv: pa.spmv_(yv, blk.own_own, xv), y.vector_partition, A.matrix_partition, x.vector_partition))
    t_mul = timed(lambda: pa.mul_c_(y, A, x))
    t_oo2 = timed(lambda: pa.pmap(lambda yv, blk, xv: pa.spmv_(yv, blk.own_own, xv), y.vector_partition, A.matrix_partition, x.vector_partition))
    print(f"P={P} n={n} per part: mul_c_ {t_mul/P:.4f} ms  own*own alone {t_oo/P:.4f} / {t_oo2/P:.4f}", flush=True)
    del A, b, x, y
