"""mul! over P parts resident on ONE GPU (DebugArray): per-part cost of pack/unpack/own*ghost next to own*own."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for P, shape in ((1, (1, 1, 1)), (2, (2, 1, 1)), (4, (2, 2, 1)), (8, (2, 2, 2))):
    ranks = pa.DebugArray(list(range(1, P + 1)))
    A, b = pa.build_p_matrix(ranks, n, n, n, *(n * s for s in shape), *shape)
    x = pa.pones(A.col_partition); y = pa.pzeros(A.row_partition)
    ctx = pa.context()
    def timed(f, reps=20):
        f(); ctx.sync()
        t = time.perf_counter()
        for _ in range(reps): f()
        ctx.sync()
        return (time.perf_counter() - t) / reps * 1e3
    t_mul = timed(lambda: pa.mul_(y, A, x))
    t_oo = timed(lambda: pa.pmap(lambda yv, blk, xv: pa.spmv_(yv, blk.own_own, xv), y.vector_partition, A.matrix_partition, x.vector_partition))
    t_oh = timed(lambda: pa.pmap(lambda yv, blk, xv: pa.spmv_(yv, blk.own_ghost, xv, L.SEG_GHOST, L.SEG_OWN, 1.0, 1.0), y.vector_partition, A.matrix_partition, x.vector_partition))
    t_cons = timed(lambda: pa.consistent_(x).wait())
    enc = A.matrix_partition.items[0].own_ghost.encoding()
    print(f"P={P} per part: mul_ {t_mul/P:.4f} ms  own*own {t_oo/P:.4f}  own*ghost {t_oh/P:.4f}  consistent! {t_cons/P:.4f}   own_ghost encoding {enc}", flush=True)
    del A, b, x, y
