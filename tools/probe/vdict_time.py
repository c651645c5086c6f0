"""SpMV / CG / MG-PCG with the optional value dictionary (PA_SPMV_VALUE_DICT=1) next to the fp64 stream."""
import os, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for vd in ("0", "1"):
    os.environ["PA_SPMV_VALUE_DICT"] = vd
    A, b = pa.build_p_matrix(pa.DebugArray([1]), n, n, n, n, n, n, 1, 1, 1)
    blk = A.matrix_partition.items[0].own_own
    x = pa.pones(A.col_partition); y = pa.pzeros(A.row_partition)
    xv, yv = x.vector_partition.items[0], y.vector_partition.items[0]
    ctx = pa.context()
    for _ in range(5): pa.spmv_(yv, blk, xv)
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    for _ in range(50): pa.spmv_(yv, blk, xv)
    e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
    ms = e0.elapsed_ms(e1) / 50
    def cg(k):
        z = pa.pzeros(A.col_partition); ctx.sync(); t = time.perf_counter(); pa.opt_cg_(z, A, b, maxiter=k); ctx.sync(); return time.perf_counter() - t
    cg(3); tcg = (cg(34) - cg(4)) / 30 * 1e3
    print(f"value_dict={vd} (dict size {blk.value_dict()}): SpMV {ms:.4f} ms = {2*blk.nnz/ms/1e6:.0f} GFLOP/s, {(blk.nnz*12 + n**3*20)/ms/1e6:.0f} GB/s algorithmic; CG iteration {tcg:.3f} ms", flush=True)
    del A, b, x, y
for vd in ("0", "1"):
    os.environ["PA_SPMV_VALUE_DICT"] = vd
    S = pa.pc_setup(pa.DebugArray([1]), 1, 4, n, n, n, ordering="multicolor_spmv")
    A, b = S.A_vec[-1], S.r[-1]
    def run(k):
        z = pa.pzeros(A.col_partition); pa.context().sync(); t = time.perf_counter(); pa.opt_cg_(z, A, b, maxiter=k, Pl=S); pa.context().sync(); return time.perf_counter() - t
    run(2); t = (run(13) - run(3)) / 10 * 1e3
    print(f"value_dict={vd}: MG-PCG iteration {t:.2f} ms", flush=True)
    del S, A, b
