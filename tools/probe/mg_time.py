import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
for n, ordering in ((128, "sequential"), (128, "multicolor"), (128, "multicolor_spmv"), (256, "multicolor"), (256, "multicolor_spmv")):
    ranks = pa.DebugArray([1])
    S = pa.pc_setup(ranks, 1, 4, n, n, n, ordering)
    A, b = S.A_vec[-1], S.r[-1]
    x = pa.pzeros(A.col_partition)
    pa.ref_cg_(x, A, b, maxiter=2, overlap=False, Pl=S)
    pa.context().sync()
    x = pa.pzeros(A.col_partition)
    t = time.perf_counter()
    x, r0, r, it = pa.ref_cg_(x, A, b, maxiter=10, overlap=False, Pl=S)
    pa.context().sync()
    dt = (time.perf_counter() - t) / 10
    print(n, ordering, 'levels', [g.info().items[0] for g in S.gs_states], 'ms per MG-PCG iteration', round(dt * 1e3, 2), 'r/r0', r / r0, flush=True)
