"""Fixed cost of one opt_cg_ solve with the multigrid preconditioner (what a CG set pays besides its iterations)."""
import sys, time, cProfile, pstats
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = pa.pc_setup(pa.DebugArray([1]), 1, 4, n, n, n, "multicolor_spmv")
A, b = S.A_vec[-1], S.r[-1]


def run(k, work=None):
    x = pa.pzeros(A.col_partition)
    pa.context().sync()
    t = time.perf_counter()
    pa.opt_cg_(x, A, b, maxiter=k, Pl=S, work=work)
    pa.context().sync()
    return (time.perf_counter() - t) * 1e3


run(2)
print("maxiter 0, 1, 2, 4, 8 (ms):", [round(run(k), 2) for k in (0, 1, 2, 4, 8)])
w = pa.cg_work(pa.pzeros(A.col_partition), b)
print("the same with reused work vectors:", [round(run(k, w), 2) for k in (0, 1, 2, 4, 8)])
pr = cProfile.Profile(); pr.enable()
run(1)
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(14)
