"""CG iteration time, eager launches vs hipGraph replay, for small parts (launch-bound) up to 256^3."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
for n in (16, 32, 64, 128, 256):
    A, b = pa.build_p_matrix(pa.DebugArray([1]), n, n, n, n, n, n, 1, 1, 1)
    res = {}
    for graph in (False, True):
        def run(k):
            x = pa.pzeros(A.col_partition); pa.context().sync()
            t = time.perf_counter(); pa.opt_cg_(x, A, b, maxiter=k, graph=graph); pa.context().sync()
            return time.perf_counter() - t
        run(9); t1 = run(9); t2 = run(309)
        res[graph] = (t2 - t1) / 300 * 1e6
    print(f"{n}^3: eager {res[False]:8.1f} us/iteration   graph {res[True]:8.1f} us/iteration   x{res[False]/res[True]:.2f}", flush=True)
