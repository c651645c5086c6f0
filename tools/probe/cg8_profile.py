"""BASELINE config 4 on ONE GPU (8 parts x 256^3 resident): CG iterations for a per-kernel rocprofv3 profile."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A, b = pa.build_p_matrix(pa.DebugArray(list(range(1, 9))), n, n, n, 2 * n, 2 * n, 2 * n, 2, 2, 2)
for fn in (pa.ref_cg_, pa.opt_cg_):
    x = pa.pzeros(A.col_partition)
    pa.context().sync()
    t = time.perf_counter()
    x, r0, r, it = fn(x, A, b, maxiter=10)
    pa.context().sync()
    print(fn.__name__, 'iters', it, 'r/r0', r / r0, 'wall ms/iter (8 parts on one GPU, incl. set-up of the loop)', (time.perf_counter() - t) / it * 1e3, flush=True)
