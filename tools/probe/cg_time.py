"""Time per CG iteration (BASELINE config 4's loop on one part): ref_cg_ with Identity preconditioner."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
ranks = pa.DebugArray([1])
A, b = pa.build_p_matrix(ranks, n, n, n, n, n, n, 1, 1, 1)
for name, fn in (("ref_cg_", pa.ref_cg_), ("opt_cg_", getattr(pa, "opt_cg_", None))):
    if fn is None:
        continue
    def run(k):
        x = pa.pzeros(A.col_partition)
        pa.context().sync()
        t = time.perf_counter()
        out = fn(x, A, b, maxiter=k)
        pa.context().sync()
        return time.perf_counter() - t, out
    run(3)
    t1, _ = run(10)
    t2, (x, r0, r, it) = run(10 + iters)        # the difference cancels the set-up (allocations, first residual)
    print(name, n, 'iters', it, 'ms per CG iteration', round((t2 - t1) / iters * 1e3, 3), 'r/r0', r / r0, flush=True)
