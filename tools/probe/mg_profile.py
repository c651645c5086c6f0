"""Kernel breakdown of one MG-PCG configuration (run under rocprofv3 --kernel-trace --stats)."""
import sys, time
sys.path.insert(0, '.')
from __graft_entry__ import load_package
pa = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ordering = sys.argv[2] if len(sys.argv) > 2 else "multicolor_spmv"
ranks = pa.DebugArray([1])
S = pa.pc_setup(ranks, 1, 4, n, n, n, ordering)
A, b = S.A_vec[-1], S.r[-1]
use_graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
def run(k):
    x = pa.pzeros(A.col_partition)
    pa.context().sync()
    t = time.perf_counter()
    out = pa.opt_cg_(x, A, b, maxiter=k, Pl=S, graph=use_graph)
    pa.context().sync()
    return time.perf_counter() - t, out
run(2)
t1, _ = run(3)
t2, (x, r0, r, it) = run(13)
print(n, ordering, 'graph' if use_graph else 'eager', 'ms per MG-PCG iteration', round((t2 - t1) / 10 * 1e3, 2), 'r/r0', r / r0, flush=True)
