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
x = pa.pzeros(A.col_partition)
pa.ref_cg_(x, A, b, maxiter=2, overlap=False, Pl=S)
pa.context().sync()
x = pa.pzeros(A.col_partition)
t = time.perf_counter()
x, r0, r, it = pa.ref_cg_(x, A, b, maxiter=10, overlap=False, Pl=S)
pa.context().sync()
print(n, ordering, 'ms per MG-PCG iteration', round((time.perf_counter() - t) / 10 * 1e3, 2), 'r/r0', r / r0, flush=True)
