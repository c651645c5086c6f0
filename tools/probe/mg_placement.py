"""Does the multicolour V-cycle's time depend on where its blocks and vectors were allocated?  The same set-up four
times in one process (earlier ones kept alive, so every repetition lands elsewhere), MG-PCG iteration time of each."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
keep = []
for rep in range(4):
    S = pa.pc_setup(pa.DebugArray([1]), 1, 4, n, n, n, "multicolor_spmv")
    A, b = S.A_vec[-1], S.r[-1]
    pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=3, Pl=S)
    pa.context().sync()
    def run(k):
        x = pa.pzeros(A.col_partition)
        pa.context().sync()
        t = time.perf_counter()
        pa.opt_cg_(x, A, b, maxiter=k, Pl=S)
        pa.context().sync()
        return time.perf_counter() - t
    ts = [(run(15) - run(3)) / 12 * 1e3 for _ in range(3)]
    print(f"set-up {rep}: MG-PCG iteration {min(ts):.3f} ms (runs {', '.join('%.3f' % t for t in ts)})", flush=True)
    keep.append(S)
