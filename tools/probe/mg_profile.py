"""20 MG-PCG iterations at 256^3 (multicolour SpMV smoother): the command rocprofv3 --kernel-trace --stats wraps."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
S = pa.pc_setup(pa.DebugArray([1]), 1, 4, n, n, n, "multicolor_spmv")
A, b = S.A_vec[-1], S.r[-1]
pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=3, Pl=S)
pa.context().sync()
x, r0, r, it = pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=20, Pl=S)
pa.context().sync()
print(it, r / r0)
