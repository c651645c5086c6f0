"""Iterations the optimised MG-PCG needs to reach the reference tolerance (50 iterations of the reference ordering) for
several sweep orders of the greedy colours.  Usage: color_order.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ranks = pa.DebugArray([1])
S = pa.pc_setup(ranks, 1, 4, n, n, n, ordering="sequential")
A, b = S.A_vec[-1], S.r[-1]
x, r0, r, it = pa.ref_cg_(pa.pzeros(A.col_partition), A, b, maxiter=50, tolerance=0.0, overlap=False, Pl=S)
tol = r / r0
print(f"reference: 50 iterations -> {tol:.6e}", flush=True)
del S, A, b, x
import itertools
if len(sys.argv) > 2 and sys.argv[2] == "ties":
    orders = ["7," + ",".join(map(str, a)) + "," + ",".join(map(str, b)) + ",0" for a in itertools.permutations((3, 5, 6)) for b in itertools.permutations((1, 2, 4))]
else:
  orders = ["affinity", "reverse", "greedy", "0,1,2,3,4,5,6,7", "7,6,5,4,3,2,1,0", "1,2,3,4,5,6,7,0", "7,1,2,3,4,5,6,0", "1,2,4,3,5,6,7,0", "7,3,5,6,1,2,4,0", "4,2,1,6,5,3,7,0",
          "1,2,4,7,3,5,6,0", "3,5,6,0,1,2,4,7", "0,7,1,6,2,5,3,4"]
for o in orders:
    os.environ["PA_GS_COLOR_ORDER"] = o
    S = pa.pc_setup(ranks, 1, 4, n, n, n, ordering="multicolor_spmv")
    A, b = S.A_vec[-1], S.r[-1]
    x, r0, r, it = pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=500, tolerance=tol, Pl=S, fuse=True)
    pa.context().sync()
    t = time.perf_counter()
    pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=20, tolerance=0.0, Pl=S, fuse=True)
    pa.context().sync()
    ms = (time.perf_counter() - t) / 20 * 1e3
    print(f"greedy colours swept in the order {o}: {it} iterations, {ms:.3f} ms each", flush=True)
    del S, A, b, x
