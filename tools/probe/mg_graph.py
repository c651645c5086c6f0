"""MG-PCG iteration at n^3 with and without the V-cycle replayed from a hipGraph."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for graph in (False, True, False, True):
    S = pa.pc_setup(pa.DebugArray([1]), 1, 4, n, n, n, "multicolor_spmv", graph=graph)
    A, b = S.A_vec[-1], S.r[-1]
    pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=25, Pl=S, fuse=True)
    out = []
    for rep in range(3):
        pa.context().sync(); t = time.perf_counter()
        pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=30, Pl=S, fuse=True)
        pa.context().sync(); out.append((time.perf_counter() - t) / 30 * 1e3)
    print(f"graph={graph}: {min(out):.3f} ms per MG-PCG iteration ({[round(v, 3) for v in out]})", flush=True)
    del S, A, b
