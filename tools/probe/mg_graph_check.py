"""MG-PCG with the V-cycle replayed from a hipGraph against the eager V-cycle: same bits, time per iteration."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
for n in (32, 128, 256):
    out = []
    for graph in (False, True):
        S = pa.pc_setup(pa.DebugArray([1]), 1, 4, n, n, n, "multicolor_spmv", graph=graph)
        A, b = S.A_vec[-1], S.r[-1]
        h = []
        x, r0, r, it = pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=8, Pl=S, history=h)
        pa.context().sync(); t = time.perf_counter()
        x2, r0, r2, it = pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=20, Pl=S)
        pa.context().sync(); dt = (time.perf_counter() - t) / 20
        out.append((h, x.own_values().items[0].copy(), dt, r2))
        del S
    same = out[0][0] == out[1][0] and np.array_equal(out[0][1], out[1][1]) and out[0][3] == out[1][3]
    print(f"{n}^3: eager {out[0][2] * 1e3:.3f} ms, graph {out[1][2] * 1e3:.3f} ms per MG-PCG iteration; same bits {same}", flush=True)
