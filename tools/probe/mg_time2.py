"""MG-PCG iteration time (opt_cg_ with the multicolour SpMV smoother), one part, 128^3 and 256^3."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
for n in (128, 256):
    t = time.perf_counter()
    S = pa.pc_setup(pa.DebugArray([1]), 1, 4, n, n, n, "multicolor_spmv")
    pa.context().sync(); ts = time.perf_counter() - t
    A, b = S.A_vec[-1], S.r[-1]
    for fuse in (True, False):
        pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=3, Pl=S, fuse=fuse)
        pa.context().sync()
        t = time.perf_counter()
        x, r0, r, it = pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=20, Pl=S, fuse=fuse)
        pa.context().sync()
        dt = (time.perf_counter() - t) / 20
        print(f"{n}^3 fuse={fuse}: {dt * 1e3:.3f} ms per MG-PCG iteration, r/r0 {r / r0:.3e}, set-up {ts:.1f} s, arena {pa.context().arena()['class_gib']}", flush=True)
