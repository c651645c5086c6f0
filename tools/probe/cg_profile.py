"""One part, 27-pt n^3: `iters` iterations of opt_cg_ (fused and unfused) for a kernel trace.
   rocprofv3 --kernel-trace --stats -d gpurun_out/prof_cg -- python tools/probe/cg_profile.py 256 30"""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 30
A, b = pa.build_p_matrix(pa.DebugArray([1]), n, n, n, n, n, n, 1, 1, 1)
ctx = pa.context()
work = pa.cg_work(pa.pzeros(A.col_partition), b, A)
for fuse in (True, False):
    for k in (3, iters):
        x = pa.pzeros(A.col_partition)
        ctx.sync(); t = time.perf_counter()
        pa.opt_cg_(x, A, b, maxiter=k, work=work, fuse=fuse)
        ctx.sync(); dt = time.perf_counter() - t
    print(f"fuse={fuse}: {dt / iters * 1e3:.4f} ms per iteration (incl. set-up of the solve)", flush=True)
