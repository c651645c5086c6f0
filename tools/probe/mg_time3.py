"""MG-PCG iteration time at 256^3 under a few set-up switches (one child process each), with a long warm-up."""
import os, subprocess, sys, time
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    from __graft_entry__ import load_package
    pa = load_package()
    n = 256
    S = pa.pc_setup(pa.DebugArray([1]), 1, 4, n, n, n, "multicolor_spmv")
    A, b = S.A_vec[-1], S.r[-1]
    pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=25, Pl=S, fuse=True)
    out = []
    for rep in range(3):
        pa.context().sync()
        t = time.perf_counter()
        pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=30, Pl=S, fuse=True)
        pa.context().sync()
        out.append((time.perf_counter() - t) / 30 * 1e3)
    ar = pa.context().arena()
    print(f"[{sys.argv[2]:28s}] 256^3 MG-PCG iteration {min(out):.3f} ms (of {[round(v, 3) for v in out]}), arena {ar['class_gib']} cells {ar['cells'][:80]}", flush=True)
else:
    for tag, env in (("default", {}), ("host encodings", {"PA_SETUP_DEVICE": "0"}), ("64 GiB extents", {"PA_ARENA_EXTENT_GIB": "64"}),
                     ("no arena", {"PA_ARENA": "0"}), ("default again", {})):
        e = dict(os.environ); e.update(env)
        subprocess.run([sys.executable, __file__, "child", tag], env=e)
