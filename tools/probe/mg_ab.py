"""MG-PCG iteration time at 256^3: this tree against another checkout of the package on the same box (A/B)."""
import os, subprocess, sys, time
HERE = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    root, tag, n = sys.argv[2], sys.argv[3], int(sys.argv[4])
    sys.path.insert(0, root)
    from __graft_entry__ import load_package
    pa = load_package()
    t0 = time.perf_counter()
    S = pa.pc_setup(pa.DebugArray([1]), 1, 4, n, n, n, "multicolor_spmv")
    pa.context().sync()
    ts = time.perf_counter() - t0
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
    print(f"[{tag:24s}] {n}^3 MG-PCG iteration {min(out):.3f} ms (of {[round(v, 3) for v in out]}), set-up {ts:.1f} s, arena {ar.get('class_gib')} checks {ar.get('checks_ok')}/{ar.get('checks_failed')}", flush=True)
else:
    other = sys.argv[1]
    n = sys.argv[2] if len(sys.argv) > 2 else "256"
    for tag, root, env in (("other", other, {}), ("this", HERE, {}), ("this, no ring", HERE, {"PA_SPMV_XRING": "0"}),
                           ("this, no x windows", HERE, {"PA_SPMV_XWIN": "0"}), ("other again", other, {}), ("this again", HERE, {})):
        e = dict(os.environ); e.update(env)
        subprocess.run([sys.executable, __file__, "child", root, tag, n], env=e)
