"""The lean form of pattern-ELL (pa_pell_slab_fast, csrc/pa_pell.h) against the masked form: 27-pt n^3 own x own product on the fp64
stream and on one bit per entry, and an MG-PCG iteration (colour sweeps / restriction on every other row: stride-2 slabs).
  python tools/probe/pell_lean_time.py 256 128 [mg]"""
import os, sys, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [256]
ctx = pa.context()
out = {}
for n in sizes:
    ys = {}
    for vd in ("0", "1"):
        for lean in ("1", "0"):
            os.environ["PA_SPMV_VALUE_DICT"] = vd
            os.environ["PA_SPMV_PELL_LEAN"] = lean
            ctx.reload_env()
            A, _ = pa.build_p_matrix(pa.DebugArray([1]), n, n, n, n, n, n, 1, 1, 1)
            blk = A.matrix_partition.items[0].own_own
            x = pa.DeviceVector(blk.n, 0).upload(np.random.default_rng(1).standard_normal(blk.n))
            y = pa.DeviceVector(blk.m, 0)
            nl = max(200, int(1.0e9 / max(blk.nnz, 1)))
            for _ in range(3 * nl): pa.spmv_(y, blk, x)
            ctx.sync()
            ts = []
            for r in range(5):
                e0 = ctx.event().record(L.STREAM_COMPUTE)
                for _ in range(nl): pa.spmv_(y, blk, x)
                e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
                ts.append(e0.elapsed_ms(e1) / nl)
            ts.sort()
            ys[(vd, lean)] = y.download()
            info = blk.pell()
            mv = blk.stream_bytes() + 16 * blk.m
            print(f"n={n} dict={vd} lean={lean} -> {info}: min {ts[0]:.4f} med {ts[len(ts)//2]:.4f} ms  {2*blk.nnz/ts[0]/1e6:.0f} GFLOP/s  "
                  f"moved {mv} B = {mv/ts[0]/1e6:.0f} GB/s", flush=True)
            out[f"n{n}_dict{vd}_lean{lean}"] = {"min_ms": ts[0], "med_ms": ts[len(ts)//2], "mode": info["mode"], "moved": mv}
            del A, blk, x, y
    ref = ys[("0", "0")]
    print(f"n={n}: bit-identical products:", {k: bool(np.array_equal(v, ref)) for k, v in ys.items()}, flush=True)
if "mg" in sys.argv:
    n = sizes[0]
    for vd in ("1", "0"):
        for lean in ("1", "0"):
            os.environ["PA_SPMV_VALUE_DICT"] = vd
            os.environ["PA_SPMV_PELL_LEAN"] = lean
            ctx.reload_env()
            S = pa.pc_setup(pa.DebugArray([1]), 1, 4, n, n, n, "multicolor_spmv")
            A, b = S.A_vec[-1], S.r[-1]
            pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=25, Pl=S, fuse=True)
            ts = []
            for rep in range(3):
                ctx.sync(); t = time.perf_counter()
                x, r0, r, it = pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=30, Pl=S, fuse=True)
                ctx.sync(); ts.append((time.perf_counter() - t) / 30 * 1e3)
            print(f"MG-PCG n={n} dict={vd} lean={lean}: {min(ts):.3f} ms per iteration (of {[round(v, 3) for v in ts]}), r/r0 {r / r0:.6e}", flush=True)
            out[f"mg{n}_dict{vd}_lean{lean}"] = min(ts)
            del S, A, b, x
print(json.dumps(out))
