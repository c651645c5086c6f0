"""Pattern-ELL (k_spmv_pell, csrc/pa_pell.h) against the row-split kernel on the HPCG 27-pt block: fp64 stream and dictionary /
one bit per entry, n^3 rows for each n given; bit-identity of the four products checked on the way.
  python tools/probe/pell_time.py 256 128 [reps<16]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
sizes = [int(a) for a in sys.argv[1:] if int(a) >= 16] or [256]
reps = next((int(a) for a in sys.argv[1:] if int(a) < 16), 5)
ctx = pa.context()
out = {}
for n in sizes:
    ys = {}
    for vd in ("0", "1"):
        for pell in ("1", "0"):
            os.environ["PA_SPMV_VALUE_DICT"] = vd
            os.environ["PA_SPMV_PELL"] = pell
            ctx.reload_env()
            A, _ = pa.build_p_matrix(pa.DebugArray([1]), n, n, n, n, n, n, 1, 1, 1)
            blk = A.matrix_partition.items[0].own_own
            x = pa.DeviceVector(blk.n, 0).upload(np.random.default_rng(1).standard_normal(blk.n))
            y = pa.DeviceVector(blk.m, 0)
            nl = max(200, int(1.0e9 / max(blk.nnz, 1)))
            for _ in range(3 * nl): pa.spmv_(y, blk, x)
            ctx.sync()
            ts = []
            for r in range(reps):
                e0 = ctx.event().record(L.STREAM_COMPUTE)
                for _ in range(nl): pa.spmv_(y, blk, x)
                e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
                ts.append(e0.elapsed_ms(e1) / nl)
            ts.sort()
            ys[(vd, pell)] = y.download()
            info = blk.pell()
            print(f"n={n} dict={vd} pell={pell} -> mode {info['mode']} (slabs {info['slabs']}, patterns {info['patterns']}, U {info['unroll']}): "
                  f"min {ts[0]:.4f} med {ts[len(ts)//2]:.4f} ms  {2*blk.nnz/ts[0]/1e6:.0f} GFLOP/s  moved {blk.stream_bytes() + 16 * blk.m} B "
                  f"= {(blk.stream_bytes() + 16 * blk.m)/ts[0]/1e6:.0f} GB/s", flush=True)
            out[f"n{n}_dict{vd}_pell{pell}"] = {"min_ms": ts[0], "med_ms": ts[len(ts)//2], "mode": info["mode"]}
            del A, blk, x, y
    ref = ys[("0", "0")]
    print(f"n={n}: bit-identical products:", {k: bool(np.array_equal(v, ref)) for k, v in ys.items()}, flush=True)
print(json.dumps(out))
