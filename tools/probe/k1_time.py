"""The product kernel K1 on the HPCG 27-pt block, fp64 value stream and value dictionary, n^3 rows for each n given.
  python tools/probe/k1_time.py 256 128 [reps]     (PA_K1_VD=0|1: just that stream, for rocprofv3 --pmc passes)"""
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
    for vd in ("0", "1"):
        if os.environ.get("PA_K1_VD") and os.environ["PA_K1_VD"] != vd: continue
        os.environ["PA_SPMV_VALUE_DICT"] = vd
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
        print(f"n={n} vd={vd} dict={blk.value_dict()}: min {ts[0]:.4f} med {ts[len(ts)//2]:.4f} ms  {2*blk.nnz/ts[0]/1e6:.0f} GFLOP/s  stream bytes {blk.stream_bytes()}  classes {blk.memory_class()} {x.memory_class()} {y.memory_class()}", flush=True)
        out[f"n{n}_vd{vd}"] = {"min_ms": ts[0], "med_ms": ts[len(ts)//2]}
        del A, blk, x, y
print(json.dumps(out))
