cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
export PA_HIP_LIBRARY=$PWD/partitionedarrays.jl_amd/libpa_hip.so.h1
timeout 900 python -m pytest tests/test_gpu_pattern_ell.py -q 2>&1 | grep -E "passed|failed|FAILED|Error|assert|tests/test" | head -20
python - <<'PY'
import os, sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
from test_gpu_pattern_ell import _stencil27
ctx = pa.context()
for n in (128, 200):
    H = _stencil27(n, n, n, few=7)
    x = pa.DeviceVector(H.n, 0).upload(np.random.default_rng(1).standard_normal(H.n))
    ys = {}
    for tag, env in (("pattern-ELL one byte", {"PA_SPMV_VALUE_DICT": "1"}), ("row-split one byte", {"PA_SPMV_VALUE_DICT": "1", "PA_SPMV_PELL_BYTES": "0"}),
                     ("pattern-ELL fp64", {"PA_SPMV_VALUE_DICT": "0"}), ("one byte, masked form", {"PA_SPMV_VALUE_DICT": "1", "PA_SPMV_PELL_LEAN": "0"})):
        for k in ("PA_SPMV_VALUE_DICT", "PA_SPMV_PELL_BYTES", "PA_SPMV_PELL_LEAN"): os.environ.pop(k, None)
        os.environ.update(env); ctx.reload_env()
        A = pa.DeviceCSR(H)
        y = pa.DeviceVector(H.m, 0)
        nl = 300
        for _ in range(2 * nl): pa.spmv_(y, A, x)
        ctx.sync(); ts = []
        for r in range(4):
            e0 = ctx.event().record(L.STREAM_COMPUTE)
            for _ in range(nl): pa.spmv_(y, A, x)
            e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync(); ts.append(e0.elapsed_ms(e1) / nl)
        ys[tag] = y.download()
        mv = A.stream_bytes() + 16 * H.m
        print(f"n={n} {tag:24s} mode {A.pell()['mode']} min {min(ts):.4f} ms {2*H.nnz/min(ts)/1e6:.0f} GFLOP/s moved {mv/1e6:.0f} MB = {mv/min(ts)/1e6:.0f} GB/s", flush=True)
        del A, y
    ref = ys["pattern-ELL fp64"]
    print("bit-identical:", {k: bool(np.array_equal(v, ref)) for k, v in ys.items()})
PY
