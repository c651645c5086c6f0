"""Where a workgroup of the product kernel spends its life (round 5).  Needs the variant library with wall-clock stamps:
    bash tools/probe/build_k1_variants.sh stamps "-include $PWD/tools/probe/pa_spmv_probe_hooks.h -DPA_PROBE_STAMPS"
    PA_HIP_LIBRARY=$PWD/tools/probe/build/libpa_hip_stamps.so python tools/probe/k1_stamps.py [n]
One product of the 27-point n^3 block per value stream (fp64 / dictionary) after a warm-up; every workgroup's lane 0 wrote the 100 MHz
wall clock at: 0 entry, 1 chunk head known, 2 its wave's products formed (loads back), 3 behind the barrier, 4 rows summed and stored."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = pa.context()
lib = C.CDLL(L.LIB_PATH)
lib.pa_probe_read_stamps.argtypes = [C.c_void_p, C.c_size_t]
out = {}
for vd in ("0", "1"):
    os.environ["PA_SPMV_VALUE_DICT"] = vd
    A, _ = pa.build_p_matrix(pa.DebugArray([1]), n, n, n, n, n, n, 1, 1, 1)
    blk = A.matrix_partition.items[0].own_own
    x = pa.DeviceVector(blk.n, 0).upload(np.random.default_rng(1).standard_normal(blk.n))
    y = pa.DeviceVector(blk.m, 0)
    for _ in range(400): pa.spmv_(y, blk, x)
    ctx.sync()
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    pa.spmv_(y, blk, x)           # (forwards or backwards: whichever is due -- the stamps are per block index)
    e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
    nb = min(1 << 19, ((blk.info()["n_chunks"] + 7) // 8) * 8)
    buf = np.zeros((nb, 8), np.uint64)
    assert lib.pa_probe_read_stamps(buf.ctypes.data, nb) == 0
    ok = buf[:, 4] > 0
    t = buf[ok, :5].astype(np.int64)
    hw = buf[ok, 7]
    t -= t[:, 0].min()
    us = lambda a: a / 100.0
    ph = np.diff(t, axis=1)
    life = t[:, 4] - t[:, 0]
    q = lambda a: [round(float(v), 2) for v in np.percentile(us(a), [10, 50, 90])]
    # per CU: (xcc, se, cu) from XCC_ID[3:0] and HW_ID (cu_id [11:8], sh [12], se [15:13])
    hwid = (hw & 0xffffffff).astype(np.int64); xcc = ((hw >> 32) & 0xf).astype(np.int64)
    cu = (xcc << 8) | (((hwid >> 13) & 7) << 5) | (((hwid >> 12) & 1) << 4) | ((hwid >> 8) & 15)
    alive, starts = [], []
    for c in np.unique(cu):
        m = cu == c
        s, e = np.sort(t[m, 0]), np.sort(t[m, 4])
        span = e[-1] - s[0]
        alive.append(life[m].sum() / max(span, 1))
        starts.append(span / max(m.sum(), 1))
    rec = {"launch_ms_events": round(e0.elapsed_ms(e1), 4), "workgroups": int(ok.sum()), "cus_seen": int(len(np.unique(cu))),
           "kernel_span_us": round(float(us(t[:, 4].max())), 1),
           "us_p10_p50_p90": {"entry_to_head": q(ph[:, 0]), "head_to_loads_back": q(ph[:, 1]), "loads_back_to_behind_barrier": q(ph[:, 2]),
                              "barrier_to_rows_stored": q(ph[:, 3]), "life": q(life)},
           "mean_workgroups_alive_per_cu": round(float(np.mean(alive)), 2), "mean_us_between_workgroup_completions_per_cu": round(float(us(np.mean(starts))), 3)}
    out["dictionary" if vd == "1" else "fp64"] = rec
    print(("dictionary" if vd == "1" else "fp64"), json.dumps(rec), flush=True)
    del A, blk, x, y
