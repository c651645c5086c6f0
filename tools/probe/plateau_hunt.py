"""Is this lease one that sits on the middle plateau (0.72 ms per launch at 27-pt 256^3)?  If so: what makes it leave?
python tools/probe/plateau_hunt.py   (prints FAST LEASE and stops, or SLOW LEASE and a series of experiments)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("PA_SPMV_VALUE_DICT", "0")
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
ctx = pa.context()
n = 256
A, _ = pa.build_p_matrix(pa.DebugArray([1]), n, n, n, n, n, n, 1, 1, 1)
blk = pa.local_items(A.matrix_partition)[0].own_own
x = pa.pvector_from_function(lambda ind: np.random.default_rng(0).standard_normal(ind.n_local), A.col_partition)
y = pa.pzeros(A.row_partition)
xv, yv = pa.local_items(x.vector_partition)[0], pa.local_items(y.vector_partition)[0]

def groups(k, per=10):
    ev = [ctx.event().record(L.STREAM_COMPUTE)]
    for _ in range(k):
        for _ in range(per): pa.spmv_(yv, blk, xv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0)
        ev.append(ctx.event().record(L.STREAM_COMPUTE))
    ctx.sync()
    return [ev[j].elapsed_ms(ev[j + 1]) / per for j in range(k)]

def show(tag, v):
    print(f"{tag:52s}", " ".join(f"{t:.4f}" for t in v), flush=True)

ctx.sync()
first = groups(15)
show("first 150 launches", first)
if np.median(first[-5:]) < 0.70:
    print("FAST LEASE", ctx.telemetry().get("sclk"), flush=True)
    sys.exit(0)
print("SLOW LEASE", ctx.telemetry(), flush=True)
show("3000 more launches (per 100)", groups(30, 100))
va, vb = pa.DeviceVector(1 << 27, 0), pa.DeviceVector(1 << 27, 0)
va.fill(1.0); vb.fill(2.0)
for _ in range(60): L.call("pa_vec_copy", vb.h, va.h, L.SEG_OWN)
ctx.sync()
show("after 60 x 1 GiB device copies", groups(15))
time.sleep(1.0)
show("after 1 s asleep", groups(15))
big = [pa.DeviceVector(1 << 29, 0) for _ in range(4)]
for b in big: b.fill(0.0)
ctx.sync(); del big
show("after 16 GiB allocated, filled, freed", groups(15))
y2 = pa.pzeros(A.row_partition); yv2 = pa.local_items(y2.vector_partition)[0]
ev = [ctx.event().record(L.STREAM_COMPUTE)]
for _ in range(15):
    for _ in range(10): pa.spmv_(yv2, blk, xv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0)
    ev.append(ctx.event().record(L.STREAM_COMPUTE))
ctx.sync()
show("into another y", [ev[j].elapsed_ms(ev[j + 1]) / 10 for j in range(15)])
print(pa.tune_output_placement(blk, xv, yv, L.SEG_OWN, reps=10, rounds=3), flush=True)
show("after the placement A/B", groups(15))
show("5000 more launches (per 250)", groups(20, 250))
print(ctx.telemetry(), flush=True)
