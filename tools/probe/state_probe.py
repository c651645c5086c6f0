"""The headline product runs in one of two states on the same box (0.720 / 0.675 ms per launch, 27-pt 256^3): which launch pattern puts
the GPU into which?  python tools/probe/state_probe.py"""
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

def launch():
    pa.spmv_(yv, blk, xv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0)

def pattern(name, groups, per_group, sync_each):
    evs = [ctx.event().record(L.STREAM_COMPUTE)]
    for _ in range(groups):
        for _ in range(per_group): launch()
        evs.append(ctx.event().record(L.STREAM_COMPUTE))
        if sync_each: ctx.sync()
    ctx.sync()
    ms = [evs[k].elapsed_ms(evs[k + 1]) / per_group for k in range(groups)]
    print(f"{name:46s}", " ".join(f"{v:.4f}" for v in ms), flush=True)

ctx.sync()
pattern("A: 15 x (10 launches, sync)", 15, 10, True)
pattern("B: 12 x 50 launches, no sync", 12, 50, False)
pattern("C: 15 x (10 launches, sync)", 15, 10, True)
time.sleep(0.3)
pattern("D: after 0.3 s idle: 12 x 50, no sync", 12, 50, False)
time.sleep(2.0)
pattern("E: after 2 s idle: 15 x (10, sync)", 15, 10, True)
pattern("F: 40 x 50 launches, no sync (1.4 s)", 40, 50, False)
pattern("G: 15 x (10 launches, sync)", 15, 10, True)
print(ctx.telemetry())
