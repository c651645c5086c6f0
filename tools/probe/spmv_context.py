"""Why is own x own slower inside the CG loop than back to back?  Event-timed product in different surroundings."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
A, b = pa.build_p_matrix(pa.DebugArray([1]), n, n, n, n, n, n, 1, 1, 1)
ctx = pa.context()
blk = A.matrix_partition.items[0].own_own
N = blk.m
vs = [pa.DeviceVector(N, 0) for _ in range(6)]
for v in vs: v.fill(1.0)
print("matrix class", blk.memory_class(), "vector classes", [v.memory_class() for v in vs], "b class", b.vector_partition.items[0].memory_class())

def timed(body, between, reps=30):
    for _ in range(3): between(); body()
    evs = []
    for _ in range(reps):
        between()
        e0 = ctx.event().record(L.STREAM_COMPUTE); body(); e1 = ctx.event().record(L.STREAM_COMPUTE)
        evs.append((e0, e1))
    ctx.sync()
    t = np.array([a.elapsed_ms(b_) for a, b_ in evs])
    return f"{np.median(t):.4f} (min {t.min():.4f})"
nothing = lambda: None
for xi, yi in ((0, 1), (1, 0), (0, 2), (2, 3)):
    x, y = vs[xi], vs[yi]
    sp = lambda: pa.spmv_(y, blk, x)
    print(f"x=v{xi} (class {x.memory_class()}), y=v{yi} (class {y.memory_class()}): back to back {timed(sp, nothing)}"
          f" | after axpby on x {timed(sp, lambda: L.call('pa_vec_axpby', x.h, 0.5, vs[4].h, 0.5, L.SEG_OWN))}"
          f" | after axpby on two other vectors {timed(sp, lambda: L.call('pa_vec_axpby', vs[5].h, 0.5, vs[4].h, 0.5, L.SEG_OWN))}"
          f" | after a dot of two other vectors {timed(sp, lambda: L.call('pa_vec_dot_slot', vs[5].h, vs[4].h, 5, 0))}", flush=True)
