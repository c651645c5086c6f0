"""Is a step of the headline loop (pa_mul5 through mul_c_) slower than the product kernel alone queued back to back?  One part."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ctx = pa.context()
A, b = pa.build_p_matrix(pa.DebugArray([1]), n, n, n, n, n, n, 1, 1, 1)
x = pa.pvector_from_function(lambda ind: np.random.default_rng(0).random(ind.n_local), A.col_partition)
y = pa.pzeros(A.row_partition)
blk = pa.local_items(A.matrix_partition)[0]
xv, yv = pa.local_items(x.vector_partition)[0], pa.local_items(y.vector_partition)[0]


def loop(f, reps=50):
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    t_host = time.perf_counter() - t0
    e1 = ctx.event().record(L.STREAM_COMPUTE)
    ctx.sync()
    return e0.elapsed_ms(e1) / reps, t_host / reps * 1e3


fs = {"pa_spmv": lambda: pa.spmv_(yv, blk.own_own, xv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0),
      "mul_c_": lambda: pa.mul_c_(y, A, x),
      "mul_ (composed)": lambda: pa.mul_(y, A, x),
      "spmv oo + spmv oh": lambda: (pa.spmv_(yv, blk.own_own, xv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0), pa.spmv_(yv, blk.own_ghost, xv, L.SEG_GHOST, L.SEG_OWN, 1.0, 1.0))}
for _ in range(300):
    fs["pa_spmv"]()
ctx.sync()
for rnd in range(3):
    for name, f in fs.items():
        for _ in range(20):
            f()
        ms, host = loop(f)
        print(f"round {rnd} {name:22s} device {ms:.4f} ms/step   host enqueue {host:.4f} ms/step", flush=True)
print(ctx.telemetry())
