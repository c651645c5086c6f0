"""Where does a CG iteration's wall time go: host enqueue time per call vs device time."""
import sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
ranks = pa.DebugArray([1])
A, b = pa.build_p_matrix(ranks, n, n, n, n, n, n, 1, 1, 1)
x = pa.pzeros(A.col_partition); u = pa.similar(x); r = pa.similar(x); c = pa.similar(x)
pa.copy_(r, b); pa.copy_(u, b)
ctx = pa.context()
pa.write_slot(1, 1.0); pa.write_slot(2, 3.0); pa.write_slot(4, 1e9)
def timed(name, f, reps=50):
    f(); ctx.sync()
    t0 = time.perf_counter()
    for _ in range(reps):
        f()
    t1 = time.perf_counter()
    ctx.sync()
    t2 = time.perf_counter()
    print(f"{name:28s} host enqueue {1e3*(t1-t0)/reps:8.3f} ms   total {1e3*(t2-t0)/reps:8.3f} ms per call", flush=True)
timed("axpby_slot_", lambda: pa.axpby_slot_(u, 1.0, -1, -1, r, 1.0, 1, 2))
timed("mul_", lambda: pa.mul_(c, A, u))
timed("mul_no_lat_", lambda: pa.mul_no_lat_(c, A, u))
timed("spmv_ own_own only", lambda: pa.spmv_(c.vector_partition.items[0], A.matrix_partition.items[0].own_own, u.vector_partition.items[0]))
timed("dot_slot", lambda: pa.dot_slot(u, c, 4))
timed("cg_update_", lambda: pa.cg_update_(x, r, u, c, 1, 4, 3))
timed("axpby_", lambda: pa.axpby_(u, 1.0, r, 0.5))
timed("dot (host read)", lambda: pa.dot(u, c))
timed("copy_", lambda: pa.copy_(c, r))
print("--- sequences, u = hashed values in [0,1)")
import numpy as np
g = A.col_partition
u = pa.pvector_from_function(lambda i: ((i.get_local_to_global().astype(np.uint64) * np.uint64(2654435761)) % np.uint64(2**32)).astype(np.float64) / 2.0**32, g)
pa.copy_(r, u)
timed("mul_ (hashed u)", lambda: pa.mul_(c, A, u))
timed("axpby_slot_ + mul_", lambda: (pa.axpby_slot_(u, 1.0, -1, -1, r, 0.0, -1, -1), pa.mul_(c, A, u)))
timed("mul_ + dot_slot", lambda: (pa.mul_(c, A, u), pa.dot_slot(u, c, 4)))
pa.write_slot(1, 1e-30)
timed("mul_ + cg_update_", lambda: (pa.mul_(c, A, u), pa.cg_update_(x, r, u, c, 1, 4, 3)))
timed("cg_update_ alone", lambda: pa.cg_update_(x, r, u, c, 1, 4, 3))
timed("full opt iteration", lambda: (pa.axpby_slot_(u, 1.0, -1, -1, r, 0.0, -1, -1), pa.mul_(c, A, u), pa.dot_slot(u, c, 4), pa.cg_update_(x, r, u, c, 1, 4, 3)))
