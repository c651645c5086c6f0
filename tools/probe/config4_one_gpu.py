"""BASELINE config 4's parts -- (2,2,2) x 256^3 rows, 27-point -- all eight on ONE GPU (29 GB of matrix values): what a part's whole mul!
(push, own x own, own x ghost from the receive buffer, unpack) costs beside its own x own alone.  The exchange runs inside the GPU here
(no xGMI), so this is the launch-chain share of an 8-GPU step, not its transport share.  python tools/probe/config4_one_gpu.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("PA_SPMV_VALUE_DICT", "0")
import numpy as np
import bench
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
ctx = pa.context()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
t = time.perf_counter()
A = pa.build_p_matrix(pa.DebugArray(range(1, 9)), n, n, n, 2 * n, 2 * n, 2 * n, 2, 2, 2)[0]
ctx.sync()
print(f"set-up of 8 parts of {n}^3: {time.perf_counter() - t:.1f} s", flush=True)
hx = lambda g: ((np.asarray(g, np.int64) * 2654435761) % 1000003) / 1000003.0 - 0.5
x = pa.pvector_from_function(lambda ind: hx(ind.get_local_to_global()) * (ind.get_local_to_owner() == ind.part), A.col_partition)
y = pa.pzeros(A.row_partition)
blocks = pa.local_items(A.matrix_partition)
print("ghosts per part", [c.n_ghost for c in pa.local_items(A.col_partition)], "own x ghost entries per part", [b.own_ghost.nnz for b in blocks], flush=True)
ms, msg, mss = bench.whole_mul_times(pa, ctx, L, A, x, y, reps=10)
print(f"mul! of all 8 parts: eager {ms / 8:.4f} ms per part, hipGraph replay {msg / 8:.4f}, own x own alone {mss / 8:.4f}  ->  mul!/own x own = {ms / mss:.4f} (eager), {msg / mss:.4f} (graph)", flush=True)
xs, ys = pa.local_items(x.vector_partition), pa.local_items(y.vector_partition)
pa.consistent_(x).wait()
e0 = ctx.event().record(L.STREAM_COMPUTE)
for _ in range(20):
    for blk, xv, yv in zip(blocks, xs, ys):
        pa.spmv_(yv, blk.own_ghost, xv, L.SEG_GHOST, L.SEG_OWN, 1.0, 1.0)
e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
print(f"own x ghost alone: {e0.elapsed_ms(e1) / 160:.4f} ms per part", flush=True)
