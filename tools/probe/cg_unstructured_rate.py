"""One CG iteration (identity preconditioner) on a symmetric, diagonally dominant matrix without any structure: 4 M rows,
~17 entries per row inside a band of +-2000, one part.  ms per iteration of ref_cg_, opt_cg_(fuse=False) and opt_cg_."""
import sys, time, functools
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
ctx = pa.context()
n = 4_000_000
rng = np.random.default_rng(53)
k = rng.integers(6, 11, n)
i0 = np.repeat(np.arange(1, n + 1), k)
j0 = i0 + rng.integers(1, 2000, len(i0))
keep = j0 <= n
i0, j0 = i0[keep], j0[keep]
v0 = -rng.random(len(i0)) - 0.1
diag = np.zeros(n + 1)
np.add.at(diag, i0, -v0); np.add.at(diag, j0, -v0)
I = np.concatenate([i0, j0, np.arange(1, n + 1)]); J = np.concatenate([j0, i0, np.arange(1, n + 1)])
V = np.concatenate([v0, v0, 2.0 * diag[1:] + 1.0])
ranks = pa.DebugArray([1])
rows = pa.uniform_partition(ranks, n)
A = pa.psparse_from_coo(pa.DebugArray([I]), pa.DebugArray([J]), pa.DebugArray([V]), rows)
blk = A.matrix_partition.items[0].own_own
print("nnz", blk.nnz, "encoding", blk.encoding(), "x windows", blk.xwin(), flush=True)
xs = pa.pvector_from_function(lambda ind: np.cos(0.001 * ind.get_local_to_global()), A.col_partition)
b = pa.pzeros(A.col_partition)
pa.mul_(b, A, xs)
for name, fn in (("ref_cg_", pa.ref_cg_), ("opt_cg_(fuse=False)", functools.partial(pa.opt_cg_, fuse=False)), ("opt_cg_", pa.opt_cg_)):
    fn(pa.pzeros(A.col_partition), A, b, maxiter=40)          # warm (also the clocks)
    ctx.sync(); t = time.perf_counter()
    x, r0, r, it = fn(pa.pzeros(A.col_partition), A, b, maxiter=40)
    ctx.sync(); dt = (time.perf_counter() - t) / 40
    print(f"{name:22s} {dt * 1e3:.4f} ms per iteration, r/r0 after 40: {r / r0:.2e}", flush=True)
y = pa.pzeros(A.row_partition)
import pa_amd._lib as L
for _ in range(200): pa.mul_c_(y, A, xs)
e0 = ctx.event().record(L.STREAM_COMPUTE)
for _ in range(50): pa.mul_c_(y, A, xs)
e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
print(f"mul! alone             {e0.elapsed_ms(e1) / 50:.4f} ms", flush=True)
