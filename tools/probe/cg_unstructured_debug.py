import os, sys, functools
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
P, n = 4, 240_000
ranks = pa.DebugArray(list(range(1, P + 1)))
rows = pa.uniform_partition(ranks, n)
rng = np.random.default_rng(53)
k = rng.integers(3, 11, n)
i0 = np.repeat(np.arange(1, n + 1), k)
j0 = i0 + rng.integers(1, 1500, len(i0))
keep = j0 <= n
i0, j0 = i0[keep], j0[keep]
v0 = -rng.random(len(i0)) - 0.1
diag = np.zeros(n + 1)
np.add.at(diag, i0, -v0); np.add.at(diag, j0, -v0)
I = np.concatenate([i0, j0, np.arange(1, n + 1)]); J = np.concatenate([j0, i0, np.arange(1, n + 1)]); V = np.concatenate([v0, v0, diag[1:] + 1.0])
order = np.lexsort((J, I)); I, J, V = I[order], J[order], V[order]
Is, Js, Vs = [], [], []
for ind in rows.items:
    g = ind.get_own_to_global() if hasattr(ind, "get_own_to_global") else ind.own_to_global
    lo, hi = g[0], g[-1]
    sel = (I >= lo) & (I <= hi)
    Is.append(I[sel].astype(np.int64)); Js.append(J[sel].astype(np.int64)); Vs.append(V[sel].copy())
for sw in ("1",):
    os.environ["PA_SPMV_XWIN"] = sw
    A = pa.psparse_from_coo(pa.DebugArray([a.copy() for a in Is]), pa.DebugArray([a.copy() for a in Js]), pa.DebugArray([a.copy() for a in Vs]), rows)
    print("xwin", sw, [b.own_own.xwin() for b in A.matrix_partition.items][:1])
    xs = pa.pvector_from_function(lambda ind: np.cos(0.001 * ind.get_local_to_global()) * (ind.get_local_to_owner() == ind.part), A.col_partition)
    b = pa.pzeros(A.col_partition); pa.mul_(b, A, xs)
    H = []
    for fn in (pa.ref_cg_, functools.partial(pa.opt_cg_, fuse=False), pa.opt_cg_):
        h = []
        x, r0, r, it = fn(pa.pzeros(A.col_partition), A, b, maxiter=60, tolerance=1e-10, history=h)
        H.append(h)
    d = [abs(p - q) / q for p, q in zip(H[2], H[0])]
    print("  fused vs ref rel diff per iteration:", " ".join(f"{v:.1e}" for v in d[:30]))
    print("  ref  :", " ".join(f"{v:.3e}" for v in H[0][8:30]))
    print("  fused:", " ".join(f"{v:.3e}" for v in H[2][8:30]))
    d = [abs(p - q) / q for p, q in zip(H[1], H[0])]
    print("  unfused vs ref max:", max(d))
    # one fused product against the separate calls
    import pa_amd.p_sparse_matrix as psm
    c1, c2 = pa.pzeros(A.col_partition), pa.pzeros(A.col_partition)
    pa.mul_c_(c1, A, xs); want = pa.dot(xs, c1)
    psm.mul_dot_(c2, A, xs, 6); got = pa.read_slots(6)[0]
    print("  mul_dot:", got, want, abs(got - want) / abs(want), all(np.array_equal(a, b_) for a, b_ in zip(c1.own_values().items, c2.own_values().items)))
