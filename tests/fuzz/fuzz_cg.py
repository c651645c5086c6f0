"""The CG loops on random symmetric, strictly diagonally dominant PSparseMatrices (random parts, sizes, bands, row lengths):
opt_cg_(fuse=False) must equal ref_cg_ bit for bit (history and solution), opt_cg_ (fused) within 1e-9 on the history, the
hipGraph replay (one part) must equal the eager fused loop bit for bit, and ref_cg_ must follow the oracle's loop.
python tests/fuzz/fuzz_cg.py [cases] [seed0]"""
import sys, time, functools
sys.path.insert(0, '.')
import numpy as np
from __graft_entry__ import load_package, load_oracle
pa = load_package()
orc = load_oracle()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t0 = time.time()
bad = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    P = int(rng.choice([1, 1, 2, 3, 4, 6]))
    n = int(rng.integers(200, 60_000))
    band = int(rng.choice([2, 30, 500, 1500, n]))
    ranks = pa.DebugArray(list(range(1, P + 1)))
    rows = pa.uniform_partition(ranks, n)
    orows = orc.uniform_partition(P, n)
    k = rng.integers(1, int(rng.integers(2, 12)), n)
    i0 = np.repeat(np.arange(1, n + 1), k)
    j0 = i0 + rng.integers(1, max(2, band), len(i0))
    keep = j0 <= n
    i0, j0 = i0[keep], j0[keep]
    v0 = -rng.random(len(i0)) - 0.1
    diag = np.zeros(n + 1)
    np.add.at(diag, i0, -v0); np.add.at(diag, j0, -v0)
    I = np.concatenate([i0, j0, np.arange(1, n + 1)]); J = np.concatenate([j0, i0, np.arange(1, n + 1)])
    V = np.concatenate([v0, v0, 2.0 * diag[1:] + 1.0])
    order = np.lexsort((J, I)); I, J, V = I[order], J[order], V[order]
    Is, Js, Vs = [], [], []
    for ind in orows:
        if ind.n_own == 0:
            Is.append(np.zeros(0, np.int64)); Js.append(np.zeros(0, np.int64)); Vs.append(np.zeros(0)); continue
        lo, hi = ind.own_to_global[0], ind.own_to_global[-1]
        sel = (I >= lo) & (I <= hi)
        Is.append(I[sel].astype(np.int64)); Js.append(J[sel].astype(np.int64)); Vs.append(V[sel].copy())
    A = pa.psparse_from_coo(pa.DebugArray([a.copy() for a in Is]), pa.DebugArray([a.copy() for a in Js]), pa.DebugArray([a.copy() for a in Vs]), rows)
    xs = pa.pvector_from_function(lambda ind: np.cos(0.01 * ind.get_local_to_global()) * (ind.get_local_to_owner() == ind.part), A.col_partition)
    b = pa.pzeros(A.col_partition)
    pa.mul_(b, A, xs)
    fails, out = [], []
    its = 14
    for fn in (pa.ref_cg_, pa.opt_cg_, functools.partial(pa.opt_cg_, fuse=True)):
        h = []
        x, r0, r, it = fn(pa.pzeros(A.col_partition), A, b, maxiter=its, history=h)
        out.append((r0, r, it, h, [v.copy() for v in x.own_values().items]))
    (r0a, ra, ita, ha, xa), (r0b, rb, itb, hb, xb), (r0c, rc, itc, hc, xc) = out
    if not ((r0a, ra, ita) == (r0b, rb, itb) and ha == hb and all(np.array_equal(u, v) for u, v in zip(xa, xb))): fails.append("unfused != ref_cg")
    if not np.allclose(hc, ha, rtol=1e-9, atol=1e-300): fails.append("fused history")
    if P == 1:
        g1 = pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=its, graph=True, fuse=True)
        g0 = pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=its, fuse=True)
        if not (g1[1:] == g0[1:] and all(np.array_equal(u, v) for u, v in zip(g1[0].own_values().items, g0[0].own_values().items))): fails.append("graph replay")
    Ao = orc.psparse_from_coo([a.copy() for a in Is], [a.copy() for a in Js], [a.copy() for a in Vs], orows)
    bo = [np.zeros(c.n_local) for c in Ao.cols]
    for dst, src, c in zip(bo, b.own_values().items, Ao.cols): dst[:c.n_own] = src
    ho = []
    orc.ref_cg([np.zeros(c.n_local) for c in Ao.cols], Ao, bo, maxiter=its, history=ho, mv=orc.mul)
    if not np.allclose(ho, ha, rtol=1e-8, atol=1e-300): fails.append("ref_cg vs oracle")
    if fails:
        bad += 1
        print(f"MISMATCH case {seed0 + case}: P {P} n {n} band {band}: {fails}", flush=True)
    if case % 20 == 19:
        print(f"{case + 1} cases, {bad} with mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"done: {n_cases} cases, {bad} with mismatches")
