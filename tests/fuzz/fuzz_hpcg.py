"""The HPCG route at random: build_p_matrix on random local grids and part grids -> mul!, mul_no_lat!, the level-scheduled
Gauss-Seidel sweeps (forward / backward, zero and non-zero guess) against the oracle, bit for bit; a few iterations of the
reference CG loop against the oracle's loop; the fused residual + restriction against the separate kernels.
python tests/fuzz/fuzz_hpcg.py [cases] [seed0]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from __graft_entry__ import load_package, load_oracle
pa = load_package()
orc = load_oracle()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t0 = time.time()
bad = 0
GRIDS = [(1, 1, 1), (2, 1, 1), (1, 2, 1), (2, 2, 1), (2, 2, 2), (3, 1, 1), (1, 1, 4), (3, 2, 1)]
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    parts = GRIDS[int(rng.integers(0, len(GRIDS)))]
    P = int(np.prod(parts))
    n = tuple(int(rng.integers(2, 11)) for _ in range(3))
    ranks = pa.DebugArray(list(range(1, P + 1)))
    fails = []
    fused = bool(rng.integers(0, 2))
    A, b = pa.build_p_matrix(ranks, *n, *(a * q for a, q in zip(n, parts)), *parts, keep_host=True, fused=fused)
    Ao, bo, _ = orc.hpcg_build_p_matrix(*n, *parts)
    xo = [rng.standard_normal(c.n_local) * (c.local_to_owner == c.part) for c in Ao.cols]
    it = iter([v.copy() for v in xo])
    x = pa.pvector_from_function(lambda ind: next(it), A.col_partition)
    y = pa.pzeros(A.row_partition)
    pa.mul_(y, A, x)
    yo = [np.zeros(r.n_local) for r in Ao.rows]
    orc.mul(yo, Ao, [v.copy() for v in xo])
    if not all(np.array_equal(g, e[:r.n_own]) for g, e, r in zip(y.own_values().items, yo, Ao.rows)): fails.append("mul!")
    y2 = pa.pzeros(A.row_partition)
    pa.mul_no_lat_(y2, A, x)
    yn = [np.zeros(r.n_local) for r in Ao.rows]
    orc.mul_no_lat(yn, Ao, [v.copy() for v in xo])
    if not all(np.array_equal(g, e[:r.n_own]) for g, e, r in zip(y2.own_values().items, yn, Ao.rows)): fails.append("mul_no_lat!")
    gs = pa.GaussSeidel(A)
    d = orc.dense_diag(Ao)
    go = [np.zeros(c.n_local) for c in Ao.cols]
    g = pa.pzeros(A.col_partition)
    for zero in (True, False):
        gs.step_(g, b, zero_guess=zero)
        orc.gauss_seidel_step(go, Ao, d, bo, zero_guess=zero)
        if not all(np.array_equal(a_, e) for a_, e in zip(g.local_values().items, go)): fails.append(f"gauss-seidel zero_guess={zero}"); break
    hist, ho = [], []
    xs, r0, r, it_ = pa.ref_cg_(pa.pzeros(A.col_partition), A, b, maxiter=6, overlap=bool(rng.integers(0, 2)), history=hist)
    orc.ref_cg([np.zeros(c.n_local) for c in Ao.cols], Ao, [v.copy() for v in bo], maxiter=6, history=ho)
    if not np.allclose(hist, ho, rtol=1e-9, atol=1e-14 * max(ho[0], 1e-300)): fails.append("ref_cg history")
    # the multigrid preconditioner (level-scheduled Gauss-Seidel = the reference's sweep) on a grid the levels divide
    levels = int(rng.integers(2, 4))
    f = 2 ** (levels - 1)
    nm = tuple(f * int(rng.integers(1, 4)) for _ in range(3))
    S = pa.pc_setup(ranks, P, levels, *nm)
    So = orc.pc_setup(tuple(pa.compute_optimal_shape_XYZ(P)), levels, *nm)   # (pc_setup picks the part grid itself, mg_preconditioner.jl:143)
    Am, bm = S.A_vec[-1], S.r[-1]
    hist, ho = [], []
    pa.ref_cg_(pa.pzeros(Am.col_partition), Am, bm, maxiter=4, overlap=False, history=hist, Pl=S)
    orc.ref_cg_mg([np.zeros(c.n_local) for c in So.A[-1].cols], So.A[-1], So.r[-1], So, maxiter=4, history=ho)
    if not np.allclose(hist, ho, rtol=1e-9, atol=1e-14 * max(ho[0], 1e-300)): fails.append(f"MG-PCG history (levels {levels}, n {nm})")
    if fails:
        bad += 1
        print(f"MISMATCH case {seed0 + case}: n {n} parts {parts} fused {fused}: {fails}", flush=True)
    if case % 20 == 19:
        print(f"{case + 1} cases, {bad} with mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"done: {n_cases} cases, {bad} with mismatches")
