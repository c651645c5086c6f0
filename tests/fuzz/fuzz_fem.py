"""The FEM route at random: laplacian_fem on random 2-D / 3-D node counts and part grids -> psparse (disassembled -> assemble),
mul!; the same matrix kept sub-assembled, mul!(...,alpha,beta) (own and ghost rows, then assemble!(c)); psparse! with new
values through the device re-assembly -- each against the oracle, bit for bit.  python tests/fuzz/fuzz_fem.py [cases] [seed0]"""
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
def upload(parts, partition):
    it = iter(parts)
    return pa.pvector_from_function(lambda ind: next(it), partition)
GRIDS = [(1, 1), (2, 1), (1, 3), (2, 2), (4, 2), (3, 2), (1, 1, 1), (2, 1, 1), (2, 2, 1), (2, 2, 2), (1, 3, 2)]
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    parts = GRIDS[int(rng.integers(0, len(GRIDS)))]
    D = len(parts)
    nodes = tuple(int(rng.integers(max(3, 2 * p + 1), 26 if D == 2 else 12)) for p in parts)
    P = int(np.prod(parts))
    ranks = pa.DebugArray(list(range(1, P + 1)))
    fails = []
    I, J, V, rows, cols = pa.laplacian_fem(nodes, parts, ranks)
    Io, Jo, Vo, orows, ocols = orc.laplacian_fem(nodes, parts)
    A, cache = pa.psparse_disassembled(I, J, V, rows, cols, reuse=True)
    Ao, (oblocks, orows_sa, ocols_sa) = orc.psparse_disassembled(Io, Jo, Vo, orows, ocols)
    xo = [rng.standard_normal(c.n_local) * (c.local_to_owner == c.part) for c in Ao.cols]
    x = upload([v.copy() for v in xo], A.col_partition)
    y = pa.pzeros(A.row_partition)
    pa.mul_(y, A, x)
    yo = [np.zeros(r.n_local) for r in Ao.rows]
    orc.mul(yo, Ao, [v.copy() for v in xo])
    if not all(np.array_equal(g, e[:r.n_own]) for g, e, r in zip(y.own_values().items, yo, Ao.rows)): fails.append("mul!")
    # new values on the same pattern: psparse!(C, V2, cache) on the device against a fresh assembly in the oracle
    V2o = [v * float(rng.standard_normal()) + rng.standard_normal(len(v)) * 1e-2 for v in Vo]
    it = iter([v.copy() for v in V2o])
    pa.psparse_(A, pa.pmap(lambda v: next(it), V), cache).wait()
    A2o, _ = orc.psparse_disassembled(Io, Jo, V2o, orows, ocols)
    pa.mul_(y, A, x)
    orc.mul(yo, A2o, [v.copy() for v in xo])
    if not all(np.array_equal(g, e[:r.n_own]) for g, e, r in zip(y.own_values().items, yo, A2o.rows)): fails.append("mul! after psparse!")
    # sub-assembled
    S = pa.psparse_disassembled(I, J, V, rows, cols, assemble=False)
    So = orc.PSparse([None] * P, oblocks, orows_sa, ocols_sa, False)
    alpha, beta = float(rng.standard_normal()), float(rng.standard_normal())
    xs = [rng.standard_normal(c.n_local) * (c.local_to_owner == c.part) for c in ocols_sa]
    ys = [rng.standard_normal(r.n_local) for r in orows_sa]
    xd, yd = upload([v.copy() for v in xs], S.col_partition), upload([v.copy() for v in ys], S.row_partition)
    pa.mul5_(yd, S, xd, alpha, beta)
    orc.mul5(ys, So, xs, alpha, beta)
    if not all(np.array_equal(g, e) for g, e in zip(yd.local_values().items, ys)): fails.append("sub-assembled mul!(alpha,beta)")
    if fails:
        bad += 1
        print(f"MISMATCH case {seed0 + case}: nodes {nodes} parts {parts}: {fails}", flush=True)
    if case % 10 == 9:
        print(f"{case + 1} cases, {bad} with mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"done: {n_cases} cases, {bad} with mismatches")
