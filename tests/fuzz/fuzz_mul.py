"""Random PSparseMatrices on random numbers of parts (1-D block partitions, rows of random length, columns at random inside a
band or anywhere: irregular ghost sets on every part) through mul!, mul!(...,alpha,beta), the one-call product, the transpose
product, consistent!, assemble! and dot against the oracle, bit for bit.  python tests/fuzz/fuzz_mul.py [cases] [seed0]"""
import sys, time
sys.path.insert(0, '.')
import numpy as np
from __graft_entry__ import load_package, load_oracle
pa = load_package()
orc = load_oracle()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
t0 = time.time()
bad = 0
def upload(parts, partition):
    it = iter(parts)
    return pa.pvector_from_function(lambda ind: next(it), partition)
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    P = int(rng.choice([1, 2, 3, 4, 5, 8]))
    n = int(rng.integers(max(64, 8 * P), 60_000))
    ranks = pa.DebugArray(list(range(1, P + 1)))
    rows = pa.uniform_partition(ranks, n)
    orows = orc.uniform_partition(P, n)
    band = int(rng.choice([3, 40, 700, 10**9]))
    if "-v" in sys.argv: print(f"case {seed0 + case}: P {P} n {n} band {band}", flush=True)
    Is, Js, Vs = [], [], []
    for ind in orows:
        g = ind.own_to_global
        lens = rng.integers(0, int(rng.integers(2, 30)), len(g))
        I = np.repeat(g, lens)
        J = rng.integers(1, n + 1, len(I)) if band >= 10**9 else np.clip(I + rng.integers(-band, band + 1, len(I)), 1, n)
        Is.append(I.astype(np.int64)); Js.append(J.astype(np.int64)); Vs.append(rng.standard_normal(len(I)))
    A = pa.psparse_from_coo(pa.DebugArray([a.copy() for a in Is]), pa.DebugArray([a.copy() for a in Js]),
                            pa.DebugArray([a.copy() for a in Vs]), rows, keep_host=True)
    Ao = orc.psparse_from_coo([a.copy() for a in Is], [a.copy() for a in Js], [a.copy() for a in Vs], orows)
    fails = []
    V_ = "-vv" in sys.argv
    def mark(what):
        if V_:
            pa.context().sync()
            print(f"   ok up to: {what}", flush=True)
    mark("psparse")
    xo = [rng.standard_normal(c.n_local) * (c.local_to_owner == c.part) for c in Ao.cols]
    x = upload([v.copy() for v in xo], A.col_partition)
    y = pa.pzeros(A.row_partition)
    pa.mul_(y, A, x)
    mark("mul!")
    yo = [np.zeros(r.n_local) for r in Ao.rows]
    orc.mul(yo, Ao, [v.copy() for v in xo])
    if not all(np.array_equal(g, e[:r.n_own]) for g, e, r in zip(y.own_values().items, yo, Ao.rows)): fails.append("mul!")
    xc = [v.copy() for v in xo]; orc.consistent(xc, Ao.cols)
    if not all(np.array_equal(g, e) for g, e in zip(x.local_values().items, xc)): fails.append("ghosts after mul!")
    y2 = pa.pzeros(A.row_partition)
    pa.mul_c_(y2, A, x)
    mark("one-call mul!")
    if not all(np.array_equal(g, e[:r.n_own]) for g, e, r in zip(y2.own_values().items, yo, Ao.rows)): fails.append("one-call mul!")
    alpha, beta = float(rng.standard_normal()), float(rng.standard_normal())
    c0 = [rng.standard_normal(r.n_local) for r in Ao.rows]
    c = upload([v.copy() for v in c0], A.row_partition)
    pa.mul5_(c, A, x, alpha, beta)
    mark("mul5")
    co = [v.copy() for v in c0]
    orc.mul5(co, Ao, [v.copy() for v in xo], alpha, beta)
    if not all(np.array_equal(g, e[:r.n_own]) for g, e, r in zip(c.own_values().items, co, Ao.rows)): fails.append("mul!(alpha,beta)")
    bt = [rng.standard_normal(r.n_local) for r in Ao.rows]
    ct0 = [rng.standard_normal(cc.n_local) for cc in Ao.cols]
    ct = upload([v.copy() for v in ct0], A.col_partition)
    pa.mul5_transpose_(ct, A, upload([v.copy() for v in bt], A.row_partition), alpha, beta)
    mark("transpose")
    cto = [v.copy() for v in ct0]
    orc.mul5_transpose(cto, Ao, [v.copy() for v in bt], alpha, beta)
    if not all(np.array_equal(g, e) for g, e in zip(ct.local_values().items, cto)): fails.append("transpose product")
    v0 = [rng.standard_normal(cc.n_local) for cc in Ao.cols]
    v = upload([w.copy() for w in v0], A.col_partition)
    pa.assemble_(v).wait()
    mark("assemble!")
    vo = [w.copy() for w in v0]; orc.assemble(vo, Ao.cols)
    if not all(np.array_equal(g, e) for g, e in zip(v.local_values().items, vo)): fails.append("assemble!")
    d, dref = pa.dot(x, x), orc.dot(xc, xc, Ao.cols)
    if abs(d - dref) > 1e-12 * abs(dref) + 1e-300: fails.append("dot")
    if fails:
        bad += 1
        print(f"MISMATCH case {seed0 + case}: P {P} n {n} band {band}: {fails}", flush=True)
    if case % 10 == 9:
        print(f"{case + 1} cases, {bad} with mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"done: {n_cases} cases, {bad} with mismatches")
