"""Block partitions with arbitrary ghosts: variable_partition (random own sizes, EMPTY parts included) or a uniform Cartesian
partition, each part then given random ghost ids through find_owner + union_ghost (duplicates, own ids and repeats in the
request, as an assembly would produce them).  Index sets, assembly neighbours and local indices against the oracle, then
consistent!, assemble!, dot on random vectors, bit for bit.  python tests/fuzz/fuzz_partitions.py [cases] [seed0]"""
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
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    fails = []
    if rng.random() < 0.6:
        P = int(rng.integers(1, 9))
        n_own = [int(rng.integers(0, 40)) if rng.random() < 0.8 else 0 for _ in range(P)]
        if sum(n_own) == 0: n_own[0] = 5
        n = sum(n_own)
        ranks = pa.DebugArray(list(range(1, P + 1)))
        parts = pa.variable_partition(pa.DebugArray(list(n_own)), n)
        oparts = orc.variable_partition(list(n_own), n)
        what = f"variable_partition {n_own}"
    else:
        D = int(rng.integers(1, 4))
        np_ = tuple(int(rng.integers(1, 4)) for _ in range(D))
        nn = tuple(int(rng.integers(p, p + 9)) for p in np_)
        P, n = int(np.prod(np_)), int(np.prod(nn))
        ranks = pa.DebugArray(list(range(1, P + 1)))
        parts = pa.uniform_partition(ranks, np_, nn)
        oparts = orc.uniform_partition(np_, nn)
        what = f"uniform_partition {np_} {nn}"
    req = [rng.integers(1, n + 1, int(rng.integers(0, 30))).astype(np.int64) for _ in range(P)]
    owners = pa.find_owner(parts, pa.DebugArray([r.copy() for r in req]))
    oowners = orc.find_owner(oparts, [r.copy() for r in req])
    if not all(np.array_equal(a, b) for a, b in zip(owners.items, oowners)): fails.append("find_owner")
    parts = pa.pmap(pa.union_ghost, parts, pa.DebugArray([r.copy() for r in req]), owners)
    oparts = [orc.union_ghost(o, r, w) for o, r, w in zip(oparts, req, oowners)]
    for i, o in zip(parts.items, oparts):
        if not (np.array_equal(i.get_local_to_global(), o.local_to_global) and np.array_equal(i.get_local_to_owner(), o.local_to_owner)):
            fails.append("index sets"); break
    snd, rcv = pa.assembly_neighbors(parts)
    osnd, orcv = orc.assembly_neighbors(oparts)
    if not all(np.array_equal(a, b) for a, b in zip(snd.items, osnd)) or not all(np.array_equal(a, b) for a, b in zip(rcv.items, orcv)):
        fails.append("assembly_neighbors")
    vo = [rng.standard_normal(o.n_local) for o in oparts]
    it = iter([v.copy() for v in vo])
    v = pa.pvector_from_function(lambda ind: next(it), parts)
    pa.consistent_(v).wait()
    orc.consistent(vo, oparts)
    if not all(np.array_equal(g, e) for g, e in zip(v.local_values().items, vo)): fails.append("consistent!")
    d, dref = pa.dot(v, v), orc.dot(vo, vo, oparts)
    if abs(d - dref) > 1e-12 * abs(dref) + 1e-300: fails.append("dot")
    wo = [rng.standard_normal(o.n_local) for o in oparts]
    it = iter([w.copy() for w in wo])
    w = pa.pvector_from_function(lambda ind: next(it), parts)
    pa.assemble_(w).wait()
    orc.assemble(wo, oparts)
    if not all(np.array_equal(g, e) for g, e in zip(w.local_values().items, wo)): fails.append("assemble!")
    if fails:
        bad += 1
        print(f"MISMATCH case {seed0 + case}: {what}: {fails}", flush=True)
    if case % 100 == 99:
        print(f"{case + 1} cases, {bad} with mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"done: {n_cases} cases, {bad} with mismatches")
