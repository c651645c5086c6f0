"""Random Cartesian partitions (1-3 dimensions, random part grids, integer ghost layers, periodic directions): the index
sets against the oracle's, then consistent!, assemble!, dot and norm on random vectors, bit for bit in LOCAL order.
python tests/fuzz/fuzz_exchange.py [cases] [seed0]"""
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
    D = int(rng.integers(1, 4))
    while True:
        np_ = tuple(int(rng.integers(1, 5)) for _ in range(D))
        if int(np.prod(np_)) <= 12: break
    ghost = tuple(int(rng.integers(0, 3)) for _ in range(D))
    per = tuple(bool(rng.integers(0, 2)) for _ in range(D))
    # every part must hold at least as many cells as ghost layers on each side (the reference asserts the same)
    n = tuple(int(rng.integers(max(2, p * max(1, 2 * g)), p * max(1, 2 * g) + 14)) for p, g in zip(np_, ghost))
    P = int(np.prod(np_))
    ranks = pa.DebugArray(list(range(1, P + 1)))
    try:
        oparts = orc.uniform_partition(np_, n, ghost, per)
    except AssertionError:
        continue
    parts = pa.uniform_partition(ranks, np_, n, ghost, per)
    fails = []
    for i, o in zip(parts.items, oparts):
        if not (np.array_equal(i.get_local_to_global(), o.local_to_global) and np.array_equal(i.get_local_to_owner(), o.local_to_owner)
                and (i.n_own, i.n_ghost) == (o.n_own, o.n_ghost)):
            fails.append("index sets"); break
    vo = [rng.standard_normal(o.n_local) for o in oparts]
    it = iter([v.copy() for v in vo])
    v = pa.pvector_from_function(lambda ind: next(it), parts)
    pa.consistent_(v).wait()
    orc.consistent(vo, oparts)
    if not all(np.array_equal(g, e) for g, e in zip(v.local_values().items, vo)): fails.append("consistent!")
    d, dref = pa.dot(v, v), orc.dot(vo, vo, oparts)
    if abs(d - dref) > 1e-12 * abs(dref): fails.append("dot")
    nr, nref = pa.norm(v), orc.norm2(vo, oparts)
    if abs(nr - nref) > 1e-12 * abs(nref): fails.append("norm")
    wo = [rng.standard_normal(o.n_local) for o in oparts]
    it = iter([w.copy() for w in wo])
    w = pa.pvector_from_function(lambda ind: next(it), parts)
    pa.assemble_(w).wait()
    orc.assemble(wo, oparts)
    if not all(np.array_equal(g, e) for g, e in zip(w.local_values().items, wo)): fails.append("assemble!")
    if fails:
        bad += 1
        print(f"MISMATCH case {seed0 + case}: np {np_} n {n} ghost {ghost} periodic {per}: {fails}", flush=True)
    if case % 50 == 49:
        print(f"{case + 1} cases, {bad} with mismatches, {time.time() - t0:.0f} s", flush=True)
print(f"done: {n_cases} cases, {bad} with mismatches")
