"""Random call sequences on valid exchange plans: pack / local transport / finish in any order and mode, destroy and reuse.
The split-phase contract (pack -> transport -> finish, each once per exchange, src/primitives.jl:119-141) is enforced with
statuses: a wrong call is refused with a message and changes nothing; a right sequence still gives the oracle's ghosts
afterwards.  python tests/fuzz/fuzz_call_order.py [cases] [seed0]"""
import sys
sys.path.insert(0, '.')
import ctypes as C
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 50
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
bad = refused = accepted = 0
for case in range(n_cases):
    rng = np.random.default_rng(seed0 + case)
    P = int(rng.integers(2, 6))
    n = int(rng.integers(4 * P, 200))
    ranks = pa.DebugArray(list(range(1, P + 1)))
    parts = pa.uniform_partition(ranks, (P,), (n,), (int(rng.integers(1, 3)),), (bool(rng.integers(0, 2)),))
    v = pa.pvector_from_function(lambda ind: np.cos(ind.get_local_to_global().astype(float)), parts)
    cache = v.cache
    plans = cache.plans.items
    vecs = v.vector_partition.items
    arr = (C.c_void_p * P)(*[h.value for h in plans])
    for step in range(int(rng.integers(5, 40))):
        op = int(rng.integers(0, 4))
        mode = int(rng.integers(0, 2))
        p = int(rng.integers(0, P))
        try:
            if op == 0: L.call("pa_exchange_pack", plans[p], vecs[p].h, mode)
            elif op == 1: L.call("pa_exchange_local", arr, P, mode)
            elif op == 2: L.call("pa_exchange_finish", plans[p], vecs[p].h, mode)
            else: pa.context().sync()
            accepted += 1
        except L.PAError:
            refused += 1
    # bring every plan back to idle whatever state the random calls left it in, then a proper exchange must be right
    for p in range(P):
        for mode in (0, 1):
            try: L.call("pa_exchange_finish", plans[p], vecs[p].h, mode)
            except L.PAError: pass
    for p in range(P):
        try: L.call("pa_exchange_pack", plans[p], vecs[p].h, L.CONSISTENT)
        except L.PAError:
            pass
    try:
        L.call("pa_exchange_local", arr, P, L.CONSISTENT)
        for p in range(P): L.call("pa_exchange_finish", plans[p], vecs[p].h, L.CONSISTENT)
    except L.PAError as e:
        bad += 1
        print(f"MISMATCH case {seed0 + case}: a clean exchange was refused after the random calls: {e}", flush=True)
        continue
    # whatever the random calls added or inserted on the way, after a proper consistent! every ghost holds its owner's value
    glob = v.collect()
    for ind, vals in zip(parts.items, v.local_values().items):
        if not np.array_equal(vals, glob[ind.get_local_to_global() - 1]):
            bad += 1
            print(f"MISMATCH case {seed0 + case}: ghosts differ from their owners after a clean consistent!", flush=True)
            break
print(f"done: {n_cases} cases, {bad} with mismatches; {accepted} calls accepted, {refused} refused with a status")
