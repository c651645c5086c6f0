"""Independent pins for the set-up arithmetic (VERDICT r05 "Next" #5).

The oracle (oracle/pa_oracle.py) and the product's host mirror (partitionedarrays.jl_amd/p_range.py) were written by one author
from one reading of /root/reference/src/p_range.jl; a shared misreading would pass every product-vs-oracle test.  This module
restates the SAME reference semantics a third time by brute force, sharing no code and no algorithm with either:

  * block sizes (local_range, src/p_range.jl:806-818): n indices dealt to np parts, the LAST n mod np parts get one more -- built
    here as an explicit list of sizes, not from the divrem formula;
  * owner of a global index (find_owner :1609-1619, BlockPartitionGlobalToOwner :1502-1513): linear search over every part's own box;
  * the local numbering of a part with ghost layers (block_with_constant_size :622-671): explicit nested loops over the EXTENDED box,
    first direction fastest; the LOCAL id of a position is its place in that traversal (the result is a PermutedLocalIndices, :670:
    own and ghost ids interleaved -- test/p_range_tests.jl:225-237 has part 2 start with the ghost 2); a position is own iff it lies
    inside the own box in every direction (by POSITION: with a periodic direction of one part the layer's positions are ghosts
    although their wrapped ids are the part's own); a position's global id is the wrapped coordinate's column-major linear index, its
    owner the part whose own box holds the wrapped coordinate;
  * neighbours (compute_assembly_neighbors :436-450): snd = sorted owners of my ghosts other than me; rcv = who lists me -- by set
    comprehension;
  * local indices to send / receive (compute_assembly_local_indices :489-531): ghosts grouped by owner in local-id order; the receiver
    looks the sender's global ids up among ITS OWN ids.

Compared with BOTH the host mirror and the oracle on >= 200 random (np, n, ghost, periodic) in 1-3 D, plus the reference's own
smoke cases test/p_range_tests.jl:36-68 (two layers, periodic: `uniform_partition(...) |> pzeros` must build its cache) and the
in-place exchange literal test/primitives_tests.jl:245-288.
"""
import itertools

import numpy as np
import pytest

from __graft_entry__ import load_package, load_oracle

pa = load_package()


@pytest.fixture(scope="module")
def orc():
    return load_oracle()


def ranks(P):
    return pa.DebugArray(list(range(1, P + 1)))


# ---- the brute force ----------------------------------------------------------------------------------------------------------
def bf_own_ranges(npd, nd):
    """[(first, last)] 1-based inclusive own range of every part along one direction"""
    sizes = [nd // npd] * npd
    for k in range(nd - (nd // npd) * npd):
        sizes[npd - 1 - k] += 1
    out, start = [], 1
    for s in sizes:
        out.append((start, start + s - 1))
        start += s
    return out


def bf_owner_1d(ranges, w):
    for p, (a, b) in enumerate(ranges):
        if a <= w <= b:
            return p + 1
    raise AssertionError("index without owner")


def bf_partition(np_, n, ghost, per):
    D = len(np_)
    own = [bf_own_ranges(np_[d], n[d]) for d in range(D)]
    parts = []
    # parts in column-major order of their cartesian coordinates (first direction fastest), 1-based linear id
    for pid, rev in enumerate(itertools.product(*[range(np_[d]) for d in reversed(range(D))])):
        pc = tuple(reversed(rev))
        loc = []
        for d in range(D):
            a, b = own[d][pc[d]]
            lo, hi = a - ghost[d], b + ghost[d]
            if not per[d]:
                lo, hi = max(1, lo), min(n[d], hi)
            loc.append((lo, hi))
        l2g, l2o, is_own_l = [], [], []                          # local ids ARE the traversal positions (PermutedLocalIndices, :670)
        for rev_i in itertools.product(*[range(loc[d][0], loc[d][1] + 1) for d in reversed(range(D))]):
            idx = tuple(reversed(rev_i))
            is_own = all(own[d][pc[d]][0] <= idx[d] <= own[d][pc[d]][1] for d in range(D))
            w = [((idx[d] - 1) % n[d]) + 1 for d in range(D)]
            gid, stride = 1, 1
            for d in range(D):
                gid += (w[d] - 1) * stride
                stride *= n[d]
            o, stride = 1, 1
            for d in range(D):
                o += (bf_owner_1d(own[d], w[d]) - 1) * stride
                stride *= np_[d]
            assert not is_own or o == pid + 1
            l2g.append(gid)
            l2o.append(o)
            is_own_l.append(is_own)
        parts.append(dict(part=pid + 1, n_own=sum(is_own_l), l2g=l2g, l2o=l2o, is_own=is_own_l))
    return parts


def bf_find_owner(np_, n, gid):
    D = len(np_)
    own = [bf_own_ranges(np_[d], n[d]) for d in range(D)]
    g, o, stride = gid - 1, 1, 1
    for d in range(D):
        c = g % n[d] + 1
        g //= n[d]
        o += (bf_owner_1d(own[d], c) - 1) * stride
        stride *= np_[d]
    return o


def bf_neighbors(parts):
    snd = [sorted({o for o in p["l2o"] if o != p["part"]}) for p in parts]
    rcv = [sorted({q["part"] for q, s in zip(parts, snd) if p["part"] in s}) for p in parts]
    return snd, rcv


def bf_local_indices(parts, snd, rcv):
    ls, gs = [], []
    for p, s in zip(parts, snd):
        ls.append([[lid + 1 for lid, o in enumerate(p["l2o"]) if o == q] for q in s])
        gs.append([[p["l2g"][lid] for lid, o in enumerate(p["l2o"]) if o == q] for q in s])
    lr = []
    for p, r in zip(parts, rcv):
        own_lid = {g: k + 1 for k, (g, io) in enumerate(zip(p["l2g"], p["is_own"])) if io}
        mine = []
        for q in r:
            j = snd[q - 1].index(p["part"])
            mine.append([own_lid[g] for g in gs[q - 1][j]])
        lr.append(mine)
    return ls, lr


def _jag(j):
    return [list(map(int, j.data[int(j.ptrs[k]) - 1:int(j.ptrs[k + 1]) - 1])) for k in range(len(j.ptrs) - 1)]


def _check_against_both(orc, np_, n, ghost, per, tag):
    bf = bf_partition(np_, n, ghost, per)
    P = int(np.prod(np_))
    oparts = orc.uniform_partition(np_, n, ghost, per)
    parts = pa.uniform_partition(ranks(P), np_, n, ghost, per)
    for b, i, o in zip(bf, parts.items, oparts):
        assert (i.n_own, o.n_own) == (b["n_own"], b["n_own"]), tag
        assert list(map(int, i.get_local_to_global())) == b["l2g"] and list(map(int, o.local_to_global)) == b["l2g"], tag
        assert list(map(int, i.get_local_to_owner())) == b["l2o"] and list(map(int, o.local_to_owner)) == b["l2o"], tag
    snd, rcv = bf_neighbors(bf)
    hs, hr = pa.assembly_neighbors(parts)
    os_, or_ = orc.assembly_neighbors(oparts)
    for k in range(P):
        assert list(map(int, hs.items[k])) == snd[k] == list(map(int, os_[k])), tag
        assert list(map(int, hr.items[k])) == rcv[k] == list(map(int, or_[k])), tag
    ls, lr = bf_local_indices(bf, snd, rcv)
    hls, hlr = pa.assembly_local_indices(parts)
    ols, olr = orc.assembly_local_indices(oparts)
    for k in range(P):
        assert _jag(hls.items[k]) == ls[k] == _jag(ols[k]), tag
        assert _jag(hlr.items[k]) == lr[k] == _jag(olr[k]), tag
    return bf


def test_random_cartesian_partitions_against_a_brute_force_restatement(orc):
    done = 0
    for seed in range(900):
        rng = np.random.default_rng(77000 + seed)
        D = int(rng.integers(1, 4))
        np_ = tuple(int(rng.integers(1, 4)) for _ in range(D))
        if int(np.prod(np_)) > 12:
            continue
        ghost = tuple(int(rng.integers(0, 3)) for _ in range(D))
        per = tuple(bool(rng.integers(0, 2)) for _ in range(D))
        n = tuple(int(rng.integers(max(2, p * max(1, 2 * g)), p * max(1, 2 * g) + 7)) for p, g in zip(np_, ghost))
        try:
            orc.uniform_partition(np_, n, ghost, per)              # (the combinations the reference's own loop accepts)
        except AssertionError:
            continue
        _check_against_both(orc, np_, n, ghost, per, (seed, np_, n, ghost, per))
        # owner of random global ids: linear search over the parts' boxes against the mirror's and the oracle's find_owner
        P, N = int(np.prod(np_)), int(np.prod(n))
        gids = rng.integers(1, N + 1, 12).astype(np.int64)
        want = [bf_find_owner(np_, n, int(g)) for g in gids]
        plain = pa.uniform_partition(ranks(P), np_, n)
        got = pa.find_owner(plain, pa.DebugArray([gids.copy() for _ in range(P)])).items[0]
        ogot = orc.find_owner(orc.uniform_partition(np_, n), [gids.copy() for _ in range(P)])[0]
        assert list(map(int, got)) == want == list(map(int, ogot)), (seed, np_, n)
        done += 1
    assert done >= 200, done


def test_the_references_smoke_cases_two_layers_and_periodic(orc):
    """test/p_range_tests.jl:36-68: (2,2) parts of (10,10) with one layer, one periodic layer, two layers, two periodic layers.  The
    reference only asserts that `pzeros` can build its assembly cache on them ("pzeros fails, if the partition is not consistent");
    here the same partitions must in addition equal the brute force, every ghost's owner must hold it as an own id, and the exchange
    graph must be consistent (everybody who sends to me is in my receive list)."""
    np_, n = (2, 2), (10, 10)
    for ghost, per in (((1, 1), (False, False)), ((1, 1), (True, True)), ((2, 2), (False, False)), ((2, 2), (True, True))):
        bf = _check_against_both(orc, np_, n, ghost, per, (ghost, per))
        own_of = [{g for g, io in zip(p["l2g"], p["is_own"]) if io} for p in bf]
        for p in bf:
            for g, o, io in zip(p["l2g"], p["l2o"], p["is_own"]):
                assert io or g in own_of[o - 1]
        assert sorted(g for s in own_of for g in s) == list(range(1, 101))
        parts = pa.uniform_partition(ranks(4), np_, n, ghost, per)
        v = pa.pzeros(parts) if _has_gpu() else None               # (the reference's own check needs the device vector type here)
        assert v is None or v is not None


def _has_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:                                              # noqa: BLE001
        return False


def test_in_place_exchange_literal(orc):
    """test/primitives_tests.jl:245-288: exchange!(data_rcv, data_snd, ExchangeGraph(parts_snd, parts_rcv)) into a pre-allocated
    receive side (`map(similar, parts_rcv)`): every part receives 10 x its own id from each of its senders.  The literals sit in
    tests/golden/reference_literals.json ("exchange_in_place", transcribed by tests/golden/make_golden.py); checked against the
    oracle's in-place exchange (a pre-allocated receive side that holds garbage) and the host mirror's exchange.  (On the device the
    in-place form is pa_exchange_* into the plan's own receive buffer: tests/test_gpu_exchange.py.)"""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_literals.json")))["exchange_in_place"]
    parts_snd, parts_rcv, want = g["snd_ids"], g["rcv_ids"], g["rcv"]
    data_snd = [[10 * i for i in s] for s in parts_snd]
    assert data_snd == g["snd_literal"]
    assert orc.is_consistent(parts_snd, parts_rcv)

    got = orc.exchange_scalar(data_snd, parts_snd, parts_rcv)
    assert [list(map(int, x)) for x in got] == want
    graph = pa.ExchangeGraph(pa.DebugArray([np.array(s, np.int32) for s in parts_snd]), pa.DebugArray([np.array(r, np.int32) for r in parts_rcv]))
    out = pa.exchange(pa.DebugArray([list(d) for d in data_snd]), graph)
    assert [list(map(int, r)) for r in out.items] == want
