"""Random inputs through the host logic (no GPU): index partitions with arbitrary ghosts against the oracle's, COO -> CSR
with duplicates and skipped ids, the split into own/ghost blocks, and the library's own self-checks of the row split, the
column encodings and the x-window groups on random blocks.  Fixed seeds; the device-side fuzzers live in tests/fuzz/."""
import ctypes as C

import numpy as np
import pytest

from __graft_entry__ import load_package, load_oracle

pa = load_package()


@pytest.fixture(scope="module")
def orc():
    return load_oracle()


def ranks(P):
    return pa.DebugArray(list(range(1, P + 1)))


def test_random_partitions_with_arbitrary_ghosts_match_the_oracle(orc):
    for seed in range(300):
        rng = np.random.default_rng(9000 + seed)
        if rng.random() < 0.5:
            P = int(rng.integers(1, 9))
            n_own = [int(rng.integers(0, 30)) if rng.random() < 0.8 else 0 for _ in range(P)]
            if sum(n_own) == 0:
                n_own[-1] = 3
            n = sum(n_own)
            parts = pa.variable_partition(pa.DebugArray(list(n_own)), n)
            oparts = orc.variable_partition(list(n_own), n)
        else:
            D = int(rng.integers(1, 4))
            np_ = tuple(int(rng.integers(1, 4)) for _ in range(D))
            nn = tuple(int(rng.integers(p, p + 8)) for p in np_)
            P, n = int(np.prod(np_)), int(np.prod(nn))
            parts = pa.uniform_partition(ranks(P), np_, nn)
            oparts = orc.uniform_partition(np_, nn)
        req = [rng.integers(1, n + 1, int(rng.integers(0, 25))).astype(np.int64) for _ in range(P)]
        owners = pa.find_owner(parts, pa.DebugArray([r.copy() for r in req]))
        oowners = orc.find_owner(oparts, [r.copy() for r in req])
        assert all(np.array_equal(a, b) for a, b in zip(owners.items, oowners)), seed
        parts = pa.pmap(pa.union_ghost, parts, pa.DebugArray([r.copy() for r in req]), owners)
        oparts = [orc.union_ghost(o, r, w) for o, r, w in zip(oparts, req, oowners)]
        for i, o in zip(parts.items, oparts):
            assert np.array_equal(i.get_local_to_global(), o.local_to_global) and np.array_equal(i.get_local_to_owner(), o.local_to_owner), seed
        snd, rcv = pa.assembly_neighbors(parts)
        osnd, orcv = orc.assembly_neighbors(oparts)
        assert all(np.array_equal(a, b) for a, b in zip(snd.items, osnd)) and all(np.array_equal(a, b) for a, b in zip(rcv.items, orcv)), seed
        ls, lr = pa.assembly_local_indices(parts)
        ols, olr = orc.assembly_local_indices(oparts)
        for a, b in zip(list(ls.items) + list(lr.items), list(ols) + list(olr)):
            assert np.array_equal(a.data, b.data) and np.array_equal(a.ptrs, b.ptrs), seed


def test_random_cartesian_partitions_with_ghost_layers_match_the_oracle(orc):
    done = 0
    for seed in range(600):
        rng = np.random.default_rng(12000 + seed)
        D = int(rng.integers(1, 4))
        np_ = tuple(int(rng.integers(1, 5)) for _ in range(D))
        if int(np.prod(np_)) > 12:
            continue
        ghost = tuple(int(rng.integers(0, 3)) for _ in range(D))
        per = tuple(bool(rng.integers(0, 2)) for _ in range(D))
        n = tuple(int(rng.integers(max(2, p * max(1, 2 * g)), p * max(1, 2 * g) + 12)) for p, g in zip(np_, ghost))
        try:
            oparts = orc.uniform_partition(np_, n, ghost, per)
        except AssertionError:
            continue
        parts = pa.uniform_partition(ranks(int(np.prod(np_))), np_, n, ghost, per)
        for i, o in zip(parts.items, oparts):
            assert (i.n_own, i.n_ghost) == (o.n_own, o.n_ghost), (seed, np_, n, ghost, per)
            assert np.array_equal(i.get_local_to_global(), o.local_to_global) and np.array_equal(i.get_local_to_owner(), o.local_to_owner), seed
        done += 1
    assert done > 300


def test_random_coo_to_csr_and_block_split_match_the_oracle(orc):
    for seed in range(200):
        rng = np.random.default_rng(15000 + seed)
        m, n = int(rng.integers(1, 60)), int(rng.integers(1, 60))
        k = int(rng.integers(0, 400))
        lo = 0 if rng.random() < 0.5 else 1                        # ids < 1 are the "skipped" entries of an assembly
        I, J, V = rng.integers(lo, m + 1, k), rng.integers(lo, n + 1, k), rng.standard_normal(k)
        A = pa.compresscoo(I, J, V, m, n, skip=True)
        O = orc.compresscoo_csr(I, J, V, m, n, skip=True)
        assert np.array_equal(A.rowptr, O.rowptr) and np.array_equal(A.colval, O.colval) and np.array_equal(A.nzval, O.nzval), seed


def _call(name, A):
    import pa_amd._lib as L
    v = [C.c_int64() for _ in range(5 if name == "pa_host_check_xw_groups" else 4)]
    L.call(name, A.m, A.n, A.nnz, L.ptr(A.rowptr), L.ptr(A.colval), 1, *[C.byref(x) for x in v])
    return [x.value for x in v]


def test_row_split_encodings_and_window_groups_hold_on_random_blocks():
    """pa_host_check_spmv_encodings decodes every entry of both column encodings back to the caller's column;
    pa_host_check_xw_groups checks that every chunk runs exactly once and every column of a group lies in its window."""
    for seed in range(60):
        rng = np.random.default_rng(18000 + seed)
        m = int(rng.integers(2_000, 120_000))
        n = m if rng.random() < 0.6 else int(m * rng.uniform(0.5, 1.6)) + 1
        law = int(rng.integers(0, 5))
        lens = (np.full(m, int(rng.integers(1, 33))) if law == 0 else rng.integers(0, int(rng.integers(2, 60)), m) if law == 1 else
                np.where(rng.random(m) < 0.01, rng.integers(500, 4000, m), rng.integers(0, 10, m)) if law == 2 else
                np.where(rng.random(m) < 0.5, 0, rng.integers(1, 20, m)) if law == 3 else
                np.repeat(rng.integers(1, 40, (m + 63) // 64), 64)[:m])
        band = int(rng.choice([8, 400, 1500, 3500, 9000, 10**9]))
        rp = np.concatenate([[1], 1 + np.cumsum(lens)]).astype(np.int32)
        rows = np.repeat(np.arange(m), lens)
        centre = (rows * (n / m)).astype(np.int64)
        col = rng.integers(0, n, len(rows)) if band >= 10**9 else np.clip(centre + rng.integers(-band, band + 1, len(rows)), 0, n - 1)
        order = np.lexsort((col, rows))
        H = pa.HostCSR(m, n, rp, (col[order] + 1).astype(np.int32), np.ones(len(rows)))
        chunks, n_pat, n_c16, _ = _call("pa_host_check_spmv_encodings", H)
        assert chunks > 0 or H.nnz == 0
        groups, in_groups, staged, entries, big = _call("pa_host_check_xw_groups", H)
        assert 0 <= big <= groups and in_groups <= max(chunks, 1) and entries <= H.nnz, seed
