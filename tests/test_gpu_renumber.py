"""Library-side renumbering for blocks without locality (BASELINE config 5: "unstructured ... irregular"):
pa_csr_locality_order (reverse Cuthill-McKee on the device), pa_csr_create_permuted (rows keep their entries in their original
order), and the host mirror's renumber_for_locality, which hides the order in local_to_device.  Bars: np.array_equal -- a
renumbered matrix gives the SAME BITS as the oracle's mul! (src/p_sparse_matrix.jl:2090-2142) on the caller's numbering."""
import ctypes as C

import numpy as np
import pytest

from gpu_helpers import pa, ranks, upload, oracle_mul
import pa_amd._lib as L

pytestmark = pytest.mark.gpu


def _shuffled_fem(orc, nodes, parts, seed):
    """The Q1 FEM Laplacian of test/fem_example.jl's kind with its global node ids shuffled: what a mesher hands over."""
    P = int(np.prod(parts))
    n = int(np.prod(nodes))
    I, J, V, rows, cols = pa.laplacian_fem(nodes, (1,) * len(nodes), ranks(1))
    perm = np.random.default_rng(seed).permutation(n) + 1
    Ip, Jp, Vp = perm[I.items[0] - 1], perm[J.items[0] - 1], V.items[0]
    return n, P, Ip, Jp, Vp


@pytest.mark.parametrize("nodes,P", [((60, 40), 1), ((50, 30), 4), ((12, 10, 9), 3)])
def test_renumbered_matrix_gives_the_oracle_s_bits(orc, nodes, P):
    n, _, Ip, Jp, Vp = _shuffled_fem(orc, nodes, (1,) * len(nodes), 5)
    rows = pa.uniform_partition(ranks(P), (P,), (n,))
    orows = orc.uniform_partition((P,), (n,))
    # every part gets the triplets of the rows it owns (assembled psparse route)
    own = [np.nonzero((Ip >= o.local_to_global[0]) & (Ip <= o.local_to_global[o.n_own - 1]))[0] for o in orows]
    A = pa.psparse_from_coo(pa.DebugArray([Ip[k] for k in own]), pa.DebugArray([Jp[k] for k in own]), pa.DebugArray([Vp[k] for k in own]), rows)
    Ao = orc.psparse_from_coo([Ip[k] for k in own], [Jp[k] for k in own], [Vp[k] for k in own], orows)
    A2 = pa.renumber_for_locality(A, force=True)
    for (b0, b1), blk in zip(A2.bandwidths.items, A2.matrix_partition.items):
        assert b1 <= b0 and blk.own_own.nnz > 0
    if P == 1:
        assert A2.bandwidths.items[0][1] * 4 < A2.bandwidths.items[0][0]          # a shuffled mesh: the band collapses
    xo = [orc.hash_x(c.local_to_global) * (c.local_to_owner == c.part) for c in Ao.cols]
    yo = oracle_mul(orc, Ao, xo)
    xc = [v.copy() for v in xo]
    orc.consistent(xc, Ao.cols)
    for M in (A, A2):
        x = upload([v.copy() for v in xo], M.col_partition)
        y = pa.pzeros(M.row_partition)
        for f in (pa.mul_, pa.mul_c_):
            f(y, M, x)
            for got, e, r in zip(y.own_values().items, yo, Ao.rows):
                assert np.array_equal(got, e[:r.n_own])
            for got, e in zip(x.local_values().items, xc):
                assert np.array_equal(got, e)
        # alpha / beta, the transpose product, dot: everything that touches the layout
        y5 = [v.copy() for v in yo]
        orc.mul5(y5, Ao, [v.copy() for v in xo], 0.3, -1.5)
        pa.mul_c_(y, M, x, 0.3, -1.5)
        for got, e, r in zip(y.own_values().items, y5, Ao.rows):
            assert np.array_equal(got, e[:r.n_own])
        bt = [orc.hash_x(r.local_to_global + 1) for r in Ao.rows]
        ct = [orc.hash_x(c.local_to_global + 9) for c in Ao.cols]
        bdev = upload([v.copy() for v in bt], M.row_partition)
        cdev = upload([v.copy() for v in ct], M.col_partition)
        pa.mul5_transpose_(cdev, M, bdev, 0.75, -1.25)
        orc.mul5_transpose(ct, Ao, bt, 0.75, -1.25)
        for got, e in zip(cdev.local_values().items, ct):
            assert np.array_equal(got, e)
        d = pa.dot(x, x)
        dref = orc.dot(xc, xc, Ao.cols)
        assert abs(d - dref) <= 1e-13 * abs(dref)


def test_vectors_of_the_unrenumbered_partition_are_refused(orc):
    n, _, Ip, Jp, Vp = _shuffled_fem(orc, (40, 30), (1, 1), 7)
    rows = pa.uniform_partition(ranks(1), (1,), (n,))
    A = pa.psparse_from_coo(pa.DebugArray([Ip]), pa.DebugArray([Jp]), pa.DebugArray([Vp]), rows)
    A2 = pa.psparse_from_coo(pa.DebugArray([Ip]), pa.DebugArray([Jp]), pa.DebugArray([Vp]), rows, renumber=True)
    assert A2.bandwidths.items[0][1] * 2 <= A2.bandwidths.items[0][0]
    x_old, y_new = pa.pones(A.col_partition), pa.pzeros(A2.row_partition)
    with pytest.raises(L.PAError):
        pa.mul_(y_new, A2, x_old)                                   # x was laid out for the unrenumbered matrix
    with pytest.raises(L.PAError):
        pa.mul_c_(pa.pzeros(A.row_partition), A2, pa.pones(A2.col_partition))
    pa.mul_(y_new, A2, pa.pones(A2.col_partition))                  # its own partitions: fine
    y = pa.pzeros(A.row_partition)
    pa.mul_(y, A, x_old)
    assert np.array_equal(y.own_values().items[0], y_new.own_values().items[0])


def test_permuted_block_keeps_every_row_s_entry_order():
    """pa_csr_create_permuted against numpy: row i of the new block = row inv[i] of the old one with its columns renamed, entries
    in the OLD order (not sorted by new column)."""
    rng = np.random.default_rng(2)
    m = 3000
    lens = rng.integers(0, 25, m)
    rows = [np.sort(rng.choice(m, size=int(k), replace=False)) for k in lens]
    rp = (1 + np.concatenate(([0], np.cumsum(lens)))).astype(np.int32)
    H = pa.HostCSR(m, m, rp, (np.concatenate(rows) + 1).astype(np.int32), rng.standard_normal(int(lens.sum())))
    B = pa.DeviceCSR(H)
    pos = rng.permutation(m).astype(np.int32)
    h = C.c_void_p()
    L.call("pa_csr_create_permuted", B.h, L.ptr(pos), L.ptr(pos), C.byref(h))
    T = pa.DeviceCSR.from_handle(h, m, m, H.nnz)
    r, c = np.zeros(H.nnz, np.int32), np.zeros(H.nnz, np.int32)
    L.call("pa_csr_download_entries", T.h, L.ptr(r), L.ptr(c))
    inv = np.empty(m, np.int64)
    inv[pos] = np.arange(m)
    want_r = np.repeat(np.arange(m), lens[inv]).astype(np.int32)
    want_c = np.concatenate([pos[rows[i]] for i in inv]).astype(np.int32) if H.nnz else np.zeros(0, np.int32)
    assert np.array_equal(r, want_r) and np.array_equal(c, want_c)
    x = rng.standard_normal(m)
    xp = np.empty(m)
    xp[pos] = x
    y, yp = pa.DeviceVector(m, 0), pa.DeviceVector(m, 0)
    pa.spmv_(y, B, pa.DeviceVector(m, 0).upload(x))
    pa.spmv_(yp, T, pa.DeviceVector(m, 0).upload(xp))
    assert np.array_equal(yp.download()[pos], y.download())
    bad = pos.copy()
    bad[0] = bad[1]
    with pytest.raises(L.PAError):
        L.call("pa_csr_create_permuted", B.h, L.ptr(bad), None, C.byref(C.c_void_p()))


def test_locality_order_of_a_shuffled_grid_and_of_two_components():
    """The device's Cuthill-McKee on a 5-point grid numbered at random: a permutation, band ~ the grid's short side; two
    disconnected grids: both numbered."""
    import scipy.sparse as sp
    nx, ny = 90, 70
    n = nx * ny
    ids = np.arange(n).reshape(ny, nx)
    e = np.concatenate([np.stack([ids[:, :-1].ravel(), ids[:, 1:].ravel()]), np.stack([ids[:-1].ravel(), ids[1:].ravel()])], axis=1)
    G = sp.coo_matrix((np.ones(e.shape[1]), (e[0], e[1])), shape=(n, n))
    G = (G + G.T + sp.eye(n)).tocsr()
    for two in (False, True):
        M = sp.block_diag([G, G]).tocsr() if two else G
        N = M.shape[0]
        p = np.random.default_rng(4).permutation(N)
        Mp = M[p][:, p].tocsr()
        Mp.sort_indices()
        H = pa.HostCSR(N, N, (Mp.indptr + 1).astype(np.int32), (Mp.indices + 1).astype(np.int32), Mp.data.copy())
        B = pa.DeviceCSR(H)
        newpos = np.zeros(N, np.int32)
        b0, b1 = C.c_int64(), C.c_int64()
        L.call("pa_csr_locality_order", B.h, L.ptr(newpos), C.byref(b0), C.byref(b1))
        assert sorted(newpos.tolist()) == list(range(N))
        rr = np.repeat(np.arange(N), np.diff(Mp.indptr))
        assert b0.value == int(np.max(np.abs(rr - Mp.indices))) and b1.value == int(np.max(np.abs(newpos[rr] - newpos[Mp.indices])))
        assert b1.value <= 2 * min(nx, ny) + 2 and b0.value > 10 * b1.value, (b0.value, b1.value)
