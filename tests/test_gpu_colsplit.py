"""Column-split blocks (round 4): unstructured rows whose band is wider than the sliding x window are stored as a
chain of column pieces, run one after the other with the later ones accumulating.  spmv_csr!'s sum (src/sparse_utils.jl:649-669)
adds a row's products in ascending column; the pieces take consecutive runs of them: the same bits (np.array_equal) as the unsplit
block and as the oracle -- for every beta, after value updates, through the fused product + dot."""
import ctypes as C

import numpy as np
import pytest

from gpu_helpers import pa, env
import pa_amd._lib as L

pytestmark = pytest.mark.gpu


def _rows(rng, m, n, per_row, band, ragged=False):
    lens = rng.integers(0, 2 * per_row, m) if ragged else np.full(m, per_row)
    centre = (np.arange(m) * (n - 1) // max(m - 1, 1)).astype(np.int64)
    rows = []
    for r in range(m):
        c = np.unique(np.clip(centre[r] + rng.integers(-band, band + 1, int(lens[r])), 0, n - 1))
        rows.append(c)
    rp = (1 + np.concatenate(([0], np.cumsum([len(c) for c in rows])))).astype(np.int32)
    cv = (np.concatenate(rows) + 1).astype(np.int32) if rp[-1] > 1 else np.zeros(0, np.int32)
    return pa.HostCSR(m, n, rp, cv, rng.standard_normal(len(cv)))


def _chain(B):
    p, g = C.c_int32(), C.c_int64()
    L.call("pa_csr_chain_info", B.h, C.byref(p), C.byref(g))
    return p.value, g.value


def _split(B, pieces):
    h = C.c_void_p()
    L.call("pa_csr_create_colsplit", B.h, pieces, C.byref(h))
    return pa.DeviceCSR.from_handle(h, B.m, B.n, B.nnz)


@pytest.mark.parametrize("m,n,per_row,band,ragged,pieces", [(4000, 4000, 12, 900, False, 2), (3000, 5000, 9, 1500, True, 3),
                                                            (2500, 2000, 20, 2000, True, 5), (64, 64, 8, 64, False, 2)])
def test_forced_column_split_gives_the_bits_of_the_unsplit_block(orc, m, n, per_row, band, ragged, pieces):
    rng = np.random.default_rng(m + pieces)
    H = _rows(rng, m, n, per_row, band, ragged)
    B = pa.DeviceCSR(H)
    S = _split(B, pieces)
    assert S.info()["nnz"] == H.nnz and S.info()["n_rows"] == m
    r, c = np.zeros(max(H.nnz, 1), np.int32), np.zeros(max(H.nnz, 1), np.int32)
    L.call("pa_csr_download_entries", S.h, L.ptr(r), L.ptr(c))
    assert np.array_equal(r[:H.nnz], np.repeat(np.arange(m), np.diff(H.rowptr.astype(np.int64)))) and np.array_equal(c[:H.nnz], H.colval - 1)
    x, y0 = rng.standard_normal(n), rng.standard_normal(m)
    xd = pa.DeviceVector(n, 0).upload(x)
    oA = orc.CSR(m, n, H.rowptr, H.colval, H.nzval)
    for alpha, beta in ((1.0, 0.0), (1.0, 1.0), (-0.7, 2.5)):
        got = []
        for blk in (B, S):
            yd = pa.DeviceVector(m, 0).upload(y0)
            pa.spmv_(yd, blk, xd, L.SEG_OWN, L.SEG_OWN, alpha, beta)
            got.append(yd.download())
        want = orc.oracle_c().mul5_csr(y0.copy(), oA, x, alpha, beta)
        assert np.array_equal(got[0], want) and np.array_equal(got[1], want), (alpha, beta)
    # new values on the same pattern: the pieces gather theirs from the caller's order
    new = rng.standard_normal(H.nnz)
    B.update_values(new)
    S.update_values(new)
    ya, yb = pa.DeviceVector(m, 0), pa.DeviceVector(m, 0)
    pa.spmv_(ya, B, xd)
    pa.spmv_(yb, S, xd)
    assert np.array_equal(ya.download(), yb.download())
    w = pa.DeviceVector(H.nnz + 5, 0).upload(np.concatenate([np.zeros(5), new * 2.0]))
    L.call("pa_csr_update_values_from", S.h, w.h, 5)
    L.call("pa_csr_update_values_from", B.h, w.h, 5)
    pa.spmv_(ya, B, xd)
    pa.spmv_(yb, S, xd)
    assert np.array_equal(ya.download(), yb.download())


def test_a_band_beyond_the_sliding_window_is_split_by_the_library(orc):
    """2 M rows x 16 entries within +-15000: no window holds the span; the library cuts the block into column pieces that run on the
    ring kernel.  Same bits as the unsplit block on the plain row split (PA_SPMV_COLSPLIT=0) and as the oracle."""
    rng = np.random.default_rng(33)
    m = 2_000_000
    col = np.repeat(np.arange(m, dtype=np.int64), 16).reshape(m, 16) + rng.integers(-15000, 15000, size=(m, 16))
    col = np.sort(np.clip(col, 0, m - 1), axis=1)
    keep = np.concatenate([np.ones((m, 1), bool), col[:, 1:] != col[:, :-1]], axis=1)
    lens = keep.sum(1)
    H = pa.HostCSR(m, m, (1 + np.concatenate(([0], np.cumsum(lens)))).astype(np.int32), (col[keep] + 1).astype(np.int32),
                   rng.standard_normal(int(lens.sum())))
    with env(PA_SPMV_COLSPLIT="0"):
        B0 = pa.DeviceCSR(H)
    B = pa.DeviceCSR(H)
    assert B0.xwin()["ring_groups"] == 0 and B.xwin()["ring_groups"] > 0, (B0.xwin(), B.xwin())
    pieces, groups = _chain(B)
    assert pieces >= 2 and groups > 0, (pieces, groups)        # (round 5: the chain runs as ONE launch, y written once)
    x = rng.standard_normal(m)
    xd = pa.DeviceVector(m, 0).upload(x)
    y, y0, y1 = pa.DeviceVector(m, 0), pa.DeviceVector(m, 0), pa.DeviceVector(m, 0)
    pa.spmv_(y, B, xd)
    pa.spmv_(y0, B0, xd)
    with env(PA_SPMV_CHAIN_FUSED="0"):                          # the same pieces, a launch each
        assert _chain(B)[1] == 0
        pa.spmv_(y1, B, xd)
    want = np.zeros(m)
    orc.oracle_c().spmv_csr(want, x, orc.CSR(m, m, H.rowptr, H.colval, H.nzval))
    assert np.array_equal(y.download(), want) and np.array_equal(y0.download(), want) and np.array_equal(y1.download(), want)


@pytest.mark.parametrize("m,per_row,band,ragged,pieces", [(200_000, 12, 1500, False, 3), (150_000, 10, 2500, True, 2),
                                                           (120_000, 16, 900, True, 5)])
def test_a_chain_cut_at_shared_rows_runs_as_one_launch_with_the_same_bits(orc, m, per_row, band, ragged, pieces):
    """Round 5.  The pieces' chunks and ring groups end at the same rows, so one workgroup takes its rows through every
    piece and the partial sums never leave L2 (k_spmv_xring_chain).  Same additions in the same order as a launch per piece: the bits
    of the unsplit block and of the oracle, for every (alpha, beta), after value updates, and with the rows above the band's first
    column empty in the lowest piece."""
    rng = np.random.default_rng(m + pieces)
    lens = rng.integers(0, 2 * per_row, m) if ragged else np.full(m, per_row)
    rows = np.repeat(np.arange(m, dtype=np.int64), lens)
    cols = np.clip(rows + rng.integers(-band, band + 1, rows.size), 0, m - 1)
    key = np.unique(rows * m + cols)
    rows, cols = key // m, key % m
    rp = (1 + np.concatenate(([0], np.cumsum(np.bincount(rows, minlength=m))))).astype(np.int32)
    H = pa.HostCSR(m, m, rp, (cols + 1).astype(np.int32), rng.standard_normal(rows.size))
    with env(PA_SPMV_COLSPLIT="0"):
        B = pa.DeviceCSR(H)
    S = _split(B, pieces)
    got_pieces, groups = _chain(S)
    assert got_pieces == pieces and groups > 1, (got_pieces, groups)
    x, y0 = rng.standard_normal(m), rng.standard_normal(m)
    xd = pa.DeviceVector(m, 0).upload(x)
    oA = orc.CSR(m, m, H.rowptr, H.colval, H.nzval)
    for alpha, beta in ((1.0, 0.0), (1.0, 1.0), (-0.7, 2.5)):
        want = orc.oracle_c().mul5_csr(y0.copy(), oA, x, alpha, beta)
        for fused in ("1", "0"):
            with env(PA_SPMV_CHAIN_FUSED=fused):
                yd = pa.DeviceVector(m, 0).upload(y0)
                pa.spmv_(yd, S, xd, L.SEG_OWN, L.SEG_OWN, alpha, beta)
                assert np.array_equal(yd.download(), want), (alpha, beta, fused)
    new = rng.standard_normal(H.nnz)
    B.update_values(new)
    S.update_values(new)
    ya, yb = pa.DeviceVector(m, 0), pa.DeviceVector(m, 0)
    pa.spmv_(ya, B, xd)
    pa.spmv_(yb, S, xd)
    assert np.array_equal(ya.download(), yb.download())


def test_fused_product_and_dot_on_a_column_split_chain(orc):
    """mul! + dot in one pass (the CG loop's fused form) over a chain of column pieces: c keeps the bits of the unsplit block, the dot --
    a sum of per-chunk partials, grouped by piece here -- agrees to rounding (its parity bar: 1e-13 relative)."""
    import pa_amd.p_sparse_matrix as psm
    rng = np.random.default_rng(9)
    m = 6000
    H = _rows(rng, m, m, 14, 1200, ragged=True)
    B = pa.DeviceCSR(H)
    S = _split(B, 3)
    ind = pa.uniform_partition(pa.DebugArray([1]), m)
    uh = rng.standard_normal(m)
    got = []
    for blk in (B, S):
        empty = pa.DeviceCSR(pa.HostCSR(m, 0, np.ones(m + 1, np.int32), np.zeros(0, np.int32), np.zeros(0)))
        A = pa.PSparseMatrix(pa.DebugArray([psm.SplitMatrixBlocks(blk, empty)]), ind, ind, True)
        u = pa.pvector_from_function(lambda i: uh, ind)
        c = pa.pzeros(ind)
        assert psm.mul_dot_(c, A, u, 5)
        got.append((c.own_values().items[0].copy(), pa.read_slots(5)[0]))
    assert np.array_equal(got[0][0], got[1][0])
    want = float(uh @ got[0][0])
    assert abs(got[0][1] - want) <= 1e-12 * max(1.0, abs(want)) and abs(got[1][1] - want) <= 1e-12 * max(1.0, abs(want))


def test_transposed_and_renumbered_twins_of_a_column_split_chain(orc):
    """The device-side transpose (pa_csr_create_transpose), the renumbered twin (pa_csr_create_permuted) and the bandwidth-reducing
    order (pa_csr_locality_order) read a chain of column pieces in the caller's entry order: the same blocks, hence the same bits,
    as from the unsplit block."""
    rng = np.random.default_rng(12)
    m = 5000
    H = _rows(rng, m, m, 10, 800, ragged=True)
    B = pa.DeviceCSR(H)
    S = _split(B, 4)
    x = rng.standard_normal(m)
    xd = pa.DeviceVector(m, 0).upload(x)
    perm = rng.permutation(m).astype(np.int32)
    outs = []
    for blk in (B, S):
        t, q = C.c_void_p(), C.c_void_p()
        L.call("pa_csr_create_transpose", blk.h, C.byref(t))
        L.call("pa_csr_create_permuted", blk.h, L.ptr(perm), L.ptr(perm), C.byref(q))
        T, Q = pa.DeviceCSR.from_handle(t, m, m, H.nnz), pa.DeviceCSR.from_handle(q, m, m, H.nnz)
        yt, yq = pa.DeviceVector(m, 0), pa.DeviceVector(m, 0)
        pa.spmv_(yt, T, xd)
        pa.spmv_(yq, Q, xd)
        order = np.zeros(m, np.int32)
        L.call("pa_csr_locality_order", blk.h, L.ptr(order), None, None)
        outs.append((yt.download(), yq.download(), order))
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
    # A' x by the reference's scatter loop (spmtv_csr!, src/sparse_utils.jl:613-647): rows ascending, y[col] += a * x[row]
    want = np.zeros(m)
    rp = H.rowptr.astype(np.int64) - 1
    for r in range(m):
        for p in range(rp[r], rp[r + 1]):
            want[H.colval[p] - 1] += H.nzval[p] * x[r]
    assert np.array_equal(outs[0][0], want)


def test_rows_whose_columns_do_not_ascend_keep_their_order_of_addition():
    """A twin with renamed columns (own x ghost reading the receive buffer, a renumbered block) stores a row's entries in the caller's
    order with columns anywhere.  Cut into column pieces, an entry follows the highest piece any entry before it in its row went to,
    so the pieces still take consecutive runs of the row: the bits of the uncut twin (found by the fuzzers with PA_SPMV_COLSPLIT=3)."""
    rng = np.random.default_rng(21)
    m = 7000
    H = _rows(rng, m, m, 12, 3000, ragged=True)
    B = pa.DeviceCSR(H)
    cperm = rng.permutation(m).astype(np.int32)
    q = C.c_void_p()
    L.call("pa_csr_create_permuted", B.h, None, L.ptr(cperm), C.byref(q))
    Q = pa.DeviceCSR.from_handle(q, m, m, H.nnz)
    x = rng.standard_normal(m)
    xd = pa.DeviceVector(m, 0).upload(x)
    xq = np.zeros(m); xq[cperm] = x                                    # x_new[col_pos[j]] = x[j]
    xqd = pa.DeviceVector(m, 0).upload(xq)
    y0, y1, y2 = pa.DeviceVector(m, 0), pa.DeviceVector(m, 0), pa.DeviceVector(m, 0)
    pa.spmv_(y0, B, xd)
    pa.spmv_(y1, Q, xqd)
    assert np.array_equal(y0.download(), y1.download())
    for pieces in (2, 3, 5):
        S = _split(Q, pieces)
        pa.spmv_(y2, S, xqd, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0)
        assert np.array_equal(y2.download(), y0.download()), pieces
