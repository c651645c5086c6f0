"""The lossless value dictionary after round 4: built on the device, ON BY DEFAULT for blocks of >= 2^18 stored
entries with at most 64 distinct values, renewed after value updates, abandoned when new values overflow it -- and always the bits of
the fp64 stream (np.array_equal).  Kernels: k_spmv_rowsplit<..., VD> of csrc/pa_spmv_kernel.h; reference loop: spmv_csr!
src/sparse_utils.jl:649-669; value updates: psparse! src/p_sparse_matrix.jl:1291-1305."""
import numpy as np
import pytest

from gpu_helpers import pa, ranks, env
import pa_amd._lib as L

pytestmark = pytest.mark.gpu


def _band(rng, m, per_row, values):
    col = np.repeat(np.arange(m, dtype=np.int64), per_row).reshape(m, per_row) + np.arange(per_row) * 37 - (per_row // 2) * 37
    col = np.clip(col, 0, m - 1)
    col.sort(axis=1)
    rows = [np.unique(c) for c in col]
    rp = (1 + np.concatenate(([0], np.cumsum([len(r) for r in rows])))).astype(np.int32)
    cv = (np.concatenate(rows) + 1).astype(np.int32)
    return pa.HostCSR(m, m, rp, cv, rng.choice(values, size=len(cv)))


def test_default_is_auto_for_big_blocks_and_the_bits_are_the_fp64_stream_s(orc):
    """64^3 x 27 points = 6.9 M stored entries, 2 distinct values: a dictionary without any switch; a 16^3 block (below 2^18
    entries) none; PA_SPMV_VALUE_DICT=0 none.  Products with and without it: the same bits as the oracle's."""
    with env(PA_SPMV_VALUE_DICT="0"):
        A0, _ = pa.build_p_matrix(ranks(1), 64, 64, 64, 64, 64, 64, 1, 1, 1)
    A1, _ = pa.build_p_matrix(ranks(1), 64, 64, 64, 64, 64, 64, 1, 1, 1)
    As, _ = pa.build_p_matrix(ranks(1), 16, 16, 16, 16, 16, 16, 1, 1, 1)
    assert A0.matrix_partition.items[0].own_own.value_dict() == 0
    assert A1.matrix_partition.items[0].own_own.value_dict() == 2
    assert As.matrix_partition.items[0].own_own.value_dict() == 0
    x = pa.pvector_from_function(lambda i: orc.hash_x(i.get_local_to_global()), A1.col_partition)
    ys = []
    for A in (A0, A1):
        y = pa.pvector_from_function(lambda i: np.cos(i.get_local_to_global().astype(float)), A.row_partition)
        pa.mul5_(y, A, x, -0.5, 1.25)
        ys.append(y.own_values().items[0].copy())
    assert np.array_equal(ys[0], ys[1])
    Ao, _, _ = orc.hpcg_build_p_matrix(64, 64, 64, 1, 1, 1)
    yo = [np.cos(r.local_to_global.astype(float)) for r in Ao.rows]
    orc.mul5(yo, Ao, [orc.hash_x(c.local_to_global) for c in Ao.cols], -0.5, 1.25)
    assert np.array_equal(ys[1], yo[0])


def test_updates_renew_the_codes_after_eight_products_and_an_overflow_ends_them():
    rng = np.random.default_rng(8)
    vals = np.array([0.5, -1.25, 3.0, 1e-3, -7.0])
    H = _band(rng, 60000, 9, vals)                                  # ~540 k stored entries: above the automatic threshold
    with env(PA_SPMV_XWIN="0"):                                      # (blocks on the x-window launches get no automatic dictionary)
        B = pa.DeviceCSR(H)
        B0 = None
        with env(PA_SPMV_VALUE_DICT="0"):
            B0 = pa.DeviceCSR(H)
    assert B.value_dict() == len(vals) and B0.value_dict() == 0
    x = pa.DeviceVector(H.n, 0).upload(rng.standard_normal(H.n))
    y, y0 = pa.DeviceVector(H.m, 0), pa.DeviceVector(H.m, 0)

    def same():
        pa.spmv_(y, B, x)
        pa.spmv_(y0, B0, x)
        return np.array_equal(y.download(), y0.download())
    assert same()
    # new values from another small set: fp64 stream at once (stale codes), a dictionary again after 8 products
    new = rng.choice(np.array([2.0, -0.125, 9.5]), size=H.nnz)
    B.update_values(new)
    B0.update_values(new)
    assert B.value_dict() == 0
    for k in range(9):
        assert same(), k
    assert B.value_dict() == 3
    assert same()
    # values that overflow 64 distinct patterns: the dictionary ends, for good
    many = rng.standard_normal(H.nnz)
    B.update_values(many)
    B0.update_values(many)
    for k in range(12):
        assert same(), k
    assert B.value_dict() == 0
    B.update_values(new)
    B0.update_values(new)
    for k in range(12):
        assert same(), k
    assert B.value_dict() == 0


def test_x_window_blocks_get_no_automatic_dictionary_but_an_explicit_one_works():
    rng = np.random.default_rng(9)
    m = 200000
    col = np.repeat(np.arange(m, dtype=np.int64), 16).reshape(m, 16) + rng.integers(-1500, 1500, size=(m, 16))
    col = np.clip(col, 0, m - 1)
    rows = [np.unique(c) for c in col]                             # no row pattern: random columns inside a band
    rp = (1 + np.concatenate(([0], np.cumsum([len(r) for r in rows])))).astype(np.int32)
    cv = (np.concatenate(rows) + 1).astype(np.int32)
    H = pa.HostCSR(m, m, rp, cv, rng.choice(np.array([1.0, -1.0]), size=len(cv)))
    B = pa.DeviceCSR(H)
    if B.xwin()["groups"] == 0:
        pytest.skip("this band does not take the x-window launches")
    assert B.value_dict() == 0
    with env(PA_SPMV_VALUE_DICT="1"):
        B1 = pa.DeviceCSR(H)
    assert B1.value_dict() == 2
    x = pa.DeviceVector(H.n, 0).upload(rng.standard_normal(H.n))
    y, y1 = pa.DeviceVector(H.m, 0), pa.DeviceVector(H.m, 0)
    pa.spmv_(y, B, x)
    pa.spmv_(y1, B1, x)
    assert np.array_equal(y.download(), y1.download())


def test_two_value_dictionaries_are_decoded_by_a_select_with_the_same_bits():
    """Round 5: a dictionary of at most two values (HPCG's 26 / -1) is decoded by a select per entry (VD = 2 of k_spmv_rowsplit) instead
    of through the lane dictionary's ds_bpermute pairs (PA_SPMV_VDICT_SELECT=0) -- the LDS pipe was the busy unit of that kernel.
    Same bits on row patterns (27-point), 16-bit windows and 32-bit columns, with alpha and beta; a hipGraph recorded through the
    select follows value updates that stay within two values and REFUSES (PA_ERR_STATE: record again) a third value instead of
    replaying a decode that cannot represent it."""
    A, _ = pa.build_p_matrix(ranks(1), 64, 64, 64, 64, 64, 64, 1, 1, 1)
    blk = A.matrix_partition.items[0].own_own
    assert blk.value_dict() == 2
    rng = np.random.default_rng(17)
    H = _band(rng, 60000, 9, np.array([0.5, -1.25]))
    with env(PA_SPMV_XWIN="0"):
        B = pa.DeviceCSR(H)
        with env(PA_SPMV_COL16="0", PA_SPMV_PATTERN="0"):
            B32 = pa.DeviceCSR(H)
        with env(PA_SPMV_VALUE_DICT="0"):
            Bf = pa.DeviceCSR(H)
    assert B.value_dict() == 2 and B32.value_dict() == 2 and Bf.value_dict() == 0
    for M, Mref in ((blk, None), (B, Bf), (B32, Bf)):
        x = pa.DeviceVector(M.n, 0).upload(rng.standard_normal(M.n))
        y0 = rng.standard_normal(M.m)
        got = []
        for sel in ("1", "0"):
            with env(PA_SPMV_VDICT_SELECT=sel):
                y = pa.DeviceVector(M.m, 0).upload(y0)
                pa.spmv_(y, M, x, L.SEG_OWN, L.SEG_OWN, -0.75, 1.5)
                got.append(y.download())
        assert np.array_equal(got[0], got[1])
        if Mref is not None:
            y = pa.DeviceVector(M.m, 0).upload(y0)
            pa.spmv_(y, Mref, x, L.SEG_OWN, L.SEG_OWN, -0.75, 1.5)
            assert np.array_equal(got[0], y.download())
    # recorded through the select: two new values are followed at once, a third one is refused
    x = pa.DeviceVector(H.n, 0).upload(rng.standard_normal(H.n))
    y, yf = pa.DeviceVector(H.m, 0), pa.DeviceVector(H.m, 0)
    pa.spmv_(y, B, x)
    with pa.Graph() as g:
        pa.spmv_(y, B, x)
    new2 = rng.choice(np.array([3.0, -8.5]), size=H.nnz)
    B.update_values(new2)
    Bf.update_values(new2)
    g.launch()
    pa.spmv_(yf, Bf, x)
    assert B.value_dict() == 2 and np.array_equal(y.download(), yf.download())
    new3 = rng.choice(np.array([3.0, -8.5, 0.25]), size=H.nnz)
    with pytest.raises(L.PAError, match="record the graph again"):
        B.update_values(new3)
