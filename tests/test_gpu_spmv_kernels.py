"""spmv! on one block (src/sparse_utils.jl:617-690) through every column encoding and launch of the row-split kernel.
Bars: np.array_equal for everything but dot / norm (1e-13).  Needs a real MI355X (-m gpu)."""
import pytest

from gpu_helpers import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", ["empty_rows", "ragged", "long_rows", "one_row", "all_empty", "wide"])
def test_spmv_irregular_bit_exact(orc, case):
    rng = np.random.default_rng(42)
    if case == "empty_rows":
        m, n = 5000, 300
        row_len = rng.integers(0, 4, m) * (rng.random(m) < 0.2)       # most rows empty -> compacted path
    elif case == "ragged":
        m, n = 3000, 4000
        row_len = rng.integers(0, 60, m)
    elif case == "long_rows":
        m, n = 40, 9000
        row_len = rng.integers(0, 50, m)
        row_len[[3, 17, 39]] = [2049, 5000, 8999]                      # longer than one 2048-entry chunk
    elif case == "one_row":
        m, n, row_len = 1, 10, np.array([7])
    elif case == "all_empty":
        m, n, row_len = 100, 10, np.zeros(100, int)
    else:
        m, n = 700, 100000
        row_len = rng.integers(1, 300, m)
    A = _random_csr(rng, m, n, row_len.astype(int))
    dA = pa.DeviceCSR(A)
    x = pa.DeviceVector(n, 0).upload(rng.standard_normal(n))
    oA = orc.CSR(A.m, A.n, A.rowptr, A.colval, A.nzval)
    for alpha, beta in [(1.0, 0.0), (1.0, 1.0), (0.5, -2.0)]:
        y0 = rng.standard_normal(m)
        y = pa.DeviceVector(m, 0).upload(y0)
        pa.spmv_(y, dA, x, alpha=alpha, beta=beta)
        exp = orc.oracle_c().mul5_csr(y0.copy(), oA, x.download(), alpha, beta)
        assert np.array_equal(y.download(), exp), (case, alpha, beta)
    # 3-arg spmv! == spmv_csr! loop
    y = pa.DeviceVector(m, 0).upload(rng.standard_normal(m))
    pa.spmv_(y, dA, x)
    assert np.array_equal(y.download(), orc.oracle_c().spmv_csr(np.zeros(m), x.download(), oA))


def test_csc_upload_gives_same_bits(orc):
    rng = np.random.default_rng(1)
    A = _random_csr(rng, 400, 300, rng.integers(0, 30, 400))
    oA = orc.CSR(A.m, A.n, A.rowptr, A.colval, A.nzval)
    colptr, rowval, nzval = orc.csr_to_csc(oA)
    import pa_amd._lib as L
    h = C.c_void_p()
    colptr, rowval = np.ascontiguousarray(colptr, np.int64), np.ascontiguousarray(rowval, np.int64)
    L.call("pa_csr_create_from_csc", pa.context().h, A.m, A.n, A.nnz, L.ptr(colptr), L.ptr(rowval), 8, 1,
           L.ptr(np.ascontiguousarray(nzval)), C.byref(h))
    x = pa.DeviceVector(A.n, 0).upload(rng.standard_normal(A.n))
    y1, y2 = pa.DeviceVector(A.m, 0), pa.DeviceVector(A.m, 0)
    pa.spmv_(y1, pa.DeviceCSR(A), x)
    L.call("pa_spmv", h, x.h, 0, y2.h, 0, 1.0, 0.0)
    assert np.array_equal(y1.download(), y2.download())
    L.call("pa_csr_destroy", h)


def test_value_dictionary_mode_is_lossless(monkeypatch, orc):
    """PA_SPMV_VALUE_DICT=1: blocks with at most 64 distinct stored values stream one byte per entry instead of eight.
    Same bits as the fp64 stream on the 27-point operator (2 values; 2 parts, mul! and the multicolour MG-PCG), on a Q1
    FEM matrix; a matrix with more distinct values keeps the fp64 stream; updating the values drops the dictionary."""
    def hpcg(P, shape):
        return pa.build_p_matrix(ranks(P), 16, 12, 10, 16 * shape[0], 12 * shape[1], 10 * shape[2], *shape, keep_host=True)
    A0, b0 = hpcg(2, (2, 1, 1))
    monkeypatch.setenv("PA_SPMV_VALUE_DICT", "1")
    A1, b1 = hpcg(2, (2, 1, 1))
    assert [bk.own_own.value_dict() for bk in A1.matrix_partition.items] == [2, 2]
    assert [bk.own_own.value_dict() for bk in A0.matrix_partition.items] == [0, 0]
    xf = lambda i: orc.hash_x(i.get_local_to_global()) * (i.get_local_to_owner() == i.part)
    ys = []
    for A in (A0, A1):
        x = pa.pvector_from_function(xf, A.col_partition)
        y = pa.pvector_from_function(lambda i: np.cos(i.get_local_to_global().astype(float)), A.row_partition)
        pa.mul5_(y, A, x, -0.5, 1.25)
        ys.append([v.copy() for v in y.own_values().items])
    for u, v in zip(*ys):
        assert np.array_equal(u, v)
    # the multicolour smoother's colour blocks and the fused restriction go through the same kernels
    S1 = pa.pc_setup(ranks(2), 2, 3, 16, 8, 8, ordering="multicolor_spmv")
    monkeypatch.delenv("PA_SPMV_VALUE_DICT")
    S0 = pa.pc_setup(ranks(2), 2, 3, 16, 8, 8, ordering="multicolor_spmv")
    res = []
    for S in (S0, S1):
        A, b = S.A_vec[-1], S.r[-1]
        h = []
        # (fuse=False: with a dictionary the dot is its own pass, so only the unfused loops share every bit)
        x, r0, r, it = pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=8, Pl=S, history=h, fuse=False)
        res.append((r0, h, [v.copy() for v in x.own_values().items]))
    assert res[0][:2] == res[1][:2] and all(np.array_equal(u, v) for u, v in zip(res[0][2], res[1][2]))
    monkeypatch.setenv("PA_SPMV_VALUE_DICT", "1")
    # FEM: a handful of distinct values; random values: too many -> fp64 stream
    I, J, V, rows, cols = pa.laplacian_fem((40, 30), (1, 1), ranks(1))
    F = pa.psparse_disassembled(I, J, V, rows, cols, keep_host=True)
    assert 2 <= F.matrix_partition.items[0].own_own.value_dict() <= 64
    rng = np.random.default_rng(5)
    R = pa.DeviceCSR(_random_csr(rng, 200, 300, rng.integers(1, 30, 200)))
    assert R.value_dict() == 0
    monkeypatch.delenv("PA_SPMV_VALUE_DICT")
    F0 = pa.psparse_disassembled(I, J, V, rows, cols, keep_host=True)
    xF = pa.pvector_from_function(xf, F.col_partition)
    yF, yF0 = pa.pzeros(F.row_partition), pa.pzeros(F0.row_partition)
    pa.mul_(yF, F, xF)
    pa.mul_(yF0, F0, xF)
    assert np.array_equal(yF.own_values().items[0], yF0.own_values().items[0])
    blk = A1.matrix_partition.items[0].own_own
    blk.update_values(np.sin(np.arange(blk.nnz, dtype=np.float64)))
    assert blk.value_dict() == 0
    A0.matrix_partition.items[0].own_own.update_values(np.sin(np.arange(blk.nnz, dtype=np.float64)))
    x = pa.pvector_from_function(xf, A1.col_partition)
    y0, y1 = pa.pzeros(A0.row_partition), pa.pzeros(A1.row_partition)
    pa.mul_(y0, A0, x)
    pa.mul_(y1, A1, x)
    assert all(np.array_equal(u, v) for u, v in zip(y0.own_values().items, y1.own_values().items))


def test_index_widths_and_bases_give_the_same_block(orc):
    """pa_csr_create / pa_csr_create_mixed accept the reference's index types as stored: Int32 or Int64, 1-based (Julia)
    or 0-based, and the mixed form (Int64 row pointers, Int32 columns).  Same device block, same product bits."""
    import pa_amd._lib as L
    rng = np.random.default_rng(11)
    A = _random_csr(rng, 300, 500, rng.integers(0, 40, 300))
    x = pa.DeviceVector(500, 0).upload(rng.standard_normal(500))
    outs = []
    for rb, cb, base in ((4, 4, 1), (8, 8, 1), (8, 4, 1), (4, 4, 0), (8, 4, 0), (4, 8, 0)):
        rp = np.ascontiguousarray(A.rowptr.astype(np.int64) - 1 + base, np.int32 if rb == 4 else np.int64)
        cv = np.ascontiguousarray(A.colval.astype(np.int64) - 1 + base, np.int32 if cb == 4 else np.int64)
        h = C.c_void_p()
        L.call("pa_csr_create_mixed", pa.context().h, A.m, A.n, A.nnz, L.ptr(rp), rb, L.ptr(cv), cb, base, L.ptr(A.nzval), C.byref(h))
        y = pa.DeviceVector(300, 0)
        L.call("pa_spmv", h, x.h, L.SEG_OWN, y.h, L.SEG_OWN, 1.0, 0.0)
        outs.append(y.own())
        L.call("pa_csr_destroy", h)
    want = np.zeros(300)
    orc.oracle_c().spmv_csr(want, x.own(), orc.CSR(A.m, A.n, A.rowptr, A.colval, A.nzval))
    for o in outs:
        assert np.array_equal(o, want)
    with pytest.raises(L.PAError):                       # 2^31 entries or more need 64-bit row pointers
        L.call("pa_csr_create_mixed", pa.context().h, 10, 10, 2 ** 31, L.ptr(np.zeros(11, np.int32)), 4,
               L.ptr(np.zeros(1, np.int32)), 4, 0, L.ptr(np.zeros(1)), C.byref(C.c_void_p()))


def test_argument_errors_are_reported():
    A = pa.DeviceCSR(pa.compresscoo([1, 2], [1, 2], [1.0, 1.0], 2, 2))
    x, y = pa.DeviceVector(3, 0), pa.DeviceVector(2, 0)
    with pytest.raises(pa.PAError, match="size"):        # @boundscheck of spmv! (src/sparse_utils.jl:618-621)
        pa.spmv_(y, A, x)


def test_block_with_more_than_2_to_31_entries_and_forced_slabs(monkeypatch, orc):
    """Device offsets are Int32; a block of 2^31 stored entries or more is kept as consecutive row slabs (Int64 row
    pointers at the boundary).  (1) forced on a small matrix (PA_CSR_MAX_SLAB_NNZ = 1000, less than a chunk): ~50 slabs give the bits of one;
    values can be updated through the slabs.  (2) for real: one part of 432^3 rows, 2 166 720 184 entries > 2^31, built
    by the native generator with Int64 row pointers: closed-form size, A*1 == b bit-exactly, patterns on both slabs."""
    A1, b1 = pa.build_p_matrix(ranks(1), 20, 12, 9, 20, 12, 9, 1, 1, 1, keep_host=True)
    monkeypatch.setenv("PA_CSR_MAX_SLAB_NNZ", "1000")
    A9, _ = pa.build_p_matrix(ranks(1), 20, 12, 9, 20, 12, 9, 1, 1, 1, keep_host=True)
    monkeypatch.delenv("PA_CSR_MAX_SLAB_NNZ")
    i1, i9 = A1.matrix_partition.items[0].own_own.info(), A9.matrix_partition.items[0].own_own.info()
    assert (i1["n_rows"], i1["nnz"]) == (i9["n_rows"], i9["nnz"]) and i9["n_chunks"] > i1["n_chunks"]
    x = pa.pvector_from_function(lambda i: orc.hash_x(i.get_local_to_global()), A1.col_partition)
    y1, y9 = pa.pzeros(A1.row_partition), pa.pzeros(A9.row_partition)
    pa.mul5_(y1, A1, x, -1.5, 0.0)
    pa.mul5_(y9, A9, x, -1.5, 0.0)
    assert np.array_equal(y1.own_values().items[0], y9.own_values().items[0])
    new_vals = np.cos(np.arange(i1["nnz"], dtype=np.float64))
    for A in (A1, A9):
        A.matrix_partition.items[0].own_own.update_values(new_vals)
    pa.mul_(y1, A1, x)
    pa.mul_(y9, A9, x)
    assert np.array_equal(y1.own_values().items[0], y9.own_values().items[0]) and np.any(y1.own_values().items[0] != 0)
    del A1, A9, y1, y9, x
    n = 432
    A, b = pa.build_p_matrix(ranks(1), n, n, n, n, n, n, 1, 1, 1)
    blk = A.matrix_partition.items[0].own_own
    info, enc = blk.info(), blk.encoding()
    assert info["nnz"] == (3 * n - 2) ** 3 > 2 ** 31 and info["n_rows"] == n ** 3
    assert enc["pattern"] >= 0.999 * info["n_chunks"]
    y = pa.pzeros(A.row_partition)
    pa.mul_(y, A, pa.pones(A.col_partition))
    assert np.array_equal(y.own_values().items[0], b.own_values().items[0])


def test_column_encodings_agree_bit_for_bit(orc, monkeypatch):
    """The three per-chunk column encodings (row patterns / 16-bit windows / 32-bit) give the same bits, on a matrix
    that mixes them: banded structured rows (patterns), rows longer than 32 (no pattern), >4 pattern runs in a chunk,
    columns scattered over 400k (no 16-bit windows).  PA_SPMV_PATTERN / PA_SPMV_COL16 switch the encodings off."""
    rng = np.random.default_rng(7)
    n = 400000
    I, J = [], []
    for r in range(1, 3001):                              # structured band: 5 deltas, boundary rows cut
        for dlt in (-700, -1, 0, 1, 700):
            if 1 <= r + dlt <= n:
                I.append(r); J.append(r + dlt)
    for r in range(3001, 3400):                           # alternating short patterns: many runs per chunk
        for dlt in ((0, 3) if r % 2 else (0, 5, 9)):
            I.append(r); J.append(r + dlt)
    for r in range(3400, 3500):                           # rows of 40 entries (longer than a pattern may be)
        for dlt in range(40):
            I.append(r); J.append(r + 2 * dlt)
    for r in range(3500, 4000):                           # scattered columns
        for c in rng.choice(n, size=rng.integers(20, 60), replace=False):
            I.append(r); J.append(int(c) + 1)
    V = rng.standard_normal(len(I))
    A = pa.compresscoo(I, J, V, n, n)
    oA = orc.CSR(A.m, A.n, A.rowptr, A.colval, A.nzval)
    xh = rng.standard_normal(n)
    exp = orc.oracle_c().mul5_csr(np.full(n, 0.5), oA, xh, -1.5, 2.0)
    x = pa.DeviceVector(n, 0).upload(xh)
    seen = set()
    for pat, c16 in (("1", "1"), ("0", "1"), ("1", "0"), ("0", "0")):
        monkeypatch.setenv("PA_SPMV_PATTERN", pat)
        monkeypatch.setenv("PA_SPMV_COL16", c16)
        dA = pa.DeviceCSR(A)
        enc = dA.encoding()
        seen.add((enc["pattern"] > 0, enc["c16"] > 0, enc["c32"] > 0))
        y = pa.DeviceVector(n, 0).upload(np.full(n, 0.5))
        pa.spmv_(y, dA, x, alpha=-1.5, beta=2.0)
        assert np.array_equal(y.download(), exp), (pat, c16, enc)
    assert (False, False, True) in seen                   # everything 32-bit when both are off
    # the default build of an HPCG matrix with long x-lines is (almost) all patterns; short lines put more than 4
    # pattern runs in a chunk and use the 16-bit stream instead
    monkeypatch.delenv("PA_SPMV_PATTERN"); monkeypatch.delenv("PA_SPMV_COL16")
    A27, b27 = pa.build_p_matrix(ranks(1), 128, 6, 5, 128, 6, 5, 1, 1, 1)
    e = A27.matrix_partition.items[0].own_own.encoding()
    assert e["pattern"] > 0 and e["c32"] == 0
    y = pa.pzeros(A27.row_partition)
    pa.mul_(y, A27, pa.pones(A27.col_partition))
    assert np.array_equal(y.own_values().items[0], b27.own_values().items[0])
    A16, _ = pa.build_p_matrix(ranks(1), 16, 16, 16, 16, 16, 16, 1, 1, 1)
    e = A16.matrix_partition.items[0].own_own.encoding()
    assert e["pattern"] == 0 and e["c16"] > 0 and e["c32"] == 0


def test_compacted_column_streams_in_a_mixed_block(orc, monkeypatch):
    """A block whose chunks are mostly described by row patterns keeps columns ONLY for the other chunks (compacted
    16-bit and 32-bit streams addressed through the chunk's descriptor slot): banded rows (patterns) + rows with
    random columns near the diagonal (16-bit windows) + scattered rows (32-bit) + one row longer than a chunk, in ONE
    block.  Same bits as the oracle and as the full-length streams; ~8 bytes of HBM per stored entry instead of 14."""
    rng = np.random.default_rng(11)
    n = 60000
    rows = {r: [r + d for d in (-300, -1, 0, 1, 300) if 1 <= r + d <= n] for r in range(1, n + 1)}
    for r in range(20000, 20400):
        rows[r] = sorted(set(int(c) for c in np.clip(r + rng.integers(-3000, 3000, 24), 1, n)))
    for r in range(40000, 40300):
        rows[r] = sorted(int(c) + 1 for c in rng.choice(n, size=int(rng.integers(20, 60)), replace=False))
    rows[50000] = sorted(int(c) + 1 for c in rng.choice(n, size=4000, replace=False))
    I = np.concatenate([np.full(len(c), r) for r, c in rows.items()])
    J = np.concatenate([np.asarray(c) for c in rows.values()])
    V = rng.standard_normal(len(I))
    A = pa.compresscoo(I, J, V, n, n)
    oA = orc.CSR(A.m, A.n, A.rowptr, A.colval, A.nzval)
    xh = rng.standard_normal(n)
    exp = orc.oracle_c().mul5_csr(np.full(n, 0.25), oA, xh, 0.75, -2.0)
    x = pa.DeviceVector(n, 0).upload(xh)
    out = {}
    for pat in ("1", "0"):
        monkeypatch.setenv("PA_SPMV_PATTERN", pat)
        dA = pa.DeviceCSR(A)
        y = pa.DeviceVector(n, 0).upload(np.full(n, 0.25))
        pa.spmv_(y, dA, x, alpha=0.75, beta=-2.0)
        assert np.array_equal(y.download(), exp), pat
        out[pat] = (dA.encoding(), dA.device_bytes(), dA.info())
    enc, nbytes, info = out["1"]
    assert enc["pattern"] > 0 and enc["c16"] > 0 and enc["c32"] > 0 and info["n_long_rows"] == 1
    assert enc["pattern"] + enc["c16"] + enc["c32"] == info["n_chunks"]
    assert out["0"][0]["pattern"] == 0 and out["0"][0]["c16"] > 0
    assert nbytes < 10 * A.nnz + 8 * n and out["0"][1] > 14 * A.nnz      # values (+ few columns) vs values + both streams
    # the same through the colour-update and restriction kernels' dispatch is covered by the MG tests (HPCG blocks
    # are compacted the same way); the 27-point operator: 8 bytes per entry + row pointers
    monkeypatch.delenv("PA_SPMV_PATTERN")
    A27, _ = pa.build_p_matrix(ranks(1), 64, 64, 64, 64, 64, 64, 1, 1, 1)
    blk = A27.matrix_partition.items[0].own_own
    # (8 B values + descriptors; round 4: + the one-byte codes of the automatic value dictionary where the block has one)
    assert blk.device_bytes() < (9.0 + (1.0 if blk.value_dict() else 0.0)) * blk.nnz


def test_unstructured_rows_in_a_band_keep_their_bits(orc):
    """Rows of 16 random columns within +-2000 of the diagonal plus a few that reach anywhere: no row pattern survives, the
    chunks ride the 16-bit window stream (or 32-bit columns where a chunk needs more than 16 windows).  spmv! and the
    alpha/beta form are bit-identical to the oracle's loops."""
    rng = np.random.default_rng(3)
    m = 300_000
    base = np.repeat(np.arange(m), 16)
    col = np.clip(base + rng.integers(-2000, 2000, size=m * 16), 0, m - 1).reshape(m, 16)
    far = rng.choice(m, size=40, replace=False)
    col[far, 0] = rng.integers(0, m, size=40)
    col = np.sort(col, axis=1)
    rp = (1 + 16 * np.arange(m + 1)).astype(np.int32)
    H = pa.HostCSR(m, m, rp, (col.ravel() + 1).astype(np.int32), rng.standard_normal(m * 16))
    xh = rng.standard_normal(m)
    Ho = orc.CSR(m, m, H.rowptr, H.colval, H.nzval)
    want = np.zeros(m)
    orc.oracle_c().spmv_csr(want, xh, Ho)
    x = pa.DeviceVector(m, 0).upload(xh)
    A = pa.DeviceCSR(H)
    enc = A.encoding()
    assert enc["pattern"] == 0 and enc["c16"] > 0
    y = pa.DeviceVector(m, 0)
    pa.spmv_(y, A, x)
    assert np.array_equal(y.download(), want)
    y.upload(np.full(m, 0.25))
    pa.spmv_(y, A, x, alpha=-2.0, beta=3.0)
    y0 = np.full(m, 0.25)
    orc.oracle_c().mul5_csr(y0, Ho, xh, -2.0, 3.0)
    assert np.array_equal(y.download(), y0)


def test_x_window_launch_of_banded_rows_is_bit_identical(orc, monkeypatch):
    """Banded rows without a pattern go through k_spmv_xwin (groups of chunks gather x from an LDS copy of their span) and
    what fits no group through k_spmv_rowsplit's chunk list: ragged rows (0..39 entries, empty ones included), a band of
    +-1500, rows that reach anywhere (their chunks leave the groups), a stretch of rows too wide for any window, signed
    zeros.  spmv! and the alpha/beta form equal the oracle's spmv_csr! / mul! loops bit for bit, with the window launch on
    and off, and on a vector segment that is only 8-byte aligned (the ghost segment of a vector with an odd own length)."""
    import pa_amd._lib as L
    rng = np.random.default_rng(11)
    m = 200_001
    lens = rng.integers(0, 40, m)
    lens[rng.choice(m, 500, replace=False)] = 0
    rp = np.concatenate([[1], 1 + np.cumsum(lens)]).astype(np.int32)
    rows = np.repeat(np.arange(m), lens)
    col = np.clip(rows + rng.integers(-1500, 1500, size=len(rows)), 0, m - 1)
    far = rng.choice(len(rows), size=60, replace=False)
    col[far] = rng.integers(0, m, size=60)
    wide = (rows >= 90_000) & (rows < 93_000)                       # spans of 20000 columns: no window holds them
    col[wide] = np.clip(rows[wide] + rng.integers(-10000, 10000, size=int(wide.sum())), 0, m - 1)
    order = np.lexsort((col, rows))
    val = rng.standard_normal(len(rows))
    val[rng.choice(len(rows), 2000, replace=False)] = -0.0
    H = pa.HostCSR(m, m, rp, (col[order] + 1).astype(np.int32), val)
    Ho = orc.CSR(m, m, H.rowptr, H.colval, H.nzval)
    xh = rng.standard_normal(m)
    xh[rng.choice(m, 300, replace=False)] = 0.0
    want = np.zeros(m)
    orc.oracle_c().spmv_csr(want, xh, Ho)
    want5 = np.full(m, 0.25)
    orc.oracle_c().mul5_csr(want5, Ho, xh, -2.0, 3.0)
    for switch in ("1", "0"):
        monkeypatch.setenv("PA_SPMV_XWIN", switch)
        A = pa.DeviceCSR(H)
        xw = A.xwin()
        if switch == "1":
            assert xw["groups"] > 0 and 0 < xw["chunks"] < A.info()["n_chunks"], xw       # both launches run
        else:
            assert xw["groups"] == 0
        x = pa.DeviceVector(m, 0).upload(xh)
        y = pa.DeviceVector(m, 0)
        pa.spmv_(y, A, x)
        assert np.array_equal(y.download(), want), switch
        y.upload(np.full(m, 0.25))
        pa.spmv_(y, A, x, alpha=-2.0, beta=3.0)
        assert np.array_equal(y.download(), want5), switch
        # x in the ghost segment of a vector with 3 own entries: the segment starts 24 bytes into the allocation
        xg = pa.DeviceVector(3, m).upload(np.concatenate([np.zeros(3), xh]))
        y2 = pa.DeviceVector(m, 0)
        pa.spmv_(y2, A, xg, x_segment=L.SEG_GHOST)
        assert np.array_equal(y2.download(), want), switch
        # new nonzeros on the same pattern (psparse!-style refresh): both launches read the block's one value stream
        A.update_values(np.ascontiguousarray(-0.5 * H.nzval))
        pa.spmv_(y, A, x)
        assert np.array_equal(y.download(), -0.5 * want), switch


@pytest.mark.parametrize("seed", range(8))
def test_x_window_launch_on_random_banded_blocks(orc, seed, monkeypatch):
    """Random banded blocks (size, band, row-length law, rectangular shapes, alpha/beta all drawn from the seed): the
    product through the library's default choice of launches (x windows where the chunks' gathers are scattered enough and the
    span fits one, the row split otherwise) AND through the forced window launches equals the oracle's loop bit for bit."""
    rng = np.random.default_rng(1000 + seed)
    m = int(rng.integers(100_000, 260_000))
    n = m + int(rng.integers(0, 5000)) * int(seed % 2)                # odd seeds: more columns than rows
    band = int(rng.choice([40, 700, 1800, 2300])) if seed < 6 else 3000        # seeds 6, 7: the 96 KiB windows
    law = seed % 3
    lens = (np.full(m, int(rng.integers(2, 30))) if law == 0 else
            rng.integers(0, int(rng.integers(5, 60)), m) if law == 1 else
            np.where(rng.random(m) < 0.02, rng.integers(200, 1600, m), rng.integers(1, 12, m)))
    rp = np.concatenate([[1], 1 + np.cumsum(lens)]).astype(np.int32)
    rows = np.repeat(np.arange(m), lens)
    col = np.clip(rows + rng.integers(-band, band + 1, size=len(rows)), 0, n - 1)
    order = np.lexsort((col, rows))
    H = pa.HostCSR(m, n, rp, (col[order] + 1).astype(np.int32), rng.standard_normal(len(rows)))
    Ho = orc.CSR(m, n, H.rowptr, H.colval, H.nzval)
    xh = rng.standard_normal(n)
    want = np.zeros(m)
    orc.oracle_c().spmv_csr(want, xh, Ho)
    alpha, beta = float(rng.standard_normal()), float(rng.standard_normal())
    y0 = rng.standard_normal(m)
    want5 = y0.copy()
    orc.oracle_c().mul5_csr(want5, Ho, xh, alpha, beta)
    x = pa.DeviceVector(n, 0).upload(xh)
    for switch in (None, "2"):
        if switch is not None:
            monkeypatch.setenv("PA_SPMV_XWIN", switch)
        A = pa.DeviceCSR(H)
        y = pa.DeviceVector(m, 0)
        pa.spmv_(y, A, x)
        assert np.array_equal(y.download(), want), (seed, band, law, switch, A.xwin())
        y.upload(y0.copy())
        pa.spmv_(y, A, x, alpha=alpha, beta=beta)
        assert np.array_equal(y.download(), want5), (seed, band, law, switch, A.xwin())


def test_fem_matrix_renumbered_by_reverse_cuthill_mckee(orc, monkeypatch):
    """What an unstructured-mesh code does before it assembles: the Q1 mesh numbered at random, then renumbered by reverse
    Cuthill-McKee (scipy).  No row pattern comes back, but the columns do fall into a band, in a few clusters per row (the
    neighbouring level sets): few lines of x per chunk, so the library keeps the block on the row split; forced onto the
    x-window launches it gives the same bits.  Both equal the oracle's spmv_csr!."""
    import scipy.sparse as sp
    from scipy.sparse.csgraph import reverse_cuthill_mckee
    I, J, V, rows, cols = pa.laplacian_fem((400, 300), (1, 1), ranks(1))
    n = 400 * 300
    perm = np.random.default_rng(29).permutation(n)                  # random id of node g (0-based)
    Ip, Jp = perm[I.items[0] - 1], perm[J.items[0] - 1]
    G = sp.csr_matrix((np.ones(len(Ip)), (Ip, Jp)), shape=(n, n))
    order = reverse_cuthill_mckee(G, symmetric_mode=True)            # order[k] = old id of the node that becomes k
    new_id = np.empty(n, np.int64)
    new_id[order] = np.arange(n)
    Hc = pa.compresscoo(new_id[Ip] + 1, new_id[Jp] + 1, V.items[0], n, n)
    band = int(np.max(np.abs(np.repeat(np.arange(n), np.diff(Hc.rowptr)) - (Hc.colval - 1))))
    assert band < 2400, band
    xh = orc.hash_x(np.arange(1, n + 1)) - 0.5
    want = np.zeros(n)
    orc.oracle_c().spmv_csr(want, xh, orc.CSR(n, n, Hc.rowptr, Hc.colval, Hc.nzval))
    x = pa.DeviceVector(n, 0).upload(xh)
    for switch in (None, "2"):
        if switch is not None:
            monkeypatch.setenv("PA_SPMV_XWIN", switch)
        A = pa.DeviceCSR(Hc)
        assert A.encoding()["pattern"] == 0 and (A.xwin()["groups"] > 0) == (switch == "2"), (A.encoding(), A.xwin())
        y = pa.DeviceVector(n, 0)
        pa.spmv_(y, A, x)
        assert np.array_equal(y.download(), want), switch


def test_fem_matrix_on_a_randomly_permuted_mesh(orc):
    """The same Q1 stiffness matrix with its nodes renumbered at random: no row pattern, no band -- every chunk falls to the
    16-bit-window / 32-bit column streams and the plain gather.  Bit-identical to the oracle's spmv_csr!."""
    I, J, V, rows, cols = pa.laplacian_fem((300, 200), (1, 1), ranks(1))
    n = 300 * 200
    perm = np.random.default_rng(17).permutation(n) + 1             # new id of node g = perm[g-1]
    Ip, Jp = perm[I.items[0] - 1], perm[J.items[0] - 1]
    Hc = pa.compresscoo(Ip, Jp, V.items[0], n, n)
    A = pa.DeviceCSR(Hc)
    enc = A.encoding()
    assert enc["pattern"] == 0, enc
    xh = orc.hash_x(np.arange(1, n + 1)) - 0.5
    want = np.zeros(n)
    orc.oracle_c().spmv_csr(want, xh, orc.CSR(n, n, Hc.rowptr, Hc.colval, Hc.nzval))
    y = pa.DeviceVector(n, 0)
    pa.spmv_(y, A, pa.DeviceVector(n, 0).upload(xh))
    assert np.array_equal(y.download(), want)
    # the unpermuted matrix for comparison: row patterns, and the product is the permuted one's, permuted (to rounding:
    # the columns of a row are visited in another order)
    H0 = pa.compresscoo(I.items[0], J.items[0], V.items[0], n, n)
    A0 = pa.DeviceCSR(H0)
    assert A0.encoding()["pattern"] > 0
    y0 = pa.DeviceVector(n, 0)
    x0 = np.zeros(n)
    x0[:] = xh[perm - 1]
    pa.spmv_(y0, A0, pa.DeviceVector(n, 0).upload(x0))
    assert np.allclose(y0.download(), want[perm - 1], rtol=0, atol=1e-11)


@pytest.mark.parametrize("sigma", [1, 256])
def test_sell_c_sigma_one_lane_per_row_is_bit_identical(orc, sigma):
    """SELL-C-sigma storage, one lane walks one row in the reference's order with its sum in a register.
    Against the oracle's spmv_csr! / mul!(y,A,x,alpha,beta) AND against the row-split kernel, bit for bit, on a 27-point
    block, ragged rows with empty ones, rows of thousands of entries, a row count that is no multiple of 64, and values
    whose row sums are -0.0 (padding must not touch them)."""
    rng = np.random.default_rng(7)
    A27, _ = pa.build_p_matrix(ranks(1), 20, 17, 13, 20, 17, 13, 1, 1, 1, keep_host=True)
    cases = [pa.local_items(A27.host_blocks)[0][0],
             _random_csr(rng, 1003, 700, rng.integers(0, 30, 1003) * (rng.random(1003) < 0.7)),
             _random_csr(rng, 130, 6000, np.concatenate([[4000, 0, 2500], rng.integers(0, 9, 127)]))]
    for H in cases:
        xh = rng.standard_normal(H.n)
        Ho = orc.CSR(H.m, H.n, H.rowptr, H.colval, H.nzval)
        want = np.zeros(H.m)
        orc.oracle_c().spmv_csr(want, xh, Ho)
        S, D = pa.DeviceSELL(H, sigma=sigma), pa.DeviceCSR(H)
        info = S.info()
        assert info["nnz"] == H.nnz and info["padded_entries"] >= H.nnz and info["n_slabs"] == (H.m + 63) // 64
        x = pa.DeviceVector(H.n, 0).upload(xh)
        ys, yd = pa.DeviceVector(H.m, 0), pa.DeviceVector(H.m, 0)
        pa.spmv_(ys, S, x)
        pa.spmv_(yd, D, x)
        assert np.array_equal(ys.download(), want) and np.array_equal(yd.download(), want)
        y0 = rng.standard_normal(H.m)
        ys.upload(y0)
        pa.spmv_(ys, S, x, alpha=-0.75, beta=2.5)
        w5 = y0.copy()
        orc.oracle_c().mul5_csr(w5, Ho, xh, -0.75, 2.5)
        assert np.array_equal(ys.download(), w5)
    if sigma > 1:                                        # sorting by length is what keeps the padding of ragged rows small
        H = cases[1]
        assert pa.DeviceSELL(H, sigma=sigma).info()["padded_entries"] < pa.DeviceSELL(H, sigma=1).info()["padded_entries"]
    # signed zeros: beta*y = -0.0 on an empty row and products that are all -0.0 must come out as the reference's loop leaves them
    H = pa.HostCSR(3, 2, np.array([1, 1, 3, 4], np.int32), np.array([1, 2, 1], np.int32), np.array([-0.0, 0.0, -1.0]))
    S = pa.DeviceSELL(H, sigma=sigma)
    x = pa.DeviceVector(2, 0).upload(np.array([1.0, 1.0]))
    y = pa.DeviceVector(3, 0).upload(np.array([0.0, 0.0, 0.0]))
    pa.spmv_(y, S, x, alpha=1.0, beta=-1.0)
    w = np.zeros(3)
    orc.oracle_c().mul5_csr(w, orc.CSR(3, 2, H.rowptr, H.colval, H.nzval), np.array([1.0, 1.0]), 1.0, -1.0)
    got = y.download()
    assert np.array_equal(got, w) and np.array_equal(np.signbit(got), np.signbit(w))


def test_new_entry_points_return_statuses_on_bad_arguments():
    """The round-2 entry points keep the ABI's convention: a bad call is a status + message, never an abort."""
    import pa_amd._lib as L
    import ctypes as C
    A, b = pa.build_p_matrix(ranks(1), 6, 6, 6, 6, 6, 6, 1, 1, 1, keep_host=True)
    H = pa.local_items(A.host_blocks)[0][0]
    x, y = pa.pzeros(A.col_partition), pa.pzeros(A.row_partition)
    xv, yv = x.vector_partition.items[0], y.vector_partition.items[0]
    import pa_amd.p_sparse_matrix as psm
    h = psm._operator_handles(A, x).items[0]
    for args, what in (((h, None, yv.h, xv.h, 99, 0), "slot"), ((h, None, xv.h, xv.h, 3, 0), "alias")):
        with pytest.raises(L.PAError, match=what):
            L.call("pa_mul_dot", *args)
    with pytest.raises(L.PAError, match="distinct"):
        L.call("pa_cg_r_update", xv.h, xv.h, 1, 2, 3, 0)
    with pytest.raises(L.PAError, match="result slot"):
        L.call("pa_cg_r_update", xv.h, yv.h, 3, 2, 3, 0)
    with pytest.raises(L.PAError, match="distinct"):
        L.call("pa_cg_xu_update", xv.h, xv.h, yv.h, 1, 2, 1, 2)
    with pytest.raises(L.PAError, match="alias"):
        L.call("pa_mul_no_lat", h, None, xv.h, xv.h)
    out = C.c_void_p()
    with pytest.raises(L.PAError, match="sigma"):
        L.call("pa_sell_create", pa.context().h, H.m, H.n, H.nnz, L.ptr(H.rowptr), L.ptr(H.colval), 4, 1, L.ptr(H.nzval), 0, C.byref(out))
    bad = H.colval.copy()
    bad[3] = H.n + 5
    with pytest.raises(L.PAError, match="column index out of range"):
        L.call("pa_sell_create", pa.context().h, H.m, H.n, H.nnz, L.ptr(H.rowptr), L.ptr(bad), 4, 1, L.ptr(H.nzval), 1, C.byref(out))
    S = pa.DeviceSELL(H)
    with pytest.raises(L.PAError, match="size"):
        pa.spmv_(pa.DeviceVector(H.m + 1, 0), S, pa.DeviceVector(H.n, 0))
    assert L.lib.pa_ctx_arena_info(None, None, None, None, None, None, None) == -2


@pytest.mark.parametrize("ring", ["1", "2"])
def test_sliding_x_window_launch_is_bit_identical(orc, monkeypatch, ring):
    """k_spmv_xring (csrc/pa_spmv_xwin.h): runs of consecutive chunks gather x from a ring of 16384 entries that every round
    tops up with the columns above the highest one loaded so far.  Bands of +-2500 / +-6000 / +-7900, ragged and empty rows,
    a stretch where the band JUMPS by 3000 columns (more new entries than one lane each can fetch), rows that reach anywhere
    and a stretch too wide for the ring (both leave the runs for the chunk list), signed zeros; spmv!, the alpha/beta form,
    x in an 8-byte-aligned ghost segment and new values on the same pattern -- bit for bit against the oracle's loops, with
    the ring behind the window tiers (1: it takes what they leave, the +-7900 stretch) and alone (2)."""
    import pa_amd._lib as L
    monkeypatch.setenv("PA_SPMV_XRING", ring)
    monkeypatch.setenv("PA_SPMV_XWIN", "2")        # groups wherever they can be formed (on a block this small the planner would
    rng = np.random.default_rng(31)                 # decline the runs of 5 chunks: more x loaded than matrix streamed)
    m = 400_003
    lens = rng.integers(0, 36, m)
    lens[rng.choice(m, 800, replace=False)] = 0
    rp = np.concatenate([[1], 1 + np.cumsum(lens)]).astype(np.int32)
    rows = np.repeat(np.arange(m), lens)
    band = np.where(rows < 120_000, 2500, np.where(rows < 260_000, 6000, 7900))
    centre = rows + np.where(rows >= 200_000, 3000, 0)                     # the band jumps at row 200000
    col = np.clip(centre + (rng.random(len(rows)) * 2 - 1) * band, 0, m - 1).astype(np.int64)
    far = rng.choice(len(rows), size=80, replace=False)
    col[far] = rng.integers(0, m, size=80)
    wide = (rows >= 300_000) & (rows < 304_000)                            # spans of 24000 columns: no ring holds them
    col[wide] = np.clip(rows[wide] + rng.integers(-12000, 12000, size=int(wide.sum())), 0, m - 1)
    order = np.lexsort((col, rows))
    val = rng.standard_normal(len(rows))
    val[rng.choice(len(rows), 3000, replace=False)] = -0.0
    H = pa.HostCSR(m, m, rp, (col[order] + 1).astype(np.int32), val)
    Ho = orc.CSR(m, m, H.rowptr, H.colval, H.nzval)
    xh = rng.standard_normal(m)
    want = np.zeros(m)
    orc.oracle_c().spmv_csr(want, xh, Ho)
    want5 = np.full(m, 0.25)
    orc.oracle_c().mul5_csr(want5, Ho, xh, -2.0, 3.0)
    A = pa.DeviceCSR(H)
    xw = A.xwin()
    assert xw["groups"] > 0 and 0 < xw["chunks"] < A.info()["n_chunks"], xw
    if ring == "2":                                 # (behind the forced windows the ring may be left with nothing on this block)
        assert xw["ring_groups"] == xw["groups"] and xw["big_groups"] == 0, xw
    x = pa.DeviceVector(m, 0).upload(xh)
    y = pa.DeviceVector(m, 0)
    pa.spmv_(y, A, x)
    assert np.array_equal(y.download(), want)
    y.upload(np.full(m, 0.25))
    pa.spmv_(y, A, x, alpha=-2.0, beta=3.0)
    assert np.array_equal(y.download(), want5)
    xg = pa.DeviceVector(3, m).upload(np.concatenate([np.zeros(3), xh]))
    y2 = pa.DeviceVector(m, 0)
    pa.spmv_(y2, A, xg, x_segment=L.SEG_GHOST)
    assert np.array_equal(y2.download(), want)
    A.update_values(np.ascontiguousarray(-0.5 * H.nzval))
    pa.spmv_(y, A, x)
    assert np.array_equal(y.download(), -0.5 * want)
    monkeypatch.setenv("PA_SPMV_XWIN", "0")                               # the row split alone on the same block: same bits
    B = pa.DeviceCSR(H)
    assert B.xwin()["groups"] == 0
    pa.spmv_(y2, B, x)
    assert np.array_equal(y2.download(), want)
