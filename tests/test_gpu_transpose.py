"""SURVEY 8(f) row f2 -- mul!(c,transpose(a),b,alpha,beta) (src/p_sparse_matrix.jl:2144-2162) and spmtv! (src/sparse_utils.jl:613-647)
with transpose(A) built ON THE DEVICE from the resident blocks (csrc/pa_transpose.hip): no host copy of the matrix exists in
any of these tests unless a test builds one to compare with.  Bit-exact bars (np.array_equal) against the oracle."""
import ctypes as C

import numpy as np
import pytest

from gpu_helpers import pa, ranks, upload, env
import pa_amd._lib as L

pytestmark = pytest.mark.gpu


def _entries(blk):
    rows, cols = np.zeros(max(blk.nnz, 1), np.int32), np.zeros(max(blk.nnz, 1), np.int32)
    L.call("pa_csr_download_entries", blk.h, L.ptr(rows), L.ptr(cols))
    return rows[:blk.nnz], cols[:blk.nnz]


def _host_entries(h):
    rows = np.repeat(np.arange(h.m, dtype=np.int32), np.diff(h.rowptr.astype(np.int64)))
    return rows, (h.colval - 1).astype(np.int32)


def _random_rows(rng, m, n, per_row, band):
    col = np.repeat(np.arange(m, dtype=np.int64) * n // m, per_row).reshape(m, per_row) + rng.integers(-band, band + 1, size=(m, per_row))
    col = np.clip(col, 0, n - 1)
    rows = []
    for r in range(m):
        rows.append(np.unique(col[r]))
    rp = np.concatenate(([0], np.cumsum([len(c) for c in rows]))).astype(np.int32) + 1
    cv = (np.concatenate(rows) + 1).astype(np.int32)
    return pa.HostCSR(m, n, rp, cv, rng.standard_normal(len(cv)))


@pytest.mark.parametrize("switches", [{}, {"PA_SPMV_PATTERN": "0"}, {"PA_SPMV_PATTERN": "0", "PA_SPMV_COL16": "0"},
                                      {"PA_SPMV_COMPACT_STREAMS": "0"}, {"PA_SETUP_DEVICE": "0"}])
def test_decoded_entries_equal_the_arrays_the_block_was_made_from(switches):
    """The decode kernel of the transpose (kt_decode) against every column encoding the product kernel reads: row patterns
    (27-point blocks, with the face / corner chunks on compacted 16-bit or Int32 streams), 16-bit windows (banded rows), Int32
    (wide rows), row-compacted blocks with a row-id stride (own|ghost, one Gauss-Seidel colour), one long row."""
    rng = np.random.default_rng(5)
    with env(**switches):
        A, _ = pa.build_p_matrix(ranks(2), 12, 10, 8, 24, 10, 8, 2, 1, 1, keep_host=True, fused=True)
        for blk, h in zip(A.matrix_partition.items, A.host_blocks.items):
            for B, H in ((blk.own_own, h[0]), (blk.own_ghost, h[1])):
                r, c = _entries(B)
                hr, hc = _host_entries(H)
                assert np.array_equal(r, hr) and np.array_equal(c, hc), switches
        # one colour of a colouring: a row-compacted block whose runs have a row-id stride
        h = A.host_blocks.items[0][0]
        keep = np.zeros(h.m, bool)
        keep[::2] = True
        lens = np.where(keep, np.diff(h.rowptr.astype(np.int64)), 0)
        rp = np.concatenate(([0], np.cumsum(lens))).astype(np.int32) + 1
        sel = np.repeat(keep, np.diff(h.rowptr.astype(np.int64)))
        Hc = pa.HostCSR(h.m, h.n, rp, h.colval[sel].copy(), h.nzval[sel].copy())
        for H in (Hc, _random_rows(rng, 5000, 5000, 12, 300), _random_rows(rng, 3000, 400000, 9, 150000),
                  pa.HostCSR(3, 6000, np.array([1, 3, 5003, 5004], np.int32),
                             np.concatenate(([1, 7], np.arange(1, 5001), [17])).astype(np.int32), rng.standard_normal(5003))):
            B = pa.DeviceCSR(H)
            r, c = _entries(B)
            hr, hc = _host_entries(H)
            assert np.array_equal(r, hr) and np.array_equal(c, hc), (switches, H.m, H.n)


def test_device_transpose_stores_the_reference_s_scatter_order():
    """A' from pa_csr_create_transpose: row j holds A's entries of column j in ascending row of A (the order spmv_csc! on the
    CSR arrays adds them in, src/sparse_utils.jl:671-690), values moved with their entries; equal to the host-side counting
    transpose (pa_csr_create_from_csc of the same arrays), product for product."""
    rng = np.random.default_rng(11)
    for H in (_random_rows(rng, 4000, 3000, 10, 200), _random_rows(rng, 700, 90000, 7, 30000)):
        B = pa.DeviceCSR(H)
        h = C.c_void_p()
        L.call("pa_csr_create_transpose", B.h, C.byref(h))
        T = pa.DeviceCSR.from_handle(h, H.n, H.m, H.nnz)
        assert T.info()["n_rows"] == H.n and T.info()["n_cols"] == H.m and T.info()["nnz"] == H.nnz
        r, c = _entries(T)
        hr, hc = _host_entries(H)
        order = np.argsort(hc, kind="stable")
        assert np.array_equal(r, hc[order]) and np.array_equal(c, hr[order])
        x = rng.standard_normal(H.m)
        y0 = rng.standard_normal(H.n)
        xd = pa.DeviceVector(H.m, 0).upload(x)
        got = []
        for blk in (T, pa.DeviceCSR.transposed(H)):
            yd = pa.DeviceVector(H.n, 0).upload(y0)
            pa.spmv_(yd, blk, xd, L.SEG_OWN, L.SEG_OWN, 0.75, -1.5)
            got.append(yd.download())
        exp = -1.5 * y0                                     # the reference's loop: rmul!(y, beta); y[col] += nz * x[row] * alpha
        rows = hr
        for p in range(H.nnz):
            exp[hc[p]] = exp[hc[p]] + (H.nzval[p] * x[rows[p]]) * 0.75
        assert np.array_equal(got[0], got[1]) and np.array_equal(got[0], exp)


@pytest.mark.parametrize("n,parts", [((6, 5, 4), (2, 2, 1)), ((32, 32, 32), (2, 2, 2)), ((64, 64, 64), (2, 2, 2))])
def test_transpose_product_of_a_generated_matrix(orc, n, parts):
    """mul!(c,transpose(a),b,alpha,beta) on HPCG 27-point matrices GENERATED IN HBM (no host copy: the round-3 wrapper raised
    here), up to BASELINE config 4's shape at 64^3 rows per part on 8 parts: bit-exact against the oracle for (1,0) and a
    general (alpha,beta); A = A', so the result also equals A*b to rounding."""
    nx, ny, nz = n
    px, py, pz = parts
    P = px * py * pz
    A, _ = pa.build_p_matrix(ranks(P), nx, ny, nz, px * nx, py * ny, pz * nz, px, py, pz)
    assert A.host_blocks is None
    Ao, _, _ = orc.hpcg_build_p_matrix(nx, ny, nz, px, py, pz)
    for alpha, beta in [(1.0, 0.0), (-0.5, 2.0), (0.3, 1.0)]:
        bo = [orc.hash_x(r.local_to_global + 1) for r in Ao.rows]
        co = [orc.hash_x(c.local_to_global + 9) for c in Ao.cols]
        b = upload([v.copy() for v in bo], A.row_partition)
        c = upload([v.copy() for v in co], A.col_partition)
        pa.mul5_transpose_(c, A, b, alpha, beta)
        orc.mul5_transpose(co, Ao, bo, alpha, beta)
        for got, exp in zip(c.local_values().items, co):
            assert np.array_equal(got, exp), (alpha, beta)
    bo = [orc.hash_x(c.local_to_global) * (c.local_to_owner == c.part) for c in Ao.cols]
    y = pa.pzeros(A.row_partition)
    pa.mul_(y, A, upload([v.copy() for v in bo], A.col_partition))
    c = pa.pzeros(A.col_partition)
    pa.mul5_transpose_(c, A, upload([v[:r.n_own].copy() for v, r in zip(bo, Ao.rows)], A.row_partition), 1.0, 0.0)
    assert np.allclose(y.collect(), c.collect(), rtol=0, atol=1e-12)


@pytest.mark.parametrize("nodes,parts", [((23, 17), (2, 2)), ((9, 7, 8), (2, 2, 2)), ((200, 160), (4, 2))])
def test_transpose_product_of_a_device_assembled_fem_matrix(orc, nodes, parts):
    """BASELINE config 5's route (disassembled psparse + assemble on the device, irregular rows, ghost-heavy) and its transpose
    product, then the same after psparse! put new values on the pattern (the cached A' must not survive the update)."""
    P = int(np.prod(parts))
    I, J, V, rows, cols = pa.laplacian_fem(nodes, parts, ranks(P))
    A, cache = pa.psparse_disassembled(I, J, V, rows, cols, reuse=True)
    Io, Jo, Vo, orows, ocols = orc.laplacian_fem(nodes, parts)
    Vcur = V
    for rep in range(2):
        Ao, _ = orc.psparse_disassembled(Io, Jo, [v.copy() for v in Vcur.items], orows, ocols)
        bo = [orc.hash_x(r.local_to_global + 3) for r in Ao.rows]
        co = [orc.hash_x(c.local_to_global + 5) for c in Ao.cols]
        b = pa.pvector_from_function(lambda ind: orc.hash_x(ind.get_local_to_global() + 3), A.row_partition)   # (only own values are read)
        c = upload([v.copy() for v in co], A.col_partition)
        pa.mul5_transpose_(c, A, b, 1.25, -0.5)
        orc.mul5_transpose(co, Ao, bo, 1.25, -0.5)
        for got, exp in zip(c.local_values().items, co):
            assert np.array_equal(got, exp), rep
        Vcur = pa.pmap(lambda v, i: v * 2.0 + orc.hash_x(np.arange(len(v)) + 7 * int(i[0])) * 1e-3, V, I)
        pa.psparse_(A, Vcur, cache).wait()
