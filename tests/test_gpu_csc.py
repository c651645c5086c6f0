"""SURVEY 8(a) rows a3/a5 at the third-party boundary: the 5-argument product of a block the caller keeps in the DEFAULT
SparseMatrixCSC storage.  mul!(y,A,x,alpha,beta) is third-party arithmetic in the reference (src/p_sparse_matrix.jl:2116-2138 call
it on the local blocks), and its two local matrix types differ by one rounding when alpha is not a power of two:
SparseMatricesCSR 0.6 adds (nz*x[col])*alpha, SparseArrays' CSC method forms axj = x[col]*alpha per column and adds nz*axj
(restated in oracle/pa_oracle.py::mul5_csc; the in-repo spmv!/spmtv! of test/sparse_utils_tests.jl:33-45 only pin alpha = 1).
pa_csr_create_from_csc blocks follow the CSC form, every other block the CSR form -- both bit for bit."""
import ctypes as C

import numpy as np
import pytest

from gpu_helpers import pa
import pa_amd._lib as L

pytestmark = pytest.mark.gpu


def _random_csr(rng, m, n, lens):
    rows = [np.sort(rng.choice(n, size=int(k), replace=False)) for k in lens]
    rp = (1 + np.concatenate(([0], np.cumsum([len(r) for r in rows])))).astype(np.int32)
    cv = (np.concatenate(rows) + 1).astype(np.int32) if rp[-1] > 1 else np.zeros(0, np.int32)
    return pa.HostCSR(m, n, rp, cv, rng.standard_normal(len(cv)))


@pytest.mark.parametrize("alpha,beta", [(1.0, 0.0), (0.3, 0.0), (0.3, -1.7), (-2.0, 1.0), (1e-3, 0.5)])
def test_five_argument_product_of_a_csc_block_follows_sparsearrays(orc, alpha, beta):
    rng = np.random.default_rng(21)
    A = _random_csr(rng, 700, 500, rng.integers(0, 40, 700))
    oA = orc.CSR(A.m, A.n, A.rowptr, A.colval, A.nzval)
    colptr, rowval, nzval = orc.csr_to_csc(oA)
    x, y0 = rng.standard_normal(A.n), rng.standard_normal(A.m)
    want_csc = orc.mul5_csc(y0.copy(), x, colptr, rowval, nzval, alpha, beta)
    want_csr = orc.mul5_csr(y0.copy(), oA, x, alpha, beta)
    xd = pa.DeviceVector(A.n, 0).upload(x)
    got = {}
    for name, blk in (("csc", pa.DeviceCSR.from_csc(A.m, A.n, colptr, rowval, nzval)), ("csr", pa.DeviceCSR(A))):
        yd = pa.DeviceVector(A.m, 0).upload(y0)
        pa.spmv_(yd, blk, xd, L.SEG_OWN, L.SEG_OWN, alpha, beta)
        got[name] = yd.download()
    assert np.array_equal(got["csc"], want_csc)
    assert np.array_equal(got["csr"], want_csr)
    if alpha in (1.0, -2.0):
        assert np.array_equal(want_csc, want_csr)                  # powers of two: the two forms agree
    elif alpha == 0.3:
        assert not np.array_equal(want_csc, want_csr)              # ... otherwise they are one rounding apart somewhere
        assert np.allclose(want_csc, want_csr, rtol=1e-13, atol=1e-13)


def test_csc_form_can_be_switched_per_block():
    rng = np.random.default_rng(3)
    A = _random_csr(rng, 300, 300, rng.integers(1, 20, 300))
    import pa_oracle as orc
    oA = orc.CSR(A.m, A.n, A.rowptr, A.colval, A.nzval)
    colptr, rowval, nzval = orc.csr_to_csc(oA)
    blk = pa.DeviceCSR.from_csc(A.m, A.n, colptr, rowval, nzval)
    x = rng.standard_normal(A.n)
    xd = pa.DeviceVector(A.n, 0).upload(x)
    yd = pa.DeviceVector(A.m, 0)
    L.call("pa_csr_set_alpha_inside", blk.h, 0)
    pa.spmv_(yd, blk, xd, L.SEG_OWN, L.SEG_OWN, 0.3, 0.0)
    assert np.array_equal(yd.download(), orc.mul5_csr(np.zeros(A.m), oA, x, 0.3, 0.0))
    L.call("pa_csr_set_alpha_inside", blk.h, 1)
    pa.spmv_(yd, blk, xd, L.SEG_OWN, L.SEG_OWN, 0.3, 0.0)
    assert np.array_equal(yd.download(), orc.mul5_csc(np.zeros(A.m), x, colptr, rowval, nzval, 0.3, 0.0))


def test_mul_of_a_matrix_whose_blocks_are_csc_follows_the_csc_form_through_every_route(orc):
    """mul!(c,a,b,alpha,beta) (src/p_sparse_matrix.jl:2105-2142) when the local blocks keep the default SparseMatrixCSC storage:
    own x own and own x ghost each follow SparseArrays' a*(x*alpha) -- also through pa_mul_all, where own x ghost is a TWIN of the
    block (columns renamed to receive-buffer positions, made at the first product) running on the comm stream beside own x own."""
    from gpu_helpers import ranks, upload
    A, _ = pa.build_p_matrix(ranks(4), 6, 5, 4, 12, 10, 4, 2, 2, 1, keep_host=True, fused=True)
    Ao, _, _ = orc.hpcg_build_p_matrix(6, 5, 4, 2, 2, 1)

    def csc_block(h):
        cp, rv, nz = orc.csr_to_csc(orc.CSR(h.m, h.n, h.rowptr, h.colval, h.nzval))
        return pa.DeviceCSR.from_csc(h.m, h.n, cp, rv, nz), (cp, rv, nz)
    blocks, host = [], []
    for h in A.host_blocks.items:
        oo, oo_h = csc_block(h[0])
        oh, oh_h = csc_block(h[1])
        blocks.append(pa.SplitMatrixBlocks(oo, oh))
        host.append((oo_h, oh_h))
    Ac = pa.PSparseMatrix(pa.DebugArray(blocks), A.row_partition, A.col_partition, True)
    alpha, beta = 0.3, -1.7
    xo = [orc.hash_x(c.local_to_global) * (c.local_to_owner == c.part) for c in Ao.cols]
    xc = [v.copy() for v in xo]
    orc.consistent(xc, Ao.cols)
    y0 = [orc.hash_x(r.local_to_global + 7) for r in Ao.rows]
    want = []
    for (oo_h, oh_h), xv, yv, c in zip(host, xc, y0, Ao.cols):
        w = orc.mul5_csc(yv[:c.n_own].copy(), xv[:c.n_own], *oo_h, alpha, beta)
        want.append(orc.mul5_csc(w, xv[c.n_own:], *oh_h, alpha, 1.0))
    for f in (pa.mul5_, pa.mul_c_):
        x = upload([v.copy() for v in xo], Ac.col_partition)
        y = upload([v.copy() for v in y0], Ac.row_partition)
        for _ in range(2 if f is pa.mul_c_ else 1):                 # (twice: the second product runs on the twin)
            y = upload([v.copy() for v in y0], Ac.row_partition)
            f(y, Ac, x, alpha, beta)
        for got, e in zip(y.own_values().items, want):
            assert np.array_equal(got, e), f.__name__
