"""SURVEY 8(a) rows a3/a5 at the third-party boundary: the 5-argument product of a block the caller keeps in the DEFAULT
SparseMatrixCSC storage.  mul!(y,A,x,alpha,beta) is third-party arithmetic in the reference (src/p_sparse_matrix.jl:2116-2138 call
it on the local blocks), and its two local matrix types differ by one rounding when alpha is not a power of two:
SparseMatricesCSR 0.6 adds (nz*x[col])*alpha, SparseArrays' CSC method forms axj = x[col]*alpha per column and adds nz*axj
(restated in oracle/pa_oracle.py::mul5_csc; the in-repo spmv!/spmtv! of test/sparse_utils_tests.jl:33-45 only pin alpha = 1).
pa_csr_create_from_csc blocks follow the CSC form, every other block the CSR form -- both bit for bit."""
import ctypes as C

import numpy as np
import pytest

from gpu_common import pa
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
