"""Float32 blocks and vectors (csrc/pa_f32.hip): the first widening beyond the path's FP64 scope (VERDICT r05 "Next" #7).

The reference's local loops are generic in the element type; its own test runs them on a fixed 7 x 6 matrix as
SparseMatrixCSC{Float32,Int32}, SparseMatrixCSR{0,Float32,Int32} and SparseMatrixCSR{1,Float32,Int32}
(test/sparse_utils_tests.jl:10-45,72-79: spmv! against the library mul!, spmtv! against mul!(transpose)).  Here pa_spmv32 against the
oracle's float loops (oracle/pa_oracle.c: orc_spmv_csr_f32 / orc_spmv_csc_f32 / orc_mul5_csr_f32, every product and sum rounded to
float, -ffp-contract=off), bit for bit: that matrix in all three storages (and its hand-worked product), irregular and empty rows,
the alpha/beta form, and the 27-point operator on the pattern-ELL structure with the 4-byte value stream.
"""
import numpy as np
import pytest

from gpu_helpers import pa
import pa_amd._lib as L

pytestmark = pytest.mark.gpu
F32 = np.float32


def _csr_from_coo(I, J, V, m, n):
    """compresscoo (src/sparse_utils.jl:313-350): duplicates combined with +, columns ascending inside a row; 1-based Int32 arrays"""
    d = {}
    for i, j, v in zip(I, J, V):
        d[(i, j)] = F32(d.get((i, j), F32(0)) + F32(v))
    keys = sorted(d)
    rowptr = np.zeros(m + 1, np.int64)
    for i, _ in keys:
        rowptr[i] += 1
    rowptr = (np.cumsum(rowptr) + 1).astype(np.int32)
    return rowptr, np.array([j for _, j in keys], np.int32), np.array([d[k] for k in keys], F32)


def _csc_of(rowptr, colval, nzval, m, n):
    ent = sorted(((int(colval[p - 1]), r + 1, nzval[p - 1]) for r in range(m) for p in range(rowptr[r], rowptr[r + 1])))
    colptr = np.zeros(n + 1, np.int64)
    for j, _, _ in ent:
        colptr[j] += 1
    return (np.cumsum(colptr) + 1).astype(np.int32), np.array([i for _, i, _ in ent], np.int32), np.array([v for _, _, v in ent], F32)


def test_the_references_7_by_6_matrix_in_float32(orc, golden):
    c = golden["sparse_utils_mat"]
    m, n = c["m"], c["n"]
    rowptr, colval, nzval = _csr_from_coo(c["I"], c["J"], c["V"], m, n)
    x = np.arange(1, n + 1, dtype=F32)                        # collect(Tv,1:size(B,2))
    K = orc.oracle_c()
    want = K.spmv_csr_f32(np.ones(m, F32), x, rowptr, colval, nzval)
    assert want.tolist() == [F32(v) for v in c["Ax"]]         # (small integers: exact in Float32 too)
    xd = pa.DeviceVector32(n).upload(x)
    colptr, rowval, cnz = _csc_of(rowptr, colval, nzval, m, n)
    assert np.array_equal(K.spmv_csc_f32(np.ones(m, F32), x, colptr, rowval, cnz), want)
    for name, A in (("CSR{1}", pa.DeviceCSR32(m, n, rowptr, colval, nzval)),
                    ("CSR{0}", pa.DeviceCSR32(m, n, rowptr - 1, colval - 1, nzval, index_base=0)),
                    ("CSR{1} Int64", pa.DeviceCSR32(m, n, rowptr.astype(np.int64), colval.astype(np.int64), nzval)),
                    ("CSC", pa.DeviceCSR32(m, n, colptr, rowval, cnz, csc=True))):
        y = pa.DeviceVector32(m).upload(np.ones(m, F32))      # b1 = ones(Tv,size(B,1)): spmv! overwrites it
        pa.spmv32_(y, A, xd)
        assert np.array_equal(y.download(), want), name
    # spmtv!(b,B,x) (src/sparse_utils.jl:625-631: the CSR arrays through spmv_csc!): the transposed block is the CSC reading of the CSR arrays
    xt = np.arange(1, m + 1, dtype=F32)
    want_t = K.spmv_csc_f32(np.ones(n, F32), xt, rowptr, colval, nzval)
    At = pa.DeviceCSR32(n, m, rowptr, colval, nzval, csc=True)
    yt = pa.DeviceVector32(n).upload(np.ones(n, F32))
    pa.spmv32_(yt, At, pa.DeviceVector32(m).upload(xt))
    assert np.array_equal(yt.download(), want_t)


@pytest.mark.parametrize("seed", range(4))
def test_random_float32_blocks_against_the_oracles_float_loops(orc, seed):
    rng = np.random.default_rng(700 + seed)
    m, n = int(rng.integers(1, 3000)), int(rng.integers(1, 3000))
    rows = [np.sort(rng.choice(n, size=int(min(n, rng.integers(0, 40))), replace=False)) if rng.random() < 0.9 else np.zeros(0, np.int64) for _ in range(m)]
    rowptr = np.concatenate([[1], 1 + np.cumsum([len(r) for r in rows])]).astype(np.int32)
    colval = (np.concatenate(rows) + 1).astype(np.int32) if rowptr[-1] > 1 else np.zeros(0, np.int32)
    nzval = (rng.standard_normal(len(colval)) * 10.0 ** rng.integers(-3, 4, len(colval))).astype(F32)
    x = rng.standard_normal(n).astype(F32)
    y0 = rng.standard_normal(m).astype(F32)
    K = orc.oracle_c()
    A = pa.DeviceCSR32(m, n, rowptr, colval, nzval)
    assert not A.info()["pattern_ell"]
    xd = pa.DeviceVector32(n).upload(x)
    y = pa.DeviceVector32(m).upload(y0)
    pa.spmv32_(y, A, xd)
    assert np.array_equal(y.download(), K.spmv_csr_f32(np.zeros(m, F32), x, rowptr, colval, nzval))
    for alpha, beta in ((1.0, 1.0), (0.3, -1.5), (-2.0, 0.0)):
        y.upload(y0)
        pa.spmv32_(y, A, xd, alpha=alpha, beta=beta)
        assert np.array_equal(y.download(), K.mul5_csr_f32(y0.copy(), x, rowptr, colval, nzval, alpha, beta)), (alpha, beta)


@pytest.mark.parametrize("n", [24, 48])
def test_the_27_point_operator_in_float32_on_the_pattern_ell_structure(orc, n):
    """The HPCG operator of one part with Float32 values (26 and -1 are exact) and a hashed Float32 x: the block qualifies for the
    fp64 path's pattern-ELL structure (9 slab patterns), its 4-byte value stream gives the oracle's bits, and so does the alpha/beta
    form; PA_SPMV_PELL=0 (SELL-64) gives the same bits again."""
    import os
    Ao = orc.hpcg_build_p_matrix(n, n, n, 1, 1, 1)[0].blocks[0].own_own
    nz32 = Ao.nzval.astype(F32)
    x = orc.hash_x(np.arange(1, n ** 3 + 1)).astype(F32)
    K = orc.oracle_c()
    want = K.spmv_csr_f32(np.zeros(Ao.m, F32), x, Ao.rowptr, Ao.colval, nz32)
    xd = pa.DeviceVector32(Ao.n).upload(x)
    got = {}
    for pell in ("1", "0"):
        os.environ["PA_SPMV_PELL"] = pell
        try:
            A = pa.DeviceCSR32(Ao.m, Ao.n, Ao.rowptr, Ao.colval, nz32)
        finally:
            os.environ.pop("PA_SPMV_PELL", None)
        assert A.info()["pattern_ell"] == (pell == "1"), A.info()
        y = pa.DeviceVector32(Ao.m)
        pa.spmv32_(y, A, xd)
        got[pell] = y.download()
        assert np.array_equal(got[pell], want), pell
        y0 = (x[:Ao.m] * F32(3)).astype(F32)
        y.upload(y0)
        pa.spmv32_(y, A, xd, alpha=0.5, beta=-1.25)
        assert np.array_equal(y.download(), K.mul5_csr_f32(y0.copy(), x, Ao.rowptr, Ao.colval, nz32, 0.5, -1.25)), pell
