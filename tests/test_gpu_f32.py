"""Float32 blocks and vectors (csrc/pa_f32.hip): the first widening beyond the path's FP64 scope (VERDICT r05 "Next" #7).

The reference's local loops are generic in the element type; its own test runs them on a fixed 7 x 6 matrix as
SparseMatrixCSC{Float32,Int32}, SparseMatrixCSR{0,Float32,Int32} and SparseMatrixCSR{1,Float32,Int32}
(test/sparse_utils_tests.jl:10-45,72-79: spmv! against the library mul!, spmtv! against mul!(transpose)).  Here pa_spmv32 against the
oracle's float loops (oracle/pa_oracle.c: orc_spmv_csr_f32 / orc_spmv_csc_f32 / orc_mul5_csr_f32, every product and sum rounded to
float, -ffp-contract=off), bit for bit: that matrix in all three storages (and its hand-worked product), irregular and empty rows,
the alpha/beta form, and the 27-point operator on the pattern-ELL structure with the 4-byte value stream.
"""
import ctypes as C

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


def test_consistent_and_assemble_of_float32_local_values(orc):
    """consistent! / assemble! (src/p_vector.jl:747-755, 695-708) of a PVector{Vector{Float32}}: pa_exchange_pack32 -> device-to-device
    slice copies -> pa_exchange_finish32 on the plans of the index partition, 27 parts of the 27-point operator's column partition (the
    middle part has 26 neighbours: messages of n^2, n and 1 values).  Against the oracle's assemble_impl! on float32 arrays: ghosts equal
    their owners after consistent!, owners hold the Float32 sums in ascending p and every ghost is zero after assemble!, bit for bit; a
    Float64 exchange on the same plans before, between and after keeps its bits (the buffers are shared)."""
    from gpu_helpers import ranks, upload
    n = 4
    A, _ = pa.build_p_matrix(ranks(27), n, n, n, 3 * n, 3 * n, 3 * n, 3, 3, 3)
    Ao, _, _ = orc.hpcg_build_p_matrix(n, n, n, 3, 3, 3)
    cols = A.col_partition
    cache = pa.pzeros(cols).cache
    host = [(orc.hash_x(c.local_to_global + 3) * 1000.0).astype(np.float32) for c in Ao.cols]
    vecs = pa.pmap(lambda i, h: pa.DeviceVector32(i.n_own, i.n_ghost).upload(h), cols, pa.DebugArray(host))
    want = [h.copy() for h in host]
    for c, w in zip(Ao.cols, want):
        w[c.ghost_to_local - 1] = np.float32(-7.0)                        # ghosts start wrong on both sides
    for v, w in zip(vecs.items, want):
        v.upload(w)
    h64 = [orc.hash_x(c.local_to_global + 11) for c in Ao.cols]
    v64 = upload([h.copy() for h in h64], cols)
    pa.assemble_(v64).wait()
    pa.consistent32_(vecs, cache).wait()
    orc.consistent(want, Ao.cols)
    for v, w in zip(vecs.items, want):
        got = v.download()
        assert got.dtype == np.float32 and np.array_equal(got, w)
    # assemble!: ghosts carry contributions
    contrib = [(orc.hash_x(c.local_to_global + 5) * 3.0 + 0.1).astype(np.float32) for c in Ao.cols]
    for v, w in zip(vecs.items, contrib):
        v.upload(w)
    pa.consistent_(v64).wait()
    pa.assemble32_(vecs, cache).wait()
    want = [w.copy() for w in contrib]
    orc.assemble(want, Ao.cols)
    for v, w, c in zip(vecs.items, want, Ao.cols):
        got = v.download()
        assert np.array_equal(got, w) and np.all(got[c.ghost_to_local - 1] == 0)
    orc.assemble(h64, Ao.cols); orc.consistent(h64, Ao.cols)
    for a_, b_ in zip(v64.local_values().items, h64):
        assert np.array_equal(a_, b_)
    # a Float32 payload must be finished as Float32
    plans = cache.plans.items
    L.call("pa_exchange_pack32", plans[0], vecs.items[0].h, L.CONSISTENT)
    with pytest.raises(L.PAError, match="Float32"):
        L.call("pa_exchange_finish", plans[0], v64.vector_partition.items[0].h, L.CONSISTENT)
    for p, v in zip(plans[1:], vecs.items[1:]):
        L.call("pa_exchange_pack32", p, v.h, L.CONSISTENT)
    arr = (C.c_void_p * len(plans))(*[p.value for p in plans])
    L.call("pa_exchange_local", arr, len(plans), L.CONSISTENT)
    for p, v in zip(plans, vecs.items):
        L.call("pa_exchange_finish32", p, v.h, L.CONSISTENT)


def test_float32_product_of_a_partitioned_matrix_through_the_float32_exchange(orc):
    """mul!(c,A,b) (src/p_sparse_matrix.jl:2090-2103) in Float32 composed from its parts: consistent!(b) on Float32 values, own x own
    on pa_spmv32, own x ghost added on pa_spmv32(beta = 1) -- 8 parts of the 27-point operator, against the oracle's float loops."""
    from gpu_helpers import ranks
    n = 6
    A, _ = pa.build_p_matrix(ranks(8), n, n, n, 2 * n, 2 * n, 2 * n, 2, 2, 2, keep_host=True)
    Ao, _, _ = orc.hpcg_build_p_matrix(n, n, n, 2, 2, 2)
    cols = A.col_partition
    cache = pa.pzeros(cols).cache
    K = orc.oracle_c()
    xs = [(orc.hash_x(c.local_to_global) * (c.local_to_owner == c.part)).astype(np.float32) for c in Ao.cols]
    xd = pa.pmap(lambda i, h: pa.DeviceVector32(i.n_own, i.n_ghost).upload(h), cols, pa.DebugArray([h.copy() for h in xs]))
    pa.consistent32_(xd, cache).wait()
    orc.consistent(xs, Ao.cols)
    for blk, xv, xo, r, c in zip(Ao.blocks, xd.items, xs, Ao.rows, Ao.cols):
        oo, oh = blk.own_own, blk.own_ghost
        Aoo = pa.DeviceCSR32(oo.m, oo.n, oo.rowptr, oo.colval, oo.nzval.astype(np.float32))
        Aoh = pa.DeviceCSR32(oh.m, oh.n, oh.rowptr, oh.colval, oh.nzval.astype(np.float32))
        y = pa.DeviceVector32(r.n_own, 0)
        pa.spmv32_(y, Aoo, xv, L.SEG_OWN, L.SEG_OWN)
        pa.spmv32_(y, Aoh, xv, L.SEG_GHOST, L.SEG_OWN, 1.0, 1.0)
        xo_own, xo_gh = np.ascontiguousarray(xo[c.own_to_local - 1]), np.ascontiguousarray(xo[c.ghost_to_local - 1])
        want = np.zeros(r.n_own, np.float32)
        K.spmv_csr_f32(want, xo_own, oo.rowptr, oo.colval, oo.nzval.astype(np.float32))
        K.mul5_csr_f32(want, xo_gh, oh.rowptr, oh.colval, oh.nzval.astype(np.float32), 1.0, 1.0)
        assert np.array_equal(y.download(), want) and np.any(want != 0)


def test_integer_payloads_travel_through_the_same_plans(orc):
    """exchange! is payload-agnostic (src/primitives.jl:1020-1042) and the reference exchanges integers at set-up (global ids, owners:
    src/p_range.jl:436-531).  pa_exchange_pack_raw / _finish_raw on the plans of the index partition: consistent! of every part's Int64
    global ids (own ids right, ghost ids wrong on purpose) must give each ghost its owner's id = local_to_global itself; assemble!(+) of
    Int32 ones counts, per own value, the parts that ghost it (+ 1), against the oracle's assemble_impl! on integer arrays."""
    from gpu_helpers import ranks
    n = 4
    A, _ = pa.build_p_matrix(ranks(27), n, n, n, 3 * n, 3 * n, 3 * n, 3, 3, 3)
    Ao, _, _ = orc.hpcg_build_p_matrix(n, n, n, 3, 3, 3)
    cols = A.col_partition
    cache = pa.pzeros(cols).cache
    gids = [c.local_to_global.astype(np.int64) for c in Ao.cols]
    wrong = [g.copy() for g in gids]
    for c, w in zip(Ao.cols, wrong):
        w[c.ghost_to_local - 1] = -1
    vecs = pa.pmap(lambda i, h: pa.DeviceVector(i.n_own, i.n_ghost).upload(h.view(np.float64)), cols, pa.DebugArray(wrong))
    pa.exchange_raw_(L.CONSISTENT, vecs, cache, "i64").wait()
    for v, g in zip(vecs.items, gids):
        assert np.array_equal(v.download().view(np.int64), g)
    ones = [np.ones(c.n_local, np.int32) for c in Ao.cols]
    v32 = pa.pmap(lambda i, h: pa.DeviceVector32(i.n_own, i.n_ghost).upload(h.view(np.float32)), cols, pa.DebugArray([o.copy() for o in ones]))
    pa.exchange_raw_(L.ASSEMBLE, v32, cache, "i32").wait()
    orc.assemble(ones, Ao.cols)
    for v, w in zip(v32.items, ones):
        got = v.download().view(np.int32)
        assert np.array_equal(got, w)
    assert max(int(w.max()) for w in ones) == 8              # a corner node of a part is ghosted by the 7 parts around it
