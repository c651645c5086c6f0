"""SURVEY 8(a) row a25 and BASELINE config 4's loop: dot / norm / broadcasts (src/p_vector.jl:1189-1277) and the CG loops of HPCG/src/ref_cg.jl.
Bars: np.array_equal for everything but dot / norm (1e-13).  Needs a real MI355X (-m gpu)."""
import pytest

from gpu_helpers import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


def test_dot_norm_axpby(orc):
    parts = pa.uniform_partition(ranks(3), (3,), (100003,))
    oparts = orc.uniform_partition((3,), (100003,))
    xo = [orc.hash_x(o.local_to_global) - 0.5 for o in oparts]
    yo = [orc.hash_x(o.local_to_global + 3) for o in oparts]
    x, y = upload([v.copy() for v in xo], parts), upload([v.copy() for v in yo], parts)
    d, dref = pa.dot(x, y), orc.dot(xo, yo, oparts)
    assert abs(d - dref) <= 1e-13 * abs(dref) * 10 + 1e-13 * sum(float(np.abs(a * b).sum()) for a, b in zip(xo, yo))
    assert abs(pa.norm(x) - orc.norm2(xo, oparts)) <= 1e-13 * orc.norm2(xo, oparts)
    pa.axpby_(y, 0.25, x, -2.0)
    for got, a, b in zip(y.local_values().items, xo, yo):
        assert np.array_equal(got, 0.25 * a + -2.0 * b)


def test_ref_cg_identity_preconditioner(orc):
    """HPCG/src/ref_cg.jl with Pl = Identity(): consistent!+mul!, 2 dots + norm, 3 axpys per iteration, on 8 parts.
    dot() reassociates, so the trajectory is compared within 1e-10 relative; the solve itself must converge to x = 1
    (b = A*1 by construction, HPCG/src/sparse_matrix.jl:75)."""
    A, b = pa.build_p_matrix(ranks(8), 8, 8, 8, 16, 16, 16, 2, 2, 2)
    Ao, bo, _ = orc.hpcg_build_p_matrix(8, 8, 8, 2, 2, 2)
    for overlap in (True, False):
        x = pa.pzeros(A.col_partition)
        hist = []
        x, r0, r, it = pa.ref_cg_(x, A, b, maxiter=25, overlap=overlap, history=hist)
        ho = []
        xo, r0o, ro, ito = orc.ref_cg([np.zeros(c.n_local) for c in Ao.cols], Ao, [v.copy() for v in bo], maxiter=25, history=ho)
        assert it == ito == 25 and abs(r0 - r0o) <= 1e-13 * r0o
        assert np.allclose(hist, ho, rtol=1e-9, atol=1e-14 * r0o)
        assert r / r0 < 1e-8
        for vals, ind in zip(x.own_values().items, A.col_partition.items):
            assert np.allclose(vals, 1.0, atol=1e-8)


@pytest.mark.parametrize("P,np3,with_mg", [(1, (1, 1, 1), False), (8, (2, 2, 2), False), (4, (2, 2, 1), True)])
def test_opt_cg_device_scalars_bit_identical_to_ref_cg(P, np3, with_mg):
    """opt_cg_(fuse=False) keeps rho, u'c and |r|^2 in device slots and fuses ref_cg.jl:64-67 into one pass; the
    arithmetic and the reduction trees are those of ref_cg_, so residual history and solution must be bit-identical.
    opt_cg_ as it runs by default (fuse=True: u'c accumulated inside the product kernels, x's update deferred into u's
    pass) sums u'c in another order: history and solution agree to rounding -- rtol 1e-9 on the residual history over 12
    iterations is the stated bar, the measured drift is ~1e-14."""
    n = (16, 16, 16)
    if with_mg:
        S = pa.pc_setup(ranks(P), P, 3, *n, ordering="multicolor_spmv")
        A, b = S.A_vec[-1], S.r[-1]
    else:
        S = None
        A, b = pa.build_p_matrix(ranks(P), *n, *(a * q for a, q in zip(n, np3)), *np3)
    out = []
    unfused = functools.partial(pa.opt_cg_, fuse=False)
    for fn in (pa.ref_cg_, unfused, pa.opt_cg_):
        x = pa.pzeros(A.col_partition)
        hist = []
        x, r0, r, it = fn(x, A, b, maxiter=12, history=hist, Pl=S)
        out.append((r0, r, it, hist, [v.copy() for v in x.own_values().items]))
    (r0a, ra, ita, ha, xa), (r0b, rb, itb, hb, xb), (r0c, rc, itc, hc, xc) = out
    assert (r0a, ra, ita) == (r0b, rb, itb) and ha == hb
    for u, v in zip(xa, xb):
        assert np.array_equal(u, v)
    assert r0c == r0a and itc == ita and np.allclose(hc, ha, rtol=1e-9, atol=0)
    drift = max(abs(p - q) / q for p, q in zip(hc, ha))
    assert drift < 1e-11, drift                             # (what is measured; the bar above is what is promised)
    scale = max(float(np.abs(u).max()) for u in xa)
    for u, v in zip(xa, xc):
        assert np.abs(u - v).max() <= 1e-11 * scale
    # without a history the host reads nothing inside the loop; the end state is the same
    for fn, want in ((unfused, (r0a, ra, ita)), (pa.opt_cg_, (r0c, rc, itc))):
        x, r0, r, it = fn(pa.pzeros(A.col_partition), A, b, maxiter=12, Pl=S)
        assert (r0, r, it) == want
    # tolerance > 0: stops at the same iteration as the reference loop
    xa_, r0a_, ra_, ita_ = pa.ref_cg_(pa.pzeros(A.col_partition), A, b, maxiter=200, tolerance=1e-6, Pl=S)
    xb_, r0b_, rb_, itb_ = unfused(pa.pzeros(A.col_partition), A, b, maxiter=200, tolerance=1e-6, Pl=S)
    assert (ita_, ra_) == (itb_, rb_) and ita_ < 200
    xc_, r0c_, rc_, itc_ = pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=200, tolerance=1e-6, Pl=S, fuse=True)
    assert itc_ == ita_ and abs(rc_ - ra_) <= 1e-9 * ra_


def test_cg_on_an_unstructured_banded_spd_matrix_four_parts(orc):
    """The CG loops on a PSparseMatrix without any structure: a symmetric, diagonally dominant matrix with 6..20 random
    couplings per row inside a band of +-1500, on 4 parts (irregular ghosts on both sides of every part boundary).  The own
    x own blocks run on the x-window launches, the fused loop on their dot variant.  ref_cg_ on the device equals the
    oracle's ref_cg (the reference loop on the host) to rounding of the reductions; opt_cg_(fuse=False) equals ref_cg_ bit
    for bit; opt_cg_ (fused) to 1e-9 on the residual history; all three converge to the solution the matrix was built for."""
    P, n = 4, 240_000
    rows = pa.uniform_partition(ranks(P), n)
    orows = orc.uniform_partition(P, n)
    rng = np.random.default_rng(53)
    k = rng.integers(3, 11, n)                                      # couplings (i, j > i) generated from the lower index
    i0 = np.repeat(np.arange(1, n + 1), k)
    j0 = i0 + rng.integers(1, 1500, len(i0))
    keep = j0 <= n                                                  # (clipping to n would give row n thousands of entries)
    i0, j0 = i0[keep], j0[keep]
    v0 = -rng.random(len(i0)) - 0.1
    diag = np.zeros(n + 1)
    np.add.at(diag, i0, -v0)
    np.add.at(diag, j0, -v0)
    I = np.concatenate([i0, j0, np.arange(1, n + 1)])
    J = np.concatenate([j0, i0, np.arange(1, n + 1)])
    V = np.concatenate([v0, v0, 2.0 * diag[1:] + 1.0])            # strictly dominant diagonal: the residual falls steadily
    order = np.lexsort((J, I))
    I, J, V = I[order], J[order], V[order]
    Is, Js, Vs = [], [], []
    for ind in orows:
        lo, hi = ind.own_to_global[0], ind.own_to_global[-1]
        sel = (I >= lo) & (I <= hi)
        Is.append(I[sel].astype(np.int64)); Js.append(J[sel].astype(np.int64)); Vs.append(V[sel].copy())
    A = pa.psparse_from_coo(pa.DebugArray([a.copy() for a in Is]), pa.DebugArray([a.copy() for a in Js]),
                            pa.DebugArray([a.copy() for a in Vs]), rows)
    for blk in A.matrix_partition.items:
        assert blk.own_own.xwin()["groups"] > 0 and blk.own_ghost.nnz > 0
    xs = pa.pvector_from_function(lambda ind: np.cos(0.001 * ind.get_local_to_global()) * (ind.get_local_to_owner() == ind.part),
                                  A.col_partition)
    b = pa.pzeros(A.col_partition)
    pa.mul_(b, A, xs)
    out = []
    for fn in (pa.ref_cg_, functools.partial(pa.opt_cg_, fuse=False), pa.opt_cg_):
        hist = []
        x, r0, r, it = fn(pa.pzeros(A.col_partition), A, b, maxiter=60, tolerance=1e-8, history=hist)
        assert r / r0 <= 1e-8 and it < 60
        for got, want in zip(x.own_values().items, xs.own_values().items):
            assert np.abs(got - want).max() <= 1e-5
        out.append((r0, r, it, hist, [v.copy() for v in x.own_values().items]))
    (r0a, ra, ita, ha, xa), (r0b, rb, itb, hb, xb), (r0c, rc, itc, hc, xc) = out
    assert (r0a, ra, ita) == (r0b, rb, itb) and ha == hb
    for u, v in zip(xa, xb):
        assert np.array_equal(u, v)
    assert itc == ita and np.allclose(hc, ha, rtol=1e-9, atol=0)
    # (on a matrix where CG's residual norm peaks -- the same construction with V = diag + 1 and the couplings clipped into
    # row n -- a peak amplifies the rounding difference of the fused u'c to percents for an iteration or two, in any pair of loops that
    # round differently; the histories meet again to 1e-14 after each peak.  tools/probe/cg_unstructured_debug.py)
    # the oracle's loop on the host: same iteration count, history to the rounding of the (differently ordered) reductions
    Ao = orc.psparse_from_coo([a.copy() for a in Is], [a.copy() for a in Js], [a.copy() for a in Vs], orows)
    bo = [np.zeros(c.n_local) for c in Ao.cols]
    for dst, src, c in zip(bo, b.own_values().items, Ao.cols):
        dst[:c.n_own] = src
    ho = []
    xo, r0o, ro, ito = orc.ref_cg([np.zeros(c.n_local) for c in Ao.cols], Ao, bo, maxiter=60, tolerance=1e-8, history=ho,
                                  mv=orc.mul)
    assert ito == ita and np.allclose(ho, ha, rtol=1e-8, atol=0)


def test_cg_with_reused_work_vectors_is_bit_identical():
    """cg_work: the work vectors allocated once -- ref_cg_ and opt_cg_ give the bits of the allocating loops, solve
    after solve."""
    A, b = pa.build_p_matrix(ranks(2), 96, 96, 64, 192, 96, 64, 2, 1, 1)         # 2 x 590k rows, 15.7 M entries per part
    opt = functools.partial(pa.opt_cg_, fuse=False)          # (the variant that shares ref_cg_'s bits)
    x0, r00, r0, it0 = opt(pa.pzeros(A.col_partition), A, b, maxiter=9)
    want = [v.copy() for v in x0.own_values().items]
    work = pa.cg_work(pa.pzeros(A.col_partition), b, A)
    for fn in (opt, pa.ref_cg_, opt):
        x, r0_, r_, it = fn(pa.pzeros(A.col_partition), A, b, maxiter=9, work=work)
        assert it == it0 == 9
        for g, e in zip(x.own_values().items, want):
            assert np.array_equal(g, e)
    assert (r0_, r_) == (r00, r0)
    # the fused loop: the same bits solve after solve on reused work vectors
    first = None
    for _ in range(2):
        x, r0_, r_, it = pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=9, work=work, fuse=True)
        got = (r0_, r_, [v.copy() for v in x.own_values().items])
        if first is None:
            first = got
        assert got[:2] == first[:2] and all(np.array_equal(u, v) for u, v in zip(got[2], first[2]))
    assert abs(first[1] - r0) <= 1e-11 * r0


def test_fused_product_and_dot_matches_the_separate_calls(orc):
    """pa_mul_dot / pa_mul_all_dot: c is bit-identical to mul!'s and the slot holds dot(b,c) to rounding -- on one part,
    on 8 parts with ghosts (own x ghost contributes its own products), on a matrix with rows longer than a chunk and
    with a chunk of more than 64 rows (the cross-wavefront path of the reduction)."""
    import pa_amd.p_sparse_matrix as psm
    for P, np3, n in ((1, (1, 1, 1), (24, 20, 16)), (8, (2, 2, 2), (12, 10, 8))):
        A, b = pa.build_p_matrix(ranks(P), *n, *(a * q for a, q in zip(n, np3)), *np3)
        u = pa.pvector_from_function(lambda i: orc.hash_x(i.get_local_to_global()) * (i.get_local_to_owner() == i.part) - 0.3, A.col_partition)
        c1, c2 = pa.pzeros(A.col_partition), pa.pzeros(A.col_partition)
        pa.mul_c_(c1, A, u)
        want = pa.dot(u, c1)
        assert psm.mul_dot_(c2, A, u, 6)
        got = pa.read_slots(6)[0]
        for g, e in zip(c2.own_values().items, c1.own_values().items):
            assert np.array_equal(g, e)
        assert abs(got - want) <= 1e-13 * abs(want), (got, want)
        assert psm.mul_dot_(c2, A, u, 6) and pa.read_slots(6)[0] == got          # deterministic
    # rows of 1 entry (hundreds of rows per chunk), of 3000 entries (longer than a chunk), empty rows
    rng = np.random.default_rng(11)
    lens = np.concatenate([np.ones(700, int), [3000, 0, 0, 5, 2000], rng.integers(0, 40, 2600)])
    m = len(lens)
    H = _random_csr(rng, m, m, lens)
    blk = pa.DeviceCSR(H)
    ind = pa.uniform_partition(ranks(1), m)
    import pa_amd.p_sparse_matrix as psm2
    empty = pa.DeviceCSR(pa.HostCSR(m, 0, np.ones(m + 1, np.int32), np.zeros(0, np.int32), np.zeros(0)))
    Ah = pa.PSparseMatrix(pa.DebugArray([psm2.SplitMatrixBlocks(blk, empty)]), ind, ind, True)
    u = pa.pvector_from_function(lambda i: rng.standard_normal(m), ind)
    c1, c2 = pa.pzeros(ind), pa.pzeros(ind)
    pa.mul_c_(c1, Ah, u)
    assert psm.mul_dot_(c2, Ah, u, 7)
    assert np.array_equal(c2.own_values().items[0], c1.own_values().items[0])
    want = pa.dot(u, c1)
    assert abs(pa.read_slots(7)[0] - want) <= 1e-12 * max(1.0, abs(want))


@pytest.mark.parametrize("m,band,tier", [(150_000, 1200, "40 KiB"), (150_000, 3000, "96 KiB"), (800_000, 6500, "128 KiB"),
                                         (150_000, 3000, "ring only"), (800_000, 7800, "ring")])
def test_fused_product_and_dot_on_banded_rows_is_the_same_on_both_launches(monkeypatch, m, band, tier):
    """Banded rows without a pattern: pa_mul_dot through k_spmv_xwin / k_spmv_xring (+ the chunk list) and through
    k_spmv_rowsplit alone give the same c AND the same dot, bit for bit (the per-chunk partial sums are formed in one order
    on all of them), with chunks of more than 64 rows (short rows) and of fewer -- on each of the three window sizes and on
    the sliding window."""
    monkeypatch.setenv("PA_SPMV_XRING", {"ring": "1", "ring only": "2"}.get(tier, "0"))
    import pa_amd.p_sparse_matrix as psm
    rng = np.random.default_rng(23)
    lens = np.where(np.arange(m) < m // 3, rng.integers(1, 6, m), rng.integers(10, 40, m))
    rp = np.concatenate([[1], 1 + np.cumsum(lens)]).astype(np.int32)
    rows = np.repeat(np.arange(m), lens)
    col = np.clip(rows + rng.integers(-band, band, size=len(rows)), 0, m - 1)
    col[rng.choice(len(rows), 30, replace=False)] = rng.integers(0, m, 30)
    order = np.lexsort((col, rows))
    H = pa.HostCSR(m, m, rp, (col[order] + 1).astype(np.int32), rng.standard_normal(len(rows)))
    ind = pa.uniform_partition(ranks(1), m)
    uh = rng.standard_normal(m)
    outs = []
    for switch in ("1", "0"):
        monkeypatch.setenv("PA_SPMV_XWIN", switch)
        blk = pa.DeviceCSR(H)
        xw = blk.xwin()
        assert (xw["groups"] > 0) == (switch == "1"), xw
        if switch == "1" and tier == "ring only":
            assert xw["ring_groups"] > 0.5 * xw["groups"], (tier, xw)
        elif switch == "1" and tier == "ring":
            assert xw["groups"] > 0, (tier, xw)              # (the windows first; the ring takes what they leave, if it pays)
        elif switch == "1":
            assert (xw["big_groups"] > 0.5 * xw["groups"]) == (tier != "40 KiB") and xw["ring_groups"] == 0, (tier, xw)
        empty = pa.DeviceCSR(pa.HostCSR(m, 0, np.ones(m + 1, np.int32), np.zeros(0, np.int32), np.zeros(0)))
        Ah = pa.PSparseMatrix(pa.DebugArray([psm.SplitMatrixBlocks(blk, empty)]), ind, ind, True)
        u = pa.pvector_from_function(lambda i: uh, ind)
        c1, c2 = pa.pzeros(ind), pa.pzeros(ind)
        pa.mul_c_(c1, Ah, u)
        assert psm.mul_dot_(c2, Ah, u, 7)
        assert np.array_equal(c2.own_values().items[0], c1.own_values().items[0])
        outs.append((c2.own_values().items[0].copy(), pa.read_slots(7)[0]))
        want = float(uh @ outs[-1][0])
        assert abs(outs[-1][1] - want) <= 1e-12 * max(1.0, abs(want))
    assert np.array_equal(outs[0][0], outs[1][0]) and outs[0][1] == outs[1][1]


@pytest.mark.parametrize("P,np3", [(1, (1, 1, 1)), (4, (2, 2, 1))])      # 4 parts: graph mode declines, eager loop runs
def test_opt_cg_replayed_from_a_hipgraph_is_bit_identical(P, np3):
    """graph=True records three CG iterations (kernels of the exchange, both SpMV blocks, the slot BLAS-1) into a
    hipGraph and replays it; 14 iterations = 4 replays + 2 eager iterations must give the bits of the eager loop."""
    n = (16, 12, 8)
    A, b = pa.build_p_matrix(ranks(P), *n, *(a * q for a, q in zip(n, np3)), *np3)
    outs = []
    for graph in (False, True):
        x, r0, r, it = pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=14, graph=graph, fuse=True)
        outs.append((r0, r, it, [v.copy() for v in x.own_values().items]))
    assert outs[0][:3] == outs[1][:3] and outs[0][2] == 14
    for u, v in zip(outs[0][3], outs[1][3]):
        assert np.array_equal(u, v)
    import pa_amd._lib as L
    with pytest.raises(L.PAError):                      # a second capture on the same context is refused
        with pa.Graph():
            L.call("pa_graph_begin", pa.context().h)
    y = pa.pzeros(A.row_partition)
    pa.mul_(y, A, pa.pones(A.col_partition))            # the context is usable after the refused capture
    assert all(np.array_equal(g, e) for g, e in zip(y.own_values().items, b.own_values().items))


def test_slot_api_errors_and_values():
    ctx = pa.context()
    pa.write_slot(5, 2.5)
    pa.write_slot(6, -4.0)
    assert pa.read_slots(5, 2) == [2.5, -4.0]
    g = pa.uniform_partition(ranks(1), (1,), (1000,))
    x = pa.pvector_from_function(lambda i: np.arange(1, i.n_local + 1, dtype=np.float64), g)
    y = pa.pones(g)
    pa.dot_slot(x, y, 7)
    assert pa.read_slots(7)[0] == 500500.0
    pa.axpby_slot_(y, 1.0, 5, 6, x, -2.0, pa._lib.SLOT_ONE, 5)            # y = (2.5/-4) x + (-2/2.5) y
    want = (2.5 / -4.0) * np.arange(1, 1001) + (-2.0 / 2.5) * 1.0
    assert np.array_equal(y.own_values().items[0], want)
    with pytest.raises(pa._lib.PAError):
        pa.write_slot(16, 1.0)
    with pytest.raises(pa._lib.PAError):
        pa.dot_slot(x, y, -1)
    with pytest.raises(pa._lib.PAError):
        pa._lib.call("pa_cg_update", x.vector_partition.items[0].h, y.vector_partition.items[0].h,
                     x.vector_partition.items[0].h, y.vector_partition.items[0].h, 1, 2, 1, 0)


def test_cg_with_a_zero_right_hand_side_behaves_like_julia(orc):
    """`residual/residual0 <= tolerance` with residual0 == 0 is 0/0 = NaN in Julia -- false, so the loop runs to
    maxiter on NaNs -- where python raised ZeroDivisionError (and took the job down under with_torchdist)."""
    A, b = pa.build_p_matrix(ranks(1), 8, 8, 8, 8, 8, 8, 1, 1, 1)
    zero = pa.pzeros(A.col_partition)
    for fn in (pa.ref_cg_, pa.opt_cg_):
        x, r0, r, it = fn(pa.pzeros(A.col_partition), A, zero, maxiter=4, tolerance=1e-6)
        assert it == 4 and r0 == 0.0 and r != r                       # NaN residual, all iterations done
    Ao, bo, _ = orc.hpcg_build_p_matrix(8, 8, 8, 1, 1, 1)
    xo, r0o, ro, ito = orc.ref_cg([np.zeros(c.n_local) for c in Ao.cols], Ao, [np.zeros_like(v) for v in bo], maxiter=4, tolerance=1e-6)
    assert ito == 4 and r0o == 0.0 and ro != ro
