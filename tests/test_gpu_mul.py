"""mul!(c,a,b[,alpha,beta]) (src/p_sparse_matrix.jl:2090-2142) and its variants against the oracle.
Bars: np.array_equal for everything but dot / norm (1e-13).  Needs a real MI355X (-m gpu)."""
import pytest

from gpu_helpers import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,parts", [((4, 4, 4), (2, 2, 2)), ((8, 8, 8), (2, 2, 1)), ((16, 8, 4), (2, 1, 1)),
                                     ((16, 16, 16), (1, 1, 1)), ((3, 5, 7), (2, 2, 2))])
def test_mul_hpcg_bit_exact(orc, n, parts):
    nx, ny, nz = n
    px, py, pz = parts
    P = px * py * pz
    A, b = pa.build_p_matrix(ranks(P), nx, ny, nz, px * nx, py * ny, pz * nz, px, py, pz)
    Ao, bo, _ = orc.hpcg_build_p_matrix(nx, ny, nz, px, py, pz)
    # G12: A*1 == b exactly
    y = pa.pzeros(A.row_partition)
    pa.mul_(y, A, pa.pones(A.col_partition))
    for got, exp in zip(y.own_values().items, b.own_values().items):
        assert np.array_equal(got, exp)
    # general x: only own values set; consistent! inside mul! must fill the ghosts
    xo = [orc.hash_x(c.local_to_global) * (c.local_to_owner == c.part) for c in Ao.cols]
    x = upload([v.copy() for v in xo], A.col_partition)
    pa.mul_(y, A, x)
    yo = _oracle_mul(orc, Ao, xo)
    for got, exp, r in zip(y.own_values().items, yo, Ao.rows):
        assert np.array_equal(got, exp[:r.n_own])
    # the ghosts of x are now consistent, bit for bit
    orc.consistent(xo, Ao.cols)
    for got, exp in zip(x.local_values().items, xo):
        assert np.array_equal(got, exp)
    # no-overlap ordering gives the same bits
    y2 = pa.pzeros(A.row_partition)
    pa.mul_no_overlap_(y2, A, x)
    for a_, b_ in zip(y.own_values().items, y2.own_values().items):
        assert np.array_equal(a_, b_)


def test_mul_diag_golden(golden):
    c = golden["mul_diag"]                                   # test/p_sparse_matrix_tests.jl:207-248
    rows = pa.uniform_partition(ranks(4), tuple(c["np"]), tuple(c["n"]))
    I = pa.pmap(lambda r: r.own_to_global.copy(), rows)
    V = pa.pmap(lambda i: np.full(len(i), c["diag"]), I)
    A = pa.psparse_from_coo(I, pa.pmap(lambda i: i.copy(), I), V, rows, keep_host=True)
    x = pa.pfill(c["x"], A.col_partition)
    b = pa.pzeros(A.row_partition)
    pa.mul_(b, A, x)
    for v in b.own_values().items:
        assert np.all(v == c["y"])
    pa.consistent_(b).wait()
    for v in b.local_values().items:
        assert np.all(v == c["y"])
    # fillstored!(A,1): :285-291
    pa.pmap(lambda blk, h: blk.own_own.update_values(np.full(h[0].nnz, c["fillstored"])), A.matrix_partition, A.host_blocks)
    pa.mul_(b, A, x)
    for v in b.own_values().items:
        assert np.all(v == c["y_fillstored"])


def test_mul5_alpha_beta(orc):
    A, _ = pa.build_p_matrix(ranks(2), 6, 5, 4, 12, 5, 4, 2, 1, 1)
    Ao, _, _ = orc.hpcg_build_p_matrix(6, 5, 4, 2, 1, 1)
    for alpha, beta in [(1.0, 0.0), (1.0, 1.0), (-0.75, 0.5), (2.0, 0.0), (0.3, -1.25)]:
        xo = [orc.hash_x(c.local_to_global) * (c.local_to_owner == c.part) for c in Ao.cols]
        yo = [orc.hash_x(r.local_to_global + 11) for r in Ao.rows]
        x = upload([v.copy() for v in xo], A.col_partition)
        y = upload([v.copy() for v in yo], A.row_partition)
        pa.mul5_(y, A, x, alpha, beta)
        orc.mul5(yo, Ao, xo, alpha, beta)
        for got, exp, r in zip(y.own_values().items, yo, Ao.rows):
            assert np.array_equal(got, exp[:r.n_own]), (alpha, beta)


def test_unsplit_local_product_equals_split_product(orc):
    """K9: HPCG's mul_no_lat! on the unsplit local CSR (one kernel launch, columns [own | ghost]) == mul! on the split
    blocks, bit for bit, 8 parts; and == the oracle's mul_no_lat!."""
    A, b = pa.build_p_matrix(ranks(8), 9, 7, 8, 18, 14, 16, 2, 2, 2, keep_host=True)
    Ao, _, _ = orc.hpcg_build_p_matrix(9, 7, 8, 2, 2, 2)
    xo = [orc.hash_x(c.local_to_global) * (c.local_to_owner == c.part) for c in Ao.cols]
    y1, y2 = pa.pzeros(A.row_partition), pa.pzeros(A.row_partition)
    pa.mul_(y1, A, upload([v.copy() for v in xo], A.col_partition))
    pa.mul_no_lat_unsplit_(y2, A, upload([v.copy() for v in xo], A.col_partition))
    yo = [np.zeros(r.n_local) for r in Ao.rows]
    orc.mul_no_lat(yo, Ao, [v.copy() for v in xo])
    for u, v, w, r in zip(y1.own_values().items, y2.own_values().items, yo, Ao.rows):
        assert np.array_equal(u, v) and np.array_equal(v, w[:r.n_own])
    enc = A._unsplit.items[0].encoding()
    assert sum(enc.values()) > 0 and A._unsplit.items[0].nnz == A.matrix_partition.items[0].own_own.nnz + A.matrix_partition.items[0].own_ghost.nnz


def test_operator_level_mul_equals_composed_mul(orc):
    """pa_mul_all / pa_mul5 (one library call per mul!) queue the kernels of mul_ / mul5_ in the same order: same bits.
    8 parts in one process, and a single part through the one-part-per-process entry point."""
    import pa_amd._lib as L
    for P, shape in ((8, (2, 2, 2)), (1, (1, 1, 1))):
        A, b = pa.build_p_matrix(ranks(P), 8, 6, 10, 8 * shape[0], 6 * shape[1], 10 * shape[2], *shape)
        g = A.col_partition
        xf = lambda i: orc.hash_x(i.get_local_to_global()) * (i.get_local_to_owner() == i.part)
        x1, x2 = pa.pvector_from_function(xf, g), pa.pvector_from_function(xf, g)
        y0 = lambda i: np.sin(i.get_local_to_global().astype(float))
        for alpha, beta in ((1.0, 0.0), (-0.75, 2.5)):
            y1, y2 = pa.pvector_from_function(y0, A.row_partition), pa.pvector_from_function(y0, A.row_partition)
            pa.mul5_(y1, A, x1, alpha, beta)
            pa.mul_c_(y2, A, x2, alpha, beta)
            for u, v in zip(y1.own_values().items, y2.own_values().items):
                assert np.array_equal(u, v)
            for u, v in zip(x1.ghost_values().items, x2.ghost_values().items):
                assert np.array_equal(u, v)
        if P == 1:                                    # the per-process entry point, no communicator needed
            blk, xv, yv = A.matrix_partition.items[0], x2.vector_partition.items[0], y2.vector_partition.items[0]
            m = C.c_void_p()
            L.call("pa_matrix_create", pa.context().h, blk.own_own.h, blk.own_ghost.h, x2.cache.plans.items[0], C.byref(m))
            L.call("pa_mul", m, None, yv.h, xv.h)
            pa.mul_(y1, A, x1)
            assert np.array_equal(y1.own_values().items[0], yv.own())
            with pytest.raises(L.PAError):            # matching_own_indices
                L.call("pa_mul", m, None, pa.DeviceVector(3, 0).h, xv.h)
            with pytest.raises(L.PAError):            # c and b alias
                L.call("pa_mul", m, None, xv.h, xv.h)
            L.call("pa_mul", m, None, yv.h, xv.h)     # the plan is still usable after the refused calls
            L.call("pa_matrix_destroy", m)
        else:
            blk = A.matrix_partition.items[1]
            m = C.c_void_p()
            L.call("pa_matrix_create", pa.context().h, blk.own_own.h, blk.own_ghost.h, x2.cache.plans.items[1], C.byref(m))
            with pytest.raises(L.PAError):            # a part with neighbours needs the communicator
                L.call("pa_mul", m, None, y2.vector_partition.items[1].h, x2.vector_partition.items[1].h)
            L.call("pa_matrix_destroy", m)


def test_config1_laplacian_64_cubed_4_parts(orc):
    """BASELINE config 1: 7-pt 64^3 on (2,2,1) parts: the reference's CPU-runnable case, vs the oracle."""
    n, parts = (64, 64, 64), (2, 2, 1)
    I, J, V, rows, _ = pa.laplacian_fdm(n, parts, ranks(4))
    A = pa.psparse_from_coo(I, J, V, rows)
    Io, Jo, Vo, orows, _ = orc.laplacian_fdm_fast(n, parts)
    Ao = orc.psparse_from_coo(Io, Jo, Vo, orows)
    assert pa.pmap(lambda b: (b.own_own.nnz, b.own_ghost.nnz), A.matrix_partition).items == [(448512, 4096)] * 4
    xo = [orc.hash_x(c.local_to_global) * (c.local_to_owner == c.part) for c in Ao.cols]
    x = upload([v.copy() for v in xo], A.col_partition)
    y = pa.pzeros(A.row_partition)
    pa.mul_(y, A, x)
    yo = _oracle_mul(orc, Ao, xo)
    for got, exp, r in zip(y.own_values().items, yo, Ao.rows):
        assert np.array_equal(got, exp[:r.n_own])
    # A*1 = alpha*(2D - #neighbours) exactly
    pa.mul_(y, A, pa.pones(A.col_partition))
    yo = _oracle_mul(orc, Ao, [np.ones(c.n_local) for c in Ao.cols])
    for got, exp, r in zip(y.own_values().items, yo, Ao.rows):
        assert np.array_equal(got, exp[:r.n_own])


def test_fdm_example_end_to_end():
    """G13: test/fdm_example.jl:11-128 -- 9^3 grid on (2,1,2) parts, 7-point stencil in the interior, identity rows on
    the boundary (a non-symmetric matrix), exact solution u = x + y imposed through the initial guess; CG must reach
    norm(x - x_hat) < 1e-5 on the own values, as the reference asserts (:128)."""
    parts_per_dir, nodes = (2, 1, 2), (9, 9, 9)
    h = 2.0 / (nodes[0] - 1)
    coeffs = np.array([-6, 1, 1, 1, 1, 1, 1]) / h ** 2
    points = [(0, 0, 0), (-1, 0, 0), (1, 0, 0), (0, -1, 0), (0, 1, 0), (0, 0, -1), (0, 0, 1)]
    rows = pa.uniform_partition(ranks(4), parts_per_dir, nodes)

    def cart(g):                                                   # CartesianIndices(nodes)[g], 0-based coordinates
        g = np.asarray(g) - 1
        return np.stack([g % nodes[0], (g // nodes[0]) % nodes[1], g // (nodes[0] * nodes[1])], axis=-1)

    def coo(ind):
        I, J, V = [], [], []
        gl = ind.get_local_to_global()
        b, xh = np.zeros(ind.n_local), np.zeros(ind.n_local)
        for k, (g, c) in enumerate(zip(gl, cart(gl))):
            xh[k] = c[0] * h + c[1] * h
            if any(ci == 0 or ci == n - 1 for ci, n in zip(c, nodes)):
                I.append(g), J.append(g), V.append(1.0)
                b[k] = xh[k]
            else:
                for v, d in zip(coeffs, points):
                    cc = c + np.array(d)
                    I.append(g), J.append(1 + cc[0] + nodes[0] * (cc[1] + nodes[1] * cc[2])), V.append(-v)
        return np.array(I), np.array(J), np.array(V), b, xh

    out = pa.pmap(coo, rows)
    I, J, V = (pa.pmap(lambda o, k=k: o[k], out) for k in range(3))
    A = pa.psparse_from_coo(I, J, V, rows)
    cols = A.col_partition

    def x0(ind):
        c = cart(ind.get_local_to_global())
        bnd = np.any((c == 0) | (c == np.array(nodes) - 1), axis=1)
        v = np.where(bnd, c[:, 0] * h + c[:, 1] * h, 0.0)
        v[ind.n_own:] = 0.0                                        # only own values are set (:104-116)
        return v
    x = pa.pvector_from_function(x0, cols)
    b = pa.pvector_from_function(lambda ind: np.concatenate([out.items[ind.part - 1][3], np.zeros(ind.n_ghost)]), cols)
    x, r0, r, it = pa.ref_cg_(x, A, b, maxiter=729, tolerance=1.4901161193847656e-08)   # IterativeSolvers: sqrt(eps)
    err = sum(float(np.sum((xv - o[4]) ** 2)) for xv, o in zip(x.own_values().items, out.items)) ** 0.5
    assert err < 1.0e-5 and it < 729


def test_empty_part_and_empty_blocks(orc):
    """Edge cases of the containers: a part that owns nothing (variable_partition([5,0,7])), hence empty vectors, 0 x n
    blocks and a plan without neighbours on that part; a matrix with empty rows; an all-zero own_ghost block."""
    n_own = [5, 0, 7]
    rows = pa.variable_partition(ranks(3).__class__(n_own), 12)
    orows = orc.variable_partition(n_own, 12)
    gi = [np.arange(1, 6), np.zeros(0, int), np.arange(6, 13)]
    # tridiagonal, rows 3 and 9 left empty
    I = [np.concatenate([[g] * 3 for g in part if g not in (3, 9)]).astype(np.int64) if len(part) else np.zeros(0, np.int64) for part in gi]
    J = [np.clip(np.concatenate([[g - 1, g, g + 1] for g in part if g not in (3, 9)]), 1, 12).astype(np.int64) if len(part) else np.zeros(0, np.int64) for part in gi]
    V = [np.tile([-1.0, 2.5, -0.75], len(i) // 3) for i in I]
    A = pa.psparse_from_coo(pa.DebugArray(I), pa.DebugArray(J), pa.DebugArray(V), rows)
    Ao = orc.psparse_from_coo(I, J, V, orows)
    assert [(b.own_own.nnz, b.own_ghost.nnz) for b in A.matrix_partition.items] == \
        [(bo.own_own.nnz, bo.own_ghost.nnz) for bo in Ao.blocks] and A.matrix_partition.items[1].own_own.nnz == 0
    xo = [orc.hash_x(c.local_to_global) * (c.local_to_owner == c.part) for c in Ao.cols]
    x = upload([v.copy() for v in xo], A.col_partition)
    y = pa.pvector_from_function(lambda i: np.full(i.n_local, 7.0), A.row_partition)
    pa.mul_(y, A, x)
    yo = _oracle_mul(orc, Ao, xo)
    for got, exp, r in zip(y.own_values().items, yo, Ao.rows):
        assert np.array_equal(got, exp[:r.n_own])
    assert y.own_values().items[0][2] == 0.0 and len(y.own_values().items[1]) == 0        # empty row -> 0, empty part
    pa.mul5_(y, A, x, -2.0, 0.5)
    assert abs(pa.norm(x) - orc.norm2(xo, Ao.cols)) <= 1e-13 * orc.norm2(xo, Ao.cols)
    pa.assemble_(x).wait()
    for vals, c in zip(x.ghost_values().items, Ao.cols):
        assert not vals.any()


def test_c_example_runs_without_python_or_torch():
    """examples/c_abi_smoke.c: two parts of a 1-D Laplacian handed over as the reference stores them, mul! through
    pa_mul_all and a dot, from a plain C program (its own process: no Python, no PyTorch in it)."""
    import subprocess
    from __graft_entry__ import ROOT
    exe = os.path.join(ROOT, "examples", "c_abi_smoke")
    if not os.path.exists(exe):
        import __graft_entry__ as g
        g.build()
    env = dict(os.environ, LD_LIBRARY_PATH=os.path.join(ROOT, "partitionedarrays.jl_amd") + ":/opt/rocm/lib:" + os.environ.get("LD_LIBRARY_PATH", ""))
    r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "c_abi_smoke: OK" in r.stdout, r.stdout + r.stderr


def test_one_call_products_give_the_bits_of_the_composed_mul(orc):
    """mul_c_ (pa_mul_all / pa_mul5) and mul_no_lat_c_ (pa_mul_no_lat; several parts in one process: the composed
    mul_no_overlap_) against mul_ on 1 and 8 parts: same kernels in the same order, so the same bits -- and the ghosts of b
    are those of the owners afterwards in every variant."""
    for P, np3, n in ((1, (1, 1, 1), (20, 16, 12)), (8, (2, 2, 2), (10, 8, 6))):
        A, _b = pa.build_p_matrix(ranks(P), *n, *(a * q for a, q in zip(n, np3)), *np3)
        mk = lambda: pa.pvector_from_function(lambda i: orc.hash_x(i.get_local_to_global()) * (i.get_local_to_owner() == i.part), A.col_partition)
        outs = []
        for f in (pa.mul_, pa.mul_c_, pa.mul_no_lat_c_, pa.mul_no_overlap_):
            x, y = mk(), pa.pzeros(A.row_partition)
            f(y, A, x)
            outs.append(([v.copy() for v in y.own_values().items], [v.copy() for v in x.local_values().items]))
        for ys, xs in outs[1:]:
            assert all(np.array_equal(a, b) for a, b in zip(ys, outs[0][0]))
            assert all(np.array_equal(a, b) for a, b in zip(xs, outs[0][1]))
        for xs, ind in zip(outs[0][1], A.col_partition.items):
            assert np.array_equal(xs, orc.hash_x(ind.get_local_to_global()))


def test_config2_laplacian_256_cubed_single_part(orc):
    """BASELINE config 2: 7-point Laplacian 256^3, one part, fp64 CSR SpMV only (no exchange), through the
    step-by-step set-up chain.  Size-independent properties: A*1 == alpha*(2D - #neighbours) bit-exactly
    (src/gallery.jl:36,65,75), and exact scaling by powers of two."""
    n = (256, 256, 256)
    I, J, V, rows, _ = pa.laplacian_fdm(n, (1, 1, 1), ranks(1))
    A = pa.psparse_from_coo(I, J, V, rows)
    blk = A.matrix_partition.items[0]
    assert (blk.own_own.nnz, blk.own_ghost.nnz) == (117047296, 0)
    del I, J, V
    y = pa.pzeros(A.row_partition)
    pa.mul_(y, A, pa.pones(A.col_partition))
    alpha = float(257 ** 3)
    ax = np.arange(256)
    nb = sum(np.meshgrid(*[2 - (ax == 0) - (ax == 255)] * 3, indexing="ij")).transpose(2, 1, 0).ravel()
    assert np.array_equal(y.own_values().items[0], alpha * (6 - nb))
    x = pa.pvector_from_function(lambda i: (i.get_local_to_global() % 5) - 2.0, A.col_partition)
    x8 = pa.pvector_from_function(lambda i: 8.0 * ((i.get_local_to_global() % 5) - 2.0), A.col_partition)
    y8 = pa.pzeros(A.row_partition)
    pa.mul_(y, A, x)
    pa.mul_(y8, A, x8)
    assert np.array_equal(8.0 * y.own_values().items[0], y8.own_values().items[0])


def test_mul_sub_assembled_matrix(orc):
    """mul!(c,a,b) with !a.assembled (src/p_sparse_matrix.jl:2094-2097,2121-2139): own and ghost rows are multiplied,
    then assemble!(c) sends the ghost-row results to their owners (test/fem_example.jl:331-338)."""
    nodes, parts = (17, 13), (2, 2)
    I, J, V, rows, cols = pa.laplacian_fem(nodes, parts, ranks(4))
    A = pa.psparse_disassembled(I, J, V, rows, cols, assemble=False)
    assert not A.assembled
    Io, Jo, Vo, orows, ocols = orc.laplacian_fem(nodes, parts)
    _, (oblocks, orows_sa, ocols_sa) = orc.psparse_disassembled(Io, Jo, Vo, orows, ocols)
    Ao = orc.PSparse([None] * 4, oblocks, orows_sa, ocols_sa, False)
    for alpha, beta in [(1.0, 0.0), (0.5, -1.0)]:
        xo = [orc.hash_x(c.local_to_global) * (c.local_to_owner == c.part) for c in ocols_sa]
        yo = [orc.hash_x(r.local_to_global + 3) for r in orows_sa]
        x = upload([v.copy() for v in xo], A.col_partition)
        y = upload([v.copy() for v in yo], A.row_partition)
        if (alpha, beta) == (1.0, 0.0):
            pa.mul_(y, A, x)                       # forwards to the 5-argument method
        else:
            pa.mul5_(y, A, x, alpha, beta)
        orc.mul5(yo, Ao, xo, alpha, beta)
        for got, exp in zip(y.local_values().items, yo):
            assert np.array_equal(got, exp), (alpha, beta)
    # and it agrees with the assembled operator up to rounding (different summation order)
    B = pa.psparse_disassembled(I, J, V, rows, cols)
    xb = pa.pvector_from_function(lambda i: orc.hash_x(i.get_local_to_global()) * (i.get_local_to_owner() == i.part), B.col_partition)
    yb = pa.pzeros(B.row_partition)
    pa.mul_(yb, B, xb)
    xs = pa.pvector_from_function(lambda i: orc.hash_x(i.get_local_to_global()) * (i.get_local_to_owner() == i.part), A.col_partition)
    ys = pa.pzeros(A.row_partition)
    pa.mul_(ys, A, xs)
    assert np.allclose(yb.collect(), ys.collect(), rtol=0, atol=1e-12)


def test_transpose_product(orc):
    """mul!(c,transpose(a),b,alpha,beta) (src/p_sparse_matrix.jl:2144-2162): ghost(c) = A_oh'*b, assemble!(c) overlapped
    with own(c) = A_oo'*b.  Bit-exact against the oracle; A = A' for the HPCG matrix, so it also equals A*b to rounding."""
    A, _ = pa.build_p_matrix(ranks(4), 6, 5, 4, 12, 10, 4, 2, 2, 1, keep_host=True, fused=True)
    Ao, _, _ = orc.hpcg_build_p_matrix(6, 5, 4, 2, 2, 1)
    for alpha, beta in [(1.0, 0.0), (-0.5, 2.0)]:
        bo = [orc.hash_x(r.local_to_global + 1) for r in Ao.rows]
        co = [orc.hash_x(c.local_to_global + 9) for c in Ao.cols]
        b = upload([v.copy() for v in bo], A.row_partition)
        c = upload([v.copy() for v in co], A.col_partition)
        pa.mul5_transpose_(c, A, b, alpha, beta)
        orc.mul5_transpose(co, Ao, bo, alpha, beta)
        for got, exp in zip(c.local_values().items, co):
            assert np.array_equal(got, exp), (alpha, beta)
    # symmetry: A'*b == A*b up to rounding
    bo = [orc.hash_x(c.local_to_global) * (c.local_to_owner == c.part) for c in Ao.cols]
    y = pa.pzeros(A.row_partition)
    pa.mul_(y, A, upload([v.copy() for v in bo], A.col_partition))
    c = pa.pzeros(A.col_partition)
    pa.mul5_transpose_(c, A, upload([v[:r.n_own].copy() for v, r in zip(bo, Ao.rows)], A.row_partition), 1.0, 0.0)
    assert np.allclose(y.collect(), c.collect(), rtol=0, atol=1e-12)
