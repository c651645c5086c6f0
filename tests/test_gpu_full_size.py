"""BASELINE.json configs at full size: size-independent properties (A*1 == b, ghosts == owners, linearity) and the end-to-end FEM / CG checks.
Bars: np.array_equal for everything but dot / norm (1e-13).  Needs a real MI355X (-m gpu)."""
import pytest

from gpu_helpers import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


def test_full_size_27pt_128_two_parts_properties():
    """BASELINE config 3 (27-pt 128^3 per part, 2 parts, here both on one GPU): size-independent properties.
    A*1 == b bit-exactly (G12), ghost values == owner values, linearity in x for power-of-two scalings."""
    A, b = pa.build_p_matrix(ranks(2), 128, 128, 128, 256, 128, 128, 2, 1, 1)
    assert pa.pmap(lambda m: (m.own_own.nnz, m.own_ghost.nnz), A.matrix_partition).items == [(55742968, 145924)] * 2
    y = pa.pzeros(A.row_partition)
    pa.mul_(y, A, pa.pones(A.col_partition))
    for got, exp in zip(y.own_values().items, b.own_values().items):
        assert np.array_equal(got, exp)
    g = A.col_partition
    x = pa.pvector_from_function(lambda i: ((i.get_local_to_global() % 7) - 3.0) * (i.get_local_to_owner() == i.part), g)
    pa.mul_(y, A, x)
    for vals, ind in zip(x.local_values().items, g.items):
        assert np.array_equal(vals, (ind.get_local_to_global() % 7) - 3.0)       # consistent!: ghosts == owners
    y4 = pa.pzeros(A.row_partition)
    x4 = pa.pvector_from_function(lambda i: 4.0 * ((i.get_local_to_global() % 7) - 3.0), g)
    pa.mul_(y4, A, x4)
    for a_, b_ in zip(y.own_values().items, y4.own_values().items):
        assert np.array_equal(4.0 * a_, b_)


def test_config4_full_size_256_cubed_eight_parts_on_one_gpu():
    """BASELINE config 4 at its full size -- 27-pt, 256^3 rows per part, 8 parts as (2,2,2), global 512^3 -- with all
    eight parts resident on ONE GPU (46 GB of HBM; the exchange is device-to-device copies instead of RCCL).
    Closed-form sizes of SURVEY 8 (C4), then size-independent properties: A*1 == b bit-exactly, ghosts == owners after
    consistent!, and three CG iterations with device scalars == the reference schedule, bit for bit."""
    n = 256
    A, b = pa.build_p_matrix(ranks(8), n, n, n, 2 * n, 2 * n, 2 * n, 2, 2, 2)
    sizes = pa.pmap(lambda m, c: (m.own_own.nnz, m.own_ghost.nnz, c.n_own, c.n_ghost), A.matrix_partition, A.col_partition)
    assert sizes.items == [(449455096, 1762567, 16777216, 197377)] * 8      # 766^3/8, (767^3 - 766^3)/8, 256^3, ghosts
    assert sum(s[0] + s[1] for s in sizes.items) == 8 * 451217663 == 1534 ** 3
    enc = A.matrix_partition.items[0].own_own.encoding()
    assert enc["pattern"] >= 0.999 * sum(enc.values())
    y = pa.pzeros(A.row_partition)
    pa.mul_(y, A, pa.pones(A.col_partition))
    for got, exp in zip(y.own_values().items, b.own_values().items):
        assert np.array_equal(got, exp)
    g = A.col_partition
    x = pa.pvector_from_function(lambda i: ((i.get_local_to_global() % 7) - 3.0) * (i.get_local_to_owner() == i.part), g)
    pa.mul_(y, A, x)
    for vals, ind in zip(x.ghost_values().items, g.items):
        assert np.array_equal(vals, (ind.get_local_to_global()[ind.n_own:] % 7) - 3.0)
    del x, y
    res = []
    for fn in (pa.ref_cg_, functools.partial(pa.opt_cg_, fuse=False), pa.opt_cg_):
        hist = []
        z, r0, r, it = fn(pa.pzeros(g), A, b, maxiter=3, history=hist)
        res.append((r0, r, hist, float(z.own_values().items[7][-1])))
        del z
    assert res[0] == res[1] and res[0][1] < res[0][0]
    # the fused loop (u'c accumulated inside the product kernels): the same numbers to rounding
    assert res[2][0] == res[0][0] and np.allclose(res[2][2], res[0][2], rtol=1e-12, atol=0) and abs(res[2][3] - res[0][3]) <= 1e-12 * abs(res[0][3])


@pytest.mark.parametrize("parts,cells", [((2, 2), (10, 10)), ((4, 2), (24, 18))])
def test_fem_example_all_variants(parts, cells):
    """BASELINE config 5 is test/fem_example.jl: ghosted cell partition, part-by-part dof numbering, cell-wise COO.
    Every solve of the reference file (:261-343) on the device path, each asserting norm(x - x_hat) < 1e-5 as it does:
    psparse + pvector; re-assembly with psparse! / pvector!; psystem; psystem with reuse and psystem! with doubled
    values; the sub-assembled system (mul! assembles the product)."""
    P = int(np.prod(parts))
    S = pa.fem_example.fem_example_system(ranks(P), parts, cells)
    I, J, V, II, VV, dofs = (S[k] for k in ("I", "J", "V", "II", "VV", "dof_partition"))
    A = pa.psparse_disassembled(I, J, V, dofs, dofs)                                    # :277,279
    b = pa.pvector_disassembled(II, VV, dofs)                                           # :280
    x = _fem_cg(A, b)
    assert _fem_error(x, S, A) < 1.0e-5                                                 # :288
    x_first = [v.copy() for v in x.own_values().items]
    A, cacheA = pa.psparse_disassembled(I, J, V, dofs, dofs, reuse=True)                # :291
    b, cacheb = pa.pvector_disassembled(II, VV, dofs, reuse=True)                       # :292
    pa.psparse_(A, V, cacheA).wait()                                                    # :293
    pa.pvector_(b, VV, cacheb)                                                          # :294
    x = _fem_cg(A, b)
    assert _fem_error(x, S, A) < 1.0e-5                                                 # :298
    for u, v in zip(x.own_values().items, x_first):
        assert np.array_equal(u, v)                                                     # re-assembly reproduces the bits
    A, b = pa.psystem(I, J, V, II, VV, dofs, dofs)                                      # :301
    assert _fem_error(_fem_cg(A, b), S, A) < 1.0e-5                                     # :303
    A, b, cache = pa.psystem(I, J, V, II, VV, dofs, dofs, reuse=True)                   # :313-317
    assert _fem_error(_fem_cg(A, b), S, A) < 1.0e-5                                     # :319
    V2, VV2 = pa.pmap(lambda v: 2 * v, V), pa.pmap(lambda v: 2 * v, VV)                 # :322-323
    pa.psystem_(A, b, V2, VV2, cache)                                                   # :325
    x = _fem_cg(A, b)
    assert _fem_error(x, S, A) < 1.0e-5                                                 # :328
    A, b = pa.psystem(I, J, V, II, VV, dofs, dofs, assemble=False)                      # :331
    assert not A.assembled and any(r.n_ghost > 0 for r in A.row_partition.items) == (P > 1)
    pa.assemble_(b).wait()                                                              # :332
    assert _fem_error(_fem_cg(A, b), S, A) < 1.0e-5                                     # :333-338


def test_config5_full_size_fem_4096_squared_eight_parts(orc):
    """BASELINE config 5 at the size SURVEY 8 names: Q1 FEM Laplacian on 4096 x 4096 nodes, 8 parts as (4,2), the
    default psparse route (disassembled COO -> assemble -> split).  Size-independent properties: ghosts == owners
    after consistent!, linearity for power-of-two scalings (bit-exact), and every part's own rows against the C oracle's
    spmv_csr!/mul!(…,1,1) run on that part's host blocks with the device's ghost values."""
    n = 4096
    I, J, V, rows, cols = pa.laplacian_fem((n, n), (4, 2), ranks(8))
    A = pa.psparse_disassembled(I, J, V, rows, cols, keep_host=True)
    del I, J, V
    assert sum(r.n_own for r in A.row_partition.items) == n * n
    g = A.col_partition
    xf = lambda i: orc.hash_x(i.get_local_to_global()) * (i.get_local_to_owner() == i.part)
    x, x4 = pa.pvector_from_function(xf, g), pa.pvector_from_function(lambda i: 4.0 * xf(i), g)
    y, y4 = pa.pzeros(A.row_partition), pa.pzeros(A.row_partition)
    pa.mul_(y, A, x)
    pa.mul_(y4, A, x4)
    K = orc.oracle_c()
    for yv, y4v, xv, ind, (oo, oh) in zip(y.own_values().items, y4.own_values().items, x.local_values().items, g.items,
                                          A.host_blocks.items):
        assert np.array_equal(xv, orc.hash_x(ind.get_local_to_global()))                 # ghosts == owners
        assert np.array_equal(4.0 * yv, y4v)
        want = np.zeros(ind.n_own)
        K.spmv_csr(want, np.ascontiguousarray(xv[:ind.n_own]), orc.CSR(oo.m, oo.n, oo.rowptr, oo.colval, oo.nzval))
        K.mul5_csr(want, orc.CSR(oh.m, oh.n, oh.rowptr, oh.colval, oh.nzval), np.ascontiguousarray(xv[ind.n_own:]), 1.0, 1.0)
        assert np.array_equal(yv, want)


@pytest.mark.gpu_extended
def test_fem_example_full_size_4096_squared_cells(orc):
    """test/fem_example.jl itself at BASELINE config 5's size: 4096 x 4096 cells on (4,2) parts (16.8 M free dofs, the
    dof partition is 1-D by part while the geometry is 2-D blocks: every part has interface dofs owned by up to three
    other parts).  psparse + pvector with the default flags, then the size-independent checks: ghosts == owners,
    linearity (bit-exact), every part's own rows against the C oracle on that part's host blocks, and the right-hand
    side against the oracle's pvector on a coarser copy of the same problem is covered by the small-size tests."""
    n = 4096
    S = pa.fem_example.fem_example_system(ranks(8), (4, 2), (n, n))
    dofs = S["dof_partition"]
    assert S["n_global_dofs"] == (n - 1) ** 2
    A = pa.psparse_disassembled(S["I"], S["J"], S["V"], dofs, dofs, keep_host=True)
    b = pa.pvector_disassembled(S["II"], S["VV"], dofs)
    assert sum(bk.own_own.nnz + bk.own_ghost.nnz for bk in A.matrix_partition.items) == (3 * (n - 1) - 2) ** 2
    g = A.col_partition
    xf = lambda i: orc.hash_x(i.get_local_to_global()) * (i.get_local_to_owner() == i.part)
    x, x4 = pa.pvector_from_function(xf, g), pa.pvector_from_function(lambda i: 4.0 * xf(i), g)
    y, y4 = pa.pzeros(A.row_partition), pa.pzeros(A.row_partition)
    pa.mul_(y, A, x)
    pa.mul_(y4, A, x4)
    K = orc.oracle_c()
    for yv, y4v, xv, ind, (oo, oh) in zip(y.own_values().items, y4.own_values().items, x.local_values().items, g.items,
                                          A.host_blocks.items):
        assert np.array_equal(xv, orc.hash_x(ind.get_local_to_global()))
        assert np.array_equal(4.0 * yv, y4v)
        want = np.zeros(ind.n_own)
        K.spmv_csr(want, np.ascontiguousarray(xv[:ind.n_own]), orc.CSR(oo.m, oo.n, oo.rowptr, oo.colval, oo.nzval))
        K.mul5_csr(want, orc.CSR(oh.m, oh.n, oh.rowptr, oh.colval, oh.nzval), np.ascontiguousarray(xv[ind.n_own:]), 1.0, 1.0)
        assert np.array_equal(yv, want)
    # the assembled right-hand side is non-zero only next to the Dirichlet boundary; A*x_hat reproduces it (the
    # discrete solution of this problem IS u = x + y: bilinear elements represent it exactly)
    xh = pa.pvector_from_function_values(pa.pmap(lambda s, c: pa.fem_example.setup_exact_solution(s, S["params"], c),
                                                 S["spaces"], g), g)
    pa.consistent_(xh).wait()
    pa.mul_(y, A, xh)
    for yv, bv in zip(y.own_values().items, b.own_values().items):
        assert np.allclose(yv, bv, rtol=0, atol=1e-12) and np.count_nonzero(bv) < 4 * 4 * n


def test_config4_shape_cg_iteration_8_parts_96_cubed():
    """BASELINE config 4's loop at 8 parts x 96^3 (7.1M rows, 190M stored entries, all parts on this GPU):
    assemble!(b) once, then CG iterations = {consistent! + mul!, 2 dots + norm, 3 axpys}, identity preconditioner
    (HPCG/src/ref_cg.jl:40-71).  Properties: A*1 == b bit-exactly on every part; assemble! leaves own values of an
    already assembled b untouched and zeroes its ghosts; the residual norm decreases monotonically for this SPD
    system (x -> 1; without the multigrid preconditioner 30 iterations only get part of the way)."""
    n = 96
    A, b = pa.build_p_matrix(ranks(8), n, n, n, 2 * n, 2 * n, 2 * n, 2, 2, 2)
    y = pa.pzeros(A.row_partition)
    pa.mul_(y, A, pa.pones(A.col_partition))
    for got, exp in zip(y.own_values().items, b.own_values().items):
        assert np.array_equal(got, exp)
    before = [v.copy() for v in b.own_values().items]
    pa.assemble_(b).wait()
    for v0, v1, g in zip(before, b.own_values().items, b.ghost_values().items):
        assert np.array_equal(v0, v1) and not g.any()
    hist = []
    x = pa.pzeros(A.col_partition)
    x, r0, r, it = pa.ref_cg_(x, A, b, maxiter=30, history=hist)
    assert it == 30 and all(h1 < h0 for h0, h1 in zip([r0] + hist[:-1], hist))
    assert r / r0 < 0.1                      # unpreconditioned CG on a 192^3 grid: slow but steady
    assert all(float(v.mean()) > 0.0 for v in x.own_values().items)   # x is moving from 0 towards the solution 1
