"""Pattern-ELL (csrc/pa_pell.h, pa_pell.hip; round 6): the lane-per-row product kernel of pattern blocks, its slab CLASSES and the
lean form (pa_pell_slab_fast) -- against the oracle's spmv_csr! / mul!(y,A,x,alpha,beta) loops (src/sparse_utils.jl:649-669, SparseMatricesCSR
mul! as called at src/p_sparse_matrix.jl:2088), bit for bit (np.array_equal), and against the masked form and the row-split kernel.
Needs a real MI355X (-m gpu)."""
import numpy as np
import pytest

from gpu_helpers import pa, ranks, env
import pa_amd._lib as L

pytestmark = pytest.mark.gpu


def _stencil27(nx, ny, nz, rng=None, drop=0.0, few=0):
    """The 27-point operator of HPCG/src/sparse_matrix.jl:56-103 on an nx x ny x nz grid (26 on the diagonal, -1 elsewhere; rng: random
    values; drop: every entry but the diagonal is left out with this probability) as a 1-based CSR, columns ascending."""
    n = nx * ny * nz
    ix, iy, iz = np.meshgrid(np.arange(nx), np.arange(ny), np.arange(nz), indexing="ij")
    row = (ix + nx * (iy + ny * iz)).ravel()
    rows, cols = [], []
    for dz in (-1, 0, 1):
        for dy in (-1, 0, 1):
            for dx in (-1, 0, 1):
                ok = ((ix + dx >= 0) & (ix + dx < nx) & (iy + dy >= 0) & (iy + dy < ny) & (iz + dz >= 0) & (iz + dz < nz)).ravel()
                if drop and (dx, dy, dz) != (0, 0, 0):
                    ok &= rng.random(n) >= drop
                rows.append(row[ok])
                cols.append(row[ok] + dx + nx * (dy + ny * dz))
    rows, cols = np.concatenate(rows), np.concatenate(cols)
    order = np.lexsort((cols, rows))
    rows, cols = rows[order], cols[order]
    rp = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=n))]) + 1
    val = np.where(rows == cols, 26.0, -1.0) if rng is None or drop else rng.standard_normal(len(rows))
    if few:                                              # a handful of distinct values (incl. a -0.0 and a denormal), which one by a hash of the entry
        table = np.array([26.0, -1.0, 0.375, -0.0, 5e-324, -3.5e10, 1.0 / 3.0])[:few]
        val = table[(rows * 7919 + cols * 104729) % few]
    return pa.HostCSR(n, n, rp.astype(np.int32), (cols + 1).astype(np.int32), val)


def _check(orc, H, x, tag, expect_lean=None, expect_mode=None):
    Ho = orc.CSR(H.m, H.n, H.rowptr, H.colval, H.nzval)
    want = np.zeros(H.m)
    with np.errstate(invalid="ignore", over="ignore"):
        orc.oracle_c().spmv_csr(want, x, Ho)
        y0 = np.cos(np.arange(H.m, dtype=float))
        want5 = y0.copy(); orc.oracle_c().mul5_csr(want5, Ho, x, -0.75, 1.5)
        want1 = y0.copy(); orc.oracle_c().mul5_csr(want1, Ho, x, 1.0, 1.0)
    A = pa.DeviceCSR(H)
    info = A.pell()
    if expect_mode is not None:
        assert info["mode"] == expect_mode, (tag, info)
    if expect_lean is not None:
        lean = info["lean_slabs_bits"] if info["mode"] == 2 else info["lean_slabs"]
        assert (lean > 0) == expect_lean, (tag, info)
    xd = pa.DeviceVector(H.n, 0).upload(x)
    y = pa.DeviceVector(H.m, 0)
    pa.spmv_(y, A, xd)
    got = y.download()
    assert np.array_equal(got, want, equal_nan=True), (tag, info, np.flatnonzero(~((got == want) | (np.isnan(got) & np.isnan(want))))[:8])
    ok = ~np.isnan(want)         # (the sign of a NaN made by an invalid operation -- Inf * 0, Inf - Inf -- is the hardware's choice: x86 sets it, gfx950 does not)
    assert np.array_equal(np.signbit(got)[ok], np.signbit(want)[ok]), (tag, "signs of zeros")
    y.upload(y0.copy()); pa.spmv_(y, A, xd, alpha=-0.75, beta=1.5)
    assert np.array_equal(y.download(), want5, equal_nan=True), (tag, "alpha, beta")
    y.upload(y0.copy()); pa.spmv_(y, A, xd, alpha=1.0, beta=1.0)
    assert np.array_equal(y.download(), want1, equal_nan=True), (tag, "muladd")
    return info


GRIDS = [(64, 5, 4), (128, 4, 3), (200, 3, 3), (70, 4, 4), (256, 3, 2), (256, 4, 4), (33, 7, 5), (64, 1, 1), (130, 1, 3)]


@pytest.mark.parametrize("grid", GRIDS)
@pytest.mark.parametrize("values", ["two", "random", "few"])
def test_the_lean_form_has_the_bits_of_the_oracle_on_the_27_point_operator(orc, grid, values):
    """Grid lines of 64 rows and more (slabs of a class: first / middle / last of a line, full and not), shorter lines and lines that
    are no multiple of 64 (slabs across lines: classes with lane ballots, or none) -- fp64 stream and one bit per entry, lean form on
    and off, the row-split kernel: every product the oracle's bits, also the signs of zeros."""
    rng = np.random.default_rng(sum(grid) + (values == "two"))
    H = _stencil27(*grid, rng=None if values == "two" else rng, few=7 if values == "few" else 0)
    x = rng.standard_normal(H.n)
    x[rng.random(H.n) < 0.02] = -0.0
    x[rng.random(H.n) < 0.02] = 0.0
    # "few": a dictionary of 7 values -> one BYTE per entry in pattern-ELL order, the dictionary in LDS (mode 3)
    with env(PA_SPMV_VALUE_DICT="0" if values == "random" else "1"):
        info = _check(orc, H, x, (grid, values, "lean"), expect_mode={"two": 2, "random": 1, "few": 3}[values])
        assert info["classes"] > 0 and (info["unroll"] == 9 or min(grid[1:]) < 3), info
        if grid[0] >= 128 and info["unroll"] == 9:       # (the commonest slab width decides the unroll: 12 offsets on thin grids -> 4, no runs of three)
            assert info["lean_slabs_bits" if values == "two" else "lean_slabs"] > 0, info
        if values == "few":
            with env(PA_SPMV_PELL_BYTES="0"):            # the row-split kernel's one-byte stream, as before round 6's third step
                _check(orc, H, x, (grid, values, "row-split bytes"), expect_mode=0)
        with env(PA_SPMV_PELL_LEAN="0"):
            _check(orc, H, x, (grid, values, "masked"), expect_lean=False)
        with env(PA_SPMV_PELL_CLASSES="0"):
            i2 = _check(orc, H, x, (grid, values, "no classes"), expect_lean=False)
            assert i2["classes"] == 0
        with env(PA_SPMV_PELL="0"):
            _check(orc, H, x, (grid, values, "row split"), expect_mode=0)


@pytest.mark.parametrize("values", ["two", "random", "few"])
def test_an_inf_or_nan_next_to_a_grid_line_s_end_stays_where_the_reference_has_it(orc, values):
    """The lean form multiplies every lane's gathered x, also where a row has no entry (the first row of a grid line has no left
    neighbour: what it gathered is the LAST entry of the line before) -- x is replaced by 0.0 before the multiply there.  Inf / NaN at
    exactly those positions must reach the rows that store an entry in that column and no other row."""
    nx, ny, nz = 128, 5, 4
    rng = np.random.default_rng(5)
    H = _stencil27(nx, ny, nz, rng=None if values == "two" else rng, few=5 if values == "few" else 0)
    x = rng.standard_normal(H.n)
    ends = np.arange(nx - 1, H.n, nx)
    x[ends[::3]] = np.inf
    x[ends[1::3]] = np.nan
    x[np.arange(0, H.n, nx)[2::5]] = -np.inf
    with env(PA_SPMV_VALUE_DICT="0" if values == "random" else "1"):
        info = _check(orc, H, x, ("inf at line ends", values), expect_lean=True)
        assert info["mode"] == {"two": 2, "random": 1, "few": 3}[values]
        with env(PA_SPMV_PELL_LEAN="0"):
            _check(orc, H, x, ("inf at line ends, masked", values))


def test_rows_that_drop_entries_at_random_keep_plain_patterns_or_their_classes(orc):
    """Every row drops some of its 26 neighbours at random: the slab unions stay the 27 offsets, the lane ballots differ from slab
    to slab -- few slabs: classes with lane ballots (the lean form selects), many: more than 4096 classes, plain patterns."""
    rng = np.random.default_rng(77)
    for grid, many in (((128, 6, 5), False), ((128, 64, 40), True)):
        H = _stencil27(*grid, rng=rng, drop=0.1)
        H.nzval[:] = rng.standard_normal(H.nnz)
        x = rng.standard_normal(H.n)
        with env(PA_SPMV_VALUE_DICT="0", PA_SPMV_XWIN="0"):      # (a big block of banded rows without row patterns would go to the x-window launches)
            info = _check(orc, H, x, ("dropped entries", grid), expect_mode=1)
        assert (info["classes"] == 0) == many, info
        assert (info["lean_slabs"] > 0) == (not many), info


def test_an_infinite_dictionary_value_keeps_the_masked_form(orc):
    """One bit per entry with a dictionary value that is not finite: value * 0.0 of an absent entry would be NaN, so no slab of such
    a block takes the lean form (set-up counts none) and the product keeps the oracle's bits."""
    H = _stencil27(128, 4, 3)
    H.nzval[H.nzval == 26.0] = np.inf
    x = np.random.default_rng(3).standard_normal(H.n)
    with env(PA_SPMV_VALUE_DICT="1"):
        info = _check(orc, H, x, "inf in the dictionary", expect_mode=2)
    assert info["lean_slabs_bits"] == 0, info


@pytest.mark.parametrize("vdict", ["1", "0"])
def test_colour_sweeps_and_restriction_on_every_other_row_of_long_grid_lines(vdict):
    """Grid lines of 128 and 256 rows: a colour of the multicolour smoother and the rows a restriction keeps are every other row of a
    line, 64 of them fill a slab (stride 2): the lean form's 16-byte gathers.  A V-cycle (colour sweeps = Gauss-Seidel update in
    place, fused residual + restriction, coarse levels with slabs across lines) and MG-PCG iterates: lean form == masked form ==
    row-split kernel, bit for bit."""
    outs = {}
    for tag, sw in (("lean", {}), ("masked", {"PA_SPMV_PELL_LEAN": "0"}), ("row split", {"PA_SPMV_PELL": "0"})):
        with env(PA_SPMV_VALUE_DICT=vdict, **sw):
            S = pa.pc_setup(ranks(1), 1, 3, 256, 8, 8, "multicolor_spmv", fuse_restriction=True)
            A, b = S.A_vec[-1], S.r[-1]
            z = pa.pzeros(A.col_partition)
            pa.ldiv_(z, S, b)
            h = []
            x, r0, r, it = pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=6, Pl=S, history=h, fuse=True)
            outs[tag] = (z.own_values().items[0].copy(), x.own_values().items[0].copy(), list(h))
            info = A.matrix_partition.items[0].own_own.pell()
            if tag == "lean":
                assert info["mode"] == (2 if vdict == "1" else 1) and info["classes"] > 0, info
                assert (info["lean_slabs_bits"] if vdict == "1" else info["lean_slabs"]) > 0, info
    for tag in ("masked", "row split"):
        assert np.array_equal(outs["lean"][0], outs[tag][0]) and np.all(np.isfinite(outs["lean"][0])), tag
    # (MG-PCG: the product + dot launch sums its partials per slab on pattern-ELL and per chunk on the row split -- the iterates agree
    #  to rounding between the two kernels, bit for bit between the two forms of one kernel)
    assert np.array_equal(outs["lean"][1], outs["masked"][1]) and outs["lean"][2] == outs["masked"][2]
    assert np.allclose(outs["lean"][1], outs["row split"][1], rtol=1e-12, atol=0) and np.allclose(outs["lean"][2], outs["row split"][2], rtol=1e-10)
    assert np.any(outs["lean"][0] != 0.0)


def test_a_colour_block_of_long_grid_lines_takes_the_lean_form(orc):
    """One colour's rows of the 27-point operator on lines of 256 rows as a block of its own (every other row, the other rows empty:
    the library row-compacts it): stride-2 classes, lean slabs; product bits == the oracle's."""
    nx, ny, nz = 256, 4, 4
    H = _stencil27(nx, ny, nz)
    n = H.n
    ix = np.arange(n) % nx; iy = (np.arange(n) // nx) % ny; iz = np.arange(n) // (nx * ny)
    keep = (ix % 2 == 1) & (iy % 2 == 0) & (iz % 2 == 1)
    lens = np.diff(H.rowptr) * keep
    rp = (np.concatenate([[0], np.cumsum(lens)]) + 1).astype(np.int32)
    sel = np.repeat(keep, np.diff(H.rowptr))
    Hc = pa.HostCSR(n, n, rp, H.colval[sel].copy(), H.nzval[sel].copy())
    x = np.random.default_rng(9).standard_normal(n)
    for vdict in ("1", "0"):
        with env(PA_SPMV_VALUE_DICT=vdict):
            info = _check(orc, Hc, x, ("colour block", vdict), expect_mode=2 if vdict == "1" else 1)
            assert info["classes"] > 0 and (info["lean_slabs_bits"] if vdict == "1" else info["lean_slabs"]) > 0, info
            with env(PA_SPMV_PELL_LEAN="0"):
                _check(orc, Hc, x, ("colour block, masked", vdict))


def test_the_one_byte_stream_follows_value_updates_and_an_infinite_dictionary_value(orc):
    """Mode 3 (one byte per entry): (i) new values of the same pattern -- products run on the row-split fp64 stream until the
    dictionary is renewed (after eight products), then on the one-byte stream again, always the oracle's bits; (ii) a dictionary value
    that is not finite: the masked form serves every slab (an absent entry's value times 0.0 would be NaN in the lean form)."""
    H = _stencil27(128, 5, 4, few=6)
    rng = np.random.default_rng(17)
    x = rng.standard_normal(H.n)
    xd = pa.DeviceVector(H.n, 0).upload(x)
    y = pa.DeviceVector(H.m, 0)
    with env(PA_SPMV_VALUE_DICT="1"):
        A = pa.DeviceCSR(H)
        assert A.pell()["mode"] == 3 and A.value_dict() == 6
        new = np.array([2.0, -7.0, 0.125, 9.0])[(np.arange(H.nnz) * 31) % 4]
        A.update_values(new)
        Ho = orc.CSR(H.m, H.n, H.rowptr, H.colval, new)
        want = np.zeros(H.m); orc.oracle_c().spmv_csr(want, x, Ho)
        modes = []
        for _ in range(12):
            pa.spmv_(y, A, xd)
            assert np.array_equal(y.download(), want)
            modes.append(A.pell()["mode"])
        assert modes[0] != 3 and modes[-1] == 3 and A.value_dict() == 4, modes
        H2 = _stencil27(128, 5, 4, few=5)
        H2.nzval[H2.nzval == 0.375] = np.inf
        info = _check(orc, H2, x, "inf in a dictionary of five", expect_mode=3)
        A2 = pa.DeviceCSR(H2)
        assert A2.pell()["mode"] == 3


def test_product_and_dot_in_one_launch_on_the_one_byte_stream(orc):
    """The epilogue forms run on every value stream of the pattern-ELL kernel: pa_mul_dot (product + this slab's term of u'c, EPI 3) on a
    few-valued 27-point block -- c equals the plain product's and the oracle's bits, the dot agrees with the host's to rounding, the
    same on the fp64 stream and on the row-split kernel."""
    import pa_amd.p_sparse_matrix as psm
    H = _stencil27(128, 6, 5, few=6)
    m = H.m
    ind = pa.uniform_partition(ranks(1), m)
    uh = np.random.default_rng(29).standard_normal(m)
    Ho = orc.CSR(H.m, H.n, H.rowptr, H.colval, H.nzval)
    want = np.zeros(m); orc.oracle_c().spmv_csr(want, uh, Ho)
    dots = {}
    for tag, sw, mode in (("one byte", {"PA_SPMV_VALUE_DICT": "1"}, 3), ("fp64", {"PA_SPMV_VALUE_DICT": "0"}, 1), ("row split", {"PA_SPMV_VALUE_DICT": "1", "PA_SPMV_PELL": "0"}, 0),
                          ("one byte, separate launches", {"PA_SPMV_VALUE_DICT": "1", "PA_MUL_FUSED": "0"}, 3),
                          ("fp64, separate launches", {"PA_SPMV_VALUE_DICT": "0", "PA_MUL_FUSED": "0"}, 1)):
        with env(**sw):
            blk = pa.DeviceCSR(H)
            assert blk.pell()["mode"] == mode, (tag, blk.pell())
            empty = pa.DeviceCSR(pa.HostCSR(m, 0, np.ones(m + 1, np.int32), np.zeros(0, np.int32), np.zeros(0)))
            Ah = pa.PSparseMatrix(pa.DebugArray([psm.SplitMatrixBlocks(blk, empty)]), ind, ind, True)
            u = pa.pvector_from_function(lambda i: uh, ind)
            c = pa.pzeros(ind)
            assert psm.mul_dot_(c, Ah, u, 5)
            assert np.array_equal(c.own_values().items[0], want), tag
            dots[tag] = pa.read_slots(5)[0]
            assert abs(dots[tag] - float(uh @ want)) <= 1e-12 * max(1.0, abs(float(uh @ want))), tag
    # the product + dot form of k_spmv_pell on both streams: the same slabs, the same partial sums (dictionary blocks on the row-split
    # kernel's one-byte stream take the plain product and a dot pass of its own: another summation order)
    assert dots["one byte"] == dots["fp64"] and dots["one byte, separate launches"] == dots["fp64, separate launches"]
    assert abs(dots["row split"] - dots["fp64"]) <= 1e-12 * abs(dots["fp64"])


def test_mul_of_a_partitioned_matrix_whose_own_blocks_run_on_the_one_byte_stream(orc):
    """mul!(c,A,b) (src/p_sparse_matrix.jl:2090-2103) of 2 parts as ONE launch per part (pa_mul_all: the interior rows on the pattern-ELL
    kernel inside the fused launch, the boundary rows as its tail) after the own x own blocks got six distinct values: the dictionary is
    renewed after eight products, the interior rows then read one byte per entry with the dictionary in LDS; every product equals the
    oracle's mul! on the same values, bit for bit -- before the renewal (fp64 row split), after it, and as separate launches."""
    Ao, _, _ = orc.hpcg_build_p_matrix(64, 6, 6, 2, 1, 1)
    table = np.array([26.0, -1.0, 0.375, -0.0, 2.5, -3.5e10])
    with env(PA_SPMV_VALUE_DICT="1"):
        A, _ = pa.build_p_matrix(ranks(2), 64, 6, 6, 128, 6, 6, 2, 1, 1)      # (two values: a dictionary, one bit per entry)
        assert A.matrix_partition.items[0].own_own.pell()["mode"] == 2
        for blk, bo in zip(A.matrix_partition.items, Ao.blocks):
            new = table[(np.arange(bo.own_own.nzval.size) * 31) % len(table)]
            bo.own_own.nzval[:] = new
            blk.own_own.update_values(new)
        xo = [orc.hash_x(c.local_to_global) * (c.local_to_owner == c.part) for c in Ao.cols]
        from gpu_helpers import upload, oracle_mul
        x = upload([v.copy() for v in xo], A.col_partition)
        want = oracle_mul(orc, Ao, xo)
        y = pa.pzeros(A.row_partition)
        modes = []
        for rep in range(12):
            pa.mul_(y, A, x)
            for got, exp, r in zip(y.own_values().items, want, Ao.rows):
                assert np.array_equal(got, exp[:r.n_own]), rep
            modes.append(A.matrix_partition.items[0].own_own.pell()["mode"])
        assert modes[0] != 3 and modes[-1] == 3, modes
        with env(PA_MUL_FUSED="0"):
            pa.mul_(y, A, x)
            for got, exp, r in zip(y.own_values().items, want, Ao.rows):
                assert np.array_equal(got, exp[:r.n_own])
