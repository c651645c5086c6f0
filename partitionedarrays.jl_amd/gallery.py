"""Workload generators (host side, native loops): the gallery Laplacian and the HPCG 27-point problem.

Mirrors /root/reference/src/gallery.jl:12-86 (laplacian_fdm) and HPCG/src/sparse_matrix.jl:27-122
(build_matrix / build_p_matrix), HPCG/src/compute_optimal_xyz.jl:8-64.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib as L
from .primitives import pmap, tuple_of_arrays
from .p_range import uniform_partition
from .p_sparse_matrix import psparse_from_coo
from .p_vector import pvector

I64, F64 = np.int64, np.float64


def laplacian_fdm(nodes_per_dir, parts_per_dir, parts):
    """laplacian_fdm(nodes_per_dir,parts_per_dir,parts) -> I,J,V,row_partition,col_partition (src/gallery.jl:12-86)."""
    n = np.array(nodes_per_dir, dtype=I64)
    D = len(n)
    node_partition = uniform_partition(parts, tuple(parts_per_dir), tuple(nodes_per_dir))

    def setup(nodes):
        lo = np.array([r[0] for r in nodes.ranges], dtype=I64)
        hi = np.array([r[1] for r in nodes.ranges], dtype=I64)
        nnz = C.c_int64()
        L.call("pa_host_laplacian_fdm", D, L.ptr(n), L.ptr(lo), L.ptr(hi), None, None, None, C.byref(nnz))
        I, J, V = np.zeros(nnz.value, I64), np.zeros(nnz.value, I64), np.zeros(nnz.value, F64)
        L.call("pa_host_laplacian_fdm", D, L.ptr(n), L.ptr(lo), L.ptr(hi), L.ptr(I), L.ptr(J), L.ptr(V), C.byref(nnz))
        return I, J, V

    I, J, V = tuple_of_arrays(pmap(setup, node_partition))
    return I, J, V, node_partition, node_partition


def build_matrix(nx, ny, nz, gnx, gny, gnz, gix0, giy0, giz0):
    """HPCG build_matrix (HPCG/src/sparse_matrix.jl:27-80) -> I, J, V, b, row_b."""
    nnz = C.c_int64()
    args = [int(x) for x in (nx, ny, nz, gnx, gny, gnz, gix0, giy0, giz0)]
    L.call("pa_host_hpcg_build_matrix", *args, None, None, None, None, None, C.byref(nnz))
    I, J, V = np.zeros(nnz.value, I64), np.zeros(nnz.value, I64), np.zeros(nnz.value, F64)
    b, row_b = np.zeros(nx * ny * nz, F64), np.zeros(nx * ny * nz, I64)
    L.call("pa_host_hpcg_build_matrix", *args, L.ptr(I), L.ptr(J), L.ptr(V), L.ptr(b), L.ptr(row_b), C.byref(nnz))
    return I, J, V, b, row_b


def compute_optimal_shape_XYZ(np_):
    """HPCG/src/compute_optimal_xyz.jl:8-64: the part grid (npx, npy, npz) of np parts -- the special cases in closed form, every
    other np by the reference's search over the ways to deal its prime powers to the directions (least surface, first found)."""
    if np_ == 1:
        return 1, 1, 1
    f, m, d = {}, np_, 2
    while m > 1:
        while m % d == 0:
            f[d] = f.get(d, 0) + 1
            m //= d
        d += 1
    primes = sorted(f)
    x = primes[0]
    if len(primes) == 1:
        e = f[x]
        z = x ** (e // 3)
        y = x ** (e // 3 + (1 if e % 3 >= 2 else 0))
        x = x ** (e // 3 + (1 if e % 3 >= 1 else 0))
        return x, y, z
    y = primes[1]
    if len(primes) == 2 and f[x] == 1 and f[y] == 1:
        return x, y, 1
    if len(primes) == 2 and f[x] + f[y] == 3:
        return x, y, (x if f[x] == 2 else y)
    if len(primes) == 3 and all(f[p] == 1 for p in primes):
        return x, y, primes[2]
    # 3 or more prime factors with repeats (np = 16 has one prime and is served above; 24, 36, 40, 48, ...): every way to deal the
    # prime powers to two of the directions is tried in the order of the reference's mixed-base counters (HPCG/src/
    # compute_optimal_xyz.jl:34-62, mixed_base_counter.jl) and the first shape of least surface tf1*tf2 + tf2*tf3 + tf1*tf3 wins;
    # the third direction gets what is left.  (x, y) are NOT sorted afterwards, as in the reference.
    powers = [f[p] for p in primes]
    n = len(primes)

    def counter_next(cur, mx):
        for i in range(n):
            cur[i] += 1
            if cur[i] > mx[i]:
                cur[i] = 0
                continue
            break

    def product(cur):
        out = 1
        for i in range(n):
            out *= primes[i] ** cur[i]
        return out

    min_area = 2.0 * np_ + 1.0
    z = 0
    c1 = [0] * n
    counter_next(c1, powers)
    while any(c1):
        mx2 = [powers[i] - c1[i] for i in range(n)]
        c2 = [0] * n
        counter_next(c2, mx2)
        while any(c2):
            tf1, tf2 = product(c1), product(c2)
            tf3 = np_ / tf1 / tf2
            area = tf1 * tf2 + tf2 * tf3 + tf1 * tf3
            if area < min_area:
                min_area, x, y, z = area, tf1, tf2, tf3
            counter_next(c2, mx2)
        counter_next(c1, powers)
    return int(x), int(y), int(z // 1)


def build_split_blocks_fused(my_rows, nx, ny, nz, gnx, gny, gnz):
    """One part of build_p_matrix without the Int64 COO triplets (csrc/pa_host.cpp, "Fused HPCG set-up"):
    returns (col LocalIndices with first-seen ghosts, own_own HostCSR, own_ghost HostCSR, b)."""
    from .p_range import LocalIndices, find_owner
    from .p_sparse_matrix import HostCSR
    from .primitives import DebugArray
    g0 = [int(my_rows.ranges[d][0]) for d in range(3)]
    args = [int(v) for v in (nx, ny, nz, gnx, gny, gnz, *g0)]
    ng, noo, noh = C.c_int64(), C.c_int64(), C.c_int64()
    L.call("pa_host_hpcg_ghosts", *args, None, C.byref(ng), C.byref(noo), C.byref(noh))
    ghosts = np.zeros(ng.value, I64)
    L.call("pa_host_hpcg_ghosts", *args, L.ptr(ghosts), C.byref(ng), C.byref(noo), C.byref(noh))
    owners = find_owner(DebugArray([my_rows]), DebugArray([ghosts])).items[0]
    cols = LocalIndices(my_rows.n_global, my_rows.part, np_=my_rows.np_, n=my_rows.n, ranges=my_rows.ranges,
                        starts=my_rows.starts, ghost_to_global=ghosts, ghost_to_owner=owners)
    n = nx * ny * nz
    # Int32 row pointers as HPCG stores them (sparse_matrix.jl:115); Int64 ones for a part of 2^31 entries or more
    big = max(noo.value, noh.value) >= 2 ** 31 - 2 ** 16
    rp_t, fn = (np.int64, "pa_host_hpcg_split_csr64") if big else (np.int32, "pa_host_hpcg_split_csr")
    oo = HostCSR(n, n, np.zeros(n + 1, rp_t), np.zeros(noo.value, np.int32), np.zeros(noo.value, F64))
    oh = HostCSR(n, ng.value, np.zeros(n + 1, rp_t), np.zeros(noh.value, np.int32), np.zeros(noh.value, F64))
    b = np.zeros(n, F64)
    L.call(fn, *args, L.ptr(ghosts), ng.value, L.ptr(oo.rowptr), L.ptr(oo.colval), L.ptr(oo.nzval),
           L.ptr(oh.rowptr), L.ptr(oh.colval), L.ptr(oh.nzval), L.ptr(b))
    return cols, oo, oh, b


def build_split_blocks_device(my_rows, nx, ny, nz, gnx, gny, gnz):
    """One part of build_p_matrix with the own|own block and b generated in HBM (csrc/pa_rowsel.hip,
    pa_hpcg_own_block_create) and only the surface -- ghost ids in first-seen order, the own|ghost block -- made by the host:
    returns (col LocalIndices, SplitMatrixBlocks, DeviceVector b).  Same arrays as build_split_blocks_fused + upload
    (tests/test_gpu_setup.py::test_hpcg_blocks_generated_on_the_device_equal_the_host_s)."""
    from .p_range import LocalIndices, find_owner
    from .p_sparse_matrix import HostCSR, DeviceCSR, SplitMatrixBlocks
    from .p_vector import DeviceVector, context
    from .primitives import DebugArray
    g0 = [int(my_rows.ranges[d][0]) for d in range(3)]
    args = [int(v) for v in (nx, ny, nz, gnx, gny, gnz, *g0)]
    ng, noo, noh = C.c_int64(), C.c_int64(), C.c_int64()
    L.call("pa_host_hpcg_ghosts", *args, None, C.byref(ng), C.byref(noo), C.byref(noh))
    ghosts = np.zeros(ng.value, I64)
    L.call("pa_host_hpcg_ghosts", *args, L.ptr(ghosts), C.byref(ng), C.byref(noo), C.byref(noh))
    owners = find_owner(DebugArray([my_rows]), DebugArray([ghosts])).items[0]
    cols = LocalIndices(my_rows.n_global, my_rows.part, np_=my_rows.np_, n=my_rows.n, ranges=my_rows.ranges,
                        starts=my_rows.starts, ghost_to_global=ghosts, ghost_to_owner=owners)
    n = nx * ny * nz
    if noh.value == 0:                                # (a part without neighbours: no surface to walk, no row pointers to upload)
        e = C.c_void_p()
        L.call("pa_csr_create_empty", context().h, n, ng.value, C.byref(e))
        oh_dev = DeviceCSR.from_handle(e, n, ng.value, 0)
    else:
        oh = HostCSR(n, ng.value, np.empty(n + 1, np.int32), np.empty(noh.value, np.int32), np.empty(noh.value, F64))
        L.call("pa_host_hpcg_ghost_block", *args, L.ptr(ghosts), ng.value, L.ptr(oh.rowptr), L.ptr(oh.colval), L.ptr(oh.nzval))
        oh_dev = None
    h = C.c_void_p()
    L.call("pa_hpcg_own_block_create", context().h, *args, C.byref(h), None)
    v = DeviceVector(cols.n_own, cols.n_ghost)        # (after the block: the arena places it away from the matrix streams' class)
    L.call("pa_hpcg_rhs", context().h, *args, v.h)
    return cols, SplitMatrixBlocks(DeviceCSR.from_handle(h, n, n, noo.value), oh_dev if oh_dev is not None else DeviceCSR(oh)), v


def build_p_matrix(ranks, nx, ny, nz, gnx, gny, gnz, npx, npy, npz, keep_host=False, fused=None, keep_raw=False):
    """HPCG build_p_matrix (HPCG/src/sparse_matrix.jl:105-122) -> A (device PSparseMatrix), b (PVector).

    fused=False: the reference's chain, step by step (build_matrix -> find_owner -> union_ghost -> psparse).
    fused=True : the same arrays produced by the fused native generator (no COO triplets in host memory);
                 default for parts of >= 2^18 rows.  tests/test_host_setup.py pins fused == chain == oracle.
    keep_raw=True: the blocks keep their raw Int32 columns in HBM (pa_ctx_keep_raw_columns) so that row subsets can be cut
                 from them on the device (the multigrid set-up, hpcg.pc_setup); drop them with DeviceCSR.drop_raw_columns()."""
    from .p_sparse_matrix import PSparseMatrix, SplitMatrixBlocks, DeviceCSR
    from .p_vector import PVector, DeviceVector
    row_partition = uniform_partition(ranks, (npx, npy, npz), (gnx, gny, gnz))
    if fused is None:
        fused = nx * ny * nz >= (1 << 18)
    import os
    if fused and not keep_host and nx * ny * nz * 27 < 2 ** 31 - 2 ** 16 and os.environ.get("PA_SETUP_DEVICE", "1") != "0":
        # nobody wants the host copy: the own|own block and b are generated in HBM (no 5.4 GB over PCIe for a 256^3 part)
        from .p_vector import context

        def one_dev(my_rows):
            if keep_raw:
                L.call("pa_ctx_keep_raw_columns", context().h, 1)
            try:
                return build_split_blocks_device(my_rows, nx, ny, nz, gnx, gny, gnz)
            finally:
                if keep_raw:
                    L.call("pa_ctx_keep_raw_columns", context().h, 0)

        cols, blocks, bvals = tuple_of_arrays(pmap(one_dev, row_partition))
        return PSparseMatrix(blocks, row_partition, cols, True, None), PVector(bvals, cols)
    if fused:
        def one(my_rows):
            cols, oo, oh, b = build_split_blocks_fused(my_rows, nx, ny, nz, gnx, gny, gnz)
            if keep_raw:
                from .p_vector import context
                L.call("pa_ctx_keep_raw_columns", context().h, 1)
            try:
                blk = SplitMatrixBlocks(DeviceCSR(oo), DeviceCSR(oh))
            finally:
                if keep_raw:
                    L.call("pa_ctx_keep_raw_columns", context().h, 0)
            v = DeviceVector(cols.n_own, cols.n_ghost)
            v.upload(b, 0)
            return cols, blk, v, ((oo, oh) if keep_host else None)

        cols, blocks, bvals, host = tuple_of_arrays(pmap(one, row_partition))
        A = PSparseMatrix(blocks, row_partition, cols, True, host if keep_host else None)
        return A, PVector(bvals, cols)

    def gen(my_rows):
        g0 = int(my_rows.ranges[0][0]), int(my_rows.ranges[1][0]), int(my_rows.ranges[2][0])   # Tuple(cis[first(my_rows)])
        return build_matrix(nx, ny, nz, gnx, gny, gnz, *g0)

    I, J, V, b, I_b = tuple_of_arrays(pmap(gen, row_partition))
    A = psparse_from_coo(I, J, V, row_partition, keep_host=keep_host)
    pb = pvector(I_b, b, A.col_partition)        # row_partition = partition(axes(A,2)) (:119)
    return A, pb


class DeviceTriplets:
    """One of the arrays I, J, V of a part's COO triplets, generated in HBM (pa_fem_triplets_device): `len`, `download()` (a numpy
    copy), and the device pointer psparse_disassembled hands to pa_coo_subassemble.  The three arrays of a part share one owner."""

    class _Owner:
        def __init__(self, ctx, ptrs):
            self.ctx, self.ptrs = ctx, ptrs

        def __del__(self):
            try:
                L.lib.pa_triplets_free(self.ctx.h, *self.ptrs)
            except Exception:                                      # noqa: BLE001
                pass

    def __init__(self, owner, ptr, n, dtype):
        self.owner, self.ptr, self.n, self.dtype = owner, ptr, int(n), dtype
        self.ids_from_one = True                                   # (node ids of interior nodes: >= 1 by construction)

    def __len__(self):
        return self.n

    def download(self):
        out = np.empty(self.n, self.dtype)
        L.call("pa_triplets_download", self.owner.ctx.h, self.ptr, self.n, L.ptr(out) if self.n else None)
        return out


def laplacian_fem(nodes_per_dir, parts_per_dir, parts, device=False):
    """laplacian_fem(nodes_per_dir,parts_per_dir,parts) (src/gallery.jl:110-239): Q1 Laplacian, interior nodes only.
    Each part loops over ITS CELLS, so the COO it returns contains rows owned by other parts (disassembled input
    for psparse).  Same entry order as the reference: cells column-major, local node i, then local node j.
    device=True: the triplets are generated in HBM (pa_fem_triplets_device) and returned as DeviceTriplets -- what
    psparse_disassembled takes without an upload; `.download()` gives the host arrays of the default route, bit for bit."""
    import itertools
    D = len(nodes_per_dir)
    nodes = tuple(int(k) for k in nodes_per_dir)
    cells_per_dir = tuple(k + 1 for k in nodes)
    h = [1.0 / (k + 1) for k in nodes]
    # ref_matrix (:123-162), literal including its index conventions
    gp = np.array([-np.sqrt(3.0) / 3.0, np.sqrt(3.0) / 3.0])
    sf = np.stack([0.5 * (1 - gp), 0.5 * (gp + 1)], axis=1)
    sg1 = np.array([[-0.5, 0.5], [-0.5, 0.5]])
    loc = [tuple(reversed(t)) for t in itertools.product(*[range(2)] * D)]         # column-major 2^D corner offsets
    nloc = len(loc)
    sg = np.zeros((nloc, nloc, D))
    for a, la in enumerate(loc):
        for b, pb in enumerate(loc):
            for d in range(D):
                v = 1.0
                for i in range(D):
                    v = v * ((2.0 / h[d]) * sg1[la[d], pb[d]] if i == d else sf[la[i], pb[i]])
                sg[a, b, d] = v
    dV = float(np.prod(h)) / (2 ** D)
    Aref = np.zeros((nloc, nloc))
    for i in range(nloc):
        for j in range(nloc):
            for k in range(nloc):
                Aref[i, j] += dV * float(np.dot(sg[k, i], sg[k, j]))
    node_partition = uniform_partition(parts, tuple(parts_per_dir), nodes)
    cell_partition = uniform_partition(parts, tuple(parts_per_dir), cells_per_dir)
    strides = [int(np.prod(nodes[:d])) for d in range(D)]

    def setup_native(cells):
        # the same triplets in the same order from the native, threaded loop (csrc/pa_host.cpp: pa_host_laplacian_fem); the numpy
        # version below stays as its checker (PA_FEM_NATIVE=0; tests/test_host_setup.py compares the two)
        lo = np.array([r[0] for r in cells.ranges], dtype=I64)
        hi = np.array([r[1] for r in cells.ranges], dtype=I64)
        nd = np.array(nodes, dtype=I64)
        Ar = np.ascontiguousarray(Aref, F64)
        nnz = C.c_int64()
        L.call("pa_host_laplacian_fem", D, L.ptr(nd), L.ptr(lo), L.ptr(hi), L.ptr(Ar), None, None, None, C.byref(nnz))
        Ii, Ji, Vi = np.empty(nnz.value, I64), np.empty(nnz.value, I64), np.empty(nnz.value, F64)
        L.call("pa_host_laplacian_fem", D, L.ptr(nd), L.ptr(lo), L.ptr(hi), L.ptr(Ar), L.ptr(Ii), L.ptr(Ji), L.ptr(Vi), C.byref(nnz))
        return Ii, Ji, Vi

    def setup(cells):
        axes = [np.arange(lo, hi + 1, dtype=I64) for lo, hi in cells.ranges]
        grids = np.meshgrid(*axes, indexing="ij")
        cc = [g.transpose(tuple(reversed(range(D)))).ravel() for g in grids]    # cells in column-major order
        ncell = len(cc[0])
        Im = np.zeros((ncell, nloc, nloc), I64)
        Jm = np.zeros((ncell, nloc, nloc), I64)
        ok = np.ones((ncell, nloc, nloc), bool)
        node_id, node_ok = [], []
        for a, la in enumerate(loc):
            coord = [cc[d] + la[d] - 1 for d in range(D)]                           # cell + local_node - offset(2)
            node_ok.append(np.all([(coord[d] >= 1) & (coord[d] <= nodes[d]) for d in range(D)], axis=0))
            node_id.append(sum((coord[d] - 1) * strides[d] for d in range(D)) + 1)
        for a in range(nloc):
            for b in range(nloc):
                Im[:, a, b], Jm[:, a, b] = node_id[a], node_id[b]
                ok[:, a, b] = node_ok[a] & node_ok[b]
        Vm = np.broadcast_to(Aref[None, :, :], ok.shape)
        return Im[ok], Jm[ok], Vm[ok].copy()

    def setup_device(cells):
        from .p_vector import context
        ctx = context()
        lo = np.array([r[0] for r in cells.ranges], dtype=I64)
        hi = np.array([r[1] for r in cells.ranges], dtype=I64)
        nd = np.array(nodes, dtype=I64)
        Ar = np.ascontiguousarray(Aref, F64)
        n = C.c_int64()
        p = [C.c_void_p() for _ in range(3)]
        L.call("pa_fem_triplets_device", ctx.h, D, L.ptr(nd), L.ptr(lo), L.ptr(hi), L.ptr(Ar), C.byref(n), *[C.byref(q) for q in p])
        owner = DeviceTriplets._Owner(ctx, [C.c_void_p(q.value) for q in p])
        return tuple(DeviceTriplets(owner, C.c_void_p(q.value), n.value, dt) for q, dt in zip(p, (I64, I64, F64)))

    native = D <= 3 and os.environ.get("PA_FEM_NATIVE", "1") != "0"
    if device and D <= 3:
        I, J, V = tuple_of_arrays(pmap(setup_device, cell_partition))
    else:
        I, J, V = tuple_of_arrays(pmap(setup_native if native else setup, cell_partition))
    return I, J, V, node_partition, node_partition
