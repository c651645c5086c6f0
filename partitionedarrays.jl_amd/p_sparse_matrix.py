"""PSparseMatrix on the device and the distributed product mul!.

Mirrors /root/reference/src/p_sparse_matrix.jl for the hot path:
    PSparseMatrix (:971), SplitMatrix blocks (:588-627), psparse(...;assembled=true) (:1249-1270),
    split_format_locally (:823-899), mul!(c,a,b) (:2090-2103), mul!(c,a,b,alpha,beta) (:2105-2142),
and the local kernels of src/sparse_utils.jl (compresscoo :313-350, spmv! :609-669), which here are
native host code (csrc/pa_host.cpp) and HIP kernels (csrc/pa_csr.hip, csrc/pa_plan.hip).
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass

import numpy as np

from . import _lib as L
from .primitives import pmap
from .p_range import PRange, find_owner, union_ghost
from .p_vector import PVector, Task, consistent_, assemble_, context

I32, I64, F64 = np.int32, np.int64, np.float64


@dataclass
class HostCSR:
    """SparseMatrixCSR{1,Float64,Int32} on the host: 1-based rowptr/colval, columns sorted per row."""
    m: int
    n: int
    rowptr: np.ndarray
    colval: np.ndarray
    nzval: np.ndarray

    @property
    def nnz(self):
        return len(self.nzval)


def compresscoo(I, J, V, m, n, skip=False) -> HostCSR:
    """compresscoo(SparseMatrixCSR{1,Float64,Int32},I,J,V,m,n;combine=+,skip) (src/sparse_utils.jl:313-350)."""
    I = np.ascontiguousarray(I, dtype=I32)
    J = np.ascontiguousarray(J, dtype=I32)
    V = np.ascontiguousarray(V, dtype=F64)
    assert len(I) == len(J) == len(V)
    rowptr = np.zeros(m + 1, dtype=I32)
    colval = np.zeros(len(I), dtype=I32)
    nzval = np.zeros(len(I), dtype=F64)
    nnz = C.c_int64()
    L.call("pa_host_compresscoo_csr", L.ptr(I), L.ptr(J), L.ptr(V), len(I), m, n, int(skip), L.ptr(rowptr),
           L.ptr(colval), L.ptr(nzval), C.byref(nnz))
    k = nnz.value
    if k != len(I):
        colval, nzval = colval[:k].copy(), nzval[:k].copy()
    return HostCSR(m, n, rowptr, colval, nzval)


def sparse_matrix(I, J, V, m, n) -> HostCSR:
    """sparse_matrix(T,I,J,V,m,n;skip=true) (src/sparse_utils.jl:398-410)."""
    return compresscoo(I, J, V, m, n, skip=True)


def split_format_locally(A: HostCSR, rows, cols):
    """split_format_locally(A,rows,cols) (src/p_sparse_matrix.jl:823-899) for own rows of a matrix whose
    local ids are [own|ghost] (block partitions: identity permutation).  Returns (own_own, own_ghost)."""
    assert rows.own_is_contiguous_prefix and cols.own_is_contiguous_prefix, \
        "the device SpMV needs [own|ghost] local ids (block partitions)"
    a, b = C.c_int64(), C.c_int64()
    n_or, n_oc, n_gc = rows.n_own, cols.n_own, cols.n_ghost
    L.call("pa_host_split_csr", n_or, n_oc, n_gc, L.ptr(A.rowptr), L.ptr(A.colval), L.ptr(A.nzval),
           None, None, None, None, None, None, C.byref(a), C.byref(b))
    oo = HostCSR(n_or, n_oc, np.zeros(n_or + 1, I32), np.zeros(a.value, I32), np.zeros(a.value, F64))
    oh = HostCSR(n_or, n_gc, np.zeros(n_or + 1, I32), np.zeros(b.value, I32), np.zeros(b.value, F64))
    L.call("pa_host_split_csr", n_or, n_oc, n_gc, L.ptr(A.rowptr), L.ptr(A.colval), L.ptr(A.nzval),
           L.ptr(oo.rowptr), L.ptr(oo.colval), L.ptr(oo.nzval), L.ptr(oh.rowptr), L.ptr(oh.colval), L.ptr(oh.nzval),
           C.byref(a), C.byref(b))
    return oo, oh


class DeviceCSR:
    """One CSR block in HBM (pa_csr): Int32 0-based indices + fp64 values + the row-split chunk table."""

    def __init__(self, A: HostCSR, ctx=None):
        self.ctx = ctx or context()
        self.m, self.n, self.nnz = A.m, A.n, A.nnz
        self.h = C.c_void_p()
        assert A.rowptr.dtype in (I32, I64) and A.colval.dtype == I32
        # Int64 row pointers: a block of 2^31 stored entries or more (kept as row slabs on the device)
        L.call("pa_csr_create_mixed", self.ctx.h, A.m, A.n, A.nnz, L.ptr(A.rowptr), A.rowptr.dtype.itemsize,
               L.ptr(A.colval), 4, 1, L.ptr(A.nzval), C.byref(self.h))

    @staticmethod
    def transposed(A: HostCSR, ctx=None) -> "DeviceCSR":
        """transpose(A) as a device CSR block: A's (rowptr, colval) ARE the CSC arrays of A'; the upload keeps, inside
        every row of A', the entries in ascending row of A -- the order SparseMatricesCSR.mul!(y,transpose(A),x,..)
        adds them in -- so the row-split kernel reproduces it bit for bit."""
        self = DeviceCSR.__new__(DeviceCSR)
        self.ctx = ctx or context()
        self.m, self.n, self.nnz = A.n, A.m, A.nnz
        self.h = C.c_void_p()
        L.call("pa_csr_create_from_csc", self.ctx.h, A.n, A.m, A.nnz, L.ptr(A.rowptr), L.ptr(A.colval), 4, 1,
               L.ptr(A.nzval), C.byref(self.h))
        L.call("pa_csr_set_alpha_inside", self.h, 0)      # (a CSR block's transpose: SparseMatricesCSR's (a*x)*alpha, not the CSC form)
        return self

    @staticmethod
    def from_csc(m, n, colptr, rowval, nzval, ctx=None) -> "DeviceCSR":
        """A block the caller keeps in the DEFAULT SparseMatrixCSC storage (1-based colptr / rowval): converted to the row-split
        layout on the way up (spmv_csc! == spmv_csr! bit for bit, src/sparse_utils.jl:671-690); its 5-argument product follows
        SparseArrays' CSC method, a*(x*alpha)."""
        self = DeviceCSR.__new__(DeviceCSR)
        self.ctx = ctx or context()
        self.m, self.n, self.nnz = int(m), int(n), len(nzval)
        self.h = C.c_void_p()
        colptr, rowval = np.ascontiguousarray(colptr, np.int32), np.ascontiguousarray(rowval, np.int32)
        L.call("pa_csr_create_from_csc", self.ctx.h, self.m, self.n, self.nnz, L.ptr(colptr), L.ptr(rowval), 4, 1,
               L.ptr(np.ascontiguousarray(nzval, F64)), C.byref(self.h))
        return self

    @staticmethod
    def from_handle(h, m, n, nnz, ctx=None) -> "DeviceCSR":
        """Adopt a pa_csr the library built itself (a block assembled on the device, pa_coo_assembly_blocks)."""
        self = DeviceCSR.__new__(DeviceCSR)
        self.ctx = ctx or context()
        self.m, self.n, self.nnz = int(m), int(n), int(nnz)
        self.h = h
        return self

    def info(self):
        v = [C.c_int64() for _ in range(6)]
        L.call("pa_csr_info", self.h, *[C.byref(x) for x in v])
        keys = ["n_rows", "n_cols", "nnz", "n_chunks", "n_nonempty_rows", "n_long_rows"]
        return dict(zip(keys, [x.value for x in v]))

    def encoding(self):
        """Chunks by column encoding: recomputed from row patterns / 16-bit windowed stream / 32-bit columns."""
        v = [C.c_int64() for _ in range(3)]
        L.call("pa_csr_encoding", self.h, *[C.byref(x) for x in v])
        return dict(zip(["pattern", "c16", "c32"], [x.value for x in v]))

    def pell(self):
        """Pattern-ELL storage of the block (pa_csr_pell_info, csrc/pa_pell.hip): mode = what a product runs on now (0 row split,
        1 pattern-ELL fp64 stream, 2 pattern-ELL one bit per entry, 3 pattern-ELL one byte per entry), slabs of 64 rows, distinct slab
        patterns, value slots, unroll, slab classes and the slabs the lean form serves."""
        mode, unroll = C.c_int(), C.c_int()
        v = [C.c_int64() for _ in range(3)]
        L.call("pa_csr_pell_info", self.h, C.byref(mode), *[C.byref(x) for x in v], C.byref(unroll))
        w = [C.c_int64() for _ in range(3)]
        L.call("pa_csr_pell_lean_info", self.h, *[C.byref(x) for x in w])
        return dict(mode=mode.value, slabs=v[0].value, patterns=v[1].value, value_slots=v[2].value, unroll=unroll.value,
                    classes=w[0].value, lean_slabs=w[1].value, lean_slabs_bits=w[2].value)

    def xwin(self):
        """x-window launch of banded rows without a pattern (pa_csr_xwin_info): groups, chunks in groups, staged x entries."""
        v = [C.c_int64() for _ in range(4)]
        L.call("pa_csr_xwin_info", self.h, *[C.byref(x) for x in v])
        r = C.c_int64()
        L.call("pa_csr_xring_info", self.h, C.byref(r))
        return dict(zip(["groups", "chunks", "staged_x", "big_groups", "ring_groups"], [x.value for x in v] + [r.value]))

    def chain(self):
        """A column-split chain (pa_csr_chain_info): its pieces (0: not a chain) and the workgroups of the ONE launch the product runs
        it as (0: a launch per piece)."""
        p, g = C.c_int32(), C.c_int64()
        L.call("pa_csr_chain_info", self.h, C.byref(p), C.byref(g))
        return {"pieces": p.value, "groups_one_launch": g.value}

    def device_bytes(self):
        """HBM bytes the block occupies (pa_csr_device_bytes)."""
        n = C.c_int64()
        L.call("pa_csr_device_bytes", self.h, C.byref(n))
        return n.value

    def stream_bytes(self):
        """Bytes one product must read from the block, each once (pa_csr_stream_bytes): values, row pointers, chunk table
        and whatever gives each chunk its columns."""
        n = C.c_int64()
        L.call("pa_csr_stream_bytes", self.h, C.byref(n))
        return n.value

    def has_raw_columns(self):
        """True when row subsets can be cut from this block on the device (pa_csr_select_rows): it kept its raw Int32
        columns (created under pa_ctx_keep_raw_columns) or never compacted them."""
        n = C.c_int()
        L.call("pa_csr_has_raw_columns", self.h, C.byref(n))
        return bool(n.value)

    def drop_raw_columns(self):
        L.call("pa_csr_drop_raw_columns", self.h)

    @staticmethod
    def select_rows(own_own, own_ghost, mask, n_sel, lower_cols=None):
        """pa_csr_select_rows: the n_sel blocks made of the rows r with mask[r] == k of own_own | own_ghost (unsplit column
        order), built on the device.  lower_cols = n: pa_csr_select_rows_lower instead -- only the entries in own columns of
        a lower mask value, blocks of n columns, None where there is no such entry."""
        mask = np.ascontiguousarray(mask, np.int32)
        if mask.shape[0] != own_own.m:
            raise L.PAError("one mask entry per row")
        out = (C.c_void_p * n_sel)()
        if lower_cols is not None:
            L.call("pa_csr_select_rows_lower", own_own.h, int(lower_cols), L.ptr(mask), n_sel, out)
        else:
            L.call("pa_csr_select_rows", own_own.h, own_ghost.h if own_ghost is not None else None, L.ptr(mask), n_sel, out)
        blocks = []
        for k in range(n_sel):
            if not out[k]:
                blocks.append(None)
                continue
            v = [C.c_int64() for _ in range(6)]
            h = C.c_void_p(out[k])
            L.call("pa_csr_info", h, *[C.byref(x) for x in v])
            blocks.append(DeviceCSR.from_handle(h, v[0].value, v[1].value, v[2].value, own_own.ctx))
        return blocks

    def memory_class(self):
        """Memory class of the value stream inside the context's arena (pa_csr_memory_class; -1: outside)."""
        n = C.c_int()
        L.call("pa_csr_memory_class", self.h, C.byref(n))
        return n.value

    def value_dict(self):
        """Distinct values held in the optional value dictionary (PA_SPMV_VALUE_DICT=1 at creation), 0 when unused."""
        n = C.c_int()
        L.call("pa_csr_value_dict", self.h, C.byref(n))
        return n.value

    def debug_arrays(self):
        """Host copies of the arrays the product kernel reads (pa_csr_debug_array; first slab): a testing aid."""
        out = {}
        for which, (name, dt) in enumerate((("crp", np.int32), ("col", np.int32), ("col16", np.uint16), ("win", np.int32),
                                            ("pdesc", np.int32), ("pdelta", np.int32), ("chunk_row", np.int32), ("row_ids", np.int32))):
            n = C.c_int64()
            L.call("pa_csr_debug_array", self.h, which, None, 0, C.byref(n))
            a = np.zeros(n.value // np.dtype(dt).itemsize, dt)
            if n.value:
                L.call("pa_csr_debug_array", self.h, which, L.ptr(a), n.value, C.byref(n))
            out[name] = a
        return out

    def update_values(self, nzval):
        nzval = np.ascontiguousarray(nzval, F64)
        assert len(nzval) == self.nnz
        L.call("pa_csr_update_values", self.h, L.ptr(nzval))

    def __del__(self):
        try:
            L.lib.pa_csr_destroy(self.h)
        except Exception:
            pass


class DeviceSELL:
    """The same block in SELL-C-sigma storage, one lane per row (pa_sell, csrc/pa_sell.hip): a second bit-exact SpMV."""

    def __init__(self, A: HostCSR, sigma=1, ctx=None):
        self.ctx = ctx or context()
        self.m, self.n, self.nnz = A.m, A.n, A.nnz
        self.h = C.c_void_p()
        L.call("pa_sell_create", self.ctx.h, A.m, A.n, A.nnz, L.ptr(A.rowptr), L.ptr(A.colval), A.colval.dtype.itemsize, 1,
               L.ptr(A.nzval), int(sigma), C.byref(self.h))

    def info(self):
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        L.call("pa_sell_info", self.h, C.byref(a), C.byref(b), C.byref(c))
        return dict(n_slabs=a.value, padded_entries=b.value, nnz=c.value)

    def __del__(self):
        try:
            L.lib.pa_sell_destroy(self.h)
        except Exception:
            pass


class DeviceVector32:
    """Local values of a PVector{Vector{Float32}} in HBM, [own | ghost] (pa_vec32, csrc/pa_f32.hip)."""

    def __init__(self, n_own, n_ghost=0, ctx=None):
        self.ctx = ctx or context()
        self.n_own, self.n_ghost = int(n_own), int(n_ghost)
        self.h = C.c_void_p()
        L.call("pa_vec32_create", self.ctx.h, self.n_own, self.n_ghost, C.byref(self.h))

    def upload(self, host, offset=0):
        host = np.ascontiguousarray(host, np.float32)
        L.call("pa_vec32_upload", self.h, L.ptr(host), int(offset), len(host))
        return self

    def data_ptr(self):
        p = C.c_void_p()
        L.call("pa_vec32_data", self.h, C.byref(p))
        return p.value

    def download(self):
        out = np.zeros(self.n_own + self.n_ghost, np.float32)
        L.call("pa_vec32_download", self.h, L.ptr(out), 0, len(out))
        return out

    def fill(self, value, segment=L.SEG_LOCAL):
        L.call("pa_vec32_fill", self.h, segment, float(value))
        return self

    def __del__(self):
        try:
            L.lib.pa_vec32_destroy(self.h)
        except Exception:
            pass


class DeviceCSR32:
    """A SparseMatrixCSR{1,Float32,Int32} (or, csc=True: the colptr / rowval / nzval of a SparseMatrixCSC{Float32}) block in HBM
    (pa_csr32, csrc/pa_f32.hip): rowptr / colval 1-based as the reference stores them."""

    def __init__(self, m, n, ptr, idx, nzval, csc=False, index_base=1, ctx=None):
        self.ctx = ctx or context()
        self.m, self.n, self.nnz = int(m), int(n), len(nzval)
        ptr, idx = np.ascontiguousarray(ptr), np.ascontiguousarray(idx, ptr.dtype)
        nzval = np.ascontiguousarray(nzval, np.float32)
        self.h = C.c_void_p()
        L.call("pa_csr32_create_from_csc" if csc else "pa_csr32_create", self.ctx.h, self.m, self.n, self.nnz, L.ptr(ptr), L.ptr(idx),
               ptr.dtype.itemsize, int(index_base), L.ptr(nzval), C.byref(self.h))

    def info(self):
        on, a, b = C.c_int(), C.c_int64(), C.c_int64()
        L.call("pa_csr32_info", self.h, C.byref(on), C.byref(a), C.byref(b))
        return dict(pattern_ell=bool(on.value), slabs=a.value, padded_entries=b.value)

    def __del__(self):
        try:
            L.lib.pa_csr32_destroy(self.h)
        except Exception:
            pass


def spmv32_(b, A, x, x_segment=L.SEG_OWN, b_segment=L.SEG_OWN, alpha=1.0, beta=0.0):
    """spmv!(b,A,x) / mul!(b,A,x,alpha,beta) in Float32 (src/sparse_utils.jl:617-690 with eltype Float32)."""
    L.call("pa_spmv32", A.h, x.h, x_segment, b.h, b_segment, float(alpha), float(beta))
    return b


def spmv_(b, A, x, x_segment=L.SEG_OWN, b_segment=L.SEG_OWN, alpha=1.0, beta=0.0):
    """spmv!(b,A,x) / mul!(b,A,x,alpha,beta) on device vectors (src/sparse_utils.jl:609-669); A: DeviceCSR or DeviceSELL."""
    L.call("pa_sell_spmv" if isinstance(A, DeviceSELL) else "pa_spmv", A.h, x.h, x_segment, b.h, b_segment, float(alpha), float(beta))
    return b


def tune_output_placement(A, x, y, x_segment=L.SEG_OWN, reps=10, rounds=3):
    """pa_spmv_tune_output: time y = A*x with y in every place the context can reach (its current one first) and move y's
    storage when another place is more than 1.5 % faster.  Returns {"places": [{"where", "ms"}...], "chosen": index, "moved"}."""
    cap = 8
    where, ms = np.zeros(cap, np.int32), np.zeros(cap, np.float64)
    n, chosen = C.c_int32(), C.c_int32()
    L.call("pa_spmv_tune_output", A.h, x.h, x_segment, y.h, int(reps), int(rounds), cap, L.ptr(where), L.ptr(ms), C.byref(n), C.byref(chosen))
    names = {-1: "plain hipMalloc", 9: "plain hipMalloc (pair-checked)"}
    return {"places": [{"where": names.get(int(w), f"arena class {int(w)}"), "ms": round(float(t), 4)} for w, t in zip(where[:n.value], ms[:n.value])],
            "chosen": int(chosen.value), "moved": bool(chosen.value != 0)}


@dataclass
class SplitMatrixBlocks:
    """SplitMatrix blocks of one part (src/p_sparse_matrix.jl:588-627); ghost rows only when sub-assembled."""
    own_own: DeviceCSR
    own_ghost: DeviceCSR
    ghost_own: DeviceCSR = None
    ghost_ghost: DeviceCSR = None


class PSparseMatrix:
    """PSparseMatrix(matrix_partition,row_partition,col_partition,assembled) (src/p_sparse_matrix.jl:971-991)
    with the split format living in HBM."""

    def __init__(self, matrix_partition, row_partition, col_partition, assembled, host_blocks=None):
        self.matrix_partition = matrix_partition      # DebugArray/TorchDistArray of SplitMatrixBlocks
        self.row_partition = row_partition
        self.col_partition = col_partition
        self.assembled = assembled
        self.host_blocks = host_blocks                # optional (own_own, own_ghost) HostCSR per part

    @property
    def axes(self):
        return (PRange(self.row_partition), PRange(self.col_partition))

    def nnz_local(self):
        return pmap(lambda b: b.own_own.nnz + b.own_ghost.nnz, self.matrix_partition)


def psparse(I, J, V, rows, cols, assembled=True, keep_host=False) -> PSparseMatrix:
    """psparse(SparseMatrixCSR{1,Float64,Int32},I,J,V,rows,cols;assembled=true)|>fetch
    (src/p_sparse_matrix.jl:1150,1249-1270): global->local ids, COO->CSR, split, upload."""
    assert assembled, "this build covers the assembled=true route (the one HPCG and the gallery tests use)"

    def build(Ii, Ji, Vi, r, c):
        li = r.global_to_local(Ii)                   # map_global_to_local! (:1253-1254)
        lj = c.global_to_local(Ji)
        A = sparse_matrix(li, lj, Vi, r.n_local, c.n_local)
        del li, lj
        oo, oh = split_format_locally(A, r, c)
        del A
        blk = SplitMatrixBlocks(DeviceCSR(oo), DeviceCSR(oh))
        return (blk, (oo, oh) if keep_host else None)

    both = pmap(build, I, J, V, rows, cols)
    blocks = pmap(lambda t: t[0], both)
    host = pmap(lambda t: t[1], both) if keep_host else None
    return PSparseMatrix(blocks, rows, cols, True, host)


def _device_assembly_applies(row_partition, I):
    """The device-side assembly (csrc/pa_assemble.hip) covers block row partitions without ghosts, parts with at least one
    triplet, on a box with a GPU; PA_SETUP_DEVICE=0 keeps the host route (the two are compared bit for bit in the tests)."""
    import os
    if os.environ.get("PA_SETUP_DEVICE", "1") == "0":
        return False
    try:
        context()
    except Exception:                                          # noqa: BLE001  (no GPU: the host route still builds host blocks)
        return False
    from .primitives import local_items
    return all(r.kind == "block" and r.n_ghost == 0 and r.n_own > 0 and len(i) > 0 and len(i) < 2 ** 31 - 10 ** 4
               for r, i in zip(local_items(row_partition), local_items(I)))


def psparse_from_coo_device(I, J, V, row_partition, keep_host=False) -> PSparseMatrix:
    """psparse_from_coo with everything per triplet on the device: one upload of (I, J, V), then kernels, scans and radix
    sorts (pa_coo_assemble); the host gets the new ghost gids and computes their owners (find_owner on a few thousand ids).
    The blocks never exist on the host unless keep_host asks for copies."""
    def build(Ii, Ji, Vi, r):
        Ii, Ji = np.ascontiguousarray(Ii, I64), np.ascontiguousarray(Ji, I64)
        Vi = np.ascontiguousarray(Vi, F64)
        D = len(r.n)
        n = np.array(r.n, I64)
        lo = np.array([a for a, _ in r.ranges], I64)
        hi = np.array([b for _, b in r.ranges], I64)
        h = C.c_void_p()
        L.call("pa_coo_assemble", context().h, len(Ii), L.ptr(Ii), L.ptr(Ji), L.ptr(Vi), D, L.ptr(n), L.ptr(lo), L.ptr(hi),
               L.ptr(n), L.ptr(lo), L.ptr(hi), 0, None, 1, C.byref(h))
        try:
            v = [C.c_int64() for _ in range(5)]
            ms = C.c_double()
            L.call("pa_coo_assembly_info", h, *[C.byref(x) for x in v], C.byref(ms))
            n_rows, n_own_cols, n_ghost, nnz_oo, nnz_oh = [x.value for x in v]
            ghosts = np.zeros(n_ghost, I64)
            L.call("pa_coo_assembly_ghosts", h, L.ptr(ghosts))
            from .primitives import DebugArray
            owners = find_owner(DebugArray([r]), DebugArray([ghosts])).items[0]
            from .p_range import LocalIndices
            c = LocalIndices(r.n_global, r.part, np_=r.np_, n=r.n, ranges=r.ranges, starts=r.starts,
                             ghost_to_global=ghosts, ghost_to_owner=owners)
            a, b = C.c_void_p(), C.c_void_p()
            L.call("pa_coo_assembly_blocks", h, C.byref(a), C.byref(b))
            blk = SplitMatrixBlocks(DeviceCSR.from_handle(a, n_rows, n_own_cols, nnz_oo), DeviceCSR.from_handle(b, n_rows, n_ghost, nnz_oh))
            host = None
            if keep_host:
                host = []
                for which, (ncol, nnz) in enumerate(((n_own_cols, nnz_oo), (n_ghost, nnz_oh))):
                    H = HostCSR(n_rows, ncol, np.zeros(n_rows + 1, I32), np.zeros(nnz, I32), np.zeros(nnz, F64))
                    L.call("pa_coo_assembly_download", h, which, L.ptr(H.rowptr), L.ptr(H.colval), L.ptr(H.nzval))
                    host.append(H)
                host = tuple(host)
        finally:
            L.lib.pa_coo_assembly_destroy(h)
        return blk, c, host

    out = pmap(build, I, J, V, row_partition)
    blocks = pmap(lambda t: t[0], out)
    cols = pmap(lambda t: t[1], out)
    host = pmap(lambda t: t[2], out) if keep_host else None
    return PSparseMatrix(blocks, row_partition, cols, True, host)


def psparse_from_coo(I, J, V, row_partition, keep_host=False, renumber=False) -> PSparseMatrix:
    """The route of HPCG.build_p_matrix / test/gallery_tests.jl:33: find_owner -> union_ghost -> psparse.
    renumber=True: the result goes through renumber_for_locality (parts whose own x own block has no locality are stored in a
    reverse Cuthill-McKee order; use the RETURNED matrix's row_partition / col_partition for the vectors)."""
    if _device_assembly_applies(row_partition, I):
        A = psparse_from_coo_device(I, J, V, row_partition, keep_host=keep_host)
    else:
        J_owner = find_owner(row_partition, J)
        cols = pmap(union_ghost, row_partition, J, J_owner)
        A = psparse(I, J, V, row_partition, cols, assembled=True, keep_host=keep_host)
    return renumber_for_locality(A) if renumber else A


def _check_axes(c: PVector, a: PSparseMatrix, b: PVector):
    """@boundscheck matching_own_indices / matching_ghost_indices (src/p_sparse_matrix.jl:2091-2093)."""

    def same_layout(u, v):
        pu, pv_ = getattr(u, "device_own_perm", None), getattr(v, "device_own_perm", None)
        return pu is pv_ or (pu is not None and pv_ is not None and np.array_equal(pu, pv_))

    def chk(ci, ri, coli, bi):
        if ci.n_own != ri.n_own:
            raise L.PAError("matching_own_indices(axes(c,1),axes(a,1)) failed")
        if bi.n_own != coli.n_own or bi.n_ghost != coli.n_ghost:
            raise L.PAError("matching_own/ghost_indices(axes(a,2),axes(b,1)) failed")
        # a renumbered matrix (renumber_for_locality) lays its vectors out in its own order: vectors must be made on ITS partitions
        if not same_layout(ci, ri) or not same_layout(bi, coli):
            raise L.PAError("the vectors' device layout is not the matrix's: make them on A.row_partition / A.col_partition of the "
                            "renumbered matrix (renumber_for_locality returns new partitions)")

    pmap(chk, c.index_partition, a.row_partition, a.col_partition, b.index_partition)


def mul_(c: PVector, a: PSparseMatrix, b: PVector) -> PVector:
    """mul!(c,a,b) (src/p_sparse_matrix.jl:2090-2103):
        t = consistent!(b)                         pack + exchange on the comm stream
        c_own  = A_oo * b_own      (spmv!)         overlaps with the exchange on the compute stream
        wait(t)                                    compute stream waits, unpack ghosts
        c_own += A_oh * b_ghost    (muladd!)
    """
    _check_axes(c, a, b)
    if not a.assembled:
        return mul5_(c, a, b, 1.0, 0.0)
    t = consistent_(b)
    pmap(lambda cv, blk, bv: spmv_(cv, blk.own_own, bv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0),
         c.vector_partition, a.matrix_partition, b.vector_partition)
    t.wait()
    pmap(lambda cv, blk, bv: spmv_(cv, blk.own_ghost, bv, L.SEG_GHOST, L.SEG_OWN, 1.0, 1.0),
         c.vector_partition, a.matrix_partition, b.vector_partition)
    return c


def _operator_handles(a: PSparseMatrix, b: PVector):
    """pa_matrix of every part for the pair (a, b): blocks of `a` + the exchange plan of b's cache; built once per pair."""
    cache = b.__dict__.setdefault("_pa_matrices", {})
    if id(a) not in cache:
        def mk(blk, plan):
            h = C.c_void_p()
            L.call("pa_matrix_create", context().h, blk.own_own.h, blk.own_ghost.h, plan, C.byref(h))
            return h
        cache[id(a)] = _OperatorHandles(pmap(mk, a.matrix_partition, b.cache.plans), a)
    return cache[id(a)].handles


class _OperatorHandles:
    def __init__(self, handles, a):
        self.handles, self.a = handles, a        # `a` stays alive as long as its handles do

    def __del__(self):
        try:
            from .primitives import local_items
            for h in local_items(self.handles):
                L.lib.pa_matrix_destroy(h)
        except Exception:
            pass


def mul_c_(c: PVector, a: PSparseMatrix, b: PVector, alpha=1.0, beta=0.0) -> PVector:
    """mul!(c,a,b[,alpha,beta]) queued by ONE library call per process (pa_mul5 / pa_mul_all, include/pa_hip.h
    "operator level") instead of five: the same kernels in the same order as mul_ / mul5_, so the same bits.
    Needs an assembled matrix and either all parts in this process or one part per process over RCCL."""
    from .primitives import DebugArray, TorchDistArray
    from . import p_vector as pv
    _check_axes(c, a, b)
    vp = b.vector_partition
    if not a.assembled or not (isinstance(vp, DebugArray) or (isinstance(vp, TorchDistArray) and (
            vp.size == 1 or (pv.TRANSPORT == "rccl" and context().comm is not None) or pv.TRANSPORT == "ipc"))):
        return mul5_(c, a, b, alpha, beta)
    hs = _operator_handles(a, b)
    if isinstance(vp, DebugArray):
        n = len(vp.items)
        arr = lambda xs: (C.c_void_p * n)(*xs)
        L.call("pa_mul_all", arr(hs.items), n, arr([v.h for v in c.vector_partition.items]), arr([v.h for v in vp.items]),
               float(alpha), float(beta))
    else:
        comm = context().comm.h if (vp.size > 1 and pv.TRANSPORT == "rccl") else None
        L.call("pa_mul5", hs.item, comm, c.vector_partition.item.h, vp.item.h, float(alpha), float(beta))
    return c


def mul_no_lat_c_(c: PVector, a: PSparseMatrix, b: PVector) -> PVector:
    """HPCG's mul_no_lat!(c,a,b) (HPCG/src/hpcg_utils.jl:6-17) queued by one library call per part (pa_mul_no_lat): the
    exchange completes before own x own.  Falls back to mul_no_overlap_ where the operator-level call does not apply."""
    from .primitives import DebugArray, TorchDistArray, local_items
    from . import p_vector as pv
    _check_axes(c, a, b)
    vp = b.vector_partition
    one_per_process = isinstance(vp, TorchDistArray) and (vp.size == 1 or (pv.TRANSPORT == "rccl" and context().comm is not None) or pv.TRANSPORT == "ipc")
    single = isinstance(vp, DebugArray) and len(vp.items) == 1
    if not a.assembled or not (one_per_process or single):
        return mul_no_overlap_(c, a, b)
    h = local_items(_operator_handles(a, b))[0]
    comm = context().comm.h if (one_per_process and vp.size > 1 and pv.TRANSPORT == "rccl") else None
    L.call("pa_mul_no_lat", h, comm, local_items(c.vector_partition)[0].h, local_items(vp)[0].h)
    return c


def mul_dot_(c: PVector, a: PSparseMatrix, b: PVector, slot: int) -> bool:
    """c = a*b as mul! does it AND slot <- dot(b,c), the dot accumulated inside the product kernels (pa_mul_dot /
    pa_mul_all_dot): the c = A*u, u'c pair of a CG iteration (HPCG/src/ref_cg.jl:59-60) without the dot's pass over u
    and c.  c is bit-identical to mul!'s; the dot agrees with dot(b,c) to rounding (another summation order).  Returns
    False -- having done nothing -- when the operator-level call does not apply (sub-assembled matrix, host-staged
    transport, rectangular operator): the caller then uses mul_c_ and dot_slot."""
    from .primitives import DebugArray, TorchDistArray
    from . import p_vector as pv
    _check_axes(c, a, b)
    vp = b.vector_partition
    if not a.assembled or not (isinstance(vp, DebugArray) or (isinstance(vp, TorchDistArray) and (
            vp.size == 1 or (pv.TRANSPORT == "rccl" and context().comm is not None)))):
        return False                     # (the ipc transport has no device-side all-reduce for the slot: host dots)
    if isinstance(vp, DebugArray) and len({id(v.ctx) for v in vp.items}) > 1:
        return False                     # (one context per part: the slot of pa_mul_all_dot lives in ONE context)
    from .primitives import local_items
    if any(bv.n_own != cv.n_own for bv, cv in zip(local_items(vp), local_items(c.vector_partition))):
        return False
    hs = _operator_handles(a, b)
    if isinstance(vp, DebugArray):
        n = len(vp.items)
        arr = lambda xs: (C.c_void_p * n)(*xs)
        L.call("pa_mul_all_dot", arr(hs.items), n, arr([v.h for v in c.vector_partition.items]), arr([v.h for v in vp.items]), slot)
    else:
        comm = context().comm.h if (vp.size > 1 and pv.TRANSPORT == "rccl") else None
        L.call("pa_mul_dot", hs.item, comm, c.vector_partition.item.h, vp.item.h, slot, 0)
        pv._slot_allreduce(vp, slot)
    return True


def mul5_(c: PVector, a: PSparseMatrix, b: PVector, alpha, beta) -> PVector:
    """mul!(c,a,b,alpha,beta) (src/p_sparse_matrix.jl:2105-2142), assembled and sub-assembled."""
    _check_axes(c, a, b)
    t = consistent_(b)
    pmap(lambda cv, blk, bv: spmv_(cv, blk.own_own, bv, L.SEG_OWN, L.SEG_OWN, alpha, beta),
         c.vector_partition, a.matrix_partition, b.vector_partition)
    if not a.assembled:
        pmap(lambda cv, blk, bv: spmv_(cv, blk.ghost_own, bv, L.SEG_OWN, L.SEG_GHOST, alpha, beta),
             c.vector_partition, a.matrix_partition, b.vector_partition)
    t.wait()
    pmap(lambda cv, blk, bv: spmv_(cv, blk.own_ghost, bv, L.SEG_GHOST, L.SEG_OWN, alpha, 1.0),
         c.vector_partition, a.matrix_partition, b.vector_partition)
    if not a.assembled:
        pmap(lambda cv, blk, bv: spmv_(cv, blk.ghost_ghost, bv, L.SEG_GHOST, L.SEG_GHOST, alpha, 1.0),
             c.vector_partition, a.matrix_partition, b.vector_partition)
        assemble_(c).wait()
    return c


def renumber_for_locality(a: PSparseMatrix, force=False, min_gain=2.0):
    """Library-side renumbering of a matrix whose own x own blocks have no locality (BASELINE config 5 says
    "unstructured ... irregular"): per part a reverse Cuthill-McKee order of the own unknowns is computed ON THE DEVICE from the
    resident block (pa_csr_locality_order), and when it narrows the band by at least `min_gain` the part's blocks are rebuilt with
    rows and own columns in that order (pa_csr_create_permuted: every row keeps its entries in their ORIGINAL order, so row sums
    keep their bits) and the part's row / column indices get a `local_to_device` map that hides the order: vectors made on the
    RETURNED matrix's partitions (pzeros(A2.col_partition), pvector_from_function(f, A2.row_partition), ...) are laid out to match,
    uploads / downloads / own_values speak the caller's local order as before, pack / unpack lists are translated when the
    exchange plan is built.  Returns a new PSparseMatrix (the old one is untouched); parts that gain nothing keep their blocks."""
    if not a.assembled:
        raise L.PAError("renumber_for_locality needs an assembled matrix")

    def one(blk, r, c):
        oo, oh = blk.own_own, blk.own_ghost
        n = r.n_own
        if oo.m != oo.n or c.n_own != n or not np.array_equal(r.own_to_global, c.own_to_global):
            raise L.PAError("renumber_for_locality: rows and columns must share their own indices (a square operator)")
        newpos = np.zeros(max(n, 1), np.int32)
        b0, b1 = C.c_int64(), C.c_int64()
        L.call("pa_csr_locality_order", oo.h, L.ptr(newpos), C.byref(b0), C.byref(b1))
        newpos = newpos[:n]
        if not force and b1.value * min_gain > b0.value:
            return blk, r, c, (b0.value, b0.value)
        out = []
        for B, colpos in ((oo, newpos), (oh, None)):
            h = C.c_void_p()
            L.call("pa_csr_create_permuted", B.h, L.ptr(newpos), None if colpos is None else L.ptr(colpos), C.byref(h))
            out.append(DeviceCSR.from_handle(h, B.m, B.n, B.nnz, B.ctx))
        return SplitMatrixBlocks(out[0], out[1]), r.with_device_own_perm(newpos), c.with_device_own_perm(newpos), (b0.value, b1.value)

    from .primitives import tuple_of_arrays
    blocks, rows, cols, bands = tuple_of_arrays(pmap(one, a.matrix_partition, a.row_partition, a.col_partition))
    A2 = PSparseMatrix(blocks, rows, cols, True)
    A2.bandwidths = bands                      # per part: (before, after) max |row - col| of own x own
    return A2


def transposed_blocks(a: PSparseMatrix):
    """(A_oo', A_oh') of every part, built ON THE DEVICE from the blocks resident in HBM (pa_csr_create_transpose: decoded column
    encoding, one stable sort by column; csrc/pa_transpose.hip) -- no host copy, so generated and device-assembled matrices
    have them too.  Built once, cached on `a` (psparse!-style value updates of `a` drop the cache)."""
    if getattr(a, "_t_blocks", None) is None:
        def mk(blk, r):
            out = []
            # (a renumbered part -- renumber_for_locality -- stores its rows in another order: the transposes add in the order
            #  of the caller's rows all the same)
            perm = getattr(r, "device_own_perm", None)
            rank = None
            if perm is not None:
                rank = np.empty(len(perm), np.int32)
                rank[np.asarray(perm, np.int64)] = np.arange(len(perm), dtype=np.int32)
            for B in (blk.own_own, blk.own_ghost):
                h = C.c_void_p()
                L.call("pa_csr_create_transpose_ranked", B.h, None if rank is None else L.ptr(rank), C.byref(h))
                out.append(DeviceCSR.from_handle(h, B.n, B.m, B.nnz, B.ctx))
            return tuple(out)
        a._t_blocks = pmap(mk, a.matrix_partition, a.row_partition)
    return a._t_blocks


def mul5_transpose_(c: PVector, a: PSparseMatrix, b: PVector, alpha, beta) -> PVector:
    """mul!(c,transpose(a),b,alpha,beta) (src/p_sparse_matrix.jl:2144-2162), a assembled; c lives on axes(a,2), b on
    axes(a,1): ghost(c) = alpha*A_oh'*own(b), assemble!(c) started, own(c) = beta*own(c) + alpha*A_oo'*own(b) under the
    exchange, wait.  One library call per process where the operator-level call applies (pa_mul5_transpose[_all])."""
    from .primitives import DebugArray, TorchDistArray
    from . import p_vector as pv
    if not a.assembled:
        raise L.PAError("mul!(c,transpose(a),b,...) needs an assembled matrix (@assert a.assembled, :2146)")

    def chk(ci, coli, bi, ri):
        if ci.n_own != coli.n_own or ci.n_ghost != coli.n_ghost or bi.n_own != ri.n_own:
            raise L.PAError("c must live on axes(a,2) and b on axes(a,1)")
        for u, v in ((ci, coli), (bi, ri)):
            pu, pv_ = getattr(u, "device_own_perm", None), getattr(v, "device_own_perm", None)
            if not (pu is pv_ or (pu is not None and pv_ is not None and np.array_equal(pu, pv_))):
                raise L.PAError("the vectors' device layout is not the matrix's (renumber_for_locality returns new partitions)")
    pmap(chk, c.index_partition, a.col_partition, b.index_partition, a.row_partition)
    tb = transposed_blocks(a)
    vp = c.vector_partition
    direct = isinstance(vp, DebugArray) or (isinstance(vp, TorchDistArray) and (
        vp.size == 1 or (pv.TRANSPORT == "rccl" and context().comm is not None) or pv.TRANSPORT == "ipc"))
    if direct:
        cache = c.__dict__.setdefault("_pa_matrices_t", {})
        if id(a) not in cache:
            def mk(t, plan):
                h = C.c_void_p()
                L.call("pa_matrix_create_transposed", context().h, t[0].h, t[1].h, plan, C.byref(h))
                return h
            cache[id(a)] = _OperatorHandles(pmap(mk, tb, c.cache.plans), a)
            import weakref
            if getattr(a, "_t_users", None) is None:
                a._t_users = weakref.WeakSet()
            a._t_users.add(c)
        hs = cache[id(a)].handles
        if isinstance(vp, DebugArray):
            n = len(vp.items)
            arr = lambda xs: (C.c_void_p * n)(*xs)
            L.call("pa_mul5_transpose_all", arr(hs.items), n, arr([v.h for v in vp.items]), arr([v.h for v in b.vector_partition.items]),
                   float(alpha), float(beta))
        else:
            comm = context().comm.h if (vp.size > 1 and pv.TRANSPORT == "rccl") else None
            L.call("pa_mul5_transpose", hs.item, comm, vp.item.h, b.vector_partition.item.h, float(alpha), float(beta))
        return c
    # (transports the library does not drive itself: the same kernels composed here)
    pmap(lambda cv, t, bv: spmv_(cv, t[1], bv, L.SEG_OWN, L.SEG_GHOST, alpha, 0.0), c.vector_partition, tb, b.vector_partition)
    tsk = assemble_(c)
    pmap(lambda cv, t, bv: spmv_(cv, t[0], bv, L.SEG_OWN, L.SEG_OWN, alpha, beta), c.vector_partition, tb, b.vector_partition)
    tsk.wait()
    return c


def mul_no_overlap_(c: PVector, a: PSparseMatrix, b: PVector) -> PVector:
    """HPCG.mul_no_lat! ordering (HPCG/src/hpcg_utils.jl:6-17): blocking consistent!, then the product.
    (Same blocks, same per-row order own-then-ghost, so same bits; kept for the overlap on/off comparison.)"""
    _check_axes(c, a, b)
    consistent_(b).wait()
    pmap(lambda cv, blk, bv: spmv_(cv, blk.own_own, bv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0),
         c.vector_partition, a.matrix_partition, b.vector_partition)
    pmap(lambda cv, blk, bv: spmv_(cv, blk.own_ghost, bv, L.SEG_GHOST, L.SEG_OWN, 1.0, 1.0),
         c.vector_partition, a.matrix_partition, b.vector_partition)
    return c


# ----------------------------------------------------------------------------------------------
# disassembled COO -> assembled split matrix (the default psparse route; FEM-style, BASELINE config 5)
# ----------------------------------------------------------------------------------------------
def _coo(A: HostCSR):
    """findnz / nziterator of a CSR block: row-major, ascending columns (src/sparse_utils.jl:96-123)."""
    rows = np.repeat(np.arange(1, A.m + 1, dtype=I64), np.diff(A.rowptr.astype(I64)))
    return rows, A.colval.astype(I64), A.nzval


def _split4(A: HostCSR, rows, cols):
    """split_format_locally with ghost rows (src/p_sparse_matrix.jl:823-899): own_own, own_ghost, ghost_own, ghost_ghost."""
    oo, oh = split_format_locally(A, rows, cols)
    i, j, v = _coo(A)
    g = i > rows.n_own
    i, j, v = i[g] - rows.n_own, j[g], v[g]
    own = j <= cols.n_own
    ho = compresscoo(i[own], j[own], v[own], rows.n_ghost, cols.n_own)
    hh = compresscoo(i[~own], j[~own] - cols.n_own, v[~own], rows.n_ghost, cols.n_ghost)
    return oo, oh, ho, hh


def _group_by_owner(owner, parts_snd, arrays, with_order=False):
    slot = np.searchsorted(parts_snd, owner)
    order = np.argsort(slot, kind="stable")
    cuts = np.searchsorted(slot[order], np.arange(len(parts_snd) + 1))
    out = [[a[order][cuts[k]:cuts[k + 1]] for k in range(len(parts_snd))] for a in arrays]
    return (out, order, cuts) if with_order else out


def _nzindex(A: HostCSR, i, j):
    """nzindex(A,i,j) (src/sparse_utils.jl:256-278), vectorised: 1-based position of (i,j) in nonzeros(A), 0 if absent."""
    ri, cj, _ = _coo(A)
    keys = (ri - 1) * max(A.n, 1) + (cj - 1)                       # ascending: rows ascending, columns sorted inside a row
    q = (np.asarray(i, I64) - 1) * max(A.n, 1) + (np.asarray(j, I64) - 1)
    if len(keys) == 0:
        return np.zeros(len(q), I64)
    pos = np.clip(np.searchsorted(keys, q), 0, len(keys) - 1)
    return np.where(keys[pos] == q, pos + 1, 0)


def _assembly_snd(ghost_own, ghost_ghost, ps, r, c):
    """setup_cache_snd (src/p_sparse_matrix.jl:1598-1650): the ghost rows' stored entries -- ghost_own's, then ghost_ghost's, each
    in CSR order -- as global triplets grouped by the owner of their row (stable inside a group)."""
    gi, gj, gv = _coo(ghost_own)
    hi, hj, hv = _coo(ghost_ghost)
    ii = np.concatenate([gi, hi])
    gI = r.ghost_to_global[ii - 1]
    gJ = np.concatenate([c.own_to_global[gj - 1], c.ghost_to_global[hj - 1]])
    gV = np.concatenate([gv, hv])
    (a, b, v), order, cuts = _group_by_owner(r.ghost_to_owner[ii - 1], np.asarray(ps), [gI, gJ, gV], with_order=True)
    return a, b, v, (order + 1, cuts + 1)            # k_snd (1-based position in [ghost_own|ghost_ghost] nz), ptrs


def _disassembled_device_applies(rows, cols, I, J):
    """The device route of psparse + assemble (csrc/pa_assemble.hip, pa_coo_subassemble / pa_coo_assemble_finish): block
    partitions without ghosts for rows and columns, every part with triplets, ids >= 1, a GPU; PA_SETUP_DEVICE=0: host route."""
    if not _device_assembly_applies(rows, I):
        return False
    from .primitives import local_items
    def ids_ok(a):
        return len(a) > 0 and (getattr(a, "ids_from_one", False) or int(np.min(a)) >= 1)      # (DeviceTriplets: node ids, >= 1 by construction)
    return (all(c.kind == "block" and c.n_ghost == 0 and c.n_own > 0 for c in local_items(cols))
            and all(ids_ok(i) and ids_ok(j) for i, j in zip(local_items(I), local_items(J))))


def psparse_disassembled_device(I, J, V, rows, cols, keep_host=False, reuse=False):
    """psparse(I,J,V,rows,cols) with the default flags, then assemble (src/p_sparse_matrix.jl:1150-1219,1590-1756), with
    everything per triplet on the device.  reuse=True: also the cache of psparse! (MatrixReassemblyCache), its per-triplet part --
    where every COO value goes -- composed on the device from what the two assembly steps remember (pa_coo_reuse_scatter); only
    the part's surface (the ghost rows' entries, the received triplets) passes through the host, as it does for the exchange.  Per part: the sub-assembled local matrix by one sort (ghost rows and ghost columns in
    first-seen order); its ghost rows -- the part's surface -- come to the host and travel to their owners exactly as
    psparse_assemble_host sends them; the own rows, still in HBM, and what arrived go through the assembled route.  Same
    blocks, same ghost order as the host route, bit for bit (tests/test_gpu_setup.py)."""
    from .primitives import exchange, ExchangeGraph, DebugArray, tuple_of_arrays
    from .p_range import assembly_neighbors, LocalIndices

    def box(ind):
        return (len(ind.n), np.array(ind.n, I64), np.array([a for a, _ in ind.ranges], I64), np.array([b for _, b in ind.ranges], I64))

    def with_ghosts(ind, gids):
        owners = find_owner(DebugArray([ind]), DebugArray([gids])).items[0]
        return LocalIndices(ind.n_global, ind.part, np_=ind.np_, n=ind.n, ranges=ind.ranges, starts=ind.starts,
                            ghost_to_global=gids, ghost_to_owner=owners)

    def sub(Ii, Ji, Vi, r, c):
        on_device = hasattr(Ii, "ptr") and not reuse              # (gallery.DeviceTriplets: generated in HBM, taken as they are)
        if on_device:
            pI, pJ, pV = Ii.ptr, Ji.ptr, Vi.ptr
        else:
            Ii, Ji, Vi = [a.download() if hasattr(a, "download") else a for a in (Ii, Ji, Vi)]
            Ii, Ji, Vi = np.ascontiguousarray(Ii, I64), np.ascontiguousarray(Ji, I64), np.ascontiguousarray(Vi, F64)
            pI, pJ, pV = L.ptr(Ii), L.ptr(Ji), L.ptr(Vi)
        D, nr, lor, hir = box(r)
        Dc, ncg, loc, hic = box(c)
        if D != Dc:
            raise L.PAError("row and column partitions of different dimension")
        h = C.c_void_p()
        L.call("pa_coo_keep_input_slots", context().h, 1 if reuse else 0)
        try:
            L.call("pa_coo_subassemble", context().h, len(Ii), pI, pJ, pV, D, L.ptr(nr), L.ptr(lor), L.ptr(hir),
                   L.ptr(ncg), L.ptr(loc), L.ptr(hic), C.byref(h))
        finally:
            L.call("pa_coo_keep_input_slots", context().h, 0)
        try:
            v = [C.c_int64() for _ in range(4)]
            L.call("pa_coo_subassembly_info", h, *[C.byref(x) for x in v])
            ngr, ngc, _n_own_entries, nge = [x.value for x in v]
            cg = np.zeros(ngc, I64)
            L.call("pa_coo_assembly_ghosts", h, L.ptr(cg))
            rg, gr, gc, gv = np.zeros(ngr, I64), np.zeros(nge, I32), np.zeros(nge, I32), np.zeros(nge, F64)
            L.call("pa_coo_subassembly_ghost_rows", h, L.ptr(rg), L.ptr(gr), L.ptr(gc), L.ptr(gv))
        except Exception:
            L.lib.pa_coo_assembly_destroy(h)
            raise
        r_sa, c_sa = with_ghosts(r, rg), with_ghosts(c, cg)
        own = gc < c.n_own                                            # (sorted by (row, column): both halves stay in CSR order)

        def csr(rows_, cols_, vals_, n_cols):
            rp = np.zeros(ngr + 1, np.int64)
            np.add.at(rp, rows_.astype(np.int64) + 1, 1)
            return HostCSR(ngr, n_cols, (np.cumsum(rp) + 1).astype(I32), (cols_ + 1).astype(I32), vals_.copy())
        # position of ghost-row entry k in [nonzeros(ghost_own) | nonzeros(ghost_ghost)] (both halves keep the (row, column) order)
        gslot = np.where(own, np.cumsum(own) - 1, int(own.sum()) + np.cumsum(~own) - 1).astype(I32)
        return h, r_sa, c_sa, csr(gr[own], gc[own], gv[own], c.n_own), csr(gr[~own], gc[~own] - c.n_own, gv[~own], ngc), gslot

    hs, rows_sa, cols_sa, g_own, g_ghost, gslots = tuple_of_arrays(pmap(sub, I, J, V, rows, cols))
    try:
        parts_snd, parts_rcv = assembly_neighbors(rows_sa)
        I_snd, J_snd, V_snd, ksnd = tuple_of_arrays(pmap(_assembly_snd, g_own, g_ghost, parts_snd, rows_sa, cols_sa))
        graph = ExchangeGraph(parts_snd, parts_rcv)
        I_rcv, J_rcv, V_rcv = exchange(I_snd, graph), exchange(J_snd, graph), exchange(V_snd, graph)

        def finish(h, Ir, Jr, Vr, r, c, gslot, ps, pr, ks, n_in):
            cat = lambda xs, dt: np.ascontiguousarray(np.concatenate([np.asarray(x, dt) for x in xs]) if len(xs) else np.zeros(0, dt))  # noqa: E731
            rcv_ptrs = np.concatenate([[1], 1 + np.cumsum([len(np.atleast_1d(x)) for x in Ir])]).astype(I32)
            Ir, Jr, Vr = cat(Ir, I64), cat(Jr, I64), cat(Vr, F64)
            f = C.c_void_p()
            L.call("pa_coo_keep_input_slots", context().h, 1 if reuse else 0)
            try:
                L.call("pa_coo_assemble_finish", h, len(Ir), L.ptr(Ir) if len(Ir) else None, L.ptr(Jr) if len(Ir) else None,
                       L.ptr(Vr) if len(Ir) else None, C.byref(f))
            finally:
                L.call("pa_coo_keep_input_slots", context().h, 0)
            try:
                v = [C.c_int64() for _ in range(5)]
                L.call("pa_coo_assembly_info", f, *[C.byref(x) for x in v], None)
                n_rows, n_own_cols, n_ghost, nnz_oo, nnz_oh = [x.value for x in v]
                ghosts = np.zeros(n_ghost, I64)
                L.call("pa_coo_assembly_ghosts", f, L.ptr(ghosts))
                c_fa = with_ghosts(c, ghosts)
                a, b = C.c_void_p(), C.c_void_p()
                L.call("pa_coo_assembly_blocks", f, C.byref(a), C.byref(b))
                blk = SplitMatrixBlocks(DeviceCSR.from_handle(a, n_rows, n_own_cols, nnz_oo), DeviceCSR.from_handle(b, n_rows, n_ghost, nnz_oh))
                host = None
                if keep_host:
                    host = []
                    for which, (ncol, nnz) in enumerate(((n_own_cols, nnz_oo), (n_ghost, nnz_oh))):
                        H = HostCSR(n_rows, ncol, np.zeros(n_rows + 1, I32), np.zeros(nnz, I32), np.zeros(nnz, F64))
                        L.call("pa_coo_assembly_download", f, which, L.ptr(H.rowptr), L.ptr(H.colval), L.ptr(H.nzval))
                        host.append(H)
                    host = tuple(host)
                cache = None
                if reuse:
                    # W = [nonzeros(own_own) | nonzeros(own_ghost) || the ghost rows' entries]; the plan that assembles it: idx_snd = the
                    # ghost-row slots in the order they are sent (k_snd), idx_rcv = the slots the received triplets landed in (k_rcv)
                    from .p_vector import DeviceVector, plan_info
                    n_own_vals, n_ghost_vals = nnz_oo + nnz_oh, len(gslot)
                    sc, k_rcv = C.c_void_p(), np.zeros(max(len(Ir), 1), I32)
                    L.call("pa_coo_reuse_scatter", h, f, L.ptr(gslot) if n_ghost_vals else None, C.byref(sc), len(Ir), L.ptr(k_rcv))
                    k_snd, ptrs_snd = ks
                    ns32, nr32 = np.ascontiguousarray(ps, I32), np.ascontiguousarray(pr, I32)
                    idx_snd = np.ascontiguousarray(n_own_vals + k_snd, I32)
                    idx_rcv = np.ascontiguousarray(k_rcv[:len(Ir)], I32)
                    p_snd, p_rcv = np.ascontiguousarray(ptrs_snd, I32), np.ascontiguousarray(rcv_ptrs, I32)
                    plan = C.c_void_p()
                    L.call("pa_plan_create", context().h, r.part, n_own_vals + n_ghost_vals, len(ns32), L.ptr(ns32), L.ptr(p_snd),
                           L.ptr(idx_snd), len(nr32), L.ptr(nr32), L.ptr(p_rcv), L.ptr(idx_rcv), 1, C.byref(plan))
                    plan_info[plan.value] = dict(
                        snd=[(int(q), int(p_snd[k]) - 1, int(p_snd[k + 1]) - 1) for k, q in enumerate(ns32)],
                        rcv=[(int(q), int(p_rcv[k]) - 1, int(p_rcv[k + 1]) - 1) for k, q in enumerate(nr32)])
                    cache = (plan, sc, DeviceVector(n_own_vals, n_ghost_vals), DeviceVector(int(n_in), 0), nnz_oo)
            finally:
                L.lib.pa_coo_assembly_destroy(f)
            return blk, c_fa, host, cache

        out = pmap(finish, hs, I_rcv, J_rcv, V_rcv, rows, cols, gslots, parts_snd, parts_rcv, ksnd, pmap(len, I))
    finally:
        pmap(lambda h: L.lib.pa_coo_assembly_destroy(h), hs)
    blocks, cols_fa, host, caches = tuple_of_arrays(out)
    C_ = PSparseMatrix(blocks, rows, cols_fa, True, host if keep_host else None)
    if not reuse:
        return C_
    plans, scs, W, Vd, nnz_oo = tuple_of_arrays(caches)
    from .p_vector import connect_ipc
    connect_ipc(plans)
    return C_, MatrixReassemblyCache(plans, scs, W, Vd, nnz_oo)


def psparse_assemble_host(blocks4, rows_sa, cols_sa, rows, reuse=False):
    """assemble(B,rows) for a split-format sub-assembled matrix, first time (psparse_assemble_impl,
    src/p_sparse_matrix.jl:1590-1756): ghost-row triplets travel to the owners of their rows, are appended to the
    owners' COO lists, the ghost columns are renumbered (union_ghost) and everything is compressed with +.
    Host-side set-up; returns (own_own, own_ghost) HostCSR per part and the final column partition."""
    from .primitives import exchange, ExchangeGraph
    from .p_range import assembly_neighbors, LocalIndices
    parts_snd, parts_rcv = assembly_neighbors(rows_sa)

    def setup_snd(blk, ps, r, c):
        return _assembly_snd(blk[2], blk[3], ps, r, c)

    from .primitives import tuple_of_arrays
    I_snd, J_snd, V_snd, ksnd = tuple_of_arrays(pmap(setup_snd, blocks4, parts_snd, rows_sa, cols_sa))
    graph = ExchangeGraph(parts_snd, parts_rcv)
    I_rcv, J_rcv, V_rcv = exchange(I_snd, graph), exchange(J_snd, graph), exchange(V_snd, graph)   # :1734-1736

    def own_triplets(blk, Ir, Jr, Vr, r, c):                        # setup_own_triplets :1656-1689
        cat = lambda xs, dt: np.concatenate([np.asarray(x, dt) for x in xs]) if len(xs) else np.zeros(0, dt)  # noqa: E731
        Ir_lists = list(Ir)
        Ir, Jr, Vr = cat(Ir, I64), cat(Jr, I64), cat(Vr, F64)
        ooI, ooJ, ooV = _coo(blk[0])
        ohI, ohJ, ohV = _coo(blk[1])
        lj = c.global_to_local(Jr).astype(I64)
        is_own = (lj >= 1) & (lj <= c.n_own)
        li = r.global_to_local(Ir).astype(I64)
        oo = (np.concatenate([ooI, li[is_own]]), np.concatenate([ooJ, lj[is_own]]), np.concatenate([ooV, Vr[is_own]]))
        og = (np.concatenate([ohI, li[~is_own]]), np.concatenate([c.ghost_to_global[ohJ - 1], Jr[~is_own]]),
              np.concatenate([ohV, Vr[~is_own]]))
        rcv_ptrs = np.concatenate([[1], 1 + np.cumsum([len(np.atleast_1d(x)) for x in Ir_lists])]).astype(I32)
        return oo, og, og[1], (li, Jr, rcv_ptrs)

    oo, og, Jg, rcvinfo = tuple_of_arrays(pmap(own_triplets, blocks4, I_rcv, J_rcv, V_rcv, rows_sa, cols_sa))
    J_owner = find_owner(cols_sa, Jg)
    cols0 = pmap(lambda c: LocalIndices(c.n_global, c.part, np_=c.np_, n=c.n, ranges=c.ranges, starts=c.starts), cols_sa)  # remove_ghost
    cols_fa = pmap(union_ghost, cols0, Jg, J_owner)

    def finalize(oo_, og_, r, c):                                   # finalize_values :1690-1723
        gj = c.global_to_local(og_[1]).astype(I64) - c.n_own        # map_global_to_ghost!
        return (compresscoo(oo_[0], oo_[1], oo_[2], r.n_own, c.n_own), compresscoo(og_[0], gj, og_[2], r.n_own, c.n_ghost))

    host = pmap(finalize, oo, og, rows, cols_fa)
    if not reuse:
        return host, cols_fa
    info = dict(parts_snd=parts_snd, parts_rcv=parts_rcv, ksnd=ksnd, rcvinfo=rcvinfo)
    return host, cols_fa, info


class MatrixReassemblyCache:
    """cache of psparse(...;reuse=true) for the device path: what psparse!(C,V,cache) needs to turn new COO values
    into the values of the assembled blocks WITHOUT leaving HBM (K7).  The stored values of one part are treated as a
    vector  W = [ nonzeros(C.own_own) | nonzeros(C.own_ghost) || nonzeros(B.ghost_own) | nonzeros(B.ghost_ghost) ]
    so that the reference's three steps are existing device primitives:
      sparse_matrix!(A,V,K) + split_format! + setup_sa  -> one deterministic scatter-add  W[dest[p]] += V[p]   (pa_scatter)
      psparse_assemble_impl! (:1762-1816)               -> assemble! of W over a pa_plan with idx_snd = ghost-row slots
                                                           (k_snd) and idx_rcv = own slots (k_rcv): pack, exchange, += in
                                                           ascending p
      nonzeros(C.blocks) .= W[own part]                 -> pa_csr_update_values_from."""

    def __init__(self, plans, scatters, W, Vdev, nnz_oo):
        self.plans, self.scatters, self.W, self.Vdev, self.nnz_oo = plans, scatters, W, Vdev, nnz_oo


def psparse_disassembled(I, J, V, rows, cols, keep_host=False, reuse=False, assemble=True):
    """psparse(SparseMatrixCSR{1,Float64,Int32},I,J,V,rows,cols)|>fetch with the DEFAULT flags
    (src/p_sparse_matrix.jl:1150-1219): every part may hold entries of rows it does not own (FEM assembly loops);
    find_owner/union_ghost for rows and columns, local compress + split, then assemble onto `rows`."""
    if assemble and _disassembled_device_applies(rows, cols, I, J) and (not reuse or os.environ.get("PA_REUSE_DEVICE", "1") != "0"):
        return psparse_disassembled_device(I, J, V, rows, cols, keep_host=keep_host, reuse=reuse)
    I, J, V = [pmap(lambda a: a.download() if hasattr(a, "download") else a, t) for t in (I, J, V)]      # (DeviceTriplets on a host route)
    I_owner = find_owner(rows, I)
    J_owner = find_owner(cols, J)
    rows_sa = pmap(union_ghost, rows, I, I_owner)
    cols_sa = pmap(union_ghost, cols, J, J_owner)

    def local(Ii, Ji, Vi, r, c):
        A = sparse_matrix(r.global_to_local(Ii), c.global_to_local(Ji), Vi, r.n_local, c.n_local)
        return _split4(A, r, c)

    blocks4 = pmap(local, I, J, V, rows_sa, cols_sa)
    if not assemble:
        # psparse(...;assemble=false): the sub-assembled matrix itself (ghost rows kept; mul! assembles the product,
        # src/p_sparse_matrix.jl:2121-2139; test/fem_example.jl:331-338)
        dev = pmap(lambda b: SplitMatrixBlocks(DeviceCSR(b[0]), DeviceCSR(b[1]), DeviceCSR(b[2]), DeviceCSR(b[3])), blocks4)
        return PSparseMatrix(dev, rows_sa, cols_sa, False, blocks4 if keep_host else None)
    if not reuse:
        host, cols_fa = psparse_assemble_host(blocks4, rows_sa, cols_sa, rows)
        dev = pmap(lambda h: SplitMatrixBlocks(DeviceCSR(h[0]), DeviceCSR(h[1])), host)
        return PSparseMatrix(dev, rows, cols_fa, True, host if keep_host else None)
    host, cols_fa, info = psparse_assemble_host(blocks4, rows_sa, cols_sa, rows, reuse=True)
    dev = pmap(lambda h: SplitMatrixBlocks(DeviceCSR(h[0]), DeviceCSR(h[1])), host)
    C_ = PSparseMatrix(dev, rows, cols_fa, True, host if keep_host else None)
    from .p_vector import DeviceVector, plan_info

    def slot_in_C(h, c_fa, li, gj):
        """1-based position of entry (own row li, global column gj) in [nonzeros(own_own) | nonzeros(own_ghost)]."""
        lc = c_fa.global_to_local(gj).astype(I64)
        own = lc <= c_fa.n_own
        out = np.zeros(len(li), I64)
        out[own] = _nzindex(h[0], li[own], lc[own])
        out[~own] = h[0].nnz + _nzindex(h[1], li[~own], lc[~own] - c_fa.n_own)
        return out

    def build(Ii, Ji, b4, h, r_sa, c_sa, c_fa, ps, pr, ks, rcv):
        n_own_vals = h[0].nnz + h[1].nnz
        n_ghost_vals = b4[2].nnz + b4[3].nnz
        li = r_sa.global_to_local(Ii).astype(I64)
        lj = c_sa.global_to_local(Ji).astype(I64)
        dest = np.zeros(len(li), I64)
        ok = (li >= 1) & (lj >= 1)
        ownrow = ok & (li <= r_sa.n_own)
        dest[ownrow] = slot_in_C(h, c_fa, li[ownrow], np.asarray(Ji, I64)[ownrow])
        gr = ok & ~ownrow
        gi = li[gr] - r_sa.n_own
        gl = lj[gr]
        d = np.zeros(len(gi), I64)
        oc = gl <= c_sa.n_own
        d[oc] = n_own_vals + _nzindex(b4[2], gi[oc], gl[oc])
        d[~oc] = n_own_vals + b4[2].nnz + _nzindex(b4[3], gi[~oc], gl[~oc] - c_sa.n_own)
        dest[gr] = d
        assert np.all(dest[ok] > 0)
        sc = C.c_void_p()
        d32 = np.ascontiguousarray(dest, I32)
        L.call("pa_scatter_create", context().h, n_own_vals + n_ghost_vals, len(d32), L.ptr(d32), 1, C.byref(sc))
        k_snd, ptrs_snd = ks
        rli, rJ, ptrs_rcv = rcv
        k_rcv = slot_in_C(h, c_fa, rli, rJ)
        assert np.all(k_rcv > 0)
        ns32, nr32 = np.ascontiguousarray(ps, I32), np.ascontiguousarray(pr, I32)
        idx_snd = np.ascontiguousarray(n_own_vals + k_snd, I32)
        idx_rcv = np.ascontiguousarray(k_rcv, I32)
        p_snd, p_rcv = np.ascontiguousarray(ptrs_snd, I32), np.ascontiguousarray(ptrs_rcv, I32)
        plan = C.c_void_p()
        L.call("pa_plan_create", context().h, r_sa.part, n_own_vals + n_ghost_vals, len(ns32), L.ptr(ns32), L.ptr(p_snd),
               L.ptr(idx_snd), len(nr32), L.ptr(nr32), L.ptr(p_rcv), L.ptr(idx_rcv), 1, C.byref(plan))
        plan_info[plan.value] = dict(
            snd=[(int(q), int(p_snd[k]) - 1, int(p_snd[k + 1]) - 1) for k, q in enumerate(ns32)],
            rcv=[(int(q), int(p_rcv[k]) - 1, int(p_rcv[k + 1]) - 1) for k, q in enumerate(nr32)])
        return plan, sc, DeviceVector(n_own_vals, n_ghost_vals), DeviceVector(len(d32), 0), h[0].nnz

    from .primitives import tuple_of_arrays
    plans, scs, W, Vd, nnz_oo = tuple_of_arrays(pmap(build, I, J, blocks4, host, rows_sa, cols_sa, cols_fa,
                                                     info["parts_snd"], info["parts_rcv"], info["ksnd"], info["rcvinfo"]))
    from .p_vector import connect_ipc
    connect_ipc(plans)                                 # (PA_TRANSPORT=ipc: the value exchange of psparse! pushes too)
    return C_, MatrixReassemblyCache(plans, scs, W, Vd, nnz_oo)


def psparse_(C_: PSparseMatrix, V, cache: MatrixReassemblyCache) -> Task:
    """psparse!(C,V,cache) (src/p_sparse_matrix.jl:1291-1305): same pattern, new COO values; everything after the
    upload of V runs on the device (see MatrixReassemblyCache).  Returns a task; wait() it before using C."""
    from .p_vector import assemble_impl

    class _P:                      # what assemble_impl needs from a cache
        pass

    pc = _P()
    pc.plans = cache.plans
    pmap(lambda vd, v: vd.upload(np.ascontiguousarray(v, F64)), cache.Vdev, V)
    pmap(lambda sc, w, vd: L.call("pa_scatter_add", sc, w.h, vd.h, 1), cache.scatters, cache.W, cache.Vdev)
    t = assemble_impl(L.ASSEMBLE, cache.W, pc)

    def finish():
        t.wait()
        C_._t_blocks = None             # (transposed copies hold the old values: rebuilt on the next transpose product)
        for v in list(getattr(C_, "_t_users", ())):
            v.__dict__.get("_pa_matrices_t", {}).pop(id(C_), None)
        pmap(lambda blk, w, k: (L.call("pa_csr_update_values_from", blk.own_own.h, w.h, 0),
                                L.call("pa_csr_update_values_from", blk.own_ghost.h, w.h, int(k))),
             C_.matrix_partition, cache.W, cache.nnz_oo)

    return Task(finish, C_)


def psystem(I, J, V, I2, V2, rows, cols, reuse=False, assemble=True):
    """psystem(I,J,V,I2,V2,rows,cols;reuse,assemble)|>fetch (src/p_sparse_matrix.jl:2475-2528): the matrix and the
    right-hand side of a linear system from the COO data of one cell loop, default (disassembled) flags.
    Returns A, b (and the pair of caches for psystem_ when reuse)."""
    from .p_vector import pvector_disassembled
    if reuse and not assemble:
        raise L.PAError("psystem(reuse=true, assemble=false): the re-assembly caches exist for the assembled system only")
    if reuse:
        A, cacheA = psparse_disassembled(I, J, V, rows, cols, reuse=True, assemble=assemble)
        b, cacheb = pvector_disassembled(I2, V2, rows, reuse=True, assemble=assemble)
        return A, b, (cacheA, cacheb)
    return (psparse_disassembled(I, J, V, rows, cols, assemble=assemble),
            pvector_disassembled(I2, V2, rows, assemble=assemble))


def psystem_(A, b, V, V2, cache):
    """psystem!(A,b,V,V2,cache) (src/p_sparse_matrix.jl:2530-2539): psparse! + pvector! on the device."""
    from .p_vector import pvector_
    cacheA, cacheb = cache
    psparse_(A, V, cacheA).wait()
    pvector_(b, V2, cacheb)
    return A, b
