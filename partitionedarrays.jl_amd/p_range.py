"""Index partitions (host side): who owns which global id, ghost numbering, neighbours.

Mirrors /root/reference/src/p_range.jl for what the mul!/consistent!/assemble! path needs.  All
ids are 1-based exactly as the reference stores them, so the arrays built here are what the Julia
glue would hand over the C ABI (index_base = 1).  The heavy loops (find_owner, first-seen ghost
filtering, global->local) run in native code (csrc/pa_host.cpp).
"""
from __future__ import annotations

import ctypes as C
import itertools

import numpy as np

from . import _lib as L
from .primitives import (exchange, exchange_graph, linear_indices, pmap, tuple_of_arrays,
                         getany, gather)

I32, I64 = np.int32, np.int64


def length_to_ptrs(ptrs):
    """src/jagged_array.jl:11-18."""
    ptrs[0] = 1
    np.cumsum(ptrs, out=ptrs)
    return ptrs


class JaggedArray:
    """src/jagged_array.jl:107-122: `data` + 1-based `ptrs` (Int32)."""

    def __init__(self, data, ptrs):
        self.data = np.ascontiguousarray(data)
        self.ptrs = np.ascontiguousarray(ptrs, dtype=I32)

    @staticmethod
    def from_lists(vv, dtype):
        ptrs = np.zeros(len(vv) + 1, dtype=I32)
        for i, v in enumerate(vv):
            ptrs[i + 1] = len(v)
        length_to_ptrs(ptrs)
        data = np.concatenate([np.asarray(v, dtype=dtype) for v in vv]) if len(vv) else np.zeros(0, dtype)
        return JaggedArray(data.astype(dtype), ptrs)

    def __len__(self):
        return len(self.ptrs) - 1

    def __getitem__(self, i):
        return self.data[self.ptrs[i] - 1: self.ptrs[i + 1] - 1]

    def tolists(self):
        return [self[i].tolist() for i in range(len(self))]

    def __eq__(self, o):
        return (isinstance(o, JaggedArray) and np.array_equal(self.data, o.data)
                and np.array_equal(self.ptrs, o.ptrs))


def local_range(p, np_, n, ghost=False, periodic=False):
    """src/p_range.jl:806-818 -> inclusive 1-based (start, stop)."""
    l, rem = divmod(n, np_)
    offset = l * (p - 1)
    if rem >= (np_ - p + 1):
        l += 1
        offset += p - (np_ - rem) - 1
    g = int(ghost)                      # a number of ghost layers (Bool or Integer in the reference: `1+offset-ghost`)
    start, stop = 1 + offset - g, l + offset + g
    if periodic:
        return start, stop
    return max(1, start), min(n, stop)


def _cartesian(rank, dims):
    r = rank - 1
    out = []
    for d in dims:
        out.append(r % d + 1)
        r //= d
    return tuple(out)


def _linear(ci, dims):
    lin, stride = 0, 1
    for c, d in zip(ci, dims):
        lin += (c - 1) * stride
        stride *= d
    return lin + 1


class LocalIndices:
    """One part of an index partition (AbstractLocalIndices, src/p_range.jl:32-160).

    kind == "block": LocalIndicesWithConstantBlockSize / VariableBlockSize (src/p_range.jl:1575,1632):
        own ids = a box of the global grid (column-major), local ids = [own | ghost] (:1711-1720).
    kind == "generic": LocalIndices / PermutedLocalIndices (:1100,:1372): arbitrary local order.
    """

    def __init__(self, n_global, part, *, np_=None, n=None, ranges=None, starts=None,
                 ghost_to_global=None, ghost_to_owner=None, local_to_global=None, local_to_owner=None, is_own=None):
        self.n_global = int(n_global)
        self.part = int(part)
        self.cache = {}                       # AssemblyCache (src/p_range.jl:354-359)
        if ranges is not None:
            self.kind = "block"
            self.np_, self.n, self.ranges, self.starts = tuple(np_), tuple(n), tuple(ranges), starts
            self.ghost_to_global = np.ascontiguousarray(
                ghost_to_global if ghost_to_global is not None else np.zeros(0), dtype=I64)
            self.ghost_to_owner = np.ascontiguousarray(
                ghost_to_owner if ghost_to_owner is not None else np.zeros(0), dtype=I32)
            self.n_own = int(np.prod([hi - lo + 1 for lo, hi in self.ranges]))
            self.n_ghost = len(self.ghost_to_global)
            self._own_to_global = None
        else:
            self.kind = "generic"
            self.local_to_global = np.ascontiguousarray(local_to_global, dtype=I64)
            self.local_to_owner = np.ascontiguousarray(local_to_owner, dtype=I32)
            # own = "inside the own box" when the constructor knows it (block_with_constant_size, src/p_range.jl:650-665:
            # a periodic direction with ONE part wraps onto ids this part owns, and the reference keeps those copies as
            # ghosts owned by self); otherwise own = owned by this part (LocalIndices, :1121)
            own = self.local_to_owner == self.part if is_own is None else np.ascontiguousarray(is_own, dtype=bool)
            self._own_to_local = (np.nonzero(own)[0] + 1).astype(I32)      # src/p_range.jl:1121
            self._ghost_to_local = (np.nonzero(~own)[0] + 1).astype(I32)   # :1122
            self.n_own, self.n_ghost = len(self._own_to_local), len(self._ghost_to_local)
            self.ghost_to_global = self.local_to_global[self._ghost_to_local - 1]
            self.ghost_to_owner = self.local_to_owner[self._ghost_to_local - 1]
            self._g2l = None

    # -- sizes / maps -------------------------------------------------------------------------
    @property
    def n_local(self):
        return self.n_own + self.n_ghost

    @property
    def own_to_local(self):
        if self.kind == "block":
            return np.arange(1, self.n_own + 1, dtype=I32)          # src/p_range.jl:1711-1714
        return self._own_to_local

    @property
    def ghost_to_local(self):
        if self.kind == "block":
            return np.arange(1, self.n_ghost + 1, dtype=I32) + I32(self.n_own)   # :1716-1720
        return self._ghost_to_local

    @property
    def own_is_contiguous_prefix(self):
        """True when local ids are [own | ghost]: the layout the SpMV kernels need."""
        return self.kind == "block" or bool(np.array_equal(self._own_to_local, np.arange(1, self.n_own + 1)))

    @property
    def local_to_device(self):
        """0-based position of every local id in the DEVICE layout, which is always [own | ghost] (what the kernels and
        the own-value reductions need); None when the local order already is that (block partitions)."""
        perm = getattr(self, "device_own_perm", None)
        if self.own_is_contiguous_prefix and perm is None:
            return None
        if getattr(self, "_l2d", None) is None:
            if self.own_is_contiguous_prefix:
                l2d = np.arange(self.n_local, dtype=np.int64)
            else:
                l2d = np.empty(self.n_local, dtype=np.int64)
                l2d[np.concatenate([self._own_to_local, self._ghost_to_local]).astype(np.int64) - 1] = np.arange(self.n_local)
            if perm is not None:
                # a library-side renumbering of the own values (renumber_for_locality, p_sparse_matrix.py): own value k of the
                # [own | ghost] layout lives at device position perm[k]; ghosts stay where they are
                own = l2d < self.n_own
                l2d[own] = np.asarray(perm, np.int64)[l2d[own]]
            self._l2d = l2d
        return self._l2d

    def with_device_own_perm(self, perm):
        """A copy of these indices (same ids, same local order, same assembly cache) whose own values are laid out in HBM in the
        order `perm` (device position of own value k); vectors and exchange plans made from the copy follow it."""
        import copy
        out = copy.copy(self)
        out.device_own_perm = np.ascontiguousarray(perm, np.int64)
        out._l2d = None
        return out

    @property
    def own_to_global(self):
        if self.kind != "block":
            return self.local_to_global[self._own_to_local - 1]
        if self._own_to_global is None:                                # src/p_range.jl:1471-1481
            gid = np.zeros((), dtype=I64)
            stride = 1
            D = len(self.n)
            for d, (lo, hi) in enumerate(self.ranges):
                shape = [1] * D
                shape[D - 1 - d] = hi - lo + 1                       # axis 0 slowest ... last fastest
                gid = gid + (np.arange(lo, hi + 1, dtype=I64).reshape(shape) - 1) * stride
                stride *= self.n[d]
            self._own_to_global = (gid + 1).ravel()
        return self._own_to_global

    def get_local_to_global(self):
        if self.kind == "block":
            return np.concatenate([self.own_to_global, self.ghost_to_global])
        return self.local_to_global

    def get_local_to_owner(self):
        if self.kind == "block":
            return np.concatenate([np.full(self.n_own, self.part, I32), self.ghost_to_owner])
        return self.local_to_owner

    def global_to_local(self, gids):
        """global_to_local(indices)[gids]; 0 when not local; ids < 1 pass through
        (map_x_to_y!, src/p_range.jl:300-309)."""
        gids = np.ascontiguousarray(gids, dtype=I64)
        out = np.zeros(len(gids), dtype=I32)
        if self.kind == "block":
            D = len(self.n)
            n = np.array(self.n, dtype=I64)
            lo = np.array([r[0] for r in self.ranges], dtype=I64)
            hi = np.array([r[1] for r in self.ranges], dtype=I64)
            L.call("pa_host_global_to_local_block", D, L.ptr(n), L.ptr(lo), L.ptr(hi),
                   L.ptr(self.ghost_to_global), self.n_ghost, L.ptr(gids), len(gids), L.ptr(out))
            return out
        if self._g2l is None:
            # an OWN id answers first, the ghost dictionary after it (BlockPartitionGlobalToLocal, src/p_range.jl:1551-1562): a part
            # alone in a periodic direction keeps copies of its own ids in its ghost layer, and what a neighbour asks for is the own one
            is_ghost = np.ones(len(self.local_to_global), dtype=np.int8)
            is_ghost[self._own_to_local - 1] = 0
            order = np.lexsort((np.arange(len(is_ghost)), is_ghost, self.local_to_global))
            self._g2l = (self.local_to_global[order], order)
        srt, order = self._g2l
        if len(srt):
            pos = np.clip(np.searchsorted(srt, gids), 0, len(srt) - 1)
            hit = srt[pos] == gids
            out = np.where(hit, order[pos] + 1, 0).astype(I32)
        out = np.where(gids < 1, gids, out).astype(I32)
        return out

    def __repr__(self):
        return f"LocalIndices(part={self.part}, n_own={self.n_own}, n_ghost={self.n_ghost}, kind={self.kind})"


def _empty_assembly_cache():
    """src/p_range.jl:385-392."""
    e = JaggedArray(np.zeros(0, I32), np.array([1], I32))
    return dict(neighbors_snd=np.zeros(0, I32), neighbors_rcv=np.zeros(0, I32),
                local_indices_snd=e, local_indices_rcv=e)


def _block_starts(np_, n):
    return tuple(np.array([local_range(p, npd, nd)[0] for p in range(1, npd + 1)] + [nd + 1], dtype=I64)
                 for npd, nd in zip(np_, n))


def uniform_partition(ranks, np_, n=None, ghost=None, periodic=None):
    """uniform_partition(ranks,np,n[,ghost[,periodic]]) (src/p_range.jl:585-671); uniform_partition(ranks,n) is
    uniform_partition(ranks,length(ranks),n) (:601-603)."""
    if n is None:
        np_, n = len(ranks), np_
    if isinstance(np_, (int, np.integer)):
        np_, n = (int(np_),), (int(n),)
        ghost = None if ghost is None else (int(ghost),)
        periodic = None if periodic is None else (bool(periodic),)
    np_, n = tuple(int(x) for x in np_), tuple(int(x) for x in n)
    assert int(np.prod(np_)) == len(ranks), "prod(np) == length(rank)"       # :586
    starts = _block_starts(np_, n)
    n_global = int(np.prod(n))

    def block(rank):
        p = _cartesian(rank, np_)
        own_ranges = tuple(local_range(pd, npd, nd) for pd, npd, nd in zip(p, np_, n))
        if ghost is None:
            ind = LocalIndices(n_global, rank, np_=np_, n=n, ranges=own_ranges, starts=starts)
            ind.cache = _empty_assembly_cache()                                  # :590-594
            return ind
        per = periodic if periodic is not None else tuple(False for _ in ghost)
        local_ranges = tuple(local_range(pd, npd, nd, g, pr) for pd, npd, nd, g, pr in zip(p, np_, n, ghost, per))
        owners = []
        for pd, npd, nd, (lo, hi) in zip(p, np_, n, local_ranges):              # :626-637
            lrv = list(range(lo, hi + 1))
            my, i = [0] * len(lrv), 0
            for q in itertools.cycle(range(1, npd + 1)):
                plo, phi = local_range(q, npd, nd)
                while i < len(my) and plo <= ((lrv[i] - 1) % nd) + 1 <= phi:
                    my[i] = q
                    i += 1
                if i >= len(my):
                    break
            owners.append(my)
        # column-major over the local box (:648): per dimension the global coordinate (CircularArray wrap) and the
        # owner coordinate of every local coordinate, combined by broadcasting (dimension 0 fastest)
        D = len(n)
        l2g = np.zeros((), I64)
        l2o = np.zeros((), I64)
        is_own = np.ones((), bool)
        gstride, ostride = 1, 1
        for d in range(D):
            lo, hi = local_ranges[d]
            shape = [1] * D
            shape[D - 1 - d] = hi - lo + 1                                     # axis 0 slowest ... last fastest
            c = np.arange(lo, hi + 1, dtype=I64)
            l2g = l2g + (((c - 1) % n[d]).reshape(shape)) * gstride
            l2o = l2o + ((np.asarray(owners[d], I64) - 1).reshape(shape)) * ostride
            is_own = is_own & ((c >= own_ranges[d][0]) & (c <= own_ranges[d][1])).reshape(shape)
            gstride *= n[d]
            ostride *= np_[d]
        l2o = np.where(is_own, rank - 1, l2o)
        return LocalIndices(n_global, rank, local_to_global=(l2g + 1).ravel().astype(I64),
                            local_to_owner=(l2o + 1).ravel().astype(I32), is_own=np.broadcast_to(is_own, l2g.shape).ravel())

    indices = pmap(block, ranks)
    if ghost is not None:
        assembly_neighbors(indices, symmetric=True)                              # :596
    return indices


def variable_partition(n_own, n_global, start=None):
    """variable_partition(n_own,n_global;start) (src/p_range.jl:705-733), 1-D, no ghost."""
    from .primitives import scan
    ranks = linear_indices(n_own)
    if start is None:
        start = scan(lambda a, b: a + b, n_own, type="exclusive", init=1)
    allstart = getany(gather(start, destination="all"))
    starts = (np.array(list(allstart) + [n_global + 1], dtype=I64),)
    P = len(n_own)

    def f(rank, k, s):
        ind = LocalIndices(n_global, rank, np_=(P,), n=(n_global,), ranges=((int(s), int(s) + int(k) - 1),), starts=starts)
        ind.cache = _empty_assembly_cache()
        return ind

    return pmap(f, ranks, n_own, start)


def find_owner(indices, global_ids):
    """find_owner(index_partition,global_ids) (src/p_range.jl:346-348,1609-1619): block partitions only."""

    def f(ind, gids):
        assert ind.kind == "block", "find_owner needs a block partition"
        gids = np.ascontiguousarray(gids, dtype=I64)
        owners = np.zeros(len(gids), dtype=I32)
        D = len(ind.n)
        n = np.array(ind.n, dtype=I64)
        npd = np.array(ind.np_, dtype=I32)
        arr = (C.c_void_p * D)(*[s.ctypes.data for s in ind.starts])
        L.call("pa_host_find_owner_block", D, L.ptr(n), L.ptr(npd), arr, L.ptr(gids), len(gids), L.ptr(owners))
        return owners

    return pmap(f, indices, global_ids)


def filter_ghost(ind, gids, owners):
    """src/p_range.jl:205-241: new ghosts in first-seen order."""
    gids = np.ascontiguousarray(gids, dtype=I64)
    owners = np.ascontiguousarray(owners, dtype=I32)
    assert len(gids) == len(owners)
    cnt = C.c_int64(0)
    known = ind.ghost_to_global
    L.call("pa_host_filter_ghost", ind.part, L.ptr(gids), L.ptr(owners), len(gids), L.ptr(known), len(known),
           None, None, C.byref(cnt))
    og, oo = np.zeros(cnt.value, I64), np.zeros(cnt.value, I32)
    L.call("pa_host_filter_ghost", ind.part, L.ptr(gids), L.ptr(owners), len(gids), L.ptr(known), len(known),
           L.ptr(og), L.ptr(oo), C.byref(cnt))
    return og, oo


def union_ghost(ind, gids, owners):
    """union_ghost(indices,gids,owners) (src/p_range.jl:252-259)."""
    if ind.kind != "block":
        raise ValueError("replace_ghost only makes sense for un-permuted local indices (src/p_range.jl:1146)")
    eg, eo = filter_ghost(ind, gids, owners)
    return LocalIndices(ind.n_global, ind.part, np_=ind.np_, n=ind.n, ranges=ind.ranges, starts=ind.starts,
                        ghost_to_global=np.concatenate([ind.ghost_to_global, eg]),
                        ghost_to_owner=np.concatenate([ind.ghost_to_owner, eo]))


def assembly_neighbors(indices, symmetric=False):
    """assembly_neighbors(index_partition) (src/p_range.jl:417-450), cached."""
    have = pmap(lambda i: "neighbors_snd" in i.cache, indices)
    if getany(have):
        return (pmap(lambda i: i.cache["neighbors_snd"], indices), pmap(lambda i: i.cache["neighbors_rcv"], indices))
    parts_snd = pmap(lambda i: np.unique(i.ghost_to_owner[i.ghost_to_owner != i.part]).astype(I32), indices)
    graph = exchange_graph(parts_snd, symmetric=symmetric)

    def store(i, s, r):
        i.cache["neighbors_snd"], i.cache["neighbors_rcv"] = np.asarray(s, I32), np.asarray(r, I32)

    pmap(store, indices, graph.snd, graph.rcv)
    return graph.snd, graph.rcv


def assembly_local_indices(indices, neighbors_snd=None, neighbors_rcv=None):
    """assembly_local_indices (src/p_range.jl:466-531), cached."""
    if neighbors_snd is None:
        neighbors_snd, neighbors_rcv = assembly_neighbors(indices)
    have = pmap(lambda i: "local_indices_snd" in i.cache, indices)
    if not getany(have):
        def snd_side(ind, parts_snd):
            # ghosts grouped by owner, in ascending local id inside a group (:506-513)
            # (a ghost owned by this very part -- the wrapped copies of a periodic direction with one part -- is not sent:
            # `if owner != rank`, :494,503)
            keep = ind.ghost_to_owner != ind.part
            gl, go = ind.ghost_to_local[keep], ind.ghost_to_owner[keep]
            slot = np.searchsorted(parts_snd, go)
            order = np.argsort(slot, kind="stable")
            ptrs = np.zeros(len(parts_snd) + 1, dtype=I32)
            np.add.at(ptrs, slot + 1, 1)
            length_to_ptrs(ptrs)
            return (JaggedArray(gl[order].astype(I32), ptrs), JaggedArray(ind.ghost_to_global[keep][order], ptrs.copy()))

        lids_snd, gids_snd = tuple_of_arrays(pmap(snd_side, indices, neighbors_snd))
        graph = exchange_graph(neighbors_snd, rcv=neighbors_rcv)
        gids_rcv = exchange(pmap(lambda g: [g[i] for i in range(len(g))], gids_snd), graph)

        def rcv_side(g, ind):
            ja = JaggedArray.from_lists(g, I64)
            return JaggedArray(ind.global_to_local(ja.data).astype(I32), ja.ptrs)

        lids_rcv = pmap(rcv_side, gids_rcv, indices)

        def store(i, s, r):
            i.cache["local_indices_snd"], i.cache["local_indices_rcv"] = s, r

        pmap(store, indices, lids_snd, lids_rcv)
    return (pmap(lambda i: i.cache["local_indices_snd"], indices), pmap(lambda i: i.cache["local_indices_rcv"], indices))


class PRange:
    """PRange(partition) (src/p_range.jl:1776-1787)."""

    def __init__(self, partition):
        self.partition = partition

    def __len__(self):
        return getany(pmap(lambda i: i.n_global, self.partition))


def partition(a):
    return a.partition
