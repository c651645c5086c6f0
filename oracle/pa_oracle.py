"""
pa_oracle -- CPU restatement of the PartitionedArrays.jl hot path (TEST INFRASTRUCTURE ONLY).

This file is the *oracle*: a plain numpy / pure-Python restatement of the reference's
algorithm for `mul!(c::PVector, A::PSparseMatrix, b::PVector)` and the ghost exchange
(`consistent!`, `assemble!`, `exchange!`) it depends on, plus the host-side set-up that
defines the data those loops run on (partitions, ghost numbering, neighbour lists, COO->CSR).

Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it.
The product (`partitionedarrays.jl_amd/`) never does.

Parity pinning: there is no Julia in this image, so the reference itself cannot be run.
The oracle is pinned against every literal golden value the reference's own tests and
docstrings hold for this path (tests/golden/*.json, checked by tests/test_oracle_golden.py).
What is NOT pinned by a literal golden is the *value* of `mul!` on a CSR-split
PSparseMatrix for general x (the reference has only type-level smoke tests there);
it is pinned by the diagonal-matrix known answers, by `A*1 == b` for the HPCG matrix
(HPCG/src/sparse_matrix.jl:60-75) and by distributed == centralised products.
The third-party `SparseMatricesCSR.mul!(y,A,x,a,b)` (v0.6, not in /root/reference) is
restated from its published source: `y[row] += nzval*x[col]*alpha` row by row.

Conventions: EVERYTHING is 1-based exactly as in the reference's data (part ids, global ids,
local ids, `ptrs`), stored in 0-based numpy arrays.  "parts" are python lists (DebugArray
semantics: `map` is a sequential loop, src/debug_array.jl:110-117).

Every function cites the reference file:line it follows (paths relative to /root/reference).
"""
from __future__ import annotations

import ctypes
import itertools
import os
from dataclasses import dataclass, field

import numpy as np

I64 = np.int64
I32 = np.int32
F64 = np.float64


# --------------------------------------------------------------------------------------
# src/jagged_array.jl
# --------------------------------------------------------------------------------------
def length_to_ptrs(ptrs):
    """src/jagged_array.jl:11-18 -- in: ptrs[i+1] = len(i); out: 1-based start offsets."""
    ptrs[0] = 1
    for i in range(len(ptrs) - 1):
        ptrs[i + 1] += ptrs[i]
    return ptrs


def rewind_ptrs(ptrs):
    """src/jagged_array.jl:26-32."""
    for i in range(len(ptrs) - 2, -1, -1):
        ptrs[i + 1] = ptrs[i]
    ptrs[0] = 1
    return ptrs


@dataclass
class Jagged:
    """src/jagged_array.jl:107-122: data + 1-based ptrs; slice i = data[ptrs[i]-1 : ptrs[i+1]-1]."""
    data: np.ndarray
    ptrs: np.ndarray

    @staticmethod
    def from_lists(vv, dtype=None, ptr_dtype=I32):
        """src/jagged_array.jl:130-152 (JaggedArray(a) for a vector of vectors)."""
        ptrs = np.zeros(len(vv) + 1, dtype=ptr_dtype)
        for i, v in enumerate(vv):
            ptrs[i + 1] = len(v)
        length_to_ptrs(ptrs)
        flat = [x for v in vv for x in v]
        if dtype is None:
            data = np.array(flat) if flat else np.zeros(0, dtype=I64)
        else:
            data = np.array(flat, dtype=dtype)
        return Jagged(data, ptrs)

    def __len__(self):
        return len(self.ptrs) - 1

    def __getitem__(self, i):  # 0-based slot i
        return self.data[self.ptrs[i] - 1: self.ptrs[i + 1] - 1]

    def tolists(self):
        return [self[i].tolist() for i in range(len(self))]


# --------------------------------------------------------------------------------------
# src/p_range.jl -- block sizes, index sets
# --------------------------------------------------------------------------------------
def local_range(p, np_, n, ghost=False, periodic=False):
    """src/p_range.jl:806-818. Returns inclusive 1-based (start, stop)."""
    l, rem = divmod(n, np_)
    offset = l * (p - 1)
    if rem >= (np_ - p + 1):
        l += 1
        offset += p - (np_ - rem) - 1
    g = int(ghost)                       # a number of ghost layers: `start = 1+offset-ghost` (:813)
    start = 1 + offset - g
    stop = l + offset + g
    if periodic:
        return start, stop
    return max(1, start), min(n, stop)


def _cartesian(rank, np_):
    """CartesianIndices(np)[rank], column-major, 1-based (src/p_range.jl:617)."""
    r = rank - 1
    out = []
    for d in np_:
        out.append(r % d + 1)
        r //= d
    return tuple(out)


def _linear(ci, n):
    """LinearIndices(n)[ci], column-major, 1-based."""
    lin, stride = 0, 1
    for c, d in zip(ci, n):
        lin += (c - 1) * stride
        stride *= d
    return lin + 1


@dataclass
class Indices:
    """One part of an index partition (the AbstractLocalIndices interface, src/p_range.jl:32-160).

    Covers LocalIndicesWithConstantBlockSize / VariableBlockSize (own ids first, ghosts after:
    src/p_range.jl:1711-1720), PermutedLocalIndices (:1372) and LocalIndices (:1100).
    """
    n_global: int
    part: int                         # 1-based owner id
    local_to_global: np.ndarray       # int64, 1-based gids
    local_to_owner: np.ndarray        # int32, 1-based part ids
    kind: str = "generic"             # "block" when a block partition (find_owner by arithmetic)
    np_: tuple = ()
    n: tuple = ()
    ranges: tuple = ()                # own box ((lo,hi),...) inclusive, block partitions only
    starts: tuple = ()                # per-dim block starts (+ n+1), for find_owner
    cache: dict = field(default_factory=dict)   # AssemblyCache, src/p_range.jl:354-359
    is_own: object = None             # own flags when the constructor decides them by the own box (src/p_range.jl:650-665)

    def __post_init__(self):
        self.local_to_global = np.asarray(self.local_to_global, dtype=I64)
        self.local_to_owner = np.asarray(self.local_to_owner, dtype=I32)
        own = self.local_to_owner == self.part if self.is_own is None else np.asarray(self.is_own, dtype=bool)
        # src/p_range.jl:1121-1128 (findall; perm)
        self.own_to_local = (np.nonzero(own)[0] + 1).astype(I32)
        self.ghost_to_local = (np.nonzero(~own)[0] + 1).astype(I32)
        n_own = len(self.own_to_local)
        self.perm = np.zeros(len(own), dtype=I32)
        self.perm[self.own_to_local - 1] = np.arange(1, n_own + 1)
        self.perm[self.ghost_to_local - 1] = np.arange(1, len(self.ghost_to_local) + 1) + n_own
        self._g2l = None

    # lengths
    @property
    def n_own(self):
        return len(self.own_to_local)

    @property
    def n_ghost(self):
        return len(self.ghost_to_local)

    @property
    def n_local(self):
        return len(self.local_to_global)

    @property
    def own_to_global(self):
        return self.local_to_global[self.own_to_local - 1]

    @property
    def ghost_to_global(self):
        return self.local_to_global[self.ghost_to_local - 1]

    @property
    def ghost_to_owner(self):
        return self.local_to_owner[self.ghost_to_local - 1]

    def global_to_local(self, gids):
        """global_to_local(indices)[gid]; 0 when gid is not local (VectorFromDict default)."""
        if self._g2l is None:
            # BlockPartitionGlobalToLocal (src/p_range.jl:1551-1562): an OWN id answers first, the ghost dictionary after it -- a part
            # alone in a periodic direction holds copies of its own ids in its ghost layer (round 6: the lowest local id used to win,
            # which is the layer's copy whenever the layer comes first in the traversal; found by tests/test_host_bruteforce.py)
            is_ghost = np.ones(len(self.local_to_global), dtype=np.int8)
            is_ghost[self.own_to_local - 1] = 0
            order = np.lexsort((np.arange(len(is_ghost)), is_ghost, self.local_to_global))
            self._g2l = (self.local_to_global[order], order)
        srt, order = self._g2l
        g = np.asarray(gids, dtype=I64).ravel()
        if len(srt) == 0:
            return np.zeros(len(g), dtype=I32)
        pos = np.clip(np.searchsorted(srt, g), 0, len(srt) - 1)
        hit = srt[pos] == g
        return np.where(hit, order[pos] + 1, 0).astype(I32)


def _block_starts(np_, n):
    """src/p_range.jl:1611-1615: per-dim first(local_range(p)) for p=1..np, then n+1."""
    return tuple(
        np.array([local_range(p, npd, nd)[0] for p in range(1, npd + 1)] + [nd + 1], dtype=I64)
        for npd, nd in zip(np_, n)
    )


def _box_gids(ranges, n):
    """Column-major enumeration of a box inside the global grid (src/p_range.jl:1471-1481)."""
    axes = [np.arange(lo, hi + 1, dtype=I64) for lo, hi in ranges]
    gid = np.zeros((), dtype=I64)
    stride = 1
    for d, ax in enumerate(axes):
        shape = [1] * len(axes)
        shape[d] = len(ax)
        gid = gid + (ax.reshape(shape) - 1) * stride
        stride *= n[d]
    # column-major: first axis fastest -> transpose then ravel in C order
    return (gid + 1).transpose(tuple(reversed(range(len(axes))))).ravel()


def uniform_partition(np_, n, ghost=None, periodic=None):
    """src/p_range.jl:585-671.  np_, n tuples (or ints for 1-D).  Returns a list of Indices."""
    if isinstance(np_, int):
        np_, n = (np_,), (n,)
        if ghost is not None and not isinstance(ghost, tuple):
            ghost = (ghost,)
        if periodic is not None and not isinstance(periodic, tuple):
            periodic = (periodic,)
    np_, n = tuple(np_), tuple(n)
    P = int(np.prod(np_))
    starts = _block_starts(np_, n)
    parts = []
    for rank in range(1, P + 1):
        p = _cartesian(rank, np_)
        own_ranges = tuple(local_range(pd, npd, nd) for pd, npd, nd in zip(p, np_, n))
        if ghost is None:
            # block_with_constant_size(rank,np,n): src/p_range.jl:615-620
            l2g = _box_gids(own_ranges, n)
            l2o = np.full(len(l2g), rank, dtype=I32)
            ind = Indices(int(np.prod(n)), rank, l2g, l2o, "block", np_, n, own_ranges, starts)
            # src/p_range.jl:590-594: an EMPTY assembly cache is installed
            ind.cache = dict(neighbors_snd=np.zeros(0, I32), neighbors_rcv=np.zeros(0, I32),
                             local_indices_snd=Jagged(np.zeros(0, I32), np.array([1], I32)),
                             local_indices_rcv=Jagged(np.zeros(0, I32), np.array([1], I32)))
            parts.append(ind)
            continue
        per = periodic if periodic is not None else tuple(False for _ in ghost)
        # block_with_constant_size(rank,np,n,ghost,periodic): src/p_range.jl:622-671
        local_ranges = tuple(local_range(pd, npd, nd, g, pr)
                             for pd, npd, nd, g, pr in zip(p, np_, n, ghost, per))
        owners = []
        for pd, npd, nd, lr in zip(p, np_, n, local_ranges):
            lo, hi = lr
            lrv = list(range(lo, hi + 1))
            my = [0] * len(lrv)
            i = 0
            done = False
            for q in itertools.cycle(range(1, npd + 1)):     # :630
                plo, phi = local_range(q, npd, nd)
                while plo <= ((lrv[i] - 1) % nd) + 1 <= phi:
                    my[i] = q
                    i += 1
                    if i >= len(my):
                        done = True
                        break
                if done:
                    break
            owners.append(my)
        lens = [hi - lo + 1 for lo, hi in local_ranges]
        l2g, l2o, own_flags = [], [], []
        for ci0 in itertools.product(*[range(L) for L in reversed(lens)]):
            ci = tuple(reversed(ci0))                         # column-major enumeration (:648)
            is_own = all(own_ranges[d][0] <= local_ranges[d][0] + ci[d] <= own_ranges[d][1]
                         for d in range(len(n)))
            gci = tuple(((local_ranges[d][0] + ci[d] - 1) % n[d]) + 1 for d in range(len(n)))  # CircularArray
            l2g.append(_linear(gci, n))
            own_flags.append(is_own)     # a wrapped copy of an own id (periodic, one part in that direction) stays a ghost
            if is_own:
                l2o.append(rank)
            else:
                o = tuple(owners[d][ci[d]] for d in range(len(n)))
                l2o.append(_linear(o, np_))
        ind = Indices(int(np.prod(n)), rank, np.array(l2g, I64), np.array(l2o, I32),
                      "block-permuted", np_, n, own_ranges, starts, is_own=np.array(own_flags, bool))
        parts.append(ind)
    if ghost is not None:
        assembly_neighbors(parts, symmetric=True)             # src/p_range.jl:596
    return parts


def variable_partition(n_own, n_global):
    """src/p_range.jl:705-733 (1-D, no ghost). start = exclusive scan with init 1."""
    P = len(n_own)
    start = [1]
    for k in n_own[:-1]:
        start.append(start[-1] + k)
    starts = (np.array(start + [n_global + 1], dtype=I64),)
    parts = []
    for rank in range(1, P + 1):
        lo = start[rank - 1]
        hi = lo + n_own[rank - 1] - 1
        l2g = np.arange(lo, hi + 1, dtype=I64)
        ind = Indices(n_global, rank, l2g, np.full(len(l2g), rank, I32), "block",
                      (P,), (n_global,), ((lo, hi),), starts)
        ind.cache = dict(neighbors_snd=np.zeros(0, I32), neighbors_rcv=np.zeros(0, I32),
                         local_indices_snd=Jagged(np.zeros(0, I32), np.array([1], I32)),
                         local_indices_rcv=Jagged(np.zeros(0, I32), np.array([1], I32)))
        parts.append(ind)
    return parts


def local_indices(n_global, owner, local_to_global, local_to_owner):
    """LocalIndices(n_global,owner,local_to_global,local_to_owner), src/p_range.jl:1119-1143."""
    return Indices(n_global, owner, local_to_global, local_to_owner)


def find_owner(indices, global_ids):
    """src/p_range.jl:346-348,1609-1619,1502-1513: owner = LinearIndices(np)[searchsortedlast per dim]."""
    out = []
    for ind, gids in zip(indices, global_ids):
        assert ind.kind.startswith("block")
        gids = np.asarray(gids, dtype=I64)
        r = gids - 1
        owner = np.zeros(len(gids), dtype=I64)
        stride = 1
        for d, nd in enumerate(ind.n):
            c = r % nd + 1
            r = r // nd
            j = np.searchsorted(ind.starts[d], c, side="right")  # searchsortedlast (1-based result)
            owner += (j - 1) * stride
            stride *= ind.np_[d]
        out.append((owner + 1).astype(I32))
    return out


def filter_ghost(ind, gids, owners):
    """src/p_range.jl:205-241: unseen non-own gids in FIRST-SEEN order; gid<1 skipped."""
    gids = np.asarray(gids, dtype=I64)
    owners = np.asarray(owners, dtype=I32)
    existing = set(int(g) for g in ind.ghost_to_global)
    mask = (gids >= 1) & (owners != ind.part)
    cand_g = gids[mask]
    cand_o = owners[mask]
    if existing:
        keep = np.array([int(g) not in existing for g in cand_g], dtype=bool)
        cand_g, cand_o = cand_g[keep], cand_o[keep]
    _, first = np.unique(cand_g, return_index=True)
    first.sort()
    return cand_g[first], cand_o[first]


def union_ghost(ind, gids, owners):
    """src/p_range.jl:252-259 (+replace_ghost :1603): same own ids, ghosts = old ++ new."""
    assert ind.kind == "block", "replace_ghost only makes sense for un-permuted local indices"
    extra_g, extra_o = filter_ghost(ind, gids, owners)
    own_g = ind.own_to_global
    l2g = np.concatenate([own_g, ind.ghost_to_global, extra_g])
    l2o = np.concatenate([np.full(len(own_g), ind.part, I32), ind.ghost_to_owner, extra_o])
    return Indices(ind.n_global, ind.part, l2g, l2o, "block", ind.np_, ind.n, ind.ranges, ind.starts)


# --------------------------------------------------------------------------------------
# src/primitives.jl -- ExchangeGraph, exchange
# --------------------------------------------------------------------------------------
def find_rcv_ids_gather_scatter(snd_ids):
    """src/primitives.jl:826-859: rcv[j] = ascending list of i with j in snd[i] (CSC order)."""
    P = len(snd_ids)
    rcv = [[] for _ in range(P)]
    for i in range(1, P + 1):
        for j in sorted(set(int(x) for x in snd_ids[i - 1])):
            rcv[j - 1].append(i)
    # adjmat = sparse(I,J,...) iterated column by column, rows ascending inside a column
    return [np.array(sorted(r), dtype=I32) for r in rcv]


def exchange_graph(snd, rcv=None, symmetric=False):
    """ExchangeGraph(snd;rcv,symmetric,...) src/primitives.jl:768-783."""
    if rcv is not None:
        return snd, rcv
    if symmetric:
        return snd, snd
    return snd, find_rcv_ids_gather_scatter(snd)


def is_consistent(snd, rcv):
    """src/primitives.jl:861-874."""
    for part in range(1, len(rcv) + 1):
        for i in rcv[part - 1]:
            if sum(1 for k in snd[i - 1] if k == part) != 1:
                return False
        for i in snd[part - 1]:
            if sum(1 for k in rcv[i - 1] if k == part) != 1:
                return False
    return True


def exchange_scalar(snd_data, snd_ids, rcv_ids):
    """src/primitives.jl:1005-1018: rcv[r][i] = snd[s][j], s = rcv_ids[r][i], snd_ids[s][j] == r."""
    assert is_consistent(snd_ids, rcv_ids)
    out = []
    for r in range(1, len(rcv_ids) + 1):
        row = []
        for s in rcv_ids[r - 1]:
            j = [k for k, v in enumerate(snd_ids[s - 1]) if v == r][0]
            row.append(snd_data[s - 1][j])
        out.append(row)
    return out


def exchange_jagged(rcv, snd, snd_ids, rcv_ids):
    """src/primitives.jl:1020-1042 (in-place on rcv[r].data). rcv/snd: lists of Jagged."""
    assert is_consistent(snd_ids, rcv_ids)
    for r in range(1, len(rcv_ids) + 1):
        for i, s in enumerate(rcv_ids[r - 1]):
            j = [k for k, v in enumerate(snd_ids[s - 1]) if v == r][0]
            pr, ps = rcv[r - 1].ptrs, snd[s - 1].ptrs
            assert pr[i + 1] - pr[i] == ps[j + 1] - ps[j]
            rcv[r - 1].data[pr[i] - 1: pr[i + 1] - 1] = snd[s - 1].data[ps[j] - 1: ps[j + 1] - 1]
    return rcv


def allocate_exchange_jagged(snd, snd_ids, rcv_ids, dtype=None):
    """src/primitives.jl:953-983: exchange the slice lengths first, then allocate."""
    n_snd = [[int(s.ptrs[j + 1] - s.ptrs[j]) for j in range(len(s))] for s in snd]
    n_rcv = exchange_scalar(n_snd, snd_ids, rcv_ids)
    out = []
    for r, counts in enumerate(n_rcv):
        ptrs = np.zeros(len(counts) + 1, dtype=I32)
        ptrs[1:] = counts
        length_to_ptrs(ptrs)
        out.append(Jagged(np.zeros(ptrs[-1] - 1, dtype=dtype or snd[r].data.dtype), ptrs))
    return out


# --------------------------------------------------------------------------------------
# src/p_range.jl -- assembly neighbours / local indices
# --------------------------------------------------------------------------------------
def assembly_neighbors(indices, symmetric=False):
    """src/p_range.jl:417-450 (cached; snd = sorted unique non-own owners)."""
    if all("neighbors_snd" in ind.cache for ind in indices):
        return [i.cache["neighbors_snd"] for i in indices], [i.cache["neighbors_rcv"] for i in indices]
    parts_snd = []
    for ind in indices:
        s = sorted(set(int(o) for o in ind.local_to_owner if o != ind.part))
        parts_snd.append(np.array(s, dtype=I32))
    snd, rcv = exchange_graph(parts_snd, symmetric=symmetric)
    for ind, s, r in zip(indices, snd, rcv):
        ind.cache["neighbors_snd"] = s
        ind.cache["neighbors_rcv"] = r
    return snd, rcv


def assembly_local_indices(indices, neighbors_snd=None, neighbors_rcv=None):
    """src/p_range.jl:466-531."""
    if neighbors_snd is None:
        neighbors_snd, neighbors_rcv = assembly_neighbors(indices)
    if all("local_indices_snd" in ind.cache for ind in indices):
        return ([i.cache["local_indices_snd"] for i in indices],
                [i.cache["local_indices_rcv"] for i in indices])
    lids_snd, gids_snd = [], []
    for ind, parts_snd in zip(indices, neighbors_snd):
        owner_to_i = {int(o): i for i, o in enumerate(parts_snd)}
        ptrs = np.zeros(len(parts_snd) + 1, dtype=I32)
        for o in ind.local_to_owner:
            if o != ind.part:
                ptrs[owner_to_i[int(o)] + 1] += 1
        length_to_ptrs(ptrs)
        data_l = np.zeros(ptrs[-1] - 1, dtype=I32)
        data_g = np.zeros(ptrs[-1] - 1, dtype=I64)
        for lid0, o in enumerate(ind.local_to_owner):
            if o != ind.part:
                k = owner_to_i[int(o)]
                p = ptrs[k]
                data_l[p - 1] = lid0 + 1
                data_g[p - 1] = ind.local_to_global[lid0]
                ptrs[k] += 1
        rewind_ptrs(ptrs)
        lids_snd.append(Jagged(data_l, ptrs))
        gids_snd.append(Jagged(data_g, ptrs.copy()))
    gids_rcv = allocate_exchange_jagged(gids_snd, neighbors_snd, neighbors_rcv, dtype=I64)
    exchange_jagged(gids_rcv, gids_snd, neighbors_snd, neighbors_rcv)
    lids_rcv = []
    for g, ind in zip(gids_rcv, indices):
        lids_rcv.append(Jagged(ind.global_to_local(g.data).astype(I32), g.ptrs))
    for ind, s, r in zip(indices, lids_snd, lids_rcv):
        ind.cache["local_indices_snd"] = s
        ind.cache["local_indices_rcv"] = r
    return lids_snd, lids_rcv


# --------------------------------------------------------------------------------------
# src/p_vector.jl -- cache, assemble!, consistent!
# --------------------------------------------------------------------------------------
@dataclass
class VectorAssemblyCache:
    """src/p_vector.jl:418-426."""
    neighbors_snd: list
    neighbors_rcv: list
    local_indices_snd: list
    local_indices_rcv: list
    buffer_snd: list
    buffer_rcv: list

    def reverse(self):
        """src/p_vector.jl:427-437."""
        return VectorAssemblyCache(self.neighbors_rcv, self.neighbors_snd,
                                   self.local_indices_rcv, self.local_indices_snd,
                                   self.buffer_rcv, self.buffer_snd)


def p_vector_cache(values, indices, dtype=F64):
    """src/p_vector.jl:451-468."""
    ns, nr = assembly_neighbors(indices)
    ls, lr = assembly_local_indices(indices, ns, nr)
    bs = [Jagged(np.zeros(l.ptrs[-1] - 1, dtype=dtype), l.ptrs) for l in ls]
    br = [Jagged(np.zeros(l.ptrs[-1] - 1, dtype=dtype), l.ptrs) for l in lr]
    return VectorAssemblyCache(ns, nr, ls, lr, bs, br)


def assemble_impl(f, values, cache):
    """src/p_vector.jl:587-612: pack, exchange!, unpack with f(old, rcv). f in {'insert', '+'}."""
    for v, lids, buf in zip(values, cache.local_indices_snd, cache.buffer_snd):
        for p, lid in enumerate(lids.data):                  # :595-599
            buf.data[p] = v[lid - 1]
    exchange_jagged(cache.buffer_rcv, cache.buffer_snd, cache.neighbors_snd, cache.neighbors_rcv)
    for v, lids, buf in zip(values, cache.local_indices_rcv, cache.buffer_rcv):
        for p, lid in enumerate(lids.data):                  # :605-609
            if f == "insert":
                v[lid - 1] = buf.data[p]
            else:
                v[lid - 1] = v[lid - 1] + buf.data[p]
    return values


def consistent(values, indices, cache=None):
    """consistent!(a), src/p_vector.jl:747-755: reversed cache + insert."""
    cache = cache or p_vector_cache(values, indices, values[0].dtype)
    return assemble_impl("insert", values, cache.reverse())


def assemble(values, indices, cache=None):
    """assemble!(+,a), src/p_vector.jl:695-708: add into owners, then zero the ghosts."""
    cache = cache or p_vector_cache(values, indices, values[0].dtype)
    assemble_impl("+", values, cache)
    for v, ind in zip(values, indices):
        v[ind.ghost_to_local - 1] = 0
    return values


def pvector_collect(values, indices):
    """collect(v): own values scattered to global positions."""
    out = np.zeros(indices[0].n_global, dtype=values[0].dtype)
    for v, ind in zip(values, indices):
        out[ind.own_to_global - 1] = v[ind.own_to_local - 1]
    return out


def dot(a, b, indices):
    """src/p_vector.jl:1189-1192: own-values dots, then sum over parts (left to right)."""
    s = 0.0
    for x, y, ind in zip(a, b, indices):
        s = s + float(np.dot(x[ind.own_to_local - 1], y[ind.own_to_local - 1]))
    return s


def norm2(a, indices):
    """src/p_vector.jl:1201-1206 with p=2."""
    s = 0.0
    for x, ind in zip(a, indices):
        s = s + float(np.linalg.norm(x[ind.own_to_local - 1])) ** 2
    return s ** 0.5


# --------------------------------------------------------------------------------------
# src/sparse_utils.jl -- COO -> CSR, local kernels
# --------------------------------------------------------------------------------------
@dataclass
class CSR:
    """SparseMatrixCSR{1,Float64,Int32}: 1-based rowptr/colval, columns sorted per row."""
    m: int
    n: int
    rowptr: np.ndarray
    colval: np.ndarray
    nzval: np.ndarray

    @property
    def nnz(self):
        return len(self.nzval)

    def to_dense(self):
        A = np.zeros((self.m, self.n))
        for r in range(self.m):
            for p in range(self.rowptr[r] - 1, self.rowptr[r + 1] - 1):
                A[r, self.colval[p] - 1] += self.nzval[p]
        return A


def compresscoo_csr(I, J, V, m, n, skip=True, index_dtype=I32):
    """src/sparse_utils.jl:313-350 -> sparsecsr(Val(1),I,J,V,m,n,+) (SparseMatricesCSR 0.6).

    Per-row sorted columns; duplicates combined with + in input order; with skip=true the
    entries with i<1||j<1 are NOT dropped for CSR but rewritten to (1,1,0.0)
    (FilteredCooVector, :330-342,370-390); m*n==0 -> empty (:334-337).
    """
    I = np.asarray(I, dtype=I64).copy()
    J = np.asarray(J, dtype=I64).copy()
    V = np.asarray(V, dtype=F64).copy()
    if skip and m * n == 0:
        I, J, V = I[:0], J[:0], V[:0]
    elif skip:
        bad = (I < 1) | (J < 1)
        I[bad] = 1
        J[bad] = 1
        V[bad] = 0.0
    order = np.lexsort((J, I))                 # stable: duplicates keep input order
    Is, Js, Vs = I[order], J[order], V[order]
    if len(Is):
        new = np.ones(len(Is), dtype=bool)
        new[1:] = (Is[1:] != Is[:-1]) | (Js[1:] != Js[:-1])
    else:
        new = np.zeros(0, dtype=bool)
    slot = np.cumsum(new) - 1
    nz = int(new.sum())
    nzval = np.zeros(nz, dtype=F64)
    nzval[slot[new]] = Vs[new]
    if (~new).any():
        np.add.at(nzval, slot[~new], Vs[~new])   # sequential, input order
    colval = Js[new].astype(index_dtype)
    rowptr = np.zeros(m + 1, dtype=index_dtype)
    np.add.at(rowptr, Is[new], 1)
    rowptr[0] = 1
    np.cumsum(rowptr, out=rowptr)
    return CSR(m, n, rowptr, colval, nzval)


def spmv_csr(b, x, rowptr, colval, nzval):
    """src/sparse_utils.jl:649-669, literal: bi = 0; bi += aij*xj ascending p; unfused."""
    for row in range(len(b)):
        bi = 0.0
        for p in range(rowptr[row] - 1, rowptr[row + 1] - 1):
            bi = bi + nzval[p] * x[colval[p] - 1]
        b[row] = bi
    return b


def spmv_csc(b, x, colptr, rowval, nzval):
    """src/sparse_utils.jl:671-690 (default CSC storage): b=0; b[row] += aij*xj column by column."""
    b[:] = 0.0
    for col in range(len(x)):
        xj = x[col]
        for p in range(colptr[col] - 1, colptr[col + 1] - 1):
            b[rowval[p] - 1] = b[rowval[p] - 1] + nzval[p] * xj
    return b


def mul5_csr(y, A: CSR, x, alpha, beta):
    """SparseMatricesCSR.mul!(y,A,x,alpha,beta) (v0.6, third-party; restated):
    beta-scale y (rmul! / fill! 0), then y[row] += nzval*x[col]*alpha row by row."""
    if beta != 1:
        if beta != 0:
            y *= beta
        else:
            y[:] = 0.0
    for row in range(A.m):
        for p in range(A.rowptr[row] - 1, A.rowptr[row + 1] - 1):
            y[row] = y[row] + A.nzval[p] * x[A.colval[p] - 1] * alpha
    return y


def mul5_csc(y, x, colptr, rowval, nzval, alpha, beta):
    """SparseArrays.mul!(C,A::SparseMatrixCSC,B,alpha,beta) (Julia stdlib, third-party to the reference: reached from
    src/p_sparse_matrix.jl:2116-2138 when the blocks keep the DEFAULT CSC storage; restated from the published algorithm):
    beta-scale C (rmul! / fill! 0), then column by column  axj = B[col]*alpha;  C[row] += nzval*axj  -- the scalar multiplies the
    VECTOR entry first, a*(x*alpha), where SparseMatricesCSR forms (a*x)*alpha.  1-based arrays."""
    if beta != 1:
        if beta != 0:
            y *= beta
        else:
            y[:] = 0.0
    for col in range(len(x)):
        axj = x[col] * alpha
        for p in range(colptr[col] - 1, colptr[col + 1] - 1):
            y[rowval[p] - 1] = y[rowval[p] - 1] + nzval[p] * axj
    return y


def csr_to_csc(A: CSR):
    """Same matrix in CSC (1-based); rows ascending inside a column."""
    rows = np.repeat(np.arange(1, A.m + 1), np.diff(A.rowptr.astype(I64)))
    order = np.lexsort((rows, A.colval))
    colptr = np.zeros(A.n + 1, dtype=I64)
    np.add.at(colptr, A.colval, 1)
    colptr[0] = 1
    np.cumsum(colptr, out=colptr)
    return colptr, rows[order], A.nzval[order]


# --------------------------------------------------------------------------------------
# src/p_sparse_matrix.jl -- psparse(assembled=true), split format, mul!
# --------------------------------------------------------------------------------------
@dataclass
class SplitBlocks:
    """src/p_sparse_matrix.jl:588-627: own_own (n x n), own_ghost (n x g), ghost_own, ghost_ghost."""
    own_own: CSR
    own_ghost: CSR
    ghost_own: CSR
    ghost_ghost: CSR


def split_format_locally(A: CSR, rows: Indices, cols: Indices):
    """src/p_sparse_matrix.jl:823-899. Routes each stored entry by (perm[i]<=n_own, perm[j]<=n_own).

    The ghost-row branches of the reference carry latent index bugs (:872,:877-878);
    they are restated as evidently intended (ip-n_own_rows, jp-n_own_cols) and only reachable
    for non-assembled matrices.
    """
    rp, cp = rows.perm, cols.perm
    nor, noc = rows.n_own, cols.n_own
    ri = np.repeat(np.arange(1, A.m + 1), np.diff(A.rowptr.astype(I64)))
    ip = rp[ri - 1].astype(I64)
    jp = cp[A.colval - 1].astype(I64)
    v = A.nzval
    oo = (ip <= nor) & (jp <= noc)
    og = (ip <= nor) & ~(jp <= noc)
    go = ~(ip <= nor) & (jp <= noc)
    gg = ~(ip <= nor) & ~(jp <= noc)
    idt = A.rowptr.dtype
    return SplitBlocks(
        compresscoo_csr(ip[oo], jp[oo], v[oo], nor, noc, skip=False, index_dtype=idt),
        compresscoo_csr(ip[og], jp[og] - noc, v[og], nor, cols.n_ghost, skip=False, index_dtype=idt),
        compresscoo_csr(ip[go] - nor, jp[go], v[go], rows.n_ghost, noc, skip=False, index_dtype=idt),
        compresscoo_csr(ip[gg] - nor, jp[gg] - noc, v[gg], rows.n_ghost, cols.n_ghost, skip=False, index_dtype=idt),
    )


@dataclass
class PSparse:
    matrix_partition: list      # list of CSR (local, n_local_rows x n_local_cols)
    blocks: list                # list of SplitBlocks
    rows: list
    cols: list
    assembled: bool


def psparse_assembled(I, J, V, rows, cols, index_dtype=I32):
    """psparse(T,I,J,V,rows,cols;assembled=true), src/p_sparse_matrix.jl:1249-1270:
    map_global_to_local! (src/p_range.jl:287,298) -> sparse_matrix -> compresscoo (skip=true)."""
    mats, blocks = [], []
    for Ii, Ji, Vi, r, c in zip(I, J, V, rows, cols):
        li = r.global_to_local(Ii)
        lj = c.global_to_local(Ji)
        A = compresscoo_csr(li, lj, Vi, r.n_local, c.n_local, skip=True, index_dtype=index_dtype)
        mats.append(A)
        blocks.append(split_format_locally(A, r, c))
    return PSparse(mats, blocks, rows, cols, True)


def mul(c, A: PSparse, b, cache=None):
    """mul!(c,a,b), src/p_sparse_matrix.jl:2090-2103 (assembled): consistent!(b);
    c_own = A_oo*b_own (spmv!); wait; c_own += A_oh*b_ghost (muladd!)."""
    assert A.assembled
    consistent(b, A.cols, cache)
    for ci, bi, blk, r, col in zip(c, b, A.blocks, A.rows, A.cols):
        co = np.zeros(r.n_own)
        oracle_c().spmv_csr(co, bi[col.own_to_local - 1].copy(), blk.own_own)
        oracle_c().mul5_csr(co, blk.own_ghost, bi[col.ghost_to_local - 1].copy(), 1.0, 1.0)
        ci[r.own_to_local - 1] = co
    return c


def mul5(c, A: PSparse, b, alpha, beta, cache_b=None, cache_c=None):
    """mul!(c,a,b,alpha,beta), src/p_sparse_matrix.jl:2105-2142 (both assembled and sub-assembled)."""
    consistent(b, A.cols, cache_b)
    K = oracle_c()
    for ci, bi, blk, r, col in zip(c, b, A.blocks, A.rows, A.cols):
        bo = bi[col.own_to_local - 1].copy()
        bh = bi[col.ghost_to_local - 1].copy()
        co = ci[r.own_to_local - 1].copy()
        K.mul5_csr(co, blk.own_own, bo, alpha, beta)            # beta-scale then += alpha*A_oo*bo
        if not A.assembled:
            ch = ci[r.ghost_to_local - 1].copy()
            K.mul5_csr(ch, blk.ghost_own, bo, alpha, beta)
        K.mul5_csr(co, blk.own_ghost, bh, alpha, 1.0)
        ci[r.own_to_local - 1] = co
        if not A.assembled:
            K.mul5_csr(ch, blk.ghost_ghost, bh, alpha, 1.0)
            ci[r.ghost_to_local - 1] = ch
    if not A.assembled:
        assemble(c, A.rows, cache_c)
    return c


def mul5_transpose(c, A: PSparse, b, alpha, beta, cache_c=None):
    """mul!(c,transpose(a),b,alpha,beta), src/p_sparse_matrix.jl:2144-2162 (assembled a): ghost(c) = alpha*A_oh'*b_own,
    assemble!(c) started, own(c) = beta*own(c) + alpha*A_oo'*b_own, then the ghost contributions arrive (+)."""
    assert A.assembled
    K = oracle_c()
    for ci, bi, blk, r, col in zip(c, b, A.blocks, A.rows, A.cols):
        bo = bi[r.own_to_local - 1].copy()
        ch = np.zeros(col.n_ghost)
        K.mul5_csr_t(ch, blk.own_ghost, bo, alpha, 1.0)
        co = ci[col.own_to_local - 1].copy()
        K.mul5_csr_t(co, blk.own_own, bo, alpha, beta)
        ci[col.ghost_to_local - 1] = ch
        ci[col.own_to_local - 1] = co
    assemble(c, A.cols, cache_c)          # unpack adds into the already updated own values, then ghosts := 0
    return c


def mul_no_lat(c, A: PSparse, b, cache=None):
    """HPCG/src/hpcg_utils.jl:6-17: blocking consistent!, then ONE unsplit local CSR spmv!."""
    consistent(b, A.cols, cache)
    for ci, bi, M, r in zip(c, b, A.matrix_partition, A.rows):
        co = np.zeros(r.n_own)
        oracle_c().spmv_csr(co, bi, CSR(r.n_own, M.n, M.rowptr[:r.n_own + 1], M.colval, M.nzval))
        ci[r.own_to_local - 1] = co
    return c


# --------------------------------------------------------------------------------------
# Workload generators
# --------------------------------------------------------------------------------------
def hpcg_build_matrix(nx, ny, nz, gnx, gny, gnz, gix0, giy0, giz0):
    """HPCG/src/sparse_matrix.jl:27-80 (vectorised; same COO stream order: iz,iy,ix then sz,sy,sx)."""
    ix = np.arange(nx, dtype=I64)
    iy = np.arange(ny, dtype=I64)
    iz = np.arange(nz, dtype=I64)
    gix = (gix0 + ix)[None, None, :, None, None, None]
    giy = (giy0 + iy)[None, :, None, None, None, None]
    giz = (giz0 + iz)[:, None, None, None, None, None]
    s = np.array([-1, 0, 1], dtype=I64)
    sx = s[None, None, None, None, None, :]
    sy = s[None, None, None, None, :, None]
    sz = s[None, None, None, :, None, None]
    row = (giz - 1) * gnx * gny + (giy - 1) * gnx + (gix - 1) + 1
    col = row + sz * gnx * gny + sy * gnx + sx
    ok = ((giz + sz > 0) & (giz + sz < gnz + 1) & (giy + sy > 0) & (giy + sy < gny + 1)
          & (gix + sx > 0) & (gix + sx < gnx + 1))
    shape = (nz, ny, nx, 3, 3, 3)
    row_b = np.broadcast_to(row, shape)[ok]
    col_b = np.broadcast_to(col, shape)[ok]
    val = np.where(col_b == row_b, 26.0, -1.0)
    cnt = np.broadcast_to(ok, shape).reshape(nz * ny * nx, 27).sum(axis=1)
    b = 27.0 - cnt
    rows_b = np.broadcast_to(row, shape)[..., 0, 0, 0].reshape(-1).copy()
    return row_b.copy(), col_b.copy(), val, b.astype(F64), rows_b


class _MixedBaseCounter:
    """HPCG/src/mixed_base_counter.jl:1-5 (1-based there; arrays of 33 as there)."""

    def __init__(self, length, max_counts, cur_counts):
        self.length, self.max_counts, self.cur_counts = length, max_counts, cur_counts


def _mbc(counts, l):
    """mixedbasecounter(counts, l), mixed_base_counter.jl:7-17."""
    mx, cur = [0] * 33, [0] * 33
    for i in range(l):
        mx[i] = counts[i]
    mx[l] = 0
    return _MixedBaseCounter(l, mx, cur)


def _mbc1(left, right):
    """mixedbasecounter1(left, right), mixed_base_counter.jl:19-27."""
    mx, cur = [0] * 33, [0] * 33
    for i in range(left.length):
        mx[i] = left.max_counts[i] - right.cur_counts[i]
    return _MixedBaseCounter(left.length, mx, cur)


def _mbc_next(c):
    """next(counter), mixed_base_counter.jl:29-39."""
    for i in range(c.length):
        c.cur_counts[i] += 1
        if c.cur_counts[i] > c.max_counts[i]:
            c.cur_counts[i] = 0
            continue
        break
    return c


def _mbc_nonzero(c):
    """is_zero(counter), mixed_base_counter.jl:41-48 -- true while some digit is NOT zero (the name says the opposite)."""
    return any(c.cur_counts[i] != 0 for i in range(c.length))


def _mbc_product(c, multipliers):
    """product(counter, multipliers), mixed_base_counter.jl:50-61: 0 for the all-zero counter."""
    k, x = 0, 1
    for i in range(c.length):
        for _ in range(int(c.cur_counts[i])):
            k = 1
            x *= multipliers[i]
    return x * k


def compute_optimal_shape_xyz(np_):
    """HPCG/src/compute_optimal_xyz.jl:8-64, statement by statement (Primes.factor into a SortedDict: primes ascending)."""
    if np_ == 1:
        return 1, 1, 1
    factors, m, d = {}, np_, 2
    while m > 1:
        while m % d == 0:
            factors[d] = factors.get(d, 0) + 1
            m //= d
        d += 1
    primes = sorted(factors)
    z = 0
    x = primes[0]
    y = primes[1] if len(primes) > 1 else None
    if len(primes) == 1:
        z = x ** int(np.floor(factors[x] / 3))
        y = x ** int(np.floor(factors[x] / 3 + (1 if (factors[x] % 3) >= 2 else 0)))
        x = x ** int(np.floor(factors[x] / 3 + (1 if (factors[x] % 3) >= 1 else 0)))
    elif len(primes) == 2 and factors[x] == 1 and factors[y] == 1:
        z = 1
    elif len(primes) == 2 and factors[x] + factors[y] == 3:
        z = x if factors[x] == 2 else y
    elif len(primes) == 3 and factors[x] == 1 and factors[y] == 1 and factors[primes[2]] == 1:
        z = primes[2]
    else:
        powers = [factors[q] for q in primes]
        c_main = _mbc(powers, len(primes))
        c1 = _mbc(powers, len(primes))
        min_area = 2.0 * np_ + 1.0
        c1 = _mbc_next(c1)
        while _mbc_nonzero(c1):
            c2 = _mbc1(c_main, c1)
            c2 = _mbc_next(c2)
            while _mbc_nonzero(c2):
                tf1 = _mbc_product(c1, primes)
                tf2 = _mbc_product(c2, primes)
                tf3 = np_ / tf1 / tf2
                area = tf1 * tf2 + tf2 * tf3 + tf1 * tf3
                if area < min_area:
                    min_area = area
                    x, y, z = tf1, tf2, tf3
                c2 = _mbc_next(c2)
            c1 = _mbc_next(c1)
    return int(x), int(y), int(np.floor(z))


def hpcg_build_p_matrix(nx, ny, nz, npx, npy, npz, index_dtype=I32):
    """HPCG/src/sparse_matrix.jl:105-122. Returns (PSparse, b values, row partition)."""
    gnx, gny, gnz = npx * nx, npy * ny, npz * nz
    row_partition = uniform_partition((npx, npy, npz), (gnx, gny, gnz))
    I, J, V, B, IB = [], [], [], [], []
    for rows in row_partition:
        g0 = int(rows.own_to_global[0])
        cx, cy, cz = _cartesian(g0, (gnx, gny, gnz))
        i, j, v, b, ib = hpcg_build_matrix(nx, ny, nz, gnx, gny, gnz, cx, cy, cz)
        I.append(i); J.append(j); V.append(v); B.append(b); IB.append(ib)
    J_owner = find_owner(row_partition, J)
    col_partition = [union_ghost(r, j, o) for r, j, o in zip(row_partition, J, J_owner)]
    A = psparse_assembled(I, J, V, row_partition, col_partition, index_dtype)
    # b = pvector(I_b,b,row_partition): values land on own rows (all I_b are own)
    bvals = []
    for rows, b, ib in zip(col_partition, B, IB):
        v = np.zeros(rows.n_local)
        np.add.at(v, rows.global_to_local(ib) - 1, b)
        bvals.append(v)
    return A, bvals, row_partition


def laplacian_fdm(nodes_per_dir, parts_per_dir):
    """src/gallery.jl:12-86: per part COO of the (2D+1)-point Laplacian, alpha = prod(n_d+1)."""
    D = len(nodes_per_dir)
    alpha = float(np.prod([n + 1 for n in nodes_per_dir]))
    node_partition = uniform_partition(tuple(parts_per_dir), tuple(nodes_per_dir))
    Is, Js, Vs = [], [], []
    for nodes in node_partition:
        I, J, V = [], [], []
        lens = [hi - lo + 1 for lo, hi in nodes.ranges]
        for ci0 in itertools.product(*[range(L) for L in reversed(lens)]):
            ci = tuple(nodes.ranges[d][0] + c for d, c in enumerate(reversed(ci0)))
            node_i = _linear(ci, nodes_per_dir)
            I.append(node_i); J.append(node_i); V.append(alpha * 2 * D)
            for d in range(D):
                for i in (-1, 1):
                    cj = list(ci)
                    cj[d] += i
                    if not (1 <= cj[d] <= nodes_per_dir[d]):
                        continue
                    I.append(node_i); J.append(_linear(cj, nodes_per_dir)); V.append(-alpha)
        Is.append(np.array(I, I64)); Js.append(np.array(J, I64)); Vs.append(np.array(V, F64))
    return Is, Js, Vs, node_partition, node_partition


def laplacian_fdm_fast(nodes_per_dir, parts_per_dir):
    """Vectorised twin of laplacian_fdm (same COO stream order), for 64^3-size cases."""
    D = len(nodes_per_dir)
    n = tuple(nodes_per_dir)
    alpha = float(np.prod([k + 1 for k in n]))
    node_partition = uniform_partition(tuple(parts_per_dir), n)
    strides = [int(np.prod(n[:d])) for d in range(D)]
    Is, Js, Vs = [], [], []
    for nodes in node_partition:
        gid = nodes.own_to_global                                   # column-major over own box
        r = gid - 1
        coords = []
        for d in range(D):
            coords.append(r % n[d] + 1)
            r = r // n[d]
        cols = [gid]
        oks = [np.ones(len(gid), dtype=bool)]
        vals = [np.full(len(gid), alpha * 2 * D)]
        for d in range(D):
            for i in (-1, 1):
                c = coords[d] + i
                oks.append((c >= 1) & (c <= n[d]))
                cols.append(gid + i * strides[d])
                vals.append(np.full(len(gid), -alpha))
        ok = np.stack(oks, axis=1)
        Jm = np.stack(cols, axis=1)
        Vm = np.stack(vals, axis=1)
        Im = np.broadcast_to(gid[:, None], Jm.shape)
        Is.append(Im[ok].copy()); Js.append(Jm[ok].copy()); Vs.append(Vm[ok].copy())
    return Is, Js, Vs, node_partition, node_partition


def psparse_from_coo(I, J, V, row_partition, index_dtype=I32):
    """The assembled=true route used by HPCG and test/gallery_tests.jl:33:
    find_owner -> union_ghost (cols) -> psparse(...;assembled=true)."""
    J_owner = find_owner(row_partition, J)
    cols = [union_ghost(r, j, o) for r, j, o in zip(row_partition, J, J_owner)]
    return psparse_assembled(I, J, V, row_partition, cols, index_dtype)


def _fdiv(a, b):
    """a / b as Julia's Float64 division gives it: 0/0 = NaN, x/0 = +-Inf (python raises ZeroDivisionError)."""
    if b != 0.0:
        return a / b
    if a == 0.0 or a != a:
        return float("nan")
    import math
    return math.copysign(float("inf"), a) * math.copysign(1.0, b)


def _converged(residual, residual0, tolerance):
    """residual/residual0 <= tolerance (HPCG/src/ref_cg.jl:23) with Julia's semantics: 0/0 is NaN (comparison false: a zero
    right-hand side iterates to maxiter), x/0 is Inf."""
    if residual0 == 0.0:
        return False if (residual == 0.0 or residual != residual) else float("inf") <= tolerance
    return residual / residual0 <= tolerance


def ref_cg(x, A: PSparse, b, maxiter=50, tolerance=0.0, history=None, mv=None):
    """HPCG/src/ref_cg.jl:40-134 with Pl = Identity(): x, residual0, residual, iters (lists of local arrays).
    mv: the product used (default mul_no_lat!, as HPCG; pass `mul` for a matrix that only has its split blocks)."""
    mul_no_lat = mv or globals()["mul_no_lat"]
    ind = A.cols
    u = [np.zeros_like(v) for v in x]
    r = [v.copy() for v in b]
    c = [np.zeros_like(v) for v in x]
    mul_no_lat(c, A, x)
    for ri, ci, i in zip(r, c, ind):
        ri[:i.n_own] -= ci[:i.n_own]
    residual0 = residual = norm2(r, ind)
    rho, iters = 1.0, 0
    while not (iters >= maxiter or _converged(residual, residual0, tolerance)):
        for ci, ri in zip(c, r):
            ci[:] = ri
        rho_prev = rho
        rho = dot(c, r, ind)
        beta = _fdiv(rho, rho_prev)
        for ui, ci, i in zip(u, c, ind):
            ui[:i.n_own] = ci[:i.n_own] + beta * ui[:i.n_own]
        mul_no_lat(c, A, u)
        alpha = _fdiv(rho, dot(u, c, ind))
        for xi, ui, ri, ci, i in zip(x, u, r, c, ind):
            xi[:i.n_own] += alpha * ui[:i.n_own]
            ri[:i.n_own] -= alpha * ci[:i.n_own]
        residual = norm2(r, ind)
        iters += 1
        if history is not None:
            history.append(residual)
    return x, residual0, residual, iters


# --------------------------------------------------------------------------------------
# Disassembled COO -> assembled split matrix (BASELINE config 5: FEM-style, ghost-heavy)
# --------------------------------------------------------------------------------------
def _coo_of(A: CSR):
    """nziterator / findnz of a CSR block: row-major, columns ascending (src/sparse_utils.jl:96-123)."""
    rows = np.repeat(np.arange(1, A.m + 1, dtype=I64), np.diff(A.rowptr.astype(I64)))
    return rows, A.colval.astype(I64), A.nzval.copy()


def _jag_by_owner(owner, parts_snd, arrays):
    """Group entries by destination part (order of parts_snd), stable inside a group."""
    slot = np.searchsorted(parts_snd, owner)
    order = np.argsort(slot, kind="stable")
    ptrs = np.zeros(len(parts_snd) + 1, dtype=I32)
    np.add.at(ptrs, slot + 1, 1)
    length_to_ptrs(ptrs)
    return [Jagged(a[order], ptrs) for a in arrays]


def psparse_disassembled(I, J, V, rows, cols, index_dtype=I32):
    """psparse(I,J,V,rows,cols) with the default flags (disassembled input, assemble=true, split format):
    src/p_sparse_matrix.jl:1183-1219, then assemble(B,rows) = psparse_assemble_impl :1590-1756."""
    I_owner = find_owner(rows, I)
    J_owner = find_owner(cols, J)
    rows_sa = [union_ghost(r, i, o) for r, i, o in zip(rows, I, I_owner)]
    cols_sa = [union_ghost(c, j, o) for c, j, o in zip(cols, J, J_owner)]
    blocks = []
    for Ii, Ji, Vi, r, c in zip(I, J, V, rows_sa, cols_sa):
        A = compresscoo_csr(r.global_to_local(Ii), c.global_to_local(Ji), Vi, r.n_local, c.n_local,
                            skip=True, index_dtype=index_dtype)
        blocks.append(split_format_locally(A, r, c))
    return psparse_assemble(blocks, rows_sa, cols_sa, rows, index_dtype), (blocks, rows_sa, cols_sa)


def psparse_assemble(blocks, rows_sa, cols_sa, rows, index_dtype=I32):
    """psparse_assemble_impl (src/p_sparse_matrix.jl:1590-1756): ship ghost-row triplets to the owners,
    concatenate own + received COO, renumber the ghost columns (union_ghost), compress with +."""
    parts_snd, parts_rcv = assembly_neighbors(rows_sa)
    I_snd, J_snd, V_snd = [], [], []
    for blk, ps, r, c in zip(blocks, parts_snd, rows_sa, cols_sa):          # setup_cache_snd :1598-1650
        gi, gj, gv = _coo_of(blk.ghost_own)
        hi, hj, hv = _coo_of(blk.ghost_ghost)
        gI = r.ghost_to_global[np.concatenate([gi, hi]) - 1]
        gJ = np.concatenate([c.own_to_global[gj - 1], c.ghost_to_global[hj - 1]])
        gV = np.concatenate([gv, hv])
        owner = r.ghost_to_owner[np.concatenate([gi, hi]) - 1]
        a, b, v = _jag_by_owner(owner, ps, [gI, gJ, gV])
        I_snd.append(a); J_snd.append(b); V_snd.append(v)
    I_rcv = allocate_exchange_jagged(I_snd, parts_snd, parts_rcv, I64)
    J_rcv = allocate_exchange_jagged(J_snd, parts_snd, parts_rcv, I64)
    V_rcv = allocate_exchange_jagged(V_snd, parts_snd, parts_rcv, F64)
    exchange_jagged(I_rcv, I_snd, parts_snd, parts_rcv)
    exchange_jagged(J_rcv, J_snd, parts_snd, parts_rcv)
    exchange_jagged(V_rcv, V_snd, parts_snd, parts_rcv)
    trip, Jg = [], []
    for blk, Ir, Jr, Vr, r, c in zip(blocks, I_rcv, J_rcv, V_rcv, rows_sa, cols_sa):   # setup_own_triplets :1656-1689
        ooI, ooJ, ooV = _coo_of(blk.own_own)
        ohI, ohJ, ohV = _coo_of(blk.own_ghost)
        lj = c.global_to_local(Jr.data).astype(I64)
        is_own = (lj >= 1) & (lj <= c.n_own)
        li = r.global_to_local(Ir.data).astype(I64)                  # received rows are own rows here
        oo = (np.concatenate([ooI, li[is_own]]), np.concatenate([ooJ, lj[is_own]]), np.concatenate([ooV, Vr.data[is_own]]))
        ohJ_g = c.ghost_to_global[ohJ - 1]
        og = (np.concatenate([ohI, li[~is_own]]), np.concatenate([ohJ_g, Jr.data[~is_own]]),
              np.concatenate([ohV, Vr.data[~is_own]]))
        trip.append((oo, og)); Jg.append(og[1])
    J_owner = find_owner(cols_sa, Jg)
    cols0 = [Indices(c.n_global, c.part, c.own_to_global, np.full(c.n_own, c.part, I32), "block", c.np_, c.n, c.ranges, c.starts)
             for c in cols_sa]                                       # remove_ghost :1727
    cols_fa = [union_ghost(c, j, o) for c, j, o in zip(cols0, Jg, J_owner)]
    out_blocks = []
    for (oo, og), r, c in zip(trip, rows, cols_fa):                                     # finalize_values :1690-1723
        gj = c.global_to_local(og[1]).astype(I64) - c.n_own           # map_global_to_ghost!
        A1 = compresscoo_csr(oo[0], oo[1], oo[2], r.n_own, c.n_own, skip=False, index_dtype=index_dtype)
        A2 = compresscoo_csr(og[0], gj, og[2], r.n_own, c.n_ghost, skip=False, index_dtype=index_dtype)
        E1 = compresscoo_csr([], [], [], 0, c.n_own, skip=False, index_dtype=index_dtype)
        E2 = compresscoo_csr([], [], [], 0, c.n_ghost, skip=False, index_dtype=index_dtype)
        out_blocks.append(SplitBlocks(A1, A2, E1, E2))
    return PSparse([None] * len(rows), out_blocks, rows, cols_fa, True)


def laplacian_fem(nodes_per_dir, parts_per_dir):
    """src/gallery.jl:110-239: Q1 Laplacian on the unit cube, free (interior) nodes only; each part loops over ITS
    CELLS (uniform cell partition) and emits element entries in global node ids -> rows of other parts appear."""
    D = len(nodes_per_dir)
    cells_per_dir = tuple(n + 1 for n in nodes_per_dir)
    h = [1.0 / (n + 1) for n in nodes_per_dir]
    gp = np.array([-np.sqrt(3.0) / 3.0, np.sqrt(3.0) / 3.0])
    sf = np.zeros((2, 2)); sf[:, 0] = 0.5 * (1 - gp); sf[:, 1] = 0.5 * (gp + 1)
    sg1 = np.zeros((2, 2)); sg1[:, 0] = -0.5; sg1[:, 1] = 0.5
    nloc = 2 ** D
    loc = [tuple(reversed(t)) for t in itertools.product(*[range(1, 3)] * D)]       # column-major local nodes / points
    sg = np.zeros((nloc, nloc, D))
    for ln, lt in enumerate(loc):
        for pt, ptt in enumerate(loc):
            for d in range(D):
                v = 1.0
                for i in range(D):                                  # prod(1:D) in index order (:141-147)
                    v = v * ((2.0 / h[d]) * sg1[lt[d] - 1, ptt[d] - 1] if i == d else sf[lt[i] - 1, ptt[i] - 1])
                sg[ln, pt, d] = v
    # NOTE restated literally, index order included: sf_1d/sg_1d are filled as [gauss point, shape function] but read
    # as [local_node, point] (:143-145), and sg is filled as [local_node, point] but summed over its FIRST index
    # (Aref[i,j] += dV*dot(sg[k,i],sg[k,j]), :154-160).
    Aref = np.zeros((nloc, nloc))
    dV = float(np.prod(h)) / (2 ** D)
    for i in range(nloc):
        for j in range(nloc):
            for k in range(nloc):
                Aref[i, j] += dV * float(np.dot(sg[k, i], sg[k, j]))
    node_partition = uniform_partition(tuple(parts_per_dir), tuple(nodes_per_dir))
    cell_partition = uniform_partition(tuple(parts_per_dir), cells_per_dir)
    Is, Js, Vs = [], [], []
    for cells in cell_partition:
        I, J, V = [], [], []
        lens = [hi - lo + 1 for lo, hi in cells.ranges]
        for c0 in itertools.product(*[range(L) for L in reversed(lens)]):
            cell = tuple(cells.ranges[d][0] + c for d, c in enumerate(reversed(c0)))
            for li, lti in enumerate(loc):
                ni = tuple(cell[d] + lti[d] - 2 for d in range(D))
                if any(not (1 <= ni[d] <= nodes_per_dir[d]) for d in range(D)):
                    continue
                for lj, ltj in enumerate(loc):
                    nj = tuple(cell[d] + ltj[d] - 2 for d in range(D))
                    if any(not (1 <= nj[d] <= nodes_per_dir[d]) for d in range(D)):
                        continue
                    I.append(_linear(ni, nodes_per_dir)); J.append(_linear(nj, nodes_per_dir)); V.append(Aref[li, lj])
        Is.append(np.array(I, I64)); Js.append(np.array(J, I64)); Vs.append(np.array(V, F64))
    return Is, Js, Vs, node_partition, node_partition


# --------------------------------------------------------------------------------------
# HPCG multigrid preconditioner: HPCG/src/mg_preconditioner.jl + PartitionedSolvers GS smoother
# --------------------------------------------------------------------------------------
def pvector_disassembled(I, V, rows):
    """pvector(I,V,rows)|>fetch, default flags (src/p_vector.jl:887-925): union_ghost, dense_vector (:853-863:
    a[i] += v in entry order, ids < 1 skipped), then assemble(A,rows) (:1331-1345).  Returns the own values per part."""
    I_owner = find_owner(rows, I)
    rows_sa = [union_ghost(r, i, o) for r, i, o in zip(rows, I, I_owner)]
    vals = []
    for r, gi, v in zip(rows_sa, I, V):
        a = np.zeros(r.n_local)
        for k, x in zip(r.global_to_local(gi), v):
            if k >= 1:
                a[k - 1] += x
        vals.append(a)
    assemble(vals, rows_sa)
    return [a[:r.n_own].copy() for a, r in zip(vals, rows_sa)]


# --------------------------------------------------------------------------------------
# test/fem_example.jl (BASELINE config 5): the set-up loops, literally
# --------------------------------------------------------------------------------------
def fem_example_setup(parts_per_dir=(2, 2), cells_per_dir=(10, 10), length_in_x=2.0):
    """test/fem_example.jl:13-29 (setup_params), :31-67 (setup_grid), :69-113 (setup_space), :115-139
    (setup_cell_dofs), :275-276 (consistent! of the cell -> global dofs table), :141-168 (finish_cell_dofs), :170-198
    (setup_IJV), :200-236 (setup_b): cell by cell, element node by element node, exactly the reference's loops.
    Returns per part I, J, V, II, VV, the dof partition and a function giving the exact solution at a global dof."""
    P = int(np.prod(parts_per_dir))
    cx, cy = cells_per_dir
    nodes = (cx + 1, cy + 1)
    h = max(length_in_x / cx, length_in_x / cy)
    Ae = (h ** 2 / 6) * np.array([[4.0, -1.0, -1.0, -2.0], [-1.0, 4.0, -2.0, -1.0], [-1.0, -2.0, 4.0, -1.0],
                                  [-2.0, -1.0, -1.0, 4.0]])
    cells = uniform_partition(parts_per_dir, cells_per_dir, (True, True))
    enodes = [(1, 1), (2, 1), (1, 2), (2, 2)]                      # CartesianIndices((2,2)), linear order
    is_bnd = lambda node_1d, nodes_1d: node_1d == 1 or node_1d == nodes_1d
    grids, spaces = [], []
    for ind in cells:
        l2g = ind.local_to_global
        first, last = int(l2g[0]), int(l2g[-1])
        fc = ((first - 1) % cx + 1, (first - 1) // cx + 1)         # linear_to_cartesian_global_cell[first] (:46-49)
        lc = ((last - 1) % cx + 1, (last - 1) // cx + 1)
        ncx, ncy = lc[0] - fc[0] + 1, lc[1] - fc[1] + 1
        nnx = ncx + 1
        loc_cells = [(i, j) for j in range(1, ncy + 1) for i in range(1, ncx + 1)]   # linear_to_cartesian_local_cell
        lnode = lambda i, j, nnx=nnx: (i - 1) + nnx * (j - 1) + 1   # cartesian_to_linear_local_node
        lcell = lambda i, j, ncx=ncx: (i - 1) + ncx * (j - 1) + 1
        gnode = lambda i, j, fc=fc: (fc[0] + i - 1, fc[1] + j - 1)  # local_to_global_cartesian_node
        # setup_space
        node_to_dof = [1] * (nnx * (ncy + 1))
        for (ci, cj) in loc_cells:
            for (ei, ej) in enodes:
                gi, gj = fc[0] + ci - 1 + (ei - 1), fc[1] + cj - 1 + (ej - 1)
                if is_bnd(gi, nodes[0]) or is_bnd(gj, nodes[1]):
                    node_to_dof[lnode(ci + ei - 1, cj + ej - 1) - 1] = 0
        dof_to_node = [k + 1 for k, v in enumerate(node_to_dof) if v == 1]
        n_local_dofs = len(dof_to_node)
        for d, nd in enumerate(dof_to_node, start=1):
            node_to_dof[nd - 1] = d
        dof_owner = [0] * n_local_dofs
        for (ci, cj) in loc_cells:
            owner = int(ind.local_to_owner[lcell(ci, cj) - 1])
            for (ei, ej) in enodes:
                d = node_to_dof[lnode(ci + ei - 1, cj + ej - 1) - 1]
                if d > 0:
                    dof_owner[d - 1] = max(dof_owner[d - 1], owner)
        n_own = sum(1 for o in dof_owner if o == ind.part)
        grids.append(dict(ind=ind, fc=fc, ncx=ncx, ncy=ncy, loc_cells=loc_cells, lnode=lnode, lcell=lcell, gnode=gnode))
        spaces.append(dict(node_to_dof=node_to_dof, dof_owner=dof_owner, n_own=n_own,
                           cell_dofs=[[0, 0, 0, 0] for _ in loc_cells]))
    n_global = sum(s["n_own"] for s in spaces)
    dofs = variable_partition([s["n_own"] for s in spaces], n_global)
    # setup_cell_dofs
    for g, s, dp in zip(grids, spaces, dofs):
        offset = int(dp.own_to_global[0]) - 1 if dp.n_own else 0
        perm = [0] * len(s["dof_owner"])
        k = 0
        for d, o in enumerate(s["dof_owner"]):
            if o == g["ind"].part:
                k += 1
                perm[d] = k
        for (ci, cj) in g["loc_cells"]:
            for e, (ei, ej) in enumerate(enodes):
                d = s["node_to_dof"][g["lnode"](ci + ei - 1, cj + ej - 1) - 1]
                if d > 0 and perm[d - 1] > 0:
                    s["cell_dofs"][g["lcell"](ci, cj) - 1][e] = perm[d - 1] + offset
    # consistent!(cell_to_global_dofs): a ghost cell's row is its owner's row
    snapshot = [[row[:] for row in s["cell_dofs"]] for s in spaces]
    for g, s in zip(grids, spaces):
        ind = g["ind"]
        for lc, (gid, owner) in enumerate(zip(ind.local_to_global, ind.local_to_owner)):
            if owner != ind.part:
                src = grids[owner - 1]["ind"]
                s["cell_dofs"][lc] = snapshot[owner - 1][int(src.global_to_local([gid])[0]) - 1][:]
    # finish_cell_dofs
    for g, s in zip(grids, spaces):
        l2g = [0] * len(s["dof_owner"])
        for (ci, cj) in g["loc_cells"]:
            for e, (ei, ej) in enumerate(enodes):
                d = s["node_to_dof"][g["lnode"](ci + ei - 1, cj + ej - 1) - 1]
                gd = s["cell_dofs"][g["lcell"](ci, cj) - 1][e]
                if d > 0 and gd > 0:
                    l2g[d - 1] = gd
        for (ci, cj) in g["loc_cells"]:
            for e, (ei, ej) in enumerate(enodes):
                d = s["node_to_dof"][g["lnode"](ci + ei - 1, cj + ej - 1) - 1]
                if d > 0:
                    assert l2g[d - 1] != 0
                    s["cell_dofs"][g["lcell"](ci, cj) - 1][e] = l2g[d - 1]
    # setup_IJV, setup_b, exact solution
    Is, Js, Vs, IIs, VVs, exact = [], [], [], [], [], {}
    for g, s in zip(grids, spaces):
        ind = g["ind"]
        I, J, V, II, VV = [], [], [], [], []
        for (ci, cj) in g["loc_cells"]:
            lc = g["lcell"](ci, cj)
            if ind.local_to_owner[lc - 1] != ind.part:
                continue
            gd = s["cell_dofs"][lc - 1]
            for er, grow in enumerate(gd):
                if grow <= 0:
                    continue
                for ec, gcol in enumerate(gd):
                    if gcol <= 0:
                        continue
                    I.append(grow), J.append(gcol), V.append(Ae[er, ec])
            ue = [0.0] * 4
            for e, (ei, ej) in enumerate(enodes):
                d = s["node_to_dof"][g["lnode"](ci + ei - 1, cj + ej - 1) - 1]
                gi, gj = g["gnode"](ci + ei - 1, cj + ej - 1)
                uval = (gi - 1) * h + (gj - 1) * h                   # u(x) = x[1] + x[2] (:11)
                if d <= 0:
                    ue[e] = uval
                else:
                    exact[gd[e]] = uval
            for er, grow in enumerate(gd):
                if grow > 0:
                    ge = ((Ae[er, 0] * ue[0] + Ae[er, 1] * ue[1]) + Ae[er, 2] * ue[2]) + Ae[er, 3] * ue[3]
                    II.append(grow), VV.append(-ge)
        Is.append(np.array(I, I64)), Js.append(np.array(J, I64)), Vs.append(np.array(V, F64))
        IIs.append(np.array(II, I64)), VVs.append(np.array(VV, F64))
    return dict(I=Is, J=Js, V=Vs, II=IIs, VV=VVs, dof_partition=dofs, exact=exact, n_global_dofs=n_global,
                n_own_dofs=[s["n_own"] for s in spaces])


def restrict_operator(nx, ny, nz):
    """HPCG/src/mg_preconditioner.jl:81-103: coarse row -> fine row (1-based), every second point per direction."""
    nxc, nyc, nzc = nx // 2, ny // 2, nz // 2
    f2c = np.zeros(nxc * nyc * nzc, dtype=I32)
    for izc in range(1, nzc + 1):
        for iyc in range(1, nyc + 1):
            for ixc in range(1, nxc + 1):
                cur = (izc - 1) * nxc * nyc + (iyc - 1) * nxc + (ixc - 1) + 1
                f2c[cur - 1] = 2 * (izc - 1) * nx * ny + 2 * (iyc - 1) * nx + 2 * (ixc - 1) + 1
    return f2c


def dense_diag(A: PSparse):
    """dense_diag!(d,A) (src/p_sparse_matrix.jl:2171-2189): diagonal of the own_own block of every part."""
    out = []
    for M, r in zip(A.matrix_partition, A.rows):
        d = np.zeros(r.n_own)
        for row in range(r.n_own):
            for p in range(M.rowptr[row] - 1, M.rowptr[row + 1] - 1):
                if M.colval[p] - 1 == row:
                    d[row] = M.nzval[p]
        out.append(d)
    return out


def gauss_seidel_step(x, A: PSparse, diag, b, zero_guess=False, cache=None):
    """gauss_seidel(p;iterations=1,sweep=:symmetric) step (PartitionedSolvers/src/smoothers.jl:105-131):
    consistent!(x) unless zero_guess; forward sweep (zero-guess variant if zero_guess); backward sweep."""
    if not zero_guess:
        consistent(x, A.cols, cache)
    lib = oracle_c().lib
    lib.orc_gs_sweep.restype = None
    lib.orc_gs_sweep.argtypes = [ctypes.c_void_p] * 6 + [ctypes.c_int64, ctypes.c_int, ctypes.c_int]
    for xi, M, d, bi, r in zip(x, A.matrix_partition, diag, b, A.rows):
        bo = np.ascontiguousarray(bi[:r.n_own])
        for backward, zero in ((0, 1 if zero_guess else 0), (1, 0)):
            lib.orc_gs_sweep(xi.ctypes.data, M.rowptr.ctypes.data, M.colval.ctypes.data, M.nzval.ctypes.data,
                             d.ctypes.data, bo.ctypes.data, r.n_own, backward, zero)
    return x


@dataclass
class MgPreconditioner:
    """Mg_preconditioner (HPCG/src/mg_preconditioner.jl:44-65), levels 1 (coarsest) .. l (finest), stored 0-based."""
    f2c: list
    A: list
    diag: list
    r: list
    x: list
    Axf: list
    l: int


def pc_setup(np3, l, nx, ny, nz):
    """pc_setup (HPCG/src/mg_preconditioner.jl:142-187): level l is the problem itself, each coarser level halves nx,ny,nz."""
    npx, npy, npz = np3
    f2c, As, diags, rs, xs, Axfs = [None] * (l - 1), [None] * l, [None] * l, [None] * l, [None] * l, [None] * l
    for lev in range(l, 0, -1):
        A, b, _ = hpcg_build_p_matrix(nx, ny, nz, npx, npy, npz)
        As[lev - 1], rs[lev - 1] = A, b
        diags[lev - 1] = dense_diag(A)
        xs[lev - 1] = [np.zeros(c.n_local) for c in A.cols]
        Axfs[lev - 1] = [np.zeros(c.n_local) for c in A.cols]
        if lev > 1:
            f2c[lev - 2] = restrict_operator(nx, ny, nz)
            nx, ny, nz = nx // 2, ny // 2, nz // 2
    return MgPreconditioner(f2c, As, diags, rs, xs, Axfs, l)


def pc_solve(x, s: MgPreconditioner, b, l, zero_guess=False):
    """pc_solve! (HPCG/src/mg_preconditioner.jl:314-329)."""
    A, d = s.A[l - 1], s.diag[l - 1]
    if l == 1:
        gauss_seidel_step(x, A, d, b, zero_guess)
    else:
        gauss_seidel_step(x, A, d, b, zero_guess)                       # presmoother
        mul_no_lat(s.Axf[l - 1], A, x)
        f2c = s.f2c[l - 2]
        for rc, rf, axf in zip(s.r[l - 2], b, s.Axf[l - 1]):             # p_restrict! on local values
            rc[:len(f2c)] = rf[f2c - 1] - axf[f2c - 1]
        for xc in s.x[l - 2]:
            xc[:] = 0.0
        pc_solve(s.x[l - 2], s, s.r[l - 2], l - 1, zero_guess=True)
        for xf, xc in zip(x, s.x[l - 2]):                                # p_prolongate!
            xf[f2c - 1] += xc[:len(f2c)]
        gauss_seidel_step(x, A, d, b)                                    # postsmoother (consistent!(x) first)
    return x


def ref_cg_mg(x, A: PSparse, b, S: MgPreconditioner, maxiter=50, tolerance=0.0, history=None):
    """ref_cg!(x,A,b,...;Pl=S) (HPCG/src/ref_cg.jl:40-134): ldiv!(c,Pl,r) = fill!(c,0); pc_solve!(c,Pl,r,l;zero_guess=true)."""
    ind = A.cols
    u = [np.zeros_like(v) for v in x]
    r = [v.copy() for v in b]
    c = [np.zeros_like(v) for v in x]
    mul(c, A, x)
    for ri, ci, i in zip(r, c, ind):
        ri[:i.n_own] -= ci[:i.n_own]
    residual0 = residual = norm2(r, ind)
    rho, iters = 1.0, 0
    while not (iters >= maxiter or _converged(residual, residual0, tolerance)):
        for ci in c:
            ci[:] = 0.0
        pc_solve(c, S, r, S.l, zero_guess=True)
        rho_prev = rho
        rho = dot(c, r, ind)
        beta = _fdiv(rho, rho_prev)
        for ui, ci, i in zip(u, c, ind):
            ui[:i.n_own] = ci[:i.n_own] + beta * ui[:i.n_own]
        mul_no_lat(c, A, u)
        alpha = _fdiv(rho, dot(u, c, ind))
        for xi, ui, ri, ci, i in zip(x, u, r, c, ind):
            xi[:i.n_own] += alpha * ui[:i.n_own]
            ri[:i.n_own] -= alpha * ci[:i.n_own]
        residual = norm2(r, ind)
        iters += 1
        if history is not None:
            history.append(residual)
    return x, residual0, residual, iters


def hash_x(gids):
    """x[gid] = ((gid*2654435761) mod 2^32)/2^32, stateless and partition independent."""
    g = np.asarray(gids, dtype=np.uint64)
    return ((g * np.uint64(2654435761)) % np.uint64(2 ** 32)).astype(F64) / float(2 ** 32)


# --------------------------------------------------------------------------------------
# C kernels of the oracle (oracle/pa_oracle.c): same loops, used for mid-size parity and timing
# --------------------------------------------------------------------------------------
class _OracleC:
    def __init__(self, path):
        self.lib = ctypes.CDLL(path)
        L = self.lib
        P = ctypes.c_void_p
        L.orc_spmv_csr.argtypes = [P, P, P, P, P, ctypes.c_int64]
        L.orc_mul5_csr.argtypes = [P, P, P, P, P, ctypes.c_int64, ctypes.c_double, ctypes.c_double]
        L.orc_mul5_csr_t.argtypes = [P, P, P, P, P, ctypes.c_int64, ctypes.c_int64, ctypes.c_double, ctypes.c_double]
        L.orc_mul5_csr_t.restype = None
        L.orc_pack.argtypes = [P, P, P, ctypes.c_int64]
        L.orc_unpack_insert.argtypes = [P, P, P, ctypes.c_int64]
        L.orc_unpack_add.argtypes = [P, P, P, ctypes.c_int64]
        for f in (L.orc_spmv_csr, L.orc_mul5_csr, L.orc_pack, L.orc_unpack_insert, L.orc_unpack_add):
            f.restype = None
        L.orc_spmv_csr_f32.argtypes = [P, P, P, P, P, ctypes.c_int64]
        L.orc_spmv_csc_f32.argtypes = [P, P, P, P, P, ctypes.c_int64, ctypes.c_int64]
        L.orc_mul5_csr_f32.argtypes = [P, P, P, P, P, ctypes.c_int64, ctypes.c_float, ctypes.c_float]
        for f in (L.orc_spmv_csr_f32, L.orc_spmv_csc_f32, L.orc_mul5_csr_f32):
            f.restype = None

    @staticmethod
    def _p(a):
        return a.ctypes.data_as(ctypes.c_void_p)

    def spmv_csr(self, b, x, A: CSR):
        assert A.rowptr.dtype == I32 and A.colval.dtype == I32
        assert b.flags.c_contiguous and x.flags.c_contiguous and b.dtype == F64 and x.dtype == F64
        self.lib.orc_spmv_csr(self._p(b), self._p(x), self._p(A.rowptr), self._p(A.colval),
                              self._p(A.nzval), len(b))
        return b

    # Float32 twins (src/sparse_utils.jl:649-690 are generic in the element type; test/sparse_utils_tests.jl:72-79 runs them in Float32)
    def spmv_csr_f32(self, b, x, rowptr, colval, nzval):
        for a, dt in ((b, np.float32), (x, np.float32), (rowptr, I32), (colval, I32), (nzval, np.float32)):
            assert a.dtype == dt and a.flags.c_contiguous
        self.lib.orc_spmv_csr_f32(self._p(b), self._p(x), self._p(rowptr), self._p(colval), self._p(nzval), len(b))
        return b

    def spmv_csc_f32(self, b, x, colptr, rowval, nzval):
        for a, dt in ((b, np.float32), (x, np.float32), (colptr, I32), (rowval, I32), (nzval, np.float32)):
            assert a.dtype == dt and a.flags.c_contiguous
        self.lib.orc_spmv_csc_f32(self._p(b), self._p(x), self._p(colptr), self._p(rowval), self._p(nzval), len(b), len(x))
        return b

    def mul5_csr_f32(self, y, x, rowptr, colval, nzval, alpha, beta):
        for a, dt in ((y, np.float32), (x, np.float32), (rowptr, I32), (colval, I32), (nzval, np.float32)):
            assert a.dtype == dt and a.flags.c_contiguous
        self.lib.orc_mul5_csr_f32(self._p(y), self._p(x), self._p(rowptr), self._p(colval), self._p(nzval), len(y), float(alpha), float(beta))
        return y

    def mul5_csr(self, y, A: CSR, x, alpha, beta):
        assert A.rowptr.dtype == I32 and A.colval.dtype == I32
        assert y.flags.c_contiguous and x.flags.c_contiguous
        self.lib.orc_mul5_csr(self._p(y), self._p(x), self._p(A.rowptr), self._p(A.colval),
                              self._p(A.nzval), A.m, float(alpha), float(beta))
        return y

    def mul5_csr_t(self, y, A: CSR, x, alpha, beta):
        assert A.rowptr.dtype == I32 and A.colval.dtype == I32 and y.flags.c_contiguous and x.flags.c_contiguous
        self.lib.orc_mul5_csr_t(self._p(y), self._p(x), self._p(A.rowptr), self._p(A.colval), self._p(A.nzval),
                                A.m, A.n, float(alpha), float(beta))
        return y

    def pack(self, buf, values, lids):
        self.lib.orc_pack(self._p(buf), self._p(values), self._p(lids), len(lids))

    def unpack_insert(self, values, buf, lids):
        self.lib.orc_unpack_insert(self._p(values), self._p(buf), self._p(lids), len(lids))

    def unpack_add(self, values, buf, lids):
        self.lib.orc_unpack_add(self._p(values), self._p(buf), self._p(lids), len(lids))


class _OraclePy:
    """Pure-python fallbacks with the same interface (tiny cases; also cross-checks the C)."""

    def spmv_csr_f32(self, b, x, rowptr, colval, nzval):
        f = np.float32
        for row in range(len(b)):
            bi = f(0)
            for p in range(rowptr[row] - 1, rowptr[row + 1] - 1):
                bi = f(bi + f(nzval[p] * x[colval[p] - 1]))
            b[row] = bi
        return b

    def spmv_csc_f32(self, b, x, colptr, rowval, nzval):
        f = np.float32
        b[:] = 0
        for col in range(len(x)):
            for p in range(colptr[col] - 1, colptr[col + 1] - 1):
                b[rowval[p] - 1] = f(b[rowval[p] - 1] + f(nzval[p] * x[col]))
        return b

    def mul5_csr_f32(self, y, x, rowptr, colval, nzval, alpha, beta):
        f = np.float32
        alpha, beta = f(alpha), f(beta)
        if beta != 1:
            y[:] = (y * beta).astype(f) if beta != 0 else 0
        for row in range(len(y)):
            for p in range(rowptr[row] - 1, rowptr[row + 1] - 1):
                y[row] = f(y[row] + f(f(nzval[p] * x[colval[p] - 1]) * alpha))
        return y

    def spmv_csr(self, b, x, A: CSR):
        return spmv_csr(b, x, A.rowptr, A.colval, A.nzval)

    def mul5_csr(self, y, A: CSR, x, alpha, beta):
        return mul5_csr(y, A, x, alpha, beta)

    def mul5_csr_t(self, y, A: CSR, x, alpha, beta):
        if beta != 1:
            y *= beta
            if beta == 0:
                y[:] = 0.0
        for row in range(A.m):
            for p in range(A.rowptr[row] - 1, A.rowptr[row + 1] - 1):
                y[A.colval[p] - 1] = y[A.colval[p] - 1] + A.nzval[p] * x[row] * alpha
        return y


_ORACLE_C = None


def oracle_c():
    """Load oracle/libpa_oracle.so (built by oracle/Makefile); pure python if it is absent."""
    global _ORACLE_C
    if _ORACLE_C is None:
        so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "libpa_oracle.so")
        _ORACLE_C = _OracleC(so) if os.path.exists(so) else _OraclePy()
    return _ORACLE_C
