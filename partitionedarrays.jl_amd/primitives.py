"""Parallel primitives on an "array of parts" and the two back-ends (host side).

Mirrors /root/reference/src/primitives.jl, src/debug_array.jl and src/mpi_array.jl for what the
mul!/consistent!/assemble! path needs at set-up time:

  DebugArray      all parts in this process, `map` is a sequential loop (src/debug_array.jl:34,110-117)
  TorchDistArray  one part per process / GPU (MPIArray analogue, src/mpi_array.jl:105); collectives go
                  through torch.distributed (gloo on CPU, RCCL = backend "nccl" on GPUs)

Part ids, MAIN and every id inside the data are 1-based, as in the reference (MAIN = 1,
src/primitives.jl:152; MPI rank = part - 1, src/mpi_array.jl:51).
"""
from __future__ import annotations

import os

from dataclasses import dataclass

import numpy as np

MAIN = 1


# ----------------------------------------------------------------------------------------------
# back-end arrays
# ----------------------------------------------------------------------------------------------
class DebugArray:
    """src/debug_array.jl:34-65: immutable wrapper; scalar indexing is an error on purpose."""

    def __init__(self, items):
        self.items = list(items)

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        raise IndexError("Scalar indexing on DebugArray is not allowed for performance reasons "
                         "(src/debug_array.jl:56-65); use pmap / gather / getany")

    def __iter__(self):
        raise TypeError("DebugArray is not iterable; use pmap (src/debug_array.jl:110)")

    def __eq__(self, other):
        return isinstance(other, DebugArray) and self.items == other.items

    def __repr__(self):
        return f"DebugArray({self.items!r})"


class TorchDistArray:
    """One item per process (MPIArray, src/mpi_array.jl:105-126)."""

    def __init__(self, item, group=None):
        import torch.distributed as dist
        self.item = item
        self.group = group
        self.rank = dist.get_rank(group)      # 0-based; part id = rank + 1
        self.size = dist.get_world_size(group)

    def __len__(self):
        return self.size

    def __getitem__(self, i):
        raise IndexError("Scalar indexing on TorchDistArray is not allowed (src/mpi_array.jl:155-157)")

    def __repr__(self):
        return f"TorchDistArray(part {self.rank + 1}/{self.size}: {self.item!r})"


def with_debug(f):
    """with_debug(f) = f(DebugArray) (src/debug_array.jl:7-9)."""
    return f(lambda a: DebugArray(list(a)))


def with_torchdist(f, group=None, abort_on_error=True):
    """with_mpi analogue (src/mpi_array.jl:64-83): `distribute` keeps this rank's item of `a`.
    An exception on any rank tears the whole job down, like with_mpi's MPI.Abort(comm,1) (:72-79): the
    traceback is printed and the process exits with status 1 (torch.distributed.run then kills the peers)."""
    import torch.distributed as dist
    if not dist.is_initialized():
        raise RuntimeError("torch.distributed is not initialised (call dist.init_process_group first)")

    def distribute(a):
        a = list(a)
        if len(a) != dist.get_world_size(group):
            raise AssertionError("number of parts must equal the number of ranks (src/mpi_array.jl:46)")
        return TorchDistArray(a[dist.get_rank(group)], group)

    if not abort_on_error:
        return f(distribute)
    try:
        return f(distribute)
    except BaseException:
        import os
        import sys
        import traceback
        traceback.print_exc()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(1)


def linear_indices(a):
    """Part ids 1..P distributed like `a` (src/primitives.jl linear_indices)."""
    if isinstance(a, DebugArray):
        return DebugArray(range(1, len(a) + 1))
    return TorchDistArray(a.rank + 1, a.group)


CURRENT_PART = [None]      # index of the part a pmap over a DebugArray is visiting (None outside)


def pmap(f, *arrays):
    """`map(f, a, b, ...)` over parts (src/debug_array.jl:110-117, src/mpi_array.jl:221-279)."""
    a0 = arrays[0]
    if isinstance(a0, DebugArray):
        n = len(a0)
        for a in arrays:
            assert isinstance(a, DebugArray) and len(a) == n
        out = []
        outer = CURRENT_PART[0]
        try:
            for i in range(n):
                # (which part's turn it is: with one device context per part -- PA_CTX_PER_PART, p_vector.context() -- everything a
                #  part creates inside the map lands in ITS context, on its GPU; a nested map over other data keeps the outer part)
                CURRENT_PART[0] = i if outer is None or n > 1 else outer
                out.append(f(*[a.items[i] for a in arrays]))
        finally:
            CURRENT_PART[0] = outer
        return DebugArray(out)
    for a in arrays:
        assert isinstance(a, TorchDistArray)
    return TorchDistArray(f(*[a.item for a in arrays]), a0.group)


def pforeach(f, *arrays):
    pmap(f, *arrays)
    return None


def tuple_of_arrays(a):
    """Array of tuples -> tuple of arrays (src/primitives.jl tuple_of_arrays)."""
    if isinstance(a, DebugArray):
        k = len(a.items[0])
        return tuple(DebugArray([t[j] for t in a.items]) for j in range(k))
    return tuple(TorchDistArray(x, a.group) for x in a.item)


def getany(a):
    """Any item (they are assumed equal), src/primitives.jl getany."""
    return a.items[0] if isinstance(a, DebugArray) else a.item


def local_items(a):
    """The items that live in THIS process (all of them for DebugArray, one for TorchDistArray)."""
    return list(a.items) if isinstance(a, DebugArray) else [a.item]


def map_main(f, *arrays, main=MAIN):
    """src/primitives.jl:185-194."""
    ranks = linear_indices(arrays[0])
    return pmap(lambda r, *xs: f(*xs) if r == main else None, ranks, *arrays)


# ----------------------------------------------------------------------------------------------
# collectives (set-up only; the per-iteration exchange runs on the device, see p_vector.py)
# ----------------------------------------------------------------------------------------------
def gather(snd, destination=MAIN):
    """src/primitives.jl:234-252: MAIN (or every part for :all) gets the list of all items."""
    if isinstance(snd, DebugArray):
        allv = list(snd.items)
        if destination == "all":
            return DebugArray([list(allv) for _ in allv])
        return DebugArray([list(allv) if p == destination else [] for p in range(1, len(allv) + 1)])
    import torch.distributed as dist
    out = [None] * snd.size
    dist.all_gather_object(out, snd.item, group=snd.group)
    if destination == "all" or snd.rank + 1 == destination:
        return TorchDistArray(out, snd.group)
    return TorchDistArray([], snd.group)


def scatter(snd, source=MAIN):
    """src/primitives.jl:357-372: part p receives snd[source][p]."""
    if isinstance(snd, DebugArray):
        src = snd.items[source - 1]
        return DebugArray([src[p] for p in range(len(snd))])
    import torch.distributed as dist
    out = [None]
    lst = list(snd.item) if snd.rank + 1 == source else None
    dist.scatter_object_list(out, lst, src=_global_rank(snd.group, source - 1), group=snd.group)
    return TorchDistArray(out[0], snd.group)


def multicast(snd, source=MAIN):
    """multicast(snd;source) (src/primitives.jl:486-561): every part receives the item part `source` holds."""
    if isinstance(snd, DebugArray):
        return DebugArray([snd.items[source - 1] for _ in snd.items])
    import torch.distributed as dist
    box = [snd.item if snd.rank + 1 == source else None]
    dist.broadcast_object_list(box, src=_global_rank(snd.group, source - 1), group=snd.group)
    return TorchDistArray(box[0], snd.group)


def _global_rank(group, group_rank):
    import torch.distributed as dist
    if group is None:
        return group_rank
    return dist.get_global_rank(group, group_rank)


def reduction(op, a, init=None, destination=MAIN):
    """src/primitives.jl:681-693: fold over parts in part order 1..P."""
    g = gather(a, destination="all")

    def fold(vals):
        acc = init
        for v in vals:
            acc = v if acc is None else op(acc, v)
        return acc

    r = pmap(fold, g)
    if destination == "all":
        return r
    return map_main(lambda x: x, r, main=destination)


def preduce(op, a, init=None):
    """reduce(op,a;init) -> plain value on every part."""
    return getany(reduction(op, a, init=init, destination="all"))


def scan(op, a, type="inclusive", init=None):
    """src/primitives.jl:599-611."""
    g = gather(a, destination="all")
    ranks = linear_indices(a)

    def f(r, vals):
        acc = init
        out = []
        for v in vals:
            if type == "exclusive":
                out.append(acc)
            acc = v if acc is None else op(acc, v)
            if type == "inclusive":
                out.append(acc)
        return out[r - 1]

    return pmap(f, ranks, g)


@dataclass
class ExchangeGraph:
    """src/primitives.jl:728-741: snd[i] / rcv[i] neighbour id lists (1-based part ids)."""
    snd: object
    rcv: object

    def reverse(self):
        return ExchangeGraph(self.rcv, self.snd)


def find_rcv_ids_gather_scatter(snd_ids):
    """src/primitives.jl:826-859: gather the graph on MAIN, transpose it, scatter the receivers back.
    The receivers of a part come out in ascending order."""
    snd_main = gather(pmap(lambda s: [int(x) for x in s], snd_ids))

    def transpose(all_snd):
        if not all_snd:
            return []
        npart = len(all_snd)
        rcv = [[] for _ in range(npart)]
        for p, lst in enumerate(all_snd, start=1):
            for j in sorted(set(lst)):
                rcv[j - 1].append(p)
        return [np.array(r, dtype=np.int32) for r in rcv]

    return scatter(pmap(transpose, snd_main))


def exchange_graph(snd, rcv=None, symmetric=False):
    """ExchangeGraph(snd; rcv, symmetric, find_rcv_ids) (src/primitives.jl:768-783)."""
    if rcv is not None:
        return ExchangeGraph(snd, rcv)
    if symmetric:
        return ExchangeGraph(snd, snd)
    return ExchangeGraph(snd, find_rcv_ids_gather_scatter(snd))


def is_consistent(graph: ExchangeGraph) -> bool:
    """src/primitives.jl:861-874."""
    snd = getany(gather(pmap(lambda s: [int(x) for x in s], graph.snd), destination="all"))
    rcv = getany(gather(pmap(lambda s: [int(x) for x in s], graph.rcv), destination="all"))
    for part in range(1, len(rcv) + 1):
        for i in rcv[part - 1]:
            if sum(1 for k in snd[i - 1] if k == part) != 1:
                return False
        for i in snd[part - 1]:
            if sum(1 for k in rcv[i - 1] if k == part) != 1:
                return False
    return True


# one part per process: 0 = trust the graph, 1 (default) = every rank checks that its receive list is exactly the set of
# ranks that list it as a destination (one all-gather of a P-byte mask + one all-reduce of the verdict: an edge only one
# end knows would otherwise leave that end waiting forever in a blocking receive), 2 = the reference's full is_consistent
CHECK_EXCHANGE_GRAPHS = int(os.environ.get("PA_CHECK_EXCHANGE_GRAPHS", "1"))


_GROUP_TOKENS = None
_GROUP_SERIAL = [0]


def _group_token(group):
    """A value that names `group` for as long as it lives and is never handed to another group (None: the default group)."""
    global _GROUP_TOKENS
    if group is None:
        return ("default-group",)
    if _GROUP_TOKENS is None:
        import weakref
        _GROUP_TOKENS = weakref.WeakKeyDictionary()
    try:
        if group not in _GROUP_TOKENS:
            _GROUP_SERIAL[0] += 1
            _GROUP_TOKENS[group] = ("group", _GROUP_SERIAL[0])
        return _GROUP_TOKENS[group]
    except TypeError:                                        # (a group type that cannot be weakly referenced: check every time)
        return object()


def _edges_match(dist, group, me, world, snd_ids, rcv_ids, kw):
    """True on every rank iff, on every rank, rcv_ids == {i : rank i sends to me} (and no neighbour is listed twice)."""
    import torch
    mask = torch.zeros(world, dtype=torch.uint8, **kw)
    for q in snd_ids:
        if 1 <= q <= world:
            mask[q - 1] = 1
    rows = [torch.zeros(world, dtype=torch.uint8, **kw) for _ in range(world)]
    dist.all_gather(rows, mask, group=group)
    senders = sorted(i + 1 for i in range(world) if int(rows[i][me - 1]))
    ok = (senders == sorted(rcv_ids) and len(set(snd_ids)) == len(snd_ids) and len(set(rcv_ids)) == len(rcv_ids)
          and all(1 <= q <= world for q in snd_ids))
    flag = torch.tensor([1 if ok else 0], **kw)
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(flag.item()), senders


def exchange(snd, graph: ExchangeGraph):
    """exchange(snd,graph)|>fetch for HOST data (src/primitives.jl:921-935,1005-1042).

    snd[i][j] (a scalar, or a sequence -> jagged exchange) goes to part graph.snd[i][j];
    the result rcv[i][j] is what part graph.rcv[i][j] sent to i.  Used at set-up (global ids of the
    ghosts, slice lengths, the I/J/V triplets of a first-time matrix assembly); the per-iteration exchange of
    values is the device path.

    One part per process (TorchDistArray): point-to-point, one message per directed edge of the graph, as the MPI
    back-end does (Irecv!/Isend per neighbour, src/mpi_array.jl:575-614) -- a rank's traffic is what ITS neighbours
    send, independent of the number of parts.  The graph's consistency (src/primitives.jl:861-874) is checked there in its
    cheap form by default -- every rank's receive list must be exactly the ranks that send to it, or the call asserts on
    EVERY rank instead of leaving one end in a blocking receive (PA_CHECK_EXCHANGE_GRAPHS=0 trusts the graph, =2 runs the
    reference's full check); the in-process DebugArray always runs the full check.  The FIRST exchange over a graph is
    therefore a collective over the whole group (every rank must enter it); the verdict is remembered on the graph.
    """
    if isinstance(snd, TorchDistArray):
        import torch.distributed as dist
        if CHECK_EXCHANGE_GRAPHS >= 2:
            assert is_consistent(graph)
        group = snd.group
        me = dist.get_rank(group) + 1
        snd_ids, rcv_ids = [int(q) for q in graph.snd.item], [int(q) for q in graph.rcv.item]
        data = list(snd.item)
        assert len(data) == len(snd_ids), "one item per send neighbour"
        out = [None] * len(rcv_ids)
        world = dist.get_world_size(group)
        # host data travels over the host back-end when the group has one ("cpu:gloo,cuda:nccl"): set-up does not depend
        # on the GPU transport, and no device memory is touched for pickled index lists
        try:
            import torch
            dev = torch.device("cpu") if "gloo" in str(dist.get_backend_config(group)) else None
        except Exception:                                    # noqa: BLE001
            dev = None
        kw = {} if dev is None else {"device": dev}
        # The check is a collective over the WHOLE group (an all-gather + an all-reduce): it runs once per graph object --
        # the first exchange over a graph must be entered by every rank of the group, later ones only involve neighbours.
        # The verdict is remembered under a TOKEN of the group, not its id() (an id can come back with another group behind it,
        # ADVICE r04): group objects get a serial number at first sight, kept in a WeakKeyDictionary.  Whether the collective check
        # runs is decided by every rank from its OWN copy of the graph: a graph object must therefore be reused (or rebuilt) by all
        # ranks of the group alike -- as every graph of this package is (they are made and used by collective calls).
        token = _group_token(group)
        checked = getattr(graph, "_edges_checked", None)
        if CHECK_EXCHANGE_GRAPHS == 1 and checked != token:
            ok, senders = _edges_match(dist, group, me, world, snd_ids, rcv_ids, kw)
            assert ok, (f"inconsistent ExchangeGraph (src/primitives.jl:861-874) seen from part {me}: it expects messages from "
                        f"{sorted(rcv_ids)}, the parts that send to it are {senders}")
            try:
                graph._edges_checked = token
            except AttributeError:                           # (a graph type without a __dict__: check every time)
                pass
        snd_at = {q: j for j, q in enumerate(snd_ids)}                 # partner -> position (a dict, not a scan per round)
        rcv_at = {q: j for j, q in enumerate(rcv_ids)}

        def grank(part):
            return part - 1 if group is None else dist.get_global_rank(group, part - 1)
        # Deadlock-free pairing without non-blocking object sends: the directed edges are served in rounds of the
        # classical pairwise schedule -- in round k rank r talks to partner (k - r) mod P; of a pair, the lower rank sends
        # first.  Every edge (i -> j) is met in exactly one round by both of its ends.
        # Only the rounds of REAL neighbours are walked (in round order, which both ends of an edge compute alike): a rank
        # with 6 neighbours among 512 parts does 6 rounds, not 512.
        # That short walk relies on both ends of an edge knowing it; an unchecked graph (PA_CHECK_EXCHANGE_GRAPHS=0) walks
        # all P rounds, where a one-sided edge cannot shift the partner's later rounds.
        if CHECK_EXCHANGE_GRAPHS == 0:
            partners = sorted(range(1, world + 1), key=lambda q: ((q - 1) + (me - 1)) % world)
        else:
            partners = sorted(set(snd_ids) | set(rcv_ids), key=lambda q: ((q - 1) + (me - 1)) % world)
        for partner in partners:
            if partner == me:
                if me in snd_at and me in rcv_at:                     # a part that lists itself (never on this path; kept exact)
                    out[rcv_at[me]] = data[snd_at[me]]
                continue
            do_send, do_recv = partner in snd_at, partner in rcv_at
            first_send = me < partner
            for phase in (0, 1):
                if (phase == 0) == first_send:
                    if do_send:
                        dist.send_object_list([data[snd_at[partner]]], dst=grank(partner), group=group, **kw)
                else:
                    if do_recv:
                        box = [None]
                        dist.recv_object_list(box, src=grank(partner), group=group, **kw)
                        out[rcv_at[partner]] = box[0]
        return TorchDistArray(out, group)
    assert is_consistent(graph)
    packed = pmap(lambda ids, data: ([int(x) for x in ids], list(data)), graph.snd, snd)
    everything = gather(packed, destination="all")
    ranks = linear_indices(snd)

    def pick(r, rcv_ids, allv):
        out = []
        for s in rcv_ids:
            ids, data = allv[int(s) - 1]
            j = ids.index(r)
            out.append(data[j])
        return out

    return pmap(pick, ranks, graph.rcv, everything)
