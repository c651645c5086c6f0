"""PVector on the device: local values in HBM, ghost exchange through HIP kernels.

Mirrors /root/reference/src/p_vector.jl for the hot path:
    PVector (:324), p_vector_cache_impl / VectorAssemblyCache (:418-468), assemble_impl! (:587-612),
    assemble! (:695-708), consistent! (:747-755), dot (:1189), norm (:1201), broadcast axpy (:1216-1277).
The local vector type is `DeviceVector` (a pa_vec of libpa_hip); this plays the role of the
`V` in `PVector{V}` that the reference dispatches on.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib as L
from .primitives import DebugArray, TorchDistArray, getany, local_items, pmap, preduce
from .p_range import PRange, assembly_local_indices, assembly_neighbors

F64 = np.float64


# ----------------------------------------------------------------------------------------------
# device context (one per process; all parts of a DebugArray share it)
# ----------------------------------------------------------------------------------------------
class Context:
    def __init__(self, device=None):
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
            n = C.c_int(0)
            L.call("pa_device_count", C.byref(n))
            if n.value == 0:
                raise L.PAError("no HIP device visible: the device path needs a GPU (there is no CPU fallback)")
            device %= n.value
        h = C.c_void_p()
        L.call("pa_ctx_create", device, C.byref(h))
        self.h = h
        self.device = device
        self.comm = None        # RcclComm, set by init_comm()

    def sync(self):
        L.call("pa_ctx_sync", self.h)
        if _PART_CTX:                       # (one context per part: "the device is idle" means all of them)
            for c in all_contexts():
                if c is not self:
                    L.call("pa_ctx_sync", c.h)

    def fused_launches(self):
        """(products run as one launch so far, of those with the exchange inside the launch) -- pa_ctx_fused_launches."""
        a, b = C.c_int64(), C.c_int64()
        L.call("pa_ctx_fused_launches", self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    def reload_env(self):
        """Read the PA_* switches of the product path from the environment again (pa_ctx_reload_env; they are read once, at creation)."""
        L.call("pa_ctx_reload_env", self.h)
        if _PART_CTX:                       # (one context per part: the switches are per context -- reload every one of them)
            for c in all_contexts():
                if c is not self:
                    L.call("pa_ctx_reload_env", c.h)

    def info(self):
        cus, xcds, hbm = C.c_int(), C.c_int(), C.c_size_t()
        name = C.create_string_buffer(128)
        L.call("pa_ctx_device_info", self.h, C.byref(cus), C.byref(xcds), C.byref(hbm), name, 128)
        return dict(cus=cus.value, xcds=xcds.value, hbm_bytes=hbm.value, name=name.value.decode())

    def event(self):
        return Event(self)

    def stream_priorities(self):
        """(compute, comm, least, greatest) stream priorities: the comm stream holds the device's greatest
        (pa_ctx_stream_priority)."""
        a, b, lo, hi = C.c_int(), C.c_int(), C.c_int(), C.c_int()
        L.call("pa_ctx_stream_priority", self.h, L.STREAM_COMPUTE, C.byref(a), C.byref(lo), C.byref(hi))
        L.call("pa_ctx_stream_priority", self.h, L.STREAM_COMM, C.byref(b), None, None)
        return dict(compute=a.value, comm=b.value, least=lo.value, greatest=hi.value)

    def arena_hint(self, vector_classes=2):
        """Announce a solver's worth of vectors (pa_ctx_arena_hint): they alternate between two memory classes of their own."""
        L.call("pa_ctx_arena_hint", self.h, int(vector_classes))

    def pci_bus_id(self):
        """PCI address of this context's GPU (pa_ctx_pci_bus_id): /sys/bus/pci/devices/<id>/ holds its clocks and power."""
        buf = C.create_string_buffer(64)
        L.call("pa_ctx_pci_bus_id", self.h, buf, 64)
        return buf.value.decode()

    def telemetry(self):
        """Clocks, power and partition modes of this context's GPU, read from its sysfs directory (what rocm-smi / amd-smi
        print, without spawning them): a dict of whatever the box exposes, {} when nothing is readable."""
        import glob
        try:
            base = "/sys/bus/pci/devices/" + self.pci_bus_id()
        except L.PAError:
            return {}
        out = {}

        def rd(path):
            try:
                return open(path).read().strip()
            except OSError:
                return None

        def cur_level(name):
            txt = rd(f"{base}/{name}")
            if txt is None:
                return None
            for line in txt.splitlines():
                if line.strip().endswith("*"):
                    return line.split(":", 1)[1].strip().rstrip("*").strip()
            return None
        for key, name in (("sclk", "pp_dpm_sclk"), ("mclk", "pp_dpm_mclk"), ("fclk", "pp_dpm_fclk"), ("socclk", "pp_dpm_socclk")):
            v = cur_level(name)
            if v is not None:
                out[key] = v
        for key, name in (("compute_partition", "current_compute_partition"), ("memory_partition", "current_memory_partition"),
                          ("perf_level", "power_dpm_force_performance_level")):
            v = rd(f"{base}/{name}")
            if v is not None:
                out[key] = v
        for hw in sorted(glob.glob(base + "/hwmon/hwmon*")):
            for key, name, scale in (("power_w", "power1_input", 1e-6), ("power_w", "power1_average", 1e-6), ("power_cap_w", "power1_cap", 1e-6),
                                     ("sclk_mhz", "freq1_input", 1e-6), ("mclk_mhz", "freq2_input", 1e-6),
                                     ("temp_junction_c", "temp2_input", 1e-3), ("temp_mem_c", "temp3_input", 1e-3)):
                v = rd(f"{hw}/{name}")
                if v is not None and key not in out:
                    try:
                        out[key] = round(int(v) * scale, 1)
                    except ValueError:
                        pass
        return out

    def arena_release(self):
        """Hand the extents of an idle arena back to the driver now (pa_ctx_arena_release; it keeps up to 24 GiB otherwise)."""
        L.call("pa_ctx_arena_release", self.h)

    def arena(self, build=False):
        """The context's HBM extents and their memory-class maps (csrc/pa_arena.hip): GiB held, classes met, GiB per class
        in the held extents, GiB in use, time spent acquiring + classifying, the class of every 512 MiB cell as a string
        ('.' = a boundary cell, '|' between two extents), and what the arena did so far (extents, GiB acquired / released,
        self-checked pairs)."""
        if build:
            L.call("pa_ctx_arena_build", self.h)
        size, used, ncls, ms, mcls = C.c_int64(), C.c_int64(), C.c_int(), C.c_double(), C.c_int()
        per = (C.c_int64 * 3)()
        L.call("pa_ctx_arena_info", self.h, C.byref(size), C.byref(ncls), per, C.byref(used), C.byref(ms), C.byref(mcls))
        n, cell = C.c_int64(), C.c_int64()
        L.call("pa_ctx_arena_map", self.h, C.byref(cell), None, 0, C.byref(n))
        cells = np.zeros(max(n.value, 1), np.int8)
        L.call("pa_ctx_arena_map", self.h, C.byref(cell), L.ptr(cells), n.value, C.byref(n))
        st = [C.c_int64() for _ in range(8)]
        L.call("pa_ctx_arena_stats", self.h, *[C.byref(v) for v in st])
        G = float(1 << 30)
        return dict(gib=round(size.value / G, 1), classes=ncls.value, matrix_class=mcls.value, class_gib=[round(v / G, 1) for v in per],
                    used_gib=round(used.value / G, 2), map_ms=round(ms.value, 1), cell_mib=cell.value >> 20,
                    cells="".join("|" if v == -2 else "." if v < 0 else str(int(v)) for v in cells[:n.value]),
                    extents=st[0].value, acquired_gib=round(st[1].value / G, 1), released_gib=round(st[2].value / G, 1),
                    peak_used_gib=round(st[3].value / G, 2), pairs_checked_ok=st[4].value, pairs_checked_same_class=st[5].value,
                    budget_gib=round(st[6].value / G, 1), plain_vectors_gib=round(st[7].value / G, 2))


class Graph:
    """hipGraph of whatever is queued between `with Graph() as g:` and its end (pa_graph_begin/_end): nothing runs while
    recording; g.launch() replays the recorded kernels and copies on the compute stream with one submission."""

    def __init__(self, ctx=None):
        self.ctx = ctx or context()
        self.h = None

    def __enter__(self):
        if len(all_contexts()) > 1:
            raise L.PAError("hipGraph capture records ONE context's streams: not with one device context per part (PA_CTX_PER_PART)")
        L.call("pa_graph_begin", self.ctx.h)
        return self

    def __exit__(self, et, ev, tb):
        h = C.c_void_p()
        L.call("pa_graph_end", self.ctx.h, C.byref(h))
        self.h = h
        return False

    def launch(self):
        L.call("pa_graph_launch", self.h)

    def __del__(self):
        try:
            if self.h is not None:
                L.lib.pa_graph_destroy(self.h)
        except Exception:
            pass


class Event:
    """HIP event on one of the context's streams (PTimer replacement, src/p_timer.jl)."""

    def __init__(self, ctx):
        self.h = C.c_void_p()
        L.call("pa_event_create", ctx.h, C.byref(self.h))

    def record(self, stream=L.STREAM_COMPUTE):
        L.call("pa_event_record", self.h, stream)
        return self

    def elapsed_ms(self, stop: "Event") -> float:
        ms = C.c_float()
        L.call("pa_event_elapsed_ms", self.h, stop.h, C.byref(ms))
        return ms.value

    def __del__(self):
        try:
            L.lib.pa_event_destroy(self.h)
        except Exception:
            pass


_CTX = None
_PART_CTX = {}             # PA_CTX_PER_PART=1: part index -> its own Context


def contexts_per_part() -> bool:
    """PA_CTX_PER_PART=1: every part of a DebugArray gets a device context of its own (compute + comm stream, arena), on GPU
    (part index mod visible GPUs) -- the single-process multi-GPU mode (DebugArray over several GPUs, src/debug_array.jl:110-117
    with the parts' data on different devices).  With one GPU the contexts share it: the same code paths (one push launch per
    context, cross-context events), which is how they are tested on a 1-GPU box."""
    return os.environ.get("PA_CTX_PER_PART", "0") == "1"


def context() -> Context:
    """The device context of the part a pmap is visiting (PA_CTX_PER_PART=1), else the process's one context."""
    global _CTX
    if contexts_per_part():
        from .primitives import CURRENT_PART
        i = CURRENT_PART[0]
        if i is not None:
            if i not in _PART_CTX:
                n = C.c_int(0)
                L.call("pa_device_count", C.byref(n))
                _PART_CTX[i] = Context(device=i % max(n.value, 1)) if i > 0 or _CTX is None else _CTX
                if i == 0 and _CTX is None:
                    _CTX = _PART_CTX[0]
            return _PART_CTX[i]
    if _CTX is None:
        _CTX = Context(device=0 if contexts_per_part() else None)
        if contexts_per_part():
            _PART_CTX.setdefault(0, _CTX)
    return _CTX


def all_contexts():
    out = [] if _CTX is None else [_CTX]
    return out + [c for c in _PART_CTX.values() if c is not _CTX]


class RcclComm:
    """RCCL communicator, one rank per part (src/mpi_array.jl:42-53 analogue)."""

    def __init__(self, ctx: Context, group=None):
        import torch.distributed as dist
        rank, size = dist.get_rank(group), dist.get_world_size(group)
        idbuf = C.create_string_buffer(L.UNIQUE_ID_BYTES)
        if rank == 0:
            L.call("pa_comm_unique_id", idbuf)
        box = [idbuf.raw if rank == 0 else None]
        src = 0 if group is None else dist.get_global_rank(group, 0)
        dist.broadcast_object_list(box, src=src, group=group)
        self.h = C.c_void_p()
        L.call("pa_comm_create", ctx.h, box[0], rank, size, C.byref(self.h))
        self.rank, self.size, self.ctx = rank, size, ctx

    def allreduce_sum(self, device_ptr, count, stream=L.STREAM_COMPUTE):
        L.call("pa_comm_allreduce_sum", self.h, device_ptr, count, stream)

    def barrier(self):
        L.call("pa_comm_barrier", self.h)

    def info(self):
        """Rank and size as the communicator itself reports them (ncclCommUserRank / ncclCommCount)."""
        r, n = C.c_int(), C.c_int()
        L.call("pa_comm_info", self.h, C.byref(r), C.byref(n))
        return dict(rank=r.value, nranks=n.value)


def init_comm(group=None):
    """Create (once) the RCCL communicator used by the device exchange of TorchDistArray back-ends."""
    ctx = context()
    if ctx.comm is None:
        ctx.comm = RcclComm(ctx, group)
    return ctx.comm


# ----------------------------------------------------------------------------------------------
# local vector type
# ----------------------------------------------------------------------------------------------
class DeviceVector:
    """Local values of one part in HBM (allocate_local_values, src/p_vector.jl:8-14)."""

    def __init__(self, n_own, n_ghost, ctx=None, l2d=None):
        self.ctx = ctx or context()
        self.n_own, self.n_ghost = int(n_own), int(n_ghost)
        # The device layout is always [own | ghost].  `l2d` (LocalIndices.local_to_device) maps a local id to its device
        # position for partitions whose local order is something else (PermutedLocalIndices, src/p_range.jl:1372:
        # uniform_partition with ghost layers); whole-vector upload/download then speak the LOCAL order.
        self.l2d = l2d
        self.h = C.c_void_p()
        L.call("pa_vec_create", self.ctx.h, self.n_own, self.n_ghost, C.byref(self.h))

    def __len__(self):
        return self.n_own + self.n_ghost

    def upload(self, host, offset=0):
        host = np.ascontiguousarray(host, dtype=F64)
        if self.l2d is not None:
            assert offset == 0 and len(host) == len(self), "a permuted local vector is uploaded whole"
            dev = np.empty(len(host), dtype=F64)
            dev[self.l2d] = host
            host = dev
        L.call("pa_vec_upload", self.h, L.ptr(host), offset, len(host))
        return self

    def download(self, offset=0, length=None):
        length = len(self) - offset if length is None else length
        out = np.empty(length, dtype=F64)
        L.call("pa_vec_download", self.h, L.ptr(out), offset, length)
        if self.l2d is not None:
            assert offset == 0 and length == len(self), "a permuted local vector is downloaded whole"
            out = out[self.l2d]
        return out

    def own(self):
        return self._device_order(0, self.n_own)

    def ghost(self):
        return self._device_order(self.n_own, self.n_ghost)

    def _device_order(self, offset, length):
        out = np.empty(length, dtype=F64)
        L.call("pa_vec_download", self.h, L.ptr(out), offset, length)
        return out

    def fill(self, value, segment=L.SEG_LOCAL):
        L.call("pa_vec_fill", self.h, segment, float(value))
        return self

    def data_ptr(self):
        p = C.c_void_p()
        L.call("pa_vec_data", self.h, C.byref(p))
        return p.value

    def memory_class(self):
        """Memory class of the storage inside the context's arena (pa_vec_memory_class; -1: outside)."""
        n = C.c_int()
        L.call("pa_vec_memory_class", self.h, C.byref(n))
        return n.value

    def __del__(self):
        try:
            L.lib.pa_vec_destroy(self.h)
        except Exception:
            pass


def connect_ipc(plans):
    """PA_TRANSPORT=ipc (csrc/pa_push.hip), one part per process: every rank publishes the ipc handles of its plan's receive
    buffers and flag words, maps its neighbours' and from then on packs straight into them.  Collective over the plans' group
    (like the neighbour discovery that precedes every plan); a no-op for other transports and back-ends."""
    if not (isinstance(plans, TorchDistArray) and TRANSPORT == "ipc"):
        return
    import torch.distributed as dist
    plan = plans.item
    n = C.c_int64()
    L.call("pa_plan_ipc_blob_size", plan, C.byref(n))
    buf = C.create_string_buffer(n.value)
    L.call("pa_plan_ipc_blob", plan, buf, n.value)
    group = plans.group
    world = dist.get_world_size(group)
    blobs = [None] * world
    dist.all_gather_object(blobs, buf.raw, group=group)
    keep = [C.create_string_buffer(b, len(b)) for b in blobs]
    ptrs = (C.c_void_p * world)(*[C.cast(k, C.c_void_p).value for k in keep])
    sizes = (C.c_int64 * world)(*[len(b) for b in blobs])
    L.call("pa_plan_ipc_connect", plan, world, ptrs, sizes)
    dist.barrier(group=group)                          # nobody pushes before everybody has mapped


# ----------------------------------------------------------------------------------------------
# VectorAssemblyCache on the device
# ----------------------------------------------------------------------------------------------
class DeviceAssemblyCache:
    """p_vector_cache_impl for DeviceVector (src/p_vector.jl:451-468): the neighbours and index lists of
    every part, uploaded once into a pa_plan that also owns buffer_snd / buffer_rcv in HBM."""

    def __init__(self, index_partition):
        self.index_partition = index_partition
        self.neighbors_snd, self.neighbors_rcv = assembly_neighbors(index_partition)
        self.local_indices_snd, self.local_indices_rcv = assembly_local_indices(
            index_partition, self.neighbors_snd, self.neighbors_rcv)

        def make(ind, ns, nr, ls, lr):
            h = C.c_void_p()
            ns32, nr32 = np.ascontiguousarray(ns, np.int32), np.ascontiguousarray(nr, np.int32)
            l2d = ind.local_to_device

            def dev(lids):           # the cache's 1-based local ids as 1-based positions of the device layout [own | ghost]
                lids = np.ascontiguousarray(lids, np.int32)
                return lids if l2d is None else np.ascontiguousarray(l2d[lids.astype(np.int64) - 1] + 1, np.int32)
            L.call("pa_plan_create", context().h, ind.part, ind.n_local, len(ns32), L.ptr(ns32), L.ptr(ls.ptrs),
                   L.ptr(dev(ls.data)), len(nr32), L.ptr(nr32), L.ptr(lr.ptrs),
                   L.ptr(dev(lr.data)), 1, C.byref(h))
            plan_info[h.value] = dict(
                snd=[(int(q), int(ls.ptrs[k]) - 1, int(ls.ptrs[k + 1]) - 1) for k, q in enumerate(ns32)],
                rcv=[(int(q), int(lr.ptrs[k]) - 1, int(lr.ptrs[k + 1]) - 1) for k, q in enumerate(nr32)])
            return h

        self.plans = pmap(make, index_partition, self.neighbors_snd, self.neighbors_rcv,
                          self.local_indices_snd, self.local_indices_rcv)
        connect_ipc(self.plans)

    def __del__(self):
        try:
            for h in local_items(self.plans):
                L.lib.pa_plan_destroy(h)
        except Exception:
            pass


class Task:
    """What `exchange!`/`consistent!`/`assemble!` return: call wait() exactly once (FakeTask,
    src/primitives.jl:119-141).  wait() queues the unpack on the compute stream; it does not block the host."""

    def __init__(self, fn, result=None):
        self._fn, self._done, self._result = fn, False, result

    def wait(self):
        if not self._done:
            self._fn()
            self._done = True
        return self._result

    fetch = wait


class _DevMem:
    """Zero-copy view of a device buffer for torch (CUDA array interface); used only by the fallback transport."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False), "version": 2}


TRANSPORT = os.environ.get("PA_TRANSPORT", "rccl")   # "rccl": ncclSend/ncclRecv issued by libpa_hip on its comm stream
                                                     # "ipc": the pack kernel pushes into the neighbours' buffers (hipIpc, pa_push.hip)
                                                     # "torch": torch.distributed batch_isend_irecv (same RCCL underneath)


def _transport_torch(plan, mode, group=None):
    """Fallback transport: the same per-neighbour slices moved by torch.distributed's NCCL(=RCCL) p2p ops.
    Host-synchronous (two stream syncs per exchange), so it does not overlap with own*own; it exists so a
    multi-GPU run still completes if the direct RCCL path cannot be initialised."""
    import torch
    import torch.distributed as dist
    snd, rcv = C.c_void_p(), C.c_void_p()
    ns, nr = C.c_int64(), C.c_int64()
    L.call("pa_plan_buffers", plan, mode, C.byref(snd), C.byref(ns), C.byref(rcv), C.byref(nr))
    context().sync()
    ops = []
    info = plan_info[plan.value]
    o, i = (info["snd"], info["rcv"]) if mode == L.ASSEMBLE else (info["rcv"], info["snd"])
    dev = torch.device("cuda", context().device)
    ts = torch.as_tensor(_DevMem(snd.value, max(ns.value, 1)), device=dev) if ns.value else None
    tr = torch.as_tensor(_DevMem(rcv.value, max(nr.value, 1)), device=dev) if nr.value else None
    for nbr, a, e in i:
        if e > a:
            ops.append(dist.P2POp(dist.irecv, tr[a:e], _grank(group, nbr - 1), group))
    for nbr, a, e in o:
        if e > a:
            ops.append(dist.P2POp(dist.isend, ts[a:e], _grank(group, nbr - 1), group))
    if ops:
        for req in dist.batch_isend_irecv(ops):
            req.wait()
        torch.cuda.current_stream(dev).synchronize()


def _transport_host(plan, mode, group=None):
    """Host-staged transport over any torch.distributed backend with CPU tensors (gloo): device -> host -> p2p ->
    device.  Slow by construction; lets several ranks share ONE GPU (RCCL refuses that), which is how the
    one-part-per-process path is exercised end to end on a single-GPU box."""
    import torch
    import torch.distributed as dist
    snd, rcv = C.c_void_p(), C.c_void_p()
    ns, nr = C.c_int64(), C.c_int64()
    L.call("pa_plan_buffers", plan, mode, C.byref(snd), C.byref(ns), C.byref(rcv), C.byref(nr))
    context().sync()
    info = plan_info[plan.value]
    o, i = (info["snd"], info["rcv"]) if mode == L.ASSEMBLE else (info["rcv"], info["snd"])
    dev = torch.device("cuda", context().device)
    hs = torch.as_tensor(_DevMem(snd.value, ns.value), device=dev).cpu() if ns.value else None
    hr = torch.zeros(nr.value, dtype=torch.float64)
    reqs = []
    for nbr, a, e in i:
        if e > a:
            reqs.append(dist.irecv(hr[a:e], _grank(group, nbr - 1), group))
    for nbr, a, e in o:
        if e > a:
            reqs.append(dist.isend(hs[a:e].contiguous(), _grank(group, nbr - 1), group))
    for r in reqs:
        r.wait()
    if nr.value:
        torch.as_tensor(_DevMem(rcv.value, nr.value), device=dev).copy_(hr)
        torch.cuda.current_stream(dev).synchronize()


def _grank(group, r):
    import torch.distributed as dist
    return r if group is None else dist.get_global_rank(group, r)


plan_info = {}   # plan handle -> neighbour slices (1-based part id, start, stop), for the fallback transport


def _transport(plans, mode):
    """exchange!(buffer_rcv,buffer_snd,graph) on the device (src/p_vector.jl:601):
    DebugArray -> device-to-device slice copies (src/debug_array.jl:250);
    TorchDistArray -> RCCL send/recv group on the comm stream (src/mpi_array.jl:575-614)."""
    if isinstance(plans, DebugArray):
        hs = plans.items
        arr = (C.c_void_p * len(hs))(*[h.value for h in hs])
        if _rccl_all():
            L.call("pa_exchange_rccl_all", arr, _all_comms(len(hs)), len(hs), mode)
        else:
            L.call("pa_exchange_local", arr, len(hs), mode)
    else:
        if TRANSPORT == "torch":
            _transport_torch(plans.item, mode, plans.group)
            return
        if TRANSPORT == "host":
            _transport_host(plans.item, mode, plans.group)
            return
        comm = context().comm
        if comm is None:
            raise L.PAError("no RCCL communicator: call init_comm() before exchanging on a TorchDistArray")
        L.call("pa_exchange_rccl", plans.item, comm.h, mode)


def _push():                 # all parts in one process: one launch packs AND delivers (csrc/pa_push.hip); PA_PUSH=0: pack + copies
    return os.environ.get("PA_PUSH", "1") != "0" and not _rccl_all()


def _rccl_all():
    """PA_TRANSPORT_ALL=rccl (with PA_CTX_PER_PART=1: every part of a DebugArray in a context on a GPU of its own): the exchange of all
    parts is ONE group of ncclSend / ncclRecv over communicators made by ncclCommInitAll (pa_comm_create_all / pa_exchange_rccl_all,
    csrc/pa_rccl.cpp) instead of peer copies.  The devices must be distinct -- on a one-GPU box only a single part can run this."""
    return os.environ.get("PA_TRANSPORT_ALL", "") == "rccl"


_ALL_COMMS = {}


def _all_comms(n):
    """the n communicators of the parts' contexts (parts 0..n-1), made once per n"""
    if n not in _ALL_COMMS:
        ctxs = [_PART_CTX[i] if i in _PART_CTX else context() for i in range(n)]
        arr = (C.c_void_p * n)(*[c.h.value for c in ctxs])
        out = (C.c_void_p * n)()
        L.call("pa_comm_create_all", arr, n, out)
        _ALL_COMMS[n] = (out, ctxs)
    return _ALL_COMMS[n][0]


def assemble_impl(mode, vector_partition, cache: DeviceAssemblyCache) -> Task:
    """assemble_impl!(f,vector_partition,cache) (src/p_vector.jl:587-612): pack, start the exchange,
    return a task whose wait() unpacks with f = insert (CONSISTENT) or + (ASSEMBLE)."""
    plans = cache.plans
    pushed = None
    if isinstance(plans, DebugArray) and _push():
        n = len(plans.items)
        pushed = ((C.c_void_p * n)(*[h.value for h in plans.items]), n, (C.c_void_p * n)(*[v.h.value for v in vector_partition.items]))
        L.call("pa_exchange_push_local", *pushed, mode)
    elif isinstance(plans, TorchDistArray) and TRANSPORT == "ipc":
        L.call("pa_exchange_push_ipc", plans.item, vector_partition.item.h, mode)
    else:
        pmap(lambda v, p: L.call("pa_exchange_pack", p, v.h, mode), vector_partition, plans)
        _transport(plans, mode)

    def finish():
        if pushed is not None:                         # all parts in one call (consistent!: one unpack launch)
            L.call("pa_exchange_finish_all", *pushed, mode)
        else:
            pmap(lambda v, p: L.call("pa_exchange_finish", p, v.h, mode), vector_partition, cache.plans)

    return Task(finish)


class PVector:
    """PVector(vector_partition,index_partition[,cache]) (src/p_vector.jl:324-345)."""

    def __init__(self, vector_partition, index_partition, cache=None):
        self.vector_partition = vector_partition
        self.index_partition = index_partition
        self._cache = cache

    @property
    def cache(self):
        if self._cache is None:
            self._cache = DeviceAssemblyCache(self.index_partition)     # p_vector_cache (:414)
        return self._cache

    @property
    def axes(self):
        return (PRange(self.index_partition),)

    def local_values(self):
        """Host copies of the local values (own and ghost) of every part."""
        return pmap(lambda v: v.download(), self.vector_partition)

    def own_values(self):
        return pmap(lambda v, i: v.download()[i.own_to_local - 1], self.vector_partition, self.index_partition)

    def ghost_values(self):
        return pmap(lambda v, i: v.download()[i.ghost_to_local - 1], self.vector_partition, self.index_partition)

    def collect(self):
        """collect(v): the global vector on the host (every process gets it)."""
        from .primitives import gather
        pieces = gather(pmap(lambda v, i: (i.own_to_global, v.download()[i.own_to_local - 1]),
                             self.vector_partition, self.index_partition), destination="all")
        out = np.zeros(getany(pmap(lambda i: i.n_global, self.index_partition)), dtype=F64)
        for g, vals in getany(pieces):
            out[g - 1] = vals
        return out


def partition(a):
    return a.vector_partition if isinstance(a, PVector) else a.partition


def allocate_local_values(ind) -> DeviceVector:
    """allocate_local_values(V, indices) (src/p_vector.jl:8-14) for V = DeviceVector: n_own + n_ghost zeroed doubles in
    HBM, device layout [own | ghost] whatever the local order of `ind` is -- dot / norm / the own-value broadcasts then
    reduce over own values only (src/p_vector.jl:1189-1206) on every kind of partition."""
    return DeviceVector(ind.n_own, ind.n_ghost, l2d=ind.local_to_device)


def pvector_from_function(f, index_partition, cache=None) -> PVector:
    """pvector(f,index_partition): f(indices) gives the host local values to upload (src/p_vector.jl:815)."""

    def make(ind):
        v = allocate_local_values(ind)
        vals = np.ascontiguousarray(f(ind), dtype=F64)
        assert len(vals) == ind.n_local
        v.upload(vals)
        return v

    return PVector(pmap(make, index_partition), index_partition, cache)


def pfill(value, index_partition) -> PVector:
    """pfill(v,index_partition) (src/p_vector.jl:1047): filled on the device (no host array, no upload)."""

    def make(ind):
        v = allocate_local_values(ind)
        if value != 0.0:                     # (pa_vec_create hands out zeroed storage)
            v.fill(float(value))
        return v

    return PVector(pmap(make, index_partition), index_partition, None)


def pzeros(index_partition) -> PVector:
    return pfill(0.0, index_partition)


def pones(index_partition) -> PVector:
    return pfill(1.0, index_partition)


def similar(a: PVector, index_partition=None) -> PVector:
    ip = a.index_partition if index_partition is None else index_partition
    return pzeros(ip)


def pvector(I, V, index_partition) -> PVector:
    """pvector(I,V,rows)|>fetch for own entries (dense_vector!, src/p_vector.jl:866-875): a[lid] += v."""

    def vals(ind, gi, v):
        out = np.zeros(ind.n_local, F64)
        lid = ind.global_to_local(gi)
        keep = lid >= 1
        np.add.at(out, lid[keep] - 1, np.asarray(v, F64)[keep])
        return out

    host = pmap(vals, index_partition, I, V)
    parts = pmap(lambda ind, h: allocate_local_values(ind).upload(h), index_partition, host)
    return PVector(parts, index_partition)


def on_partition(v: PVector, index_partition) -> PVector:
    """A vector on `index_partition` with v's own values (ghosts 0): `w .= v` between partitions whose own indices
    match (src/p_vector.jl:1331-1345); used to move a right-hand side from the row to the column partition."""
    w = pzeros(index_partition)
    pmap(lambda d, s: L.call("pa_vec_copy", d.h, s.h, L.SEG_OWN), w.vector_partition, v.vector_partition)
    return w


class VectorReassemblyCache:
    """cache of pvector(...;reuse=true) (src/p_vector.jl:887-987): K = local ids of the entries on the sub-assembled
    partition (as a deterministic device scatter), the sub-assembled vector A and the work copy of assemble(v,rows)."""

    def __init__(self, A_sa: PVector, work: PVector, scatters, n_entries):
        self.A_sa, self.work, self.scatters, self.n_entries = A_sa, work, scatters, n_entries

    def __del__(self):
        try:
            for s in local_items(self.scatters):
                L.lib.pa_scatter_destroy(s)
        except Exception:
            pass


def pvector_disassembled(I, V, rows, reuse=False, assemble=True):
    """pvector(I,V,rows)|>fetch with the DEFAULT flags (src/p_vector.jl:887-925,966-987): the entries of a part may
    touch rows it does not own (cell-wise FEM loops).  find_owner -> union_ghost -> dense_vector on the sub-assembled
    partition (a[lid] += v in entry order, :853-863) -> assemble(A,rows) (:1331-1345: copy, assemble!, own values).
    assemble=False returns the sub-assembled vector itself.  reuse=True also returns the cache of pvector_."""
    from .p_range import find_owner, union_ghost
    rows_sa = pmap(union_ghost, rows, I, find_owner(rows, I))
    K = pmap(lambda r, gi: r.global_to_local(gi).astype(np.int32), rows_sa, I)

    def dense(ind, k, v):
        out = np.zeros(ind.n_local, F64)
        keep = k >= 1
        np.add.at(out, k[keep].astype(np.int64) - 1, np.asarray(v, F64)[keep])     # sequential, entry order
        return out
    A_sa = pvector_from_function_values(pmap(dense, rows_sa, K, V), rows_sa)
    if not assemble:
        return (A_sa, None) if reuse else A_sa
    work = similar(A_sa)
    copy_(work, A_sa)
    assemble_(work).wait()
    w = on_partition(work, rows)
    if not reuse:
        return w

    def mk(ind, k):
        s = C.c_void_p()
        L.call("pa_scatter_create", context().h, ind.n_local, len(k), L.ptr(np.ascontiguousarray(k, np.int32)), 1, C.byref(s))
        return s
    return w, VectorReassemblyCache(A_sa, work, pmap(mk, rows_sa, K), pmap(len, K))


def pvector_(b: PVector, V, cache: VectorReassemblyCache) -> PVector:
    """pvector!(b,V,cache) (src/p_vector.jl:990-1008): dense_vector!(A,K,V) (:865-873: fill 0, then a[k] += v in entry
    order -- here one deterministic scatter-add on the device), assemble!(B,A,cacheB), own values into b."""
    def scatter(s, a, v, n):
        v = np.ascontiguousarray(v, F64)
        assert len(v) == n, "V must have the length it had when the cache was built"
        src = DeviceVector(n, 0).upload(v)
        L.call("pa_scatter_add", s, a.h, src.h, 1)
    pmap(scatter, cache.scatters, cache.A_sa.vector_partition, V, cache.n_entries)
    copy_(cache.work, cache.A_sa)
    assemble_(cache.work).wait()
    pmap(lambda d, s: L.call("pa_vec_copy", d.h, s.h, L.SEG_OWN), b.vector_partition, cache.work.vector_partition)
    return b


def pvector_from_function_values(host_values, index_partition) -> PVector:
    """PVector(values,index_partition) from host arrays in local order."""
    it = pmap(lambda h: h, host_values)
    parts = pmap(lambda ind, h: allocate_local_values(ind).upload(h), index_partition, it)
    return PVector(parts, index_partition)


def consistent_(a: PVector) -> Task:
    """consistent!(a) (src/p_vector.jl:747-755): ghost <- owner.  Returns a task; wait() it."""
    t = assemble_impl(L.CONSISTENT, a.vector_partition, a.cache)
    return Task(t.wait, a)


def assemble_(a: PVector) -> Task:
    """assemble!(a) (src/p_vector.jl:695-708): owner += ghost copies (ascending p), then ghosts := 0."""
    t = assemble_impl(L.ASSEMBLE, a.vector_partition, a.cache)
    return Task(t.wait, a)


def exchange32_(mode, vector_partition, cache: DeviceAssemblyCache) -> Task:
    """assemble_impl!(f,vector_partition,cache) (src/p_vector.jl:587-612) for a PVector{Vector{Float32}}: vector_partition holds one
    DeviceVector32 per part ([own | ghost] like DeviceVector), cache is the VectorAssemblyCache of the index partition (the one a
    Float64 PVector on it uses).  Pack, exchange (device-to-device slice copies for a DebugArray, an RCCL group of ncclFloat
    send / recv for a TorchDistArray), and a task whose wait() unpacks: insert (CONSISTENT) or + in ascending p, then ghosts := 0."""
    plans = cache.plans
    if not isinstance(plans, DebugArray) and TRANSPORT in ("torch", "host"):
        raise L.PAError("Float32 payloads travel over the device-to-device copies of a DebugArray or over RCCL, not over the "
                        f"'{TRANSPORT}' staging transport")
    pmap(lambda v, p: L.call("pa_exchange_pack32", p, v.h, mode), vector_partition, plans)
    _transport(plans, mode)
    return Task(lambda: pmap(lambda v, p: L.call("pa_exchange_finish32", p, v.h, mode), vector_partition, plans))


_DTYPES = {"f64": (0, 8), "f32": (1, 4), "i32": (2, 4), "i64": (3, 8)}


def exchange_raw_(mode, vector_partition, cache: DeviceAssemblyCache, dtype: str) -> Task:
    """assemble_impl! (src/p_vector.jl:587-612) on local values of another element type held in the device arrays of `vector_partition`
    (DeviceVector: 8-byte values, DeviceVector32: 4-byte values; integers are uploaded as the bits of the float type of their width,
    `a.view(np.float64)` / `a.view(np.float32)`): dtype in "f64", "f32", "i32", "i64".  consistent!: the bits travel; assemble!: + in
    ascending p in the dtype's arithmetic, then ghosts := 0 (pa_exchange_pack_raw / pa_exchange_finish_raw)."""
    code, width = _DTYPES[dtype]
    plans = cache.plans
    if not isinstance(plans, DebugArray) and TRANSPORT in ("torch", "host"):
        raise L.PAError(f"raw payloads travel over the device-to-device copies of a DebugArray or over RCCL, not over the '{TRANSPORT}' staging transport")
    pmap(lambda v, p: L.call("pa_exchange_pack_raw", p, C.c_void_p(v.data_ptr()), v.n_own + v.n_ghost, code, mode), vector_partition, plans)
    _transport(plans, mode)
    return Task(lambda: pmap(lambda v, p: L.call("pa_exchange_finish_raw", p, C.c_void_p(v.data_ptr()), v.n_own, v.n_own + v.n_ghost, code, mode),
                             vector_partition, plans))


def consistent32_(vector_partition, cache: DeviceAssemblyCache) -> Task:
    """consistent!(a) (src/p_vector.jl:747-755) on Float32 local values: ghost <- owner."""
    return exchange32_(L.CONSISTENT, vector_partition, cache)


def assemble32_(vector_partition, cache: DeviceAssemblyCache) -> Task:
    """assemble!(a) (src/p_vector.jl:695-708) on Float32 local values: owner += ghost copies (ascending p, Float32 sums), ghosts := 0."""
    return exchange32_(L.ASSEMBLE, vector_partition, cache)


def _part_sum(parts_values, scalar_on_device=False):
    return preduce(lambda x, y: x + y, parts_values, init=0.0)


def dot(a: PVector, b: PVector) -> float:
    """dot(a,b) (src/p_vector.jl:1189-1192): per-part own-value dot on the device, then sum over parts
    (reduction(+,...;destination=:all), src/mpi_array.jl:494): one process per part with an RCCL communicator ->
    ncclAllReduce of the device scalar on the compute stream; otherwise a host sum in part order."""
    vp = a.vector_partition
    if isinstance(vp, TorchDistArray) and TRANSPORT == "rccl" and context().comm is not None:
        ctx = context()
        L.call("pa_vec_dot", vp.item.h, b.vector_partition.item.h, None)
        ptr = C.c_void_p()
        L.call("pa_vec_dot_result", ctx.h, C.byref(ptr))
        ctx.comm.allreduce_sum(ptr, 1, L.STREAM_COMPUTE)
        out = C.c_double()
        L.call("pa_ctx_read_scalar", ctx.h, C.byref(out))
        return out.value

    def local(x, y):
        out = C.c_double()
        L.call("pa_vec_dot", x.h, y.h, C.byref(out))
        return out.value

    return _part_sum(pmap(local, a.vector_partition, b.vector_partition))


def norm(a: PVector) -> float:
    """norm(a,2) (src/p_vector.jl:1201-1206)."""
    return dot(a, a) ** 0.5


def axpby_(y: PVector, alpha, x: PVector, beta) -> PVector:
    """y .= alpha .* x .+ beta .* y on own values (the broadcasts of a CG loop, src/p_vector.jl:1216-1277)."""
    pmap(lambda yv, xv: L.call("pa_vec_axpby", yv.h, float(alpha), xv.h, float(beta), L.SEG_OWN),
         y.vector_partition, x.vector_partition)
    return y


# ---- solver scalars that stay on the device (include/pa_hip.h "slots") ---------------------------------------------
def slots_supported(a: PVector) -> bool:
    """Device-resident scalars need every part's context reachable without a host round trip: all parts in this
    process (DebugArray), or one part per process with an RCCL communicator (or a single process)."""
    vp = a.vector_partition
    if isinstance(vp, DebugArray):
        return len({id(v.ctx) for v in vp.items}) == 1          # (one context per part: its scalars live apart -> host sums)
    if isinstance(vp, TorchDistArray):
        import torch.distributed as dist
        return dist.get_world_size(vp.group) == 1 or (TRANSPORT == "rccl" and context().comm is not None)
    return False


def _slot_allreduce(vp, slot):
    import torch.distributed as dist
    if dist.get_world_size(vp.group) > 1:
        ctx = context()
        ptr = C.c_void_p()
        L.call("pa_ctx_slot_ptr", ctx.h, slot, C.byref(ptr))
        ctx.comm.allreduce_sum(ptr, 1, L.STREAM_COMPUTE)


def dot_slot(a: PVector, b: PVector, slot: int) -> None:
    """slot <- dot(a,b) without a host read-back: per-part dots accumulate in part order (DebugArray) or are
    all-reduced by RCCL on the compute stream (one part per process)."""
    vp, vq = a.vector_partition, b.vector_partition
    if isinstance(vp, TorchDistArray):
        L.call("pa_vec_dot_slot", vp.item.h, vq.item.h, slot, 0)
        _slot_allreduce(vp, slot)
    else:
        for k, (x, y) in enumerate(zip(vp.items, vq.items)):
            L.call("pa_vec_dot_slot", x.h, y.h, slot, int(k > 0))


def axpby_slot_(y: PVector, ca, a_num, a_den, x: PVector, cb, b_num, b_den) -> PVector:
    """y .= (ca*s[a_num]/s[a_den]) .* x .+ (cb*s[b_num]/s[b_den]) .* y with the scalars read on the device."""
    pmap(lambda yv, xv: L.call("pa_vec_axpby_slot", yv.h, float(ca), a_num, a_den, xv.h, float(cb), b_num, b_den,
                               L.SEG_OWN), y.vector_partition, x.vector_partition)
    return y


def cg_update_(x: PVector, r: PVector, u: PVector, c: PVector, num: int, den: int, rr_slot: int) -> None:
    """x .+= alpha .* u; r .-= alpha .* c; slot[rr_slot] = dot(r,r) with alpha = s[num]/s[den] (ref_cg.jl:64-67)."""
    vx = x.vector_partition
    if isinstance(vx, TorchDistArray):
        L.call("pa_cg_update", vx.item.h, r.vector_partition.item.h, u.vector_partition.item.h,
               c.vector_partition.item.h, num, den, rr_slot, 0)
        _slot_allreduce(vx, rr_slot)
    else:
        for k, (a, b, d, e) in enumerate(zip(vx.items, r.vector_partition.items, u.vector_partition.items,
                                             c.vector_partition.items)):
            L.call("pa_cg_update", a.h, b.h, d.h, e.h, num, den, rr_slot, int(k > 0))


def cg_r_update_(r: PVector, c: PVector, num: int, den: int, rr_slot: int) -> None:
    """r .-= alpha .* c; slot[rr_slot] = dot(r,r), alpha = s[num]/s[den] (ref_cg.jl:65-67; pa_cg_r_update)."""
    vr = r.vector_partition
    if isinstance(vr, TorchDistArray):
        L.call("pa_cg_r_update", vr.item.h, c.vector_partition.item.h, num, den, rr_slot, 0)
        _slot_allreduce(vr, rr_slot)
    else:
        for k, (a, b) in enumerate(zip(vr.items, c.vector_partition.items)):
            L.call("pa_cg_r_update", a.h, b.h, num, den, rr_slot, int(k > 0))


def cg_xu_update_(x: PVector, u: PVector, z: PVector, a_num: int, a_den: int, b_num: int, b_den: int) -> None:
    """x .+= (s[a_num]/s[a_den]) .* u, then u .= z .+ (s[b_num]/s[b_den]) .* u, one pass (pa_cg_xu_update)."""
    pmap(lambda xv, uv, zv: L.call("pa_cg_xu_update", xv.h, uv.h, zv.h, a_num, a_den, b_num, b_den),
         x.vector_partition, u.vector_partition, z.vector_partition)


def write_slot(slot: int, value: float) -> None:
    L.call("pa_ctx_write_slot", context().h, slot, float(value))


def read_slots(first: int, n: int = 1):
    out = (C.c_double * n)()
    L.call("pa_ctx_read_slots", context().h, first, n, out)
    return list(out)


def copy_(dst: PVector, src: PVector) -> PVector:
    pmap(lambda d, s: L.call("pa_vec_copy", d.h, s.h, L.SEG_LOCAL), dst.vector_partition, src.vector_partition)
    return dst
