"""VERDICT r03 #4a: what do RCCL's send/recv kernels cost a product kernel that wants all 32 waves per CU?  One rank, one GPU:
own x own of the 27-point operator (256^3 and 128^3) alone, and with -- per product, queued on the high-priority comm stream
just before it, as pa_mul5 does -- a 1-rank RCCL group of self-addressed ncclSend / ncclRecv pairs carrying BASELINE config
4's message set (3 x 65536 + 3 x 256 + 1 doubles: the 7 neighbours of a corner part of (2,2,2) x 256^3), pack kernel
included.  Third column: the same messages through the push transport's single launch (pa_exchange_push_local on a
self-addressed plan; csrc/pa_push.hip) -- what PA_TRANSPORT=ipc puts beside own x own instead.
    python tools/probe/rccl_interference.py [reps]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package

pa = load_package()
import pa_amd._lib as L

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
ctx = pa.context()
idbuf = C.create_string_buffer(L.UNIQUE_ID_BYTES)
L.call("pa_comm_unique_id", idbuf)
comm = C.c_void_p()
L.call("pa_comm_create", ctx.h, idbuf.raw, 0, 1, C.byref(comm))
out = {}
for n in (256, 128):
    A, _ = pa.build_p_matrix(pa.DebugArray([1]), n, n, n, n, n, n, 1, 1, 1)
    blk = pa.local_items(A.matrix_partition)[0].own_own
    rows = blk.m
    face, edge = n * n, n
    lens = [face] * 3 + [edge] * 3 + [1]
    g = sum(lens)
    # a self-addressed plan: the part "ghosts" g of its own values, 7 slices to / from itself
    nbr = np.ones(7, np.int32)
    ptrs = np.concatenate(([1], 1 + np.cumsum(lens))).astype(np.int32)
    own_ids = (1 + (np.arange(g, dtype=np.int64) * 7919) % rows).astype(np.int32)
    ghost_ids = (rows + 1 + np.arange(g)).astype(np.int32)
    plan = C.c_void_p()
    L.call("pa_plan_create", ctx.h, 1, rows + g, 7, L.ptr(nbr), L.ptr(ptrs), L.ptr(ghost_ids), 7, L.ptr(nbr), L.ptr(ptrs), L.ptr(own_ids), 1, C.byref(plan))
    x = pa.DeviceVector(rows, g).upload(np.random.default_rng(0).random(rows + g))
    y = pa.DeviceVector(rows, 0)
    xo = pa.DeviceVector(rows, 0).upload(np.random.default_rng(1).random(rows))

    def product():
        pa.spmv_(y, blk, xo, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0)

    def with_rccl():
        L.call("pa_exchange_pack", plan, x.h, L.CONSISTENT)
        L.call("pa_exchange_rccl", plan, comm, L.CONSISTENT)
        product()
        L.call("pa_exchange_finish", plan, x.h, L.CONSISTENT)

    # (the push tables name one slice per neighbour: the same bytes as ONE self-addressed slice)
    plan_p = C.c_void_p()
    ptrs1 = np.array([1, 1 + g], np.int32)
    L.call("pa_plan_create", ctx.h, 1, rows + g, 1, L.ptr(nbr[:1].copy()), L.ptr(ptrs1), L.ptr(ghost_ids), 1, L.ptr(nbr[:1].copy()), L.ptr(ptrs1), L.ptr(own_ids), 1,
           C.byref(plan_p))
    plans1 = (C.c_void_p * 1)(plan_p.value)
    vecs1 = (C.c_void_p * 1)(x.h.value)

    def with_push():
        L.call("pa_exchange_push_local", plans1, 1, vecs1, L.CONSISTENT)
        product()
        L.call("pa_exchange_finish", plan_p, x.h, L.CONSISTENT)

    def timed(f):
        for _ in range(200 if n == 256 else 600):
            f()
        ctx.sync()
        best = 1e9
        for _ in range(3):
            e0 = ctx.event().record(L.STREAM_COMPUTE)
            for _ in range(reps):
                f()
            e1 = ctx.event().record(L.STREAM_COMPUTE)
            ctx.sync()
            best = min(best, e0.elapsed_ms(e1) / reps)
        return best
    t_alone = timed(product)
    t_rccl = timed(with_rccl)
    t_push = timed(with_push)
    t_alone2 = timed(product)
    base = min(t_alone, t_alone2)
    out[f"{n}^3"] = {"own_own_alone_ms": round(base, 4), "with_rccl_group_ms": round(t_rccl, 4), "with_push_launch_ms": round(t_push, 4),
                     "overlap_interference_pct": round(100 * (t_rccl - base) / base, 2),
                     "overlap_interference_pct_push": round(100 * (t_push - base) / base, 2),
                     "messages_doubles": lens}
    L.call("pa_plan_destroy", plan)
    L.call("pa_plan_destroy", plan_p)
    del A, blk, x, y, xo
print(json.dumps(out))
L.call("pa_comm_destroy", comm)
