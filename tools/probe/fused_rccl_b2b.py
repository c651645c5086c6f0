"""Back-to-back products of a part that is its own neighbour (tests/test_gpu_self_exchange.py) over a 1-rank RCCL communicator and
over the ipc link to itself: K products queued WITHOUT a host synchronisation in between, an event between every two, per-product ms.
  python tools/probe/fused_rccl_b2b.py [n=128] [K=12]     (PA_IPC_TIMEOUT_S bounds an in-launch wait; default here 2 s)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ.setdefault("PA_IPC_TIMEOUT_S", "2")
import numpy as np
import test_gpu_self_exchange as T
from gpu_helpers import pa, reload_switches
import pa_amd._lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
K = int(sys.argv[2]) if len(sys.argv) > 2 else 12
P = T.SelfPeriodicPart(n)
ctx = P.ctx
comm = T._comm(ctx)
x = np.random.default_rng(1).integers(-3, 4, P.n_own).astype(np.float64)
P.b.upload(np.concatenate([x, np.zeros(P.n_ghost)]))
want = P.expected_integer(x)
for link in ("rccl", "ipc"):
    if link == "ipc":
        P.connect_ipc_to_itself()
    cm = comm if link == "rccl" else None
    for fused in ("0", "1"):
        os.environ["PA_MUL_FUSED"] = fused
        reload_switches()
        for sync_between in (True, False):
            evs = [ctx.event().record(L.STREAM_COMPUTE)]
            t0 = time.perf_counter()
            err = None
            try:
                for k in range(K):
                    L.call("pa_mul5", P.m, cm, P.c.h, P.b.h, 1.0, 0.0)
                    evs.append(ctx.event().record(L.STREAM_COMPUTE))
                    if sync_between:
                        ctx.sync()
                ctx.sync()
            except L.PAError as e:
                err = str(e)[:160]
                try:
                    ctx.sync()
                except L.PAError:
                    pass
            wall = time.perf_counter() - t0
            ms = [round(evs[k].elapsed_ms(evs[k + 1]), 3) for k in range(len(evs) - 1)]
            ok = bool(np.array_equal(P.c.download(), want))
            print(f"{link} fused={fused} sync_between={sync_between}: wall {wall:.3f} s, ok={ok}, err={err}\n   per product ms: {ms}", flush=True)
