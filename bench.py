#!/usr/bin/env python
"""bench.py -- HPCG 27-point distributed SpMV (mul!) on N MI355X, one process per GPU.

    python bench.py --gpus 1 --steps 50 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A "step" is one mul!(c, A, b) of the reference (src/p_sparse_matrix.jl:2090-2103) on the HPCG 27-point
matrix, 256^3 rows per part (BASELINE.json: the size the metric is quoted on): consistent!(b) [pack ->
RCCL neighbour exchange -> unpack] overlapped with own x own, then own x ghost.  Weak scaling: part p of an
(npx,npy,npz) = compute_optimal_shape_XYZ(N) grid lives on GPU p-1.  Inputs are resident in HBM before the
timed region.  Rank 0 prints ONE JSON line.

Order of the run: set-up, parity gate, warm-up, the K timed steps (wall clock between barriers, nothing else in the
loop), then -- outside the headline's timed region -- a second pass of K steps with HIP events around the dominant
kernel (roofline), the overlap-off comparison (N > 1), the CG loop, the other BASELINE configs (N = 1) and the CPU baseline.
"""
import argparse
import json
import os
import sys
import time

# dmabuf IPC is the only mode the host driver supports (RCCL / cross-process GPU memory); harmless for one process
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))


class _Phase(list):
    """PHASE[0] = name: where the run is (watchdog, error messages) -- and, since round 6, how long every phase took (the line's
    `phase_seconds`: the driver's budget for this command is wall clock)"""

    def __init__(self):
        super().__init__(["start"])
        self.t0 = time.perf_counter()
        self.log = []

    def __setitem__(self, i, name):
        now = time.perf_counter()
        self.log.append((self[0], round(now - self.t0, 1)))
        self.t0 = now
        super().__setitem__(i, name)

    def seconds(self):
        out = {}
        for name, dt in self.log + [(self[0], round(time.perf_counter() - self.t0, 1))]:
            if dt >= 0.5:
                out[name] = round(out.get(name, 0.0) + dt, 1)
        return out


PHASE = _Phase()


def watchdog(seconds):
    """A multi-process run that deadlocks (a neighbour exchange that never completes) would sit until the driver's
    own limit; instead say where it hung and exit non-zero.  Generous: set-up + 4 x the default run fit many times."""
    import threading

    def fire():
        print(f"[bench watchdog] no progress after {seconds} s, phase = {PHASE[0]!r}; aborting", file=sys.stderr, flush=True)
        if LINE[0] is not None:                     # (rank 0, headline measured: the line as far as it is known, and status 0)
            LINE[0].setdefault("optional_sections_unfinished", []).append(f"watchdog in phase {PHASE[0]!r}")
            print(json.dumps(LINE[0]), flush=True)
            os._exit(0)
        os._exit(4)
    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t
sys.path.insert(0, ROOT)

LINE = [None]            # rank 0: the JSON line as far as it is known (the headline first, optional sections add to it)


class optional_section:
    """Nothing after the headline measurement may cost the run its line.  With one process an exception is enough to skip
    a section; with N processes a rank that fails or hangs inside a section leaves the others waiting in a collective,
    so every rank arms the same timer: when it fires (or the section raises), rank 0 prints the line it has -- headline
    complete, the section listed under `optional_sections_unfinished` -- and every rank leaves with status 0."""

    def __init__(self, name, seconds, N, rank):
        seconds = float(os.environ.get("PA_BENCH_SECTION_TIMEOUT_S", seconds))
        self.name, self.seconds, self.N, self.rank, self.timer = name, seconds, N, rank, None

    def _bail(self, why):
        print(f"[bench rank {self.rank}] optional section {self.name!r} {why}; the line goes out without it", file=sys.stderr, flush=True)
        if self.rank == 0 and LINE[0] is not None:
            LINE[0].setdefault("optional_sections_unfinished", []).append(self.name)
            print(json.dumps(LINE[0]), flush=True)
        os._exit(0)

    def __enter__(self):
        PHASE[0] = self.name
        if self.N > 1:
            import threading
            self.timer = threading.Timer(self.seconds, self._bail, args=(f"did not finish within {self.seconds:.0f} s",))
            self.timer.daemon = True
            self.timer.start()
            if os.environ.get("PA_BENCH_FAULT") == f"{self.name}:hang:{self.rank}":     # (tests: a rank that never comes back)
                time.sleep(1e6)
        return self

    def __exit__(self, et, ev, tb):
        if self.timer is not None:
            self.timer.cancel()
        if et is None or not issubclass(et, Exception):
            return False
        if self.N > 1:
            import traceback
            traceback.print_exception(et, ev, tb)
            self._bail(f"raised {et.__name__}: {ev}")
        print(f"[bench] optional section {self.name!r} skipped: {ev}", file=sys.stderr, flush=True)
        return True


FP64_VECTOR_PEAK_TFLOPS = 78.6      # MI355X vector fp64 with FMA (256 CUs x 4 SIMDs x 16 lanes x 2 flops x 2.4 GHz); unfused multiply + add: half
HBM_PEAK_GBPS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy kernel achieves


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--grid", dest="n", type=int, default=256, help="grid points per direction per part")
    ap.add_argument("--cg-iters", type=int, default=int(os.environ.get("PA_BENCH_CG", "20")),
                    help="iterations of the CG loop timed after the headline measurement (0: skip)")
    ap.add_argument("--no-value-dict", dest="value_dict", action="store_false",
                    help="skip the extra measurement of the optional value-dictionary mode (N=1 only)")
    ap.add_argument("--no-overlap", action="store_true",
                    help="the timed step waits for the ghost exchange BEFORE own x own (HPCG's mul_no_lat! order): "
                         "what the overlap is worth is `overlap.ms_per_step_off` vs `_on` of a default run")
    ap.add_argument("--no-extra", dest="extra", action="store_false", help="skip BASELINE configs 2, 3, 5 (N=1 only)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-live-traffic", dest="live_traffic", action="store_false",
                    help="do not measure roofline.traffic with rocprofv3 --pmc child passes of this command (N=1 only); quote the "
                         "committed profiles/rNN_summary.json instead")
    ap.add_argument("--traffic-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="CPU work of the baseline sample (per rank)")
    return ap.parse_args()


def hash_x(gids):
    """x[gid] = ((gid*2654435761) mod 2^32)/2^32: stateless, identical on any number of parts."""
    g = np.asarray(gids).astype(np.uint64)
    return ((g * np.uint64(2654435761)) % np.uint64(2 ** 32)).astype(np.float64) / float(2 ** 32)


# ---------------------------------------------------------------------------------------------------------------------
# CPU baseline: the reference's mul! as its own processes would run it -- one part per process, one core per process
# (mpiexec -n P with single-threaded ranks, src/mpi_array.jl:42-53), every loop the oracle's C restatement
# ---------------------------------------------------------------------------------------------------------------------
def cpu_mul_baseline(pa, A, N, rank, seconds, host_shape=None):
    """pack -> exchange -> spmv_csr!(own x own) -> unpack -> muladd!(own x ghost) of THIS rank's part at the bench's own
    size, on one pinned core, with the oracle's C loops (oracle/pa_oracle.c; src/p_vector.jl:587-612,
    src/sparse_utils.jl:649-669, src/p_sparse_matrix.jl:2088,2098-2101); the exchange between the ranks' buffers goes
    through torch.distributed's CPU backend (gloo) the way the reference's goes through MPI.  All ranks run
    concurrently; the reported time per mul! is the slowest rank's median."""
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_oracle
    orc = load_oracle()
    K = orc.oracle_c()
    if not hasattr(K, "lib"):
        return None
    if A.host_blocks is not None:
        oo, oh = pa.local_items(A.host_blocks)[0]
    else:       # the product's blocks were generated in HBM: the baseline makes its own host copy (outside every timed region)
        from pa_amd.gallery import build_split_blocks_fused
        _, oo, oh, _ = build_split_blocks_fused(pa.local_items(A.row_partition)[0], *host_shape)
    ind = pa.local_items(A.col_partition)[0]
    cache = ind.cache
    nbr_snd, nbr_rcv = np.asarray(cache["neighbors_snd"]), np.asarray(cache["neighbors_rcv"])
    lsnd, lrcv = cache["local_indices_snd"], cache["local_indices_rcv"]

    # one PHYSICAL core per rank: of every set of hyper-thread siblings only the first logical CPU counts (Linux numbers
    # the siblings of core i as i and i + n/2, so "every (n/N)-th logical CPU" would pin two ranks to one core)
    allowed = sorted(os.sched_getaffinity(0))
    cores, seen = [], set()
    for cpu in allowed:
        try:
            sib = open(f"/sys/devices/system/cpu/cpu{cpu}/topology/thread_siblings_list").read().strip()
        except OSError:
            sib = str(cpu)
        if sib not in seen:
            seen.add(sib)
            cores.append(cpu)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    mine = cores[(local * len(cores)) // max(N, 1) % len(cores)]     # spread over the sockets / CCDs, as `--map-by` would
    old = os.sched_getaffinity(0)
    os.sched_setaffinity(0, {mine})
    torch_threads = torch.get_num_threads()
    torch.set_num_threads(1)
    # the operands are (re)created by the pinned thread: first touch puts them on this core's NUMA node, where an MPI rank
    # of the reference would have allocated them
    # consistent! uses the reversed cache (src/p_vector.jl:748): pack the own ids others ghost (rcv side), unpack into my ghosts
    buf_out = np.zeros(len(lrcv.data))
    buf_in = np.zeros(len(lsnd.data))
    x = np.zeros(ind.n_local)
    x[:ind.n_own] = hash_x(ind.own_to_global)
    y = np.zeros(oo.m)
    Aoo = orc.CSR(oo.m, oo.n, oo.rowptr.copy(), oo.colval.copy(), oo.nzval.copy())
    Aoh = orc.CSR(oh.m, oh.n, oh.rowptr.copy(), oh.colval.copy(), oh.nzval.copy())
    t_out, t_in = torch.from_numpy(buf_out), torch.from_numpy(buf_in)
    xg = x[ind.n_own:]                       # ghost values: a view (the device layout [own | ghost] is the host layout too)
    glids = (np.asarray(lsnd.data, np.int64) - ind.n_own).astype(np.int32)
    plids = np.ascontiguousarray(lrcv.data, np.int32)

    def one():
        K.pack(buf_out, x, plids)
        reqs = []
        for k, q in enumerate(nbr_snd):     # my ghosts' owners send to me
            a, e = int(lsnd.ptrs[k]) - 1, int(lsnd.ptrs[k + 1]) - 1
            if e > a:
                reqs.append(dist.irecv(t_in[a:e], int(q) - 1))
        for k, q in enumerate(nbr_rcv):
            a, e = int(lrcv.ptrs[k]) - 1, int(lrcv.ptrs[k + 1]) - 1
            if e > a:
                reqs.append(dist.isend(t_out[a:e], int(q) - 1))
        K.spmv_csr(y, x[:ind.n_own], Aoo)                      # own x own while the messages travel
        for r in reqs:
            r.wait()
        K.unpack_insert(xg, buf_in, glids)
        K.mul5_csr(y, Aoh, xg, 1.0, 1.0)                       # muladd!
    try:
        one()
        if N > 1:
            dist.barrier()
        times = []
        t_begin = time.perf_counter()
        while True:
            t0 = time.perf_counter()
            one()
            times.append(time.perf_counter() - t0)
            stop = 1 if (time.perf_counter() - t_begin > seconds and len(times) >= 3) or len(times) >= 200 else 0
            if N > 1:                        # every rank does the same number of products (they exchange messages)
                flag = torch.tensor([stop])
                dist.all_reduce(flag, op=dist.ReduceOp.MAX)
                stop = int(flag.item())
            if stop:
                break
    finally:
        os.sched_setaffinity(0, old)
        torch.set_num_threads(torch_threads)
    med = float(np.median(times))
    nnz = oo.nnz + oh.nnz
    if N > 1:
        tt = torch.tensor([med, float(nnz)], dtype=torch.float64)
        mx = tt.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tt, op=dist.ReduceOp.SUM)
        med, nnz_total = float(mx[0].item()), int(tt[1].item())
    else:
        nnz_total = nnz
    bytes_part = nnz * 12 + (oo.m + 1) * 4 + ind.n_local * 8 + oo.m * 8
    cpu_mul_baseline.y = y            # the oracle's mul! of the bench's own x at the bench's own size: the full-size parity gate
    return {"value": round(2.0 * nnz_total / med / 1e9, 3), "unit": "GFLOP/s", "cores": N, "kind": "port",
            "ms_per_mul": round(med * 1e3, 2), "gflops": round(2.0 * nnz_total / med / 1e9, 3),
            "gbps_algorithmic_per_core": round(bytes_part / med / 1e9, 2),
            "pinned_to_cpu": int(mine), "physical_cores_available": len(cores),
            "sample": f"the bench's own workload ({oo.m} rows, {nnz} stored entries per part, {N} part(s)): {len(times)} x mul! per "
                      f"rank = pack / exchange (gloo p2p) / spmv_csr! / unpack / muladd!, oracle/pa_oracle.c loops (-O3 "
                      f"-ffp-contract=off), one process pinned to one core per part (mpiexec -n {N} of the reference), median of "
                      f"the slowest rank"}


def cpu_c1_debugarray(seconds=3.0):
    """BASELINE config 1 the way DebugArray runs it (src/debug_array.jl:110-117): gallery laplacian 7-point 64^3 on 4
    parts (2,2,1), ONE core looping over the parts, mul! = consistent! + spmv! + muladd! with the oracle's loops."""
    from __graft_entry__ import load_oracle
    orc = load_oracle()
    if not hasattr(orc.oracle_c(), "lib"):
        return None
    Io, Jo, Vo, rows, _ = orc.laplacian_fdm_fast((64, 64, 64), (2, 2, 1))        # src/gallery.jl:12-86, docs/examples.jl:223
    A = orc.psparse_from_coo(Io, Jo, Vo, rows)
    K = orc.oracle_c()
    x = [np.ascontiguousarray(orc.hash_x(c.local_to_global) * (c.local_to_owner == c.part)) for c in A.cols]
    y = [np.zeros(r.n_own) for r in A.rows]
    cache = orc.p_vector_cache(x, A.cols).reverse()             # consistent! = reversed cache + insert (src/p_vector.jl:748)
    lsnd = [np.ascontiguousarray(l.data, np.int32) for l in cache.local_indices_snd]
    lrcv = [np.ascontiguousarray(l.data, np.int32) for l in cache.local_indices_rcv]
    copies = []                                                 # exchange_impl! of DebugArray: src/primitives.jl:1020-1042
    for r, rids in enumerate(cache.neighbors_rcv):
        for i, sp in enumerate(rids):
            j = list(cache.neighbors_snd[sp - 1]).index(r + 1)
            pr, ps = cache.buffer_rcv[r].ptrs, cache.buffer_snd[sp - 1].ptrs
            copies.append((cache.buffer_rcv[r].data, int(pr[i]) - 1, int(pr[i + 1]) - 1, cache.buffer_snd[sp - 1].data, int(ps[j]) - 1, int(ps[j + 1]) - 1))
    P = len(x)

    def mul():                                                  # mul!(c,a,b), src/p_sparse_matrix.jl:2098-2101, parts in sequence
        for p in range(P):
            K.pack(cache.buffer_snd[p].data, x[p], lsnd[p])
        for dst, a0, a1, src, b0, b1 in copies:
            dst[a0:a1] = src[b0:b1]
        for p in range(P):
            K.spmv_csr(y[p], x[p][:A.cols[p].n_own], A.blocks[p].own_own)
        for p in range(P):
            K.unpack_insert(x[p], cache.buffer_rcv[p].data, lrcv[p])
        for p in range(P):
            K.mul5_csr(y[p], A.blocks[p].own_ghost, x[p][A.cols[p].n_own:], 1.0, 1.0)
    mul()
    want = [np.zeros(r.n_local) for r in A.rows]
    orc.mul(want, A, [v.copy() for v in x])                     # the oracle's own mul! on the same input: same bits
    assert all(np.array_equal(g, w[:len(g)]) for g, w in zip(y, want)), "the timed C1 loop differs from the oracle's mul!"
    times, t_begin = [], time.perf_counter()
    while time.perf_counter() - t_begin < seconds or len(times) < 3:
        t0 = time.perf_counter()
        mul()
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    nnz = sum(b.own_own.nnz + b.own_ghost.nnz for b in A.blocks)
    return {"what": "config 1: 7-pt 64^3, 4 parts (2,2,1), DebugArray order (parts one after the other), 1 core",
            "cores": 1, "ms_per_mul": round(med * 1e3, 3), "gflops": round(2.0 * nnz / med / 1e9, 3), "nnz": int(nnz),
            "reps": len(times)}


def live_traffic(n, timeout_s=120):
    """HBM bytes per launch of the dominant kernel FROM THIS BOX, THIS COMMAND: two child processes of bench.py (headline only,
    10 timed steps) under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` -- separate passes, counters only, as
    MI355X_MICROARCH.md prescribes (KiB units; FETCH_SIZE doubled on gfx950 for wide streaming reads) -- averaged over the
    k_spmv_rowsplit launches of each pass.  Returns (bytes, source) or (None, why)."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or ("/opt/rocm/bin/rocprofv3" if os.path.exists("/opt/rocm/bin/rocprofv3") else None)
    if exe is None:
        return None, "rocprofv3 not found"
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ):
        return None, "this process is itself running under a profiler"
    got = {}
    work = tempfile.mkdtemp(prefix="pa_bench_pmc_", dir="/tmp")
    try:
        for counter in ("FETCH_SIZE", "WRITE_SIZE"):
            out = os.path.join(work, counter)
            cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", out, "-o", "c", "--", sys.executable,
                   os.path.join(ROOT, "bench.py"), "--grid", str(n), "--steps", "10", "--warmup", "2", "--no-cpu-baseline",
                   "--no-value-dict", "--no-extra", "--cg-iters", "0", "--traffic-child"]
            env = dict(os.environ, TMPDIR="/tmp")
            r = subprocess.run(cmd, cwd=work, env=env, capture_output=True, text=True, timeout=timeout_s)
            vals = []
            for f in glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True):
                for row in csv.DictReader(open(f)):
                    if any(k in row.get("Kernel_Name", "") for k in ("k_spmv_pell", "k_spmv_rowsplit")) and row.get("Counter_Name") == counter:
                        vals.append(float(row["Counter_Value"]))
            if not vals:
                return None, f"no {counter} rows (rocprofv3 rc {r.returncode})"
            got[counter] = (sum(vals) / len(vals), len(vals))
    except Exception as e:                                        # noqa: BLE001
        return None, f"{type(e).__name__}: {e}"
    finally:
        shutil.rmtree(work, ignore_errors=True)
    fetch, write = 2 * got["FETCH_SIZE"][0] * 1024, got["WRITE_SIZE"][0] * 1024
    return round(fetch + write), (f"this box, this command: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE over two child runs of bench.py "
                                  f"(headline only; {got['FETCH_SIZE'][1]} / {got['WRITE_SIZE'][1]} launches of the product kernel averaged; KiB units, "
                                  f"FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950: fetch {fetch / 1e9:.3f} GB + write {write / 1e9:.3f} GB)")


def calibrate_box(pa, ctx, L):
    """What this box's HBM delivers to the simplest kernels ("re-confirm on the box with a device triad"): a
    device-to-device copy and a two-stream read (dot) over vectors far larger than the 256 MiB Infinity Cache."""
    PHASE[0] = "copy / read calibration"
    m = 1 << 27                                              # 1 GiB per vector
    va, vb = pa.DeviceVector(m, 0), pa.DeviceVector(m, 0)
    va.fill(1.0), vb.fill(2.0)

    def ev_time(f, reps=5):
        f()
        e0 = ctx.event().record(L.STREAM_COMPUTE)
        for _ in range(reps):
            f()
        e1 = ctx.event().record(L.STREAM_COMPUTE)
        ctx.sync()
        return e0.elapsed_ms(e1) / reps
    t_copy = ev_time(lambda: L.call("pa_vec_copy", vb.h, va.h, L.SEG_OWN))
    t_read = ev_time(lambda: L.call("pa_vec_dot_slot", va.h, vb.h, 5, 0))
    return {"copy_gbps": round(2 * 8 * m / t_copy / 1e6, 1), "read_gbps": round(2 * 8 * m / t_read / 1e6, 1),
            "what": "1 GiB vectors: hipMemcpyAsync device-to-device (read+write bytes) and k_dot_partial (two read streams)"}


def spin_up(ctx, f, busy_ms=200.0):
    """Bring the GPU back to its working clocks after host-side work left it idle: repeat f until the device has been busy
    for `busy_ms` (the product needs ~40 launches = 30 ms at 256^3 after the short idle of the parity gate, see the clock
    ramp in main(); after the seconds of idle while the host builds another matrix 45 ms were not enough -- a 0.115 ms kernel
    measured 0.125 ms, tools/probe/xwin_rate.py)."""
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < busy_ms:
        for _ in range(10):
            f()
        ctx.sync()


def time_block(pa, ctx, L, blk, n_rows, n_cols, reps=30):
    """Events around `reps` products of one block with its own vectors: ms per product."""
    x = pa.DeviceVector(n_cols, 0).upload(hash_x(np.arange(1, n_cols + 1)))
    y = pa.DeviceVector(n_rows, 0)
    spin_up(ctx, lambda: pa.spmv_(y, blk, x))
    for _ in range(5):
        pa.spmv_(y, blk, x)
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    for _ in range(reps):
        pa.spmv_(y, blk, x)
    e1 = ctx.event().record(L.STREAM_COMPUTE)
    ctx.sync()
    return e0.elapsed_ms(e1) / reps


def chain_info(P, fused=True):
    """Launches of one mul! of P parts in one process (pa_mul_all).  Round 5 (csrc/pa_fused.hip): 1 push launch that packs, delivers
    AND writes b's ghost entries (consistent! complete), then ONE launch per part -- own x own's chunks with the part's boundary rows
    (own_own entries, then own_ghost entries read from the receive buffer) as the launch's tail; one stream, no events.  Round 4
    (PA_MUL_FUSED=0): 1 push, P own x own, P own x ghost, 1 unpack."""
    if fused:
        return {"launches_per_step": P + 1, "launches_per_step_round4": 2 * P + 2,
                "critical_path_launches": {"before_own_own": 1, "beside_own_own": 0, "after_own_own": 0}}
    return {"launches_per_step": 2 * P + 2, "launches_per_step_round3": "4 per part + one copy per directed edge",
            "critical_path_launches": {"before_own_own": 0, "beside_own_own": 1, "after_own_own": 1, "after_own_ghost_off_path": 1}}


def fused_info(pa, ctx, L, A, x, y, reps=30):
    """Which parts' mul! runs as one launch (pa_matrix_fused), their boundary rows, and the same mul! with PA_MUL_FUSED=0 (round 4's
    separate launches) timed right beside it."""
    import ctypes as C
    import pa_amd.p_sparse_matrix as psm
    fused, nb = [], []
    for h in pa.local_items(psm._operator_handles(A, x)):
        yes, n = C.c_int(), C.c_int64()
        L.call("pa_matrix_fused", h, C.byref(yes), C.byref(n))
        fused.append(bool(yes.value)); nb.append(int(n.value))
    os.environ["PA_MUL_FUSED"] = "0"
    ctx.reload_env()
    try:
        for _ in range(5):
            pa.mul_c_(y, A, x)
        e0 = ctx.event().record(L.STREAM_COMPUTE)
        for _ in range(reps):
            pa.mul_c_(y, A, x)
        e1 = ctx.event().record(L.STREAM_COMPUTE)
        ctx.sync()
        ms_sep = e0.elapsed_ms(e1) / reps
    finally:
        os.environ.pop("PA_MUL_FUSED", None)
        ctx.reload_env()
    return all(fused), {"parts_as_one_launch": int(sum(fused)), "boundary_rows_per_part": nb,
                        "ms_per_part_mul_separate_launches": round(ms_sep / len(fused), 4)}


def whole_mul_times(pa, ctx, L, A, x, y, reps=30, graph=True):
    """ms per mul! of all parts of A (eager: one library call; replayed from a hipGraph) and ms of the parts' own x own alone."""
    spin_up(ctx, lambda: pa.mul_c_(y, A, x))
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    for _ in range(reps):
        pa.mul_c_(y, A, x)
    e1 = ctx.event().record(L.STREAM_COMPUTE)
    ctx.sync()
    ms = e0.elapsed_ms(e1) / reps
    ms_graph = float("nan")
    try:
        if not graph:
            raise RuntimeError("not asked for")
        with pa.Graph() as g:
            pa.mul_c_(y, A, x)
        for _ in range(10):
            g.launch()
        e0 = ctx.event().record(L.STREAM_COMPUTE)
        for _ in range(reps):
            g.launch()
        e1 = ctx.event().record(L.STREAM_COMPUTE)
        ctx.sync()
        ms_graph = e0.elapsed_ms(e1) / reps
    except Exception as e:                                       # noqa: BLE001
        print(f"[bench] hipGraph replay of mul! skipped: {e}", file=sys.stderr)
    blocks, xs, ys = pa.local_items(A.matrix_partition), pa.local_items(x.vector_partition), pa.local_items(y.vector_partition)

    def spmv_only():
        for blk, xv, yv in zip(blocks, xs, ys):
            pa.spmv_(yv, blk.own_own, xv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0)
    for _ in range(5):
        spmv_only()
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    for _ in range(reps):
        spmv_only()
    e1 = ctx.event().record(L.STREAM_COMPUTE)
    ctx.sync()
    return ms, ms_graph, e0.elapsed_ms(e1) / reps


def kernel_of(blk):
    """which product kernel pa_spmv runs this block on now (pa_csr_pell_info)"""
    try:
        p = blk.pell()
    except Exception:                                          # noqa: BLE001
        return "k_spmv_rowsplit"
    if p["mode"] == 0:
        return "k_spmv_rowsplit (row split, LDS-staged products)"
    lean = p.get("lean_slabs_bits" if p["mode"] == 2 else "lean_slabs", 0)
    return (f"k_spmv_pell (pattern-ELL, one lane per row; {p['slabs']} slabs of 64 rows, {p['patterns']} slab patterns in {p.get('classes', 0)} classes, "
            f"lean form in {lean} slabs, unroll {p['unroll']}; "
            + ("fp64 value stream" if p["mode"] == 1 else "one bit per entry: a two-value dictionary" if p["mode"] == 2
               else "one byte per entry: a dictionary of 3 .. 64 values") + ")")


def extra_configs(pa, ctx, L, out):
    """BASELINE configs 2, 3 (one part's size) and 5 (one part's rows) on this GPU, one entry each: ms per product,
    GFLOP/s, algorithmic GB/s (12 B per entry + 20 B per row), moved GB/s (bytes the block's encoding makes the
    kernel read + x once + y once), column encoding of the chunks -- plus rows without a pattern inside a band.  Entries
    are appended to `out` as they are measured (a failure keeps what came before it)."""
    ranks1 = pa.DebugArray([1])

    def entry(workload, blk, n_rows, n_cols, ms, t_setup):
        nnz = blk.nnz
        alg = nnz * 12 + (n_rows + 1) * 4 + n_cols * 8 + n_rows * 8
        moved = blk.stream_bytes() + n_cols * 8 + n_rows * 8
        return {"workload": workload, "rows": int(n_rows), "nnz": int(nnz), "ms": round(ms, 4),
                "gflops": round(2.0 * nnz / ms / 1e6, 1), "algorithmic_gbps": round(alg / ms / 1e6, 1),
                "moved_bytes_per_launch": int(moved), "moved_gbps": round(moved / ms / 1e6, 1),
                "frac_moved": round(moved / ms / 1e6 / HBM_PEAK_GBPS, 4), "frac_spec_8d": round(alg / ms / 1e6 / HBM_PEAK_GBPS, 4),
                "encoding": blk.encoding(), "kernel": kernel_of(blk), "setup_s": round(t_setup, 1)}
    # config 2: 7-point Laplacian 256^3, one part, SpMV only (gallery laplacian_fdm -> psparse, Int32 CSR)
    PHASE[0] = "extra: config 2"
    t = time.perf_counter()
    I, J, V, rows, cols = pa.laplacian_fdm((256, 256, 256), (1, 1, 1), ranks1)
    A2 = pa.psparse_from_coo(I, J, V, rows)
    del I, J, V
    b2 = A2.matrix_partition.items[0].own_own
    ts = time.perf_counter() - t
    out.append(entry("config 2: gallery 7-pt Laplacian 256^3, 1 part, pa_spmv only", b2, b2.m, b2.n,
                     time_block(pa, ctx, L, b2, b2.m, b2.n), ts))
    del A2, b2
    # config 3's size: 27-point 128^3, one part, mul!
    PHASE[0] = "extra: config 3 size"
    t = time.perf_counter()
    A3, _ = pa.build_p_matrix(ranks1, 128, 128, 128, 128, 128, 128, 1, 1, 1)
    b3 = A3.matrix_partition.items[0].own_own
    ts = time.perf_counter() - t
    x3, y3 = pa.pvector_from_function(lambda ind: hash_x(ind.get_local_to_global()), A3.col_partition), pa.pzeros(A3.row_partition)
    spin_up(ctx, lambda: pa.mul_(y3, A3, x3))
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    for _ in range(50):
        pa.mul_(y3, A3, x3)
    e1 = ctx.event().record(L.STREAM_COMPUTE)
    ctx.sync()
    out.append(entry("config 3's size: HPCG 27-pt 128^3, 1 part, mul!", b3, b3.m, b3.n, e0.elapsed_ms(e1) / 50, ts))
    del A3, b3, x3, y3
    # config 3 itself: 27-point 128^3 per part, 2 parts as (2,1,1), both on this ONE GPU: the whole mul! (push launch, own x own,
    # own x ghost from the receive buffers, unpack) per part against own x own alone (<= 1.15 x)
    PHASE[0] = "extra: config 3 on 2 parts"
    t = time.perf_counter()
    ranks2 = pa.DebugArray([1, 2])
    A32, _ = pa.build_p_matrix(ranks2, 128, 128, 128, 256, 128, 128, 2, 1, 1)
    ts = time.perf_counter() - t
    x32 = pa.pvector_from_function(lambda ind: hash_x(ind.get_local_to_global()) * (ind.get_local_to_owner() == ind.part), A32.col_partition)
    y32 = pa.pzeros(A32.row_partition)
    ms, ms_graph, ms_spmv = whole_mul_times(pa, ctx, L, A32, x32, y32)
    nnz32 = sum(b.own_own.nnz + b.own_ghost.nnz for b in pa.local_items(A32.matrix_partition))
    all_fused, finfo = fused_info(pa, ctx, L, A32, x32, y32)
    out.append({"workload": "config 3 whole: HPCG 27-pt 128^3 per part, 2 parts (2,1,1) BOTH on this one GPU, mul! = one push launch (pack + "
                            "deliver + b's ghosts) + ONE launch per part (own x own's chunks, boundary rows as the tail) (pa_mul_all)",
                **finfo,
                "parts": 2, "nnz": int(nnz32), "ghosts_per_part": [c.n_ghost for c in pa.local_items(A32.col_partition)],
                "ms_per_part_mul": round(ms / 2, 4), "ms_per_part_mul_hipgraph": round(ms_graph / 2, 4), "ms_per_part_spmv": round(ms_spmv / 2, 4),
                "mul_over_spmv": round(ms / ms_spmv, 3), **chain_info(2, all_fused), "gflops": round(2.0 * nnz32 / ms / 1e6, 1), "setup_s": round(ts, 1)})
    del A32, x32, y32
    # consistent!(v) and assemble!(v) on their own (rows a7-a9 of SURVEY 8): config 4's partition shape, (2,2,2) parts of 128^3
    # rows with all 26-neighbour ghost layers, the 8 parts on this ONE GPU -- one push launch (pack + deliver) and one unpack launch
    # per call, whatever the number of parts and neighbours
    PHASE[0] = "extra: consistent! / assemble! alone"
    try:
        t = time.perf_counter()
        ranks8 = pa.DebugArray(range(1, 9))
        A48, _ = pa.build_p_matrix(ranks8, 128, 128, 128, 256, 256, 256, 2, 2, 2)
        ts = time.perf_counter() - t
        v48 = pa.pvector_from_function(lambda ind: hash_x(ind.get_local_to_global()) * (ind.get_local_to_owner() == ind.part), A48.col_partition)
        ghosts48 = [c.n_ghost for c in pa.local_items(A48.col_partition)]
        res48 = {}
        for name, op in (("consistent", pa.consistent_), ("assemble", pa.assemble_)):
            for _ in range(10):
                op(v48).wait()
            e0 = ctx.event().record(L.STREAM_COMPUTE)
            for _ in range(50):
                op(v48).wait()
            e1 = ctx.event().record(L.STREAM_COMPUTE)
            ctx.sync()
            res48[name] = e0.elapsed_ms(e1) / 50
        out.append({"workload": "consistent!(v) / assemble!(v) alone: (2,2,2) parts of 128^3 rows (config 4's shape, 26 neighbour relations per "
                                "interior corner), the 8 parts on this one GPU, push transport (one launch packs and delivers, one unpacks / adds in order)",
                    "parts": 8, "ghosts_per_part": ghosts48, "ms_per_consistent_all_parts": round(res48["consistent"], 4),
                    "ms_per_assemble_all_parts": round(res48["assemble"], 4),
                    "ghost_values_moved_per_call": int(sum(ghosts48)),
                    "gbps_consistent": round(3 * 8 * sum(ghosts48) / res48["consistent"] / 1e6, 1),
                    "what": "bytes: every ghost value is read from its owner, written to the receive buffer and read + written by the unpack "
                            "(3 x 8 B counted); these calls are latency-bound launches (2 per call), not bandwidth", "setup_s": round(ts, 1)})
        del A48, v48
    except Exception as ex:                                     # noqa: BLE001
        print(f"[bench] exchange-only entry skipped: {ex}", file=sys.stderr)
    # config 5: the rows one part of the 4096^2-node Q1 FEM matrix on (4,2) parts holds: 1024 x 2048 nodes
    PHASE[0] = "extra: config 5 part"
    t = time.perf_counter()
    I, J, V, rows, cols = pa.laplacian_fem((1024, 2048), (1, 1), ranks1)
    A5 = pa.psparse_disassembled(I, J, V, rows, cols)
    del I, J, V
    b5 = A5.matrix_partition.items[0].own_own
    ts = time.perf_counter() - t
    out.append(entry("config 5: Q1 FEM Laplacian 2-D, 1024 x 2048 nodes (= one part's rows of the 4096^2 instance on (4,2) "
                     "parts), disassembled psparse route, pa_spmv", b5, b5.m, b5.n, time_block(pa, ctx, L, b5, b5.m, b5.n), ts))
    del A5, b5
    # rows without any pattern inside a band (an unstructured mesh in a bandwidth-reducing numbering):
    # 4 M rows of 16 entries, columns drawn at random within +-2000 of the diagonal
    PHASE[0] = "extra: banded unstructured rows"
    t = time.perf_counter()
    rng = np.random.default_rng(0)
    m = 4_000_000
    col = np.repeat(np.arange(m, dtype=np.int32), 16).reshape(m, 16)
    col += rng.integers(-2000, 2000, size=(m, 16), dtype=np.int32)
    np.clip(col, 0, m - 1, out=col)
    col.sort(axis=1)
    col += 1
    Hb = pa.HostCSR(m, m, (1 + 16 * np.arange(m + 1)).astype(np.int32), col.ravel(), rng.standard_normal(m * 16))
    bb = pa.DeviceCSR(Hb)
    del Hb, col
    ts = time.perf_counter() - t
    e = entry("unstructured rows in a band: 4 M rows x 16 entries, random columns within +-2000 of the diagonal, pa_spmv",
              bb, m, m, time_block(pa, ctx, L, bb, m, m), ts)
    e["x_window_launch"] = bb.xwin()
    out.append(e)
    del bb
    # the same beyond every LDS window (VERDICT r02 #6a): columns within +-7900 -- a span of 15 900 entries of x per chunk; runs
    # of chunks gather from the sliding 128 KiB ring (k_spmv_xring), nothing else but the plain row split holds such a band
    PHASE[0] = "extra: wide band"
    t = time.perf_counter()
    col = np.repeat(np.arange(m, dtype=np.int32), 16).reshape(m, 16)
    col += rng.integers(-7900, 7900, size=(m, 16), dtype=np.int32)
    np.clip(col, 0, m - 1, out=col)
    col.sort(axis=1)
    col += 1
    Hw = pa.HostCSR(m, m, (1 + 16 * np.arange(m + 1)).astype(np.int32), col.ravel(), rng.standard_normal(m * 16))
    bw = pa.DeviceCSR(Hw)
    del Hw, col
    ts = time.perf_counter() - t
    e = entry("unstructured rows in a WIDE band: 4 M rows x 16 entries, random columns within +-7900 of the diagonal (beyond every LDS "
              "window), pa_spmv", bw, m, m, time_block(pa, ctx, L, bw, m, m), ts)
    e["x_window_launch"] = bw.xwin()
    out.append(e)
    del bw
    # beyond the ring too: +-16 000 -- the block is stored as a chain of three column pieces, each on the ring
    PHASE[0] = "extra: band beyond the ring"
    t = time.perf_counter()
    col = np.repeat(np.arange(m, dtype=np.int32), 16).reshape(m, 16)
    col += rng.integers(-16000, 16000, size=(m, 16), dtype=np.int32)
    np.clip(col, 0, m - 1, out=col)
    col.sort(axis=1)
    col += 1
    Hw = pa.HostCSR(m, m, (1 + 16 * np.arange(m + 1)).astype(np.int32), col.ravel(), rng.standard_normal(m * 16))
    bw = pa.DeviceCSR(Hw)
    del Hw, col
    ts = time.perf_counter() - t
    e = entry("unstructured rows in a band BEYOND THE RING: 4 M rows x 16 entries, random columns within +-16000 of the diagonal, stored as a "
              "chain of column pieces (pa_csr_colsplit_if_wide), pa_spmv", bw, m, m, time_block(pa, ctx, L, bw, m, m), ts)
    e["x_window_launch"] = bw.xwin()
    e["chain"] = bw.chain()
    # (the same pieces a launch each, y written and re-read between them: round 4's way)
    os.environ["PA_SPMV_CHAIN_FUSED"] = "0"
    ctx.reload_env()
    e["ms_a_launch_per_piece"] = time_block(pa, ctx, L, bw, m, m)
    os.environ.pop("PA_SPMV_CHAIN_FUSED")
    ctx.reload_env()
    out.append(e)
    del bw
    # a non-pattern FEM matrix (VERDICT r02 #6b): the Q1 mesh numbered at random, then renumbered by reverse Cuthill-McKee --
    # what an unstructured-mesh code hands over.  No row pattern survives; the columns fall into a band in a few clusters.
    PHASE[0] = "extra: FEM mesh after RCM"
    try:
        import scipy.sparse as sp
        from scipy.sparse.csgraph import reverse_cuthill_mckee
        t = time.perf_counter()
        nxm, nym = 1000, 800
        I, J, V, rows, cols = pa.laplacian_fem((nxm, nym), (1, 1), ranks1)
        nn = nxm * nym
        perm = np.random.default_rng(29).permutation(nn)
        Ip, Jp = perm[I.items[0] - 1], perm[J.items[0] - 1]
        G = sp.csr_matrix((np.ones(len(Ip), np.float32), (Ip, Jp)), shape=(nn, nn))
        order = reverse_cuthill_mckee(G, symmetric_mode=True)
        new_id = np.empty(nn, np.int64)
        new_id[order] = np.arange(nn)
        Hc = pa.compresscoo(new_id[Ip] + 1, new_id[Jp] + 1, V.items[0], nn, nn)
        band = int(np.max(np.abs(np.repeat(np.arange(nn), np.diff(Hc.rowptr)) - (Hc.colval - 1))))
        del I, J, V, G
        br = pa.DeviceCSR(Hc)
        ts = time.perf_counter() - t
        e = entry(f"Q1 FEM Laplacian on a {nxm} x {nym} mesh numbered at random, then reverse Cuthill-McKee (band {band}): no row "
                  "patterns, pa_spmv", br, nn, nn, time_block(pa, ctx, L, br, nn, nn), ts)
        e["x_window_launch"] = br.xwin()
        out.append(e)
        del br, Hc
    except ImportError:
        pass
    # the same mesh as the mesher left it -- numbered at random, NO external RCM: the library renumbers:
    # pa_csr_locality_order (reverse Cuthill-McKee by level sets, on the device) + pa_csr_create_permuted, hidden in local_to_device
    PHASE[0] = "extra: randomly numbered FEM mesh, library-side renumbering"
    try:
        nxm, nym = 1000, 800
        nn = nxm * nym
        I, J, V, rows, cols = pa.laplacian_fem((nxm, nym), (1, 1), ranks1)
        perm = np.random.default_rng(29).permutation(nn) + 1
        Ip, Jp, Vp = perm[I.items[0] - 1], perm[J.items[0] - 1], V.items[0]
        del I, J, V
        rows1 = pa.uniform_partition(ranks1, (1,), (nn,))
        Ar = pa.psparse_from_coo(pa.DebugArray([Ip]), pa.DebugArray([Jp]), pa.DebugArray([Vp]), rows1)
        b_raw = pa.local_items(Ar.matrix_partition)[0].own_own
        ms_raw = time_block(pa, ctx, L, b_raw, nn, nn)
        ctx.sync()
        t = time.perf_counter()
        Ar2 = pa.renumber_for_locality(Ar)
        ctx.sync()
        t_ren = time.perf_counter() - t
        b_ren = pa.local_items(Ar2.matrix_partition)[0].own_own
        xr = pa.pvector_from_function(lambda ind: hash_x(ind.get_local_to_global()), Ar.col_partition)
        xr2 = pa.pvector_from_function(lambda ind: hash_x(ind.get_local_to_global()), Ar2.col_partition)
        yr, yr2 = pa.pzeros(Ar.row_partition), pa.pzeros(Ar2.row_partition)
        pa.mul_(yr, Ar, xr)
        pa.mul_(yr2, Ar2, xr2)
        same = bool(np.array_equal(pa.local_items(yr.own_values())[0], pa.local_items(yr2.own_values())[0]))
        e = entry(f"Q1 FEM Laplacian on a {nxm} x {nym} mesh numbered at RANDOM, no external RCM: renumber_for_locality (device-side reverse "
                  "Cuthill-McKee, rows keep their entry order, hidden in local_to_device), pa_spmv", b_ren, nn, nn,
                  time_block(pa, ctx, L, b_ren, nn, nn), t_ren)
        e["x_window_launch"] = b_ren.xwin()
        e["band_before_after"] = [int(v) for v in pa.local_items(Ar2.bandwidths)[0]]
        e["ms_as_numbered_by_the_mesher"] = round(ms_raw, 4)
        e["algorithmic_gbps_as_numbered_by_the_mesher"] = round((b_raw.nnz * 12 + nn * 20) / ms_raw / 1e6, 1)
        e["renumbering_s"] = round(t_ren, 2)
        e["same_bits_as_the_unrenumbered_product"] = same
        out.append(e)
        del Ar, Ar2, xr, xr2, yr, yr2, b_raw, b_ren
    except Exception as ex:                                     # noqa: BLE001
        print(f"[bench] renumbering entry skipped: {ex}", file=sys.stderr)
    # BASELINE config 5 as a whole: Q1 FEM Laplacian, 8 parts as (4,2) resident on this ONE GPU, the disassembled psparse
    # route, full mul! = pack / device-to-device exchange / own x own / unpack / own x ghost of all 8 parts in one call
    PHASE[0] = "extra: config 5 on 8 parts"
    t = time.perf_counter()
    n5 = int(os.environ.get("PA_BENCH_C5_NODES", "4096"))
    ranks8 = pa.DebugArray(range(1, 9))
    fem_device = os.environ.get("PA_BENCH_FEM_DEVICE", "1") != "0"       # (the triplets generated in HBM, round 6; 0: native host threads + upload)
    I, J, V, rows, cols = pa.laplacian_fem((n5, n5), (4, 2), ranks8, device=fem_device)
    A5 = pa.psparse_disassembled(I, J, V, rows, cols)
    del I, J, V
    ts = time.perf_counter() - t
    x5 = pa.pvector_from_function(lambda ind: hash_x(ind.get_local_to_global()) * (ind.get_local_to_owner() == ind.part), A5.col_partition)
    y5 = pa.pzeros(A5.row_partition)
    ms, ms_graph, ms_spmv = whole_mul_times(pa, ctx, L, A5, x5, y5)
    blocks = pa.local_items(A5.matrix_partition)
    nnz5 = sum(b.own_own.nnz + b.own_ghost.nnz for b in blocks)
    ghosts = [c.n_ghost for c in pa.local_items(A5.col_partition)]
    rows5 = sum(r.n_own for r in pa.local_items(A5.row_partition))
    moved = sum(b.own_own.stream_bytes() + b.own_ghost.stream_bytes() for b in blocks) + 16 * rows5 + 3 * 8 * sum(ghosts)
    all_fused, finfo = fused_info(pa, ctx, L, A5, x5, y5)
    out.append({"workload": f"config 5 whole: Q1 FEM Laplacian {n5} x {n5} nodes, 8 parts (4,2) ALL on this one GPU, disassembled psparse "
                            "route, mul! = one push launch + ONE launch per part (pa_mul_all)",
                **finfo, "mul_over_spmv": round(ms / ms_spmv, 3),
                "parts": 8, "rows": int(rows5), "nnz": int(nnz5), "ghosts_per_part": ghosts, "ms_all_parts": round(ms, 4),
                "ms_per_part": round(ms / 8, 4), "ms_per_part_mul": round(ms / 8, 4), "ms_per_part_mul_hipgraph": round(ms_graph / 8, 4),
                "ms_per_part_spmv": round(ms_spmv / 8, 4), **chain_info(8, all_fused),
                "gflops": round(2.0 * nnz5 / ms / 1e6, 1), "moved_gbps": round(moved / ms / 1e6, 1),
                "encoding_own_own": blocks[0].own_own.encoding(), "encoding_own_ghost": blocks[0].own_ghost.encoding(), "setup_s": round(ts, 2),
                "triplets_generated_in_hbm": bool(fem_device)})
    del A5, x5, y5, blocks
    # the caller of the hot path that §8(f) names next: one MG-PCG iteration of the HPCG driver (4 levels, multicolour
    # Gauss-Seidel as SpMV + update, opt_cg_) at the headline's size, with the set-up it needs (everything made in HBM)
    PHASE[0] = "extra: MG-PCG 256^3"
    ctx.sync()
    t = time.perf_counter()
    S = pa.pc_setup(ranks1, 1, 4, 256, 256, 256, ordering="multicolor_spmv")
    ctx.sync()
    ts = time.perf_counter() - t
    Amg, bmg = S.A_vec[-1], S.r[-1]
    pa.opt_cg_(pa.pzeros(Amg.col_partition), Amg, bmg, maxiter=10, Pl=S, fuse=True)
    ctx.sync()
    t = time.perf_counter()
    _x, r0, r, it = pa.opt_cg_(pa.pzeros(Amg.col_partition), Amg, bmg, maxiter=30, Pl=S, fuse=True)
    ctx.sync()
    dt = (time.perf_counter() - t) / 30
    entry_mg = {"workload": "HPCG MG-PCG iteration, 27-pt 256^3, 1 part: 4-level V-cycle (multicolour Gauss-Seidel as SpMV + update, fused "
                            "restriction) + opt_cg_ (tools/hpcg_driver.py runs the three-phase benchmark around it)",
                "ms_per_iteration": round(dt * 1e3, 3), "pc_setup_s": round(ts, 2), "iterations": int(it), "residual_reduction": float(r / r0)}
    out.append(entry_mg)
    del S, Amg, bmg, _x
    try:                                                     # the same with the library's default value dictionary (what a caller gets)
        PHASE[0] = "extra: MG-PCG 256^3, value dictionary"
        os.environ.pop("PA_SPMV_VALUE_DICT")
        ctx.sync()
        t = time.perf_counter()
        S = pa.pc_setup(ranks1, 1, 4, 256, 256, 256, ordering="multicolor_spmv")
        ctx.sync()
        ts = time.perf_counter() - t
        Amg, bmg = S.A_vec[-1], S.r[-1]
        pa.opt_cg_(pa.pzeros(Amg.col_partition), Amg, bmg, maxiter=10, Pl=S, fuse=True)
        ctx.sync()
        t = time.perf_counter()
        _x, r0b, rb, itb = pa.opt_cg_(pa.pzeros(Amg.col_partition), Amg, bmg, maxiter=30, Pl=S, fuse=True)
        ctx.sync()
        entry_mg["with_default_value_dictionary"] = {"ms_per_iteration": round((time.perf_counter() - t) / 30 * 1e3, 3), "pc_setup_s": round(ts, 2),
                                                     "iterations": int(itb),
                                                     "residual_rel_diff_vs_fp64_stream": float(abs(rb - r) / max(abs(r), 1e-300)),
                                                     "note": "same products, same bits of every vector; the fused u'c of opt_cg_ runs on pattern-ELL's bit / byte "
                                                             "streams too (round 6; blocks on the row-split kernel's one-byte stream take a dot pass of their own)"}
    finally:
        os.environ["PA_SPMV_VALUE_DICT"] = "0"
    return out


def float32_entry(pa, ctx, L, host_oo, xv, y_head, n_own):
    """Beside the f64 headline, never instead of it: the same own x own block with Float32 values and Float32 vectors
    (csrc/pa_f32.hip: the fp64 path's pattern-ELL structure with a 4-byte value stream; every product and sum rounded to float in
    spmv_csr!'s order, src/sparse_utils.jl:649-669 with eltype Float32).  The f64 headline's y is the yardstick: the Float32 result
    must agree with it to Float32 accuracy."""
    PHASE[0] = "Float32 block"
    t = time.perf_counter()
    A32 = pa.DeviceCSR32(host_oo.m, host_oo.n, host_oo.rowptr, host_oo.colval, host_oo.nzval.astype(np.float32))
    ts = time.perf_counter() - t
    x32 = pa.DeviceVector32(host_oo.n).upload(xv.download()[:host_oo.n].astype(np.float32))
    y32 = pa.DeviceVector32(host_oo.m)
    pa.spmv32_(y32, A32, x32)
    got = y32.download().astype(np.float64)
    err = float(np.max(np.abs(got - y_head)) / max(1e-300, float(np.max(np.abs(y_head)))))
    spin_up(ctx, lambda: pa.spmv32_(y32, A32, x32))
    reps = 30
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    for _ in range(reps):
        pa.spmv32_(y32, A32, x32)
    e1 = ctx.event().record(L.STREAM_COMPUTE)
    ctx.sync()
    ms = e0.elapsed_ms(e1) / reps
    info = A32.info()
    nnz = len(host_oo.nzval)
    moved = 4 * info["padded_entries"] + 8 * info["slabs"] + 4 * n_own + 2 * 4 * n_own      # values + slab descriptors + row masks + x once + y once
    return {"workload": "the headline's own x own block with Float32 values, Float32 x and y (pa_spmv32)", "dtype": "f32", "nnz": int(nnz),
            "storage": "pattern-ELL structure, 4-byte value stream" if info["pattern_ell"] else "SELL-64, Float32 values + Int32 columns",
            "ms": round(ms, 4), "gflops": round(2.0 * nnz / ms / 1e6, 1), "moved_bytes_per_launch": int(moved),
            "moved_gbps": round(moved / ms / 1e6, 1), "frac_moved": round(moved / ms / 1e6 / HBM_PEAK_GBPS, 4),
            "max_rel_diff_vs_the_f64_headline": err, "agrees_to_float32_accuracy": bool(err < 5e-6), "setup_s": round(ts, 1),
            "what": "reported beside `value` (f64), never instead of it"}


def general_csr_entries(pa, ctx, L, host_oo, xv, y_head, n_own, out):
    """The headline's own x own block WITHOUT row patterns (VERDICT r02 #1b): the same host CSR uploaded with
    PA_SPMV_PATTERN=0 (16-bit windowed column stream: 10 B per stored entry) and with PA_SPMV_PATTERN=0 PA_SPMV_COL16=0
    (Int32 columns: the 12 B per entry spmv_csr! of src/sparse_utils.jl:649-669 really reads) -- what a user matrix the
    pattern detector misses gets.  ms, GFLOP/s, moved GB/s, fraction of the 8 TB/s peak, bit-identical to the headline."""
    # (PA_SPMV_PELL=0 at creation: pattern-ELL storage is tried on EVERY block since round 6 -- a slab's union of offsets decides, not
    #  the row split's pattern detector -- and would otherwise serve these two as well; they are here to time the row-split kernel
    #  on explicit column streams)
    for name, env in (("c16: PA_SPMV_PATTERN=0 PA_SPMV_PELL=0 (2-byte windowed column stream, row-split kernel)", {"PA_SPMV_PATTERN": "0", "PA_SPMV_PELL": "0"}),
                      ("c32: PA_SPMV_PATTERN=0 PA_SPMV_COL16=0 PA_SPMV_PELL=0 (Int32 columns, the reference's CSR bytes, row-split kernel)",
                       {"PA_SPMV_PATTERN": "0", "PA_SPMV_COL16": "0", "PA_SPMV_PELL": "0"})):
        PHASE[0] = "general CSR: " + name
        old = {k: os.environ.get(k) for k in env}
        os.environ.update(env)
        try:
            t = time.perf_counter()
            blk = pa.DeviceCSR(host_oo)
            ts = time.perf_counter() - t
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
        y2 = pa.DeviceVector(n_own, 0)
        pa.spmv_(y2, blk, xv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0)
        ctx.sync()
        same = bool(np.array_equal(y2.download(), y_head))
        spin_up(ctx, lambda: pa.spmv_(y2, blk, xv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0))
        reps = 30
        e0 = ctx.event().record(L.STREAM_COMPUTE)
        for _ in range(reps):
            pa.spmv_(y2, blk, xv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0)
        e1 = ctx.event().record(L.STREAM_COMPUTE)
        ctx.sync()
        ms = e0.elapsed_ms(e1) / reps
        moved = blk.stream_bytes() + 2 * 8 * n_own
        alg = blk.nnz * 12 + (n_own + 1) * 4 + 2 * 8 * n_own
        out.append({"workload": "the headline's own x own block, " + name, "nnz": int(blk.nnz), "ms": round(ms, 4),
                    "gflops": round(2.0 * blk.nnz / ms / 1e6, 1), "moved_bytes_per_launch": int(moved),
                    "moved_gbps": round(moved / ms / 1e6, 1), "frac_moved": round(moved / ms / 1e6 / HBM_PEAK_GBPS, 4),
                    "algorithmic_gbps": round(alg / ms / 1e6, 1), "frac_spec_8d": round(alg / ms / 1e6 / HBM_PEAK_GBPS, 4),
                    "kernel": kernel_of(blk), "bit_identical_to_headline_product": same,
                    "encoding": blk.encoding(), "setup_s": round(ts, 1)})
        del blk, y2
    return out


def main():
    args = parse()
    # ranks sharing a GPU (the test mode: never a production run): the fused launch's tail blocks may spin on their neighbours'
    # arrival flags, and eight ranks' worth of spinners must not fill the one GPU the neighbours' pushing blocks have to get onto
    try:
        import torch as _t
        if int(os.environ.get("WORLD_SIZE", "1")) > max(1, _t.cuda.device_count()):
            os.environ.setdefault("PA_FUSED_TAIL_BLOCKS", "32")
    except Exception:                                           # noqa: BLE001
        pass
    N = args.gpus
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    import torch.distributed as dist
    from __graft_entry__ import load_package
    if N > 1 or world > 1:
        watchdog(float(os.environ.get("PA_BENCH_WATCHDOG_S", "1500")))
        assert world == N, f"--gpus {N} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {N}"
        local = int(os.environ.get("LOCAL_RANK", str(rank))) % max(1, torch.cuda.device_count())
        torch.cuda.set_device(local)
        backend = os.environ.get("PA_BENCH_BACKEND", "cpu:gloo,cuda:nccl")   # "gloo" + PA_TRANSPORT=host: ranks may share a GPU
        if "nccl" in backend:
            dist.init_process_group(backend, rank=rank, world_size=world, device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    # `value` and every standard entry of the line stay on the fp64 value stream: the library's value dictionary --
    # on by default for big blocks since round 4, lossless -- is switched off here and measured beside them, labelled
    vdict_env = os.environ.get("PA_SPMV_VALUE_DICT")
    os.environ["PA_SPMV_VALUE_DICT"] = "0" if vdict_env is None else vdict_env
    # N > 1 (VERDICT r05 "Next" #1a): the parity gate and the first timed loop run on the CONSERVATIVE product -- separate launches,
    # stream order and events around the RCCL group (PA_MUL_FUSED=0: round 4's chain, the `north_star` design) -- and rank 0 holds a
    # complete line from then on; the library's default (one launch per part, the exchange waited for INSIDE the launch) is gated and
    # timed behind it as a section that can fail or hang without costing the line.  `value` = the faster of the two that passed its
    # gate; config.product_path says which.  An in-launch wait gives up after 10 s here (the library's default is 30).
    want_fused = (N > 1 or world > 1) and os.environ.get("PA_MUL_FUSED", "1") != "0"
    if N > 1 or world > 1:
        os.environ.setdefault("PA_IPC_TIMEOUT_S", "10")
        os.environ["PA_MUL_FUSED"] = "0"
    pa = load_package()
    ctx = pa.context()
    import pa_amd._lib as L

    n = args.n
    npx, npy, npz = pa.compute_optimal_shape_XYZ(N)
    gn = (npx * n, npy * n, npz * n)
    transport = "none(1 part)"
    rccl_ranks_seen = None
    if N > 1:
        import pa_amd.p_vector as pv
        transport = os.environ.get("PA_TRANSPORT", "rccl")
        allow_fallback = os.environ.get("PA_ALLOW_TRANSPORT_FALLBACK", "0") == "1"
        if transport == "rccl":
            PHASE[0] = "RCCL communicator"
            err = ""
            try:                       # direct RCCL (ncclSend/ncclRecv issued by libpa_hip on its comm stream)
                comm = pa.init_comm()
                rccl_ranks_seen = comm.info()["nranks"]
                good = 1 if rccl_ranks_seen == N else 0
                if not good:
                    err = f"the communicator reports {rccl_ranks_seen} ranks, expected {N}"
            except Exception as e:     # noqa: BLE001
                good, err = 0, str(e)
            if not good:
                print(f"[rank {rank}] direct RCCL communicator failed: {err}", file=sys.stderr, flush=True)
            flag = torch.tensor([good])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if not flag.item():
                # the RCCL neighbour exchange IS what an N > 1 line measures: a silent downgrade would report another
                # path under the same name.  PA_ALLOW_TRANSPORT_FALLBACK=1 lets the run continue on torch.distributed p2p,
                # labelled as such in config.transport.
                if not allow_fallback:
                    print("[bench] RCCL transport unavailable on at least one rank; refusing to downgrade "
                          "(PA_ALLOW_TRANSPORT_FALLBACK=1 overrides)", file=sys.stderr, flush=True)
                    dist.barrier()
                    os._exit(3)
                transport = "torch"
        pv.TRANSPORT = transport
        ranks = pa.with_torchdist(lambda distribute: distribute(range(1, N + 1)))
    else:
        ranks = pa.DebugArray([1])

    PHASE[0] = "matrix set-up"
    t_setup = time.perf_counter()
    want_cpu = not args.no_cpu_baseline
    A, b = pa.build_p_matrix(ranks, n, n, n, *gn, npx, npy, npz)      # (no host copy: own|own and b are generated in HBM)

    # x[gid] = hash on OWN entries only: mul! must bring the ghosts
    def xfun(ind):
        v = hash_x(ind.get_local_to_global())
        v[ind.n_own:] = 0.0
        return v
    x = pa.pvector_from_function(xfun, A.col_partition)
    y = pa.pzeros(A.row_partition)
    ctx.sync()
    t_setup = time.perf_counter() - t_setup

    # ---- parity gate before any timing counts: A*1 == b bit-exactly; ghosts == owner values
    def gate():
        pa.mul_(y, A, pa.pones(A.col_partition))
        ok = all(np.array_equal(g, e) for g, e in zip(pa.local_items(y.own_values()), pa.local_items(b.own_values())))
        x.vector_partition = pa.pmap(lambda v, ind: v.upload(xfun(ind)), x.vector_partition, A.col_partition)
        pa.mul_(y, A, x)
        for vals, ind in zip(pa.local_items(x.local_values()), pa.local_items(A.col_partition)):
            ok = ok and np.array_equal(vals, hash_x(ind.get_local_to_global()))
        if N > 1:
            flag = torch.tensor([1 if ok else 0])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item())
        return ok

    PHASE[0] = f"parity gate (transport {transport})"
    if not gate():
        print(f"[bench] parity gate failed (transport {transport}): A*1 != b or ghost values differ from their owners",
              file=sys.stderr, flush=True)
        if N > 1:
            dist.barrier()
        os._exit(5)

    blk = pa.local_items(A.matrix_partition)[0]
    ind = pa.local_items(A.col_partition)[0]
    nnz_oo, nnz_oh, n_own, n_ghost = blk.own_own.nnz, blk.own_ghost.nnz, ind.n_own, ind.n_ghost
    xv, yv = pa.local_items(x.vector_partition)[0], pa.local_items(y.vector_partition)[0]

    def step(overlap=True, ev=None):
        # mul!(c,a,b): src/p_sparse_matrix.jl:2098-2101.  The timed loops queue it with ONE library call per step (pa_mul5 /
        # pa_mul_all; overlap off: pa_mul_no_lat), so that the host's share of a step -- seven neighbours' worth of
        # send/recv bookkeeping on an 8-GPU run -- is C, not Python; the event pass composes the same calls itself to put
        # its events around own x own.  Same kernels, same order, same bits.
        if ev is None:
            if overlap:
                pa.mul_c_(y, A, x)
            else:
                pa.mul_no_lat_c_(y, A, x)
            return
        t = pa.consistent_(x)
        if not overlap:
            t.wait()                     # mul_no_lat! (HPCG/src/hpcg_utils.jl:6-17): exchange first, then the products
        if ev:
            ev[0].record(L.STREAM_COMPUTE)
        pa.spmv_(yv, blk.own_own, xv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0)
        if ev:
            ev[1].record(L.STREAM_COMPUTE)
        if overlap:
            t.wait()
        pa.spmv_(yv, blk.own_ghost, xv, L.SEG_GHOST, L.SEG_OWN, 1.0, 1.0)

    def barrier():
        ctx.sync()
        if N > 1:
            dist.barrier()
            ctx.sync()

    def timed(steps, overlap):
        """EXACTLY `steps` steps between barrier + synchronize on both sides; max over ranks."""
        barrier()
        m0 = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
        t0 = time.perf_counter()
        ea = ctx.event().record(L.STREAM_COMPUTE)       # (one HIP event before the first step, one behind the last: nothing inside)
        for _ in range(steps):
            step(overlap)
        eb = ctx.event().record(L.STREAM_COMPUTE)
        ctx.sync()
        barrier()
        dt = time.perf_counter() - t0
        m1 = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
        timed.device_ms = ea.elapsed_ms(eb)
        timed.local_s = dt                              # this rank's own wall clock (the line's per_rank section)
        if N > 1:
            tt = torch.tensor([dt], dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        return dt, m0, m1

    overlap_on = not args.no_overlap
    # ---- clock ramp, before the W warm-up steps and outside every timed region: after the host-side parity gate the GPU has
    # been idle for seconds and needs ~40 launches (30 ms) of this kernel to be back at its working clocks -- the first
    # launch after the idle takes 0.83 ms, the 40th 0.68 (rocprofv3 trace of round 2, profiles/r02_summary.json).  W = 5
    # warm-up steps do not cover that; a fixed 150 steps do, the same number on every rank.
    # Round 4: the ramp has TWO plateaus (27-pt 256^3: 0.80 -> 0.72 -> 0.675 ms per step) and how long the GPU sits on the middle
    # one differs from lease to lease -- 50 steps on most boxes, more than the 150 of round 3 on others (the driver's round-3 line,
    # 0.7202 ms, and one of this round's leases were timed there; docs/LAB_NOTEBOOK.md R4.8).  The ramp is now ONE SECOND of steps
    # queued back to back (no host synchronisation inside: an idle stream is what lets the clocks fall), the same number on every rank.
    PHASE[0] = "clock ramp"
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    for _ in range(10):
        step(overlap_on)
    e1 = ctx.event().record(L.STREAM_COMPUTE)
    ctx.sync()
    ramp = [e0.elapsed_ms(e1) / 10]
    n_groups = int(min(400, max(14, np.ceil(float(os.environ.get("PA_BENCH_RAMP_S", "1.0")) * 1e3 / (10 * max(ramp[0], 1e-3))))))
    if N > 1:
        tg = torch.tensor([n_groups], dtype=torch.int64)
        dist.all_reduce(tg, op=dist.ReduceOp.MAX)
        n_groups = int(tg.item())
    gev = [ctx.event().record(L.STREAM_COMPUTE)]
    for _ in range(n_groups):
        for _ in range(10):
            step(overlap_on)
        gev.append(ctx.event().record(L.STREAM_COMPUTE))
    ctx.sync()
    ramp += [gev[k].elapsed_ms(gev[k + 1]) / 10 for k in range(n_groups)]
    ms_after_fixed_ramp = float(np.mean(ramp[-2:]))     # the last 20 steps of the FIXED ramp: no selection rule has acted yet
    # One part, no neighbours (the N = 1 headline): a step IS one launch of the product kernel, which on every lease seen so far
    # settles at 0.90-0.91 of what the box's own two-stream read kernel streams (roofline.frac_vs_this_box_read) and sits at 0.85-0.86
    # while the GPU is still on the middle plateau.  Below 0.875 after the ramp the warm-up goes on, a quarter of a second at a
    # time, for at most two more seconds -- and the line says so.
    box_early, ramp_extended_s, ramp_frac_end = None, 0.0, None
    if N == 1 and nnz_oh == 0 and nnz_oo > 0 and os.environ.get("PA_BENCH_RAMP_EXTEND", "1") != "0":
        try:
            box_early = calibrate_box(pa, ctx, L)
            moved_chk = blk.own_own.stream_bytes() + 16 * n_own

            def more_groups(k):
                ev_ = [ctx.event().record(L.STREAM_COMPUTE)]
                for _ in range(k):
                    for _ in range(10):
                        step(overlap_on)
                    ev_.append(ctx.event().record(L.STREAM_COMPUTE))
                ctx.sync()
                return [ev_[j].elapsed_ms(ev_[j + 1]) / 10 for j in range(k)]
            ramp += more_groups(20)                       # (the calibration kernels ran in between: back onto the product first)
            frac_now = lambda: moved_chk / (float(np.median(ramp[-10:])) * 1e-3) / 1e9 / box_early["read_gbps"]   # noqa: E731
            per_quarter = max(10, int(np.ceil(250.0 / (10 * max(ramp[-1], 1e-3)))))
            while frac_now() < 0.875 and ramp_extended_s < 2.0:
                ramp += more_groups(per_quarter)
                ramp_extended_s += 0.25
            ramp_frac_end = round(frac_now(), 4)
        except Exception as e:                            # noqa: BLE001  (a diagnostic: never costs the run its line)
            print(f"[bench] ramp extension skipped: {e}", file=sys.stderr, flush=True)
    # ---- placement A/B with the product kernel itself (VERDICT r03 #1a), at working clocks, outside every timed region: y
    # where the arena's rule put it, in every other memory class the held extents have room in (the matrix streams' own class is
    # the control that should lose ~13 %) and in a plain hipMalloc; y moves when another place is > 1.5 % faster
    placement_ab = None
    if os.environ.get("PA_BENCH_PLACEMENT_AB", "1") != "0" and nnz_oo > 0:
        PHASE[0] = "placement A/B"
        try:
            placement_ab = pa.tune_output_placement(blk.own_own, xv, yv, L.SEG_OWN, reps=10, rounds=3)
            pa.mul_(y, A, x)                      # (the A/B leaves y = A_oo * x_own; the full product again before anything reads y)
        except Exception as e:                    # noqa: BLE001  (a diagnostic: never costs the run its line)
            print(f"[bench rank {rank}] placement A/B skipped: {e}", file=sys.stderr, flush=True)
    # (a box of round 4 ran the ramp and the A/B at 0.7215 ms per step and the timed region, seconds later, at 0.675: the same
    # ten-step groups once more behind the A/B say on which side of it the change happened)
    ramp_after_ab = []
    for _ in range(3):
        e0 = ctx.event().record(L.STREAM_COMPUTE)
        for _ in range(10):
            step(overlap_on)
        e1 = ctx.event().record(L.STREAM_COMPUTE)
        ctx.sync()
        ramp_after_ab.append(e0.elapsed_ms(e1) / 10)
    # clocks / power / partition modes under load, before and after the timed region (sysfs of this context's GPU)
    def telemetry_under_load():
        for _ in range(30):
            step(overlap_on)
        t = ctx.telemetry()                       # (read while the 30 queued steps keep the GPU busy)
        ctx.sync()
        return t
    PHASE[0] = "telemetry"
    try:
        telemetry = {"before_timed_region": telemetry_under_load()}
    except Exception as e:                        # noqa: BLE001
        telemetry = {"error": str(e)}
    PHASE[0] = f"warm-up (transport {transport})"
    for _ in range(args.warmup):
        step(overlap_on)
    PHASE[0] = f"timed mul! loop (transport {transport})"
    dt, mono0, mono1 = timed(args.steps, overlap_on)
    local_ms_headline = timed.local_s / args.steps * 1e3
    try:
        telemetry["after_timed_region"] = telemetry_under_load()
        telemetry["what"] = ("sysfs of this context's GPU (/sys/bus/pci/devices/<pci id>/: pp_dpm_*, hwmon power1_input / power1_cap / "
                             "freq*_input / temp*_input, current_*_partition), read while 30 queued steps keep the GPU busy, right before the W "
                             "warm-up steps and right after the timed region")
    except Exception as e:                        # noqa: BLE001
        telemetry["error"] = str(e)
    ms_per_step = dt / args.steps * 1e3
    device_ms_per_step = timed.device_ms / args.steps       # HIP events around the whole timed region, compute stream

    # ---- second pass, outside the headline's timed region: HIP events (compute stream) around own x own and around the
    # whole step, every launch on its own; median and mean of max(50, steps) launches
    PHASE[0] = "kernel-event pass"
    K2 = max(50, args.steps)
    evs = [[ctx.event() for _ in range(2)] for _ in range(K2 + 1)]
    barrier()
    for k in range(K2 + 1):                     # (two event records per step and no more: each one costs the stream ~10 us)
        step(overlap_on, ev=evs[k])
    ctx.sync()
    barrier()
    kern = np.array([e[0].elapsed_ms(e[1]) for e in evs[:K2]])
    whole = np.array([evs[k][0].elapsed_ms(evs[k + 1][0]) for k in range(K2)])     # start of own x own to the next one: a step
    kern_ms_events, kern_med = float(kern.mean()), float(np.median(kern))
    # The dominant kernel's average launch duration over the timed region: with one part a step IS one launch of it (no
    # neighbours: no pack, no unpack, an empty own x ghost block), so the HIP events around the timed region give it without
    # putting anything between the launches; with several parts a step has more kernels, and the event pass's per-launch
    # figure is what there is (each of its launches sits behind an event record: ~10 us of idle stream and colder caches).
    kern_ms = device_ms_per_step if (N == 1 and nnz_oh == 0) else kern_ms_events

    nnz = nnz_oo + nnz_oh
    if N > 1:
        tot = torch.tensor([nnz], dtype=torch.int64)
        dist.all_reduce(tot)
        nnz_total = int(tot.item())
    else:
        nnz_total = nnz
    flops_total = 2.0 * nnz_total
    value = flops_total / (ms_per_step * 1e-3) / 1e9

    # algorithmic bytes: whole mul! per part, and the dominant kernel's share; moved bytes: what the
    # block's encoding makes the kernel read (pa_csr_stream_bytes) + x once + y once
    bytes_mul = nnz * 12 + (n_own + 1) * 4 + (n_own + n_ghost) * 8 + n_own * 8
    bytes_oo = nnz_oo * 12 + (n_own + 1) * 4 + n_own * 8 + n_own * 8
    moved_oo = blk.own_own.stream_bytes() + n_own * 8 + n_own * 8
    ach = bytes_oo / (kern_ms * 1e-3) / 1e9
    ach_moved = moved_oo / (kern_ms * 1e-3) / 1e9

    # measured HBM traffic of the dominant kernel: PMC passes are separate rocprofv3 runs (never inside a timed
    # run); their per-launch summary is committed under profiles/ and quoted here (null when absent).
    traffic, traffic_src = None, None
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_summary.json")))
        for cand in reversed(cands if n == 256 else []):          # the newest round's summary that holds the counters
            pml = json.load(open(cand)).get("pmc_per_launch", {})
            pm = pml.get("k_spmv_pell") or pml.get("k_spmv_rowsplit")
            if pm and "fetch_bytes_gfx950_corrected" in pm and "write_bytes" in pm:
                traffic = round(pm["fetch_bytes_gfx950_corrected"] + pm["write_bytes"])
                traffic_src = (os.path.relpath(cand, ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE of a separate 1-GPU 256^3 "
                               "run of this command, not of this process)")
                break
    except Exception:
        pass

    if N == 1 and rank == 0 and args.live_traffic and not args.traffic_child:
        PHASE[0] = "PMC child passes"
        t_pmc = time.perf_counter()
        lt, lsrc = live_traffic(n)
        if lt is not None:
            traffic, traffic_src = lt, lsrc + f"; {time.perf_counter() - t_pmc:.0f} s"
        else:
            traffic_src = (traffic_src or "none") + f" [live PMC passes unavailable: {lsrc}]"
    box = box_early
    if rank == 0 and box is None:
        try:
            box = calibrate_box(pa, ctx, L)
        except Exception as e:                                   # noqa: BLE001  (an extra; no collectives inside)
            print(f"[bench] HBM calibration skipped: {e}", file=sys.stderr)

    if rank == 0:
        prio = ctx.stream_priorities()
        out = {
            "metric": "HPCG 27-pt SpMV GFLOP/s + achieved HBM GB/s per GPU",
            "value": round(value, 2), "unit": "GFLOP/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "warmup_effective": args.warmup + 10 * len(ramp) + 10 * len(ramp_after_ab),
            #  the unconditioned figures beside `value`: the 20 steps that END the fixed one-second ramp -- before the
            # N = 1 extension rule (clock_ramp.extension_rule) can have acted -- and how long that rule then kept warming up
            "value_after_fixed_ramp": round(flops_total / (ms_after_fixed_ramp * 1e-3) / 1e9, 2),
            "ms_per_step_after_fixed_ramp": round(ms_after_fixed_ramp, 4),
            "clock_ramp_extended_s": ramp_extended_s,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"HPCG 27-pt stencil {n}^3 rows per part, {N} part(s) as ({npx},{npy},{npz}), "
                                   "mul! = consistent!(pack+exchange+unpack) " + ("overlapped with own*own, then own*ghost" if overlap_on else
                                                                                   "completed BEFORE own*own (--no-overlap), then own*ghost"),
                       "rows_per_part": n_own, "nnz_per_part": nnz, "nnz_own_own": nnz_oo, "nnz_own_ghost": nnz_oh,
                       "ghosts_per_part": n_ghost, "index_type": "Int32",
                       "transport": {"rccl": "rccl-p2p (ncclSend/ncclRecv group on the comm stream)",
                                     "torch": "torch.distributed p2p (FALLBACK, PA_ALLOW_TRANSPORT_FALLBACK=1: not the RCCL row)",
                                     "host": "host-staged gloo (test transport: ranks may share a GPU)",
                                     "ipc": "ipc push (PA_TRANSPORT=ipc: the pack kernel stores into the neighbours' hipIpc-mapped receive "
                                            "buffers, csrc/pa_push.hip; ranks may share a GPU)"}.get(transport, transport),
                       "rccl_ranks_seen": rccl_ranks_seen, "overlap": overlap_on,
                       "product_path": ("one part, no neighbours: a step is one launch of the product kernel" if N == 1 else
                                        "separate launches (PA_MUL_FUSED=0: pack + exchange on the comm stream | own x own, own x ghost from the "
                                        "receive buffer, unpack -- stream order and events)"),
                       "stream_priority": prio},
            "gflops_per_gpu": round(value / N, 2),
            "hbm_gbps_per_gpu_algorithmic": round(bytes_mul / (ms_per_step * 1e-3) / 1e9, 1),
            "ms_per_step_median_events": round(float(np.median(whole)), 4),
            "roofline": {"bound": "hbm", "kernel": "own x own on " + kernel_of(blk.own_own) + "; column encoding of the row-split chunks: "
                                                      + json.dumps(blk.own_own.encoding()),
                         # `achieved` / `frac`: the bytes this kernel must MOVE per launch (values + row pointers + chunk table +
                         # descriptors / kept column streams, pa_csr_stream_bytes, + x once + y once) over its average launch time:
                         # a physical fraction, <= 1.  Reproduce: moved_bytes_per_launch / avg_launch_ms / 1e6 / peak.
                         "achieved": round(ach_moved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach_moved / HBM_PEAK_GBPS, 4),
                         "moved_bytes_per_launch": int(moved_oo),
                         "traffic": traffic, "traffic_source": traffic_src,
                         "avg_launch_ms": round(kern_ms, 4),
                         "avg_launch_ms_source": ("HIP events around the timed region / steps (one part: a step is one launch)"
                                                  if (N == 1 and nnz_oh == 0) else "event pass: HIP events around every own x own launch"),
                         "event_pass": {"avg_launch_ms": round(kern_ms_events, 4), "median_launch_ms": round(kern_med, 4),
                                        "launches": int(K2)},
                         "median_launch_ms": round(kern_med, 4), "launches_timed": int(args.steps if (N == 1 and nnz_oh == 0) else K2),
                         # the SURVEY 8(d) accounting beside it: the reference's CSR bytes (12 B per stored entry + 20 B per row).
                         # It exceeds 1 on this matrix because row patterns regenerate the columns: 4 of those 12 bytes per entry are
                         # never read (all 2*nnz flops are done, all fp64 values are read, y is bit-identical).
                         "algorithmic_bytes_per_launch": bytes_oo, "achieved_algorithmic_csr": round(ach, 1),
                         "frac_algorithmic_csr": round(ach / HBM_PEAK_GBPS, 4),
                         #  the three accountings side by side: SURVEY 8(d)'s algorithmic bytes (can exceed 1: the 4-byte column
                         # of every entry is regenerated from the slab's pattern, never read), the bytes the kernel must move (`frac`), and the
                         # bytes the HBM counters saw
                         "frac_spec_8d": round(ach / HBM_PEAK_GBPS, 4),
                         "frac_spec_8d_note": "SURVEY 8(d) bytes (12 B per stored entry + 20 B per row) over the launch time; above 1 is not skipped "
                                              "work -- columns are recomputed from row patterns, so 4 of those 12 bytes per entry are never read",
                         "frac_counter": (round(traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if traffic else None),
                         "frac_vs_this_box_read": (round(ach_moved / box["read_gbps"], 4) if box else None),
                         "frac_back_to_back_wall_clock": (round(moved_oo / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4) if N == 1 else None),
                         "what": "`achieved`/`frac`: bytes the kernel must move (no column stream where a row pattern describes the "
                                 "chunk) over the kernel's average launch time, against the 8 TB/s spec peak; `*_algorithmic_csr`: the "
                                 "reference's CSR bytes over the same time (can exceed 1, see DESIGN.md section 3); "
                                 "`frac_back_to_back_wall_clock` uses the host's wall clock (ms_per_step) instead of the device's events; "
                                 "`general_csr` (below) is the same block with explicit column streams",
                         "timed_region_monotonic_ns": [mono0, mono1],
                         "this_box": box,
                         "placement_ab": placement_ab,
                         "telemetry": telemetry,
                         "memory_classes": {"arena": ctx.arena(), "value_stream": blk.own_own.memory_class(),
                                            "x": xv.memory_class(), "y": yv.memory_class(),
                                            "what": "csrc/pa_arena.hip: matrix streams and vectors live in different memory classes "
                                                    "of one contiguous arena (a write stream in its read stream's class costs 13-15 %)"}},
            "parity_gate": "A*1==b bit-exact; ghosts==owners bit-exact",
            "cold": {"ms_per_step_first_10": round(ramp[0], 4), "gflops_first_10": round(flops_total / (ramp[0] * 1e-3) / 1e9, 1),
                     "what": "the first 10 steps after the host-side parity gate (GPU idle for seconds): what a solver that calls mul! "
                             "right after a host-side pause sees; `value` is the steady state after `warmup_effective` steps"},
            "clock_ramp": {"steps": 10 * len(ramp), "first_10_ms_per_step": round(ramp[0], 4), "last_10_ms_per_step": round(ramp[-1], 4),
                           "ms_per_step_by_10": [round(v, 4) for v in (ramp if len(ramp) <= 30 else ramp[:15] + ramp[15:-5:10] + ramp[-5:])],
                           "ms_per_step_by_10_is": "every group of ten steps" if len(ramp) <= 30 else "the first 15 groups of ten steps, every tenth group after them, the last 5",
                           "ms_per_step_by_10_after_placement_ab": [round(v, 4) for v in ramp_after_ab],
                           "what": "untimed steps run before the W warm-up steps (they ARE warm-up: warmup_effective counts them): the "
                                   "GPU idles during the host-side parity gate and comes back to its working clocks over two plateaus; how "
                                   "long it stays on the second differs between leases, so the ramp is one second of steps queued back to back "
                                   "(PA_BENCH_RAMP_S)",
                           "plateau_ms_per_step": {"min": round(min(ramp), 4), "median_last_20_groups": round(float(np.median(ramp[-20:])), 4)},
                           "extended_s": ramp_extended_s, "frac_vs_this_box_read_at_its_end": ramp_frac_end,
                           "extension_rule": "N = 1 only: while the product moves less than 0.875 of this box's two-stream read rate "
                                             "(calibrate_box) the warm-up goes on in quarters of a second, two seconds at most"},
            "setup_s": round(t_setup, 1),
        }
        LINE[0] = out

    # ---------------- everything below is optional: it adds to the line, it can never take the line away ----------------
    # First of all the library's DEFAULT product for one part per process: mul! as ONE launch with the exchange waited for inside it
    # (csrc/pa_fused.hip).  Its own parity gate, its own warm-up, the same timed loop; it becomes `value` when it passed its gate and
    # was faster than the separate launches measured above.  A time-out of an in-launch wait surfaces as an exception from ctx.sync()
    # (the library reports it once and the handle continues on separate launches), a hang ends with the section's timer: either way
    # the line above goes out.
    if N > 1 and want_fused:
        with optional_section("fused product", 180, N, rank):
            os.environ.pop("PA_MUL_FUSED", None)
            os.environ["PA_MUL_FUSED_RCCL"] = "1"        # (over RCCL the one launch is opt-in since round 6: here it is opted into, behind its gate)
            ctx.reload_env()
            ms_sep = ms_per_step
            ok_fused, t_fused, fl0, fl1 = False, None, ctx.fused_launches(), None
            try:
                ok_fused = gate()
                if ok_fused:
                    for _ in range(5):
                        step(overlap_on)
                    ctx.sync()
                    fl1 = ctx.fused_launches()
                    for _ in range(30 + args.warmup):
                        step(overlap_on)
                    t_fused, _, _ = timed(args.steps, overlap_on)
                    ok_fused = gate()                    # (and still right after the loop)
            finally:
                ms_fused = None if t_fused is None else t_fused / args.steps * 1e3
                use_fused = bool(ok_fused and ms_fused is not None and ms_fused < ms_sep)
                if not use_fused:                        # the rest of the run stays on the path `value` was measured on
                    os.environ["PA_MUL_FUSED"] = "0"
                    os.environ.pop("PA_MUL_FUSED_RCCL", None)
                    ctx.reload_env()
            if use_fused:
                ms_per_step = ms_fused
                value = flops_total / (ms_per_step * 1e-3) / 1e9
            if rank == 0:
                o = LINE[0]
                o["fused_ab"] = {"mul_as_one_launch_rank0": bool(fl1 is not None and fl1[0] - fl0[0] >= 5),
                                 "exchange_inside_the_launch_rank0": bool(fl1 is not None and fl1[1] - fl0[1] >= 5),
                                 "parity_gate_one_launch": bool(ok_fused),
                                 "ms_per_step_one_launch": None if ms_fused is None else round(ms_fused, 4),
                                 "ms_per_step_separate_launches": round(ms_sep, 4),
                                 "value_uses": "one launch" if use_fused else "separate launches",
                                 "what": "the separate launches (PA_MUL_FUSED=0) are gated and timed FIRST; the library's default -- one launch "
                                         "per part, the exchange waited for inside it (csrc/pa_fused.hip) -- behind them by the same loop; "
                                         "`value` is the faster one that passed its parity gate"}
                if use_fused:
                    o["value"] = round(value, 2)
                    o["ms_per_step"] = round(ms_per_step, 4)
                    o["gflops_per_gpu"] = round(value / N, 2)
                    o["hbm_gbps_per_gpu_algorithmic"] = round(bytes_mul / (ms_per_step * 1e-3) / 1e9, 1)
                    o["config"]["product_path"] = ("one launch per part, the exchange inside it (the library's default; csrc/pa_fused.hip) -- "
                                                   "gated and timed behind the separate launches, see fused_ab")
    overlap = None
    if N > 1:
        with optional_section("overlap on/off comparison", 120, N, rank):
            t_other, _, _ = timed(args.steps, not overlap_on)
            on, off = (ms_per_step, t_other / args.steps * 1e3) if overlap_on else (t_other / args.steps * 1e3, ms_per_step)
            overlap = {"ms_per_step_on": round(on, 4), "ms_per_step_off": round(off, 4), "headline_uses": "on" if overlap_on else "off",
                       "what": "mul! with the ghost exchange under own x own (src/p_sparse_matrix.jl:2098-2100) vs exchange first "
                               "(HPCG mul_no_lat!), same steps, same barriers"}
            if rank == 0:
                LINE[0]["overlap"] = overlap
            local_ms_other = timed.local_s / args.steps * 1e3
            # every rank's own view, so that the first real multi-GPU record diagnoses itself: device,
            # communicator size seen, neighbours, what its arena holds, its own ms per step with the exchange under own x own
            # and with the exchange first, its own x own launch time
            ar = ctx.arena()
            cache = ind.cache or {}
            mine = torch.tensor([rank, torch.cuda.current_device(), -1 if rccl_ranks_seen is None else rccl_ranks_seen,
                                 len(cache.get("neighbors_snd", ())), len(cache.get("neighbors_rcv", ())), n_ghost, ar["gib"], ar["used_gib"],
                                 local_ms_headline if overlap_on else local_ms_other, local_ms_other if overlap_on else local_ms_headline,
                                 kern_ms_events], dtype=torch.float64)
            # (through the rendezvous store, not a collective: a rank that hangs or died costs ITS row, not the section -- rank 0
            # waits twenty seconds per row and marks the ones that never came)
            rows = []
            try:
                import datetime
                store = dist.distributed_c10d._get_default_store()
                store.set(f"pa_bench_row_{rank}", json.dumps(mine.tolist()))
                if rank == 0:
                    for r_ in range(N):
                        try:
                            store.wait([f"pa_bench_row_{r_}"], datetime.timedelta(seconds=20))
                            rows.append(torch.tensor(json.loads(store.get(f"pa_bench_row_{r_}")), dtype=torch.float64))
                        except Exception:                      # noqa: BLE001
                            miss = torch.full_like(mine, float("nan")); miss[0] = r_
                            rows.append(miss)
            except Exception as e:                             # noqa: BLE001  (no store: the collective, as before)
                print(f"[bench rank {rank}] per-rank rows over the store failed ({e}): all_gather", file=sys.stderr, flush=True)
                rows = [torch.zeros_like(mine) for _ in range(N)]
                dist.all_gather(rows, mine)
            if rank == 0:
                keys = ("rank", "cuda_device", "rccl_ranks_seen", "neighbors_snd", "neighbors_rcv", "ghosts", "arena_held_gib", "arena_used_gib",
                        "ms_per_step_overlap_on", "ms_per_step_overlap_off", "own_own_launch_ms")
                LINE[0]["per_rank"] = [({"rank": int(r[0]), "missing": True} if bool(torch.isnan(r[1])) else
                                        {k: (int(v) if k in keys[:6] else round(float(v), 4)) for k, v in zip(keys, r.tolist())}) for r in rows]
                LINE[0]["per_rank_transport"] = transport

    # ---- optional mode, reported beside the headline and never part of `value`: the same product with the lossless value
    # dictionary (PA_SPMV_VALUE_DICT=1: one byte per stored entry instead of eight when a block has <= 64 distinct values)
    vdict = None
    if N == 1 and args.value_dict:
        try:                                                   # an optional extra never costs the headline its line
            PHASE[0] = "value-dictionary mode"
            os.environ.pop("PA_SPMV_VALUE_DICT")              # the library's default (auto: big blocks with <= 64 distinct values)
            A2, _b2 = pa.build_p_matrix(ranks, n, n, n, *gn, npx, npy, npz)
            os.environ["PA_SPMV_VALUE_DICT"] = "0"
            blk2 = pa.local_items(A2.matrix_partition)[0]
            y2 = pa.pzeros(A2.row_partition)
            pa.mul_(y2, A2, x)
            same = all(np.array_equal(a_, b_) for a_, b_ in zip(pa.local_items(y2.own_values()), pa.local_items(y.own_values())))
            y2v = pa.local_items(y2.vector_partition)[0]
            spin_up(ctx, lambda: pa.spmv_(y2v, blk2.own_own, xv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0))
            for _ in range(args.warmup):
                pa.spmv_(y2v, blk2.own_own, xv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0)
            e0 = ctx.event().record(L.STREAM_COMPUTE)
            for _ in range(args.steps):
                pa.spmv_(y2v, blk2.own_own, xv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0)
            e1 = ctx.event().record(L.STREAM_COMPUTE)
            ctx.sync()
            ms2 = e0.elapsed_ms(e1) / args.steps
            moved2 = blk2.own_own.stream_bytes() + 16 * n_own
            vdict = {"kernel": kernel_of(blk2.own_own), "moved_bytes_per_launch": int(moved2),
                     "moved_gbps": round(moved2 / (ms2 * 1e-3) / 1e9, 1), "frac_moved": round(moved2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                     "frac_spec_8d": round(bytes_oo / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                     "what": "own x own SpMV with the library's DEFAULT value dictionary (lossless, built on the device for big blocks with <= 64 "
                             "distinct values; bench.py switches it off -- PA_SPMV_VALUE_DICT=0 -- for `value` and every other entry)",
                     "distinct_values": blk2.own_own.value_dict(), "bit_identical_to_headline_product": bool(same),
                     "avg_launch_ms": round(ms2, 4), "gflops": round(2.0 * nnz_oo / (ms2 * 1e-3) / 1e9, 1),
                     "algorithmic_gbps": round(bytes_oo / (ms2 * 1e-3) / 1e9, 1),
                     # fp64 adds and multiplies kept apart (-ffp-contract=off: the reference's roundings): 2 vector instructions per entry
                     "frac_of_unfused_fp64_vector_peak": round(2.0 * nnz_oo / (ms2 * 1e-3) / 1e12 / (FP64_VECTOR_PEAK_TFLOPS / 2.0), 4),
                     "bound": "neither HBM nor the vector ALU: with one bit per entry the kernel moves x, y and 16 bytes per 64 rows; SQ counters "
                              "(profiles/r06_lean_k1_sq.json, docs/LAB_NOTEBOOK.md R6.5): 171 vector + 138 scalar instructions per wavefront of 64 rows, "
                              "both pipes ~55 % busy, wavefronts waiting on their three dependent groups of gathers half of their time at 8 waves per SIMD"}
            # the same block with SEVEN distinct values (a stencil with a handful of coefficients: anisotropic / layered media): the
            # dictionary is renewed after eight products on the new values, then one BYTE per entry in pattern-ELL order (round 6)
            try:
                PHASE[0] = "value-dictionary mode: seven values"
                os.environ.pop("PA_SPMV_VALUE_DICT", None)       # (the renewal of the dictionary reads it)
                oo = blk2.own_own
                table = np.array([26.0, -1.0, 0.375, -2.5, 1.0 / 3.0, 7.0, -0.125])
                pick = np.random.default_rng(7).integers(0, len(table), 1009)
                oo.update_values(np.resize(table[pick], oo.nnz))
                want7 = pa.DeviceVector(n_own, 0)
                with_env = dict(PA_SPMV_PELL_BYTES="0")
                for _ in range(10):
                    pa.spmv_(y2v, oo, xv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0)
                seven = {"distinct_values": oo.value_dict(), "kernel": kernel_of(oo)}
                for tag, sw in (("pattern_ell_one_byte", {}), ("row_split_one_byte", with_env)):
                    os.environ.update(sw); ctx.reload_env()
                    for _ in range(30):
                        pa.spmv_(y2v, oo, xv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0)
                    e0 = ctx.event().record(L.STREAM_COMPUTE)
                    for _ in range(30):
                        pa.spmv_(y2v, oo, xv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0)
                    e1 = ctx.event().record(L.STREAM_COMPUTE)
                    ctx.sync()
                    ms7 = e0.elapsed_ms(e1) / 30
                    mv7 = oo.stream_bytes() + 16 * n_own
                    seven[tag] = {"ms": round(ms7, 4), "gflops": round(2.0 * nnz_oo / (ms7 * 1e-3) / 1e9, 1), "moved_bytes_per_launch": int(mv7),
                                  "frac_moved": round(mv7 / (ms7 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "pell_mode": oo.pell()["mode"]}
                    if tag == "pattern_ell_one_byte":
                        pa.spmv_(want7, oo, xv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0)
                    for k in sw:
                        os.environ.pop(k, None)
                    ctx.reload_env()
                ctx.sync()
                seven["bit_identical_products"] = bool(np.array_equal(want7.download(), y2v.download()[:n_own]))
                seven["what"] = ("the headline block's pattern with 7 distinct values: library defaults keep one BYTE per entry in pattern-ELL order, the dictionary "
                                 "in 512 bytes of LDS (pa_pell.h, VM 2); beside it the row-split kernel's one-byte stream (PA_SPMV_PELL_BYTES=0: what such "
                                 "blocks ran on before)")
                vdict["seven_values"] = seven
                del want7
            except Exception as e:                             # noqa: BLE001
                print(f"[bench] seven-values extra skipped: {e}", file=sys.stderr)
            os.environ["PA_SPMV_VALUE_DICT"] = "0"
            os.environ.pop("PA_SPMV_PELL_BYTES", None)
            ctx.reload_env()
            del A2, _b2, y2, blk2
        except Exception as e:                                 # noqa: BLE001
            os.environ["PA_SPMV_VALUE_DICT"] = "0"
            print(f"[bench] value-dictionary extra skipped: {e}", file=sys.stderr)
            vdict = None

    # ---- BASELINE config 4's loop, reported beside the headline (never part of `value`): one CG iteration of
    # HPCG/src/ref_cg.jl (consistent!+mul!, 2 dots + norm, 3 axpys; Identity preconditioner), as the reference
    # schedules it (ref_cg_: a blocking reduction per dot) and as opt_cg_ does (scalars stay on the device).
    cg = None
    if args.cg_iters > 0:
        with optional_section("CG loop", 240, N, rank):
            work = pa.cg_work(pa.pzeros(A.col_partition), b, A)     # the loop's own work vectors, allocated once

            def cg_time(fn, k):
                xx = pa.pzeros(A.col_partition)
                for _ in range(60):                     # (working clocks: the same number of steps on every rank)
                    step(overlap_on)
                barrier()
                t = time.perf_counter()
                fn(xx, A, b, maxiter=k, work=work)
                ctx.sync()
                barrier()
                return time.perf_counter() - t
            res = {}
            import functools
            for name, fn in (("ref_cg", pa.ref_cg_), ("opt_cg_unfused", functools.partial(pa.opt_cg_, fuse=False)),
                             ("opt_cg", functools.partial(pa.opt_cg_, fuse=True))):
                cg_time(fn, 2)
                for attempt in range(3):
                    t_long = cg_time(fn, 4 + args.cg_iters)
                    d = t_long - cg_time(fn, 4)                         # the difference cancels allocation + first residual
                    if N > 1:
                        tt = torch.tensor([d, t_long], dtype=torch.float64)
                        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                        d, t_long = float(tt[0].item()), float(tt[1].item())
                    if d > 0:
                        break
                    # (a few iterations of a tiny part with the ranks sharing one GPU: the two wall-clock spans can come out the wrong way
                    #  round; measured again, and after three tries the long run alone, pro rata -- the same decision on every rank)
                    d = t_long * args.cg_iters / (4 + args.cg_iters)
                res[name] = d / args.cg_iters * 1e3
            cg = res
            if rank == 0:
                n_rows_total = n_own * N
                cg_flops = 2.0 * nnz_total + 12.0 * n_rows_total      # SpMV + 3 dots + 3 axpys (HPCG/src/report_results.jl)
                LINE[0]["cg_loop"] = {"what": "one CG iteration of HPCG ref_cg.jl on the same matrix (BASELINE config 4 loop, "
                                              "Identity preconditioner): consistent!+mul!, 2 dots + norm, 3 axpys",
                                      "iterations_timed": args.cg_iters,
                                      "ms_per_iteration_ref_cg": round(cg["ref_cg"], 4),
                                      "ms_per_iteration_opt_cg_unfused": round(cg["opt_cg_unfused"], 4),
                                      "ms_per_iteration_opt_cg": round(cg["opt_cg"], 4),
                                      "gflops_opt_cg": round(cg_flops / (cg["opt_cg"] * 1e-3) / 1e9, 1),
                                      "note": "ref_cg_ = the reference's schedule, a blocking reduction to the host per dot; opt_cg_(fuse=False) = the "
                                              "same kernels in the same order with the scalars kept on the device: the iterates of ref_cg_ bit for "
                                              "bit (tests/test_gpu_blas1_cg.py); opt_cg_(fuse=True) = also u'c accumulated inside the product "
                                              "kernels and x's update fused into u's pass (iterates within 1e-9)"}

    general, f32 = None, None
    if N == 1 and args.extra and rank == 0:
        general = []
        try:
            if A.host_blocks is not None:
                host_oo = pa.local_items(A.host_blocks)[0][0]
            else:
                from pa_amd.gallery import build_split_blocks_fused
                host_oo = build_split_blocks_fused(pa.local_items(A.row_partition)[0], n, n, n, *gn)[1]
            pa.mul_(y, A, x)
            ctx.sync()
            y_head = pa.local_items(y.own_values())[0]
            general_csr_entries(pa, ctx, L, host_oo, xv, y_head, n_own, general)
            try:
                f32 = float32_entry(pa, ctx, L, host_oo, xv, y_head, n_own)
            except Exception as e:                             # noqa: BLE001
                print(f"[bench] Float32 entry skipped: {e}", file=sys.stderr)
            del host_oo
        except Exception as e:                                 # noqa: BLE001
            print(f"[bench] general-CSR entries stopped at {PHASE[0]!r}: {e}", file=sys.stderr)
        general = general or None

    extras = None
    if N == 1 and args.extra and rank == 0:
        extras = []
        try:
            extra_configs(pa, ctx, L, extras)
        except Exception as e:                                 # noqa: BLE001
            print(f"[bench] extra configs stopped at {PHASE[0]!r}: {e}", file=sys.stderr)
        extras = extras or None

    # (behind the extras: its sort scratch -- ~16 GB of plain allocations -- makes the driver wipe memory the extras' extents would take)
    # ---- SURVEY 8(f) row f2 at the headline's size: mul!(c,transpose(A),b,1,0) with A' built on the device from the resident
    # block (no host copy exists: the headline matrix was generated in HBM)
    transpose = None
    if N == 1 and args.extra and rank == 0:
        try:
            PHASE[0] = "transpose product"
            ctx.sync()
            t = time.perf_counter()
            pa.transposed_blocks(A)
            ctx.sync()
            t_build = time.perf_counter() - t
            ct = pa.pzeros(A.col_partition)
            bt = pa.pvector_from_function(lambda ind: hash_x(ind.get_local_to_global()), A.row_partition)
            pa.mul5_transpose_(ct, A, bt, 1.0, 0.0)
            pa.mul_(y, A, x)                                   # A = A' for this operator and x = bt on the own entries
            ctx.sync()
            sym_err = float(np.max(np.abs(pa.local_items(ct.own_values())[0] - pa.local_items(y.own_values())[0])))
            spin_up(ctx, lambda: pa.mul5_transpose_(ct, A, bt, 1.0, 0.0))
            e0 = ctx.event().record(L.STREAM_COMPUTE)
            for _ in range(30):
                pa.mul5_transpose_(ct, A, bt, 1.0, 0.0)
            e1 = ctx.event().record(L.STREAM_COMPUTE)
            ctx.sync()
            ms_t = e0.elapsed_ms(e1) / 30
            tb = pa.local_items(pa.transposed_blocks(A))[0][0]
            transpose = {"workload": f"mul!(c,transpose(A),b,1,0) on the headline matrix (27-pt {n}^3, 1 part), A' built on the device "
                                     "(pa_csr_create_transpose: decode + one stable radix sort by column + device-side block constructor)",
                         "ms": round(ms_t, 4), "gflops": round(2.0 * nnz / ms_t / 1e6, 1), "rate_vs_forward": round(ms_per_step / ms_t, 3),
                         "build_s": round(t_build, 2), "encoding": tb.encoding(), "kernel": kernel_of(tb),
                         "moved_bytes_per_launch": int(tb.stream_bytes() + 16 * n_own),
                         "frac_moved": round((tb.stream_bytes() + 16 * n_own) / ms_t / 1e6 / HBM_PEAK_GBPS, 4),
                         "max_abs_diff_vs_forward_product": sym_err}
            del ct, bt
            A._t_blocks = None
        except Exception as e:                                 # noqa: BLE001
            print(f"[bench] transpose product skipped: {e}", file=sys.stderr)

    if want_cpu:
        with optional_section("CPU baseline", 3 * args.cpu_seconds + 240, N, rank):
            cpu = cpu_mul_baseline(pa, A, N, rank, args.cpu_seconds, (n, n, n, *gn))
            # the free full-size gate (VERDICT r03 #1c): the oracle's y for the bench's own x at the bench's own size against
            # the device's, bit for bit, on every rank
            same = None
            if cpu is not None and getattr(cpu_mul_baseline, "y", None) is not None:
                x.vector_partition = pa.pmap(lambda v, ind: v.upload(xfun(ind)), x.vector_partition, A.col_partition)
                pa.mul_(y, A, x)
                ctx.sync()
                same = bool(np.array_equal(pa.local_items(y.own_values())[0], cpu_mul_baseline.y))
                if N > 1:
                    flag = torch.tensor([1 if same else 0])
                    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
                    same = bool(flag.item())
            if rank == 0:
                LINE[0]["parity_full_size"] = same
                LINE[0]["parity_gate"] += ("; full size: mul!(y, A, x) for the hashed x == the CPU baseline's (oracle C loops) y bit for bit on every "
                                           "part" if same else "; full size vs the CPU baseline: " + ("MISMATCH" if same is False else "not run"))
            if cpu is not None and rank == 0:
                c1 = cpu_c1_debugarray()
                if c1:
                    cpu["c1_debugarray"] = c1
                LINE[0]["cpu_baseline"] = cpu

    # ---- N > 1, last of all (a hang in here costs nothing that came before): the same timed loop over the OTHER device transport,
    # the ipc push of csrc/pa_push.hip (pack kernel storing straight into the neighbours' hipIpc-mapped receive buffers), so that the
    # first real multi-GPU record says what each transport costs beside own x own (profiles/r04_rccl_interference.json: one GPU)
    if N > 1 and transport in ("rccl", "host", "torch") and os.environ.get("PA_BENCH_TRANSPORT_AB", "1") != "0":
        with optional_section("transport A/B: ipc push", 120, N, rank):
            import pa_amd.p_vector as pv
            os.environ.setdefault("PA_IPC_TIMEOUT_S", "5")
            was = pv.TRANSPORT
            pv.TRANSPORT = "ipc"
            try:
                pv.connect_ipc(x.cache.plans)
                ok_ipc = gate()
                for _ in range(30):
                    step(overlap_on)
                t_ipc, _, _ = timed(args.steps, overlap_on)
            finally:
                pv.TRANSPORT = was
            if rank == 0:
                ms_ipc_ = t_ipc / args.steps * 1e3
                LINE[0]["config"]["transport_chosen"] = {
                    "headline": transport, "faster_on_this_run": ("ipc" if (ok_ipc and ms_ipc_ < ms_per_step) else transport),
                    "ms_per_step": {transport: round(ms_per_step, 4), "ipc": round(ms_ipc_, 4)}, "rccl_ranks_seen": rccl_ranks_seen,
                    "rule": "`value` is ALWAYS the headline transport's (RCCL on a multi-GPU node, as north_star names it); the ipc push is "
                            "timed by the same loop behind it and named here when it is the faster one -- PA_TRANSPORT=ipc selects it"}
                LINE[0]["transport_ab"] = {"headline_transport": transport, "ms_per_step_headline": round(ms_per_step, 4),
                                           "ms_per_step_ipc_push": round(t_ipc / args.steps * 1e3, 4), "parity_gate_over_ipc_push": bool(ok_ipc),
                                           "gflops_ipc_push": round(flops_total / (t_ipc / args.steps) / 1e9, 2),
                                           "what": "the timed mul! loop again with PA_TRANSPORT=ipc (csrc/pa_push.hip): same matrix, same "
                                                   "vectors, same barriers; `value` is the headline transport's"}

    if rank == 0:
        out = LINE[0]
        if vdict:
            out["value_dictionary_mode"] = vdict
            # the product a user of the library gets WITHOUT switches (the dictionary is the library's default for big blocks with
            # <= 64 distinct values; `value` switches it off to keep the fp64 stream): first-class beside `value`
            out["value_library_defaults"] = vdict["gflops"]
            out["ms_per_step_library_defaults"] = vdict["avg_launch_ms"]
        if general:
            out["general_csr"] = general
        if f32:
            out["float32_mode"] = f32
        if transpose:
            out["transpose_product"] = transpose
        if extras:
            out["extra_configs"] = extras
        out["phase_seconds"] = PHASE.seconds()
        print(json.dumps(out), flush=True)
    if N > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
