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
PHASE = ["start"]


def watchdog(seconds):
    """A multi-process run that deadlocks (a neighbour exchange that never completes) would sit until the driver's
    own limit; instead say where it hung and exit non-zero.  Generous: set-up + 4 x the default run fit many times."""
    import threading

    def fire():
        print(f"[bench watchdog] no progress after {seconds} s, phase = {PHASE[0]!r}; aborting", file=sys.stderr, flush=True)
        os._exit(4)
    t = threading.Timer(seconds, fire)
    t.daemon = True
    t.start()
    return t
sys.path.insert(0, ROOT)

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
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=160, help="grid size of the bounded CPU-baseline sample")
    return ap.parse_args()


def cpu_baseline(n):
    """The reference's spmv_csr! loop (src/sparse_utils.jl:649-669) restated in C (oracle/pa_oracle.c,
    -O3 -ffp-contract=off), one core, on a bounded sample: the same 27-point matrix at n^3 rows."""
    import ctypes as C
    from __graft_entry__ import load_oracle
    orc = load_oracle()
    K = orc.oracle_c()
    if not hasattr(K, "lib"):
        return None
    lib = K.lib
    lib.orc_hpcg_csr_single.restype = C.c_int64
    lib.orc_hpcg_csr_single.argtypes = [C.c_int32] * 3 + [C.c_void_p] * 3
    rows = n ** 3
    rowptr = np.zeros(rows + 1, np.int32)
    nnz = lib.orc_hpcg_csr_single(n, n, n, rowptr.ctypes.data, None, None)
    colval, nzval = np.zeros(nnz, np.int32), np.zeros(nnz, np.float64)
    lib.orc_hpcg_csr_single(n, n, n, rowptr.ctypes.data, colval.ctypes.data, nzval.ctypes.data)
    x = orc.hash_x(np.arange(1, rows + 1))
    y = np.zeros(rows)
    A = orc.CSR(rows, rows, rowptr, colval, nzval)
    K.spmv_csr(y, x, A)                       # warm
    reps, t0 = 0, time.perf_counter()
    while True:
        K.spmv_csr(y, x, A)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > 12.0 or reps >= 200:
            break
    t = dt / reps
    return {"value": round(2.0 * nnz / t / 1e9, 4), "unit": "GFLOP/s", "cores": 1, "kind": "port",
            "sample": f"27-pt HPCG matrix {n}^3 rows ({nnz} nnz), {reps} x spmv_csr! (oracle/pa_oracle.c) in {dt:.1f} s",
            "gbps_algorithmic": round((nnz * 12 + (rows + 1) * 4 + rows * 16) / t / 1e9, 3)}


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


def main():
    args = parse()
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
    pa = load_package()
    ctx = pa.context()

    n = args.n
    npx, npy, npz = pa.compute_optimal_shape_XYZ(N)
    gn = (npx * n, npy * n, npz * n)
    transport = "none(1 part)"
    if N > 1:
        import pa_amd.p_vector as pv
        transport = os.environ.get("PA_TRANSPORT", "rccl")
        if transport == "rccl":
            PHASE[0] = "RCCL communicator"
            try:                       # direct RCCL (ncclSend/ncclRecv issued by libpa_hip on its comm stream)
                pa.init_comm()
                good = 1
            except Exception as e:     # noqa: BLE001
                print(f"[rank {rank}] direct RCCL communicator failed: {e}", file=sys.stderr)
                good = 0
            flag = torch.tensor([good])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if not flag.item():
                transport = "torch"
        pv.TRANSPORT = transport
        ranks = pa.with_torchdist(lambda distribute: distribute(range(1, N + 1)))
    else:
        ranks = pa.DebugArray([1])

    PHASE[0] = "matrix set-up"
    t_setup = time.perf_counter()
    A, b = pa.build_p_matrix(ranks, n, n, n, *gn, npx, npy, npz)
    # x[gid] = ((gid*2654435761) mod 2^32)/2^32 on OWN entries only: mul! must bring the ghosts
    def xfun(ind):
        g = ind.get_local_to_global().astype(np.uint64)
        v = ((g * np.uint64(2654435761)) % np.uint64(2 ** 32)).astype(np.float64) / float(2 ** 32)
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
            g = ind.get_local_to_global().astype(np.uint64)
            want = ((g * np.uint64(2654435761)) % np.uint64(2 ** 32)).astype(np.float64) / float(2 ** 32)
            ok = ok and np.array_equal(vals, want)
        if N > 1:
            flag = torch.tensor([1 if ok else 0])
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            ok = bool(flag.item())
        return ok

    PHASE[0] = f"parity gate (transport {transport})"
    ok = gate()
    if not ok and N > 1 and transport == "rccl":
        print(f"[rank {rank}] parity gate failed with the direct RCCL transport; retrying with torch.distributed p2p",
              file=sys.stderr)
        transport = pv.TRANSPORT = "torch"
        ok = gate()
    if not ok:
        raise SystemExit("parity gate failed: A*1 != b or ghost values differ from their owners")

    blk = pa.local_items(A.matrix_partition)[0]
    ind = pa.local_items(A.col_partition)[0]
    nnz_oo, nnz_oh, n_own, n_ghost = blk.own_own.nnz, blk.own_ghost.nnz, ind.n_own, ind.n_ghost

    # ---- HIP events around the dominant kernel (own x own SpMV) inside the timed region, compute stream
    import pa_amd._lib as L
    xv, yv = pa.local_items(x.vector_partition)[0], pa.local_items(y.vector_partition)[0]
    ev0 = [ctx.event() for _ in range(args.steps)]
    ev1 = [ctx.event() for _ in range(args.steps)]

    def step(k=None):
        # mul!(c,a,b): src/p_sparse_matrix.jl:2098-2101
        t = pa.consistent_(x)
        if k is not None:
            ev0[k].record(L.STREAM_COMPUTE)
        pa.spmv_(yv, blk.own_own, xv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0)
        if k is not None:
            ev1[k].record(L.STREAM_COMPUTE)
        t.wait()
        pa.spmv_(yv, blk.own_ghost, xv, L.SEG_GHOST, L.SEG_OWN, 1.0, 1.0)

    def barrier():
        ctx.sync()
        if N > 1:
            dist.barrier()
            ctx.sync()

    # optional, measured, result-neutral: keep the value stream in the allocation on which own x own runs fastest with
    # THESE x and y (pa_csr_tune_placement).  PA_PLACEMENT_TRIES=0 turns it off.
    tries = int(os.environ.get("PA_PLACEMENT_TRIES", "16"))
    placement_searches = 0
    tries0 = tries                         # (the same on every rank: what the collective decisions below depend on)
    for attempt in range(3):
        if tries > 1:
            PHASE[0] = "value-stream placement"
            try:                               # an optional, result-neutral step never costs the run its line
                kept = blk.own_own.tune_placement(xv, yv, tries=tries)["kept_ms"]
                placement_searches += 1
            except Exception as e:             # noqa: BLE001
                print(f"[rank {rank}] placement search skipped: {e}", file=sys.stderr)
                tries = 0
        PHASE[0] = f"warm-up (transport {transport})"
        for _ in range(args.warmup):
            step()
        if tries0 <= 1 or args.warmup < 1 or args.steps < 1:
            break
        # a placement can stop holding when other allocations come and go: one untimed step with the
        # kernel's events tells; search again (at most twice) if the product is 4 % slower than the search left it
        step(0)
        ctx.sync()
        again = 1 if tries > 1 and ev0[0].elapsed_ms(ev1[0]) > 1.04 * kept else 0
        if N > 1:                          # (every rank decides the same: the search is a collective no-op otherwise)
            flag = torch.tensor([again])
            dist.all_reduce(flag, op=dist.ReduceOp.MAX)
            again = int(flag.item())
        if not again:
            break
    PHASE[0] = f"timed mul! loop (transport {transport})"
    barrier()
    mono0 = time.clock_gettime_ns(time.CLOCK_MONOTONIC)       # (lets profiles/summarize.py find the timed launches in a trace)
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    ctx.sync()
    barrier()
    dt = time.perf_counter() - t0
    mono1 = time.clock_gettime_ns(time.CLOCK_MONOTONIC)
    if N > 1:
        tt = torch.tensor([dt], dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    ms_per_step = dt / args.steps * 1e3
    kern_ms = float(np.mean([a.elapsed_ms(b_) for a, b_ in zip(ev0, ev1)]))

    nnz = nnz_oo + nnz_oh
    if N > 1:
        tot = torch.tensor([nnz], dtype=torch.int64)
        dist.all_reduce(tot)
        nnz_total = int(tot.item())
    else:
        nnz_total = nnz
    flops_total = 2.0 * nnz_total
    value = flops_total / (ms_per_step * 1e-3) / 1e9

    # algorithmic bytes: whole mul! per part, and the dominant kernel's share
    bytes_mul = nnz * 12 + (n_own + 1) * 4 + (n_own + n_ghost) * 8 + n_own * 8
    bytes_oo = nnz_oo * 12 + (n_own + 1) * 4 + n_own * 8 + n_own * 8
    ach = bytes_oo / (kern_ms * 1e-3) / 1e9

    # measured HBM traffic of the dominant kernel: PMC passes are separate rocprofv3 runs (never inside a timed
    # run); their per-launch summary is committed under profiles/ and quoted here (null when absent).
    traffic, traffic_src = None, None
    try:
        import glob
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_summary.json")))
        if cands and n == 256:
            pm = json.load(open(cands[-1]))["pmc_per_launch"]["k_spmv_rowsplit"]
            traffic = round(pm["fetch_bytes_gfx950_corrected"] + pm["write_bytes"])
            traffic_src = os.path.relpath(cands[-1], ROOT) + " (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, 1-GPU 256^3 run)"
    except Exception:
        pass

    # ---- what this box's HBM delivers to the simplest kernels ("re-confirm on the box with a device triad"):
    # a device-to-device copy and a two-stream read (dot) over vectors far larger than the 256 MiB Infinity Cache
    box = None
    try:
        box = calibrate_box(pa, ctx, L) if (rank == 0 or N > 1) else None
    except Exception as e:                                   # noqa: BLE001  (an extra; no collectives inside)
        print(f"[bench] HBM calibration skipped: {e}", file=sys.stderr)
    # ---- optional mode, reported beside the headline and never part of `value`: the same mul! with the lossless value
    # dictionary (PA_SPMV_VALUE_DICT=1: one byte per stored entry instead of eight when a block has <= 64 distinct values)
    vdict = None
    if N == 1 and args.value_dict:
        try:                                                   # an optional extra never costs the headline its line
            PHASE[0] = "value-dictionary mode"
            os.environ["PA_SPMV_VALUE_DICT"] = "1"
            A2, _b2 = pa.build_p_matrix(ranks, n, n, n, *gn, npx, npy, npz)
            os.environ.pop("PA_SPMV_VALUE_DICT")
            blk2 = pa.local_items(A2.matrix_partition)[0]
            y2 = pa.pzeros(A2.row_partition)
            pa.mul_(y2, A2, x)
            same = all(np.array_equal(a_, b_) for a_, b_ in zip(pa.local_items(y2.own_values()), pa.local_items(y.own_values())))
            y2v = pa.local_items(y2.vector_partition)[0]
            if tries > 1:                                      # the code stream and y2 placed by measurement as well
                blk2.own_own.tune_placement(xv, y2v, tries=tries)
            for _ in range(args.warmup):
                pa.spmv_(y2v, blk2.own_own, xv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0)
            e0 = ctx.event().record(L.STREAM_COMPUTE)
            for _ in range(args.steps):
                pa.spmv_(y2v, blk2.own_own, xv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0)
            e1 = ctx.event().record(L.STREAM_COMPUTE)
            ctx.sync()
            ms2 = e0.elapsed_ms(e1) / args.steps
            vdict = {"what": "own x own SpMV with PA_SPMV_VALUE_DICT=1 (optional, lossless; NOT used for `value`)",
                     "distinct_values": blk2.own_own.value_dict(), "bit_identical_to_headline_product": bool(same),
                     "avg_launch_ms": round(ms2, 4), "gflops": round(2.0 * nnz_oo / (ms2 * 1e-3) / 1e9, 1),
                     "algorithmic_gbps": round(bytes_oo / (ms2 * 1e-3) / 1e9, 1)}
            del A2, _b2, y2, blk2
        except Exception as e:                                 # noqa: BLE001
            os.environ.pop("PA_SPMV_VALUE_DICT", None)
            print(f"[bench] value-dictionary extra skipped: {e}", file=sys.stderr)
            vdict = None

    # ---- BASELINE config 4's loop, reported beside the headline (never part of `value`): one CG iteration of
    # HPCG/src/ref_cg.jl (consistent!+mul!, 2 dots + norm, 3 axpys; Identity preconditioner), as the reference
    # schedules it (ref_cg_: a blocking reduction per dot) and as opt_cg_ does (scalars stay on the device).
    cg = None
    PHASE[0] = "CG loop"
    if args.cg_iters > 0:
        # the loop's own work vectors, allocated once; the value stream and c placed for c = A*u (result-neutral)
        work = pa.cg_work(pa.pzeros(A.col_partition), b, A, tune_placement=tries)

        def cg_time(fn, k):
            xx = pa.pzeros(A.col_partition)
            barrier()
            t = time.perf_counter()
            fn(xx, A, b, maxiter=k, work=work)
            ctx.sync()
            barrier()
            return time.perf_counter() - t
        cg = {}
        for name, fn in (("ref_cg", pa.ref_cg_), ("opt_cg", pa.opt_cg_)):
            cg_time(fn, 2)
            d = cg_time(fn, 4 + args.cg_iters) - cg_time(fn, 4)     # the difference cancels allocation + first residual
            if N > 1:
                tt = torch.tensor([d], dtype=torch.float64)
                dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                d = float(tt.item())
            cg[name] = d / args.cg_iters * 1e3

    if rank == 0:
        out = {
            "metric": "HPCG 27-pt SpMV GFLOP/s + achieved HBM GB/s per GPU",
            "value": round(value, 2), "unit": "GFLOP/s", "n_gpus": N, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"HPCG 27-pt stencil {n}^3 rows per part, {N} part(s) as ({npx},{npy},{npz}), "
                                   "mul! = consistent!(pack+exchange+unpack) overlapped with own*own, then own*ghost",
                       "rows_per_part": n_own, "nnz_per_part": nnz, "nnz_own_own": nnz_oo, "nnz_own_ghost": nnz_oh,
                       "ghosts_per_part": n_ghost, "index_type": "Int32", "transport": {"rccl": "rccl-p2p (ncclSend/ncclRecv group on the comm stream)", "torch": "torch.distributed p2p (fallback)"}.get(transport, transport)},
            "gflops_per_gpu": round(value / N, 2),
            "hbm_gbps_per_gpu_algorithmic": round(bytes_mul / (ms_per_step * 1e-3) / 1e9, 1),
            "roofline": {"bound": "hbm", "kernel": "k_spmv_rowsplit<256,6,nt> (own x own): CSR row split, LDS-staged products; "
                                                      "column encoding of the chunks: " + json.dumps(blk.own_own.encoding()),
                         "achieved": round(ach, 1),
                         "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_bytes_per_launch": bytes_oo, "avg_launch_ms": round(kern_ms, 4),
                         "timed_region_monotonic_ns": [mono0, mono1],
                         "this_box": box,
                         "value_stream_placement": dict(blk.own_own.placement(), searches=placement_searches, what="allocations of the value stream timed with the "
                                                        "bench's own x and y before the warm-up, fastest kept "
                                                        "(pa_csr_tune_placement, PA_PLACEMENT_TRIES; DESIGN.md 3)")},
            "parity_gate": "A*1==b bit-exact; ghosts==owners bit-exact",
            "setup_s": round(t_setup, 1),
        }
        if vdict:
            out["value_dictionary_mode"] = vdict
        if cg:
            n_rows_total = n_own * N
            cg_flops = 2.0 * nnz_total + 12.0 * n_rows_total      # SpMV + 3 dots + 3 axpys (HPCG/src/report_results.jl)
            out["cg_loop"] = {"what": "one CG iteration of HPCG ref_cg.jl on the same matrix (BASELINE config 4 loop, "
                                      "Identity preconditioner): consistent!+mul!, 2 dots + norm, 3 axpys",
                              "iterations_timed": args.cg_iters,
                              "ms_per_iteration_ref_cg": round(cg["ref_cg"], 4),
                              "ms_per_iteration_opt_cg": round(cg["opt_cg"], 4),
                              "gflops_opt_cg": round(cg_flops / (cg["opt_cg"] * 1e-3) / 1e9, 1),
                              "note": "opt_cg_ = same arithmetic (bit-identical iterates), scalars kept on the device"}
        if N == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(args.cpu_n)
            if cb:
                out["cpu_baseline"] = cb
        print(json.dumps(out), flush=True)
    if N > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
