"""HPCG's benchmark driver and report (HPCG/src/hpcg_benchmark.jl, HPCG/src/report_results.jl) on top of the package:
three phases (reference solve, optimised solve to the reference tolerance, timed sets) and the flop / byte models of the
report.  A development tool beside the probes -- SURVEY section 2 marks the report writer out of the hot path's scope, so it
is not part of `partitionedarrays.jl_amd/`.

    python tools/hpcg_driver.py <parts> <n> [runtime_s]
"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# (the context keeps up to this much idle memory between the phases instead of 24 GiB: the extents the reference solver gives back
#  are the ones the optimised solver's set-up takes, without another pass of the driver's wiping and of the class search)
os.environ.setdefault("PA_ARENA_SPARE_GIB", "96")
os.environ.setdefault("PA_ARENA_SPARE", "8")          # (... and up to 8 emptied extents while others are still in use)
# the forward half of a zero-guess sweep on blocks of the lower colours only saves 0.3 ms of an iteration and costs 0.13 s of set-up,
# which the rating charges per set: 1051 against 1075 GFLOP/s with / without them at 256^3 (round 5) -- the driver goes without
os.environ.setdefault("PA_GS_LOWER", "0")
from __graft_entry__ import load_package  # noqa: E402

pa = load_package()
from pa_amd.hpcg import CgTimer, cg_work, opt_cg_, pc_setup, ref_cg_   # noqa: E402
from pa_amd.p_vector import context, pzeros                    # noqa: E402
from pa_amd.gallery import compute_optimal_shape_XYZ           # noqa: E402
from pa_amd.primitives import local_items, pmap                # noqa: E402


# ----------------------------------------------------------------------------------------------
# the benchmark driver and its report (HPCG/src/hpcg_benchmark.jl, HPCG/src/report_results.jl)
# ----------------------------------------------------------------------------------------------
def hpcg_geometry(np_, l, nx, ny, nz):
    """Geometry (HPCG/src/mg_preconditioner.jl:17-26): per level (index 0 = coarsest) global rows and stored entries
    of the 27-point operator, closed form: prod(g_d) rows, prod(3 g_d - 2) entries."""
    npx, npy, npz = compute_optimal_shape_XYZ(np_)
    nrows, nnz = [], []
    for lev in range(l):
        f = 2 ** (l - 1 - lev)
        g = (npx * nx // f, npy * ny // f, npz * nz // f)
        nrows.append(g[0] * g[1] * g[2])
        nnz.append((3 * g[0] - 2) * (3 * g[1] - 2) * (3 * g[2] - 2))
    return dict(nx=nx, ny=ny, nz=nz, npx=npx, npy=npy, npz=npz, nnz=nnz, nrows=nrows)


def hpcg_report(np_, times, levels, ref_max_iters, opt_max_iters, nr_cg_sets, norm_data, geom):
    """report_results (HPCG/src/report_results.jl:21-144): the flop and byte models of the official benchmark and its
    GFLOP/s rating.  `times`: dict total/DDOT/WAXPBY/SPMV/MG/setup/opt_time/ref_time in seconds (timing_data[1..10])."""
    fniters = nr_cg_sets * opt_max_iters
    fnrow, fnnz = float(geom["nrows"][levels - 1]), float(geom["nnz"][levels - 1])
    ops_ddot = (3.0 * fniters + nr_cg_sets) * 2.0 * fnrow
    ops_waxpby = (3.0 * fniters + nr_cg_sets) * 2.0 * fnrow
    ops_spmv = (fniters + nr_cg_sets) * 2.0 * fnnz
    ops_mg = 0.0
    for i in range(1, levels):                               # levels 2..l of the reference
        ops_mg += fniters * (4.0 + 2.0 + 4.0) * geom["nnz"][i]
    ops_mg += fniters * 4.0 * geom["nnz"][0]
    ops = ops_ddot + ops_waxpby + ops_spmv + ops_mg
    ref_ops = ops * (ref_max_iters / opt_max_iters)
    f8, i8 = 8.0, 8.0                                        # sizeof(Float64), sizeof(Int64): the model's, not ours
    reads = (3.0 * fniters + nr_cg_sets) * 2.0 * fnrow * f8 * 2 + (fniters + nr_cg_sets) * (fnnz * (f8 + i8) + fnrow * f8)
    writes = (3.0 * fniters + nr_cg_sets) * f8 + (3.0 * fniters + nr_cg_sets) * fnrow * f8 + (fniters + nr_cg_sets) * fnrow * f8
    for i in range(1, levels):
        nz, nr = float(geom["nnz"][i]), float(geom["nrows"][i])
        reads += fniters * (2.0 * nz * (f8 + i8) + nr * f8) * 2 + fniters * (nz * (f8 + i8) + nr * f8)
        writes += fniters * nz * f8 * 3
    reads += fniters * (2.0 * geom["nnz"][0] * (f8 + i8) + geom["nrows"][0] * f8)
    writes += fniters * geom["nrows"][0] * f8
    ref_rw = (reads + writes) * ref_max_iters / opt_max_iters
    overhead = times["total"] + nr_cg_sets * (times["opt_time"] / 10.0 + times["setup"] / 10.0)
    norm_data = np.asarray(norm_data, dtype=np.float64)
    gf = lambda o, t: (o / t / 1e9) if t > 0 else float("nan")
    return {
        "procs": np_, "times": dict(times), "nr_equations": int(fnrow), "non_zeros": int(fnnz),
        "multigrid_data": {f"level_{i + 1}": {"non_zeros": geom["nnz"][i], "nr_equations": geom["nrows"][i]} for i in range(levels)},
        "geometry": {k: geom[k] for k in ("npx", "npy", "npz", "nx", "ny", "nz")},
        "iter_data": {"ref_iters_set": ref_max_iters, "opt_iters_set": opt_max_iters,
                      "ref_iters_total": ref_max_iters * nr_cg_sets, "opt_iters_total": opt_max_iters * nr_cg_sets},
        "reproducibility_data": {"mean": float(norm_data.mean()), "var": float(norm_data.var(ddof=1)) if len(norm_data) > 1 else 0.0},
        "flops": {"DDOT": ops_ddot, "WAXPBY": ops_waxpby, "SpMV": ops_spmv, "MG": ops_mg, "Total": ops, "Total_conv": ref_ops},
        "GB/s": {"Read": reads / times["total"] / 1e9, "Write": writes / times["total"] / 1e9,
                 "Total": (reads + writes) / times["total"] / 1e9, "Total_conv_opt": ref_rw / overhead / 1e9},
        "GFLOP/s": {"DDOT": gf(ops_ddot, times["DDOT"]), "WAXPBY": gf(ops_waxpby, times["WAXPBY"]),
                    "SpMV": gf(ops_spmv, times["SPMV"]), "MG": gf(ops_mg, times["MG"]),
                    "Total": ops / times["total"] / 1e9, "Total_conv": ref_ops / times["total"] / 1e9,
                    "Total_conv_opt": ref_ops / overhead / 1e9},
        "Overview": {"GFLOP/s": ref_ops / overhead / 1e9, "time": times["total"]},
    }


def hpcg_benchmark(ranks, np_, nx, ny, nz, total_runtime=60.0, levels=4, ref_max_iters=50, ref_ordering="sequential",
                   opt_ordering="multicolor_spmv", max_sets=None, output_type="none", output_folder="results"):
    """hpcg_benchmark(distribute,np,nx,ny,nz;total_runtime) (HPCG/src/hpcg_benchmark.jl:28-116), three phases:
      reference   : two sets of `ref_max_iters` MG-PCG iterations with the reference's smoother -> ref_tol = |r|/|r0|;
      optimisation: the optimised solver (multicolour smoother, opt_cg_) runs to ref_tol; the iterations it needs
                    (>= ref_max_iters) are what every timed set must perform -- extra iterations are charged;
      timing      : ceil(total_runtime / worst set time) sets of that many iterations.
    Returns the report dictionary of hpcg_report (and writes it when output_type is "json" or "txt")."""
    from pa_amd.primitives import getany, reduction
    ctx = context()
    # What is timed as "set-up".  The reference times its one and only set-up (HPCG/src/hpcg_benchmark.jl:35-40), and the rating
    # charges it per set; here the FIRST set-up of a process also pays for things that are the box's history rather than the
    # benchmark's work -- the driver wipes memory other processes have used when it hands it out (up to a second per 16 GiB) and
    # the context has to find its memory classes among the extents it gets (0.1 s on one lease, 2.4 s on another).  Both views
    # are reported: every set-up is run TWICE, the first time as the process finds the device
    # ("first encounter": pool acquisition, code-object loading, scratch allocations all inside the timed region -- the
    # reference's procedure, and the OFFICIAL rating of this report), the second time with the pool in hand ("pool in hand").
    # PA_HPCG_POOL_WARMUP=1 restores round 4's order (one untimed set-up first; then both views coincide).
    t_pool = time.perf_counter()
    if os.environ.get("PA_HPCG_POOL_WARMUP", "0") == "1":
        ctx.arena(build=True)
        warm = pc_setup(ranks, np_, levels, nx, ny, nz, ordering=opt_ordering)
        ctx.sync()
        del warm
    t_pool = time.perf_counter() - t_pool

    def elapsed(f):
        ctx.sync()
        t = time.perf_counter()
        out = f()
        ctx.sync()
        return time.perf_counter() - t, out

    pmax = lambda v: float(getany(reduction(max, pmap(lambda _r: v, ranks), destination="all")))
    # (the hierarchy keeps its raw columns: the optimised phase takes it over and cuts its colours' rows from it, below)
    t_setup_first, S_ref = elapsed(lambda: pc_setup(ranks, np_, levels, nx, ny, nz, ordering=ref_ordering, keep_raw_columns=True))
    del S_ref
    import gc
    gc.collect()                    # (the hierarchy's objects reference each other: without this the first one is still in HBM while
    #                                  the second is built, and the second timing pays for a fresh extent -- 0.40 s against 0.22)
    t_setup, S_ref = elapsed(lambda: pc_setup(ranks, np_, levels, nx, ny, nz, ordering=ref_ordering, keep_raw_columns=True))
    geom = hpcg_geometry(np_, levels, nx, ny, nz)
    A, b = S_ref.A_vec[-1], S_ref.r[-1]
    ref_timer = CgTimer()
    t_ref = 0.0
    for _ in range(2):
        dt, (x, normr0, normr, iters) = elapsed(lambda: ref_cg_(pzeros(A.col_partition), A, b, maxiter=ref_max_iters,
                                                                 tolerance=0.0, overlap=False, Pl=S_ref, timer=ref_timer))
        t_ref += dt
    ref_ms = ref_timer.resolve()
    ref_tol = normr / normr0
    del x

    # (one part: the V-cycle is replayed from a hipGraph -- same kernels, same bits, less launch cost on the coarse levels; the
    #  graph holds the addresses of the vectors it was recorded with, so the CG work vectors are allocated once: cg_work)
    # The optimised phase works on the SAME hierarchy as the reference phase, as in the reference's driver (one pc_setup serves
    # ref_cg! and opt_cg!, HPCG/src/hpcg_benchmark.jl:35-60): its set-up is what the optimisation adds -- the multicolour smoothers
    # and the restriction's row blocks, cut from the operators that are already in HBM (PA_HPCG_SHARE_HIERARCHY=0: a second
    # hierarchy from scratch, as rounds 3-4 did and charged).
    share = os.environ.get("PA_HPCG_SHARE_HIERARCHY", "1") != "0"
    mk_opt = lambda prev: pc_setup(ranks, np_, levels, nx, ny, nz, ordering=opt_ordering, graph=(np_ == 1),
                                   reuse=prev if share else None, keep_raw_columns=share)
    seq_smoothers = list(S_ref.gs_states) if share else None      # (the second timing must see what the first saw: below)
    t_opt_setup_first, S = elapsed(lambda: mk_opt(S_ref))
    del S_ref
    S_prev = S
    del S
    if share:
        # the second timing builds what the first built, from what the first was given: its own smoothers and row blocks go first, and
        # the hierarchy carries the reference phase's sequential smoothers again (the optimised set-up colours level by level from
        # their dependency levels; without them the re-timing would measure the discovery by rounds, which no first set-up runs)
        S_prev.gs_states, S_prev.row_blocks, S_prev._graphs = seq_smoothers, None, {}
    import gc
    gc.collect()
    t_opt_setup, S = elapsed(lambda: mk_opt(S_prev))
    del S_prev, seq_smoothers
    gc.collect()
    A, b = S.A_vec[-1], S.r[-1]
    work = cg_work(pzeros(A.col_partition), b, A)
    opt_n_iters, worst = ref_max_iters, 0.0
    for _ in range(2):
        dt, (x, normr0, normr, iters) = elapsed(lambda: opt_cg_(pzeros(A.col_partition), A, b, maxiter=10 * ref_max_iters,
                                                                 tolerance=ref_tol, Pl=S, fuse=True, work=work))
        if normr / normr0 > ref_tol:
            raise pa.PAError(f"the optimised solver did not reach the reference tolerance {ref_tol:.3e} in {iters} iterations")
        opt_n_iters, worst = max(opt_n_iters, iters), max(worst, dt)
    worst = pmax(worst)
    nr_sets = max(1, int(np.ceil(total_runtime / worst)))
    if max_sets is not None:
        nr_sets = min(nr_sets, max_sets)
    timer = CgTimer()
    norm_data, total = [], 0.0
    for _ in range(nr_sets):
        dt, (x, normr0, normr, iters) = elapsed(lambda: opt_cg_(pzeros(A.col_partition), A, b, maxiter=opt_n_iters,
                                                                 tolerance=0.0, Pl=S, timer=timer, fuse=True, work=work))
        norm_data.append(normr / normr0)
        total += dt
    ms = timer.resolve()
    times = {"total": pmax(total), "DDOT": ms["DDOT"] / 1e3, "WAXPBY": ms["WAXPBY"] / 1e3, "SPMV": ms["SPMV"] / 1e3,
             "MG": ms["MG"] / 1e3, "setup": pmax(t_setup), "opt_time": pmax(t_opt_setup),
             "ref_time": (ref_ms["SPMV"] + ref_ms["MG"]) / 1e3 / 2}
    print(f"[hpcg_driver] untimed pool warm-up {t_pool:.2f} s; set-up {times['setup']:.3f} s (reference ordering), {times['opt_time']:.3f} s (optimised), {nr_sets} sets of "
          f"{opt_n_iters} iterations in {times['total']:.2f} s, arena {ctx.arena()['acquired_gib']} GiB acquired in {ctx.arena()['map_ms']:.0f} ms",
          file=sys.stderr, flush=True)
    times_first = dict(times, setup=pmax(t_setup_first), opt_time=pmax(t_opt_setup_first))
    rep = hpcg_report(np_, times_first, levels, ref_max_iters, opt_n_iters, nr_sets, norm_data, geom)
    rep_warm = hpcg_report(np_, times, levels, ref_max_iters, opt_n_iters, nr_sets, norm_data, geom)
    rep["ratings"] = {
        "official": {"GFLOP/s": rep["Overview"]["GFLOP/s"], "setup_s": times_first["setup"], "opt_setup_s": times_first["opt_time"],
                     "what": "the reference's procedure (HPCG/src/hpcg_benchmark.jl:35-40, report_results.jl:140-147): each set-up timed the "
                             "first time this process runs it -- memory pool acquisition and classification, code-object loading and "
                             "scratch allocations inside the timed region"},
        "pool_in_hand": {"GFLOP/s": rep_warm["Overview"]["GFLOP/s"], "setup_s": times["setup"], "opt_setup_s": times["opt_time"],
                         "what": "the same set-ups timed a second time, with the context's memory pool acquired and every kernel "
                                 "loaded: what a second solve in the same process pays"},
        "raw": rep["GFLOP/s"]["Total"], "after_convergence_penalty": rep["GFLOP/s"]["Total_conv"]}
    print(f"[hpcg_driver] rating OFFICIAL (set-ups as first encountered: {times_first['setup']:.3f} + {times_first['opt_time']:.3f} s) "
          f"{rep['Overview']['GFLOP/s']:.1f} GFLOP/s; with the pool in hand ({times['setup']:.3f} + {times['opt_time']:.3f} s) "
          f"{rep_warm['Overview']['GFLOP/s']:.1f} GFLOP/s; raw {rep['GFLOP/s']['Total']:.1f}, after the convergence penalty "
          f"{rep['GFLOP/s']['Total_conv']:.1f}", file=sys.stderr, flush=True)
    rep["reference_phase"] = {"ref_tol": ref_tol, "seconds_per_set": t_ref / 2, "ordering": ref_ordering}
    rep["pool_warmup"] = {"seconds": t_pool, "what": "PA_HPCG_POOL_WARMUP=1 only: one untimed pc_setup of the optimised solver before the three phases "
                                                     "(round 4's order); by default nothing runs untimed and `ratings` carries both views"}
    rep["optimised_phase"] = {"ordering": opt_ordering, "iterations_to_ref_tol": opt_n_iters, "worst_set_seconds": worst}
    if output_type != "none" and getany(pmap(lambda r: r, ranks)) == 1:
        os.makedirs(output_folder, exist_ok=True)
        stamp = time.strftime("%Y-%m-%d_%H-%M-%S")
        path = os.path.join(output_folder, f"hpcg-benchmark_results{stamp}.{output_type}")
        with open(path, "w") as f:
            if output_type == "json":
                json.dump(rep, f, indent=1)
            else:
                f.write("########## Problem Summary  ##########\n")
                for k in ("procs", "nr_equations", "non_zeros", "geometry", "iter_data", "reproducibility_data", "times",
                          "flops", "GB/s", "GFLOP/s"):
                    f.write(f"{k}: {rep[k]}\n")
                f.write(f"HPCG result is VALID with a GFLOP/s rating of: {rep['Overview']['GFLOP/s']}\n")
        rep["file"] = path
    return rep


if __name__ == "__main__":
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    rt = float(sys.argv[3]) if len(sys.argv) > 3 else 10.0
    rep = hpcg_benchmark(pa.DebugArray(list(range(1, P + 1))), P, n, n, n, total_runtime=rt)
    print(json.dumps(rep["GFLOP/s"]), json.dumps(rep["Overview"]), json.dumps(rep["ratings"]))
