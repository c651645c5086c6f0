"""The one-part-per-process device path on the 1-GPU box: P ranks share cuda:0, transport = host-staged gloo.
(RCCL itself needs distinct GPUs: test_rccl_exchange_between_distinct_gpus and test_bench_two_gpus_over_rccl run where there
are some and skip otherwise; on one GPU its call sequence is pinned by test_rccl_single_rank_loopback.)"""
import pytest

from test_multiprocess_gloo import _run

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nproc", [2, 4, 8])
def test_device_path_one_part_per_process(nproc):
    _run("device_path_driver.py", nproc, {"PA_TRANSPORT": "host"})


@pytest.mark.parametrize("nproc", [2, 4, 8])
def test_device_path_over_the_ipc_push_transport(nproc):
    """PA_TRANSPORT=ipc (csrc/pa_push.hip): one part per process, every pack kernel storing straight into its neighbours' receive
    buffers (hipIpcOpenMemHandle), arrival and flow control by sequence numbers in device memory -- a device-only transport
    that, unlike RCCL, also runs with all ranks on ONE GPU.  mul! (composed and pa_mul5 with own x ghost reading the receive
    buffer), 12 products in a row, the alpha/beta form, the transpose product, consistent!, assemble!, psparse!: bit-exact
    against the sequential oracle on every rank."""
    _run("device_path_driver.py", nproc, {"PA_TRANSPORT": "ipc"})


def test_random_matrices_over_the_ipc_push_transport():
    """tests/fuzz/fuzz_dist_driver.py with PA_TRANSPORT=ipc, 4 ranks on this box's GPU: 40 random PSparseMatrices -- hundreds of
    exchange plans made and dropped, each with a region cut from the exported pool chunks (a region of its own per plan made
    hipIpcGetMemHandle fail after a few dozen: found by the round-4 fuzz campaign)."""
    import os
    import subprocess
    import sys
    from test_multiprocess_gloo import ROOT, _free_port
    e = dict(os.environ, OMP_NUM_THREADS="1", PA_HOST_THREADS="1", PA_TRANSPORT="ipc")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=4", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "fuzz", "fuzz_dist_driver.py"), "40", "31500"]
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "0 with mismatches" in r.stdout, (r.stdout + r.stderr)[-3000:]


def _gpus():
    import torch
    return torch.cuda.device_count()


@pytest.mark.parametrize("nproc", [2, 4, 8])
def test_rccl_exchange_between_distinct_gpus(nproc):
    """The RCCL transport itself (csrc/pa_rccl.cpp: one ncclGroup of ncclSend/ncclRecv per exchange, src/mpi_array.jl:575-614):
    one part per process, one GPU per process -- mul!, consistent!, assemble!, dot and psparse! against the sequential oracle,
    bit-exact, exactly as the host-staged runs above.  Needs `nproc` GPUs in this box: skipped on the 1-GPU lease the builder
    has, runs wherever the suite meets a multi-GPU node."""
    if _gpus() < nproc:
        pytest.skip(f"{_gpus()} GPU(s) visible, {nproc} needed")
    _run("device_path_driver.py", nproc, {"PA_TRANSPORT": "rccl"})


@pytest.mark.parametrize("nproc", [2, 4, 8])
def test_random_matrices_over_rccl_between_distinct_gpus(nproc):
    """tests/fuzz/fuzz_dist_driver.py with the RCCL transport: 60 random PSparseMatrices, every rank on its own GPU checking its
    part of mul!, the alpha/beta form, consistent!, assemble!, dot and the CG loops against the sequential oracle."""
    if _gpus() < nproc:
        pytest.skip(f"{_gpus()} GPU(s) visible, {nproc} needed")
    import os
    import subprocess
    import sys
    from test_multiprocess_gloo import ROOT, _free_port
    e = dict(os.environ, OMP_NUM_THREADS="1", PA_HOST_THREADS="1", PA_TRANSPORT="rccl")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "fuzz", "fuzz_dist_driver.py"), "60", "31000"]
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "0 with mismatches" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_bench_two_gpus_over_rccl():
    """`bench.py --gpus 2` the way the driver launches it, on two GPUs: the line says RCCL and two ranks."""
    if _gpus() < 2:
        pytest.skip(f"{_gpus()} GPU(s) visible, 2 needed")
    r, d = _bench(2, {}, ("--no-cpu-baseline",))
    assert r.returncode == 0 and d is not None, r.stdout[-2000:] + r.stderr[-3000:]
    assert d["n_gpus"] == 2 and d["config"]["transport"].startswith("rccl") and d["config"]["rccl_ranks_seen"] == 2, d["config"]


def _bench(nproc, env, args=(), grid="32", timeout=600):
    import json
    import os
    import subprocess
    import sys
    from test_multiprocess_gloo import ROOT, _free_port
    e = dict(os.environ, OMP_NUM_THREADS="1", PA_HOST_THREADS="1")
    e.update(env)
    # the child ranks share this box's ONE GPU with this pytest process: hand back what earlier tests of the session left in its arena
    # (an idle context keeps up to 24 GiB of extents) before eight ranks take 20 GiB each at --grid 256
    try:
        import gc
        gc.collect()
        from gpu_helpers import pa as _pa
        import pa_amd.p_vector as _pv
        for c in _pv.all_contexts():
            c.sync()
            c.arena_release()
    except Exception:                                   # noqa: BLE001
        pass
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), "--steps", "5", "--warmup", "1",
           "--grid", grid, "--cg-iters", "3", "--cpu-seconds", "0.5", *args]
    r = subprocess.run(cmd, cwd=ROOT, env=e, capture_output=True, text=True, timeout=timeout)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    return r, (json.loads(lines[-1]) if lines else None)


def test_bench_line_of_a_multi_rank_run_carries_every_field():
    """`bench.py --gpus 2` with both ranks on this box's one GPU (host-staged transport): the line carries
    the transport, the overlap on/off comparison, the stream priorities, the moved-bytes roofline and a CPU baseline on 2
    cores (one pinned process per part, exchanging over gloo)."""
    r, d = _bench(2, {"PA_TRANSPORT": "host", "PA_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0 and d is not None, r.stdout[-2000:] + r.stderr[-3000:]
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["transport"].startswith("host-staged")
    ab = d["transport_ab"]                                  # round 4: the ipc push transport timed beside the headline's
    assert ab["parity_gate_over_ipc_push"] is True and ab["ms_per_step_ipc_push"] > 0 and ab["headline_transport"] == "host"
    assert d["config"]["overlap"] is True and set(d["overlap"]) >= {"ms_per_step_on", "ms_per_step_off"}
    pr = d["config"]["stream_priority"]
    assert pr["comm"] == pr["greatest"] and pr["compute"] == pr["least"]
    rf = d["roofline"]
    assert rf["moved_bytes_per_launch"] < rf["algorithmic_bytes_per_launch"] and 0 < rf["frac"] <= 1.0 and rf["median_launch_ms"] > 0
    cb = d["cpu_baseline"]
    assert cb["cores"] == 2 and cb["kind"] == "port" and cb["ms_per_mul"] > 0 and cb["c1_debugarray"]["cores"] == 1
    assert d["cg_loop"]["ms_per_iteration_opt_cg"] > 0
    r2, d2 = _bench(2, {"PA_TRANSPORT": "host", "PA_BENCH_BACKEND": "gloo"}, ("--no-overlap", "--no-cpu-baseline"))
    assert r2.returncode == 0 and d2["config"]["overlap"] is False and d2["overlap"]["headline_uses"] == "off"


def test_bench_refuses_to_downgrade_the_rccl_transport_silently():
    """Two ranks on ONE GPU cannot form an RCCL communicator: the run must end with a non-zero status and no line --
    never a line that measured another transport under the RCCL row's name.  With PA_ALLOW_TRANSPORT_FALLBACK=1 it
    continues on torch.distributed p2p and says so (that, too, needs distinct GPUs: only the refusal is checked here)."""
    # (torch's own process group on gloo, so that the first RCCL communicator of the run is libpa_hip's: ncclCommInitRank
    # with two ranks on one device fails, which is the situation the refusal exists for)
    r, d = _bench(2, {"PA_TRANSPORT": "rccl", "PA_BENCH_BACKEND": "gloo", "PA_BENCH_WATCHDOG_S": "240"}, ("--no-cpu-baseline",))
    assert r.returncode != 0 and d is None, r.stdout[-2000:]
    assert "refusing to downgrade" in r.stderr, r.stderr[-3000:]


def test_a_rank_lost_in_an_optional_section_does_not_cost_the_line():
    """Everything after the headline measurement is optional: rank 1 never returns from the CG section (injected), every
    rank's section timer fires, rank 0 prints the line it has -- headline complete, `cg_loop` absent, the section named --
    and the job ends with status 0."""
    r, d = _bench(2, {"PA_TRANSPORT": "host", "PA_BENCH_BACKEND": "gloo", "PA_BENCH_FAULT": "CG loop:hang:1",
                      "PA_BENCH_SECTION_TIMEOUT_S": "10"}, ("--no-cpu-baseline",))
    assert r.returncode == 0 and d is not None, r.stdout[-2000:] + r.stderr[-3000:]
    assert d["value"] > 0 and "cg_loop" not in d and d["optional_sections_unfinished"] == ["CG loop"]
    assert "overlap" in d and 0 < d["roofline"]["frac"] <= 1.0


def test_a_rank_lost_in_the_fused_product_section_leaves_the_line_of_the_separate_launches():
    """VERDICT r05 "Next" #1a: for N > 1 the parity gate and the first timed loop run on the conservative product (separate launches,
    PA_MUL_FUSED=0) and rank 0 holds a complete line BEFORE the library's default -- one launch per part with the exchange waited for
    inside the launch -- is gated and timed.  Rank 1 never returns from that section (injected): the line goes out with the value of
    the separate launches, says so, names the section, status 0."""
    r, d = _bench(2, {"PA_TRANSPORT": "ipc", "PA_BENCH_BACKEND": "gloo", "PA_BENCH_FAULT": "fused product:hang:1",
                      "PA_BENCH_SECTION_TIMEOUT_S": "10"}, ("--no-cpu-baseline",))
    assert r.returncode == 0 and d is not None, r.stdout[-2000:] + r.stderr[-3000:]
    assert d["value"] > 0 and d["config"]["product_path"].startswith("separate launches"), d["config"]
    assert d["optional_sections_unfinished"] == ["fused product"] and "fused_ab" not in d
    assert d["parity_gate"].startswith("A*1==b") and 0 < d["roofline"]["frac"] <= 1.0


def test_the_fused_product_becomes_the_value_only_behind_its_own_gate():
    """The same run without the fault: `fused_ab` holds both loops and both gates; `value` is the faster product that passed --
    config.product_path and fused_ab.value_uses agree, and the one-launch product really was one launch with the exchange inside."""
    r, d = _bench(2, {"PA_TRANSPORT": "ipc", "PA_BENCH_BACKEND": "gloo"}, ("--no-cpu-baseline",))
    assert r.returncode == 0 and d is not None, r.stdout[-2000:] + r.stderr[-3000:]
    fa = d["fused_ab"]
    assert fa["parity_gate_one_launch"] and fa["mul_as_one_launch_rank0"] and fa["exchange_inside_the_launch_rank0"], fa
    assert fa["ms_per_step_one_launch"] > 0 and fa["ms_per_step_separate_launches"] > 0
    one = fa["value_uses"] == "one launch"
    assert one == (fa["ms_per_step_one_launch"] < fa["ms_per_step_separate_launches"])
    assert d["config"]["product_path"].startswith("one launch per part" if one else "separate launches")
    assert abs(d["ms_per_step"] - (fa["ms_per_step_one_launch"] if one else fa["ms_per_step_separate_launches"])) < 1e-3


def test_bench_eight_ranks_sharing_the_gpu_print_a_self_diagnosing_line():
    """VERDICT r02 #9 (first-contact insurance for the 8-GPU record): `bench.py --gpus 8 --grid 32` the way the driver
    launches it, all eight ranks on this box's one GPU over the host-staged transport: part grid (2,2,2), every part has
    its 7 neighbours, the line carries one `per_rank` entry per rank (device, communicator size seen, neighbours, ghosts,
    what its arena holds, its own ms per step with the exchange under own x own and with the exchange first), and no
    rank's context holds HBM it does not use (eight ranks next to each other on one device: the on-demand arena)."""
    r, d = _bench(8, {"PA_TRANSPORT": "host", "PA_BENCH_BACKEND": "gloo"}, ("--no-cpu-baseline",))
    assert r.returncode == 0 and d is not None, r.stdout[-2000:] + r.stderr[-3000:]
    assert d["n_gpus"] == 8 and "(2,2,2)" in d["config"]["workload"] and d["config"]["transport"].startswith("host-staged")
    assert d["parity_gate"].startswith("A*1==b")
    pr = d["per_rank"]
    assert [e["rank"] for e in pr] == list(range(8)) and d["per_rank_transport"] == "host"
    for e in pr:
        assert e["neighbors_snd"] == 7 and e["neighbors_rcv"] == 7 and e["ghosts"] == 3 * 32 * 32 + 3 * 32 + 1, e
        assert e["arena_held_gib"] <= 1.0, e                         # (32^3 rows per part: nothing big enough to start an arena)
        assert e["ms_per_step_overlap_on"] > 0 and e["ms_per_step_overlap_off"] > 0 and e["own_own_launch_ms"] > 0, e
    assert set(d["overlap"]) >= {"ms_per_step_on", "ms_per_step_off"}
    assert abs(d["value"] - 2 * d["config"]["nnz_per_part"] * 8 / d["ms_per_step"] / 1e6) / d["value"] < 0.05   # (parts differ by their ghosts)


@pytest.mark.gpu_extended
def test_bench_line_of_an_8_rank_run_over_the_ipc_push_transport():
    """`bench.py --gpus 8` with all ranks on this box's one GPU over the device-only ipc transport: the parity gate passes on every
    rank (A*1 == b, ghosts == owners) and the line names the transport."""
    r, d = _bench(8, {"PA_TRANSPORT": "ipc", "PA_BENCH_BACKEND": "gloo"}, ("--no-cpu-baseline",))
    assert r.returncode == 0 and d is not None, r.stdout[-2000:] + r.stderr[-3000:]
    assert d["n_gpus"] == 8 and d["config"]["transport"].startswith("ipc"), d["config"]
    assert len(d["per_rank"]) == 8 and all(p["neighbors_snd"] == 7 for p in d["per_rank"])
    fa = d.get("fused_ab")
    assert fa and fa["mul_as_one_launch_rank0"] and fa["exchange_inside_the_launch_rank0"], fa      # round 5: one launch per part


def test_bench_config_4_at_full_size_eight_ranks_on_one_gpu_over_ipc():
    """VERDICT r04 #3b: `bench.py --gpus 8 --grid 256` -- BASELINE config 4's parts, (2,2,2) x 256^3 rows, 197 377 ghosts each --
    with all eight ranks on this box's ONE GPU (46 GB) over the ipc push transport, every mul! one launch per rank with the exchange
    inside it: the parity gate at full size, a complete line, every rank's row.  The line goes to gpurun_out/ (copied to profiles/)."""
    import json
    import os
    from test_multiprocess_gloo import ROOT
    r, d = _bench(8, {"PA_TRANSPORT": "ipc", "PA_BENCH_BACKEND": "gloo", "PA_BENCH_RAMP_S": "0.3"}, ("--no-cpu-baseline", "--no-extra"), grid="256",
                  timeout=1500)
    assert r.returncode == 0 and d is not None, r.stdout[-2000:] + r.stderr[-3000:]
    assert d["n_gpus"] == 8 and d["config"]["rows_per_part"] == 256 ** 3 and d["config"]["ghosts_per_part"] == 197377, d["config"]
    assert len(d["per_rank"]) == 8 and not any(p.get("missing") for p in d["per_rank"])
    assert d["fused_ab"]["exchange_inside_the_launch_rank0"]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "bench_n8_one_gpu_ipc_256.json"), "w") as f:
        json.dump(d, f, indent=1)
