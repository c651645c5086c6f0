"""N>1 path on CPU: one part per process over torch.distributed/gloo (world_size 2 and 4).
Spawned like the reference spawns `mpiexec -n 4 julia driver.jl` (test/mpi_array/run_mpi_driver.jl:3-15)."""
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(driver, nproc, extra_env=None):
    env = dict(os.environ, OMP_NUM_THREADS="1", PA_HOST_THREADS="1")
    env.update(extra_env or {})
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "drivers", driver)]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.parametrize("nproc", [2, 4])
def test_host_setup_one_part_per_process(nproc):
    _run("host_setup_driver.py", nproc)


def test_exception_on_one_rank_fails_the_job():
    """test/mpi_array/exception_tests.jl:5-11: an error on one rank must bring the whole job down."""
    env = dict(os.environ, OMP_NUM_THREADS="1", PA_FAIL_RANK="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "drivers", "exception_driver.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


def test_inconsistent_exchange_graph_asserts_on_every_rank_instead_of_hanging():
    """an edge only one end knows (a send nobody receives) used to leave the sender in a blocking call for the
    group's timeout; the cheap consistency check that is now on by default makes every rank raise together."""
    env = dict(os.environ, OMP_NUM_THREADS="1")
    env.pop("PA_CHECK_EXCHANGE_GRAPHS", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "drivers", "bad_graph_driver.py")]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "exitcode: 7" in (r.stdout + r.stderr).replace("exitcode  : 7", "exitcode: 7"), (r.stdout + r.stderr)[-3000:]
