"""The one-part-per-process device path on the 1-GPU box: P ranks share cuda:0, transport = host-staged gloo.
(RCCL itself needs distinct GPUs; its call sequence is pinned by test_rccl_single_rank_loopback.)"""
import pytest

from test_multiprocess_gloo import _run

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nproc", [2, 4, 8])
def test_device_path_one_part_per_process(nproc):
    _run("device_path_driver.py", nproc, {"PA_TRANSPORT": "host"})
