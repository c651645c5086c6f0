"""One rank raises inside with_torchdist: the job must exit non-zero (with_mpi -> MPI.Abort, src/mpi_array.jl:64-83)."""
import os
import sys

import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pa = load_package()
dist.init_process_group("gloo")


def body(distribute):
    ranks = distribute(range(1, dist.get_world_size() + 1))
    if pa.getany(ranks) == int(os.environ.get("PA_FAIL_RANK", "1")) + 1:
        raise RuntimeError("boom on one rank")
    return pa.getany(pa.gather(ranks, destination="all"))


try:
    pa.with_torchdist(body)
except BaseException:
    os._exit(1)
