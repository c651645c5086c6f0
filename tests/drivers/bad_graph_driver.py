"""An ExchangeGraph only one end knows about (rank 1 sends to 2, rank 2 expects nothing; a third part sends both ways):
exchange() must ASSERT on every rank instead of leaving the sender in a blocking send / the receiver in a blocking
receive (src/primitives.jl:861-874, is_consistent).  Exit status 7 = every rank saw the assertion."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package  # noqa: E402

pa = load_package()
dist.init_process_group("gloo")


def body(distribute):
    P = dist.get_world_size()
    ranks = distribute(range(1, P + 1))
    me = pa.getany(ranks)
    # a consistent ring first: every part sends to its right neighbour (docs/examples.jl:58-91)
    snd = pa.pmap(lambda r: np.array([r % P + 1], np.int32), ranks)
    rcv = pa.pmap(lambda r: np.array([(r - 2) % P + 1], np.int32), ranks)
    got = pa.exchange(pa.pmap(lambda r: [10 * r], ranks), pa.ExchangeGraph(snd, rcv))
    assert pa.getany(got) == [10 * ((me - 2) % P + 1)]
    # now part 1 also sends to part 2's right neighbour... which does not expect it
    snd_bad = pa.pmap(lambda r: np.array([2, 3] if r == 1 else [r % P + 1], np.int32), ranks)
    data = pa.pmap(lambda r: [1, 2] if r == 1 else [10 * r], ranks)
    try:
        pa.exchange(data, pa.ExchangeGraph(snd_bad, rcv))
    except AssertionError as e:
        assert "inconsistent ExchangeGraph" in str(e)
        return 7
    return 0


status = pa.with_torchdist(body)
dist.barrier()
os._exit(status)
