"""Helpers shared by the -m gpu test modules (one module per row family of SURVEY 8; tests/gpu_helpers.py serves the modules split off the former test_gpu_parity.py)."""
import numpy as np

from __graft_entry__ import load_package

pa = load_package()


def ranks(n):
    return pa.DebugArray(range(1, n + 1))


def upload(host_parts, index_partition):
    it = iter(host_parts)
    return pa.pvector_from_function(lambda ind: next(it), index_partition)


def oracle_mul(orc, Ao, xo):
    yo = [np.zeros(r.n_local) for r in Ao.rows]
    orc.mul(yo, Ao, [v.copy() for v in xo])
    return yo


class env:
    """with env(PA_X="0"): ... -- environment switches of the library for the duration of a block."""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        import os
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)

    def __exit__(self, *a):
        import os
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        return False
