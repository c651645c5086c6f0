"""Helpers of the -m gpu parity modules (tests/test_gpu_*.py; one module per row family of SURVEY 8 since round 4)."""
import ctypes as C
import functools
import os

import numpy as np
import pathlib
import pytest

from __graft_entry__ import load_package

pa = load_package()



def hpcg_driver():
    """tools/hpcg_driver.py: HPCG's benchmark driver and report (a tool beside the probes, not part of the package)."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "hpcg_driver.py")
    spec = importlib.util.spec_from_file_location("hpcg_driver", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def ranks(n):
    return pa.DebugArray(range(1, n + 1))


def upload(host_parts, index_partition):
    it = iter(host_parts)
    return pa.pvector_from_function(lambda ind: next(it), index_partition)


def _hand(golden):
    c = golden["p_vector_local_indices"]
    return c, pa.DebugArray([pa.LocalIndices(c["n"], p + 1, local_to_global=g, local_to_owner=o)
                             for p, (g, o) in enumerate(zip(c["local_to_global"], c["local_to_owner"]))])


def _oracle_mul(orc, Ao, xo):
    yo = [np.zeros(r.n_local) for r in Ao.rows]
    orc.mul(yo, Ao, [v.copy() for v in xo])
    return yo


def _random_csr(rng, m, n, row_len):
    I = np.repeat(np.arange(1, m + 1), row_len)
    J = np.concatenate([rng.choice(n, size=k, replace=False) + 1 if k else np.zeros(0, int) for k in row_len])
    V = rng.standard_normal(len(I))
    return pa.compresscoo(I, J, V, m, n)


def _fem_error(x, S, A):
    """norm(x - x_hat) over own values, x_hat from setup_exact_solution on A's column partition (fem_example.jl:284-288)."""
    xh = pa.pmap(lambda s, c: pa.fem_example.setup_exact_solution(s, S["params"], c), S["spaces"], A.col_partition)
    return sum(float(np.sum((xv - h[:len(xv)]) ** 2)) for xv, h in zip(x.own_values().items, xh.items)) ** 0.5


def _fem_cg(A, b):
    x, r0, r, it = pa.ref_cg_(pa.pzeros(A.col_partition), A, b, maxiter=400, tolerance=1.4901161193847656e-08)
    assert it < 400
    return x


def _encoding_cases(orc):
    """Blocks that exercise every branch of the column encoders: stencils (row patterns), a 7-point Laplacian (patterns + a few
    explicit chunks), ragged and empty rows, rows longer than a chunk and longer than a pattern, columns all over (32-bit
    chunks), banded random rows (16-bit windows, x-window groups), a block that is mostly empty rows (row-compacted, patterns
    with a row-id stride: one Gauss-Seidel colour) and a mixed block (half stencil, half random)."""
    rng = np.random.default_rng(11)

    def csr(m, n, rows):
        rp = np.zeros(m + 1, np.int64)
        for i, r in enumerate(rows):
            rp[i + 1] = rp[i] + len(r)
        cols = np.concatenate([np.asarray(r, np.int64) for r in rows]) if rp[-1] else np.zeros(0, np.int64)
        return pa.HostCSR(m, n, (rp + 1).astype(np.int32), (cols + 1).astype(np.int32), rng.standard_normal(int(rp[-1])))
    Ao, _, _ = orc.hpcg_build_p_matrix(24, 24, 24, 1, 1, 1)
    oo = Ao.blocks[0].own_own
    yield "27-point 24^3", pa.HostCSR(oo.m, oo.n, oo.rowptr, oo.colval, oo.nzval)
    Io, Jo, Vo, rows, _ = orc.laplacian_fdm_fast((40, 40, 40), (1, 1, 1))
    B = orc.psparse_from_coo(Io, Jo, Vo, rows).blocks[0].own_own
    yield "7-point 40^3", pa.HostCSR(B.m, B.n, B.rowptr, B.colval, B.nzval)
    m = 60000
    yield "ragged rows", csr(m, m, [np.sort(rng.choice(m, size=int(k), replace=False)) for k in rng.integers(0, 40, size=m)])
    rows = [np.sort(np.clip(i + rng.integers(-1500, 1500, size=16), 0, m - 1)) for i in range(m)]
    rows = [np.unique(r) for r in rows]
    yield "banded random rows", csr(m, m, rows)
    rows = [np.arange(max(0, i - 1), min(m, i + 2)) for i in range(m)]
    rows[100] = np.sort(rng.choice(m, size=5000, replace=False))           # a row longer than a chunk
    rows[2000] = np.sort(rng.choice(m, size=40, replace=False))            # longer than a pattern
    rows[3000] = np.zeros(0, np.int64)
    yield "tridiagonal with long rows", csr(m, m, rows)
    oo_rows = [oo.colval[oo.rowptr[r] - 1:oo.rowptr[r + 1] - 1] - 1 if (r % 2 == 0 and (r // 24) % 2 == 0 and (r // 576) % 2 == 0) else np.zeros(0, np.int64)
               for r in range(oo.m)]
    yield "one colour of the 27-point operator (row-compacted, strided patterns)", csr(oo.m, oo.n, oo_rows)
    half = [oo.colval[oo.rowptr[r] - 1:oo.rowptr[r + 1] - 1] - 1 if r < oo.m // 2 else np.sort(rng.choice(oo.n, size=20, replace=False))
            for r in range(oo.m)]
    yield "half stencil, half scattered rows", csr(oo.m, oo.n, half)
    yield "scattered rows", csr(20000, 300000, [np.sort(rng.choice(300000, size=12, replace=False)) for _ in range(20000)])


def oracle_mul(orc, Ao, xo):
    return _oracle_mul(orc, Ao, xo)


def reload_switches():
    """The library reads its product-path switches (PA_PUSH, PA_MUL_FUSED, ...) when a context is created: after changing the
    environment inside a test, every live context reads them again."""
    import pa_amd.p_vector as pv
    for c in pv.all_contexts():
        c.reload_env()


class env:
    """with env(PA_X="0"): ... -- environment switches of the library for the duration of a block."""

    def __init__(self, **kv):
        self.kv = kv

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.kv}
        os.environ.update(self.kv)
        reload_switches()

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        reload_switches()
        return False


__all__ = [n for n in dir() if not n.startswith("__")]
