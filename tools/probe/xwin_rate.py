"""Banded rows without a pattern: the x-window launch (pa_spmv_xwin.h) against k_spmv_rowsplit on the same block."""
import os, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
ctx = pa.context()
tag = sys.argv[1] if len(sys.argv) > 1 else "product"

def rate(name, H):
    x = pa.DeviceVector(H.n, 0).upload(np.random.default_rng(1).standard_normal(H.n))
    out = []
    for sw in ("0", None, "2"):                       # row split only / the library's choice / x windows wherever groups exist
        if sw is None: os.environ.pop("PA_SPMV_XWIN", None)
        else: os.environ["PA_SPMV_XWIN"] = sw
        blk = pa.DeviceCSR(H)
        y = pa.DeviceVector(H.m, 0)
        import time
        t_end = time.perf_counter() + 0.25               # the GPU idled while the host built the matrix: 250 ms of launches
        while time.perf_counter() < t_end:               # bring it back to its working clocks (60 launches are not enough
            for _ in range(20): pa.spmv_(y, blk, x)      # after ten seconds of idle: 0.70 ms measured for a 0.144 ms kernel)
            ctx.sync()
        e0 = ctx.event().record(L.STREAM_COMPUTE)
        for _ in range(50): pa.spmv_(y, blk, x)
        e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
        ms = e0.elapsed_ms(e1) / 50
        out.append((ms, y.download(), blk.xwin(), (blk.memory_class(), x.memory_class(), y.memory_class())))
        del blk, y
    os.environ.pop("PA_SPMV_XWIN", None)
    same = np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][1], out[2][1])
    alg = (H.nnz * 12 + H.m * 20) / 1e6
    print(f"[{tag:10s}] {name:40s} row split {out[0][0]:7.4f} ms {alg/out[0][0]:6.0f} GB/s | default {out[1][0]:7.4f} ms {alg/out[1][0]:6.0f} GB/s "
          f"| forced {out[2][0]:7.4f} ms {alg/out[2][0]:6.0f} GB/s  same bits {same}  default {out[1][2]}  classes val/x/y {[o[3] for o in out]}", flush=True)

rng = np.random.default_rng(0)
m = 4_000_000
bands = tuple(int(b) for b in sys.argv[2].split(",")) if len(sys.argv) > 2 else (8000, 2000, 500, 100)
for band in bands:
    base = np.repeat(np.arange(m), 16)
    col = np.sort(np.clip(base + rng.integers(-band, band, size=m * 16), 0, m - 1).reshape(m, 16), axis=1).ravel().astype(np.int32) + 1
    H = pa.HostCSR(m, m, (1 + 16 * np.arange(m + 1)).astype(np.int32), col, rng.standard_normal(m * 16))
    rate(f"4M rows x 16 within +-{band}", H)
    del H, col, base
if len(sys.argv) > 3: sys.exit(0)
m = 2_000_000
lens = rng.integers(1, 40, m)
rp = np.concatenate([[1], 1 + np.cumsum(lens)]).astype(np.int32)
rows = np.repeat(np.arange(m), lens)
colr = np.clip(rows + rng.integers(-2000, 2000, size=len(rows)), 0, m - 1)
order = np.lexsort((colr, rows))
rate("2M ragged rows (1..39) within +-2000", pa.HostCSR(m, m, rp, (colr[order] + 1).astype(np.int32), rng.standard_normal(len(rows))))
m = 2_000_000
lens = rng.integers(3, 9, m)
rp = np.concatenate([[1], 1 + np.cumsum(lens)]).astype(np.int32)
rows = np.repeat(np.arange(m), lens)
colr = np.clip(rows + rng.integers(-1000, 1000, size=len(rows)), 0, m - 1)
order = np.lexsort((colr, rows))
rate("2M short rows (3..8) within +-1000", pa.HostCSR(m, m, rp, (colr[order] + 1).astype(np.int32), rng.standard_normal(len(rows))))

# a Q1 mesh numbered at random, then renumbered by reverse Cuthill-McKee: what an unstructured-mesh code hands over
import scipy.sparse as sp
from scipy.sparse.csgraph import reverse_cuthill_mckee
for dims in ((600, 4000), (1000, 4000)):
    I, J, V, rows_, cols_ = pa.laplacian_fem(dims, (1, 1), pa.DebugArray([1]))
    n = dims[0] * dims[1]
    perm = np.random.default_rng(29).permutation(n)
    Ip, Jp = perm[I.items[0] - 1], perm[J.items[0] - 1]
    G = sp.csr_matrix((np.ones(len(Ip), np.int8), (Ip, Jp)), shape=(n, n))
    order = reverse_cuthill_mckee(G, symmetric_mode=True)
    new_id = np.empty(n, np.int64); new_id[order] = np.arange(n)
    Hc = pa.compresscoo(new_id[Ip] + 1, new_id[Jp] + 1, V.items[0], n, n)
    band = int(np.max(np.abs(np.repeat(np.arange(n), np.diff(Hc.rowptr)) - (Hc.colval - 1))))
    rate(f"Q1 FEM {dims[0]}x{dims[1]} nodes, random then RCM (band {band})", Hc)
    del I, J, V, G, Hc
