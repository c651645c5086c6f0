"""What would the 27-point 256^3 product gain if a grid line of x did not start every 2048 bytes?  The same operator with every line of the
vectors padded from 256 to 256 + pad entries (the pad unknowns have empty rows and no columns): a (256+pad) x 256 x 256 device layout.
python tools/probe/padded_lines.py [pad ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("PA_SPMV_VALUE_DICT", "0")
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
from pa_amd.gallery import build_split_blocks_fused
ctx = pa.context()
n = 256
rows = pa.uniform_partition(pa.DebugArray([1]), (1, 1, 1), (n, n, n)).items[0]
_, oo, _, _ = build_split_blocks_fused(rows, n, n, n, n, n, n)
N = n ** 3
rp = oo.rowptr.astype(np.int64) - 1
lens = np.diff(rp).astype(np.int32)
col0 = oo.colval.astype(np.int64) - 1
val = oo.nzval
rng = np.random.default_rng(1)
xh = rng.standard_normal(N)
for pad in [int(a) for a in sys.argv[1:]] or [0, 8]:
    w = n + pad
    Nd = w * n * n
    pos = lambda i: i + pad * (i // n)                     # device position of unknown i
    lens_d = np.zeros(Nd, np.int32)
    lens_d[pos(np.arange(N, dtype=np.int64))] = lens
    rp_d = np.concatenate(([1], 1 + np.cumsum(lens_d, dtype=np.int64))).astype(np.int32)
    H = pa.HostCSR(Nd, Nd, rp_d, (pos(col0) + 1).astype(np.int32), val)
    blk = pa.DeviceCSR(H)
    xd_h = np.zeros(Nd); xd_h[pos(np.arange(N, dtype=np.int64))] = xh
    x, y = pa.DeviceVector(Nd, 0).upload(xd_h), pa.DeviceVector(Nd, 0)
    for _ in range(400): pa.spmv_(y, blk, x)
    best = []
    for rep in range(3):
        e0 = ctx.event().record(L.STREAM_COMPUTE)
        for _ in range(50): pa.spmv_(y, blk, x)
        e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
        best.append(e0.elapsed_ms(e1) / 50)
    got = y.download()[pos(np.arange(N, dtype=np.int64))]
    if pad == 0: ref = got
    print(f"pad {pad:3d}: {min(best):.4f} ms = {2 * oo.nnz / min(best) / 1e6:6.0f} GFLOP/s  encoding {blk.encoding()}  same bits as pad 0: {np.array_equal(got, ref)}", flush=True)
    del blk, x, y, H
