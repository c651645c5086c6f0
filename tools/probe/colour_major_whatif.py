"""What would colour-major numbering buy a colour launch?  One colour's rows of the 27-point operator as a block in the natural
numbering (rows every other node, line and plane; what the smoother launches today) and the same block with rows and columns
renumbered colour by colour (the colour's rows contiguous); product time of each.  Usage: colour_major_whatif.py [n]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from __graft_entry__ import load_package
pa = load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 192
A, b = pa.build_p_matrix(pa.DebugArray([1]), n, n, n, n, n, n, 1, 1, 1, keep_host=True)
oo = pa.local_items(A.host_blocks)[0][0]
N = oo.m
ix, iy, iz = np.arange(N) % n, (np.arange(N) // n) % n, np.arange(N) // (n * n)
colour = (ix & 1) + 2 * (iy & 1) + 4 * (iz & 1)
order = np.argsort(colour, kind="stable")                 # colour-major: new position -> old row
P = np.empty(N, np.int64); P[order] = np.arange(N)        # old -> new
rp = oo.rowptr.astype(np.int64) - 1
lens = np.diff(rp)
ctx = pa.context()
def time_block(H, x, reps=40):
    blk = pa.DeviceCSR(H)
    y = pa.DeviceVector(H.m, 0)
    for _ in range(10): pa.spmv_(y, blk, x)
    ctx.sync(); t = time.perf_counter()
    for _ in range(reps): pa.spmv_(y, blk, x)
    ctx.sync()
    return (time.perf_counter() - t) / reps * 1e3, blk.encoding(), blk.stream_bytes()
x = pa.DeviceVector(N, 0).upload(np.random.default_rng(1).standard_normal(N))
for k in (0, 3, 7):
    rows = np.nonzero(colour == k)[0]
    sel = np.concatenate([np.arange(rp[r], rp[r + 1]) for r in rows[:0]]) if False else None
    mask = np.repeat(colour == k, lens)
    cols, vals = oo.colval[mask].astype(np.int64) - 1, oo.nzval[mask]
    l2 = np.where(colour == k, lens, 0)
    rp_nat = np.concatenate([[0], np.cumsum(l2)])
    H_nat = pa.HostCSR(N, N, (rp_nat + 1).astype(np.int32), (cols + 1).astype(np.int32), vals)
    # colour-major: rows of this colour are new rows P[rows] (contiguous); columns P[cols]
    l3 = np.zeros(N, np.int64); l3[P[rows]] = lens[rows]
    rp_cm = np.concatenate([[0], np.cumsum(l3)])
    H_cm = pa.HostCSR(N, N, (rp_cm + 1).astype(np.int32), (P[cols] + 1).astype(np.int32), vals)     # (rows keep their relative order)
    a, b_ = time_block(H_nat, x), time_block(H_cm, x)
    print(f"colour {k}: natural numbering {a[0]:.4f} ms {a[1]} {a[2] / 1e6:.0f} MB | colour-major {b_[0]:.4f} ms {b_[1]} {b_[2] / 1e6:.0f} MB | ratio {b_[0] / a[0]:.3f}", flush=True)
