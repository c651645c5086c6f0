"""What a caller pays who keeps b and c on the HOST around every product (the boundary's operands are device-resident handles: this is
the cost of NOT using it that way): pa_vec_upload of b + mul! + pa_vec_download of c at 27-pt 256^3, pageable host memory.
  python tools/probe/pcie_rate.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
os.environ["PA_SPMV_VALUE_DICT"] = "0"
ctx = pa.context()
A, _ = pa.build_p_matrix(pa.DebugArray([1]), n, n, n, n, n, n, 1, 1, 1)
blk = A.matrix_partition.items[0].own_own
hb = np.random.default_rng(1).standard_normal(blk.n)
b = pa.DeviceVector(blk.n, 0).upload(hb)
c = pa.DeviceVector(blk.m, 0)
for _ in range(50): pa.spmv_(c, blk, b)
ctx.sync()
def timed(f, reps=10):
    ts = []
    for _ in range(reps):
        ctx.sync(); t = time.perf_counter(); f(); ctx.sync(); ts.append(time.perf_counter() - t)
    return min(ts) * 1e3
t_mul = timed(lambda: pa.spmv_(c, blk, b), 30)
t_up = timed(lambda: b.upload(hb))
t_down = timed(lambda: c.download())
def whole():
    b.upload(hb); pa.spmv_(c, blk, b); c.download()
t_all = timed(whole)
gb = blk.n * 8 / 1e9
print(f"27-pt {n}^3: product alone {t_mul:.3f} ms (host clock, one launch); upload of b ({gb*1e3:.0f} MB) {t_up:.2f} ms = {gb/t_up*1e3:.1f} GB/s; "
      f"download of c {t_down:.2f} ms = {gb/t_down*1e3:.1f} GB/s; upload + product + download {t_all:.2f} ms = {2*blk.nnz/t_all/1e6:.0f} GFLOP/s PCIe-inclusive")
