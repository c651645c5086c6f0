"""A part of 2^31 stored entries or more (a chain of row slabs) on pattern-ELL and on the row-split kernel: 27-pt n^3, fp64 streams.
  python tools/probe/big_pell.py 448"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
n = int(sys.argv[1]) if len(sys.argv) > 1 else 448
os.environ["PA_SPMV_VALUE_DICT"] = "0"
ctx = pa.context()
ys = {}
for big in ("1", "0"):
    os.environ["PA_SPMV_PELL_BIG"] = big
    t = time.perf_counter()
    A, _ = pa.build_p_matrix(pa.DebugArray([1]), n, n, n, n, n, n, 1, 1, 1)
    ctx.sync(); ts = time.perf_counter() - t
    blk = A.matrix_partition.items[0].own_own
    x = pa.DeviceVector(blk.n, 0).upload(np.random.default_rng(1).standard_normal(blk.n))
    y = pa.DeviceVector(blk.m, 0)
    for _ in range(60): pa.spmv_(y, blk, x)
    ctx.sync(); tt = []
    for r in range(4):
        e0 = ctx.event().record(L.STREAM_COMPUTE)
        for _ in range(40): pa.spmv_(y, blk, x)
        e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync(); tt.append(e0.elapsed_ms(e1) / 40)
    ys[big] = y.download()
    print(f"27-pt {n}^3 ({blk.nnz / 1e9:.2f} G entries, info {blk.info()}) PA_SPMV_PELL_BIG={big}: pell {blk.pell()['mode']}, set-up {ts:.1f} s, "
          f"{min(tt):.3f} ms = {2 * blk.nnz / min(tt) / 1e6:.0f} GFLOP/s, arena {ctx.arena()['class_gib']}", flush=True)
    del A, blk, x, y
print("bit-identical:", bool(np.array_equal(ys["1"], ys["0"])))
