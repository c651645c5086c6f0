"""27-point parts of the same size and different shapes: is it the distance between grid planes (nx*ny*8 bytes) that costs big parts their rate?
python tools/probe/shape_rate.py 256x256x1024 512x512x256 ..."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("PA_SPMV_VALUE_DICT", "0")
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
ctx = pa.context()
for shape in sys.argv[1:]:
    nx, ny, nz = (int(v) for v in shape.split("x"))
    A, b = pa.build_p_matrix(pa.DebugArray([1]), nx, ny, nz, nx, ny, nz, 1, 1, 1)
    blk = A.matrix_partition.items[0].own_own
    x = pa.pones(A.col_partition)
    y = pa.pzeros(A.row_partition)
    for _ in range(300): pa.mul_c_(y, A, x)
    e0 = ctx.event().record(L.STREAM_COMPUTE)
    for _ in range(30): pa.mul_c_(y, A, x)
    e1 = ctx.event().record(L.STREAM_COMPUTE); ctx.sync()
    ms = e0.elapsed_ms(e1) / 30
    print(f"{shape:16s} rows {blk.m:10d} line {nx * 8 / 1024:6.1f} KiB plane {nx * ny * 8 / 1024:8.1f} KiB  mul! {ms:7.3f} ms = {2 * blk.nnz / ms / 1e6:6.0f} GFLOP/s, moved {(blk.stream_bytes() + 16 * blk.m) / ms / 1e6:5.0f} GB/s  {blk.encoding()}", flush=True)
    del A, b, blk, x, y
