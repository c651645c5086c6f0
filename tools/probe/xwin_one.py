"""One banded-unstructured block (4 M rows x 16 within +-2000), 40 products: the command the rocprofv3 passes of
profiles/r02_xwin_* wrap."""
import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
rng = np.random.default_rng(0)
m = 4_000_000
col = np.repeat(np.arange(m, dtype=np.int32), 16).reshape(m, 16)
col += rng.integers(-2000, 2000, size=(m, 16), dtype=np.int32)
np.clip(col, 0, m - 1, out=col); col.sort(axis=1); col += 1
H = pa.HostCSR(m, m, (1 + 16 * np.arange(m + 1)).astype(np.int32), col.ravel(), rng.standard_normal(m * 16))
blk = pa.DeviceCSR(H)
x = pa.DeviceVector(m, 0).upload(rng.standard_normal(m))
y = pa.DeviceVector(m, 0)
for _ in range(40): pa.spmv_(y, blk, x)
pa.context().sync()
print(blk.xwin(), blk.stream_bytes() + 16 * m, "bytes moved per product (streams + x + y)")
