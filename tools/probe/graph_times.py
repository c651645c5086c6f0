import os, sys
sys.path.insert(0, "/root/repo" if os.path.isdir("/root/repo") else ".")
os.environ.setdefault("PA_SPMV_VALUE_DICT", "0")
import numpy as np
import bench
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
ctx = pa.context()
def hx(g): return ((np.asarray(g, np.int64) * 2654435761) % 1000003) / 1000003.0 - 0.5
for name, build in (("config 3 on 2 parts", lambda: pa.build_p_matrix(pa.DebugArray([1, 2]), 128, 128, 128, 256, 128, 128, 2, 1, 1)[0]),
                    ("config 5 on 8 parts", lambda: (lambda t: pa.psparse_disassembled(t[0], t[1], t[2], t[3], t[4]))(pa.laplacian_fem((4096, 4096), (4, 2), pa.DebugArray(range(1, 9)))))):
    A = build()
    x = pa.pvector_from_function(lambda ind: hx(ind.get_local_to_global()) * (ind.get_local_to_owner() == ind.part), A.col_partition)
    y = pa.pzeros(A.row_partition)
    P = len(pa.local_items(A.matrix_partition))
    for one in ("1", "0"):
        os.environ["PA_GRAPH_ONE_STREAM"] = one
        ms, msg, mss = bench.whole_mul_times(pa, ctx, L, A, x, y)
        print(f"{name}: PA_GRAPH_ONE_STREAM={one}: eager {ms / P:.4f} ms per part, hipGraph replay {msg / P:.4f}, own x own alone {mss / P:.4f}", flush=True)
