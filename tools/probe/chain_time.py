"""mul! of all parts of a process against the parts' own x own alone: BASELINE config 3 on 2 parts and config 5 on 8 parts, all
on this one GPU (the two `extra_configs` entries of bench.py, without the rest of the bench).
    python tools/probe/chain_time.py [nodes_per_dir_of_config_5]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
import bench
ctx = pa.context()
n5 = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
out = {}
A, _ = pa.build_p_matrix(pa.DebugArray([1, 2]), 128, 128, 128, 256, 128, 128, 2, 1, 1)
x = pa.pvector_from_function(lambda ind: bench.hash_x(ind.get_local_to_global()) * (ind.get_local_to_owner() == ind.part), A.col_partition)
y = pa.pzeros(A.row_partition)
ms, msg, mss = bench.whole_mul_times(pa, ctx, L, A, x, y)
print("config 3 done", ms, msg, mss, flush=True)
out["config 3 on 2 parts"] = dict(ms_per_part_mul=round(ms / 2, 4), ms_per_part_mul_hipgraph=round(msg / 2, 4), ms_per_part_spmv=round(mss / 2, 4), mul_over_spmv=round(ms / mss, 3))
os.environ["PA_MUL_FUSED"] = "0"; ctx.reload_env()
A2, _ = pa.build_p_matrix(pa.DebugArray([1, 2]), 128, 128, 128, 256, 128, 128, 2, 1, 1)
x2 = pa.pvector_from_function(lambda ind: bench.hash_x(ind.get_local_to_global()) * (ind.get_local_to_owner() == ind.part), A2.col_partition)
ms, msg, mss = bench.whole_mul_times(pa, ctx, L, A2, x2, y)
out["config 3 on 2 parts, PA_MUL_FUSED=0"] = dict(ms_per_part_mul=round(ms / 2, 4), ms_per_part_mul_hipgraph=round(msg / 2, 4), ms_per_part_spmv=round(mss / 2, 4), mul_over_spmv=round(ms / mss, 3))
os.environ.pop("PA_MUL_FUSED"); ctx.reload_env()
del A, x, y, A2, x2
I, J, V, rows, cols = pa.laplacian_fem((n5, n5), (4, 2), pa.DebugArray(range(1, 9)))
A = pa.psparse_disassembled(I, J, V, rows, cols)
del I, J, V
x = pa.pvector_from_function(lambda ind: bench.hash_x(ind.get_local_to_global()) * (ind.get_local_to_owner() == ind.part), A.col_partition)
y = pa.pzeros(A.row_partition)
for sw in ({}, {"PA_MUL_FUSED": "0"}, {"PA_PUSH": "0"}):
    os.environ.update(sw)
    ctx.reload_env()
    print("config 5", sw, flush=True)
    ms, msg, mss = bench.whole_mul_times(pa, ctx, L, A, x, y, graph=not sw)
    out[f"config 5 on 8 parts {sw}"] = dict(ms_per_part_mul=round(ms / 8, 4), ms_per_part_mul_hipgraph=round(msg / 8, 4), ms_per_part_spmv=round(mss / 8, 4), mul_over_spmv=round(ms / mss, 3))
    for k in sw:
        os.environ.pop(k)
    ctx.reload_env()
print(json.dumps(out, indent=1))
