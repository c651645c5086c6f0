"""config 3 whole (27-pt 128^3 x 2 parts on one GPU) and config 5 whole (Q1 FEM 4096^2 on 8 parts): 50 mul! each, then the parts'
own x own alone -- for `rocprofv3 --kernel-trace --stats` (per-kernel durations of the fused launches against k_spmv_rowsplit).
    PA_SPMV_VALUE_DICT=0 rocprofv3 --kernel-trace --stats --output-format csv -d out -o ft -- python tools/probe/fused_trace.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package
pa = load_package()
import pa_amd._lib as L
import bench
ctx = pa.context()
def run(A):
    x = pa.pvector_from_function(lambda ind: bench.hash_x(ind.get_local_to_global()) * (ind.get_local_to_owner() == ind.part), A.col_partition)
    y = pa.pzeros(A.row_partition)
    for _ in range(300): pa.mul_c_(y, A, x)
    ctx.sync()
    blocks, xs, ys = pa.local_items(A.matrix_partition), pa.local_items(x.vector_partition), pa.local_items(y.vector_partition)
    for _ in range(300):
        for blk, xv, yv in zip(blocks, xs, ys):
            pa.spmv_(yv, blk.own_own, xv, L.SEG_OWN, L.SEG_OWN, 1.0, 0.0)
    ctx.sync()
A, _ = pa.build_p_matrix(pa.DebugArray([1, 2]), 128, 128, 128, 256, 128, 128, 2, 1, 1)
run(A)
if len(sys.argv) > 1:
    I, J, V, rows, cols = pa.laplacian_fem((4096, 4096), (4, 2), pa.DebugArray(range(1, 9)))
    run(pa.psparse_disassembled(I, J, V, rows, cols))
print("done")
