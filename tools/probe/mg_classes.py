"""Memory class of every array of the 256^3 multigrid hierarchy (matrix blocks, colour blocks, vectors), per level:
what the MG-PCG iteration time depends on.  Usage: mg_classes.py <package root> <tag> [n]"""
import os, sys, time
root, tag = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 256
sys.path.insert(0, root)
from __graft_entry__ import load_package
pa = load_package()
S = pa.pc_setup(pa.DebugArray([1]), 1, 4, n, n, n, "multicolor_spmv")
A, b = S.A_vec[-1], S.r[-1]
x = pa.pzeros(A.col_partition)
pa.opt_cg_(x, A, b, maxiter=25, Pl=S, fuse=True)
out = []
for rep in range(3):
    pa.context().sync()
    t = time.perf_counter()
    pa.opt_cg_(pa.pzeros(A.col_partition), A, b, maxiter=30, Pl=S, fuse=True)
    pa.context().sync()
    out.append((time.perf_counter() - t) / 30 * 1e3)
print(f"[{tag}] {n}^3 MG-PCG iteration {min(out):.3f} ms; arena {pa.context().arena().get('class_gib')}")
vc = lambda v: v.vector_partition.items[0].memory_class()
for lev in range(S.l - 1, -1, -1):
    Al = S.A_vec[lev]
    blk = Al.matrix_partition.items[0]
    mats = [blk.own_own.memory_class(), blk.own_ghost.memory_class()]
    p = S.gs_states[lev].parts.items[0]
    print(f"[{tag}] level {lev}: A {mats}  colour blocks {[bb.memory_class() for bb in p[0]]}  d {p[1].memory_class()}  b/r {vc(S.r[lev])}  x {vc(S.x[lev])}  Axf {vc(S.Axf[lev])}"
          + (f"  row block {[q.memory_class() for q in S.row_blocks[lev - 1].items]}" if lev >= 1 and S.row_blocks and S.row_blocks[lev - 1] is not None else ""))
print(f"[{tag}] CG x {vc(x)}")
