"""One part per process (torch.distributed); by default every rank on the SAME GPU, host-staged transport (PA_TRANSPORT=host);
with PA_TRANSPORT=rccl one GPU per rank and the RCCL neighbour exchange of csrc/pa_rccl.cpp (needs as many GPUs as ranks);
with PA_TRANSPORT=ipc the push transport of csrc/pa_push.hip (hipIpc-mapped receive buffers; ranks may share a GPU):
the full N>1 device path -- pack kernel, exchange, unpack kernel, own*own / own*ghost SpMV, dot -- against the
sequential oracle, bit-exact.  Run by tests/test_gpu_multiprocess.py on the 1-GPU box."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ.setdefault("PA_TRANSPORT", "host")       # "rccl": one GPU per rank (LOCAL_RANK), ncclSend/ncclRecv between them
from __graft_entry__ import load_package, load_oracle  # noqa: E402

pa = load_package()
orc = load_oracle()


def body(distribute):
    P, me = dist.get_world_size(), dist.get_rank() + 1
    ranks = distribute(range(1, P + 1))
    npx, npy, npz = pa.compute_optimal_shape_XYZ(P)
    nx, ny, nz = 8, 6, 5
    for fused in (False, True):
        A, b = pa.build_p_matrix(ranks, nx, ny, nz, npx * nx, npy * ny, npz * nz, npx, npy, npz, fused=fused)
        Ao, bo, _ = orc.hpcg_build_p_matrix(nx, ny, nz, npx, npy, npz)
        k = me - 1
        y = pa.pzeros(A.row_partition)
        pa.mul_(y, A, pa.pones(A.col_partition))
        assert np.array_equal(pa.getany(y.own_values()), pa.getany(b.own_values()))          # A*1 == b
        xo = [orc.hash_x(c.local_to_global) * (c.local_to_owner == c.part) for c in Ao.cols]
        x = pa.pvector_from_function(lambda ind: xo[ind.part - 1].copy(), A.col_partition)
        pa.mul_(y, A, x)
        yo = [np.zeros(r.n_local) for r in Ao.rows]
        orc.mul(yo, Ao, [v.copy() for v in xo])
        assert np.array_equal(pa.getany(y.own_values()), yo[k][:Ao.rows[k].n_own]), "mul! differs from the oracle"
        orc.consistent(xo, Ao.cols)
        assert np.array_equal(pa.getany(x.local_values()), xo[k]), "ghosts differ after consistent!"
        # assemble!: ghost contributions travel back and are added in the reference's order
        v = pa.pvector_from_function(lambda ind: orc.hash_x(ind.get_local_to_global() + 5 * ind.part), A.col_partition)
        vo = [orc.hash_x(c.local_to_global + 5 * c.part) for c in Ao.cols]
        pa.assemble_(v).wait()
        orc.assemble(vo, Ao.cols)
        assert np.array_equal(pa.getany(v.local_values()), vo[k]), "assemble! differs from the oracle"
        d = pa.dot(x, x)
        dref = orc.dot(xo, xo, Ao.cols)
        assert abs(d - dref) <= 1e-13 * abs(dref)
        assert np.array_equal(y.collect(), orc.pvector_collect(yo, Ao.rows))
        # the operator-level call (pa_mul5: own x ghost may read the receive buffer, the unpack follows it) and many exchanges
        # in a row (the push transport's flow control: every buffer is reused 12 times)
        x2 = pa.pvector_from_function(lambda ind: xo[ind.part - 1] * (ind.get_local_to_owner() == ind.part), A.col_partition)
        y2 = pa.pzeros(A.row_partition)
        fused_before = pa.context().fused_launches()[1]
        for _ in range(12):
            pa.mul_c_(y2, A, x2)
        assert np.array_equal(pa.getany(y2.own_values()), yo[k][:Ao.rows[k].n_own]), "mul_c_ differs from the oracle"
        assert np.array_equal(pa.getany(x2.local_values()), xo[k]), "ghosts differ after mul_c_"
        pa.mul_c_(y2, A, x2, -0.5, 2.0)
        y5 = [v.copy() for v in yo]
        orc.mul5(y5, Ao, [v.copy() for v in xo], -0.5, 2.0)
        assert np.array_equal(pa.getany(y2.own_values()), y5[k][:Ao.rows[k].n_own]), "mul_c_(alpha,beta) differs from the oracle"
        if os.environ["PA_TRANSPORT"] == "ipc" and os.environ.get("PA_MUL_FUSED", "1") != "0" and P > 1:
            # round 5: over the ipc link mul! is ONE launch per part -- push, both products, unpack and acknowledgement inside it
            inside = pa.context().fused_launches()[1] - fused_before
            assert inside == 13, f"{inside} of 13 products ran as one launch with the exchange inside"
        # mul!(c,transpose(a),b,alpha,beta): assemble!(c) under A_oo'*b
        bt = [orc.hash_x(r.local_to_global + 1) for r in Ao.rows]
        ct = [orc.hash_x(c.local_to_global + 9) for c in Ao.cols]
        bdev = pa.pvector_from_function(lambda ind: bt[ind.part - 1].copy(), A.row_partition)
        cdev = pa.pvector_from_function(lambda ind: ct[ind.part - 1].copy(), A.col_partition)
        pa.mul5_transpose_(cdev, A, bdev, 0.75, -1.25)
        orc.mul5_transpose(ct, Ao, bt, 0.75, -1.25)
        assert np.array_equal(pa.getany(cdev.local_values()), ct[k]), "transpose product differs from the oracle"
    # K7 across processes: psparse!(C,V2,cache) with the triplet values exchanged by the device plan
    if P in (2, 4):
        fparts = {2: (2, 1), 4: (2, 2)}[P]
        nodes = (13, 9)
        I, J, V, frows, fcols = pa.laplacian_fem(nodes, fparts, ranks)
        C_, cache = pa.psparse_disassembled(I, J, V, frows, fcols, reuse=True)
        Io, Jo, Vo, orows, ocols = orc.laplacian_fem(nodes, fparts)
        V2o = [v * 3.0 + orc.hash_x(np.arange(len(v))) * 1e-2 for v in Vo]
        pa.psparse_(C_, pa.pmap(lambda v: v * 3.0 + orc.hash_x(np.arange(len(v))) * 1e-2, V), cache).wait()
        Af, _ = orc.psparse_disassembled(Io, Jo, V2o, orows, ocols)
        exp = np.concatenate([Af.blocks[k].own_own.nzval, Af.blocks[k].own_ghost.nzval])
        assert np.array_equal(pa.getany(cache.W).download()[:len(exp)], exp), "psparse! differs from a fresh assembly"
    return True


if __name__ == "__main__":
    dist.init_process_group("gloo")
    if os.environ["PA_TRANSPORT"] == "rccl":
        from pa_amd.p_vector import init_comm
        comm = init_comm()
        assert comm.info() == {"rank": dist.get_rank(), "nranks": dist.get_world_size()}
    pa.with_torchdist(body)
    dist.barrier()
    dist.destroy_process_group()
