"""Driver run under `python -m torch.distributed.run --nproc-per-node P` (gloo, CPU): the host-side set-up of the
N>1 path (one part per process) against the oracle and the reference's literal goldens.  Mirrors the reference's
test/mpi_array/drivers/*.jl: "passed" == every rank exits 0."""
import json
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from __graft_entry__ import load_package, load_oracle  # noqa: E402

pa = load_package()
orc = load_oracle()
golden = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_literals.json")))


def main():
    dist.init_process_group("gloo")
    P, me = dist.get_world_size(), dist.get_rank() + 1

    def body(distribute):
        ranks = distribute(range(1, P + 1))
        assert pa.getany(ranks) == me
        # collectives (test/primitives_tests.jl:40-150,290-300)
        g = pa.gather(ranks)
        assert pa.getany(g) == (list(range(1, P + 1)) if me == pa.MAIN else [])
        assert pa.getany(pa.gather(ranks, destination="all")) == list(range(1, P + 1))
        assert pa.getany(pa.scatter(pa.pmap(lambda r: [10 * q for q in range(1, P + 1)], ranks))) == 10 * me
        assert pa.preduce(lambda a, b: a + b, ranks, init=0) == P * (P + 1) // 2
        assert pa.getany(pa.scan(lambda a, b: a + b, ranks, type="inclusive", init=0)) == me * (me + 1) // 2
        assert pa.getany(pa.scan(lambda a, b: a + b, ranks, type="exclusive", init=1)) == 1 + me * (me - 1) // 2
        if P == 4:
            c6 = golden["collectives"]                       # the literal expectations, one part per process
            b10 = pa.pmap(lambda r: 10 * r, ranks)
            assert pa.getany(pa.gather(b10, destination="all")) == c6["gather_10rank"]["rcv"]
            snd = pa.pmap(lambda r: list(range(1, r + 1)), ranks)
            assert pa.getany(pa.gather(snd, destination="all")) == c6["gather_ragged"]["rcv_all"]
            assert pa.getany(pa.scatter(pa.gather(snd))) == list(range(1, me + 1))
            assert pa.getany(pa.multicast(ranks, source=2)) == c6["multicast_rank_source2"]
            assert pa.getany(pa.multicast(snd, source=2)) == c6["multicast_ragged_source2"]
            a3 = pa.pmap(lambda r: 3 * (r % 3), ranks)
            plus = lambda x, y: x + y
            assert pa.getany(pa.scan(plus, a3, type="inclusive", init=0)) == c6["scan"]["inclusive_init0"][me - 1]
            assert pa.getany(pa.scan(plus, a3, type="exclusive", init=1)) == c6["scan"]["exclusive_init1"][me - 1]
            assert pa.getany(pa.reduction(plus, ranks, init=10, destination="all")) == c6["reduction"]["sum_init10_all"]
            for c in golden["exchange"]:
                snd_ids = distribute(c["snd_ids"])
                graph = pa.exchange_graph(snd_ids, None if c["rcv_ids"] is None else distribute(c["rcv_ids"]))
                if c["rcv_ids"] is not None:
                    assert [int(v) for v in pa.getany(pa.find_rcv_ids_gather_scatter(snd_ids))] == c["rcv_ids"][me - 1]
                assert pa.getany(pa.exchange(distribute(c["snd_literal"]), graph)) == c["rcv"][me - 1]
            c = golden["exchange_jagged"]
            rcv = pa.exchange(distribute(c["snd"]), pa.ExchangeGraph(distribute(c["snd_ids"]), distribute(c["rcv_ids"])))
            assert pa.getany(rcv) == c["rcv"][me - 1]
            c = golden["p_vector_local_indices"]
            parts = pa.pmap(lambda r: pa.LocalIndices(c["n"], r, local_to_global=c["local_to_global"][r - 1],
                                                      local_to_owner=c["local_to_owner"][r - 1]), ranks)
            oparts = [orc.local_indices(c["n"], p + 1, g_, o_) for p, (g_, o_) in
                      enumerate(zip(c["local_to_global"], c["local_to_owner"]))]
            ns, nr = pa.assembly_neighbors(parts)
            ls, lr = pa.assembly_local_indices(parts, ns, nr)
            ons, onr = orc.assembly_neighbors(oparts)
            ols, olr = orc.assembly_local_indices(oparts, ons, onr)
            assert np.array_equal(pa.getany(ns), ons[me - 1]) and np.array_equal(pa.getany(nr), onr[me - 1])
            assert pa.getany(ls).tolists() == ols[me - 1].tolists() and pa.getany(lr).tolists() == olr[me - 1].tolists()
            vp = pa.variable_partition(distribute([4, 2, 6, 3]), 15)
            assert pa.getany(vp).get_local_to_global().tolist() == golden["variable_partition"][1]["local_to_global"][me - 1]
        # HPCG set-up, one part per process, vs the sequential oracle
        npx, npy, npz = pa.compute_optimal_shape_XYZ(P)
        nx, ny, nz = 4, 3, 5
        gn = (npx * nx, npy * ny, npz * nz)
        Ao, bo, _ = orc.hpcg_build_p_matrix(nx, ny, nz, npx, npy, npz)
        rows = pa.uniform_partition(ranks, (npx, npy, npz), gn)

        def gen(r):
            return pa.build_matrix(nx, ny, nz, *gn, r.ranges[0][0], r.ranges[1][0], r.ranges[2][0])

        I, J, V, b, Ib = pa.tuple_of_arrays(pa.pmap(gen, rows))
        cols = pa.pmap(pa.union_ghost, rows, J, pa.find_owner(rows, J))
        ns, nr = pa.assembly_neighbors(cols)
        ls, lr = pa.assembly_local_indices(cols, ns, nr)
        ons, onr = orc.assembly_neighbors(Ao.cols)
        ols, olr = orc.assembly_local_indices(Ao.cols, ons, onr)
        k = me - 1
        c = pa.getany(cols)
        assert np.array_equal(c.get_local_to_global(), Ao.cols[k].local_to_global)
        assert np.array_equal(c.get_local_to_owner(), Ao.cols[k].local_to_owner)
        assert np.array_equal(pa.getany(ns), ons[k]) and np.array_equal(pa.getany(nr), onr[k])
        assert np.array_equal(pa.getany(ls).data, ols[k].data) and np.array_equal(pa.getany(ls).ptrs, ols[k].ptrs)
        assert np.array_equal(pa.getany(lr).data, olr[k].data) and np.array_equal(pa.getany(lr).ptrs, olr[k].ptrs)
        assert pa.is_consistent(pa.ExchangeGraph(ns, nr))
        # the fused generator gives the same part
        c2, oo, oh, b2 = pa.build_split_blocks_fused(pa.getany(rows), nx, ny, nz, *gn)
        assert np.array_equal(c2.get_local_to_global(), Ao.cols[k].local_to_global)
        assert np.array_equal(oo.colval, Ao.blocks[k].own_own.colval) and np.array_equal(oh.colval, Ao.blocks[k].own_ghost.colval)
        assert np.array_equal(oh.rowptr, Ao.blocks[k].own_ghost.rowptr) and np.array_equal(b2, bo[k][:c2.n_own])
        # disassembled COO -> assembled (config 5 route): the triplet exchange crosses processes
        if P in (2, 4):
            fparts = {2: (2, 1), 4: (2, 2)}[P]
            nodes = (7, 5)
            I, J, V, frows, fcols = pa.laplacian_fem(nodes, fparts, ranks)
            Io, Jo, Vo, orows, ocols = orc.laplacian_fem(nodes, fparts)
            Af, _ = orc.psparse_disassembled(Io, Jo, Vo, orows, ocols)
            rows_sa = pa.pmap(pa.union_ghost, frows, I, pa.find_owner(frows, I))
            cols_sa = pa.pmap(pa.union_ghost, fcols, J, pa.find_owner(fcols, J))
            import pa_amd.p_sparse_matrix as psm
            b4 = pa.pmap(lambda Ii, Ji, Vi, r, c: psm._split4(
                pa.sparse_matrix(r.global_to_local(Ii), c.global_to_local(Ji), Vi, r.n_local, c.n_local), r, c),
                I, J, V, rows_sa, cols_sa)
            host, cols_fa = pa.psparse_assemble_host(b4, rows_sa, cols_sa, frows)
            assert np.array_equal(pa.getany(cols_fa).get_local_to_global(), Af.cols[k].local_to_global)
            for mine, ref in zip(pa.getany(host), (Af.blocks[k].own_own, Af.blocks[k].own_ghost)):
                assert np.array_equal(mine.rowptr, ref.rowptr) and np.array_equal(mine.colval, ref.colval)
                assert np.array_equal(mine.nzval, ref.nzval)
            # test/fem_example.jl's set-up: the cell -> dof table of the ghost cells crosses processes (jagged consistent!)
            S = pa.fem_example.fem_example_system(ranks, fparts, (9, 6))
            O = orc.fem_example_setup(fparts, (9, 6))
            assert S["n_global_dofs"] == O["n_global_dofs"]
            for key in ("I", "J", "V", "II", "VV"):
                assert np.array_equal(pa.getany(S[key]), O[key][k]), key
            assert np.array_equal(pa.getany(S["dof_partition"]).own_to_global, O["dof_partition"][k].own_to_global)
        # random block partitions with arbitrary ghosts, one part per process: find_owner / union_ghost are local, the assembly
        # neighbours and local indices come out of point-to-point exchanges over random graphs (every rank draws the same
        # numbers and checks its own part against the sequential oracle)
        for seed in range(40):
            rng = np.random.default_rng(77000 + seed)
            n_own = [int(rng.integers(0, 25)) if rng.random() < 0.85 else 0 for _ in range(P)]
            if sum(n_own) == 0:
                n_own[0] = 4
            n = sum(n_own)
            parts = pa.variable_partition(distribute(list(n_own)), n)
            oparts = orc.variable_partition(list(n_own), n)
            req = [rng.integers(1, n + 1, int(rng.integers(0, 20))).astype(np.int64) for _ in range(P)]
            owners = pa.find_owner(parts, distribute([r.copy() for r in req]))
            oowners = orc.find_owner(oparts, [r.copy() for r in req])
            assert np.array_equal(pa.getany(owners), oowners[k]), seed
            parts = pa.pmap(pa.union_ghost, parts, distribute([r.copy() for r in req]), owners)
            oparts = [orc.union_ghost(o, r, w) for o, r, w in zip(oparts, req, oowners)]
            assert np.array_equal(pa.getany(parts).get_local_to_global(), oparts[k].local_to_global), seed
            snd, rcv = pa.assembly_neighbors(parts)
            osnd, orcv = orc.assembly_neighbors(oparts)
            assert np.array_equal(pa.getany(snd), osnd[k]) and np.array_equal(pa.getany(rcv), orcv[k]), seed
            ls, lr = pa.assembly_local_indices(parts)
            ols, olr = orc.assembly_local_indices(oparts)
            for mine, ref in ((pa.getany(ls), ols[k]), (pa.getany(lr), olr[k])):
                assert np.array_equal(mine.data, ref.data) and np.array_equal(mine.ptrs, ref.ptrs), seed
        return True

    ok = pa.with_torchdist(body)
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        sys.exit(1)


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        import traceback
        traceback.print_exc()
        os._exit(1)          # MPI.Abort analogue (src/mpi_array.jl:72-79): do not leave peers hanging
