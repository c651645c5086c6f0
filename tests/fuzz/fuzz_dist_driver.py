"""One part per PROCESS (torch.distributed; by default the ranks share this box's GPU and the exchange is staged through the
host, with PA_TRANSPORT=rccl every rank takes its own GPU and the exchange is csrc/pa_rccl.cpp's): random
PSparseMatrices as in fuzz_mul.py, every rank building its own part and checking it against the sequential oracle run
redundantly on every rank -- mul!, mul!(...,alpha,beta), the one-call product, consistent!, assemble!, dot, a few iterations of
the CG loops.  Run:  python -m torch.distributed.run --nproc-per-node P --master-addr 127.0.0.1 --master-port N
tests/fuzz/fuzz_dist_driver.py [cases] [seed0]"""
import os, sys, time, functools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # tools/probe/ -> repo
sys.path.insert(0, ROOT)
os.environ.setdefault("PA_TRANSPORT", "host")
import numpy as np
import torch.distributed as dist
from __graft_entry__ import load_package, load_oracle
pa = load_package()
orc = load_oracle()
n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0


def body(distribute):
    P, me = dist.get_world_size(), dist.get_rank() + 1
    k = me - 1
    ranks = distribute(range(1, P + 1))
    bad = 0
    t0 = time.time()
    for case in range(n_cases):
        rng = np.random.default_rng(seed0 + case)
        n = int(rng.integers(max(64, 8 * P), 30_000))
        band = int(rng.choice([3, 40, 700, 10**9]))
        rows = pa.uniform_partition(ranks, n)
        orows = orc.uniform_partition(P, n)
        sym = bool(rng.integers(0, 2))
        Is, Js, Vs = [], [], []
        if sym:                                        # symmetric, diagonally dominant: the CG loops run on it
            kk = rng.integers(1, 8, n)
            i0 = np.repeat(np.arange(1, n + 1), kk)
            j0 = i0 + rng.integers(1, max(2, min(band, n)), len(i0))
            keep = j0 <= n
            i0, j0 = i0[keep], j0[keep]
            v0 = -rng.random(len(i0)) - 0.1
            diag = np.zeros(n + 1)
            np.add.at(diag, i0, -v0); np.add.at(diag, j0, -v0)
            I = np.concatenate([i0, j0, np.arange(1, n + 1)]); J = np.concatenate([j0, i0, np.arange(1, n + 1)])
            V = np.concatenate([v0, v0, 2.0 * diag[1:] + 1.0])
            order = np.lexsort((J, I)); I, J, V = I[order], J[order], V[order]
            for ind in orows:
                lo, hi = ind.own_to_global[0], ind.own_to_global[-1]
                sel = (I >= lo) & (I <= hi)
                Is.append(I[sel].astype(np.int64)); Js.append(J[sel].astype(np.int64)); Vs.append(V[sel].copy())
        else:
            for ind in orows:
                g = ind.own_to_global
                lens = rng.integers(0, int(rng.integers(2, 30)), len(g))
                I = np.repeat(g, lens)
                J = rng.integers(1, n + 1, len(I)) if band >= 10**9 else np.clip(I + rng.integers(-band, band + 1, len(I)), 1, n)
                Is.append(I.astype(np.int64)); Js.append(J.astype(np.int64)); Vs.append(rng.standard_normal(len(I)))
        A = pa.psparse_from_coo(distribute([a.copy() for a in Is]), distribute([a.copy() for a in Js]), distribute([a.copy() for a in Vs]), rows)
        Ao = orc.psparse_from_coo([a.copy() for a in Is], [a.copy() for a in Js], [a.copy() for a in Vs], orows)
        fails = []
        xo = [rng.standard_normal(c.n_local) * (c.local_to_owner == c.part) for c in Ao.cols]
        x = pa.pvector_from_function(lambda ind: xo[ind.part - 1].copy(), A.col_partition)
        y = pa.pzeros(A.row_partition)
        pa.mul_(y, A, x)
        yo = [np.zeros(r.n_local) for r in Ao.rows]
        orc.mul(yo, Ao, [v.copy() for v in xo])
        if not np.array_equal(pa.getany(y.own_values()), yo[k][:Ao.rows[k].n_own]): fails.append("mul!")
        xc = [v.copy() for v in xo]; orc.consistent(xc, Ao.cols)
        if not np.array_equal(pa.getany(x.local_values()), xc[k]): fails.append("ghosts after mul!")
        y2 = pa.pzeros(A.row_partition)
        pa.mul_c_(y2, A, x)
        if not np.array_equal(pa.getany(y2.own_values()), yo[k][:Ao.rows[k].n_own]): fails.append("one-call mul!")
        alpha, beta = float(rng.standard_normal()), float(rng.standard_normal())
        c0 = [rng.standard_normal(r.n_local) for r in Ao.rows]
        c = pa.pvector_from_function(lambda ind: c0[ind.part - 1].copy(), A.row_partition)
        pa.mul5_(c, A, x, alpha, beta)
        co = [v.copy() for v in c0]
        orc.mul5(co, Ao, [v.copy() for v in xo], alpha, beta)
        if not np.array_equal(pa.getany(c.own_values()), co[k][:Ao.rows[k].n_own]): fails.append("mul!(alpha,beta)")
        v0_ = [rng.standard_normal(cc.n_local) for cc in Ao.cols]
        v = pa.pvector_from_function(lambda ind: v0_[ind.part - 1].copy(), A.col_partition)
        pa.assemble_(v).wait()
        vo = [w.copy() for w in v0_]; orc.assemble(vo, Ao.cols)
        if not np.array_equal(pa.getany(v.local_values()), vo[k]): fails.append("assemble!")
        d, dref = pa.dot(x, x), orc.dot(xc, xc, Ao.cols)
        if abs(d - dref) > 1e-12 * abs(dref) + 1e-300: fails.append("dot")
        if sym:
            b = pa.pzeros(A.col_partition)
            pa.mul_(b, A, x)
            outs = []
            for fn in (pa.ref_cg_, functools.partial(pa.opt_cg_, fuse=False), pa.opt_cg_):
                h = []
                xs, r0, r, it = fn(pa.pzeros(A.col_partition), A, b, maxiter=8, history=h)
                outs.append((r0, r, it, h, pa.getany(xs.own_values()).copy()))
            if not (outs[0][:4] == outs[1][:4] and np.array_equal(outs[0][4], outs[1][4])): fails.append("unfused CG != ref_cg")
            if not np.allclose(outs[2][3], outs[0][3], rtol=1e-9, atol=1e-300): fails.append("fused CG history")
        flag = np.array([len(fails)], dtype=np.int64)
        import torch
        tflag = torch.from_numpy(flag)
        dist.all_reduce(tflag)
        if fails:
            print(f"MISMATCH rank {me} case {seed0 + case}: P {P} n {n} band {band} sym {sym}: {fails}", flush=True)
        bad += int(tflag.item() > 0)
        if me == 1 and case % 10 == 9:
            print(f"{case + 1} cases, {bad} with mismatches, {time.time() - t0:.0f} s", flush=True)
    if me == 1:
        print(f"done: P {P}, {n_cases} cases, {bad} with mismatches", flush=True)
    return bad == 0


if __name__ == "__main__":
    dist.init_process_group("gloo")
    if os.environ["PA_TRANSPORT"] == "rccl":               # one GPU per rank (LOCAL_RANK): the RCCL neighbour exchange itself
        from pa_amd.p_vector import init_comm
        init_comm()
    ok = pa.with_torchdist(body)
    dist.barrier()
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)
