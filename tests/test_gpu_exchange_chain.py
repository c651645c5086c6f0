"""The exchange chain of mul! after round 4: pack and exchange! fused into one push launch for all parts of a
process (csrc/pa_push.hip), own x ghost reading consistent!'s receive buffer, the unpack behind it -- every variant against the
oracle and against the round-3 order (PA_PUSH=0 / PA_MUL_GHOST_FROM_BUFFER=0), bit for bit (np.array_equal).
Reference: assemble_impl! src/p_vector.jl:587-612, mul! src/p_sparse_matrix.jl:2090-2142."""
import ctypes as C
import os

import numpy as np
import pytest

from gpu_helpers import pa, ranks, upload, oracle_mul, env, reload_switches
import pa_amd._lib as L
import pa_amd.p_sparse_matrix as psm

pytestmark = pytest.mark.gpu


def _from_buffer(A, b):
    out = []
    for h in psm._operator_handles(A, b).items:
        yes = C.c_int()
        L.call("pa_matrix_ghost_from_buffer", h, C.byref(yes))
        out.append(bool(yes.value))
    return out


def _fused(A, b):
    out = []
    for h in psm._operator_handles(A, b).items:
        yes, nb = C.c_int(), C.c_int64()
        L.call("pa_matrix_fused", h, C.byref(yes), C.byref(nb))
        out.append((bool(yes.value), int(nb.value)))
    return out


@pytest.mark.parametrize("shape", ["27 parts", "fem (4,2)", "periodic ghost layers"])
def test_push_exchange_equals_pack_and_copies(orc, shape):
    """consistent! and assemble! through ONE push launch (PA_PUSH=1, the default) and through pack + device-to-device copies
    (PA_PUSH=0): the same local values as the oracle's, three exchanges in a row each (buffers are reused)."""
    if shape == "27 parts":
        A, _ = pa.build_p_matrix(ranks(27), 4, 4, 4, 12, 12, 12, 3, 3, 3)
        parts, oparts = A.col_partition, orc.hpcg_build_p_matrix(4, 4, 4, 3, 3, 3)[0].cols
    elif shape == "fem (4,2)":
        I, J, V, rows, cols = pa.laplacian_fem((40, 24), (4, 2), ranks(8))
        A = pa.psparse_disassembled(I, J, V, rows, cols)
        Io, Jo, Vo, orows, ocols = orc.laplacian_fem((40, 24), (4, 2))
        parts, oparts = A.col_partition, orc.psparse_disassembled(Io, Jo, Vo, orows, ocols)[0].cols
    else:
        parts = pa.uniform_partition(ranks(4), (2, 2), (6, 6), (True, True), (True, True))
        oparts = orc.uniform_partition((2, 2), (6, 6), (True, True), (True, True))
    host = [orc.hash_x(o.local_to_global + 3 * o.part) for o in oparts]
    want_c = [h * (o.local_to_owner == o.part) for h, o in zip(host, oparts)]
    orc.consistent(want_c, oparts)
    want_a = [h.copy() for h in host]
    orc.assemble(want_a, oparts)
    for push in ("1", "0"):
        with env(PA_PUSH=push):
            v = upload([h * (o.local_to_owner == o.part) for h, o in zip(host, oparts)], parts)
            for _ in range(3):
                pa.consistent_(v).wait()
            for got, exp in zip(v.local_values().items, want_c):
                assert np.array_equal(got, exp), (shape, push)
            w = upload([h.copy() for h in host], parts)
            pa.assemble_(w).wait()
            for got, exp in zip(w.local_values().items, want_a):
                assert np.array_equal(got, exp), (shape, push)


@pytest.mark.parametrize("case", ["hpcg 27 parts", "hpcg (2,2,2) 12^3", "fem (4,2)"])
def test_mul_with_own_x_ghost_from_the_receive_buffer(orc, case):
    """pa_mul_all with the renamed own x ghost block: y AND b's ghosts equal the oracle's; the alpha/beta form; the same bits as
    the round-3 order; the handle says which route it took."""
    if case.startswith("hpcg"):
        n, np3 = ((4, 4, 4), (3, 3, 3)) if "27" in case else ((12, 12, 12), (2, 2, 2))
        P = int(np.prod(np3))
        build = lambda: pa.build_p_matrix(ranks(P), *n, *(a * q for a, q in zip(n, np3)), *np3)[0]
        Ao = orc.hpcg_build_p_matrix(*n, *np3)[0]
    else:
        def build():
            I, J, V, rows, cols = pa.laplacian_fem((48, 36), (4, 2), ranks(8))
            return pa.psparse_disassembled(I, J, V, rows, cols)
        Io, Jo, Vo, orows, ocols = orc.laplacian_fem((48, 36), (4, 2))
        Ao = orc.psparse_disassembled(Io, Jo, Vo, orows, ocols)[0]
    xo = [orc.hash_x(c.local_to_global) * (c.local_to_owner == c.part) for c in Ao.cols]
    yo = oracle_mul(orc, Ao, xo)
    xc = [v.copy() for v in xo]
    orc.consistent(xc, Ao.cols)
    y5 = [v.copy() for v in yo]
    orc.mul5(y5, Ao, [v.copy() for v in xo], 0.3, -1.5)
    outs = {}
    for tag, switches in (("buffer", {}), ("separate launches", {"PA_MUL_FUSED": "0"}), ("round 3", {"PA_PUSH": "0"}),
                          ("unpack first", {"PA_MUL_GHOST_FROM_BUFFER": "0"})):
        with env(**switches):
            A = build()
            x = upload([v.copy() for v in xo], A.col_partition)
            y = pa.pzeros(A.row_partition)
            for _ in range(3):
                pa.mul_c_(y, A, x)
            used = _from_buffer(A, x)
            assert all(used) if tag in ("buffer", "separate launches") else not any(used), (tag, used)
            fused = _fused(A, x)
            # round 5: one launch per part, the boundary rows (rows with stored entries in own_ghost) as its tail
            import pa_amd.p_vector as pv
            want_fused = tag == "buffer" and not pv.contexts_per_part()      # (one context per part: the parts' launches are not one chain)
            assert all(f for f, _ in fused) if want_fused else not any(f for f, _ in fused), (tag, fused)
            if want_fused:
                for (_, nb), blk in zip(fused, Ao.blocks):
                    assert nb == int(np.count_nonzero(np.diff(blk.own_ghost.rowptr)))
            for got, e, r in zip(y.own_values().items, yo, Ao.rows):
                assert np.array_equal(got, e[:r.n_own]), tag
            for got, e in zip(x.local_values().items, xc):
                assert np.array_equal(got, e), tag                     # consistent!(b) still happened
            pa.mul_c_(y, A, x, 0.3, -1.5)
            for got, e, r in zip(y.own_values().items, y5, Ao.rows):
                assert np.array_equal(got, e[:r.n_own]), tag
            outs[tag] = [v.copy() for v in y.own_values().items]
    for tag in ("separate launches", "round 3", "unpack first"):
        assert all(np.array_equal(a, b) for a, b in zip(outs["buffer"], outs[tag]))


def test_a_ghost_nobody_sends_keeps_the_unpack_route():
    """A ghost column with stored entries that no message carries (a periodic direction with a single part makes ghosts whose
    owner is the part itself, src/p_range.jl:441-445: they are ghosts, they are never exchanged) has no slot in the receive
    buffer: such a handle keeps the unpack-first route, silently, and the product reads what b's ghost holds.  Two parts driven
    through the C ABI: part 1 has 3 ghost columns, only 2 of them arrive from part 2."""
    ctx = pa.context()
    P = C.c_void_p
    i32 = lambda *v: np.array(v, np.int32)
    # part 1: 2 own rows/cols, 3 ghosts; own x ghost has an entry in EVERY ghost column.  part 2: 2 own, no ghosts, no coupling.
    oo1 = pa.DeviceCSR(pa.HostCSR(2, 2, i32(1, 2, 3), i32(1, 2), np.array([2.0, 3.0])))
    oh1 = pa.DeviceCSR(pa.HostCSR(2, 3, i32(1, 3, 4), i32(1, 3, 2), np.array([0.5, 0.25, -1.0])))
    oo2 = pa.DeviceCSR(pa.HostCSR(2, 2, i32(1, 2, 3), i32(1, 2), np.array([1.0, 1.0])))
    oh2 = pa.DeviceCSR(pa.HostCSR(2, 0, i32(1, 1, 1), i32(), np.zeros(0)))
    plans = [P(), P()]
    # assembly orientation (src/p_vector.jl:418-426): snd = ghost lids grouped by owner, rcv = own lids others ghost
    L.call("pa_plan_create", ctx.h, 1, 5, 1, L.ptr(i32(2)), L.ptr(i32(1, 3)), L.ptr(i32(3, 4)), 0, L.ptr(i32()), L.ptr(i32(1)), L.ptr(i32()), 1,
           C.byref(plans[0]))
    L.call("pa_plan_create", ctx.h, 2, 2, 0, L.ptr(i32()), L.ptr(i32(1)), L.ptr(i32()), 1, L.ptr(i32(1)), L.ptr(i32(1, 3)), L.ptr(i32(2, 1)), 1,
           C.byref(plans[1]))
    ms = [P(), P()]
    L.call("pa_matrix_create", ctx.h, oo1.h, oh1.h, plans[0], C.byref(ms[0]))
    L.call("pa_matrix_create", ctx.h, oo2.h, oh2.h, plans[1], C.byref(ms[1]))
    b1 = pa.DeviceVector(2, 3).upload(np.array([1.0, 2.0, 0.0, 0.0, 7.0]))      # ghost 3 (local id 5) is nobody's message: stays 7
    b2 = pa.DeviceVector(2, 0).upload(np.array([10.0, 20.0]))
    c1, c2 = pa.DeviceVector(2, 0), pa.DeviceVector(2, 0)
    arr = lambda xs: (C.c_void_p * 2)(*[x.value if isinstance(x, C.c_void_p) else x.h.value for x in xs])
    for _ in range(2):
        L.call("pa_mul_all", arr(ms), 2, arr([c1, c2]), arr([b1, b2]), 1.0, 0.0)
    ctx.sync()
    yes = C.c_int()
    L.call("pa_matrix_ghost_from_buffer", ms[0], C.byref(yes))
    assert yes.value == 0
    assert b1.download().tolist() == [1.0, 2.0, 20.0, 10.0, 7.0]                 # part 2 sends its own ids (2, 1)
    assert c1.download().tolist() == [2.0 * 1.0 + 0.5 * 20.0 + 0.25 * 7.0, 3.0 * 2.0 - 1.0 * 10.0]
    assert c2.download().tolist() == [10.0, 20.0]
    for m in ms:
        L.call("pa_matrix_destroy", m)
    for p in plans:
        L.call("pa_plan_destroy", p)


@pytest.mark.parametrize("one_stream", ["1", "0"])
def test_mul_all_replayed_from_a_hipgraph(orc, monkeypatch, one_stream):
    """The whole mul! of 8 parts (push launch, 8 own x own, 8 own x ghost from the buffers, one unpack launch) captured once and
    replayed: the bits of the eager call, also after x changed between replays, and x's ghosts made consistent by the replay.
    Recorded as ONE chain on the compute stream (the default inside a capture: a graph with edges between two streams replays
    2.6 x slower than the eager calls) and, PA_GRAPH_ONE_STREAM=0, with the eager call's two streams."""
    with env(PA_GRAPH_ONE_STREAM=one_stream):
        n, np3 = (10, 8, 6), (2, 2, 2)
        A = pa.build_p_matrix(ranks(8), *n, *(a * q for a, q in zip(n, np3)), *np3)[0]
        Ao = orc.hpcg_build_p_matrix(*n, *np3)[0]
        xo = [orc.hash_x(c.local_to_global) * (c.local_to_owner == c.part) for c in Ao.cols]
        x = upload([v.copy() for v in xo], A.col_partition)
        y = pa.pzeros(A.row_partition)
        pa.mul_c_(y, A, x)                                   # eager once: tables and the renamed block are made outside the capture
        with pa.Graph() as g:
            pa.mul_c_(y, A, x)
        for rep in range(3):
            xr = [v * (1.0 + rep) for v in xo]
            for dv, h in zip(x.vector_partition.items, xr):
                dv.upload(h)
            pa.pfill(0.0, A.row_partition)
            g.launch()
            yo = oracle_mul(orc, Ao, xr)
            for got, e, r in zip(y.own_values().items, yo, Ao.rows):
                assert np.array_equal(got, e[:r.n_own]), rep
            xc = [v.copy() for v in xr]
            orc.consistent(xc, Ao.cols)
            for got, e in zip(x.local_values().items, xc):
                assert np.array_equal(got, e), rep


def test_a_recorded_hipgraph_of_the_fused_product_follows_value_updates_in_place(orc):
    """the fused launch's boundary-row block `bd` holds COPIES of own_own's and own_ghost's values.  A graph
    recorded through it and replayed after pa_csr_update_values must multiply with the NEW values in boundary rows too (it summed
    them from the old copies), and the eager product after the update must not free what the recorded graph still reads (bd was
    destroyed and rebuilt).  Now bd and the twin of own_ghost follow every update in place, at the update: replay, eager product and
    replay again all equal the oracle on the updated values, bit for bit; the handle stays fused with the same block."""
    n, np3 = (10, 8, 6), (2, 2, 2)
    A = pa.build_p_matrix(ranks(8), *n, *(a * q for a, q in zip(n, np3)), *np3)[0]
    Ao = orc.hpcg_build_p_matrix(*n, *np3)[0]
    xo = [orc.hash_x(c.local_to_global) * (c.local_to_owner == c.part) for c in Ao.cols]
    x = upload([v.copy() for v in xo], A.col_partition)
    y = pa.pzeros(A.row_partition)
    pa.mul_c_(y, A, x)
    assert all(f for f, _ in _fused(A, x))
    with pa.Graph() as g:
        pa.mul_c_(y, A, x)
    rng = np.random.default_rng(5)
    for rep in range(3):
        for blk, blko in zip(A.matrix_partition.items, Ao.blocks):
            for dev, host in ((blk.own_own, blko.own_own), (blk.own_ghost, blko.own_ghost)):
                if rep == 1 and dev is blk.own_own:
                    continue                                         # (an update of own_ghost alone, too)
                host.nzval[:] = rng.integers(-8, 9, size=len(host.nzval)) * 0.25 + (rep + 1)
                dev.update_values(host.nzval)
        yo = oracle_mul(orc, Ao, xo)
        for how in ("replay", "eager", "replay"):
            for dv in y.vector_partition.items:
                dv.fill(-7.0)
            if how == "replay":
                g.launch()
            else:
                pa.mul_c_(y, A, x)
            for got, e, r in zip(y.own_values().items, yo, Ao.rows):
                assert np.array_equal(got, e[:r.n_own]), (rep, how)
        assert all(f for f, _ in _fused(A, x)), rep


def test_pa_mul5_over_a_one_rank_rccl_communicator():
    """The RCCL branch of the operator-level call on the one GPU there is: a part that "ghosts" three of its own values over a
    1-rank communicator (self-addressed ncclSend / ncclRecv, as test_rccl_single_rank_loopback) -- pack, RCCL group, own x own,
    own x ghost from the receive buffer, the unpack behind it on the comm stream -- five products in a row with changing x, plus
    the alpha/beta form and pa_mul_no_lat: against numpy, exactly.  (Between distinct GPUs the same calls run in
    tests/test_gpu_multiprocess.py where the box has them.)"""
    ctx = pa.context()
    os.environ["PA_MUL_FUSED_RCCL"] = "1"            # (opt-in since round 6; the handle decides at its first product)
    reload_switches()
    idbuf = C.create_string_buffer(L.UNIQUE_ID_BYTES)
    L.call("pa_comm_unique_id", idbuf)
    comm = C.c_void_p()
    L.call("pa_comm_create", ctx.h, idbuf.raw, 0, 1, C.byref(comm))
    i32 = lambda *v: np.array(v, np.int32)
    rng = np.random.default_rng(6)
    n, g = 6, 3
    oo_dense = np.where(rng.random((n, n)) < 0.6, rng.integers(-4, 5, (n, n)).astype(float), 0.0)
    oh_dense = np.where(rng.random((n, g)) < 0.7, rng.integers(-4, 5, (n, g)).astype(float), 0.0)
    oh_dense[0, :] = [1.0, -2.0, 3.0]

    def csr(D):
        rp = (1 + np.concatenate(([0], np.cumsum((D != 0).sum(1))))).astype(np.int32)
        cv = (np.nonzero(D)[1] + 1).astype(np.int32)
        return pa.DeviceCSR(pa.HostCSR(D.shape[0], D.shape[1], rp, cv, D[D != 0].astype(float)))
    oo, oh = csr(oo_dense), csr(oh_dense)
    one, ptrs = i32(1), i32(1, 4)
    plan = C.c_void_p()
    L.call("pa_plan_create", ctx.h, 1, n + g, 1, L.ptr(one), L.ptr(ptrs), L.ptr(i32(7, 8, 9)), 1, L.ptr(one), L.ptr(ptrs), L.ptr(i32(2, 4, 6)), 1,
           C.byref(plan))
    m = C.c_void_p()
    L.call("pa_matrix_create", ctx.h, oo.h, oh.h, plan, C.byref(m))
    b, c = pa.DeviceVector(n, g), pa.DeviceVector(n, 0)
    inside0 = ctx.fused_launches()[1]
    for rep in range(5):
        xo = rng.integers(-3, 4, n).astype(float) + rep
        b.upload(np.concatenate([xo, [99.0, 98.0, 97.0]]))            # stale ghosts: the exchange must replace them
        L.call("pa_mul5", m, comm, c.h, b.h, 1.0, 0.0)
        ghosts = xo[[1, 3, 5]]
        assert c.download().tolist() == (oo_dense @ xo + oh_dense @ ghosts).tolist(), rep
        assert b.download().tolist() == np.concatenate([xo, ghosts]).tolist(), rep
    yes = C.c_int()
    L.call("pa_matrix_ghost_from_buffer", m, C.byref(yes))
    assert yes.value == 1
    # round 5: each of those products was ONE launch on the compute stream beside the RCCL group -- its tail acquired the flag the
    # comm stream raises behind the receives, summed the boundary rows from the receive buffer and unpacked b's ghosts
    assert ctx.fused_launches()[1] - inside0 == 5
    c0 = c.download()
    L.call("pa_mul5", m, comm, c.h, b.h, 2.0, -1.0)
    assert c.download().tolist() == (-c0 + 2.0 * (oo_dense @ xo + oh_dense @ ghosts)).tolist()
    L.call("pa_mul_no_lat", m, comm, c.h, b.h)
    assert c.download().tolist() == (oo_dense @ xo + oh_dense @ ghosts).tolist()
    L.call("pa_matrix_destroy", m)
    L.call("pa_plan_destroy", plan)
    L.call("pa_comm_destroy", comm)
    os.environ.pop("PA_MUL_FUSED_RCCL", None)
    reload_switches()


def test_one_device_context_per_part():
    """PA_CTX_PER_PART=1: every part of a DebugArray in a device context of its own (own streams, own arena; GPU = part index
    mod visible GPUs) -- the single-process multi-GPU mode of SURVEY 8(b) (`DebugArray` over several GPUs).  On a 1-GPU box the
    contexts share the device and the same paths run: one push launch per context, events across contexts, per-context joins.
    The exchange-chain, transpose and renumbering modules and a slice of the parity module, in a child process under the switch."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PA_CTX_PER_PART="1")
    cmd = [sys.executable, "-m", "pytest", "-q", "-x", "-m", "gpu", "tests/test_gpu_exchange_chain.py", "tests/test_gpu_transpose.py",
           "tests/test_gpu_renumber.py", "tests/test_gpu_exchange.py", "tests/test_gpu_mul.py", "tests/test_gpu_setup.py", "-k",
           "not one_device_context and not hipgraph and (exchange or push or buffer or transpose or renumber or hand_partition or doc_examples or "
           "mul_hpcg or 27_parts or sub_assembled or periodic or reassembly)"]
    r = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=1500)
    assert r.returncode == 0 and " passed" in r.stdout, (r.stdout[-3000:] + r.stderr[-2000:])
