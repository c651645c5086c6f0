"""consistent! / assemble! / exchange! (src/p_vector.jl:587-755, src/primitives.jl:1020-1042) against the reference's literals and the oracle.
Bars: np.array_equal for everything but dot / norm (1e-13).  Needs a real MI355X (-m gpu)."""
import pytest

from gpu_helpers import *  # noqa: F401,F403

pytestmark = pytest.mark.gpu


def test_consistent_hand_partition(golden):
    c, parts = _hand(golden)
    v = pa.pvector_from_function(lambda i: 10.0 * i.part * (i.get_local_to_owner() == i.part), parts)
    pa.consistent_(v).wait()
    for vals, ind in zip(v.local_values().items, parts.items):
        assert vals.tolist() == (10.0 * ind.get_local_to_owner()).tolist()      # test/p_vector_tests.jl:116-124


def test_assemble_hand_partition(golden):
    c, parts = _hand(golden)
    v = pa.pfill(c["assemble_input"], parts)
    pa.assemble_(v).wait()
    assert [x.tolist() for x in v.local_values().items] == c["assemble_local_values"]   # :126-141
    assert v.collect().tolist() == c["assemble_collect"]                                  # :142


def test_doc_examples(golden):
    c = golden["doc_consistent"]
    parts = pa.uniform_partition(ranks(2), tuple(c["np"]), tuple(c["n"]), tuple(c["ghost"]))
    v = upload([np.array(b, float) for b in c["before"]], parts)
    pa.consistent_(v).wait()
    assert [x.tolist() for x in v.local_values().items] == c["after"]
    c = golden["doc_assemble"]
    v = upload([np.array(b, float) for b in c["before"]], parts)
    pa.assemble_(v).wait()
    assert [x.tolist() for x in v.local_values().items] == c["after"]


def test_repeated_exchanges_and_periodic_partition(orc):
    """Jacobi-style use (docs/jacobi_tutorial.jl:239-263): ghosted, periodic partition; many consistent! in a row."""
    parts = pa.uniform_partition(ranks(4), (2, 2), (6, 6), (True, True), (True, True))
    oparts = orc.uniform_partition((2, 2), (6, 6), (True, True), (True, True))
    host = [orc.hash_x(o.local_to_global) * (o.local_to_owner == o.part) for o in oparts]
    v = upload([h.copy() for h in host], parts)
    for _ in range(3):
        pa.consistent_(v).wait()
    orc.consistent(host, oparts)
    for a, b in zip(v.local_values().items, host):
        assert np.array_equal(a, b)
    pa.assemble_(v).wait()
    orc.assemble(host, oparts)
    for a, b in zip(v.local_values().items, host):
        assert np.array_equal(a, b)


def test_jacobi_tutorial_equals_serial_jacobi_bit_for_bit():
    """G14: docs/jacobi_tutorial.jl:239-263, jacobi_par(10,100,3) on uniform_partition(ranks,3,10,true) -- local ranges
    1:4, 3:7, 6:10, local order = global order (ghosts at both ends) -- with consistent! on the device every sweep and
    the tutorial's local update on the host.  The same operations as a serial Jacobi: own values equal bit for bit."""
    n, niters, p = 10, 100, 3
    parts = pa.uniform_partition(ranks(p), (p,), (n,), (True,))
    assert [i.get_local_to_global().tolist() for i in parts.items] == [[1, 2, 3, 4], [3, 4, 5, 6, 7], [6, 7, 8, 9, 10]]

    def init(ind):
        a = np.zeros(ind.n_local)
        a[0], a[-1] = 1.0, -1.0
        return a
    u, u_new = pa.pvector_from_function(init, parts), pa.pvector_from_function(init, parts)
    for _ in range(niters):
        pa.consistent_(u).wait()
        for dv, dn in zip(u.vector_partition.items, u_new.vector_partition.items):
            a, b = dv.download(0, len(dv)), dn.download(0, len(dn))
            b[1:-1] = 0.5 * (a[:-2] + a[2:])
            dn.upload(b)
        u, u_new = u_new, u
    s = np.zeros(n)
    s[0], s[-1] = 1.0, -1.0
    s_new = s.copy()
    for _ in range(niters):
        s_new[1:-1] = 0.5 * (s[:-2] + s[2:])
        s, s_new = s_new, s
    for dv, ind in zip(u.vector_partition.items, parts.items):
        own = ind.get_local_to_owner() == ind.part
        assert np.array_equal(dv.download(0, len(dv))[own], s[ind.get_local_to_global()[own] - 1])
    assert abs(s[4]) < 0.5 and s[1] > s[8]                  # the profile relaxes from +1 towards -1


def test_rccl_single_rank_loopback():
    """librccl is dlopen'ed, a 1-rank communicator works, and a self-addressed exchange moves the bytes.
    (Multi-GPU runs are the driver's; this pins the API plumbing on the 1-GPU box.)"""
    import pa_amd._lib as L
    ctx = pa.context()
    idbuf = C.create_string_buffer(L.UNIQUE_ID_BYTES)
    L.call("pa_comm_unique_id", idbuf)
    comm = C.c_void_p()
    L.call("pa_comm_create", ctx.h, idbuf.raw, 0, 1, C.byref(comm))
    v = pa.DeviceVector(6, 3).upload(np.arange(9, dtype=float))
    # part 1 "ghosts" three of its own values: snd side = ghost lids 7..9, rcv side = own lids 2,4,6
    one, ptrs = np.array([1], np.int32), np.array([1, 4], np.int32)
    plan = C.c_void_p()
    L.call("pa_plan_create", ctx.h, 1, 9, 1, L.ptr(one), L.ptr(ptrs), L.ptr(np.array([7, 8, 9], np.int32)),
           1, L.ptr(one), L.ptr(ptrs), L.ptr(np.array([2, 4, 6], np.int32)), 1, C.byref(plan))
    L.call("pa_exchange_pack", plan, v.h, L.CONSISTENT)
    L.call("pa_exchange_rccl", plan, comm, L.CONSISTENT)
    L.call("pa_exchange_finish", plan, v.h, L.CONSISTENT)
    assert v.download().tolist() == [0, 1, 2, 3, 4, 5, 1, 3, 5]
    L.call("pa_exchange_pack", plan, v.h, L.ASSEMBLE)
    L.call("pa_exchange_rccl", plan, comm, L.ASSEMBLE)
    L.call("pa_exchange_finish", plan, v.h, L.ASSEMBLE)
    assert v.download().tolist() == [0, 2, 2, 6, 4, 10, 0, 0, 0]
    # the same plan with a Float32 payload (pa_exchange_pack32 / _finish32: ncclFloat send / recv at the same element offsets)
    w = pa.DeviceVector32(6, 3).upload(np.arange(9, dtype=np.float32) + np.float32(0.5))
    L.call("pa_exchange_pack32", plan, w.h, L.CONSISTENT)
    L.call("pa_exchange_rccl", plan, comm, L.CONSISTENT)
    L.call("pa_exchange_finish32", plan, w.h, L.CONSISTENT)
    assert w.download().tolist() == [0.5, 1.5, 2.5, 3.5, 4.5, 5.5, 1.5, 3.5, 5.5]
    L.call("pa_exchange_pack32", plan, w.h, L.ASSEMBLE)
    L.call("pa_exchange_rccl", plan, comm, L.ASSEMBLE)
    L.call("pa_exchange_finish32", plan, w.h, L.ASSEMBLE)
    assert w.download().tolist() == [0.5, 3.0, 2.5, 7.0, 4.5, 11.0, 0, 0, 0]
    L.call("pa_exchange_pack", plan, v.h, L.CONSISTENT)             # (and Float64 again on the shared buffers)
    L.call("pa_exchange_rccl", plan, comm, L.CONSISTENT)
    L.call("pa_exchange_finish", plan, v.h, L.CONSISTENT)
    assert v.download().tolist() == [0, 2, 2, 6, 4, 10, 2, 6, 10]
    d = pa.DeviceVector(4, 0).upload(np.array([1.5, 2.0, 0.0, -1.0]))
    L.call("pa_comm_allreduce_sum", comm, C.c_void_p(d.data_ptr()), 4, L.STREAM_COMPUTE)
    assert d.download().tolist() == [1.5, 2.0, 0.0, -1.0]
    # dot -> device scalar -> all-reduce -> read back (the N>1 route of dot())
    a = pa.DeviceVector(4, 0).upload(np.array([1.0, 2.0, 3.0, 4.0]))
    L.call("pa_vec_dot", a.h, a.h, None)
    sp = C.c_void_p()
    L.call("pa_vec_dot_result", ctx.h, C.byref(sp))
    L.call("pa_comm_allreduce_sum", comm, sp, 1, L.STREAM_COMPUTE)
    out = C.c_double()
    L.call("pa_ctx_read_scalar", ctx.h, C.byref(out))
    assert out.value == 30.0
    L.call("pa_comm_barrier", comm)
    L.call("pa_plan_destroy", plan)
    L.call("pa_comm_destroy", comm)


def test_27_parts_26_neighbours(orc):
    """27-pt stencil on 3 x 3 x 3 parts: the middle part exchanges with all 26 neighbours (faces, edges, corners --
    messages of n^2, n and 1 values).  mul!, consistent! and assemble! against the oracle, bit-exact."""
    n = 5
    A, b = pa.build_p_matrix(ranks(27), n, n, n, 3 * n, 3 * n, 3 * n, 3, 3, 3)
    Ao, bo, _ = orc.hpcg_build_p_matrix(n, n, n, 3, 3, 3)
    nb = pa.assembly_neighbors(A.col_partition)[0].items
    assert len(nb[13]) == 26 and len(nb[0]) == 7                       # middle part / corner part
    xo = [orc.hash_x(c.local_to_global) * (c.local_to_owner == c.part) for c in Ao.cols]
    x = upload([v.copy() for v in xo], A.col_partition)
    y = pa.pzeros(A.row_partition)
    pa.mul_c_(y, A, x)
    yo = _oracle_mul(orc, Ao, xo)
    for got, exp, r in zip(y.own_values().items, yo, Ao.rows):
        assert np.array_equal(got, exp[:r.n_own])
    pa.mul_(y, A, pa.pones(A.col_partition))
    for got, exp in zip(y.own_values().items, b.own_values().items):
        assert np.array_equal(got, exp)
    host = [orc.hash_x(c.local_to_global + 7) for c in Ao.cols]          # ghosts carry their own values: assemble! adds them
    v = upload([h.copy() for h in host], A.col_partition)
    pa.assemble_(v).wait()
    orc.assemble(host, Ao.cols)
    for a_, b_ in zip(v.local_values().items, host):
        assert np.array_equal(a_, b_)


def test_norm_on_a_ghosted_uniform_partition_counts_own_values_only(orc):
    """uniform_partition(ranks,(2,2),(6,6),(true,true)) gives PermutedLocalIndices -- the local order is the
    extended box, own ids are not a prefix.  The device stores [own | ghost] whatever the local order is, so after a
    consistent! (every ghost holds its owner's value) dot / norm still reduce over own values only
    (src/p_vector.jl:1189-1206), the ghost exchange is bit-exact in LOCAL order, and axpby touches own values only."""
    for np_, n, ghost, per in (((2, 2), (6, 6), (True, True), None), ((2, 2), (10, 10), (2, 2), (True, True)), ((1, 2), (4, 4), (True, True), (True, True))):
        P = int(np.prod(np_))
        parts = pa.uniform_partition(ranks(P), np_, n, ghost, per)
        oparts = orc.uniform_partition(np_, n, ghost, per)
        assert [(i.n_own, i.n_ghost) for i in parts.items] == [(o.n_own, o.n_ghost) for o in oparts]
        assert not parts.items[0].own_is_contiguous_prefix
        f = lambda g: np.sin(g.astype(float)) + 2.0
        v = pa.pvector_from_function(lambda i: f(i.get_local_to_global()) * (i.get_local_to_owner() == i.part), parts)
        vo = [f(o.local_to_global) * (o.local_to_owner == o.part) for o in oparts]
        pa.consistent_(v).wait()
        orc.consistent(vo, oparts)
        for got, want in zip(v.local_values().items, vo):
            assert np.array_equal(got, want)                                # local order, ghosts included
        want = orc.norm2(vo, oparts)
        assert abs(pa.norm(v) - want) <= 1e-13 * want
        assert abs(pa.dot(v, v) - orc.dot(vo, vo, oparts)) <= 1e-13 * want * want
        w = pa.pzeros(parts)
        pa.axpby_(w, 2.0, v, 0.0)                                            # own values only: w's ghosts stay 0
        for got, o, src in zip(w.local_values().items, oparts, vo):
            exp = np.zeros(o.n_local)
            exp[o.own_to_local - 1] = 2.0 * src[o.own_to_local - 1]
            assert np.array_equal(got, exp)
        assert all(np.array_equal(g, src[o.own_to_local - 1]) for g, o, src in zip(v.own_values().items, oparts, vo))


@pytest.mark.parametrize("np_,n,ghost,per", [((1,), (6,), (1,), (True,)), ((2, 1), (14, 16), (1, 2), (True, True)),
                                            ((2, 1, 1), (10, 14, 14), (0, 0, 2), (True, True, True)),
                                            ((1, 3), (6, 4), (1, 0), (True, False)), ((4, 2, 1), (18, 9, 7), (2, 2, 1), (False, True, True))])
def test_assemble_zeroes_every_ghost_also_the_self_owned_ones(orc, np_, n, ghost, per):
    """assemble!(a) ends with fill!(ghost_values(a),0) (src/p_vector.jl:703-705).  A periodic direction with ONE part makes
    wrap-around copies owned by the part itself: ghosts that no message carries (compute_assembly_neighbors skips owner ==
    rank, src/p_range.jl:441-445) -- they are zeroed like the others, also on a part that exchanges nothing at all.
    (Found by tests/fuzz/fuzz_exchange.py in round 2: the device zeroed the ids of its send side only.)"""
    P = int(np.prod(np_))
    parts = pa.uniform_partition(ranks(P), np_, n, ghost, per)
    oparts = orc.uniform_partition(np_, n, ghost, per)
    assert any((o.local_to_owner[o.ghost_to_local - 1] == o.part).any() for o in oparts)      # self-owned ghosts exist
    rng = np.random.default_rng(3)
    wo = [rng.standard_normal(o.n_local) for o in oparts]
    it = iter([w.copy() for w in wo])
    w = pa.pvector_from_function(lambda ind: next(it), parts)
    pa.assemble_(w).wait()
    orc.assemble(wo, oparts)
    for got, want in zip(w.local_values().items, wo):
        assert np.array_equal(got, want)
    vo = [rng.standard_normal(o.n_local) for o in oparts]
    it = iter([v.copy() for v in vo])
    v = pa.pvector_from_function(lambda ind: next(it), parts)
    pa.consistent_(v).wait()
    orc.consistent(vo, oparts)
    for got, want in zip(v.local_values().items, vo):
        assert np.array_equal(got, want)


def test_all_parts_of_one_process_over_one_rccl_group_single_part():
    """pa_comm_create_all / pa_exchange_rccl_all (csrc/pa_rccl.cpp): the single-process multi-GPU form -- communicators of all parts
    from ONE ncclCommInitAll, the exchange ONE group over them.  A one-GPU box can run it for one part only (RCCL wants distinct
    devices): a part that ghosts three of its own values, Float64 and Float32 payloads; two parts on the one device are refused with a
    message that names the transports that do serve them."""
    import pa_amd._lib as L
    ctx = pa.context()
    arr = (C.c_void_p * 1)(ctx.h.value)
    comms = (C.c_void_p * 1)()
    L.call("pa_comm_create_all", arr, 1, comms)
    one, ptrs = np.array([1], np.int32), np.array([1, 4], np.int32)
    plan = C.c_void_p()
    L.call("pa_plan_create", ctx.h, 1, 9, 1, L.ptr(one), L.ptr(ptrs), L.ptr(np.array([7, 8, 9], np.int32)),
           1, L.ptr(one), L.ptr(ptrs), L.ptr(np.array([2, 4, 6], np.int32)), 1, C.byref(plan))
    plans = (C.c_void_p * 1)(plan.value)
    v = pa.DeviceVector(6, 3).upload(np.arange(9, dtype=float))
    for mode, want in ((L.CONSISTENT, [0, 1, 2, 3, 4, 5, 1, 3, 5]), (L.ASSEMBLE, [0, 2, 2, 6, 4, 10, 0, 0, 0])):
        L.call("pa_exchange_pack", plan, v.h, mode)
        L.call("pa_exchange_rccl_all", plans, comms, 1, mode)
        L.call("pa_exchange_finish", plan, v.h, mode)
        assert v.download().tolist() == want
    w = pa.DeviceVector32(6, 3).upload(np.arange(9, dtype=np.float32) + np.float32(0.25))
    L.call("pa_exchange_pack32", plan, w.h, L.CONSISTENT)
    L.call("pa_exchange_rccl_all", plans, comms, 1, L.CONSISTENT)
    L.call("pa_exchange_finish32", plan, w.h, L.CONSISTENT)
    assert w.download().tolist() == [0.25, 1.25, 2.25, 3.25, 4.25, 5.25, 1.25, 3.25, 5.25]
    two = (C.c_void_p * 2)(ctx.h.value, ctx.h.value)
    out2 = (C.c_void_p * 2)()
    with pytest.raises(L.PAError, match="share device"):
        L.call("pa_comm_create_all", two, 2, out2)
    L.call("pa_plan_destroy", plan)
    L.call("pa_comm_destroy", C.c_void_p(comms[0]))
