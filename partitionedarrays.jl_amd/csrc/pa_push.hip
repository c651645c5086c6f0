// pa_push.hip -- the "push" transport of the ghost exchange (round 4): the pack kernel stores every send slice STRAIGHT INTO THE
// RECEIVE BUFFER of the part it goes to.  No send buffer, no copy engine, no library kernels between pack and unpack.
//
// Reference: exchange!(buffer_rcv, buffer_snd, graph) inside assemble_impl! (src/p_vector.jl:595-601): pack loop, then
// rcv[i].data[ptrs_rcv[k]..] <- snd[j].data[ptrs_snd[l]..] per directed edge (src/primitives.jl:1020-1042; MPI: Isend / Irecv per
// neighbour, src/mpi_array.jl:575-614).  Fusing the two is invisible to the caller: buffer_rcv holds the same values when wait(t)
// returns; buffer_snd is a private scratch of the cache that nothing else reads.
//
// Two shapes:
//  (a) every part in ONE process (DebugArray, src/debug_array.jl:110-117,250): the neighbours' buffers are ordinary device
//      pointers.  pa_exchange_push_local packs AND delivers all parts with ONE launch (per device) where the round-3 path
//      queued one pack kernel per part and one hipMemcpyAsync per directed edge (config 5 on 8 parts: 8 + ~40 operations).
//  (b) one part per PROCESS (MPIArray): the neighbours' receive buffers are mapped with hipIpcOpenMemHandle
//      (pa_plan_ipc_blob / pa_plan_ipc_connect; the host language moves the blobs, as it moves the ncclUniqueId), the stores
//      travel over xGMI (or stay inside the GPU when ranks share one: how this is tested on 1-GPU boxes), and arrival is a
//      sequence number the LAST block of the push kernel writes behind its payload -- release at system scope -- into a flag
//      word in the receiver's memory; the receiver's comm stream runs a one-wavefront wait kernel before the event wait(t)
//      waits on.  Flow control: the receiver acknowledges (pa_exchange_finish -> pa_ipc_ack) after its unpack, and the next
//      push into that buffer spins on the acknowledgement of the one before (it has long arrived in any iterative solver: there
//      is a dot product between two products).  PA_TRANSPORT=ipc; RCCL stays the default transport (`north_star`).
//      A wait gives up after PA_IPC_TIMEOUT_S (default 30) and raises the link's status word instead of hanging the GPU.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstring>
#include <map>
#include <mutex>
#include <vector>

#include "pa_internal.h"

#include "pa_push_dev.h"

// (a) all parts of a process: block -> part through block_part
__global__ __launch_bounds__(256) void k_push_local(const pa_push_part *__restrict__ parts, const pa_push_seg *__restrict__ segs,
                                                    const int32_t *__restrict__ block_part, pa_push_vecs vecs) {
  const int pi = block_part[blockIdx.x];
  const pa_push_part P = parts[pi];
  const int p = ((int)blockIdx.x - P.blk0) * 256 + (int)threadIdx.x;
  if (p >= P.n) return;
  const double val = vecs.v[pi][P.idx[p]];
  const int s = push_find_seg(segs, P.seg0, P.nseg, p);
  segs[s].dst[p - segs[s].start] = val;
}

// the same launch completing consistent! as well: every delivered value is also stored into the receiving part's ghost entry -- the
// unpack loop of src/p_vector.jl:603-611 with f = insert, done by the lane that has the value in a register already
__global__ __launch_bounds__(256) void k_push_unpack(const pa_push_part *__restrict__ parts, const pa_push_seg *__restrict__ segs,
                                                     const int32_t *__restrict__ block_part, pa_push_vecs vecs) {
  const int pi = block_part[blockIdx.x];
  const pa_push_part P = parts[pi];
  const int p = ((int)blockIdx.x - P.blk0) * 256 + (int)threadIdx.x;
  if (p >= P.n) return;
  const double val = vecs.v[pi][P.idx[p]];
  const int s = push_find_seg(segs, P.seg0, P.nseg, p);
  const pa_push_seg S = segs[s];
  S.dst[p - S.start] = val;
  const_cast<double *>(vecs.v[S.upart])[S.uidx[p - S.start]] = val;
}

struct pa_unpack_part { const int32_t *idx; const double *buf; int32_t n, blk0; };
struct pa_unpack_vecs { double *v[PA_PUSH_MAX_PARTS]; };
__global__ __launch_bounds__(256) void k_unpack_insert_multi(const pa_unpack_part *__restrict__ parts, const int32_t *__restrict__ block_part,
                                                             pa_unpack_vecs vecs) {
  const int pi = block_part[blockIdx.x];
  const pa_unpack_part P = parts[pi];
  const int p = ((int)blockIdx.x - P.blk0) * 256 + (int)threadIdx.x;
  if (p < P.n) vecs.v[pi][P.idx[p]] = P.buf[p];
}

// assemble!'s wait(t) of all parts of a device in one launch: the ordered adds of every part (k_unpack_add: one lane per target adds
// its sources in ascending order) in the first blocks, fill!(ghost_values, 0) of every part (src/p_vector.jl:703-705) in the rest
struct pa_add_part { const int32_t *tgt, *tptr, *tp; const double *buf; int32_t n_tgt, blk0; };
struct pa_add_vecs { double *v[PA_PUSH_MAX_PARTS]; int32_t ghost0[PA_PUSH_MAX_PARTS], n_ghost[PA_PUSH_MAX_PARTS], zblk0[PA_PUSH_MAX_PARTS + 1]; int32_t n_parts, n_add_blocks; };
__global__ __launch_bounds__(256) void k_unpack_add_multi(const pa_add_part *__restrict__ parts, const int32_t *__restrict__ block_part, pa_add_vecs vecs) {
  const int b = (int)blockIdx.x;
  if (b < vecs.n_add_blocks) {
    const int pi = block_part[b];
    const pa_add_part P = parts[pi];
    const int k = (b - P.blk0) * 256 + (int)threadIdx.x;
    if (k >= P.n_tgt) return;
    double *v = vecs.v[pi];
    const int lid = P.tgt[k];
    // (a target can be a GHOST entry -- the wrap-around copies of a periodic direction: the reference adds into it and then zeroes
    // every ghost; the zeroing runs in this very launch, so such a target is simply left to it)
    if (lid >= vecs.ghost0[pi]) return;
    double acc = v[lid];
    for (int j = P.tptr[k]; j < P.tptr[k + 1]; ++j) acc = acc + P.buf[P.tp[j]];
    v[lid] = acc;
    return;
  }
  const int zb = b - vecs.n_add_blocks;
  int pi = 0;
  while (pi + 1 < vecs.n_parts && zb >= vecs.zblk0[pi + 1]) ++pi;
  const int g = (zb - vecs.zblk0[pi]) * 256 + (int)threadIdx.x;
  if (g < vecs.n_ghost[pi]) vecs.v[pi][vecs.ghost0[pi] + g] = 0.0;
}

// (b) one part per process
__global__ __launch_bounds__(256) void k_push_ipc(const int32_t *__restrict__ idx, int n, const pa_push_seg *__restrict__ segs, int nseg,
                                                  const double *__restrict__ v, unsigned long long seq, unsigned *done, long long ticks,
                                                  int *status) {
  __shared__ int ok;
  (void)pa_push_ipc_block(&ok, idx, n, segs, nseg, v, seq, done, ticks, status, (int)blockIdx.x, (int)gridDim.x);
}

__global__ void k_wait_flags(const unsigned long long *__restrict__ flags, const int32_t *__restrict__ which, int n, unsigned long long seq,
                             long long ticks, int *status) {
  for (int i = threadIdx.x; i < n; i += blockDim.x)
    if (!flag_wait(flags + which[i], seq, ticks)) atomicExch(status, 1);
}

__global__ void k_write_flags(unsigned long long *const *__restrict__ dst, int n, unsigned long long seq) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) flag_store(dst[i], seq);
}

// ---------------------------------------------------------------------------------------------------------------------------
static inline pa_plan::side &out_side(pa_plan *p, int mode) { return mode == PA_ASSEMBLE ? p->snd : p->rcv; }
static inline pa_plan::side &in_side(pa_plan *p, int mode) { return mode == PA_ASSEMBLE ? p->rcv : p->snd; }

struct pa_push_table {
  std::vector<pa_plan *> key;
  std::vector<uint64_t> serials;    // the plans' serial numbers: an address may come back with another plan behind it
  // one launch per device (context) that holds parts of the group
  struct launch {
    pa_ctx *ctx = nullptr;
    std::vector<int> parts;           // group indices, in order
    pa_push_part *d_parts = nullptr;
    pa_push_seg *d_segs = nullptr;
    int32_t *d_block_part = nullptr;
    int n_blocks = 0;
    // the matching multi-part unpack (insert) of the parts on this device
    pa_unpack_part *d_uparts = nullptr;
    int32_t *d_ublock_part = nullptr;
    int n_ublocks = 0;
    // (assemble! tables) the matching multi-part ordered add
    pa_add_part *d_aparts = nullptr;
    int32_t *d_ablock_part = nullptr;
    int n_ablocks = 0;
    hipEvent_t ev = nullptr;
  };
  std::vector<launch> launches;
  void free_all() {
    for (launch &l : launches) {
      (void)hipSetDevice(l.ctx->device);
      (void)pa_raw_free(l.d_parts); (void)pa_raw_free(l.d_segs); (void)pa_raw_free(l.d_block_part);
      (void)pa_raw_free(l.d_uparts); (void)pa_raw_free(l.d_ublock_part);
      (void)pa_raw_free(l.d_aparts); (void)pa_raw_free(l.d_ablock_part);
      if (l.ev) (void)hipEventDestroy(l.ev);
    }
    launches.clear();
  }
};

template <class T>
static int upload_vec(const std::vector<T> &h, T **d) {
  PA_HIP(pa_raw_malloc(d, sizeof(T) * std::max<size_t>(1, h.size())));
  if (!h.empty()) PA_HIP(pa_h2d(*d, h.data(), sizeof(T) * h.size()));
  return PA_OK;
}

static int build_local_table(pa_plan *const *plans, int n_parts, int mode, pa_push_table **out) {
  pa_push_table *T = new pa_push_table();
  T->key.assign(plans, plans + n_parts);
  for (int r = 0; r < n_parts; ++r) T->serials.push_back(plans[r]->serial);
  std::map<pa_ctx *, int> at;
  for (int r = 0; r < n_parts; ++r) {
    pa_ctx *c = plans[r]->ctx;
    if (!at.count(c)) { at[c] = (int)T->launches.size(); T->launches.emplace_back(); T->launches.back().ctx = c; }
    T->launches[at[c]].parts.push_back(r);
  }
  for (pa_push_table::launch &l : T->launches) {
    if ((int)l.parts.size() > PA_PUSH_MAX_PARTS) { delete T; pa_set_err("more than %d parts on one device", PA_PUSH_MAX_PARTS); return PA_ERR_ARG; }
    std::vector<pa_push_part> parts;
    std::vector<pa_push_seg> segs;
    std::vector<int32_t> bp, ubp, abp;
    std::vector<pa_unpack_part> uparts;
    std::vector<pa_add_part> aparts;
    for (size_t k = 0; k < l.parts.size(); ++k) {
      pa_plan *ps = plans[l.parts[k]];
      pa_plan::side &o = out_side(ps, mode);
      pa_push_part P;
      P.idx = o.d_idx; P.n = (int32_t)o.n; P.seg0 = (int32_t)segs.size(); P.nseg = 0; P.blk0 = (int32_t)bp.size();
      for (size_t j = 0; j < o.nbr.size(); ++j) {
        const int len = o.ptrs[j + 1] - o.ptrs[j];
        if (!len) continue;
        const int q = o.nbr[j];
        if (q < 0 || q >= n_parts) { T->free_all(); delete T; pa_set_err("part %d: neighbour %d out of range", ps->part, q); return PA_ERR_ARG; }
        pa_plan *pr = plans[q];
        pa_plan::side &in = in_side(pr, mode);
        auto it = std::find(in.nbr.begin(), in.nbr.end(), ps->part);
        if (it == in.nbr.end()) { T->free_all(); delete T; pa_set_err("inconsistent ExchangeGraph: part %d sends to %d, which does not receive from it", ps->part, q); return PA_ERR_ARG; }
        const size_t i = it - in.nbr.begin();
        if (in.ptrs[i + 1] - in.ptrs[i] != len) { T->free_all(); delete T; pa_set_err("slice length mismatch between parts %d and %d", ps->part, q); return PA_ERR_ARG; }
        pa_push_seg S;
        S.start = o.ptrs[j]; S.len = len; S.dst = in.d_buf + in.ptrs[i]; S.arrive = nullptr; S.ack = nullptr;
        // (the receiving part's place among this launch's vectors, when it is on this device)
        S.uidx = nullptr; S.upart = -1;
        for (size_t kk = 0; kk < l.parts.size(); ++kk)
          if (l.parts[kk] == q) { S.upart = (int32_t)kk; S.uidx = in.d_idx + in.ptrs[i]; }
        segs.push_back(S);
        ++P.nseg;
      }
      const int nb = (int)((o.n + 255) / 256);
      for (int b = 0; b < nb; ++b) bp.push_back((int32_t)k);
      if (P.nseg == 0) P.n = 0;
      parts.push_back(P);
      pa_plan::side &in = in_side(ps, mode);
      pa_unpack_part U;
      U.idx = in.d_idx; U.buf = in.d_buf; U.n = (int32_t)in.n; U.blk0 = (int32_t)ubp.size();
      const int nub = (int)((in.n + 255) / 256);
      for (int b = 0; b < nub; ++b) ubp.push_back((int32_t)k);
      uparts.push_back(U);
      if (mode == PA_ASSEMBLE) {
        pa_add_part Ap;
        Ap.tgt = ps->d_tgt; Ap.tptr = ps->d_tptr; Ap.tp = ps->d_tp; Ap.buf = in.d_buf; Ap.n_tgt = (int32_t)ps->n_tgt; Ap.blk0 = (int32_t)abp.size();
        for (int b = 0; b < (int)((ps->n_tgt + 255) / 256); ++b) abp.push_back((int32_t)k);
        aparts.push_back(Ap);
      }
    }
    // every receiving slice must have a sender inside the group
    (void)hipSetDevice(l.ctx->device);
    int st = upload_vec(parts, &l.d_parts);
    if (st == PA_OK) st = upload_vec(segs, &l.d_segs);
    if (st == PA_OK) st = upload_vec(bp, &l.d_block_part);
    if (st == PA_OK) st = upload_vec(uparts, &l.d_uparts);
    if (st == PA_OK) st = upload_vec(ubp, &l.d_ublock_part);
    if (st == PA_OK && mode == PA_ASSEMBLE) st = upload_vec(aparts, &l.d_aparts);
    if (st == PA_OK && mode == PA_ASSEMBLE) st = upload_vec(abp, &l.d_ablock_part);
    l.n_ablocks = (int)abp.size();
    if (st == PA_OK && hipEventCreateWithFlags(&l.ev, hipEventDisableTiming) != hipSuccess) { pa_set_err("hipEventCreate failed"); st = PA_ERR_HIP; }
    if (st != PA_OK) { T->free_all(); delete T; return st; }
    l.n_blocks = (int)bp.size();
    l.n_ublocks = (int)ubp.size();
  }
  // slices expected by a receiver that no sender of the group provides (an inconsistent graph seen from the other end)
  for (int r = 0; r < n_parts; ++r) {
    pa_plan::side &in = in_side(plans[r], mode);
    for (size_t i = 0; i < in.nbr.size(); ++i) {
      const int s = in.nbr[i];
      bool ok = s >= 0 && s < n_parts;
      if (ok) { pa_plan::side &o = out_side(plans[s], mode); ok = std::find(o.nbr.begin(), o.nbr.end(), r) != o.nbr.end(); }
      if (!ok) { T->free_all(); delete T; pa_set_err("inconsistent ExchangeGraph: part %d receives from %d, which does not send to it", r, s); return PA_ERR_ARG; }
    }
  }
  *out = T;
  return PA_OK;
}

static int local_table(pa_plan *const *plans, int n_parts, int mode, pa_push_table **out) {
  pa_push_table *&T = plans[0]->push[mode];
  bool same = T && (int)T->key.size() == n_parts && std::equal(T->key.begin(), T->key.end(), plans);
  for (int r = 0; same && r < n_parts; ++r) same = T->serials[r] == plans[r]->serial;
  if (T && !same) {
    T->free_all();
    delete T;
    T = nullptr;
  }
  if (!T) PA_TRY(build_local_table(plans, n_parts, mode, &T));
  *out = T;
  return PA_OK;
}

// pack + exchange! of every part of this process in one launch per device: what pa_exchange_pack on every part followed by
// pa_exchange_local does, minus the send buffers.  Afterwards every plan is in the state pa_exchange_local leaves it in
// (pa_exchange_finish is next).
static int push_local_impl(pa_plan *const *plans, int32_t n_parts, pa_vec *const *v, int mode, bool one_stream);
extern "C" int pa_exchange_push_local(pa_plan *const *plans, int32_t n_parts, pa_vec *const *v, int mode) {
  return push_local_impl(plans, n_parts, v, mode, false);
}
// The same launch ON THE COMPUTE STREAM, no events: what pa_mul_all records into a hipGraph (a graph with edges between two streams
// replays 2.6 x slower than the eager calls; one chain of kernels replays faster than they launch).  One device only; the readers of
// the receive buffers and pa_exchange_finish_all_insert(..., 3) follow on the same stream.
int pa_exchange_push_local_one_stream(pa_plan *const *plans, int32_t n_parts, pa_vec *const *v, int mode) {
  return push_local_impl(plans, n_parts, v, mode, true);
}
static int push_local_impl(pa_plan *const *plans, int32_t n_parts, pa_vec *const *v, int mode, bool one_stream) {
  PA_REQUIRE(plans && v && n_parts > 0 && (mode == PA_ASSEMBLE || mode == PA_CONSISTENT), "bad arguments");
  for (int r = 0; r < n_parts; ++r) {
    PA_REQUIRE(plans[r] && v[r] && plans[r]->part == r, "plans[%d] is not the plan of part %d", r, r);
    PA_REQUIRE(v[r]->n_own + v[r]->n_ghost == plans[r]->n_local, "part %d: vector has %lld local values, plan expects %lld", r,
               (long long)(v[r]->n_own + v[r]->n_ghost), (long long)plans[r]->n_local);
    PA_REQUIRE(plans[r]->phase == 0, "part %d: exchange already in flight on this plan (missing pa_exchange_finish)", r);
  }
  pa_push_table *T = nullptr;
  PA_TRY(local_table(plans, n_parts, mode, &T));
  // (a group nothing travels in -- the only part of a run -- costs no stream operation at all: a cross-stream event pair between
  //  two products is ~8 us of idle GPU, measured as 0.689 against 0.673 ms per step of the 256^3 headline loop)
  bool any = false;
  for (pa_push_table::launch &l : T->launches) any = any || l.n_blocks || l.n_ublocks;
  if (!any) {
    for (int r = 0; r < n_parts; ++r) { plans[r]->phase = 2; plans[r]->mode = mode; plans[r]->own_comm_stream = false; plans[r]->ev_wait = nullptr; }
    return PA_OK;
  }
  if (one_stream) {
    PA_REQUIRE(T->launches.size() == 1, "the one-stream order needs all parts on one device");
    pa_push_table::launch &l = T->launches[0];
    pa_ctx *c = l.ctx;
    PA_HIP(hipSetDevice(c->device));
    if (l.n_blocks) {
      pa_push_vecs vv;
      for (size_t k = 0; k < l.parts.size(); ++k) vv.v[k] = v[l.parts[k]]->d;
      hipLaunchKernelGGL(k_push_local, dim3(l.n_blocks), dim3(256), 0, c->s[0], l.d_parts, l.d_segs, l.d_block_part, vv);
      PA_HIP(hipGetLastError());
    }
    for (int r : l.parts) { plans[r]->phase = 2; plans[r]->mode = mode; plans[r]->own_comm_stream = false; plans[r]->ev_wait = nullptr; }
    return PA_OK;
  }
  for (pa_push_table::launch &l : T->launches) {
    pa_ctx *c = l.ctx;
    PA_HIP(hipSetDevice(c->device));
    // the comm stream must see everything the compute stream wrote into the vectors so far
    PA_HIP(hipEventRecord(c->ev_compute, c->s[0]));
    PA_HIP(hipStreamWaitEvent(c->s[1], c->ev_compute, 0));
    if (l.n_blocks) {
      pa_push_vecs vv;
      for (size_t k = 0; k < l.parts.size(); ++k) vv.v[k] = v[l.parts[k]]->d;
      hipLaunchKernelGGL(k_push_local, dim3(l.n_blocks), dim3(256), 0, c->s[1], l.d_parts, l.d_segs, l.d_block_part, vv);
      PA_HIP(hipGetLastError());
    }
    PA_HIP(hipEventRecord(l.ev, c->s[1]));
  }
  // a part's slices have arrived when every device that sends to it has finished its launch: with one device that is the
  // launch's own event (recorded once, shared by the parts); with several, each comm stream waits for the others' launches
  for (pa_push_table::launch &l : T->launches) {
    if (T->launches.size() > 1) {
      PA_HIP(hipSetDevice(l.ctx->device));
      for (pa_push_table::launch &o : T->launches) if (&o != &l) PA_HIP(hipStreamWaitEvent(l.ctx->s[1], o.ev, 0));
      PA_HIP(hipEventRecord(l.ev, l.ctx->s[1]));
    }
    for (int r : l.parts) {
      pa_plan *p = plans[r];
      p->phase = 2; p->mode = mode; p->own_comm_stream = false;
      p->ev_wait = l.ev;
    }
  }
  return PA_OK;
}

int pa_exchange_push_unpack_one_stream(pa_plan *const *plans, int32_t n_parts, pa_vec *const *v) {
  PA_REQUIRE(plans && v && n_parts > 0, "bad arguments");
  for (int r = 0; r < n_parts; ++r) {
    PA_REQUIRE(plans[r] && v[r] && plans[r]->part == r, "plans[%d] is not the plan of part %d", r, r);
    PA_REQUIRE(v[r]->n_own + v[r]->n_ghost == plans[r]->n_local, "part %d: vector has %lld local values, plan expects %lld", r,
               (long long)(v[r]->n_own + v[r]->n_ghost), (long long)plans[r]->n_local);
    PA_REQUIRE(plans[r]->phase == 0, "part %d: exchange already in flight on this plan (missing pa_exchange_finish)", r);
  }
  pa_push_table *T = nullptr;
  PA_TRY(local_table(plans, n_parts, PA_CONSISTENT, &T));
  PA_REQUIRE(T->launches.size() == 1, "the one-stream order needs all parts on one device");
  pa_push_table::launch &l = T->launches[0];
  if (l.n_blocks) {
    pa_ctx *c = l.ctx;
    PA_HIP(hipSetDevice(c->device));
    pa_push_vecs vv;
    for (size_t k = 0; k < l.parts.size(); ++k) vv.v[k] = v[l.parts[k]]->d;
    hipLaunchKernelGGL(k_push_unpack, dim3(l.n_blocks), dim3(256), 0, c->s[0], l.d_parts, l.d_segs, l.d_block_part, vv);
    PA_HIP(hipGetLastError());
    // (a later exchange that comes by the comm stream orders itself behind the compute stream when it starts: pa_exchange_pack /
    //  push_local_impl record ev_compute first -- the products of this step that still read the receive buffers are in front of it)
  }
  for (int r = 0; r < n_parts; ++r) { plans[r]->phase = 0; plans[r]->mode = PA_CONSISTENT; plans[r]->own_comm_stream = false; plans[r]->ev_wait = nullptr; }
  return PA_OK;
}

// unpack (insert) of every part of the group with one launch per device, behind whatever read the receive buffers since the
// arrival.  on_comm_stream = 0: on the compute streams (the caller's readers ran there); 1: on the comm streams, and the compute
// streams wait for it; 2: on the comm streams, the compute streams join later (pa_exchange_join_all); 3: on the compute stream,
// nothing to wait for (the one-stream order of a graph capture).  consistent! only.
int pa_exchange_finish_all_insert(pa_plan *const *plans, int32_t n_parts, pa_vec *const *v, int on_comm_stream) {
  pa_push_table *T = plans[0]->push[PA_CONSISTENT];
  PA_REQUIRE(T && (int)T->key.size() == n_parts && std::equal(T->key.begin(), T->key.end(), plans), "no push table for these plans");
  bool any = false;
  for (pa_push_table::launch &l : T->launches) any = any || l.n_blocks || l.n_ublocks;
  for (pa_push_table::launch &l : T->launches) {
    pa_ctx *c = l.ctx;
    if (!any) { for (int r : l.parts) { plans[r]->phase = 0; plans[r]->ev_wait = nullptr; } continue; }
    PA_HIP(hipSetDevice(c->device));
    const bool one_stream = on_comm_stream == 3;         // behind pa_exchange_push_local_one_stream: everything is on the compute stream
    if (one_stream) on_comm_stream = 0;
    hipStream_t st = on_comm_stream ? c->s[1] : c->s[0];
    if (!on_comm_stream && !one_stream) PA_HIP(hipStreamWaitEvent(c->s[0], l.ev, 0));
    if (l.n_ublocks) {
      pa_unpack_vecs vv;
      for (size_t k = 0; k < l.parts.size(); ++k) vv.v[k] = v[l.parts[k]]->d;
      hipLaunchKernelGGL(k_unpack_insert_multi, dim3(l.n_ublocks), dim3(256), 0, st, l.d_uparts, l.d_ublock_part, vv);
      PA_HIP(hipGetLastError());
    }
    if (on_comm_stream) {
      PA_HIP(hipEventRecord(l.ev, c->s[1]));             // wait(t) of the whole product: the compute stream joins here ...
      if (on_comm_stream == 2) continue;                 // ... or later, in pa_exchange_join_all (the plans stay "in flight")
      PA_HIP(hipStreamWaitEvent(c->s[0], l.ev, 0));      // (the next push is on the comm stream, behind this unpack)
    } else if (!c->capturing) {                          // the next pack (comm stream) must not overwrite buffers this unpack reads
      PA_HIP(hipEventRecord(c->ev_compute, c->s[0]));
      PA_HIP(hipStreamWaitEvent(c->s[1], c->ev_compute, 0));
    }
    for (int r : l.parts) { plans[r]->phase = 0; plans[r]->ev_wait = nullptr; }
  }
  return PA_OK;
}

// wait(t) of every part behind pa_exchange_push_local, one call: consistent! -- one unpack launch per device for all its parts;
// assemble! -- the parts' ordered adds one after the other (k_unpack_add has no multi-part form).  What pa_exchange_finish on every
// part does, without a host round trip per part.
extern "C" int pa_exchange_finish_all(pa_plan *const *plans, int32_t n_parts, pa_vec *const *v, int mode) {
  PA_REQUIRE(plans && v && n_parts > 0 && (mode == PA_ASSEMBLE || mode == PA_CONSISTENT), "bad arguments");
  for (int r = 0; r < n_parts; ++r) PA_REQUIRE(plans[r] && v[r] && plans[r]->phase >= 1 && plans[r]->mode == mode, "part %d: no exchange of this mode in flight", r);
  pa_push_table *T = plans[0]->push[PA_CONSISTENT];
  bool table = mode == PA_CONSISTENT && T && (int)T->key.size() == n_parts && std::equal(T->key.begin(), T->key.end(), plans);
  for (int r = 0; r < n_parts && table; ++r) table = plans[r]->phase == 2 && !plans[r]->own_comm_stream;
  if (table) return pa_exchange_finish_all_insert(plans, n_parts, v, 0);
  pa_push_table *Ta = plans[0]->push[PA_ASSEMBLE];
  bool atable = mode == PA_ASSEMBLE && Ta && (int)Ta->key.size() == n_parts && std::equal(Ta->key.begin(), Ta->key.end(), plans);
  for (int r = 0; r < n_parts && atable; ++r) atable = plans[r]->phase == 2 && !plans[r]->ipc && Ta->serials[r] == plans[r]->serial;
  if (atable) {
    for (pa_push_table::launch &l : Ta->launches) {
      pa_ctx *c = l.ctx;
      PA_HIP(hipSetDevice(c->device));
      bool traffic = l.n_blocks > 0;
      pa_add_vecs vv;
      vv.n_parts = (int32_t)l.parts.size(); vv.n_add_blocks = l.n_ablocks;
      int zb = 0;
      for (size_t k = 0; k < l.parts.size(); ++k) {
        pa_vec *w = v[l.parts[k]];
        vv.v[k] = w->d; vv.ghost0[k] = (int32_t)w->n_own; vv.n_ghost[k] = (int32_t)w->n_ghost; vv.zblk0[k] = zb;
        zb += (int)((w->n_ghost + 255) / 256);
      }
      vv.zblk0[l.parts.size()] = zb;
      if (traffic || plans[l.parts[0]]->ev_wait) PA_HIP(hipStreamWaitEvent(c->s[0], l.ev, 0));        // wait(t)
      if (l.n_ablocks + zb > 0)
        hipLaunchKernelGGL(k_unpack_add_multi, dim3(l.n_ablocks + zb), dim3(256), 0, c->s[0], l.d_aparts, l.d_ablock_part, vv);
      PA_HIP(hipGetLastError());
      if (!c->capturing) {                               // the next pack (comm stream) must not overwrite buffers this launch reads
        PA_HIP(hipEventRecord(c->ev_compute, c->s[0]));
        PA_HIP(hipStreamWaitEvent(c->s[1], c->ev_compute, 0));
      }
      for (int r : l.parts) { plans[r]->phase = 0; plans[r]->ev_wait = nullptr; }
    }
    return PA_OK;
  }
  for (int r = 0; r < n_parts; ++r) PA_TRY(pa_exchange_finish(plans[r], v[r], mode));
  return PA_OK;
}

// wait(t) after pa_exchange_finish_all_insert(..., 2): the compute streams wait for the unpack queued on the comm streams
int pa_exchange_join_all(pa_plan *const *plans, int32_t n_parts) {
  pa_push_table *T = plans[0]->push[PA_CONSISTENT];
  PA_REQUIRE(T && (int)T->key.size() == n_parts && std::equal(T->key.begin(), T->key.end(), plans), "no push table for these plans");
  bool any = false;
  for (pa_push_table::launch &l : T->launches) any = any || l.n_blocks || l.n_ublocks;
  for (pa_push_table::launch &l : T->launches) {
    if (any) {
      PA_HIP(hipSetDevice(l.ctx->device));
      PA_HIP(hipStreamWaitEvent(l.ctx->s[0], l.ev, 0));
    }
    for (int r : l.parts) { plans[r]->phase = 0; plans[r]->ev_wait = nullptr; }
  }
  return PA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// (b) hipIpc link of one part per process
// ---------------------------------------------------------------------------------------------------------------------------
#define PA_IPC_MAGIC 0x70614970u   /* "paIp" */
struct ipc_header {
  uint32_t magic;
  int32_t part;
  int32_t n_snd_nbr, n_rcv_nbr;
  int64_t n_snd, n_rcv;
  int64_t off_snd, off_rcv, off_flags;   // byte offsets of the two receive buffers and of the flag words inside the CHUNK
  hipIpcMemHandle_t h_region;            // the pool chunk the plan's region is cut from
};

struct pa_ipc_link {
  // ONE allocation per plan holds everything a neighbour touches -- both buffers and the flag words -- so that one ipc handle names
  // it (hipIpcGetMemHandle wants the base of an allocation of its own: the 8-byte buffer of a side without traffic, made by an
  // ordinary hipMalloc, was refused with "invalid argument")
  char *d_region = nullptr;
  size_t region_bytes = 0, chunk_off = 0;
  int chunk = -1;                          // the pool chunk the region is cut from (exported once, ipc_region_take)
  int64_t off_snd = 0, off_rcv = 0, off_flags = 0;
  unsigned long long *d_flags = nullptr;   // [arrive CONSISTENT: NS | arrive ASSEMBLE: NR | ack CONSISTENT: NR | ack ASSEMBLE: NS]
  int64_t n_flags = 0;
  unsigned *d_done = nullptr;              // [0] the push kernel's block counter, [1] the fused launch's tail-block counter
  int *h_status = nullptr;                 // host-pinned, device-visible: 0 ok, 1 an arrival wait timed out, 2 an acknowledgement wait did
  struct peer { void *region = nullptr, *snd = nullptr, *rcv = nullptr, *flags = nullptr; };
  std::map<int, peer> peers;               // part -> its mapped buffers
  struct per_mode {
    pa_push_seg *d_segs = nullptr;
    int nseg = 0;
    int32_t *d_wait = nullptr;             // flag indices (in my d_flags) of the senders with a non-empty slice
    int n_wait = 0;
    unsigned long long **d_ack_dst = nullptr;   // where I acknowledge: one flag per sender, in the senders' memory
    int n_ack = 0;
  } m[2];
  long long ticks = 0;
  bool connected = false;
  pa_fused_comm *d_xcomm = nullptr;         // device copy of what the fused product launch does for consistent! over this link
  std::vector<int64_t> ack_idx[2];          // (host) my flag words the receivers of my slices acknowledge in, per mode
};

static int ipc_flags_layout(const pa_plan *p, int64_t *aC, int64_t *aA, int64_t *kC, int64_t *kA) {
  const int64_t NS = (int64_t)p->snd.nbr.size(), NR = (int64_t)p->rcv.nbr.size();
  *aC = 0; *aA = NS; *kC = NS + NR; *kA = NS + 2 * NR;
  return (int)(2 * NS + 2 * NR);
}

// The regions come out of a POOL of big chunks that a process exports ONCE each and never frees, and a process maps a peer's chunk
// ONCE and keeps it mapped: a plan per PVector cache means hundreds of short-lived regions in a solver, and exporting / importing /
// closing / freeing each of them made hipIpcGetMemHandle fail with "invalid argument" after a few dozen (an address handed out
// again while a peer still held the old mapping).  A freed region goes to a free list of its size and is zeroed when taken again.
struct ipc_chunk { char *base = nullptr; size_t size = 0, used = 0; hipIpcMemHandle_t handle; int device = 0; bool finegrained = false; };
static std::mutex g_ipc_mu;
static std::vector<ipc_chunk> g_ipc_chunks;
static std::multimap<size_t, std::pair<int, size_t>> g_ipc_free;        // bytes -> (chunk, offset)
struct handle_key { char b[sizeof(hipIpcMemHandle_t)]; bool operator<(const handle_key &o) const { return memcmp(b, o.b, sizeof b) < 0; } };
static std::map<handle_key, void *> g_ipc_open;                          // a peer's chunk -> where it is mapped here

static int ipc_region_take(int device, size_t bytes, int *chunk, size_t *off) {
  std::lock_guard<std::mutex> lk(g_ipc_mu);
  for (auto it = g_ipc_free.lower_bound(bytes); it != g_ipc_free.end() && it->first == bytes; ++it)
    if (g_ipc_chunks[it->second.first].device == device) { *chunk = it->second.first; *off = it->second.second; g_ipc_free.erase(it); return PA_OK; }
  for (size_t k = 0; k < g_ipc_chunks.size(); ++k) {
    ipc_chunk &C = g_ipc_chunks[k];
    if (C.device == device && C.used + bytes <= C.size) { *chunk = (int)k; *off = C.used; C.used += bytes; return PA_OK; }
  }
  ipc_chunk C;
  C.device = device;
  C.size = std::max<size_t>((size_t)32 << 20, bytes);
  // Fine-grained device memory where the runtime gives it (what RCCL takes for its own buffers): stores that come over xGMI from a
  // peer GPU are then visible to a kernel that is ALREADY RUNNING here and acquires at system scope -- the fused product launch
  // polls its arrival flags and reads the receive buffer without a kernel boundary in between (pa_fused.hip).  Coarse-grained
  // memory only promises that at kernel boundaries.  PA_IPC_FINEGRAINED=0, or a runtime that refuses: plain hipMalloc.
  const char *fg = getenv("PA_IPC_FINEGRAINED");
  C.base = nullptr;
  if (!(fg && atoi(fg) == 0)) {
    if (hipExtMallocWithFlags((void **)&C.base, C.size, hipDeviceMallocFinegrained) != hipSuccess ||
        hipIpcGetMemHandle(&C.handle, C.base) != hipSuccess) {
      (void)hipGetLastError();
      if (C.base) (void)hipFree(C.base);
      C.base = nullptr;
    } else C.finegrained = true;
  }
  if (!C.base) {
    PA_HIP(hipMalloc((void **)&C.base, C.size));
    PA_HIP(hipIpcGetMemHandle(&C.handle, C.base));
  }
  C.used = bytes;
  g_ipc_chunks.push_back(C);
  *chunk = (int)g_ipc_chunks.size() - 1; *off = 0;
  return PA_OK;
}
static void ipc_region_give_back(int chunk, size_t off, size_t bytes) {
  std::lock_guard<std::mutex> lk(g_ipc_mu);
  g_ipc_free.insert({bytes, {chunk, off}});
}
static int ipc_open_chunk(const hipIpcMemHandle_t &h, void **base) {
  std::lock_guard<std::mutex> lk(g_ipc_mu);
  handle_key k;
  memcpy(k.b, &h, sizeof k.b);
  auto it = g_ipc_open.find(k);
  if (it != g_ipc_open.end()) { *base = it->second; return PA_OK; }
  PA_HIP(hipIpcOpenMemHandle(base, h, hipIpcMemLazyEnablePeerAccess));
  g_ipc_open[k] = *base;
  return PA_OK;
}

static int ipc_prepare(pa_plan *p) {
  if (p->ipc) return PA_OK;
  PA_REQUIRE(p->phase == 0, "an exchange is in flight on this plan: its buffers cannot move into an ipc region now");
  PA_HIP(hipSetDevice(p->ctx->device));
  PA_HIP(hipStreamSynchronize(p->ctx->s[0]));
  PA_HIP(hipStreamSynchronize(p->ctx->s[1]));
  pa_ipc_link *L = new pa_ipc_link();
  // everything that can fail happens BEFORE the plan is touched: a failure leaves the plan on its own buffers and nothing behind
  auto fail = [&](int st) {
    if (L->d_done) (void)hipFree(L->d_done);
    if (L->h_status) (void)hipHostFree(L->h_status);
    if (L->chunk >= 0) ipc_region_give_back(L->chunk, L->chunk_off, L->region_bytes);
    delete L;
    (void)hipGetLastError();
    return st;
  };
  int64_t a, b, c, d;
  L->n_flags = std::max(1, ipc_flags_layout(p, &a, &b, &c, &d));
  auto up = [](size_t x) { return (x + 255) / 256 * 256; };
  L->off_snd = 0;
  L->off_rcv = (int64_t)up(sizeof(double) * std::max<int64_t>(1, p->snd.n));
  L->off_flags = L->off_rcv + (int64_t)up(sizeof(double) * std::max<int64_t>(1, p->rcv.n));
  L->region_bytes = ((size_t)L->off_flags + up(sizeof(unsigned long long) * L->n_flags) + 4095) / 4096 * 4096;
  if (int st = ipc_region_take(p->ctx->device, L->region_bytes, &L->chunk, &L->chunk_off)) { L->chunk = -1; return fail(st); }
  { std::lock_guard<std::mutex> lk(g_ipc_mu); L->d_region = g_ipc_chunks[L->chunk].base + L->chunk_off; }
  if (hipMemset(L->d_region, 0, L->region_bytes) != hipSuccess || hipMalloc((void **)&L->d_done, 2 * sizeof(unsigned)) != hipSuccess ||
      hipMemset(L->d_done, 0, 2 * sizeof(unsigned)) != hipSuccess ||
      hipHostMalloc((void **)&L->h_status, sizeof(int), hipHostMallocMapped) != hipSuccess || hipDeviceSynchronize() != hipSuccess) {
    pa_set_err("ipc link: device memory for the region's bookkeeping could not be set up");
    return fail(PA_ERR_HIP);
  }
  *L->h_status = 0;
  // the plan's buffers move into the region (they hold nothing between two exchanges); tables that cached the old addresses go
  for (int m = 0; m < 2; ++m) if (p->push[m]) { p->push[m]->free_all(); delete p->push[m]; p->push[m] = nullptr; }
  (void)pa_raw_free(p->snd.d_buf);
  (void)pa_raw_free(p->rcv.d_buf);
  p->snd.d_buf = (double *)(L->d_region + L->off_snd);
  p->rcv.d_buf = (double *)(L->d_region + L->off_rcv);
  p->bufs_in_ipc_region = true;
  L->d_flags = (unsigned long long *)(L->d_region + L->off_flags);
  double secs = 30.0;
  if (const char *e = getenv("PA_IPC_TIMEOUT_S")) secs = std::max(0.001, atof(e));
  L->ticks = (long long)(secs * 1e8);                  // wall_clock64: 100 MHz
  p->ipc = L;
  return PA_OK;
}

extern "C" int pa_plan_ipc_blob_size(pa_plan *p, int64_t *bytes) {
  PA_REQUIRE(p && bytes, "bad arguments");
  *bytes = (int64_t)sizeof(ipc_header) + (int64_t)sizeof(int32_t) * (int64_t)(2 * p->snd.nbr.size() + 2 * p->rcv.nbr.size() + 2);
  return PA_OK;
}

// What a neighbour needs to push into this part's buffers: ipc handles of the two receive buffers and of the flag words, the
// neighbour lists and slice offsets of both sides.  Opaque bytes; the host language carries them to the neighbours (or to
// everybody) the way it carries the RCCL unique id.
extern "C" int pa_plan_ipc_blob(pa_plan *p, void *out, int64_t capacity) {
  PA_REQUIRE(p && out, "bad arguments");
  int64_t need = 0;
  PA_TRY(pa_plan_ipc_blob_size(p, &need));
  PA_REQUIRE(capacity >= need, "the blob needs %lld bytes", (long long)need);
  PA_TRY(ipc_prepare(p));
  ipc_header h;
  memset(&h, 0, sizeof h);
  h.magic = PA_IPC_MAGIC; h.part = p->part;
  h.n_snd_nbr = (int32_t)p->snd.nbr.size(); h.n_rcv_nbr = (int32_t)p->rcv.nbr.size();
  h.n_snd = p->snd.n; h.n_rcv = p->rcv.n;
  h.off_snd = (int64_t)p->ipc->chunk_off + p->ipc->off_snd; h.off_rcv = (int64_t)p->ipc->chunk_off + p->ipc->off_rcv;
  h.off_flags = (int64_t)p->ipc->chunk_off + p->ipc->off_flags;
  { std::lock_guard<std::mutex> lk(g_ipc_mu); h.h_region = g_ipc_chunks[p->ipc->chunk].handle; }
  char *q = (char *)out;
  memcpy(q, &h, sizeof h); q += sizeof h;
  auto put = [&](const std::vector<int32_t> &v) { if (!v.empty()) memcpy(q, v.data(), sizeof(int32_t) * v.size()); q += sizeof(int32_t) * v.size(); };
  put(p->snd.nbr); put(p->snd.ptrs); put(p->rcv.nbr); put(p->rcv.ptrs);
  return PA_OK;
}

struct peer_view {
  ipc_header h;
  std::vector<int32_t> snd_nbr, snd_ptrs, rcv_nbr, rcv_ptrs;
};

static int parse_blob(const void *blob, int64_t bytes, peer_view &V) {
  PA_REQUIRE(blob && bytes >= (int64_t)sizeof(ipc_header), "a blob is too short");
  const char *q = (const char *)blob;
  memcpy(&V.h, q, sizeof V.h); q += sizeof V.h;
  PA_REQUIRE(V.h.magic == PA_IPC_MAGIC, "not a pa_plan_ipc_blob");
  PA_REQUIRE(V.h.n_snd_nbr >= 0 && V.h.n_rcv_nbr >= 0 &&
             bytes >= (int64_t)sizeof(ipc_header) + (int64_t)sizeof(int32_t) * (2 * (int64_t)V.h.n_snd_nbr + 2 * (int64_t)V.h.n_rcv_nbr + 2), "a blob is truncated");
  auto get = [&](std::vector<int32_t> &v, size_t n) { v.resize(n); if (n) memcpy(v.data(), q, sizeof(int32_t) * n); q += sizeof(int32_t) * n; };
  get(V.snd_nbr, V.h.n_snd_nbr); get(V.snd_ptrs, V.h.n_snd_nbr + 1); get(V.rcv_nbr, V.h.n_rcv_nbr); get(V.rcv_ptrs, V.h.n_rcv_nbr + 1);
  return PA_OK;
}

// blobs[k] (sizes[k] bytes): the pa_plan_ipc_blob of some part (any order, this part's own and strangers' are ignored).  Opens the
// neighbours' buffers and builds the push tables of both modes.  Collective in spirit: every neighbour must do the same before
// the first pa_exchange_push_ipc.
extern "C" int pa_plan_ipc_connect(pa_plan *p, int32_t n_blobs, const void *const *blobs, const int64_t *sizes) {
  PA_REQUIRE(p && (n_blobs == 0 || (blobs && sizes)) && n_blobs >= 0, "bad arguments");
  PA_TRY(ipc_prepare(p));
  pa_ipc_link *L = p->ipc;
  PA_REQUIRE(!L->connected, "this plan is connected already");
  PA_HIP(hipSetDevice(p->ctx->device));
  std::map<int, peer_view> views;
  for (int k = 0; k < n_blobs; ++k) {
    peer_view V;
    PA_TRY(parse_blob(blobs[k], sizes[k], V));
    // (its own blob matters only to a part that is its own neighbour: a periodic direction with one part -- the link then points
    //  at this process's own region, no handle is opened)
    const bool nb = std::find(p->snd.nbr.begin(), p->snd.nbr.end(), V.h.part) != p->snd.nbr.end() ||
                    std::find(p->rcv.nbr.begin(), p->rcv.nbr.end(), V.h.part) != p->rcv.nbr.end();
    if (nb) views[V.h.part] = V;
  }
  auto open_peer = [&](int q) -> int {
    if (L->peers.count(q)) return PA_OK;
    auto it = views.find(q);
    PA_REQUIRE(it != views.end(), "no blob of neighbour part %d", q);
    pa_ipc_link::peer P;
    const ipc_header &h = it->second.h;
    if (q == p->part) { std::lock_guard<std::mutex> lk(g_ipc_mu); P.region = g_ipc_chunks[L->chunk].base; }
    else PA_TRY(ipc_open_chunk(h.h_region, &P.region));
    P.snd = (char *)P.region + h.off_snd;
    P.rcv = (char *)P.region + h.off_rcv;
    P.flags = (char *)P.region + h.off_flags;
    L->peers[q] = P;
    return PA_OK;
  };
  int64_t aC, aA, kC, kA;
  ipc_flags_layout(p, &aC, &aA, &kC, &kA);
  for (int mode = 0; mode < 2; ++mode) {
    pa_plan::side &o = out_side(p, mode), &in = in_side(p, mode);
    const int64_t my_ack0 = mode == PA_CONSISTENT ? kC : kA, my_arr0 = mode == PA_CONSISTENT ? aC : aA;
    std::vector<pa_push_seg> segs;
    for (size_t j = 0; j < o.nbr.size(); ++j) {
      const int len = o.ptrs[j + 1] - o.ptrs[j];
      if (!len) continue;
      const int q = o.nbr[j];
      PA_TRY(open_peer(q));
      const peer_view &V = views[q];
      // the receiver's in-side of this mode: its snd side for consistent!, its rcv side for assemble!
      const std::vector<int32_t> &qn = mode == PA_CONSISTENT ? V.snd_nbr : V.rcv_nbr, &qp = mode == PA_CONSISTENT ? V.snd_ptrs : V.rcv_ptrs;
      auto it = std::find(qn.begin(), qn.end(), p->part);
      PA_REQUIRE(it != qn.end(), "inconsistent ExchangeGraph: part %d sends to %d, which does not receive from it", p->part, q);
      const size_t i = it - qn.begin();
      PA_REQUIRE(qp[i + 1] - qp[i] == len, "slice length mismatch between parts %d and %d", p->part, q);
      const int64_t qNS = V.h.n_snd_nbr;
      const int64_t q_arr0 = mode == PA_CONSISTENT ? 0 : qNS;
      pa_push_seg S;
      S.uidx = nullptr; S.upart = -1;
      S.start = o.ptrs[j]; S.len = len;
      S.dst = (double *)(mode == PA_CONSISTENT ? L->peers[q].snd : L->peers[q].rcv) + qp[i];
      S.arrive = (unsigned long long *)L->peers[q].flags + q_arr0 + (int64_t)i;
      S.ack = L->d_flags + my_ack0 + (int64_t)j;
      segs.push_back(S);
      L->ack_idx[mode].push_back(my_ack0 + (int64_t)j);
    }
    std::vector<int32_t> wait;
    std::vector<unsigned long long *> ackdst;
    for (size_t i = 0; i < in.nbr.size(); ++i) {
      const int len = in.ptrs[i + 1] - in.ptrs[i];
      if (!len) continue;
      const int q = in.nbr[i];
      PA_TRY(open_peer(q));
      const peer_view &V = views[q];
      // the sender's out-side of this mode: its rcv side for consistent!, its snd side for assemble!
      const std::vector<int32_t> &qn = mode == PA_CONSISTENT ? V.rcv_nbr : V.snd_nbr, &qp = mode == PA_CONSISTENT ? V.rcv_ptrs : V.snd_ptrs;
      auto it = std::find(qn.begin(), qn.end(), p->part);
      PA_REQUIRE(it != qn.end(), "inconsistent ExchangeGraph: part %d receives from %d, which does not send to it", p->part, q);
      const size_t j = it - qn.begin();
      PA_REQUIRE(qp[j + 1] - qp[j] == len, "slice length mismatch between parts %d and %d", q, p->part);
      const int64_t qNS = V.h.n_snd_nbr, qNR = V.h.n_rcv_nbr;
      const int64_t q_ack0 = mode == PA_CONSISTENT ? qNS + qNR : qNS + 2 * qNR;
      wait.push_back((int32_t)(my_arr0 + (int64_t)i));
      ackdst.push_back((unsigned long long *)L->peers[q].flags + q_ack0 + (int64_t)j);
    }
    pa_ipc_link::per_mode &M = L->m[mode];
    M.nseg = (int)segs.size(); M.n_wait = (int)wait.size(); M.n_ack = (int)ackdst.size();
    PA_TRY(upload_vec(segs, &M.d_segs));
    PA_TRY(upload_vec(wait, &M.d_wait));
    PA_TRY(upload_vec(ackdst, &M.d_ack_dst));
  }
  L->connected = true;
  return PA_OK;
}

bool pa_plan_ipc_connected(const pa_plan *p) { return p && p->ipc && p->ipc->connected; }

extern "C" int pa_plan_ipc_status(pa_plan *p, int *status) {
  PA_REQUIRE(p && status, "bad arguments");
  *status = (p->ipc && p->ipc->h_status) ? *(volatile int *)p->ipc->h_status : 0;
  return PA_OK;
}

// pack + exchange! of this process's part over the ipc link (the one-part-per-process twin of pa_exchange_push_local)
extern "C" int pa_exchange_push_ipc(pa_plan *p, const pa_vec *v, int mode) {
  PA_REQUIRE(p && v && (mode == PA_ASSEMBLE || mode == PA_CONSISTENT), "bad arguments");
  PA_REQUIRE(p->ipc && p->ipc->connected, "the plan has no ipc link (pa_plan_ipc_connect)");
  PA_REQUIRE(v->n_own + v->n_ghost == p->n_local, "vector has %lld local values, plan expects %lld", (long long)(v->n_own + v->n_ghost),
             (long long)p->n_local);
  PA_REQUIRE(p->phase == 0, "exchange already in flight on this plan (missing pa_exchange_finish)");
  pa_ipc_link *L = p->ipc;
  const int st = *(volatile int *)L->h_status;
  if (st != 0) { pa_set_err("the ipc link of part %d timed out earlier (%s wait): a neighbour is gone or out of step", p->part, st == 1 ? "arrival" : "acknowledgement"); return PA_ERR_STATE; }
  pa_ctx *c = p->ctx;
  PA_REQUIRE(!c->capturing, "the ipc transport carries a sequence number per exchange: not inside a graph capture");
  pa_ipc_link::per_mode &M = L->m[mode];
  p->mode = mode;
  if (p->snd.n == 0 && p->rcv.n == 0) { p->phase = 1; return PA_OK; }
  PA_HIP(hipSetDevice(c->device));
  const unsigned long long seq = ++p->seq[mode];
  PA_HIP(hipEventRecord(c->ev_compute, c->s[0]));
  PA_HIP(hipStreamWaitEvent(c->s[1], c->ev_compute, 0));
  pa_plan::side &o = out_side(p, mode);
  if (M.nseg && o.n) {
    hipLaunchKernelGGL(k_push_ipc, dim3((unsigned)((o.n + 255) / 256)), dim3(256), 0, c->s[1], o.d_idx, (int)o.n, M.d_segs, M.nseg, v->d, seq,
                       L->d_done, L->ticks, L->h_status);
  }
  if (M.n_wait) hipLaunchKernelGGL(k_wait_flags, dim3(1), dim3(64), 0, c->s[1], L->d_flags, M.d_wait, M.n_wait, seq, L->ticks, L->h_status);
  PA_HIP(hipGetLastError());
  p->own_comm_stream = true;
  p->ipc_ack_due = M.n_ack > 0;
  p->ev_wait = nullptr;
  return pa_plan_mark_arrived(p);
}

// mul!(c,a,b) of this process's part as ONE launch on the compute stream: its first blocks push b's own values into the neighbours'
// receive buffers, its tail acquires the neighbours' arrival flags, sums the boundary rows from the receive buffer, unpacks b's
// ghost entries and acknowledges (pa_fused.hip).  The plan is idle again when this returns: consistent!(b) is part of the launch.
int pa_mul_fused_ipc(pa_matrix *m, pa_vec *c, pa_vec *b, double alpha, double beta) {
  pa_plan *p = m->plan;
  PA_REQUIRE(p->ipc && p->ipc->connected, "the plan has no ipc link (pa_plan_ipc_connect)");
  PA_REQUIRE(b->n_own + b->n_ghost == p->n_local, "vector has %lld local values, plan expects %lld", (long long)(b->n_own + b->n_ghost),
             (long long)p->n_local);
  PA_REQUIRE(p->phase == 0, "exchange already in flight on this plan (missing pa_exchange_finish)");
  pa_ipc_link *L = p->ipc;
  const int st = *(volatile int *)L->h_status;
  if (st != 0) { pa_set_err("the ipc link of part %d timed out earlier (%s wait): a neighbour is gone or out of step", p->part, st == 1 ? "arrival" : "acknowledgement"); return PA_ERR_STATE; }
  pa_ctx *cx = p->ctx;
  PA_REQUIRE(!cx->capturing, "the ipc transport carries a sequence number per exchange: not inside a graph capture");
  pa_ipc_link::per_mode &M = L->m[PA_CONSISTENT];
  PA_HIP(hipSetDevice(cx->device));
  pa_plan::side &o = p->rcv, &in = p->snd;             // consistent!: the cache reversed (src/p_vector.jl:747-755)
  int n_push_blocks = 0;
  if (M.nseg && o.n) n_push_blocks = (int)((o.n + 255) / 256);
  if (!L->d_xcomm) {                                   // what the launch does for the exchange: static per link, kept in device memory
    pa_fused_comm X;
    if (n_push_blocks) { X.p_idx = o.d_idx; X.p_n = (int)o.n; X.p_segs = M.d_segs; X.p_nseg = M.nseg; X.p_done = L->d_done; }
    X.flags = L->d_flags; X.wait_idx = M.d_wait; X.n_wait = M.n_wait; X.ticks = L->ticks; X.status = L->h_status;
    X.u_idx = in.d_idx; X.u_n = (int)in.n;
    X.ack_dst = M.d_ack_dst; X.n_ack = M.n_ack; X.t_done = L->d_done + 1;
    PA_HIP(pa_raw_malloc(&L->d_xcomm, sizeof X));
    PA_HIP(pa_h2d(L->d_xcomm, &X, sizeof X));
  }
  const unsigned long long seq = ++p->seq[PA_CONSISTENT];
  p->mode = PA_CONSISTENT;
  return pa_mul_fused_launch(m, c, b, alpha, beta, cx->s[0], L->d_xcomm, seq, n_push_blocks, cx->sw.fused_tail_blocks);
}

bool pa_fused_ipc_fits(const pa_matrix *m) {
  const pa_plan *p = m->plan;
  const int64_t groups = ((m->oo->n_crows + 63) / 64 + 3) / 4;          // (the main blocks of the pattern-ELL form: four slabs each)
  const int64_t main_blocks = std::min<int64_t>((m->oo->n_chunks + 7) / 8 * 8, (groups + 7) / 8 * 8);
  return (int64_t)((p->rcv.n + 255) / 256) <= main_blocks;
}

// compute stream, behind the unpack (and whatever read the receive buffer): the senders may overwrite it now
int pa_ipc_ack(pa_plan *p, int mode) {
  if (!p->ipc || !p->ipc_ack_due) return PA_OK;
  p->ipc_ack_due = false;
  pa_ipc_link::per_mode &M = p->ipc->m[mode];
  if (M.n_ack == 0) return PA_OK;
  hipLaunchKernelGGL(k_write_flags, dim3((M.n_ack + 63) / 64), dim3(64), 0, p->ctx->s[0], M.d_ack_dst, M.n_ack, p->seq[mode]);
  PA_HIP(hipGetLastError());
  return PA_OK;
}

void pa_push_release(pa_plan *p) {
  for (int m = 0; m < 2; ++m)
    if (p->push[m]) { p->push[m]->free_all(); delete p->push[m]; p->push[m] = nullptr; }
  if (pa_ipc_link *L = p->ipc) {
    (void)hipSetDevice(p->ctx->device);
    // (the peers' chunks stay mapped for the life of the process: ipc_open_chunk)
    for (int m = 0; m < 2; ++m) {
      (void)pa_raw_free(L->m[m].d_segs); (void)pa_raw_free(L->m[m].d_wait); (void)pa_raw_free(L->m[m].d_ack_dst);
    }
    // The region goes back to the pool only when nobody can write into it any more.  What may still be ON ITS WAY when this plan goes
    // is the receivers' acknowledgement of the last slices this part pushed (they acknowledge behind their own unpack, on their own
    // stream, in their own time): a region handed out again and zeroed would take that late store for a flag of the NEW plan, whose
    // sequence numbers start over.  So: wait, bounded, until every acknowledgement of the last exchange of either mode
    // is in; a region whose acknowledgements never come is not reused.
    bool quiet = true;
    if (L->connected && L->chunk >= 0) {
      for (int m = 0; m < 2 && quiet; ++m) {
        const unsigned long long want = p->seq[m];
        if (want == 0 || L->ack_idx[m].empty()) continue;
        std::vector<unsigned long long> flags((size_t)L->n_flags);
        const auto t0 = std::chrono::steady_clock::now();
        for (;;) {
          if (hipMemcpy(flags.data(), L->d_flags, sizeof(unsigned long long) * flags.size(), hipMemcpyDeviceToHost) != hipSuccess) { (void)hipGetLastError(); quiet = false; break; }
          bool all = true;
          for (int64_t k : L->ack_idx[m]) all = all && flags[(size_t)k] >= want;
          if (all) break;
          if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) { quiet = false; break; }
        }
      }
    }
    if (L->chunk >= 0 && quiet) ipc_region_give_back(L->chunk, L->chunk_off, L->region_bytes);   // (the plan's two buffers live in it)
    if (L->d_done) (void)hipFree(L->d_done);
    if (L->h_status) (void)hipHostFree(L->h_status);
    if (L->d_xcomm) (void)pa_raw_free(L->d_xcomm);
    delete L;
    p->ipc = nullptr;
  }
}
