// pa_plan.hip -- exchange plans (consistent! / assemble! / exchange!) and the operator level: mul!(c,a,b), mul! + dot
// (one of the units pa_device.hip was split into in round 5; compiled with -ffp-contract=off like all of them)
#include <hip/hip_runtime.h>

#include <atomic>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <thread>
#include <cstring>
#include <memory>
#include <numeric>
#include <iterator>
#include <string>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "pa_internal.h"
#include "pa_setup.h"
#include "pa_dev_kernels.h"

// ------------------------------------------------------------------------------------------------
// exchange plans
// ------------------------------------------------------------------------------------------------

extern "C" int pa_plan_create(pa_ctx *c, int32_t part, int64_t n_local, int32_t n_snd, const int32_t *nbr_snd,
                              const int32_t *ptrs_snd, const int32_t *idx_snd, int32_t n_rcv, const int32_t *nbr_rcv,
                              const int32_t *ptrs_rcv, const int32_t *idx_rcv, int index_base, pa_plan **out) {
  PA_REQUIRE(c && out && ptrs_snd && ptrs_rcv, "bad arguments");
  PA_REQUIRE(index_base == 0 || index_base == 1, "index_base must be 0 or 1");
  PA_REQUIRE(n_snd >= 0 && n_rcv >= 0 && n_local >= 0, "negative size");
  PA_REQUIRE((n_snd == 0 || nbr_snd) && (n_rcv == 0 || nbr_rcv), "neighbour arrays are NULL");
  pa_plan *p = new pa_plan();
  static std::atomic<uint64_t> next_serial{1};
  p->serial = next_serial++;
  p->ctx = c; p->part = part - index_base; p->n_local = n_local;
  auto side = [&](pa_plan::side &s, int32_t n, const int32_t *nbr, const int32_t *ptrs, const int32_t *idx) -> int {
    s.nbr.assign(nbr, nbr + n);
    for (auto &q : s.nbr) q -= index_base;
    for (int i = 0; i < n; ++i) {
      PA_REQUIRE(s.nbr[i] >= 0, "neighbour %d is not a part id (%d with index base %d)", i + 1, s.nbr[i] + index_base, index_base);
    }
    s.ptrs.resize(n + 1);
    for (int i = 0; i <= n; ++i) s.ptrs[i] = ptrs[i] - index_base;
    PA_REQUIRE(s.ptrs[0] == 0, "ptrs[1] must be the index base");
    for (int i = 0; i < n; ++i) PA_REQUIRE(s.ptrs[i + 1] >= s.ptrs[i], "ptrs not monotone");
    s.n = s.ptrs[n];
    PA_REQUIRE(s.n == 0 || idx, "index array is NULL");
    s.idx.resize(s.n);
    for (int64_t k = 0; k < s.n; ++k) {
      s.idx[k] = idx[k] - index_base;
      PA_REQUIRE(s.idx[k] >= 0 && s.idx[k] < n_local, "local index out of range at position %lld", (long long)k);
    }
    return PA_OK;
  };
  PA_TRY(side(p->snd, n_snd, nbr_snd, ptrs_snd, idx_snd));
  PA_TRY(side(p->rcv, n_rcv, nbr_rcv, ptrs_rcv, idx_rcv));
  PA_HIP(hipSetDevice(c->device));
  for (pa_plan::side *s : {&p->snd, &p->rcv}) {
    PA_TRY(upload_i32(s->idx, &s->d_idx));
    PA_HIP(pa_raw_malloc(&s->d_buf, sizeof(double) * std::max<int64_t>(1, s->n)));
    PA_HIP(hipMemsetAsync(s->d_buf, 0, sizeof(double) * std::max<int64_t>(1, s->n), c->s[1]));   // (the stream the pack kernel writes it on)
    PA_HIP(hipStreamSynchronize(c->s[1]));
  }
  // inverse map of the rcv side for the deterministic assemble!(+): target lid -> its p's, ascending
  {
    std::vector<int32_t> order(p->rcv.n);
    std::iota(order.begin(), order.end(), 0);
    std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return p->rcv.idx[a] < p->rcv.idx[b]; });
    std::vector<int32_t> tgt, tptr;
    tptr.push_back(0);
    for (int64_t k = 0; k < p->rcv.n; ++k) {
      if (k == 0 || p->rcv.idx[order[k]] != p->rcv.idx[order[k - 1]]) {
        if (k) tptr.push_back((int32_t)k);
        tgt.push_back(p->rcv.idx[order[k]]);
      }
    }
    if (p->rcv.n) tptr.push_back((int32_t)p->rcv.n);
    p->n_tgt = (int64_t)tgt.size();
    PA_TRY(upload_i32(tgt, &p->d_tgt));
    PA_TRY(upload_i32(tptr, &p->d_tptr));
    PA_TRY(upload_i32(order, &p->d_tp));
  }
  PA_HIP(hipEventCreateWithFlags(&p->ev_packed, hipEventDisableTiming));
  PA_HIP(hipEventCreateWithFlags(&p->ev_arrived, hipEventDisableTiming));
  PA_HIP(hipStreamSynchronize(nullptr));  // buffers were zeroed on the default stream; the ctx streams do not wait for it
  *out = p;
  return PA_OK;
}

extern "C" int pa_plan_destroy(pa_plan *p) {
  if (!p) return PA_OK;
  (void)hipSetDevice(p->ctx->device);
  (void)hipStreamSynchronize(p->ctx->s[0]);
  (void)hipStreamSynchronize(p->ctx->s[1]);
  pa_push_release(p);
  pa_fused_plan_release(p);
  for (pa_plan::side *s : {&p->snd, &p->rcv}) {
    (void)pa_raw_free(s->d_idx);
    if (!p->bufs_in_ipc_region) (void)pa_raw_free(s->d_buf);
  }
  (void)pa_raw_free(p->d_tgt);
  (void)pa_raw_free(p->d_tptr);
  (void)pa_raw_free(p->d_tp);
  (void)hipEventDestroy(p->ev_packed);
  (void)hipEventDestroy(p->ev_arrived);
  delete p;
  return PA_OK;
}

// Roles of the two sides per mode (reverse(cache), src/p_vector.jl:427-437,748):
//   PA_ASSEMBLE  : pack from snd side (ghost lids), receive into rcv side (own lids)
//   PA_CONSISTENT: pack from rcv side (own lids),   receive into snd side (ghost lids)
static inline pa_plan::side &out_side(pa_plan *p, int mode) { return mode == PA_ASSEMBLE ? p->snd : p->rcv; }
static inline pa_plan::side &in_side(pa_plan *p, int mode) { return mode == PA_ASSEMBLE ? p->rcv : p->snd; }

extern "C" int pa_plan_buffers(pa_plan *p, int mode, void **snd, int64_t *snd_len, void **rcv, int64_t *rcv_len) {
  PA_REQUIRE(p && (mode == PA_ASSEMBLE || mode == PA_CONSISTENT), "bad arguments");
  if (snd) *snd = out_side(p, mode).d_buf;
  if (snd_len) *snd_len = out_side(p, mode).n;
  if (rcv) *rcv = in_side(p, mode).d_buf;
  if (rcv_len) *rcv_len = in_side(p, mode).n;
  return PA_OK;
}

extern "C" int pa_exchange_pack(pa_plan *p, const pa_vec *v, int mode) {
  PA_REQUIRE(p && v && (mode == PA_ASSEMBLE || mode == PA_CONSISTENT), "bad arguments");
  PA_REQUIRE(v->n_own + v->n_ghost == p->n_local, "vector has %lld local values, plan expects %lld",
             (long long)(v->n_own + v->n_ghost), (long long)p->n_local);
  PA_REQUIRE(p->phase == 0, "exchange already in flight on this plan (missing pa_exchange_finish)");
  pa_ctx *c = p->ctx;
  p->ev_wait = nullptr;
  p->elem = 8;
  if (p->snd.n == 0 && p->rcv.n == 0) {  // a part without neighbours (e.g. the only part): nothing to move, no stream traffic
    p->phase = 1;
    p->mode = mode;
    return PA_OK;
  }
  PA_HIP(hipSetDevice(c->device));
  // the comm stream must see everything the compute stream wrote into v so far
  PA_HIP(hipEventRecord(c->ev_compute, c->s[0]));
  PA_HIP(hipStreamWaitEvent(c->s[1], c->ev_compute, 0));
  pa_plan::side &o = out_side(p, mode);
  if (o.n) hipLaunchKernelGGL(k_pack, dim3((o.n + 255) / 256), dim3(256), 0, c->s[1], o.d_buf, v->d, o.d_idx, (int)o.n);
  PA_HIP(hipGetLastError());
  PA_HIP(hipEventRecord(p->ev_packed, c->s[1]));
  p->phase = 1;
  p->mode = mode;
  return PA_OK;
}

extern "C" int pa_exchange_local(pa_plan *const *plans, int32_t n_parts, int mode) {
  PA_REQUIRE(plans && n_parts > 0 && (mode == PA_ASSEMBLE || mode == PA_CONSISTENT), "bad arguments");
  for (int r = 0; r < n_parts; ++r) {
    PA_REQUIRE(plans[r] && plans[r]->part == r, "plans[%d] is not the plan of part %d", r, r);
    PA_REQUIRE(plans[r]->phase == 1 && plans[r]->mode == mode, "part %d: pa_exchange_pack(mode) must come first", r);
    // (measured on ROCm 7.0: hipStreamEndCapture recurses without end -- a segfault -- over this transport's comm-stream waits)
    PA_REQUIRE(!(plans[r]->ctx->capturing && n_parts > 1), "the copy transport is not capturable into a hipGraph: use pa_exchange_push_local");
  }
  // src/primitives.jl:1020-1042: rcv[r].data[ptrs_rcv[i]..] = snd[s].data[ptrs_snd[j]..], snd_ids[s][j] == r
  for (int r = 0; r < n_parts; ++r) {
    pa_plan *pr = plans[r];
    pa_plan::side &in = in_side(pr, mode);
    PA_HIP(hipSetDevice(pr->ctx->device));
    for (size_t i = 0; i < in.nbr.size(); ++i) {
      const int s = in.nbr[i];
      PA_REQUIRE(s >= 0 && s < n_parts, "part %d: neighbour %d out of range", r, s);
      pa_plan *ps = plans[s];
      pa_plan::side &o = out_side(ps, mode);
      auto it = std::find(o.nbr.begin(), o.nbr.end(), r);
      PA_REQUIRE(it != o.nbr.end(), "inconsistent ExchangeGraph: part %d receives from %d, which does not send to it", r, s);
      const size_t j = it - o.nbr.begin();
      const int64_t len = in.ptrs[i + 1] - in.ptrs[i];
      PA_REQUIRE(len == o.ptrs[j + 1] - o.ptrs[j], "slice length mismatch between parts %d and %d", s, r);
      PA_HIP(hipStreamWaitEvent(pr->ctx->s[1], ps->ev_packed, 0));
      PA_REQUIRE(pr->elem == ps->elem, "parts %d and %d packed payloads of different element types", s, r);
      const size_t eb = (size_t)pr->elem;     // (8; 4 for a Float32 payload, pa_exchange_pack32: the buffers hold floats at the same element offsets)
      if (len)
        PA_HIP(hipMemcpyAsync(reinterpret_cast<char *>(in.d_buf) + eb * in.ptrs[i], reinterpret_cast<const char *>(o.d_buf) + eb * o.ptrs[j],
                              eb * len, hipMemcpyDeviceToDevice, pr->ctx->s[1]));
    }
    if (in.n || out_side(pr, mode).n) PA_HIP(hipEventRecord(pr->ev_arrived, pr->ctx->s[1]));
    pr->ev_wait = nullptr;
    pr->phase = 2;
  }
  return PA_OK;
}


// ---- other payload types (round 6, second widening): consistent! / assemble! of a PVector{Vector{T}}, T = Float32, Int32, Int64 ------
// assemble_impl! (src/p_vector.jl:587-612) is generic in the element type and exchange! in the payload (src/primitives.jl:1020-1042:
// Int64 ids at set-up, Float32 values, ...); the plan's index lists serve any element type, its buffers hold values of 4 or 8 bytes at
// the same element offsets.  pack_raw / pack32 -> one of the pack-then-transport transports (pa_exchange_local, pa_exchange_rccl(_all), or
// the caller's own copies between pa_plan_buffers) -> finish_raw / finish32.  The transports move bytes: integers travel as the float type
// of their width (ncclFloat / ncclDouble are never reduced here, only sent).
template <class T>
static __global__ void k_pack_t(T *__restrict__ buf, const T *__restrict__ v, const int *__restrict__ idx, int n) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) buf[p] = v[idx[p]];
}
template <class T>
static __global__ void k_unpack_insert_t(T *__restrict__ v, const T *__restrict__ buf, const int *__restrict__ idx, int n) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) v[idx[p]] = buf[p];
}
// one lane per distinct target; its contributions are added in ascending p (the reference's order), every sum rounded to T
template <class T>
static __global__ void k_unpack_add_t(T *__restrict__ v, const T *__restrict__ buf, const int *__restrict__ tgt, const int *__restrict__ tptr,
                                      const int *__restrict__ tp, int n_tgt) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n_tgt) {
    const int lid = tgt[k];
    T acc = v[lid];
    for (int j = tptr[k]; j < tptr[k + 1]; ++j) acc = acc + buf[tp[j]];
    v[lid] = acc;
  }
}
template <class T>
static __global__ void k_zero_t(T *__restrict__ v, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = (T)0;
}

// dtype of a raw payload: PA_DTYPE_F64 0, PA_DTYPE_F32 1, PA_DTYPE_I32 2, PA_DTYPE_I64 3 (include/pa_hip_experimental.h)
static inline int dtype_bytes(int dtype) { return dtype == 0 || dtype == 3 ? 8 : dtype == 1 || dtype == 2 ? 4 : 0; }

// values: n_local values of the dtype in HBM, device layout [own | ghost] (what a pa_vec / pa_vec32 holds; integers alike)
extern "C" int pa_exchange_pack_raw(pa_plan *p, const void *values, int64_t n_local, int dtype, int mode) {
  PA_REQUIRE(p && values && (mode == PA_ASSEMBLE || mode == PA_CONSISTENT), "bad arguments");
  PA_REQUIRE(dtype_bytes(dtype) != 0, "unknown dtype %d", dtype);
  PA_REQUIRE(n_local == p->n_local, "vector has %lld local values, plan expects %lld", (long long)n_local, (long long)p->n_local);
  PA_REQUIRE(p->phase == 0, "exchange already in flight on this plan (missing pa_exchange_finish)");
  pa_ctx *c = p->ctx;
  PA_REQUIRE(!c->capturing, "raw-payload exchanges are not recorded into graphs");
  p->ev_wait = nullptr;
  p->elem = dtype_bytes(dtype);
  p->raw_dtype = dtype;
  p->phase = 1;
  p->mode = mode;
  if (p->snd.n == 0 && p->rcv.n == 0) return PA_OK;
  PA_HIP(hipSetDevice(c->device));
  PA_HIP(hipEventRecord(c->ev_compute, c->s[0]));
  PA_HIP(hipStreamWaitEvent(c->s[1], c->ev_compute, 0));
  pa_plan::side &o = out_side(p, mode);
  if (o.n) {
    const dim3 g((unsigned)((o.n + 255) / 256));
    if (p->elem == 4)
      hipLaunchKernelGGL(k_pack_t<uint32_t>, g, dim3(256), 0, c->s[1], reinterpret_cast<uint32_t *>(o.d_buf), (const uint32_t *)values, (const int *)o.d_idx, (int)o.n);
    else
      hipLaunchKernelGGL(k_pack_t<uint64_t>, g, dim3(256), 0, c->s[1], reinterpret_cast<uint64_t *>(o.d_buf), (const uint64_t *)values, (const int *)o.d_idx, (int)o.n);
  }
  PA_HIP(hipGetLastError());
  PA_HIP(hipEventRecord(p->ev_packed, c->s[1]));
  return PA_OK;
}

template <class T>
static void finish_typed(pa_plan *p, T *v, int64_t n_own, int64_t n_ghost, int mode, bool none, hipStream_t st) {
  pa_plan::side &in = in_side(p, mode);
  if (mode == PA_CONSISTENT) {
    if (!none && in.n)
      hipLaunchKernelGGL(k_unpack_insert_t<T>, dim3((unsigned)((in.n + 255) / 256)), dim3(256), 0, st, v, reinterpret_cast<const T *>(in.d_buf),
                         (const int *)in.d_idx, (int)in.n);
  } else {
    if (!none && p->n_tgt)
      hipLaunchKernelGGL(k_unpack_add_t<T>, dim3((unsigned)((p->n_tgt + 255) / 256)), dim3(256), 0, st, v, reinterpret_cast<const T *>(in.d_buf),
                         (const int *)p->d_tgt, (const int *)p->d_tptr, (const int *)p->d_tp, (int)p->n_tgt);
    // fill!(ghost_values(a),0) (src/p_vector.jl:703-705): every ghost value, also the ones no message carries
    if (n_ghost > 0) hipLaunchKernelGGL(k_zero_t<T>, dim3((unsigned)((n_ghost + 255) / 256)), dim3(256), 0, st, v + n_own, (int64_t)n_ghost);
  }
}

// n_own: the own values come first in `values` (the ghosts behind them are what assemble! zeroes)
extern "C" int pa_exchange_finish_raw(pa_plan *p, void *values, int64_t n_own, int64_t n_local, int dtype, int mode) {
  PA_REQUIRE(p && values && (mode == PA_ASSEMBLE || mode == PA_CONSISTENT), "bad arguments");
  PA_REQUIRE(p->phase >= 1 && p->mode == mode, "pa_exchange_finish_raw without a matching pa_exchange_pack_raw");
  PA_REQUIRE(p->raw_dtype == dtype && p->elem == dtype_bytes(dtype),
             "the exchange in flight carries another payload type (%s): finish it as it was packed", p->raw_dtype < 0 ? "a pa_vec" : "another dtype");
  PA_REQUIRE(n_local == p->n_local && n_own >= 0 && n_own <= n_local, "vector/plan size mismatch");
  pa_ctx *c = p->ctx;
  PA_HIP(hipSetDevice(c->device));
  const bool none = p->snd.n == 0 && p->rcv.n == 0;
  if (!none) {
    if (p->phase == 1) PA_HIP(hipEventRecord(p->ev_arrived, c->s[1]));      // (caller-driven transport on the comm stream)
    PA_HIP(hipStreamWaitEvent(c->s[0], p->ev_wait ? p->ev_wait : p->ev_arrived, 0));  // wait(t)
  }
  p->ev_wait = nullptr;
  p->own_comm_stream = false;
  const int64_t n_ghost = n_local - n_own;
  switch (dtype) {
    case 0: finish_typed<double>(p, (double *)values, n_own, n_ghost, mode, none, c->s[0]); break;
    case 1: finish_typed<float>(p, (float *)values, n_own, n_ghost, mode, none, c->s[0]); break;
    case 2: finish_typed<int32_t>(p, (int32_t *)values, n_own, n_ghost, mode, none, c->s[0]); break;
    default: finish_typed<long long>(p, (long long *)values, n_own, n_ghost, mode, none, c->s[0]); break;
  }
  PA_HIP(hipGetLastError());
  // the next pack (on the comm stream) must not overwrite buffers this unpack still reads
  PA_HIP(hipEventRecord(c->ev_compute, c->s[0]));
  PA_HIP(hipStreamWaitEvent(c->s[1], c->ev_compute, 0));
  p->elem = 8;
  p->raw_dtype = -1;
  p->phase = 0;
  return PA_OK;
}

extern "C" int pa_exchange_pack32(pa_plan *p, const pa_vec32 *v, int mode) {
  PA_REQUIRE(p && v, "bad arguments");
  PA_REQUIRE(p->ctx == v->ctx, "plan and vector live on different contexts");
  return pa_exchange_pack_raw(p, v->d, v->n_own + v->n_ghost, 1, mode);
}

extern "C" int pa_exchange_finish32(pa_plan *p, pa_vec32 *v, int mode) {
  PA_REQUIRE(p && v, "bad arguments");
  PA_REQUIRE(p->phase < 1 || p->elem == 4, "the exchange in flight carries a Float64 payload: finish it with pa_exchange_finish");
  return pa_exchange_finish_raw(p, v->d, v->n_own, v->n_own + v->n_ghost, 1, mode);
}

int pa_plan_mark_arrived(pa_plan *p) {
  PA_HIP(hipEventRecord(p->ev_arrived, p->ctx->s[1]));
  p->ev_wait = nullptr;
  p->phase = 2;
  return PA_OK;
}

// the compute stream waits for the arrival of the exchange in flight, nothing else (no unpack): what a kernel that reads the
// RECEIVE BUFFER itself needs (own x ghost with renamed columns, pa_mul5)
static int exchange_wait_arrived(pa_plan *p) {
  if (p->snd.n == 0 && p->rcv.n == 0) return PA_OK;
  pa_ctx *c = p->ctx;
  if (p->phase == 1) PA_HIP(hipEventRecord(p->ev_arrived, c->s[1]));
  PA_HIP(hipStreamWaitEvent(c->s[0], p->ev_wait ? p->ev_wait : p->ev_arrived, 0));
  return PA_OK;
}

extern "C" int pa_exchange_finish(pa_plan *p, pa_vec *v, int mode) {
  PA_REQUIRE(p && v && (mode == PA_ASSEMBLE || mode == PA_CONSISTENT), "bad arguments");
  PA_REQUIRE(p->phase >= 1 && p->mode == mode, "pa_exchange_finish without a matching pa_exchange_pack");
  PA_REQUIRE(v->n_own + v->n_ghost == p->n_local, "vector/plan size mismatch");
  PA_REQUIRE(p->elem == 8 && p->raw_dtype < 0, "the exchange in flight carries a Float32 / raw payload (pa_exchange_pack32, pa_exchange_pack_raw): finish it with pa_exchange_finish32 / _finish_raw");
  pa_ctx *c = p->ctx;
  if (p->snd.n == 0 && p->rcv.n == 0) {                     // nothing travels; assemble! still zeroes the ghosts (below)
    if (mode == PA_ASSEMBLE && v->n_ghost > 0) {
      PA_HIP(hipSetDevice(c->device));
      hipLaunchKernelGGL(k_fill, dim3(grid_for(v->n_ghost, 256)), dim3(256), 0, c->s[0], v->d + v->n_own, (int64_t)v->n_ghost, 0.0);
      PA_HIP(hipGetLastError());
    }
    p->phase = 0;
    return PA_OK;
  }
  PA_HIP(hipSetDevice(c->device));
  pa_plan::side &in = in_side(p, mode);
  const bool early = mode == PA_CONSISTENT && p->own_comm_stream;
  p->own_comm_stream = false;
  if (early) {
    // One part per process (RCCL): the unpack writes ghost entries only, which nothing queued between pack and finish
    // may touch (the reference's wait(t) contract), so it runs on the comm stream right behind the receives, in the
    // shadow of own x own, and the compute stream waits for it: only own x ghost is left after the big kernel.  (With
    // all parts of a DebugArray on one GPU the comm stream is shared and this order measured 15-50 % slower.)
    if (in.n) hipLaunchKernelGGL(k_unpack_insert, dim3((in.n + 255) / 256), dim3(256), 0, c->s[1], v->d, in.d_buf, in.d_idx, (int)in.n);
    PA_HIP(hipEventRecord(p->ev_arrived, c->s[1]));
    PA_HIP(hipStreamWaitEvent(c->s[0], p->ev_arrived, 0));  // wait(t)
    PA_HIP(hipGetLastError());
    PA_TRY(pa_ipc_ack(p, mode));                            // (push transport: the senders may reuse the buffer; compute stream)
    p->phase = 0;
    return PA_OK;                                           // (the next pack is on the comm stream too: ordered)
  }
  if (p->phase == 1) {  // caller-driven transport on the comm stream: everything queued there so far counts
    PA_HIP(hipEventRecord(p->ev_arrived, c->s[1]));
  }
  PA_HIP(hipStreamWaitEvent(c->s[0], p->ev_wait ? p->ev_wait : p->ev_arrived, 0));  // wait(t)
  p->ev_wait = nullptr;
  if (mode == PA_CONSISTENT) {
    if (in.n) hipLaunchKernelGGL(k_unpack_insert, dim3((in.n + 255) / 256), dim3(256), 0, c->s[0], v->d, in.d_buf, in.d_idx, (int)in.n);
  } else {
    if (p->n_tgt)
      hipLaunchKernelGGL(k_unpack_add, dim3((p->n_tgt + 255) / 256), dim3(256), 0, c->s[0], v->d, in.d_buf, p->d_tgt, p->d_tptr,
                         p->d_tp, (int)p->n_tgt);
    // fill!(ghost_values(a),0) (src/p_vector.jl:703-705): EVERY ghost value, also the ones no message carries -- a periodic
    // direction with a single part makes wrap-around copies whose owner is the part itself; they are ghosts, are not
    // exchanged (compute_assembly_neighbors skips owner == rank, src/p_range.jl:441-445) and are zeroed all the same.  The
    // device layout is [own | ghost], so that is the tail of the vector.
    if (v->n_ghost > 0)
      hipLaunchKernelGGL(k_fill, dim3(grid_for(v->n_ghost, 256)), dim3(256), 0, c->s[0], v->d + v->n_own, (int64_t)v->n_ghost, 0.0);
  }
  PA_HIP(hipGetLastError());
  PA_TRY(pa_ipc_ack(p, mode));
  // the next pack (on the comm stream) must not overwrite buffers this unpack still reads (inside a capture the next pack's own
  // fork from the compute stream orders it; a trailing fork would be left unjoined)
  if (!c->capturing) {
    PA_HIP(hipEventRecord(c->ev_compute, c->s[0]));
    PA_HIP(hipStreamWaitEvent(c->s[1], c->ev_compute, 0));
  }
  p->phase = 0;
  return PA_OK;
}

// ------------------------------------------------------------------------------------------------
// operator level: mul!(c,a,b) of one part (or of all parts of a process) in one call
// ------------------------------------------------------------------------------------------------
extern "C" int pa_matrix_create(pa_ctx *c, const pa_csr *own_own, const pa_csr *own_ghost, pa_plan *col_plan, pa_matrix **out) {
  PA_REQUIRE(c && own_own && own_ghost && col_plan && out, "bad arguments");
  PA_REQUIRE(own_own->ctx == c && own_ghost->ctx == c && col_plan->ctx == c, "operands live in different contexts");
  PA_REQUIRE(own_own->t_rows == own_ghost->t_rows, "own_own has %lld rows, own_ghost %lld", (long long)own_own->t_rows,
             (long long)own_ghost->t_rows);
  PA_REQUIRE(own_own->n_cols + own_ghost->n_cols == col_plan->n_local,
             "blocks have %lld own + %lld ghost columns, the column plan %lld local ids", (long long)own_own->n_cols,
             (long long)own_ghost->n_cols, (long long)col_plan->n_local);
  pa_matrix *m = new pa_matrix();
  m->ctx = c; m->oo = own_own; m->oh = own_ghost; m->plan = col_plan;
  pa_watch_add(own_own, m);
  pa_watch_add(own_ghost, m);
  *out = m;
  return PA_OK;
}

// ---- who copies a block's values -------------------------------------------------------------------------------------------------
static std::mutex g_watch_mu;
static std::unordered_multimap<const pa_csr *, pa_matrix *> g_watch;
void pa_watch_add(const pa_csr *A, pa_matrix *m) {
  std::lock_guard<std::mutex> lk(g_watch_mu);
  g_watch.emplace(A, m);
}
void pa_watch_drop_matrix(pa_matrix *m) {
  std::lock_guard<std::mutex> lk(g_watch_mu);
  for (auto it = g_watch.begin(); it != g_watch.end();) it = it->second == m ? g_watch.erase(it) : std::next(it);
}
void pa_watch_drop_csr(const pa_csr *A) {
  std::lock_guard<std::mutex> lk(g_watch_mu);
  g_watch.erase(A);
}
static int matrix_rb(pa_matrix *m);
int pa_csr_values_changed(const pa_csr *A) {
  std::vector<pa_matrix *> ms;
  {
    std::lock_guard<std::mutex> lk(g_watch_mu);
    if (g_watch.empty()) return PA_OK;
    auto r = g_watch.equal_range(A);
    for (auto it = r.first; it != r.second; ++it) ms.push_back(it->second);
  }
  for (pa_matrix *m : ms) {
    if (m->transposed || (!m->oh_rb && !m->bd)) continue;     // nothing derived yet: the first product builds from current values
    // in place where that is possible (always for what a recorded graph may read: see matrix_rb / pa_matrix_fused_refresh)
    if (m->oh_rb && m->rb_epoch != m->oh->val_epoch && !m->oh->next && !m->oh->colsplit) PA_TRY(matrix_rb(m));
    if (m->bd) PA_TRY(pa_matrix_fused_refresh(m));
  }
  return PA_OK;
}

extern "C" int pa_matrix_destroy(pa_matrix *m) {
  if (m) pa_watch_drop_matrix(m);
  if (m && m->oh_rb) pa_csr_destroy(m->oh_rb);     // (own x ghost with renamed columns: made for this handle, matrix_rb)
  if (m) pa_matrix_fused_release(m);
  delete m;
  return PA_OK;
}

static int mul_check(const pa_matrix *m, const pa_vec *c, const pa_vec *b) {
  PA_REQUIRE(m && c && b, "bad arguments");
  PA_REQUIRE(!m->transposed, "a transposed matrix handle takes pa_mul5_transpose");
  // @boundscheck matching_own_indices / matching_ghost_indices (src/p_sparse_matrix.jl:2091-2093)
  PA_REQUIRE(c->n_own == m->oo->t_rows, "matching_own_indices(axes(c,1),axes(a,1)) failed");
  PA_REQUIRE(b->n_own == m->oo->n_cols && b->n_ghost == m->oh->n_cols, "matching_own/ghost_indices(axes(a,2),axes(b,1)) failed");
  return PA_OK;
}

// t = consistent!(b) / assemble!(c) of ONE part of this process: pack + transport over whichever link there is -- the RCCL
// communicator (one part per process), the plan's ipc link (pa_plan_ipc_connect: the pack kernel pushes into the neighbours'
// buffers), or nothing (the only part)
int pa_exchange_start(pa_plan *p, pa_comm *comm, pa_vec *v, int mode) {
  if (comm) {
    PA_TRY(pa_exchange_pack(p, v, mode));
    return pa_exchange_rccl(p, comm, mode);
  }
  if (pa_plan_ipc_connected(p)) return pa_exchange_push_ipc(p, v, mode);
  PA_REQUIRE(p->snd.nbr.empty() && p->rcv.nbr.empty(), "the plan has neighbours: pass the communicator (or connect the plans over ipc)");
  PA_REQUIRE(p->part == 0, "without a communicator the part must be the only one");
  PA_TRY(pa_exchange_pack(p, v, mode));
  pa_plan *one[1] = {p};
  return pa_exchange_local(one, 1, mode);
}

// own x ghost with its columns renamed to positions of consistent!'s receive buffer (built once per handle): the product then
// gathers b's ghost values straight from buffer_rcv, and the unpack that makes b itself consistent moves behind it, off the
// critical path of mul! (src/p_vector.jl:603-611 still runs, later).  Same entries, same order, same values gathered: same bits.
// Not built when a ghost column with stored entries gets no message (it would have nothing to read), inside a graph capture, or
// with PA_MUL_GHOST_FROM_BUFFER=0.
static int matrix_rb(pa_matrix *m) {
  if (m->transposed) return PA_OK;
  if (m->oh_rb && m->rb_epoch != m->oh->val_epoch) {
    // own_ghost's values were updated (pa_csr_update_values*, psparse!) since the twin copied them: the twin follows IN PLACE (the
    // stored entries keep their order, and a recorded graph keeps the twin's address) -- ADVICE r04: the product silently went on
    // multiplying own x ghost with the old values.  A column-split original has its values in piece order: that twin is rebuilt.
    if (!m->oh->next && !m->oh->colsplit) {
      pa_vec src;
      src.ctx = m->ctx; src.d = m->oh->d_val; src.n_own = m->oh->nnz; src.n_ghost = 0; src.owned = false;
      PA_TRY(pa_csr_update_values_from(m->oh_rb, &src, 0));
      m->rb_epoch = m->oh->val_epoch;
      return PA_OK;
    }
    PA_REQUIRE(!m->ctx->capturing, "own_ghost's values changed: the first product afterwards must run outside a graph capture");
    PA_HIP(hipStreamSynchronize(m->ctx->s[0]));
    PA_HIP(hipStreamSynchronize(m->ctx->s[1]));
    pa_csr_destroy(m->oh_rb);
    m->oh_rb = nullptr;
    m->rb_tried = false;
  }
  if (m->rb_tried) return PA_OK;
  if (m->ctx->capturing) return PA_OK;
  m->rb_tried = true;
  if (!m->ctx->sw.ghost_from_buffer) return PA_OK;
  pa_plan *p = m->plan;
  const pa_plan::side &in = p->snd;                           // the receiving side of consistent! (ghost lids)
  if (in.n == 0 || m->oh->t_nnz == 0 || (m->oh->next && !m->oh->colsplit)) return PA_OK;
  const int64_t n_own = m->oo->n_cols, n_ghost = m->oh->n_cols;
  std::vector<int32_t> map((size_t)n_ghost, -1);
  for (int64_t k = 0; k < in.n; ++k) {
    const int64_t g = (int64_t)in.idx[k] - n_own;
    if (g < 0 || g >= n_ghost || map[g] != -1) return PA_OK;  // (not a plain ghost list: the unpack path serves)
    map[g] = (int32_t)k;
  }
  pa_csr *rb = nullptr;
  if (pa_csr_create_remapped(m->oh, map.data(), in.n, &rb) != PA_OK) { (void)hipGetLastError(); return PA_OK; }
  m->oh_rb = rb;
  m->rb_epoch = m->oh->val_epoch;
  return PA_OK;
}

// own x ghost of one part after its exchange has been started and own x own queued
static int mul_ghost_part(pa_matrix *m, pa_vec *c, pa_vec *b, double alpha) {
  pa_plan *p = m->plan;
  if (m->oh_rb && (p->snd.n || p->rcv.n)) {
    PA_TRY(exchange_wait_arrived(p));                                        // wait(t), without the unpack
    pa_vec buf;
    buf.ctx = m->ctx; buf.d = p->snd.d_buf; buf.n_own = p->snd.n; buf.n_ghost = 0; buf.owned = false;
    PA_TRY(pa_spmv(m->oh_rb, &buf, PA_SEG_OWN, c, PA_SEG_OWN, alpha, 1.0));   // own x ghost from buffer_rcv
    return pa_exchange_finish(p, b, PA_CONSISTENT);                           // b's ghosts, behind it
  }
  PA_TRY(pa_exchange_finish(p, b, PA_CONSISTENT));                           // wait(t)
  return pa_spmv(m->oh, b, PA_SEG_GHOST, c, PA_SEG_OWN, alpha, 1.0);         // own x ghost
}

// src/p_sparse_matrix.jl:2105-2142 (assembled branch); alpha = 1, beta = 0 is :2090-2103
extern "C" int pa_mul5(pa_matrix *m, pa_comm *comm, pa_vec *c, pa_vec *b, double alpha, double beta) {
  PA_TRY(mul_check(m, c, b));
  PA_REQUIRE(c->d != b->d, "c and b alias");
  PA_TRY(matrix_rb(m));
  if (!comm && m->ctx->sw.mul_fused && pa_plan_ipc_connected(m->plan) && !m->ctx->capturing && (m->plan->snd.n || m->plan->rcv.n)) {
    // one part per process over the ipc link: push, both products, unpack and acknowledgement are ONE launch (pa_fused.hip)
    PA_TRY(pa_matrix_fused_build(m));
    const bool scaled = (m->oo->alpha_inside || m->oh->alpha_inside) && alpha != 1.0;
    if (pa_matrix_fused_ready(m) && !scaled && pa_fused_ipc_fits(m)) {
      pa_csr_before_product(m->oo);
      return pa_mul_fused_ipc(m, c, b, alpha, beta);
    }
  }
  if (comm && m->ctx->sw.mul_fused && m->ctx->sw.mul_fused_rccl && !m->fused_off && !m->ctx->capturing && (m->plan->snd.n || m->plan->rcv.n)) {
    // one part per process over RCCL: the transport on the comm stream, the whole product ONE launch beside it (pa_fused.hip)
    PA_TRY(pa_matrix_fused_build(m));
    const bool scaled = (m->oo->alpha_inside || m->oh->alpha_inside) && alpha != 1.0;
    if (pa_matrix_fused_ready(m) && !scaled) {
      pa_csr_before_product(m->oo);
      const int st = pa_mul_fused_rccl(m, comm, c, b, alpha, beta);
      if (st != PA_FUSED_GAVE_UP) return st;                                 // (an earlier product timed out: this one on the chain below)
    }
  }
  PA_TRY(pa_exchange_start(m->plan, comm, b, PA_CONSISTENT));                // t = consistent!(b)
  PA_TRY(pa_spmv(m->oo, b, PA_SEG_OWN, c, PA_SEG_OWN, alpha, beta));        // own x own, overlaps the exchange
  return mul_ghost_part(m, c, b, alpha);
}

extern "C" int pa_mul(pa_matrix *m, pa_comm *comm, pa_vec *c, pa_vec *b) { return pa_mul5(m, comm, c, b, 1.0, 0.0); }

// *yes = 1 when own x ghost of this handle reads consistent!'s receive buffer (decided at the first product; 0 before it)
extern "C" int pa_matrix_ghost_from_buffer(const pa_matrix *m, int *yes) {
  PA_REQUIRE(m && yes, "bad arguments");
  *yes = m->oh_rb != nullptr;
  return PA_OK;
}

// mul_no_lat!(c,a,b) (HPCG/src/hpcg_utils.jl:6-17): consistent!(b) |> wait FIRST, then the two local products -- the order
// HPCG's reference solver uses, and the "overlap off" side of bench.py's comparison.  Same kernels, same bits as pa_mul.
extern "C" int pa_mul_no_lat(pa_matrix *m, pa_comm *comm, pa_vec *c, pa_vec *b) {
  PA_TRY(mul_check(m, c, b));
  PA_REQUIRE(c->d != b->d, "c and b alias");
  PA_TRY(pa_exchange_start(m->plan, comm, b, PA_CONSISTENT));
  PA_TRY(pa_exchange_finish(m->plan, b, PA_CONSISTENT));
  PA_TRY(pa_spmv(m->oo, b, PA_SEG_OWN, c, PA_SEG_OWN, 1.0, 0.0));
  PA_TRY(pa_spmv(m->oh, b, PA_SEG_GHOST, c, PA_SEG_OWN, 1.0, 1.0));
  return PA_OK;
}

// Every part of this process.  Round 4: ONE push launch packs and delivers all parts (pa_push.hip), own x ghost reads the receive
// buffers, ONE launch unpacks b's ghosts behind it: 2 + 2 per part launches and no copies where round 3 queued 4 per part + one
// copy per directed edge.  PA_PUSH=0: the round-3 order (pack per part, device-to-device copies, unpack before own x ghost).
extern "C" int pa_mul_all(pa_matrix *const *m, int32_t n_parts, pa_vec *const *c, pa_vec *const *b, double alpha, double beta) {
  PA_REQUIRE(m && c && b && n_parts > 0, "bad arguments");
  std::vector<pa_plan *> plans(n_parts);
  for (int r = 0; r < n_parts; ++r) {
    PA_TRY(mul_check(m[r], c[r], b[r]));
    PA_REQUIRE(c[r]->d != b[r]->d, "c and b alias (part %d)", r);
    plans[r] = m[r]->plan;
  }
  const int push = m[0]->ctx->sw.push;
  bool all_rb = push != 0;
  if (push) {
    for (int r = 0; r < n_parts; ++r) {
      PA_TRY(matrix_rb(m[r]));
      if ((plans[r]->snd.n || plans[r]->rcv.n) && !m[r]->oh_rb && m[r]->oh->t_nnz) all_rb = false;
    }
    // Inside a graph capture (all parts in one context): ONE chain of kernels on the compute stream -- push, then per part own x own
    // and own x ghost from the receive buffers, then the unpack.  Nothing overlaps inside the chain (replayed, the kernels follow
    // each other without launch gaps), and no edge between two streams is recorded: such a graph replays 2.6 x slower than the eager
    // calls (config 5 on 8 parts: 0.118 ms per part against 0.045).
    bool one_ctx = true;
    for (int r = 1; r < n_parts; ++r) one_ctx = one_ctx && m[r]->ctx == m[0]->ctx;
    // Round 5: P + 1 launches on ONE stream, no events -- the push launch completes consistent!(b) of all parts (receive buffers AND
    // b's ghost entries), then every part is one launch: own x own's chunks, the boundary rows as the launch's tail (pa_fused.hip).
    // Parts whose handle cannot be fused (see pa_matrix_fused_build) run their two products separately behind the same push.
    if (all_rb && one_ctx && m[0]->ctx->sw.mul_fused) {
      bool any_fused = false, traffic = false;
      for (int r = 0; r < n_parts; ++r) {
        PA_TRY(pa_matrix_fused_build(m[r]));
        const bool nb = plans[r]->snd.n || plans[r]->rcv.n;
        traffic = traffic || nb;
        any_fused = any_fused || (nb && pa_matrix_fused_ready(m[r]));
      }
      if (any_fused) {
        PA_TRY(pa_exchange_push_unpack_one_stream(plans.data(), n_parts, b));
        for (int r = 0; r < n_parts; ++r) {
          pa_plan *p = plans[r];
          const bool nb = p->snd.n || p->rcv.n;
          const bool scaled = (m[r]->oo->alpha_inside || m[r]->oh->alpha_inside) && alpha != 1.0;
          if (nb && !scaled && pa_matrix_fused_ready(m[r])) {
            pa_csr_before_product(m[r]->oo);
            PA_TRY(pa_mul_fused_launch(m[r], c[r], b[r], alpha, beta, m[r]->ctx->s[0]));
            continue;
          }
          PA_TRY(pa_spmv(m[r]->oo, b[r], PA_SEG_OWN, c[r], PA_SEG_OWN, alpha, beta));
          if (!nb || !m[r]->oh_rb) continue;
          pa_vec buf;
          buf.ctx = m[r]->ctx; buf.d = p->snd.d_buf; buf.n_own = p->snd.n; buf.n_ghost = 0; buf.owned = false;
          PA_TRY(pa_spmv(m[r]->oh_rb, &buf, PA_SEG_OWN, c[r], PA_SEG_OWN, alpha, 1.0));
        }
        return PA_OK;
      }
    }
    if (all_rb && one_ctx && m[0]->ctx->capturing && m[0]->ctx->sw.graph_one_stream) {
      PA_TRY(pa_exchange_push_local_one_stream(plans.data(), n_parts, b, PA_CONSISTENT));
      for (int r = 0; r < n_parts; ++r) {
        pa_plan *p = plans[r];
        PA_TRY(pa_spmv(m[r]->oo, b[r], PA_SEG_OWN, c[r], PA_SEG_OWN, alpha, beta));
        if (!(p->snd.n || p->rcv.n) || !m[r]->oh_rb) continue;
        pa_vec buf;
        buf.ctx = m[r]->ctx; buf.d = p->snd.d_buf; buf.n_own = p->snd.n; buf.n_ghost = 0; buf.owned = false;
        PA_TRY(pa_spmv(m[r]->oh_rb, &buf, PA_SEG_OWN, c[r], PA_SEG_OWN, alpha, 1.0));
      }
      return pa_exchange_finish_all_insert(plans.data(), n_parts, b, 3);
    }
    PA_TRY(pa_exchange_push_local(plans.data(), n_parts, b, PA_CONSISTENT));
  } else {
    for (int r = 0; r < n_parts; ++r) PA_TRY(pa_exchange_pack(plans[r], b[r], PA_CONSISTENT));
    PA_TRY(pa_exchange_local(plans.data(), n_parts, PA_CONSISTENT));
  }
  if (all_rb) {
    // own x own of the parts one after the other on the compute stream; a part's own x ghost goes to the COMM stream, behind the
    // push launch (its data) and an event behind the part's own x own (its accumulator): the small kernel runs beside the next
    // part's own x own instead of between two of them.  The unpack of all parts follows there, and the compute stream joins.
    // The LAST part's own x ghost stays on the compute stream (nothing is left to run beside it, and a cross-stream hop costs ~8 us:
    // with everything on the comm stream config 3 on two parts measured 1.23 x own x own, 1.19 x with nothing there), the compute
    // stream then waits for the comm stream's products (long done) and the unpack of all parts follows on it.
    // The unpack that makes b itself consistent (src/p_vector.jl:603-611) reads the receive buffers and writes b's ghosts, which no
    // product of this call reads any more: it follows the push launch on the comm stream at once, beside own x own of the first
    // part, and the compute streams join it at the very end (wait(t)) -- nothing of consistent! is left on the critical path.
    PA_TRY(pa_exchange_finish_all_insert(plans.data(), n_parts, b, 2));
    // (per device context: with the parts on several GPUs -- one context each -- every GPU keeps ITS last own x ghost at home)
    std::vector<char> is_last(n_parts, 0);
    for (int r = n_parts - 1; r >= 0; --r) {
      if (!((plans[r]->snd.n || plans[r]->rcv.n) && m[r]->oh_rb)) continue;
      bool later = false;
      for (int q = r + 1; q < n_parts && !later; ++q) later = is_last[q] && m[q]->ctx == m[r]->ctx;
      if (!later) is_last[r] = 1;
    }
    std::vector<pa_ctx *> forked;
    for (int r = 0; r < n_parts; ++r) {
      pa_plan *p = plans[r];
      pa_ctx *cx = m[r]->ctx;
      PA_TRY(pa_spmv(m[r]->oo, b[r], PA_SEG_OWN, c[r], PA_SEG_OWN, alpha, beta));
      if (!(p->snd.n || p->rcv.n) || !m[r]->oh_rb) continue;
      pa_vec buf;
      buf.ctx = cx; buf.d = p->snd.d_buf; buf.n_own = p->snd.n; buf.n_ghost = 0; buf.owned = false;
      if (is_last[r]) {
        PA_TRY(exchange_wait_arrived(p));
        PA_TRY(pa_spmv(m[r]->oh_rb, &buf, PA_SEG_OWN, c[r], PA_SEG_OWN, alpha, 1.0));
        continue;
      }
      PA_HIP(hipEventRecord(p->ev_packed, cx->s[0]));
      PA_HIP(hipStreamWaitEvent(cx->s[1], p->ev_packed, 0));
      PA_TRY(pa_spmv_on(m[r]->oh_rb, &buf, PA_SEG_OWN, c[r], PA_SEG_OWN, alpha, 1.0, cx->s[1]));
      if (std::find(forked.begin(), forked.end(), cx) == forked.end()) forked.push_back(cx);
      PA_HIP(hipEventRecord(p->ev_arrived, cx->s[1]));            // (the newest of these per device is what the compute stream joins on)
      p->ev_wait = p->ev_arrived;
    }
    for (pa_ctx *cx : forked) {                                    // join: the products queued on the comm streams
      int newest = -1;
      for (int r = 0; r < n_parts; ++r) if (m[r]->ctx == cx && !is_last[r] && plans[r]->ev_wait == plans[r]->ev_arrived && m[r]->oh_rb) newest = r;
      if (newest >= 0) PA_HIP(hipStreamWaitEvent(cx->s[0], plans[newest]->ev_arrived, 0));
    }
    return pa_exchange_join_all(plans.data(), n_parts);
  }
  for (int r = 0; r < n_parts; ++r) PA_TRY(pa_spmv(m[r]->oo, b[r], PA_SEG_OWN, c[r], PA_SEG_OWN, alpha, beta));
  for (int r = 0; r < n_parts; ++r) {
    if (push) PA_TRY(mul_ghost_part(m[r], c[r], b[r], alpha));
    else {
      PA_TRY(pa_exchange_finish(plans[r], b[r], PA_CONSISTENT));
      PA_TRY(pa_spmv(m[r]->oh, b[r], PA_SEG_GHOST, c[r], PA_SEG_OWN, alpha, 1.0));
    }
  }
  return PA_OK;
}

// ---- mul!(c,a,b) that also leaves dot(b,c) in a slot: the CG loop's c = A*u and u'c (HPCG/src/ref_cg.jl:59-60) with no pass
// over u and c for the dot.  Every chunk of the product kernels (EPI 3) writes its partial sum of b_own[row] * (row's
// products); own x own and own x ghost each contribute their own products, so the total is b_own'(A_oo b_own + A_oh b_ghost).
static int dot_scratch(pa_ctx *c, int64_t n) {
  if (n <= c->n_dotpart) return PA_OK;
  if (c->capturing) { pa_set_err("the fused product + dot needs its scratch before a capture opens (run it once eagerly)"); return PA_ERR_STATE; }
  PA_HIP(hipStreamSynchronize(c->s[0]));
  if (c->d_dotpart) pa_dev_free(c, c->d_dotpart);
  c->d_dotpart = nullptr;
  c->n_dotpart = 0;
  // a write stream of the product kernels like y: it must not sit in the matrix streams' memory class either
  const int64_t cap = std::max<int64_t>(n + n / 4 + 64, (int64_t)1 << 17);
  PA_TRY(pa_dev_alloc(c, (void **)&c->d_dotpart, sizeof(double) * (size_t)cap, PA_MEM_VECTOR));
  c->n_dotpart = cap;
  return PA_OK;
}

// one block (all of its slabs): y_seg = beta*y_seg + A*x_seg, partial[off + chunk] = that chunk's share of u'(A x)
static int spmv_dot_block(const pa_csr *A, const double *x, double *y, double beta, const double *u, double *partial) {
  pa_ctx *c = A->ctx;
  int64_t off = 0;
  for (const pa_csr *S = A; S; S = S->next) {
    double *ys = y + S->row0;
    const double *us = u + S->row0;
    double kbeta = S->accumulate ? 1.0 : beta;
    if (S->compact && kbeta != 1.0) {
      if (S->n_rows) hipLaunchKernelGGL(k_scale, dim3(grid_for(S->n_rows, 256)), dim3(256), 0, c->s[0], ys, S->n_rows, beta);
      kbeta = 1.0;
    }
    if (const int pm = pa_pell_mode(S)) {                // a pattern block: one partial per slab of 64 rows (pa_pell.h)
      PA_TRY(pa_pell_launch(S, pm, 3, x, ys, 1.0, kbeta, partial + off, us, nullptr, c->s[0]));
      off += pa_pell_partials(S);
      continue;
    }
    if (S->n_xw_groups > 0) {
      pa_launch_xwin(S, x, ys, 1.0, kbeta, us, partial + off);
    } else if (S->n_chunks > 0) {
      const int cpx = (int)((S->n_chunks + 7) / 8);
#define PA_LAUNCH_DOT(C16, PAT)                                                                                           \
  hipLaunchKernelGGL((k_spmv_rowsplit<SPMV_BLK, SPMV_NPT, SPMV_NT, C16, PAT, 3, false>), dim3(cpx * 8), dim3(SPMV_BLK), 0, \
                     c->s[0], S->d_crp, S->d_col, S->d_col16, S->d_win, S->d_pdesc, S->d_pdelta, S->d_val, x, ys,     \
                     S->d_chunk_rp, S->d_row_ids, (int)S->n_chunks, cpx, 1.0, kbeta, partial + off, us,              \
                     (const double *)nullptr, (const unsigned char *)nullptr, (const double *)nullptr,                 \
                     (const int *)nullptr, (int)S->n_cols - 1)
      switch ((S->use_pattern ? (S->compact ? 2 : 1) : 0) * 2 + (S->use_c16 ? 1 : 0)) {
        case 5: PA_LAUNCH_DOT(true, 2); break;
        case 4: PA_LAUNCH_DOT(false, 2); break;
        case 3: PA_LAUNCH_DOT(true, 1); break;
        case 2: PA_LAUNCH_DOT(false, 1); break;
        case 1: PA_LAUNCH_DOT(true, 0); break;
        default: PA_LAUNCH_DOT(false, 0); break;
      }
#undef PA_LAUNCH_DOT
    }
    off += S->n_chunks;
  }
  PA_HIP(hipGetLastError());
  return PA_OK;
}

static int64_t chunks_of(const pa_csr *A) {             // partial sums a fused product + dot of this block writes (spmv_dot_block)
  int64_t n = 0;
  for (const pa_csr *S = A; S; S = S->next) n += pa_pell_mode(S) ? pa_pell_partials(S) : S->n_chunks;
  return n;
}
static bool has_vdict(const pa_csr *A) {             // a slab on the ROW-SPLIT kernel's one-byte stream (it has no product + dot form; pattern-ELL's
  for (const pa_csr *S = A; S; S = S->next)            // bit and byte streams do: pa_pell_launch, epilogue 3)
    if (S->use_vdict && (S->vdict_stale || pa_pell_mode(S) == 0)) return true;      // (stale: pa_spmv counts the products towards the renewal)
  return false;
}

// the part's share of dot(b,c), reduced into the slot (two small launches)
static int dot_finish(pa_ctx *c, int64_t n_partials, int slot, int accumulate) {
  if (n_partials == 0) {
    if (!accumulate) hipLaunchKernelGGL(k_fill, dim3(1), dim3(64), 0, c->s[0], c->d_scalar + slot, (int64_t)1, 0.0);
  } else {
    const int nb = grid_for(n_partials, 256 * 8, c->n_partials);
    hipLaunchKernelGGL(k_sum_partial, dim3(nb), dim3(256), 0, c->s[0], c->d_dotpart, n_partials, c->d_partials);
    hipLaunchKernelGGL(k_dot_final_slot, dim3(1), dim3(256), 0, c->s[0], c->d_partials, nb, c->d_scalar + slot, accumulate);
  }
  PA_HIP(hipGetLastError());
  return PA_OK;
}

static int mul_dot_part(pa_matrix *m, pa_vec *cv, pa_vec *b, int slot, int accumulate, bool first_half, bool second_half) {
  pa_ctx *c = m->ctx;
  if (first_half) pa_csr_before_product(m->oo);        // (a block whose values were updated counts its products towards its dictionary's renewal)
  const int64_t noo = chunks_of(m->oo), noh = chunks_of(m->oh);
  if (has_vdict(m->oo) || has_vdict(m->oh)) {       // (value-dictionary blocks: the plain product, then the dot as its own pass)
    if (first_half) PA_TRY(pa_spmv(m->oo, b, PA_SEG_OWN, cv, PA_SEG_OWN, 1.0, 0.0));
    if (second_half) {
      PA_TRY(pa_spmv(m->oh, b, PA_SEG_GHOST, cv, PA_SEG_OWN, 1.0, 1.0));
      PA_TRY(pa_vec_dot_slot(b, cv, slot, accumulate));
    }
    return PA_OK;
  }
  if (first_half) {
    PA_TRY(dot_scratch(c, noo + noh));
    PA_TRY(spmv_dot_block(m->oo, b->d, cv->d, 0.0, b->d, c->d_dotpart));
  }
  if (second_half) {
    PA_TRY(spmv_dot_block(m->oh, b->d + b->n_own, cv->d, 1.0, b->d, c->d_dotpart + noo));
    PA_TRY(dot_finish(c, noo + noh, slot, accumulate));
  }
  return PA_OK;
}

extern "C" int pa_mul_dot(pa_matrix *m, pa_comm *comm, pa_vec *c, pa_vec *b, int slot, int accumulate) {
  PA_TRY(mul_check(m, c, b));
  PA_REQUIRE(c->d != b->d, "c and b alias");
  PA_REQUIRE(PA_SLOT_OK(slot), "slot %d out of range [0,%d)", slot, PA_N_SLOTS);
  PA_REQUIRE(b->n_own == c->n_own, "dot(b,c) needs a square operator: %lld columns, %lld rows", (long long)b->n_own, (long long)c->n_own);
  PA_HIP(hipSetDevice(m->ctx->device));
  PA_TRY(pa_exchange_start(m->plan, comm, b, PA_CONSISTENT));
  PA_TRY(mul_dot_part(m, c, b, slot, accumulate, true, false));
  PA_TRY(pa_exchange_finish(m->plan, b, PA_CONSISTENT));
  PA_TRY(mul_dot_part(m, c, b, slot, accumulate, false, true));
  return PA_OK;
}

// every part of one process: the slot ends up holding the sum over the parts, added in part order
extern "C" int pa_mul_all_dot(pa_matrix *const *m, int32_t n_parts, pa_vec *const *c, pa_vec *const *b, int slot) {
  PA_REQUIRE(m && c && b && n_parts > 0, "bad arguments");
  PA_REQUIRE(PA_SLOT_OK(slot), "slot %d out of range [0,%d)", slot, PA_N_SLOTS);
  std::vector<pa_plan *> plans(n_parts);
  for (int r = 0; r < n_parts; ++r) {
    PA_TRY(mul_check(m[r], c[r], b[r]));
    PA_REQUIRE(c[r]->d != b[r]->d, "c and b alias (part %d)", r);
    PA_REQUIRE(b[r]->n_own == c[r]->n_own, "dot(b,c) needs a square operator (part %d)", r);
    PA_REQUIRE(m[r]->ctx == m[0]->ctx, "the parts of one call share a context");
    plans[r] = m[r]->plan;
  }
  const int push = m[0]->ctx->sw.push;
  if (push) PA_TRY(pa_exchange_push_local(plans.data(), n_parts, b, PA_CONSISTENT));
  else {
    for (int r = 0; r < n_parts; ++r) PA_TRY(pa_exchange_pack(plans[r], b[r], PA_CONSISTENT));
    PA_TRY(pa_exchange_local(plans.data(), n_parts, PA_CONSISTENT));
  }
  // the parts share the context's partial-sum scratch: part r's product + reduction run before part r+1's first half
  // overwrites it (one stream: in order), so own x own of part r cannot wait for ALL exchanges as pa_mul_all's does --
  // one part (the benchmark's case) loses nothing
  for (int r = 0; r < n_parts; ++r) {
    PA_TRY(mul_dot_part(m[r], c[r], b[r], slot, r > 0, true, false));
    PA_TRY(pa_exchange_finish(plans[r], b[r], PA_CONSISTENT));
    PA_TRY(mul_dot_part(m[r], c[r], b[r], slot, r > 0, false, true));
  }
  return PA_OK;
}
