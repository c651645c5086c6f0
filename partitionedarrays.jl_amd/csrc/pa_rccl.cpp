// pa_rccl.cpp -- RCCL transport of the ghost exchange: one process per part / GPU, point-to-point
// over xGMI.  Replaces the MPI backend's exchange_impl! (src/mpi_array.jl:575-614: Irecv!/Isend per
// neighbour, Waitall at wait(t)) with ONE ncclGroup of ncclRecv/ncclSend per neighbour on the comm
// stream; "Waitall" is the event the compute stream waits on in pa_exchange_finish.
//
// librccl is dlopen'ed at first use: inside a PyTorch process that resolves to the librccl torch
// already loaded (same SONAME, librccl.so.1), in a Julia process to /opt/rocm/lib/librccl.so.1.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <cstring>
#include <mutex>
#include <vector>

#include "pa_internal.h"

namespace {
struct Api {
  void *h = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  bool ok = false;
};
Api g_api;
std::once_flag g_once;

void load_api() {
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  for (const char *n : names) {
    g_api.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (g_api.h) break;
  }
  if (!g_api.h) return;
#define PA_SYM(field, sym) g_api.field = (decltype(g_api.field))dlsym(g_api.h, #sym)
  PA_SYM(GetUniqueId, ncclGetUniqueId);
  PA_SYM(CommInitRank, ncclCommInitRank);
  PA_SYM(CommInitAll, ncclCommInitAll);
  PA_SYM(CommDestroy, ncclCommDestroy);
  PA_SYM(GroupStart, ncclGroupStart);
  PA_SYM(GroupEnd, ncclGroupEnd);
  PA_SYM(Send, ncclSend);
  PA_SYM(Recv, ncclRecv);
  PA_SYM(AllReduce, ncclAllReduce);
  PA_SYM(GetErrorString, ncclGetErrorString);
  PA_SYM(CommCount, ncclCommCount);
  PA_SYM(CommUserRank, ncclCommUserRank);
#undef PA_SYM
  g_api.ok = g_api.GetUniqueId && g_api.CommInitRank && g_api.CommDestroy && g_api.GroupStart && g_api.GroupEnd &&
             g_api.Send && g_api.Recv && g_api.AllReduce && g_api.GetErrorString && g_api.CommCount && g_api.CommUserRank;
}

int need_api() {
  std::call_once(g_once, load_api);
  if (!g_api.ok) {
    pa_set_err("librccl could not be loaded (%s)", dlerror() ? dlerror() : "missing symbols");
    return PA_ERR_RCCL;
  }
  return PA_OK;
}
}  // namespace

#define PA_NCCL(call)                                                                              \
  do {                                                                                             \
    ncclResult_t pa_r_ = (call);                                                                   \
    if (pa_r_ != ncclSuccess) {                                                                    \
      pa_set_err("%s failed: %s (%s:%d)", #call, g_api.GetErrorString(pa_r_), __FILE__, __LINE__); \
      return PA_ERR_RCCL;                                                                          \
    }                                                                                              \
  } while (0)

struct pa_comm {
  pa_ctx *ctx = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, nranks = 1;
  double *d_token = nullptr;
};

static_assert(sizeof(ncclUniqueId) == PA_UNIQUE_ID_BYTES, "ncclUniqueId size");

extern "C" int pa_comm_unique_id(char id[PA_UNIQUE_ID_BYTES]) {
  PA_REQUIRE(id != nullptr, "id is NULL");
  PA_TRY(need_api());
  ncclUniqueId u;
  PA_NCCL(g_api.GetUniqueId(&u));
  memcpy(id, &u, sizeof u);
  return PA_OK;
}

extern "C" int pa_comm_create(pa_ctx *c, const char id[PA_UNIQUE_ID_BYTES], int rank, int nranks, pa_comm **out) {
  PA_REQUIRE(c && id && out, "bad arguments");
  PA_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "rank %d outside [0,%d)", rank, nranks);
  PA_TRY(need_api());
  PA_HIP(hipSetDevice(c->device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  pa_comm *m = new pa_comm();
  m->ctx = c; m->rank = rank; m->nranks = nranks;
  PA_NCCL(g_api.CommInitRank(&m->comm, nranks, u, rank));
  PA_HIP(pa_raw_malloc(&m->d_token, sizeof(double)));
  PA_HIP(hipMemsetAsync(m->d_token, 0, sizeof(double), c->s[1]));
  PA_HIP(hipStreamSynchronize(c->s[1]));
  *out = m;
  return PA_OK;
}

extern "C" int pa_comm_destroy(pa_comm *m) {
  if (!m) return PA_OK;
  if (need_api() != PA_OK) return PA_ERR_RCCL;
  (void)hipSetDevice(m->ctx->device);
  (void)hipStreamSynchronize(m->ctx->s[0]);
  (void)hipStreamSynchronize(m->ctx->s[1]);
  (void)pa_raw_free(m->d_token);
  g_api.CommDestroy(m->comm);
  delete m;
  return PA_OK;
}

extern "C" int pa_comm_allreduce_sum(pa_comm *m, void *ptr, int64_t count, int which) {
  PA_REQUIRE(m && ptr && count >= 0 && (which == 0 || which == 1), "bad arguments");
  PA_TRY(need_api());
  PA_HIP(hipSetDevice(m->ctx->device));
  PA_NCCL(g_api.AllReduce(ptr, ptr, (size_t)count, ncclDouble, ncclSum, m->comm, m->ctx->s[which]));
  return PA_OK;
}

extern "C" int pa_comm_barrier(pa_comm *m) {
  PA_REQUIRE(m != nullptr, "comm is NULL");
  PA_TRY(pa_comm_allreduce_sum(m, m->d_token, 1, PA_STREAM_COMM));
  PA_HIP(hipStreamSynchronize(m->ctx->s[1]));
  return PA_OK;
}

extern "C" int pa_comm_info(pa_comm *m, int *rank, int *nranks) {
  PA_REQUIRE(m != nullptr, "comm is NULL");
  PA_TRY(need_api());
  int r = -1, n = -1;
  PA_NCCL(g_api.CommUserRank(m->comm, &r));
  PA_NCCL(g_api.CommCount(m->comm, &n));
  if (rank) *rank = r;
  if (nranks) *nranks = n;
  return PA_OK;
}

extern "C" int pa_exchange_rccl(pa_plan *p, pa_comm *m, int mode) {
  PA_REQUIRE(p && m && (mode == PA_ASSEMBLE || mode == PA_CONSISTENT), "bad arguments");
  PA_REQUIRE(p->phase == 1 && p->mode == mode, "pa_exchange_pack(mode) must come first");
  PA_REQUIRE(p->ctx == m->ctx, "plan and communicator live on different contexts");
  PA_REQUIRE(p->part == m->rank, "plan of part %d driven by rank %d", p->part, m->rank);
  if (p->snd.n == 0 && p->rcv.n == 0) { p->phase = 2; return PA_OK; }
  PA_TRY(need_api());
  pa_plan::side &o = (mode == PA_ASSEMBLE) ? p->snd : p->rcv;
  pa_plan::side &in = (mode == PA_ASSEMBLE) ? p->rcv : p->snd;
  hipStream_t st = p->ctx->s[1];
  PA_HIP(hipSetDevice(p->ctx->device));
  for (int32_t q : o.nbr) PA_REQUIRE(q >= 0 && q < m->nranks, "bad send neighbour %d", q);
  for (int32_t q : in.nbr) PA_REQUIRE(q >= 0 && q < m->nranks, "bad receive neighbour %d", q);
  const size_t eb = (size_t)p->elem;                                 // (a Float32 payload, pa_exchange_pack32: floats at the same element offsets)
  const ncclDataType_t dt = p->elem == 4 ? ncclFloat : ncclDouble;
  PA_NCCL(g_api.GroupStart());
  // (a failing ncclRecv / ncclSend must not leave the group OPEN -- the next RCCL call of this thread would be swallowed by it:
  // the group is always closed, then the first failure is reported)
  ncclResult_t first = ncclSuccess;
  const char *what = "";
  for (size_t i = 0; i < in.nbr.size() && first == ncclSuccess; ++i) {
    const size_t len = (size_t)(in.ptrs[i + 1] - in.ptrs[i]);
    if (len) { first = g_api.Recv(reinterpret_cast<char *>(in.d_buf) + eb * in.ptrs[i], len, dt, in.nbr[i], m->comm, st); what = "ncclRecv"; }
  }
  for (size_t j = 0; j < o.nbr.size() && first == ncclSuccess; ++j) {
    const size_t len = (size_t)(o.ptrs[j + 1] - o.ptrs[j]);
    if (len) { first = g_api.Send(reinterpret_cast<const char *>(o.d_buf) + eb * o.ptrs[j], len, dt, o.nbr[j], m->comm, st); what = "ncclSend"; }
  }
  const ncclResult_t closed = g_api.GroupEnd();
  if (first != ncclSuccess) {
    pa_set_err("%s failed inside the exchange group of part %d: %s", what, p->part, g_api.GetErrorString(first));
    return PA_ERR_RCCL;
  }
  if (closed != ncclSuccess) {
    pa_set_err("ncclGroupEnd failed for the exchange of part %d: %s", p->part, g_api.GetErrorString(closed));
    return PA_ERR_RCCL;
  }
  p->own_comm_stream = true;
  return pa_plan_mark_arrived(p);
}

// ---- all parts in ONE process, one GPU each (SURVEY 8(b)'s sketch: pa_ctx_create per device, ONE ncclGroup across the parts) ------
// The DebugArray model over several GPUs (the Python mirror's PA_CTX_PER_PART=1): part p has its own context on device p and the
// exchange is ONE group of ncclSend / ncclRecv over every part's communicator (ncclCommInitAll: the single-process multi-GPU form of
// RCCL).  Beside the peer copies of pa_exchange_local; same plans, same buffers, same finish.
extern "C" int pa_comm_create_all(pa_ctx *const *ctxs, int32_t n, pa_comm **out) {
  PA_REQUIRE(ctxs && out && n >= 1, "bad arguments");
  PA_TRY(need_api());
  PA_REQUIRE(g_api.CommInitAll != nullptr, "this librccl has no ncclCommInitAll");
  std::vector<int> devs((size_t)n);
  for (int i = 0; i < n; ++i) {
    PA_REQUIRE(ctxs[i] != nullptr, "ctxs[%d] is NULL", i);
    devs[(size_t)i] = ctxs[i]->device;
    for (int j = 0; j < i; ++j)
      PA_REQUIRE(devs[(size_t)j] != devs[(size_t)i], "parts %d and %d share device %d: RCCL wants one device per rank (use pa_exchange_local / "
                 "pa_exchange_push_local for parts that share a GPU)", j, i, devs[(size_t)i]);
  }
  std::vector<ncclComm_t> comms((size_t)n, nullptr);
  PA_NCCL(g_api.CommInitAll(comms.data(), n, devs.data()));
  for (int i = 0; i < n; ++i) {
    pa_comm *m = new pa_comm();
    m->ctx = ctxs[i]; m->comm = comms[(size_t)i]; m->rank = i; m->nranks = n;
    PA_HIP(hipSetDevice(ctxs[i]->device));
    PA_HIP(pa_raw_malloc(&m->d_token, sizeof(double)));
    PA_HIP(hipMemsetAsync(m->d_token, 0, sizeof(double), ctxs[i]->s[1]));
    PA_HIP(hipStreamSynchronize(ctxs[i]->s[1]));
    out[i] = m;
  }
  return PA_OK;
}

extern "C" int pa_exchange_rccl_all(pa_plan *const *plans, pa_comm *const *comms, int32_t n, int mode) {
  PA_REQUIRE(plans && comms && n >= 1 && (mode == PA_ASSEMBLE || mode == PA_CONSISTENT), "bad arguments");
  PA_TRY(need_api());
  for (int r = 0; r < n; ++r) {
    pa_plan *p = plans[r];
    pa_comm *m = comms[r];
    PA_REQUIRE(p && m, "part %d: NULL plan or communicator", r);
    PA_REQUIRE(p->phase == 1 && p->mode == mode, "part %d: pa_exchange_pack(mode) must come first", r);
    PA_REQUIRE(p->ctx == m->ctx && p->part == m->rank && m->nranks == n, "part %d: plan and communicator do not belong together", r);
    PA_REQUIRE(p->elem == plans[0]->elem, "the parts packed payloads of different element types");
    const pa_plan::side &o = (mode == PA_ASSEMBLE) ? p->snd : p->rcv, &in = (mode == PA_ASSEMBLE) ? p->rcv : p->snd;
    for (int32_t q : o.nbr) PA_REQUIRE(q >= 0 && q < n, "part %d: bad send neighbour %d", r, q);
    for (int32_t q : in.nbr) PA_REQUIRE(q >= 0 && q < n, "part %d: bad receive neighbour %d", r, q);
  }
  const size_t eb = (size_t)plans[0]->elem;
  const ncclDataType_t dt = plans[0]->elem == 4 ? ncclFloat : ncclDouble;
  PA_NCCL(g_api.GroupStart());
  ncclResult_t first = ncclSuccess;
  int bad_part = -1;
  for (int r = 0; r < n && first == ncclSuccess; ++r) {
    pa_plan *p = plans[r];
    pa_comm *m = comms[r];
    if (p->snd.n == 0 && p->rcv.n == 0) continue;
    pa_plan::side &o = (mode == PA_ASSEMBLE) ? p->snd : p->rcv;
    pa_plan::side &in = (mode == PA_ASSEMBLE) ? p->rcv : p->snd;
    hipStream_t st = p->ctx->s[1];
    for (size_t i = 0; i < in.nbr.size() && first == ncclSuccess; ++i) {
      const size_t len = (size_t)(in.ptrs[i + 1] - in.ptrs[i]);
      if (len) first = g_api.Recv(reinterpret_cast<char *>(in.d_buf) + eb * in.ptrs[i], len, dt, in.nbr[i], m->comm, st);
    }
    for (size_t j = 0; j < o.nbr.size() && first == ncclSuccess; ++j) {
      const size_t len = (size_t)(o.ptrs[j + 1] - o.ptrs[j]);
      if (len) first = g_api.Send(reinterpret_cast<const char *>(o.d_buf) + eb * o.ptrs[j], len, dt, o.nbr[j], m->comm, st);
    }
    if (first != ncclSuccess) bad_part = r;
  }
  const ncclResult_t closed = g_api.GroupEnd();          // (the group is always closed, then the first failure is reported)
  if (first != ncclSuccess) {
    pa_set_err("ncclSend / ncclRecv failed inside the exchange group at part %d: %s", bad_part, g_api.GetErrorString(first));
    return PA_ERR_RCCL;
  }
  if (closed != ncclSuccess) {
    pa_set_err("ncclGroupEnd failed for the exchange of %d parts: %s", (int)n, g_api.GetErrorString(closed));
    return PA_ERR_RCCL;
  }
  for (int r = 0; r < n; ++r) {
    pa_plan *p = plans[r];
    if (p->snd.n == 0 && p->rcv.n == 0) { p->phase = 2; continue; }
    PA_HIP(hipSetDevice(p->ctx->device));
    p->own_comm_stream = true;
    PA_TRY(pa_plan_mark_arrived(p));
  }
  return PA_OK;
}
