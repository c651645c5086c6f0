// pa_device.hip -- gfx950 kernels and the device half of the C ABI declared in include/pa_hip.h.
//
// Hot path (reference file:line in /root/reference):
//   K1/K2  k_spmv_rowsplit   spmv_csr! src/sparse_utils.jl:649-669, muladd! src/p_sparse_matrix.jl:2088
//   K3     k_pack            src/p_vector.jl:595-599
//   K4     k_unpack_insert   src/p_vector.jl:605-609 with f = insert (:755)
//   K5     k_unpack_add      same loop with f = + (:695-697), deterministic (ascending p per target)
//   K6     k_fill (ghosts)   src/p_vector.jl:703-705
//   K8     k_axpby / k_dot_* src/p_vector.jl:1189-1277
//
// This file is compiled with -ffp-contract=off: every product and every sum is rounded once, in
// the reference's order, so SpMV is bit-identical to the CPU loop (no FMA contraction).
//
// SpMV design (bandwidth-bound; no MFMA on purpose):
//   * host-side "row split": consecutive rows are grouped into chunks of <= 1536 stored entries
//     (PA_SPMV_CHUNK_NNZ); one 256-thread workgroup per chunk.
//   * load phase: every lane streams 16-byte value pairs + 8-byte column pairs (fully coalesced,
//     non-temporal: the matrix is read once and must not evict x from L2), gathers x through
//     L1/L2, multiplies, and stages the products in LDS (12 KiB per workgroup).
//   * reduce phase: one lane per row walks its products in LDS in ascending p -- the reference's
//     left-to-right order -- and writes y.  64-wide wavefronts: lanes of a wave own consecutive rows,
//     so their LDS reads are stride-(row length) apart: conflict-free for 27 (odd), 2-way for 18.
//   * blockIdx -> chunk map is XCD-aware: block b runs on XCD b%8, so XCD i gets the i-th contiguous
//     eighth of the rows; neighbouring workgroups of one XCD share x lines in that XCD's 4 MiB L2.
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
#include <vector>

#include "pa_internal.h"
#include "pa_setup.h"
#include "pa_scratch.h"

thread_local std::string g_pa_err;
thread_local int pa_tls_plain_encoding = 0;   // > 0 while pa_matrix_fused_build makes its block: Int32 columns, nothing else
thread_local int pa_tls_piece_build = 0;       // > 0 while pa_csr_colsplit_if_wide builds its pieces through csr_build
thread_local const std::vector<int32_t> *pa_tls_row_breaks = nullptr;   // ... and the rows every piece's chunks and ring groups are cut at

void pa_set_err(const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_pa_err = buf;
}

extern "C" const char *pa_last_error(void) { return g_pa_err.c_str(); }
extern "C" int pa_version(void) { return 100; }

extern "C" int pa_device_count(int *count) {
  PA_REQUIRE(count != nullptr, "count is NULL");
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    n = 0;
  }
  *count = n;
  return PA_OK;
}

#include "pa_dev_kernels.h"

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
static void enable_peer_access(int device);

static void read_switches(pa_ctx *c) {
  auto flag = [](const char *name, int dflt) { const char *e = getenv(name); return e ? atoi(e) : dflt; };
  c->sw.push = flag("PA_PUSH", 1);
  c->sw.graph_one_stream = flag("PA_GRAPH_ONE_STREAM", 1);
  c->sw.ghost_from_buffer = flag("PA_MUL_GHOST_FROM_BUFFER", 1);
  c->sw.mul_fused = flag("PA_MUL_FUSED", 1);
  c->sw.mul_fused_rccl = flag("PA_MUL_FUSED_RCCL", 0);
  c->sw.fused_tail_blocks = std::max(1, flag("PA_FUSED_TAIL_BLOCKS", 1024));
  c->sw.spmv_alternate = flag("PA_SPMV_ALTERNATE", 1);
  c->sw.chain_fused = flag("PA_SPMV_CHAIN_FUSED", 1);
  c->sw.vd_select = flag("PA_SPMV_VDICT_SELECT", 1);
  c->sw.pell = flag("PA_SPMV_PELL", 1);
  c->sw.pell_lean = flag("PA_SPMV_PELL_LEAN", 1);
  c->sw.pell_bytes = flag("PA_SPMV_PELL_BYTES", 1);
  c->sw.test_skip_raise = flag("PA_TEST_FUSED_SKIP_RAISE", 0);
}
extern "C" int pa_ctx_reload_env(pa_ctx *c) {
  PA_REQUIRE(c != nullptr, "bad arguments");
  read_switches(c);
  return PA_OK;
}

extern "C" int pa_ctx_create(int device, pa_ctx **out) {
  PA_REQUIRE(out != nullptr, "ctx out pointer is NULL");
  int n = 0;
  PA_HIP(hipGetDeviceCount(&n));
  PA_REQUIRE(device >= 0 && device < n, "device %d out of range (have %d)", device, n);
  if (n > 1) enable_peer_access(device);
  PA_HIP(hipSetDevice(device));
  pa_ctx *c = new pa_ctx();
  c->device = device;
  read_switches(c);
  // The comm stream gets the highest priority the device offers: own x own puts ~300 k workgroups in front of the
  // dispatcher, and the pack kernel, RCCL's send/recv kernels and the unpack have to get CUs while it runs or the
  // exchange does not hide under it (mul!: src/p_sparse_matrix.jl:2098-2100).  Numerically lower = higher priority.
  int prio_least = 0, prio_greatest = 0;
  PA_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
  c->comm_priority = prio_greatest;
  int prio_compute = prio_least;
  if (const char *e = getenv("PA_COMPUTE_PRIORITY")) prio_compute = atoi(e);     // (measurements)
  PA_HIP(hipStreamCreateWithPriority(&c->s[0], hipStreamNonBlocking, prio_compute));
  PA_HIP(hipStreamCreateWithPriority(&c->s[1], hipStreamNonBlocking, prio_greatest));
  PA_HIP(hipEventCreateWithFlags(&c->ev_compute, hipEventDisableTiming));
  hipDeviceProp_t prop;
  PA_HIP(hipGetDeviceProperties(&prop, device));
  c->cus = prop.multiProcessorCount;
  c->xcds = 8;  // gfx950: 8 XCDs x 32 CUs
  c->hbm = prop.totalGlobalMem;
  snprintf(c->name, sizeof c->name, "%s (%s)", prop.name, prop.gcnArchName);
  c->n_partials = 1024;
  PA_HIP(pa_raw_malloc(&c->d_partials, sizeof(double) * c->n_partials));
  PA_HIP(pa_raw_malloc(&c->d_scalar, sizeof(double) * PA_N_SLOTS));
  PA_HIP(hipMemsetAsync(c->d_scalar, 0, sizeof(double) * PA_N_SLOTS, c->s[0]));   // (on the stream the slot kernels run on: the
  PA_HIP(hipStreamSynchronize(c->s[0]));                                            // null stream does not order with it)
  PA_HIP(hipDeviceSynchronize());  // the context's streams are non-blocking: do not race with default-stream set-up
  *out = c;
  return PA_OK;
}

// Several contexts of one process on different GPUs (a DebugArray over several GPUs): the push kernels store into, and the copy
// transport copies between, buffers of other devices -- peer access both ways, enabled once per pair, failures ignored (a pair
// without a link keeps the staged copies of hipMemcpyPeer; the push transport then fails at its first store, loudly).
static void enable_peer_access(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return; }
  for (int d = 0; d < n; ++d) {
    if (d == device) continue;
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, device, d) == hipSuccess && can) {
      if (hipSetDevice(device) == hipSuccess) (void)hipDeviceEnablePeerAccess(d, 0);
      if (hipSetDevice(d) == hipSuccess) (void)hipDeviceEnablePeerAccess(device, 0);
    }
    (void)hipGetLastError();
  }
  (void)hipSetDevice(device);
}

extern "C" int pa_ctx_destroy(pa_ctx *c) {
  if (!c) return PA_OK;
  (void)hipSetDevice(c->device);
  (void)hipStreamSynchronize(c->s[0]);
  (void)hipStreamSynchronize(c->s[1]);
  (void)pa_raw_free(c->d_partials);
  (void)pa_raw_free(c->d_scalar);
  if (c->d_dotpart) pa_dev_free(c, c->d_dotpart);
  for (int k = 0; k < 2; ++k) if (c->d_xalpha[k]) pa_dev_free(c, c->d_xalpha[k]);
  if (c->d_vdict_scratch) (void)pa_raw_free(c->d_vdict_scratch);
  pa_arena_destroy(c);
  pa_scratch().trim();                 // (the set-up routes' cached temporaries: pa_scratch.h)
  (void)hipEventDestroy(c->ev_compute);
  (void)hipStreamDestroy(c->s[0]);
  (void)hipStreamDestroy(c->s[1]);
  delete c;
  return PA_OK;
}

extern "C" int pa_ctx_sync(pa_ctx *c) {
  PA_REQUIRE(c != nullptr, "ctx is NULL");
  PA_HIP(hipSetDevice(c->device));
  PA_HIP(hipStreamSynchronize(c->s[1]));
  PA_HIP(hipStreamSynchronize(c->s[0]));
  // a fused product whose tail gave up waiting for its exchange left boundary rows unsummed: said here, once, where a host would
  // read the result (1 -> 2: reported; the handle's next product drains, clears the word and continues with separate launches)
  int64_t lost = c->n_fused_timeouts;
  c->n_fused_timeouts = 0;
  for (int *w : c->fused_status)
    if (*(volatile int *)w == 1) { *(volatile int *)w = 2; ++lost; }
  if (lost) {
    pa_set_err("%lld fused product(s) gave up waiting for their RCCL receives (PA_IPC_TIMEOUT_S): their boundary rows were not summed; "
               "the handles continue with separate launches", (long long)lost);
    return PA_ERR_STATE;
  }
  return PA_OK;
}

extern "C" int pa_ctx_stream(pa_ctx *c, int which, void **s) {
  PA_REQUIRE(c && s && (which == 0 || which == 1), "bad arguments");
  *s = (void *)c->s[which];
  return PA_OK;
}

extern "C" int pa_ctx_stream_priority(pa_ctx *c, int which, int *priority, int *least, int *greatest) {
  PA_REQUIRE(c && (which == 0 || which == 1), "bad arguments");
  PA_HIP(hipSetDevice(c->device));
  int lo = 0, hi = 0, pr = 0;
  PA_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  PA_HIP(hipStreamGetPriority(c->s[which], &pr));
  if (priority) *priority = pr;
  if (least) *least = lo;
  if (greatest) *greatest = hi;
  return PA_OK;
}

extern "C" int pa_ctx_device_info(pa_ctx *c, int *cus, int *xcds, size_t *hbm, char *name, size_t name_len) {
  PA_REQUIRE(c != nullptr, "ctx is NULL");
  if (cus) *cus = c->cus;
  if (xcds) *xcds = c->xcds;
  if (hbm) *hbm = c->hbm;
  if (name && name_len) snprintf(name, name_len, "%s", c->name);
  return PA_OK;
}

// ------------------------------------------------------------------------------------------------
// events
// ------------------------------------------------------------------------------------------------
extern "C" int pa_event_create(pa_ctx *c, pa_event **ev) {
  PA_REQUIRE(c && ev, "bad arguments");
  PA_HIP(hipSetDevice(c->device));
  pa_event *e = new pa_event();
  e->ctx = c;
  // timing events: no system-scope fence when they complete (the default one writes back / invalidates the caches between
  // two launches: a product bracketed by default events ran 3 % slower than the same product queued back to back)
  PA_HIP(hipEventCreateWithFlags(&e->ev, hipEventDisableSystemFence));
  *ev = e;
  return PA_OK;
}
extern "C" int pa_event_destroy(pa_event *e) {
  if (!e) return PA_OK;
  (void)hipEventDestroy(e->ev);
  delete e;
  return PA_OK;
}
extern "C" int pa_event_record(pa_event *e, int which) {
  PA_REQUIRE(e && (which == 0 || which == 1), "bad arguments");
  PA_HIP(hipSetDevice(e->ctx->device));
  PA_HIP(hipEventRecord(e->ev, e->ctx->s[which]));
  return PA_OK;
}
extern "C" int pa_event_elapsed_ms(pa_event *a, pa_event *b, float *ms) {
  PA_REQUIRE(a && b && ms, "bad arguments");
  PA_HIP(hipEventSynchronize(b->ev));
  PA_HIP(hipEventElapsedTime(ms, a->ev, b->ev));
  return PA_OK;
}

// ------------------------------------------------------------------------------------------------
// vectors
// ------------------------------------------------------------------------------------------------

extern "C" int pa_vec_create(pa_ctx *c, int64_t n_own, int64_t n_ghost, pa_vec **out) {
  PA_REQUIRE(c && out && n_own >= 0 && n_ghost >= 0, "bad arguments");
  PA_HIP(hipSetDevice(c->device));
  pa_vec *v = new pa_vec();
  v->ctx = c; v->n_own = n_own; v->n_ghost = n_ghost; v->owned = true;
  const size_t bytes = sizeof(double) * (size_t)(n_own + n_ghost + 2);
  if (const int st = pa_dev_alloc(c, (void **)&v->d, bytes, PA_MEM_VECTOR)) { delete v; return st; }   // a memory class no matrix stream lives in (pa_arena.hip)
  if (hipMemsetAsync(v->d, 0, bytes, c->s[0]) != hipSuccess) {
    pa_set_err("hipMemsetAsync failed on a new vector of %lld values", (long long)(n_own + n_ghost));
    pa_dev_free(c, v->d);
    delete v;
    return PA_ERR_HIP;
  }
  *out = v;
  return PA_OK;
}

extern "C" int pa_vec_wrap(pa_ctx *c, void *ptr, int64_t n_own, int64_t n_ghost, pa_vec **out) {
  PA_REQUIRE(c && out && (ptr || n_own + n_ghost == 0) && n_own >= 0 && n_ghost >= 0, "bad arguments");
  pa_vec *v = new pa_vec();
  v->ctx = c; v->n_own = n_own; v->n_ghost = n_ghost; v->owned = false; v->d = (double *)ptr;
  *out = v;
  return PA_OK;
}

extern "C" int pa_vec_destroy(pa_vec *v) {
  if (!v) return PA_OK;
  if (v->owned) {
    (void)hipSetDevice(v->ctx->device);
    (void)hipStreamSynchronize(v->ctx->s[0]);
    (void)hipStreamSynchronize(v->ctx->s[1]);
    pa_dev_free(v->ctx, v->d);
  }
  delete v;
  return PA_OK;
}

extern "C" int pa_vec_sizes(const pa_vec *v, int64_t *n_own, int64_t *n_ghost) {
  PA_REQUIRE(v != nullptr, "vec is NULL");
  if (n_own) *n_own = v->n_own;
  if (n_ghost) *n_ghost = v->n_ghost;
  return PA_OK;
}
extern "C" int pa_vec_data(pa_vec *v, void **p) {
  PA_REQUIRE(v && p, "bad arguments");
  *p = v->d;
  return PA_OK;
}

extern "C" int pa_vec_upload(pa_vec *v, const double *host, int64_t off, int64_t len) {
  PA_REQUIRE(v && (host || len == 0), "bad arguments");
  PA_REQUIRE(off >= 0 && len >= 0 && off + len <= v->n_own + v->n_ghost, "range [%lld,+%lld) outside the local vector",
             (long long)off, (long long)len);
  if (len == 0) return PA_OK;
  PA_HIP(hipSetDevice(v->ctx->device));
  PA_HIP(hipMemcpyAsync(v->d + off, host, sizeof(double) * len, hipMemcpyHostToDevice, v->ctx->s[0]));
  PA_HIP(hipStreamSynchronize(v->ctx->s[0]));
  return PA_OK;
}

extern "C" int pa_vec_download(const pa_vec *v, double *host, int64_t off, int64_t len) {
  PA_REQUIRE(v && (host || len == 0), "bad arguments");
  PA_REQUIRE(off >= 0 && len >= 0 && off + len <= v->n_own + v->n_ghost, "range [%lld,+%lld) outside the local vector",
             (long long)off, (long long)len);
  PA_HIP(hipSetDevice(v->ctx->device));
  PA_HIP(hipStreamSynchronize(v->ctx->s[1]));
  if (len) PA_HIP(hipMemcpyAsync(host, v->d + off, sizeof(double) * len, hipMemcpyDeviceToHost, v->ctx->s[0]));
  PA_HIP(hipStreamSynchronize(v->ctx->s[0]));
  return PA_OK;
}


extern "C" int pa_vec_fill(pa_vec *v, int seg, double value) {
  PA_REQUIRE(v != nullptr, "vec is NULL");
  int64_t off, len;
  PA_TRY(seg_range(v, seg, &off, &len));
  if (len == 0) return PA_OK;
  PA_HIP(hipSetDevice(v->ctx->device));
  hipLaunchKernelGGL(k_fill, dim3(grid_for(len, 256)), dim3(256), 0, v->ctx->s[0], v->d + off, len, value);
  PA_HIP(hipGetLastError());
  return PA_OK;
}

extern "C" int pa_vec_copy(pa_vec *dst, const pa_vec *src, int seg) {
  PA_REQUIRE(dst && src, "bad arguments");
  // own values of vectors on different index partitions with matching own indices may be copied (w .= v2 in
  // assemble(v,rows), src/p_vector.jl:1331-1345); the other segments need identical layouts
  PA_REQUIRE(dst->n_own == src->n_own && (seg == PA_SEG_OWN || dst->n_ghost == src->n_ghost), "size mismatch");
  int64_t off, len;
  PA_TRY(seg_range(dst, seg, &off, &len));
  if (len == 0) return PA_OK;
  PA_HIP(hipSetDevice(dst->ctx->device));
  PA_HIP(hipMemcpyAsync(dst->d + off, src->d + off, sizeof(double) * len, hipMemcpyDeviceToDevice, dst->ctx->s[0]));
  return PA_OK;
}

extern "C" int pa_vec_axpby(pa_vec *y, double a, const pa_vec *x, double b, int seg) {
  PA_REQUIRE(y && x, "bad arguments");
  PA_REQUIRE(y->n_own == x->n_own && (seg == PA_SEG_OWN || y->n_ghost == x->n_ghost), "size mismatch");
  int64_t off, len;
  PA_TRY(seg_range(y, seg, &off, &len));
  if (len == 0) return PA_OK;
  PA_HIP(hipSetDevice(y->ctx->device));
  hipLaunchKernelGGL(k_axpby, dim3(grid_for(len, 256)), dim3(256), 0, y->ctx->s[0], y->d + off, x->d + off, len, a, b);
  PA_HIP(hipGetLastError());
  return PA_OK;
}

extern "C" int pa_vec_dot(const pa_vec *x, const pa_vec *y, double *host_out) {
  PA_REQUIRE(x && y, "bad arguments");
  PA_REQUIRE(x->n_own == y->n_own, "own-size mismatch (%lld vs %lld)", (long long)x->n_own, (long long)y->n_own);
  pa_ctx *c = x->ctx;
  PA_HIP(hipSetDevice(c->device));
  const int nb = grid_for(x->n_own, 256 * 8, c->n_partials);
  hipLaunchKernelGGL(k_dot_partial, dim3(nb), dim3(256), 0, c->s[0], x->d, y->d, x->n_own, c->d_partials);
  hipLaunchKernelGGL(k_dot_final, dim3(1), dim3(256), 0, c->s[0], c->d_partials, nb, c->d_scalar);
  PA_HIP(hipGetLastError());
  if (host_out) {
    PA_HIP(hipMemcpyAsync(host_out, c->d_scalar, sizeof(double), hipMemcpyDeviceToHost, c->s[0]));
    PA_HIP(hipStreamSynchronize(c->s[0]));
  }
  return PA_OK;
}

extern "C" int pa_ctx_read_scalar(pa_ctx *c, double *host_out) {
  PA_REQUIRE(c && host_out, "bad arguments");
  PA_HIP(hipSetDevice(c->device));
  PA_HIP(hipMemcpyAsync(host_out, c->d_scalar, sizeof(double), hipMemcpyDeviceToHost, c->s[0]));
  PA_HIP(hipStreamSynchronize(c->s[0]));
  return PA_OK;
}

// ---- device-resident solver scalars -------------------------------------------------------------

extern "C" int pa_vec_dot_slot(const pa_vec *x, const pa_vec *y, int slot, int accumulate) {
  PA_REQUIRE(x && y, "bad arguments");
  PA_REQUIRE(x->n_own == y->n_own, "own-size mismatch (%lld vs %lld)", (long long)x->n_own, (long long)y->n_own);
  PA_REQUIRE(PA_SLOT_OK(slot), "slot %d out of range [0,%d)", slot, PA_N_SLOTS);
  pa_ctx *c = x->ctx;
  PA_HIP(hipSetDevice(c->device));
  const int nb = grid_for(x->n_own, 256 * 8, c->n_partials);
  hipLaunchKernelGGL(k_dot_partial, dim3(nb), dim3(256), 0, c->s[0], x->d, y->d, x->n_own, c->d_partials);
  hipLaunchKernelGGL(k_dot_final_slot, dim3(1), dim3(256), 0, c->s[0], c->d_partials, nb, c->d_scalar + slot, accumulate);
  PA_HIP(hipGetLastError());
  return PA_OK;
}

extern "C" int pa_vec_axpby_slot(pa_vec *y, double ca, int a_num, int a_den, const pa_vec *x, double cb, int b_num,
                                 int b_den, int seg) {
  PA_REQUIRE(y && x, "bad arguments");
  PA_REQUIRE(y->n_own == x->n_own && (seg == PA_SEG_OWN || y->n_ghost == x->n_ghost), "size mismatch");
  PA_REQUIRE(PA_COEF_OK(a_num) && PA_COEF_OK(a_den) && PA_COEF_OK(b_num) && PA_COEF_OK(b_den), "slot out of range");
  int64_t off, len;
  PA_TRY(seg_range(y, seg, &off, &len));
  if (len == 0) return PA_OK;
  PA_HIP(hipSetDevice(y->ctx->device));
  hipLaunchKernelGGL(k_axpby_slot, dim3(grid_for(len, 256)), dim3(256), 0, y->ctx->s[0], y->d + off, x->d + off, len,
                     y->ctx->d_scalar, ca, a_num, a_den, cb, b_num, b_den);
  PA_HIP(hipGetLastError());
  return PA_OK;
}

extern "C" int pa_cg_update(pa_vec *x, pa_vec *r, const pa_vec *u, const pa_vec *cv, int num, int den, int rr_slot,
                            int accumulate) {
  PA_REQUIRE(x && r && u && cv, "bad arguments");
  PA_REQUIRE(x->n_own == r->n_own && x->n_own == u->n_own && x->n_own == cv->n_own, "own-size mismatch");
  PA_REQUIRE(PA_COEF_OK(num) && PA_COEF_OK(den) && PA_SLOT_OK(rr_slot), "slot out of range");
  PA_REQUIRE(rr_slot != num && rr_slot != den, "the result slot must differ from the coefficient slots");
  PA_REQUIRE(x->d != r->d && x->d != u->d && x->d != cv->d && r->d != u->d && r->d != cv->d, "x, r, u, c must be distinct vectors");
  pa_ctx *c = x->ctx;
  PA_HIP(hipSetDevice(c->device));
  const int nb = grid_for(x->n_own, 256 * 8, c->n_partials);
  hipLaunchKernelGGL(k_cg_update, dim3(nb), dim3(256), 0, c->s[0], x->d, r->d, u->d, cv->d, x->n_own, c->d_scalar, num,
                     den, c->d_partials);
  hipLaunchKernelGGL(k_dot_final_slot, dim3(1), dim3(256), 0, c->s[0], c->d_partials, nb, c->d_scalar + rr_slot, accumulate);
  PA_HIP(hipGetLastError());
  return PA_OK;
}

extern "C" int pa_cg_r_update(pa_vec *r, const pa_vec *cv, int num, int den, int rr_slot, int accumulate) {
  PA_REQUIRE(r && cv, "bad arguments");
  PA_REQUIRE(r->n_own == cv->n_own, "own-size mismatch");
  PA_REQUIRE(PA_COEF_OK(num) && PA_COEF_OK(den) && PA_SLOT_OK(rr_slot), "slot out of range");
  PA_REQUIRE(rr_slot != num && rr_slot != den, "the result slot must differ from the coefficient slots");
  PA_REQUIRE(r->d != cv->d, "r and c must be distinct vectors");
  pa_ctx *c = r->ctx;
  PA_HIP(hipSetDevice(c->device));
  const int nb = grid_for(r->n_own, 256 * 8, c->n_partials);
  hipLaunchKernelGGL(k_cg_r_update, dim3(nb), dim3(256), 0, c->s[0], r->d, cv->d, r->n_own, c->d_scalar, num, den, c->d_partials);
  hipLaunchKernelGGL(k_dot_final_slot, dim3(1), dim3(256), 0, c->s[0], c->d_partials, nb, c->d_scalar + rr_slot, accumulate);
  PA_HIP(hipGetLastError());
  return PA_OK;
}

extern "C" int pa_cg_xu_update(pa_vec *x, pa_vec *u, const pa_vec *z, int a_num, int a_den, int b_num, int b_den) {
  PA_REQUIRE(x && u && z, "bad arguments");
  PA_REQUIRE(x->n_own == u->n_own && x->n_own == z->n_own, "own-size mismatch");
  PA_REQUIRE(PA_COEF_OK(a_num) && PA_COEF_OK(a_den) && PA_COEF_OK(b_num) && PA_COEF_OK(b_den), "slot out of range");
  PA_REQUIRE(x->d != u->d && x->d != z->d && u->d != z->d, "x, u, z must be distinct vectors");
  if (x->n_own == 0) return PA_OK;
  pa_ctx *c = x->ctx;
  PA_HIP(hipSetDevice(c->device));
  hipLaunchKernelGGL(k_cg_xu_update, dim3(grid_for(x->n_own, 256)), dim3(256), 0, c->s[0], x->d, u->d, z->d, x->n_own, c->d_scalar,
                     a_num, a_den, b_num, b_den);
  PA_HIP(hipGetLastError());
  return PA_OK;
}

extern "C" int pa_ctx_slot_ptr(pa_ctx *c, int slot, void **p) {
  PA_REQUIRE(c && p, "bad arguments");
  PA_REQUIRE(PA_SLOT_OK(slot), "slot %d out of range [0,%d)", slot, PA_N_SLOTS);
  *p = c->d_scalar + slot;
  return PA_OK;
}

extern "C" int pa_ctx_write_slot(pa_ctx *c, int slot, double value) {
  PA_REQUIRE(c != nullptr, "ctx is NULL");
  PA_REQUIRE(PA_SLOT_OK(slot), "slot %d out of range [0,%d)", slot, PA_N_SLOTS);
  PA_HIP(hipSetDevice(c->device));
  hipLaunchKernelGGL(k_fill, dim3(1), dim3(64), 0, c->s[0], c->d_scalar + slot, (int64_t)1, value);
  PA_HIP(hipGetLastError());
  return PA_OK;
}

extern "C" int pa_ctx_read_slots(pa_ctx *c, int first, int n, double *host_out) {
  PA_REQUIRE(c && host_out, "bad arguments");
  PA_REQUIRE(first >= 0 && n >= 1 && first + n <= PA_N_SLOTS, "slots [%d,%d) out of range", first, first + n);
  PA_HIP(hipSetDevice(c->device));
  PA_HIP(hipMemcpyAsync(host_out, c->d_scalar + first, sizeof(double) * n, hipMemcpyDeviceToHost, c->s[0]));
  PA_HIP(hipStreamSynchronize(c->s[0]));
  return PA_OK;
}

extern "C" int pa_vec_dot_result(pa_ctx *c, void **p) {
  PA_REQUIRE(c && p, "bad arguments");
  *p = c->d_scalar;
  return PA_OK;
}

// ------------------------------------------------------------------------------------------------
// hipGraph capture of whatever the entry points queue (launch-bound loops: a CG iteration of a small part is
// ~10 kernels of a few microseconds each)
// ------------------------------------------------------------------------------------------------
extern "C" int pa_graph_begin(pa_ctx *c) {
  PA_REQUIRE(c != nullptr, "ctx is NULL");
  PA_REQUIRE(!c->capturing, "a capture is already open on this context");
  PA_HIP(hipSetDevice(c->device));
  PA_HIP(hipStreamBeginCapture(c->s[0], hipStreamCaptureModeThreadLocal));
  c->capturing = true;
  return PA_OK;
}

extern "C" int pa_graph_end(pa_ctx *c, pa_graph **out) {
  PA_REQUIRE(c && out, "bad arguments");
  PA_REQUIRE(c->capturing, "pa_graph_end without pa_graph_begin");
  c->capturing = false;
  hipGraph_t graph = nullptr;
  if (hipError_t e = hipStreamEndCapture(c->s[0], &graph)) {
    (void)hipGetLastError();                           // (the failed capture's error must not surface in the next, unrelated call)
    pa_set_err("hipStreamEndCapture failed: %s", hipGetErrorString(e));
    return PA_ERR_HIP;
  }
  hipGraphExec_t exec = nullptr;
  hipError_t e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) {
    pa_set_err("hipGraphInstantiate failed: %s", hipGetErrorString(e));
    return PA_ERR_HIP;
  }
  pa_graph *g = new pa_graph();
  g->ctx = c; g->exec = exec;
  *out = g;
  return PA_OK;
}

extern "C" int pa_graph_launch(pa_graph *g) {
  PA_REQUIRE(g != nullptr, "graph is NULL");
  PA_HIP(hipSetDevice(g->ctx->device));
  PA_HIP(hipGraphLaunch(g->exec, g->ctx->s[0]));
  return PA_OK;
}

extern "C" int pa_graph_destroy(pa_graph *g) {
  if (!g) return PA_OK;
  (void)hipSetDevice(g->ctx->device);
  (void)hipStreamSynchronize(g->ctx->s[0]);
  (void)hipGraphExecDestroy(g->exec);
  delete g;
  return PA_OK;
}
