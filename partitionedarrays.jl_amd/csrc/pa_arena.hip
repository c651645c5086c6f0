// pa_arena.hip -- where the big arrays of a context live in the 288 GB of HBM3E: one physically contiguous arena with a
// measured map of its memory classes, and the rule "a product's write stream never shares a class with its read stream".
//
// What was measured on MI355X (tools/probe/ystore_probe.hip, gpurun_out of round 2): device memory
// falls into THREE classes of about a third each, laid out in physically contiguous regions of 2 ... 96 GiB whose
// boundaries differ from box to box.  A kernel that streams reads from one class while it writes 64-byte lines into the
// SAME class loses 13-15 % (27-point 256^3 product: 0.765 ms against 0.670 ms; the kernel without its store: 0.63 ms);
// with the write stream in either of the other two classes the penalty is gone, wherever x lives.  A plain hipMalloc
// of a few GB may straddle classes (then no place for y is fast), so the value stream must be ONE contiguous piece of one
// class.  Neither virtual addresses nor allocation order predict the class; a 40 us stand-in kernel does (a 1 : 27
// write : read stream pair, the product's ratio) -- so the context maps its arena once (~0.2 s, at the first allocation
// of 256 MiB or more) and then places by rule, with no timing of the caller's kernels and no moving of vectors:
//     matrix streams (values, columns, row pointers, descriptors)  -> the class with the most room in the arena
//     vectors                                                        -> the two other classes, alternating
// A request that no run of its preferred classes holds takes another class (matrix streams the one with fewer vectors,
// vectors the one with fewer matrix streams), then plain hipMalloc.  Classes are numbered in the order the map meets them.
// Replaces round 1's pa_csr_tune_placement (a search over hipMalloc'ed copies that found a fast pair on two boxes of three).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <mutex>
#include <unordered_map>
#include <vector>

#include "pa_internal.h"

typedef double pa_d2 __attribute__((ext_vector_type(2)));

// block b streams 12 KiB of `rd` and writes 56 doubles to `wr`: the 27-point product's traffic without a matrix behind it
__global__ __launch_bounds__(256) void k_class_probe(const pa_d2 *__restrict__ rd, int n_blocks, int per_xcd,
                                                     double *__restrict__ wr) {
  const int b = blockIdx.x;
  const int blk = (b & 7) * per_xcd + (b >> 3);          // dealt to the XCDs like the product's chunks
  if (blk >= n_blocks || (b >> 3) >= per_xcd) return;
  const pa_d2 *p = rd + (size_t)blk * 768;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const pa_d2 v = __builtin_nontemporal_load(p + k * 256 + threadIdx.x);
    s += v.x + v.y;
  }
  if (threadIdx.x < 56) __builtin_nontemporal_store(s, wr + (size_t)blk * 56 + threadIdx.x);
  else if (s == 123.456) wr[(size_t)blk * 56] = s;       // keeps every lane's loads alive
}

struct pa_arena {
  char *base = nullptr;
  size_t size = 0, cell = 0;
  std::vector<int8_t> cls;                 // per cell: 0, 1, 2, or -1 (a class boundary runs through it: not handed out)
  int n_classes = 1;
  double map_ms = 0;
  size_t class_bytes[3] = {0, 0, 0};       // usable bytes by class
  struct blk { size_t len; int cls; };
  std::map<size_t, blk> free_;             // offset -> free block (never spans a class change)
  std::map<size_t, blk> live_;             // offset -> allocated block
  std::map<size_t, int> live_kind_;        // offset -> PA_MEM_MATRIX / PA_MEM_VECTOR
  size_t used = 0, peak = 0;
  int matrix_class = 0;                    // the class with the most room: where matrix streams go
  size_t mat_bytes[3] = {0, 0, 0}, vec_bytes[3] = {0, 0, 0};   // what lives where (by kind)
  unsigned vec_turn = 0;                   // vectors alternate between the two other classes while both are free of matrix streams
};

static constexpr size_t ARENA_ALIGN = (size_t)256 << 10;

static int probe_ms(pa_ctx *c, hipEvent_t e0, hipEvent_t e1, const char *rd, char *wr, int nb, float *ms) {
  const int per_xcd = (nb + 7) / 8;
  hipLaunchKernelGGL(k_class_probe, dim3(per_xcd * 8), dim3(256), 0, c->s[0], (const pa_d2 *)rd, nb, per_xcd, (double *)wr);
  PA_HIP(hipEventRecord(e0, c->s[0]));
  for (int r = 0; r < 3; ++r)
    hipLaunchKernelGGL(k_class_probe, dim3(per_xcd * 8), dim3(256), 0, c->s[0], (const pa_d2 *)rd, nb, per_xcd, (double *)wr);
  PA_HIP(hipEventRecord(e1, c->s[0]));
  PA_HIP(hipEventSynchronize(e1));
  PA_HIP(hipGetLastError());
  PA_HIP(hipEventElapsedTime(ms, e0, e1));
  *ms /= 3;
  return PA_OK;
}

// One pass: the read stream sits at `rd`, the write stream at the END of every cell listed in `cells`.  slow[c] = the end
// of cell c is in the read stream's class.  Returns false when the times do not separate (one class, or no signal).
static int class_pass(pa_ctx *c, pa_arena *a, hipEvent_t e0, hipEvent_t e1, const char *rd, int nb, size_t wr_bytes,
                      const std::vector<int> &cells, std::vector<char> &slow, bool *separated) {
  std::vector<float> t(cells.size());
  for (size_t i = 0; i < cells.size(); ++i)
    PA_TRY(probe_ms(c, e0, e1, rd, a->base + (size_t)(cells[i] + 1) * a->cell - wr_bytes, nb, &t[i]));
  // Two clusters about 11 % apart (other class / same class) are what the hardware gives; anything far above that is the
  // same-class case plus interference from whoever else uses the device, so the slow end is capped at 1.3 x the fastest.
  float mn = 1e30f, mx = 0;
  for (float v : t) { mn = std::min(mn, v); mx = std::max(mx, v); }
  mx = std::min(mx, 1.3f * mn);
  *separated = !cells.empty() && mx > 1.06f * mn;
  if (!*separated) return PA_OK;
  const float thr = 0.5f * (mn + mx);
  for (size_t i = 0; i < cells.size(); ++i) {
    float v = t[i];
    for (int again = 0; again < 2 && v > 0.97f * thr && v < 1.03f * thr; ++again) {     // too close to call: measure again
      float w = 0;
      PA_TRY(probe_ms(c, e0, e1, rd, a->base + (size_t)(cells[i] + 1) * a->cell - wr_bytes, nb, &w));
      v = 0.5f * (v + w);
    }
    slow[cells[i]] = v > thr;
  }
  return PA_OK;
}

static int arena_build(pa_ctx *c) {
  if (c->capturing) return PA_OK;             // (not now: mapping the classes launches and synchronises; the next request tries again)
  c->arena_tried = true;
  const char *on = getenv("PA_ARENA");
  if (on && atoi(on) == 0) return PA_OK;
  PA_HIP(hipSetDevice(c->device));
  const size_t G = (size_t)1 << 30;
  size_t fr = 0, tot = 0;
  PA_HIP(hipMemGetInfo(&fr, &tot));
  double frac = 0.70;
  if (const char *e = getenv("PA_ARENA_FRACTION")) frac = std::min(0.95, std::max(0.05, atof(e)));
  size_t want = (size_t)(frac * (double)fr);
  if (const char *e = getenv("PA_ARENA_GIB")) want = std::min<size_t>((size_t)atol(e) * G, (size_t)(0.95 * (double)fr));
  const size_t cell = (size_t)512 << 20;
  want = want / cell * cell;
  hipEvent_t e0, e1;
  PA_HIP(hipEventCreate(&e0));
  PA_HIP(hipEventCreate(&e1));
  char *base = nullptr;
  while (want >= 8 * G) {       // physically contiguous: positions inside it are physical offsets, classes are regions
    if (hipExtMallocWithFlags((void **)&base, want, hipDeviceMallocContiguous) == hipSuccess) break;
    (void)hipGetLastError();
    base = nullptr;
    want = (want * 3 / 4) / cell * cell;
  }
  if (!base) {                  // no arena: every request falls through to hipMalloc
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return PA_OK;
  }
  pa_arena *a = new pa_arena();
  a->base = base; a->size = want; a->cell = cell;
  const int ncell = (int)(want / cell);
  a->cls.assign(ncell, 0);
  const auto t0 = std::chrono::steady_clock::now();
  const size_t rd_bytes = cell;                               // (twice the Infinity Cache: the stream comes from HBM)
  const int nb = (int)(rd_bytes / 12288);
  const size_t wr_bytes = ((size_t)nb * 56 * 8 + 4095) / 4096 * 4096;
  // end[c] = class at the end of cell c.  Pass 0: read stream in cell 0.
  std::vector<int> endc(ncell, 0), all(ncell);
  for (int i = 0; i < ncell; ++i) all[i] = i;
  std::vector<char> slow0(ncell, 1), slow1(ncell, 0);
  bool sep0 = false, sep1 = false;
  int st = class_pass(c, a, e0, e1, base, nb, wr_bytes, all, slow0, &sep0);
  if (st == PA_OK && sep0) {
    // Pass 1: read stream in the first cell whose both ends are outside class 0; splits the rest into classes 1 and 2
    std::vector<int> rest;
    for (int i = 0; i < ncell; ++i) if (!slow0[i]) rest.push_back(i);
    for (int tries = 0, from = 1; tries < 3 && st == PA_OK && !sep1; ++tries) {
      int ref = -1;
      for (int i = from; i < ncell; ++i) if (!slow0[i - 1] && !slow0[i]) { ref = i; break; }
      if (ref < 0) break;
      std::fill(slow1.begin(), slow1.end(), 0);
      bool sep = false;
      st = class_pass(c, a, e0, e1, base + (size_t)ref * cell, nb, wr_bytes, rest, slow1, &sep);
      if (st != PA_OK) break;
      if (!sep) {                                     // nothing stands out against the reference: the rest is ONE class
        for (int i : rest) slow1[i] = 1;
        sep1 = true;
      } else if (slow1[ref - 1] && slow1[ref]) {      // (the reference cell itself must come out as "same class")
        sep1 = true;
      } else {
        from = ref + 1;
      }
    }
    if (!sep1) for (int i : rest) slow1[i] = 1;
    sep1 = true;
    for (int i = 0; i < ncell; ++i) endc[i] = slow0[i] ? 0 : (sep1 ? (slow1[i] ? 1 : 2) : 1);
    a->n_classes = 1 + (rest.empty() ? 0 : 1);
    for (int i : rest) if (endc[i] == 2) { a->n_classes = 3; break; }
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  if (st != PA_OK) { (void)hipFree(base); delete a; return st; }
  // a cell is handed out when both of its ends are in one class
  for (int i = 0; i < ncell; ++i) a->cls[i] = (int8_t)((i == 0 ? endc[0] : endc[i - 1]) == endc[i] ? endc[i] : -1);
  if (endc[0] != 0) a->cls[0] = -1;
  for (int i = 0; i < ncell;) {
    int j = i;
    while (j < ncell && a->cls[j] == a->cls[i]) ++j;
    if (a->cls[i] >= 0) {
      a->free_[(size_t)i * cell] = {(size_t)(j - i) * cell, a->cls[i]};
      a->class_bytes[a->cls[i]] += (size_t)(j - i) * cell;
    }
    i = j;
  }
  for (int k = 1; k < 3; ++k)
    if (a->class_bytes[k] > a->class_bytes[a->matrix_class]) a->matrix_class = k;
  a->map_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  c->arena = a;
  if (getenv("PA_SETUP_TIMING")) {
    fprintf(stderr, "[pa arena] %.1f GiB contiguous at %p, %d classes (%.1f / %.1f / %.1f GiB usable), mapped in %.1f ms:", want / (double)G,
            (void *)base, a->n_classes, a->class_bytes[0] / (double)G, a->class_bytes[1] / (double)G, a->class_bytes[2] / (double)G, a->map_ms);
    for (int i = 0; i < ncell; ++i) fputc(a->cls[i] < 0 ? '.' : (char)('0' + a->cls[i]), stderr);
    fputc('\n', stderr);
  }
  return PA_OK;
}

static void *arena_take(pa_arena *a, size_t bytes, int cls) {
  bytes = (bytes + ARENA_ALIGN - 1) / ARENA_ALIGN * ARENA_ALIGN;
  for (auto it = a->free_.begin(); it != a->free_.end(); ++it) {
    if (it->second.cls != cls || it->second.len < bytes) continue;
    const size_t off = it->first, len = it->second.len;
    a->free_.erase(it);
    if (len > bytes) a->free_[off + bytes] = {len - bytes, cls};
    a->live_[off] = {bytes, cls};
    a->used += bytes;
    a->peak = std::max(a->peak, a->used);
    return a->base + off;
  }
  return nullptr;
}

// ---- PA_DEBUG_GUARD: one mapping per buffer, the buffer flush with its end, nothing mapped behind it ----
static int guard_level() {                                  // 0 off, 1 guarded mappings, 2 + every new buffer filled with 0xFF bytes
  static int g = -1;
  if (g < 0) { const char *e = getenv("PA_DEBUG_GUARD"); g = e ? atoi(e) : 0; }
  return g;
}
static bool guard_mode() { return guard_level() >= 1; }
struct guard_rec { void *va; size_t total, mapped; hipMemGenericAllocationHandle_t h; };
static std::mutex g_guard_mu;
static std::unordered_map<void *, guard_rec> g_guard_live;

hipError_t pa_raw_malloc_impl(void **p, size_t bytes) {
  if (!guard_mode()) return hipMalloc(p, bytes);
  int dev = 0;
  hipError_t st = hipGetDevice(&dev);
  if (st != hipSuccess) return st;
  hipMemAllocationProp prop = {};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  if ((st = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum)) != hipSuccess) return st;
  if (bytes == 0) bytes = 8;
  guard_rec r;
  r.mapped = (bytes + gran - 1) / gran * gran;
  r.total = r.mapped + gran;                               // the granule behind the mapping stays unmapped
  if ((st = hipMemAddressReserve(&r.va, r.total, gran, nullptr, 0)) != hipSuccess) return st;
  if ((st = hipMemCreate(&r.h, r.mapped, &prop, 0)) != hipSuccess) { (void)hipMemAddressFree(r.va, r.total); return st; }
  if ((st = hipMemMap(r.va, r.mapped, 0, r.h, 0)) != hipSuccess) { (void)hipMemRelease(r.h); (void)hipMemAddressFree(r.va, r.total); return st; }
  hipMemAccessDesc acc = {};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  if ((st = hipMemSetAccess(r.va, r.mapped, &acc, 1)) != hipSuccess) return st;
  if (guard_level() >= 2) {                                  // (a buffer the library forgets to initialise then reads as NaNs)
    (void)hipMemset(r.va, 0xFF, r.mapped);
    (void)hipDeviceSynchronize();
  }
  *p = (char *)r.va + (r.mapped - (bytes + 15) / 16 * 16);    // 16-byte aligned, ends within 16 bytes of the mapping's end
  std::lock_guard<std::mutex> lk(g_guard_mu);
  g_guard_live[*p] = r;
  return hipSuccess;
}

hipError_t pa_raw_free(void *p) {
  if (!p) return hipSuccess;
  if (guard_mode()) {
    std::lock_guard<std::mutex> lk(g_guard_mu);
    auto it = g_guard_live.find(p);
    if (it != g_guard_live.end()) {
      const guard_rec r = it->second;
      g_guard_live.erase(it);
      (void)hipDeviceSynchronize();
      // The mapping is NOT handed back: unmapping and re-using the physical pages gave wrong products and faults that
      // disappear when nothing is ever unmapped (stale lines of the old owner written back over the new one's data, as far as
      // could be told) -- an artefact of this debugging allocator, not of the library.  Guarded runs are short; they leak.
      (void)r;
      return hipSuccess;
    }
  }
  return hipFree(p);
}

int pa_dev_alloc(pa_ctx *c, void **p, size_t bytes, int kind) {
  *p = nullptr;
  if (bytes == 0) bytes = 8;
  if (guard_mode()) {                                      // no arena: every buffer gets its own guarded mapping
    PA_HIP(pa_raw_malloc_impl(p, bytes));
    return PA_OK;
  }
  const size_t small = (size_t)1 << 20;          // below 1 MiB the class of a buffer does not matter
  size_t first_big = (size_t)256 << 20;          // the arena is built when the first allocation this large arrives
  if (const char *e = getenv("PA_ARENA_MIN_MIB")) first_big = (size_t)atol(e) << 20;
  if (kind != PA_MEM_PLAIN && bytes >= small) {
    if (!c->arena && !c->arena_tried && bytes >= first_big) PA_TRY(arena_build(c));
    if (pa_arena *a = c->arena) {
      // The rule: a vector never shares a class with a matrix stream.  Matrix streams fill the roomiest class first and
      // spill into the class that holds fewer vectors; vectors take the other classes -- alternating while both are
      // free of matrix streams (BLAS-1 kernels, too, run 3-6 % faster when what they write is not where they read) --
      // and end up next to matrix streams only when nothing else is left.
      const int M = a->matrix_class, o1 = (M + 1) % 3, o2 = (M + 2) % 3;
      int order[3];
      if (kind == PA_MEM_MATRIX) {
        order[0] = M;
        order[1] = a->vec_bytes[o1] <= a->vec_bytes[o2] ? o1 : o2;
        order[2] = order[1] == o1 ? o2 : o1;
      } else {
        int first = a->mat_bytes[o1] < a->mat_bytes[o2] ? o1 : a->mat_bytes[o2] < a->mat_bytes[o1] ? o2 : ((a->vec_turn++ & 1) ? o2 : o1);
        order[0] = first;
        order[1] = first == o1 ? o2 : o1;
        order[2] = M;
      }
      for (int k = 0; k < 3; ++k)
        if ((*p = arena_take(a, bytes, order[k])) != nullptr) {
          (kind == PA_MEM_MATRIX ? a->mat_bytes : a->vec_bytes)[order[k]] += (bytes + ARENA_ALIGN - 1) / ARENA_ALIGN * ARENA_ALIGN;
          a->live_kind_[(size_t)((char *)*p - a->base)] = kind;
          return PA_OK;
        }
    }
  }
  PA_HIP(hipMalloc(p, bytes));
  return PA_OK;
}

void pa_dev_free(pa_ctx *c, void *p) {
  if (!p) return;
  if (guard_mode()) { (void)pa_raw_free(p); return; }
  pa_arena *a = c ? c->arena : nullptr;
  if (a && (char *)p >= a->base && (char *)p < a->base + a->size) {
    const size_t off = (size_t)((char *)p - a->base);
    auto it = a->live_.find(off);
    if (it == a->live_.end()) return;            // not ours (cannot happen for pointers pa_dev_alloc handed out)
    size_t len = it->second.len;
    const int cls = it->second.cls;
    a->live_.erase(it);
    a->used -= len;
    auto kd = a->live_kind_.find(off);
    if (kd != a->live_kind_.end()) {
      size_t *acct = kd->second == PA_MEM_MATRIX ? a->mat_bytes : a->vec_bytes;
      acct[cls] -= std::min(acct[cls], len);
      a->live_kind_.erase(kd);
    }
    size_t start = off;
    auto nx = a->free_.find(off + len);           // merge with free neighbours of the same class (never across a boundary cell)
    if (nx != a->free_.end() && nx->second.cls == cls && a->cls[(off + len) / a->cell] == cls && a->cls[(off + len - 1) / a->cell] == cls) {
      len += nx->second.len;
      a->free_.erase(nx);
    }
    auto pv = a->free_.lower_bound(off);
    if (pv != a->free_.begin()) {
      --pv;
      if (pv->first + pv->second.len == off && pv->second.cls == cls && a->cls[(off - 1) / a->cell] == cls) {
        start = pv->first;
        len += pv->second.len;
        a->free_.erase(pv);
      }
    }
    a->free_[start] = {len, cls};
    return;
  }
  (void)hipFree(p);
}

int pa_mem_class(const pa_ctx *c, const void *p) {
  const pa_arena *a = c ? c->arena : nullptr;
  if (!a || !p || (const char *)p < a->base || (const char *)p >= a->base + a->size) return -1;
  return a->cls[(size_t)((const char *)p - a->base) / a->cell];
}

void pa_arena_destroy(pa_ctx *c) {
  if (!c || !c->arena) return;
  (void)hipFree(c->arena->base);
  delete c->arena;
  c->arena = nullptr;
}

extern "C" int pa_ctx_arena_info(pa_ctx *c, int64_t *bytes, int *n_classes, int64_t class_bytes[3], int64_t *used,
                                 double *map_ms, int *matrix_class) {
  PA_REQUIRE(c != nullptr, "ctx is NULL");
  const pa_arena *a = c->arena;
  if (bytes) *bytes = a ? (int64_t)a->size : 0;
  if (n_classes) *n_classes = a ? a->n_classes : 0;
  if (class_bytes) for (int k = 0; k < 3; ++k) class_bytes[k] = a ? (int64_t)a->class_bytes[k] : 0;
  if (used) *used = a ? (int64_t)a->used : 0;
  if (map_ms) *map_ms = a ? a->map_ms : 0.0;
  if (matrix_class) *matrix_class = a ? a->matrix_class : -1;
  return PA_OK;
}

extern "C" int pa_ctx_arena_map(pa_ctx *c, int64_t *cell_bytes, int8_t *classes, int64_t capacity, int64_t *n_cells) {
  PA_REQUIRE(c && n_cells, "bad arguments");
  const pa_arena *a = c->arena;
  *n_cells = a ? (int64_t)a->cls.size() : 0;
  if (cell_bytes) *cell_bytes = a ? (int64_t)a->cell : 0;
  if (a && classes) for (int64_t i = 0; i < std::min<int64_t>(capacity, *n_cells); ++i) classes[i] = a->cls[i];
  return PA_OK;
}

extern "C" int pa_ctx_arena_build(pa_ctx *c) {
  PA_REQUIRE(c != nullptr, "ctx is NULL");
  if (c->arena || c->arena_tried) return PA_OK;
  return arena_build(c);
}
