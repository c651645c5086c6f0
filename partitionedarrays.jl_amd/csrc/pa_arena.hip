// pa_arena.hip -- where the big arrays of a context live in the 288 GB of HBM3E: physically contiguous extents acquired ON
// DEMAND, each with a measured map of its memory classes, and the rule "a product's write stream never shares a class
// with its read stream".
//
// What was measured on MI355X (tools/probe/ystore_probe.hip, extent_probe.hip): device memory falls into THREE
// classes of about a third each, laid out in physically contiguous regions of tens of GiB.  A kernel that streams reads
// from one class while it writes 64-byte lines into the SAME class loses 13-15 % (27-point 256^3 product: 0.765 ms
// against 0.670 ms; without its store 0.63 ms); with the write stream in either other class the penalty is gone, wherever
// x lives.  A plain hipMalloc of a few GB may straddle classes, so the value stream must be ONE contiguous piece of one
// class.  Neither virtual addresses nor allocation order predict the class; a 40 us stand-in kernel does (a 1 : 27
// write : read stream pair, the product's ratio).
//
// Round 3: the single grab of 70 % of the free memory (7 s, hostile to anything else on the
// device) is gone.  The arena is a list of EXTENTS (16 GiB by default, PA_ARENA_EXTENT_GIB; a bigger request gets an extent
// of its own), each acquired when a class runs out of room and classified at once against the reference cell of every
// class met so far.  Placement by rule, no timing of the caller's kernels, nothing ever moves:
//     matrix streams (values, columns, row pointers, descriptors) -> the class the first one landed in
//     vectors                                                      -> anywhere else: a plain allocation the pair check
//                                                                     finds clear of that class (the driver serves plain
//                                                                     and contiguous requests from different ends of the
//                                                                     memory), else a class that holds no matrix stream
// When neither is at hand the arena WALKS: it acquires extents one after the other until one shows another class,
// keeps that one and hands the ones it walked over back to the driver at once (transient; bounded by PA_ARENA_WALK_GIB =
// 160 and by the budget PA_ARENA_FRACTION = 0.70 of the free memory / PA_ARENA_GIB).  An extent nothing lives in any more
// is released (PA_ARENA_SPARE = n keeps the newest n: re-allocating memory that has been used costs a driver-side wipe of
// 30-75 ms per GiB).  The vectors' extents, and the steps of their walk, are 4 GiB (or 8 x the request): the extent the walk
// ends in stays.  A big vector handed out is checked against the matrix streams' class with the same stand-in kernel
// (~3 ms; once per cell it touches -- a solver that allocates its work vectors on every call pays nothing after the first;
// PA_ARENA_SELFCHECK=0 disables): a pair that times as "same class" although the map says otherwise is moved to the other
// clean class or reported.  All of it under the context's mutex; any failure (no contiguous memory, a
// probe error) freezes growth and falls back to hipMalloc -- never an error of the caller's allocation.
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
#include "pa_scratch.h"

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

// One physically contiguous piece of the arena, acquired on demand and classified when acquired.
struct pa_extent {
  char *base = nullptr;
  size_t size = 0;
  std::vector<int8_t> cls;                 // per cell: global class 0..2, or -1 (a boundary runs through it / not told: not handed out)
  size_t live = 0;                         // bytes handed out from it (a class's scratch counts)
  std::vector<uint8_t> clear_of;           // per cell: bit m set = a vector here passed the pair check against matrix class m
};

struct pa_arena {
  size_t cell = (size_t)512 << 20;
  std::vector<pa_extent *> ext;
  int n_classes = 0;                       // global classes met so far (<= 3), numbered in the order they were met
  const char *ref[3] = {nullptr, nullptr, nullptr};   // a cell wholly inside class k: the read stream of a classification pass
  char *scr[3] = {nullptr, nullptr, nullptr};         // the tail of that cell, never handed out: a write stream KNOWN to be in class k
  double map_ms = 0;                       // time spent acquiring and classifying, so far
  size_t budget = 0;                       // bytes the extents together may hold
  size_t held = 0, acquired = 0, released = 0;
  int n_acquired = 0, n_released = 0;
  bool frozen = false;                     // an acquisition or a classification failed: no more growth (what exists keeps serving)
  struct blk { size_t len; int cls; int kind; pa_extent *e; };
  std::map<uintptr_t, blk> free_;          // address -> free block (never spans a class change or two extents)
  std::map<uintptr_t, blk> live_;          // address -> allocated block
  size_t used = 0, peak = 0;
  int matrix_class = -1;                   // where matrix streams go: the class the first one landed in
  size_t mat_bytes[3] = {0, 0, 0}, vec_bytes[3] = {0, 0, 0};   // what lives where (by kind)
  unsigned vec_turn = 0;
  int last_matrix_cls = -1;                // class of the newest big matrix stream handed out (what vectors stay away from; it
                                           // outlives the stream: a block's set-up frees temporaries bigger than anything it keeps)
  bool release_idle = false;               // pa_ctx_arena_release: an idle arena hands everything back, spare or not
  bool walking = false;                    // a walk is under way: what it went over is held until it has ended (arena_trim waits)
  int want_vec_classes = 1;                // pa_ctx_arena_hint: 2 = a solver's vectors alternate between two classes of their own
  bool second_walk_done = false;
  long check_ok = 0, check_failed = 0;     // vectors whose (matrix stream, vector) pair timed as "different classes" / "same class"
  std::map<uintptr_t, size_t> foreign_;    // plain hipMalloc'ed vectors the pair check found clear of the matrix streams' class
  size_t foreign_bytes = 0;
  long plain_rejected = 0;
  bool warned = false;
};

static constexpr size_t ARENA_ALIGN = (size_t)256 << 10;
static constexpr size_t GIB = (size_t)1 << 30;

static int probe_nb(size_t rd_bytes) { return (int)(rd_bytes / 12288); }
static size_t probe_wr_bytes(int nb) { return ((size_t)nb * 56 * 8 + 4095) / 4096 * 4096; }

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

struct probe_events {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int make() {
    PA_HIP(hipEventCreate(&e0));
    PA_HIP(hipEventCreate(&e1));
    return PA_OK;
  }
  ~probe_events() {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
  }
};

// One pass over the cell ENDS listed in `cells` of extent X: the read stream sits at `rd`; `slow_ctl` is a write position
// known to lie in rd's class, `fast_ctl` (or NULL) one known to lie in another.  same[c] = the end of cell c is in rd's
// class.  The two clusters are ~11 % apart (0.182 / 0.164 ms); the controls, measured before and after the pass, fix the
// threshold -- a pass over cells that are ALL in one class (nothing separates) is decided by them too.
static int class_pass(pa_ctx *c, pa_arena *a, probe_events &ev, const pa_extent *X, const char *rd, char *slow_ctl, char *fast_ctl,
                      const std::vector<int> &cells, std::vector<char> &same) {
  const int nb = probe_nb(a->cell);
  const size_t wr_bytes = probe_wr_bytes(nb);
  auto wr_of = [&](int cell) { return X->base + (size_t)(cell + 1) * a->cell - wr_bytes; };
  float s0 = 0, s1 = 0, f0 = 0, f1 = 0, warm = 0;
  for (int k = 0; k < 4; ++k) PA_TRY(probe_ms(c, ev.e0, ev.e1, rd, slow_ctl, nb, &warm));     // (working clocks)
  PA_TRY(probe_ms(c, ev.e0, ev.e1, rd, slow_ctl, nb, &s0));
  if (fast_ctl) PA_TRY(probe_ms(c, ev.e0, ev.e1, rd, fast_ctl, nb, &f0));
  std::vector<float> t(cells.size());
  for (size_t i = 0; i < cells.size(); ++i) PA_TRY(probe_ms(c, ev.e0, ev.e1, rd, wr_of(cells[i]), nb, &t[i]));
  PA_TRY(probe_ms(c, ev.e0, ev.e1, rd, slow_ctl, nb, &s1));
  if (fast_ctl) PA_TRY(probe_ms(c, ev.e0, ev.e1, rd, fast_ctl, nb, &f1));
  const float slow = 0.5f * (s0 + s1);
  float fast = fast_ctl ? 0.5f * (f0 + f1) : 0.9f * slow;
  if (fast > 0.96f * slow) fast = 0.9f * slow;          // (the "fast" control did not separate: fall back on the known ratio)
  const float thr = 0.5f * (slow + fast);
  for (size_t i = 0; i < cells.size(); ++i) {
    float v = t[i];
    // Interference only ever makes a probe SLOWER (tools/probe/extent_probe.hip: about one cell in 60 of a pass shows a
    // spike of the other cluster's size), so a "same class" verdict is confirmed by a second measurement and the smaller of
    // the two counts; a value too close to call is measured again and averaged.
    if (v > thr) {
      float w = 0;
      PA_TRY(probe_ms(c, ev.e0, ev.e1, rd, wr_of(cells[i]), nb, &w));
      v = std::min(v, w);
    }
    for (int again = 0; again < 2 && v > 0.98f * thr && v < 1.02f * thr; ++again) {
      float w = 0;
      PA_TRY(probe_ms(c, ev.e0, ev.e1, rd, wr_of(cells[i]), nb, &w));
      v = 0.5f * (v + w);
    }
    same[cells[i]] = v > thr;
  }
  return PA_OK;
}

static void arena_add_free(pa_arena *a, pa_extent *X, size_t off, size_t len, int cls) {
  if (len) a->free_[(uintptr_t)(X->base + off)] = {len, cls, 0, X};
}

// Classes of a freshly acquired extent (nothing of it is handed out yet, so the passes may write into it): first against
// the reference cell of every class met so far; what matches none is a class not met before -- a cell of it becomes that
// class's reference, its tail the class's scratch.  A cell whose two ends disagree is not handed out.
static int classify_extent(pa_ctx *c, pa_arena *a, pa_extent *X) {
  probe_events ev;
  PA_TRY(ev.make());
  const int ncell = (int)(X->size / a->cell);
  const int nb = probe_nb(a->cell);
  const size_t wr_bytes = probe_wr_bytes(nb);
  const size_t scr_bytes = (wr_bytes + ARENA_ALIGN - 1) / ARENA_ALIGN * ARENA_ALIGN;
  std::vector<int> endc(ncell, -2);                     // class at the END of each cell; -2: not told yet
  auto unknown = [&]() {
    std::vector<int> u;
    for (int i = 0; i < ncell; ++i) if (endc[i] == -2) u.push_back(i);
    return u;
  };
  for (int k = 0; k < a->n_classes; ++k) {
    std::vector<int> u = unknown();
    if (u.empty()) break;
    std::vector<char> same(ncell, 0);
    char *fast_ctl = nullptr;
    for (int j = 0; j < a->n_classes; ++j) if (j != k) { fast_ctl = a->scr[j]; break; }
    PA_TRY(class_pass(c, a, ev, X, a->ref[k], a->scr[k], fast_ctl, u, same));
    for (int i : u) if (same[i]) endc[i] = k;
  }
  int ref_cell[3] = {-1, -1, -1};
  for (int from = 1; a->n_classes < 3;) {
    // a cell whose both ends are still untold lies wholly in a class not met yet (one boundary per cell at most)
    int r = -1;
    for (int i = from; i < ncell; ++i) if (endc[i - 1] == -2 && endc[i] == -2) { r = i; break; }
    if (r < 0) break;
    std::vector<int> u = unknown();
    std::vector<char> same(ncell, 0);
    char *own_tail = X->base + (size_t)(r + 1) * a->cell - wr_bytes;
    PA_TRY(class_pass(c, a, ev, X, X->base + (size_t)r * a->cell, own_tail, a->n_classes ? a->scr[0] : nullptr, u, same));
    if (!(same[r - 1] && same[r])) { from = r + 1; continue; }     // (the cell straddles something after all: try the next one)
    const int k = a->n_classes++;
    for (int i : u) if (same[i]) endc[i] = k;
    a->ref[k] = X->base + (size_t)r * a->cell;
    a->scr[k] = X->base + (size_t)(r + 1) * a->cell - scr_bytes;
    ref_cell[k] = r;
    from = 1;
  }
  X->cls.assign(ncell, -1);
  for (int i = 0; i < ncell; ++i) {
    const int e = endc[i] < 0 ? -1 : endc[i];
    X->cls[i] = (int8_t)((i == 0 || endc[i - 1] == endc[i]) ? e : -1);
  }
  // free runs of one class; the scratch at the tail of a reference cell stays out (it counts as live: the extent is kept)
  for (int i = 0; i < ncell;) {
    int j = i;
    while (j < ncell && X->cls[j] == X->cls[i]) ++j;
    if (X->cls[i] >= 0) {
      size_t lo = (size_t)i * a->cell;
      const size_t hi = (size_t)j * a->cell;
      for (int k = 0; k < 3; ++k)
        if (ref_cell[k] >= i && ref_cell[k] < j) {
          const size_t s0 = (size_t)(ref_cell[k] + 1) * a->cell - scr_bytes;
          arena_add_free(a, X, lo, s0 - lo, X->cls[i]);
          lo = s0 + scr_bytes;
          X->live += scr_bytes;
        }
      arena_add_free(a, X, lo, hi - lo, X->cls[i]);
    }
    i = j;
  }
  return PA_OK;
}

static void arena_log_extent(const pa_arena *a, const pa_extent *X, const char *why, double ms) {
  if (!getenv("PA_SETUP_TIMING")) return;
  fprintf(stderr, "[pa arena] +%.1f GiB contiguous at %p (%s), %.1f ms, %d classes known, holding %.1f GiB in %zu extents: ", X->size / (double)GIB,
          (void *)X->base, why, ms, a->n_classes, a->held / (double)GIB, a->ext.size());
  for (int8_t v : X->cls) fputc(v < 0 ? '.' : (char)('0' + v), stderr);
  fputc('\n', stderr);
}

// Acquire and classify one extent of `bytes` (a multiple of the cell).  NULL (and the arena frozen) when the driver has no
// contiguous memory left or a probe failed: the callers fall back to hipMalloc.
static pa_extent *arena_acquire(pa_ctx *c, pa_arena *a, size_t bytes, const char *why) {
  if (a->frozen || c->capturing) return nullptr;       // (classifying launches and synchronises: not inside a capture)
  bytes = (bytes + a->cell - 1) / a->cell * a->cell;
  if (a->held + bytes > a->budget) return nullptr;
  const auto t0 = std::chrono::steady_clock::now();
  char *base = nullptr;
  if (hipExtMallocWithFlags((void **)&base, bytes, hipDeviceMallocContiguous) != hipSuccess) {
    (void)hipGetLastError();
    return nullptr;
  }
  pa_extent *X = new pa_extent();
  X->base = base; X->size = bytes;
  const int nc0 = a->n_classes;
  const char *ref0[3] = {a->ref[0], a->ref[1], a->ref[2]};
  char *scr0[3] = {a->scr[0], a->scr[1], a->scr[2]};
  if (classify_extent(c, a, X) != PA_OK) {             // leave the arena as it was and stop growing
    for (auto it = a->free_.begin(); it != a->free_.end();) it = it->second.e == X ? a->free_.erase(it) : std::next(it);
    a->n_classes = nc0;
    for (int k = 0; k < 3; ++k) { a->ref[k] = ref0[k]; a->scr[k] = scr0[k]; }
    (void)hipGetLastError();
    (void)hipFree(base);
    delete X;
    a->frozen = true;
    if (getenv("PA_SETUP_TIMING")) fprintf(stderr, "[pa arena] classification failed (%s): no more growth, hipMalloc from here on\n", pa_last_error());
    return nullptr;
  }
  a->ext.push_back(X);
  a->held += bytes; a->acquired += bytes; a->n_acquired++;
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  a->map_ms += ms;
  arena_log_extent(a, X, why, ms);
  return X;
}

static void arena_release(pa_arena *a, pa_extent *X) {
  for (auto it = a->free_.begin(); it != a->free_.end();) it = it->second.e == X ? a->free_.erase(it) : std::next(it);
  a->ext.erase(std::find(a->ext.begin(), a->ext.end(), X));
  a->held -= X->size; a->released += X->size; a->n_released++;
  (void)hipFree(X->base);
  if (getenv("PA_SETUP_TIMING")) fprintf(stderr, "[pa arena] -%.1f GiB at %p released, holding %.1f GiB\n", X->size / (double)GIB, (void *)X->base, a->held / (double)GIB);
  delete X;
}

// Extents nothing lives in are handed back to the driver -- all but the newest PA_ARENA_SPARE of them (default 0: what a
// context holds is what it uses; memory this process has freed is wiped by the driver when it is allocated again, 30-75 ms
// per GiB -- extent_probe (a): 16 GiB 0.84 s, 96 GiB 4.8 s -- so a caller that creates, destroys and re-creates big blocks
// in a loop may want a spare).
static void arena_trim(pa_arena *a) {
  if (a->walking) return;                  // (the extents a walk went over keep the driver from handing their memory out again)
  static const int spare = getenv("PA_ARENA_SPARE") ? std::max(0, atoi(getenv("PA_ARENA_SPARE"))) : 0;
  std::vector<pa_extent *> empty;
  for (pa_extent *X : a->ext) if (X->live == 0) empty.push_back(X);
  for (size_t i = 0; i + spare < empty.size(); ++i) arena_release(a, empty[i]);     // (ext is in order of acquisition: the oldest go)
  // Nothing of the context lives in the arena any more: up to PA_ARENA_SPARE_GIB (24) stay as they are, classes and all, for
  // whatever the caller builds next (a solver set up again, the next matrix of a parameter sweep: the driver wipes memory that
  // has been used when it is allocated again, 0.9 s for a 16 GiB extent -- a third of the multigrid set-up at 256^3);
  // beyond that, or on pa_ctx_arena_release, everything goes back, the extents that hold the classes' reference cells
  // included (a later allocation starts over: classes are only ever compared inside one context's lifetime of live buffers).
  static const size_t spare_bytes = (size_t)(getenv("PA_ARENA_SPARE_GIB") ? std::max(0, atoi(getenv("PA_ARENA_SPARE_GIB"))) : 24) * GIB;
  if (a->used == 0 && (a->held > spare_bytes || a->release_idle)) {
    while (!a->ext.empty()) arena_release(a, a->ext.back());
    a->n_classes = 0;
    for (int k = 0; k < 3; ++k) { a->ref[k] = nullptr; a->scr[k] = nullptr; a->mat_bytes[k] = a->vec_bytes[k] = 0; }
    a->matrix_class = -1;
    a->last_matrix_cls = -1;
    a->second_walk_done = false;
  }
}

static int arena_init(pa_ctx *c) {
  c->arena_tried = true;
  const char *on = getenv("PA_ARENA");
  if (on && atoi(on) == 0) return PA_OK;
  PA_HIP(hipSetDevice(c->device));
  size_t fr = 0, tot = 0;
  PA_HIP(hipMemGetInfo(&fr, &tot));
  double frac = 0.70;
  if (const char *e = getenv("PA_ARENA_FRACTION")) frac = std::min(0.95, std::max(0.05, atof(e)));
  pa_arena *a = new pa_arena();
  a->budget = (size_t)(frac * (double)fr);
  if (const char *e = getenv("PA_ARENA_GIB")) a->budget = std::min<size_t>((size_t)atol(e) * GIB, (size_t)(0.95 * (double)fr));
  c->arena = a;
  return PA_OK;
}

static size_t extent_bytes(const pa_arena *a, size_t request) {
  // 16 GiB: room for a 256^3 part's streams (3.6 GB of values, the columns while they are encoded, row pointers,
  // descriptors) and for the next block or two -- a block that does not fit makes the arena walk.  Not more: memory that
  // has been used before is wiped by the driver when it is allocated again, 30-75 ms per GiB (a 16 GiB extent: 15 ms
  // on clean memory, 0.5 s after other processes had used the device).
  size_t e = (size_t)16 * GIB;
  if (const char *s = getenv("PA_ARENA_EXTENT_GIB")) e = std::max<size_t>(1, (size_t)atol(s)) * GIB;
  // a request that does not fit an extent of the usual size gets one of its own: the buffer + a cell at either end (a
  // boundary cell is not handed out)
  const size_t need = (request + a->cell - 1) / a->cell * a->cell + 2 * a->cell;
  return std::max(e, need);
}

static void *arena_take(pa_arena *a, size_t bytes, int cls, int kind) {
  bytes = (bytes + ARENA_ALIGN - 1) / ARENA_ALIGN * ARENA_ALIGN;
  for (auto it = a->free_.begin(); it != a->free_.end(); ++it) {
    if (it->second.cls != cls || it->second.len < bytes) continue;
    const uintptr_t at = it->first;
    const pa_arena::blk b = it->second;
    a->free_.erase(it);
    if (b.len > bytes) a->free_[at + bytes] = {b.len - bytes, cls, 0, b.e};
    a->live_[at] = {bytes, cls, kind, b.e};
    b.e->live += bytes;
    a->used += bytes;
    a->peak = std::max(a->peak, a->used);
    (kind == PA_MEM_MATRIX ? a->mat_bytes : a->vec_bytes)[cls] += bytes;
    return (void *)at;
  }
  return nullptr;
}

static bool class_has_room(const pa_arena *a, size_t bytes, int cls) {
  bytes = (bytes + ARENA_ALIGN - 1) / ARENA_ALIGN * ARENA_ALIGN;
  for (auto &kv : a->free_) if (kv.second.cls == cls && kv.second.len >= bytes) return true;
  return false;
}

static void arena_give_back(pa_arena *a, void *p);

// Does the pair (newest big matrix stream, this vector) time as "different classes"?  The stand-in kernel streams the
// matrix buffer and writes into windows of the vector (up to 8, evenly spread, each what the kernel writes for the bytes it
// reads: ~20 MB); a control -- the same read stream against the scratch of the matrix streams' own class -- is taken before
// and after.  *same = some window times like the control (the clusters are ~15 % apart; a window that is half in the class
// sits half-way).  ~0.5 ms per window.  The map was measured cell by cell when the extents were acquired; this checks the
// one thing it is FOR, on the buffers actually handed out -- and it is what lets a plain hipMalloc'ed vector be used at all.
static int pair_check(pa_ctx *c, pa_arena *a, char *vec, size_t vec_bytes, bool *same) {
  probe_events ev;
  PA_TRY(ev.make());
  // The read stream: the reference cell of the matrix streams' class (512 MiB from HBM; a smaller matrix buffer would be
  // served by the 256 MiB Infinity Cache and show no class at all) -- "is this vector in the matrix streams' class" is the
  // question a classification pass asks of a cell.
  const int cls = a->last_matrix_cls;
  const char *rd = a->ref[cls];
  const int nb = std::min(probe_nb(a->cell), (int)(vec_bytes / 448));
  const size_t wr = (size_t)nb * 448;
  const int n_win = (int)std::min<size_t>(8, (vec_bytes + wr - 1) / wr);
  auto off_of = [&](int w) { return n_win == 1 ? (size_t)0 : ((vec_bytes - wr) * (size_t)w / (size_t)(n_win - 1)) & ~(size_t)4095; };
  char *slow_ctl = a->scr[cls], *fast_ctl = nullptr;
  for (int j = 0; j < a->n_classes; ++j) if (j != cls) { fast_ctl = a->scr[j]; break; }
  float s0 = 0, s1 = 0, f0 = 0, f1 = 0, worst = 0;
  PA_TRY(probe_ms(c, ev.e0, ev.e1, rd, slow_ctl, nb, &s0));
  PA_TRY(probe_ms(c, ev.e0, ev.e1, rd, slow_ctl, nb, &s0));
  if (fast_ctl) PA_TRY(probe_ms(c, ev.e0, ev.e1, rd, fast_ctl, nb, &f0));
  std::vector<float> t(n_win);
  for (int w = 0; w < n_win; ++w) PA_TRY(probe_ms(c, ev.e0, ev.e1, rd, vec + off_of(w), nb, &t[w]));
  PA_TRY(probe_ms(c, ev.e0, ev.e1, rd, slow_ctl, nb, &s1));
  if (fast_ctl) PA_TRY(probe_ms(c, ev.e0, ev.e1, rd, fast_ctl, nb, &f1));
  const float slow = 0.5f * (s0 + s1);
  float fast = fast_ctl ? 0.5f * (f0 + f1) : 0.87f * slow;
  if (fast > 0.95f * slow) fast = 0.87f * slow;
  const float thr = 0.5f * (slow + fast);
  for (int w = 0; w < n_win; ++w) {
    if (t[w] > thr) {                                   // (interference only ever slows a probe down: confirm)
      float again = 0;
      PA_TRY(probe_ms(c, ev.e0, ev.e1, rd, vec + off_of(w), nb, &again));
      t[w] = std::min(t[w], again);
    }
    worst = std::max(worst, t[w]);
  }
  *same = worst > thr;
  if (getenv("PA_SETUP_TIMING")) {
    fprintf(stderr, "[pa arena] pair check of a %.0f MiB vector at %p against class %d: same-class control %.4f ms, other-class %.4f ms, windows",
            vec_bytes / 1048576.0, (void *)vec, cls, slow, fast);
    for (float v : t) fprintf(stderr, " %.4f", v);
    fprintf(stderr, " -> %s\n", *same ? "SAME class" : "clear of it");
  }
  return PA_OK;
}

static void *arena_alloc(pa_ctx *c, pa_arena *a, size_t bytes, int kind) {
  void *p = nullptr;
  if (kind == PA_MEM_MATRIX) {
    // Matrix streams stay in ONE class (the one the first landed in) as long as the device has memory of that class; a
    // new extent comes before a spill into a class that holds vectors.
    if (a->matrix_class >= 0 && (p = arena_take(a, bytes, a->matrix_class, kind))) return p;
    if (a->matrix_class < 0)
      for (int k = 0; k < a->n_classes && !p; ++k) if (a->vec_bytes[k] == 0) p = arena_take(a, bytes, k, kind);
    if (!p) {
      // the class is full: further extents, one after the other, until one offers the matrix streams' class again or a
      // class no vector lives in (a vector must never find a matrix stream moving into its class: a product writing it
      // would lose 13-16 % -- measured when 8 GiB extents made the second block of bench.py land next to the vectors:
      // 0.95 instead of 0.81 ms); what the walk went over is handed back
      size_t walk_budget = (size_t)64 * GIB, walked_bytes = 0;
      if (const char *sw = getenv("PA_ARENA_WALK_GIB")) walk_budget = (size_t)atol(sw) * GIB;
      auto take_clean = [&]() -> void * {
        void *q = a->matrix_class >= 0 ? arena_take(a, bytes, a->matrix_class, kind) : nullptr;
        for (int k = 0; k < a->n_classes && !q; ++k) if (a->vec_bytes[k] == 0) q = arena_take(a, bytes, k, kind);
        return q;
      };
      a->walking = true;
      while (!p && walked_bytes < walk_budget) {
        pa_extent *X = arena_acquire(c, a, extent_bytes(a, bytes), "matrix streams");
        if (!X) break;
        walked_bytes += X->size;
        p = take_clean();
      }
      a->walking = false;
      arena_trim(a);
    }
    if (!p) {                                           // spill: the class with the fewest vector bytes first
      int order[3] = {0, 1, 2};
      std::sort(order, order + 3, [&](int x, int y) { return a->vec_bytes[x] < a->vec_bytes[y]; });
      for (int k = 0; k < 3 && !p; ++k) if (order[k] < a->n_classes) p = arena_take(a, bytes, order[k], kind);
    }
    if (p) {
      const int cls = a->live_[(uintptr_t)p].cls;
      if (a->matrix_class < 0) a->matrix_class = cls;
      if (bytes >= ((size_t)32 << 20)) a->last_matrix_cls = cls;
    }
    return p;
  }
  // Vectors: a class that holds no matrix stream -- alternating while there are two of those (BLAS-1 kernels, too, run
  // 3-6 % faster when what they write is not where they read).  When there is none, the arena WALKS: extents are acquired
  // one after the other (the driver hands out physical memory in order, and a class is a region of tens of GiB) until one
  // shows a class without matrix streams; the extents walked over are handed back at once.
  auto try_clean = [&]() -> void * {
    int cand[3], n = 0;
    for (int k = 0; k < a->n_classes; ++k) if (a->mat_bytes[k] == 0 && k != a->matrix_class && class_has_room(a, bytes, k)) cand[n++] = k;
    if (n == 0) return nullptr;
    return arena_take(a, bytes, cand[(a->vec_turn++) % (unsigned)n], kind);
  };
  static const int check_mode = getenv("PA_ARENA_SELFCHECK") ? atoi(getenv("PA_ARENA_SELFCHECK")) : 1;
  static const int plain_first = getenv("PA_ARENA_PLAIN_VECTORS") ? atoi(getenv("PA_ARENA_PLAIN_VECTORS")) : 0;   // (experimental, below)
  if (a->last_matrix_cls < 0) return nullptr;           // no matrix stream to stay away from (yet): a plain allocation
  // A caller that asks for it (pa_ctx_arena_hint(2)) gets TWO vector classes: kernels that read vectors and write one (BLAS-1,
  // the Gauss-Seidel colour updates) run a little faster when what they write is not where they read -- 1.2 % of an MG-PCG
  // iteration at 256^3 (5.78 vs 5.85 ms; the 5.78 / 6.16 measured in round 2 was mostly the self-check then run per
  // allocation, profiles/r03_mg_ab.md).  Costs one more walk (<= 16 GiB), once: up to a second on memory other processes have
  // used, which is why hpcg.pc_setup no longer asks by default.
  auto take_new = [&]() -> void * {
    for (int k = 0; k < a->n_classes; ++k)
      if (a->mat_bytes[k] == 0 && a->vec_bytes[k] == 0 && k != a->matrix_class && class_has_room(a, bytes, k)) return arena_take(a, bytes, k, kind);
    return nullptr;
  };
  int n_vec_classes = 0;
  for (int k = 0; k < a->n_classes; ++k) n_vec_classes += a->mat_bytes[k] == 0 && k != a->matrix_class && (a->vec_bytes[k] > 0 || class_has_room(a, bytes, k));
  const bool second = a->want_vec_classes >= 2 && !a->second_walk_done && n_vec_classes == 1 && !a->frozen && !c->capturing;
  if (second) {
    a->second_walk_done = true;
    p = take_new();
  } else {
    p = try_clean();
  }
  auto accept = [&]() -> void * { return second ? take_new() : try_clean(); };
  if (!p && plain_first && check_mode && bytes >= ((size_t)32 << 20) && !c->capturing) {
    // PA_ARENA_PLAIN_VECTORS=1 (off by default: a vector verified against the matrix streams' class today is not verified
    // against the class a LATER block may have to take).  Before walking (which acquires -- and makes the driver wipe --
    // tens of GiB): the driver serves plain allocations
    // from another end of the memory than the contiguous extents (tools/probe/extent_probe.hip (d): 128 MiB ... 4 GiB
    // buffers never shared the first extent's class), so a plain buffer that the pair check finds clear of the matrix
    // streams' class, window by window, is as good a home for a vector as a mapped extent -- and costs nothing to hold.
    void *q = nullptr;
    if (hipMalloc(&q, bytes) == hipSuccess) {
      bool same = true;
      if (pair_check(c, a, (char *)q, bytes, &same) == PA_OK && !same) {
        a->foreign_[(uintptr_t)q] = bytes;
        a->foreign_bytes += bytes;
        a->check_ok++;
        return q;
      }
      (void)hipGetLastError();
      (void)hipFree(q);
      a->plain_rejected++;
    } else (void)hipGetLastError();
  }
  if (!p && !a->frozen && !c->capturing) {
    size_t walk_budget = (size_t)160 * GIB;
    if (const char *s = getenv("PA_ARENA_WALK_GIB")) walk_budget = (size_t)atol(s) * GIB;
    if (second) walk_budget = std::min(walk_budget, (size_t)16 * GIB);   // (a second class is worth 1 % to a solver, not a long walk)
    size_t walked_bytes = 0;
    pa_extent *found = nullptr;
    // The vectors' extents are small (4 GiB, or 8 x the request: the vectors of a part are a fraction of its matrix
    // streams), and the walk steps in that size: the extent it ends in is the one that stays.  (Handing a big step back
    // and taking a smaller extent "in its place" does not work: the driver serves the next request from elsewhere --
    // measured, profiles/r03_mg_ab.md -- and wipes what was handed back before it is used again, 1 s per 16 GiB.)
    const size_t step = std::max<size_t>((size_t)4 * GIB, (8 * bytes + a->cell - 1) / a->cell * a->cell + 2 * a->cell);
    a->walking = true;
    while (!p && walked_bytes < walk_budget) {
      pa_extent *X = arena_acquire(c, a, step, second ? "vectors: looking for a second class of their own" : "vectors: looking for a class without matrix streams");
      if (!X) break;
      walked_bytes += X->size;
      if ((p = accept()) != nullptr) found = X;
    }
    (void)found;
    a->walking = false;
    arena_trim(a);                                      // what the walk went over goes back at once
  }
  if (!p && second) p = try_clean();                    // (no second class within reach: the first one serves)
  if (!p) {                                             // nothing clean anywhere: next to the fewest matrix bytes
    int order[3] = {0, 1, 2};
    std::sort(order, order + 3, [&](int x, int y) { return a->mat_bytes[x] < a->mat_bytes[y]; });
    for (int k = 0; k < 3 && !p; ++k) if (order[k] < a->n_classes) p = arena_take(a, bytes, order[k], kind);
    return p;                                           // (knowingly next to matrix streams: nothing to check)
  }
  // self-check of the pair actually handed out (big vectors only: what a product writes)
  // (once per cell and matrix class: a solver that allocates its work vectors on every call must not pay a dozen probe
  // launches per vector each time -- 0.57 ms per MG-PCG iteration of a 30-iteration solve at 256^3 when it did)
  if (check_mode && a->last_matrix_cls >= 0 && bytes >= ((size_t)32 << 20) && !c->capturing) {
    const pa_arena::blk &bk = a->live_[(uintptr_t)p];
    const int cls = bk.cls;
    bool same = false;
    const uint8_t bit = (uint8_t)(1u << a->last_matrix_cls);
    const size_t c0 = (size_t)((char *)p - bk.e->base) / a->cell, c1 = (size_t)((char *)p + bytes - 1 - bk.e->base) / a->cell;
    bool known = cls != a->last_matrix_cls;
    for (size_t k = c0; k <= c1 && known; ++k) known = k < bk.e->clear_of.size() && (bk.e->clear_of[k] & bit);
    if (known) {
      // every cell under this vector has been checked against the matrix streams' class before
    } else if (cls != a->last_matrix_cls && pair_check(c, a, (char *)p, bytes, &same) == PA_OK) {
      if (!same) {
        a->check_ok++;
        if (bk.e->clear_of.size() < bk.e->cls.size()) bk.e->clear_of.resize(bk.e->cls.size(), 0);
        for (size_t k = c0; k <= c1 && k < bk.e->clear_of.size(); ++k) bk.e->clear_of[k] |= bit;
      } else {
        a->check_failed++;
        // the map says "different classes", the pair says "same": try the other clean class once, else keep it and say so
        void *q = nullptr;
        for (int k = 0; k < a->n_classes && !q; ++k)
          if (k != cls && k != a->last_matrix_cls && a->mat_bytes[k] == 0) q = arena_take(a, bytes, k, kind);
        bool same2 = true;
        if (q && pair_check(c, a, (char *)q, bytes, &same2) == PA_OK && !same2) {
          arena_give_back(a, p);
          p = q;
        } else {
          if (q) arena_give_back(a, q);
          if (!a->warned) {
            a->warned = true;
            fprintf(stderr, "[pa arena] warning: a vector placed in memory class %d times as if it shared the matrix streams' class %d "
                            "(products writing it may run ~13 %% slower); PA_ARENA_SELFCHECK=0 silences the check\n", cls, a->last_matrix_cls);
          }
        }
      }
    } else (void)hipGetLastError();
  }
  return p;
}

static void arena_give_back(pa_arena *a, void *p) {
  auto it = a->live_.find((uintptr_t)p);
  if (it == a->live_.end()) return;
  pa_arena::blk b = it->second;
  a->live_.erase(it);
  a->used -= b.len;
  b.e->live -= b.len;
  size_t *acct = b.kind == PA_MEM_MATRIX ? a->mat_bytes : a->vec_bytes;
  acct[b.cls] -= std::min(acct[b.cls], b.len);
  uintptr_t start = (uintptr_t)p;
  size_t len = b.len;
  auto nx = a->free_.find(start + len);                 // merge with free neighbours of the same extent and class
  if (nx != a->free_.end() && nx->second.e == b.e && nx->second.cls == b.cls) {
    len += nx->second.len;
    a->free_.erase(nx);
  }
  auto pv = a->free_.lower_bound(start);
  if (pv != a->free_.begin()) {
    --pv;
    if (pv->first + pv->second.len == start && pv->second.e == b.e && pv->second.cls == b.cls) {
      start = pv->first;
      len += pv->second.len;
      a->free_.erase(pv);
    }
  }
  a->free_[start] = {len, b.cls, 0, b.e};
  // (the matrix streams' class stays what it is while extents are held: the next block goes where the room for it is)
  if (b.e->live == 0 || a->used == 0) arena_trim(a);    // nothing of the extent (or of the context) is in use any more
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
  if (!guard_mode()) {
    hipError_t st = hipMalloc(p, bytes);
    if (st == hipErrorOutOfMemory && pa_scratch().held_bytes() > 0) {   // (the scratch cache holds freed set-up temporaries;
      (void)hipGetLastError();                                          //  out of memory anywhere in the library empties it and retries)
      (void)hipDeviceSynchronize();
      pa_scratch().trim();
      st = hipMalloc(p, bytes);
    }
    return st;
  }
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
  size_t first_big = (size_t)256 << 20;          // the arena comes into being when the first allocation this large arrives
  if (const char *e = getenv("PA_ARENA_MIN_MIB")) first_big = (size_t)atol(e) << 20;
  if (kind != PA_MEM_PLAIN && bytes >= small) {
    // (pa_*_create / pa_*_destroy may be reached from any host thread -- finalizers of a managed host language)
    std::lock_guard<std::mutex> lk(c->mem_mu);
    if (!c->arena && !c->arena_tried && bytes >= first_big && !c->capturing) {
      if (arena_init(c) != PA_OK) (void)hipGetLastError();    // no arena: plain hipMalloc below
    }
    if (pa_arena *a = c->arena) {
      PA_HIP(hipSetDevice(c->device));
      if ((*p = arena_alloc(c, a, bytes, kind)) != nullptr) return PA_OK;
    }
  }
  PA_HIP(hipMalloc(p, bytes));
  return PA_OK;
}

// A buffer in a NAMED place, for the placement A/B of a product's write stream (pa_spmv_tune_output): memory class `cls` of the
// extents the arena HOLDS (nothing is acquired, no walk, no pair check), or a plain hipMalloc (cls < 0).  *p = NULL (and
// PA_OK) when that place has no room.  Freed with pa_dev_free like everything else.
int pa_dev_alloc_at(pa_ctx *c, void **p, size_t bytes, int cls) {
  *p = nullptr;
  if (bytes == 0) bytes = 8;
  PA_HIP(hipSetDevice(c->device));
  if (cls < 0 || guard_mode()) {
    if (guard_mode()) { PA_HIP(pa_raw_malloc_impl(p, bytes)); return PA_OK; }
    if (hipMalloc(p, bytes) != hipSuccess) { (void)hipGetLastError(); *p = nullptr; }
    return PA_OK;
  }
  std::lock_guard<std::mutex> lk(c->mem_mu);
  pa_arena *a = c->arena;
  if (!a || cls >= a->n_classes) return PA_OK;
  *p = arena_take(a, bytes, cls, PA_MEM_VECTOR);
  return PA_OK;
}

void pa_dev_free(pa_ctx *c, void *p) {
  if (!p) return;
  if (guard_mode()) { (void)pa_raw_free(p); return; }
  if (c) {
    std::lock_guard<std::mutex> lk(c->mem_mu);
    pa_arena *a = c->arena;
    if (a && a->live_.count((uintptr_t)p)) {
      arena_give_back(a, p);
      return;
    }
    if (a) {
      auto f = a->foreign_.find((uintptr_t)p);
      if (f != a->foreign_.end()) { a->foreign_bytes -= f->second; a->foreign_.erase(f); }
    }
  }
  (void)hipFree(p);
}

int pa_mem_class(const pa_ctx *c, const void *p) {
  if (!c || !p) return -1;
  std::lock_guard<std::mutex> lk(const_cast<pa_ctx *>(c)->mem_mu);
  const pa_arena *a = c->arena;
  if (!a) return -1;
  for (const pa_extent *X : a->ext)
    if ((const char *)p >= X->base && (const char *)p < X->base + X->size) return X->cls[(size_t)((const char *)p - X->base) / a->cell];
  if (a->foreign_.count((uintptr_t)p)) return PA_MEM_CLASS_PLAIN_VERIFIED;
  return -1;
}

void pa_arena_destroy(pa_ctx *c) {
  if (!c || !c->arena) return;
  for (pa_extent *X : c->arena->ext) {
    (void)hipFree(X->base);
    delete X;
  }
  delete c->arena;
  c->arena = nullptr;
}

extern "C" int pa_ctx_arena_info(pa_ctx *c, int64_t *bytes, int *n_classes, int64_t class_bytes[3], int64_t *used,
                                 double *map_ms, int *matrix_class) {
  PA_REQUIRE(c != nullptr, "ctx is NULL");
  std::lock_guard<std::mutex> lk(c->mem_mu);
  const pa_arena *a = c->arena;
  if (bytes) *bytes = a ? (int64_t)a->held : 0;
  if (n_classes) *n_classes = a ? a->n_classes : 0;
  if (class_bytes) {
    for (int k = 0; k < 3; ++k) class_bytes[k] = 0;
    if (a) for (const pa_extent *X : a->ext) for (int8_t v : X->cls) if (v >= 0) class_bytes[v] += (int64_t)a->cell;
  }
  if (used) *used = a ? (int64_t)a->used : 0;
  if (map_ms) *map_ms = a ? a->map_ms : 0.0;
  if (matrix_class) *matrix_class = a ? a->matrix_class : -1;
  return PA_OK;
}

extern "C" int pa_ctx_arena_map(pa_ctx *c, int64_t *cell_bytes, int8_t *classes, int64_t capacity, int64_t *n_cells) {
  PA_REQUIRE(c && n_cells, "bad arguments");
  std::lock_guard<std::mutex> lk(c->mem_mu);
  const pa_arena *a = c->arena;
  int64_t n = 0;
  if (a) for (const pa_extent *X : a->ext) {             // extent after extent, -2 between two of them
    if (n && classes && n < capacity) classes[n] = -2;
    if (n) ++n;
    for (int8_t v : X->cls) { if (classes && n < capacity) classes[n] = v; ++n; }
  }
  *n_cells = n;
  if (cell_bytes) *cell_bytes = a ? (int64_t)a->cell : 0;
  return PA_OK;
}

extern "C" int pa_ctx_arena_stats(pa_ctx *c, int64_t *n_extents, int64_t *bytes_acquired, int64_t *bytes_released,
                                  int64_t *peak_used, int64_t *pairs_ok, int64_t *pairs_failed, int64_t *budget,
                                  int64_t *plain_vector_bytes) {
  PA_REQUIRE(c != nullptr, "ctx is NULL");
  std::lock_guard<std::mutex> lk(c->mem_mu);
  const pa_arena *a = c->arena;
  if (n_extents) *n_extents = a ? (int64_t)a->ext.size() : 0;
  if (bytes_acquired) *bytes_acquired = a ? (int64_t)a->acquired : 0;
  if (bytes_released) *bytes_released = a ? (int64_t)a->released : 0;
  if (peak_used) *peak_used = a ? (int64_t)a->peak : 0;
  if (pairs_ok) *pairs_ok = a ? a->check_ok : 0;
  if (plain_vector_bytes) *plain_vector_bytes = a ? (int64_t)a->foreign_bytes : 0;
  if (pairs_failed) *pairs_failed = a ? a->check_failed : 0;
  if (budget) *budget = a ? (int64_t)a->budget : 0;
  return PA_OK;
}

// A caller about to allocate a solver's worth of vectors (a multigrid hierarchy) asks for `vector_classes` = 2: the vectors
// then alternate between two memory classes of their own (see arena_alloc); 1 = the default.
extern "C" int pa_ctx_arena_release(pa_ctx *c) {
  PA_REQUIRE(c != nullptr, "ctx is NULL");
  std::lock_guard<std::mutex> lk(c->mem_mu);
  if (pa_arena *a = c->arena) {
    PA_HIP(hipSetDevice(c->device));
    a->release_idle = true;
    arena_trim(a);
    a->release_idle = false;
  }
  return PA_OK;
}

extern "C" int pa_ctx_arena_hint(pa_ctx *c, int vector_classes) {
  PA_REQUIRE(c && (vector_classes == 1 || vector_classes == 2), "bad arguments");
  std::lock_guard<std::mutex> lk(c->mem_mu);
  if (!c->arena && !c->arena_tried) PA_TRY(arena_init(c));
  if (c->arena) c->arena->want_vec_classes = vector_classes;
  return PA_OK;
}

// (kept for callers that want the first extent before their first big allocation: it acquires one extent of the usual size)
extern "C" int pa_ctx_arena_build(pa_ctx *c) {
  PA_REQUIRE(c != nullptr, "ctx is NULL");
  std::lock_guard<std::mutex> lk(c->mem_mu);
  if (c->arena || c->arena_tried) return PA_OK;
  PA_TRY(arena_init(c));
  if (c->arena) (void)arena_acquire(c, c->arena, extent_bytes(c->arena, 0), "pa_ctx_arena_build");
  return PA_OK;
}
