// pa_scratch.h -- where the set-up kernels' temporaries come from (round 5).
//
// The set-up routes (pa_setup / pa_assemble / pa_rowsel / pa_transpose / pa_fused) take their temporaries through `scratch`
// (pa_dev_util.h): ~1500 hipMalloc / hipFree pairs per HPCG set-up pair, 66 ms of a 0.54 s total, almost all of it in hipFree
// (the unmapping; `tools/probe/pc_setup_pair.py` with -DPA_SCRATCH_TIMING).  Freed temporaries of up to PA_SCRATCH_BLOCK_MIB now
// wait in a small per-process cache (PA_SCRATCH_CACHE_MIB in all, per device) for the next request of their size class.
// A block goes back to the cache only behind a device synchronise (what hipFree implied), so nothing in flight can still be
// using it when the next owner -- whatever its stream or context -- gets it.  PA_SCRATCH_CACHE_MIB=0: hipMalloc / hipFree as before.
#ifndef PA_SCRATCH_H
#define PA_SCRATCH_H

#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <vector>

struct pa_scratch_cache {
  struct blk { void *p; size_t n; int dev; };
  std::mutex m;
  std::vector<blk> free_;
  size_t held = 0, cap = 0, max_block = 0;
  long hits = 0, misses = 0;
  pa_scratch_cache() {
    const char *e = getenv("PA_SCRATCH_CACHE_MIB"), *b = getenv("PA_SCRATCH_BLOCK_MIB");
    cap = (size_t)(e ? std::max(0, atoi(e)) : 4096) << 20;
    max_block = (size_t)(b ? std::max(1, atoi(b)) : 256) << 20;
  }
  ~pa_scratch_cache() { free_.clear(); }                 // (process exit: the runtime is going away, nothing to hand back)
  static size_t round_up(size_t n) {                     // size classes: 64 KiB steps up to 1 MiB, then 1/8 of the leading power of two
    if (n <= (1u << 16)) return 1u << 16;
    if (n <= (1u << 20)) return (n + 0xffff) & ~(size_t)0xffff;
    size_t p = (size_t)1 << 20;
    while ((p << 1) <= n) p <<= 1;
    const size_t step = p >> 3;
    return (n + step - 1) / step * step;
  }
  void *take(size_t n, int dev) {
    std::lock_guard<std::mutex> g(m);
    for (size_t i = 0; i < free_.size(); ++i)
      if (free_[i].dev == dev && free_[i].n == n) {
        void *p = free_[i].p;
        held -= n;
        free_[i] = free_.back();
        free_.pop_back();
        ++hits;
        return p;
      }
    ++misses;
    return nullptr;
  }
  // false: the caller frees the block itself
  bool give(void *p, size_t n, int dev) {
    if (cap == 0 || n > max_block) return false;
    std::vector<void *> out;
    {
      std::lock_guard<std::mutex> g(m);
      while (held + n > cap && !free_.empty()) {         // oldest first
        out.push_back(free_.front().p);
        held -= free_.front().n;
        free_.erase(free_.begin());
      }
      free_.push_back(blk{p, n, dev});
      held += n;
    }
    for (void *q : out) (void)hipFree(q);
    return true;
  }
  size_t held_bytes() {
    std::lock_guard<std::mutex> g(m);
    return held;
  }
  void trim() {
    std::vector<void *> out;
    {
      std::lock_guard<std::mutex> g(m);
      if (getenv("PA_SETUP_TIMING")) fprintf(stderr, "[pa scratch] %ld requests served from the cache, %ld by hipMalloc; holding %.1f MiB in %zu blocks\n", hits, misses, held / 1048576.0, free_.size());
      for (auto &b : free_) out.push_back(b.p);
      free_.clear();
      held = 0;
    }
    for (void *q : out) (void)hipFree(q);
  }
};
inline pa_scratch_cache &pa_scratch() { static pa_scratch_cache c; return c; }

#endif
