// pa_dev_util.h -- small device-side building blocks of the set-up kernels (pa_setup.hip, pa_assemble.hip): temporaries
// that free themselves, rocPRIM scans and radix sorts on the context's compute stream.  Private to libpa_hip.so.
#ifndef PA_DEV_UTIL_H
#define PA_DEV_UTIL_H

#include <hip/hip_runtime.h>

#include <cstring>   // (rocprim's texture iterator calls memset from host code)
#include <rocprim/rocprim.hpp>

#include <algorithm>
#include <vector>

#include "pa_internal.h"
#include "pa_scratch.h"

namespace pa_util {

struct scratch {                                   // temporaries that go back when the owner goes out of scope (pa_scratch.h: a small cache
  struct blk { void *p; size_t n; int dev; };      // in front of hipMalloc / hipFree)
  std::vector<blk> p;
  template <class T> int get(T **out, size_t n) {
    const size_t bytes = pa_scratch_cache::round_up(std::max<size_t>(sizeof(T) * n, 16));
    int dev = 0;
    (void)hipGetDevice(&dev);
    void *q = pa_scratch().cap ? pa_scratch().take(bytes, dev) : nullptr;
    if (!q) {
      if (hipMalloc(&q, bytes) != hipSuccess) {      // (what the cache holds may be what is missing)
        (void)hipGetLastError();
        pa_scratch().trim();
        PA_HIP(hipMalloc(&q, bytes));
      }
    }
    p.push_back(blk{q, bytes, dev});
    *out = (T *)q;
    return PA_OK;
  }
  static void back(const blk &b) {                   // (behind a device synchronise: see pa_scratch.h)
    if (!pa_scratch().give(b.p, b.n, b.dev)) (void)hipFree(b.p);
  }
  void release(void *q) {
    for (size_t i = 0; i < p.size(); ++i)
      if (p[i].p == q) {
        (void)hipDeviceSynchronize();
        back(p[i]);
        p.erase(p.begin() + i);
        return;
      }
  }
  ~scratch() {
    if (p.empty()) return;
    (void)hipDeviceSynchronize();
    for (const blk &b : p) back(b);
  }
};

inline dim3 grid1(int64_t n, int t = 256) { return dim3((unsigned)std::max<int64_t>(1, (n + t - 1) / t)); }

template <class T>
int scan_exclusive(scratch &sc, hipStream_t s, const T *in, T *out, size_t n) {
  size_t tb = 0;
  PA_HIP(rocprim::exclusive_scan((void *)nullptr, tb, in, out, (T)0, n, rocprim::plus<T>(), s));
  char *tmp = nullptr;
  PA_TRY(sc.get(&tmp, tb));
  PA_HIP(rocprim::exclusive_scan((void *)tmp, tb, in, out, (T)0, n, rocprim::plus<T>(), s));
  return PA_OK;
}

inline int scan_inclusive(scratch &sc, hipStream_t s, const int *in, int *out, size_t n) {
  size_t tb = 0;
  PA_HIP(rocprim::inclusive_scan((void *)nullptr, tb, in, out, n, rocprim::plus<int>(), s));
  char *tmp = nullptr;
  PA_TRY(sc.get(&tmp, tb));
  PA_HIP(rocprim::inclusive_scan((void *)tmp, tb, in, out, n, rocprim::plus<int>(), s));
  return PA_OK;
}

// stable: equal keys keep the order of the input
template <class K>
int sort_pairs(scratch &sc, hipStream_t s, const K *ki, K *ko, const int *vi, int *vo, size_t n, unsigned end_bit = 8 * sizeof(K)) {
  size_t tb = 0;
  PA_HIP(rocprim::radix_sort_pairs((void *)nullptr, tb, ki, ko, vi, vo, n, 0, end_bit, s));
  char *tmp = nullptr;
  PA_TRY(sc.get(&tmp, tb));
  PA_HIP(rocprim::radix_sort_pairs((void *)tmp, tb, ki, ko, vi, vo, n, 0, end_bit, s));
  sc.release(tmp);
  return PA_OK;
}

template <class T>
int d2h(hipStream_t s, T *host, const T *dev, size_t n) {
  PA_HIP(hipMemcpyAsync(host, dev, sizeof(T) * n, hipMemcpyDeviceToHost, s));
  PA_HIP(hipStreamSynchronize(s));
  return PA_OK;
}

}  // namespace pa_util

#endif
