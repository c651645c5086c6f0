// pa_push_dev.h -- device-side pieces of the push transport shared by pa_push.hip (the transport's own launches) and pa_fused.hip
// (the same work done by the first / last blocks of the fused product launch).
#ifndef PA_PUSH_DEV_H
#define PA_PUSH_DEV_H

#include <hip/hip_runtime.h>

#include <cstdint>

#define PA_PUSH_MAX_PARTS 32

struct pa_push_seg {
  int32_t start, len;                 // entries [start, start + len) of the part's send list ...
  double *dst;                        // ... go to dst[0 .. len)
  const int32_t *uidx;                // (one device, consistent!) and on into ghost entry uidx[k] of the receiving part's vector,
  int32_t upart;                      //     which is vector `upart` of the launch (k_push_unpack); NULL: no unpack table
  unsigned long long *arrive;         // ipc: the flag in the receiver's memory this slice's arrival is announced in
  const unsigned long long *ack;      // ipc: the flag in MY memory the receiver acknowledges the previous slice in
};
struct pa_push_part {
  const int32_t *idx;                 // send list (local ids)
  int32_t n, seg0, nseg, blk0;
};
struct pa_push_vecs { const double *v[PA_PUSH_MAX_PARTS]; };

__device__ __forceinline__ int push_find_seg(const pa_push_seg *__restrict__ segs, int s0, int ns, int p) {
  int lo = s0, hi = s0 + ns - 1;                       // the last segment whose start <= p
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (segs[mid].start <= p) lo = mid; else hi = mid - 1;
  }
  return lo;
}

__device__ __forceinline__ unsigned long long flag_load(const unsigned long long *p) {
  return __hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void flag_store(unsigned long long *p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// spin until *p >= want; false after `ticks` of the 100 MHz wall clock
__device__ __forceinline__ bool flag_wait(const unsigned long long *p, unsigned long long want, long long ticks) {
  if (flag_load(p) >= want) return true;
  const long long t0 = (long long)wall_clock64();
  for (;;) {
    for (int k = 0; k < 64; ++k) {
      if (flag_load(p) >= want) return true;
      __builtin_amdgcn_s_sleep(8);
    }
    if ((long long)wall_clock64() - t0 > ticks) return false;
  }
}

// pack + deliver of one 256-entry block of this part's send list over the ipc link (k_push_ipc's body; also the first blocks of the
// fused product launch): flow control on the receivers' acknowledgements of the previous exchange, the stores, and -- the last block
// to finish -- the arrival announcement to every receiver.  `ok`: a shared int of the caller.  false: an acknowledgement timed out.
__device__ __forceinline__ bool pa_push_ipc_block(int *ok, const int32_t *__restrict__ idx, int n, const pa_push_seg *__restrict__ segs,
                                                  int nseg, const double *__restrict__ v, unsigned long long seq, unsigned *done,
                                                  long long ticks, int *status, int block, int n_blocks) {
  const int p0 = block * 256, p = p0 + (int)threadIdx.x;
  if (threadIdx.x == 0) {
    // flow control: the receivers of the slices this block writes must have consumed what the previous exchange put there
    int good = 1;
    if (seq > 1 && n > 0) {
      const int sa = push_find_seg(segs, 0, nseg, min(p0, n - 1)), sb = push_find_seg(segs, 0, nseg, min(p0 + 255, n - 1));
      for (int s = sa; s <= sb && good; ++s) good = flag_wait(segs[s].ack, seq - 1, ticks) ? 1 : 0;
    }
    *ok = good;
  }
  __syncthreads();
  const bool good = *ok != 0;
  if (!good) { if (threadIdx.x == 0) atomicExch(status, 2); }
  else if (p < n) {
    const double val = v[idx[p]];
    const int s = push_find_seg(segs, 0, nseg, p);
    segs[s].dst[p - segs[s].start] = val;
  }
  __threadfence_system();                              // my stores are in memory before I count myself done
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = atomicAdd(done, 1u);
    if (t == (unsigned)n_blocks - 1) {                 // the last block of the launch announces every slice
      *done = 0;
      __threadfence_system();
      if (good) for (int s = 0; s < nseg; ++s) flag_store(segs[s].arrive, seq);
    }
  }
  return good;
}

#endif
