// pa_spmv_kernel.h -- the row-split CSR SpMV kernel (K1/K2) and its host-side row split.
// Shared by pa_device.hip (the product) and probe/spmv_probe.hip (A/B tuning harness).
//
// Reference loops: spmv_csr! src/sparse_utils.jl:649-669; muladd! src/p_sparse_matrix.jl:2088.
// Must be compiled with -ffp-contract=off (one rounding per multiply and per add).
#ifndef PA_SPMV_KERNEL_H
#define PA_SPMV_KERNEL_H

#include <hip/hip_runtime.h>

#include <cstdint>
#include <thread>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));

template <bool NT, typename T>
__device__ __forceinline__ T pa_stream_load(const T *p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);  // read-once matrix stream: keep x in L2
  else return *p;
}

typedef unsigned short us4 __attribute__((ext_vector_type(4)));

// Column encodings of a chunk
//   32-bit : col[p]                                            (always present; fallback and long rows)
//   "c16"  : col[p] = win[chunk][c16[p] >> 12] + (c16[p] & 4095)  -- up to PA_C16_WINDOWS 4096-aligned column
//            windows per chunk (stencil / FEM / banded rows touch a handful of narrow column clusters), 2 bytes per
//            stored entry instead of 4.  Pure index compression: values stay fp64, pa_csr_update_values is unaffected,
//            and a chunk whose columns need more windows keeps the 32-bit path (win[chunk][0] < 0).
#define PA_C16_WINDOWS 16

// y[row] = beta*y[row] + sum_p (val[p]*x[col[p]])*alpha, products summed in ascending p.
//   BLK  threads per workgroup, NPT stored entries per lane (chunk capacity CAP = BLK*NPT products in LDS),
//   NT   non-temporal matrix loads, C16 use the 16-bit column stream where the chunk has one.
template <int BLK, int NPT, bool NT, bool C16>
__global__ __launch_bounds__(BLK) void k_spmv_rowsplit(
    const int *__restrict__ crp, const int *__restrict__ col, const unsigned short *__restrict__ col16,
    const int *__restrict__ win, const double *__restrict__ val, const double *__restrict__ x,
    double *__restrict__ y, const int *__restrict__ chunk_row, const int *__restrict__ row_ids, int n_chunks,
    int chunks_per_xcd, double alpha, double beta) {
  constexpr int CAP = BLK * NPT;
  static_assert(NPT % 2 == 0, "pairs");
  __shared__ __attribute__((aligned(16))) double prod[CAP];
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int chunk = (b & 7) * chunks_per_xcd + (b >> 3);  // XCD-aware: block b sits on XCD b%8
  if (chunk >= n_chunks || (b >> 3) >= chunks_per_xcd) return;
  const int r0 = chunk_row[chunk];
  const int r1 = chunk_row[chunk + 1];
  const int p0 = crp[r0];
  const int p1 = crp[r1];
  const int base = p0 & ~1;  // 16-byte aligned value pairs, 4-byte aligned c16 pairs

  if (p1 - base <= CAP) {
    // my first row's extent, fetched early so the latency hides under the matrix stream
    int ra = 0, re = 0;
    if (r0 + tid < r1) {
      ra = crp[r0 + tid];
      re = crp[r0 + tid + 1];
    }
    // Unconditional loads: lanes past the chunk's end re-read its last pair (same address => no extra
    // traffic) and their products are never summed.  A guarded load would make the compiler wait for
    // each load before issuing the next (one HBM round trip per k instead of one per chunk).
    // Every load instruction is contiguous across the 64 lanes (16 B, 8 B or 4 B per lane).
    const int last = max((p1 - 1) & ~1, 0);
    int mywin = 0;
    if (C16) mywin = win[chunk * PA_C16_WINDOWS + (tid & (PA_C16_WINDOWS - 1))];
    const bool use16 = C16 && (__builtin_amdgcn_readfirstlane(mywin) >= 0);   // lane 0 holds window 0
    d2 v[NPT / 2];
    int c0[NPT / 2], c1[NPT / 2];
    if (use16) {
      unsigned q[NPT / 2];
#pragma unroll
      for (int k = 0; k < NPT / 2; ++k) {
        const int idx = min(base + (k * BLK + tid) * 2, last);
        v[k] = pa_stream_load<NT>(reinterpret_cast<const d2 *>(val + idx));
        q[k] = pa_stream_load<NT>(reinterpret_cast<const unsigned *>(col16 + idx));
      }
#pragma unroll
      for (int k = 0; k < NPT / 2; ++k) {
        // window base of slot s lives in lane s of `mywin` (any 16-lane group): fetch it with ds_bpermute
        const unsigned lo = q[k] & 0xffffu, hi = q[k] >> 16;
        c0[k] = __builtin_amdgcn_ds_bpermute((lo >> 12) << 2, mywin) + (lo & 4095);
        c1[k] = __builtin_amdgcn_ds_bpermute((hi >> 12) << 2, mywin) + (hi & 4095);
      }
    } else {
#pragma unroll
      for (int k = 0; k < NPT / 2; ++k) {
        const int idx = min(base + (k * BLK + tid) * 2, last);
        v[k] = pa_stream_load<NT>(reinterpret_cast<const d2 *>(val + idx));
        const i2 c = pa_stream_load<NT>(reinterpret_cast<const i2 *>(col + idx));
        c0[k] = c.x; c1[k] = c.y;
      }
    }
#pragma unroll
    for (int k = 0; k < NPT / 2; ++k) {
      d2 pr;
      pr.x = v[k].x * x[c0[k]];
      pr.y = v[k].y * x[c1[k]];
      if (alpha != 1.0) {
        pr.x = pr.x * alpha;
        pr.y = pr.y * alpha;
      }
      *reinterpret_cast<d2 *>(&prod[(k * BLK + tid) * 2]) = pr;
    }
    __syncthreads();
    for (int r = r0 + tid; r < r1; r += BLK) {
      if (r != r0 + tid) {
        ra = crp[r];
        re = crp[r + 1];
      }
      const int row = row_ids ? row_ids[r] : r;
      double acc = (beta == 0.0) ? 0.0 : beta * y[row];
      const int a = ra - base, e = re - base;
#pragma unroll 4
      for (int p = a; p < e; ++p) acc = acc + prod[p];
      y[row] = acc;
    }
  } else {
    // one long row (more stored entries than a chunk holds): windows of CAP products, summed by
    // lane 0 in ascending p so that even this path keeps the reference's order.
    const int row = row_ids ? row_ids[r0] : r0;
    double acc = 0.0;
    if (tid == 0) acc = (beta == 0.0) ? 0.0 : beta * y[row];
    for (int w = p0; w < p1; w += CAP) {
      const int wend = min(w + CAP, p1);
      for (int idx = w + tid; idx < wend; idx += BLK) {
        double pr = val[idx] * x[col[idx]];
        if (alpha != 1.0) pr = pr * alpha;
        prod[idx - w] = pr;
      }
      __syncthreads();
      if (tid == 0)
        for (int p = 0; p < wend - w; ++p) acc = acc + prod[p];
      __syncthreads();
    }
    if (tid == 0) y[row] = acc;
  }
}

// Host-side row split: greedy chunks of consecutive (compacted) rows whose stored entries, counted from
// the 2-aligned start, fit `cap` products; a row longer than that is a chunk on its own.
inline void pa_build_chunks(const int32_t *crp, int64_t nc, int cap, int max_rows, std::vector<int32_t> &chunk_row,
                            int64_t *n_long) {
  chunk_row.clear();
  chunk_row.push_back(0);
  *n_long = 0;
  int64_t r = 0;
  while (r < nc) {
    const int64_t base = crp[r] & ~1;
    int64_t e = r + 1;
    if ((int64_t)crp[e] - base > cap) {
      ++*n_long;
    } else {
      while (e < nc && (int64_t)crp[e + 1] - base <= cap && e - r < max_rows) ++e;
    }
    chunk_row.push_back((int32_t)e);
    r = e;
  }
}

// Host-side c16 encoding of every chunk (multi-threaded over chunks). win has n_chunks*PA_C16_WINDOWS entries;
// win[c*16] = -1 marks a chunk that keeps 32-bit columns (too many windows, or a long row).
inline int64_t pa_encode_col16(const int32_t *crp, const int32_t *col, const std::vector<int32_t> &chunk_row, int cap,
                               uint16_t *c16, int32_t *win, int n_threads) {
  const int64_t n_chunks = (int64_t)chunk_row.size() - 1;
  std::vector<int64_t> fallback(n_threads > 0 ? n_threads : 1, 0);
  auto work = [&](int t, int T) {
    for (int64_t c = n_chunks * t / T; c < n_chunks * (t + 1) / T; ++c) {
      int32_t *w = win + c * PA_C16_WINDOWS;
      for (int s = 0; s < PA_C16_WINDOWS; ++s) w[s] = 0;
      const int64_t p0 = crp[chunk_row[c]], p1 = crp[chunk_row[c + 1]];
      bool ok = (p1 - (p0 & ~1)) <= cap;
      int n = 0;
      int32_t tags[PA_C16_WINDOWS];
      for (int64_t p = p0; ok && p < p1; ++p) {
        const int32_t tag = col[p] >> 12;
        int s = 0;
        while (s < n && tags[s] != tag) ++s;
        if (s == n) {
          if (n == PA_C16_WINDOWS) { ok = false; break; }
          tags[n++] = tag;
        }
        c16[p] = (uint16_t)((s << 12) | (col[p] & 4095));
      }
      if (ok) for (int s = 0; s < n; ++s) w[s] = tags[s] << 12;
      else { w[0] = -1; ++fallback[t]; }
    }
  };
  if (n_threads <= 1) work(0, 1);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < n_threads; ++t) th.emplace_back(work, t, n_threads);
    for (auto &x : th) x.join();
  }
  int64_t nf = 0;
  for (auto v : fallback) nf += v;
  return nf;
}

#endif
