// pa_spmv_kernel.h -- the row-split CSR SpMV kernel (K1/K2) and its host-side row split.
// Shared by pa_device.hip (the product) and probe/spmv_probe.hip (A/B tuning harness).
//
// Reference loops: spmv_csr! src/sparse_utils.jl:649-669; muladd! src/p_sparse_matrix.jl:2088.
// Must be compiled with -ffp-contract=off (one rounding per multiply and per add).
#ifndef PA_SPMV_KERNEL_H
#define PA_SPMV_KERNEL_H

#include <hip/hip_runtime.h>

#include <cstdint>
#include <vector>

typedef double d2 __attribute__((ext_vector_type(2)));
typedef int i2 __attribute__((ext_vector_type(2)));

template <bool NT, typename T>
__device__ __forceinline__ T pa_stream_load(const T *p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);  // read-once matrix stream: keep x in L2
  else return *p;
}

// y[row] = beta*y[row] + sum_p (val[p]*x[col[p]])*alpha, products summed in ascending p.
//   BLK  threads per workgroup, NPT stored entries per lane (chunk capacity CAP = BLK*NPT products in LDS),
//   NT   non-temporal matrix loads.
template <int BLK, int NPT, bool NT>
__global__ __launch_bounds__(BLK) void k_spmv_rowsplit(
    const int *__restrict__ crp, const int *__restrict__ col, const double *__restrict__ val,
    const double *__restrict__ x, double *__restrict__ y, const int *__restrict__ chunk_row,
    const int *__restrict__ row_ids, int n_chunks, int chunks_per_xcd, double alpha, double beta) {
  constexpr int CAP = BLK * NPT;
  static_assert(NPT % 2 == 0, "pairs");
  __shared__ double prod[CAP];
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int chunk = (b & 7) * chunks_per_xcd + (b >> 3);  // XCD-aware: block b sits on XCD b%8
  if (chunk >= n_chunks || (b >> 3) >= chunks_per_xcd) return;
  const int r0 = chunk_row[chunk];
  const int r1 = chunk_row[chunk + 1];
  const int p0 = crp[r0];
  const int p1 = crp[r1];
  const int base = p0 & ~1;  // 16-byte aligned value pairs

  if (p1 - base <= CAP) {
    // my first row's extent, fetched early so the latency hides under the matrix stream
    int ra = 0, re = 0;
    if (r0 + tid < r1) {
      ra = crp[r0 + tid];
      re = crp[r0 + tid + 1];
    }
    d2 v[NPT / 2];
    i2 c[NPT / 2];
    // Unconditional loads: lanes past the chunk's end re-read its last pair (same address => no extra
    // traffic) and their products are never summed.  A guarded load would make the compiler wait for
    // each load before issuing the next (one HBM round trip per k instead of one per chunk).
    const int last = max((p1 - 1) & ~1, 0);
#pragma unroll
    for (int k = 0; k < NPT / 2; ++k) {
      const int idx = min(base + (k * BLK + tid) * 2, last);
      v[k] = pa_stream_load<NT>(reinterpret_cast<const d2 *>(val + idx));
      c[k] = pa_stream_load<NT>(reinterpret_cast<const i2 *>(col + idx));
    }
#pragma unroll
    for (int k = 0; k < NPT / 2; ++k) {
      d2 pr;
      pr.x = v[k].x * x[c[k].x];
      pr.y = v[k].y * x[c[k].y];
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

#endif
