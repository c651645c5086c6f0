// spmv_probe.hip -- A/B harness for the row-split SpMV kernel on the HPCG 27-point matrix (one part,
// n^3 rows).  Development tool: builds the matrix on the device, runs interleaved rounds of kernel
// variants / ablations, prints median and min time and the algorithmic GB/s.  Not part of libpa_hip.so.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../partitionedarrays.jl_amd/csrc -I../../include spmv_probe.hip -o spmv_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <map>
#include <string>
#include <vector>

#include "pa_spmv_probe_hooks.h"   // the product kernel + the lab's hooks

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

// ---- device-side generator of the single-part 27-point CSR (0-based) -------------------------------
__global__ void k_gen(int n, const int *__restrict__ rp, int *__restrict__ col, double *__restrict__ val) {
  const long row = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long nrows = (long)n * n * n;
  if (row >= nrows) return;
  const int ix = row % n, iy = (row / n) % n, iz = row / ((long)n * n);
  int p = rp[row];
  for (int sz = -1; sz <= 1; ++sz) { if (iz + sz < 0 || iz + sz >= n) continue;
    for (int sy = -1; sy <= 1; ++sy) { if (iy + sy < 0 || iy + sy >= n) continue;
      for (int sx = -1; sx <= 1; ++sx) { if (ix + sx < 0 || ix + sx >= n) continue;
        const long c = row + (long)sz * n * n + (long)sy * n + sx;
        col[p] = (int)c; val[p] = (c == row) ? 26.0 : -1.0; ++p; } } }
}
__global__ void k_hashx(double *x, long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = (double)((unsigned)((unsigned long)(i + 1) * 2654435761ul)) / 4294967296.0;
}

// ---- ablations ----------------------------------------------------------------------------------------
template <int BLK, int NPT, bool NT, bool GATHER>
__global__ __launch_bounds__(BLK) void k_stream(const int *__restrict__ col, const double *__restrict__ val,
                                                const double *__restrict__ x, double *__restrict__ y, long nnz) {
  const long base = (long)blockIdx.x * BLK * NPT;
  double acc = 0.0;
#pragma unroll
  for (int k = 0; k < NPT / 2; ++k) {
    const long idx = base + (long)(k * BLK + threadIdx.x) * 2;
    if (idx < nnz) {
      d2 v = pa_stream_load<NT>(reinterpret_cast<const d2 *>(val + idx));
      i2 c = pa_stream_load<NT>(reinterpret_cast<const i2 *>(col + idx));
      if (GATHER) acc += v.x * x[c.x] + v.y * x[c.y];
      else acc += v.x * (double)c.x + v.y * (double)c.y;
    }
  }
  if (acc == 123.456) y[blockIdx.x] = acc;   // keep the loads alive, (almost) never store
}

__global__ void k_copy(const d2 *__restrict__ a, d2 *__restrict__ b, long n2) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n2; i += stride) b[i] = a[i];
}
__global__ void k_readsum(const d2 *__restrict__ a, double *out, long n2) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  double s = 0;
  for (; i < n2; i += stride) { d2 v = __builtin_nontemporal_load(a + i); s += v.x + v.y; }
  if (s == 123.456) out[0] = s;
}

// ---- ablation copy of the product kernel: XCD map on/off and pieces switched off (timing only) -------
//   ABL bit0: reduce reads one product per row instead of walking the row   bit1: no y store
//       bit4: nontemporal y store   bit5: y store into a 2 MiB window   bit6: (unused)
//       bit2: no LDS write / barrier                                         bit3: no x gather

// ---- persistent variant with the row sums parked in LDS and written in bursts ----------------------------------------
// grid = 8 XCDs x WPX workgroups; workgroup i of XCD k walks chunks k*cpx + i, + WPX, + 2*WPX, ... (so that at any moment
// the workgroups of an XCD work on consecutive chunks, as in the one-workgroup-per-chunk kernel); the row sums of KF
// consecutive iterations stay in LDS (ROWS_MAX rows per chunk) and are stored together.
template <int NPT, int KF, int ROWS_MAX>
__global__ __launch_bounds__(256) void k_spmv_persist(
    const int *__restrict__ crp, const int *__restrict__ pdesc, const int *__restrict__ pdelta,
    const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y,
    const int *__restrict__ chunk_row, int n_chunks, int cpx, int wpx) {
  constexpr int BLK = 256, CAP = BLK * NPT;
  __shared__ __attribute__((aligned(16))) double prod[CAP];
  __shared__ double ybuf[KF * ROWS_MAX];
  __shared__ int yrow0[KF], ynr[KF];
  const int tid = threadIdx.x;
  const int xcd = blockIdx.x & 7, wi = blockIdx.x >> 3;
  int kslot = 0;
  for (int j = wi; j < cpx; j += wpx) {
    const int chunk = xcd * cpx + j;
    const bool live = chunk < n_chunks;
    if (live) {
      const int r0 = chunk_row[chunk], r1 = chunk_row[chunk + 1];
      const int p0 = crp[r0], p1 = crp[r1];
      const int base = p0 & ~1;
      int ra = 0, re = 0;
      if (r0 + tid < r1) { ra = crp[r0 + tid]; re = crp[r0 + tid + 1]; }
      const int last = max((p1 - 1) & ~1, 0);
      const int *d = pdesc + chunk * PA_PDESC_INTS;
      const int q1 = d[1], q2 = d[2], q3 = d[3];
      const int s0r = d[4], s1r = d[5], s2r = d[6], s3r = d[7];
      const int L0 = d[8], L1 = d[9], L2 = d[10], L3 = d[11];
      const int pt0 = d[12], pt1 = d[13], pt2 = d[14], pt3 = d[15];
      const unsigned M0 = d[16], M1 = d[17], M2 = d[18], M3 = d[19];
      const int lane = tid & 63;
      const int dA = pdelta[((lane >> 5) ? pt1 : pt0) * PA_PAT_MAXLEN + (lane & 31)];
      const int dB = pdelta[((lane >> 5) ? pt3 : pt2) * PA_PAT_MAXLEN + (lane & 31)];
      d2 v[NPT / 2];
#pragma unroll
      for (int k = 0; k < NPT / 2; ++k) {
        const int idx = min(base + (k * BLK + tid) * 2, last);
        v[k] = pa_stream_load<true>(reinterpret_cast<const d2 *>(val + idx));
      }
      const int nq = p1 - p0 - 1;
#pragma unroll
      for (int k = 0; k < NPT / 2; ++k) {
        const int idx = min(base + (k * BLK + tid) * 2, last);
        const int c0 = pa_pattern_col<false>(idx - p0, nq, q1, q2, q3, L0, L1, L2, L3, s0r, s1r, s2r, s3r, M0, M1, M2, M3, dA, dB);
        const int c1 = pa_pattern_col<false>(idx + 1 - p0, nq, q1, q2, q3, L0, L1, L2, L3, s0r, s1r, s2r, s3r, M0, M1, M2, M3, dA, dB);
        d2 pr;
        pr.x = v[k].x * x[c0];
        pr.y = v[k].y * x[c1];
        *reinterpret_cast<d2 *>(&prod[(k * BLK + tid) * 2]) = pr;
      }
      __syncthreads();
      if (r0 + tid < r1) {
        double acc = 0.0;
        const int a = ra - base, e = re - base;
#pragma unroll 4
        for (int p = a; p < e; ++p) acc = acc + prod[p];
        ybuf[kslot * ROWS_MAX + tid] = acc;
      }
      if (tid == 0) { yrow0[kslot] = r0; ynr[kslot] = r1 - r0; }
      ++kslot;
    }
    __syncthreads();
    if (kslot == KF || j + wpx >= cpx) {           // burst: KF chunks' row sums leave together
      for (int s = 0; s < kslot; ++s) {
        const int nr = ynr[s];
        if (tid < nr) y[yrow0[s] + tid] = ybuf[s * ROWS_MAX + tid];
      }
      kslot = 0;
      __syncthreads();
    }
  }
}

template <int BLK, int NPT, bool XCD, int ABL, int SL = 0>
__global__ __launch_bounds__(BLK) void k_spmv_abl(
    const int *__restrict__ crp, const int *__restrict__ col, const double *__restrict__ val,
    const double *__restrict__ x, double *__restrict__ y, const int *__restrict__ chunk_row,
    int n_chunks, int chunks_per_xcd) {
  constexpr int CAP = BLK * NPT;
  __shared__ double prod[CAP];
  const int tid = threadIdx.x;
  const int b = blockIdx.x;
  const int chunk = XCD ? (b & 7) * chunks_per_xcd + (b >> 3) : b;
  if (chunk >= n_chunks || (XCD && (b >> 3) >= chunks_per_xcd)) return;
  const int r0 = chunk_row[chunk], r1 = chunk_row[chunk + 1];
  const int p0 = crp[r0], p1 = crp[r1];
  const int base = p0 & ~1;
  int ra = 0, re = 0;
  if (r0 + tid < r1) { ra = crp[r0 + tid]; re = crp[r0 + tid + 1]; }
  d2 v[NPT / 2]; i2 c[NPT / 2];
  const int last = max((p1 - 1) & ~1, 0);
#pragma unroll
  for (int k = 0; k < NPT / 2; ++k) {
    const int idx = min(base + (k * BLK + tid) * 2, last);
    v[k] = pa_stream_load<true>(reinterpret_cast<const d2 *>(val + idx));
    c[k] = pa_stream_load<true>(reinterpret_cast<const i2 *>(col + idx));
  }
  double keep = 0.0;
#pragma unroll
  for (int k = 0; k < NPT / 2; ++k) {
    d2 pr;
    if (ABL & 8) { pr.x = v[k].x * (double)c[k].x; pr.y = v[k].y * (double)c[k].y; }
    else { pr.x = v[k].x * x[c[k].x]; pr.y = v[k].y * x[c[k].y]; }
    if (ABL & 4) keep += pr.x + pr.y;
    else *reinterpret_cast<d2 *>(&prod[(k * BLK + tid) * 2]) = pr;
  }
  if (!(ABL & 4)) __syncthreads();
  for (int r = r0 + tid; r < r1; r += BLK) {
    if (r != r0 + tid) { ra = crp[r]; re = crp[r + 1]; }
    double acc = keep;
    const int a = ra - base, e = re - base;
    if (ABL & 1) { if (!(ABL & 4)) acc += prod[a]; }
    else {
#pragma unroll 4
      for (int p = a; p < e; ++p) acc = acc + prod[p];
    }
    if (ABL & 8192) {      // burst emulation: only every SL-th dispatch wave of 2048 blocks stores, SL times as much
      if (((b >> 11) % SL) == SL - 1) {
        const long o = ((long)(r0 / 64) * 64 * SL) % 16000000l;
        if (ABL & 65536) { for (int g = 0; g < SL; ++g) y[(o + (long)g * 500009l * 8) % 16000000l + (r - r0)] = acc; }   // same bursts in time, scattered in space
        else for (int g = 0; g < SL; ++g) y[o + (long)g * (r1 - r0) + (r - r0)] = acc;
      }
    }
    else if (ABL & 32768) {   // scalar stores: the row sums leave through the scalar data cache, not the TA/TCP path
      const unsigned lo = (unsigned)__double_as_longlong(acc), hi = (unsigned)(__double_as_longlong(acc) >> 32);
      const int wbase = __builtin_amdgcn_readfirstlane(r - (tid & 63));
      const int n = min(64, __builtin_amdgcn_readfirstlane(r1) - wbase);
      double *yb = y + wbase;
      for (int i = 0; i < n; ++i) {
        const unsigned long long v = ((unsigned long long)__builtin_amdgcn_readlane(hi, i) << 32) | __builtin_amdgcn_readlane(lo, i);
        double *p = yb + i;
        asm volatile("s_store_dwordx2 %0, %1, 0x0" ::"s"(v), "s"(p) : "memory");
        if ((i & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
    }
    else if (ABL & 16384) {   // time-gated store: every block stores at the next multiple of SL x 10 ns
      while ((wall_clock64() % SL) > SL / 8) __builtin_amdgcn_s_sleep(2);
      y[r] = acc;
    }
    else if (ABL & 1024) reinterpret_cast<float *>(y)[r] = (float)acc;                 // half the bytes
    else if (ABL & 2048) unsafeAtomicAdd(&y[r], acc);                            // L2 atomic instead of a store
    else if (ABL & 4096) y[(long)chunk * 512 + (r - r0)] = acc;                   // every chunk writes into its own 4 KiB
    else if (ABL & 512) {            // staggered store: at most 256 B (4 write requests) leave the CU at a time
      const int lane = tid & 63, wave = tid >> 6;
      for (int w = 0; w < wave; ++w) { __builtin_amdgcn_s_sleep(SL); __builtin_amdgcn_s_sleep(SL); }
      if (lane < 32) y[r] = acc;
      __builtin_amdgcn_s_sleep(SL);
      if (lane >= 32) y[r] = acc;
    }
    else if (ABL & 2) { if (acc == 123.456) y[r] = acc; }
    else if (ABL & 16) __builtin_nontemporal_store(acc, &y[r]);
    else if (ABL & 32) y[r & 0x3ffff] = acc;                       // 2 MiB region: stays in L2/MALL
    else if (ABL & 64) __hip_atomic_store(&y[r], acc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (ABL & 128) prod[CAP - 1 - (r - r0)] = acc;
    else if (ABL & 256) prod[r - r0] = acc;   // safe: row r's products start at or after slot r-r0 only if rows are non-empty (27-pt)
    else y[r] = acc;
  }
  if (ABL & 256) {   // wide (16 B/lane) stores by wave 0; needs r0 even and every row non-empty
    __syncthreads();
    const int nr = r1 - r0;
    if (tid < 64) for (int i = tid * 2; i < nr; i += 128) {
      if (i + 1 < nr) *reinterpret_cast<d2 *>(&y[r0 + i]) = *reinterpret_cast<d2 *>(&prod[i]);
      else y[r0 + i] = prod[i];
    }
  }
  if (ABL & 128) {   // repack through LDS (top of prod[] is free after the reduce) and store with one wave
    __syncthreads();
    const int nr = r1 - r0;
    if (tid < 64) for (int i = tid; i < nr; i += 64) y[r0 + i] = prod[CAP - 1 - i];
  }
}


__global__ void k_cmp(const double *a, const double *b, long n, unsigned long long *bad) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && __double_as_longlong(a[i]) != __double_as_longlong(b[i])) atomicAdd(bad, 1ull);
}

// {first row, its row pointer} pairs of a row split: what k_spmv_rowsplit reads per chunk (pa_spmv_kernel.h, round 5)
static int *dev_chunk_rp(const std::vector<int32_t> &cr, const std::vector<int> &rp) {
  std::vector<int> h(2 * cr.size());
  for (size_t k = 0; k < cr.size(); ++k) { h[2 * k] = cr[k]; h[2 * k + 1] = rp[cr[k]]; }
  int *d; CK(hipMalloc(&d, sizeof(int) * h.size()));
  CK(hipMemcpy(d, h.data(), sizeof(int) * h.size(), hipMemcpyHostToDevice));
  return d;
}

struct Variant { std::string name; std::function<void()> run; double bytes; std::vector<float> ms; };

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 256;
  const int rounds = argc > 2 ? atoi(argv[2]) : 7;
  const long nrows = (long)n * n * n;
  std::vector<int> rp(nrows + 1);
  {
    long k = 0; rp[0] = 0; long row = 0;
    for (int iz = 0; iz < n; ++iz) { const int cz = 3 - (iz == 0) - (iz == n - 1);
      for (int iy = 0; iy < n; ++iy) { const int cy = 3 - (iy == 0) - (iy == n - 1);
        for (int ix = 0; ix < n; ++ix) { const int cx = 3 - (ix == 0) - (ix == n - 1);
          k += (long)cx * cy * cz; rp[++row] = (int)k; } } }
  }
  const long nnz = rp[nrows];
  printf("27-pt %d^3: rows %ld nnz %ld\n", n, nrows, nnz);
  int *d_rp, *d_col; double *d_val, *d_x, *d_y, *d_y2;
  CK(hipMalloc(&d_rp, sizeof(int) * (nrows + 1))); CK(hipMalloc(&d_col, sizeof(int) * (nnz + 8)));
  CK(hipMalloc(&d_val, sizeof(double) * (nnz + 8))); CK(hipMalloc(&d_x, sizeof(double) * (nrows + 2)));
  CK(hipMalloc(&d_y, sizeof(double) * nrows)); CK(hipMalloc(&d_y2, sizeof(double) * std::max<long>(nrows, (nnz / 2000 + 8) * 512)));
  double *d_ybig; CK(hipMalloc(&d_ybig, sizeof(double) * nrows + (80l << 20)));
  printf("ptrs val %p col %p x %p y %p y2 %p ybig %p rp %p\n", (void*)d_val, (void*)d_col, (void*)d_x, (void*)d_y, (void*)d_y2, (void*)d_ybig, (void*)d_rp);
  CK(hipMemset(d_col + nnz, 0, 32)); CK(hipMemset(d_val + nnz, 0, 64));
  CK(hipMemcpy(d_rp, rp.data(), sizeof(int) * (nrows + 1), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_gen, dim3((nrows + 255) / 256), dim3(256), 0, 0, n, d_rp, d_col, d_val);
  hipLaunchKernelGGL(k_hashx, dim3((nrows + 255) / 256), dim3(256), 0, 0, d_x, nrows);
  CK(hipDeviceSynchronize());

  const double bytes_spmv = (double)nnz * 12 + (nrows + 1) * 4.0 + nrows * 16.0;
  std::vector<Variant> V;
  auto chunks_for = [&](int cap, int **d_chunks, int *nch, int max_rows = 4096) {
    std::vector<int32_t> cr; int64_t nl;
    pa_build_chunks(rp.data(), nrows, cap, max_rows, cr, &nl);
    *nch = (int)cr.size() - 1;
    CK(hipMalloc(d_chunks, sizeof(int) * cr.size()));
    CK(hipMemcpy(*d_chunks, cr.data(), sizeof(int) * cr.size(), hipMemcpyHostToDevice));
  };
  // host copy of the columns for the c16 encoder
  std::vector<int> hcol(nnz);
  CK(hipMemcpy(hcol.data(), d_col, sizeof(int) * nnz, hipMemcpyDeviceToHost));
#define ADD_SPMV4(BLK, NPT, NT, C16)                                                                          \
  { std::vector<int32_t> cr; int64_t nl; pa_build_chunks(rp.data(), nrows, BLK * NPT, 4096, cr, &nl);           \
    const int nch = (int)cr.size() - 1; int *dc; CK(hipMalloc(&dc, sizeof(int) * cr.size()));                   \
    CK(hipMemcpy(dc, cr.data(), sizeof(int) * cr.size(), hipMemcpyHostToDevice)); int *drp = dev_chunk_rp(cr, rp); (void)drp;                               \
    unsigned short *d16 = nullptr; int *dwin = nullptr;                                                         \
    if (C16) { std::vector<uint16_t> c16(nnz + 8, 0); std::vector<int32_t> win((size_t)nch * 16, 0);            \
      const int64_t nf = pa_encode_col16(rp.data(), hcol.data(), cr, BLK * NPT, c16.data(), win.data(), 32);    \
      printf("c16<%d,%d>: %d chunks, %lld fall back to 32-bit columns\n", BLK, NPT, nch, (long long)nf);        \
      CK(hipMalloc(&d16, 2 * (nnz + 8))); CK(hipMalloc(&dwin, 4 * win.size()));                                 \
      CK(hipMemcpy(d16, c16.data(), 2 * (nnz + 8), hipMemcpyHostToDevice));                                     \
      CK(hipMemcpy(dwin, win.data(), 4 * win.size(), hipMemcpyHostToDevice)); }                                 \
    const int cpx = (nch + 7) / 8;                                                                              \
    V.push_back({"spmv<" #BLK "," #NPT "," #NT ",c16=" #C16 ">", [=]() {                                        \
      hipLaunchKernelGGL((k_spmv_rowsplit<BLK, NPT, NT, C16, 0>), dim3(cpx * 8), dim3(BLK), 0, 0, d_rp, d_col, d16, dwin, \
                         (const int *)nullptr, (const int *)nullptr, d_val, d_x, (C16 ? d_y2 : d_y), drp, (const int *)nullptr, nch, cpx, 1.0, 0.0, (double *)nullptr, (const double *)nullptr, (const double *)nullptr); }, bytes_spmv, {}}); }
  ADD_SPMV4(256, 4, true, false)
  ADD_SPMV4(256, 6, true, true)
  ADD_SPMV4(256, 6, true, false)
  ADD_SPMV4(256, 10, true, true)
  ADD_SPMV4(192, 8, true, true)
  // ---- allocation-attribute experiments: matrix stream and/or y in "uncached" (MTYPE_UC) memory -----------
  {
    std::vector<int32_t> cr; int64_t nl; pa_build_chunks(rp.data(), nrows, 256 * 6, 4096, cr, &nl);
    const int nch = (int)cr.size() - 1; int *dc; CK(hipMalloc(&dc, sizeof(int) * cr.size()));
    CK(hipMemcpy(dc, cr.data(), sizeof(int) * cr.size(), hipMemcpyHostToDevice)); int *drp = dev_chunk_rp(cr, rp); (void)drp;
    std::vector<uint16_t> c16(nnz + 8, 0); std::vector<int32_t> win((size_t)nch * 16, 0);
    pa_encode_col16(rp.data(), hcol.data(), cr, 256 * 6, c16.data(), win.data(), 32);
    unsigned short *d16, *u16; int *dwin; double *uval, *uy;
    CK(hipMalloc(&d16, 2 * (nnz + 8))); CK(hipMalloc(&dwin, 4 * win.size()));
    CK(hipMemcpy(d16, c16.data(), 2 * (nnz + 8), hipMemcpyHostToDevice));
    CK(hipMemcpy(dwin, win.data(), 4 * win.size(), hipMemcpyHostToDevice));
    CK(hipExtMallocWithFlags((void **)&u16, 2 * (nnz + 8), hipDeviceMallocUncached));
    CK(hipExtMallocWithFlags((void **)&uval, 8 * (nnz + 8), hipDeviceMallocUncached));
    CK(hipExtMallocWithFlags((void **)&uy, 8 * nrows, hipDeviceMallocUncached));
    double *fy; CK(hipExtMallocWithFlags((void **)&fy, 8 * nrows, hipDeviceMallocFinegrained));
    CK(hipMemcpy(u16, d16, 2 * (nnz + 8), hipMemcpyDeviceToDevice));
    CK(hipMemcpy(uval, d_val, 8 * (nnz + 8), hipMemcpyDeviceToDevice));
    const int cpx = (nch + 7) / 8;
#define ADD_ATTR(NAME, VALP, C16P, YP, NT)                                                                   \
    V.push_back({NAME, [=]() {                                                                                 \
      hipLaunchKernelGGL((k_spmv_rowsplit<256, 6, NT, true, false>), dim3(cpx * 8), dim3(256), 0, 0, d_rp, d_col, C16P, dwin, \
                         (const int *)nullptr, (const int *)nullptr, VALP, d_x, YP, drp, (const int *)nullptr, nch, cpx, 1.0, 0.0, (double *)nullptr, (const double *)nullptr, (const double *)nullptr); }, bytes_spmv, {}});
    ADD_ATTR("c16 base (cached all, nt)", d_val, d16, d_y2, true)
    ADD_ATTR("c16 matrix uncached, nt", uval, u16, d_y2, true)
    ADD_ATTR("c16 matrix uncached, plain", uval, u16, d_y2, false)
    ADD_ATTR("c16 y uncached, nt", d_val, d16, uy, true)
    ADD_ATTR("c16 matrix+y uncached, nt", uval, u16, uy, true)
    ADD_ATTR("c16 y finegrained, nt", d_val, d16, fy, true)
  }

  // ---- row-pattern mode of the product kernel ------------------------------------------------------------------
#define ADD_PAT(BLK, NPT, NT)                                                                                     \
  { std::vector<int32_t> cr; int64_t nl; pa_build_chunks(rp.data(), nrows, BLK * NPT, 4096, cr, &nl);           \
    const int nch = (int)cr.size() - 1; std::vector<int32_t> pdesc, pdelta;                                     \
    const int64_t ng = pa_encode_patterns(rp.data(), hcol.data(), nullptr, nrows, cr, BLK * NPT, pdesc, pdelta, 32);     \
    printf("pattern<%d,%d>: %d chunks, %lld with a descriptor, %zu patterns\n", BLK, NPT, nch, (long long)ng, pdelta.size() / 32); \
    int *dc, *ddesc, *ddel; CK(hipMalloc(&dc, 4 * cr.size())); CK(hipMalloc(&ddesc, 4 * pdesc.size())); CK(hipMalloc(&ddel, 4 * pdelta.size())); \
    CK(hipMemcpy(dc, cr.data(), 4 * cr.size(), hipMemcpyHostToDevice)); int *drp = dev_chunk_rp(cr, rp); (void)drp; CK(hipMemcpy(ddesc, pdesc.data(), 4 * pdesc.size(), hipMemcpyHostToDevice)); \
    CK(hipMemcpy(ddel, pdelta.data(), 4 * pdelta.size(), hipMemcpyHostToDevice));                               \
    const int cpx = (nch + 7) / 8;                                                                              \
    V.push_back({"pattern<" #BLK "," #NPT ",nt=" #NT ">c16=true", [=]() {                                                  \
      hipLaunchKernelGGL((k_spmv_rowsplit<BLK, NPT, NT, false, true>), dim3(cpx * 8), dim3(BLK), 0, 0, d_rp, d_col, \
                         (const unsigned short *)nullptr, (const int *)nullptr, ddesc, ddel, d_val, d_x, d_y2, drp,  \
                         (const int *)nullptr, nch, cpx, 1.0, 0.0, (double *)nullptr, (const double *)nullptr, (const double *)nullptr); }, bytes_spmv, {}}); }
  ADD_PAT(256, 8, true)
  ADD_PAT(256, 8, false)
  ADD_PAT(256, 4, true)
  ADD_PAT(256, 6, true)
  ADD_PAT(128, 8, true)
  ADD_PAT(128, 4, true)
  ADD_PAT(512, 4, true)
  ADD_PAT(192, 8, true)
#define ADD_PAT_EPI(BLK, NPT, EPI)                                                                                \
  { std::vector<int32_t> cr; int64_t nl; pa_build_chunks(rp.data(), nrows, BLK * NPT, 4096, cr, &nl);           \
    const int nch = (int)cr.size() - 1; std::vector<int32_t> pdesc, pdelta;                                     \
    pa_encode_patterns(rp.data(), hcol.data(), nullptr, nrows, cr, BLK * NPT, pdesc, pdelta, 32);               \
    int *dc, *ddesc, *ddel; CK(hipMalloc(&dc, 4 * cr.size())); CK(hipMalloc(&ddesc, 4 * pdesc.size())); CK(hipMalloc(&ddel, 4 * pdelta.size())); \
    CK(hipMemcpy(dc, cr.data(), 4 * cr.size(), hipMemcpyHostToDevice)); int *drp = dev_chunk_rp(cr, rp); (void)drp; CK(hipMemcpy(ddesc, pdesc.data(), 4 * pdesc.size(), hipMemcpyHostToDevice)); \
    CK(hipMemcpy(ddel, pdelta.data(), 4 * pdelta.size(), hipMemcpyHostToDevice));                               \
    const int cpx = (nch + 7) / 8;                                                                              \
    V.push_back({"pattern<" #BLK "," #NPT ",epi=" #EPI ">", [=]() {                                             \
      hipLaunchKernelGGL((k_spmv_rowsplit<BLK, NPT, true, false, 1, EPI>), dim3(cpx * 8), dim3(BLK), 0, 0, d_rp, d_col, \
                         (const unsigned short *)nullptr, (const int *)nullptr, ddesc, ddel, d_val, d_x, d_y2, drp,  \
                         (const int *)nullptr, nch, cpx, 1.0, 0.0, (double *)nullptr, (const double *)nullptr, (const double *)nullptr); }, bytes_spmv, {}}); }
  ADD_PAT_EPI(256, 6, 0)
  ADD_PAT_EPI(256, 6, 7)
  ADD_PAT_EPI(256, 8, 7)
  ADD_PAT_EPI(256, 6, 8)
  ADD_PAT_EPI(256, 6, 9)
  ADD_PAT_EPI(256, 6, 10)
#define ADD_PAT_UNR(UNR)                                                                                          \
  { std::vector<int32_t> cr; int64_t nl; pa_build_chunks(rp.data(), nrows, 1536, 4096, cr, &nl);                \
    const int nch = (int)cr.size() - 1; std::vector<int32_t> pdesc, pdelta;                                     \
    pa_encode_patterns(rp.data(), hcol.data(), nullptr, nrows, cr, 1536, pdesc, pdelta, 32);                    \
    int *dc, *ddesc, *ddel; CK(hipMalloc(&dc, 4 * cr.size())); CK(hipMalloc(&ddesc, 4 * pdesc.size())); CK(hipMalloc(&ddel, 4 * pdelta.size())); \
    CK(hipMemcpy(dc, cr.data(), 4 * cr.size(), hipMemcpyHostToDevice)); int *drp = dev_chunk_rp(cr, rp); (void)drp; CK(hipMemcpy(ddesc, pdesc.data(), 4 * pdesc.size(), hipMemcpyHostToDevice)); \
    CK(hipMemcpy(ddel, pdelta.data(), 4 * pdelta.size(), hipMemcpyHostToDevice));                               \
    const int cpx = (nch + 7) / 8;                                                                              \
    V.push_back({"pattern<256,6,unroll=" #UNR ">", [=]() {                                                      \
      hipLaunchKernelGGL((k_spmv_rowsplit<256, 6, true, false, 1, 0, false, UNR>), dim3(cpx * 8), dim3(256), 0, 0, d_rp, d_col, \
                         (const unsigned short *)nullptr, (const int *)nullptr, ddesc, ddel, d_val, d_x, d_y2, drp,  \
                         (const int *)nullptr, nch, cpx, 1.0, 0.0, (double *)nullptr, (const double *)nullptr, (const double *)nullptr); }, bytes_spmv, {}}); }
  ADD_PAT_UNR(1)
  ADD_PAT_UNR(2)
  ADD_PAT_UNR(4)
  ADD_PAT_UNR(9)
  ADD_PAT_UNR(16)
  ADD_PAT_UNR(27)
  ADD_PAT_UNR(4)
#define ADD_PAT_ROWS(MAXR)                                                                                         \
  { std::vector<int32_t> cr; int64_t nl; pa_build_chunks(rp.data(), nrows, 1536, MAXR, cr, &nl);                \
    const int nch = (int)cr.size() - 1; std::vector<int32_t> pdesc, pdelta;                                     \
    pa_encode_patterns(rp.data(), hcol.data(), nullptr, nrows, cr, 1536, pdesc, pdelta, 32);                    \
    int *dc, *ddesc, *ddel; CK(hipMalloc(&dc, 4 * cr.size())); CK(hipMalloc(&ddesc, 4 * pdesc.size())); CK(hipMalloc(&ddel, 4 * pdelta.size())); \
    CK(hipMemcpy(dc, cr.data(), 4 * cr.size(), hipMemcpyHostToDevice)); int *drp = dev_chunk_rp(cr, rp); (void)drp; CK(hipMemcpy(ddesc, pdesc.data(), 4 * pdesc.size(), hipMemcpyHostToDevice)); \
    CK(hipMemcpy(ddel, pdelta.data(), 4 * pdelta.size(), hipMemcpyHostToDevice));                               \
    const int cpx = (nch + 7) / 8;                                                                              \
    V.push_back({"pattern<256,6,maxrows=" #MAXR ">", [=]() {                                                    \
      hipLaunchKernelGGL((k_spmv_rowsplit<256, 6, true, false, 1, 0>), dim3(cpx * 8), dim3(256), 0, 0, d_rp, d_col, \
                         (const unsigned short *)nullptr, (const int *)nullptr, ddesc, ddel, d_val, d_x, d_y2, drp,  \
                         (const int *)nullptr, nch, cpx, 1.0, 0.0, (double *)nullptr, (const double *)nullptr, (const double *)nullptr); }, bytes_spmv, {}}); }
  ADD_PAT_ROWS(56)
  ADD_PAT_ROWS(48)
#define ADD_PAT_ROWS8(MAXR)                                                                                        \
  { std::vector<int32_t> cr; int64_t nl; pa_build_chunks(rp.data(), nrows, 2048, MAXR, cr, &nl, 16);            \
    const int nch = (int)cr.size() - 1; std::vector<int32_t> pdesc, pdelta;                                     \
    pa_encode_patterns(rp.data(), hcol.data(), nullptr, nrows, cr, 2048, pdesc, pdelta, 32);                    \
    int *dc, *ddesc, *ddel; CK(hipMalloc(&dc, 4 * cr.size())); CK(hipMalloc(&ddesc, 4 * pdesc.size())); CK(hipMalloc(&ddel, 4 * pdelta.size())); \
    CK(hipMemcpy(dc, cr.data(), 4 * cr.size(), hipMemcpyHostToDevice)); int *drp = dev_chunk_rp(cr, rp); (void)drp; CK(hipMemcpy(ddesc, pdesc.data(), 4 * pdesc.size(), hipMemcpyHostToDevice)); \
    CK(hipMemcpy(ddel, pdelta.data(), 4 * pdelta.size(), hipMemcpyHostToDevice));                               \
    const int cpx = (nch + 7) / 8;                                                                              \
    V.push_back({"pattern<256,8,maxrows=" #MAXR ",align16>", [=]() {                                            \
      hipLaunchKernelGGL((k_spmv_rowsplit<256, 8, true, false, 1, 0>), dim3(cpx * 8), dim3(256), 0, 0, d_rp, d_col, \
                         (const unsigned short *)nullptr, (const int *)nullptr, ddesc, ddel, d_val, d_x, d_y2, drp,  \
                         (const int *)nullptr, nch, cpx, 1.0, 0.0, (double *)nullptr, (const double *)nullptr, (const double *)nullptr); }, bytes_spmv, {}}); }
  ADD_PAT_ROWS8(64)
  ADD_PAT_ROWS8(72)
  ADD_PAT_ROWS8(4096)
#define ADD_PERSIST(NPT, KF, WPX)                                                                                 \
  { std::vector<int32_t> cr; int64_t nl; pa_build_chunks(rp.data(), nrows, 256 * NPT, 96, cr, &nl);             \
    const int nch = (int)cr.size() - 1; std::vector<int32_t> pdesc, pdelta;                                     \
    const int64_t ng = pa_encode_patterns(rp.data(), hcol.data(), nullptr, nrows, cr, 256 * NPT, pdesc, pdelta, 32, 4096, 1); \
    if (ng != nch) printf("persist: %lld of %d chunks have a descriptor -- variant skipped\n", (long long)ng, nch);  \
    else {                                                                                                        \
    int *dc, *ddesc, *ddel; CK(hipMalloc(&dc, 4 * cr.size())); CK(hipMalloc(&ddesc, 4 * pdesc.size())); CK(hipMalloc(&ddel, 4 * pdelta.size())); \
    CK(hipMemcpy(dc, cr.data(), 4 * cr.size(), hipMemcpyHostToDevice)); int *drp = dev_chunk_rp(cr, rp); (void)drp; CK(hipMemcpy(ddesc, pdesc.data(), 4 * pdesc.size(), hipMemcpyHostToDevice)); \
    CK(hipMemcpy(ddel, pdelta.data(), 4 * pdelta.size(), hipMemcpyHostToDevice));                               \
    const int cpx = (nch + 7) / 8;                                                                              \
    V.push_back({"persist<" #NPT ",KF=" #KF ",wpx=" #WPX ">c16=true", [=]() {                                   \
      hipLaunchKernelGGL((k_spmv_persist<NPT, KF, 96>), dim3(8 * WPX), dim3(256), 0, 0, d_rp, ddesc, ddel, d_val, d_x, d_y2, dc, \
                         nch, cpx, WPX); }, bytes_spmv, {}}); } }
  ADD_PERSIST(6, 1, 256)
  ADD_PERSIST(6, 4, 256)
  ADD_PERSIST(6, 8, 256)
  ADD_PERSIST(6, 16, 256)
  ADD_PERSIST(6, 8, 224)
  ADD_PERSIST(8, 8, 256)
  ADD_PERSIST(6, 32, 192)
  ADD_PAT(256, 12, true)
  ADD_PAT(256, 16, true)
  ADD_PAT(128, 16, true)
  ADD_PAT(512, 8, true)

#define ADD_ABLS(BLK, NPT, ABL, SL)                                                                          \
  { int *dc; int nch; chunks_for(BLK * NPT, &dc, &nch); const int cpx = (nch + 7) / 8;                       \
    V.push_back({"abl<" #BLK "," #NPT ",abl=" #ABL ",sleep=" #SL ">", [=]() {                                  \
      hipLaunchKernelGGL((k_spmv_abl<BLK, NPT, true, ABL, SL>), dim3(cpx * 8), dim3(BLK), 0, 0, d_rp, d_col, d_val, d_x, \
                         d_y2, dc, nch, cpx); }, bytes_spmv, {}}); }
  ADD_ABLS(256, 8, 0, 0)
  ADD_ABLS(256, 8, 2, 0)
  ADD_ABLS(256, 8, 32768, 0)
  ADD_ABLS(256, 8, 73728, 8)
  ADD_ABLS(256, 8, 73728, 16)
  ADD_ABLS(256, 8, 73728, 32)
  ADD_ABLS(256, 8, 16384, 50)
  ADD_ABLS(256, 8, 16384, 100)
  ADD_ABLS(256, 8, 16384, 200)
  ADD_ABLS(256, 8, 16384, 400)
  ADD_ABLS(256, 8, 16384, 1600)
  ADD_ABLS(256, 8, 8192, 2)
  ADD_ABLS(256, 8, 8192, 4)
  ADD_ABLS(256, 8, 8192, 8)
  ADD_ABLS(256, 8, 8192, 16)
  ADD_ABLS(256, 8, 8192, 32)
  {
    const long nb = (nnz + 2047) / 2048;
    V.push_back({"stream_only<256,8,nt>", [=]() { hipLaunchKernelGGL((k_stream<256, 8, true, false>), dim3(nb), dim3(256), 0, 0, d_col, d_val, d_x, d_y2, nnz); }, (double)nnz * 12, {}});
    V.push_back({"stream_only<256,8,plain>", [=]() { hipLaunchKernelGGL((k_stream<256, 8, false, false>), dim3(nb), dim3(256), 0, 0, d_col, d_val, d_x, d_y2, nnz); }, (double)nnz * 12, {}});
    V.push_back({"stream+gather<256,8,nt>", [=]() { hipLaunchKernelGGL((k_stream<256, 8, true, true>), dim3(nb), dim3(256), 0, 0, d_col, d_val, d_x, d_y2, nnz); }, (double)nnz * 12 + nrows * 8.0, {}});
    const long n2 = nnz / 2;
    V.push_back({"readsum(val) nt", [=]() { hipLaunchKernelGGL(k_readsum, dim3(2048), dim3(256), 0, 0, (const d2 *)d_val, d_y2, n2); }, (double)n2 * 16, {}});
  }
  {  // bitwise check of every full-SpMV variant against the first one
    unsigned long long *d_bad; CK(hipMalloc(&d_bad, 8));
    V[0].run(); CK(hipDeviceSynchronize());
    for (size_t i = 1; i < V.size(); ++i) {
      if (V[i].name.find("c16=true") == std::string::npos && V[i].name.find("abl=32768") == std::string::npos && (V[i].name.find("epi=") == std::string::npos || V[i].name.find("epi=7") != std::string::npos)) continue;
      CK(hipMemset(d_y2, 0xff, sizeof(double) * nrows)); CK(hipMemset(d_bad, 0, 8));
      V[i].run();
      hipLaunchKernelGGL(k_cmp, dim3((nrows + 255) / 256), dim3(256), 0, 0, d_y, d_y2, nrows, d_bad);
      unsigned long long bad; CK(hipMemcpy(&bad, d_bad, 8, hipMemcpyDeviceToHost));
      printf("check %-28s mismatches vs %s: %llu\n", V[i].name.c_str(), V[0].name.c_str(), bad);
    }
  }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int r = 0; r < rounds + 1; ++r)
    for (auto &v : V) {
      CK(hipEventRecord(e0, 0)); v.run(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) v.ms.push_back(ms);
    }
  // correctness of the spmv variants vs the first one
  printf("%-28s %9s %9s %10s %10s\n", "variant", "med ms", "min ms", "GB/s(med)", "GB/s(min)");
  for (auto &v : V) {
    std::sort(v.ms.begin(), v.ms.end());
    const float med = v.ms[v.ms.size() / 2], mn = v.ms[0];
    printf("%-28s %9.4f %9.4f %10.1f %10.1f\n", v.name.c_str(), med, mn, v.bytes / med / 1e6, v.bytes / mn / 1e6);
  }
  {
    // bitwise check of the last wave variant against the baseline kernel (run both again on intact data)
  }
  // note: the copy variant above overwrote part of val; do not reuse the matrix after this point
  return 0;
}
