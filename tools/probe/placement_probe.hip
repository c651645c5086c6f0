// placement_probe.hip -- does the product kernel's time depend on WHERE hipMalloc put the value stream?
// 27-point 256^3 operator, the shipped kernel configuration (256 threads, 6 per lane, row patterns), one x / y pair,
// and the SAME values copied into (a) several separate allocations, (b) one arena at offsets of different alignment.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -I../../partitionedarrays.jl_amd/csrc -I../../include placement_probe.hip -o placement_probe
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "pa_spmv_probe_hooks.h"   // the product kernel + the lab's hooks

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1);} } while (0)

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
__global__ void k_readsum(const d2 *__restrict__ a, double *out, long n2) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  double s = 0;
  for (; i < n2; i += stride) { d2 v = __builtin_nontemporal_load(a + i); s += v.x + v.y; }
  if (s == 123.456) out[0] = s;
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 256;
  const int copies = argc > 2 ? atoi(argv[2]) : 6;
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
  const size_t vbytes = sizeof(double) * (nnz + 8);
  int *d_rp, *d_col; double *d_val, *d_x, *d_y;
  CK(hipMalloc(&d_rp, sizeof(int) * (nrows + 1))); CK(hipMalloc(&d_col, sizeof(int) * (nnz + 8)));
  CK(hipMalloc(&d_val, vbytes)); CK(hipMalloc(&d_x, sizeof(double) * (nrows + 2))); CK(hipMalloc(&d_y, sizeof(double) * nrows));
  CK(hipMemset(d_val + nnz, 0, 64));
  CK(hipMemcpy(d_rp, rp.data(), sizeof(int) * (nrows + 1), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(k_gen, dim3((nrows + 255) / 256), dim3(256), 0, 0, n, d_rp, d_col, d_val);
  hipLaunchKernelGGL(k_hashx, dim3((nrows + 255) / 256), dim3(256), 0, 0, d_x, nrows);
  CK(hipDeviceSynchronize());
  std::vector<int> hcol(nnz);
  CK(hipMemcpy(hcol.data(), d_col, sizeof(int) * nnz, hipMemcpyDeviceToHost));
  CK(hipFree(d_col));
  constexpr int BLK = 256, NPT = 6;
  std::vector<int32_t> cr, pdesc, pdelta; int64_t nl;
  pa_build_chunks(rp.data(), nrows, BLK * NPT, 4096, cr, &nl);
  const int nch = (int)cr.size() - 1;
  pa_encode_patterns(rp.data(), hcol.data(), nullptr, nrows, cr, BLK * NPT, pdesc, pdelta, 32);
  // the 8 chunks without a descriptor: give them 32-bit columns through the compacted stream
  std::vector<int32_t> c32(8, 0);
  for (int c = 0; c < nch; ++c)
    if (pdesc[(size_t)c * PA_PDESC_INTS] == 0) {
      const long b = rp[cr[c]] & ~1, e = rp[cr[c + 1]];
      pdesc[(size_t)c * PA_PDESC_INTS + 2] = (int32_t)((long)c32.size() - b);
      for (long p = b; p < e + 2 && p < nnz; ++p) c32.push_back(hcol[p]);
      while (c32.size() & 1) c32.push_back(0);
    }
  int *dc, *ddesc, *ddel, *dc32;
  CK(hipMalloc(&dc, 4 * cr.size())); CK(hipMalloc(&ddesc, 4 * pdesc.size())); CK(hipMalloc(&ddel, 4 * pdelta.size())); CK(hipMalloc(&dc32, 4 * c32.size() + 64));
  CK(hipMemcpy(dc, cr.data(), 4 * cr.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(ddesc, pdesc.data(), 4 * pdesc.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(ddel, pdelta.data(), 4 * pdelta.size(), hipMemcpyHostToDevice)); CK(hipMemcpy(dc32, c32.data(), 4 * c32.size(), hipMemcpyHostToDevice));
  const int cpx = (nch + 7) / 8;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto run = [&](const double *val, int reps) {
    for (int w = 0; w < 3; ++w)
      hipLaunchKernelGGL((k_spmv_rowsplit<BLK, NPT, true, false, 1>), dim3(cpx * 8), dim3(BLK), 0, 0, d_rp, dc32, (const unsigned short *)nullptr,
                         (const int *)nullptr, ddesc, ddel, val, d_x, d_y, dc, (const int *)nullptr, nch, cpx, 1.0, 0.0, (double *)nullptr,
                         (const double *)nullptr, (const double *)nullptr);
    CK(hipEventRecord(e0, 0));
    for (int w = 0; w < reps; ++w)
      hipLaunchKernelGGL((k_spmv_rowsplit<BLK, NPT, true, false, 1>), dim3(cpx * 8), dim3(BLK), 0, 0, d_rp, dc32, (const unsigned short *)nullptr,
                         (const int *)nullptr, ddesc, ddel, val, d_x, d_y, dc, (const int *)nullptr, nch, cpx, 1.0, 0.0, (double *)nullptr,
                         (const double *)nullptr, (const double *)nullptr);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
  };
  auto readsum = [&](const double *val, int reps) {
    hipLaunchKernelGGL(k_readsum, dim3(4096), dim3(256), 0, 0, (const d2 *)val, d_y, nnz / 2);
    CK(hipEventRecord(e0, 0));
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL(k_readsum, dim3(4096), dim3(256), 0, 0, (const d2 *)val, d_y, nnz / 2);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
  };
  for (int w = 0; w < 3; ++w) run(d_val, 20);     // clocks up
  printf("27-pt %d^3: nnz %ld, %d chunks; value stream %.2f GB\n", n, nnz, nch, vbytes / 1e9);
  if (argc > 3 && std::string(argv[3]) == "pmc") {   // counter mode (run under rocprofv3 --pmc): N copies, 2 launches each, event time per copy
    std::vector<double *> V(1, d_val);
    for (int k = 1; k < copies; ++k) { double *v; CK(hipMalloc(&v, vbytes)); CK(hipMemcpy(v, d_val, vbytes, hipMemcpyDeviceToDevice)); V.push_back(v); }
    for (int round = 0; round < 2; ++round)
      for (int k = 0; k < copies; ++k) printf("round %d copy %d: %.4f ms\n", round, k, run(V[k], 2));
    return 0;
  }
  if (argc > 2 && copies == 3) {   // y carved out of the allocation that holds the values: before them, after them, or apart
    const size_t ysz = ((sizeof(double) * nrows + ((size_t)2 << 20) - 1) >> 21) << 21;
    double *y_orig = d_y;
    printf("one allocation [y | values | y]: ms with y before the values, after them, and in the original separate allocation\n");
    for (int k = 0; k < 10; ++k) {
      char *a; CK(hipMalloc(&a, vbytes + 2 * ysz + ((size_t)4 << 20)));
      double *yb = (double *)a, *v = (double *)(a + ysz), *ya = (double *)(a + ysz + ((vbytes + ((size_t)2 << 20) - 1) >> 21 << 21));
      CK(hipMemcpy(v, d_val, vbytes, hipMemcpyDeviceToDevice));
      d_y = yb; const float t0 = run(v, 10);
      d_y = ya; const float t1 = run(v, 10);
      d_y = y_orig; const float t2 = run(v, 10);
      double *ys; CK(hipMalloc(&ys, sizeof(double) * nrows)); d_y = ys; const float t3 = run(v, 10);
      printf("allocation %d at %p: %.4f %.4f %.4f | y allocated right after it: %.4f\n", k, (void *)a, t0, t1, t2, t3);
    }
    return 0;
  }
  if (argc > 3 && copies == 2) {   // y map: result vectors allocated one after the other through the device memory
    const int NY = atoi(argv[3]);
    double *v1; CK(hipMalloc(&v1, vbytes)); CK(hipMemcpy(v1, d_val, vbytes, hipMemcpyDeviceToDevice));
    printf("y index: GiB allocated so far | ms with the original values | ms with a second copy\n");
    size_t total = 0;
    for (int j = 0; j < NY; ++j) {
      double *yn; if (hipMalloc(&yn, sizeof(double) * nrows) != hipSuccess) break;
      total += sizeof(double) * nrows;
      d_y = yn;
      const float a = run(d_val, 2), b = run(v1, 2);
      printf("%4d %7.2f %.4f %.4f %p\n", j, total / 1073741824.0, a, b, (void *)yn);
    }
    return 0;
  }
  if (argc > 3 && copies == 1) {   // ballast mode: hold b GiB, then time exact-size copies allocated after it
    const size_t G = (size_t)1 << 30;
    const size_t b = (size_t)atol(argv[3]);
    size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot));
    char *ballast = nullptr; if (b) CK(hipMalloc(&ballast, b * G));
    printf("free %.1f of %.1f GiB before; ballast %zu GiB:", fr / (double)G, tot / (double)G, b);
    std::vector<double *> V;
    for (int k = 0; k < 4; ++k) {
      double *v; CK(hipMalloc(&v, vbytes)); CK(hipMemcpy(v, d_val, vbytes, hipMemcpyDeviceToDevice)); V.push_back(v);
      printf(" %.4f", run(v, 20));
    }
    if (ballast) CK(hipFree(ballast));
    printf(" | after freeing the ballast:");
    for (int k = 0; k < 4; ++k) printf(" %.4f", run(V[k], 20));
    double *xn, *yn; CK(hipMalloc(&xn, sizeof(double) * (nrows + 2))); CK(hipMalloc(&yn, sizeof(double) * nrows));
    CK(hipMemcpy(xn, d_x, sizeof(double) * (nrows + 2), hipMemcpyDeviceToDevice));
    d_x = xn; d_y = yn;
    printf(" | with x,y allocated now:");
    for (int k = 0; k < 4; ++k) printf(" %.4f", run(V[k], 20));
    printf("\n");
    return 0;
  }
  if (argc > 2 && copies == 0) {   // values carved out of arenas of different sizes vs separate allocations, interleaved
    const size_t G = (size_t)1 << 30;
    const size_t extra[] = {0, 0, 4 * G, 0, 12 * G, 0, 28 * G, 0, 60 * G, 0, 0, 4 * G, 12 * G};
    for (size_t ex : extra) {
      char *a; CK(hipMalloc(&a, vbytes + ex));
      float t[3]; size_t offs[3] = {0, ex / 2 & ~((size_t)4095), ex};
      for (int k = 0; k < 3; ++k) {
        CK(hipMemcpy(a + offs[k], d_val, vbytes, hipMemcpyDeviceToDevice));
        t[k] = run((double *)(a + offs[k]), 20);
        if (ex == 0) { t[1] = t[2] = t[0]; break; }
      }
      printf("allocation of values + %2zu GiB at %p: values at the start %.4f, in the middle %.4f, at the end %.4f ms\n", ex / G, (void *)a, t[0], t[1], t[2]);
    }
    return 0;
  }
  if (argc > 2 && copies < 0) {   // what changes a kept copy's speed: frees?  new vectors?
    const int K = -copies;
    std::vector<double *> V(1, d_val);
    for (int k = 1; k < K; ++k) { double *v; CK(hipMalloc(&v, vbytes)); CK(hipMemcpy(v, d_val, vbytes, hipMemcpyDeviceToDevice)); V.push_back(v); }
    std::vector<float> t(K);
    for (int round = 0; round < 2; ++round) { printf("all %d copies alive:", K); for (int k = 0; k < K; ++k) { t[k] = run(V[k], 10); printf(" %.4f", t[k]); } printf("\n"); }
    const int best = (int)(std::min_element(t.begin(), t.end()) - t.begin()), worst = (int)(std::max_element(t.begin(), t.end()) - t.begin());
    for (int k = 0; k < K; ++k) if (k != best && k != worst && k != 0) CK(hipFree(V[k]));
    printf("after freeing the others: best(#%d) %.4f worst(#%d) %.4f original %.4f\n", best, run(V[best], 10), worst, run(V[worst], 10), run(V[0], 10));
    double *x0 = d_x, *y0 = d_y;
    for (int j = 0; j < 4; ++j) {
      double *xn, *yn; CK(hipMalloc(&xn, sizeof(double) * (nrows + 2))); CK(hipMalloc(&yn, sizeof(double) * nrows));
      CK(hipMemcpy(xn, x0, sizeof(double) * (nrows + 2), hipMemcpyDeviceToDevice));
      d_x = xn; d_y = yn;
      printf("new x,y #%d (%p %p): best %.4f worst %.4f original %.4f", j, (void *)xn, (void *)yn, run(V[best], 10), run(V[worst], 10), run(V[0], 10));
      d_x = x0; d_y = y0;
      printf(" | first x,y again: best %.4f worst %.4f original %.4f\n", run(V[best], 10), run(V[worst], 10), run(V[0], 10));
    }
    // launches separated by a device synchronisation (what a profiler with counters does)
    auto run_sync = [&](const double *val) {
      float tot = 0;
      for (int w = 0; w < 8; ++w) {
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL((k_spmv_rowsplit<BLK, NPT, true, false, 1>), dim3(cpx * 8), dim3(BLK), 0, 0, d_rp, dc32, (const unsigned short *)nullptr,
                           (const int *)nullptr, ddesc, ddel, val, d_x, d_y, dc, (const int *)nullptr, nch, cpx, 1.0, 0.0, (double *)nullptr,
                           (const double *)nullptr, (const double *)nullptr);
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (w >= 2) tot += ms;
      }
      return tot / 6;
    };
    printf("one launch at a time (sync between): best %.4f worst %.4f original %.4f\n", run_sync(V[best]), run_sync(V[worst]), run_sync(V[0]));
    printf("back to back again:                  best %.4f worst %.4f original %.4f\n", run(V[best], 10), run(V[worst], 10), run(V[0], 10));
    return 0;
  }
  // N value copies x M (x,y) pairs, allocated interleaved: the full matrix of times
  const int N = copies, M = copies;
  std::vector<double *> V(1, d_val), X(1, d_x), Y(1, d_y);
  for (int k = 1; k < std::max(N, M); ++k) {
    if (k < M) {
      double *xn, *yn; CK(hipMalloc(&xn, sizeof(double) * (nrows + 2))); CK(hipMalloc(&yn, sizeof(double) * nrows));
      CK(hipMemcpy(xn, d_x, sizeof(double) * (nrows + 2), hipMemcpyDeviceToDevice));
      X.push_back(xn); Y.push_back(yn);
    }
    if (k < N) { double *v; CK(hipMalloc(&v, vbytes)); CK(hipMemcpy(v, d_val, vbytes, hipMemcpyDeviceToDevice)); V.push_back(v); }
  }
  for (int round = 0; round < 2; ++round) {
    printf("round %d: rows = value copy, columns = (x,y) pair; then x of pair j with y of pair 0, then x of pair 0 with y of pair j\n", round);
    for (int i = 0; i < N; ++i) {
      printf("val %p |", (void *)V[i]);
      for (int j = 0; j < M; ++j) { d_x = X[j]; d_y = Y[j]; printf(" %.4f", run(V[i], 12)); }
      printf(" |");
      for (int j = 0; j < M; ++j) { d_x = X[j]; d_y = Y[0]; printf(" %.4f", run(V[i], 12)); }
      printf(" |");
      for (int j = 0; j < M; ++j) { d_x = X[0]; d_y = Y[j]; printf(" %.4f", run(V[i], 12)); }
      printf("\n");
    }
  }
  for (int j = 0; j < M; ++j) printf("pair %d: x %p y %p\n", j, (void *)X[j], (void *)Y[j]);
  return 0;
}
