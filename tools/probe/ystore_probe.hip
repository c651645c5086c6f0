// ystore_probe.hip -- is there a DETERMINISTIC home for the result vector of the 27-point product?
// (round 2, VERDICT "Next" #3).  The product kernel's time depends on which allocations hold the value stream and y
// (DESIGN.md section 3).  This probe times, in one process and interleaved:
//   A  plain hipMalloc pairs (what the box gives by default: the spread to beat)
//   B  y in hipExtMallocWithFlags memory: uncached, fine-grained
//   C  values and y carved out of ONE physically contiguous allocation (hipDeviceMallocContiguous), y at a ladder of
//      fixed offsets -- if the time is a function of the offset only, the same on every arena and every box, the placement
//      lottery can become a rule
//   D  the same inside one hipMemCreate handle mapped at a reserved address (VMM)
//   E  the kernel with one dwordx4 store per two rows (EPI 11) on the pairs of A
//   make -C tools/probe ystore_probe && tools/probe/ystore_probe [n=256]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
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
__global__ void k_cmp(const double *a, const double *b, long n, unsigned long long *bad) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && a[i] != b[i]) atomicAdd(bad, 1ull);
}

// A stand-in for the product's memory behaviour with no matrix behind it: block b streams 12 KiB (768 value pairs) of `rd`
// and writes 56 doubles to `wr` (the 27-point operator's 1 : 27 ratio), dealt to the XCDs like the product's chunks.
__global__ __launch_bounds__(256) void k_rank_probe(const d2 *__restrict__ rd, int n_blocks, int per_xcd, double *__restrict__ wr) {
  const int b = blockIdx.x;
  const int blk = (b & 7) * per_xcd + (b >> 3);
  if (blk >= n_blocks || (b >> 3) >= per_xcd) return;
  const d2 *p = rd + (size_t)blk * 768;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) { const d2 v = __builtin_nontemporal_load(p + k * 256 + threadIdx.x); s += v.x + v.y; }
  if (threadIdx.x < 56) __builtin_nontemporal_store(s, wr + (size_t)blk * 56 + threadIdx.x);
  else if (s == 123.456) wr[(size_t)blk * 56] = s;      // (keeps the other lanes' loads alive; the buffer is zeros)
}
__global__ void k_axpby_p(double *__restrict__ y, const double *__restrict__ x, long n, double a, double b) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = a * x[i] + b * y[i];
}
__global__ void k_wsum_p(double *__restrict__ w, const double *__restrict__ x, const double *__restrict__ y, long n) {
  long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long stride = (long)gridDim.x * blockDim.x;
  for (; i < n; i += stride) w[i] = x[i] + y[i];
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 256;
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
  const size_t vbytes = sizeof(double) * (nnz + 8), ybytes = sizeof(double) * (nrows + 2);
  int *d_rp, *d_col; double *d_val, *d_x, *d_y;
  CK(hipMalloc(&d_rp, sizeof(int) * (nrows + 1))); CK(hipMalloc(&d_col, sizeof(int) * (nnz + 8)));
  CK(hipMalloc(&d_val, vbytes)); CK(hipMalloc(&d_x, ybytes)); CK(hipMalloc(&d_y, ybytes));
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
#define LAUNCH(EPI_, val_, y_)                                                                                                      \
  hipLaunchKernelGGL((k_spmv_rowsplit<BLK, NPT, true, false, 1, EPI_>), dim3(cpx * 8), dim3(BLK), 0, 0, d_rp, dc32,                 \
                     (const unsigned short *)nullptr, (const int *)nullptr, ddesc, ddel, val_, d_x, y_, dc, (const int *)nullptr, nch, \
                     cpx, 1.0, 0.0, (double *)nullptr, (const double *)nullptr, (const double *)nullptr)
  auto run = [&](const double *val, double *y, int epi = 0, int reps = 10) {
    auto go = [&]() { if (epi == 11) LAUNCH(11, val, y); else if (epi == 7) LAUNCH(7, val, y); else if (epi == 8) LAUNCH(8, val, y);
                      else if (epi == 12) LAUNCH(12, val, y); else if (epi == 13) LAUNCH(13, val, y); else LAUNCH(0, val, y); };
    for (int w = 0; w < 2; ++w) go();
    CK(hipEventRecord(e0, 0));
    for (int w = 0; w < reps; ++w) go();
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
  };
  for (int w = 0; w < 4; ++w) run(d_val, d_y, 0, 20);     // clocks up
  printf("27-pt %d^3: nnz %ld, %d chunks; value stream %.2f GB, y %.1f MB\n", n, nnz, nch, vbytes / 1e9, ybytes / 1e6);
  printf("kernel without its y store: %.4f ms\n", run(d_val, d_y, 7));
  // the dwordx4 variant gives the same bits?
  {
    double *y2; CK(hipMalloc(&y2, ybytes));
    unsigned long long *bad; CK(hipMalloc(&bad, 8)); CK(hipMemset(bad, 0, 8));
    run(d_val, d_y, 0, 1); run(d_val, y2, 11, 1);
    hipLaunchKernelGGL(k_cmp, dim3((nrows + 255) / 256), dim3(256), 0, 0, d_y, y2, nrows, bad);
    unsigned long long hb; CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
    printf("EPI 11 (dwordx4 store, two rows per lane): %llu rows differ from the shipped kernel\n", hb);
    CK(hipFree(y2));
  }

  // ---- L: one large physically contiguous arena; values at its start, y on a ladder through it; then y at the start and
  //         the values on the ladder.  Is the product's time a function of the PHYSICAL distance (a DRAM rank / stack-id
  //         boundary would show as a period of many GiB)?   ystore_probe 256 ladder <arena GiB> <step MiB>
  if (argc > 2 && std::string(argv[2]) == "ladder") {
    const size_t M = (size_t)1 << 20, G = (size_t)1 << 30;
    size_t want = (argc > 3 ? (size_t)atol(argv[3]) : 128) * G;
    const size_t step = (argc > 4 ? (size_t)atol(argv[4]) : 2048) * M;
    size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot));
    printf("free %.1f of %.1f GiB\n", fr / (double)G, tot / (double)G);
    char *a = nullptr;
    while (want >= 8 * G) {
      hipError_t e = hipExtMallocWithFlags((void **)&a, want, hipDeviceMallocContiguous);
      if (e == hipSuccess) break;
      (void)hipGetLastError(); a = nullptr;
      printf("contiguous %.0f GiB failed: %s\n", want / (double)G, hipGetErrorString(e));
      want -= 16 * G;
    }
    if (!a) { printf("no contiguous arena\n"); return 0; }
    printf("contiguous arena of %.0f GiB at %p; values at its start, y at offset ->  ms   |  y at its start, values at offset -> ms\n", want / (double)G, (void *)a);
    const size_t vspan = (vbytes + 2 * M - 1) / (2 * M) * (2 * M);
    CK(hipMemcpy(a, d_val, vbytes, hipMemcpyDeviceToDevice));
    std::vector<float> t1, t2; std::vector<size_t> o1, o2;
    for (size_t o = vspan; o + ybytes <= want; o += step) { o1.push_back(o); t1.push_back(run((double *)a, (double *)(a + o))); }
    // second half: y at the start (over the first bytes of the values: they are re-copied further along)
    for (size_t o = ((ybytes + 2 * M - 1) / (2 * M)) * 2 * M; o + vbytes <= want; o += step) {
      CK(hipMemcpy(a + o, d_val, vbytes, hipMemcpyDeviceToDevice));
      o2.push_back(o); t2.push_back(run((double *)(a + o), (double *)a));
    }
    for (size_t i = 0; i < std::max(o1.size(), o2.size()); ++i) {
      if (i < o1.size()) printf("  y @ %7.2f GiB: %.4f", o1[i] / (double)G, t1[i]); else printf("  %26s", "");
      if (i < o2.size()) printf("   |  values @ %7.2f GiB: %.4f", o2[i] / (double)G, t2[i]);
      printf("\n");
    }
    // x as well?  x at the far end, y next to the values
    CK(hipMemcpy(a, d_val, vbytes, hipMemcpyDeviceToDevice));
    double *xs = d_x;
    double *xfar = (double *)(a + want - ((ybytes + 2 * M - 1) / (2 * M)) * 2 * M);
    CK(hipMemcpy(xfar, d_x, ybytes, hipMemcpyDeviceToDevice));
    d_x = xfar;
    printf("x at the far end of the arena, y right after the values: %.4f; y in hipMalloc: %.4f\n", run((double *)a, (double *)(a + vspan)), run((double *)a, d_y));
    d_x = xs;
    return 0;
  }
  // ---- MAP: the class map of one contiguous arena from the stand-in kernel (cells of `cell` GiB: read stream in the
  //          reference cell, write stream in the cell under test), checked against the real product on the same cells;
  //          then the placement a library would derive from it, and the BLAS-1 kernels across classes.
  if (argc > 2 && std::string(argv[2]) == "map") {
    const size_t M = (size_t)1 << 20, G = (size_t)1 << 30;
    size_t want = (argc > 3 ? (size_t)atol(argv[3]) : 128) * G;
    const size_t cell = (argc > 4 ? (size_t)atol(argv[4]) : 2048) * M;
    char *a = nullptr;
    while (want >= 8 * G) {
      if (hipExtMallocWithFlags((void **)&a, want, hipDeviceMallocContiguous) == hipSuccess) break;
      (void)hipGetLastError(); a = nullptr; want -= 16 * G;
    }
    if (!a) { printf("no contiguous arena\n"); return 0; }
    const int ncell = (int)(want / cell);
    const size_t rd_bytes = (size_t)1 << 30;
    const int nb = (int)(rd_bytes / 12288), per_xcd = (nb + 7) / 8;
    auto probe = [&](int rc, int wc) {        // read stream at the start of cell rc, write stream in the second half of cell wc
      const d2 *rd = (const d2 *)(a + rc * cell);
      double *wr = (double *)(a + wc * cell + cell / 2);
      hipLaunchKernelGGL(k_rank_probe, dim3(per_xcd * 8), dim3(256), 0, 0, rd, nb, per_xcd, wr);
      CK(hipEventRecord(e0, 0));
      for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(k_rank_probe, dim3(per_xcd * 8), dim3(256), 0, 0, rd, nb, per_xcd, wr);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      return ms / 3;
    };
    CK(hipMemset(a, 0, rd_bytes));
    const auto m0 = std::chrono::steady_clock::now();
    std::vector<float> p0(ncell), p1(ncell, 0.f);
    for (int c = 0; c < ncell; ++c) p0[c] = probe(0, c);
    const float lo = *std::min_element(p0.begin(), p0.end());
    int ref2 = -1;
    for (int c = 1; c + 1 < ncell; ++c) if (p0[c] < 1.04f * lo && p0[c + 1] < 1.04f * lo) { ref2 = c; break; }
    std::vector<int> cls(ncell, -1);
    if (ref2 >= 0) {
      CK(hipMemset(a + ref2 * cell, 0, rd_bytes));
      for (int c = 0; c < ncell; ++c) p1[c] = probe(ref2, c);
      const float lo1 = *std::min_element(p1.begin(), p1.end());
      for (int c = 0; c < ncell; ++c) {
        const bool s0 = p0[c] > 1.08f * lo, s1 = p1[c] > 1.08f * lo1, f0 = p0[c] < 1.04f * lo, f1 = p1[c] < 1.04f * lo1;
        cls[c] = (s0 && f1) ? 0 : (f0 && s1) ? 1 : (f0 && f1) ? 2 : -1;
      }
    }
    printf("map of %d cells of %.1f GiB built in %.1f ms (two passes of the stand-in kernel, 1 GiB read stream); second reference = cell %d\n",
           ncell, cell / (double)G, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - m0).count(), ref2);
    // the real product on the same cells: values at the start of the arena (cells 0-1), y in each cell
    CK(hipMemcpy(a, d_val, vbytes, hipMemcpyDeviceToDevice));
    printf("cell: stand-in vs cell 0 | stand-in vs cell %d | class | product with the values at the arena's start and y in this cell\n", ref2);
    for (int c = 0; c < ncell; ++c) {
      float t = 0;
      if ((size_t)c * cell >= vbytes) t = run((double *)a, (double *)(a + c * cell + cell / 2), 0, 3);
      printf("  %3d  %.4f | %.4f | %2d | %.4f\n", c, p0[c], p1[c], cls[c], t);
    }
    // the placement: values in the first run of class-0 cells that holds them, y and x in the first class-1 cell, then class 2
    auto first_run = [&](int k, size_t bytes, int from) {
      const int need = (int)((bytes + cell - 1) / cell);
      for (int c = from; c + need <= ncell; ++c) { bool ok = true; for (int j = 0; j < need; ++j) ok = ok && cls[c + j] == k; if (ok) return c; }
      return -1;
    };
    const int cv = first_run(0, vbytes, 0), c1 = first_run(1, ybytes * 2, 0), c2 = first_run(2, ybytes * 2, 0);
    printf("placement: values in cell %d (class 0), vectors in cell %d (class 1) / cell %d (class 2)\n", cv, c1, c2);
    if (cv >= 0 && c1 >= 0) {
      double *v = (double *)(a + cv * cell);
      if (cv != 0) CK(hipMemcpy(v, d_val, vbytes, hipMemcpyDeviceToDevice));
      double *y1 = (double *)(a + c1 * cell), *x1 = (double *)(a + c1 * cell + ((ybytes + 2 * M - 1) / (2 * M)) * 2 * M);
      CK(hipMemcpy(x1, d_x, ybytes, hipMemcpyDeviceToDevice));
      double *xs = d_x;
      printf("  product, y in class 1, x where hipMalloc put it: %.4f\n", run(v, y1));
      d_x = x1;
      printf("  product, y and x both in class 1:               %.4f\n", run(v, y1));
      if (c2 >= 0) {
        double *y2 = (double *)(a + c2 * cell);
        printf("  product, y in class 2, x in class 1:            %.4f\n", run(v, y2));
      }
      d_x = xs;
      for (int round = 0; round < 3; ++round)
        printf("  store flavours, y in class 1 (round %d): nt %.4f | plain %.4f | sc1 %.4f | sc0 sc1 %.4f | no store %.4f\n", round,
               run(v, y1, 0, 20), run(v, y1, 8, 20), run(v, y1, 12, 20), run(v, y1, 13, 20), run(v, y1, 7, 20));
      printf("  product, y in class 0 (next to the values):     %.4f\n", run(v, (double *)(a + cv * cell + ((vbytes + 2 * M - 1) / (2 * M)) * 2 * M)));
      // BLAS-1 across classes: y = a*x + b*y (read 2, write 1) and w = x + y (read 2, write 1, distinct)
      auto tb = [&](auto f) { f(); CK(hipEventRecord(e0, 0)); for (int w = 0; w < 5; ++w) f(); CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); return ms / 5; };
      const long nv = (long)1 << 27;                               // 1 GiB vectors
      auto cellp = [&](int c, int slot) { return (double *)(a + c * cell) + (size_t)slot * 0; };
      const int cA = first_run(0, 3 * G, 0), cB = first_run(1, 3 * G, 0), cC = c2 >= 0 ? first_run(2, 3 * G, 0) : -1;
      if (cA >= 0 && cB >= 0) {
        double *A0 = cellp(cA, 0), *A1 = A0 + nv, *A2 = A1 + nv, *B0 = cellp(cB, 0), *B1 = B0 + nv, *C0 = cC >= 0 ? cellp(cC, 0) : B1;
        const int g = 4096;
        printf("  axpby y(A)=x(A):   %.4f ms | y(A)=x(B): %.4f | w = x + y with w,x,y in AAA %.4f  ABB %.4f  ABC %.4f  (1 GiB vectors: 3 GiB of traffic each)\n",
               tb([&] { hipLaunchKernelGGL(k_axpby_p, dim3(g), dim3(256), 0, 0, A0, A1, nv, 2.0, 0.5); }),
               tb([&] { hipLaunchKernelGGL(k_axpby_p, dim3(g), dim3(256), 0, 0, A0, B0, nv, 2.0, 0.5); }),
               tb([&] { hipLaunchKernelGGL(k_wsum_p, dim3(g), dim3(256), 0, 0, A0, A1, A2, nv); }),
               tb([&] { hipLaunchKernelGGL(k_wsum_p, dim3(g), dim3(256), 0, 0, A0, B0, B1, nv); }),
               tb([&] { hipLaunchKernelGGL(k_wsum_p, dim3(g), dim3(256), 0, 0, A0, B0, C0, nv); }));
      }
    }
    return 0;
  }
  // ---- L3: how many classes?  Map y against values at the arena's start, move the values into the first "other" region
  //          and map y again: a third class would show as a region that is fast against BOTH value positions.
  if (argc > 2 && std::string(argv[2]) == "ladder3") {
    const size_t M = (size_t)1 << 20, G = (size_t)1 << 30;
    size_t want = (argc > 3 ? (size_t)atol(argv[3]) : 128) * G;
    const size_t step = (argc > 4 ? (size_t)atol(argv[4]) : 4096) * M;
    char *a = nullptr;
    while (want >= 8 * G) {
      if (hipExtMallocWithFlags((void **)&a, want, hipDeviceMallocContiguous) == hipSuccess) break;
      (void)hipGetLastError(); a = nullptr; want -= 16 * G;
    }
    if (!a) { printf("no contiguous arena\n"); return 0; }
    const size_t vspan = (vbytes + 2 * M - 1) / (2 * M) * (2 * M);
    CK(hipMemcpy(a, d_val, vbytes, hipMemcpyDeviceToDevice));
    std::vector<size_t> offs; std::vector<float> t0;
    for (size_t o = vspan; o + ybytes <= want; o += step) { offs.push_back(o); t0.push_back(run((double *)a, (double *)(a + o))); }
    float lo = *std::min_element(t0.begin(), t0.end()), hi = *std::max_element(t0.begin(), t0.end());
    size_t vpos = 0;
    for (size_t i = 0; i + 3 < offs.size(); ++i) if (t0[i] < 0.5f * (lo + hi) && t0[i + 1] < 0.5f * (lo + hi) && t0[i + 2] < 0.5f * (lo + hi)) { vpos = offs[i + 1]; break; }
    printf("contiguous arena of %.0f GiB; pass 1 values at 0; pass 2 values at %.2f GiB; pass 3: x moved next to y as well\n", want / (double)G, vpos / (double)G);
    std::vector<float> t1(offs.size(), 0.f), t2(offs.size(), 0.f);
    if (vpos) {
      CK(hipMemcpy(a + vpos, d_val, vbytes, hipMemcpyDeviceToDevice));
      for (size_t i = 0; i < offs.size(); ++i) {
        if (offs[i] + ybytes > vpos && offs[i] < vpos + vspan) continue;     // inside the moved values
        t1[i] = run((double *)(a + vpos), (double *)(a + offs[i]));
      }
      // x in the same place class as y: x right behind y (same region), values at the start again
      CK(hipMemcpy(a, d_val, vbytes, hipMemcpyDeviceToDevice));
      double *xs = d_x;
      for (size_t i = 0; i < offs.size(); ++i) {
        if (offs[i] + 2 * ((ybytes + 2 * M - 1) / (2 * M)) * 2 * M > want) continue;
        double *xn = (double *)(a + offs[i] + ((ybytes + 2 * M - 1) / (2 * M)) * 2 * M);
        CK(hipMemcpy(xn, xs, ybytes, hipMemcpyDeviceToDevice));
        d_x = xn;
        t2[i] = run((double *)a, (double *)(a + offs[i]));
      }
      d_x = xs;
    }
    for (size_t i = 0; i < offs.size(); ++i) printf("  y @ %7.2f GiB: %.4f | %.4f | %.4f\n", offs[i] / (double)G, t0[i], t1[i], t2[i]);
    return 0;
  }
  // ---- W: the walk a library could do: values / x where hipMalloc puts them; then rungs of [spacer | candidate], both
  //         physically contiguous allocations, each candidate timed; everything printed.  ystore_probe 256 walk <stride GiB> <rungs>
  if (argc > 2 && std::string(argv[2]) == "walk") {
    const size_t G = (size_t)1 << 30;
    const size_t stride = (argc > 3 ? (size_t)atol(argv[3]) : 8) * G;
    const int rungs = argc > 4 ? atoi(argv[4]) : 16;
    const size_t cand_bytes = 2 * G;
    const float t_nostore = run(d_val, d_y, 7);
    printf("no-store %.4f; default y (hipMalloc) %.4f\n", t_nostore, run(d_val, d_y));
    std::vector<void *> held;
    for (int contiguous = 1; contiguous >= 0; --contiguous) {
      const auto w0 = std::chrono::steady_clock::now();
      for (int r = 0; r < rungs; ++r) {
        void *cand = nullptr, *sp = nullptr;
        const auto a0 = std::chrono::steady_clock::now();
        hipError_t e = contiguous ? hipExtMallocWithFlags(&cand, cand_bytes, hipDeviceMallocContiguous) : hipMalloc(&cand, cand_bytes);
        if (e != hipSuccess) { (void)hipGetLastError(); printf("  rung %d: candidate allocation failed\n", r); break; }
        const double alloc_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - a0).count();
        const float t = run(d_val, (double *)cand, 0, 3);
        const float t_end = run(d_val, (double *)((char *)cand + cand_bytes - ybytes), 0, 3);
        printf("  %s rung %2d (%.0f GiB walked) cand %p: %.4f (start) %.4f (end)  ratio to no-store %.3f  [alloc %.2f ms]\n", contiguous ? "contiguous" : "hipMalloc ",
               r, r * (stride + cand_bytes) / (double)G, cand, t, t_end, t / t_nostore, alloc_ms);
        held.push_back(cand);
        const auto s0 = std::chrono::steady_clock::now();
        e = contiguous ? hipExtMallocWithFlags(&sp, stride, hipDeviceMallocContiguous) : hipMalloc(&sp, stride);
        if (e != hipSuccess) { (void)hipGetLastError(); printf("  rung %d: spacer allocation failed\n", r); break; }
        if (r == 0) printf("  (spacer of %.0f GiB allocated in %.2f ms)\n", stride / (double)G, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - s0).count());
        held.push_back(sp);
      }
      printf("  walk took %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - w0).count());
      const auto f0 = std::chrono::steady_clock::now();
      for (void *q : held) CK(hipFree(q));
      held.clear();
      printf("  frees took %.1f ms\n", std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - f0).count());
    }
    return 0;
  }
  // ---- A: plain hipMalloc pairs ------------------------------------------------------------------------------------
  std::vector<double *> V(1, d_val), Y(1, d_y);
  for (int k = 1; k < 4; ++k) {
    double *yn; CK(hipMalloc(&yn, ybytes)); Y.push_back(yn);
    if (k < 3) { double *v; CK(hipMalloc(&v, vbytes)); CK(hipMemcpy(v, d_val, vbytes, hipMemcpyDeviceToDevice)); V.push_back(v); }
  }
  for (int round = 0; round < 2; ++round) {
    printf("A round %d: rows = value copy, columns = y allocation (hipMalloc); then the same with the dwordx4 store\n", round);
    for (size_t i = 0; i < V.size(); ++i) {
      printf("  val %p |", (void *)V[i]);
      for (size_t j = 0; j < Y.size(); ++j) printf(" %.4f", run(V[i], Y[j]));
      printf(" |");
      for (size_t j = 0; j < Y.size(); ++j) printf(" %.4f", run(V[i], Y[j], 11));
      printf("\n");
    }
  }
  // ---- B: y in uncached / fine-grained device memory ---------------------------------------------------------------
  {
    const unsigned flags[] = {hipDeviceMallocUncached, hipDeviceMallocFinegrained};
    const char *names[] = {"uncached", "fine-grained"};
    for (int f = 0; f < 2; ++f)
      for (int k = 0; k < 2; ++k) {
        double *yn = nullptr;
        hipError_t e = hipExtMallocWithFlags((void **)&yn, ybytes, flags[f]);
        if (e != hipSuccess) { (void)hipGetLastError(); printf("B %s y: allocation failed (%s)\n", names[f], hipGetErrorString(e)); continue; }
        printf("B y %-12s #%d %p:", names[f], k, (void *)yn);
        for (size_t i = 0; i < V.size(); ++i) printf(" %.4f", run(V[i], yn));
        printf(" | dwordx4:");
        for (size_t i = 0; i < V.size(); ++i) printf(" %.4f", run(V[i], yn, 11));
        printf("\n");
      }
    // and the VALUES in uncached memory (reads that bypass the caches' allocation policy), y plain
    double *vn = nullptr;
    hipError_t e = hipExtMallocWithFlags((void **)&vn, vbytes, hipDeviceMallocUncached);
    if (e == hipSuccess) {
      CK(hipMemcpy(vn, d_val, vbytes, hipMemcpyDeviceToDevice));
      printf("B values uncached %p:", (void *)vn);
      for (size_t j = 0; j < Y.size(); ++j) printf(" %.4f", run(vn, Y[j]));
      printf("\n");
      CK(hipFree(vn));
    } else (void)hipGetLastError();
  }
  // ---- C / D: one arena, values at a fixed place, y on a ladder of fixed offsets ------------------------------------
  const size_t M = (size_t)1 << 20, G = (size_t)1 << 30;
  const size_t vspan = (vbytes + 2 * M - 1) / (2 * M) * (2 * M), yspan = (ybytes + 2 * M - 1) / (2 * M) * (2 * M);
  // arena = [front: 2 GiB of y slots | values | back: 6 GiB of y slots]
  const size_t front = 2 * G, back = 6 * G, arena = front + vspan + back;
  const long offs[] = {-(long)(2 * G), -(long)G, -(long)(256 * M), -(long)yspan, 0, (long)(64 * M), (long)(256 * M), (long)(512 * M), (long)G,
                       (long)(G + 512 * M), (long)(2 * G), (long)(3 * G), (long)(4 * G), (long)(5 * G), (long)(6 * G - yspan)};
  auto ladder = [&](char *a, const char *what) {
    double *v = (double *)(a + front);
    CK(hipMemcpy(v, d_val, vbytes, hipMemcpyDeviceToDevice));
    printf("%s arena %p: y offset from the end of the values (negative: before their start) -> ms | dwordx4\n", what, (void *)a);
    for (long o : offs) {
      double *y = o < 0 ? (double *)(a + front + o) : (double *)(a + front + vspan + o);
      printf("   %+8.0f MiB: %.4f | %.4f\n", o / (double)M, run(v, y), run(v, y, 11));
    }
    printf("   with a separate hipMalloc y: %.4f %.4f\n", run(v, Y[0]), run(v, Y[1]));
  };
  for (int k = 0; k < 2; ++k) {
    char *a = nullptr;
    hipError_t e = hipExtMallocWithFlags((void **)&a, arena, hipDeviceMallocContiguous);
    if (e != hipSuccess) { (void)hipGetLastError(); printf("C contiguous allocation of %.1f GiB failed: %s\n", arena / (double)G, hipGetErrorString(e)); break; }
    ladder(a, "C contiguous");
    if (k == 1) CK(hipFree(a));    // (the first stays allocated so that the second is a different physical range)
  }
  for (int k = 0; k < 2; ++k) {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) { (void)hipGetLastError(); printf("D no VMM granularity\n"); break; }
    const size_t sz = (arena + gran - 1) / gran * gran;
    hipMemGenericAllocationHandle_t h;
    hipError_t e = hipMemCreate(&h, sz, &prop, 0);
    if (e != hipSuccess) { (void)hipGetLastError(); printf("D hipMemCreate(%.1f GiB) failed: %s\n", sz / (double)G, hipGetErrorString(e)); break; }
    void *va = nullptr;
    CK(hipMemAddressReserve(&va, sz, 0, nullptr, 0));
    CK(hipMemMap(va, sz, 0, h, 0));
    hipMemAccessDesc acc = {}; acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    CK(hipMemSetAccess(va, sz, &acc, 1));
    printf("D granularity %zu KiB\n", gran >> 10);
    ladder((char *)va, "D one hipMemCreate handle");
  }
  printf("A again (after the arenas):");
  for (size_t i = 0; i < V.size(); ++i) for (size_t j = 0; j < Y.size(); ++j) printf(" %.4f", run(V[i], Y[j]));
  printf("\n");
  return 0;
}
