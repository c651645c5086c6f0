// extent_probe.hip -- how physically contiguous extents behave on MI355X, for the on-demand arena (csrc/pa_arena.hip):
//   (a) what hipExtMallocWithFlags(hipDeviceMallocContiguous) costs by size,
//   (b) which memory class consecutive extents of E GiB land in (held all at once) -- i.e. how far a search has to walk
//       before it meets another class,
//   (c) whether a freed extent's memory comes back to the next request,
//   (d) whether a plain hipMalloc of a vector's size sits in one class.
// Classes are told apart with the arena's stand-in kernel: a 512 MiB read stream + a 1 : 27 write stream; the pair is
// ~11 % slower when both lie in one class.
//   ./extent_probe [extent GiB = 16] [extents = 14]
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef double d2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_probe(const d2 *__restrict__ rd, int n_blocks, int per_xcd, double *__restrict__ wr) {
  const int b = blockIdx.x;
  const int blk = (b & 7) * per_xcd + (b >> 3);
  if (blk >= n_blocks || (b >> 3) >= per_xcd) return;
  const d2 *p = rd + (size_t)blk * 768;
  double s = 0.0;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const d2 v = __builtin_nontemporal_load(p + k * 256 + threadIdx.x);
    s += v.x + v.y;
  }
  if (threadIdx.x < 56) __builtin_nontemporal_store(s, wr + (size_t)blk * 56 + threadIdx.x);
  else if (s == 123.456) wr[(size_t)blk * 56] = s;
}

static const size_t G = (size_t)1 << 30, CELL = (size_t)512 << 20;
static const int NB = (int)(CELL / 12288);
static const size_t WR = ((size_t)NB * 56 * 8 + 4095) / 4096 * 4096;
static hipEvent_t e0, e1;

static float pair_ms(const char *rd, char *wr) {
  const int per = (NB + 7) / 8;
  hipLaunchKernelGGL(k_probe, dim3(per * 8), dim3(256), 0, 0, (const d2 *)rd, NB, per, (double *)wr);
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < 3; ++r) hipLaunchKernelGGL(k_probe, dim3(per * 8), dim3(256), 0, 0, (const d2 *)rd, NB, per, (double *)wr);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / 3;
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  const int EG = argc > 1 ? atoi(argv[1]) : 16, NE = argc > 2 ? atoi(argv[2]) : 14;
  CK(hipSetDevice(0));
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  size_t fr = 0, tot = 0;
  CK(hipMemGetInfo(&fr, &tot));
  printf("free %.1f GiB of %.1f\n", fr / (double)G, tot / (double)G);
  // (a) cost of a contiguous allocation by size (freed right away)
  for (int g : {4, 8, 16, 32, 48, 64, 96}) {
    char *p = nullptr;
    const double t = now();
    hipError_t e = hipExtMallocWithFlags((void **)&p, (size_t)g * G, hipDeviceMallocContiguous);
    const double t1 = now();
    if (e != hipSuccess) { printf("(a) %3d GiB: %s\n", g, hipGetErrorString(e)); (void)hipGetLastError(); continue; }
    CK(hipFree(p));
    printf("(a) %3d GiB contiguous: alloc %.3f s, free %.3f s\n", g, t1 - t, now() - t1);
  }
  // (b) NE extents of EG GiB held together; class of every cell end against a reference read stream
  std::vector<char *> ext;
  for (int i = 0; i < NE; ++i) {
    char *p = nullptr;
    const double t = now();
    if (hipExtMallocWithFlags((void **)&p, (size_t)EG * G, hipDeviceMallocContiguous) != hipSuccess) { (void)hipGetLastError(); printf("(b) extent %d: no memory\n", i); break; }
    printf("(b) extent %2d at %p: alloc %.3f s\n", i, (void *)p, now() - t);
    ext.push_back(p);
  }
  const int cells = (int)((size_t)EG * G / CELL);
  auto classify = [&](const char *rd, const char *label) {
    // times of every cell end of every extent against rd; '#' = slow (same class as rd), '.' = fast
    std::vector<float> t;
    for (char *b : ext) for (int c = 0; c < cells; ++c) t.push_back(pair_ms(rd, b + (size_t)(c + 1) * CELL - WR));
    float mn = 1e30f, mx = 0;
    for (float v : t) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    printf("(b) read stream %s: %.4f .. %.4f ms\n", label, mn, mx);
    const float thr = 0.5f * (mn + (mx < 1.3f * mn ? mx : 1.3f * mn));
    for (size_t i = 0; i < ext.size(); ++i) {
      printf("    extent %2zu: ", i);
      for (int c = 0; c < cells; ++c) putchar(mx > 1.06f * mn ? (t[i * cells + c] > thr ? '#' : '.') : '?');
      putchar('\n');
    }
    return t;
  };
  if (!ext.empty()) {
    auto t0 = classify(ext[0], "extent 0 cell 0");
    // a second reference: the first cell (from the back) that is fast against extent 0
    float mn = 1e30f, mx = 0;
    for (float v : t0) { mn = v < mn ? v : mn; mx = v > mx ? v : mx; }
    const float thr = 0.5f * (mn + mx);
    int ref = -1;
    for (size_t k = 1; k < t0.size(); ++k) if (t0[k] < thr && t0[k - 1] < thr && (k % cells) != 0) { ref = (int)k; break; }
    if (ref >= 0) {
      char lab[64];
      snprintf(lab, sizeof lab, "extent %d cell %d", ref / cells, ref % cells);
      classify(ext[ref / cells] + (size_t)(ref % cells) * CELL, lab);
    }
    // (c) free the middle extents and ask again: which class does the next request land in?
    if (ext.size() >= 4) {
      for (size_t i = 1; i + 1 < ext.size(); ++i) CK(hipFree(ext[i]));
      char *last = ext.back(), *first = ext[0];
      ext.assign({first, last});
      char *p = nullptr;
      if (hipExtMallocWithFlags((void **)&p, (size_t)EG * G, hipDeviceMallocContiguous) == hipSuccess) {
        printf("(c) after freeing the middle extents, a new extent sits at %p\n", (void *)p);
        ext.push_back(p);
        classify(first, "extent 0 cell 0 (extents: first, last, new)");
      }
    }
    // (d) plain hipMalloc'ed buffers of vector size: class of start / middle / end against extent 0
    for (size_t mb : {128, 1024, 4096}) {
      char *p = nullptr;
      CK(hipMalloc((void **)&p, mb << 20));
      printf("(d) hipMalloc %4zu MiB at %p vs extent 0:", mb, (void *)p);
      for (int k = 0; k < 5; ++k) printf(" %.4f", pair_ms(ext[0], p + ((mb << 20) - WR) / 4 * k / 4096 * 4096));
      printf(" ms (slow ~ same class)\n");
      CK(hipFree(p));
    }
  }
  return 0;
}
