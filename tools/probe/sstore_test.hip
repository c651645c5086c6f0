// sstore_test.hip -- does gfx950 execute scalar stores (s_store_dwordx2 + s_dcache_wb)?  Prints OK / FAIL.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(double *y, int n) {
  const int tid = threadIdx.x;
  const double acc = 1000.0 * blockIdx.x + tid + 0.5;
  const unsigned lo = (unsigned)__double_as_longlong(acc), hi = (unsigned)(__double_as_longlong(acc) >> 32);
  double *yb = y + (size_t)blockIdx.x * 64;
  const int nn = __builtin_amdgcn_readfirstlane(n);
  for (int i = 0; i < nn; ++i) {
    const unsigned long long v = ((unsigned long long)__builtin_amdgcn_readlane(hi, i) << 32) | __builtin_amdgcn_readlane(lo, i);
    double *p = yb + i;
    asm volatile("s_store_dwordx2 %0, %1, 0x0" ::"s"(v), "s"(p) : "memory");
    if ((i & 7) == 7) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::: "memory");
}
int main() {
  double *d, h[64 * 4];
  if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 2;
  hipMemset(d, 0, sizeof(h));
  hipLaunchKernelGGL(k, dim3(4), dim3(64), 0, 0, d, 64);
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { printf("FAIL: %s\n", hipGetErrorString(e)); return 1; }
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int b = 0; b < 4; ++b) for (int i = 0; i < 64; ++i) bad += h[b * 64 + i] != 1000.0 * b + i + 0.5;
  printf(bad ? "FAIL: %d wrong values\n" : "OK scalar stores work%.0d\n", bad);
  return bad != 0;
}
