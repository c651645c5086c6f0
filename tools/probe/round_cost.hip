// What does one ROUND cost?  (a) a dependent launch of a small kernel that writes a little (the set-up's rounds: pa_rowsel.hip), (b) one
// iteration of a persistent kernel with a device-wide barrier (agent-scope release / acquire around an atomic counter).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/round_cost.hip -o /tmp/round_cost && /tmp/round_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ void k_small(int *a, int n, int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) a[(i * 97 + k) % n] += 1;
}

__global__ void k_persist(int *a, int n, int rounds, unsigned *bar, int work_blocks) {
  const unsigned G = gridDim.x;
  for (int k = 0; k < rounds; ++k) {
    if ((int)blockIdx.x < work_blocks) {
      const int i = blockIdx.x * blockDim.x + threadIdx.x;
      if (i < n) a[(i * 97 + k) % n] += 1;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      atomicAdd(bar, 1u);
      const unsigned want = (unsigned)(k + 1) * G;
      const long long t0 = (long long)wall_clock64();
      while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
        __builtin_amdgcn_s_sleep(2);
        if ((long long)wall_clock64() - t0 > 200000000ll) { bar[8] = 1; return; }       // 2 s: give up (never spin for ever on a shared box)
      }
      __threadfence();
    }
    __syncthreads();
  }
}

int main() {
  const int n = 1 << 20;
  int *a; unsigned *bar;
  CK(hipMalloc(&a, sizeof(int) * n)); CK(hipMemset(a, 0, sizeof(int) * n));
  CK(hipMalloc(&bar, 64)); CK(hipMemset(bar, 0, 64)); CK(hipDeviceSynchronize());
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int blocks : {64, 256, 1024}) {
    for (int w = 0; w < 200; ++w) hipLaunchKernelGGL(k_small, dim3(blocks), dim3(256), 0, s, a, n, w);
    CK(hipStreamSynchronize(s));
    auto t0 = std::chrono::steady_clock::now();
    const int R = 2000;
    for (int k = 0; k < R; ++k) hipLaunchKernelGGL(k_small, dim3(blocks), dim3(256), 0, s, a, n, k);
    CK(hipStreamSynchronize(s));
    const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / R;
    printf("dependent launches of %4d blocks: %.2f us per round\n", blocks, us); fflush(stdout);
  }
  for (int G : {256, 512, 1024}) {
    for (int wb : {64, G}) {
      CK(hipMemsetAsync(bar, 0, 64, s));
      hipLaunchKernelGGL(k_persist, dim3(G), dim3(256), 0, s, a, n, 10, bar, wb);
      CK(hipStreamSynchronize(s));
      CK(hipMemsetAsync(bar, 0, 64, s));
      CK(hipStreamSynchronize(s));
      auto t0 = std::chrono::steady_clock::now();
      const int R = 2000;
      hipLaunchKernelGGL(k_persist, dim3(G), dim3(256), 0, s, a, n, R, bar, wb);
      CK(hipStreamSynchronize(s));
      const double us = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 1e6 / R;
      unsigned gave_up = 0; CK(hipMemcpy(&gave_up, bar + 8, 4, hipMemcpyDeviceToHost));
      printf("persistent, %4d workgroups (%4d working): %.2f us per round%s\n", G, wb, us, gave_up ? "  (GAVE UP)" : ""); fflush(stdout);
    }
  }
  return 0;
}
