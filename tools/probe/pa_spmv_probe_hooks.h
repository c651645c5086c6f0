// pa_spmv_probe_hooks.h -- the what-if experiments of rounds 1-2 on the product kernel, OUTSIDE the product.
//
// csrc/pa_spmv_kernel.h has a handful of hook points (PA_HOOK_*) that are empty in libpa_hip.so.  This header defines them
// and then includes the product kernel, so that tools/probe/*.hip and the variant libraries of tools/probe/Makefile get the
// same kernel with ONE thing changed.  Every experiment gives WRONG results on purpose and answers one question about
// where the time goes; the answers are in docs/LAB_NOTEBOOK.md and profiles/r02_*whatif*.log.
//
//   -DPA_PROBE_IDENTITY_CHUNK_MAP  workgroup b runs chunk b (no XCD-aware dealing)          placement_probe.hip
//   -DPA_PROBE_NO_GATHER           lane-contiguous x reads in place of the decoded columns
//   -DPA_PROBE_ONE_PLANE           every gather folded into the row's own grid plane (27-point 256^3)
//   -DPA_PROBE_TILE_X              the x footprint of a 4 x 14 tile instead of 57 consecutive nodes
//   -DPA_PROBE_LDS_X               three coalesced x loads per lane into LDS, gathers from there
//   -DPA_PROBE_PAIR_GATHER         (round 5) the second entry of every pair reads x[c0 + 1]: three 16-byte gathers per lane instead of six
//   -DPA_PROBE_STAMPS              (round 5; right results) wall-clock stamps of a workgroup's phases, tools/probe/k1_stamps.py
//   -DPA_PROBE_TILE_LDS            round 3, VERDICT r02 #5 "tile + LDS together": the x footprint of a 4 x 14 tile (3 planes x 6
//                                  lines x 16 nodes = 288 doubles, 2.3 KB instead of the 4.2 KB of 57 consecutive nodes) staged
//                                  ONCE per chunk with coalesced 128-byte runs, every gather a ds_read_b64
//   EPI 7..13 (template argument)  variants of the y store: none / plain / 2 MiB window / nt window / 16-byte pairs / sc1 / sc0 sc1
#ifndef PA_SPMV_PROBE_HOOKS_H
#define PA_SPMV_PROBE_HOOKS_H
#include <hip/hip_runtime.h>

#ifdef PA_PROBE_STAMPS
// round 5: when does a workgroup of the product kernel reach which phase?  Lane 0 writes the 100 MHz wall clock at five points of
// its workgroup's life (0 entry, 1 chunk head known, 2 wave 0's products formed = its loads are back, 3 behind the barrier, 4 rows
// summed and stored) plus where it ran (HW_ID, XCC_ID) into a device array that pa_probe_read_stamps copies out
// (tools/probe/k1_stamps.py).  Only pa_csr.hip is built with this (tools/probe/build_k1_variants.sh).
#define PA_PROBE_STAMP_BLOCKS (1 << 19)
__device__ unsigned long long pa_probe_stamps[(size_t)PA_PROBE_STAMP_BLOCKS * 8];
template <typename... T>
__device__ __forceinline__ void pa_probe_stamp(int k, T... dep) {
  if (threadIdx.x != 0 || blockIdx.x >= PA_PROBE_STAMP_BLOCKS) return;
  ((void)(dep), ...);
  unsigned long long *rec = pa_probe_stamps + (size_t)blockIdx.x * 8;
  rec[k] = wall_clock64();
  if (k == 0) rec[7] = ((unsigned long long)__builtin_amdgcn_s_getreg(63508) << 32) | (unsigned)__builtin_amdgcn_s_getreg(63492);
}
__device__ __forceinline__ void pa_probe_pin(double a) { asm volatile("" ::"v"(a)); }
__device__ __forceinline__ void pa_probe_pin(int a) { asm volatile("" ::"s"(a)); }
#define PA_HOOK_STAMP(k, ...) { pa_probe_pin_all(__VA_ARGS__); pa_probe_stamp(k); }
template <typename... T>
__device__ __forceinline__ void pa_probe_pin_all(T... a) { (pa_probe_pin(a), ...); }
extern "C" int pa_probe_read_stamps(unsigned long long *out, size_t n_blocks) {
  return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(pa_probe_stamps), n_blocks * 8 * sizeof(unsigned long long), 0, hipMemcpyDeviceToHost);
}
#endif

#ifdef PA_PROBE_IDENTITY_CHUNK_MAP
#define PA_HOOK_CHUNK_MAP 1
#endif

#if defined(PA_PROBE_TILE_X)
// 27-point 256^3: the x footprint a chunk WOULD have if its rows were a tile of 4 grid lines x 14 nodes
#define PA_HOOK_PATTERN_COLS(c0, c1, r0, r1, tid)                                                       \
  {                                                                                                     \
    auto tile = [&](int c) {                                                                            \
      const int off = c - r0;                                                                           \
      const int dz = (off + 32768) >> 16, rem = off - (dz << 16);                                       \
      const int dy = (rem + 64) >> 8, qdx = rem - (dy << 8) + 14;                                       \
      const int ly = (qdx * 4682) >> 16;                                                                \
      return max(r0 + (qdx - ly * 14) + ((ly - 1 + dy) << 8) + (dz << 16), 0);                          \
    };                                                                                                  \
    c0 = tile(c0);                                                                                      \
    c1 = tile(c1);                                                                                      \
  }
#elif defined(PA_PROBE_ONE_PLANE)
// the dz = -1 / +1 entries read where the dz = 0 entries do: same instruction count, a third of the pages per gather
#define PA_HOOK_PATTERN_COLS(c0, c1, r0, r1, tid)                                                       \
  {                                                                                                     \
    auto flat = [&](int c) { const int dz = ((c - r0) + 32768) >> 16; return max(c - (dz << 16), 0); }; \
    c0 = flat(c0);                                                                                      \
    c1 = flat(c1);                                                                                      \
  }
#elif defined(PA_PROBE_PAIR_GATHER)
// round 5: what would it buy if a lane's two entries were always neighbours in x (one 16-byte gather instead of two 8-byte ones)?
#define PA_HOOK_PATTERN_COLS(c0, c1, r0, r1, tid) { c1 = c0 + 1; }
#elif defined(PA_PROBE_NO_GATHER)
#define PA_HOOK_PATTERN_COLS(c0, c1, r0, r1, tid)                                                       \
  {                                                                                                     \
    c0 = min(max(c0, 0) & 1, 1) + min(r0 + (tid & 63), r1 - 1);                                         \
    c1 = min(max(c1, 0) & 1, 1) + min(r0 + (tid & 63), r1 - 1);                                         \
  }
#define PA_HOOK_C16_COLS(c0, c1, lo, hi, r0, r1, tid)                                                   \
  {                                                                                                     \
    c0 = min(r0 + (tid & 63) + (int)(lo & 1), r1 - 1);                                                  \
    c1 = min(r0 + (tid & 63) + (int)(hi & 1), r1 - 1);                                                  \
  }
#endif

#ifdef PA_PROBE_TILE_LDS
// lanes 0..287 fetch run = lane / 16 (plane = run / 6, line = run % 6), node = lane % 16 of the tile's footprint: 18 runs of
// 128 bytes; a gather's true offset (dz, dy, q + dx) is folded into the tile (line = q / 14 + dy, node = q % 14 + dx).
#define PA_HOOK_X_STAGE(x, r0, r1, tid)                                                                 \
  __shared__ double xtile[3 * 6 * 16];                                                                  \
  if (tid < 288) {                                                                                      \
    const int run_ = tid >> 4, k_ = tid & 15, pl_ = run_ / 6, ln_ = run_ - pl_ * 6;                     \
    xtile[tid] = x[min(max(r0 + k_ - 1 + (ln_ - 1) * 256 + (pl_ - 1) * 65536, 0), max_col)];            \
  }                                                                                                     \
  __syncthreads();
__device__ __forceinline__ int pa_probe_tile_slot(int off) {
  const int dz = (off + 32768) >> 16, rem = off - (dz << 16);
  const int dy = (rem + 64) >> 8, qdx = rem - (dy << 8) + 14;                 // q + dx + 14 in [13, 71]
  const int ly = (qdx * 4682) >> 16;                                          // qdx / 14
  const int line = min(max(ly - 1 + dy + 1, 0), 5), node = qdx - ly * 14 + 1; // node in [1, 14]
  return (dz + 1) * 96 + line * 16 + node;
}
#define PA_HOOK_X_AT(x, c, r0) xtile[min(max(pa_probe_tile_slot((c) - (r0)), 0), 287)]
#endif

#ifdef PA_PROBE_LDS_X
#define PA_HOOK_X_STAGE(x, r0, r1, tid)                                                                 \
  __shared__ double xprobe[3 * BLK];                                                                    \
  _Pragma("unroll") for (int k_ = 0; k_ < 3; ++k_) xprobe[tid + k_ * BLK] = x[min(max(r0 - BLK + tid + k_ * BLK, 0), r1 - 1)]; \
  __syncthreads();
#define PA_HOOK_X_AT(x, c, r0) xprobe[((c) - (r0)) & 511]
#endif

// EPI 11: a lane owns the two rows of a 16-byte slot of y and stores them with one dwordx4
#define PA_HOOK_ALT_REDUCE()                                                                            \
  if (EPI == 11) {                                                                                      \
    for (int rb = (r0 & ~1) + 2 * tid; rb < r1; rb += 2 * BLK) {                                        \
      double acc2[2] = {0.0, 0.0};                                                                      \
      _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                                   \
        const int r = rb + h;                                                                           \
        if (r < r0 || r >= r1) continue;                                                                \
        const int a = crp[r] - base, e = crp[r + 1] - base;                                             \
        double acc = 0.0;                                                                               \
        for (int p = a; p < e; ++p) acc = acc + prod[PA_PSLOT(p)];                                      \
        acc2[h] = acc;                                                                                  \
      }                                                                                                 \
      if (rb >= r0 && rb + 1 < r1) {                                                                    \
        d2 o; o.x = acc2[0]; o.y = acc2[1];                                                             \
        __builtin_nontemporal_store(o, reinterpret_cast<d2 *>(&y[rb]));                                 \
      } else if (rb >= r0) __builtin_nontemporal_store(acc2[0], &y[rb]);                                \
      else __builtin_nontemporal_store(acc2[1], &y[rb + 1]);                                            \
    }                                                                                                   \
    return;                                                                                             \
  }

#define PA_HOOK_STORE_Y(EPI, y, row, acc)                                                                                     \
  if (EPI == 7) { if (acc == 123.456) (y)[row] = acc; }                     /* the kernel without its y store */              \
  else if (EPI == 8) (y)[row] = acc;                                        /* plain (cached) store */                        \
  else if (EPI == 9) (y)[(row) & 0x3ffff] = acc;                            /* plain store into a 2 MiB window */             \
  else if (EPI == 10) __builtin_nontemporal_store(acc, &(y)[(row) & 0x3ffff]);                                               \
  else if (EPI == 12) asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(&(y)[row]), "v"(acc) : "memory");            \
  else if (EPI == 13) asm volatile("global_store_dwordx2 %0, %1, off sc0 sc1" ::"v"(&(y)[row]), "v"(acc) : "memory");        \
  else __builtin_nontemporal_store(acc, &(y)[row])

#include "pa_spmv_kernel.h"
#endif
