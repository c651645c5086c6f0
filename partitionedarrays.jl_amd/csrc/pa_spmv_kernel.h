// pa_spmv_kernel.h -- the row-split CSR SpMV kernel (K1/K2) and its host-side row split.
// Shared by pa_csr.hip (the product), pa_mg.hip (smoother / restriction epilogues), pa_plan.hip (product + dot), pa_fused.hip and tools/probe/spmv_probe.hip.
//
// Reference loops: spmv_csr! src/sparse_utils.jl:649-669; muladd! src/p_sparse_matrix.jl:2088.
// Must be compiled with -ffp-contract=off (one rounding per multiply and per add).
//
// This header holds the PRODUCT kernel only.  The what-if experiments of rounds 1-2 (kernels that give wrong results on
// purpose to answer one question about where the time goes: no gather, one plane, tile footprint, x through LDS, variants
// of the y store) live in tools/probe/pa_spmv_probe_hooks.h, which includes this file after defining the PA_HOOK_* macros
// below; here every hook is empty, so nothing of the lab can reach libpa_hip.so.
#ifndef PA_SPMV_KERNEL_H
#define PA_SPMV_KERNEL_H

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstring>
#include <thread>
#include <type_traits>
#include <unordered_map>
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
//   32-bit : col[p]                                            (fallback and long rows; see pa_encode_columns for which
//            chunks keep a column stream at all)
//   "c16"  : col[p] = win[chunk][c16[p] >> 12] + (c16[p] & 4095)  -- up to PA_C16_WINDOWS 4096-aligned column
//            windows per chunk (stencil / FEM / banded rows touch a handful of narrow column clusters), 2 bytes per
//            stored entry instead of 4.  Pure index compression: values stay fp64, pa_csr_update_values is unaffected,
//            and a chunk whose columns need more windows keeps the 32-bit path (win[chunk][0] < 0).
#define PA_C16_WINDOWS 16
//   "row patterns": no per-entry column at all.  A row's pattern is its list of (col - row) deltas; stencil / FEM
//            matrices on structured grids have a handful of distinct patterns (27-pt: 27).  A chunk made of at most
//            PA_PAT_SEGMENTS runs of consecutive rows with one pattern each is described by 16 ints (pdesc) and its
//            columns are recomputed: col = first_row(seg) + (t / L) * stride(seg) + delta[pat(seg)][t % L], t = entry
//            index inside the segment (stride: row-id step of a run, 1 unless the block is row-compacted, e.g. 2 along
//            a grid line for the rows of one Gauss-Seidel colour).  Chunks that do not fit (more runs, rows longer than
//            PA_PAT_MAXLEN, rows whose pattern is rare) use c16 / 32-bit columns.
#define PA_PAT_SEGMENTS 4
#define PA_PAT_MAXLEN 32
#define PA_PDESC_INTS 20   // {nseg, q1..q3, first row x4, (L | stride<<8) x4, pattern x4, magic x4 (2^32/L rounded up)}

// column of entry q (relative to the chunk's first entry) from the chunk's pattern descriptor
template <bool STRIDED>
__device__ __forceinline__ int pa_pattern_col(int q, int nq, int q1, int q2, int q3, int L0, int L1, int L2, int L3,
                                              int s0r, int s1r, int s2r, int s3r, unsigned M0, unsigned M1, unsigned M2,
                                              unsigned M3, int dA, int dB) {
  q = max(0, min(q, nq));
  const bool g1 = q >= q1, g2 = q >= q2, g3 = q >= q3;  // chained selects (a runtime-indexed array would go to scratch)
  int qs = g1 ? q1 : 0, Ls = g1 ? L1 : L0, rs = g1 ? s1r : s0r;
  unsigned M = g1 ? M1 : M0;                  // 2^32 / L rounded up, from the descriptor (a division per entry otherwise)
  qs = g2 ? q2 : qs; Ls = g2 ? L2 : Ls; rs = g2 ? s2r : rs; M = g2 ? M2 : M;
  qs = g3 ? q3 : qs; Ls = g3 ? L3 : Ls; rs = g3 ? s3r : rs; M = g3 ? M3 : M;
  const int s = (int)g1 + (int)g2 + (int)g3;
  const int t = q - qs;
  // descriptor word: row length | row-id stride << 8 (the stride bits are only set for row-compacted blocks)
  const int L = STRIDED ? (Ls & 255) : Ls, stride = STRIDED ? (Ls >> 8) : 1;
  const int rr = L == 1 ? t : (int)__umulhi((unsigned)t, M);  // t / L, exact for t < 2^32 / L
  const int kk = t - rr * L;
  // lanes 0-31 / 32-63 of dA hold the deltas of segment 0 / 1, of dB those of segment 2 / 3.  ds_bpermute reads the
  // SOURCE lane's register, so both are fetched and the requesting lane selects.
  const int sel = (((s & 1) << 5) + kk) << 2;
  const int delA = __builtin_amdgcn_ds_bpermute(sel, dA);
  const int delB = __builtin_amdgcn_ds_bpermute(sel, dB);
  return rs + rr * stride + ((s & 2) ? delB : delA);
}

// The same when every entry a wavefront decodes in one step lies in ONE run of the chunk (the common case: a chunk has
// at most 4 runs and a step covers 128 consecutive entries): the run's parameters are wave-uniform (scalar registers),
// so the per-entry compare/select chains disappear -- 9 vector instructions and one ds_bpermute per entry instead of ~35
// and two.  dsrc = the register (dA or dB) that holds the run's deltas, half = which 32-lane half of it.
template <bool STRIDED>
__device__ __forceinline__ int pa_pattern_col_uniform(int q, int nq, int qs, int Lw, int rs, unsigned M, int dsrc,
                                                      int half) {
  q = max(0, min(q, nq));
  const int t = q - qs;
  const int L = STRIDED ? (Lw & 255) : Lw, stride = STRIDED ? (Lw >> 8) : 1;
  const int rr = L == 1 ? t : (int)__umulhi((unsigned)t, M);
  const int kk = t - (int)__umul24((unsigned)rr, (unsigned)L);
  const int del = __builtin_amdgcn_ds_bpermute(((half << 5) + kk) << 2, dsrc);
  return rs + (STRIDED ? (int)__umul24((unsigned)rr, (unsigned)stride) : rr) + del;
}

// The same for a chunk that is ONE run from its first entry on, rows of L >= 2 entries (round 5): no run to look up, no L == 1 case.
template <bool STRIDED>
__device__ __forceinline__ int pa_pattern_col_run(int q, int nq, int Lw, int rs, unsigned M, int dsrc) {
  const int t = max(0, min(q, nq));
  const int L = STRIDED ? (Lw & 255) : Lw, stride = STRIDED ? (Lw >> 8) : 1;
  const int rr = (int)__umulhi((unsigned)t, M);
  const int kk = t - (int)__umul24((unsigned)rr, (unsigned)L);
  const int del = __builtin_amdgcn_ds_bpermute(kk << 2, dsrc);
  return rs + (STRIDED ? (int)__umul24((unsigned)rr, (unsigned)stride) : rr) + del;
}

// Sum of a double over the 64 lanes of a wavefront, the same value returned in every lane, in a fixed order: four DPP
// row_shr steps inside each row of 16 lanes (a few cycles each; __shfl_down goes through the LDS crossbar, ~100 cycles a
// step -- on the one wavefront whose exit frees a workgroup's LDS that is 8 % of the product kernel), then the four
// row totals read with v_readlane and added left to right.
template <int CTRL>
__device__ __forceinline__ double pa_dpp_move(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xF, 0xF, true);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double pa_wave_sum(double v) {
  v = v + pa_dpp_move<0x111>(v);   // row_shr:1 (lanes without a source get 0)
  v = v + pa_dpp_move<0x112>(v);   // row_shr:2
  v = v + pa_dpp_move<0x114>(v);   // row_shr:4
  v = v + pa_dpp_move<0x118>(v);   // row_shr:8 -> lane 15 of every row holds the row's sum
  const int lo = __double2loint(v), hi = __double2hiint(v);
  const double a = __hiloint2double(__builtin_amdgcn_readlane(hi, 15), __builtin_amdgcn_readlane(lo, 15));
  const double b = __hiloint2double(__builtin_amdgcn_readlane(hi, 31), __builtin_amdgcn_readlane(lo, 31));
  const double c = __hiloint2double(__builtin_amdgcn_readlane(hi, 47), __builtin_amdgcn_readlane(lo, 47));
  const double d = __hiloint2double(__builtin_amdgcn_readlane(hi, 63), __builtin_amdgcn_readlane(lo, 63));
  return ((a + b) + c) + d;
}

// y[row] = beta*y[row] + sum_p (val[p]*x[col[p]])*alpha, products summed in ascending p.
//   BLK  threads per workgroup, NPT stored entries per lane (chunk capacity CAP = BLK*NPT products in LDS),
//   NT   non-temporal matrix loads, C16 use the 16-bit column stream where the chunk has one,
//   PAT  0: no row patterns (pdesc/pdelta may be NULL); 1: use row-pattern descriptors where the chunk has one, rows
//        of a run are consecutive; 2: the same for row-compacted blocks, a run has a row-id stride.
//   EPI  0: y[row] = beta*y[row] + sum (x, y distinct).
//        1: Gauss-Seidel colour update in place, gs_x[row] += (gs_b[row] - sum) / gs_diag[row] with the products
//           gathered from gs_x itself (x, y unused).  Race-free when the block's rows form one colour of a proper
//           colouring: no row of the launch reads another row of the launch, only itself.
//        2: fused residual + restriction, gs_x[r] = gs_b[row] - sum for the r-th stored (compacted) row: the coarse
//           residual r_c = (r_f - A x_f) at the fine rows a coarse grid keeps (gs_x = r_c, gs_b = r_f, x_in = x_f).
//        3: the product plus one term of a dot product: y as for EPI 0 (alpha = 1), and gs_x[chunk] = sum over the chunk's
//           rows of gs_b[row] * (sum of the row's products) -- the per-chunk partial sums of u'(A x), u = gs_b, summed in a
//           fixed order (row order per lane, shuffle tree, wave sums): deterministic.  The CG loop's u'c = dot(u, A u)
//           (HPCG/src/ref_cg.jl:60) costs no pass over u and c this way.
//   chunk_list (or NULL): the launch covers these chunks only -- what k_spmv_xwin (pa_spmv_xwin.h) leaves over.
//   max_col: n_cols - 1 of the block (bounds the columns lanes outside the chunk decode on the 16-bit path).
//   VD   value dictionary (optional, lossless): a block with at most PA_VDICT_MAX distinct stored values (bit patterns)
//        keeps one byte per entry (`code`) and the values in `dict`; lane l holds dict[l] and an entry's value is fetched
//        from the lane that holds it with ds_bpermute -- 1 byte of matrix stream per entry instead of 8.  The 27-point
//        HPCG operator has 2 distinct values, a Q1 stiffness matrix on a uniform grid about a dozen.
//        VD = 2: a dictionary of at most two values, decoded by a select (no LDS traffic).
#define PA_VDICT_MAX 64

// ---- hook points (empty in the product; see the comment at the top) ------------------------------------------------------
#ifndef PA_HOOK_CHUNK_MAP            /* blockIdx -> chunk: the product's XCD-aware map */
#define PA_HOOK_CHUNK_MAP 0
#endif
#ifndef PA_HOOK_STAMP                /* phase k of a workgroup's life reached (the lab records the time) */
#define PA_HOOK_STAMP(k, ...)
#endif
#ifndef PA_HOOK_PATTERN_COLS         /* after a pair of pattern columns has been decoded */
#define PA_HOOK_PATTERN_COLS(c0, c1, r0, r1, tid)
#endif
#ifndef PA_HOOK_C16_COLS             /* after a pair of 16-bit columns has been decoded */
#define PA_HOOK_C16_COLS(c0, c1, lo, hi, r0, r1, tid)
#endif
#ifndef PA_HOOK_X_STAGE              /* before the products: a chance to stage x somewhere */
#define PA_HOOK_X_STAGE(x, r0, r1, tid)
#define PA_HOOK_X_AT(x, c, r0) (x)[c]
#endif
#ifndef PA_HOOK_ALT_REDUCE           /* a whole other reduce phase for some EPI values (returns from the kernel) */
#define PA_HOOK_ALT_REDUCE()
#endif
#ifndef PA_HOOK_STORE_Y              /* the y store of EPI values the product does not know */
#define PA_HOOK_STORE_Y(EPI, y, row, acc) __builtin_nontemporal_store(acc, &(y)[row])
#endif
// What the fused product mul!(c,a,b) (k_mul_fused, pa_mul_fused.h) adds to a chunk's work:
//   FX 1 (own x own inside the fused launch): rows whose bit is set in `rowmask` are the part's BOUNDARY rows -- rows with stored
//        entries in own_ghost; they are summed and stored by the launch's tail (FX 2) once b's ghost values are there, so this
//        role leaves them alone: not stored, and with beta != 0 not read either;
//   FX 2 (the tail: the boundary rows, own_own entries then own_ghost entries of each): columns >= n_split are positions of
//        consistent!'s receive buffer x2.
struct pa_fx {
  const unsigned *rowmask = nullptr;
  const double *x2 = nullptr;
  int n_split = 0x7fffffff;
};

// one chunk of the row split (everything k_spmv_rowsplit does once it knows its chunk); prod / wsum: the workgroup's LDS
template <int BLK, int NPT, bool NT, bool C16, int PAT, int EPI, int VD, int UNR, bool PADP, int FX>
__device__ __forceinline__ void pa_rowsplit_chunk(
    double *prod, double *wsum, const int *__restrict__ crp, const int *__restrict__ col, const unsigned short *__restrict__ col16,
    const int *__restrict__ win, const int *__restrict__ pdesc, const int *__restrict__ pdelta,
    const double *__restrict__ val, const double *__restrict__ x_in,
    double *__restrict__ y, const int *__restrict__ chunk_rp, const int *__restrict__ row_ids,
    double alpha, double beta, double *gs_x, const double *__restrict__ gs_b,
    const double *__restrict__ gs_diag, const unsigned char *__restrict__ code,
    const double *__restrict__ dict, int max_col, int chunk, const pa_fx fx) {
  constexpr int CAP = BLK * NPT;
  const double *x = EPI == 1 ? gs_x : x_in;   // EPI 1 reads and writes the same vector: no restrict promise on it
  static_assert(NPT % 2 == 0, "pairs");
  static_assert(FX == 0 || EPI == 0, "the fused roles are roles of the plain product");
  constexpr bool PAD = PADP;
#define PA_PSLOT(p) (PAD ? (p) + 2 * ((p) >> 5) : (p))
  const int tid = threadIdx.x;
  // chunk_rp[2c .. 2c+3] = {first row, its row pointer} of chunk c and of chunk c+1: the chunk's rows AND entries from one 16-byte
  // scalar read (round 5; chunk_row -> crp[r0], crp[r1] was a second, dependent round trip before the value stream could be
  // requested).  Eight chunks share a 64-byte line, so seven reads in eight hit in the scalar cache / L2.  (A 96-byte record with
  // the descriptor inside, read by one vector load per wavefront, measured 2-4 % SLOWER: it misses to HBM every time.)
  const int r0 = chunk_rp[2 * chunk], p0 = chunk_rp[2 * chunk + 1];
  const int r1 = chunk_rp[2 * chunk + 2], p1 = chunk_rp[2 * chunk + 3];
  const int base = p0 & ~1;  // 16-byte aligned value pairs, 4-byte aligned c16 pairs
  PA_HOOK_STAMP(1, p1);

  // The body of a chunk, once per column encoding (MODE 2: row patterns, 1: 16-bit windowed stream, 0: 32-bit columns): each copy
  // runs from its first load to the end of the kernel and the copies never meet again.  (Round 5.  As one body with the encodings
  // as branches inside it, the compiler had to assume at every join that the loads of EITHER side were pending and protect their
  // registers: the pattern path waited for the row-extent load -- a whole memory round trip -- before it even requested the
  // values, and a load hoisted or sunk across a join drained the other side's prefetch.  No joins, no such waits.)
  auto body = [&](auto mode_tag, int nseg, int mywin) __attribute__((always_inline)) {
    constexpr int MODE = decltype(mode_tag)::value;
    // Unconditional loads: lanes past the chunk's end re-read its last pair (same address => no extra
    // traffic) and their products are never summed.  A guarded load would make the compiler wait for
    // each load before issuing the next (one HBM round trip per k instead of one per chunk).
    // Every load instruction is contiguous across the 64 lanes (16 B, 8 B or 4 B per lane).
    const int last = max((p1 - 1) & ~1, 0);
    d2 v[NPT / 2];
    unsigned cc[NPT / 2];
    int c0[NPT / 2], c1[NPT / 2];
    int dict_lo = 0, dict_hi = 0;
    double d0 = 0.0, d1 = 0.0;
    if (VD == 1) {
      const double dv = dict[tid & 63];
      dict_lo = __double2loint(dv);
      dict_hi = __double2hiint(dv);
    }
    if (VD == 2) { d0 = dict[0]; d1 = dict[1]; }        // (scalar loads: the same two values for every lane)
    // the order of the requests is the order their answers are waited for: what the decode needs (pattern deltas: an L2 hit),
    // the value stream (HBM; the wait for the deltas leaves it in flight), my row's extent (needed by the row sums only)
    int dA = 0, dB = 0;
#define PA_D(k) pdesc[chunk * PA_PDESC_INTS + (k)]
    if (MODE == 2) {
      const int pt0 = PA_D(12), pt1 = PA_D(13), pt2 = PA_D(14), pt3 = PA_D(15);
      const int lane = tid & 63;
      dA = pdelta[((lane >> 5) ? pt1 : pt0) * PA_PAT_MAXLEN + (lane & 31)];
      if (nseg > 2) dB = pdelta[((lane >> 5) ? pt3 : pt2) * PA_PAT_MAXLEN + (lane & 31)];
    }
#pragma unroll
    for (int k = 0; k < NPT / 2; ++k) {
      const int idx = min(base + (k * BLK + tid) * 2, last);
      if (VD) cc[k] = pa_stream_load<NT>(reinterpret_cast<const unsigned short *>(code + idx));
      else v[k] = pa_stream_load<NT>(reinterpret_cast<const d2 *>(val + idx));
    }
    if (MODE == 2) {
      const int q1 = PA_D(1), q2 = PA_D(2), q3 = PA_D(3);
      const int s0r = PA_D(4), s1r = PA_D(5), s2r = PA_D(6), s3r = PA_D(7);
      const int L0 = PA_D(8), L1 = PA_D(9), L2 = PA_D(10), L3 = PA_D(11);
      const unsigned M0 = PA_D(16), M1 = PA_D(17), M2 = PA_D(18), M3 = PA_D(19);
      const int nq = p1 - p0 - 1;
      const int wave0 = __builtin_amdgcn_readfirstlane(tid & ~63);   // first thread of this wavefront (uniform)
      if (VD && nseg == 1 && (L0 & 255) != 1) {
        // ONE run of one pattern (four chunks in five of a stencil block): nothing to look up per step, where the general path
        // below finds the run of every 128-entry step with ~45 scalar instructions (238 scalar against 188 vector instructions
        // per wavefront, profiles/r05_k1_sq.json).  Worth 2.5 % on the one-byte value stream and COSTS 2 % on the fp64 stream
        // (interleaved A/B, 256^3: 0.540 / 0.553 ms and 0.683 / 0.668 ms with / without), hence only with the dictionary.
#pragma unroll
        for (int k = 0; k < NPT / 2; ++k) {
          const int idx = min(base + (k * BLK + tid) * 2, last);
          c0[k] = pa_pattern_col_run<PAT == 2>(idx - p0, nq, L0, s0r, M0, dA);
          c1[k] = pa_pattern_col_run<PAT == 2>(idx + 1 - p0, nq, L0, s0r, M0, dA);
          PA_HOOK_PATTERN_COLS(c0[k], c1[k], r0, r1, tid);
        }
      } else
#pragma unroll
      for (int k = 0; k < NPT / 2; ++k) {
        const int idx = min(base + (k * BLK + tid) * 2, last);
        // the entries this wavefront decodes in this step: [ulo, uhi] relative to the chunk's first entry
        const int ulo = max(0, min(min(base + (k * BLK + wave0) * 2, last) - p0, nq));
        const int uhi = max(0, min(min(base + (k * BLK + wave0 + 63) * 2, last) + 1 - p0, nq));
        const int sa = (int)(ulo >= q1) + (int)(ulo >= q2) + (int)(ulo >= q3);
        const int sb = (int)(uhi >= q1) + (int)(uhi >= q2) + (int)(uhi >= q3);
        if (sa == sb) {                                               // wave-uniform branch
          const int qs = sa == 0 ? 0 : sa == 1 ? q1 : sa == 2 ? q2 : q3;
          const int Lw = sa == 0 ? L0 : sa == 1 ? L1 : sa == 2 ? L2 : L3;
          const int rs = sa == 0 ? s0r : sa == 1 ? s1r : sa == 2 ? s2r : s3r;
          const unsigned Mw = sa == 0 ? M0 : sa == 1 ? M1 : sa == 2 ? M2 : M3;
          const int dsrc = (sa & 2) ? dB : dA;
          c0[k] = pa_pattern_col_uniform<PAT == 2>(idx - p0, nq, qs, Lw, rs, Mw, dsrc, sa & 1);
          c1[k] = pa_pattern_col_uniform<PAT == 2>(idx + 1 - p0, nq, qs, Lw, rs, Mw, dsrc, sa & 1);
        } else {
          c0[k] = pa_pattern_col<PAT == 2>(idx - p0, nq, q1, q2, q3, L0, L1, L2, L3, s0r, s1r, s2r, s3r, M0, M1, M2, M3, dA, dB);
          c1[k] = pa_pattern_col<PAT == 2>(idx + 1 - p0, nq, q1, q2, q3, L0, L1, L2, L3, s0r, s1r, s2r, s3r, M0, M1, M2, M3, dA, dB);
        }
        PA_HOOK_PATTERN_COLS(c0[k], c1[k], r0, r1, tid);
      }
    } else {
      // A block with row patterns keeps columns only for its chunks WITHOUT a descriptor (compacted streams): such a
      // chunk's descriptor slot holds where its columns sit relative to its entry offsets (0: full-length streams).
      const int sh16 = PAT ? PA_D(1) : 0, sh32 = PAT ? PA_D(2) : 0;
      if (MODE == 1) {
        unsigned q[NPT / 2];
#pragma unroll
        for (int k = 0; k < NPT / 2; ++k) {
          const int idx = min(base + (k * BLK + tid) * 2, last);
          q[k] = pa_stream_load<NT>(reinterpret_cast<const unsigned *>(col16 + (idx + sh16)));
        }
#pragma unroll
        for (int k = 0; k < NPT / 2; ++k) {
          // window base of slot s lives in lane s of `mywin` (any 16-lane group): fetch it with ds_bpermute
          const unsigned lo = q[k] & 0xffffu, hi = q[k] >> 16;
          // max_col = n_cols - 1: the entry before an odd first entry and the one after an odd last entry belong to the
          // neighbouring chunks; decoded with THIS chunk's windows their codes can point up to 4095 columns past the end of x
          // (their products are never summed, but the load must stay inside the vector)
          c0[k] = min(__builtin_amdgcn_ds_bpermute((lo >> 12) << 2, mywin) + (int)(lo & 4095), max_col);
          c1[k] = min(__builtin_amdgcn_ds_bpermute((hi >> 12) << 2, mywin) + (int)(hi & 4095), max_col);
          PA_HOOK_C16_COLS(c0[k], c1[k], lo, hi, r0, r1, tid);
        }
      } else {
#pragma unroll
        for (int k = 0; k < NPT / 2; ++k) {
          const int idx = min(base + (k * BLK + tid) * 2, last);
          const i2 c = pa_stream_load<NT>(reinterpret_cast<const i2 *>(col + (idx + sh32)));
          c0[k] = c.x; c1[k] = c.y;
        }
      }
    }
#undef PA_D
    // my first row's extent (and, EPI 3, u[my row]): requested now so that the row sums do not wait for it
    const int rmine = min(r0 + tid, r1 - 1);     // (lanes past the chunk's rows: its last row, never summed)
    int ra = crp[rmine], re = crp[rmine + 1];
    unsigned mword = 0;
    if (FX == 1) mword = fx.rowmask[rmine >> 5];  // (non-compact block: row = stored row)
    double urow = 0.0;
    if (EPI == 3) urow = gs_b[row_ids ? row_ids[rmine] : rmine];
    if (VD == 1) {
#pragma unroll
      for (int k = 0; k < NPT / 2; ++k) {
        const int s0 = (cc[k] & 0xffu) << 2, s1 = (cc[k] >> 8) << 2;
        v[k].x = __hiloint2double(__builtin_amdgcn_ds_bpermute(s0, dict_hi), __builtin_amdgcn_ds_bpermute(s0, dict_lo));
        v[k].y = __hiloint2double(__builtin_amdgcn_ds_bpermute(s1, dict_hi), __builtin_amdgcn_ds_bpermute(s1, dict_lo));
      }
    }
    if (VD == 2) {
      // Two stored values (HPCG's operator: 26 and -1): a select per entry.  The lane dictionary's four ds_bpermute per pair were
      // half of this kernel's LDS instructions, and on the one-byte stream it is the LDS pipe that is busy (SQ_ACTIVE_INST_LDS x
      // resident waves = 0.9 of the launch, profiles/r05_k1_sq.json), not the vector ALU.
#pragma unroll
      for (int k = 0; k < NPT / 2; ++k) {
        v[k].x = (cc[k] & 0xffu) ? d1 : d0;
        v[k].y = (cc[k] >> 8) ? d1 : d0;
      }
    }
    PA_HOOK_X_STAGE(x, r0, r1, tid);
#pragma unroll
    for (int k = 0; k < NPT / 2; ++k) {
      d2 pr;
      if (FX == 2) {
        const double *x2m = fx.x2 - fx.n_split;
        pr.x = v[k].x * (c0[k] < fx.n_split ? x : x2m)[c0[k]];
        pr.y = v[k].y * (c1[k] < fx.n_split ? x : x2m)[c1[k]];
      } else {
        pr.x = v[k].x * PA_HOOK_X_AT(x, c0[k], r0);
        pr.y = v[k].y * PA_HOOK_X_AT(x, c1[k], r0);
      }
      v[k] = pr;
    }
    // (alpha: multiply-and-select per product, not a branch around the scaling -- the branch form waits for ALL gathers before the
    // first LDS write and measured 3 % slower, 0.707 against 0.684 ms)
#pragma unroll
    for (int k = 0; k < NPT / 2; ++k) if (alpha != 1.0) { v[k].x = v[k].x * alpha; v[k].y = v[k].y * alpha; }
    PA_HOOK_STAMP(2, v[0].x, v[NPT / 2 - 1].y);
#pragma unroll
    for (int k = 0; k < NPT / 2; ++k) *reinterpret_cast<d2 *>(&prod[PA_PSLOT((k * BLK + tid) * 2)]) = v[k];
    __syncthreads();
    PA_HOOK_STAMP(3, 0);
    PA_HOOK_ALT_REDUCE();
    double dacc = 0.0;                       // EPI 3: this lane's share of the dot product
    for (int r = r0 + tid; r < r1; r += BLK) {
      if (r != r0 + tid) {
        ra = crp[r];
        re = crp[r + 1];
      }
      const int row = row_ids ? row_ids[r] : r;
      if (FX == 1) {
        if (r != r0 + tid) mword = fx.rowmask[row >> 5];
        if ((mword >> (row & 31)) & 1u) continue;          // a boundary row: the launch's tail sums and stores it
      }
      double acc = ((EPI != 0 && EPI != 3) || beta == 0.0) ? 0.0 : beta * y[row];
      const int a = ra - base, e = re - base;
#pragma unroll UNR
      for (int p = a; p < e; ++p) acc = acc + prod[PA_PSLOT(p)];
      if (EPI == 3) {
        double pr = acc;                     // the row's products alone (beta = 0: that is acc itself)
        if (beta != 0.0) {
          pr = 0.0;
          for (int p = a; p < e; ++p) pr = pr + prod[PA_PSLOT(p)];
        }
        dacc = dacc + (r == r0 + tid ? urow : gs_b[row]) * pr;
        __builtin_nontemporal_store(acc, &y[row]);
      } else
      if (EPI == 1) gs_x[row] = gs_x[row] + (gs_b[row] - acc) / gs_diag[row];
      else if (EPI == 2) gs_x[r] = gs_b[row] - acc;
      // non-temporal: y is written once and not read again by this kernel; measured 3.6 % faster than the plain
      // store (0.791 vs 0.820 ms; sc1 0.797, sc0 sc1 0.807, sc0 sc1 nt 0.842, no store at all 0.681)
      else if (EPI == 0) __builtin_nontemporal_store(acc, &y[row]);
      else { PA_HOOK_STORE_Y(EPI, y, row, acc); }
    }
    PA_HOOK_STAMP(4, 0);
    if (EPI == 3) {
      dacc = pa_wave_sum(dacc);
      if (r1 - r0 <= 64) {                   // (block-uniform) every row sat in wavefront 0: the other three just leave
        if (tid == 0) gs_x[chunk] = dacc;
      } else {
        if ((tid & 63) == 0) wsum[tid >> 6] = dacc;
        __syncthreads();
        if (tid == 0) {
          double t = 0.0;
          for (int w = 0; w < BLK / 64; ++w) t = t + wsum[w];
          gs_x[chunk] = t;
        }
      }
    }
  };
  if (p1 - base <= CAP) {
    const int nseg = PAT ? pdesc[chunk * PA_PDESC_INTS] : 0;
    if (PAT && nseg > 0) {
      body(std::integral_constant<int, 2>{}, nseg, 0);
    } else {
      int mywin = -1;
      if (C16) mywin = win[chunk * PA_C16_WINDOWS + (tid & (PA_C16_WINDOWS - 1))];
      if (C16 && __builtin_amdgcn_readfirstlane(mywin) >= 0) body(std::integral_constant<int, 1>{}, nseg, mywin);   // lane 0 holds window 0
      else body(std::integral_constant<int, 0>{}, nseg, mywin);
    }
  } else {
    // one long row (more stored entries than a chunk holds): windows of CAP products, summed by
    // lane 0 in ascending p so that even this path keeps the reference's order.
    const int row = row_ids ? row_ids[r0] : r0;
    if (FX == 1 && ((fx.rowmask[row >> 5] >> (row & 31)) & 1u)) return;      // (a long boundary row: the tail's, whole)
    const int *lcol = col + (PAT ? pdesc[chunk * PA_PDESC_INTS + 2] : 0);   // compacted 32-bit stream, see above
    double acc = 0.0, accp = 0.0;
    if (tid == 0) acc = ((EPI != 0 && EPI != 3) || beta == 0.0) ? 0.0 : beta * y[row];
    for (int w = p0; w < p1; w += CAP) {
      const int wend = min(w + CAP, p1);
      for (int idx = w + tid; idx < wend; idx += BLK) {
        const int lc = lcol[idx];
        double pr = val[idx] * (FX == 2 && lc >= fx.n_split ? fx.x2[lc - fx.n_split] : x[lc]);
        if (alpha != 1.0) pr = pr * alpha;
        prod[idx - w] = pr;
      }
      __syncthreads();
      if (tid == 0)
        for (int p = 0; p < wend - w; ++p) {
          acc = acc + prod[p];
          if (EPI == 3) accp = accp + prod[p];
        }
      __syncthreads();
    }
    if (tid == 0) {
      if (EPI == 1) gs_x[row] = gs_x[row] + (gs_b[row] - acc) / gs_diag[row];
      else if (EPI == 2) gs_x[r0] = gs_b[row] - acc;
      else {
        y[row] = acc;
        if (EPI == 3) gs_x[chunk] = gs_b[row] * accp;
      }
    }
  }
}

#undef PA_PSLOT

template <int BLK, int NPT, bool NT, bool C16, int PAT, int EPI = 0, int VD = 0, int UNR = 4, bool PADP = false>
__global__ __launch_bounds__(BLK) void k_spmv_rowsplit(
    const int *__restrict__ crp, const int *__restrict__ col, const unsigned short *__restrict__ col16,
    const int *__restrict__ win, const int *__restrict__ pdesc, const int *__restrict__ pdelta,
    const double *__restrict__ val, const double *__restrict__ x_in,
    double *__restrict__ y, const int *__restrict__ chunk_rp, const int *__restrict__ row_ids, int n_chunks,
    int chunks_per_xcd, double alpha, double beta, double *gs_x, const double *__restrict__ gs_b,
    const double *__restrict__ gs_diag, const unsigned char *__restrict__ code = nullptr,
    const double *__restrict__ dict = nullptr, const int *__restrict__ chunk_list = nullptr, int max_col = 0x7fffffff) {
  constexpr int CAP = BLK * NPT;
  // PADP: two pad slots per 32 products, for blocks whose rows mostly hold a multiple of 8 stored entries: the lanes of the
  // reduce phase otherwise sit on the same banks (rows of 16: two banks, 16-way; with the pad 2-way; 4 M x 16 within +-500:
  // 0.145 -> 0.136 ms).  Two slots, not one, so that a lane's pair of products stays 16-byte aligned and goes out as one
  // ds_write_b128.  Other row lengths (18, 27, 81, ragged) lose 2-3 % to the slot arithmetic: the library picks the variant
  // per block (pa_csr::pad_products).
  __shared__ __attribute__((aligned(16))) double prod[PADP ? CAP + CAP / 16 + 2 : CAP];
  __shared__ double wsum[EPI == 3 ? BLK / 64 : 1];
  PA_HOOK_STAMP(0, 0);
  const int b = blockIdx.x;
  // (chunks_per_xcd < 0: the same map walked BACKWARDS -- every other product of a block streams its arrays from the end, so that what
  // the last product left in the translation caches and in the Infinity Cache is what this one reads first)
  const bool backwards = chunks_per_xcd < 0;
  if (backwards) chunks_per_xcd = -chunks_per_xcd;
  int chunk = PA_HOOK_CHUNK_MAP ? b : (b & 7) * chunks_per_xcd + (b >> 3);  // XCD-aware: block b sits on XCD b%8
  if (chunk >= n_chunks || (!PA_HOOK_CHUNK_MAP && (b >> 3) >= chunks_per_xcd)) return;
  if (backwards) chunk = n_chunks - 1 - chunk;
  if (chunk_list) chunk = chunk_list[chunk];       // a launch over some of the block's chunks (n_chunks = length of the list)
  pa_rowsplit_chunk<BLK, NPT, NT, C16, PAT, EPI, VD, UNR, PADP, 0>(prod, wsum, crp, col, col16, win, pdesc, pdelta, val, x_in, y, chunk_rp,
                                                                    row_ids, alpha, beta, gs_x, gs_b, gs_diag, code, dict, max_col, chunk,
                                                                    pa_fx());
}

// Host-side row split: greedy chunks of consecutive (compacted) rows whose stored entries, counted from
// the 2-aligned start, fit `cap` products; a row longer than that is a chunk on its own.
inline void pa_build_chunks(const int32_t *crp, int64_t nc, int cap, int max_rows, std::vector<int32_t> &chunk_row,
                            int64_t *n_long, int align_rows = 8) {
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
      // end the chunk on a multiple of `align_rows` rows (64 bytes of y) when that costs less than a quarter of it:
      // every chunk then starts its y store on a full 64-byte line (measured 2 % on the 27-point operator, where
      // the chunks next to the grid's faces hold more, shorter rows than the 56 of an interior chunk)
      if (align_rows > 1 && e < nc) {
        const int64_t ea = e - (e % align_rows);
        if (ea > r && (ea - r) * 4 >= (e - r) * 3) e = ea;
      }
    }
    chunk_row.push_back((int32_t)e);
    r = e;
  }
}

// Host-side c16 encoding of every chunk (multi-threaded over chunks). win has n_chunks*PA_C16_WINDOWS entries;
// win[c*16] = -1 marks a chunk that keeps 32-bit columns (too many windows, or a long row).
// pos (or NULL): where chunk c's entries go in c16, pos[c] = slot of the entry at its 2-aligned start, or -1 = this chunk
// gets no 16-bit columns (not counted as a fallback).  NULL: entry p goes to c16[p].
inline int64_t pa_encode_col16(const int32_t *crp, const int32_t *col, const std::vector<int32_t> &chunk_row, int cap,
                               uint16_t *c16, int32_t *win, int n_threads, const int64_t *pos = nullptr) {
  const int64_t n_chunks = (int64_t)chunk_row.size() - 1;
  std::vector<int64_t> fallback(n_threads > 0 ? n_threads : 1, 0);
  auto work = [&](int t, int T) {
    for (int64_t c = n_chunks * t / T; c < n_chunks * (t + 1) / T; ++c) {
      int32_t *w = win + c * PA_C16_WINDOWS;
      for (int s = 0; s < PA_C16_WINDOWS; ++s) w[s] = 0;
      const int64_t p0 = crp[chunk_row[c]], p1 = crp[chunk_row[c + 1]];
      if (pos && pos[c] < 0) { w[0] = -1; continue; }
      const int64_t shift = pos ? pos[c] - (p0 & ~(int64_t)1) : 0;
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
        c16[p + shift] = (uint16_t)((s << 12) | (col[p] & 4095));
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

// Host-side row-pattern analysis.  pdesc: n_chunks*PA_PDESC_INTS ints ({nseg | 0, q1..q3, r0..r3, (L | stride<<8)0..3,
// pat0..3, magic0..3});
// pdelta: PA_PAT_MAXLEN ints per pattern.  row_ids (or NULL) maps a stored (compacted) row to its row id; deltas are
// col - row id.  Only patterns shared by at least `min_rows` rows enter the table (an unstructured row is its own
// pattern and keeps explicit columns).  Returns the number of chunks that got a
// descriptor.
inline int64_t pa_encode_patterns(const int32_t *crp, const int32_t *col, const int32_t *row_ids, int64_t n_rows,
                                  const std::vector<int32_t> &chunk_row, int cap, std::vector<int32_t> &pdesc,
                                  std::vector<int32_t> &pdelta, int n_threads, int max_patterns = 4096,
                                  int min_rows = 2) {
  const int64_t n_chunks = (int64_t)chunk_row.size() - 1;
  pdesc.assign((size_t)n_chunks * PA_PDESC_INTS, 0);
  pdelta.clear();
  if (n_threads < 1) n_threads = 1;
  auto rid = [&](int64_t r) -> int32_t { return row_ids ? row_ids[r] : (int32_t)r; };
  // phase A (parallel): a 64-bit hash of every row's delta list
  std::vector<uint64_t> h(n_rows);
  auto hash_rows = [&](int t) {
    for (int64_t r = n_rows * t / n_threads; r < n_rows * (t + 1) / n_threads; ++r) {
      uint64_t x = 1469598103934665603ull ^ (uint64_t)(crp[r + 1] - crp[r]);
      const int32_t id = rid(r);
      for (int64_t p = crp[r]; p < crp[r + 1]; ++p) { x ^= (uint64_t)(uint32_t)(col[p] - id); x *= 1099511628211ull; x ^= x >> 29; }
      h[r] = x;
    }
  };
  {
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(hash_rows, t);
    hash_rows(0);
    for (auto &x : th) x.join();
  }
  // (a block of unstructured rows has as many delta lists as rows: a sample of 65536 evenly spaced rows tells -- more than
  // half of them distinct means no descriptor would cover half of the chunks, and the sequential phase below, a hash map
  // over every row, is skipped: 0.6 s of a 64 M-entry block's set-up)
  if (n_rows >= (1 << 18)) {
    const int64_t ns = 1 << 16, step = n_rows / ns;
    std::vector<uint64_t> sample(ns);
    for (int64_t k = 0; k < ns; ++k) sample[k] = h[k * step];
    std::sort(sample.begin(), sample.end());
    const int64_t distinct = std::unique(sample.begin(), sample.end()) - sample.begin();
    if (distinct * 2 > ns) { pdelta.assign(PA_PAT_MAXLEN, 0); return 0; }
  }
  // phase B (sequential): how many rows share each hash, then pattern ids for the frequent ones, verified against the
  // stored deltas
  std::unordered_map<uint64_t, int32_t> freq;
  freq.reserve(1 << 16);
  for (int64_t r = 0; r < n_rows; ++r) {
    const int len = crp[r + 1] - crp[r];
    if (len < 1 || len > PA_PAT_MAXLEN) continue;
    int32_t &f = freq[h[r]];
    if (f < min_rows) ++f;
    if ((int64_t)freq.size() > (int64_t)1 << 22 && n_rows > ((int64_t)1 << 23)) break;   // unstructured: stop counting
  }
  std::vector<int32_t> rowpat(n_rows);
  std::vector<int32_t> plen;
  std::unordered_map<uint64_t, std::vector<int32_t>> ids;
  for (int64_t r = 0; r < n_rows; ++r) {
    const int len = crp[r + 1] - crp[r];
    rowpat[r] = -1;
    if (len < 1 || len > PA_PAT_MAXLEN) continue;
    auto fi = freq.find(h[r]);
    if (fi == freq.end() || fi->second < min_rows) continue;
    auto &cand = ids[h[r]];
    const int32_t id_r = rid(r);
    int32_t id = -1;
    for (int32_t c : cand) {
      if (plen[c] != len) continue;
      bool same = true;
      for (int k = 0; k < len && same; ++k) same = pdelta[(size_t)c * PA_PAT_MAXLEN + k] == col[crp[r] + k] - id_r;
      if (same) { id = c; break; }
    }
    if (id < 0) {
      if ((int)plen.size() >= max_patterns) continue;   // table full: this row keeps explicit columns
      id = (int32_t)plen.size();
      plen.push_back(len);
      pdelta.resize((size_t)(id + 1) * PA_PAT_MAXLEN, 0);
      for (int k = 0; k < len; ++k) pdelta[(size_t)id * PA_PAT_MAXLEN + k] = col[crp[r] + k] - id_r;
      cand.push_back(id);
    }
    rowpat[r] = id;
  }
  if (pdelta.empty()) { pdelta.assign(PA_PAT_MAXLEN, 0); return 0; }
  // phase C (parallel): runs of equal pattern and constant row-id stride inside every chunk
  std::vector<int64_t> good(n_threads, 0);
  auto segs = [&](int t) {
    for (int64_t c = n_chunks * t / n_threads; c < n_chunks * (t + 1) / n_threads; ++c) {
      int32_t *d = &pdesc[(size_t)c * PA_PDESC_INTS];
      const int64_t r0 = chunk_row[c], r1 = chunk_row[c + 1];
      const int64_t p0 = crp[r0], p1 = crp[r1];
      bool ok = (p1 - (p0 & ~1)) <= cap && p1 > p0;
      int ns = 0;
      int64_t run_rows = 0;
      int32_t stride = 1;
      for (int64_t r = r0; ok && r < r1; ++r) {
        if (rowpat[r] < 0) { ok = false; break; }
        bool extend = ns > 0 && rowpat[r] == d[12 + ns - 1];
        if (extend) {
          const int32_t step = rid(r) - rid(r - 1);
          if (run_rows == 1) {
            if (step < 1 || step >= (1 << 20)) extend = false;
            else stride = step;
          } else if (step != stride) {
            extend = false;
          }
        }
        if (extend) {
          ++run_rows;
          if (row_ids) d[8 + ns - 1] = (d[8 + ns - 1] & 255) | (stride << 8);
          continue;
        }
        if (ns == PA_PAT_SEGMENTS) { ok = false; break; }
        if (ns) d[ns] = (int32_t)(crp[r] - p0);
        d[4 + ns] = rid(r);
        d[8 + ns] = (crp[r + 1] - crp[r]) | (row_ids ? 1 << 8 : 0);
        d[12 + ns] = rowpat[r];
        ++ns;
        run_rows = 1;
        stride = 1;
      }
      if (!ok) { for (int k = 0; k < PA_PDESC_INTS; ++k) d[k] = 0; continue; }
      for (int s = ns; s < PA_PAT_SEGMENTS; ++s) { if (s) d[s] = 1 << 30; d[4 + s] = 0; d[8 + s] = 1 | (row_ids ? 1 << 8 : 0); d[12 + s] = 0; }
      for (int s = 0; s < PA_PAT_SEGMENTS; ++s) {
        const uint32_t L = (uint32_t)(d[8 + s] & 255);
        d[16 + s] = (int32_t)(L > 1 ? 0xFFFFFFFFu / L + 1u : 0u);
      }
      d[0] = ns;
      ++good[t];
    }
  };
  {
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(segs, t);
    segs(0);
    for (auto &x : th) x.join();
  }
  int64_t ng = 0;
  for (auto v : good) ng += v;
  return ng;
}

// Column streams of one block exactly as the kernel reads them.
//   block without row patterns: `c16` (when wanted) is full length, entry p at c16[p], and the 32-bit columns are the
//     caller's own array (`full`);
//   block with row patterns (descriptors on at least half of its chunks): columns are kept ONLY for the chunks without
//     a descriptor -- `c16` for those that encode in 16-bit windows, `c32` for the rest (too many windows, long rows) --
//     and such a chunk's descriptor slot holds {0, shift16, shift32}: its entry p sits at c16[p + shift16] or
//     c32[p + shift32].  A stencil operator then stores 8 bytes per entry (the value) instead of 14.
struct pa_col_streams {
  bool use_pattern = false, use_c16 = false, full = true;
  std::vector<int32_t> pdesc, pdelta, win, c32;
  std::vector<uint16_t> c16;
  int64_t n_pattern = 0, n_c16 = 0, n_c32 = 0;   // chunks by column encoding
};

inline void pa_encode_columns(const int32_t *crp, const int32_t *col, const int32_t *row_ids, int64_t n_rows,
                              const std::vector<int32_t> &chunk_row, int cap, bool want_pattern, bool want_c16,
                              int n_threads, pa_col_streams &S, bool compact_streams = true) {
  const int64_t n_chunks = (int64_t)chunk_row.size() - 1;
  const int64_t nnz = n_rows > 0 ? crp[n_rows] : 0;
  const size_t pad = 8;
  S = pa_col_streams();
  if (nnz == 0 || n_chunks == 0) { S.n_c32 = n_chunks; return; }
  if (want_pattern) {
    S.n_pattern = pa_encode_patterns(crp, col, row_ids, n_rows, chunk_row, cap, S.pdesc, S.pdelta, n_threads);
    S.use_pattern = S.n_pattern > 0 && S.n_pattern * 2 >= n_chunks;   // worth it only when it covers most of the block
    if (!S.use_pattern) { S.n_pattern = 0; S.pdesc.clear(); S.pdelta.clear(); }
  }
  // A handful of chunks outside the patterns (the 8 corner chunks of a 298 197-chunk stencil block) do not justify the
  // 16-bit path: the kernel variant that carries it runs the PATTERN chunks 1.2 % slower (0.686 vs 0.678 ms at 27-point
  // 256^3, interleaved A/B), so such a block keeps 32-bit columns for them.
  if (S.use_pattern && want_c16 && (n_chunks - S.n_pattern) * 64 < n_chunks) want_c16 = false;
  S.use_c16 = want_c16;
  if (want_c16) S.win.assign((size_t)n_chunks * PA_C16_WINDOWS, 0);
  if (!S.use_pattern || !compact_streams) {
    // full-length streams (with descriptors only for measurements: their column slots stay 0 = no shift)
    if (want_c16) {
      S.c16.assign(nnz + pad, 0);
      pa_encode_col16(crp, col, chunk_row, cap, S.c16.data(), S.win.data(), n_threads);
    }
    for (int64_t c = 0; c < n_chunks; ++c) {
      if (S.use_pattern && S.pdesc[(size_t)c * PA_PDESC_INTS] > 0) continue;
      if (want_c16 && S.win[(size_t)c * PA_C16_WINDOWS] >= 0) ++S.n_c16; else ++S.n_c32;
    }
    return;
  }
  S.full = false;
  auto span = [&](int64_t c) {   // slots of chunk c counted from its 2-aligned start, plus the pair the clamped loads may touch
    const int64_t p0 = crp[chunk_row[c]], p1 = crp[chunk_row[c + 1]];
    return ((p1 - (p0 & ~(int64_t)1) + 1) & ~(int64_t)1) + 2;
  };
  auto start = [&](int64_t c) { return (int64_t)crp[chunk_row[c]] & ~(int64_t)1; };
  if (want_c16) {
    std::vector<int64_t> pos(n_chunks, -1);
    int64_t n16 = 0;
    for (int64_t c = 0; c < n_chunks; ++c) {
      if (S.pdesc[(size_t)c * PA_PDESC_INTS] > 0) continue;
      if (crp[chunk_row[c + 1]] - start(c) > cap) continue;           // a long row keeps 32-bit columns
      pos[c] = n16;
      n16 += span(c);
    }
    S.c16.assign(n16 + pad, 0);
    pa_encode_col16(crp, col, chunk_row, cap, S.c16.data(), S.win.data(), n_threads, pos.data());
    for (int64_t c = 0; c < n_chunks; ++c)
      if (pos[c] >= 0 && S.win[(size_t)c * PA_C16_WINDOWS] >= 0) {
        S.pdesc[(size_t)c * PA_PDESC_INTS + 1] = (int32_t)(pos[c] - start(c));
        ++S.n_c16;
      }
  }
  int64_t n32 = 0;
  std::vector<int64_t> pos32(n_chunks, -1);
  for (int64_t c = 0; c < n_chunks; ++c) {
    if (S.pdesc[(size_t)c * PA_PDESC_INTS] > 0) continue;
    if (want_c16 && S.win[(size_t)c * PA_C16_WINDOWS] >= 0) continue;
    pos32[c] = n32;
    n32 += span(c);
    ++S.n_c32;
  }
  S.c32.assign(n32 + pad, 0);
  for (int64_t c = 0; c < n_chunks; ++c) {
    if (pos32[c] < 0) continue;
    const int64_t b = start(c), p1 = crp[chunk_row[c + 1]];
    std::memcpy(&S.c32[pos32[c]], col + b, sizeof(int32_t) * (size_t)(p1 - b));
    S.pdesc[(size_t)c * PA_PDESC_INTS + 2] = (int32_t)(pos32[c] - b);
  }
}

#endif
