// pa_pell.h -- "pattern-ELL": the byte-bound SpMV kernel for blocks whose rows follow a handful of column patterns (round 6).
//
// Reference loop: spmv_csr! src/sparse_utils.jl:649-669 (y[row] = sum over the row's stored entries, ascending p, one rounding per
// multiply and per add); mul!(y,A,x,alpha,beta) of SparseMatricesCSR as called at src/p_sparse_matrix.jl:2088.
//
// Why (VERDICT r05 "What's weak" 3, profiles/r05_k1_sq.json): on pattern blocks the row-split kernel k_spmv_rowsplit no longer reads a
// column stream, and with the value dictionary not even fp64 values -- yet it cannot go below ~0.45 ms at 256^3: products staged in
// LDS, a ds_bpermute per decoded delta, a barrier, and one wave in four adding 27 dependent LDS-fed products per row.  Its floor is
// the LDS pipe and a chain of dependent round trips, not bytes.
//
// Here ONE LANE owns ONE ROW and a wavefront owns a SLAB of 64 consecutive stored rows:
//   * a slab's column pattern is the UNION of its rows' (column - row id) deltas, ascending: D[0..W), W <= 32, wave-uniform -- the
//     deltas sit in SCALAR registers (s_load from a small pattern table), lane l gathers x[rid(l) + D[k]]: for consecutive rows
//     that is one contiguous 512-byte load per k -- no ds_bpermute, no decode, no LDS, no barrier;
//   * row l takes part in delta k when bit k of its 32-bit mask is set (rows at the ends of a grid line lack some neighbours): the k-th
//     set bit is the row's k-th stored entry, so walking k upwards adds the row's products in stored order -- the additions of
//     spmv_csr!, same bits (file compiled -ffp-contract=off); absent entries are neither loaded (the gather index falls back to 0) nor
//     added (an added +0.0 could turn a -0.0 sum into +0.0);
//   * values are re-laid per slab, delta-major: val[(off + k) * 64 + lane] -- every value load of a wavefront is one contiguous
//     512 bytes (SELL-64 without its column array, csrc/pa_sell.hip); slots of absent entries hold 0.0 and are read, never used;
//     a slab's width is padded to a multiple of the unroll U (the block's commonest width divides by it: no padding there);
//   * VM 1: a block with at most TWO distinct stored values (HPCG: 26 and -1 on every level, in every colour) keeps ONE BIT per
//     entry -- a second 32-bit word per row -- and no value stream at all: 8 bytes of matrix per ROW.
// Moved bytes per row at 27 entries: fp64 stream 216 + 4 (mask) (+ 4 row id when the block is row-compacted) against the row-split
// kernel's 216 + 20; one-bit stream 8 against 27 + 20.
#ifndef PA_PELL_H
#define PA_PELL_H

#include <hip/hip_runtime.h>

#include <cstdint>
#include <type_traits>

#include "pa_internal.h"
#include "pa_spmv_kernel.h"

#define PA_PELL_MAXW 32                  /* deltas of a slab's union (bits of a row mask) */
#define PA_PELL_TW 44                    /* ints per pattern in the table: the deltas, padded with 0 up to the padded width (<= 36), then: */
#define PA_PELL_T_STRIDE 40              /*   slab CLASSES only (see below): the rows of the slab are row0 + stride * lane (1 or 2; 0: not so) */
#define PA_PELL_T_FLAGS 41               /*   bit 0: every lane has every delta of the padded width (no lane masks needed) */
#define PA_PELL_T_MIN 42                 /*   lowest and highest delta of the union: a slab whose row0 + min >= 0 and */
#define PA_PELL_T_MAX 43                 /*   row0 + 63 * stride + max < n_cols gathers without clamping */

struct pa_pell_dev {
  const int2 *desc = nullptr;            // per slab {pattern | padded width << 20, first value slot / 64}
  const int *pdelta = nullptr;           // n_patterns x PA_PELL_TW
  const unsigned *mask = nullptr;        // per stored row: which deltas of its slab's pattern it has
  const unsigned *bits = nullptr;        // VM 1: per stored row, bit k = dictionary code of the entry at delta k
  const double *val = nullptr;           // VM 0: slab-major, delta-major, lane-minor
  const double *dict = nullptr;          // VM 1: the two values; VM 2: the dictionary (PA_VDICT_MAX values)
  const unsigned *codes = nullptr;       // VM 2: one BYTE per entry (its dictionary code): per group of U deltas and lane ceil(U / 4) dwords,
                                         //       group-major, lane-minor -- a wavefront's codes of a group are contiguous (768 bytes at U = 9)
  const int *row_ids = nullptr;          // row-compacted block: stored row -> row
  // slab classes (round 6, second step): table rows are then not just the union of deltas but (union, which LANES have each delta,
  // stride of the row ids), so that what was a 32-bit mask per row becomes a 64-bit lane ballot per delta and CLASS, read as scalars
  const unsigned long long *plane = nullptr;   // n_classes x PA_PELL_TW: bit l of [k] = lane l of such a slab has delta k; NULL: no classes
  const unsigned *prel = nullptr;        // classes: (delta - lowest delta of the class) * 8, n_classes x PA_PELL_TW (the lean form's byte offsets)
  const uint2 *sbits = nullptr;          // VM 1, per slab: {the rows' bits OR-ed, 1 when every row's bits are that word under its mask}
  int n_slabs = 0, n_crows = 0, n_cols = 0;
};

typedef double pa_d2u __attribute__((ext_vector_type(2), aligned(8)));      // two consecutive doubles at any 8-byte boundary

// value slot of delta k of a slab whose first slot is `first` (in 64-entry units), lane `lane`: inside every group of U deltas the
// lanes hold PAIRS of consecutive deltas next to each other (one 16-byte load per lane and pair, 1 KiB per wavefront) and, U odd,
// the group's last delta on its own (512 bytes per wavefront) -- a group is U * 512 contiguous bytes either way
template <int U>
__host__ __device__ __forceinline__ size_t pa_pell_slot(unsigned first, int k, int lane) {
  const int g = k / U, j = k - g * U;
  const size_t B = ((size_t)first + (size_t)g * U) * 64;
  constexpr int UE = U & ~1;
  return j < UE ? B + (size_t)(j >> 1) * 128 + (size_t)lane * 2 + (j & 1) : B + (size_t)UE * 64 + lane;
}

// lane l <- v of lane l + 1; lane 63 <- e (DPP wave_shl:1, the whole wavefront as one row of 64)
__device__ __forceinline__ double pa_wave_shl1(double v, double e) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(e), __double2loint(v), 0x130, 0xF, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(e), __double2hiint(v), 0x130, 0xF, 0xF, false);
  return __hiloint2double(hi, lo);
}

// VM 2 (round 6, third step): a block whose value dictionary has 3 .. 64 values keeps ONE BYTE per entry in pattern-ELL order; the
// dictionary sits in 512 bytes of LDS per workgroup (staged once per launch), a value is a byte extract + one ds_read_b64.
template <int U>
struct pa_pell_cw { unsigned w[(U + 3) / 4]; } __attribute__((aligned(4)));
template <int U>
__device__ __forceinline__ pa_pell_cw<U> pa_pell_codes(const unsigned *codes, unsigned first, int k0, int lane) {
  const size_t g = (size_t)(first + (unsigned)k0) / U;           // (first and k0 are multiples of U)
  return *reinterpret_cast<const pa_pell_cw<U> *>(codes + (g * 64 + (size_t)lane) * ((U + 3) / 4));
}
template <int U>
__device__ __forceinline__ double pa_pell_value(const pa_pell_cw<U> &c, int j, const double *sd) {      // (j: a constant after unrolling)
  const unsigned w = c.w[j >> 2];
  const unsigned off = (j & 3) == 0 ? (w << 3) & 0x7f8u : (w >> ((j & 3) * 8 - 3)) & 0x7f8u;             // code * 8
  return *reinterpret_cast<const double *>(reinterpret_cast<const char *>(sd) + off);
}

// bit `bit` of the (scalar) word sb set ? d1 : d0 -- two scalar instructions (the compiler makes three: a 64-bit select as two halves)
__device__ __forceinline__ double pa_uniform(double v) {        // (a wave-uniform value the compiler may hold in vector registers -> scalar)
  return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)), __builtin_amdgcn_readfirstlane(__double2loint(v)));
}
__device__ __forceinline__ double pa_sel_bit_at(unsigned sb, int bit, double d0, double d1) {      // (bit: a constant after unrolling)
  double r;
  asm("s_bitcmp1_b32 %3, %4\n\ts_cselect_b64 %0, %1, %2" : "=s"(r) : "s"(d1), "s"(d0), "s"(sb), "n"(bit) : "scc");
  return r;
}

// The slab of a CLASS with runs of three, alpha = 1, nothing to add to (beta = 0 or an epilogue form), gathers in range: the
// instruction-lean form of the slab below (notebook R6.5; the one-bit stream was bound by the vector ALU at ~17 instructions per entry).
//   * rows row0 + S * lane: the gather of a run's first delta is ONE scalar base + lane offset (no per-lane address arithmetic, no clamps);
//     S = 1: 8-byte gather + two wave shifts (as R3 below); S = 2 (the colour and restriction blocks of the multigrid: every other row of a
//     grid line): ONE 16-byte gather gives columns d and d + 1, column d + 2 is the next lane's first -- one wave shift;
//   * which lanes have delta k is a 64-bit ballot of the class, a scalar pair: x of an absent entry is replaced by 0.0 with two
//     v_cndmask on that pair and the product added unconditionally.  Same bits as adding only the present products: the sum starts at
//     +0.0 and can never become -0.0 (x + (-x) and (+0) + (-0) are +0 in round-to-nearest), so adding a product of +-0.0 changes nothing;
//     the value of an absent entry is finite (a stored 0.0, or one of the two dictionary values, checked finite by the caller) and its x
//     is selected away BEFORE the multiply, so an Inf / NaN elsewhere in x stays where the reference has it;  FULL: no selects at all;
//   * VM 1: the bits of a slab whose rows all carry the same word are ONE scalar (P.sbits): the value of delta k is a scalar select.
template <int U, int VM, int S, int EPI, int FX, bool FULL, bool R3>
__device__ __forceinline__ void pa_pell_slab_fast(const pa_pell_dev P, int slab, int pat, int Wp, unsigned first, int row0, unsigned sb,
                                                  const double *x, double *__restrict__ y, double *gs_x, const double *__restrict__ gs_b,
                                                  const double *__restrict__ gs_diag, const pa_fx fx, const double *sd) {
  static_assert(!R3 || U == 9, "runs of three: groups of nine");
  const int lane = threadIdx.x & 63;
  const int r = slab * 64 + lane;
  const bool live = r < P.n_crows;
  const int row = row0 + S * lane;
  bool mine = live;
  if (FX == 1) {
    const int rw = live ? row : 0;
    if ((fx.rowmask[rw >> 5] >> (rw & 31)) & 1u) mine = false;
  }
  const int *dl = P.pdelta + (size_t)pat * PA_PELL_TW;
  const unsigned long long *pl = P.plane + (size_t)pat * PA_PELL_TW;
  const unsigned *rel = P.prel + (size_t)pat * PA_PELL_TW;
  // every gather is [scalar base xm = column of lane 0 at the class's lowest delta] + [32-bit offset: lane * 8 S + rel[k]]: one 32-bit
  // vector add per gather, one scalar add per run for the elements past the wavefront's end (the scalar pipe, one per CU, had become
  // the busy unit of the one-bit stream: 254 scalar instructions per wavefront, notebook R6.5)
  const char *xm = reinterpret_cast<const char *>(x + row0 + dl[PA_PELL_T_MIN]);
  const unsigned lane_off = (unsigned)lane * (8u * S);
  const double *vp = P.val + (size_t)first * 64;
  double d0 = 0.0, d1 = 0.0;
  if (VM == 1) { d0 = pa_uniform(P.dict[0]); d1 = pa_uniform(P.dict[1]); sb = __builtin_amdgcn_readfirstlane(sb); }
  double acc = 0.0;
  constexpr int UE = U & ~1;
  // a group of U deltas from k0 on (kb: k0 when it is known at compile time, else -1): every gather and scalar load of the group is
  // requested before its first product.  (One-bit stream: the whole row as ONE group of 27 measured slower, 0.155 against 0.146 ms at
  // 256^3 -- 106 scalar registers, 7 waves per SIMD instead of 8.)
  auto group = [&](auto kb_tag, int k0) {
    constexpr int KB = decltype(kb_tag)::value;
    double v[U], a[U];
    pa_pell_cw<U> cw;
    if (VM == 0) {
      const double *g = vp + (size_t)k0 * 64;
#pragma unroll
      for (int j = 0; j < UE; j += 2) {
        const d2 pr = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(g + (size_t)(j >> 1) * 128) + lane);
        v[j] = pr.x; v[j + 1] = pr.y;
      }
      if (U & 1) v[U - 1] = __builtin_nontemporal_load(g + (size_t)UE * 64 + lane);
    } else if (VM == 2) {
      cw = pa_pell_codes<U>(P.codes, first, k0, lane);
    }
    if constexpr (R3) {
      double ee[3], e2[3];
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const unsigned ro = rel[k0 + 3 * t];             // (scalar: the run's first column of lane 0, relative to xm)
        if (S == 1) {
          a[3 * t] = *reinterpret_cast<const double *>(xm + (lane_off + ro));
          const double *xe = reinterpret_cast<const double *>(xm + (ro + 512u));
          ee[t] = xe[0]; e2[t] = xe[1];
        } else {
          const pa_d2u pr = *reinterpret_cast<const pa_d2u *>(xm + (lane_off + ro));
          a[3 * t] = pr.x; a[3 * t + 1] = pr.y;
          ee[t] = *reinterpret_cast<const double *>(xm + (ro + 1024u));
        }
      }
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        if (S == 1) {
          a[3 * t + 1] = pa_wave_shl1(a[3 * t], ee[t]);
          a[3 * t + 2] = pa_wave_shl1(a[3 * t + 1], e2[t]);
        } else {
          a[3 * t + 2] = pa_wave_shl1(a[3 * t], ee[t]);
        }
      }
    } else {
      // no runs: one gather per delta (the 7-point operator, the smoother's lower-colour blocks, ...)
#pragma unroll
      for (int j = 0; j < U; ++j) a[j] = *reinterpret_cast<const double *>(xm + (lane_off + rel[k0 + j]));
    }
    if (VM == 1) {
      if constexpr (KB >= 0) {
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = pa_sel_bit_at(sb, KB + j, d0, d1);
      } else {
        const unsigned sk = sb >> k0;
#pragma unroll
        for (int j = 0; j < U; ++j) v[j] = pa_sel_bit_at(sk, j, d0, d1);
      }
    }
    if (VM == 2) {
#pragma unroll
      for (int j = 0; j < U; ++j) v[j] = pa_pell_value<U>(cw, j, sd);
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      if (!FULL) {
        if (!__builtin_amdgcn_inverse_ballot_w64(pl[k0 + j])) a[j] = 0.0;
      }
      acc = acc + v[j] * a[j];
    }
  };
  if (R3 && Wp == 27) {                                 // (the 27-point row: three groups without the loop around them)
    group(std::integral_constant<int, 0>(), 0);
    group(std::integral_constant<int, 9>(), 9);
    group(std::integral_constant<int, 18>(), 18);
  } else {
    for (int k0 = 0; k0 < Wp; k0 += U) group(std::integral_constant<int, -1>(), k0);
  }
  if (EPI == 0) {
    if (mine) __builtin_nontemporal_store(acc, &y[row]);
  } else if (EPI == 1) {
    if (mine) gs_x[row] = gs_x[row] + (gs_b[row] - acc) / gs_diag[row];
  } else if (EPI == 2) {
    if (mine) gs_x[r] = gs_b[row] - acc;
  } else {
    double dacc = 0.0;
    if (mine) {
      __builtin_nontemporal_store(acc, &y[row]);
      dacc = gs_b[row] * acc;                        // (beta = 0: the product IS the row's new value)
    }
    dacc = pa_wave_sum(dacc);
    if (lane == 0) gs_x[slab] = dacc;
  }
}

// one slab.  EPI / FX as in pa_rowsplit_chunk (pa_spmv_kernel.h): EPI 0 product, 1 Gauss-Seidel colour update in place, 2 residual +
// restriction, 3 product + this slab's term of a dot product (partial[slab]); FX 1: rows whose bit is set in fx.rowmask are left
// alone (the fused launch's tail sums them).
// R3 (U = 9, rows of the slab consecutive: not a row-compacted block): every pattern of the block is made of RUNS OF THREE consecutive
// deltas (d, d+1, d+2 -- the three neighbours along a grid line).  Lane l then needs x[row_l + d + j], j = 0..2, and x[row_l + d + j] IS
// what lane l + j loads for j = 0: ONE gather per run, the other two columns by a wave shift (DPP, a VALU move), the two elements
// past the wavefront's end from two scalar loads -- 9 gathers per row of the 27-point operator instead of 27.  (What-if of round 5,
// notebook R5.8: it is the NUMBER of gather instructions through the texture addresser that costs, not the lines they touch.)
template <int U, int VM, bool COMPACT, int EPI, int FX, bool R3, bool A1 = false>
__device__ __forceinline__ void pa_pell_slab(const pa_pell_dev P, int slab, const double *__restrict__ x_in, double *__restrict__ y,
                                             double alpha, double beta, double *gs_x, const double *__restrict__ gs_b,
                                             const double *__restrict__ gs_diag, const pa_fx fx, const double *sd = nullptr) {
  static_assert(!R3 || U % 9 == 0, "runs of three: unroll 9 (27 on the one-bit stream)");
  // (the clamped runs-of-three form below: consecutive rows, x not written by the launch; a row-compacted block or the Gauss-Seidel
  //  update take runs of three only in the slabs pa_pell_slab_fast serves)
  constexpr bool R3G = R3 && !COMPACT && EPI != 1;
  const double *x = EPI == 1 ? gs_x : x_in;          // EPI 1 reads and writes the same vector: no restrict promise on it
  const int lane = threadIdx.x & 63;
  const int2 d = P.desc[slab];                       // (slab is wave-uniform: scalar loads)
  // (everything that depends on the slab's number alone is requested WITH the descriptor, not behind it: the slab's bits and its first
  //  row id come from HBM like the descriptor, and a wavefront's life is a chain of such round trips)
  uint2 q = make_uint2(0u, 0u);
  int row0 = slab * 64;
  if constexpr (A1) {
    if (P.plane != nullptr) {
      if (VM == 1) q = P.sbits[slab];
      if (COMPACT) row0 = P.row_ids[slab * 64];
    }
  }
  const int pat = d.x & 0xfffff, Wp = d.x >> 20;
  const int *dl = P.pdelta + (size_t)pat * PA_PELL_TW;
  if constexpr (A1) {
    // a slab of a class (P.plane), nothing to add to, every gather of every lane in range: the lean form (all of this is scalar)
    if (P.plane != nullptr && (EPI == 1 || EPI == 2 || beta == 0.0)) {
      const int st = dl[PA_PELL_T_STRIDE];
      bool go = st != 0 && row0 + dl[PA_PELL_T_MIN] >= 0 && row0 + 63 * st + dl[PA_PELL_T_MAX] < P.n_cols;
      const unsigned sb = q.x;
      if (VM == 1) go = go && q.y != 0;
      if (go) {
        const bool full = (dl[PA_PELL_T_FLAGS] & 1) != 0;
        constexpr int UF = R3 ? 9 : U;
        if (COMPACT && st == 2) {
          if (full) pa_pell_slab_fast<UF, VM, 2, EPI, FX, true, R3>(P, slab, pat, Wp, (unsigned)d.y, row0, sb, x, y, gs_x, gs_b, gs_diag, fx, sd);
          else pa_pell_slab_fast<UF, VM, 2, EPI, FX, false, R3>(P, slab, pat, Wp, (unsigned)d.y, row0, sb, x, y, gs_x, gs_b, gs_diag, fx, sd);
          return;
        }
        if (st == 1) {
          if (full) pa_pell_slab_fast<UF, VM, 1, EPI, FX, true, R3>(P, slab, pat, Wp, (unsigned)d.y, row0, sb, x, y, gs_x, gs_b, gs_diag, fx, sd);
          else pa_pell_slab_fast<UF, VM, 1, EPI, FX, false, R3>(P, slab, pat, Wp, (unsigned)d.y, row0, sb, x, y, gs_x, gs_b, gs_diag, fx, sd);
          return;
        }
      }
    }
  }
  const int r = slab * 64 + lane;
  const bool live = r < P.n_crows;
  const int rc = live ? r : P.n_crows - 1;
  unsigned long long m = live ? P.mask[rc] : 0u;
  const int row = COMPACT ? P.row_ids[rc] : rc;
  bool mine = live;
  if (FX == 1) {
    if ((fx.rowmask[row >> 5] >> (row & 31)) & 1u) { mine = false; m = 0; }      // a boundary row: the launch's tail sums and stores it
  }
  unsigned long long vb = 0;
  double d0 = 0.0, d1 = 0.0;
  if (VM == 1) { vb = P.bits[rc]; d0 = P.dict[0]; d1 = P.dict[1]; }
  const double *vp = P.val + (size_t)(unsigned)d.y * 64;
  double acc = 0.0, accp = 0.0;
  if ((EPI == 0 || EPI == 3) && beta != 0.0 && mine) acc = beta * y[row];
  constexpr int UE = U & ~1;
  for (int k0 = 0; k0 < Wp; k0 += U) {
    double v[U], xv[U];
    bool on[U];
    if (VM == 0) {
      const double *g = vp + (size_t)k0 * 64;
#pragma unroll
      for (int j = 0; j < UE; j += 2) {
        const d2 pr = __builtin_nontemporal_load(reinterpret_cast<const d2 *>(g + (size_t)(j >> 1) * 128) + lane);
        v[j] = pr.x; v[j + 1] = pr.y;
      }
      if (U & 1) v[U - 1] = __builtin_nontemporal_load(g + (size_t)UE * 64 + lane);
    } else if (VM == 1) {
#pragma unroll
      for (int j = 0; j < U; ++j) v[j] = ((vb >> (k0 + j)) & 1ull) ? d1 : d0;
    } else {
      const pa_pell_cw<U> cw = pa_pell_codes<U>(P.codes, (unsigned)d.y, k0, lane);
#pragma unroll
      for (int j = 0; j < U; ++j) v[j] = pa_pell_value<U>(cw, j, sd);
    }
#pragma unroll
    for (int j = 0; j < U; ++j) on[j] = (m >> (k0 + j)) & 1ull;
    if (R3G) {
#pragma unroll
      for (int t = 0; t < U / 3; ++t) {
        const int dk = dl[k0 + 3 * t];               // the run's first delta (scalar)
        const int hi = P.n_cols - 1;
        const double xb = x[min(max(r + dk, 0), hi)];
        const int e = slab * 64 + 64 + dk;           // the two columns past the wavefront's last lane (scalar loads)
        const double e1 = x[min(max(e, 0), hi)], e2 = x[min(max(e + 1, 0), hi)];
        xv[3 * t] = xb;
        xv[3 * t + 1] = pa_wave_shl1(xb, e1);
        xv[3 * t + 2] = pa_wave_shl1(xv[3 * t + 1], e2);
      }
    } else {
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int c = on[j] ? row + dl[k0 + j] : 0;
        xv[j] = x[c];
      }
    }
#pragma unroll
    for (int j = 0; j < U; ++j) {
      double pr = v[j] * xv[j];
      if (!A1 && alpha != 1.0) pr = pr * alpha;   // (A1: the launch knows alpha = 1 -- a multiply and a select per product less; on the
      if (on[j]) {                                //  one-bit stream the vector ALU is the busy unit, notebook R6.1)
        acc = acc + pr;
        if (EPI == 3) accp = accp + pr;
      }
    }
  }
  if (EPI == 0) {
    if (mine) __builtin_nontemporal_store(acc, &y[row]);
  } else if (EPI == 1) {
    if (mine) gs_x[row] = gs_x[row] + (gs_b[row] - acc) / gs_diag[row];
  } else if (EPI == 2) {
    if (mine) gs_x[r] = gs_b[row] - acc;
  } else {
    double dacc = 0.0;
    if (mine) {
      __builtin_nontemporal_store(acc, &y[row]);
      dacc = gs_b[row] * accp;
    }
    dacc = pa_wave_sum(dacc);
    if (lane == 0) gs_x[slab] = dacc;
  }
}

// host side of a block's pattern-ELL storage (pa_pell.hip builds it; pa_f32.hip shares the structure for Float32 blocks)
struct pa_pell {
  int U = 9;
  int64_t n_slabs = 0, n_patterns = 0, slots = 0;      // slots: 64-entry units of the value stream
  int64_t n_table = 0, n_classes = 0;                  // rows of the table (classes when there are any, else patterns)
  int64_t n_lean = 0, n_lean_bits = 0;                 // slabs the lean form serves on the fp64 stream / on the one-bit stream (set-up counts)
  int max_w = 0;
  bool runs3 = false;                                  // every pattern is made of runs of three consecutive deltas (and U = 9)
  int2 *d_desc = nullptr;
  int *d_pdelta = nullptr;
  unsigned *d_mask = nullptr, *d_bits = nullptr;
  unsigned *d_codes = nullptr;                         // one byte per entry (dictionaries of 3 .. 64 values): slots / U groups x 64 lanes x ceil(U / 4) dwords
  uint64_t codes_epoch = ~(uint64_t)0;                 // A->val_epoch the codes were made at
  unsigned long long *d_plane = nullptr;               // classes: lane ballots, n_table x PA_PELL_TW
  unsigned *d_prel = nullptr;                          // classes: byte offsets relative to the lowest delta, n_table x PA_PELL_TW
  uint2 *d_sbits = nullptr;                            // one-bit stream: per slab {bits, uniform}, and behind them the count of uniform lean slabs
  double *d_val = nullptr;
  uint64_t bits_epoch = ~(uint64_t)0;                  // A->val_epoch the bits were made at
  uint64_t n_launched = 0;
};

pa_pell *pa_pell_structure(pa_ctx *c, const int32_t *d_crp, const int32_t *d_col, const int32_t *d_row_ids, int64_t n_crows, int64_t n_cols,
                           int64_t nnz, bool compact, const char **why);
void pa_pell_struct_free(pa_ctx *c, pa_pell *P);

// blockIdx -> slabs: four slabs per workgroup (one per wavefront), consecutive workgroups of an XCD take consecutive slabs (block b
// sits on XCD b % 8; each XCD has its own L2, and the rows of neighbouring grid lines and planes share their x).  bpx < 0: the same
// map walked backwards (every other product of a big block: what the last product left in the caches is read first, PA_SPMV_ALTERNATE).
template <int U, int VM, bool COMPACT, int EPI, bool R3 = false, bool A1 = false>
__global__ __launch_bounds__(256) void k_spmv_pell(const pa_pell_dev P, const double *__restrict__ x, double *__restrict__ y, int bpx,
                                                   double alpha, double beta, double *gs_x, const double *__restrict__ gs_b,
                                                   const double *__restrict__ gs_diag) {
  __shared__ double sdict[VM == 2 ? PA_VDICT_MAX : 1];
  if (VM == 2) {                                     // (before anyone leaves: the workgroup's copy of the dictionary)
    if (threadIdx.x < PA_VDICT_MAX) sdict[threadIdx.x] = P.dict[threadIdx.x];
    __syncthreads();
  }
  const int b = blockIdx.x;
  const bool backwards = bpx < 0;
  if (backwards) bpx = -bpx;
  int g = (b & 7) * bpx + (b >> 3);
  const int n_groups = (P.n_slabs + 3) >> 2;
  if (g >= n_groups) return;
  if (backwards) g = n_groups - 1 - g;
  const int slab = __builtin_amdgcn_readfirstlane(g * 4 + (int)(threadIdx.x >> 6));
  if (slab >= P.n_slabs) return;
  pa_pell_slab<U, VM, COMPACT, EPI, 0, R3, A1>(P, slab, x, y, alpha, beta, gs_x, gs_b, gs_diag, pa_fx(), sdict);
}

#endif
