// Row-split CSR SpMV for banded but unstructured rows: the same chunks, the same streams and the same arithmetic as
// k_spmv_rowsplit (pa_spmv_kernel.h), with the gathers served from LDS.
//
// Why: a block whose rows follow no pattern reads x through one gather per stored entry.  With columns spread over a
// band of a few thousand (an unstructured mesh in a bandwidth-reducing numbering), the 64 lanes of one gather touch ~60
// different cache lines, and the ~32 KiB of x one chunk needs does not survive in a 32 KiB L1 shared by 4-5 resident
// workgroups: most gathers become L2 requests of a whole line for 8 useful bytes (measured: 0.216 ms for 64 M entries,
// 0.131 ms with the same kernel reading x lane-contiguously).  Here a workgroup owns a GROUP of consecutive chunks whose
// columns span at most PA_XW_CAP entries of x; it copies that span into LDS once, with full-width coalesced loads, and
// every gather of the group's chunks is a ds_read_b64.  x then costs (span / entries of the group) * 8 bytes per entry of
// coalesced L2 traffic instead of a line per gather.
//
// Bit-exactness: unchanged.  Products val[p] * x[col[p]] are formed one by one (no FMA), land in their own LDS slot and
// are summed by the row's lane in ascending p (src/sparse_utils.jl:649-669, spmv_csr!).
//
// A workgroup is SUB sub-groups of 256 lanes; sub-group s works on chunks first+s, first+s+SUB, ... of the group, all
// share the x window.  Chunks that do not fit a group (a span wider than the cap, 32-bit columns, a row longer than a
// chunk) stay on k_spmv_rowsplit through its chunk list.
#pragma once
#include "pa_spmv_kernel.h"

// Three window sizes: 40 KiB (two workgroups = 16 waves per CU) and, for the chunks whose span does not fit that, 96 KiB
// with four sub-groups and 128 KiB with two (one workgroup per CU: 4 M rows x 16 within +-4000 / +-6000 0.158 / 0.164 ms against 0.245 / 0.263 on the row split;
// within +-2000 the small window is the faster one, 0.126 against 0.158).
#ifndef PA_XW_CAP
#define PA_XW_CAP 5120      // doubles of x a workgroup stages (40 KiB)
#endif
// ... 96 KiB with four sub-groups (one workgroup of 16 waves per CU) and 128 KiB with two (8 waves): what the 160 KiB of
// LDS leave next to the sub-groups' product slots (12284 / 16380 entries with the shipped 1536 products per chunk)
constexpr int pa_xw_cap_for(int sub, int want) {
  const int pcap = PA_SPMV_CHUNK_NNZ + PA_SPMV_CHUNK_NNZ / 16 + 2;
  const int room = ((163840 - 256 - sub * pcap * 8 - sub * 32) / 8 - 4) & ~3;
  return room < want ? room : want;
}
#define PA_XW_CAP_MID pa_xw_cap_for(4, 12284)
#define PA_XW_CAP_BIG pa_xw_cap_for(2, 16380)
#define PA_XW_TIERS 3
#ifndef PA_XW_MAXG
#define PA_XW_MAXG 16       // chunks per group at most (fewer on a small block: see pa_build_xw_groups)
#endif
#define PA_XW_WANT_GROUPS 900    // a launch of fewer workgroups than about two rounds of the 512 resident ones shows its tail
#ifndef PA_XW_UNROLL
#define PA_XW_UNROLL 4      // products a lane fetches ahead of its running sum in the reduce phase
#endif
#ifndef PA_XW_SUB
#define PA_XW_SUB 2         // sub-groups of 256 lanes per workgroup (chunks of a group in flight at a time)
#endif
#define PA_XW_PSLOT(p) ((p) + 2 * ((p) >> 5))   // two pad slots per 32 products: pairs stay 16-byte aligned (see pa_spmv_kernel.h)
#define PA_XW_MIN_LINES 48  // a group's chunks must touch, on average, this many 128-byte lines of x each (6 KiB): below that
                            // the lines stay in L1 between the gathers of a chunk and k_spmv_rowsplit is the faster kernel
                            // (a grid with 2 interleaved unknowns per node: ~20 lines per chunk, 0.077 ms against 0.083 here)
#define PA_XW_MING 4        // a shorter group would move more x than matrix: its chunks go to k_spmv_rowsplit

struct pa_xw_group { int first, cnt, wlo, wlen; };

// DOT: also partial[chunk] = sum over the chunk's rows of u[row] * (sum of the row's products), in the order of
// k_spmv_rowsplit's EPI 3 (per lane in row order, pa_wave_sum, the four wave sums left to right): the same bits whichever
// kernel a chunk runs on.
// BLK lanes work on one chunk: 256 x 6 entries, or 512 x 4 (the lanes past the chunk's 1536 entries idle) where the LDS
// leaves room for one workgroup per CU only and twice the waves help (the 128 KiB windows).
template <int SUB, int NPT, bool NT, bool DOT = false, int XCAP = PA_XW_CAP, int BLK = 256>
__global__ __launch_bounds__(BLK * SUB) void k_spmv_xwin(
    const int *__restrict__ crp, const unsigned short *__restrict__ col16, const int *__restrict__ win,
    const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y,
    const int *__restrict__ chunk_row, const int *__restrict__ chunk_p, const pa_xw_group *__restrict__ grp,
    int n_groups, int groups_per_xcd, int n_cols, double alpha, double beta, const double *__restrict__ u = nullptr,
    double *__restrict__ partial = nullptr) {
  constexpr int CAP = PA_SPMV_CHUNK_NNZ, NTHR = BLK * SUB;
  constexpr int PCAP = CAP + CAP / 16 + 2;            // padded product slots: rows of 2^k entries miss each other's banks
  static_assert(BLK * NPT >= CAP, "the lanes of a sub-group cover a chunk");
  __shared__ __attribute__((aligned(16))) double xs[XCAP + 4];
  __shared__ __attribute__((aligned(16))) double prod_all[SUB * PCAP];
  __shared__ double wsum[SUB * (BLK / 64)];
  const int tid = threadIdx.x;
  const int t = tid & (BLK - 1);
  const int sub = __builtin_amdgcn_readfirstlane(tid / BLK);
  double *prod = prod_all + sub * PCAP;
  const int b = blockIdx.x;
  const int g = (b & 7) * groups_per_xcd + (b >> 3);  // XCD-aware: workgroup b sits on XCD b%8, neighbours in g share an L2
  if (g >= n_groups || (b >> 3) >= groups_per_xcd) return;
  const pa_xw_group G = grp[g];
  const int ch_end = G.first + G.cnt;

  d2 v[NPT / 2];
  unsigned q[NPT / 2];
  int mywin = 0, ra = 0, re = 0;
  double ur = 0.0;                                    // DOT: u of this lane's first row
  int r0 = 0, r1 = 0, p0 = 0, p1 = 0;                 // the chunk whose loads are in flight
  int nr0 = 0, nr1 = 0, np0 = 0, np1 = 0;             // the one after it (row and entry bounds only)
  auto meta = [&](int ch, int &a0, int &a1, int &b0, int &b1) {
    a0 = chunk_row[ch]; a1 = chunk_row[ch + 1]; b0 = chunk_p[ch]; b1 = chunk_p[ch + 1];
  };
  auto issue = [&](int ch) {                          // every load of chunk ch this lane needs; none is waited for here
    const int base = p0 & ~1, last = max((p1 - 1) & ~1, 0);
#pragma unroll
    for (int k = 0; k < NPT / 2; ++k) {
      const int idx = min(base + (k * BLK + t) * 2, last);
      v[k] = pa_stream_load<NT>(reinterpret_cast<const d2 *>(val + idx));
      q[k] = pa_stream_load<NT>(reinterpret_cast<const unsigned *>(col16 + idx));
    }
    mywin = win[ch * PA_C16_WINDOWS + (t & (PA_C16_WINDOWS - 1))];
    if (r0 + t < r1) {
      ra = crp[r0 + t];
      re = crp[r0 + t + 1];
      if (DOT) ur = u[r0 + t];
    }
  };
  int ch = G.first + sub;
  if (ch < ch_end) {
    meta(ch, r0, r1, p0, p1);
    issue(ch);
    if (ch + SUB < ch_end) meta(ch + SUB, nr0, nr1, np0, np1);
  }
  // the x window: element wl of x sits at xs[0]; wl <= wlo is chosen so that x + wl is 16-byte aligned
  const int wl = G.wlo - (int)(((reinterpret_cast<uintptr_t>(x) >> 3) + (uintptr_t)G.wlo) & 1);
  {
    const int npair = (G.wlo + G.wlen - wl + 1) >> 1;
    if (wl >= 0 && wl + 2 * npair <= n_cols) {        // (block-uniform) every pair lies inside x: 16-byte loads
      constexpr int KX = (XCAP / 2 + 2 + NTHR - 1) / NTHR;
#ifndef PA_XW_NO_GLDS
      // global_load_lds_dwordx4: 16 bytes per lane straight into LDS at (wave-uniform base) + lane * 16 -- the window is
      // lane-linear, so no register round trip and no ds_write pass; lanes past the window's end are switched off by the
      // guard (a masked lane neither loads nor writes).  The compiler drains these loads (vmcnt) before the barrier below.
      const int wave0 = __builtin_amdgcn_readfirstlane(tid & ~63);
#pragma unroll
      for (int k = 0; k < KX; ++k)
        if (tid + k * NTHR < npair)
          __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(x + wl + 2 * (tid + k * NTHR)),
                                           (__attribute__((address_space(3))) void *)(xs + 2 * (k * NTHR + wave0)), 16, 0, 0);
#else
      d2 xv[KX];
#pragma unroll
      for (int k = 0; k < KX; ++k) xv[k] = *reinterpret_cast<const d2 *>(x + wl + 2 * min(tid + k * NTHR, npair - 1));
#pragma unroll
      for (int k = 0; k < KX; ++k)
        if (tid + k * NTHR < npair) *reinterpret_cast<d2 *>(&xs[2 * (tid + k * NTHR)]) = xv[k];
#endif
    } else {                                          // the first / last window of the vector
      for (int i = tid; i < 2 * npair; i += NTHR) {
        const int e = wl + i;
        xs[i] = (e >= 0 && e < n_cols) ? x[e] : 0.0;
      }
    }
  }
  __syncthreads();
  for (; __builtin_amdgcn_readfirstlane(ch - sub) < ch_end; ch += SUB) {   // every sub-group runs the same number of rounds
    const bool act = ch < ch_end;                                           // wave-uniform
    const int cr0 = r0, cr1 = r1, cbase = p0 & ~1, cra = ra, cre = re;
    const double cur = ur;
    if (act) {
      const int wrel = mywin - wl;
#pragma unroll
      for (int k = 0; k < NPT / 2; ++k) {
        const unsigned lo = q[k] & 0xffffu, hi = q[k] >> 16;
        // (the entry before an odd first entry and the one after an odd last entry belong to the neighbouring chunks: decoded
        // with this chunk's windows they give any index; their products are never summed, but the reads stay inside xs)
        const int c0 = min(max(__builtin_amdgcn_ds_bpermute((lo >> 12) << 2, wrel) + (int)(lo & 4095), 0), XCAP + 3);
        const int c1 = min(max(__builtin_amdgcn_ds_bpermute((hi >> 12) << 2, wrel) + (int)(hi & 4095), 0), XCAP + 3);
        double a = v[k].x * xs[c0];
        double c = v[k].y * xs[c1];
        if (alpha != 1.0) {
          a = a * alpha;
          c = c * alpha;
        }
        d2 pr;
        pr.x = a;
        pr.y = c;
        if (BLK * NPT == CAP || (k * BLK + t) * 2 < CAP) *reinterpret_cast<d2 *>(&prod[PA_XW_PSLOT((k * BLK + t) * 2)]) = pr;
      }
      // the next chunk's loads go out before this one's row sums: they fly during the barrier and the reduce phase
      if (ch + SUB < ch_end) {
        r0 = nr0; r1 = nr1; p0 = np0; p1 = np1;
        issue(ch + SUB);
        if (ch + 2 * SUB < ch_end) meta(ch + 2 * SUB, nr0, nr1, np0, np1);
      }
    }
    __syncthreads();
    if (act) {
      int a = cra - cbase, e = cre - cbase;
      double dacc = 0.0;
      for (int r = cr0 + t; r < cr1; r += BLK) {
        if (r != cr0 + t) {
          a = crp[r] - cbase;
          e = crp[r + 1] - cbase;
        }
        double acc = beta == 0.0 ? 0.0 : beta * y[r];
#pragma unroll PA_XW_UNROLL
        for (int p = a; p < e; ++p) acc = acc + prod[PA_XW_PSLOT(p)];
        if (DOT) {
          double pr = acc;                            // the row's products alone
          if (beta != 0.0) {
            pr = 0.0;
            for (int p = a; p < e; ++p) pr = pr + prod[PA_XW_PSLOT(p)];
          }
          dacc = dacc + (r == cr0 + t ? cur : u[r]) * pr;
        }
        __builtin_nontemporal_store(acc, &y[r]);
      }
      if (DOT) {
        dacc = pa_wave_sum(dacc);
        if (cr1 - cr0 <= 64) {
          if (t == 0) partial[ch] = dacc;
        } else if ((t & 63) == 0) wsum[sub * (BLK / 64) + (t >> 6)] = dacc;
      }
    }
    __syncthreads();
    if (DOT && act && cr1 - cr0 > 64 && t == 0) {
      double sum = 0.0;
      for (int w = 0; w < BLK / 64; ++w) sum = sum + wsum[sub * (BLK / 64) + w];
      partial[ch] = sum;
    }
  }
}

// ---- the sliding x window ("ring") -------------------------------------------------------------------------------------------
// Round 3 (VERDICT r02 #6a).  The windows above copy the span of a group of <= 16 chunks: a band of +-B costs 2B + rows of x
// per group, and beyond +-7000 no group fits the LDS.  But consecutive chunks of a banded block need ALMOST THE SAME x: the
// span moves by the chunk's rows (~100 entries) while it is 2B wide.  Here a workgroup owns a long run of consecutive chunks
// and keeps x in a RING of PA_XR_CAP entries (128 KiB): entry e of x lives in slot e mod PA_XR_CAP, every round (SUB chunks)
// appends the few new entries above the highest one loaded so far -- fetched at the top of the round, written behind the
// round's first barrier, when nobody gathers any more -- and overwrites the oldest ones, which no later chunk reads (the host
// checks: a chunk's smallest column is less than PA_XR_CAP below the highest column of its round and of all before it).  x
// is then read from L2 once per run of chunks plus once per entry, whatever the band (up to +-8000), and a gather's slot is
// the column itself masked -- no window base to subtract.  Same streams, same products, same order as k_spmv_rowsplit: same bits.
#define PA_XR_CAP 16384
#define PA_XR_MAXG 1024     // chunks per ring group at most (fewer on a small block)
#define PA_XR_WANT_GROUPS 256    // one workgroup per CU is resident (157 KB of LDS) and every run pays a first fill of its whole span
                                 // (a latency-bound phase nothing overlaps): ONE run per CU.  Measured (round 4, 4 M rows x 16, +-7900):
                                 // 256 runs 0.1537 ms = 5.5 TB/s algorithmic, 512 0.1593, 768 0.1651, 1024 0.1685, 2048 0.1918
                                 // (PA_SPMV_XRING_GROUPS); 772 runs -- 3 per CU and 4 left over -- had run a quarter longer than 768

// BLK lanes work on one chunk (256 x 6 entries each as everywhere else, or 512 x 4 with the lanes past the chunk's 1536
// entries idle: twice the waves on the CU for the same LDS)
// One run of chunks (a ring group G) by the whole workgroup; the ring xs, the product slots and wsum are the caller's.  keep: y is
// read again by THIS workgroup (the next column piece of a chain, k_spmv_xring_chain): a plain store that stays in L2 instead of
// the streaming one.  Returns behind a barrier: the LDS is free.
template <int SUB, int NPT, bool NT, bool DOT, int BLK>
__device__ __forceinline__ void pa_xring_run(
    double *__restrict__ xs, double *__restrict__ prod_all, double *__restrict__ wsum,
    const int *__restrict__ crp, const unsigned short *__restrict__ col16, const int *__restrict__ win,
    const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y,
    const int *__restrict__ chunk_row, const int *__restrict__ chunk_p, const int *__restrict__ chunk_cmax,
    const pa_xw_group G, int n_cols, double alpha, double beta, const double *__restrict__ u, double *__restrict__ partial,
    const bool keep) {
  constexpr int CAP = PA_SPMV_CHUNK_NNZ, NTHR = BLK * SUB, C = PA_XR_CAP;
  constexpr int PCAP = CAP + CAP / 16 + 2;
  // rows per pass of the row phase.  The fused dot must form a chunk's partial sum in k_spmv_rowsplit's order (lane t adds the rows
  // t, t + 256, ..., then the wave sums, then the four waves in turn): with 512 lanes on a chunk of more than 256 rows a lane per
  // row gives other bits, so the dot variant sums rows with the first 256 lanes only (round 4: found when one run per CU brought
  // chunks of short rows into the ring)
  constexpr int RB = DOT && BLK > 256 ? 256 : BLK;
  static_assert(BLK * NPT >= CAP, "the lanes of a sub-group cover a chunk");
  const int tid = threadIdx.x;
  const int t = tid & (BLK - 1);
  const int sub = __builtin_amdgcn_readfirstlane(tid / BLK);
  double *prod = prod_all + sub * PCAP;
  const int ch_end = G.first + G.cnt;

  d2 v[NPT / 2];
  unsigned q[NPT / 2];
  int mywin = 0, ra = 0, re = 0;
  double ur = 0.0, yr = 0.0;
  int r0 = 0, r1 = 0, p0 = 0, p1 = 0;
  int nr0 = 0, nr1 = 0, np0 = 0, np1 = 0;
  auto meta = [&](int ch, int &a0, int &a1, int &b0, int &b1) {
    a0 = chunk_row[ch]; a1 = chunk_row[ch + 1]; b0 = chunk_p[ch]; b1 = chunk_p[ch + 1];
  };
  auto issue = [&](int ch) {
    const int base = p0 & ~1, last = max((p1 - 1) & ~1, 0);
#pragma unroll
    for (int k = 0; k < NPT / 2; ++k) {
      const int idx = min(base + (k * BLK + t) * 2, last);
      v[k] = pa_stream_load<NT>(reinterpret_cast<const d2 *>(val + idx));
      q[k] = pa_stream_load<NT>(reinterpret_cast<const unsigned *>(col16 + idx));
    }
    mywin = win[ch * PA_C16_WINDOWS + (t & (PA_C16_WINDOWS - 1))];
    if (r0 + t < r1) {
      ra = crp[r0 + t];
      re = crp[r0 + t + 1];
      if (DOT) ur = u[r0 + t];
      // (an accumulating run -- a column piece, own x ghost -- reads y: a round ahead like the row bounds, one workgroup per CU
      // has nobody to hide the load behind)
      if (beta != 0.0) yr = y[r0 + t];
    }
  };
  // highest column of the chunks of the round that starts at chunk c0 (block-uniform)
  auto round_hi = [&](int c0) {
    int h = -1;
#pragma unroll
    for (int s = 0; s < SUB; ++s) if (c0 + s < ch_end) h = max(h, chunk_cmax[c0 + s]);
    return h;
  };
  int ch = G.first + sub;
  if (ch < ch_end) {
    meta(ch, r0, r1, p0, p1);
    issue(ch);
    if (ch + SUB < ch_end) meta(ch + SUB, nr0, nr1, np0, np1);
  }
  // first fill: [wlo, highest column of the first round]
  int hcur = min(round_hi(G.first), n_cols - 1);
  int hi1 = G.first + SUB < ch_end ? round_hi(G.first + SUB) : -1;           // highest column of the round after this one,
  for (int e = G.wlo + tid; e <= hcur; e += NTHR) xs[e & (C - 1)] = x[e];      // fetched a round ahead (its load must not
  __syncthreads();                                                             // sit in front of the ring's own)
  for (int c0 = G.first; c0 < ch_end; c0 += SUB, ch += SUB) {                 // every sub-group runs the same rounds
    const bool act = ch < ch_end;                                              // wave-uniform
    const int cr0 = r0, cr1 = r1, cbase = p0 & ~1, cra = ra, cre = re;
    const double cur = ur, cyr = yr;
    // the ring's new entries for the NEXT round: fetched now, they fly while this round's products are formed
    const int hnext = c0 + SUB < ch_end ? min(max(hi1, hcur), n_cols - 1) : hcur;
    hi1 = c0 + 2 * SUB < ch_end ? round_hi(c0 + 2 * SUB) : -1;
    const int e_new = hcur + 1 + tid;
    double xnew = 0.0;
    if (e_new <= hnext) xnew = x[e_new];
    if (act) {
      // Round 4: no ds_bpermute for the window bases.  A gather's ring slot is its column mod 16384 = ((window base + offset) mod
      // 16384), and a window base is a multiple of 4096: only TWO BITS of each of the chunk's 16 bases matter.  They travel as two
      // wave-wide ballots (lanes 0..15 hold the bases of slots 0..15), and a slot's pair of bits is picked with shifts -- VALU work
      // that was idle, where the two bpermutes per pair of entries went through the LDS pipeline the gathers are waiting on:
      // +-7900 0.1723 -> 0.1685 ms = 5.03 TB/s algorithmic.  (The same with three bits per base in k_spmv_xwin, where two workgroups
      // share a CU and the VALU is not idle, LOST 3-9 %: not there.)
      static_assert(C == 16384, "the two-bit window trick assumes a ring of 4 x 4096 entries");
      const unsigned long long wb0 = __ballot((mywin & 4096) != 0), wb1 = __ballot((mywin & 8192) != 0);
      const unsigned w0 = (unsigned)wb0 & 0xffffu, w1 = (unsigned)wb1 & 0xffffu;
#pragma unroll
      for (int k = 0; k < NPT / 2; ++k) {
        const unsigned lo = q[k] & 0xffffu, hi = q[k] >> 16;
        // (lanes outside the chunk decode a neighbour's code with this chunk's windows: any column, masked into the ring;
        // their products are never summed)
        const unsigned s0 = lo >> 12, s1 = hi >> 12;
        const int c0_ = (int)(((((w0 >> s0) & 1u) | (((w1 >> s0) & 1u) << 1)) << 12) | (lo & 4095u));
        const int c1_ = (int)(((((w0 >> s1) & 1u) | (((w1 >> s1) & 1u) << 1)) << 12) | (hi & 4095u));
        double a = v[k].x * xs[c0_];
        double c = v[k].y * xs[c1_];
        if (alpha != 1.0) {
          a = a * alpha;
          c = c * alpha;
        }
        d2 pr;
        pr.x = a;
        pr.y = c;
        if (BLK * NPT == CAP || (k * BLK + t) * 2 < CAP) *reinterpret_cast<d2 *>(&prod[PA_XW_PSLOT((k * BLK + t) * 2)]) = pr;
      }
      if (ch + SUB < ch_end) {
        r0 = nr0; r1 = nr1; p0 = np0; p1 = np1;
        issue(ch + SUB);
        if (ch + 2 * SUB < ch_end) meta(ch + 2 * SUB, nr0, nr1, np0, np1);
      }
    }
    __syncthreads();                                                           // nobody gathers any more: the ring may move
    if (e_new <= hnext) xs[e_new & (C - 1)] = xnew;
    for (int e = e_new + NTHR; e <= hnext; e += NTHR) xs[e & (C - 1)] = x[e];    // (a jump of more than 512 columns: rare)
    hcur = hnext;
    if (act) {
      int a = cra - cbase, e = cre - cbase;
      double dacc = 0.0;
      for (int r = cr0 + t; r < cr1 && t < RB; r += RB) {
        if (r != cr0 + t) {
          a = crp[r] - cbase;
          e = crp[r + 1] - cbase;
        }
        double acc = beta == 0.0 ? 0.0 : beta * (r == cr0 + t ? cyr : y[r]);
#pragma unroll PA_XW_UNROLL
        for (int p = a; p < e; ++p) acc = acc + prod[PA_XW_PSLOT(p)];
        if (DOT) {
          double pr = acc;
          if (beta != 0.0) {
            pr = 0.0;
            for (int p = a; p < e; ++p) pr = pr + prod[PA_XW_PSLOT(p)];
          }
          dacc = dacc + (r == cr0 + t ? cur : u[r]) * pr;
        }
        if (keep) y[r] = acc;
        else __builtin_nontemporal_store(acc, &y[r]);
      }
      if (DOT) {
        dacc = pa_wave_sum(dacc);
        if (cr1 - cr0 <= 64) {
          if (t == 0) partial[ch] = dacc;
        } else if ((t & 63) == 0) wsum[sub * (BLK / 64) + (t >> 6)] = dacc;
      }
    }
    __syncthreads();
    if (DOT && act && cr1 - cr0 > 64 && t == 0) {
      double sum = 0.0;
      for (int w = 0; w < RB / 64; ++w) sum = sum + wsum[sub * (BLK / 64) + w];
      partial[ch] = sum;
    }
  }
}

template <int SUB, int NPT, bool NT, bool DOT = false, int BLK = 256>
__global__ __launch_bounds__(BLK * SUB) void k_spmv_xring(
    const int *__restrict__ crp, const unsigned short *__restrict__ col16, const int *__restrict__ win,
    const double *__restrict__ val, const double *__restrict__ x, double *__restrict__ y,
    const int *__restrict__ chunk_row, const int *__restrict__ chunk_p, const int *__restrict__ chunk_cmax,
    const pa_xw_group *__restrict__ grp, int n_groups, int groups_per_xcd, int n_cols, double alpha, double beta,
    const double *__restrict__ u = nullptr, double *__restrict__ partial = nullptr) {
  constexpr int PCAP = PA_SPMV_CHUNK_NNZ + PA_SPMV_CHUNK_NNZ / 16 + 2;
  __shared__ __attribute__((aligned(16))) double xs[PA_XR_CAP];
  __shared__ __attribute__((aligned(16))) double prod_all[SUB * PCAP];
  __shared__ double wsum[SUB * (BLK / 64)];
  const int b = blockIdx.x;
  const int g = (b & 7) * groups_per_xcd + (b >> 3);
  if (g >= n_groups || (b >> 3) >= groups_per_xcd) return;
  pa_xring_run<SUB, NPT, NT, DOT, BLK>(xs, prod_all, wsum, crp, col16, win, val, x, y, chunk_row, chunk_p, chunk_cmax, grp[g], n_cols,
                                       alpha, beta, u, partial, false);
}

// ---- a column-split chain in ONE launch (round 5) ----------------------------------------------------------------
// The pieces of a chain (pa_transpose.hip) hold all rows each; run as k launches, y goes to HBM and back between them.  When the
// pieces' chunks and ring groups are cut at the SAME rows (pa_csr_colsplit_if_wide passes the row breaks to the chunker), group g of
// every piece covers the same rows, and one workgroup runs group g of piece 0, then of piece 1, ...: the rows' partial sums are
// stored plainly and read back by the same workgroup a piece later -- from L2 (a group's rows: ~16 K x 8 B), y reaches HBM once.
// The products of a row are added in the order the separate launches add them (piece after piece, each onto the stored fp64 sum
// of those before): the same bits.
// (the table holds ADDRESSES: a pointer loaded from memory is a generic pointer to the compiler -- flat loads, which count on both
// wait counters and drain the LDS queue with every use -- where a kernel argument is known to point to global memory; pa_global
// says so for these)
struct pa_chain_piece {
  unsigned long long crp, col16, win, val, chunk_row, chunk_p, chunk_cmax, grp;
};
template <typename T>
__device__ __forceinline__ const T *pa_global(unsigned long long a) {
  return (const T *)(const T __attribute__((address_space(1))) *)a;
}
template <int SUB, int NPT, bool NT, int BLK>
__global__ __launch_bounds__(BLK * SUB) void k_spmv_xring_chain(const pa_chain_piece *__restrict__ pieces, int n_pieces,
                                                               const double *__restrict__ x, double *__restrict__ y, int n_groups,
                                                               int groups_per_xcd, int n_cols, double alpha, double beta) {
  constexpr int PCAP = PA_SPMV_CHUNK_NNZ + PA_SPMV_CHUNK_NNZ / 16 + 2;
  __shared__ __attribute__((aligned(16))) double xs[PA_XR_CAP];
  __shared__ __attribute__((aligned(16))) double prod_all[SUB * PCAP];
  __shared__ double wsum[SUB * (BLK / 64)];
  const int b = blockIdx.x;
  const int g = (b & 7) * groups_per_xcd + (b >> 3);
  if (g >= n_groups || (b >> 3) >= groups_per_xcd) return;
  for (int j = 0; j < n_pieces; ++j) {
    const pa_chain_piece P = pieces[j];
    pa_xring_run<SUB, NPT, NT, false, BLK>(xs, prod_all, wsum, pa_global<int>(P.crp), pa_global<unsigned short>(P.col16), pa_global<int>(P.win),
                                           pa_global<double>(P.val), x, y, pa_global<int>(P.chunk_row), pa_global<int>(P.chunk_p),
                                           pa_global<int>(P.chunk_cmax), pa_global<pa_xw_group>(P.grp)[g], n_cols, alpha, j == 0 ? beta : 1.0,
                                           nullptr, nullptr, j + 1 < n_pieces);
  }
}

// Host side, per chunk (multi-threaded over chunks): first and last column, and how many distinct 128-byte lines of x (16
// entries) its gathers touch -- only for chunks on the 16-bit stream whose span fits the largest window.
struct pa_xw_chunk_stats { std::vector<int32_t> cmin, cmax, lines; };
inline void pa_xw_scan_chunks(const int32_t *crp, const int32_t *col, const std::vector<int32_t> &chunk_row, const int32_t *win,
                              int max_cap, int n_threads, pa_xw_chunk_stats &S) {
  const int64_t n_chunks = (int64_t)chunk_row.size() - 1;
  S.cmin.assign(n_chunks, INT32_MAX); S.cmax.assign(n_chunks, -1); S.lines.assign(n_chunks, 0);
  auto work = [&](int t, int T) {
    std::vector<uint64_t> bits;
    for (int64_t c = n_chunks * t / T; c < n_chunks * (t + 1) / T; ++c) {
      if (win[c * PA_C16_WINDOWS] < 0) continue;
      int32_t lo = INT32_MAX, hi = -1;
      for (int64_t p = crp[chunk_row[c]]; p < crp[chunk_row[c + 1]]; ++p) {
        lo = std::min(lo, col[p]);
        hi = std::max(hi, col[p]);
      }
      S.cmin[c] = lo; S.cmax[c] = hi;
      if (hi < 0 || hi - lo + 2 > max_cap - 2) continue;
      const int32_t l0 = lo >> 4;
      bits.assign((size_t)(((hi >> 4) - l0) >> 6) + 1, 0);
      for (int64_t p = crp[chunk_row[c]]; p < crp[chunk_row[c + 1]]; ++p) {
        const int32_t l = (col[p] >> 4) - l0;
        bits[l >> 6] |= 1ull << (l & 63);
      }
      int n = 0;
      for (uint64_t w : bits) n += __builtin_popcountll(w);
      S.lines[c] = n;
    }
  };
  if (n_threads <= 1) work(0, 1);
  else {
    std::vector<std::thread> th;
    for (int t = 1; t < n_threads; ++t) th.emplace_back(work, t, n_threads);
    work(0, n_threads);
    for (auto &x : th) x.join();
  }
}

// Groups of consecutive 16-bit chunks, not yet `taken`, whose columns span at most cap - 2 entries; the chunks
// of every group are marked taken.  Appends to `groups`; returns the entries of x these groups stage in total (the extra
// L2 traffic the window path pays) and, in *grouped_entries, the stored entries they hold.
inline int64_t pa_build_xw_groups(const int32_t *crp, const std::vector<int32_t> &chunk_row, const pa_xw_chunk_stats &S,
                                  int cap, int max_ratio_16ths, std::vector<char> &taken,
                                  std::vector<pa_xw_group> &groups, int64_t *grouped_entries) {
  const int64_t n_chunks = (int64_t)chunk_row.size() - 1;
  const std::vector<int32_t> &cmin = S.cmin, &lines = S.lines;
  std::vector<int32_t> cmax = S.cmax;
  for (int64_t c = 0; c < n_chunks; ++c)
    if (taken[c]) cmax[c] = -1;
  // a small block gets shorter groups, so that the launch still has about two rounds of workgroups per CU (2 M short rows,
  // 7268 chunks: 0.0402 ms with groups of 4, 0.0335 with groups of 8, 0.0381 with groups of 16)
  const int64_t maxg = std::max<int64_t>(PA_XW_MING, std::min<int64_t>(PA_XW_MAXG, n_chunks / PA_XW_WANT_GROUPS));
  int64_t staged = 0;
  *grouped_entries = 0;
  int64_t c = 0;
  while (c < n_chunks) {
    if (cmax[c] < 0) { ++c; continue; }
    int32_t lo = cmin[c], hi = cmax[c];
    int64_t e = c + 1;
    if (hi - lo + 2 <= cap - 2)
      while (e < n_chunks && e - c < maxg && cmax[e] >= 0) {
        const int32_t l2 = std::min(lo, cmin[e]), h2 = std::max(hi, cmax[e]);
        if (h2 - l2 + 2 > cap - 2) break;
        lo = l2; hi = h2; ++e;
      }
    // a group pays when the x it stages (8 B per column of the span) is at most max_ratio x the matrix bytes it streams
    // (10 B per stored entry); a shorter group over the same span would only be worse, so a failed group's chunks are all left
    const int64_t ent = (int64_t)crp[chunk_row[e]] - crp[chunk_row[c]];
    int64_t touched = 0;
    for (int64_t k = c; k < e; ++k) touched += lines[k];
    const bool scattered = max_ratio_16ths >= (1 << 20) || touched >= (int64_t)PA_XW_MIN_LINES * (e - c);
    if (hi - lo + 2 <= cap - 2 && e - c >= PA_XW_MING && scattered && (int64_t)(hi - lo + 1) * 8 * 16 <= ent * 10 * max_ratio_16ths) {
      groups.push_back(pa_xw_group{(int)c, (int)(e - c), lo, hi - lo + 1});
      for (int64_t k = c; k < e; ++k) taken[k] = 1;
      staged += hi - lo + 1;
      *grouped_entries += (int64_t)crp[chunk_row[e]] - crp[chunk_row[c]];
    }
    c = e;
  }
  return staged;
}

// Ring groups (k_spmv_xring): maximal runs of consecutive 16-bit chunks, not yet taken, in which every chunk's smallest
// column is less than PA_XR_CAP (minus a margin) below the highest column of its round and of all rounds before it -- a
// round being `sub` consecutive chunks counted from the run's first; conservatively, of the sub - 1 chunks after it too.
inline int64_t pa_build_xring_groups(const int32_t *crp, const std::vector<int32_t> &chunk_row, const pa_xw_chunk_stats &S, int sub,
                                     std::vector<char> &taken, std::vector<pa_xw_group> &groups, int64_t *grouped_entries,
                                     bool forced = false, const std::vector<int32_t> *breaks = nullptr) {
  const int64_t n_chunks = (int64_t)chunk_row.size() - 1;
  const int C = PA_XR_CAP - 64;
  static const int64_t want = getenv("PA_SPMV_XRING_GROUPS") ? std::max(1, atoi(getenv("PA_SPMV_XRING_GROUPS"))) : PA_XR_WANT_GROUPS;
  // breaks (a column piece of a chain that is to run in one launch, k_spmv_xring_chain): a run ends where the next chunk starts on
  // one of these rows and nowhere else -- every piece then has the same runs of rows.  A chunk without entries (the rows above the
  // band's first column in the lowest piece) is part of its run there: the kernel writes beta * y for its rows.
  const int64_t maxg = breaks ? PA_XR_MAXG : std::max<int64_t>(PA_XW_MING, std::min<int64_t>(PA_XR_MAXG, (n_chunks + want - 1) / want));
  auto is_break = [&](int32_t row) { return breaks && std::binary_search(breaks->begin(), breaks->end(), row); };
  auto usable = [&](int64_t k) {
    if (taken[k]) return false;
    if (S.cmax[k] >= 0) return true;
    return breaks != nullptr && crp[chunk_row[k + 1]] == crp[chunk_row[k]];
  };
  int64_t staged = 0;
  *grouped_entries = 0;
  int64_t c = 0;
  while (c < n_chunks) {
    if (!usable(c)) { ++c; continue; }
    int64_t e = c;
    int32_t runmax = -1, wlo = INT32_MAX;
    int64_t touched = 0;
    while (e < n_chunks && e - c < maxg && usable(e) && !(e > c && is_break(chunk_row[e]))) {
      const int32_t rm = std::max(runmax, S.cmax[e]);
      bool ok = true;
      for (int64_t j = std::max(c, e - (sub - 1)); j <= e && ok; ++j) ok = S.cmax[j] < 0 || (int64_t)rm - S.cmin[j] < C;
      if (!ok) break;
      runmax = rm;
      if (S.cmax[e] >= 0) wlo = std::min(wlo, S.cmin[e]);
      touched += S.lines[e];
      ++e;
    }
    if (runmax < 0) wlo = 0;                         // (a run of empty chunks: nothing to stage)
    // worth it when the chunks' gathers are scattered (as for the windows) and the run is long enough to pay its first fill:
    // the x it loads in all (8 B per column of its span) at most the matrix bytes it streams (10 B per stored entry)
    const int64_t ent = e > c ? (int64_t)crp[chunk_row[e]] - crp[chunk_row[c]] : 0;
    const bool take = breaks ? e > c
                             : e - c >= PA_XW_MING && (forced || (touched >= (int64_t)PA_XW_MIN_LINES * (e - c) && (int64_t)(runmax - wlo + 1) * 8 <= ent * 10));
    if (take) {
      groups.push_back(pa_xw_group{(int)c, (int)(e - c), wlo, runmax - wlo + 1});
      for (int64_t k = c; k < e; ++k) taken[k] = 1;
      staged += runmax - wlo + 1;
      *grouped_entries += (int64_t)crp[chunk_row[e]] - crp[chunk_row[c]];
      c = e;
    } else {
      c = std::max(e, c + 1);
    }
  }
  return staged;
}

// The tiers of one block: 40 KiB groups first, then 96 KiB and 128 KiB groups over what the smaller windows left (every
// group passes the staged-x test of pa_build_xw_groups unless `forced`).
// groups = [tier 0..., tier 1..., tier 2...].
struct pa_xw_plan {
  std::vector<pa_xw_group> groups;      // [40 KiB windows..., 96 KiB..., 128 KiB..., ring groups...]
  std::vector<int32_t> rest;
  int64_t n_tier[PA_XW_TIERS] = {0, 0, 0}, n_ring = 0, staged = 0, grouped = 0;
};
// (the per-chunk statistics S come from pa_xw_scan_chunks on the host or from the device kernel of pa_setup.hip: the same numbers)
// ring: 0 = windows only (the three tiers), 1 = the three tiers first, then ring groups over what they left (measured, 4 M rows
// x 16: the ring runs at 4.1-4.2 TB/s algorithmic whatever the band -- one workgroup of 8 waves per CU, two barriers per
// round -- against 5.1-5.9 on the 96 / 128 KiB windows and 7.4 on the 40 KiB ones where those fit, and against 3.05 on the
// row split at +-7900, where nothing else fits), 2 = ring groups only
inline void pa_plan_xw_from_stats(const int32_t *crp, const std::vector<int32_t> &chunk_row, const pa_xw_chunk_stats &S, bool forced,
                                  pa_xw_plan &P, int ring = 1, const std::vector<int32_t> *breaks = nullptr) {
  const int64_t n_chunks = (int64_t)chunk_row.size() - 1;
  P = pa_xw_plan();
  std::vector<char> taken(n_chunks, 0);
  const int caps[PA_XW_TIERS] = {PA_XW_CAP, PA_XW_CAP_MID, PA_XW_CAP_BIG};
  std::vector<pa_xw_group> ring_groups;
  for (int tier = 0; tier < PA_XW_TIERS; ++tier) {
    if (ring == 2) break;
    std::vector<char> t2 = taken;
    std::vector<pa_xw_group> g;
    int64_t grouped = 0;
    // 40 KiB windows compete with a row split that is not bad on such spans (5.9 TB/s within +-500): staged x <= 0.625 x
    // the matrix bytes.  Spans beyond that are where the row split's gathers go to L2 (3.1-3.6 TB/s): there even as much x
    // as matrix pays (+-7000, ratio 0.63: 0.172 ms against 0.269; +-8000 forced, ratio 2.1: 0.259 against 0.274).
    const int ratio16 = forced ? 1 << 20 : tier == 0 ? 10 : 16;
    const int64_t staged = pa_build_xw_groups(crp, chunk_row, S, caps[tier], ratio16, t2, g, &grouped);
    if (g.empty()) continue;
    // (a tier that would launch a handful of workgroups -- the clipped ends of a band the ring takes -- is not worth its
    // launch: 12 groups of the 128 KiB tier next to 1012 ring groups cost 0.212 ms instead of 0.189)
    if (ring >= 1 && !forced && tier >= 1 && (int64_t)g.size() < 64) continue;
    taken.swap(t2);
    P.groups.insert(P.groups.end(), g.begin(), g.end());
    P.n_tier[tier] = (int64_t)g.size();
    P.staged += staged;
    P.grouped += grouped;
  }
  if (ring >= 1) {
    int64_t grouped = 0;
    const int64_t staged = pa_build_xring_groups(crp, chunk_row, S, 2, taken, ring_groups, &grouped, forced, breaks);
    P.n_ring = (int64_t)ring_groups.size();
    P.staged += staged;
    P.grouped += grouped;
  }
  P.groups.insert(P.groups.end(), ring_groups.begin(), ring_groups.end());
  for (int64_t c = 0; c < n_chunks; ++c)
    if (!taken[c]) P.rest.push_back((int32_t)c);
}
inline void pa_plan_xw(const int32_t *crp, const int32_t *col, const std::vector<int32_t> &chunk_row, const int32_t *win,
                       bool forced, pa_xw_plan &P, int n_threads = 1, int ring = 1, std::vector<int32_t> *cmax_out = nullptr,
                       const std::vector<int32_t> *breaks = nullptr) {
  pa_xw_chunk_stats S;
  pa_xw_scan_chunks(crp, col, chunk_row, win, PA_XR_CAP, n_threads, S);       // (spans up to the ring's capacity are counted)
  pa_plan_xw_from_stats(crp, chunk_row, S, forced, P, ring, breaks);
  if (cmax_out) *cmax_out = S.cmax;
}
