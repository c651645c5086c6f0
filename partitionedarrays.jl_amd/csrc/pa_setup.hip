// pa_setup.hip -- device-side set-up of a block's column encodings.
//
// What csr_fill_slab (pa_csr.hip) used to do with host threads over the block's 1-based arrays -- row-pattern
// detection, the windowed 16-bit column stream, the compacted 32-bit stream (pa_encode_columns, pa_spmv_kernel.h) -- as
// kernels over the raw CSR already in HBM.  The result is the host encoder's, array for array: the same pattern table
// (ids in order of the first row that shows the pattern, at most 4096, patterns shared by fewer than 2 rows left out),
// the same descriptors, windows in first-seen order, the same compacted positions.  The host encoder stays as the
// checker (pa_host_check_spmv_encodings) and as the PA_SETUP_DEVICE=0 path the tests compare against.
//
//   rows -> 64-bit hash of (length, col - row id ...)                          one lane per row
//   (hash, row) radix-sorted (rocPRIM; stable: rows ascend inside a hash)      groups = runs of equal hash
//   groups of >= 2 rows ordered by their first row                              = the host's first-seen pattern ids
//   every row checked against its group's first row (a colliding hash keeps explicit columns)
//   chunks -> descriptors (runs of one pattern, constant row-id stride)        one lane per chunk
//   chunks without a descriptor -> window tags in first-seen order, 16-bit codes, or a copy of their 32-bit columns
//                                                                               one lane (codes) / one wavefront (copy) per chunk
// Reference loops these arrays serve: spmv_csr! src/sparse_utils.jl:649-669.
#include "pa_dev_util.h"   // hip_runtime, rocprim, the scratch / scan / sort helpers

#include "pa_setup.h"
#include "pa_spmv_kernel.h"

static constexpr uint64_t KEY_NONE = ~0ull;     // rows that can have no pattern (empty, or longer than PA_PAT_MAXLEN)

__global__ void ks_row_hash(const int *__restrict__ crp, const int *__restrict__ col, const int *__restrict__ row_ids, int n,
                            uint64_t *__restrict__ key, int *__restrict__ row) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int a = crp[r], e = crp[r + 1], len = e - a;
  uint64_t x = KEY_NONE;
  if (len >= 1 && len <= PA_PAT_MAXLEN) {
    const int id = row_ids ? row_ids[r] : r;
    x = 1469598103934665603ull ^ (uint64_t)len;                 // (pa_encode_patterns' hash: FNV-1a over the deltas)
    for (int p = a; p < e; ++p) { x ^= (uint64_t)(uint32_t)(col[p] - id); x *= 1099511628211ull; x ^= x >> 29; }
    if (x == KEY_NONE) x = 0;
  }
  key[r] = x;
  row[r] = r;
}

__global__ void ks_heads(const uint64_t *__restrict__ key, int n, int *__restrict__ head) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) head[i] = (i == 0 || key[i] != key[i - 1]) ? 1 : 0;
}

// gidx = inclusive scan of head - 1; gstart[g] = first sorted position of group g
__global__ void ks_group_starts(const int *__restrict__ head, const int *__restrict__ gscan, int n, int *__restrict__ gstart) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && head[i]) gstart[gscan[i] - 1] = i;
}

// a group may give a pattern when its rows can have one and there are at least min_rows of them
__global__ void ks_group_valid(const uint64_t *__restrict__ key, const int *__restrict__ gstart, int n_groups, int n, int min_rows,
                               int *__restrict__ valid) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups) return;
  const int a = gstart[g], e = g + 1 < n_groups ? gstart[g + 1] : n;
  valid[g] = (key[a] != KEY_NONE && e - a >= min_rows) ? 1 : 0;
}

// compact the valid groups: (first row, group) pairs, to be sorted by first row
__global__ void ks_group_compact(const int *__restrict__ valid, const int *__restrict__ vscan, const int *__restrict__ gstart,
                                 const int *__restrict__ srow, int n_groups, int *__restrict__ first_row, int *__restrict__ group) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n_groups || !valid[g]) return;
  const int k = vscan[g];                          // exclusive scan
  first_row[k] = srow[gstart[g]];                  // stable sort: the smallest row of the group
  group[k] = g;
}

__global__ void ks_fill_i32(int *__restrict__ p, int64_t n, int v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) p[i] = v;
}

__global__ void ks_assign_pattern(const int *__restrict__ group_sorted, int n_valid, int max_patterns, int *__restrict__ gpat) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n_valid && k < max_patterns) gpat[group_sorted[k]] = k;
}

// pattern table: the deltas of the pattern's first row
__global__ void ks_pdelta(const int *__restrict__ crp, const int *__restrict__ col, const int *__restrict__ row_ids,
                          const int *__restrict__ first_row_sorted, int n_patterns, int *__restrict__ pdelta) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int id = t / PA_PAT_MAXLEN, k = t % PA_PAT_MAXLEN;
  if (id >= n_patterns) return;
  const int r = first_row_sorted[id], a = crp[r], len = crp[r + 1] - a;
  pdelta[t] = k < len ? col[a + k] - (row_ids ? row_ids[r] : r) : 0;
}

// rowpat[row] = the row's pattern id, checked entry by entry against the table (a hash collision keeps explicit columns)
__global__ void ks_row_pattern(const int *__restrict__ crp, const int *__restrict__ col, const int *__restrict__ row_ids,
                               const int *__restrict__ srow, const int *__restrict__ gscan, const int *__restrict__ gpat,
                               const int *__restrict__ pdelta, const int *__restrict__ first_row_sorted, int n,
                               int *__restrict__ rowpat) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int r = srow[i];
  int pat = gpat[gscan[i] - 1];
  if (pat >= 0) {
    const int a = crp[r], len = crp[r + 1] - a, id = row_ids ? row_ids[r] : r;
    const int fr = first_row_sorted[pat];
    bool same = crp[fr + 1] - crp[fr] == len;
    for (int k = 0; k < len && same; ++k) same = pdelta[pat * PA_PAT_MAXLEN + k] == col[a + k] - id;
    if (!same) pat = -1;
  }
  rowpat[r] = pat;
}

// Per chunk: runs of equal pattern and constant row-id stride (phase C of pa_encode_patterns, statement for statement)
__global__ void ks_chunk_desc(const int *__restrict__ crp, const int *__restrict__ row_ids, const int *__restrict__ chunk_row,
                              const int *__restrict__ rowpat, int n_chunks, int cap, int *__restrict__ pdesc,
                              unsigned long long *__restrict__ n_good) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_chunks) return;
  int d[PA_PDESC_INTS];
#pragma unroll
  for (int k = 0; k < PA_PDESC_INTS; ++k) d[k] = 0;
  const int r0 = chunk_row[c], r1 = chunk_row[c + 1];
  const int p0 = crp[r0], p1 = crp[r1];
  bool ok = (p1 - (p0 & ~1)) <= cap && p1 > p0;
  int ns = 0, run_rows = 0, stride = 1, last_pat = -1;
  auto rid = [&](int r) { return row_ids ? row_ids[r] : r; };
  auto setd = [&](int k, int v) {                   // d[k] = v with a runtime k (kept in registers by the chains below)
#pragma unroll
    for (int j = 0; j < PA_PDESC_INTS; ++j) if (j == k) d[j] = v;
  };
  auto getd = [&](int k) {
    int v = 0;
#pragma unroll
    for (int j = 0; j < PA_PDESC_INTS; ++j) if (j == k) v = d[j];
    return v;
  };
  for (int r = r0; ok && r < r1; ++r) {
    const int rp = rowpat[r];
    if (rp < 0) { ok = false; break; }
    bool extend = ns > 0 && rp == last_pat;
    if (extend) {
      const int step = rid(r) - rid(r - 1);
      if (run_rows == 1) {
        if (step < 1 || step >= (1 << 20)) extend = false;
        else stride = step;
      } else if (step != stride) {
        extend = false;
      }
    }
    if (extend) {
      ++run_rows;
      if (row_ids) setd(8 + ns - 1, (getd(8 + ns - 1) & 255) | (stride << 8));
      continue;
    }
    if (ns == PA_PAT_SEGMENTS) { ok = false; break; }
    if (ns) setd(ns, crp[r] - p0);
    setd(4 + ns, rid(r));
    setd(8 + ns, (crp[r + 1] - crp[r]) | (row_ids ? 1 << 8 : 0));
    setd(12 + ns, rp);
    last_pat = rp;
    ++ns;
    run_rows = 1;
    stride = 1;
  }
  int *out = pdesc + (size_t)c * PA_PDESC_INTS;
  if (!ok) {
#pragma unroll
    for (int k = 0; k < PA_PDESC_INTS; ++k) out[k] = 0;
    return;
  }
  for (int s = ns; s < PA_PAT_SEGMENTS; ++s) {
    if (s) setd(s, 1 << 30);
    setd(4 + s, 0);
    setd(8 + s, 1 | (row_ids ? 1 << 8 : 0));
    setd(12 + s, 0);
  }
#pragma unroll
  for (int s = 0; s < PA_PAT_SEGMENTS; ++s) {
    const unsigned L = (unsigned)(d[8 + s] & 255);
    d[16 + s] = (int)(L > 1 ? 0xFFFFFFFFu / L + 1u : 0u);
  }
  d[0] = ns;
#pragma unroll
  for (int k = 0; k < PA_PDESC_INTS; ++k) out[k] = d[k];
  atomicAdd(n_good, 1ull);
}

// slots a chunk takes in a compacted stream (its entries from the 2-aligned start, plus the pair the clamped loads may
// touch), for the chunks the stream keeps; 0 for the others.  which = 16: no descriptor and not a long row; which = 32: no
// descriptor and no 16-bit windows (win == NULL: every chunk without a descriptor).
__global__ void ks_chunk_span(const int *__restrict__ crp, const int *__restrict__ chunk_row, const int *__restrict__ pdesc,
                              const int *__restrict__ win, int n_chunks, int cap, int which, long long *__restrict__ span) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_chunks) return;
  const long long p0 = crp[chunk_row[c]], p1 = crp[chunk_row[c + 1]];
  const long long start = p0 & ~1ll;
  const long long sp = ((p1 - start + 1) & ~1ll) + 2;
  bool keep = pdesc[(size_t)c * PA_PDESC_INTS] <= 0;
  if (which == 16) keep = keep && (p1 - start <= cap);
  else keep = keep && !(win && win[(size_t)c * PA_C16_WINDOWS] >= 0);
  span[c] = keep ? sp : 0;
}

// 16-bit codes of a chunk, window tags in first-seen order (pa_encode_col16).  pos (or NULL): slot of the chunk's
// 2-aligned start in the compacted stream; keep[c] == 0: the chunk gets no 16-bit columns (win[c*16] = -1, not a fallback).
__global__ void ks_encode_c16(const int *__restrict__ crp, const int *__restrict__ col, const int *__restrict__ chunk_row,
                              int n_chunks, int cap, const long long *__restrict__ pos, const long long *__restrict__ keep,
                              unsigned short *__restrict__ c16, int *__restrict__ win) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_chunks) return;
  int *w = win + (size_t)c * PA_C16_WINDOWS;
#pragma unroll
  for (int s = 0; s < PA_C16_WINDOWS; ++s) w[s] = 0;
  const long long p0 = crp[chunk_row[c]], p1 = crp[chunk_row[c + 1]];
  if (keep && keep[c] == 0) { w[0] = -1; return; }
  const long long shift = pos ? pos[c] - (p0 & ~1ll) : 0;
  bool ok = (p1 - (p0 & ~1ll)) <= cap;
  int n = 0;
  int tags[PA_C16_WINDOWS];
  for (long long p = p0; ok && p < p1; ++p) {
    const int cj = col[p], tag = cj >> 12;
    int s = 0;
#pragma unroll
    for (int j = 0; j < PA_C16_WINDOWS; ++j) if (j < n && tags[j] != tag && s == j) s = j + 1;   // first slot holding the tag
    if (s == n) {
      if (n == PA_C16_WINDOWS) { ok = false; break; }
#pragma unroll
      for (int j = 0; j < PA_C16_WINDOWS; ++j) if (j == n) tags[j] = tag;
      ++n;
    }
    c16[p + shift] = (unsigned short)((s << 12) | (cj & 4095));
  }
  if (ok) {
#pragma unroll
    for (int s = 0; s < PA_C16_WINDOWS; ++s) if (s < n) w[s] = tags[s] << 12;
  } else {
    w[0] = -1;
  }
}

// compacted 16-bit stream: the descriptor slot of a chunk that got windows says where its codes sit
__global__ void ks_mark_c16(const int *__restrict__ crp, const int *__restrict__ chunk_row, const long long *__restrict__ span,
                            const long long *__restrict__ pos, const int *__restrict__ win, int n_chunks, int *__restrict__ pdesc) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_chunks || span[c] == 0 || win[(size_t)c * PA_C16_WINDOWS] < 0) return;
  pdesc[(size_t)c * PA_PDESC_INTS + 1] = (int)(pos[c] - ((long long)crp[chunk_row[c]] & ~1ll));
}

// compacted 32-bit stream: one wavefront copies a chunk's columns (a long row is a chunk of any length)
__global__ void ks_copy_c32(const int *__restrict__ crp, const int *__restrict__ col, const int *__restrict__ chunk_row,
                            const long long *__restrict__ span, const long long *__restrict__ pos, int n_chunks,
                            int *__restrict__ c32, int *__restrict__ pdesc) {
  const int c = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (c >= n_chunks || span[c] == 0) return;
  const long long b = (long long)crp[chunk_row[c]] & ~1ll, p1 = crp[chunk_row[c + 1]];
  for (long long k = lane; k < p1 - b; k += 64) c32[pos[c] + k] = col[b + k];
  if (lane == 0) pdesc[(size_t)c * PA_PDESC_INTS + 2] = (int)(pos[c] - b);
}

// chunks and stored entries by the column encoding the kernel will read: cnt = {n_c16, n_c32, nnz_c16, nnz_c32}
__global__ void ks_count(const int *__restrict__ crp, const int *__restrict__ chunk_row, const int *__restrict__ pdesc,
                         const int *__restrict__ win, int n_chunks, int cap, unsigned long long *__restrict__ cnt) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_chunks) return;
  if (pdesc && pdesc[(size_t)c * PA_PDESC_INTS] > 0) return;
  const long long p0 = crp[chunk_row[c]], ne = (long long)crp[chunk_row[c + 1]] - p0;
  const bool w16 = win && win[(size_t)c * PA_C16_WINDOWS] >= 0;
  atomicAdd(&cnt[w16 ? 0 : 1], 1ull);
  atomicAdd(&cnt[(w16 && ne + (p0 & 1) <= cap) ? 2 : 3], (unsigned long long)ne);
}

// rows holding a multiple of 8 entries / non-empty rows (the padded product slots of blocks without row patterns)
__global__ void ks_mult8(const int *__restrict__ crp, int n, unsigned long long *__restrict__ cnt) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  const int len = crp[r + 1] - crp[r];
  if (len > 0) {
    atomicAdd(&cnt[0], 1ull);
    if ((len & 7) == 0) atomicAdd(&cnt[1], 1ull);
  }
}

// ---- the row split on the device (pa_build_chunks, pa_spmv_kernel.h) -------------------------------------------------------
// The host loop is greedy and sequential: a chunk starts where the previous one ended.  In parallel: nxt[r] = the end of the
// chunk that WOULD start at row r (independent per row: a binary search in the row pointers, the max-rows cap, the 64-byte
// alignment rule), and the chunk starts are the orbit of row 0 under nxt.  The orbit is marked by pointer doubling: with
// jump = nxt^(2^k), "every marked row marks jump[row]", then jump := jump o jump; after ceil(log2(rows)) + 1 rounds every
// element of the orbit is marked (each index of the orbit is a sum of distinct powers of two) and nothing else is.
__global__ void ks_chunk_next(const int *__restrict__ crp, int nc, int cap, int max_rows, int align_rows, int *__restrict__ nxt) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > nc) return;
  if (r == nc) { nxt[r] = nc; return; }
  const long long base = crp[r] & ~1;
  int e = r + 1;
  if ((long long)crp[e] - base <= cap) {
    int lo = r + 1, hi = min(nc, r + max_rows);               // largest e in [lo, hi] with crp[e] - base <= cap
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if ((long long)crp[mid] - base <= cap) lo = mid; else hi = mid - 1;
    }
    e = lo;
    if (align_rows > 1 && e < nc) {
      const int ea = e - (e % align_rows);
      if (ea > r && (long long)(ea - r) * 4 >= (long long)(e - r) * 3) e = ea;
    }
  }
  nxt[r] = e;
}
__global__ void ks_orbit_mark(const int *__restrict__ jump, int n, int *__restrict__ mark) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n && mark[r]) mark[jump[r]] = 1;                    // (in place: a row marked in this very round is on the orbit too)
}
__global__ void ks_jump_square(const int *__restrict__ in, int n, int *__restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) out[r] = in[in[r]];
}
__global__ void ks_chunk_rows(const int *__restrict__ mark, const int *__restrict__ mscan, const int *__restrict__ crp, int n, int nc,
                              int cap, int *__restrict__ chunk_row, unsigned long long *__restrict__ n_long) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n || !mark[r]) return;
  chunk_row[mscan[r]] = r;                                     // exclusive scan
  if (r < nc && (long long)crp[r + 1] - (crp[r] & ~1) > cap) atomicAdd(n_long, 1ull);
}

// Per chunk on the 16-bit stream: first and last column and the number of distinct 128-byte lines of x (16 entries) its
// gathers touch -- pa_xw_scan_chunks (pa_spmv_xwin.h) on the device; the line set is a bitmap in registers (a chunk whose
// span exceeds the largest window is not counted: it cannot join a group anyway).
__global__ void ks_xw_stats(const int *__restrict__ crp, const int *__restrict__ col, const int *__restrict__ chunk_row,
                            const int *__restrict__ win, int n_chunks, int max_cap, int *__restrict__ cmin, int *__restrict__ cmax,
                            int *__restrict__ lines) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n_chunks) return;
  int lo = 0x7fffffff, hi = -1, n = 0;
  if (win[(size_t)c * PA_C16_WINDOWS] >= 0) {
    const int p0 = crp[chunk_row[c]], p1 = crp[chunk_row[c + 1]];
    for (int p = p0; p < p1; ++p) { const int j = col[p]; lo = min(lo, j); hi = max(hi, j); }
    if (hi >= 0 && hi - lo + 2 <= max_cap - 2) {
      constexpr int W = 17;                                    // 17 x 64 lines x 16 entries >= the largest window + one line
      unsigned long long bits[W];
#pragma unroll
      for (int k = 0; k < W; ++k) bits[k] = 0ull;
      const int l0 = lo >> 4;
      for (int p = p0; p < p1; ++p) {
        const int l = (col[p] >> 4) - l0, w = l >> 6;
        const unsigned long long m = 1ull << (l & 63);
#pragma unroll
        for (int k = 0; k < W; ++k) if (k == w) bits[k] |= m;
      }
#pragma unroll
      for (int k = 0; k < W; ++k) n += __popcll(bits[k]);
    }
  }
  cmin[c] = lo; cmax[c] = hi; lines[c] = n;
}

__global__ void ks_minmax(const int *__restrict__ p, int64_t n, int *__restrict__ out) {
  int mn = 0x7fffffff, mx = (int)0x80000000;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) { const int v = p[i]; mn = min(mn, v); mx = max(mx, v); }
  for (int off = 32; off > 0; off >>= 1) { mn = min(mn, __shfl_down(mn, off, 64)); mx = max(mx, __shfl_down(mx, off, 64)); }
  if ((threadIdx.x & 63) == 0) { atomicMin(&out[0], mn); atomicMax(&out[1], mx); }
}

// ---- host side -----------------------------------------------------------------------------------------------------------
namespace {
using namespace pa_util;

// Row patterns of the block: pdesc (n_chunks x PA_PDESC_INTS, zeroed for chunks without a descriptor) and pdelta in S;
// returns in *n_good the number of chunks that got a descriptor (0: no table worth having; S holds a one-pattern dummy
// table as the host encoder leaves one).
int encode_patterns(pa_ctx *c, const int32_t *d_crp, const int32_t *d_col, const int32_t *d_row_ids, int64_t nc,
                    const int32_t *d_chunk_row, int64_t n_chunks, int cap, pa_dev_streams &S, int64_t *n_good,
                    int max_patterns = 4096, int min_rows = 2) {
  hipStream_t s = c->s[0];
  scratch sc;
  *n_good = 0;
  const int n = (int)nc;
  PA_TRY(pa_dev_alloc(c, (void **)&S.d_pdesc, sizeof(int32_t) * (size_t)n_chunks * PA_PDESC_INTS, PA_MEM_MATRIX));
  PA_HIP(hipMemsetAsync(S.d_pdesc, 0, sizeof(int32_t) * (size_t)n_chunks * PA_PDESC_INTS, s));
  auto dummy_table = [&]() -> int {
    S.n_pdelta = PA_PAT_MAXLEN;
    PA_TRY(pa_dev_alloc(c, (void **)&S.d_pdelta, sizeof(int32_t) * PA_PAT_MAXLEN, PA_MEM_MATRIX));
    PA_HIP(hipMemsetAsync(S.d_pdelta, 0, sizeof(int32_t) * PA_PAT_MAXLEN, s));
    return PA_OK;
  };
  uint64_t *key = nullptr, *skey = nullptr;
  int *row = nullptr, *srow = nullptr;
  PA_TRY(sc.get(&key, nc));
  PA_TRY(sc.get(&row, nc));
  hipLaunchKernelGGL(ks_row_hash, grid1(nc), dim3(256), 0, s, d_crp, d_col, d_row_ids, n, key, row);
  // (a block of unstructured rows has as many delta lists as rows: a sample of 65536 evenly spaced rows tells, as in
  // pa_encode_patterns -- more than half of them distinct means no descriptor would cover half of the chunks)
  if (nc >= (1 << 18)) {
    const int64_t ns = 1 << 16, step = nc / ns;
    std::vector<uint64_t> sample(ns);
    PA_HIP(hipMemcpy2DAsync(sample.data(), sizeof(uint64_t), key, sizeof(uint64_t) * (size_t)step, sizeof(uint64_t), ns, hipMemcpyDeviceToHost, s));
    PA_HIP(hipStreamSynchronize(s));
    std::sort(sample.begin(), sample.end());
    const int64_t distinct = std::unique(sample.begin(), sample.end()) - sample.begin();
    if (distinct * 2 > ns) return dummy_table();
  }
  PA_TRY(sc.get(&skey, nc));
  PA_TRY(sc.get(&srow, nc));
  PA_TRY(sort_pairs<uint64_t>(sc, s, key, skey, row, srow, (size_t)nc));
  int *head = nullptr, *gscan = nullptr, *gstart = nullptr;
  PA_TRY(sc.get(&head, nc));
  PA_TRY(sc.get(&gscan, nc));
  hipLaunchKernelGGL(ks_heads, grid1(nc), dim3(256), 0, s, skey, n, head);
  PA_TRY(scan_inclusive(sc, s, head, gscan, (size_t)nc));
  int n_groups = 0;
  PA_TRY(d2h(s, &n_groups, gscan + (nc - 1), 1));
  PA_TRY(sc.get(&gstart, (size_t)n_groups + 1));
  hipLaunchKernelGGL(ks_group_starts, grid1(nc), dim3(256), 0, s, head, gscan, n, gstart);
  int *valid = nullptr, *vscan = nullptr, *first_row = nullptr, *group = nullptr, *first_row_s = nullptr, *group_s = nullptr, *gpat = nullptr;
  PA_TRY(sc.get(&valid, (size_t)n_groups + 1));
  PA_TRY(sc.get(&vscan, (size_t)n_groups + 1));
  PA_HIP(hipMemsetAsync(valid + n_groups, 0, sizeof(int), s));
  hipLaunchKernelGGL(ks_group_valid, grid1(n_groups), dim3(256), 0, s, skey, gstart, n_groups, n, min_rows, valid);
  PA_TRY(scan_exclusive<int>(sc, s, valid, vscan, (size_t)n_groups + 1));
  int n_valid = 0;
  PA_TRY(d2h(s, &n_valid, vscan + n_groups, 1));
  if (n_valid == 0) return dummy_table();
  PA_TRY(sc.get(&first_row, n_valid));
  PA_TRY(sc.get(&group, n_valid));
  PA_TRY(sc.get(&first_row_s, n_valid));
  PA_TRY(sc.get(&group_s, n_valid));
  PA_TRY(sc.get(&gpat, n_groups));
  hipLaunchKernelGGL(ks_group_compact, grid1(n_groups), dim3(256), 0, s, valid, vscan, gstart, srow, n_groups, first_row, group);
  PA_TRY(sort_pairs<int>(sc, s, first_row, first_row_s, group, group_s, (size_t)n_valid));      // (row ids are >= 0: unsigned order = order)
  hipLaunchKernelGGL(ks_fill_i32, grid1(n_groups), dim3(256), 0, s, gpat, (int64_t)n_groups, -1);
  hipLaunchKernelGGL(ks_assign_pattern, grid1(n_valid), dim3(256), 0, s, group_s, n_valid, max_patterns, gpat);
  const int n_patterns = std::min(n_valid, max_patterns);
  S.n_pdelta = (int64_t)n_patterns * PA_PAT_MAXLEN;
  PA_TRY(pa_dev_alloc(c, (void **)&S.d_pdelta, sizeof(int32_t) * (size_t)S.n_pdelta, PA_MEM_MATRIX));
  hipLaunchKernelGGL(ks_pdelta, grid1(S.n_pdelta), dim3(256), 0, s, d_crp, d_col, d_row_ids, first_row_s, n_patterns, S.d_pdelta);
  int *rowpat = nullptr;
  PA_TRY(sc.get(&rowpat, nc));
  hipLaunchKernelGGL(ks_row_pattern, grid1(nc), dim3(256), 0, s, d_crp, d_col, d_row_ids, srow, gscan, gpat, S.d_pdelta, first_row_s, n, rowpat);
  unsigned long long *good = nullptr;
  PA_TRY(sc.get(&good, 1));
  PA_HIP(hipMemsetAsync(good, 0, sizeof(unsigned long long), s));
  hipLaunchKernelGGL(ks_chunk_desc, grid1(n_chunks, 128), dim3(128), 0, s, d_crp, d_row_ids, d_chunk_row, rowpat, (int)n_chunks, cap, S.d_pdesc, good);
  PA_HIP(hipGetLastError());
  unsigned long long g = 0;
  PA_TRY(d2h(s, &g, good, 1));
  *n_good = (int64_t)g;
  return PA_OK;
}

int encode_impl(pa_ctx *c, const int32_t *d_crp, const int32_t *d_col, const int32_t *d_row_ids, int64_t nc, int64_t nnz,
                const int32_t *d_chunk_row, int64_t n_chunks, int cap, bool want_pattern, bool want_c16,
                bool compact_streams, pa_dev_streams &S) {
  hipStream_t s = c->s[0];
  const size_t pad = 8;
  if (nnz == 0 || n_chunks == 0) { S.n_c32 = n_chunks; return PA_OK; }
  scratch sc;
  if (want_pattern) {
    PA_TRY(encode_patterns(c, d_crp, d_col, d_row_ids, nc, d_chunk_row, n_chunks, cap, S, &S.n_pattern));
    S.use_pattern = S.n_pattern > 0 && S.n_pattern * 2 >= n_chunks;    // worth it only when it covers most of the block
    if (!S.use_pattern) {
      S.n_pattern = 0;
      pa_dev_free(c, S.d_pdesc); S.d_pdesc = nullptr;
      pa_dev_free(c, S.d_pdelta); S.d_pdelta = nullptr;
      S.n_pdelta = 0;
    }
  }
  if (S.use_pattern && want_c16 && (n_chunks - S.n_pattern) * 64 < n_chunks) want_c16 = false;   // (see pa_encode_columns)
  S.use_c16 = want_c16;
  if (want_c16) PA_TRY(pa_dev_alloc(c, (void **)&S.d_win, sizeof(int32_t) * (size_t)n_chunks * PA_C16_WINDOWS, PA_MEM_MATRIX));
  unsigned long long *cnt = nullptr;
  PA_TRY(sc.get(&cnt, 4));
  PA_HIP(hipMemsetAsync(cnt, 0, 4 * sizeof(unsigned long long), s));
  if (!S.use_pattern || !compact_streams) {
    // full-length streams (with descriptors only for measurements: their column slots stay 0 = no shift)
    if (want_c16) {
      S.n_c16_slots = nnz + (int64_t)pad;
      PA_TRY(pa_dev_alloc(c, (void **)&S.d_c16, sizeof(uint16_t) * (size_t)S.n_c16_slots, PA_MEM_MATRIX));
      PA_HIP(hipMemsetAsync(S.d_c16, 0, sizeof(uint16_t) * (size_t)S.n_c16_slots, s));
      hipLaunchKernelGGL(ks_encode_c16, grid1(n_chunks, 64), dim3(64), 0, s, d_crp, d_col, d_chunk_row, (int)n_chunks, cap,
                         (const long long *)nullptr, (const long long *)nullptr, S.d_c16, S.d_win);
    }
  } else {
    S.full = false;
    long long *span = nullptr, *pos = nullptr;
    PA_TRY(sc.get(&span, (size_t)n_chunks + 1));
    PA_TRY(sc.get(&pos, (size_t)n_chunks + 1));
    if (want_c16) {
      PA_HIP(hipMemsetAsync(span + n_chunks, 0, sizeof(long long), s));
      hipLaunchKernelGGL(ks_chunk_span, grid1(n_chunks), dim3(256), 0, s, d_crp, d_chunk_row, S.d_pdesc, (const int *)nullptr, (int)n_chunks, cap, 16, span);
      PA_TRY(scan_exclusive<long long>(sc, s, span, pos, (size_t)n_chunks + 1));
      long long n16 = 0;
      PA_TRY(d2h(s, &n16, pos + n_chunks, 1));
      S.n_c16_slots = (int64_t)n16 + (int64_t)pad;
      PA_TRY(pa_dev_alloc(c, (void **)&S.d_c16, sizeof(uint16_t) * (size_t)S.n_c16_slots, PA_MEM_MATRIX));
      PA_HIP(hipMemsetAsync(S.d_c16, 0, sizeof(uint16_t) * (size_t)S.n_c16_slots, s));
      hipLaunchKernelGGL(ks_encode_c16, grid1(n_chunks, 64), dim3(64), 0, s, d_crp, d_col, d_chunk_row, (int)n_chunks, cap, pos, span, S.d_c16, S.d_win);
      hipLaunchKernelGGL(ks_mark_c16, grid1(n_chunks), dim3(256), 0, s, d_crp, d_chunk_row, span, pos, S.d_win, (int)n_chunks, S.d_pdesc);
    }
    long long *span32 = nullptr, *pos32 = nullptr;
    PA_TRY(sc.get(&span32, (size_t)n_chunks + 1));
    PA_TRY(sc.get(&pos32, (size_t)n_chunks + 1));
    PA_HIP(hipMemsetAsync(span32 + n_chunks, 0, sizeof(long long), s));
    hipLaunchKernelGGL(ks_chunk_span, grid1(n_chunks), dim3(256), 0, s, d_crp, d_chunk_row, S.d_pdesc, want_c16 ? S.d_win : (const int *)nullptr,
                       (int)n_chunks, cap, 32, span32);
    PA_TRY(scan_exclusive<long long>(sc, s, span32, pos32, (size_t)n_chunks + 1));
    long long n32 = 0;
    PA_TRY(d2h(s, &n32, pos32 + n_chunks, 1));
    S.n_c32_slots = (int64_t)n32;
    PA_TRY(pa_dev_alloc(c, (void **)&S.d_c32, sizeof(int32_t) * (size_t)(n32 + (long long)pad), PA_MEM_MATRIX));
    PA_HIP(hipMemsetAsync(S.d_c32, 0, sizeof(int32_t) * (size_t)(n32 + (long long)pad), s));
    hipLaunchKernelGGL(ks_copy_c32, grid1(n_chunks * 64, 256), dim3(256), 0, s, d_crp, d_col, d_chunk_row, span32, pos32, (int)n_chunks, S.d_c32, S.d_pdesc);
  }
  hipLaunchKernelGGL(ks_count, grid1(n_chunks), dim3(256), 0, s, d_crp, d_chunk_row, S.use_pattern ? S.d_pdesc : (const int *)nullptr,
                     want_c16 ? S.d_win : (const int *)nullptr, (int)n_chunks, cap, cnt);
  PA_HIP(hipGetLastError());
  unsigned long long h[4];
  PA_TRY(d2h(s, h, cnt, 4));
  S.n_c16 = (int64_t)h[0]; S.n_c32 = (int64_t)h[1]; S.nnz_c16 = (int64_t)h[2]; S.nnz_c32 = (int64_t)h[3];
  if (!S.use_pattern && nc > 0) {                          // (see PADP in pa_spmv_kernel.h)
    PA_HIP(hipMemsetAsync(cnt, 0, 2 * sizeof(unsigned long long), s));
    hipLaunchKernelGGL(ks_mult8, grid1(nc), dim3(256), 0, s, d_crp, (int)nc, cnt);
    PA_TRY(d2h(s, h, cnt, 2));
    S.pad_products = h[1] * 2 > h[0];
  }
  return PA_OK;
}

}  // namespace

int pa_dev_row_split(pa_ctx *c, const int32_t *d_crp, int64_t nc, int cap, int max_rows, int align_rows,
                     std::vector<int32_t> &chunk_row, int64_t *n_long) {
  using namespace pa_util;
  hipStream_t s = c->s[0];
  scratch sc;
  const int n = (int)nc + 1;
  int *jump = nullptr, *jump2 = nullptr, *mark = nullptr, *mscan = nullptr, *out = nullptr;
  unsigned long long *nl = nullptr;
  PA_TRY(sc.get(&jump, n));
  PA_TRY(sc.get(&jump2, n));
  PA_TRY(sc.get(&mark, (size_t)n + 1));
  PA_TRY(sc.get(&mscan, (size_t)n + 1));
  PA_TRY(sc.get(&nl, 1));
  PA_HIP(hipMemsetAsync(mark, 0, sizeof(int) * ((size_t)n + 1), s));
  PA_HIP(hipMemsetAsync(nl, 0, sizeof(unsigned long long), s));
  const int one = 1;
  PA_HIP(hipMemcpyAsync(mark, &one, sizeof(int), hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(ks_chunk_next, grid1(n), dim3(256), 0, s, d_crp, (int)nc, cap, max_rows, align_rows, jump);
  int rounds = 1;
  while (((int64_t)1 << rounds) < (int64_t)n) ++rounds;
  for (int k = 0; k <= rounds; ++k) {
    hipLaunchKernelGGL(ks_orbit_mark, grid1(n), dim3(256), 0, s, jump, n, mark);
    hipLaunchKernelGGL(ks_jump_square, grid1(n), dim3(256), 0, s, jump, n, jump2);
    std::swap(jump, jump2);
  }
  PA_HIP(hipGetLastError());
  PA_TRY(scan_exclusive<int>(sc, s, mark, mscan, (size_t)n + 1));
  int count = 0;
  PA_TRY(d2h(s, &count, mscan + n, 1));
  PA_REQUIRE(count >= 1, "the device row split lost its first row");
  PA_TRY(sc.get(&out, count));
  hipLaunchKernelGGL(ks_chunk_rows, grid1(n), dim3(256), 0, s, mark, mscan, d_crp, n, (int)nc, cap, out, nl);
  PA_HIP(hipGetLastError());
  chunk_row.resize(count);
  PA_TRY(d2h(s, chunk_row.data(), out, (size_t)count));
  unsigned long long h = 0;
  PA_TRY(d2h(s, &h, nl, 1));
  *n_long = (int64_t)h;
  PA_REQUIRE(chunk_row.front() == 0 && chunk_row.back() == (int32_t)nc, "the device row split does not span the rows");
  return PA_OK;
}

int pa_dev_xw_chunk_stats(pa_ctx *c, const int32_t *d_crp, const int32_t *d_col, const int32_t *d_chunk_row, const int32_t *d_win,
                          int64_t n_chunks, int max_cap, int32_t *cmin, int32_t *cmax, int32_t *lines) {
  using namespace pa_util;
  scratch sc;
  int *a = nullptr, *b = nullptr, *l = nullptr;
  PA_TRY(sc.get(&a, n_chunks));
  PA_TRY(sc.get(&b, n_chunks));
  PA_TRY(sc.get(&l, n_chunks));
  hipLaunchKernelGGL(ks_xw_stats, grid1(n_chunks, 64), dim3(64), 0, c->s[0], d_crp, d_col, d_chunk_row, d_win, (int)n_chunks, max_cap, a, b, l);
  PA_HIP(hipGetLastError());
  PA_TRY(d2h(c->s[0], cmin, a, (size_t)n_chunks));
  PA_TRY(d2h(c->s[0], cmax, b, (size_t)n_chunks));
  PA_TRY(d2h(c->s[0], lines, l, (size_t)n_chunks));
  return PA_OK;
}

void pa_dev_streams_free(pa_ctx *c, pa_dev_streams &S) {
  pa_dev_free(c, S.d_pdesc); pa_dev_free(c, S.d_pdelta); pa_dev_free(c, S.d_win); pa_dev_free(c, S.d_c16); pa_dev_free(c, S.d_c32);
  S = pa_dev_streams();
}

int pa_dev_encode_columns(pa_ctx *c, const int32_t *d_crp, const int32_t *d_col, const int32_t *d_row_ids, int64_t nc,
                          int64_t nnz, const int32_t *d_chunk_row, int64_t n_chunks, int cap, bool want_pattern,
                          bool want_c16, bool compact_streams, pa_dev_streams &S) {
  S = pa_dev_streams();
  hipEvent_t e0 = nullptr, e1 = nullptr;
  PA_HIP(hipEventCreate(&e0));
  PA_HIP(hipEventCreate(&e1));
  (void)hipEventRecord(e0, c->s[0]);
  int st = encode_impl(c, d_crp, d_col, d_row_ids, nc, nnz, d_chunk_row, n_chunks, cap, want_pattern, want_c16, compact_streams, S);
  if (st == PA_OK && hipStreamSynchronize(c->s[0]) != hipSuccess) { pa_set_err("device-side encoding failed: %s", hipGetErrorString(hipGetLastError())); st = PA_ERR_HIP; }
  if (st == PA_OK) {
    (void)hipEventRecord(e1, c->s[0]);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    S.ms = ms;
  } else {
    (void)hipGetLastError();
    pa_dev_streams_free(c, S);
  }
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  return st;
}

int pa_dev_minmax_i32(pa_ctx *c, const int32_t *d, int64_t n, int32_t *mn, int32_t *mx) {
  int *out = nullptr;
  PA_HIP(hipMalloc((void **)&out, 2 * sizeof(int)));
  const int init[2] = {0x7fffffff, (int)0x80000000};
  hipError_t e = hipMemcpyAsync(out, init, sizeof init, hipMemcpyHostToDevice, c->s[0]);
  if (e == hipSuccess && n > 0) hipLaunchKernelGGL(ks_minmax, dim3((unsigned)std::min<int64_t>(4096, (n + 255) / 256)), dim3(256), 0, c->s[0], d, n, out);
  int h[2] = {0, 0};
  if (e == hipSuccess) e = hipMemcpyAsync(h, out, sizeof h, hipMemcpyDeviceToHost, c->s[0]);
  if (e == hipSuccess) e = hipStreamSynchronize(c->s[0]);
  (void)hipFree(out);
  PA_HIP(e);
  *mn = h[0]; *mx = h[1];
  return PA_OK;
}
