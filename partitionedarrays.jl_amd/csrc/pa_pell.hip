// pa_pell.hip -- pattern-ELL storage of a CSR block (pa_pell.h: the kernel and why), built in HBM from the block's own arrays.
//
// Reference: spmv_csr! src/sparse_utils.jl:649-669; mul!(y,A,x,alpha,beta) as called at src/p_sparse_matrix.jl:2088.
//
// A block qualifies when every slab of 64 consecutive stored rows has at most 32 distinct (column - row id) deltas and the padding
// the slab-wide union costs stays small (<= 25 % more slots than stored entries): stencil and structured-FEM operators, their
// row-compacted colour blocks and restriction subsets.  Anything else stays on the row-split kernel; nothing here can fail a block's
// creation (no room, too irregular: the block simply has no pattern-ELL storage).
//   build   : decode the columns (pa_dev_decode_entries) -> kp_union: per slab the ascending union of deltas by repeated wave-minimum
//             (the step at which a lane's next entry IS the minimum is that entry's bit in the row mask) -> host: the distinct unions
//             (a few dozen) become the pattern table, slab widths padded to the unroll -> kp_verify (a hash collision would be caught
//             here) -> kp_fill: values delta-major per slab, or one bit per entry when the block's dictionary has at most two values.
//   updates : pa_csr_update_values* / psparse! write d_val; pa_pell_after_update re-runs kp_fill behind them on the compute stream
//             (same pattern, same addresses: recorded graphs stay valid).
#include "pa_dev_util.h"

#include "pa_setup.h"
#include "pa_pell.h"
#include "pa_pell_launch.h"

#include <chrono>
#include <map>
#include <tuple>

using namespace pa_util;

__device__ __forceinline__ int kp_wave_min(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, __shfl_xor(v, o, 64));
  return v;
}

// one wavefront per slab: D = ascending union of the rows' deltas (repeated wave minimum), mask[r] bit k = row r has delta D[k];
// out_lhash: a hash of WHICH LANES have each delta (the ballots, in order) and of the stride of the row ids -- what makes two slabs
// of one union members of the same CLASS; out_stride: the rows are row0 + stride * lane (1 or 2), 0: they are not
__global__ __launch_bounds__(256) void kp_union(const int *__restrict__ crp, const int *__restrict__ col, const int *__restrict__ row_ids,
                                                int n_crows, int n_slabs, int *__restrict__ out_len, int *__restrict__ out_D,
                                                unsigned *__restrict__ out_mask, unsigned long long *__restrict__ out_hash,
                                                unsigned long long *__restrict__ out_lhash, int *__restrict__ out_stride) {
  const int slab = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (slab >= n_slabs) return;
  const int r = slab * 64 + lane;
  const bool live = r < n_crows;
  int p = live ? crp[r] : 0;
  const int e = live ? crp[r + 1] : 0;
  const int rid = live ? (row_ids ? row_ids[r] : r) : 0;
  unsigned mask = 0;
  int count = 0;
  bool fail = false;
  unsigned long long lh = 1469598103934665603ull;
  for (;;) {
    const int cur = p < e ? col[p] - rid : 0x7fffffff;
    const int mn = kp_wave_min(cur);
    if (mn == 0x7fffffff) break;
    if (count == PA_PELL_MAXW) { fail = true; break; }
    if (lane == 0) out_D[(size_t)slab * PA_PELL_MAXW + count] = mn;
    const bool hit = cur == mn;
    lh ^= __ballot(hit);
    lh *= 1099511628211ull;
    lh ^= lh >> 29;
    if (hit) { mask |= 1u << count; ++p; }
    ++count;
  }
  if (live) out_mask[r] = fail ? 0u : mask;
  const int rid0 = __shfl(rid, 0, 64), rid1 = __shfl(rid, 1, 64);
  const int n_live = min(64, n_crows - slab * 64);
  int st = n_live > 1 ? rid1 - rid0 : 1;
  if (st != 1 && st != 2) st = 0;
  if (__ballot(live && rid != rid0 + st * lane) != 0ull) st = 0;
  if (lane == 0) {
    out_len[slab] = fail ? -1 : count;
    unsigned long long h = 1469598103934665603ull ^ (unsigned long long)count;
    if (!fail)
      for (int k = 0; k < count; ++k) {
        h ^= (unsigned long long)(unsigned)out_D[(size_t)slab * PA_PELL_MAXW + k];
        h *= 1099511628211ull;
        h ^= h >> 29;
      }
    out_hash[slab] = h;
    out_lhash[slab] = (lh ^ (unsigned long long)st) * 1099511628211ull;
    out_stride[slab] = st;
  }
}

// the lane ballots of a class, from the row masks of the slab that stands for it (one wavefront per class)
__global__ __launch_bounds__(64) void kp_class_planes(const unsigned *__restrict__ mask, const long long *__restrict__ rep, const int *__restrict__ len,
                                                      int n_crows, unsigned long long *__restrict__ plane) {
  const int cls = blockIdx.x, lane = threadIdx.x;
  const long long slab = rep[cls];
  const long long r = slab * 64 + lane;
  const unsigned m = r < n_crows ? mask[r] : 0u;
  const int L = len[slab];
  for (int k = 0; k < PA_PELL_TW; ++k) {
    const unsigned long long b = k < L ? __ballot((m >> k) & 1u) : 0ull;
    if (lane == 0) plane[(size_t)cls * PA_PELL_TW + k] = b;
  }
}

// classes: every slab's lane ballots and stride against its class's (a hash collision would be caught here)
__global__ __launch_bounds__(256) void kp_verify_classes(const unsigned *__restrict__ mask, const int *__restrict__ len, const int *__restrict__ stride,
                                                         const int2 *__restrict__ desc, const int *__restrict__ pdelta,
                                                         const unsigned long long *__restrict__ plane, const int *__restrict__ row_ids, int n_crows,
                                                         int n_slabs, int n_cols, int *__restrict__ bad) {
  const int slab = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (slab >= n_slabs) return;
  const int r = slab * 64 + lane;
  const unsigned m = r < n_crows ? mask[r] : 0u;
  const int cls = desc[slab].x & 0xfffff;
  const int L = len[slab];
  bool wrong = stride[slab] != pdelta[(size_t)cls * PA_PELL_TW + PA_PELL_T_STRIDE];
  for (int k = 0; k < L; ++k)
    if (__ballot((m >> k) & 1u) != plane[(size_t)cls * PA_PELL_TW + k]) wrong = true;
  if (wrong && lane == 0) atomicOr(bad, 1);
  // bad[1]: slabs the lean form can serve (a stride, every gather of every lane in range: pa_pell_slab)
  const int *dl = pdelta + (size_t)cls * PA_PELL_TW;
  const int st = dl[PA_PELL_T_STRIDE], row0 = row_ids ? row_ids[slab * 64] : slab * 64;
  if (lane == 0 && st != 0 && row0 + dl[PA_PELL_T_MIN] >= 0 && row0 + 63 * st + dl[PA_PELL_T_MAX] < n_cols) atomicAdd(bad + 1, 1);
}

// every slab's union against the table row its hash was mapped to
__global__ void kp_verify(const int *__restrict__ len, const int *__restrict__ D, const int2 *__restrict__ desc, const int *__restrict__ pdelta,
                          int n_slabs, int *__restrict__ bad) {
  const int slab = blockIdx.x * blockDim.x + threadIdx.x;
  if (slab >= n_slabs) return;
  const int pat = desc[slab].x & 0xfffff;
  for (int k = 0; k < len[slab]; ++k)
    if (D[(size_t)slab * PA_PELL_MAXW + k] != pdelta[(size_t)pat * PA_PELL_TW + k]) { atomicOr(bad, 1); return; }
}

// VM 0: the values, delta-major per slab; VM 1: one bit per entry (its dictionary code), a word per row -- and per slab {the rows'
// words OR-ed, 1 when every row's word is that one under its mask and both dictionary values are finite}: such a slab's bits are a
// scalar to the product kernel (pa_pell_slab_fast); n_uniform counts them
template <int VM, int U>
__global__ __launch_bounds__(256) void kp_fill(const int *__restrict__ crp, const double *__restrict__ val, const unsigned char *__restrict__ code,
                                               const unsigned *__restrict__ mask, const int2 *__restrict__ desc, int n_crows, int n_slabs,
                                               double *__restrict__ out_val, unsigned *__restrict__ out_bits, const double *__restrict__ dict,
                                               uint2 *__restrict__ out_sbits, unsigned long long *__restrict__ n_uniform,
                                               const int *__restrict__ pdelta, const int *__restrict__ row_ids, int n_cols) {
  const int slab = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (slab >= n_slabs) return;
  const int r = slab * 64 + lane;
  const bool live = r < n_crows;
  const unsigned m0 = live ? mask[r] : 0u;
  unsigned m = m0;
  int p = live ? crp[r] : 0;
  const unsigned first = (unsigned)desc[slab].y;
  unsigned bits = 0;
  while (m) {
    const int k = __builtin_ctz(m);
    m &= m - 1;
    if (VM == 0) out_val[pa_pell_slot<U>(first, k, lane)] = val[p];
    else if (VM == 1) bits |= (unsigned)(code[p] & 1) << k;
    else {
      // VM 2: the entry's dictionary code, a byte of the lane's dwords of group (first + k) / U (pa_pell_codes)
      unsigned char *cb = reinterpret_cast<unsigned char *>(out_bits) + (((size_t)(first + (unsigned)k) / U) * 64 + (size_t)lane) * (((U + 3) / 4) * 4);
      cb[(first + (unsigned)k) % U] = code[p];
    }
    ++p;
  }
  if (VM == 1) {
    if (live) out_bits[r] = bits;
    unsigned all = bits;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) all |= __shfl_xor(all, o, 64);
    const bool same = __ballot(bits != (all & m0)) == 0ull;
    const double d0 = dict[0], d1 = dict[1];
    const bool fin = d0 - d0 == 0.0 && d1 - d1 == 0.0;
    if (lane == 0) {
      out_sbits[slab] = make_uint2(all, same && fin ? 1u : 0u);
      const int *dl = pdelta + (size_t)(desc[slab].x & 0xfffff) * PA_PELL_TW;      // (counted: the slabs the lean form serves, as kp_verify_classes)
      const int st = dl[PA_PELL_T_STRIDE], row0 = row_ids ? row_ids[slab * 64] : slab * 64;
      if (same && fin && st != 0 && row0 + dl[PA_PELL_T_MIN] >= 0 && row0 + 63 * st + dl[PA_PELL_T_MAX] < n_cols) atomicAdd(n_uniform, 1ull);
    }
  }
}

void pa_pell_struct_free(pa_ctx *c, pa_pell *P);
void pa_pell_free(pa_csr *A) {
  pa_pell_struct_free(A->ctx, A->pell);
  A->pell = nullptr;
}

static bool pell_wanted() {
  const char *e = getenv("PA_SPMV_PELL");
  return !(e && atoi(e) == 0);
}

// bytes of the one-byte stream: per group of U deltas and lane ceil(U / 4) dwords (pa_pell_codes, pa_pell.h)
static size_t pell_codes_bytes(const pa_pell *P) { return (size_t)std::max<int64_t>(P->slots / P->U, 1) * 64 * (size_t)(((P->U + 3) / 4) * 4); }

static int pell_fill(pa_csr *A, bool bits, bool codes = false) {
  pa_pell *P = A->pell;
  hipStream_t s = A->ctx->s[0];
  const dim3 grid((unsigned)((P->n_slabs + 3) / 4));
#define PA_FILL0(UU) hipLaunchKernelGGL((kp_fill<0, UU>), grid, dim3(256), 0, s, A->d_crp, A->d_val, (const unsigned char *)nullptr, P->d_mask, \
                                        P->d_desc, (int)A->n_crows, (int)P->n_slabs, P->d_val, (unsigned *)nullptr, (const double *)nullptr,   \
                                        (uint2 *)nullptr, (unsigned long long *)nullptr, (const int *)nullptr, (const int *)nullptr, 0)
#define PA_FILL2(UU) hipLaunchKernelGGL((kp_fill<2, UU>), grid, dim3(256), 0, s, A->d_crp, (const double *)nullptr, A->d_code, P->d_mask,      \
                                        P->d_desc, (int)A->n_crows, (int)P->n_slabs, (double *)nullptr, P->d_codes, (const double *)nullptr,   \
                                        (uint2 *)nullptr, (unsigned long long *)nullptr, (const int *)nullptr, (const int *)nullptr, 0)
  if (codes) {
    // (bytes of absent entries stay code 0: a value of the dictionary, multiplied by an x of 0.0 in the lean form, never used in the masked one)
    PA_HIP(hipMemsetAsync(P->d_codes, 0, pell_codes_bytes(P), s));
    switch (P->U) {
      case 9: PA_FILL2(9); break;
      case 7: PA_FILL2(7); break;
      case 5: PA_FILL2(5); break;
      default: PA_FILL2(4); break;
    }
  } else if (bits) {
    PA_HIP(hipMemsetAsync(P->d_sbits + P->n_slabs, 0, sizeof(unsigned long long), s));      // (the counter sits behind the last slab's pair)
    hipLaunchKernelGGL((kp_fill<1, 4>), grid, dim3(256), 0, s, A->d_crp, (const double *)nullptr, A->d_code, P->d_mask, P->d_desc, (int)A->n_crows,
                       (int)P->n_slabs, (double *)nullptr, P->d_bits, (const double *)A->d_dict, P->d_sbits,
                       reinterpret_cast<unsigned long long *>(P->d_sbits + P->n_slabs), P->d_pdelta, A->d_row_ids, (int)A->n_cols);
  } else switch (P->U) {
    case 9: PA_FILL0(9); break;
    case 7: PA_FILL0(7); break;
    case 5: PA_FILL0(5); break;
    default: PA_FILL0(4); break;
  }
#undef PA_FILL0
#undef PA_FILL2
  PA_HIP(hipGetLastError());
  if (codes) P->codes_epoch = A->val_epoch;
  else if (bits) P->bits_epoch = A->val_epoch;
  return PA_OK;
}

// (the stream is idle) how many slabs the lean form serves on the one-bit stream: the byte accounting's only use of it
static void pell_read_lean_bits(pa_csr *A) {
  pa_pell *P = A->pell;
  unsigned long long n = 0;
  if (P->d_sbits && hipMemcpy(&n, P->d_sbits + P->n_slabs, sizeof(n), hipMemcpyDeviceToHost) == hipSuccess) P->n_lean_bits = (int64_t)n;
  else (void)hipGetLastError();
}

void pa_pell_struct_free(pa_ctx *c, pa_pell *P) {
  if (!P) return;
  pa_dev_free(c, P->d_desc);
  pa_dev_free(c, P->d_pdelta);
  pa_dev_free(c, P->d_mask);
  pa_dev_free(c, P->d_plane);
  pa_dev_free(c, P->d_prel);
  pa_dev_free(c, P->d_sbits);
  if (P->d_bits) pa_dev_free(c, P->d_bits);
  if (P->d_codes) pa_dev_free(c, P->d_codes);
  if (P->d_val) pa_dev_free(c, P->d_val);
  delete P;
}

// The structure of a block's pattern-ELL storage -- slab descriptors, pattern table, row masks; no values -- from its row pointers and
// columns in HBM (d_col: every stored entry's 0-based column in storage order).  NULL (and *why) when the block does not qualify.
pa_pell *pa_pell_structure(pa_ctx *c, const int32_t *d_crp, const int32_t *d_col, const int32_t *d_row_ids, int64_t n_crows, int64_t n_cols,
                           int64_t nnz, bool compact, const char **why) {
  hipStream_t s = c->s[0];
  const int64_t n_slabs = (n_crows + 63) / 64;
  pa_pell *P = new pa_pell();
  P->n_slabs = n_slabs;
  auto give_up = [&](const char *w) -> pa_pell * {
    (void)hipGetLastError();
    pa_pell_struct_free(c, P);
    *why = w;
    return nullptr;
  };
  scratch sc;
  int *d_len = nullptr, *d_D = nullptr, *d_bad = nullptr, *d_stride = nullptr;
  unsigned long long *d_hash = nullptr, *d_lhash = nullptr;
  if (sc.get(&d_len, (size_t)n_slabs) || sc.get(&d_D, (size_t)n_slabs * PA_PELL_MAXW) || sc.get(&d_hash, (size_t)n_slabs) || sc.get(&d_bad, 4) ||
      sc.get(&d_lhash, (size_t)n_slabs) || sc.get(&d_stride, (size_t)n_slabs))
    return give_up("no room for the set-up's temporaries");
  if (pa_dev_alloc(c, (void **)&P->d_mask, sizeof(unsigned) * (size_t)n_crows, PA_MEM_MATRIX)) return give_up("no room");
  hipLaunchKernelGGL(kp_union, dim3((unsigned)((n_slabs + 3) / 4)), dim3(256), 0, s, d_crp, d_col, d_row_ids, (int)n_crows, (int)n_slabs,
                     d_len, d_D, P->d_mask, d_hash, d_lhash, d_stride);
  std::vector<int> len((size_t)n_slabs);
  std::vector<unsigned long long> hash((size_t)n_slabs);
  if (d2h(s, len.data(), d_len, (size_t)n_slabs) || d2h(s, hash.data(), d_hash, (size_t)n_slabs)) return give_up("read-back failed");
  // the unroll: the commonest width decides (27 -> 9, 7 -> 7, 5 -> 5, else 4); widths are padded to it
  std::map<int, int64_t> freq;
  int max_w = 0;
  for (int64_t k = 0; k < n_slabs; ++k) {
    if (len[k] < 0) return give_up("a slab of 64 rows has more than 32 distinct column offsets");
    ++freq[len[k]];
    max_w = std::max(max_w, len[k]);
  }
  int common = 0;
  int64_t best = -1;
  for (auto &kv : freq) if (kv.second > best) { best = kv.second; common = kv.first; }
  const int U = common % 9 == 0 && common ? 9 : common % 7 == 0 && common ? 7 : common % 5 == 0 && common ? 5 : 4;
  P->U = U; P->max_w = max_w;
  // distinct unions -> pattern ids (by hash and length; kp_verify compares the lists themselves); and, finer, slab CLASSES: slabs of one
  // union whose deltas sit in the same LANES and whose row ids have the same stride (a structured grid has a few dozen: the slabs at
  // the start, in the middle and at the end of a grid line, times the 9 kinds of lines).  With classes the table rows are classes; an
  // operator whose rows drop entries at random has as many classes as slabs and keeps the plain patterns (no lean form, pa_pell.h).
  std::vector<unsigned long long> lhash((size_t)n_slabs);
  std::vector<int> stride((size_t)n_slabs);
  if (d2h(s, lhash.data(), d_lhash, (size_t)n_slabs) || d2h(s, stride.data(), d_stride, (size_t)n_slabs)) return give_up("read-back failed");
  std::map<std::pair<unsigned long long, int>, int> ids;
  std::map<std::tuple<unsigned long long, int, unsigned long long>, int> cids;
  std::vector<int64_t> rep, crep;
  std::vector<int> pat_of((size_t)n_slabs), cls_of((size_t)n_slabs);
  const char *ec = getenv("PA_SPMV_PELL_CLASSES");
  bool classes = !(ec && atoi(ec) == 0);
  for (int64_t k = 0; k < n_slabs; ++k) {
    auto key = std::make_pair(hash[k], len[k]);
    auto it = ids.find(key);
    if (it == ids.end()) {
      if (ids.size() >= 4096) return give_up("more than 4096 distinct slab patterns");
      it = ids.emplace(key, (int)ids.size()).first;
      rep.push_back(k);
    }
    pat_of[(size_t)k] = it->second;
    if (classes) {
      auto ckey = std::make_tuple(hash[k], len[k], lhash[k]);
      auto ct = cids.find(ckey);
      if (ct == cids.end()) {
        if (cids.size() >= 4096) { classes = false; continue; }
        ct = cids.emplace(ckey, (int)cids.size()).first;
        crep.push_back(k);
      }
      cls_of[(size_t)k] = ct->second;
    }
  }
  const std::vector<int64_t> &trep = classes ? crep : rep;            // the slab that stands for a table row
  std::vector<int2> desc((size_t)n_slabs);
  int64_t slots = 0;
  for (int64_t k = 0; k < n_slabs; ++k) {
    const int wp = (len[k] + U - 1) / U * U;
    if (slots + wp >= ((int64_t)1 << 32)) return give_up("value stream too long for 32-bit slab offsets");
    desc[(size_t)k].x = (classes ? cls_of[(size_t)k] : pat_of[(size_t)k]) | (wp << 20);
    desc[(size_t)k].y = (int)(unsigned)slots;
    slots += wp;
  }
  if (slots * 64 > nnz + nnz / 4 + 64 * 64) return give_up("the slab-wide unions would pad the value stream by more than 25 %");
  P->n_patterns = (int64_t)ids.size(); P->n_table = (int64_t)trep.size(); P->n_classes = classes ? (int64_t)crep.size() : 0; P->slots = slots;
  std::vector<int> table((size_t)P->n_table * PA_PELL_TW, 0);
  for (size_t i = 0; i < trep.size(); ++i)
    if (len[(size_t)trep[i]] > 0 &&
        hipMemcpyAsync(&table[i * PA_PELL_TW], d_D + (size_t)trep[i] * PA_PELL_MAXW, sizeof(int) * (size_t)len[(size_t)trep[i]], hipMemcpyDeviceToHost, s) != hipSuccess)
      return give_up("read-back failed");
  if (hipStreamSynchronize(s) != hipSuccess) return give_up("read-back failed");
  // (slots between a union's length and its padded width repeat the FIRST delta: no row has them -- masks and ballots say so --, but the
  //  lean form gathers before it selects, and a gather at the first run's columns is in range whenever the slab's real gathers are)
  for (size_t i = 0; i < trep.size(); ++i) {
    const int L = len[(size_t)trep[i]];
    for (int k = L; L > 0 && k < PA_PELL_T_STRIDE; ++k) table[i * PA_PELL_TW + k] = table[i * PA_PELL_TW];
  }
  // runs of three consecutive deltas in every pattern (the 27-point operator: nine per row): the kernel's R3 form, one gather per run
  {
    const char *e3 = getenv("PA_SPMV_PELL_RUNS3");
    bool r3 = U == 9 && (!compact || classes) && !(e3 && atoi(e3) == 0);
    for (size_t i = 0; i < trep.size() && r3; ++i) {
      const int L = len[(size_t)trep[i]];
      if (L % 3) { r3 = false; break; }
      for (int k = 0; k < L; k += 3)
        if (table[i * PA_PELL_TW + k + 1] != table[i * PA_PELL_TW + k] + 1 || table[i * PA_PELL_TW + k + 2] != table[i * PA_PELL_TW + k] + 2) { r3 = false; break; }
    }
    P->runs3 = r3;
  }
  if (pa_dev_alloc(c, (void **)&P->d_desc, sizeof(int2) * (size_t)n_slabs, PA_MEM_MATRIX) ||
      pa_dev_alloc(c, (void **)&P->d_pdelta, sizeof(int) * table.size(), PA_MEM_MATRIX))
    return give_up("no room");
  if (pa_h2d(P->d_desc, desc.data(), sizeof(int2) * (size_t)n_slabs) != hipSuccess) return give_up("upload failed");
  if (classes) {
    // a class's row: stride, lowest / highest delta, and whether every lane has every delta (flags bit 0); its lane ballots in d_plane
    long long *d_rep = nullptr;
    if (sc.get(&d_rep, trep.size()) || pa_dev_alloc(c, (void **)&P->d_plane, sizeof(unsigned long long) * table.size(), PA_MEM_MATRIX))
      return give_up("no room");
    std::vector<long long> hrep(trep.begin(), trep.end());
    if (pa_h2d(d_rep, hrep.data(), sizeof(long long) * hrep.size()) != hipSuccess) return give_up("upload failed");
    hipLaunchKernelGGL(kp_class_planes, dim3((unsigned)trep.size()), dim3(64), 0, s, P->d_mask, d_rep, d_len, (int)n_crows, P->d_plane);
    std::vector<unsigned long long> plane(table.size());
    std::vector<unsigned> rel(table.size(), 0u);
    if (d2h(s, plane.data(), P->d_plane, plane.size())) return give_up("read-back failed");
    for (size_t i = 0; i < trep.size(); ++i) {
      const int L = len[(size_t)trep[i]];
      int *row = &table[i * PA_PELL_TW];
      bool full = L > 0 && L % U == 0;
      for (int k = 0; k < L && full; ++k) full = plane[i * PA_PELL_TW + k] == ~0ull;
      row[PA_PELL_T_STRIDE] = L > 0 ? stride[(size_t)trep[i]] : 0;
      row[PA_PELL_T_FLAGS] = full ? 1 : 0;
      row[PA_PELL_T_MIN] = L > 0 ? row[0] : 0;
      row[PA_PELL_T_MAX] = L > 0 ? row[L - 1] : 0;
      // the lean form's byte offsets from the column of lane 0 at the lowest delta (32 bits: a class whose span does not fit has no lean form)
      if (((int64_t)row[PA_PELL_T_MAX] - row[PA_PELL_T_MIN] + 130) * 8 >= ((int64_t)1 << 32)) row[PA_PELL_T_STRIDE] = 0;
      for (int k = 0; k < PA_PELL_T_STRIDE; ++k)
        rel[i * PA_PELL_TW + k] = row[PA_PELL_T_STRIDE] ? (unsigned)(((int64_t)row[k] - row[PA_PELL_T_MIN]) * 8) : 0u;
    }
    if (pa_dev_alloc(c, (void **)&P->d_prel, sizeof(unsigned) * rel.size(), PA_MEM_MATRIX)) return give_up("no room");
    if (pa_h2d(P->d_prel, rel.data(), sizeof(unsigned) * rel.size()) != hipSuccess) return give_up("upload failed");
  }
  if (pa_h2d(P->d_pdelta, table.data(), sizeof(int) * table.size()) != hipSuccess || hipMemsetAsync(d_bad, 0, 3 * sizeof(int), s) != hipSuccess)
    return give_up("upload failed");
  hipLaunchKernelGGL(kp_verify, grid1(n_slabs), dim3(256), 0, s, d_len, d_D, P->d_desc, P->d_pdelta, (int)n_slabs, d_bad);
  if (classes)
    hipLaunchKernelGGL(kp_verify_classes, dim3((unsigned)((n_slabs + 3) / 4)), dim3(256), 0, s, P->d_mask, d_len, d_stride, P->d_desc, P->d_pdelta,
                       P->d_plane, d_row_ids, (int)n_crows, (int)n_slabs, (int)n_cols, d_bad);
  int bad[3] = {1, 0, 0};
  if (d2h(s, bad, d_bad, 3) || bad[0]) return give_up("two different slab patterns share a hash");
  P->n_lean = bad[1];
  return P;
}

// Called at the end of a slab's creation (pa_csr.hip).  Never an error to the caller: a block that does not qualify, or a device
// without room for the second value stream, keeps the row-split kernel.
int pa_pell_build(pa_csr *A) {
  if (A->pell || !pell_wanted() || pa_tls_plain_encoding || A->nnz == 0 || A->n_crows == 0 || A->next || A->n_xw_groups > 0) return PA_OK;
  // (every block is looked at -- a decode and one pass over its entries: the row split's pattern detector wants whole rows to repeat, this
  //  storage only a slab's UNION of offsets to stay within 32, e.g. rows that each drop a few entries of a stencil; blocks of scattered rows
  //  fail at their first slab's 33rd offset)
  if (A->ctx->capturing) return PA_OK;
  pa_ctx *c = A->ctx;
  PA_HIP(hipSetDevice(c->device));
  hipStream_t s = c->s[0];
  const auto t_begin = std::chrono::steady_clock::now();
  auto give_up = [&](const char *why) {
    (void)hipGetLastError();
    pa_pell_free(A);
    if (getenv("PA_SETUP_TIMING")) fprintf(stderr, "[pa setup] pattern-ELL of %lld entries: none (%s)\n", (long long)A->nnz, why);
    return PA_OK;
  };
  const char *why = "";
  {
    scratch sc;
    int32_t *d_row = nullptr, *d_col = nullptr;
    if (sc.get(&d_row, (size_t)A->nnz + 8) || sc.get(&d_col, (size_t)A->nnz + 8)) return give_up("no room for the set-up's temporaries");
    if (pa_dev_decode_entries(A, d_row, d_col) != PA_OK) return give_up("decode failed");
    A->pell = pa_pell_structure(c, A->d_crp, d_col, A->d_row_ids, A->n_crows, A->n_cols, A->nnz, A->compact, &why);
  }
  if (!A->pell) return give_up(why);
  pa_pell *P = A->pell;
  const int64_t slots = P->slots;
  // the value stream: one bit per entry when the block's dictionary (built just before) has at most two values, else fp64
  const bool two = A->use_vdict && A->n_dict <= 2;
  if (two) {
    if (pa_dev_alloc(c, (void **)&P->d_bits, sizeof(unsigned) * (size_t)A->n_crows, PA_MEM_MATRIX) ||
        pa_dev_alloc(c, (void **)&P->d_sbits, sizeof(uint2) * ((size_t)P->n_slabs + 1), PA_MEM_MATRIX))
      return give_up("no room");
  } else if (!A->use_vdict) {
    if (pa_dev_alloc(c, (void **)&P->d_val, sizeof(double) * (size_t)std::max<int64_t>(slots, 1) * 64, PA_MEM_MATRIX)) return give_up("no room");
    if (hipMemsetAsync(P->d_val, 0, sizeof(double) * (size_t)std::max<int64_t>(slots, 1) * 64, s) != hipSuccess) return give_up("memset failed");
  } else {
    // a dictionary of 3 .. 64 values: one BYTE per entry in pattern-ELL order (round 6, third step; PA_SPMV_PELL_BYTES=0: the row-split
    // kernel's one-byte stream serves such blocks as before)
    const char *eb = getenv("PA_SPMV_PELL_BYTES");
    if (eb && atoi(eb) == 0) return give_up("a dictionary of more than two values: the row-split kernel's one-byte stream serves (PA_SPMV_PELL_BYTES=0)");
    if (pa_dev_alloc(c, (void **)&P->d_codes, pell_codes_bytes(P), PA_MEM_MATRIX)) return give_up("no room");
  }
  const bool bytes = P->d_codes != nullptr;
  if (pell_fill(A, two, bytes) != PA_OK || hipStreamSynchronize(s) != hipSuccess) return give_up("fill failed");
  if (two) pell_read_lean_bits(A);
  if (getenv("PA_SETUP_TIMING"))
    fprintf(stderr, "[pa setup] pattern-ELL of %lld entries: %lld slabs, %lld patterns, %lld classes, lean form in %lld slabs, width <= %d, unroll %d, %lld slots (%.3f x the entries), %s%s, %.3f ms\n",
            (long long)A->nnz, (long long)P->n_slabs, (long long)P->n_patterns, (long long)P->n_classes, (long long)(two ? P->n_lean_bits : P->n_lean),
            P->max_w, P->U, (long long)slots * 64, slots * 64.0 / A->nnz,
            P->runs3 ? "runs of three, " : "", two ? "one bit per entry" : bytes ? "one byte per entry" : "fp64 stream",
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
  return PA_OK;
}

// behind a value update of slab A (queued on the compute stream): the fp64 stream follows in place
int pa_pell_after_update(pa_csr *A) {
  pa_pell *P = A->pell;
  if (P && P->d_val) PA_TRY(pell_fill(A, false));
  return PA_OK;
}

// behind a (re)built value dictionary of slab A (vdict_build, pa_csr.hip; the codes are in HBM): with at most two values the bits are
// made again from the codes, on the compute stream, complete when this returns (the next product may run on the comm stream)
int pa_pell_bits_refresh(pa_csr *A) {
  pa_pell *P = A->pell;
  if (!P || !A->use_vdict) return PA_OK;
  // 3 .. 64 values: the one-byte stream again (made at creation when the block had such a dictionary then; a block that got its
  // dictionary later gets the stream here, outside captures).  A stream that exists follows EVERY renewal, also one that left two
  // values or fewer: a recorded graph may still replay the one-byte kernel, and codes + dictionary of any size give it the new values.
  const char *eb = getenv("PA_SPMV_PELL_BYTES");
  const bool bytes_on = !(eb && atoi(eb) == 0);
  if (bytes_on && (A->n_dict > 2 || P->d_codes)) {
    if (!P->d_codes) {
      if (A->ctx->capturing) return PA_OK;
      if (pa_dev_alloc(A->ctx, (void **)&P->d_codes, pell_codes_bytes(P), PA_MEM_MATRIX) != PA_OK) { (void)hipGetLastError(); P->d_codes = nullptr; return PA_OK; }
    }
    PA_TRY(pell_fill(A, false, true));
    if (A->n_dict > 2) {
      if (!A->ctx->capturing) PA_HIP(hipStreamSynchronize(A->ctx->s[0]));
      return PA_OK;
    }
  }
  if (A->n_dict > 2) return PA_OK;
  if (!P->d_bits) {
    if (A->ctx->capturing) return PA_OK;
    if (pa_dev_alloc(A->ctx, (void **)&P->d_bits, sizeof(unsigned) * (size_t)A->n_crows, PA_MEM_MATRIX) != PA_OK) { (void)hipGetLastError(); P->d_bits = nullptr; return PA_OK; }
    if (pa_dev_alloc(A->ctx, (void **)&P->d_sbits, sizeof(uint2) * ((size_t)P->n_slabs + 1), PA_MEM_MATRIX) != PA_OK) {
      (void)hipGetLastError();
      pa_dev_free(A->ctx, P->d_bits);
      P->d_bits = nullptr; P->d_sbits = nullptr;
      return PA_OK;
    }
  }
  PA_TRY(pell_fill(A, true));
  if (!A->ctx->capturing) {
    PA_HIP(hipStreamSynchronize(A->ctx->s[0]));
    pell_read_lean_bits(A);
  }
  return PA_OK;
}

// 0: the row-split kernel serves; 1: fp64 stream; 2: one bit per entry; 3: one byte per entry
int pa_pell_mode(const pa_csr *A) {
  const pa_pell *P = A->pell;
  if (!P || A->alpha_inside || A->accumulate || !A->ctx->sw.pell) return 0;
  if (P->d_bits && A->use_vdict && !A->vdict_stale && A->n_dict <= 2 && P->bits_epoch == A->val_epoch) return 2;
  if (P->d_codes && A->use_vdict && !A->vdict_stale && A->n_dict > 2 && P->codes_epoch == A->val_epoch && A->ctx->sw.pell_bytes) return 3;
  return P->d_val ? 1 : 0;
}

static pa_pell_dev pell_dev(const pa_csr *A, int mode) {
  const pa_pell *P = A->pell;
  pa_pell_dev D;
  D.desc = P->d_desc; D.pdelta = P->d_pdelta; D.mask = P->d_mask; D.bits = P->d_bits; D.val = P->d_val; D.dict = A->d_dict;
  D.row_ids = A->d_row_ids; D.n_slabs = (int)P->n_slabs; D.n_crows = (int)A->n_crows; D.n_cols = (int)A->n_cols;
  D.plane = A->ctx->sw.pell_lean ? P->d_plane : nullptr; D.prel = P->d_prel; D.sbits = P->d_sbits; D.codes = P->d_codes;
  // (one byte per entry: the lean form multiplies an absent entry's value -- code 0's -- by an x of 0.0: every dictionary value must be finite)
  if (mode == 3 && !A->dict_finite) D.plane = nullptr;
  return D;
}

// the product (or one of its epilogue forms) of slab A on the pattern-ELL kernel; mode from pa_pell_mode (1 or 2).  epi as
// k_spmv_rowsplit's EPI; EPI 3 writes one partial per SLAB (pa_pell_partials of them) into gs_x.
int pa_pell_launch(const pa_csr *A, int mode, int epi, const double *x, double *y, double alpha, double beta, double *gs_x,
                   const double *gs_b, const double *gs_diag, hipStream_t st) {
  pa_pell *P = A->pell;
  PA_REQUIRE(epi == 0 || alpha == 1.0, "the epilogue forms of the pattern-ELL kernel take alpha = 1");
  if (!st) st = A->ctx->s[0];
  const pa_pell_dev D = pell_dev(A, mode);
  const int n_groups = (int)((P->n_slabs + 3) / 4);
  int bpx = (n_groups + 7) / 8;
  const int nblk = bpx * 8;
  if (A->ctx->sw.spmv_alternate && epi == 0 && ((P->n_launched++) & 1)) bpx = -bpx;
  if (mode == 2 && A->ctx->capturing) { const_cast<pa_csr *>(A)->vd_captured = true; const_cast<pa_csr *>(A)->vd_captured_two = true; }
  if (mode == 3 && A->ctx->capturing) const_cast<pa_csr *>(A)->vd_captured = true;
  if (mode == 3) pa_pell_launch_v2(A, D, epi, nblk, bpx, x, y, alpha, beta, gs_x, gs_b, gs_diag, st);
  else if (mode == 2) pa_pell_launch_v1(A, D, epi, nblk, bpx, x, y, alpha, beta, gs_x, gs_b, gs_diag, st);
  else pa_pell_launch_v0(A, D, epi, nblk, bpx, x, y, alpha, beta, gs_x, gs_b, gs_diag, st);
  PA_HIP(hipGetLastError());
  return PA_OK;
}

// what a launch that runs the slabs itself needs (the fused product, pa_fused.hip): mode from pa_pell_mode (1 or 2)
void pa_pell_describe(const pa_csr *A, int mode, pa_pell_dev *D, int *U, bool *runs3, int64_t *n_slabs) {
  *D = pell_dev(A, mode);
  *U = A->pell->U; *runs3 = A->pell->runs3 && !A->compact; *n_slabs = A->pell->n_slabs;
}

int64_t pa_pell_partials(const pa_csr *A) { return A->pell ? A->pell->n_slabs : 0; }

// bytes the pattern-ELL product of this slab reads from the matrix side (pa_csr_stream_bytes counts the row-split kernel's)
int64_t pa_pell_stream_bytes(const pa_csr *A, int mode) {
  const pa_pell *P = A->pell;
  // (the lean form reads neither the row masks nor, on the one-bit stream, the rows' bits: its slabs cost their descriptor, the
  //  class table and 8 bytes of slab bits)
  const bool lean_on = A->ctx->sw.pell_lean && P->d_plane;
  const int64_t lean = !lean_on || (mode == 3 && !A->dict_finite) ? 0 : mode == 2 ? P->n_lean_bits : P->n_lean;
  const int64_t rows_masked = std::max<int64_t>(0, A->n_crows - 64 * lean);
  int64_t t = 8 * P->n_slabs + 4 * P->n_table * PA_PELL_TW + 4 * rows_masked;
  if (lean_on) t += 12 * P->n_table * PA_PELL_TW;
  t += mode == 2 ? 4 * rows_masked + 8 * P->n_slabs + 16 : mode == 3 ? (int64_t)pell_codes_bytes(P) + 8 * PA_VDICT_MAX : 8 * 64 * P->slots;
  if (A->compact) t += 4 * rows_masked + (lean ? 4 * lean : 0);
  return t;
}

// *mode: what pa_spmv runs this block on now (0 row split, 1 pattern-ELL fp64, 2 pattern-ELL one bit per entry); slabs / patterns /
// value slots / unroll of the pattern-ELL storage (0 when the block has none)
extern "C" int pa_csr_pell_info(const pa_csr *A, int *mode, int64_t *n_slabs, int64_t *n_patterns, int64_t *value_slots, int *unroll) {
  PA_REQUIRE(A != nullptr, "csr is NULL");
  const pa_pell *P = A->pell;
  if (mode) *mode = pa_pell_mode(A);
  if (n_slabs) *n_slabs = P ? P->n_slabs : 0;
  if (n_patterns) *n_patterns = P ? P->n_patterns : 0;
  if (value_slots) *value_slots = P ? P->slots * 64 : 0;
  if (unroll) *unroll = P ? P->U : 0;
  return PA_OK;
}

extern "C" int pa_csr_pell_lean_info(const pa_csr *A, int64_t *n_classes, int64_t *n_lean, int64_t *n_lean_bits) {
  PA_REQUIRE(A != nullptr, "csr is NULL");
  const pa_pell *P = A->pell;
  const bool on = P && P->d_plane && A->ctx->sw.pell_lean;
  if (n_classes) *n_classes = P ? P->n_classes : 0;
  if (n_lean) *n_lean = on ? P->n_lean : 0;
  if (n_lean_bits) *n_lean_bits = on ? P->n_lean_bits : 0;
  return PA_OK;
}
