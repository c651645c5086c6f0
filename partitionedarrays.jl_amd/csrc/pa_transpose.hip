// pa_transpose.hip -- transpose(A) of a block that is resident in HBM, built on the device (round 4).
//
// Reference: mul!(c, transpose(a), b, alpha, beta)  src/p_sparse_matrix.jl:2144-2162
//              ghost(c) = alpha * A_oh' * own(b);  t = assemble!(c);  own(c) = beta*own(c) + alpha * A_oo' * own(b);  wait(t)
//            spmtv!(b, A::SparseMatrixCSR, x) = spmv_csc!(b, x, A.rowptr, A.colval, A.nzval)  src/sparse_utils.jl:613-647,671-690
// The transposed product of a CSR block is the scatter loop "for row ascending, for p ascending: b[col[p]] += nz[p] * x[row]":
// an output entry j receives its contributions in ascending ROW of A.  A' stored as CSR with, inside each of its rows j, the
// entries ordered by ascending row of A therefore reproduces that sum with the row-split kernel (accumulator starts at
// beta*b[j], products added in stored order) bit for bit.  Built here without a host copy of anything per entry:
//     1. the block's column encoding (row patterns / 16-bit windows / Int32) is decoded back into (row, column) per entry
//     2. ONE stable radix sort by column (rocPRIM): equal columns keep the input order = ascending row
//     3. the sorted entries' rows / values are the columns / values of A'; its row pointer = lower bounds of the sorted keys
//     4. the block constructor of every other device-made block (pa_csr_from_device: row split + column encodings as kernels)
// Works for uploaded, generated (pa_hpcg_own_block_create) and device-assembled (pa_coo_assemble) blocks alike; a block of 2^31
// stored entries or more (a chain of slabs) is refused.
#include "pa_dev_util.h"

#include "pa_setup.h"
#include "pa_spmv_kernel.h"
#include "pa_spmv_xwin.h"

using namespace pa_util;

// (row, column) of every stored entry of one slab, decoded from whatever the slab keeps: one workgroup per chunk.  The
// arithmetic is the product kernel's (pa_spmv_kernel.h) written out per entry -- no ds_bpermute tricks, this runs once.
__global__ __launch_bounds__(256) void kt_decode(const int *__restrict__ crp, const int *__restrict__ chunk_row, const int *__restrict__ row_ids,
                                                 const int *__restrict__ col32, const unsigned short *__restrict__ col16,
                                                 const int *__restrict__ win, const int *__restrict__ pdesc, const int *__restrict__ pdelta,
                                                 int use_pattern, int use_c16, int cap, int n_chunks, int *__restrict__ out_row,
                                                 int *__restrict__ out_col) {
  const int chunk = blockIdx.x;
  if (chunk >= n_chunks) return;
  const int r0 = chunk_row[chunk], r1 = chunk_row[chunk + 1];
  const int p0 = crp[r0], p1 = crp[r1];
  const int base = p0 & ~1;
  const bool is_long = p1 - base > cap;
  const int *d = use_pattern ? pdesc + (size_t)chunk * PA_PDESC_INTS : nullptr;
  const int nseg = d ? d[0] : 0;
  const int sh16 = (d && nseg <= 0) ? d[1] : 0, sh32 = (d && nseg <= 0) ? d[2] : 0;
  const bool c16 = use_c16 && nseg <= 0 && !is_long && win[(size_t)chunk * PA_C16_WINDOWS] >= 0;
  for (int p = p0 + (int)threadIdx.x; p < p1; p += blockDim.x) {
    // the (compacted) row that holds entry p: the last r in [r0, r1) with crp[r] <= p
    int lo = r0, hi = r1 - 1;
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (crp[mid] <= p) lo = mid; else hi = mid - 1;
    }
    out_row[p] = row_ids ? row_ids[lo] : lo;
    int col;
    if (nseg > 0) {
      const int q = p - p0;
      int s = 0;
      if (nseg > 1 && q >= d[1]) s = 1;
      if (nseg > 2 && q >= d[2]) s = 2;
      if (nseg > 3 && q >= d[3]) s = 3;
      const int qs = s ? d[s] : 0;
      const int Lw = d[8 + s], L = Lw & 255, stride = (Lw >> 8) ? (Lw >> 8) : 1;
      const int t = q - qs, rr = t / L, kk = t - rr * L;
      col = d[4 + s] + rr * stride + pdelta[(size_t)d[12 + s] * PA_PAT_MAXLEN + kk];
    } else if (c16) {
      const unsigned code = col16[(size_t)p + sh16];
      col = win[(size_t)chunk * PA_C16_WINDOWS + (code >> 12)] + (int)(code & 4095u);
    } else {
      col = col32[(size_t)p + sh32];
    }
    out_col[p] = col;
  }
}

__global__ void kt_iota(int *__restrict__ v, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) v[i] = i;
}

// A' entry k = A's entry perm[k]: its column in A' is that entry's row in A
__global__ void kt_gather(const int *__restrict__ perm, const int *__restrict__ row, const double *__restrict__ val, int n,
                          int *__restrict__ out_col, double *__restrict__ out_val) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int p = perm[k];
  out_col[k] = row[p];
  out_val[k] = val[p];
}

// rp[j] = number of sorted keys < j, j = 0..n_keys (keys ascending): the row pointer of A'
__global__ void kt_lower_bounds(const int *__restrict__ keys, int n, int n_rows, int *__restrict__ rp) {
  const int j = blockIdx.x * blockDim.x + threadIdx.x;
  if (j > n_rows) return;
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (keys[mid] < j) lo = mid + 1; else hi = mid;
  }
  rp[j] = lo;
}

int pa_dev_decode_entries(const pa_csr *A, int32_t *d_row, int32_t *d_col) {
  pa_ctx *c = A->ctx;
  if (A->nnz == 0 || A->n_chunks == 0) return PA_OK;
  hipLaunchKernelGGL(kt_decode, dim3((unsigned)A->n_chunks), dim3(256), 0, c->s[0], A->d_crp, A->d_chunk_row, A->d_row_ids, A->d_col,
                     A->d_col16, A->d_win, A->d_pdesc, A->d_pdelta, A->use_pattern ? 1 : 0, A->use_c16 ? 1 : 0, PA_SPMV_CHUNK_NNZ, (int)A->n_chunks,
                     d_row, d_col);
  PA_HIP(hipGetLastError());
  return PA_OK;
}

// The whole block as (row, column, value) arrays in the CALLER'S entry order: one node -- its decoded entries and its value stream
// as they are; a column-split chain -- every piece's entries put back where the caller had them (d_src).  Chains of row slabs (2^31
// entries or more) are refused by the callers.  d_row / d_col: block_nnz(A) entries each; *d_val: A->d_val or a scratch array.
static inline int64_t block_nnz(const pa_csr *A) { return A->next ? A->t_nnz : A->nnz; }
__global__ void kt_put_back(const int *__restrict__ src, const int *__restrict__ row, const int *__restrict__ col, const double *__restrict__ val,
                            int n, int *__restrict__ out_row, int *__restrict__ out_col, double *__restrict__ out_val) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int p = src[q];
  out_row[p] = row[q]; out_col[p] = col[q]; out_val[p] = val[q];
}
static int decode_block(const pa_csr *A, scratch &sc, int32_t *d_row, int32_t *d_col, const double **d_val) {
  if (!A->next) {
    *d_val = A->d_val;
    return pa_dev_decode_entries(A, d_row, d_col);
  }
  PA_REQUIRE(A->colsplit, "a chain of row slabs");
  double *d_all = nullptr;
  PA_TRY(sc.get(&d_all, (size_t)A->t_nnz + 1));
  for (const pa_csr *S = A; S; S = S->next) {
    if (S->nnz == 0) continue;
    int32_t *t_row = nullptr, *t_col = nullptr;
    PA_TRY(sc.get(&t_row, (size_t)S->nnz));
    PA_TRY(sc.get(&t_col, (size_t)S->nnz));
    PA_TRY(pa_dev_decode_entries(S, t_row, t_col));
    hipLaunchKernelGGL(kt_put_back, grid1(S->nnz), dim3(256), 0, A->ctx->s[0], S->d_src, t_row, t_col, S->d_val, (int)S->nnz, d_row, d_col, d_all);
    PA_HIP(hipGetLastError());
    PA_HIP(hipStreamSynchronize(A->ctx->s[0]));
    sc.release(t_row); sc.release(t_col);
  }
  *d_val = d_all;
  return PA_OK;
}

// 0-based (row, column) of every stored entry in storage order, as the product kernel decodes them (tests compare this with
// the arrays the block was made from; nothing on the product path reads it)
extern "C" int pa_csr_download_entries(const pa_csr *A, int32_t *rows, int32_t *cols) {
  PA_REQUIRE(A && (A->t_nnz == 0 || (rows && cols)), "bad arguments");
  pa_ctx *c = A->ctx;
  PA_HIP(hipSetDevice(c->device));
  for (const pa_csr *S = A; S; S = S->next) {
    if (S->nnz == 0) continue;
    scratch sc;
    int32_t *d_row = nullptr, *d_col = nullptr;
    PA_TRY(sc.get(&d_row, (size_t)S->nnz));
    PA_TRY(sc.get(&d_col, (size_t)S->nnz));
    PA_TRY(pa_dev_decode_entries(S, d_row, d_col));
    if (A->colsplit) {                                     // a column piece: its entries go back to where the caller had them
      std::vector<int32_t> r((size_t)S->nnz), cc((size_t)S->nnz), src((size_t)S->nnz);
      PA_TRY(d2h(c->s[0], r.data(), d_row, (size_t)S->nnz));
      PA_TRY(d2h(c->s[0], cc.data(), d_col, (size_t)S->nnz));
      PA_TRY(d2h(c->s[0], src.data(), S->d_src, (size_t)S->nnz));
      for (int64_t p = 0; p < S->nnz; ++p) { rows[src[p]] = r[p]; cols[src[p]] = cc[p]; }
      continue;
    }
    PA_TRY(d2h(c->s[0], rows + S->nnz0, d_row, (size_t)S->nnz));
    PA_TRY(d2h(c->s[0], cols + S->nnz0, d_col, (size_t)S->nnz));
    if (S->row0) for (int64_t p = 0; p < S->nnz; ++p) rows[S->nnz0 + p] += (int32_t)S->row0;
  }
  return PA_OK;
}

extern "C" int pa_csr_create_transpose(const pa_csr *A, pa_csr **out) { return pa_csr_create_transpose_ranked(A, nullptr, out); }

__global__ void kt_rank_keys(const int *__restrict__ row, const int *__restrict__ rank, int n, int *__restrict__ key) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) key[k] = rank[row[k]];
}
__global__ void kt_gather_keys(const int *__restrict__ perm, const int *__restrict__ col, int n, int *__restrict__ key) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) key[k] = col[perm[k]];
}

// row_rank (host, n_rows entries, a permutation; or NULL): inside a row of A' the entries are ordered by ascending row_rank[row of A]
// instead of ascending row -- for a block whose rows were renumbered (pa_csr_create_permuted) row_rank = the ORIGINAL row of every
// stored row, and A' adds in the order the reference's scatter loop has on the caller's numbering.
extern "C" int pa_csr_create_transpose_ranked(const pa_csr *A, const int32_t *row_rank, pa_csr **out) {
  PA_REQUIRE(A && out, "bad arguments");
  PA_REQUIRE(!A->next || A->colsplit, "a block of 2^31 stored entries or more (a chain of slabs) has no device-side transpose");
  pa_ctx *c = A->ctx;
  PA_REQUIRE(!c->capturing, "not inside a graph capture");
  PA_HIP(hipSetDevice(c->device));
  hipStream_t s = c->s[0];
  const int64_t nnz = block_nnz(A), n_rows_t = A->n_cols, n_cols_t = A->n_rows;
  PA_REQUIRE(nnz < (int64_t)2147483000 && n_rows_t < (int64_t)2147483000, "too large for Int32 offsets");
  scratch sc;
  int32_t *d_rp = nullptr;
  PA_TRY(sc.get(&d_rp, (size_t)n_rows_t + 1));
  if (nnz == 0) {
    PA_HIP(hipMemsetAsync(d_rp, 0, sizeof(int32_t) * (n_rows_t + 1), s));
    PA_HIP(hipStreamSynchronize(s));
    return pa_csr_from_device(c, n_rows_t, n_cols_t, 0, d_rp, nullptr, nullptr, out);
  }
  int32_t *d_row = nullptr, *d_col = nullptr, *d_keys = nullptr, *d_iota = nullptr, *d_perm = nullptr, *d_tcol = nullptr;
  double *d_tval = nullptr;
  PA_TRY(sc.get(&d_row, (size_t)nnz));
  PA_TRY(sc.get(&d_col, (size_t)nnz));
  const double *d_aval = nullptr;
  PA_TRY(decode_block(A, sc, d_row, d_col, &d_aval));
  PA_TRY(sc.get(&d_keys, (size_t)nnz));
  PA_TRY(sc.get(&d_iota, (size_t)nnz));
  PA_TRY(sc.get(&d_perm, (size_t)nnz));
  hipLaunchKernelGGL(kt_iota, grid1(nnz), dim3(256), 0, s, d_iota, (int)nnz);
  // (only the bits a column index can have: fewer radix passes)
  unsigned bits = 1;
  while (bits < 32 && ((int64_t)1 << bits) < n_rows_t) ++bits;
  if (row_rank) {
    // two stable sorts: by the rows' rank first, by column second -- equal columns then stand in ascending rank
    unsigned rbits = 1;
    while (rbits < 32 && ((int64_t)1 << rbits) < n_cols_t) ++rbits;
    int32_t *d_rank = nullptr, *d_k1 = nullptr, *d_p1 = nullptr;
    PA_TRY(sc.get(&d_rank, (size_t)n_cols_t + 1));
    PA_TRY(sc.get(&d_k1, (size_t)nnz));
    PA_TRY(sc.get(&d_p1, (size_t)nnz));
    PA_HIP(hipMemcpyAsync(d_rank, row_rank, sizeof(int32_t) * n_cols_t, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(kt_rank_keys, grid1(nnz), dim3(256), 0, s, d_row, d_rank, (int)nnz, d_k1);
    PA_TRY(sort_pairs(sc, s, d_k1, d_keys, d_iota, d_p1, (size_t)nnz, rbits));
    hipLaunchKernelGGL(kt_gather_keys, grid1(nnz), dim3(256), 0, s, d_p1, d_col, (int)nnz, d_k1);
    PA_TRY(sort_pairs(sc, s, d_k1, d_keys, d_p1, d_perm, (size_t)nnz, bits));
    sc.release(d_rank); sc.release(d_k1); sc.release(d_p1);
  } else
  PA_TRY(sort_pairs(sc, s, d_col, d_keys, d_iota, d_perm, (size_t)nnz, bits));
  sc.release(d_col);
  sc.release(d_iota);
  hipLaunchKernelGGL(kt_lower_bounds, grid1(n_rows_t + 1), dim3(256), 0, s, d_keys, (int)nnz, (int)n_rows_t, d_rp);
  PA_TRY(sc.get(&d_tcol, (size_t)nnz));
  PA_TRY(sc.get(&d_tval, (size_t)nnz));
  hipLaunchKernelGGL(kt_gather, grid1(nnz), dim3(256), 0, s, d_perm, d_row, d_aval, (int)nnz, d_tcol, d_tval);
  PA_HIP(hipGetLastError());
  PA_HIP(hipStreamSynchronize(s));
  sc.release(d_keys);
  sc.release(d_perm);
  sc.release(d_row);
  return pa_csr_from_device(c, n_rows_t, n_cols_t, nnz, d_rp, d_tcol, d_tval, out);
}

__global__ void kt_remap(int *__restrict__ col, int n, const int *__restrict__ map, int *__restrict__ bad) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int c = map[col[k]];
  if (c < 0) atomicAdd(bad, 1);
  col[k] = c < 0 ? 0 : c;
}

// The same stored entries, in the same order, with every column j renamed map[j] (host array of A->n_cols entries, values in
// [0, n_cols_new); -1 = "no such column": allowed only for columns without stored entries, else PA_ERR_ARG and *out = NULL).
// Row sums keep their order of addition, so products through the new block gather x_new[map[j]] where the old one gathered
// x[j] and are otherwise the same bits.  Used by pa_mul5 to let own x ghost read the RECEIVE BUFFER of consistent! directly.
int pa_csr_create_remapped(const pa_csr *A, const int32_t *map, int64_t n_cols_new, pa_csr **out) {
  PA_REQUIRE(A && map && out && n_cols_new >= 0, "bad arguments");
  PA_REQUIRE(!A->next || A->colsplit, "a chain of slabs has no remapped twin");
  *out = nullptr;
  pa_ctx *c = A->ctx;
  PA_HIP(hipSetDevice(c->device));
  hipStream_t s = c->s[0];
  const int64_t nnz = block_nnz(A), n_rows = A->n_rows;
  scratch sc;
  int32_t *d_rp = nullptr, *d_row = nullptr, *d_col = nullptr, *d_map = nullptr;
  int *d_bad = nullptr;
  double *d_val = nullptr;
  PA_TRY(sc.get(&d_rp, (size_t)n_rows + 1));
  PA_TRY(sc.get(&d_row, (size_t)nnz + 1));
  PA_TRY(sc.get(&d_col, (size_t)nnz + 1));
  PA_TRY(sc.get(&d_val, (size_t)nnz + 1));
  PA_TRY(sc.get(&d_map, (size_t)A->n_cols + 1));
  PA_TRY(sc.get(&d_bad, 1));
  PA_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), s));
  if (A->n_cols) PA_HIP(hipMemcpyAsync(d_map, map, sizeof(int32_t) * A->n_cols, hipMemcpyHostToDevice, s));
  if (nnz) {
    const double *d_aval = nullptr;
    PA_TRY(decode_block(A, sc, d_row, d_col, &d_aval));
    hipLaunchKernelGGL(kt_remap, grid1(nnz), dim3(256), 0, s, d_col, (int)nnz, d_map, d_bad);
    PA_HIP(hipMemcpyAsync(d_val, d_aval, sizeof(double) * nnz, hipMemcpyDeviceToDevice, s));
  }
  hipLaunchKernelGGL(kt_lower_bounds, grid1(n_rows + 1), dim3(256), 0, s, d_row, (int)nnz, (int)n_rows, d_rp);   // rows ascend in storage order
  int bad = 0;
  PA_TRY(d2h(s, &bad, d_bad, 1));
  PA_HIP(hipGetLastError());
  PA_REQUIRE(bad == 0, "%d stored entries sit in columns the map does not carry", bad);
  PA_TRY(pa_csr_from_device(c, n_rows, n_cols_new, nnz, d_rp, d_col, d_val, out));
  for (pa_csr *S = *out; S; S = S->next) S->alpha_inside = A->alpha_inside;   // (the twin of a CSC-made block keeps its 5-argument form)
  return PA_OK;
}

// ---- the same block with its rows and / or columns renumbered (round 4: library-side renumbering for blocks without locality) ----
// row_pos[i] = new position of row i (a permutation of 0..n_rows-1, host, or NULL = rows stay), col_pos[j] = new name of column j
// (host, or NULL).  Every row keeps its stored entries IN THEIR ORIGINAL ORDER -- the sort by new row is stable -- so a row's sum
// adds the same products in the same order: y_new[row_pos[i]] has the bits y[i] had when x_new[col_pos[j]] = x[j].
__global__ void kt_rename(int *__restrict__ row, int *__restrict__ col, int n, const int *__restrict__ row_pos, const int *__restrict__ col_pos) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  if (row_pos) row[k] = row_pos[row[k]];
  if (col_pos) col[k] = col_pos[col[k]];
}
__global__ void kt_gather2(const int *__restrict__ perm, const int *__restrict__ col, const double *__restrict__ val, int n,
                           int *__restrict__ out_col, double *__restrict__ out_val) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int p = perm[k];
  out_col[k] = col[p];
  out_val[k] = val[p];
}

extern "C" int pa_csr_create_permuted(const pa_csr *A, const int32_t *row_pos, const int32_t *col_pos, pa_csr **out) {
  PA_REQUIRE(A && out, "bad arguments");
  PA_REQUIRE(!A->next || A->colsplit, "a chain of slabs has no renumbered twin");
  pa_ctx *c = A->ctx;
  PA_REQUIRE(!c->capturing, "not inside a graph capture");
  PA_HIP(hipSetDevice(c->device));
  hipStream_t s = c->s[0];
  const int64_t nnz = block_nnz(A), n_rows = A->n_rows, n_cols = A->n_cols;
  if (row_pos) {                                               // a permutation, or rows would collide
    std::vector<char> seen((size_t)n_rows, 0);
    for (int64_t i = 0; i < n_rows; ++i) {
      PA_REQUIRE(row_pos[i] >= 0 && row_pos[i] < n_rows && !seen[row_pos[i]], "row_pos is not a permutation (row %lld)", (long long)i);
      seen[row_pos[i]] = 1;
    }
  }
  if (col_pos) for (int64_t j = 0; j < n_cols; ++j) PA_REQUIRE(col_pos[j] >= 0 && col_pos[j] < n_cols, "col_pos[%lld] out of range", (long long)j);
  scratch sc;
  int32_t *d_rp = nullptr, *d_row = nullptr, *d_col = nullptr, *d_rpos = nullptr, *d_cpos = nullptr;
  PA_TRY(sc.get(&d_rp, (size_t)n_rows + 1));
  PA_TRY(sc.get(&d_row, (size_t)nnz + 1));
  PA_TRY(sc.get(&d_col, (size_t)nnz + 1));
  if (row_pos) { PA_TRY(sc.get(&d_rpos, (size_t)n_rows + 1)); PA_HIP(hipMemcpyAsync(d_rpos, row_pos, sizeof(int32_t) * n_rows, hipMemcpyHostToDevice, s)); }
  if (col_pos) { PA_TRY(sc.get(&d_cpos, (size_t)n_cols + 1)); PA_HIP(hipMemcpyAsync(d_cpos, col_pos, sizeof(int32_t) * n_cols, hipMemcpyHostToDevice, s)); }
  const int32_t *d_fcol = d_col;
  const double *d_fval = A->d_val, *d_aval = A->d_val;
  int32_t *d_keys = nullptr, *d_iota = nullptr, *d_perm = nullptr, *d_tcol = nullptr;
  double *d_tval = nullptr;
  if (nnz) {
    PA_TRY(decode_block(A, sc, d_row, d_col, &d_aval));
    d_fval = d_aval;
    hipLaunchKernelGGL(kt_rename, grid1(nnz), dim3(256), 0, s, d_row, d_col, (int)nnz, d_rpos, d_cpos);
    if (row_pos) {
      PA_TRY(sc.get(&d_keys, (size_t)nnz));
      PA_TRY(sc.get(&d_iota, (size_t)nnz));
      PA_TRY(sc.get(&d_perm, (size_t)nnz));
      hipLaunchKernelGGL(kt_iota, grid1(nnz), dim3(256), 0, s, d_iota, (int)nnz);
      unsigned bits = 1;
      while (bits < 32 && ((int64_t)1 << bits) < n_rows) ++bits;
      PA_TRY(sort_pairs(sc, s, d_row, d_keys, d_iota, d_perm, (size_t)nnz, bits));
      PA_TRY(sc.get(&d_tcol, (size_t)nnz));
      PA_TRY(sc.get(&d_tval, (size_t)nnz));
      hipLaunchKernelGGL(kt_gather2, grid1(nnz), dim3(256), 0, s, d_perm, d_col, d_aval, (int)nnz, d_tcol, d_tval);
      d_fcol = d_tcol; d_fval = d_tval;
    }
  }
  hipLaunchKernelGGL(kt_lower_bounds, grid1(n_rows + 1), dim3(256), 0, s, row_pos && nnz ? d_keys : d_row, (int)nnz, (int)n_rows, d_rp);
  PA_HIP(hipGetLastError());
  PA_HIP(hipStreamSynchronize(s));
  PA_TRY(pa_csr_from_device(c, n_rows, n_cols, nnz, d_rp, d_fcol, d_fval, out));
  for (pa_csr *S = *out; S; S = S->next) S->alpha_inside = A->alpha_inside;
  return PA_OK;
}

// ---- a bandwidth-reducing order of a square block, computed on the device: reverse Cuthill-McKee by level sets -------------------
// A block whose rows gather x from all over (an unstructured mesh numbered as the mesher left it) gets 64 bytes of HBM traffic per
// gathered entry; in a Cuthill-McKee order the rows of a chunk share their columns with the chunks around them and the x windows of
// pa_spmv_xwin.h apply.  Breadth-first from a pseudo-peripheral vertex (the last vertex of a first sweep from vertex 0); a level's
// vertices are ordered by the position of their first-numbered neighbour in the level before, ties by vertex id (deterministic);
// every level is two frontier kernels, a count read-back and a radix sort of the level.  Unreached vertices (another component)
// start a new sweep at the lowest unnumbered id.  new_pos[v] = n - 1 - (Cuthill-McKee position of v).
__global__ void kr_claim(const int *__restrict__ rp, const int *__restrict__ col, const int *__restrict__ frontier, int f, int base,
                         const int *__restrict__ pos, int *__restrict__ key) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= f) return;
  const int u = frontier[i];
  for (int p = rp[u]; p < rp[u + 1]; ++p) {
    const int v = col[p];
    if (pos[v] < 0) atomicMin(&key[v], base + i);
  }
}
__global__ void kr_collect(const int *__restrict__ rp, const int *__restrict__ col, const int *__restrict__ frontier, int f, int base,
                           const int *__restrict__ pos, const int *__restrict__ key, int *__restrict__ count,
                           unsigned long long *__restrict__ ckey, int *__restrict__ cand) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= f) return;
  const int u = frontier[i];
  for (int p = rp[u]; p < rp[u + 1]; ++p) {
    const int v = col[p];
    if (v != u && pos[v] < 0 && key[v] == base + i) {           // (v's first-numbered neighbour is u: exactly one claim per vertex --
      bool first = true;                                        //  unless the row holds v twice, which CSR rows here do not)
      for (int q = rp[u]; q < p && first; ++q) first = col[q] != v;
      if (!first) continue;
      const int k = atomicAdd(count, 1);
      cand[k] = v;
      ckey[k] = ((unsigned long long)(unsigned)(base + i) << 32) | (unsigned)v;
    }
  }
}
__global__ void kr_number(const int *__restrict__ cand, int n, int base, int *__restrict__ pos) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) pos[cand[k]] = base + k;
}
__global__ void kr_first_unnumbered(const int *__restrict__ pos, int n, int *__restrict__ out) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n && pos[v] < 0) atomicMin(out, v);
}
__global__ void kr_band(const int *__restrict__ row, const int *__restrict__ col, int n, const int *__restrict__ newpos, int *__restrict__ out) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int d0 = abs(row[k] - col[k]);
  const int d1 = abs(newpos[row[k]] - newpos[col[k]]);
  if (d0 > out[0]) atomicMax(&out[0], d0);
  if (d1 > out[1]) atomicMax(&out[1], d1);
}
__global__ void kr_reverse(const int *__restrict__ pos, int n, int *__restrict__ newpos) {
  const int v = blockIdx.x * blockDim.x + threadIdx.x;
  if (v < n) newpos[v] = n - 1 - pos[v];
}

extern "C" int pa_csr_locality_order(const pa_csr *A, int32_t *new_pos, int64_t *band_before, int64_t *band_after) {
  PA_REQUIRE(A && new_pos, "bad arguments");
  PA_REQUIRE((!A->next || A->colsplit) && A->n_rows == A->n_cols, "a square single-slab block is needed");
  pa_ctx *c = A->ctx;
  PA_REQUIRE(!c->capturing, "not inside a graph capture");
  PA_HIP(hipSetDevice(c->device));
  hipStream_t s = c->s[0];
  const int64_t n = A->n_rows, nnz = block_nnz(A);
  if (n == 0) return PA_OK;
  scratch sc;
  int32_t *d_rp = nullptr, *d_row = nullptr, *d_col = nullptr, *d_pos = nullptr, *d_key = nullptr, *d_fa = nullptr, *d_fb = nullptr, *d_cand = nullptr,
          *d_small = nullptr, *d_newpos = nullptr;
  unsigned long long *d_ckey = nullptr, *d_ckey2 = nullptr;
  PA_TRY(sc.get(&d_rp, (size_t)n + 1));
  PA_TRY(sc.get(&d_row, (size_t)nnz + 1));
  PA_TRY(sc.get(&d_col, (size_t)nnz + 1));
  PA_TRY(sc.get(&d_pos, (size_t)n));
  PA_TRY(sc.get(&d_key, (size_t)n));
  PA_TRY(sc.get(&d_fa, (size_t)n));
  PA_TRY(sc.get(&d_fb, (size_t)n));
  PA_TRY(sc.get(&d_cand, (size_t)n));
  PA_TRY(sc.get(&d_ckey, (size_t)n));
  PA_TRY(sc.get(&d_ckey2, (size_t)n));
  PA_TRY(sc.get(&d_small, 4));
  PA_TRY(sc.get(&d_newpos, (size_t)n));
  const double *d_unused = nullptr;
  if (nnz) PA_TRY(decode_block(A, sc, d_row, d_col, &d_unused));
  hipLaunchKernelGGL(kt_lower_bounds, grid1(n + 1), dim3(256), 0, s, d_row, (int)nnz, (int)n, d_rp);
  size_t sort_tb = 0;
  PA_HIP(rocprim::radix_sort_pairs((void *)nullptr, sort_tb, d_ckey, d_ckey2, d_cand, d_fb, (size_t)n, 0, 64, s));
  char *d_sort_tmp = nullptr;
  PA_TRY(sc.get(&d_sort_tmp, sort_tb));
  unsigned pos_bits = 1;
  while (pos_bits < 32 && ((int64_t)1 << pos_bits) < n) ++pos_bits;
  int last_vertex = 0;
  // one Cuthill-McKee sweep over the whole block starting at `seed`; leaves positions in d_pos, the last numbered vertex in last_vertex
  auto sweep = [&](int seed) -> int {
    PA_HIP(hipMemsetAsync(d_pos, 0xFF, sizeof(int32_t) * n, s));             // -1
    PA_HIP(hipMemsetAsync(d_key, 0x7F, sizeof(int32_t) * n, s));             // 0x7F7F7F7F: larger than any position
    int base = 0, f = 1;
    int32_t *F = d_fa, *Fn = d_fb;
    PA_HIP(hipMemcpyAsync(F, &seed, sizeof(int), hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(kr_number, dim3(1), dim3(64), 0, s, F, 1, 0, d_pos);
    int numbered = 1;
    last_vertex = seed;
    while (numbered < n) {
      PA_HIP(hipMemsetAsync(d_small, 0, sizeof(int32_t), s));
      hipLaunchKernelGGL(kr_claim, grid1(f), dim3(256), 0, s, d_rp, d_col, F, f, base, d_pos, d_key);
      hipLaunchKernelGGL(kr_collect, grid1(f), dim3(256), 0, s, d_rp, d_col, F, f, base, d_pos, d_key, d_small, d_ckey, d_cand);
      int cnt = 0;
      PA_TRY(d2h(s, &cnt, d_small, 1));
      if (cnt == 0) {                                           // this component is numbered: the lowest unnumbered vertex starts the next
        int first = 0x7FFFFFFF;
        PA_HIP(hipMemcpyAsync(d_small + 1, &first, sizeof(int), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(kr_first_unnumbered, grid1(n), dim3(256), 0, s, d_pos, (int)n, d_small + 1);
        PA_TRY(d2h(s, &first, d_small + 1, 1));
        PA_REQUIRE(first >= 0 && first < n, "Cuthill-McKee: no unnumbered vertex left although %d of %lld are numbered", numbered, (long long)n);
        base = numbered; f = 1;
        PA_HIP(hipMemcpyAsync(F, &first, sizeof(int), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(kr_number, dim3(1), dim3(64), 0, s, F, 1, base, d_pos);
        numbered += 1;
        last_vertex = first;
        continue;
      }
      PA_HIP(rocprim::radix_sort_pairs((void *)d_sort_tmp, sort_tb, d_ckey, d_ckey2, d_cand, Fn, (size_t)cnt, 0, 32 + pos_bits, s));
      base += f;
      hipLaunchKernelGGL(kr_number, grid1(cnt), dim3(256), 0, s, Fn, cnt, base, d_pos);
      std::swap(F, Fn);
      f = cnt;
      numbered += cnt;
    }
    PA_TRY(d2h(s, &last_vertex, F + (f - 1), 1));
    PA_HIP(hipGetLastError());
    return PA_OK;
  };
  PA_TRY(sweep(0));
  PA_TRY(sweep(last_vertex));                                   // from a pseudo-peripheral vertex: narrower levels
  hipLaunchKernelGGL(kr_reverse, grid1(n), dim3(256), 0, s, d_pos, (int)n, d_newpos);
  int band[2] = {0, 0};
  PA_HIP(hipMemcpyAsync(d_small, band, 2 * sizeof(int), hipMemcpyHostToDevice, s));
  if (nnz) hipLaunchKernelGGL(kr_band, grid1(nnz), dim3(256), 0, s, d_row, d_col, (int)nnz, d_newpos, d_small);
  PA_TRY(d2h(s, band, d_small, 2));
  PA_TRY(d2h(s, new_pos, d_newpos, (size_t)n));
  PA_HIP(hipGetLastError());
  if (band_before) *band_before = band[0];
  if (band_after) *band_after = band[1];
  return PA_OK;
}

// ---- column split of a wide band (round 4) --------------------------------------------------------------------
// The sliding x window (k_spmv_xring) holds 16384 entries of x: rows whose columns spread over more than that (+-8000 around the
// diagonal) fall back to the plain row split and its one L2 line per gather (3.0 TB/s algorithmic).  Such a block is cut into k
// COLUMN PIECES: entry (r, c) goes to piece floor((c - t0(r)) / w), t0(r) = the lower edge of the band at row r; every piece holds
// all rows, is a band of width w <= ~11000 that the window holds, and is a block of its own (row split, 16-bit windows, ring
// groups).  The product runs the pieces one after the other, the first with the caller's beta, the others accumulating: a row's
// columns ascend, so the pieces take consecutive runs of its entries and the sum adds the same products in the same order (the
// intermediate y is a stored fp64: exact).  Costs y read and written k times and the pieces' x ranges read once each.
// (Rows whose columns do NOT ascend in storage order -- renamed twins -- see the running maximum in pa_csr_colsplit_if_wide.)
__global__ void kt_piece_keys(const int *__restrict__ row, const int *__restrict__ col, int n, double slope, int half, int w, int k,
                              int *__restrict__ key) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const long long t0 = (long long)((double)row[p] * slope) - half;
  long long j = ((long long)col[p] - t0) / w;
  key[p] = (int)(j < 0 ? 0 : j >= k ? k - 1 : j);
}
// entries whose piece lies below that of the entry before them in the same row (rows whose columns do not ascend)
__global__ void kt_count_descents(const int *__restrict__ row, const int *__restrict__ key, int n, int *__restrict__ out) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < 1 || p >= n) return;
  if (row[p] == row[p - 1] && key[p] < key[p - 1]) atomicAdd(out, 1);
}
__global__ void kt_take(const int *__restrict__ perm, int first, int n, const int *__restrict__ row, const int *__restrict__ col,
                        const double *__restrict__ val, int *__restrict__ out_row, int *__restrict__ out_col, double *__restrict__ out_val,
                        int *__restrict__ out_src) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n) return;
  const int p = perm[first + q];
  out_row[q] = row[p]; out_col[q] = col[p]; out_val[q] = val[p]; out_src[q] = p;
}

int pa_csr_colsplit_if_wide(const pa_csr *A, pa_csr **out, int force_pieces) {
  *out = nullptr;
  if (!A || A->next || A->nnz == 0) return PA_OK;
  const char *e = getenv("PA_SPMV_COLSPLIT");
  const int mode = e ? atoi(e) : 1;
  const int64_t window = PA_XR_CAP - 64;
  int k = force_pieces;
  if (k <= 0 && mode >= 2 && mode <= 8) {
    if (A->nnz >= 64) k = mode;                      // (fuzzers: every block of 64 entries or more becomes a chain of `mode` pieces)
  } else if (k <= 0) {
    if (mode == 0 || A->use_pattern || A->compact || !A->use_c16 || A->n_xw_groups > 0 || A->nnz < ((int64_t)1 << 21)) return PA_OK;
    if (A->ctx->keep_raw_columns) return PA_OK;        // (the caller is about to run the row-selection set-up on this block: whole blocks only)
    if (A->xw_max_span <= window || A->xw_max_span > 4 * 15000) return PA_OK;
    k = (int)((A->xw_max_span + 14999) / 15000);
  }
  if (k < 2) return PA_OK;
  pa_ctx *c = A->ctx;
  if (c->capturing) return PA_OK;
  PA_HIP(hipSetDevice(c->device));
  hipStream_t s = c->s[0];
  const int64_t nnz = A->nnz, n_rows = A->n_rows, n_cols = A->n_cols;
  const int64_t span = std::max<int64_t>(A->xw_max_span > 0 ? A->xw_max_span : n_cols, k);
  const int w = (int)((span + k - 1) / k), half = (int)(span / 2);
  const double slope = n_rows > 1 ? (double)(n_cols - 1) / (double)(n_rows - 1) : 0.0;
  scratch sc;
  int32_t *d_row = nullptr, *d_col = nullptr, *d_key = nullptr, *d_ks = nullptr, *d_iota = nullptr, *d_perm = nullptr, *d_first = nullptr;
  PA_TRY(sc.get(&d_row, (size_t)nnz));
  PA_TRY(sc.get(&d_col, (size_t)nnz));
  PA_TRY(sc.get(&d_key, (size_t)nnz));
  PA_TRY(sc.get(&d_ks, (size_t)nnz));
  PA_TRY(sc.get(&d_iota, (size_t)nnz));
  PA_TRY(sc.get(&d_perm, (size_t)nnz));
  PA_TRY(sc.get(&d_first, (size_t)k + 2));
  PA_TRY(pa_dev_decode_entries(A, d_row, d_col));
  hipLaunchKernelGGL(kt_piece_keys, grid1(nnz), dim3(256), 0, s, d_row, d_col, (int)nnz, slope, half, w, k, d_key);
  if (force_pieces <= 0 && mode == 1) {
    // (an automatic split of rows whose columns do not ascend buys nothing: the running maximum below sweeps most of such a row into
    // its last piece, which is then as wide as the block was)
    int *d_cnt = nullptr, cnt = 0;
    PA_TRY(sc.get(&d_cnt, 1));
    PA_HIP(hipMemsetAsync(d_cnt, 0, sizeof(int), s));
    hipLaunchKernelGGL(kt_count_descents, grid1(nnz), dim3(256), 0, s, d_row, d_key, (int)nnz, d_cnt);
    PA_TRY(d2h(s, &cnt, d_cnt, 1));
    if ((int64_t)cnt * 100 > nnz) return PA_OK;
  }
  {
    // The pieces must take CONSECUTIVE runs of a row's entries, or the row's sum changes its order.  A row whose columns ascend
    // has that by itself; a twin with renamed columns (own x ghost reading the receive buffer, a renumbered block: entries in the
    // caller's order, columns anywhere) does not -- an entry goes to the highest piece any entry before it in the row went to (a
    // running maximum along the row; such a piece may then be wider than the window and its chunks fall to the row split).
    size_t tb = 0;
    PA_HIP(rocprim::inclusive_scan_by_key((void *)nullptr, tb, d_row, d_key, d_key, (size_t)nnz, rocprim::maximum<int>(), rocprim::equal_to<int>(), s));
    char *tmp = nullptr;
    PA_TRY(sc.get(&tmp, tb));
    PA_HIP(rocprim::inclusive_scan_by_key((void *)tmp, tb, d_row, d_key, d_key, (size_t)nnz, rocprim::maximum<int>(), rocprim::equal_to<int>(), s));
    PA_HIP(hipStreamSynchronize(s));
    sc.release(tmp);
  }
  hipLaunchKernelGGL(kt_iota, grid1(nnz), dim3(256), 0, s, d_iota, (int)nnz);
  PA_TRY(sort_pairs(sc, s, d_key, d_ks, d_iota, d_perm, (size_t)nnz, 3));      // stable: inside a piece the entries keep their (row, column) order
  hipLaunchKernelGGL(kt_lower_bounds, grid1(k + 1), dim3(256), 0, s, d_ks, (int)nnz, k, d_first);
  std::vector<int32_t> first((size_t)k + 1);
  PA_TRY(d2h(s, first.data(), d_first, (size_t)k + 1));
  PA_HIP(hipGetLastError());
  sc.release(d_key); sc.release(d_ks); sc.release(d_iota);
  pa_csr *head = nullptr, *tail = nullptr;
  // rows at which EVERY piece ends a chunk and a ring group (the chain then runs as one launch, k_spmv_xring_chain): equal shares of
  // the block's entries, on multiples of 8 rows (64 bytes of y)
  std::vector<int32_t> breaks;
  if (c->sw.chain_fused && n_rows + 1 < ((int64_t)1 << 31)) {
    std::vector<int32_t> hrp((size_t)n_rows + 1);
    PA_TRY(d2h(s, hrp.data(), A->d_crp, (size_t)n_rows + 1));
    static const int64_t want = getenv("PA_SPMV_XRING_GROUPS") ? std::max(1, atoi(getenv("PA_SPMV_XRING_GROUPS"))) : PA_XR_WANT_GROUPS;
    const int64_t piece_chunks = std::max<int64_t>(1, nnz / k / PA_SPMV_CHUNK_NNZ);
    const int64_t G = std::max<int64_t>(std::min<int64_t>(want, piece_chunks / 4), (piece_chunks + 799) / 800);
    breaks.push_back(0);
    for (int64_t g = 1; g < G; ++g) {
      int64_t r = std::lower_bound(hrp.begin(), hrp.end(), (int32_t)(nnz * g / G)) - hrp.begin();
      r &= ~(int64_t)7;
      if (r > breaks.back() && r < n_rows) breaks.push_back((int32_t)r);
    }
    breaks.push_back((int32_t)n_rows);
  }
  struct tls_guard { ~tls_guard() { pa_tls_row_breaks = nullptr; } } guard;
  pa_tls_row_breaks = breaks.size() > 1 ? &breaks : nullptr;
  auto fail = [&](int st) { if (head) pa_csr_destroy(head); return st; };
  for (int j = 0; j < k; ++j) {
    const int64_t cnt = (int64_t)first[j + 1] - first[j];
    scratch pc;
    int32_t *p_row = nullptr, *p_col = nullptr, *p_src = nullptr, *p_rp = nullptr;
    double *p_val = nullptr;
    if (int st = pc.get(&p_row, (size_t)cnt + 1)) return fail(st);
    if (int st = pc.get(&p_col, (size_t)cnt + 1)) return fail(st);
    if (int st = pc.get(&p_val, (size_t)cnt + 1)) return fail(st);
    if (int st = pc.get(&p_src, (size_t)cnt + 1)) return fail(st);
    if (int st = pc.get(&p_rp, (size_t)n_rows + 1)) return fail(st);
    if (cnt) hipLaunchKernelGGL(kt_take, grid1(cnt), dim3(256), 0, s, d_perm, first[j], (int)cnt, d_row, d_col, A->d_val, p_row, p_col, p_val, p_src);
    hipLaunchKernelGGL(kt_lower_bounds, grid1(n_rows + 1), dim3(256), 0, s, p_row, (int)cnt, (int)n_rows, p_rp);
    if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) { pa_set_err("column split: a kernel failed"); return fail(PA_ERR_HIP); }
    pa_csr *P = nullptr;
    ++pa_tls_piece_build;
    const int st_p = pa_csr_from_device(c, n_rows, n_cols, cnt, p_rp, p_col, p_val, &P);
    --pa_tls_piece_build;
    if (st_p) return fail(st_p);
    if (P->next) { pa_csr_destroy(P); pa_set_err("column split: a piece came out as a chain"); return fail(PA_ERR_ARG); }
    if (int st = pa_dev_alloc(c, (void **)&P->d_src, sizeof(int32_t) * (size_t)std::max<int64_t>(cnt, 1), PA_MEM_MATRIX)) { pa_csr_destroy(P); return fail(st); }
    if (cnt && hipMemcpyAsync(P->d_src, p_src, sizeof(int32_t) * (size_t)cnt, hipMemcpyDeviceToDevice, s) != hipSuccess) { pa_csr_destroy(P); return fail(PA_ERR_HIP); }
    if (hipStreamSynchronize(s) != hipSuccess) { pa_csr_destroy(P); return fail(PA_ERR_HIP); }
    P->accumulate = j > 0;
    P->alpha_inside = A->alpha_inside;
    P->row0 = 0; P->nnz0 = first[j];
    if (tail) tail->next = P; else head = P;
    tail = P;
  }
  head->colsplit = true;
  head->t_rows = A->t_rows; head->t_nnz = A->t_nnz;
  head->xw_max_span = A->xw_max_span;
  // one launch for the chain when every piece is ring groups only and group g of every piece starts on the same row
  if (breaks.size() > 1) {
    bool aligned = true;
    int64_t ng = -1;
    std::vector<int32_t> rows0;
    std::vector<pa_chain_piece> tab;
    for (pa_csr *P = head; P && aligned; P = P->next) {
      aligned = !P->compact && P->n_xw_ring > 0 && P->n_xw_groups == P->n_xw_ring && P->n_xw_rest == 0 && (ng < 0 || P->n_xw_ring == ng);
      if (!aligned) break;
      ng = P->n_xw_ring;
      std::vector<pa_xw_group> g((size_t)ng);
      std::vector<int32_t> cr((size_t)P->n_chunks + 1), r0((size_t)ng);
      if (int st = d2h(s, (char *)g.data(), (const char *)P->d_xw_grp, sizeof(pa_xw_group) * (size_t)ng)) return fail(st);
      if (int st = d2h(s, cr.data(), P->d_chunk_row, (size_t)P->n_chunks + 1)) return fail(st);
      int64_t covered = 0;
      for (int64_t i = 0; i < ng; ++i) {
        r0[i] = cr[g[i].first];
        covered += g[i].cnt;
        if (i > 0 && g[i].first != g[i - 1].first + g[i - 1].cnt) aligned = false;
      }
      if (covered != P->n_chunks || (ng > 0 && g[0].first != 0)) aligned = false;
      if (rows0.empty()) rows0.swap(r0);
      else if (rows0 != r0) aligned = false;
      auto adr = [](const void *q) { return (unsigned long long)(uintptr_t)q; };
      tab.push_back(pa_chain_piece{adr(P->d_crp), adr(P->d_col16), adr(P->d_win), adr(P->d_val), adr(P->d_chunk_row), adr(P->d_chunk_p),
                                   adr(P->d_chunk_cmax), adr(P->d_xw_grp)});
    }
    if (aligned && ng > 0 && (int)tab.size() == k) {
      if (int st = pa_dev_alloc(c, &head->d_chain, sizeof(pa_chain_piece) * tab.size(), PA_MEM_MATRIX)) return fail(st);
      if (hipMemcpyAsync(head->d_chain, tab.data(), sizeof(pa_chain_piece) * tab.size(), hipMemcpyHostToDevice, s) != hipSuccess ||
          hipStreamSynchronize(s) != hipSuccess) { pa_set_err("column split: the chain table did not upload"); return fail(PA_ERR_HIP); }
      head->chain_groups = ng;
      head->chain_pieces = k;
    }
    if (getenv("PA_SETUP_TIMING")) fprintf(stderr, "[pa setup] column split: %d pieces, %lld row breaks, %s\n", k, (long long)breaks.size() - 1,
                                            head->d_chain ? "one launch" : "a launch per piece");
  }
  *out = head;
  return PA_OK;
}

// the block as a chain of `pieces` column pieces whatever its spans (tests; NULL out when it cannot be split)
extern "C" int pa_csr_create_colsplit(const pa_csr *A, int pieces, pa_csr **out) {
  PA_REQUIRE(A && out && pieces >= 2 && pieces <= 8, "bad arguments");
  PA_REQUIRE(!A->next, "the block is a chain already");
  PA_TRY(pa_csr_colsplit_if_wide(A, out, pieces));
  PA_REQUIRE(*out != nullptr, "the block could not be split");
  return PA_OK;
}

extern "C" int pa_csr_chain_info(const pa_csr *A, int32_t *pieces, int64_t *groups_one_launch) {
  PA_REQUIRE(A && pieces && groups_one_launch, "bad arguments");
  *pieces = 0; *groups_one_launch = 0;
  if (!A->colsplit) return PA_OK;
  for (const pa_csr *S = A; S; S = S->next) ++*pieces;
  if (A->d_chain && A->ctx->sw.chain_fused) *groups_one_launch = A->chain_groups;
  return PA_OK;
}

// ---- mul!(c, transpose(a), b, alpha, beta) of one part / of all parts of a process ------------------------------------------
// src/p_sparse_matrix.jl:2144-2162.  c lives on axes(a,2) (own + ghost columns), b on axes(a,1); the matrix must be assembled.
extern "C" int pa_matrix_create_transposed(pa_ctx *c, const pa_csr *own_own_t, const pa_csr *own_ghost_t, pa_plan *col_plan, pa_matrix **out) {
  PA_REQUIRE(c && own_own_t && own_ghost_t && col_plan && out, "bad arguments");
  PA_REQUIRE(own_own_t->ctx == c && own_ghost_t->ctx == c && col_plan->ctx == c, "operands live in different contexts");
  // own_own_t = A_oo' (own columns x own rows), own_ghost_t = A_oh' (ghost columns x own rows): pa_csr_create_transpose of a's blocks
  PA_REQUIRE(own_own_t->n_cols == own_ghost_t->n_cols, "A_oo' has %lld columns, A_oh' %lld (both: the own rows of a)", (long long)own_own_t->n_cols,
             (long long)own_ghost_t->n_cols);
  PA_REQUIRE(own_own_t->t_rows + own_ghost_t->t_rows == col_plan->n_local, "the transposed blocks have %lld + %lld rows, the column plan %lld local ids",
             (long long)own_own_t->t_rows, (long long)own_ghost_t->t_rows, (long long)col_plan->n_local);
  pa_matrix *m = new pa_matrix();
  m->ctx = c; m->oo = own_own_t; m->oh = own_ghost_t; m->plan = col_plan;   // (blocks and plan stay the caller's, as for pa_matrix_create)
  m->transposed = true;
  *out = m;
  return PA_OK;
}

static int mul_t_check(const pa_matrix *m, const pa_vec *c, const pa_vec *b) {
  PA_REQUIRE(m && c && b, "bad arguments");
  PA_REQUIRE(m->transposed, "not a transposed matrix handle (pa_matrix_create_transposed)");
  // oo = A_oo' (n_own_cols x n_own_rows), oh = A_oh' (n_ghost_cols x n_own_rows)
  PA_REQUIRE(c->n_own == m->oo->t_rows && c->n_ghost == m->oh->t_rows, "c does not live on axes(a,2)");
  PA_REQUIRE(b->n_own == m->oo->n_cols, "matching_own_indices(axes(a,1),axes(b,1)) failed");
  PA_REQUIRE(c->d != b->d, "c and b alias");
  return PA_OK;
}

extern "C" int pa_mul5_transpose(pa_matrix *m, pa_comm *comm, pa_vec *c, pa_vec *b, double alpha, double beta) {
  PA_TRY(mul_t_check(m, c, b));
  PA_TRY(pa_spmv(m->oh, b, PA_SEG_OWN, c, PA_SEG_GHOST, alpha, 0.0));        // fill!(ch,0); mul!(ch,atoh,bo,alpha,1)
  PA_TRY(pa_exchange_start(m->plan, comm, c, PA_ASSEMBLE));                   // t = assemble!(c)
  PA_TRY(pa_spmv(m->oo, b, PA_SEG_OWN, c, PA_SEG_OWN, alpha, beta));         // rmul!(co,beta); mul!(co,atoo,bo,alpha,1): overlaps
  PA_TRY(pa_exchange_finish(m->plan, c, PA_ASSEMBLE));                        // wait(t): owners += ghost contributions; ghosts := 0
  return PA_OK;
}

extern "C" int pa_mul5_transpose_all(pa_matrix *const *m, int32_t n_parts, pa_vec *const *c, pa_vec *const *b, double alpha, double beta) {
  PA_REQUIRE(m && c && b && n_parts > 0, "bad arguments");
  std::vector<pa_plan *> plans(n_parts);
  for (int r = 0; r < n_parts; ++r) {
    PA_TRY(mul_t_check(m[r], c[r], b[r]));
    plans[r] = m[r]->plan;
  }
  for (int r = 0; r < n_parts; ++r) PA_TRY(pa_spmv(m[r]->oh, b[r], PA_SEG_OWN, c[r], PA_SEG_GHOST, alpha, 0.0));
  const int push = getenv("PA_PUSH") ? atoi(getenv("PA_PUSH")) : 1;      // (read per call: tests switch it)
  if (push) PA_TRY(pa_exchange_push_local(plans.data(), n_parts, c, PA_ASSEMBLE));   // one launch packs and delivers every part's ghosts
  else {
    for (int r = 0; r < n_parts; ++r) PA_TRY(pa_exchange_pack(plans[r], c[r], PA_ASSEMBLE));
    PA_TRY(pa_exchange_local(plans.data(), n_parts, PA_ASSEMBLE));
  }
  for (int r = 0; r < n_parts; ++r) PA_TRY(pa_spmv(m[r]->oo, b[r], PA_SEG_OWN, c[r], PA_SEG_OWN, alpha, beta));
  if (push) return pa_exchange_finish_all(plans.data(), n_parts, c, PA_ASSEMBLE);     // one launch: every part's ordered adds + ghost zeroing
  for (int r = 0; r < n_parts; ++r) PA_TRY(pa_exchange_finish(plans[r], c[r], PA_ASSEMBLE));
  return PA_OK;
}
