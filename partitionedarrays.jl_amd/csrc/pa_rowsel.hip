// pa_rowsel.hip -- blocks made of SOME ROWS of a part's split matrix, built on the device from the blocks already in HBM
// ("the colour split as kernels over the uploaded CSR").
//
// The multigrid set-up of the HPCG driver needs, per level, nine row subsets of the level's matrix: the eight colours of
// the multicolour Gauss-Seidel smoother (the sweep of PartitionedSolvers/src/smoothers.jl:98-176 written as SpMV + update)
// and the fine rows the coarse grid keeps (the residual that restrict! reads, HPCG/src/mg_preconditioner.jl:224-251,314-329).
// Each is an n_own x n_local block in the UNSPLIT column order HPCG stores (own columns, then ghost columns shifted by the
// number of own columns) whose other rows are empty.  Until round 3 the host copied the rows (pa_host_color_split) and every
// subset went over PCIe again: 6 GB for a 256^3 part, 1.7 of the 3.4 s of pc_setup.  Here the part's own|own and own|ghost
// blocks keep their raw Int32 columns in HBM while the set-up lasts (pa_ctx_keep_raw_columns), and a subset is
//     lengths of the selected rows -> exclusive scan -> one lane per row copies its entries (own block first, then ghost)
// handed to the same block constructor as an uploaded matrix (pa_csr_from_device: row split and column encodings on the
// device).  Entry order inside a row = the host route's, so the blocks are the host route's, array for array
// (tests/test_gpu_setup.py::test_device_side_row_subsets_equal_the_host_route).
#include "pa_dev_util.h"

#include <chrono>

#include "pa_setup.h"

using namespace pa_util;

extern thread_local const pa_csr *pa_tls_vdict_parent;   // pa_csr.hip

// per row of the (possibly row-compacted) block: where its entries start and how many there are
__global__ void kr_spans(const int32_t *__restrict__ crp, const int32_t *__restrict__ row_ids, int nc, int32_t *__restrict__ start,
                         int32_t *__restrict__ len) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc) return;
  const int r = row_ids ? row_ids[c] : c;
  start[r] = crp[c];
  len[r] = crp[c + 1] - crp[c];
}

// stored entries of every subset (64-bit: the sum over a subset must be checked against Int32 row pointers, not wrap)
__global__ void kr_totals(const int32_t *__restrict__ mask, const int32_t *__restrict__ len_a, const int32_t *__restrict__ len_b, int n,
                          int n_sel, unsigned long long *__restrict__ tot) {
  __shared__ unsigned long long h[64];
  if (threadIdx.x < 64) h[threadIdx.x] = 0;
  __syncthreads();
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) {
    const int k = mask[r];
    if (k >= 0 && k < n_sel) atomicAdd(&h[k], (unsigned long long)(len_a[r] + (len_b ? len_b[r] : 0)));
  }
  __syncthreads();
  if (threadIdx.x < n_sel && h[threadIdx.x]) atomicAdd(&tot[threadIdx.x], h[threadIdx.x]);
}

__global__ void kr_bad_mask(const int32_t *__restrict__ mask, int n, int n_sel, int *__restrict__ bad) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n && (mask[r] < -1 || mask[r] >= n_sel)) *bad = 1;
}

__global__ void kr_len(const int32_t *__restrict__ mask, int k, const int32_t *__restrict__ len_a, const int32_t *__restrict__ len_b,
                       int n, int32_t *__restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n) return;
  out[r] = (r < n && mask[r] == k) ? len_a[r] + (len_b ? len_b[r] : 0) : 0;
}

// 8 lanes per selected row: lane j copies entries j, j + 8, ... (the own block's, then the ghost block's shifted)
__global__ void kr_fill(const int32_t *__restrict__ mask, int k, int n, const int32_t *__restrict__ rp_out,
                        const int32_t *__restrict__ start_a, const int32_t *__restrict__ len_a, const int32_t *__restrict__ col_a,
                        const double *__restrict__ val_a, const int32_t *__restrict__ start_b, const int32_t *__restrict__ len_b,
                        const int32_t *__restrict__ col_b, const double *__restrict__ val_b, int shift_b, int32_t *__restrict__ col_out,
                        double *__restrict__ val_out) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int r = (int)(t >> 3), j = (int)(t & 7);
  if (r >= n || mask[r] != k) return;
  const int dst = rp_out[r], la = len_a[r], sa = start_a[r];
  for (int e = j; e < la; e += 8) {
    col_out[dst + e] = col_a[sa + e];
    val_out[dst + e] = val_a[sa + e];
  }
  if (len_b) {
    const int lb = len_b[r], sb = start_b[r];
    for (int e = j; e < lb; e += 8) {
      col_out[dst + la + e] = col_b[sb + e] + shift_b;
      val_out[dst + la + e] = val_b[sb + e];
    }
  }
}

// the same for the entries of a selected row whose (own) column has a LOWER mask value than the row (mask in 0..k-1): what a
// forward sweep over x == 0 can see when it reaches colour k.  One lane per row: the entries keep their order.
__global__ void kr_len_lower(const int32_t *__restrict__ mask, int k, const int32_t *__restrict__ start, const int32_t *__restrict__ len,
                             const int32_t *__restrict__ col, int n, int32_t *__restrict__ out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n) return;
  int cnt = 0;
  if (r < n && mask[r] == k)
    for (int p = start[r], e = p + len[r]; p < e; ++p) {
      const int m = mask[col[p]];
      cnt += m >= 0 && m < k;
    }
  out[r] = cnt;
}

__global__ void kr_fill_lower(const int32_t *__restrict__ mask, int k, int n, const int32_t *__restrict__ rp_out,
                              const int32_t *__restrict__ start, const int32_t *__restrict__ len, const int32_t *__restrict__ col,
                              const double *__restrict__ val, int32_t *__restrict__ col_out, double *__restrict__ val_out) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n || mask[r] != k) return;
  int dst = rp_out[r];
  for (int p = start[r], e = p + len[r]; p < e; ++p) {
    const int m = mask[col[p]];
    if (m >= 0 && m < k) {
      col_out[dst] = col[p];
      val_out[dst] = val[p];
      ++dst;
    }
  }
}

__global__ void kr_flag(const int32_t *__restrict__ len, int n, int32_t *__restrict__ flag) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r <= n) flag[r] = r < n && len[r] > 0;
}

// the non-empty rows of a subset, in order: their ids and their row pointer (csr_fill_slab's compaction, on the device)
__global__ void kr_compact(const int32_t *__restrict__ len, const int32_t *__restrict__ pos, const int32_t *__restrict__ rp, int n,
                           int32_t *__restrict__ row_ids, int32_t *__restrict__ crp) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n) return;
  if (r == n) { crp[pos[n]] = rp[n]; return; }
  if (len[r] > 0) {
    row_ids[pos[r]] = r;
    crp[pos[r]] = rp[r];
  }
}

__global__ void kr_diag(const int32_t *__restrict__ start, const int32_t *__restrict__ len, const int32_t *__restrict__ col,
                        const double *__restrict__ val, int n, double *__restrict__ d) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  double v = 0.0;
  for (int p = start[r], e = p + len[r]; p < e; ++p)
    if (col[p] == r) v = val[p];                       // (pa_host_color_split: the last stored (r,r) entry)
  d[r] = v;
}

static const int32_t *raw_columns(const pa_csr *A) {
  if (A->d_raw_col) return A->d_raw_col;
  return A->n_col32 == A->nnz ? A->d_col : nullptr;    // (a block without compacted streams holds them all anyway)
}

extern "C" int pa_ctx_keep_raw_columns(pa_ctx *c, int on) {
  PA_REQUIRE(c != nullptr, "ctx is NULL");
  c->keep_raw_columns = on != 0;
  return PA_OK;
}

extern "C" int pa_csr_has_raw_columns(const pa_csr *A, int *yes) {
  PA_REQUIRE(A && yes, "bad arguments");
  *yes = !A->next && (A->nnz == 0 || raw_columns(A) != nullptr);
  return PA_OK;
}

extern "C" int pa_csr_drop_raw_columns(pa_csr *A) {
  PA_REQUIRE(A != nullptr, "block is NULL");
  for (pa_csr *S = A; S; S = S->next)
    if (S->d_raw_col) {
      PA_HIP(hipSetDevice(S->ctx->device));
      PA_HIP(hipStreamSynchronize(S->ctx->s[0]));
      pa_dev_free(S->ctx, S->d_raw_col);
      S->d_raw_col = nullptr;
    }
  return PA_OK;
}

struct row_spans {
  int32_t *start = nullptr, *len = nullptr;
};

static int spans_of(pa_ctx *c, scratch &sc, const pa_csr *A, int64_t n, row_spans &S) {
  PA_TRY(sc.get(&S.start, (size_t)n + 1));
  PA_TRY(sc.get(&S.len, (size_t)n + 1));
  PA_HIP(hipMemsetAsync(S.start, 0, sizeof(int32_t) * (n + 1), c->s[0]));
  PA_HIP(hipMemsetAsync(S.len, 0, sizeof(int32_t) * (n + 1), c->s[0]));
  if (A->n_crows > 0)
    hipLaunchKernelGGL(kr_spans, grid1(A->n_crows), dim3(256), 0, c->s[0], A->d_crp, A->d_row_ids, (int)A->n_crows, S.start, S.len);
  PA_HIP(hipGetLastError());
  return PA_OK;
}

// out[k] (k = 0..n_sel-1) = the rows r of the part with mask[r] == k (mask: n_rows host entries in -1..n_sel-1; -1 = in no
// block), columns: oo's, then oh's shifted by oo's column count.  oh may be NULL (a part without ghost columns).
static int select_rows_impl(const pa_csr *oo, const pa_csr *oh, const int32_t *mask, int32_t n_sel, pa_csr **out, bool lower,
                            int64_t n_cols_total) {
  PA_REQUIRE(oo && mask && out && n_sel > 0 && n_sel <= 64, "bad arguments");
  PA_REQUIRE(!lower || oo->n_rows == oo->n_cols, "the own|own block is not square");
  PA_REQUIRE(!oo->next && !(oh && oh->next), "a block stored as a chain (2^31 stored entries or more: row slabs; a band beyond the sliding x window: column pieces, PA_SPMV_COLSPLIT=0 keeps it whole) takes the host route");
  PA_REQUIRE(!oh || (oh->n_rows == oo->n_rows && oh->ctx == oo->ctx), "the own|ghost block does not match the own|own block");
  pa_ctx *c = oo->ctx;
  const int64_t n_ghost_cols = oh ? oh->n_cols : 0;      // (counted before an entry-less own|ghost block is dropped below)
  if (oh && oh->nnz == 0) oh = nullptr;
  const int32_t *col_a = raw_columns(oo), *col_b = oh ? raw_columns(oh) : nullptr;
  PA_REQUIRE(oo->nnz == 0 || col_a, "the own|own block did not keep its raw columns (create it under pa_ctx_keep_raw_columns)");
  PA_REQUIRE(!oh || col_b, "the own|ghost block did not keep its raw columns (create it under pa_ctx_keep_raw_columns)");
  const int64_t n = oo->n_rows, n_cols = n_cols_total >= 0 ? n_cols_total : oo->n_cols + n_ghost_cols;
  PA_REQUIRE(n_cols >= oo->n_cols, "fewer columns than the own|own block has");
  PA_REQUIRE(oo->nnz + (oh ? oh->nnz : 0) < (int64_t)2147483000 && n_cols < (int64_t)2147483000, "too large for Int32 offsets");
  for (int k = 0; k < n_sel; ++k) out[k] = nullptr;
  const auto t_begin = std::chrono::steady_clock::now();
  PA_HIP(hipSetDevice(c->device));
  hipStream_t s = c->s[0];
  scratch sc;
  row_spans A, B;
  PA_TRY(spans_of(c, sc, oo, n, A));
  if (oh) PA_TRY(spans_of(c, sc, oh, n, B));
  int32_t *d_mask = nullptr, *d_len = nullptr, *d_rp = nullptr, *d_flag = nullptr, *d_pos = nullptr, *d_ids = nullptr, *d_crp = nullptr;
  int *d_bad = nullptr;
  unsigned long long *d_tot = nullptr;
  PA_TRY(sc.get(&d_mask, (size_t)n + 1));
  PA_TRY(sc.get(&d_len, (size_t)n + 1));
  PA_TRY(sc.get(&d_rp, (size_t)n + 1));
  PA_TRY(sc.get(&d_flag, (size_t)n + 1));
  PA_TRY(sc.get(&d_pos, (size_t)n + 1));
  PA_TRY(sc.get(&d_ids, (size_t)n + 1));
  PA_TRY(sc.get(&d_crp, (size_t)n + 1));
  PA_TRY(sc.get(&d_bad, 1));
  PA_TRY(sc.get(&d_tot, 64));
  PA_HIP(hipMemsetAsync(d_bad, 0, sizeof(int), s));
  PA_HIP(hipMemsetAsync(d_tot, 0, sizeof(unsigned long long) * 64, s));
  if (n) PA_HIP(hipMemcpyAsync(d_mask, mask, sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
  if (n) {
    hipLaunchKernelGGL(kr_bad_mask, grid1(n), dim3(256), 0, s, d_mask, (int)n, (int)n_sel, d_bad);
    hipLaunchKernelGGL(kr_totals, grid1(n), dim3(256), 0, s, d_mask, A.len, oh ? B.len : nullptr, (int)n, (int)n_sel, d_tot);
  }
  int bad = 0;
  unsigned long long tot[64];
  PA_TRY(d2h(s, &bad, d_bad, 1));
  PA_TRY(d2h(s, tot, d_tot, 64));
  PA_REQUIRE(!bad, "a mask entry outside -1..n_sel-1");
  unsigned long long most = 0;
  for (int k = 0; k < n_sel; ++k) most = std::max(most, tot[k]);
  PA_REQUIRE(most < 2147483000ull, "a row subset holds more entries than Int32 row pointers address");
  // one pair of output buffers for all subsets (from the arena: no driver-side wipe per subset)
  int32_t *d_col = nullptr;
  double *d_val = nullptr;
  PA_TRY(pa_dev_alloc(c, (void **)&d_col, sizeof(int32_t) * (most + 8), PA_MEM_MATRIX));
  int st = pa_dev_alloc(c, (void **)&d_val, sizeof(double) * (most + 8), PA_MEM_MATRIX);
  for (int k = 0; k < n_sel && st == PA_OK; ++k) {
    if (lower) hipLaunchKernelGGL(kr_len_lower, grid1(n + 1), dim3(256), 0, s, d_mask, k, A.start, A.len, col_a, (int)n, d_len);
    else hipLaunchKernelGGL(kr_len, grid1(n + 1), dim3(256), 0, s, d_mask, k, A.len, oh ? B.len : nullptr, (int)n, d_len);
    st = scan_exclusive(sc, s, d_len, d_rp, (size_t)n + 1);
    if (st != PA_OK) break;
    if (lower) {                                        // (the subset's size is known only now)
      int32_t t = 0;
      st = d2h(s, &t, d_rp + n, 1);
      if (st != PA_OK) break;
      tot[k] = (unsigned long long)t;
      if (t == 0) continue;                             // no block: out[k] stays NULL
      hipLaunchKernelGGL(kr_fill_lower, grid1(n), dim3(256), 0, s, d_mask, k, (int)n, d_rp, A.start, A.len, col_a, oo->d_val, d_col, d_val);
    } else if (tot[k])
      hipLaunchKernelGGL(kr_fill, grid1(n * 8), dim3(256), 0, s, d_mask, k, (int)n, d_rp, A.start, A.len, col_a, oo->d_val,
                         oh ? B.start : nullptr, oh ? B.len : nullptr, col_b, oh ? oh->d_val : nullptr, (int)oo->n_cols, d_col, d_val);
    // non-empty rows counted, and compacted when most rows are empty (csr_fill_slab's rule), here: the host gets the final
    // row pointer only (a colour of the 256^3 operator: 8 MB instead of 67, and no passes over 16.8 M rows)
    hipLaunchKernelGGL(kr_flag, grid1(n + 1), dim3(256), 0, s, d_len, (int)n, d_flag);
    st = scan_exclusive(sc, s, d_flag, d_pos, (size_t)n + 1);
    if (st != PA_OK) break;
    int32_t n_nonempty = 0;
    st = d2h(s, &n_nonempty, d_pos + n, 1);
    if (st != PA_OK) break;
    const bool compact = n > 0 && (int64_t)n_nonempty * 2 < n;
    std::vector<int32_t> crp((size_t)(compact ? n_nonempty : n) + 1);
    if (compact) {
      hipLaunchKernelGGL(kr_compact, grid1(n + 1), dim3(256), 0, s, d_len, d_pos, d_rp, (int)n, d_ids, d_crp);
      st = d2h(s, crp.data(), d_crp, crp.size());
    } else st = d2h(s, crp.data(), d_rp, crp.size());
    if (st != PA_OK || hipGetLastError() != hipSuccess) { st = PA_ERR_HIP; break; }
    // (a subset of own|own's rows holds own|own's values: it takes that block's value dictionary -- pa_csr.hip, vdict_build)
    pa_tls_vdict_parent = (!oh && oo->use_vdict) ? oo : nullptr;
    st = pa_csr_from_device_rows(c, n, n_cols, (int64_t)tot[k], n_nonempty, crp, compact ? d_ids : nullptr, d_col, d_val, &out[k]);
    pa_tls_vdict_parent = nullptr;
  }
  (void)hipStreamSynchronize(s);
  if (getenv("PA_SETUP_TIMING"))
    fprintf(stderr, "[pa setup] %d row subset(s) of a %lld-row part on the device: %.3f s\n", (int)n_sel, (long long)n,
            std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count());
  pa_dev_free(c, d_col);
  if (d_val) pa_dev_free(c, d_val);
  if (st != PA_OK)
    for (int k = 0; k < n_sel; ++k)
      if (out[k]) { pa_csr_destroy(out[k]); out[k] = nullptr; }
  return st;
}

extern "C" int pa_csr_select_rows(const pa_csr *oo, const pa_csr *oh, const int32_t *mask, int32_t n_sel, pa_csr **out) {
  return select_rows_impl(oo, oh, mask, n_sel, out, false, -1);
}

// out[k] = the rows with mask == k again, but only their entries in own columns j with 0 <= mask[j] < k (no ghost columns):
// all a forward multicolour sweep over x == 0 reads when it reaches colour k.  out[k] is NULL when there is no such entry
// (k == 0 always).  n_cols: the column count the blocks get (own + ghost columns of the part, so that they take the same x).
extern "C" int pa_csr_select_rows_lower(const pa_csr *oo, int64_t n_cols, const int32_t *mask, int32_t n_sel, pa_csr **out) {
  return select_rows_impl(oo, nullptr, mask, n_sel, out, true, n_cols);
}

// d[r] = the stored (r,r) entry of the own|own block (0.0 when the row holds none): the smoother's diagonal
extern "C" int pa_csr_diagonal(const pa_csr *oo, pa_vec *d) {
  PA_REQUIRE(oo && d && !oo->next, "bad arguments");
  PA_REQUIRE(d->n_own + d->n_ghost >= oo->n_rows, "the vector is shorter than the block has rows");
  const int32_t *col = raw_columns(oo);
  PA_REQUIRE(oo->nnz == 0 || col, "the block did not keep its raw columns (create it under pa_ctx_keep_raw_columns)");
  pa_ctx *c = oo->ctx;
  PA_HIP(hipSetDevice(c->device));
  scratch sc;
  row_spans A;
  PA_TRY(spans_of(c, sc, oo, oo->n_rows, A));
  if (oo->n_rows) hipLaunchKernelGGL(kr_diag, grid1(oo->n_rows), dim3(256), 0, c->s[0], A.start, A.len, col, oo->d_val, (int)oo->n_rows, d->d);
  PA_HIP(hipGetLastError());
  PA_HIP(hipStreamSynchronize(c->s[0]));
  return PA_OK;
}

// ------------------------------------------------------------------------------------------------
// The level-scheduled Gauss-Seidel smoother (pa_gs, pa_mg.hip) from the blocks already in HBM: the unsplit local CSR by
// the kernels above (one subset: every row), the diagonal, and the dependency levels of the sequential sweep
// (PartitionedSolvers/src/smoothers.jl:144-160: row i needs every own column j < i) by rounds over the rows whose lower
// neighbours are all done -- level(i) = 1 + max level(j), the same numbers as pa_gs_create's loop over the rows in order.
// A round finds the rows it frees through the UPPER entries of the rows it finishes, which is right when the own|own pattern
// is structurally symmetric (what pa_gs_create demands anyway); the result is verified entry by entry against the definition
// and the caller takes the host route when it does not hold.
// ------------------------------------------------------------------------------------------------
__global__ void kg_lower_count(const int32_t *__restrict__ start, const int32_t *__restrict__ len, const int32_t *__restrict__ col, int n,
                               int32_t *__restrict__ cnt) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  int k = 0;
  for (int p = start[r], e = p + len[r]; p < e; ++p) k += col[p] < r;
  cnt[r] = k;
}

// lanes that share a frontier row in a round (each takes every PA_ROUND_LANES-th entry of the row).  256^3, four levels, both traversals
// (sweep levels / greedy colouring), ms: 1 lane 94 / 112, 4 lanes 63 / 86, 8 lanes 70 / 100, 32 lanes 115 / 133 -- a round is a small
// launch whose cost grows with its grid, not a chain of atomics
#ifndef PA_ROUND_LANES
#define PA_ROUND_LANES 4
#endif
__global__ void kg_first(const int32_t *__restrict__ cnt, int n, int32_t *__restrict__ frontier, int *__restrict__ count) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n && cnt[r] == 0) frontier[atomicAdd(count, 1)] = r;
}

// One round.  The frontier's size is read from device memory (sizes[lv], written by the round before) and the next one's
// accumulates in sizes[lv + 1]: the host queues rounds without waiting for any of them and looks at a size only now and then
// (a launch's grid is a guess; the loop strides over whatever the frontier holds).
__global__ void kg_round(const int32_t *__restrict__ frontier, int *__restrict__ sizes, int lv, const int32_t *__restrict__ start,
                         const int32_t *__restrict__ len, const int32_t *__restrict__ col, int n, int32_t *__restrict__ level,
                         int32_t *__restrict__ cnt, int32_t *__restrict__ next) {
  const int size = sizes[lv];
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < (long long)size * PA_ROUND_LANES; t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t / PA_ROUND_LANES), lane = (int)(t % PA_ROUND_LANES);
    const int r = frontier[i];
    if (lane == 0) level[r] = lv;
    for (int p = start[r] + lane, e = start[r] + len[r]; p < e; p += PA_ROUND_LANES) {
      const int j = col[p];
      if (j > r && j < n && atomicSub(&cnt[j], 1) == 1) next[atomicAdd(&sizes[lv + 1], 1)] = j;
    }
  }
}

// level(r) == 1 + max level(own j < r) (0 without such an entry), every own j > r is swept later, the diagonal is there
__global__ void kg_verify(const int32_t *__restrict__ start, const int32_t *__restrict__ len, const int32_t *__restrict__ col,
                          const double *__restrict__ diag, const int32_t *__restrict__ level, int n, int *__restrict__ bad) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  int want = 0;
  bool has_diag = false, ok = true;
  for (int p = start[r], e = p + len[r]; p < e; ++p) {
    const int j = col[p];
    if (j == r) has_diag = true;
    if (j < r) want = max(want, level[j] + 1);
    if (j > r && j < n && level[j] <= level[r]) ok = false;
  }
  if (!ok || want != level[r]) atomicOr(bad, 1);
  if (!has_diag || diag[r] == 0.0) atomicOr(bad, 2);
}

__global__ void kg_iota(int32_t *__restrict__ v, int n) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) v[r] = r;
}

// Queues the rounds in batches of 32 launches and reads the frontier sizes back once per batch (d_sizes: n + 64 zeroed ints,
// d_sizes[0] set by kg_first; round k works on d_sizes[k] rows and counts the next frontier in d_sizes[k + 1]).  launch(k, cur,
// nxt, blocks) enqueues round k.  sizes: the non-empty rounds' sizes; false when more rows were freed than there are.
template <class F>
static int run_rounds(hipStream_t s, int64_t n, int *d_sizes, int32_t *d_f0, int32_t *d_f1, F launch, std::vector<int> &sizes, bool *ok) {
  const int BATCH = 32;
  int buf[BATCH];
  int64_t done = 0, k = 0;
  int guess = 0;
  PA_TRY(d2h(s, &guess, d_sizes, 1));
  sizes.clear();
  *ok = true;
  if (guess == 0) return PA_OK;
  while (true) {
    const int blocks = (int)std::min<int64_t>(65535, std::max<int64_t>(64, ((int64_t)guess * PA_ROUND_LANES + 255) / 256 * 2));
    for (int b = 0; b < BATCH; ++b) launch((int)(k + b), ((k + b) & 1) ? d_f1 : d_f0, ((k + b) & 1) ? d_f0 : d_f1, blocks);
    PA_HIP(hipGetLastError());
    PA_TRY(d2h(s, buf, d_sizes + k, (size_t)BATCH));
    for (int b = 0; b < BATCH; ++b) {
      if (buf[b] == 0) return PA_OK;                   // (the rounds queued behind it found nothing to do)
      done += buf[b];
      if (done > n) { *ok = false; return PA_OK; }
      sizes.push_back(buf[b]);
      guess = std::max(buf[b], guess / 2);
    }
    k += BATCH;
    if (k > n) { *ok = false; return PA_OK; }
  }
}

extern "C" int pa_gs_create_from_blocks(const pa_csr *oo, const pa_csr *oh, int ordering, pa_gs **out) {
  PA_REQUIRE(oo && out, "bad arguments");
  PA_REQUIRE(ordering == PA_GS_SEQUENTIAL, "the device route builds the sequential ordering only");
  PA_REQUIRE(!oo->next && !(oh && oh->next), "a block stored as a chain (2^31 stored entries or more: row slabs; a band beyond the sliding x window: column pieces, PA_SPMV_COLSPLIT=0 keeps it whole) takes the host route");
  PA_REQUIRE(!oh || (oh->n_rows == oo->n_rows && oh->ctx == oo->ctx), "the own|ghost block does not match the own|own block");
  PA_REQUIRE(oo->n_rows == oo->n_cols, "the own|own block is not square");
  pa_ctx *c = oo->ctx;
  const int64_t n_ghost_cols = oh ? oh->n_cols : 0;      // (counted before an entry-less own|ghost block is dropped below)
  if (oh && oh->nnz == 0) oh = nullptr;
  const int32_t *col_a = raw_columns(oo), *col_b = oh ? raw_columns(oh) : nullptr;
  PA_REQUIRE(oo->nnz == 0 || col_a, "the own|own block did not keep its raw columns (create it under pa_ctx_keep_raw_columns)");
  PA_REQUIRE(!oh || col_b, "the own|ghost block did not keep its raw columns (create it under pa_ctx_keep_raw_columns)");
  const int64_t n = oo->n_rows, n_local = oo->n_cols + n_ghost_cols, nnz = oo->nnz + (oh ? oh->nnz : 0);
  PA_REQUIRE(nnz < (int64_t)2147483000 && n_local < (int64_t)2147483000, "too large for Int32 offsets");
  PA_HIP(hipSetDevice(c->device));
  hipStream_t s = c->s[0];
  scratch sc;
  row_spans A, B;
  PA_TRY(spans_of(c, sc, oo, n, A));
  if (oh) PA_TRY(spans_of(c, sc, oh, n, B));
  int32_t *d_mask = nullptr, *d_len = nullptr, *d_cnt = nullptr, *d_level = nullptr, *d_f0 = nullptr, *d_f1 = nullptr, *d_iota = nullptr,
          *d_lsorted = nullptr;
  int *d_count = nullptr, *d_sizes = nullptr;
  PA_TRY(sc.get(&d_sizes, (size_t)n + 64));
  PA_TRY(sc.get(&d_mask, (size_t)n + 1));
  PA_TRY(sc.get(&d_len, (size_t)n + 1));
  PA_TRY(sc.get(&d_cnt, (size_t)n + 1));
  PA_TRY(sc.get(&d_level, (size_t)n + 1));
  PA_TRY(sc.get(&d_f0, (size_t)n + 1));
  PA_TRY(sc.get(&d_f1, (size_t)n + 1));
  PA_TRY(sc.get(&d_iota, (size_t)n + 1));
  PA_TRY(sc.get(&d_lsorted, (size_t)n + 1));
  PA_TRY(sc.get(&d_count, 4));
  pa_gs *g = new pa_gs();
  g->ctx = c; g->n_own = n; g->n_local = n_local; g->nnz = nnz;
  auto fail = [&](int st) { pa_gs_destroy(g); return st; };
#define PA_G(call) do { if ((call) != hipSuccess) { pa_set_err("HIP error in %s: %s", #call, hipGetErrorString(hipGetLastError())); return fail(PA_ERR_HIP); } } while (0)
#define PA_GT(call) do { const int st_ = (call); if (st_ != PA_OK) return fail(st_); } while (0)
  PA_G(pa_raw_malloc(&g->d_rowptr, sizeof(int32_t) * (n + 1)));
  PA_G(pa_raw_malloc(&g->d_col, sizeof(int32_t) * std::max<int64_t>(1, nnz)));
  PA_G(pa_raw_malloc(&g->d_val, sizeof(double) * std::max<int64_t>(1, nnz)));
  PA_G(pa_raw_malloc(&g->d_diag, sizeof(double) * std::max<int64_t>(1, n)));
  PA_G(pa_raw_malloc(&g->d_rows, sizeof(int32_t) * std::max<int64_t>(1, n)));
  // the unsplit CSR: every row, own entries then ghost entries shifted
  PA_G(hipMemsetAsync(d_mask, 0, sizeof(int32_t) * (n + 1), s));
  hipLaunchKernelGGL(kr_len, grid1(n + 1), dim3(256), 0, s, d_mask, 0, A.len, oh ? B.len : nullptr, (int)n, d_len);
  PA_GT(scan_exclusive(sc, s, d_len, g->d_rowptr, (size_t)n + 1));
  if (nnz)
    hipLaunchKernelGGL(kr_fill, grid1(n * 8), dim3(256), 0, s, d_mask, 0, (int)n, g->d_rowptr, A.start, A.len, col_a, oo->d_val,
                       oh ? B.start : nullptr, oh ? B.len : nullptr, col_b, oh ? oh->d_val : nullptr, (int)oo->n_cols, g->d_col, g->d_val);
  if (n) hipLaunchKernelGGL(kr_diag, grid1(n), dim3(256), 0, s, A.start, A.len, col_a, oo->d_val, (int)n, g->d_diag);
  // dependency levels by rounds
  if (n) hipLaunchKernelGGL(kg_lower_count, grid1(n), dim3(256), 0, s, A.start, A.len, col_a, (int)n, d_cnt);
  PA_G(hipMemsetAsync(d_count, 0, sizeof(int) * 4, s));
  PA_G(hipMemsetAsync(d_sizes, 0, sizeof(int) * (size_t)(n + 64), s));
  PA_G(hipMemsetAsync(d_level, 0xFF, sizeof(int32_t) * (n + 1), s));                        // -1: not reached
  if (n) hipLaunchKernelGGL(kg_first, grid1(n), dim3(256), 0, s, d_cnt, (int)n, d_f0, d_sizes);
  std::vector<int> round_sizes;
  bool rounds_ok = true;
  PA_GT(run_rounds(s, n, d_sizes, d_f0, d_f1, [&](int k, int32_t *cur, int32_t *nxt, int blocks) {
    hipLaunchKernelGGL(kg_round, dim3(blocks), dim3(256), 0, s, cur, d_sizes, k, A.start, A.len, col_a, (int)n, d_level, d_cnt, nxt);
  }, round_sizes, &rounds_ok));
  g->lev_ptr.assign(1, 0);
  int64_t done = 0;
  for (int sz : round_sizes) {
    done += sz;
    g->max_level_rows = std::max<int64_t>(g->max_level_rows, sz);
    g->lev_ptr.push_back((int32_t)done);
  }
  if (!rounds_ok) done = -1;
  int bad = done == n ? 0 : 1;
  if (!bad && n) {
    int *d_bad = d_count + 3;
    PA_G(hipMemsetAsync(d_bad, 0, sizeof(int), s));
    hipLaunchKernelGGL(kg_verify, grid1(n), dim3(256), 0, s, A.start, A.len, col_a, g->d_diag, d_level, (int)n, d_bad);
    PA_GT(d2h(s, &bad, d_bad, 1));
  }
  if (bad) {
    pa_set_err(bad & 2 ? "a row has no (non-zero) diagonal entry"
                       : "own x own pattern is not structurally symmetric: the rounds do not reproduce the sequential sweep's levels");
    return fail(PA_ERR_ARG);
  }
  // rows by level, ascending inside a level (stable sort of the row ids by level)
  if (n) {
    hipLaunchKernelGGL(kg_iota, grid1(n), dim3(256), 0, s, d_iota, (int)n);
    unsigned bits = 1;
    while (((int64_t)1 << bits) < (int64_t)g->lev_ptr.size()) ++bits;
    PA_GT(sort_pairs(sc, s, d_level, d_lsorted, d_iota, g->d_rows, (size_t)n, bits));
  }
  PA_G(hipGetLastError());
  PA_G(hipStreamSynchronize(s));
#undef PA_G
#undef PA_GT
  *out = g;
  return PA_OK;
}

// ------------------------------------------------------------------------------------------------
// Greedy colouring in natural order (pa_host_greedy_coloring: row r takes the smallest colour no own neighbour j < r has)
// by the same rounds: a row is coloured in the round after its last lower neighbour, from colours that are final by then.
// Verified against the definition row by row (which has one solution); PA_ERR_ARG when the rounds do not get there (a
// pattern that is not structurally symmetric): the caller colours on the host.
// ------------------------------------------------------------------------------------------------
__global__ void kg_round_color(const int32_t *__restrict__ frontier, int *__restrict__ sizes, int lv, const int32_t *__restrict__ start,
                               const int32_t *__restrict__ len, const int32_t *__restrict__ col, int n, int32_t *__restrict__ color,
                               int32_t *__restrict__ cnt, int32_t *__restrict__ next) {
  const int size = sizes[lv];
  for (long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x; t < (long long)size * PA_ROUND_LANES; t += (long long)gridDim.x * blockDim.x) {
    const int i = (int)(t / PA_ROUND_LANES), lane = (int)(t % PA_ROUND_LANES);
    const int r = frontier[i];
    if (lane == 0) {
      unsigned long long used = 0;
      for (int p = start[r], e = p + len[r]; p < e; ++p) {
        const int j = col[p];
        if (j < r && color[j] < 64) used |= 1ull << color[j];
      }
      int c = 0;
      while (c < 63 && (used >> c) & 1ull) ++c;
      color[r] = c;
    }
    for (int p = start[r] + lane, e = start[r] + len[r]; p < e; p += PA_ROUND_LANES) {
      const int j = col[p];
      if (j > r && j < n && atomicSub(&cnt[j], 1) == 1) next[atomicAdd(&sizes[lv + 1], 1)] = j;
    }
  }
}

__global__ void kg_verify_color(const int32_t *__restrict__ start, const int32_t *__restrict__ len, const int32_t *__restrict__ col,
                                const int32_t *__restrict__ color, int n, int *__restrict__ bad, int *__restrict__ n_colors) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n) return;
  unsigned long long used = 0;
  for (int p = start[r], e = p + len[r]; p < e; ++p) {
    const int j = col[p];
    if (j < r && color[j] >= 0 && color[j] < 64) used |= 1ull << color[j];
    if (j < r && color[j] < 0) atomicOr(bad, 1);
  }
  int c = 0;
  while (c < 63 && (used >> c) & 1ull) ++c;
  if (c != color[r]) atomicOr(bad, 1);
  atomicMax(n_colors, color[r] + 1);
}

extern "C" int pa_csr_greedy_coloring(const pa_csr *oo, int32_t *color, int32_t *n_colors) {
  PA_REQUIRE(oo && color && n_colors && !oo->next, "bad arguments");
  const int32_t *col = raw_columns(oo);
  PA_REQUIRE(oo->nnz == 0 || col, "the block did not keep its raw columns (create it under pa_ctx_keep_raw_columns)");
  pa_ctx *c = oo->ctx;
  const int64_t n = oo->n_rows;
  PA_HIP(hipSetDevice(c->device));
  hipStream_t s = c->s[0];
  scratch sc;
  row_spans A;
  PA_TRY(spans_of(c, sc, oo, n, A));
  int32_t *d_cnt = nullptr, *d_color = nullptr, *d_f0 = nullptr, *d_f1 = nullptr;
  int *d_count = nullptr, *d_sizes = nullptr;
  PA_TRY(sc.get(&d_sizes, (size_t)n + 64));
  PA_TRY(sc.get(&d_cnt, (size_t)n + 1));
  PA_TRY(sc.get(&d_color, (size_t)n + 1));
  PA_TRY(sc.get(&d_f0, (size_t)n + 1));
  PA_TRY(sc.get(&d_f1, (size_t)n + 1));
  PA_TRY(sc.get(&d_count, 8));
  if (n) hipLaunchKernelGGL(kg_lower_count, grid1(n), dim3(256), 0, s, A.start, A.len, col, (int)n, d_cnt);
  PA_HIP(hipMemsetAsync(d_count, 0, sizeof(int) * 8, s));
  PA_HIP(hipMemsetAsync(d_sizes, 0, sizeof(int) * (size_t)(n + 64), s));
  PA_HIP(hipMemsetAsync(d_color, 0xFF, sizeof(int32_t) * (n + 1), s));
  if (n) hipLaunchKernelGGL(kg_first, grid1(n), dim3(256), 0, s, d_cnt, (int)n, d_f0, d_sizes);
  std::vector<int> round_sizes;
  bool rounds_ok = true;
  PA_TRY(run_rounds(s, n, d_sizes, d_f0, d_f1, [&](int k, int32_t *cur, int32_t *nxt, int blocks) {
    hipLaunchKernelGGL(kg_round_color, dim3(blocks), dim3(256), 0, s, cur, d_sizes, k, A.start, A.len, col, (int)n, d_color, d_cnt, nxt);
  }, round_sizes, &rounds_ok));
  int64_t done = 0;
  for (int sz : round_sizes) done += sz;
  if (!rounds_ok) done = -1;
  int res[2] = {done == n ? 0 : 1, 0};
  if (!res[0] && n) {
    hipLaunchKernelGGL(kg_verify_color, grid1(n), dim3(256), 0, s, A.start, A.len, col, d_color, (int)n, d_count + 3, d_count + 4);
    PA_TRY(d2h(s, res, d_count + 3, 2));
  }
  PA_HIP(hipGetLastError());
  PA_REQUIRE(!res[0], "the rounds do not reproduce the greedy colouring in natural order (own x own pattern not structurally symmetric)");
  if (n) PA_TRY(d2h(s, color, d_color, (size_t)n));
  *n_colors = res[1];
  return PA_OK;
}

// The same colouring when the dependency levels of the sequential sweep over this block are already known (a pa_gs made from it by
// pa_gs_create_from_blocks: rows sorted by level, the levels' bounds on the host): a row's lower neighbours all sit in earlier
// levels, so the levels are coloured one after the other -- one small launch each, queued without a look at the device in between
// (no frontier to discover, no atomics: a level costs ~10 us where a round of the discovery costs ~25).  Same definition, same
// verification; PA_ERR_ARG when the levels are not this block's.
__global__ void kg_color_level(const int32_t *__restrict__ rows, int first, int count, const int32_t *__restrict__ start,
                               const int32_t *__restrict__ len, const int32_t *__restrict__ col, int32_t *__restrict__ color) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const int r = rows[first + i];
  unsigned long long used = 0;
  for (int p = start[r], e = p + len[r]; p < e; ++p) {
    const int j = col[p];
    if (j < r) { const int cj = color[j]; if (cj >= 0 && cj < 64) used |= 1ull << cj; }
  }
  int c = 0;
  while (c < 63 && (used >> c) & 1ull) ++c;
  color[r] = c;
}

extern "C" int pa_csr_greedy_coloring_by_levels(const pa_csr *oo, const pa_gs *gs, int32_t *color, int32_t *n_colors) {
  PA_REQUIRE(oo && gs && color && n_colors && !oo->next, "bad arguments");
  PA_REQUIRE(gs->ctx == oo->ctx && gs->n_own == oo->n_rows, "the smoother was not made from this block");
  PA_REQUIRE(!gs->lev_ptr.empty() && gs->lev_ptr.back() == oo->n_rows && gs->d_rows, "the smoother holds no dependency levels (sequential ordering, device route)");
  const int32_t *col = raw_columns(oo);
  PA_REQUIRE(oo->nnz == 0 || col, "the block did not keep its raw columns (create it under pa_ctx_keep_raw_columns)");
  pa_ctx *c = oo->ctx;
  const int64_t n = oo->n_rows;
  PA_HIP(hipSetDevice(c->device));
  hipStream_t s = c->s[0];
  scratch sc;
  row_spans A;
  PA_TRY(spans_of(c, sc, oo, n, A));
  int32_t *d_color = nullptr;
  int *d_count = nullptr;
  PA_TRY(sc.get(&d_color, (size_t)n + 1));
  PA_TRY(sc.get(&d_count, 8));
  PA_HIP(hipMemsetAsync(d_count, 0, sizeof(int) * 8, s));
  PA_HIP(hipMemsetAsync(d_color, 0xFF, sizeof(int32_t) * (n + 1), s));
  for (size_t l = 0; l + 1 < gs->lev_ptr.size(); ++l) {
    const int first = gs->lev_ptr[l], count = gs->lev_ptr[l + 1] - first;
    if (count > 0) hipLaunchKernelGGL(kg_color_level, grid1(count), dim3(256), 0, s, gs->d_rows, first, count, A.start, A.len, col, d_color);
  }
  int res[2] = {0, 0};
  if (n) {
    hipLaunchKernelGGL(kg_verify_color, grid1(n), dim3(256), 0, s, A.start, A.len, col, d_color, (int)n, d_count + 3, d_count + 4);
    PA_TRY(d2h(s, res, d_count + 3, 2));
  }
  PA_HIP(hipGetLastError());
  PA_REQUIRE(!res[0], "these levels do not give the greedy colouring in natural order (not this block's levels?)");
  if (n) PA_TRY(d2h(s, color, d_color, (size_t)n));
  *n_colors = res[1];
  return PA_OK;
}

// ------------------------------------------------------------------------------------------------
// HPCG's 27-point operator of one part, own|own block and right-hand side, generated in HBM
// (HPCG/src/sparse_matrix.jl:28-122: build_matrix loops over the part's cells and their 27 neighbours in (sz, sy, sx)
// order, 26.0 on the diagonal, -1.0 elsewhere, b = 27 - #neighbours inside the global grid; psparse then keeps the own
// columns in own|own, in ascending local id -- which IS the (sz, sy, sx) order).  The host's fused generator
// (pa_host.cpp, hpcg_split_csr_impl) writes the same arrays; the surface-only own|ghost block stays with it.
// ------------------------------------------------------------------------------------------------
struct hpcg_box { int nx, ny, nz; long long gnx, gny, gnz, gx0, gy0, gz0; };

__device__ inline void dim_cnt(long long g, int i, int n, long long gn, int &own, int &grid) {
  grid = 1 + (g > 0) + (g < gn - 1);
  own = 1 + (i > 0) + (i < n - 1);
}

__global__ void kh_count(hpcg_box B, int n, int32_t *__restrict__ len, double *__restrict__ b) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n) return;
  if (r == n) { if (len) len[r] = 0; return; }
  const int ix = r % B.nx, iy = (r / B.nx) % B.ny, iz = r / (B.nx * B.ny);
  int ax, bx, ay, by, az, bz;
  dim_cnt(B.gx0 + ix, ix, B.nx, B.gnx, ax, bx);
  dim_cnt(B.gy0 + iy, iy, B.ny, B.gny, ay, by);
  dim_cnt(B.gz0 + iz, iz, B.nz, B.gnz, az, bz);
  if (len) len[r] = ax * ay * az;
  if (b) b[r] = 27.0 - (double)(bx * by * bz);
}

// 32 lanes per row: lane k < 27 is the neighbour (sz, sy, sx) = (k / 9 - 1, (k / 3) % 3 - 1, k % 3 - 1); its slot in the row is the number
// of neighbours before it that lie inside the part -- a row's entries leave the wave as one contiguous store (one thread per row wrote
// 27 entries 216 bytes apart from its neighbours': 25 ms for the 450 M entries of a 256^3 part, 5.4 GB)
__global__ __launch_bounds__(256) void kh_fill(hpcg_box B, int n, const int32_t *__restrict__ rp, int32_t *__restrict__ col, double *__restrict__ val) {
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int r = (int)(t >> 5), k = (int)(t & 31);
  const bool row_ok = r < n;
  const int rr = row_ok ? r : 0;
  const int ix = rr % B.nx, iy = (rr / B.nx) % B.ny, iz = rr / (B.nx * B.ny);
  const int sz = k / 9 - 1, sy = (k / 3) % 3 - 1, sx = k % 3 - 1;
  const int cz = iz + sz, cy = iy + sy, cx = ix + sx;
  const bool valid = row_ok && k < 27 && cz >= 0 && cz < B.nz && cy >= 0 && cy < B.ny && cx >= 0 && cx < B.nx;
  const unsigned long long m = __ballot(valid);
  const unsigned mine = (unsigned)(m >> (threadIdx.x & 32));            // my row's 32 lanes
  if (!valid) return;
  const int p = rp[r] + __popc(mine & ((1u << k) - 1u));
  col[p] = (cz * B.ny + cy) * B.nx + cx;
  val[p] = k == 13 ? 26.0 : -1.0;
}

extern "C" int pa_hpcg_own_block_create(pa_ctx *c, int64_t nx, int64_t ny, int64_t nz, int64_t gnx, int64_t gny, int64_t gnz,
                                        int64_t gix0, int64_t giy0, int64_t giz0, pa_csr **own_own, pa_vec *b) {
  PA_REQUIRE(c && own_own && nx > 0 && ny > 0 && nz > 0, "bad arguments");
  // (gix0, giy0, giz0: global coordinates of the part's first node, 1-based like pa_host_hpcg_split_csr's)
  PA_REQUIRE(gix0 >= 1 && giy0 >= 1 && giz0 >= 1 && gix0 - 1 + nx <= gnx && giy0 - 1 + ny <= gny && giz0 - 1 + nz <= gnz, "the part's box is not inside the grid");
  const int64_t n = nx * ny * nz;
  PA_REQUIRE(n < (int64_t)2147483000 / 27, "a part of this size takes the host generator (Int64 row pointers, slabs)");
  PA_REQUIRE(!b || b->n_own + b->n_ghost >= n, "b is shorter than the part has rows");
  PA_HIP(hipSetDevice(c->device));
  hipStream_t s = c->s[0];
  const bool tm_ = getenv("PA_SETUP_TIMING") != nullptr;
  auto t0_ = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!tm_) return;
    (void)hipStreamSynchronize(s);
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[pa setup] own block %-12s %8.3f s\n", what, std::chrono::duration<double>(t1 - t0_).count());
    t0_ = t1;
  };
  scratch sc;
  int32_t *d_len = nullptr, *d_rp = nullptr;
  PA_TRY(sc.get(&d_len, (size_t)n + 1));
  PA_TRY(sc.get(&d_rp, (size_t)n + 1));
  const hpcg_box B{(int)nx, (int)ny, (int)nz, (long long)gnx, (long long)gny, (long long)gnz, (long long)gix0 - 1, (long long)giy0 - 1, (long long)giz0 - 1};
  hipLaunchKernelGGL(kh_count, grid1(n + 1), dim3(256), 0, s, B, (int)n, d_len, b ? b->d : nullptr);
  PA_TRY(scan_exclusive(sc, s, d_len, d_rp, (size_t)n + 1));
  std::vector<int32_t> crp((size_t)n + 1);
  lap("count+scan");
  PA_TRY(d2h(s, crp.data(), d_rp, crp.size()));
  lap("crp to host");
  const int64_t nnz = crp.back();
  int32_t *d_col = nullptr;
  double *d_val = nullptr;
  PA_TRY(pa_dev_alloc(c, (void **)&d_col, sizeof(int32_t) * (nnz + 8), PA_MEM_MATRIX));
  int st = pa_dev_alloc(c, (void **)&d_val, sizeof(double) * (nnz + 8), PA_MEM_MATRIX);
  if (st == PA_OK) {
    hipLaunchKernelGGL(kh_fill, grid1(n * 32), dim3(256), 0, s, B, (int)n, d_rp, d_col, d_val);
    if (hipGetLastError() != hipSuccess || hipStreamSynchronize(s) != hipSuccess) st = PA_ERR_HIP;
  }
  lap("fill");
  if (st == PA_OK) st = pa_csr_from_device_rows(c, n, n, nnz, n, crp, nullptr, d_col, d_val, own_own);
  (void)hipStreamSynchronize(s);
  lap("block");
  pa_dev_free(c, d_col);
  if (d_val) pa_dev_free(c, d_val);
  return st;
}

// b alone (27 - the number of neighbours inside the global grid): for a vector created after the block, so that the arena
// knows the matrix streams' class by the time it places it
extern "C" int pa_hpcg_rhs(pa_ctx *c, int64_t nx, int64_t ny, int64_t nz, int64_t gnx, int64_t gny, int64_t gnz, int64_t gix0,
                           int64_t giy0, int64_t giz0, pa_vec *b) {
  PA_REQUIRE(c && b && nx > 0 && ny > 0 && nz > 0, "bad arguments");
  PA_REQUIRE(gix0 >= 1 && giy0 >= 1 && giz0 >= 1 && gix0 - 1 + nx <= gnx && giy0 - 1 + ny <= gny && giz0 - 1 + nz <= gnz, "the part's box is not inside the grid");
  const int64_t n = nx * ny * nz;
  PA_REQUIRE(n < (int64_t)2147483000 && b->n_own + b->n_ghost >= n, "b is shorter than the part has rows");
  PA_HIP(hipSetDevice(c->device));
  const hpcg_box B{(int)nx, (int)ny, (int)nz, (long long)gnx, (long long)gny, (long long)gnz, (long long)gix0 - 1, (long long)giy0 - 1, (long long)giz0 - 1};
  hipLaunchKernelGGL(kh_count, grid1(n + 1), dim3(256), 0, c->s[0], B, (int)n, (int32_t *)nullptr, b->d);
  PA_HIP(hipGetLastError());
  PA_HIP(hipStreamSynchronize(c->s[0]));
  return PA_OK;
}

// ------------------------------------------------------------------------------------------------
// In which order to sweep the colours of a multicolour smoother inside a multigrid cycle: affinity[k] = the mean number of
// stored entries of a colour-k row whose column is a row the coarse grid keeps.  Sweeping in order of decreasing affinity
// puts the kept rows' own colour at the turn of the symmetric sweep instead of at its end, where their residual -- all the
// restriction injects -- would be zero up to rounding (hpcg.py, ColoredGaussSeidelSpMV).
// ------------------------------------------------------------------------------------------------
__global__ void ka_mark(const int32_t *__restrict__ rows, int n_rows, int n, int32_t *__restrict__ mark) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_rows && rows[i] >= 0 && rows[i] < n) mark[rows[i]] = 1;
}

__global__ void ka_affinity(const int32_t *__restrict__ color, const int32_t *__restrict__ mark, const int32_t *__restrict__ start,
                            const int32_t *__restrict__ len, const int32_t *__restrict__ col, int n, int n_colors,
                            unsigned long long *__restrict__ sums, unsigned long long *__restrict__ rows) {
  __shared__ unsigned long long hs[64], hr[64];
  if (threadIdx.x < 64) { hs[threadIdx.x] = 0; hr[threadIdx.x] = 0; }
  __syncthreads();
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r < n) {
    const int k = color[r];
    if (k >= 0 && k < n_colors) {
      unsigned cnt = 0;
      for (int p = start[r], e = p + len[r]; p < e; ++p) cnt += col[p] < n && mark[col[p]] != 0;
      atomicAdd(&hs[k], (unsigned long long)cnt);
      atomicAdd(&hr[k], 1ull);
    }
  }
  __syncthreads();
  if (threadIdx.x < n_colors) {
    if (hs[threadIdx.x]) atomicAdd(&sums[threadIdx.x], hs[threadIdx.x]);
    if (hr[threadIdx.x]) atomicAdd(&rows[threadIdx.x], hr[threadIdx.x]);
  }
}

extern "C" int pa_csr_color_affinity(const pa_csr *oo, const int32_t *color, int32_t n_colors, const int32_t *kept_rows, int64_t n_kept,
                                     double *affinity) {
  PA_REQUIRE(oo && color && affinity && n_colors > 0 && n_colors <= 64 && n_kept >= 0 && (n_kept == 0 || kept_rows) && !oo->next, "bad arguments");
  const int32_t *col = raw_columns(oo);
  PA_REQUIRE(oo->nnz == 0 || col, "the block did not keep its raw columns (create it under pa_ctx_keep_raw_columns)");
  pa_ctx *c = oo->ctx;
  const int64_t n = oo->n_rows;
  PA_HIP(hipSetDevice(c->device));
  hipStream_t s = c->s[0];
  scratch sc;
  row_spans A;
  PA_TRY(spans_of(c, sc, oo, n, A));
  int32_t *d_color = nullptr, *d_mark = nullptr, *d_rows = nullptr;
  unsigned long long *d_acc = nullptr;
  PA_TRY(sc.get(&d_color, (size_t)n + 1));
  PA_TRY(sc.get(&d_mark, (size_t)n + 1));
  PA_TRY(sc.get(&d_rows, (size_t)n_kept + 1));
  PA_TRY(sc.get(&d_acc, 128));
  PA_HIP(hipMemsetAsync(d_mark, 0, sizeof(int32_t) * (n + 1), s));
  PA_HIP(hipMemsetAsync(d_acc, 0, sizeof(unsigned long long) * 128, s));
  if (n) PA_HIP(hipMemcpyAsync(d_color, color, sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
  if (n_kept) {
    PA_HIP(hipMemcpyAsync(d_rows, kept_rows, sizeof(int32_t) * n_kept, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(ka_mark, grid1(n_kept), dim3(256), 0, s, d_rows, (int)n_kept, (int)n, d_mark);
  }
  if (n) hipLaunchKernelGGL(ka_affinity, grid1(n), dim3(256), 0, s, d_color, d_mark, A.start, A.len, col, (int)n, (int)n_colors, d_acc, d_acc + 64);
  unsigned long long acc[128];
  PA_TRY(d2h(s, acc, d_acc, 128));
  PA_HIP(hipGetLastError());
  for (int k = 0; k < n_colors; ++k) affinity[k] = acc[64 + k] ? (double)acc[k] / (double)acc[64 + k] : 0.0;
  return PA_OK;
}
