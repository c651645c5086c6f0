// pa_sell.hip -- SELL-C-sigma storage of one CSR block with one lane per row.
//
// The reference's loop, spmv_csr! (src/sparse_utils.jl:649-669) / SparseMatricesCSR.mul!(y,A,x,alpha,beta), walks a row's
// stored entries left to right.  Here a wavefront owns a slab of C = 64 rows and every lane walks ITS row in exactly that
// order with its accumulator in a register -- no LDS, no barrier, no cross-lane sum -- so the result is bit-identical to
// the reference's (and to pa_spmv's) by construction; the file is compiled -ffp-contract=off like the rest.
// Storage: inside a slab the k-th stored entries of the 64 rows sit next to each other (val[slab_ptr + k*64 + lane]), so
// every value / column load of a wavefront is one contiguous 512 / 256 bytes; a slab is as wide as its longest row.  Rows
// are sorted by length inside windows of sigma rows before they are dealt to slabs (sigma = 1: none), which keeps the
// padding small on ragged matrices; padding slots are never touched arithmetically (each lane stops at its own length:
// adding a padded 0.0 could turn a -0.0 row result into +0.0).
// What it is for: a second, structurally different bit-exact SpMV (parity / debugging mode), and short irregular rows.
// It streams 12 bytes per stored entry plus padding, so the row-split kernel with its row patterns (8 bytes per entry on
// stencils) stays the product path; tests/test_gpu_spmv_kernels.py compares the two bit for bit.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <numeric>
#include <vector>

#include "pa_internal.h"

constexpr int SELL_C = 64;

struct pa_sell {
  pa_ctx *ctx = nullptr;
  int64_t n_rows = 0, n_cols = 0, nnz = 0, n_slabs = 0, padded = 0;
  int sigma = 1;
  int64_t *d_slab_ptr = nullptr;   // n_slabs + 1: first slot of every slab
  int32_t *d_len = nullptr;        // n_slabs * 64: stored entries of the row in this lane (0: no row)
  int32_t *d_row = nullptr;        // n_slabs * 64: the row this lane owns
  int32_t *d_col = nullptr;        // padded slots, 0-based columns
  double *d_val = nullptr;
};

// y[row] = beta*y[row] + sum_k (val*x[col])*alpha, k ascending: one wavefront per slab, one lane per row
__global__ __launch_bounds__(256) void k_sell_spmv(const int64_t *__restrict__ slab_ptr, const int32_t *__restrict__ len,
                                                   const int32_t *__restrict__ rows, const int32_t *__restrict__ col,
                                                   const double *__restrict__ val, const double *__restrict__ x,
                                                   double *__restrict__ y, int64_t n_slabs, double alpha, double beta) {
  const int64_t slab = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (slab >= n_slabs) return;
  const int lane = threadIdx.x & 63;
  const int n = len[slab * SELL_C + lane];
  const int row = rows[slab * SELL_C + lane];
  const int64_t base = slab_ptr[slab] + lane;
  const int width = (int)((slab_ptr[slab + 1] - slab_ptr[slab]) / SELL_C);
  double acc = 0.0;
  if (n >= 0 && beta != 0.0 && row >= 0) acc = beta * y[row];
  // groups of 4 steps: all loads of a group first (clamped inside the slab: no guard between them), then the gathers,
  // then the ordered adds of the steps this lane really has
  for (int k0 = 0; k0 < width; k0 += 4) {
    double v[4], xv[4];
    int c[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int64_t p = base + (int64_t)min(k0 + j, width - 1) * SELL_C;
      v[j] = __builtin_nontemporal_load(&val[p]);
      c[j] = __builtin_nontemporal_load(&col[p]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) xv[j] = x[c[j]];
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (k0 + j < n) {
        double pr = v[j] * xv[j];
        if (alpha != 1.0) pr = pr * alpha;
        acc = acc + pr;
      }
  }
  if (row >= 0) y[row] = acc;
}

extern "C" int pa_sell_create(pa_ctx *c, int64_t n_rows, int64_t n_cols, int64_t nnz, const void *rowptr, const void *colval,
                              int index_bytes, int index_base, const double *nzval, int sigma, pa_sell **out) {
  PA_REQUIRE(c && out && rowptr, "bad arguments");
  PA_REQUIRE(index_bytes == 4 || index_bytes == 8, "index_bytes must be 4 or 8");
  PA_REQUIRE(index_base == 0 || index_base == 1, "index_base must be 0 or 1");
  PA_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0 && n_rows < (int64_t)2147483000 && n_cols < (int64_t)2147483000, "bad sizes");
  PA_REQUIRE(nnz == 0 || (colval && nzval), "colval/nzval are NULL");
  PA_REQUIRE(sigma >= 1, "sigma must be at least 1");
  auto rd = [&](const void *a, int64_t i) { return index_bytes == 4 ? (int64_t)((const int32_t *)a)[i] : ((const int64_t *)a)[i]; };
  std::vector<int64_t> rp(n_rows + 1);
  for (int64_t r = 0; r <= n_rows; ++r) rp[r] = rd(rowptr, r) - index_base;
  PA_REQUIRE(rp[0] == 0 && rp[n_rows] == nnz, "rowptr does not span [base, base+nnz]");
  for (int64_t r = 0; r < n_rows; ++r) PA_REQUIRE(rp[r + 1] >= rp[r], "rowptr not monotone at row %lld", (long long)r);
  // sigma-sort: inside every window of sigma rows, longest first (stable: equal lengths keep their order)
  std::vector<int32_t> order(n_rows);
  std::iota(order.begin(), order.end(), 0);
  if (sigma > 1)
    for (int64_t w = 0; w < n_rows; w += sigma)
      std::stable_sort(order.begin() + w, order.begin() + std::min<int64_t>(n_rows, w + sigma),
                       [&](int32_t a, int32_t b) { return rp[a + 1] - rp[a] > rp[b + 1] - rp[b]; });
  const int64_t n_slabs = (n_rows + SELL_C - 1) / SELL_C;
  std::vector<int64_t> slab_ptr(n_slabs + 1, 0);
  std::vector<int32_t> len(n_slabs * SELL_C, 0), row(n_slabs * SELL_C, -1);
  for (int64_t s = 0; s < n_slabs; ++s) {
    int64_t w = 0;
    for (int l = 0; l < SELL_C && s * SELL_C + l < n_rows; ++l) {
      const int32_t r = order[s * SELL_C + l];
      row[s * SELL_C + l] = r;
      len[s * SELL_C + l] = (int32_t)(rp[r + 1] - rp[r]);
      w = std::max<int64_t>(w, rp[r + 1] - rp[r]);
    }
    slab_ptr[s + 1] = slab_ptr[s] + w * SELL_C;
  }
  const int64_t padded = slab_ptr[n_slabs];
  std::vector<int32_t> col(std::max<int64_t>(padded, 1), 0);
  std::vector<double> val(std::max<int64_t>(padded, 1), 0.0);
  for (int64_t s = 0; s < n_slabs; ++s)
    for (int l = 0; l < SELL_C; ++l) {
      const int32_t r = row[s * SELL_C + l];
      if (r < 0) continue;
      for (int64_t k = 0; k < len[s * SELL_C + l]; ++k) {
        const int64_t j = rd(colval, rp[r] + k) - index_base;
        PA_REQUIRE(j >= 0 && j < n_cols, "column index out of range at entry %lld", (long long)(rp[r] + k));
        col[slab_ptr[s] + k * SELL_C + l] = (int32_t)j;
        val[slab_ptr[s] + k * SELL_C + l] = nzval[rp[r] + k];
      }
    }
  pa_sell *A = new pa_sell();
  A->ctx = c; A->n_rows = n_rows; A->n_cols = n_cols; A->nnz = nnz; A->n_slabs = n_slabs; A->padded = padded; A->sigma = sigma;
  auto fill = [&]() -> int {
    PA_HIP(hipSetDevice(c->device));
    PA_TRY(pa_dev_alloc(c, (void **)&A->d_val, sizeof(double) * val.size(), PA_MEM_MATRIX));
    PA_TRY(pa_dev_alloc(c, (void **)&A->d_col, sizeof(int32_t) * col.size(), PA_MEM_MATRIX));
    PA_TRY(pa_dev_alloc(c, (void **)&A->d_slab_ptr, sizeof(int64_t) * slab_ptr.size(), PA_MEM_MATRIX));
    PA_TRY(pa_dev_alloc(c, (void **)&A->d_len, sizeof(int32_t) * std::max<size_t>(1, len.size()), PA_MEM_MATRIX));
    PA_TRY(pa_dev_alloc(c, (void **)&A->d_row, sizeof(int32_t) * std::max<size_t>(1, row.size()), PA_MEM_MATRIX));
    PA_HIP(pa_h2d(A->d_val, val.data(), sizeof(double) * val.size()));
    PA_HIP(pa_h2d(A->d_col, col.data(), sizeof(int32_t) * col.size()));
    PA_HIP(pa_h2d(A->d_slab_ptr, slab_ptr.data(), sizeof(int64_t) * slab_ptr.size()));
    if (!len.empty()) {
      PA_HIP(pa_h2d(A->d_len, len.data(), sizeof(int32_t) * len.size()));
      PA_HIP(pa_h2d(A->d_row, row.data(), sizeof(int32_t) * row.size()));
    }
    return PA_OK;
  };
  if (const int st = fill()) {               // half-built: hand back what was taken (pa_dev_free ignores NULL)
    (void)hipGetLastError();
    (void)pa_sell_destroy(A);
    return st;
  }
  *out = A;
  return PA_OK;
}

extern "C" int pa_sell_destroy(pa_sell *A) {
  if (!A) return PA_OK;
  (void)hipSetDevice(A->ctx->device);
  (void)hipStreamSynchronize(A->ctx->s[0]);
  (void)hipStreamSynchronize(A->ctx->s[1]);
  pa_dev_free(A->ctx, A->d_val);
  pa_dev_free(A->ctx, A->d_col);
  pa_dev_free(A->ctx, A->d_slab_ptr);
  pa_dev_free(A->ctx, A->d_len);
  pa_dev_free(A->ctx, A->d_row);
  delete A;
  return PA_OK;
}

extern "C" int pa_sell_info(const pa_sell *A, int64_t *n_slabs, int64_t *padded_entries, int64_t *nnz) {
  PA_REQUIRE(A != nullptr, "sell is NULL");
  if (n_slabs) *n_slabs = A->n_slabs;
  if (padded_entries) *padded_entries = A->padded;
  if (nnz) *nnz = A->nnz;
  return PA_OK;
}

extern "C" int pa_sell_spmv(const pa_sell *A, const pa_vec *x, int xseg, pa_vec *y, int yseg, double alpha, double beta) {
  PA_REQUIRE(A && x && y, "bad arguments");
  auto seg = [](const pa_vec *v, int s, int64_t *off, int64_t *len) {
    if (s == PA_SEG_OWN) { *off = 0; *len = v->n_own; }
    else if (s == PA_SEG_GHOST) { *off = v->n_own; *len = v->n_ghost; }
    else if (s == PA_SEG_LOCAL) { *off = 0; *len = v->n_own + v->n_ghost; }
    else return false;
    return true;
  };
  int64_t xoff, xlen, yoff, ylen;
  PA_REQUIRE(seg(x, xseg, &xoff, &xlen) && seg(y, yseg, &yoff, &ylen), "unknown segment");
  PA_REQUIRE(ylen == A->n_rows, "length(b)=%lld != size(A,1)=%lld", (long long)ylen, (long long)A->n_rows);
  PA_REQUIRE(xlen == A->n_cols, "length(x)=%lld != size(A,2)=%lld", (long long)xlen, (long long)A->n_cols);
  PA_REQUIRE(x->d != y->d || xseg != yseg, "x and y alias");
  if (A->n_slabs == 0) return PA_OK;
  pa_ctx *c = A->ctx;
  PA_HIP(hipSetDevice(c->device));
  hipLaunchKernelGGL(k_sell_spmv, dim3((unsigned)((A->n_slabs + 3) / 4)), dim3(256), 0, c->s[0], A->d_slab_ptr, A->d_len, A->d_row,
                     A->d_col, A->d_val, x->d + xoff, y->d + yoff, A->n_slabs, alpha, beta);
  PA_HIP(hipGetLastError());
  return PA_OK;
}
