// pa_mg.hip -- Gauss-Seidel smoother and grid transfer (the HPCG multigrid preconditioner)
// (one of the units pa_device.hip was split into in round 5; compiled with -ffp-contract=off like all of them)
#include <hip/hip_runtime.h>

#include <atomic>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <thread>
#include <cstring>
#include <memory>
#include <numeric>
#include <iterator>
#include <string>
#include <vector>

#include "pa_internal.h"
#include "pa_setup.h"
#include "pa_dev_kernels.h"

// ------------------------------------------------------------------------------------------------
// Gauss-Seidel smoother (level scheduled) and grid transfer: HPCG multigrid preconditioner
// ------------------------------------------------------------------------------------------------
extern "C" int pa_gs_create(pa_ctx *c, int64_t n_own, int64_t n_local, int64_t nnz, const int32_t *rowptr,
                            const int32_t *colval, const double *nzval, int index_base, int ordering, pa_gs **out) {
  PA_REQUIRE(c && out && rowptr && (nnz == 0 || (colval && nzval)), "bad arguments");
  PA_REQUIRE(ordering == PA_GS_SEQUENTIAL || ordering == PA_GS_MULTICOLOR, "unknown ordering %d", ordering);
  PA_REQUIRE(index_base == 0 || index_base == 1, "index_base must be 0 or 1");
  PA_REQUIRE(n_own >= 0 && n_local >= n_own && nnz < (int64_t)2147483000, "bad sizes");
  std::vector<int32_t> rp(n_own + 1), col(nnz), level(n_own, 0);
  std::vector<double> diag(n_own, 0.0);
  for (int64_t r = 0; r <= n_own; ++r) rp[r] = rowptr[r] - index_base;
  PA_REQUIRE(rp[0] == 0 && rp[n_own] == nnz, "rowptr does not span the stored entries");
  int32_t n_levels = 0;
  for (int64_t r = 0; r < n_own; ++r) {
    int32_t lv = 0;
    bool has_diag = false;
    for (int64_t p = rp[r]; p < rp[r + 1]; ++p) {
      const int64_t j = (int64_t)colval[p] - index_base;
      PA_REQUIRE(j >= 0 && j < n_local, "column out of range at entry %lld", (long long)p);
      col[p] = (int32_t)j;
      if (j == r) { diag[r] = nzval[p]; has_diag = true; }
      if (j < r) lv = std::max(lv, level[j] + 1);
    }
    PA_REQUIRE(has_diag && diag[r] != 0.0, "row %lld has no (non-zero) diagonal entry", (long long)r);
    if (ordering == PA_GS_MULTICOLOR) {
      // greedy colouring in natural order: smallest colour no already-coloured own neighbour uses (<= 64 colours)
      uint64_t used = 0;
      for (int64_t p = rp[r]; p < rp[r + 1]; ++p)
        if (col[p] < r && level[col[p]] < 64) used |= 1ull << level[col[p]];
      lv = 0;
      while (lv < 63 && (used >> lv) & 1ull) ++lv;
    }
    level[r] = lv;
    n_levels = std::max(n_levels, lv + 1);
  }
  // the parallel schedule equals the sequential sweep only if every own column j > i of row i is swept later;
  // a colouring only needs neighbours to differ
  for (int64_t r = 0; r < n_own; ++r)
    for (int64_t p = rp[r]; p < rp[r + 1]; ++p)
      PA_REQUIRE(!(col[p] > r && col[p] < n_own) ||
                     (ordering == PA_GS_SEQUENTIAL ? level[col[p]] > level[r] : level[col[p]] != level[r]),
                 "own x own pattern is not structurally symmetric at (%lld,%d): level scheduling would change the sweep order",
                 (long long)r, col[p]);
  pa_gs *g = new pa_gs();
  g->ctx = c; g->n_own = n_own; g->n_local = n_local; g->nnz = nnz;
  g->lev_ptr.assign(n_levels + 1, 0);
  for (int64_t r = 0; r < n_own; ++r) g->lev_ptr[level[r] + 1]++;
  for (int l = 0; l < n_levels; ++l) {
    g->max_level_rows = std::max<int64_t>(g->max_level_rows, g->lev_ptr[l + 1]);
    g->lev_ptr[l + 1] += g->lev_ptr[l];
  }
  std::vector<int32_t> rows(n_own), fill(g->lev_ptr.begin(), g->lev_ptr.end() - (n_levels ? 1 : 0));
  for (int64_t r = 0; r < n_own; ++r) rows[fill[level[r]]++] = (int32_t)r;  // ascending row inside a level
  PA_HIP(hipSetDevice(c->device));
  PA_TRY(upload_i32(rp, &g->d_rowptr));
  PA_TRY(upload_i32(col, &g->d_col));
  PA_TRY(upload_i32(rows, &g->d_rows));
  PA_HIP(pa_raw_malloc(&g->d_val, sizeof(double) * std::max<int64_t>(1, nnz)));
  PA_HIP(pa_raw_malloc(&g->d_diag, sizeof(double) * std::max<int64_t>(1, n_own)));
  if (nnz) PA_HIP(pa_h2d(g->d_val, nzval, sizeof(double) * nnz));
  if (n_own) PA_HIP(pa_h2d(g->d_diag, diag.data(), sizeof(double) * n_own));
  *out = g;
  return PA_OK;
}

extern "C" int pa_gs_destroy(pa_gs *g) {
  if (!g) return PA_OK;
  (void)hipSetDevice(g->ctx->device);
  (void)hipStreamSynchronize(g->ctx->s[0]);
  for (auto &e : g->graphs) (void)hipGraphExecDestroy(e.exec);
  (void)pa_raw_free(g->d_rowptr); (void)pa_raw_free(g->d_col); (void)pa_raw_free(g->d_rows); (void)pa_raw_free(g->d_val); (void)pa_raw_free(g->d_diag);
  delete g;
  return PA_OK;
}

extern "C" int pa_gs_info(const pa_gs *g, int64_t *n_levels, int64_t *max_rows) {
  PA_REQUIRE(g != nullptr, "gs is NULL");
  if (n_levels) *n_levels = (int64_t)g->lev_ptr.size() - 1;
  if (max_rows) *max_rows = g->max_level_rows;
  return PA_OK;
}

extern "C" int pa_gs_sweep(pa_gs *g, pa_vec *x, const pa_vec *b, int backward, int zero_guess) {
  PA_REQUIRE(g && x && b, "bad arguments");
  PA_REQUIRE(x->n_own + x->n_ghost == g->n_local && x->n_own == g->n_own, "x does not match the matrix (%lld own, %lld local)",
             (long long)g->n_own, (long long)g->n_local);
  PA_REQUIRE(b->n_own == g->n_own, "b does not match the matrix");
  PA_REQUIRE(x->d != b->d, "x and b alias");
  pa_ctx *c = g->ctx;
  PA_HIP(hipSetDevice(c->device));
  const int nl = (int)g->lev_ptr.size() - 1;
  auto launch_levels = [&]() {
    for (int k = 0; k < nl; ++k) {
      const int l = backward ? nl - 1 - k : k;
      const int n = g->lev_ptr[l + 1] - g->lev_ptr[l];
      if (n == 0) continue;
      hipLaunchKernelGGL(k_gs_level, dim3((n + 127) / 128), dim3(128), 0, c->s[0], x->d, b->d, g->d_rowptr, g->d_col, g->d_val,
                         g->d_diag, g->d_rows + g->lev_ptr[l], n, zero_guess);
    }
  };
  // A sweep is a chain of hundreds of tiny dependent launches.  PA_GS_GRAPH=1 captures it once per
  // (x, b, direction, zero_guess) into a hipGraph and replays it; measured neutral on MI355X (47.4 vs 47.6 ms per
  // MG-PCG iteration at 128^3: the cost is the ~7 us dependent-kernel boundary + row latency on the GPU, not the host
  // launch), so eager launches stay the default.
  static const bool use_graph = getenv("PA_GS_GRAPH") && atoi(getenv("PA_GS_GRAPH")) == 1;
  if (!use_graph || nl < 8) {
    launch_levels();
    PA_HIP(hipGetLastError());
    return PA_OK;
  }
  for (auto &e : g->graphs)
    if (e.x == x->d && e.b == b->d && e.backward == (backward != 0) && e.zero_guess == (zero_guess != 0)) {
      PA_HIP(hipGraphLaunch(e.exec, c->s[0]));
      return PA_OK;
    }
  hipGraph_t graph = nullptr;
  PA_HIP(hipStreamBeginCapture(c->s[0], hipStreamCaptureModeThreadLocal));
  launch_levels();
  PA_HIP(hipStreamEndCapture(c->s[0], &graph));
  hipGraphExec_t exec = nullptr;
  PA_HIP(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
  PA_HIP(hipGraphDestroy(graph));
  if (g->graphs.size() >= 16) {  // bounded cache: drop the oldest
    (void)hipGraphExecDestroy(g->graphs.front().exec);
    g->graphs.erase(g->graphs.begin());
  }
  g->graphs.push_back({x->d, b->d, backward != 0, zero_guess != 0, exec});
  PA_HIP(hipGraphLaunch(exec, c->s[0]));
  return PA_OK;
}

extern "C" int pa_host_greedy_coloring(int64_t n_own, const int32_t *rowptr, const int32_t *colval, int index_base,
                                       int32_t *color, int32_t *n_colors) {
  PA_REQUIRE(rowptr && color && n_colors && (index_base == 0 || index_base == 1), "bad arguments");
  int32_t nc = 0;
  for (int64_t r = 0; r < n_own; ++r) {
    uint64_t used = 0;
    for (int64_t p = rowptr[r] - index_base; p < rowptr[r + 1] - index_base; ++p) {
      const int64_t j = (int64_t)colval[p] - index_base;
      if (j < r && color[j] < 64) used |= 1ull << color[j];
    }
    int32_t c = 0;
    while (c < 63 && (used >> c) & 1ull) ++c;
    color[r] = c;
    nc = std::max(nc, c + 1);
  }
  *n_colors = nc;
  return PA_OK;
}

extern "C" int pa_rowset_create(pa_ctx *c, int64_t n, const int32_t *rows, int index_base, pa_rowset **out) {
  PA_REQUIRE(c && out && n >= 0 && (n == 0 || rows) && (index_base == 0 || index_base == 1), "bad arguments");
  std::vector<int32_t> h(n);
  for (int64_t i = 0; i < n; ++i) {
    h[i] = rows[i] - index_base;
    PA_REQUIRE(h[i] >= 0, "negative row id at %lld", (long long)i);
  }
  pa_rowset *r = new pa_rowset();
  r->ctx = c; r->n = n;
  PA_HIP(hipSetDevice(c->device));
  PA_TRY(upload_i32(h, &r->d_rows));
  *out = r;
  return PA_OK;
}

extern "C" int pa_rowset_destroy(pa_rowset *r) {
  if (!r) return PA_OK;
  (void)hipSetDevice(r->ctx->device);
  (void)hipStreamSynchronize(r->ctx->s[0]);
  (void)pa_raw_free(r->d_rows);
  delete r;
  return PA_OK;
}

extern "C" int pa_gs_color_update(pa_rowset *r, pa_vec *x, const pa_vec *b, pa_vec *t, const pa_vec *diag) {
  PA_REQUIRE(r && x && b && t && diag, "bad arguments");
  if (r->n == 0) return PA_OK;
  PA_HIP(hipSetDevice(r->ctx->device));
  hipLaunchKernelGGL(k_gs_color_update, dim3((r->n + 255) / 256), dim3(256), 0, r->ctx->s[0], x->d, b->d, t->d, diag->d, r->d_rows,
                     (int)r->n);
  PA_HIP(hipGetLastError());
  return PA_OK;
}

extern "C" int pa_transfer_create(pa_ctx *c, int64_t n_coarse, const int32_t *f2c, int index_base, pa_transfer **out) {
  PA_REQUIRE(c && out && n_coarse >= 0 && (n_coarse == 0 || f2c), "bad arguments");
  PA_REQUIRE(index_base == 0 || index_base == 1, "index_base must be 0 or 1");
  std::vector<int32_t> h(n_coarse);
  for (int64_t i = 0; i < n_coarse; ++i) {
    h[i] = f2c[i] - index_base;
    PA_REQUIRE(h[i] >= 0, "negative fine index at %lld", (long long)i);
  }
  pa_transfer *t = new pa_transfer();
  t->ctx = c; t->n_coarse = n_coarse;
  PA_HIP(hipSetDevice(c->device));
  PA_TRY(upload_i32(h, &t->d_f2c));
  *out = t;
  return PA_OK;
}

extern "C" int pa_transfer_destroy(pa_transfer *t) {
  if (!t) return PA_OK;
  (void)hipSetDevice(t->ctx->device);
  (void)hipStreamSynchronize(t->ctx->s[0]);
  (void)pa_raw_free(t->d_f2c);
  delete t;
  return PA_OK;
}

extern "C" int pa_transfer_restrict(pa_transfer *t, pa_vec *rc, const pa_vec *rf, const pa_vec *axf) {
  PA_REQUIRE(t && rc && rf && axf, "bad arguments");
  PA_REQUIRE(rc->n_own + rc->n_ghost >= t->n_coarse && rf->n_own + rf->n_ghost == axf->n_own + axf->n_ghost, "vector sizes");
  if (t->n_coarse == 0) return PA_OK;
  PA_HIP(hipSetDevice(t->ctx->device));
  hipLaunchKernelGGL(k_restrict, dim3((t->n_coarse + 255) / 256), dim3(256), 0, t->ctx->s[0], rc->d, rf->d, axf->d, t->d_f2c,
                     (int)t->n_coarse);
  PA_HIP(hipGetLastError());
  return PA_OK;
}

// Fused residual + restriction (the reference computes Axf = A*x on every fine row and then keeps one row in eight,
// HPCG/src/mg_preconditioner.jl:320-321,224-237): `rows` holds the stored entries of the fine rows f2c only, and the
// row-split kernel's epilogue writes r_c[i] = r_f[f2c[i]] - (A x_f)[f2c[i]] -- the same row sums, one eighth of the work.
extern "C" int pa_transfer_attach_rows(pa_transfer *t, const pa_csr *rows) {
  PA_REQUIRE(t && rows, "bad arguments");
  PA_REQUIRE(rows->ctx == t->ctx, "transfer and block live in different contexts");
  PA_REQUIRE(rows->next == nullptr, "a block stored in several slabs (>= 2^31 entries) is not supported by the fused restriction");
  PA_REQUIRE(rows->compact && rows->n_crows == t->n_coarse,
             "the block must store exactly the %lld fine rows of the coarse grid (it stores %lld%s)", (long long)t->n_coarse,
             (long long)rows->n_crows, rows->compact ? "" : ", not compacted");
  PA_HIP(hipSetDevice(t->ctx->device));
  std::vector<int32_t> a(t->n_coarse), b(t->n_coarse);
  PA_HIP(hipMemcpy(a.data(), t->d_f2c, sizeof(int32_t) * t->n_coarse, hipMemcpyDeviceToHost));
  PA_HIP(hipMemcpy(b.data(), rows->d_row_ids, sizeof(int32_t) * t->n_coarse, hipMemcpyDeviceToHost));
  for (int64_t i = 0; i < t->n_coarse; ++i)
    PA_REQUIRE(a[i] == b[i], "stored row %lld of the block is fine row %d, the transfer expects %d", (long long)i, b[i], a[i]);
  t->rows = rows;
  return PA_OK;
}

extern "C" int pa_transfer_restrict_fused(pa_transfer *t, pa_vec *rc, const pa_vec *rf, const pa_vec *xf) {
  PA_REQUIRE(t && rc && rf && xf, "bad arguments");
  PA_REQUIRE(t->rows != nullptr, "no row block attached (pa_transfer_attach_rows)");
  const pa_csr *A = t->rows;
  PA_REQUIRE(rc->n_own + rc->n_ghost >= t->n_coarse, "coarse vector too short");
  PA_REQUIRE(rf->n_own == A->t_rows && xf->n_own + xf->n_ghost == A->n_cols, "fine vector sizes do not match the block");
  PA_REQUIRE(rc->d != xf->d && rc->d != rf->d, "r_c aliases a fine vector");
  if (A->n_chunks == 0) return PA_OK;
  pa_ctx *c = t->ctx;
  PA_HIP(hipSetDevice(c->device));
  if (const int pm = pa_pell_mode(A)) {                // a pattern block: one lane per row (pa_pell.h)
    PA_TRY(pa_pell_launch(A, pm, 2, xf->d, nullptr, 1.0, 0.0, rc->d, rf->d, nullptr, c->s[0]));
    return PA_OK;
  }
  const int cpx = (int)((A->n_chunks + 7) / 8);
#define PA_LAUNCH_RR(C16, PAT, VD)                                                                                       \
  hipLaunchKernelGGL((k_spmv_rowsplit<SPMV_BLK, SPMV_NPT, SPMV_NT, C16, PAT, 2, VD>), dim3(cpx * 8), dim3(SPMV_BLK), 0,    \
                     c->s[0], A->d_crp, A->d_col, A->d_col16, A->d_win, A->d_pdesc, A->d_pdelta, A->d_val,           \
                     (const double *)xf->d, (double *)nullptr, A->d_chunk_rp, A->d_row_ids, (int)A->n_chunks, cpx,  \
                     1.0, 0.0, rc->d, (const double *)rf->d, (const double *)nullptr, A->d_code, A->d_dict,                \
                     (const int *)nullptr, (int)A->n_cols - 1)
  const int sel_ = (A->use_pattern ? (A->compact ? 2 : 1) : 0) * 2 + (A->use_c16 ? 1 : 0);
  if (A->use_vdict) {
    const bool vd_two = pa_vd_two(A);
    if (c->capturing) { const_cast<pa_csr *>(A)->vd_captured = true; if (vd_two) const_cast<pa_csr *>(A)->vd_captured_two = true; }
    switch (sel_) {
      case 5: if (vd_two) PA_LAUNCH_RR(true, 2, 2); else PA_LAUNCH_RR(true, 2, 1); break;
      case 4: if (vd_two) PA_LAUNCH_RR(false, 2, 2); else PA_LAUNCH_RR(false, 2, 1); break;
      case 3: if (vd_two) PA_LAUNCH_RR(true, 1, 2); else PA_LAUNCH_RR(true, 1, 1); break;
      case 2: if (vd_two) PA_LAUNCH_RR(false, 1, 2); else PA_LAUNCH_RR(false, 1, 1); break;
      case 1: if (vd_two) PA_LAUNCH_RR(true, 0, 2); else PA_LAUNCH_RR(true, 0, 1); break;
      default: if (vd_two) PA_LAUNCH_RR(false, 0, 2); else PA_LAUNCH_RR(false, 0, 1); break;
    }
  } else {
    switch (sel_) {
      case 5: PA_LAUNCH_RR(true, 2, 0); break;
      case 4: PA_LAUNCH_RR(false, 2, 0); break;
      case 3: PA_LAUNCH_RR(true, 1, 0); break;
      case 2: PA_LAUNCH_RR(false, 1, 0); break;
      case 1: PA_LAUNCH_RR(true, 0, 0); break;
      default: PA_LAUNCH_RR(false, 0, 0); break;
    }
  }
#undef PA_LAUNCH_RR
  PA_HIP(hipGetLastError());
  return PA_OK;
}

extern "C" int pa_transfer_prolongate(pa_transfer *t, pa_vec *xf, const pa_vec *xc) {
  PA_REQUIRE(t && xf && xc, "bad arguments");
  PA_REQUIRE(xc->n_own + xc->n_ghost >= t->n_coarse, "coarse vector too short");
  if (t->n_coarse == 0) return PA_OK;
  PA_HIP(hipSetDevice(t->ctx->device));
  hipLaunchKernelGGL(k_prolongate, dim3((t->n_coarse + 255) / 256), dim3(256), 0, t->ctx->s[0], xf->d, xc->d, t->d_f2c,
                     (int)t->n_coarse);
  PA_HIP(hipGetLastError());
  return PA_OK;
}
