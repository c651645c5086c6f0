// pa_pell_launch.h -- the launches of k_spmv_pell (pa_pell.h), one value stream per translation unit.
//
// Reference loops: spmv_csr! src/sparse_utils.jl:649-669, mul!(y,A,x,alpha,beta) as called at src/p_sparse_matrix.jl:2088.
// The kernel is a template over unroll x value stream x row compaction x epilogue x runs of three x alpha: some two hundred
// instantiations.  The HIP runtime loads a translation unit's code objects when its first kernel is launched, so the three value
// streams -- fp64 (VM 0: the headline), one bit per entry (VM 1: HPCG and its multigrid), one byte per entry (VM 2) -- are
// instantiated in three translation units (pa_pell_v0.hip, _v1, _v2): a process pays for the stream it runs (the HPCG driver's
// first set-up is timed as first met: tools/hpcg_driver.py, profiles/r06_hpcg256.log).
#ifndef PA_PELL_LAUNCH_H
#define PA_PELL_LAUNCH_H

#include "pa_pell.h"

template <int U, int VM, int EPI>
static void pell_launch_uv(const pa_csr *A, const pa_pell_dev &D, int nblk, int bpx, const double *x, double *y, double alpha, double beta,
                           double *gs_x, const double *gs_b, const double *gs_diag, hipStream_t st) {
  // alpha = 1 (every epilogue form, and the plain product of mul!(c,a,b)) is compiled in: no multiply-and-select per product
#define PA_PELL_GO(UU, CC, RR)                                                                                                                     \
  do {                                                                                                                                             \
    if (alpha == 1.0)                                                                                                                              \
      hipLaunchKernelGGL((k_spmv_pell<UU, VM, CC, EPI, RR, true>), dim3(nblk), dim3(256), 0, st, D, x, y, bpx, alpha, beta, gs_x, gs_b, gs_diag);   \
    else if constexpr (EPI == 0)                                                                                                                   \
      hipLaunchKernelGGL((k_spmv_pell<UU, VM, CC, EPI, RR, false>), dim3(nblk), dim3(256), 0, st, D, x, y, bpx, alpha, beta, gs_x, gs_b, gs_diag);  \
  } while (0)
  if constexpr (U == 9 && EPI != 1) {
    if constexpr (VM == 1) {
      // (the whole row as ONE group of 27 -- every gather requested before the first product -- measured SLOWER than three groups of
      //  nine, 0.2212 against 0.2127 ms at 256^3, 66 VGPRs: PA_SPMV_PELL_BITS_U27=1 keeps the experiment reachable)
      static const bool u27 = getenv("PA_SPMV_PELL_BITS_U27") && atoi(getenv("PA_SPMV_PELL_BITS_U27")) != 0;
      if (A->pell->runs3 && !A->compact && A->pell->max_w <= 27 && u27) { PA_PELL_GO(27, false, true); return; }
    }
    if (A->pell->runs3 && !A->compact) { PA_PELL_GO(9, false, true); return; }
  }
  if constexpr (U == 9) {
    // (a row-compacted block -- a colour of the smoother, the rows a restriction keeps -- or the Gauss-Seidel update: runs of three in
    //  the slabs of a class only, the lean form of pa_pell_slab_fast; every other slab one gather per entry as before)
    if (A->pell->runs3 && A->pell->d_plane && A->ctx->sw.pell_lean && alpha == 1.0) {
      if (A->compact) hipLaunchKernelGGL((k_spmv_pell<9, VM, true, EPI, true, true>), dim3(nblk), dim3(256), 0, st, D, x, y, bpx, alpha, beta, gs_x, gs_b, gs_diag);
      else hipLaunchKernelGGL((k_spmv_pell<9, VM, false, EPI, true, true>), dim3(nblk), dim3(256), 0, st, D, x, y, bpx, alpha, beta, gs_x, gs_b, gs_diag);
      return;
    }
  }
  if (A->compact) PA_PELL_GO(U, true, false);
  else PA_PELL_GO(U, false, false);
#undef PA_PELL_GO
}

template <int VM, int EPI>
static void pell_launch_epi(const pa_csr *A, const pa_pell_dev &D, int nblk, int bpx, const double *x, double *y, double alpha,
                            double beta, double *gs_x, const double *gs_b, const double *gs_diag, hipStream_t st) {
  switch (A->pell->U) {
    case 9: pell_launch_uv<9, VM, EPI>(A, D, nblk, bpx, x, y, alpha, beta, gs_x, gs_b, gs_diag, st); break;
    case 7: pell_launch_uv<7, VM, EPI>(A, D, nblk, bpx, x, y, alpha, beta, gs_x, gs_b, gs_diag, st); break;
    case 5: pell_launch_uv<5, VM, EPI>(A, D, nblk, bpx, x, y, alpha, beta, gs_x, gs_b, gs_diag, st); break;
    default: pell_launch_uv<4, VM, EPI>(A, D, nblk, bpx, x, y, alpha, beta, gs_x, gs_b, gs_diag, st); break;
  }
}

// every epilogue form of value stream VM (what pa_pell_launch_v0 / _v1 / _v2 are made of)
template <int VM>
static void pell_launch_vm(const pa_csr *A, const pa_pell_dev &D, int epi, int nblk, int bpx, const double *x, double *y, double alpha,
                           double beta, double *gs_x, const double *gs_b, const double *gs_diag, hipStream_t st) {
  switch (epi) {
    case 0: pell_launch_epi<VM, 0>(A, D, nblk, bpx, x, y, alpha, beta, gs_x, gs_b, gs_diag, st); break;
    case 1: pell_launch_epi<VM, 1>(A, D, nblk, bpx, x, y, alpha, beta, gs_x, gs_b, gs_diag, st); break;
    case 2: pell_launch_epi<VM, 2>(A, D, nblk, bpx, x, y, alpha, beta, gs_x, gs_b, gs_diag, st); break;
    default: pell_launch_epi<VM, 3>(A, D, nblk, bpx, x, y, alpha, beta, gs_x, gs_b, gs_diag, st); break;
  }
}

#define PA_PELL_LAUNCH_ARGS const pa_csr *A, const pa_pell_dev &D, int epi, int nblk, int bpx, const double *x, double *y, double alpha, \
                            double beta, double *gs_x, const double *gs_b, const double *gs_diag, hipStream_t st
void pa_pell_launch_v0(PA_PELL_LAUNCH_ARGS);
void pa_pell_launch_v1(PA_PELL_LAUNCH_ARGS);
void pa_pell_launch_v2(PA_PELL_LAUNCH_ARGS);

#endif
