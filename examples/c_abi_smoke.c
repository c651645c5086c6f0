/* c_abi_smoke.c -- libpa_hip.so from plain C (no Python, no PyTorch): two parts of a 1-D Laplacian on one GPU.
 *
 *   part 1 owns rows 1..4, part 2 rows 5..8 of  A = tridiag(-1, 2, -1);  each has one ghost column (the other part's
 *   nearest row).  Data is handed over exactly as PartitionedArrays.jl stores it: 1-based Int32 CSR blocks
 *   (own_own, own_ghost of the split format, src/p_sparse_matrix.jl:588-627) and the VectorAssemblyCache lists
 *   (neighbours, local ids to send / receive, src/p_vector.jl:418-426).  Then mul!(c,a,b) = pa_mul_all and a dot.
 *
 *   build: gcc -std=c99 -I include examples/c_abi_smoke.c -L partitionedarrays.jl_amd -lpa_hip -o c_abi_smoke
 *   run  : LD_LIBRARY_PATH=partitionedarrays.jl_amd:/opt/rocm/lib ./c_abi_smoke
 */
#include <stdio.h>
#include <stdlib.h>

#include "pa_hip.h"

#define CHECK(call)                                                              \
  do {                                                                           \
    int st_ = (call);                                                            \
    if (st_ != PA_OK) {                                                          \
      fprintf(stderr, "%s -> %d: %s\n", #call, st_, pa_last_error());            \
      return 1;                                                                  \
    }                                                                            \
  } while (0)

int main(void) {
  int ndev = 0;
  CHECK(pa_device_count(&ndev));
  if (ndev == 0) { fprintf(stderr, "no HIP device\n"); return 2; }
  pa_ctx *ctx;
  CHECK(pa_ctx_create(0, &ctx));

  /* own_own (4 x 4) is the same for both parts; own_ghost (4 x 1): part 1's last row, part 2's first row */
  const int32_t oo_rowptr[5] = {1, 3, 6, 9, 11};
  const int32_t oo_colval[10] = {1, 2, 1, 2, 3, 2, 3, 4, 3, 4};
  const double oo_nzval[10] = {2, -1, -1, 2, -1, -1, 2, -1, -1, 2};
  const int32_t oh_rowptr[2][5] = {{1, 1, 1, 1, 2}, {1, 2, 2, 2, 2}};
  const int32_t oh_colval[1] = {1};
  const double oh_nzval[1] = {-1};

  pa_csr *oo[2], *oh[2];
  pa_vec *b[2], *c[2];
  pa_plan *plan[2];
  pa_matrix *A[2];
  for (int p = 0; p < 2; ++p) {
    CHECK(pa_csr_create(ctx, 4, 4, 10, oo_rowptr, oo_colval, 4, 1, oo_nzval, &oo[p]));
    CHECK(pa_csr_create(ctx, 4, 1, 1, oh_rowptr[p], oh_colval, 4, 1, oh_nzval, &oh[p]));
    CHECK(pa_vec_create(ctx, 4, 1, &b[p]));
    CHECK(pa_vec_create(ctx, 4, 0, &c[p]));
    /* assembly orientation: snd = my ghosts grouped by owner, rcv = my own ids that the neighbour ghosts */
    const int32_t nbr[1] = {p == 0 ? 2 : 1};
    const int32_t ptrs[2] = {1, 2};
    const int32_t idx_snd[1] = {5};                 /* local id of my only ghost */
    const int32_t idx_rcv[1] = {p == 0 ? 4 : 1};    /* my own row next to the interface */
    CHECK(pa_plan_create(ctx, p + 1, 5, 1, nbr, ptrs, idx_snd, 1, nbr, ptrs, idx_rcv, 1, &plan[p]));
    CHECK(pa_matrix_create(ctx, oo[p], oh[p], plan[p], &A[p]));
    double x[5];
    for (int i = 0; i < 4; ++i) x[i] = (double)(4 * p + i + 1) * (4 * p + i + 1);   /* x[g] = g^2 */
    x[4] = -1000.0;                                                                  /* ghost: must be overwritten */
    CHECK(pa_vec_upload(b[p], x, 0, 5));
  }
  CHECK(pa_mul_all(A, 2, c, b, 1.0, 0.0));          /* consistent!(b) + own*own + own*ghost for both parts */

  int bad = 0;
  for (int p = 0; p < 2; ++p) {
    double y[4];
    CHECK(pa_vec_download(c[p], y, 0, 4));
    for (int i = 0; i < 4; ++i) {
      const int g = 4 * p + i + 1;
      /* (A x)[g] = -(g-1)^2 + 2 g^2 - (g+1)^2 = -2 inside; boundary rows miss one neighbour */
      double want = -2.0;
      if (g == 1) want = 2.0 * 1 - 4;
      if (g == 8) want = -49.0 + 2.0 * 64;
      if (y[i] != want) { fprintf(stderr, "row %d: got %g, want %g\n", g, y[i], want); bad = 1; }
    }
  }
  double d = 0.0, total = 0.0;
  for (int p = 0; p < 2; ++p) { CHECK(pa_vec_dot(c[p], c[p], &d)); total += d; }
  if (total != 4.0 + 6 * 4.0 + 79.0 * 79.0) { fprintf(stderr, "dot: got %g\n", total); bad = 1; }

  for (int p = 0; p < 2; ++p) {
    pa_matrix_destroy(A[p]); pa_plan_destroy(plan[p]); pa_vec_destroy(b[p]); pa_vec_destroy(c[p]);
    pa_csr_destroy(oo[p]); pa_csr_destroy(oh[p]);
  }
  pa_ctx_destroy(ctx);
  printf(bad ? "c_abi_smoke: FAILED\n" : "c_abi_smoke: OK (mul! on 2 parts and dot through the C ABI, version %d)\n", pa_version());
  return bad;
}
