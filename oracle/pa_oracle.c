/*
 * pa_oracle.c -- C twins of the oracle's hot loops (TEST INFRASTRUCTURE ONLY).
 *
 * Same loops as oracle/pa_oracle.py, compiled with -O3 -ffp-contract=off so that every
 * product and every sum is rounded separately, in the reference's order.  Used (a) for
 * parity at sizes python loops cannot reach and (b) as bench.py's `cpu_baseline` ("port").
 * Never linked into, or loaded by, the product.
 *
 * All indices are 1-based Int32 as in SparseMatrixCSR{1,Float64,Int32}
 * (/root/reference/HPCG/src/sparse_matrix.jl:115).
 */
#include <stdint.h>

/* src/sparse_utils.jl:649-669  spmv_csr!(b,x,rowptr,colval,nzval) */
void orc_spmv_csr(double *b, const double *x, const int32_t *rowptr, const int32_t *colval,
                  const double *nzval, int64_t nrows) {
  for (int64_t row = 0; row < nrows; ++row) {
    int64_t pini = rowptr[row], pend = rowptr[row + 1];
    double bi = 0.0;
    for (int64_t p = pini; p < pend; ++p) {
      double aij = nzval[p - 1];
      double xj = x[colval[p - 1] - 1];
      bi += aij * xj;
    }
    b[row] = bi;
  }
}

/* SparseMatricesCSR.mul!(y,A,x,alpha,beta) (v0.6, third-party; called from
 * src/p_sparse_matrix.jl:2088,2119,2126,2133,2137): beta-scale, then
 * y[row] += nzval*x[col]*alpha, entry by entry. */
void orc_mul5_csr(double *y, const double *x, const int32_t *rowptr, const int32_t *colval,
                  const double *nzval, int64_t nrows, double alpha, double beta) {
  if (beta != 1.0) {
    if (beta != 0.0) { for (int64_t r = 0; r < nrows; ++r) y[r] *= beta; }
    else             { for (int64_t r = 0; r < nrows; ++r) y[r] = 0.0; }
  }
  for (int64_t row = 0; row < nrows; ++row) {
    int64_t pini = rowptr[row], pend = rowptr[row + 1];
    for (int64_t p = pini; p < pend; ++p)
      y[row] += nzval[p - 1] * x[colval[p - 1] - 1] * alpha;
  }
}

/* SparseMatricesCSR.mul!(y,transpose(A),x,alpha,beta) (v0.6, third-party; called from
 * src/p_sparse_matrix.jl:2150,2159): beta-scale, then for every row of A, y[col] += nzval*x[row]*alpha. */
void orc_mul5_csr_t(double *y, const double *x, const int32_t *rowptr, const int32_t *colval,
                    const double *nzval, int64_t nrows_A, int64_t ncols_A, double alpha, double beta) {
  if (beta != 1.0) {
    if (beta != 0.0) { for (int64_t c = 0; c < ncols_A; ++c) y[c] *= beta; }
    else             { for (int64_t c = 0; c < ncols_A; ++c) y[c] = 0.0; }
  }
  for (int64_t row = 0; row < nrows_A; ++row)
    for (int64_t p = rowptr[row]; p < rowptr[row + 1]; ++p)
      y[colval[p - 1] - 1] += nzval[p - 1] * x[row] * alpha;
}

/* src/p_vector.jl:595-599  buffer_snd.data[p] = values[lid] */
void orc_pack(double *buf, const double *values, const int32_t *lids, int64_t n) {
  for (int64_t p = 0; p < n; ++p) buf[p] = values[lids[p] - 1];
}
/* src/p_vector.jl:605-609 with f = insert (:755) */
void orc_unpack_insert(double *values, const double *buf, const int32_t *lids, int64_t n) {
  for (int64_t p = 0; p < n; ++p) values[lids[p] - 1] = buf[p];
}
/* src/p_vector.jl:605-609 with f = + (:695-697); ascending p, duplicates allowed */
void orc_unpack_add(double *values, const double *buf, const int32_t *lids, int64_t n) {
  for (int64_t p = 0; p < n; ++p) values[lids[p] - 1] = values[lids[p] - 1] + buf[p];
}

/* HPCG/src/sparse_matrix.jl:27-80 restricted to what the CPU baseline needs: the local CSR
 * (columns [own|ghost], 1-based) of ONE part that owns the whole grid (npx=npy=npz=1), written
 * directly in row order.  With a single part there are no ghosts and the COO stream is already
 * row-major with ascending columns, so COO->CSR is the identity on the stream. */
int64_t orc_hpcg_csr_single(int32_t nx, int32_t ny, int32_t nz, int32_t *rowptr, int32_t *colval,
                            double *nzval) {
  int64_t k = 0, row = 0;
  rowptr[0] = 1;
  for (int32_t iz = 0; iz < nz; ++iz)
    for (int32_t iy = 0; iy < ny; ++iy)
      for (int32_t ix = 0; ix < nx; ++ix) {
        int64_t cur = (int64_t)iz * nx * ny + (int64_t)iy * nx + ix;
        for (int sz = -1; sz <= 1; ++sz) {
          if (iz + sz < 0 || iz + sz >= nz) continue;
          for (int sy = -1; sy <= 1; ++sy) {
            if (iy + sy < 0 || iy + sy >= ny) continue;
            for (int sx = -1; sx <= 1; ++sx) {
              if (ix + sx < 0 || ix + sx >= nx) continue;
              int64_t col = cur + (int64_t)sz * nx * ny + (int64_t)sy * nx + sx;
              if (colval) { colval[k] = (int32_t)(col + 1); nzval[k] = (col == cur) ? 26.0 : -1.0; }
              ++k;
            }
          }
        }
        ++row;
        rowptr[row] = (int32_t)(k + 1);
      }
  return k;
}

/* PartitionedSolvers/src/smoothers.jl:144-160 gauss_seidel_sweep!(x,A::SparseMatrixCSR,diagA,b,rows) and
 * :236-259 gauss_seidel_sweep_zero! (only columns < row, no diagonal correction).  A is the UNSPLIT local CSR of one
 * part (n_own x n_local, 1-based), x its local values [own|ghost]; forward = rows 1:n, backward = n:-1:1. */
void orc_gs_sweep(double *x, const int32_t *rowptr, const int32_t *colval, const double *nzval,
                  const double *diag, const double *b, int64_t n, int backward, int zero_guess) {
  for (int64_t k = 0; k < n; ++k) {
    const int64_t row = backward ? n - 1 - k : k;
    double s = b[row];
    for (int64_t p = rowptr[row]; p < rowptr[row + 1]; ++p) {
      const int64_t col = colval[p - 1] - 1;
      if (zero_guess) { if (col < row) s -= nzval[p - 1] * x[col]; }
      else s -= nzval[p - 1] * x[col];
    }
    const double d = diag[row];
    if (!zero_guess) s += d * x[row];
    s = s / d;
    x[row] = s;
  }
}

/* ---- Float32 twins (round 6: the reference's local loops are generic in the element type; its own test runs them in Float32,
 * test/sparse_utils_tests.jl:33-45,72-79).  Every product and sum is rounded to float (the file is compiled -ffp-contract=off and
 * x86-64 gcc evaluates float expressions in float). */
/* src/sparse_utils.jl:649-669 with eltype Float32 */
void orc_spmv_csr_f32(float *b, const float *x, const int32_t *rowptr, const int32_t *colval, const float *nzval, int64_t nrows) {
  for (int64_t row = 0; row < nrows; ++row) {
    float bi = 0.0f;
    for (int64_t p = rowptr[row]; p < rowptr[row + 1]; ++p) {
      float t = nzval[p - 1] * x[colval[p - 1] - 1];
      bi = bi + t;
    }
    b[row] = bi;
  }
}
/* src/sparse_utils.jl:671-690 with eltype Float32 (colptr / rowval / nzval of a CSC matrix, or the CSR arrays for the transposed product) */
void orc_spmv_csc_f32(float *b, const float *x, const int32_t *colptr, const int32_t *rowval, const float *nzval, int64_t nrows, int64_t ncols) {
  for (int64_t r = 0; r < nrows; ++r) b[r] = 0.0f;
  for (int64_t col = 0; col < ncols; ++col) {
    float xj = x[col];
    for (int64_t p = colptr[col]; p < colptr[col + 1]; ++p) {
      float t = nzval[p - 1] * xj;
      b[rowval[p - 1] - 1] = b[rowval[p - 1] - 1] + t;
    }
  }
}
/* SparseMatricesCSR.mul!(y,A,x,alpha,beta) with Float32 arrays and Float32 scalars */
void orc_mul5_csr_f32(float *y, const float *x, const int32_t *rowptr, const int32_t *colval, const float *nzval, int64_t nrows,
                      float alpha, float beta) {
  if (beta != 1.0f) {
    if (beta != 0.0f) { for (int64_t r = 0; r < nrows; ++r) y[r] = y[r] * beta; }
    else              { for (int64_t r = 0; r < nrows; ++r) y[r] = 0.0f; }
  }
  for (int64_t row = 0; row < nrows; ++row)
    for (int64_t p = rowptr[row]; p < rowptr[row + 1]; ++p) {
      float t = nzval[p - 1] * x[colval[p - 1] - 1];
      t = t * alpha;
      y[row] = y[row] + t;
    }
}
