// pa_setup.h -- device-side set-up of a CSR block's column encodings (pa_setup.hip); private to libpa_hip.so.
#ifndef PA_SETUP_H
#define PA_SETUP_H

#include <cstdint>
#include <vector>

#include "pa_internal.h"

// The column streams of one block exactly as k_spmv_rowsplit reads them, built ON THE DEVICE from the block's raw CSR
// arrays (0-based Int32 row pointers and columns already in HBM): the device twin of pa_encode_columns
// (pa_spmv_kernel.h), array for array and byte for byte (tests/test_gpu_setup.py::test_device_side_encoding_equals_the_host_s).
struct pa_dev_streams {
  bool use_pattern = false, use_c16 = false, full = true;
  int32_t *d_pdesc = nullptr;    // n_chunks * PA_PDESC_INTS (only when use_pattern)
  int32_t *d_pdelta = nullptr;   // n_pdelta ints
  int32_t *d_win = nullptr;      // n_chunks * PA_C16_WINDOWS (only when use_c16)
  uint16_t *d_c16 = nullptr;     // n_c16_slots entries (padding included)
  int32_t *d_c32 = nullptr;      // compacted 32-bit columns, n_c32_slots entries + padding (only when !full)
  int64_t n_pdelta = 0, n_c16_slots = 0, n_c32_slots = 0;
  int64_t n_pattern = 0, n_c16 = 0, n_c32 = 0;   // chunks by column encoding
  int64_t nnz_c16 = 0, nnz_c32 = 0;              // stored entries whose chunk reads the 16-bit stream / 32-bit columns
  bool pad_products = false;
  double ms = 0;                                 // device time of the encoding (events)
};

// d_crp[nc + 1], d_col[nnz (+ padding)], d_row_ids[nc] or NULL, d_chunk_row[n_chunks + 1]: device arrays, 0-based.
// Output buffers are taken with pa_dev_alloc(..., PA_MEM_MATRIX); on failure everything taken so far is handed back.
int pa_dev_encode_columns(pa_ctx *c, const int32_t *d_crp, const int32_t *d_col, const int32_t *d_row_ids, int64_t nc,
                          int64_t nnz, const int32_t *d_chunk_row, int64_t n_chunks, int cap, bool want_pattern,
                          bool want_c16, bool compact_streams, pa_dev_streams &S);
void pa_dev_streams_free(pa_ctx *c, pa_dev_streams &S);

// the row split of pa_build_chunks (pa_spmv_kernel.h) on the device: chunk_row (host) = the chunk boundaries, the host
// loop's array exactly
int pa_dev_row_split(pa_ctx *c, const int32_t *d_crp, int64_t nc, int cap, int max_rows, int align_rows,
                     std::vector<int32_t> &chunk_row, int64_t *n_long);

// per-chunk statistics of the x-window planning (pa_xw_scan_chunks of pa_spmv_xwin.h) computed on the device; host arrays of
// n_chunks entries each
int pa_dev_xw_chunk_stats(pa_ctx *c, const int32_t *d_crp, const int32_t *d_col, const int32_t *d_chunk_row, const int32_t *d_win,
                          int64_t n_chunks, int max_cap, int32_t *cmin, int32_t *cmax, int32_t *lines);

// pa_csr.hip: a pa_csr from entries that are already in HBM (0-based; the arrays are copied, the caller keeps its own)
int pa_csr_from_device(pa_ctx *c, int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t *d_rowptr, const int32_t *d_col,
                       const double *d_val, pa_csr **out);

// the same from rows the caller counted (and, when most are empty, compacted: d_row_ids) on the device; crp = the final row
// pointer on the host (n_nonempty + 1 entries with d_row_ids, n_rows + 1 without; moved from).  One slab only.
int pa_csr_from_device_rows(pa_ctx *c, int64_t n_rows, int64_t n_cols, int64_t nnz, int64_t n_nonempty, std::vector<int32_t> &crp,
                            const int32_t *d_row_ids, const int32_t *d_col, const double *d_val, pa_csr **out);

// pa_transpose.hip: 0-based (row, column) of every stored entry of ONE slab in storage order, decoded on the device from whatever
// column encoding the slab keeps (d_row / d_col: nnz entries each, device)
int pa_dev_decode_entries(const pa_csr *A, int32_t *d_row, int32_t *d_col);
// the same entries in the same order with column j renamed map[j] (host, A->n_cols entries, -1 = absent): see pa_transpose.hip
int pa_csr_create_remapped(const pa_csr *A, const int32_t *map, int64_t n_cols_new, pa_csr **out);

// pa_transpose.hip: when A (one slab, unstructured rows on the 16-bit stream) has chunks whose columns span more than the sliding x
// window holds (and PA_SPMV_COLSPLIT != 0), *out = the same block as a chain of column pieces (pa_internal.h: accumulate / colsplit /
// d_src), else *out = NULL.  force_pieces > 0 (tests): that many pieces whatever the spans.
int pa_csr_colsplit_if_wide(const pa_csr *A, pa_csr **out, int force_pieces = 0);
extern thread_local int pa_tls_piece_build;
extern thread_local const std::vector<int32_t> *pa_tls_row_breaks;
int pa_scatter_from_device_dest(pa_ctx *c, int64_t n_dst, int64_t n_src, const int32_t *d_dest, pa_scatter **out);

// min / max of a device Int32 array (column range check of an uploaded block)
int pa_dev_minmax_i32(pa_ctx *c, const int32_t *d, int64_t n, int32_t *mn, int32_t *mx);

#endif
