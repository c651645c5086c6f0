// pa_fused.hip -- mul!(c,a,b) of one part as ONE launch (round 5, VERDICT r04 "Next" #1).
//
// Reference: mul!(c,a,b[,alpha,beta])  src/p_sparse_matrix.jl:2090-2142
//              t = consistent!(b);  own(c) = beta*own(c) + alpha*A_oo*own(b);  wait(t);  own(c) += alpha*A_oh*ghost(b)
//            consistent! / wait(t)      src/p_vector.jl:595-611, 747-755
//
// Round 4 queued, per part: the push launch, own x own, [event hop] own x ghost from the receive buffer, the unpack.  Beside a
// 128^3 own x own (88 us) that chain cost +19 % (config 3 whole 1.19 x own x own, config 5 whole 1.21 x): launch boundaries, a
// drained and refilled GPU around a 3 us kernel, cross-stream event hops of ~8 us.  Here the part's rows are cut differently:
//
//   INTERIOR rows  (no stored entry in own_ghost)  : summed from own_own alone -- they never wait for anything;
//   BOUNDARY rows  (>= 1 stored entry in own_ghost): summed from a small block of their own, `bd`, that holds for each of them
//                  its own_own entries FOLLOWED BY its own_ghost entries (columns renamed: own column j stays j, ghost column
//                  -> n_own + its position in consistent!'s receive buffer).  One lane adds beta*c[row], then the row's products
//                  in that stored order: exactly the additions of spmv!(own_own) followed by muladd!(own_ghost) -- same bits.
//
// k_mul_fused is one grid: the first blocks are own x own's chunks, the row-split kernel as it is (pa_spmv_kernel.h, FX 1), except
// that they neither store nor (beta != 0) read the boundary rows; the LAST blocks are bd's chunks (FX 2: columns >= n_own gather
// the receive buffer).  No block of the launch depends on another block of the launch.  What the tail blocks do depend on is the
// arrival of b's ghost values:
//   * all parts in one process (pa_mul_all): the push launch in front (k_push_unpack: it delivers into the receive buffers AND
//     writes b's ghost entries, so consistent!(b) is complete with it) -- launches per step: P + 1, one stream, no events;
//   * one part per process: see pa_mul_fused_ipc below (the tail acquires the arrival flags inside the launch).
// The own_own entries of boundary rows are read twice (by the chunk they sit in, which still needs them to find its other rows'
// products, and by bd): 1-3 % of the matrix stream of a 3-D part.
#include "pa_dev_util.h"

#include "pa_setup.h"
#include "pa_spmv_kernel.h"
#include "pa_pell.h"
#include "pa_push_dev.h"

using namespace pa_util;


// ---- bd: the boundary rows' block, built in HBM from the two blocks' own encodings ---------------------------------------------
struct kf_block {                                  // what decoding a stored entry's column needs (kt_decode's arguments)
  const int *crp, *chunk_row, *col32, *win, *pdesc, *pdelta;
  const unsigned short *col16;
  const double *val;
  int use_pattern, use_c16, n_chunks;
};
static kf_block kf_of(const pa_csr *A) {
  kf_block k;
  k.crp = A->d_crp; k.chunk_row = A->d_chunk_row; k.col32 = A->d_col; k.win = A->d_win; k.pdesc = A->d_pdesc; k.pdelta = A->d_pdelta;
  k.col16 = A->d_col16; k.val = A->d_val;
  k.use_pattern = A->use_pattern ? 1 : 0; k.use_c16 = A->use_c16 ? 1 : 0; k.n_chunks = (int)A->n_chunks;
  return k;
}
// column of stored entry p of stored row r (the arithmetic of kt_decode, pa_transpose.hip, for one entry)
__device__ static int kf_col(const kf_block &B, int r, int p) {
  int lo = 0, hi = B.n_chunks - 1;                 // the chunk that holds stored row r: the last c with chunk_row[c] <= r
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (B.chunk_row[mid] <= r) lo = mid; else hi = mid - 1;
  }
  const int chunk = lo;
  const int p0 = B.crp[B.chunk_row[chunk]], p1 = B.crp[B.chunk_row[chunk + 1]];
  const bool is_long = p1 - (p0 & ~1) > PA_SPMV_CHUNK_NNZ;
  const int *d = B.use_pattern ? B.pdesc + (size_t)chunk * PA_PDESC_INTS : nullptr;
  const int nseg = d ? d[0] : 0;
  const int sh16 = (d && nseg <= 0) ? d[1] : 0, sh32 = (d && nseg <= 0) ? d[2] : 0;
  if (nseg > 0) {
    const int q = p - p0;
    int s = 0;
    if (nseg > 1 && q >= d[1]) s = 1;
    if (nseg > 2 && q >= d[2]) s = 2;
    if (nseg > 3 && q >= d[3]) s = 3;
    const int qs = s ? d[s] : 0;
    const int Lw = d[8 + s], L = Lw & 255, stride = (Lw >> 8) ? (Lw >> 8) : 1;
    const int t = q - qs, rr = t / L, kk = t - rr * L;
    return d[4 + s] + rr * stride + B.pdelta[(size_t)d[12 + s] * PA_PAT_MAXLEN + kk];
  }
  if (B.use_c16 && !is_long && B.win[(size_t)chunk * PA_C16_WINDOWS] >= 0) {
    const unsigned code = B.col16[(size_t)p + sh16];
    return B.win[(size_t)chunk * PA_C16_WINDOWS + (code >> 12)] + (int)(code & 4095u);
  }
  return B.col32[(size_t)p + sh32];
}

// rows[i] = the i-th boundary row; hrow[i] = its stored row in the own_ghost twin.  len[i] = its entries in both blocks.
__global__ void kf_lens(const int *__restrict__ rows, const int *__restrict__ hrow, int n, const int *__restrict__ crp_oo,
                        const int *__restrict__ crp_oh, int *__restrict__ len) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  len[i] = (crp_oo[rows[i] + 1] - crp_oo[rows[i]]) + (crp_oh[hrow[i] + 1] - crp_oh[hrow[i]]);
}
__global__ void kf_fill(const int *__restrict__ rows, const int *__restrict__ hrow, int n, kf_block OO, kf_block OH, int n_own,
                        const int *__restrict__ brp, int *__restrict__ out_col, double *__restrict__ out_val) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int q = brp[i];
  const int r = rows[i];
  for (int p = OO.crp[r]; p < OO.crp[r + 1]; ++p, ++q) { out_col[q] = kf_col(OO, r, p); out_val[q] = OO.val[p]; }
  const int h = hrow[i];
  for (int p = OH.crp[h]; p < OH.crp[h + 1]; ++p, ++q) { out_col[q] = n_own + kf_col(OH, h, p); out_val[q] = OH.val[p]; }
}
// the values of bd again, in place: bd's i-th row holds the values of own_own's row rows[i] followed by those of the twin's stored
// row hrow[i], in stored order (kf_fill's order; the pattern of neither block changes under a value update)
__global__ void kf_refill(const int *__restrict__ rows, const int *__restrict__ hrow, int n, const int *__restrict__ crp_oo,
                          const double *__restrict__ val_oo, const int *__restrict__ crp_oh, const double *__restrict__ val_oh,
                          const int *__restrict__ brp, double *__restrict__ out_val) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  int q = brp[i];
  const int r = rows[i], h = hrow[i];
  for (int p = crp_oo[r]; p < crp_oo[r + 1]; ++p, ++q) out_val[q] = val_oo[p];
  for (int p = crp_oh[h]; p < crp_oh[h + 1]; ++p, ++q) out_val[q] = val_oh[p];
}
__global__ void kf_mask(const int *__restrict__ rows, int n, unsigned *__restrict__ mask) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicOr(&mask[rows[i] >> 5], 1u << (rows[i] & 31));
}

void pa_matrix_fused_release(pa_matrix *m) {
  if (m->bd) pa_csr_destroy(m->bd);
  if (m->d_rowmask) (void)pa_raw_free(m->d_rowmask);
  if (m->d_bd_hrow) (void)pa_raw_free(m->d_bd_hrow);
  m->bd = nullptr; m->d_rowmask = nullptr; m->d_bd_hrow = nullptr; m->n_bd_rows = 0;
}

// bd follows a value update of own_own / own_ghost IN PLACE: same pattern, same addresses -- a recorded graph that
// holds bd's pointers multiplies with the new values at its next replay, and nothing it reads is ever freed under it.  Needs the twin
// of own_ghost to be current (matrix_rb refreshes it in place first).  Returns PA_OK without doing anything when bd cannot follow in
// place (a rebuilt twin): the next eager product rebuilds it -- unless a graph has recorded it, which is an error to say aloud.
int pa_matrix_fused_refresh(pa_matrix *m) {
  if (!m->bd) return PA_OK;
  const pa_csr *oo = m->oo, *oh = m->oh_rb;
  if (m->bd_epoch_oo == oo->val_epoch && m->bd_epoch_oh == m->oh->val_epoch) return PA_OK;
  const bool can = oh && m->rb_epoch == m->oh->val_epoch && m->d_bd_hrow && m->bd->compact && m->bd->d_row_ids && !m->bd->next;
  if (!can) {
    PA_REQUIRE(!m->bd_captured, "the values of a block changed under a recorded fused product and its boundary rows' block cannot "
                                "follow in place: record the graph again (pa_graph_begin / pa_graph_end)");
    return PA_OK;
  }
  pa_ctx *c = m->ctx;
  PA_HIP(hipSetDevice(c->device));
  const int n = (int)m->n_bd_rows;
  hipLaunchKernelGGL(kf_refill, grid1(n, 64), dim3(64), 0, c->s[0], m->bd->d_row_ids, m->d_bd_hrow, n, oo->d_crp, oo->d_val, oh->d_crp,
                     oh->d_val, m->bd->d_crp, m->bd->d_val);
  PA_HIP(hipGetLastError());
  m->bd->val_epoch++;
  m->bd_epoch_oo = oo->val_epoch; m->bd_epoch_oh = m->oh->val_epoch;
  return PA_OK;
}

// Builds (or, after a value update of either block, rebuilds) what the fused launch needs.  m->bd == NULL afterwards: this handle
// stays on the separate launches (not an error) -- own_own is not a plain row-split block (x windows, column pieces, slabs, a
// row-compacted block), no twin of own_ghost on receive-buffer positions (see matrix_rb), or PA_MUL_FUSED=0.
int pa_matrix_fused_build(pa_matrix *m) {
  const pa_csr *oo = m->oo, *oh = m->oh_rb;
  if (m->bd && m->bd_epoch_oo == oo->val_epoch && m->bd_epoch_oh == m->oh->val_epoch) return PA_OK;
  pa_ctx *c = m->ctx;
  if (m->bd) {                                               // the values changed: in place when possible (also inside a capture)
    PA_TRY(pa_matrix_fused_refresh(m));
    if (m->bd_epoch_oo == oo->val_epoch && m->bd_epoch_oh == m->oh->val_epoch) return PA_OK;
  }
  if (c->capturing) return PA_OK;                            // (a stale bd is never used: see pa_matrix_fused_ready)
  if (m->bd) {
    PA_REQUIRE(!m->bd_captured, "a recorded fused product holds this handle's boundary rows' block: record the graph again");
    PA_HIP(hipStreamSynchronize(c->s[0]));
    PA_HIP(hipStreamSynchronize(c->s[1]));
    pa_matrix_fused_release(m);
    m->fuse_tried = false;
  }
  if (m->fuse_tried) return PA_OK;
  m->fuse_tried = true;
  if (!c->sw.mul_fused || !oh || m->transposed) return PA_OK;
  if (oo->next || oo->compact || oo->n_xw_groups > 0 || oo->pad_products || oo->n_chunks == 0 || oh->next || oh->nnz == 0) return PA_OK;
  PA_HIP(hipSetDevice(c->device));
  hipStream_t s = c->s[0];
  // the boundary rows, ascending, and where each sits in the twin: from the twin's row pointer (small: the part's surface)
  std::vector<int32_t> hcrp((size_t)oh->n_crows + 1), hids;
  PA_HIP(hipMemcpyAsync(hcrp.data(), oh->d_crp, sizeof(int32_t) * hcrp.size(), hipMemcpyDeviceToHost, s));
  if (oh->compact) {
    hids.resize((size_t)oh->n_crows);
    PA_HIP(hipMemcpyAsync(hids.data(), oh->d_row_ids, sizeof(int32_t) * hids.size(), hipMemcpyDeviceToHost, s));
  }
  PA_HIP(hipStreamSynchronize(s));
  std::vector<int32_t> rows, hrow;
  for (int64_t k = 0; k < oh->n_crows; ++k)
    if (hcrp[k + 1] > hcrp[k]) { rows.push_back(oh->compact ? hids[k] : (int32_t)k); hrow.push_back((int32_t)k); }
  const int n = (int)rows.size();
  if (n == 0) return PA_OK;
  scratch sc;
  int32_t *d_rows = nullptr, *d_hrow = nullptr, *d_len = nullptr, *d_brp = nullptr, *d_col = nullptr;
  double *d_val = nullptr;
  PA_TRY(sc.get(&d_rows, (size_t)n));
  PA_TRY(sc.get(&d_hrow, (size_t)n));
  PA_TRY(sc.get(&d_len, (size_t)n + 1));
  PA_TRY(sc.get(&d_brp, (size_t)n + 1));
  PA_HIP(hipMemcpyAsync(d_rows, rows.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
  PA_HIP(hipMemcpyAsync(d_hrow, hrow.data(), sizeof(int32_t) * n, hipMemcpyHostToDevice, s));
  PA_HIP(hipMemsetAsync(d_len, 0, sizeof(int32_t) * ((size_t)n + 1), s));
  hipLaunchKernelGGL(kf_lens, grid1(n), dim3(256), 0, s, d_rows, d_hrow, n, oo->d_crp, oh->d_crp, d_len);
  PA_TRY(scan_exclusive(sc, s, d_len, d_brp, (size_t)n + 1));
  std::vector<int32_t> brp((size_t)n + 1);
  PA_TRY(d2h(s, brp.data(), d_brp, (size_t)n + 1));
  const int64_t nnz = brp[n];
  PA_TRY(sc.get(&d_col, (size_t)nnz + 8));
  PA_TRY(sc.get(&d_val, (size_t)nnz + 8));
  hipLaunchKernelGGL(kf_fill, grid1(n, 64), dim3(64), 0, s, d_rows, d_hrow, n, kf_of(oo), kf_of(oh), (int)oo->n_cols, d_brp, d_col, d_val);
  PA_HIP(hipGetLastError());
  PA_HIP(hipStreamSynchronize(s));
  pa_csr *bd = nullptr;
  ++pa_tls_plain_encoding;
  const int st = pa_csr_from_device_rows(c, oo->n_rows, oo->n_cols + oh->n_cols, nnz, n, brp, d_rows, d_col, d_val, &bd);
  --pa_tls_plain_encoding;
  if (st != PA_OK) { (void)hipGetLastError(); return PA_OK; }               // (no room: the separate launches serve)
  const size_t words = ((size_t)oo->n_rows + 31) / 32 + 1;
  if (pa_raw_malloc(&m->d_rowmask, sizeof(unsigned) * words) != hipSuccess) { (void)hipGetLastError(); pa_csr_destroy(bd); return PA_OK; }
  PA_HIP(hipMemsetAsync(m->d_rowmask, 0, sizeof(unsigned) * words, s));
  hipLaunchKernelGGL(kf_mask, grid1(n), dim3(256), 0, s, d_rows, n, m->d_rowmask);
  PA_HIP(hipGetLastError());
  PA_HIP(hipStreamSynchronize(s));
  if (pa_raw_malloc(&m->d_bd_hrow, sizeof(int32_t) * (size_t)n) == hipSuccess)      // (without it: no in-place refresh, a rebuild)
    PA_HIP(hipMemcpy(m->d_bd_hrow, d_hrow, sizeof(int32_t) * (size_t)n, hipMemcpyDeviceToDevice));
  else (void)hipGetLastError();
  m->bd = bd; m->n_bd_rows = n;
  m->bd_epoch_oo = oo->val_epoch; m->bd_epoch_oh = m->oh->val_epoch;
  return PA_OK;
}

// *yes = 1: this handle's products run as one launch (decided at the first product; 0 before it); *n_boundary_rows = rows of the
// part with stored entries in own_ghost (the launch's tail)
extern "C" int pa_matrix_fused(const pa_matrix *m, int *yes, int64_t *n_boundary_rows) {
  PA_REQUIRE(m && yes, "bad arguments");
  *yes = pa_matrix_fused_ready(m) ? 1 : 0;
  if (n_boundary_rows) *n_boundary_rows = m->n_bd_rows;
  return PA_OK;
}

extern "C" int pa_ctx_fused_launches(const pa_ctx *c, int64_t *all, int64_t *with_exchange) {
  PA_REQUIRE(c != nullptr, "bad arguments");
  if (all) *all = c->n_fused;
  if (with_exchange) *with_exchange = c->n_fused_exchange;
  return PA_OK;
}

bool pa_matrix_fused_ready(const pa_matrix *m) {
  return m->bd && m->bd_epoch_oo == m->oo->val_epoch && m->bd_epoch_oh == m->oh->val_epoch && m->oh_rb &&
         m->rb_epoch == m->oh->val_epoch;
}

// ---- the launch ------------------------------------------------------------------------------------------------------------------
struct pa_fused_args {
  int n_main_blocks;                               // the grid's first blocks: own x own's chunks (8 * chunks per XCD)
  int n_tail_chunks, n_tail_blocks;                // bd's chunks, walked by the grid's last blocks (block t: chunks t, t + blocks, ...)
  const unsigned *rowmask;                         // bit r: row r is a boundary row
  // bd: Int32 columns, compacted rows
  const int *b_crp, *b_col, *b_chunk_rp, *b_row_ids;
  const double *b_val;
  const double *rbuf;                              // consistent!'s receive buffer (buffer_rcv of the reversed cache)
  int n_split, b_max_col;
  double *bvec;                                    // b's local values (the tail's unpack)
  // the exchange inside the launch (XCH): what to do sits in DEVICE memory (static per link; read by the blocks that need it, when
  // they need it -- as kernel arguments its fifteen words stayed in scalar registers for the whole kernel: 84 SGPRs, and from 81 on
  // the hardware admits 7 workgroups of 256 per CU instead of 8), this exchange's sequence number and the pushing blocks by value
  const pa_fused_comm *X;
  unsigned long long seq;
  int n_push_blocks;
};

// One part per process: the exchange lives INSIDE the launch.
//   first blocks : pack + push over the ipc link (stores into the neighbours' receive buffers, arrival flags behind them), then
//                  they take own x own chunks like every other block;
//   tail blocks  : one lane per awaited neighbour polls its arrival flag (bounded: the link's time-out raises the status word and
//                  the block gives up), ONE system-scope acquire, then the boundary rows from the receive buffer, the unpack of
//                  b's ghost entries, and -- the last tail block to finish -- the acknowledgement to the senders.
// The tail blocks are the LAST blocks of the grid: by the time one is dispatched every own x own block is running or done, so a
// spinning tail block never holds a slot an own x own block of THIS launch is waiting for.  They are at most PA_FUSED_TAIL_BLOCKS
// (default 1024, half the GPU's 2048 resident workgroups): the other half stays free for what may have to run for the arrival to
// happen at all -- RCCL's receive kernels on the comm stream; with SEVERAL RANKS ON ONE GPU (the tests, never a production run) the
// other ranks' launches with their pushing blocks, which is why those runs set it to a few dozen.  A neighbour that is ahead or in
// step costs no spinning at all: the tail is dispatched ~a product's duration after the launch began, its flags are long raised.
template <bool C16, int PAT, int VD, bool XCH>
__global__ __launch_bounds__(256) void k_mul_fused(
    const int *__restrict__ crp, const int *__restrict__ col, const unsigned short *__restrict__ col16,
    const int *__restrict__ win, const int *__restrict__ pdesc, const int *__restrict__ pdelta, const double *__restrict__ val,
    const double *__restrict__ x, double *__restrict__ y, const int *__restrict__ chunk_rp, int n_chunks, int chunks_per_xcd,
    double alpha, double beta, const unsigned char *__restrict__ code, const double *__restrict__ dict, int max_col,
    const pa_fused_args F) {
  constexpr int BLK = 256, NPT = PA_SPMV_CHUNK_NNZ / 256;
  __shared__ __attribute__((aligned(16))) double prod[BLK * NPT];
  __shared__ double wsum[1];
  __shared__ int ok;
  const int b = blockIdx.x;
  if (b >= F.n_main_blocks) {
    const int tb = b - F.n_main_blocks;
    if (XCH) {
      const int n_wait = F.X->n_wait;
      if (n_wait > 0) {
        const unsigned long long *flags = F.X->flags;
        const int32_t *wait_idx = F.X->wait_idx;
        const long long ticks = F.X->ticks;
        if (threadIdx.x == 0) ok = 1;
        __syncthreads();
        for (int i = threadIdx.x; i < n_wait; i += BLK)
          if (!flag_wait(flags + wait_idx[i], F.seq, ticks)) { atomicExch(F.X->status, 1); ok = 0; }
        __syncthreads();
        if (!ok) return;                                     // (a neighbour is gone or out of step: the status word says so)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");       // what the neighbours stored in front of their flags is visible now
        __syncthreads();
      }
    }
    pa_fx fx;
    fx.x2 = F.rbuf; fx.n_split = F.n_split;
    for (int t = tb; t < F.n_tail_chunks; t += F.n_tail_blocks) {
      pa_rowsplit_chunk<BLK, NPT, true, false, 0, 0, false, 4, false, 2>(
          prod, wsum, F.b_crp, F.b_col, nullptr, nullptr, nullptr, nullptr, F.b_val, x, y, F.b_chunk_rp, F.b_row_ids, alpha, beta, nullptr,
          nullptr, nullptr, nullptr, nullptr, F.b_max_col, t, fx);
      __syncthreads();                                       // (the next chunk's products go where this one's row sums read)
    }
    if (XCH) {
      const int u_n = F.X->u_n;
      const int32_t *u_idx = F.X->u_idx;
      for (int k = tb * BLK + (int)threadIdx.x; k < u_n; k += F.n_tail_blocks * BLK) F.bvec[u_idx[k]] = F.rbuf[k];
      const int n_ack = F.X->n_ack;
      if (n_ack > 0) {
        __syncthreads();                                     // every lane of this block has read what it needs of the buffer
        if (threadIdx.x == 0) {
          unsigned *t_done = F.X->t_done;
          const unsigned done = atomicAdd(t_done, 1u);
          if (done == (unsigned)F.n_tail_blocks - 1) {       // the last tail block: the senders may overwrite the buffer
            *t_done = 0;
            __threadfence_system();
            unsigned long long *const *ack_dst = F.X->ack_dst;
            for (int i = 0; i < n_ack; ++i) flag_store(ack_dst[i], F.seq);
          }
        }
      }
    }
    return;
  }
  if (XCH && b < F.n_push_blocks) {
    const pa_fused_comm X = *F.X;
    (void)pa_push_ipc_block(&ok, X.p_idx, X.p_n, X.p_segs, X.p_nseg, x, F.seq, X.p_done, X.ticks, X.status, b, F.n_push_blocks);
  }
  const bool backwards = chunks_per_xcd < 0;
  if (backwards) chunks_per_xcd = -chunks_per_xcd;
  int chunk = (b & 7) * chunks_per_xcd + (b >> 3);           // XCD-aware, as k_spmv_rowsplit
  if (chunk >= n_chunks || (b >> 3) >= chunks_per_xcd) return;
  if (backwards) chunk = n_chunks - 1 - chunk;
  pa_fx fx;
  fx.rowmask = F.rowmask;
  pa_rowsplit_chunk<BLK, NPT, true, C16, PAT, 0, VD, 4, false, 1>(prod, wsum, crp, col, col16, win, pdesc, pdelta, val, x, y, chunk_rp,
                                                                  nullptr, alpha, beta, nullptr, nullptr, nullptr, code, dict, max_col,
                                                                  chunk, fx);
}

// The same launch with own x own's INTERIOR rows on the pattern-ELL kernel (pa_pell.h; round 6): the first blocks are groups of four
// slabs (one lane per row, rows of the boundary bitmap left alone), the last blocks the boundary rows' block as before.  With the
// one-bit value stream the row-split main blocks had become the slower half of the fused launch.
void pa_pell_describe(const pa_csr *A, int mode, pa_pell_dev *D, int *U, bool *runs3, int64_t *n_slabs);
template <int U, int VM, bool R3, bool XCH, bool A1>
__global__ __launch_bounds__(256) void k_mul_fused_pell(const pa_pell_dev P, const double *__restrict__ x, double *__restrict__ y, int bpx,
                                                        double alpha, double beta, const pa_fused_args F) {
  constexpr int BLK = 256, NPT = PA_SPMV_CHUNK_NNZ / 256;
  __shared__ __attribute__((aligned(16))) double prod[BLK * NPT];
  __shared__ double wsum[1];
  __shared__ int ok;
  __shared__ double sdict[VM == 2 ? PA_VDICT_MAX : 1];
  if (VM == 2) {                                     // (one byte per entry: the workgroup's copy of the dictionary, before anyone leaves)
    if (threadIdx.x < PA_VDICT_MAX) sdict[threadIdx.x] = P.dict[threadIdx.x];
    __syncthreads();
  }
  const int b = blockIdx.x;
  if (b >= F.n_main_blocks) {
    const int tb = b - F.n_main_blocks;
    if (XCH) {
      const int n_wait = F.X->n_wait;
      if (n_wait > 0) {
        const unsigned long long *flags = F.X->flags;
        const int32_t *wait_idx = F.X->wait_idx;
        const long long ticks = F.X->ticks;
        if (threadIdx.x == 0) ok = 1;
        __syncthreads();
        for (int i = threadIdx.x; i < n_wait; i += BLK)
          if (!flag_wait(flags + wait_idx[i], F.seq, ticks)) { atomicExch(F.X->status, 1); ok = 0; }
        __syncthreads();
        if (!ok) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
        __syncthreads();
      }
    }
    pa_fx fx;
    fx.x2 = F.rbuf; fx.n_split = F.n_split;
    for (int t = tb; t < F.n_tail_chunks; t += F.n_tail_blocks) {
      pa_rowsplit_chunk<BLK, NPT, true, false, 0, 0, false, 4, false, 2>(
          prod, wsum, F.b_crp, F.b_col, nullptr, nullptr, nullptr, nullptr, F.b_val, x, y, F.b_chunk_rp, F.b_row_ids, alpha, beta, nullptr,
          nullptr, nullptr, nullptr, nullptr, F.b_max_col, t, fx);
      __syncthreads();
    }
    if (XCH) {
      const int u_n = F.X->u_n;
      const int32_t *u_idx = F.X->u_idx;
      for (int k = tb * BLK + (int)threadIdx.x; k < u_n; k += F.n_tail_blocks * BLK) F.bvec[u_idx[k]] = F.rbuf[k];
      const int n_ack = F.X->n_ack;
      if (n_ack > 0) {
        __syncthreads();
        if (threadIdx.x == 0) {
          unsigned *t_done = F.X->t_done;
          const unsigned done = atomicAdd(t_done, 1u);
          if (done == (unsigned)F.n_tail_blocks - 1) {
            *t_done = 0;
            __threadfence_system();
            unsigned long long *const *ack_dst = F.X->ack_dst;
            for (int i = 0; i < n_ack; ++i) flag_store(ack_dst[i], F.seq);
          }
        }
      }
    }
    return;
  }
  if (XCH && b < F.n_push_blocks) {
    const pa_fused_comm X = *F.X;
    (void)pa_push_ipc_block(&ok, X.p_idx, X.p_n, X.p_segs, X.p_nseg, x, F.seq, X.p_done, X.ticks, X.status, b, F.n_push_blocks);
  }
  const bool backwards = bpx < 0;
  if (backwards) bpx = -bpx;
  int g = (b & 7) * bpx + (b >> 3);
  const int n_groups = (P.n_slabs + 3) >> 2;
  if (g >= n_groups) return;
  if (backwards) g = n_groups - 1 - g;
  const int slab = __builtin_amdgcn_readfirstlane(g * 4 + (int)(threadIdx.x >> 6));
  if (slab >= P.n_slabs) return;
  pa_fx fx;
  fx.rowmask = F.rowmask;
  pa_pell_slab<U, VM, false, 0, 1, R3, A1>(P, slab, x, y, alpha, beta, nullptr, nullptr, nullptr, fx, sdict);
}

// own(c) = beta*own(c) + alpha*(A_oo*own(b) + A_oh*ghost(b)) of one part in one launch on stream st.  comm == NULL: the receive
// buffer of consistent!(b) holds b's ghost values by stream order (the push launch is in front, pa_mul_all); else the exchange
// happens inside the launch as *comm says.
int pa_mul_fused_launch(pa_matrix *m, pa_vec *c, pa_vec *b, double alpha, double beta, hipStream_t st, const pa_fused_comm *comm,
                        unsigned long long seq, int n_push_blocks, int max_tail_blocks) {
  const pa_csr *S = m->oo, *B = m->bd;
  pa_plan *p = m->plan;
  pa_fused_args F;
  int cpx = (int)((S->n_chunks + 7) / 8);
  F.n_main_blocks = cpx * 8;
  F.rowmask = m->d_rowmask;
  F.b_crp = B->d_crp; F.b_col = B->d_col; F.b_chunk_rp = B->d_chunk_rp; F.b_row_ids = B->d_row_ids; F.b_val = B->d_val;
  F.rbuf = p->snd.d_buf;
  F.n_split = (int)S->n_cols; F.b_max_col = (int)B->n_cols - 1;
  F.bvec = b->d;
  F.X = comm; F.seq = seq; F.n_push_blocks = n_push_blocks;
  F.n_tail_chunks = (int)B->n_chunks;
  F.n_tail_blocks = max_tail_blocks > 0 ? std::min(F.n_tail_chunks, max_tail_blocks) : F.n_tail_chunks;
  PA_REQUIRE(n_push_blocks <= F.n_main_blocks, "more pushing blocks than own x own has chunks");
  const int n_tail = F.n_tail_blocks;
  if (const int pm = pa_pell_mode(S)) {                      // own x own has pattern-ELL storage: its interior rows run there
    pa_pell_dev D;
    int U = 0;
    bool r3 = false;
    int64_t n_slabs = 0;
    pa_pell_describe(S, pm, &D, &U, &r3, &n_slabs);
    if (U == 9 || U == 7) {
      const int n_groups = (int)((n_slabs + 3) / 4);
      int bpx = (n_groups + 7) / 8;
      F.n_main_blocks = bpx * 8;
      PA_REQUIRE(n_push_blocks <= F.n_main_blocks, "more pushing blocks than own x own has slab groups");
      if (m->ctx->sw.spmv_alternate && ((const_cast<pa_csr *>(S)->n_launched++) & 1)) bpx = -bpx;
      if (pm == 2 && m->ctx->capturing) { const_cast<pa_csr *>(S)->vd_captured = true; const_cast<pa_csr *>(S)->vd_captured_two = true; }
      if (pm == 3 && m->ctx->capturing) const_cast<pa_csr *>(S)->vd_captured = true;
      // (runs of three with alpha = 1 -- mul!(c,a,b) -- compiled in: the interior rows' lean form, pa_pell_slab_fast)
#define PA_LAUNCH_FP(UU, VM, R3, XCH)                                                                                                       \
  do {                                                                                                                                      \
    if (R3 && alpha == 1.0)                                                                                                                 \
      hipLaunchKernelGGL((k_mul_fused_pell<UU, VM, R3, XCH, R3>), dim3(F.n_main_blocks + n_tail), dim3(256), 0, st, D, (const double *)b->d, \
                         c->d, bpx, alpha, beta, F);                                                                                        \
    else                                                                                                                                    \
      hipLaunchKernelGGL((k_mul_fused_pell<UU, VM, R3, XCH, false>), dim3(F.n_main_blocks + n_tail), dim3(256), 0, st, D,                    \
                         (const double *)b->d, c->d, bpx, alpha, beta, F);                                                                  \
  } while (0)
#define PA_FP_CASES(XCH)                                                                              \
      if (U == 7) { if (pm == 3) PA_LAUNCH_FP(7, 2, false, XCH); else if (pm == 2) PA_LAUNCH_FP(7, 1, false, XCH); else PA_LAUNCH_FP(7, 0, false, XCH); } \
      else if (r3) { if (pm == 3) PA_LAUNCH_FP(9, 2, true, XCH); else if (pm == 2) PA_LAUNCH_FP(9, 1, true, XCH); else PA_LAUNCH_FP(9, 0, true, XCH); }   \
      else { if (pm == 3) PA_LAUNCH_FP(9, 2, false, XCH); else if (pm == 2) PA_LAUNCH_FP(9, 1, false, XCH); else PA_LAUNCH_FP(9, 0, false, XCH); }
      if (comm) { PA_FP_CASES(true) } else { PA_FP_CASES(false) }
#undef PA_FP_CASES
#undef PA_LAUNCH_FP
      PA_HIP(hipGetLastError());
      if (m->ctx->capturing) m->bd_captured = true;
      m->ctx->n_fused++;
      if (comm) m->ctx->n_fused_exchange++;
      return PA_OK;
    }
  }
  if (m->ctx->sw.spmv_alternate && ((const_cast<pa_csr *>(S)->n_launched++) & 1)) cpx = -cpx;
#define PA_LAUNCH_FUSED(C16, PAT, VD)                                                                                           \
  hipLaunchKernelGGL((k_mul_fused<C16, PAT, VD, XCH>), dim3(F.n_main_blocks + n_tail), dim3(256), 0, st, S->d_crp, S->d_col, S->d_col16, \
                     S->d_win, S->d_pdesc, S->d_pdelta, S->d_val, (const double *)b->d, c->d, S->d_chunk_rp, (int)S->n_chunks, cpx,   \
                     alpha, beta, S->d_code, S->d_dict, (int)S->n_cols - 1, F)
  const int sel = (S->use_pattern ? 4 : 0) + (S->use_c16 ? 2 : 0) + (S->use_vdict ? 1 : 0);
  const bool vd_two = pa_vd_two(S);
  if (S->use_vdict && m->ctx->capturing) { const_cast<pa_csr *>(S)->vd_captured = true; if (vd_two) const_cast<pa_csr *>(S)->vd_captured_two = true; }
#define PA_FUSED_CASES(XCH_)                                \
  { constexpr bool XCH = XCH_;                              \
    switch (sel) {                                          \
      case 7: if (vd_two) PA_LAUNCH_FUSED(true, 1, 2); else PA_LAUNCH_FUSED(true, 1, 1); break;        \
      case 6: PA_LAUNCH_FUSED(true, 1, 0); break;       \
      case 5: if (vd_two) PA_LAUNCH_FUSED(false, 1, 2); else PA_LAUNCH_FUSED(false, 1, 1); break;       \
      case 4: PA_LAUNCH_FUSED(false, 1, 0); break;      \
      case 3: if (vd_two) PA_LAUNCH_FUSED(true, 0, 2); else PA_LAUNCH_FUSED(true, 0, 1); break;        \
      case 2: PA_LAUNCH_FUSED(true, 0, 0); break;       \
      case 1: if (vd_two) PA_LAUNCH_FUSED(false, 0, 2); else PA_LAUNCH_FUSED(false, 0, 1); break;       \
      default: PA_LAUNCH_FUSED(false, 0, 0); break;     \
    } }
  if (comm) PA_FUSED_CASES(true) else PA_FUSED_CASES(false)
#undef PA_FUSED_CASES
#undef PA_LAUNCH_FUSED
  PA_HIP(hipGetLastError());
  if (m->ctx->capturing) m->bd_captured = true;
  m->ctx->n_fused++;
  if (comm) m->ctx->n_fused_exchange++;
  return PA_OK;
}

// ---- one part per process over RCCL ---------------------------------------------------------------------------------------------
// The transport stays what `north_star` names -- pack, ONE group of ncclSend / ncclRecv per neighbour on the comm stream
// (pa_rccl.cpp; src/mpi_array.jl:575-614) -- and behind the receives the comm stream raises a flag word.  The product is ONE launch
// on the compute stream, queued at once: its own x own blocks overlap the transport, its tail acquires the flag, sums the boundary
// rows from the receive buffer and unpacks b's ghost entries.  Launches per step: pack + RCCL's own + flag | 1.
__global__ void kf_raise(unsigned long long *flag, unsigned long long seq) { flag_store(flag, seq); }

void pa_fused_plan_release(pa_plan *p) {
  if (p->h_rstatus) {
    auto &w = p->ctx->fused_status;
    w.erase(std::remove(w.begin(), w.end(), p->h_rstatus), w.end());
  }
  if (p->d_rflag) (void)hipFree(p->d_rflag);
  if (p->h_rstatus) (void)hipHostFree(p->h_rstatus);
  p->d_rflag = nullptr; p->h_rstatus = nullptr;
}

int pa_mul_fused_rccl(pa_matrix *m, pa_comm *comm, pa_vec *c, pa_vec *b, double alpha, double beta) {
  pa_plan *p = m->plan;
  pa_ctx *cx = p->ctx;
  PA_HIP(hipSetDevice(cx->device));
  if (!p->d_rflag) {
    // [0] the flag, [1] (as Int32) the index 0 of the tail's one-entry wait list, [2..] what the launch's tail does (pa_fused_comm)
    PA_HIP(hipMalloc((void **)&p->d_rflag, 16 + sizeof(pa_fused_comm)));
    PA_HIP(hipMemset(p->d_rflag, 0, 16 + sizeof(pa_fused_comm)));
    PA_HIP(hipHostMalloc((void **)&p->h_rstatus, sizeof(int), hipHostMallocMapped));
    *p->h_rstatus = 0;
    pa_fused_comm X;
    X.flags = p->d_rflag; X.wait_idx = (const int32_t *)(p->d_rflag + 1); X.n_wait = 1;
    double secs = 30.0;
    if (const char *e = getenv("PA_IPC_TIMEOUT_S")) secs = std::max(0.001, atof(e));
    X.ticks = (long long)(secs * 1e8);
    X.status = p->h_rstatus;
    X.u_idx = p->snd.d_idx; X.u_n = (int)p->snd.n;
    PA_HIP(pa_h2d(p->d_rflag + 2, &X, sizeof X));
    PA_HIP(hipDeviceSynchronize());
    cx->fused_status.push_back(p->h_rstatus);
  }
  if (*(volatile int *)p->h_rstatus != 0) {
    // An earlier product's tail gave up waiting for the flag behind the RCCL receives (VERDICT r05 "Next" #1b).  That product is lost
    // (pa_ctx_sync says so, once); the HANDLE is not: drain both streams -- the receives it waited for have either completed by now
    // or hipStreamSynchronize reports what RCCL died of --, and this and every later product of the handle run as separate launches
    // (stream order + events instead of an in-launch wait: round 4's chain, PA_MUL_FUSED=0's path).
    PA_HIP(hipStreamSynchronize(cx->s[1]));
    PA_HIP(hipStreamSynchronize(cx->s[0]));
    if (*(volatile int *)p->h_rstatus == 1) cx->n_fused_timeouts++;          // (not yet seen by a pa_ctx_sync: it will report it)
    *(volatile int *)p->h_rstatus = 0;
    m->fused_off = true;
    fprintf(stderr, "[pa] part %d: a fused product gave up waiting for its RCCL receives (PA_IPC_TIMEOUT_S); this handle continues "
                    "with separate launches\n", p->part);
    return PA_FUSED_GAVE_UP;
  }
  PA_TRY(pa_exchange_pack(p, b, PA_CONSISTENT));
  PA_TRY(pa_exchange_rccl(p, comm, PA_CONSISTENT));
  const unsigned long long seq = ++p->rseq;
  if (!(cx->sw.test_skip_raise && seq == (unsigned long long)cx->sw.test_skip_raise))
    hipLaunchKernelGGL(kf_raise, dim3(1), dim3(1), 0, cx->s[1], p->d_rflag, seq);
  PA_HIP(hipGetLastError());
  PA_TRY(pa_mul_fused_launch(m, c, b, alpha, beta, cx->s[0], (const pa_fused_comm *)(p->d_rflag + 2), seq, 0, cx->sw.fused_tail_blocks));
  // the exchange is complete with the launch (wait(t) and the unpack are its tail); the next pack orders itself behind the compute
  // stream (pa_exchange_pack records ev_compute), so nothing overwrites the buffers this launch still reads
  p->phase = 0; p->own_comm_stream = false; p->ev_wait = nullptr;
  return PA_OK;
}
