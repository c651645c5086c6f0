// pa_csr.hip -- CSR blocks: construction, column encodings, value dictionary, the product launch (pa_spmv), introspection; scatter-add maps
// (one of the units pa_device.hip was split into in round 5; compiled with -ffp-contract=off like all of them)
#include <hip/hip_runtime.h>

#include <atomic>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cmath>
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
// CSR blocks
// ------------------------------------------------------------------------------------------------
static inline int64_t read_index(const void *a, int bytes, int64_t i) {
  return bytes == 4 ? (int64_t)((const int32_t *)a)[i] : ((const int64_t *)a)[i];
}

static void csr_free_chain(pa_csr *A);

// Where a block's stored entries come from: host arrays (an upload) or device arrays (a block assembled on the device,
// pa_setup.hip); 0-based columns either way.
struct csr_src {
  const int32_t *col0 = nullptr;
  const double *nzval = nullptr;
  const int32_t *d_col = nullptr;
  const double *d_val = nullptr;
  // rows counted (and, when most are empty, compacted) on the device already (pa_rowsel.hip): the row pointer handed to
  // csr_fill_slab is the final one (n_nonempty + 1 entries with d_row_ids, n_rows + 1 without) and the host passes are skipped
  int64_t pre_nonempty = -1;
  bool pre_compact = false;
  const int32_t *d_pre_row_ids = nullptr;
  bool on_device() const { return d_col != nullptr || d_val != nullptr; }
  csr_src at(int64_t off) const {
    csr_src o;
    o.pre_nonempty = pre_nonempty; o.pre_compact = pre_compact; o.d_pre_row_ids = d_pre_row_ids;
    o.col0 = col0 ? col0 + off : nullptr; o.nzval = nzval ? nzval + off : nullptr;
    o.d_col = d_col ? d_col + off : nullptr; o.d_val = d_val ? d_val + off : nullptr;
    return o;
  }
};

// ---- the lossless value dictionary (round 4: built on the device, on by default for big blocks) --------------------------
// A block whose stored values take at most PA_VDICT_MAX = 64 distinct bit patterns (27-point HPCG: 2; 7-point Laplacian: 2; a Q1
// stiffness matrix on a uniform grid: about a dozen) also keeps ONE BYTE per stored entry, and the product kernels stream that
// instead of the 8-byte value (k_spmv_rowsplit<..., VD = true>: the values sit in the lanes of a register, an entry's value is
// fetched with ds_bpermute).  Same values, same products, same order: same bits; 0.567 against 0.673 ms on the 256^3 operator.
//   PA_SPMV_VALUE_DICT unset: AUTO -- blocks of >= 2^18 stored entries that do not run on the x-window launches;
//                      = 1  : every block that qualifies;  = 0: never (bench.py's headline: `value` stays on the fp64 stream).
// Two passes over the values: (1) every distinct bit pattern is inserted into a 256-slot open-addressing table with atomicCAS
// (a lane first compares with the last two patterns it saw: a stencil operator costs two compares per entry), more than 64 -> no
// dictionary; (2) the sorted patterns become the dictionary and every entry its code.  pa_csr_update_values* leave the codes
// stale: the block continues on the fp64 stream and is re-encoded once it has served 8 products on the new values (a caller that
// re-assembles every step never pays for codes it will not use); new values that overflow the dictionary end it for good.
#define PA_VDICT_SLOTS 256
#define PA_VDICT_EMPTY 0x7FF8DEADBEEF0001ull   /* (a NaN payload no assembled matrix holds; a block that does gets no dictionary) */
__device__ __forceinline__ int vdict_hash(unsigned long long b) { return (int)((b * 0x9E3779B97F4A7C15ull) >> 56); }

__global__ __launch_bounds__(256) void k_vdict_collect(const double *__restrict__ val, int64_t n, unsigned long long *__restrict__ table,
                                                       int *__restrict__ count) {
  unsigned long long seen0 = PA_VDICT_EMPTY, seen1 = PA_VDICT_EMPTY;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(val[p]);
    if (b == seen0 || b == seen1) continue;
    if (b == PA_VDICT_EMPTY || *(volatile int *)count > PA_VDICT_MAX) { atomicMax(count, PA_VDICT_MAX + 1); return; }
    int h = vdict_hash(b);
    for (int k = 0; k < PA_VDICT_SLOTS; ++k) {
      // (a plain look first: after the first few hundred lanes every pattern of a stencil operator is in the table, and four
      //  million lanes doing an atomic on the same two words cost ~10 ms where the loads cost nothing)
      unsigned long long old = *(volatile const unsigned long long *)&table[h];
      if (old == PA_VDICT_EMPTY) {
        old = atomicCAS(&table[h], PA_VDICT_EMPTY, b);
        if (old == PA_VDICT_EMPTY) { atomicAdd(count, 1); break; }
      }
      if (old == b) break;
      h = (h + 1) & (PA_VDICT_SLOTS - 1);
    }
    seen1 = seen0; seen0 = b;
  }
}

__global__ __launch_bounds__(256) void k_vdict_encode(const double *__restrict__ val, int64_t n, const unsigned long long *__restrict__ table,
                                                      const unsigned char *__restrict__ slot_code, unsigned char *__restrict__ code,
                                                      int *__restrict__ missing) {
  __shared__ unsigned long long t[PA_VDICT_SLOTS];
  __shared__ unsigned char sc[PA_VDICT_SLOTS];
  t[threadIdx.x] = table[threadIdx.x];
  sc[threadIdx.x] = slot_code[threadIdx.x];
  __syncthreads();
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < n; p += stride) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(val[p]);
    int h = vdict_hash(b);
    // (a table built from THESE values holds every one of them; a table inherited from the block this one was cut from holds them
    //  unless the caller changed values in between: a value that is not there raises `missing` and the caller builds afresh)
    for (int k = 0; k < PA_VDICT_SLOTS && t[h] != b; ++k) h = (h + 1) & (PA_VDICT_SLOTS - 1);
    if (t[h] != b) { *missing = 1; code[p] = 0; continue; }
    code[p] = sc[h];
  }
}

// A block cut from another block (a colour's rows, pa_rowsel.hip) takes that block's dictionary instead of finding its own: same
// values, so the same table serves -- one pass over the values (their codes) instead of two plus a read-back (round 5: 50 dictionary
// builds were 0.18 s of the HPCG driver's 0.72 s optimised set-up, which the rating charges per set).
thread_local const pa_csr *pa_tls_vdict_parent = nullptr;

// (re)build the dictionary of one slab from its value stream; `rebuild`: the codes exist and the values changed
static int vdict_build(pa_ctx *c, pa_csr *A, bool rebuild) {
  const char *e = getenv("PA_SPMV_VALUE_DICT");
  const int mode = e ? atoi(e) : -1;                           // -1 auto, 0 off, 1 on
  A->use_vdict = false;
  A->vdict_stale = false;
  if (mode == 0 || A->nnz == 0 || A->vdict_dead || c->capturing || pa_tls_plain_encoding) return PA_OK;
  if (mode < 0 && !rebuild && (A->nnz < ((int64_t)1 << 18) || A->n_xw_groups > 0)) return PA_OK;
  const size_t pad = 8;
  hipStream_t s = c->s[0];
  auto alloc_codes = [&]() -> int {
    if (A->d_code) return PA_OK;
    PA_TRY(pa_dev_alloc(c, (void **)&A->d_code, A->nnz + pad, PA_MEM_MATRIX));
    // the dictionary (PA_VDICT_MAX values) and, behind it, what built it: the hash table's slots and their codes
    PA_TRY(pa_dev_alloc(c, (void **)&A->d_dict, sizeof(double) * PA_VDICT_MAX + sizeof(unsigned long long) * PA_VDICT_SLOTS + PA_VDICT_SLOTS + 64, PA_MEM_MATRIX));
    PA_HIP(hipMemsetAsync(A->d_code + A->nnz, 0, pad, s));
    return PA_OK;
  };
  auto table_of = [](const pa_csr *B) { return (unsigned long long *)(B->d_dict + PA_VDICT_MAX); };
  auto slots_of = [&](const pa_csr *B) { return (unsigned char *)(table_of(B) + PA_VDICT_SLOTS); };
  auto flag_of = [&](const pa_csr *B) { return (int *)(slots_of(B) + PA_VDICT_SLOTS); };
  const int enc_blocks = (int)std::min<int64_t>((A->nnz + 256 * 8 - 1) / (256 * 8), 256 * 64);
  if (const pa_csr *P = pa_tls_vdict_parent) {
    if (!rebuild && P != A && P->ctx == c && P->use_vdict && P->d_dict && mode != 0) {
      if (alloc_codes() == PA_OK &&
          hipMemcpyAsync(A->d_dict, P->d_dict, sizeof(double) * PA_VDICT_MAX + sizeof(unsigned long long) * PA_VDICT_SLOTS + PA_VDICT_SLOTS,
                         hipMemcpyDeviceToDevice, s) == hipSuccess &&
          hipMemsetAsync(flag_of(A), 0, sizeof(int), s) == hipSuccess) {
        hipLaunchKernelGGL(k_vdict_encode, dim3(enc_blocks), dim3(256), 0, s, A->d_val, A->nnz, table_of(A), slots_of(A), A->d_code, flag_of(A));
        int missing = 1;
        if (hipMemcpyAsync(&missing, flag_of(A), sizeof(int), hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess &&
            !missing) {
          A->use_vdict = true;
          A->n_dict = P->n_dict;
          A->dict_finite = P->dict_finite;
          return pa_pell_bits_refresh(A);
        }
      }
      (void)hipGetLastError();                               // (anything amiss: the block finds its own dictionary below)
    }
  }
  // (scratch of the context, made once: a multigrid set-up builds dozens of blocks, and three hipMallocs per block showed)
  if (!c->d_vdict_scratch) PA_HIP(pa_raw_malloc(&c->d_vdict_scratch, sizeof(unsigned long long) * PA_VDICT_SLOTS + PA_VDICT_SLOTS + 64));
  unsigned long long *d_table = (unsigned long long *)c->d_vdict_scratch;
  unsigned char *d_slot = (unsigned char *)(d_table + PA_VDICT_SLOTS);
  int *d_count = (int *)(d_slot + PA_VDICT_SLOTS);
  const auto t_begin = std::chrono::steady_clock::now();
  auto done = [&](int st) {
    if (getenv("PA_SETUP_TIMING"))
      fprintf(stderr, "[pa setup] value dictionary of %lld entries: %s, %.3f ms\n", (long long)A->nnz, A->use_vdict ? "built" : "none",
              std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count());
    return st;
  };
  std::vector<unsigned long long> table(PA_VDICT_SLOTS, PA_VDICT_EMPTY);
  if (hipMemcpyAsync(d_table, table.data(), sizeof(unsigned long long) * PA_VDICT_SLOTS, hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemsetAsync(d_count, 0, sizeof(int), s) != hipSuccess) { pa_set_err("value dictionary: upload failed"); return done(PA_ERR_HIP); }
  const int blocks = (int)std::min<int64_t>((A->nnz + 256 * 8 - 1) / (256 * 8), 256 * 64);
  hipLaunchKernelGGL(k_vdict_collect, dim3(blocks), dim3(256), 0, s, A->d_val, A->nnz, d_table, d_count);
  int count = 0;
  if (hipMemcpyAsync(&count, d_count, sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipMemcpyAsync(table.data(), d_table, sizeof(unsigned long long) * PA_VDICT_SLOTS, hipMemcpyDeviceToHost, s) != hipSuccess ||
      hipStreamSynchronize(s) != hipSuccess) { pa_set_err("value dictionary: read-back failed"); return done(PA_ERR_HIP); }
  if (count > PA_VDICT_MAX) {                                  // more distinct values than lanes: this block streams fp64 for good
    if (rebuild) A->vdict_dead = true;
    return done(PA_OK);
  }
  std::vector<unsigned long long> dict;
  for (unsigned long long b : table) if (b != PA_VDICT_EMPTY) dict.push_back(b);
  std::sort(dict.begin(), dict.end());
  std::vector<unsigned char> slot(PA_VDICT_SLOTS, 0);
  for (int h = 0; h < PA_VDICT_SLOTS; ++h)
    if (table[h] != PA_VDICT_EMPTY) slot[h] = (unsigned char)(std::lower_bound(dict.begin(), dict.end(), table[h]) - dict.begin());
  std::vector<double> dv(PA_VDICT_MAX, 0.0);
  memcpy(dv.data(), dict.data(), 8 * dict.size());
  if (const int st = alloc_codes()) return done(st);
  if (hipMemcpyAsync(d_slot, slot.data(), PA_VDICT_SLOTS, hipMemcpyHostToDevice, s) != hipSuccess ||
      hipMemcpyAsync(A->d_dict, dv.data(), sizeof(double) * PA_VDICT_MAX, hipMemcpyHostToDevice, s) != hipSuccess ||
      // (the table and its codes stay with the block: a block cut from this one inherits them, pa_tls_vdict_parent)
      hipMemcpyAsync(table_of(A), d_table, sizeof(unsigned long long) * PA_VDICT_SLOTS, hipMemcpyDeviceToDevice, s) != hipSuccess ||
      hipMemcpyAsync(slots_of(A), d_slot, PA_VDICT_SLOTS, hipMemcpyDeviceToDevice, s) != hipSuccess ||
      hipMemsetAsync(flag_of(A), 0, sizeof(int), s) != hipSuccess) {
    pa_set_err("value dictionary: upload failed");
    return done(PA_ERR_HIP);
  }
  hipLaunchKernelGGL(k_vdict_encode, dim3(blocks), dim3(256), 0, s, A->d_val, A->nnz, d_table, d_slot, A->d_code, flag_of(A));
  if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) { pa_set_err("value dictionary: encoding failed"); return done(PA_ERR_HIP); }
  A->use_vdict = true;
  A->n_dict = (int)dict.size();
  A->dict_finite = true;
  for (size_t k = 0; k < dict.size(); ++k) A->dict_finite = A->dict_finite && std::isfinite(dv[k]);
  return done(pa_pell_bits_refresh(A));       // (a block with pattern-ELL storage and <= 2 values: its one-bit stream again)
}

// a block whose values were updated runs on the fp64 stream; once it has served 8 products on the new values its codes are renewed
static void vdict_maintain(const pa_csr *A) {
  for (const pa_csr *S0 = A; S0; S0 = S0->next) {
    pa_csr *S = const_cast<pa_csr *>(S0);
    if (!S->vdict_stale || S->ctx->capturing) continue;
    if (++S->vdict_products < 8) continue;
    if (vdict_build(S->ctx, S, true) != PA_OK) { (void)hipGetLastError(); S->vdict_dead = true; S->vdict_stale = false; }
  }
}

// Behind a value update (the new values are queued on the compute stream).  A slab whose one-byte-stream product sits in a recorded
// hipGraph gets its codes and dictionary renewed NOW, in place: the replay reads them, and nothing eager may run in between to
// renew them lazily (the replay multiplied with the codes of the old values).  When the new values no longer fit a
// dictionary the recorded graph cannot be served any more: an error, not a silent wrong product.
static int vdict_after_update(pa_csr *A) {
  A->val_epoch++;
  for (pa_csr *S = A; S; S = S->next) PA_TRY(pa_pell_after_update(S));
  for (pa_csr *S = A; S; S = S->next) {
    if (!S->vd_captured) continue;
    PA_REQUIRE(!S->ctx->capturing, "values of a block whose product is already recorded must not be updated inside a capture");
    S->vdict_dead = false;
    PA_TRY(vdict_build(S->ctx, S, true));
    if (!S->use_vdict || (S->vd_captured_two && S->n_dict > 2)) {
      pa_set_err("the new values take more than %d distinct bit patterns, but a recorded hipGraph multiplies through this block's "
                 "value dictionary: record the graph again (pa_graph_begin / pa_graph_end)", S->use_vdict ? 2 : PA_VDICT_MAX);
      S->vd_captured = false; S->vd_captured_two = false;
      return PA_ERR_STATE;
    }
  }
  return pa_csr_values_changed(A);           // the handles that hold copies of these values (twin, boundary rows' block) follow now
}

void pa_csr_before_product(const pa_csr *A) { vdict_maintain(A); }

// fills the freshly created slab A; on any failure the caller (csr_build_slab) hands back whatever A holds by then
static int csr_fill_slab(pa_ctx *c, pa_csr *A, int64_t n_rows, int64_t n_cols, int64_t nnz, std::vector<int32_t> &rp,
                         const csr_src &src) {
  const int32_t *col0 = src.col0;          // 0-based, host (NULL when the entries are already on the device)
  const double *nzval = src.nzval;
  // non-empty rows; compact when most rows are empty (the own_ghost block: only boundary rows)
  const bool tm_ = getenv("PA_SETUP_TIMING") != nullptr;   // stderr: seconds per phase of this function
  auto t0_ = std::chrono::steady_clock::now();
  auto lap = [&](const char *what) {
    if (!tm_) return;
    auto t1 = std::chrono::steady_clock::now();
    fprintf(stderr, "[pa setup] %-10s %8.3f s  (nnz %lld)\n", what, std::chrono::duration<double>(t1 - t0_).count(), (long long)nnz);
    t0_ = t1;
  };
  std::vector<int32_t> row_ids;
  int64_t n_nonempty = 0;
  bool compact = false;
  std::vector<int32_t> crp;
  if (src.pre_nonempty >= 0) {
    n_nonempty = src.pre_nonempty;
    compact = src.pre_compact;
    crp.swap(rp);
  } else {
    std::vector<int64_t> part_cnt(33, 0);          // (host threads over row ranges: a colour block of the 256^3 operator has 16.8 M
    host_parallel(n_rows, n_rows * 4, [&](int t, int64_t lo, int64_t hi) {      // rows, one in eight non-empty)
      int64_t k = 0;
      for (int64_t r = lo; r < hi; ++r) k += rp[r + 1] > rp[r];
      part_cnt[t] = k;
    });
    for (int t = 0; t < 33; ++t) n_nonempty += part_cnt[t];
    compact = n_rows > 0 && n_nonempty * 2 < n_rows;
    if (compact) {
      row_ids.resize(n_nonempty);
      crp.resize(n_nonempty + 1);
      crp[0] = 0;
      std::vector<int64_t> first(34, 0);
      for (int t = 0; t < 33; ++t) first[t + 1] = first[t] + part_cnt[t];
      host_parallel(n_rows, n_rows * 4, [&](int t, int64_t lo, int64_t hi) {
        int64_t k = first[t];
        for (int64_t r = lo; r < hi; ++r)
          if (rp[r + 1] > rp[r]) {
            row_ids[k] = (int32_t)r;
            crp[k + 1] = rp[r + 1];
            ++k;
          }
      });
    } else {
      crp.swap(rp);
    }
  }
  const int64_t nc = (int64_t)crp.size() - 1;
  std::vector<int32_t> chunk_row;
  int64_t n_long = 0;
  A->n_rows = n_rows; A->n_cols = n_cols; A->nnz = nnz;
  A->n_crows = nc; A->n_nonempty = n_nonempty;
  A->compact = compact;
  PA_HIP(hipSetDevice(c->device));
  const size_t pad = 8;
  // Column streams: they decide what has to live in HBM at all (see pa_encode_columns).  PA_SPMV_PATTERN=0 /
  // PA_SPMV_COL16=0 disable the row-pattern descriptors / the 16-bit windowed stream.
  // Round 3: the encoding runs ON THE DEVICE (pa_setup.hip) over the raw CSR uploaded first -- row hashes, a radix sort,
  // per-chunk descriptors, window tags and codes as kernels; PA_SETUP_DEVICE=0 keeps the host encoder below, whose arrays
  // the device's equal byte for byte (tests/...test_device_side_encoding_equals_the_host_s).
  const char *ep = getenv("PA_SPMV_PATTERN"), *e16 = getenv("PA_SPMV_COL16"), *ec = getenv("PA_SPMV_COMPACT_STREAMS"), *ed = getenv("PA_SETUP_DEVICE");
  const bool want_pattern = !(ep && atoi(ep) == 0) && nnz > 0 && !pa_tls_plain_encoding;
  const bool want_c16 = !(e16 && atoi(e16) == 0) && nnz > 0 && !pa_tls_plain_encoding;
  const bool compact_streams = !(ec && atoi(ec) == 0), on_device = (!(ed && atoi(ed) == 0) || src.on_device()) && nnz > 0;
  // (the value stream first: it is the allocation that brings the context's arena into being, pa_arena.hip)
  PA_TRY(pa_dev_alloc(c, (void **)&A->d_val, sizeof(double) * (nnz + pad), PA_MEM_MATRIX));
  PA_TRY(pa_dev_alloc(c, (void **)&A->d_crp, sizeof(int32_t) * (nc + 1), PA_MEM_MATRIX));
  PA_HIP(hipMemsetAsync(A->d_val + nnz, 0, sizeof(double) * pad, c->s[0]));            // (the streams are non-blocking: a null-stream
  PA_HIP(hipStreamSynchronize(c->s[0]));                                               // memset would not be ordered with the kernels)
  PA_HIP(pa_h2d(A->d_crp, crp.data(), sizeof(int32_t) * (nc + 1)));
  // the row split: on the device from the row pointers just uploaded (pointer doubling, pa_setup.hip) or the host's greedy loop
  if (on_device) PA_TRY(pa_dev_row_split(c, A->d_crp, nc, PA_SPMV_CHUNK_NNZ, 4096, 8, chunk_row, &n_long));
  else pa_build_chunks(crp.data(), nc, PA_SPMV_CHUNK_NNZ, 4096, chunk_row, &n_long);
  if (pa_tls_piece_build && pa_tls_row_breaks && !compact) {
    // a column piece of a chain meant for one launch: its chunks also end at the rows all pieces share (a chunk cut in two at a row
    // boundary is two valid chunks)
    std::vector<int32_t> merged;
    merged.reserve(chunk_row.size() + pa_tls_row_breaks->size());
    std::set_union(chunk_row.begin(), chunk_row.end(), pa_tls_row_breaks->begin(), pa_tls_row_breaks->end(), std::back_inserter(merged));
    while (!merged.empty() && merged.back() > nc) merged.pop_back();
    chunk_row.swap(merged);
  }
  lap("chunks");
  A->n_chunks = (int64_t)chunk_row.size() - 1; A->n_long = n_long;
  PA_TRY(pa_dev_alloc(c, (void **)&A->d_chunk_row, sizeof(int32_t) * chunk_row.size(), PA_MEM_MATRIX));
  if (nnz && src.on_device()) {
    PA_HIP(hipMemcpyAsync(A->d_val, src.d_val, sizeof(double) * nnz, hipMemcpyDeviceToDevice, c->s[0]));
    PA_HIP(hipStreamSynchronize(c->s[0]));
  } else if (nnz) PA_HIP(pa_h2d(A->d_val, nzval, sizeof(double) * nnz));
  PA_HIP(pa_h2d(A->d_chunk_row, chunk_row.data(), sizeof(int32_t) * chunk_row.size()));
  if (compact) {
    PA_TRY(pa_dev_alloc(c, (void **)&A->d_row_ids, sizeof(int32_t) * std::max<int64_t>(1, nc), PA_MEM_MATRIX));
    if (nc && src.d_pre_row_ids) {
      PA_HIP(hipMemcpyAsync(A->d_row_ids, src.d_pre_row_ids, sizeof(int32_t) * nc, hipMemcpyDeviceToDevice, c->s[0]));
      PA_HIP(hipStreamSynchronize(c->s[0]));
    } else if (nc) PA_HIP(pa_h2d(A->d_row_ids, row_ids.data(), sizeof(int32_t) * nc));
  }
  lap("upload");
  pa_col_streams cs;                       // host encoder's arrays (PA_SETUP_DEVICE=0); `win` also when the x windows are planned
  if (on_device) {
    // the raw columns go up whole; a block with row patterns keeps only the compacted streams made from them
    int32_t *d_colfull = nullptr;
    PA_TRY(pa_dev_alloc(c, (void **)&d_colfull, sizeof(int32_t) * (nnz + pad), PA_MEM_MATRIX));
    A->d_col = d_colfull;                  // (owned by A from here on: a failure below frees it with the block)
    PA_HIP(hipMemsetAsync(d_colfull + nnz, 0, sizeof(int32_t) * pad, c->s[0]));
    PA_HIP(hipStreamSynchronize(c->s[0]));
    if (src.on_device()) {
      PA_HIP(hipMemcpyAsync(d_colfull, src.d_col, sizeof(int32_t) * nnz, hipMemcpyDeviceToDevice, c->s[0]));
      PA_HIP(hipStreamSynchronize(c->s[0]));
    } else PA_HIP(pa_h2d(d_colfull, col0, sizeof(int32_t) * nnz));
    lap("columns up");
    pa_dev_streams ds;
    PA_TRY(pa_dev_encode_columns(c, A->d_crp, d_colfull, A->d_row_ids, nc, nnz, A->d_chunk_row, A->n_chunks, PA_SPMV_CHUNK_NNZ,
                                 want_pattern, want_c16, compact_streams, ds));
    A->use_pattern = ds.use_pattern; A->use_c16 = ds.use_c16; A->pad_products = ds.pad_products;
    A->n_pattern_chunks = ds.n_pattern; A->n_c16_chunks = ds.n_c16; A->n_c32_chunks = ds.n_c32;
    A->n_c16_fallback = ds.use_c16 ? A->n_chunks - ds.n_pattern - ds.n_c16 : 0;
    A->nnz_c16 = ds.nnz_c16; A->nnz_c32 = ds.nnz_c32;
    A->d_pdesc = ds.d_pdesc; A->d_pdelta = ds.d_pdelta; A->n_pdelta = ds.n_pdelta;
    A->d_win = ds.d_win; A->d_col16 = ds.d_c16; A->n_col16 = ds.n_c16_slots;
    if (ds.full) A->n_col32 = nnz;
    else {
      A->d_col = ds.d_c32; A->n_col32 = ds.n_c32_slots;
      PA_HIP(hipStreamSynchronize(c->s[0]));
      if (c->keep_raw_columns) A->d_raw_col = d_colfull;
      else pa_dev_free(c, d_colfull);
    }
    cs.use_pattern = ds.use_pattern; cs.use_c16 = ds.use_c16; cs.full = ds.full;
    if (tm_) fprintf(stderr, "[pa setup] device encode %.3f ms: %lld pattern / %lld c16 / %lld c32 chunks\n", ds.ms, (long long)ds.n_pattern,
                     (long long)ds.n_c16, (long long)ds.n_c32);
    lap("encode");
  } else {
    pa_encode_columns(crp.data(), col0, compact ? row_ids.data() : nullptr, nc, chunk_row, PA_SPMV_CHUNK_NNZ, want_pattern, want_c16,
                      host_threads(nnz), cs, compact_streams);
    lap("encode");
    A->use_pattern = cs.use_pattern; A->use_c16 = cs.use_c16;
    if (!cs.use_pattern && nc > 0) {                         // (see PADP in pa_spmv_kernel.h)
      int64_t mult8 = 0, nonempty = 0;
      for (int64_t r = 0; r < nc; ++r) {
        const int32_t len = crp[r + 1] - crp[r];
        nonempty += len > 0;
        mult8 += len > 0 && (len & 7) == 0;
      }
      A->pad_products = mult8 * 2 > nonempty;
    }
    A->n_pattern_chunks = cs.n_pattern; A->n_c16_chunks = cs.n_c16; A->n_c32_chunks = cs.n_c32;
    A->n_c16_fallback = cs.use_c16 ? A->n_chunks - cs.n_pattern - cs.n_c16 : 0;
    A->n_col32 = cs.full ? nnz : (int64_t)cs.c32.size() - (int64_t)pad;
    for (int64_t ch = 0; ch < A->n_chunks; ++ch) {           // stored entries by the column encoding their chunk reads
      const int64_t ne = (int64_t)crp[chunk_row[ch + 1]] - crp[chunk_row[ch]];
      if (cs.use_pattern && cs.pdesc[(size_t)ch * PA_PDESC_INTS] > 0) continue;
      if (cs.use_c16 && cs.win[(size_t)ch * PA_C16_WINDOWS] >= 0 && ne + (crp[chunk_row[ch]] & 1) <= PA_SPMV_CHUNK_NNZ) A->nnz_c16 += ne;
      else A->nnz_c32 += ne;
    }
    PA_TRY(pa_dev_alloc(c, (void **)&A->d_col, sizeof(int32_t) * (A->n_col32 + pad), PA_MEM_MATRIX));
    PA_HIP(hipMemsetAsync(A->d_col + A->n_col32, 0, sizeof(int32_t) * pad, c->s[0]));
    PA_HIP(hipStreamSynchronize(c->s[0]));
    if (nnz) {
      if (cs.full) PA_HIP(pa_h2d(A->d_col, col0, sizeof(int32_t) * nnz));
      else if (A->n_col32) PA_HIP(pa_h2d(A->d_col, cs.c32.data(), sizeof(int32_t) * A->n_col32));
    }
    if (cs.use_c16) {
      A->n_col16 = (int64_t)cs.c16.size();
      PA_TRY(pa_dev_alloc(c, (void **)&A->d_col16, sizeof(uint16_t) * cs.c16.size(), PA_MEM_MATRIX));
      PA_TRY(pa_dev_alloc(c, (void **)&A->d_win, sizeof(int32_t) * std::max<size_t>(1, cs.win.size()), PA_MEM_MATRIX));
      PA_HIP(pa_h2d(A->d_col16, cs.c16.data(), sizeof(uint16_t) * cs.c16.size()));
      if (!cs.win.empty()) PA_HIP(pa_h2d(A->d_win, cs.win.data(), sizeof(int32_t) * cs.win.size()));
    }
    if (cs.use_pattern) {
      PA_TRY(pa_dev_alloc(c, (void **)&A->d_pdesc, sizeof(int32_t) * cs.pdesc.size(), PA_MEM_MATRIX));
      A->n_pdelta = (int64_t)cs.pdelta.size();
      PA_TRY(pa_dev_alloc(c, (void **)&A->d_pdelta, sizeof(int32_t) * cs.pdelta.size(), PA_MEM_MATRIX));
      PA_HIP(pa_h2d(A->d_pdesc, cs.pdesc.data(), sizeof(int32_t) * cs.pdesc.size()));
      PA_HIP(pa_h2d(A->d_pdelta, cs.pdelta.data(), sizeof(int32_t) * cs.pdelta.size()));
    }
    lap("streams up");
  }
  // Rows without a pattern whose columns stay within a band: groups of chunks read x from an LDS copy of their span
  // (pa_spmv_xwin.h).  Taken when most of the block's chunks fall into groups and the staged x is a fraction of the matrix
  // bytes the groups stream; PA_SPMV_XWIN=0 keeps every chunk on k_spmv_rowsplit.
  {
    const char *ex = getenv("PA_SPMV_XWIN");
    if (cs.use_c16 && !cs.use_pattern && !compact && !(ex && atoi(ex) == 0) && A->n_chunks >= 64) {
      const bool forced = ex && atoi(ex) == 2;
      const char *er = getenv("PA_SPMV_XRING");              // 0: windows only, 1 (default): the window tiers, then the ring, 2: ring only
      const int ring = pa_tls_piece_build ? 2 : er ? atoi(er) : 1;     // (a column piece is cut for the ring)
      std::vector<int32_t> cmax_host;
      pa_xw_plan P;
      if (on_device) {
        // the windows and the raw columns are on the device: the per-entry part of the planning (first / last column and
        // distinct lines of x per chunk) is a kernel, the greedy grouping over the chunks stays here
        pa_xw_chunk_stats S;
        S.cmin.resize(A->n_chunks); S.cmax.resize(A->n_chunks); S.lines.resize(A->n_chunks);
        PA_TRY(pa_dev_xw_chunk_stats(c, A->d_crp, A->d_col, A->d_chunk_row, A->d_win, A->n_chunks, PA_XR_CAP, S.cmin.data(),
                                     S.cmax.data(), S.lines.data()));
        pa_plan_xw_from_stats(crp.data(), chunk_row, S, forced, P, ring, pa_tls_piece_build ? pa_tls_row_breaks : nullptr);
        for (int64_t k = 0; k < A->n_chunks; ++k)
          if (S.cmax[k] >= 0) A->xw_max_span = std::max<int64_t>(A->xw_max_span, (int64_t)S.cmax[k] - S.cmin[k] + 1);
        cmax_host.swap(S.cmax);
      } else {
        pa_plan_xw(crp.data(), col0, chunk_row, cs.win.data(), forced, P, host_threads(nnz), ring, &cmax_host,
                   pa_tls_piece_build ? pa_tls_row_breaks : nullptr);
      }
      const std::vector<pa_xw_group> &groups = P.groups;
      const std::vector<int32_t> &rest = P.rest;
      const int64_t grouped = P.grouped, staged = P.staged;
      if (!groups.empty() && (forced || grouped * 2 >= nnz)) {
        std::vector<int32_t> chunk_p(chunk_row.size());
        for (size_t k = 0; k < chunk_row.size(); ++k) chunk_p[k] = crp[chunk_row[k]];
        A->n_xw_groups = (int64_t)groups.size(); A->n_xw_rest = (int64_t)rest.size();
        for (int t = 0; t < PA_XW_TIERS; ++t) A->n_xw_tier[t] = P.n_tier[t];
        A->n_xw_ring = P.n_ring;
        if (P.n_ring > 0) {
          PA_TRY(pa_dev_alloc(c, (void **)&A->d_chunk_cmax, sizeof(int32_t) * cmax_host.size(), PA_MEM_MATRIX));
          PA_HIP(pa_h2d(A->d_chunk_cmax, cmax_host.data(), sizeof(int32_t) * cmax_host.size()));
        }
        A->n_xw_chunks = A->n_chunks - A->n_xw_rest; A->xw_staged = staged;
        PA_TRY(pa_dev_alloc(c, (void **)&A->d_chunk_p, sizeof(int32_t) * chunk_p.size(), PA_MEM_MATRIX));
        PA_TRY(pa_dev_alloc(c, (void **)&A->d_xw_grp, sizeof(pa_xw_group) * groups.size(), PA_MEM_MATRIX));
        PA_HIP(pa_h2d(A->d_chunk_p, chunk_p.data(), sizeof(int32_t) * chunk_p.size()));
        PA_HIP(pa_h2d(A->d_xw_grp, groups.data(), sizeof(pa_xw_group) * groups.size()));
        if (!rest.empty()) {
          PA_TRY(pa_dev_alloc(c, (void **)&A->d_xw_rest, sizeof(int32_t) * rest.size(), PA_MEM_MATRIX));
          PA_HIP(pa_h2d(A->d_xw_rest, rest.data(), sizeof(int32_t) * rest.size()));
        }
      }
      lap("x windows");
      if (tm_) fprintf(stderr, "[pa setup] x windows: %lld + %lld + %lld groups (40 / 96 / 128 KiB) + %lld ring groups, %lld of %lld entries, %lld staged x entries, %s\n",
                       (long long)P.n_tier[0], (long long)P.n_tier[1], (long long)P.n_tier[2], (long long)P.n_ring, (long long)grouped, (long long)nnz,
                       (long long)staged, A->n_xw_groups ? "used" : "not used");
    }
  }
  lap("x windows");
  if (tm_) fprintf(stderr, "[pa setup] val %p (%lld B, memory class %d) col %p crp %p chunk_row %p pdesc %p\n", (void *)A->d_val,
                   (long long)(8 * (nnz + pad)), pa_mem_class(c, A->d_val), (void *)A->d_col, (void *)A->d_crp, (void *)A->d_chunk_row, (void *)A->d_pdesc);
  // what k_spmv_rowsplit reads first of a chunk, in one piece: {first row, its row pointer} pairs
  {
    std::vector<int32_t> rp2(2 * chunk_row.size());
    for (size_t k = 0; k < chunk_row.size(); ++k) { rp2[2 * k] = chunk_row[k]; rp2[2 * k + 1] = crp[chunk_row[k]]; }
    PA_TRY(pa_dev_alloc(c, (void **)&A->d_chunk_rp, sizeof(int32_t) * rp2.size(), PA_MEM_MATRIX));
    PA_HIP(pa_h2d(A->d_chunk_rp, rp2.data(), sizeof(int32_t) * rp2.size()));
  }
  // lossless value dictionary (see vdict_build): built by kernels from the value stream that is in HBM by now
  PA_TRY(vdict_build(c, A, false));
  // pattern blocks: second storage for the lane-per-row kernel (pa_pell.h); never an error
  (void)pa_pell_build(A);
  return PA_OK;
}

static int csr_build_slab(pa_ctx *c, int64_t n_rows, int64_t n_cols, int64_t nnz, std::vector<int32_t> &rp,
                          const csr_src &src, pa_csr **out) {
  pa_csr *A = new pa_csr();
  A->ctx = c;
  const int st = csr_fill_slab(c, A, n_rows, n_cols, nnz, rp, src);
  if (st != PA_OK) {                 // a failed allocation or upload half-way: nothing stays behind (device buffers, arena blocks)
    (void)hipGetLastError();
    csr_free_chain(A);
    return st;
  }
  *out = A;
  return PA_OK;
}

// Stored entries per slab: Int32 offsets (plus the padding) must stay below 2^31.  PA_CSR_MAX_SLAB_NNZ lowers the
// limit (tests force several slabs on small matrices).
static int64_t slab_limit() {
  const char *e = getenv("PA_CSR_MAX_SLAB_NNZ");
  const int64_t hard = ((int64_t)1 << 31) - ((int64_t)1 << 16);
  if (e && atoll(e) > 0 && atoll(e) < hard) return atoll(e);
  return hard;
}

// rp: 0-based Int64 row pointers of the whole block.  One slab when the block fits Int32 offsets, else consecutive
// row slabs (greedy, whole rows).
static int csr_build(pa_ctx *c, int64_t n_rows, int64_t n_cols, int64_t nnz, const std::vector<int64_t> &rp,
                     const csr_src &src, pa_csr **out) {
  const int64_t limit = slab_limit();
  pa_csr *head = nullptr, *tail = nullptr;
  int64_t r0 = 0;
  do {
    int64_t r1 = r0;
    if (nnz - rp[r0] <= limit) r1 = n_rows;
    else {
      // largest r1 with rp[r1] - rp[r0] <= limit
      r1 = std::upper_bound(rp.begin() + r0, rp.end(), rp[r0] + limit) - rp.begin() - 1;
      if (r1 <= r0) {
        csr_free_chain(head);
        pa_set_err("row %lld alone has more stored entries than a slab holds (%lld)", (long long)r0, (long long)limit);
        return PA_ERR_ARG;
      }
    }
    std::vector<int32_t> rp32(r1 - r0 + 1);
    host_parallel(r1 - r0 + 1, (r1 - r0 + 1) * 4, [&](int, int64_t lo, int64_t hi) {
      for (int64_t k = lo; k < hi; ++k) rp32[k] = (int32_t)(rp[r0 + k] - rp[r0]);
    });
    pa_csr *S = nullptr;
    const int64_t snnz = rp[r1] - rp[r0];
    const int st = csr_build_slab(c, r1 - r0, n_cols, snnz, rp32, src.at(rp[r0]), &S);
    if (st != PA_OK) { csr_free_chain(head); return st; }
    S->row0 = r0; S->nnz0 = rp[r0];
    if (tail) tail->next = S; else head = S;
    tail = S;
    r0 = r1;
  } while (r0 < n_rows);
  head->t_rows = n_rows;
  head->t_nnz = nnz;
  *out = head;
  // A block of 2^31 entries or more (a chain of row slabs, each with its own pattern-ELL storage by now): the fp64 stream of pattern-ELL
  // is a SECOND copy of tens of GB of values.  It stays when the device still has a quarter of its memory free behind it (round 6:
  // 27-pt 320^3 runs 1627 GFLOP/s on it against 1205-1250 on the row split -- big parts are where it pays most); else the slabs
  // keep the row-split kernel.  PA_SPMV_PELL_BIG=1 / 0: always / never.
  if (head->next && nnz >= ((int64_t)1 << 31) - ((int64_t)1 << 16)) {
    size_t fr = 0, tot = 0;
    bool keep = hipMemGetInfo(&fr, &tot) == hipSuccess && fr > tot / 4;
    if (const char *e = getenv("PA_SPMV_PELL_BIG")) keep = atoi(e) != 0;
    if (!keep) for (pa_csr *S = head; S; S = S->next) pa_pell_free(S);
  }
  // a block of unstructured rows whose band is wider than the sliding x window holds: split by columns into pieces the window does
  // hold (pa_transpose.hip; the pieces are built through this function again, hence the guard)
  if (!pa_tls_piece_build && !head->next) {
    pa_csr *split = nullptr;
    const int st = pa_csr_colsplit_if_wide(head, &split);
    if (st != PA_OK) (void)hipGetLastError();          // (the unsplit block serves)
    else if (split) { csr_free_chain(head); *out = split; }
  }
  return PA_OK;
}

extern "C" int pa_csr_create(pa_ctx *c, int64_t n_rows, int64_t n_cols, int64_t nnz, const void *rowptr,
                             const void *colval, int index_bytes, int index_base, const double *nzval, pa_csr **out) {
  return pa_csr_create_mixed(c, n_rows, n_cols, nnz, rowptr, index_bytes, colval, index_bytes, index_base, nzval, out);
}

extern "C" int pa_csr_create_mixed(pa_ctx *c, int64_t n_rows, int64_t n_cols, int64_t nnz, const void *rowptr,
                                   int rowptr_bytes, const void *colval, int colval_bytes, int index_base,
                                   const double *nzval, pa_csr **out) {
  PA_REQUIRE(c && out && rowptr, "bad arguments");
  PA_REQUIRE((rowptr_bytes == 4 || rowptr_bytes == 8) && (colval_bytes == 4 || colval_bytes == 8), "index bytes must be 4 or 8");
  PA_REQUIRE(index_base == 0 || index_base == 1, "index_base must be 0 or 1");
  PA_REQUIRE(n_rows >= 0 && n_cols >= 0 && nnz >= 0, "negative size");
  PA_REQUIRE(n_rows < (int64_t)2147483000 && n_cols < (int64_t)2147483000, "block too large for Int32 device indices");
  PA_REQUIRE(rowptr_bytes == 8 || nnz < (int64_t)2147483000, "2^31 stored entries or more need 64-bit row pointers");
  PA_REQUIRE(nnz == 0 || (colval && nzval), "colval/nzval are NULL");
  const auto t0_ = std::chrono::steady_clock::now();
  if (nnz == 0) {
    // a block without stored entries (the own|ghost block of a part without ghost columns: 16.8 M rows at 256^3): every row
    // pointer must equal the base; nothing else to look at, no Int64 copy of them
    std::vector<int64_t> bad_row(33, -1);
    host_parallel(n_rows + 1, (n_rows + 1) * 2, [&](int t, int64_t lo, int64_t hi) {
      for (int64_t r = lo; r < hi; ++r) if (read_index(rowptr, rowptr_bytes, r) != index_base) { bad_row[t] = r; return; }
    });
    for (int t = 0; t < 33; ++t) PA_REQUIRE(bad_row[t] < 0, "rowptr does not span [base, base+nnz] (row %lld)", (long long)bad_row[t]);
    std::vector<int32_t> crp(1, 0);
    csr_src src;
    src.pre_nonempty = 0; src.pre_compact = n_rows > 0;
    pa_csr *S = nullptr;
    PA_TRY(csr_build_slab(c, n_rows, n_cols, 0, crp, src, &S));
    S->t_rows = n_rows; S->t_nnz = 0;
    *out = S;
    return PA_OK;
  }
  std::vector<int64_t> rp(n_rows + 1);
  host_parallel(n_rows + 1, (n_rows + 1) * 4, [&](int, int64_t lo, int64_t hi) {
    for (int64_t r = lo; r < hi; ++r) rp[r] = read_index(rowptr, rowptr_bytes, r) - index_base;
  });
  PA_REQUIRE(rp[0] == 0 && rp[n_rows] == nnz, "rowptr does not span [base, base+nnz]");
  {
    std::vector<int64_t> bad_row(33, -1);
    host_parallel(n_rows, n_rows * 4, [&](int t, int64_t lo, int64_t hi) {
      for (int64_t r = lo; r < hi; ++r) if (rp[r + 1] < rp[r]) { bad_row[t] = r; return; }
    });
    for (int t = 0; t < 33; ++t) PA_REQUIRE(bad_row[t] < 0, "rowptr not monotone at row %lld", (long long)bad_row[t]);
  }
  std::unique_ptr<int32_t[]> colbuf;                   // (not a vector: no single-threaded zero fill of a multi-GB array)
  const int32_t *col0 = nullptr;
  if (colval_bytes == 4 && index_base == 0) {
    col0 = (const int32_t *)colval;                      // already what the device wants: no copy of a multi-GB array
    const int T = host_threads(nnz);
    std::vector<int64_t> bad(T, -1);
    auto chk = [&](int t) {
      for (int64_t p = nnz * t / T; p < nnz * (t + 1) / T; ++p)
        if (col0[p] < 0 || col0[p] >= n_cols) { bad[t] = p; return; }
    };
    {
      std::vector<std::thread> th;
      for (int t = 1; t < T; ++t) th.emplace_back(chk, t);
      chk(0);
      for (auto &x : th) x.join();
    }
    for (int t = 0; t < T; ++t) PA_REQUIRE(bad[t] < 0, "column index out of range at entry %lld", (long long)bad[t]);
  } else {
    colbuf.reset(new int32_t[std::max<int64_t>(1, nnz)]);
    const int T = host_threads(nnz);
    std::vector<int64_t> bad(T, -1);
    auto conv = [&](int t) {
      for (int64_t p = nnz * t / T; p < nnz * (t + 1) / T; ++p) {
        const int64_t j = read_index(colval, colval_bytes, p) - index_base;
        if (j < 0 || j >= n_cols) { if (bad[t] < 0) bad[t] = p; continue; }
        colbuf[p] = (int32_t)j;
      }
    };
    {
      std::vector<std::thread> th;
      for (int t = 1; t < T; ++t) th.emplace_back(conv, t);
      conv(0);
      for (auto &x : th) x.join();
    }
    for (int t = 0; t < T; ++t) PA_REQUIRE(bad[t] < 0, "column index out of range at entry %lld", (long long)bad[t]);
    col0 = colbuf.get();
  }
  if (getenv("PA_SETUP_TIMING"))
    fprintf(stderr, "[pa setup] %-10s %8.3f s  (nnz %lld)\n", "validate", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0_).count(), (long long)nnz);
  csr_src src;
  src.col0 = col0; src.nzval = nzval;
  return csr_build(c, n_rows, n_cols, nnz, rp, src, out);
}

// A block whose stored entries are already in HBM (0-based Int32 row pointers and columns, made by the device-side
// assembly of pa_setup.hip): the row split needs the row pointers on the host (one small download), everything per entry
// stays on the device.
int pa_csr_from_device(pa_ctx *c, int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t *d_rowptr, const int32_t *d_col,
                       const double *d_val, pa_csr **out) {
  std::vector<int32_t> rp32(n_rows + 1);
  PA_HIP(hipSetDevice(c->device));
  PA_HIP(hipMemcpy(rp32.data(), d_rowptr, sizeof(int32_t) * (n_rows + 1), hipMemcpyDeviceToHost));
  std::vector<int64_t> rp(rp32.begin(), rp32.end());
  PA_REQUIRE(rp[0] == 0 && rp[n_rows] == nnz, "device row pointers do not span the stored entries");
  csr_src src;
  src.d_col = d_col; src.d_val = d_val;
  return csr_build(c, n_rows, n_cols, nnz, rp, src, out);
}

// the same from rows the caller has counted and compacted on the device: crp = the final row pointer on the host (moved from)
int pa_csr_from_device_rows(pa_ctx *c, int64_t n_rows, int64_t n_cols, int64_t nnz, int64_t n_nonempty, std::vector<int32_t> &crp,
                            const int32_t *d_row_ids, const int32_t *d_col, const double *d_val, pa_csr **out) {
  PA_REQUIRE(nnz < slab_limit(), "a block of this size is a chain of slabs: the general constructor builds those");
  PA_REQUIRE((int64_t)crp.size() == (d_row_ids ? n_nonempty : n_rows) + 1 && crp.front() == 0 && crp.back() == nnz,
             "row pointers do not span the stored entries");
  csr_src src;
  src.d_col = d_col; src.d_val = d_val;
  src.pre_nonempty = n_nonempty; src.pre_compact = d_row_ids != nullptr; src.d_pre_row_ids = d_row_ids;
  pa_csr *S = nullptr;
  PA_TRY(csr_build_slab(c, n_rows, n_cols, nnz, crp, src, &S));
  S->t_rows = n_rows; S->t_nnz = nnz;
  *out = S;
  return PA_OK;
}

// a block without stored entries (the own|ghost block of a part without neighbours): what pa_csr_create_mixed builds from n_rows + 1
// equal row pointers, without the caller making, and the library reading, that array (16.8 M rows: 67 MB)
extern "C" int pa_csr_create_empty(pa_ctx *c, int64_t n_rows, int64_t n_cols, pa_csr **out) {
  PA_REQUIRE(c && out && n_rows >= 0 && n_cols >= 0, "bad arguments");
  PA_REQUIRE(n_rows < (int64_t)2147483000 && n_cols < (int64_t)2147483000, "block too large for Int32 device indices");
  csr_src src;
  src.pre_nonempty = 0;
  src.pre_compact = n_rows > 0;                          // (csr_fill_slab's rule: fewer than half of the rows hold entries)
  std::vector<int32_t> crp((size_t)(src.pre_compact ? 0 : n_rows) + 1, 0);
  pa_csr *S = nullptr;
  PA_TRY(csr_build_slab(c, n_rows, n_cols, 0, crp, src, &S));
  S->t_rows = n_rows; S->t_nnz = 0;
  *out = S;
  return PA_OK;
}

extern "C" int pa_csr_create_from_csc(pa_ctx *c, int64_t n_rows, int64_t n_cols, int64_t nnz, const void *colptr,
                                      const void *rowval, int index_bytes, int index_base, const double *nzval,
                                      pa_csr **out) {
  PA_REQUIRE(c && out && colptr, "bad arguments");
  PA_REQUIRE(index_bytes == 4 || index_bytes == 8, "index_bytes must be 4 or 8");
  PA_REQUIRE(index_base == 0 || index_base == 1, "index_base must be 0 or 1");
  PA_REQUIRE(nnz == 0 || (rowval && nzval), "rowval/nzval are NULL");
  PA_REQUIRE(n_rows < (int64_t)2147483000 && n_cols < (int64_t)2147483000, "block too large for Int32 device indices");
  PA_REQUIRE(index_bytes == 8 || nnz < (int64_t)2147483000, "2^31 stored entries or more need 64-bit pointers");
  // counting transpose; columns end up ascending inside each row because we sweep columns in order
  std::vector<int64_t> rp(n_rows + 1, 0);
  for (int64_t p = 0; p < nnz; ++p) {
    const int64_t i = read_index(rowval, index_bytes, p) - index_base;
    PA_REQUIRE(i >= 0 && i < n_rows, "row index out of range at entry %lld", (long long)p);
    rp[i + 1]++;
  }
  for (int64_t r = 0; r < n_rows; ++r) rp[r + 1] += rp[r];
  std::vector<int32_t> col(nnz);
  std::vector<int64_t> fill(rp.begin(), rp.end() - 1);
  std::vector<double> val(nnz);
  for (int64_t j = 0; j < n_cols; ++j) {
    const int64_t a = read_index(colptr, index_bytes, j) - index_base, e = read_index(colptr, index_bytes, j + 1) - index_base;
    for (int64_t p = a; p < e; ++p) {
      const int64_t i = read_index(rowval, index_bytes, p) - index_base;
      const int64_t q = fill[i]++;
      col[q] = (int32_t)j;
      val[q] = nzval[p];
    }
  }
  csr_src src;
  src.col0 = col.data(); src.nzval = val.data();
  PA_TRY(csr_build(c, n_rows, n_cols, nnz, rp, src, out));
  // the caller's storage was CSC: its 5-argument product is SparseArrays' (alpha multiplies the vector entry first); pa_spmv follows
  const char *e = getenv("PA_CSC_ALPHA_INSIDE");
  if (!(e && atoi(e) == 0)) for (pa_csr *S = *out; S; S = S->next) S->alpha_inside = true;
  return PA_OK;
}

// Which of the two third-party 5-argument products a block follows when alpha != 1: 1 = SparseArrays' CSC method, a*(x*alpha)
// (the default of blocks made by pa_csr_create_from_csc), 0 = SparseMatricesCSR's, (a*x)*alpha (every other block).
extern "C" int pa_csr_set_alpha_inside(pa_csr *A, int on) {
  PA_REQUIRE(A != nullptr, "block is NULL");
  for (pa_csr *S = A; S; S = S->next) S->alpha_inside = on != 0;
  return PA_OK;
}

extern "C" int pa_csr_update_values(pa_csr *A, const double *nzval) {
  PA_REQUIRE(A && (nzval || A->t_nnz == 0), "bad arguments");
  if (A->t_nnz == 0) return PA_OK;
  PA_HIP(hipSetDevice(A->ctx->device));
  double *d_all = nullptr;                                  // column split: the pieces gather from the caller's order
  if (A->colsplit) {
    PA_HIP(pa_raw_malloc(&d_all, sizeof(double) * (size_t)A->t_nnz));
    if (hipMemcpyAsync(d_all, nzval, sizeof(double) * (size_t)A->t_nnz, hipMemcpyHostToDevice, A->ctx->s[0]) != hipSuccess) {
      (void)pa_raw_free(d_all);
      pa_set_err("pa_csr_update_values: upload failed");
      return PA_ERR_HIP;
    }
  }
  for (pa_csr *S = A; S; S = S->next) {
    if (S->use_vdict || S->vdict_stale) { S->vdict_stale = true; S->vdict_products = 0; }
    S->use_vdict = false;            // the codes describe the old values: back to the fp64 stream (vdict_maintain renews them)
    if (!S->nnz) continue;
    if (A->colsplit) hipLaunchKernelGGL(k_gather_values, dim3(grid_for(S->nnz, 256)), dim3(256), 0, A->ctx->s[0], S->d_val, (const double *)d_all, S->d_src, S->nnz);
    else PA_HIP(hipMemcpyAsync(S->d_val, nzval + S->nnz0, sizeof(double) * S->nnz, hipMemcpyHostToDevice, A->ctx->s[0]));
  }
  PA_HIP(hipStreamSynchronize(A->ctx->s[0]));
  if (d_all) (void)pa_raw_free(d_all);
  PA_HIP(hipGetLastError());
  return vdict_after_update(A);
}

extern "C" int pa_csr_update_values_from(pa_csr *A, const pa_vec *src, int64_t offset) {
  PA_REQUIRE(A && src && offset >= 0, "bad arguments");
  PA_REQUIRE(offset + A->t_nnz <= src->n_own + src->n_ghost, "source vector too short for nnz=%lld at offset %lld",
             (long long)A->t_nnz, (long long)offset);
  if (A->t_nnz == 0) return PA_OK;
  PA_HIP(hipSetDevice(A->ctx->device));
  for (pa_csr *S = A; S; S = S->next) {
    if (S->use_vdict || S->vdict_stale) { S->vdict_stale = true; S->vdict_products = 0; }
    S->use_vdict = false;
    if (!S->nnz) continue;
    if (A->colsplit) hipLaunchKernelGGL(k_gather_values, dim3(grid_for(S->nnz, 256)), dim3(256), 0, A->ctx->s[0], S->d_val, (const double *)(src->d + offset), S->d_src, S->nnz);
    else PA_HIP(hipMemcpyAsync(S->d_val, src->d + offset + S->nnz0, sizeof(double) * S->nnz, hipMemcpyDeviceToDevice, A->ctx->s[0]));
  }
  PA_HIP(hipGetLastError());
  return vdict_after_update(A);
}

static void csr_free_chain(pa_csr *A) {
  while (A) {
    pa_csr *n = A->next;
    pa_dev_free(A->ctx, A->d_crp);
    pa_dev_free(A->ctx, A->d_col);
    if (A->d_raw_col) pa_dev_free(A->ctx, A->d_raw_col);
    if (A->d_src) pa_dev_free(A->ctx, A->d_src);
    if (A->d_chain) pa_dev_free(A->ctx, A->d_chain);
    pa_dev_free(A->ctx, A->d_val);
    pa_dev_free(A->ctx, A->d_chunk_row);
    if (A->d_chunk_rp) pa_dev_free(A->ctx, A->d_chunk_rp);
    if (A->d_row_ids) pa_dev_free(A->ctx, A->d_row_ids);
    if (A->d_col16) pa_dev_free(A->ctx, A->d_col16);
    if (A->d_win) pa_dev_free(A->ctx, A->d_win);
    if (A->d_chunk_p) pa_dev_free(A->ctx, A->d_chunk_p);
    if (A->d_chunk_cmax) pa_dev_free(A->ctx, A->d_chunk_cmax);
    if (A->d_xw_grp) pa_dev_free(A->ctx, A->d_xw_grp);
    if (A->d_xw_rest) pa_dev_free(A->ctx, A->d_xw_rest);
    if (A->d_pdesc) pa_dev_free(A->ctx, A->d_pdesc);
    if (A->d_pdelta) pa_dev_free(A->ctx, A->d_pdelta);
    if (A->d_code) pa_dev_free(A->ctx, A->d_code);
    if (A->d_dict) pa_dev_free(A->ctx, A->d_dict);
    pa_pell_free(A);
    delete A;
    A = n;
  }
}

extern "C" int pa_csr_destroy(pa_csr *A) {
  if (!A) return PA_OK;
  (void)hipSetDevice(A->ctx->device);
  (void)hipStreamSynchronize(A->ctx->s[0]);      // (both: the arena hands these blocks to the next caller at once, whereas
  (void)hipStreamSynchronize(A->ctx->s[1]);      // hipFree used to synchronise the whole device)
  pa_watch_drop_csr(A);
  csr_free_chain(A);
  return PA_OK;
}

extern "C" int pa_csr_info(const pa_csr *A, int64_t *n_rows, int64_t *n_cols, int64_t *nnz, int64_t *n_chunks,
                           int64_t *n_nonempty, int64_t *n_long) {
  PA_REQUIRE(A != nullptr, "csr is NULL");
  int64_t ch = 0, ne = 0, nl = 0;
  for (const pa_csr *S = A; S; S = S->next) { ch += S->n_chunks; ne += S->n_nonempty; nl += S->n_long; }
  if (n_rows) *n_rows = A->t_rows;
  if (n_cols) *n_cols = A->n_cols;
  if (nnz) *nnz = A->t_nnz;
  if (n_chunks) *n_chunks = ch;
  if (n_nonempty) *n_nonempty = ne;
  if (n_long) *n_long = nl;
  return PA_OK;
}

// Host-only self-check of the row split and of the two column encoders: build them exactly as csr_build does and
// decode every entry on the host with the kernel's arithmetic; any mismatch with colval is an error.
extern "C" int pa_host_check_spmv_encodings(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t *rowptr,
                                            const int32_t *colval, int index_base, int64_t *n_chunks, int64_t *n_pattern,
                                            int64_t *n_c16, int64_t *n_patterns) {
  PA_REQUIRE(rowptr && (nnz == 0 || colval) && (index_base == 0 || index_base == 1), "bad arguments");
  std::vector<int32_t> crp(n_rows + 1), col(nnz), row_ids;
  for (int64_t r = 0; r <= n_rows; ++r) crp[r] = rowptr[r] - index_base;
  for (int64_t p = 0; p < nnz; ++p) col[p] = colval[p] - index_base;
  {  // the compaction rule of csr_build
    int64_t n_nonempty = 0;
    for (int64_t r = 0; r < n_rows; ++r) n_nonempty += crp[r + 1] > crp[r];
    if (n_rows > 0 && n_nonempty * 2 < n_rows) {
      std::vector<int32_t> c2(1, 0);
      for (int64_t r = 0; r < n_rows; ++r)
        if (crp[r + 1] > crp[r]) { row_ids.push_back((int32_t)r); c2.push_back(crp[r + 1]); }
      crp.swap(c2);
      n_rows = (int64_t)row_ids.size();
    }
  }
  std::vector<int32_t> chunk_row;
  int64_t n_long = 0;
  pa_build_chunks(crp.data(), n_rows, PA_SPMV_CHUNK_NNZ, 4096, chunk_row, &n_long);
  const int64_t nch = (int64_t)chunk_row.size() - 1;
  // both forms of the streams: full length (no descriptors) and compacted next to the row patterns
  pa_col_streams full, cs;
  pa_encode_columns(crp.data(), col.data(), nullptr, n_rows, chunk_row, PA_SPMV_CHUNK_NNZ, false, true, 1, full);
  pa_encode_columns(crp.data(), col.data(), row_ids.empty() ? nullptr : row_ids.data(), n_rows, chunk_row, PA_SPMV_CHUNK_NNZ,
                    true, true, 1, cs);
  PA_REQUIRE(full.full && full.n_c16 + full.n_c32 == nch, "chunk counts of the full-length streams");
  PA_REQUIRE(cs.n_pattern + cs.n_c16 + cs.n_c32 == nch, "chunk counts of the compacted streams");
  const int64_t npat = cs.n_pattern;
  const std::vector<int32_t> &pdesc = cs.pdesc, &pdelta = cs.pdelta;
  for (int64_t c = 0; c < nch; ++c) {
    const int64_t r0 = chunk_row[c], r1 = chunk_row[c + 1], p0 = crp[r0], p1 = crp[r1];
    const bool is_long = (p1 - (p0 & ~1)) > PA_SPMV_CHUNK_NNZ;
    PA_REQUIRE(r1 > r0, "empty chunk %lld", (long long)c);
    PA_REQUIRE(!is_long || r1 - r0 == 1, "chunk %lld overflows the LDS stage", (long long)c);
    if (full.win[c * PA_C16_WINDOWS] >= 0 && !is_long)
      for (int64_t p = p0; p < p1; ++p) {
        const int32_t dec = full.win[c * PA_C16_WINDOWS + (full.c16[p] >> 12)] + (full.c16[p] & 4095);
        PA_REQUIRE(dec == col[p], "c16 decode mismatch at entry %lld", (long long)p);
      }
    if (!cs.use_pattern) continue;
    const int32_t *d = &pdesc[(size_t)c * PA_PDESC_INTS];
    if (d[0] > 0) {
      for (int64_t p = p0; p < p1; ++p) {
        const int q = (int)(p - p0);
        const int s = (q >= d[1]) + (q >= d[2]) + (q >= d[3]);
        const bool strided = !row_ids.empty();
        const int t = q - (s ? d[s] : 0), L = strided ? (d[8 + s] & 255) : d[8 + s], stride = strided ? (d[8 + s] >> 8) : 1;
        const int rr = L == 1 ? t : (int)(((uint64_t)(uint32_t)t * (uint64_t)(0xFFFFFFFFu / (uint32_t)L + 1u)) >> 32);
        const int32_t dec = d[4 + s] + rr * stride + pdelta[(size_t)d[12 + s] * PA_PAT_MAXLEN + (t - rr * L)];
        PA_REQUIRE(dec == col[p], "pattern decode mismatch at entry %lld (chunk %lld)", (long long)p, (long long)c);
      }
    } else if (cs.use_c16 && cs.win[c * PA_C16_WINDOWS] >= 0 && !is_long) {     // compacted 16-bit stream, the kernel's indexing
      for (int64_t p = p0; p < p1; ++p) {
        const int64_t k = p + d[1];
        PA_REQUIRE(k >= 0 && k + 1 < (int64_t)cs.c16.size(), "compacted c16 slot out of range (chunk %lld)", (long long)c);
        const int32_t dec = cs.win[c * PA_C16_WINDOWS + (cs.c16[k] >> 12)] + (cs.c16[k] & 4095);
        PA_REQUIRE(dec == col[p], "compacted c16 decode mismatch at entry %lld", (long long)p);
      }
    } else {                                                       // compacted 32-bit stream
      for (int64_t p = p0; p < p1; ++p) {
        const int64_t k = p + d[2];
        PA_REQUIRE(k >= 0 && k + 1 < (int64_t)cs.c32.size(), "compacted 32-bit slot out of range (chunk %lld)", (long long)c);
        PA_REQUIRE(cs.c32[k] == col[p], "compacted 32-bit column mismatch at entry %lld", (long long)p);
      }
    }
  }
  const int64_t nfall = full.n_c32;
  if (n_chunks) *n_chunks = nch;
  if (n_pattern) *n_pattern = npat;
  if (n_c16) *n_c16 = nch - nfall;             // (of the full-length encoding: what the 16-bit windows COULD carry)
  if (n_patterns) *n_patterns = (int64_t)pdelta.size() / PA_PAT_MAXLEN;
  (void)n_cols;
  return PA_OK;
}

// Debugging / testing: a copy of one of the arrays the product kernel reads (first slab), so that two ways of building
// them can be compared byte for byte.  which: 0 row pointers, 1 32-bit columns, 2 16-bit codes, 3 windows, 4 pattern
// descriptors, 5 pattern table, 6 chunk table, 7 compacted row ids.  *bytes = size of the array; copied when it fits.
extern "C" int pa_csr_debug_array(const pa_csr *A, int which, void *host, int64_t capacity, int64_t *bytes) {
  PA_REQUIRE(A && bytes, "bad arguments");
  const int64_t pad = 8;
  const void *d = nullptr;
  int64_t n = 0;
  switch (which) {
    case 0: d = A->d_crp; n = 4 * (A->n_crows + 1); break;
    case 1: d = A->d_col; n = 4 * (A->n_col32 + pad); break;
    case 2: d = A->d_col16; n = A->d_col16 ? 2 * A->n_col16 : 0; break;
    case 3: d = A->d_win; n = A->d_win ? 4 * A->n_chunks * PA_C16_WINDOWS : 0; break;
    case 4: d = A->d_pdesc; n = A->d_pdesc ? 4 * A->n_chunks * PA_PDESC_INTS : 0; break;
    case 5: d = A->d_pdelta; n = A->d_pdelta ? 4 * A->n_pdelta : 0; break;
    case 6: d = A->d_chunk_row; n = 4 * (A->n_chunks + 1); break;
    case 7: d = A->d_row_ids; n = A->d_row_ids ? 4 * A->n_crows : 0; break;
    default: pa_set_err("unknown array %d", which); return PA_ERR_ARG;
  }
  *bytes = n;
  if (host && n > 0 && n <= capacity) {
    PA_HIP(hipSetDevice(A->ctx->device));
    PA_HIP(hipStreamSynchronize(A->ctx->s[0]));
    PA_HIP(hipMemcpy(host, d, (size_t)n, hipMemcpyDeviceToHost));
  }
  return PA_OK;
}

extern "C" int pa_csr_encoding(const pa_csr *A, int64_t *n_pattern, int64_t *n_c16, int64_t *n_c32) {
  PA_REQUIRE(A != nullptr, "csr is NULL");
  int64_t tp = 0, t16 = 0, t32 = 0;
  for (const pa_csr *S = A; S; S = S->next) {
    tp += S->n_pattern_chunks; t16 += S->n_c16_chunks; t32 += S->n_c32_chunks;
  }
  if (n_pattern) *n_pattern = tp;
  if (n_c16) *n_c16 = t16;
  if (n_c32) *n_c32 = t32;
  return PA_OK;
}

extern "C" int pa_csr_xwin_info(const pa_csr *A, int64_t *n_groups, int64_t *n_chunks, int64_t *staged_x_entries,
                                int64_t *n_big_groups) {
  PA_REQUIRE(A != nullptr, "csr is NULL");
  int64_t g = 0, k = 0, st = 0, big = 0;
  for (const pa_csr *S = A; S; S = S->next) {
    g += S->n_xw_groups; k += S->n_xw_chunks; st += S->xw_staged; big += S->n_xw_tier[1] + S->n_xw_tier[2];
  }
  if (n_groups) *n_groups = g;
  if (n_chunks) *n_chunks = k;
  if (staged_x_entries) *staged_x_entries = st;
  if (n_big_groups) *n_big_groups = big;
  return PA_OK;
}

extern "C" int pa_csr_xring_info(const pa_csr *A, int64_t *n_ring_groups) {
  PA_REQUIRE(A && n_ring_groups, "bad arguments");
  int64_t n = 0;
  for (const pa_csr *S = A; S; S = S->next) n += S->n_xw_ring;
  *n_ring_groups = n;
  return PA_OK;
}

extern "C" int pa_csr_device_bytes(const pa_csr *A, int64_t *bytes) {
  PA_REQUIRE(A && bytes, "bad arguments");
  int64_t t = 0;
  for (const pa_csr *S = A; S; S = S->next) {
    const int64_t pad = 8;
    t += 4 * (S->n_crows + 1) + 4 * (S->n_col32 + pad) + 8 * (S->nnz + pad) + 12 * (S->n_chunks + 1);
    if (S->use_c16) t += 2 * S->n_col16 + 4 * S->n_chunks * PA_C16_WINDOWS;
    if (S->n_xw_groups) t += 4 * (S->n_chunks + 1) + 16 * S->n_xw_groups + 4 * S->n_xw_rest + (S->n_xw_ring ? 4 * S->n_chunks : 0);
    if (S->use_pattern) t += 4 * S->n_chunks * PA_PDESC_INTS + 4 * S->n_pdelta;
    if (S->use_vdict) t += S->nnz + pad + 8 * PA_VDICT_MAX;
    if (S->compact) t += 4 * S->n_crows;
  }
  *bytes = t;
  return PA_OK;
}

// Bytes one product MUST read from the block's own storage (each exactly once): values, the row pointers, the chunk
// table, and per chunk whatever gives it its columns -- a pattern descriptor (no column stream), the window table + the
// 16-bit stream, or 32-bit columns.  With x read once and y written once this is the compulsory HBM traffic of pa_spmv
// ("moved bytes"), as opposed to the reference's CSR bytes (12 per stored entry) the SURVEY's roofline is quoted on.
extern "C" int pa_csr_stream_bytes(const pa_csr *A, int64_t *bytes) {
  PA_REQUIRE(A && bytes, "bad arguments");
  int64_t t = 0;
  for (const pa_csr *S = A; S; S = S->next) {
    if (const int pm = pa_pell_mode(S)) { t += pa_pell_stream_bytes(S, pm); continue; }
    t += (S->use_vdict ? 1 : 8) * S->nnz + 4 * (S->n_crows + 1);
    if (S->use_vdict) t += 8 * PA_VDICT_MAX;
    t += 8 * (S->n_chunks + 1);                                                    // {row, pointer} pairs
    if (S->use_pattern) t += 4 * S->n_chunks * PA_PDESC_INTS + 4 * S->n_pdelta;
    if (S->use_c16) t += 4 * (S->n_chunks - S->n_pattern_chunks) * PA_C16_WINDOWS;
    if (S->n_xw_groups) t += 4 * (S->n_chunks + 1) + 16 * S->n_xw_groups + 4 * S->n_xw_rest + (S->n_xw_ring ? 4 * S->n_chunks : 0);
    t += 2 * S->nnz_c16 + 4 * S->nnz_c32;
    if (S->compact) t += 4 * S->n_crows;
  }
  *bytes = t;
  return PA_OK;
}

// Host-only self-check of the x-window groups (pa_spmv_xwin.h): built as csr_build_slab builds them; every chunk is in
// exactly one group or in the rest list, a group's chunks all read the 16-bit stream, every column of a group lies in its
// window, and the window fits the kernel's LDS stage.
extern "C" int pa_host_check_xw_groups(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t *rowptr, const int32_t *colval,
                                       int index_base, int64_t *n_groups, int64_t *n_grouped_chunks, int64_t *staged_x_entries,
                                       int64_t *grouped_entries, int64_t *n_big_groups) {
  PA_REQUIRE(rowptr && (nnz == 0 || colval) && (index_base == 0 || index_base == 1), "bad arguments");
  std::vector<int32_t> crp(n_rows + 1), col(nnz);
  for (int64_t r = 0; r <= n_rows; ++r) crp[r] = rowptr[r] - index_base;
  for (int64_t p = 0; p < nnz; ++p) col[p] = colval[p] - index_base;
  std::vector<int32_t> chunk_row;
  int64_t n_long = 0;
  pa_build_chunks(crp.data(), n_rows, PA_SPMV_CHUNK_NNZ, 4096, chunk_row, &n_long);
  const int64_t nch = (int64_t)chunk_row.size() - 1;
  pa_col_streams full;
  pa_encode_columns(crp.data(), col.data(), nullptr, n_rows, chunk_row, PA_SPMV_CHUNK_NNZ, false, true, 1, full);
  pa_xw_plan P;
  const char *er = getenv("PA_SPMV_XRING");
  std::vector<int32_t> cmaxv;
  if (full.use_c16) pa_plan_xw(crp.data(), col.data(), chunk_row, full.win.data(), false, P, 1, er ? atoi(er) : 1, &cmaxv);
  else for (int64_t c = 0; c < nch; ++c) P.rest.push_back((int32_t)c);
  const std::vector<pa_xw_group> &groups = P.groups;
  const std::vector<int32_t> &rest = P.rest;
  const int64_t grouped = P.grouped, staged = P.staged;
  int64_t in_groups = 0;
  const int64_t n_windows = P.n_tier[0] + P.n_tier[1] + P.n_tier[2];
  PA_REQUIRE(n_windows + P.n_ring == (int64_t)groups.size(), "tier counts");
  std::vector<char> seen(nch, 0);
  int64_t check_staged = 0, check_grouped = 0;
  for (size_t gi = 0; gi < groups.size(); ++gi) {
    const pa_xw_group &g = groups[gi];
    if ((int64_t)gi >= n_windows) {
      // a ring group: replay k_spmv_xring's rounds (2 chunks each) -- every column a chunk gathers must have been loaded
      // (>= the group's first column, <= the highest column loaded by its round) and not yet overwritten (within one ring
      // capacity below that highest column)
      PA_REQUIRE(g.cnt >= PA_XW_MING && g.cnt <= PA_XR_MAXG && g.first >= 0 && g.first + g.cnt <= nch, "ring group of %d chunks at %d", g.cnt, g.first);
      int hcur = -1;
      for (int c0 = g.first; c0 < g.first + g.cnt; c0 += 2) {
        for (int c = c0; c < std::min(c0 + 2, g.first + g.cnt); ++c) hcur = std::max(hcur, cmaxv[c]);
        for (int c = c0; c < std::min(c0 + 2, g.first + g.cnt); ++c) {
          PA_REQUIRE(!seen[c], "chunk %d in two groups", c);
          seen[c] = 1;
          const int64_t p0 = crp[chunk_row[c]], p1 = crp[chunk_row[c + 1]];
          PA_REQUIRE(full.win[(size_t)c * PA_C16_WINDOWS] >= 0 && p1 - (p0 & ~1) <= PA_SPMV_CHUNK_NNZ, "chunk %d has no 16-bit columns", c);
          for (int64_t p = p0; p < p1; ++p)
            PA_REQUIRE(col[p] >= g.wlo && col[p] <= hcur && col[p] > hcur - PA_XR_CAP, "column %d of chunk %d is not in the ring (loaded up to %d)", col[p], c, hcur);
          check_grouped += p1 - p0;
        }
      }
      PA_REQUIRE(g.wlen == hcur - g.wlo + 1, "ring group span");
      check_staged += g.wlen;
      in_groups += g.cnt;
      continue;
    }
    const int cap = (int64_t)gi < P.n_tier[0] ? PA_XW_CAP : (int64_t)gi < P.n_tier[0] + P.n_tier[1] ? PA_XW_CAP_MID : PA_XW_CAP_BIG;
    PA_REQUIRE(g.cnt >= PA_XW_MING && g.cnt <= PA_XW_MAXG, "group of %d chunks", g.cnt);
    PA_REQUIRE(g.first >= 0 && g.first + g.cnt <= nch, "group outside the block");
    PA_REQUIRE(g.wlo >= 0 && g.wlen >= 1 && g.wlo + g.wlen <= n_cols && g.wlen + 2 <= cap, "window [%d,+%d) does not fit", g.wlo, g.wlen);
    for (int c = g.first; c < g.first + g.cnt; ++c) {
      PA_REQUIRE(!seen[c], "chunk %d in two groups", c);
      seen[c] = 1;
      const int64_t p0 = crp[chunk_row[c]], p1 = crp[chunk_row[c + 1]];
      PA_REQUIRE(full.win[(size_t)c * PA_C16_WINDOWS] >= 0 && p1 - (p0 & ~1) <= PA_SPMV_CHUNK_NNZ, "chunk %d has no 16-bit columns", c);
      for (int64_t p = p0; p < p1; ++p)
        PA_REQUIRE(col[p] >= g.wlo && col[p] < g.wlo + g.wlen, "column %d of chunk %d outside its window", col[p], c);
      check_grouped += p1 - p0;
    }
    check_staged += g.wlen;
    in_groups += g.cnt;
  }
  for (int32_t c : rest) {
    PA_REQUIRE(c >= 0 && c < nch && !seen[c], "chunk %d listed twice", c);
    seen[c] = 1;
  }
  for (int64_t c = 0; c < nch; ++c) PA_REQUIRE(seen[c], "chunk %lld in no launch", (long long)c);
  for (size_t k = 1; k < rest.size(); ++k) PA_REQUIRE(rest[k] > rest[k - 1], "rest list not ascending");
  PA_REQUIRE(check_staged == staged && check_grouped == grouped, "group totals");
  if (n_groups) *n_groups = (int64_t)groups.size();
  if (n_big_groups) *n_big_groups = P.n_tier[1] + P.n_tier[2];
  if (n_grouped_chunks) *n_grouped_chunks = in_groups;
  if (staged_x_entries) *staged_x_entries = staged;
  if (grouped_entries) *grouped_entries = grouped;
  return PA_OK;
}

extern "C" int pa_csr_memory_class(const pa_csr *A, int *cls) {
  PA_REQUIRE(A && cls, "bad arguments");
  *cls = pa_mem_class(A->ctx, A->d_val);
  return PA_OK;
}

extern "C" int pa_vec_memory_class(const pa_vec *v, int *cls) {
  PA_REQUIRE(v && cls, "bad arguments");
  *cls = pa_mem_class(v->ctx, v->d);
  return PA_OK;
}

extern "C" int pa_csr_value_dict(const pa_csr *A, int *n_values) {
  PA_REQUIRE(A && n_values, "bad arguments");
  int n = 0;
  bool all = true;
  for (const pa_csr *S = A; S; S = S->next) {
    if (S->nnz == 0) continue;
    if (!S->use_vdict) all = false;
    n = std::max(n, S->n_dict);
  }
  *n_values = all ? n : 0;
  return PA_OK;
}

// the x-window launches of a slab (pa_spmv_xwin.h): small-window groups, big-window groups, and k_spmv_rowsplit over the
// chunks that are in no group; u != NULL: the fused dot (partial[chunk] as k_spmv_rowsplit's EPI 3 writes it)
void pa_launch_xwin(const pa_csr *S, const double *xs, double *ys, double alpha, double kbeta, const double *u, double *partial,
                    hipStream_t st) {
  if (!st) st = S->ctx->s[0];
  const pa_xw_group *grp = (const pa_xw_group *)S->d_xw_grp;
#define PA_LAUNCH_XW(SUB, DOT, XCAP, G, NG)                                                                                \
  hipLaunchKernelGGL((k_spmv_xwin<SUB, SPMV_NPT, SPMV_NT, DOT, XCAP>), dim3((((NG) + 7) / 8) * 8), dim3(256 * SUB), 0, st,  \
                     S->d_crp, S->d_col16, S->d_win, S->d_val, xs, ys, S->d_chunk_row, S->d_chunk_p, (G), (int)(NG),           \
                     (int)(((NG) + 7) / 8), (int)S->n_cols, alpha, kbeta, u, partial)
  const int64_t n0 = S->n_xw_tier[0], n1 = S->n_xw_tier[1], n2 = S->n_xw_tier[2];
  if (n0 > 0) {
    if (u) PA_LAUNCH_XW(PA_XW_SUB, true, PA_XW_CAP, grp, n0);
    else PA_LAUNCH_XW(PA_XW_SUB, false, PA_XW_CAP, grp, n0);
  }
  if (n1 > 0) {
    if (u) PA_LAUNCH_XW(4, true, PA_XW_CAP_MID, grp + n0, n1);
    else PA_LAUNCH_XW(4, false, PA_XW_CAP_MID, grp + n0, n1);
  }
#undef PA_LAUNCH_XW
  if (n2 > 0) {                                        // 128 KiB windows: one workgroup per CU of 2 x 256 lanes (512 lanes per chunk,
    // which lifts the ring kernel by 10 %, measured neutral here: +-7000 0.168 / 0.170 ms, +-5000 0.142 / 0.144; PA_SPMV_XWIN_BIG_LANES=512)
    static const int wide2 = getenv("PA_SPMV_XWIN_BIG_LANES") ? atoi(getenv("PA_SPMV_XWIN_BIG_LANES")) : 256;
#define PA_LAUNCH_XW2(DOT, BLKX, NPTX)                                                                                                  \
  hipLaunchKernelGGL((k_spmv_xwin<2, NPTX, SPMV_NT, DOT, PA_XW_CAP_BIG, BLKX>), dim3(((n2 + 7) / 8) * 8), dim3(2 * BLKX), 0, st,   \
                     S->d_crp, S->d_col16, S->d_win, S->d_val, xs, ys, S->d_chunk_row, S->d_chunk_p, grp + n0 + n1, (int)n2,            \
                     (int)((n2 + 7) / 8), (int)S->n_cols, alpha, kbeta, u, partial)
    if (wide2 == 512) { if (u) PA_LAUNCH_XW2(true, 512, 4); else PA_LAUNCH_XW2(false, 512, 4); }
    else { if (u) PA_LAUNCH_XW2(true, 256, SPMV_NPT); else PA_LAUNCH_XW2(false, 256, SPMV_NPT); }
#undef PA_LAUNCH_XW2
  }
  if (S->n_xw_ring > 0) {                              // runs of chunks served from the sliding x window
    const int ng = (int)S->n_xw_ring, gpx = (ng + 7) / 8;
    const pa_xw_group *rg = grp + n0 + n1 + n2;
    // lanes per chunk: 512 (4 entries each, the lanes past the chunk's 1536 entries idle) put 16 waves on the CU for the same LDS:
    // +-7900 0.172 ms = 4.9 TB/s algorithmic against 0.189 / 4.5 with 256 lanes (PA_SPMV_XRING_LANES=256)
    static const int wide = getenv("PA_SPMV_XRING_LANES") ? atoi(getenv("PA_SPMV_XRING_LANES")) : 512;
#define PA_LAUNCH_XR(DOT, BLKX, NPTX, UU, PP)                                                                                          \
  hipLaunchKernelGGL((k_spmv_xring<2, NPTX, SPMV_NT, DOT, BLKX>), dim3(gpx * 8), dim3(2 * BLKX), 0, st, S->d_crp, S->d_col16, S->d_win, \
                     S->d_val, xs, ys, S->d_chunk_row, S->d_chunk_p, S->d_chunk_cmax, rg, ng, gpx, (int)S->n_cols, alpha, kbeta, UU, PP)
    if (wide == 512) {
      if (u) PA_LAUNCH_XR(true, 512, 4, u, partial);
      else PA_LAUNCH_XR(false, 512, 4, (const double *)nullptr, (double *)nullptr);
    } else {
      if (u) PA_LAUNCH_XR(true, 256, SPMV_NPT, u, partial);
      else PA_LAUNCH_XR(false, 256, SPMV_NPT, (const double *)nullptr, (double *)nullptr);
    }
#undef PA_LAUNCH_XR
  }
  if (S->n_xw_rest > 0) {                              // what fits no group: the general kernel over a chunk list
    const int cpx = (int)((S->n_xw_rest + 7) / 8);
    if (u)
      hipLaunchKernelGGL((k_spmv_rowsplit<SPMV_BLK, SPMV_NPT, SPMV_NT, true, 0, 3, false>), dim3(cpx * 8), dim3(SPMV_BLK), 0,
                         st, S->d_crp, S->d_col, S->d_col16, S->d_win, S->d_pdesc, S->d_pdelta, S->d_val, xs, ys,
                         S->d_chunk_rp, S->d_row_ids, (int)S->n_xw_rest, cpx, 1.0, kbeta, partial, u,
                         (const double *)nullptr, (const unsigned char *)nullptr, (const double *)nullptr, S->d_xw_rest,
                         (int)S->n_cols - 1);
    else
      hipLaunchKernelGGL((k_spmv_rowsplit<SPMV_BLK, SPMV_NPT, SPMV_NT, true, 0, 0, false>), dim3(cpx * 8), dim3(SPMV_BLK), 0,
                         st, S->d_crp, S->d_col, S->d_col16, S->d_win, S->d_pdesc, S->d_pdelta, S->d_val, xs, ys,
                         S->d_chunk_rp, S->d_row_ids, (int)S->n_xw_rest, cpx, alpha, kbeta, (double *)nullptr,
                         (const double *)nullptr, (const double *)nullptr, (const unsigned char *)nullptr,
                         (const double *)nullptr, S->d_xw_rest, (int)S->n_cols - 1);
  }
}

// the product kernel on one slab, raw pointers (x: the block's column segment, ys: this slab's rows)
static void spmv_launch_slab(const pa_csr *S, const double *xs, double *ys, double alpha, double kbeta, hipStream_t st = nullptr) {
  if (!st) st = S->ctx->s[0];
  if (const int pm = pa_pell_mode(S)) {                // a pattern block: one lane per row, no LDS (pa_pell.h)
    (void)pa_pell_launch(S, pm, 0, xs, ys, alpha, kbeta, nullptr, nullptr, nullptr, st);
    return;
  }
  if (S->n_xw_groups > 0 && !S->use_vdict) {
    pa_launch_xwin(S, xs, ys, alpha, kbeta, nullptr, nullptr, st);
    return;
  }
  if (S->n_chunks > 0) {
      int cpx = (int)((S->n_chunks + 7) / 8);
      const int gcpx = cpx;
      if (S->ctx->sw.spmv_alternate && ((const_cast<pa_csr *>(S)->n_launched++) & 1)) cpx = -cpx;
#define PA_LAUNCH_SPMV(C16, PAT, VD)                                                                                     \
  hipLaunchKernelGGL((k_spmv_rowsplit<SPMV_BLK, SPMV_NPT, SPMV_NT, C16, PAT, 0, VD>), dim3(gcpx * 8), dim3(SPMV_BLK), 0,    \
                     st, S->d_crp, S->d_col, S->d_col16, S->d_win, S->d_pdesc, S->d_pdelta, S->d_val,           \
                     xs, ys, S->d_chunk_rp, S->d_row_ids, (int)S->n_chunks, cpx, alpha, kbeta,             \
                     (double *)nullptr, (const double *)nullptr, (const double *)nullptr, S->d_code, S->d_dict,         \
                     (const int *)nullptr, (int)S->n_cols - 1)
      const int sel_ = (S->use_pattern ? (S->compact ? 2 : 1) : 0) * 2 + (S->use_c16 ? 1 : 0);
      if (S->pad_products && !S->use_vdict && sel_ < 2) {
        if (sel_ == 1)
          hipLaunchKernelGGL((k_spmv_rowsplit<SPMV_BLK, SPMV_NPT, SPMV_NT, true, 0, 0, false, 4, true>), dim3(gcpx * 8), dim3(SPMV_BLK),
                             0, st, S->d_crp, S->d_col, S->d_col16, S->d_win, S->d_pdesc, S->d_pdelta, S->d_val, xs, ys,
                             S->d_chunk_rp, S->d_row_ids, (int)S->n_chunks, cpx, alpha, kbeta, (double *)nullptr,
                             (const double *)nullptr, (const double *)nullptr, S->d_code, S->d_dict, (const int *)nullptr,
                             (int)S->n_cols - 1);
        else
          hipLaunchKernelGGL((k_spmv_rowsplit<SPMV_BLK, SPMV_NPT, SPMV_NT, false, 0, 0, false, 4, true>), dim3(gcpx * 8), dim3(SPMV_BLK),
                             0, st, S->d_crp, S->d_col, S->d_col16, S->d_win, S->d_pdesc, S->d_pdelta, S->d_val, xs, ys,
                             S->d_chunk_rp, S->d_row_ids, (int)S->n_chunks, cpx, alpha, kbeta, (double *)nullptr,
                             (const double *)nullptr, (const double *)nullptr, S->d_code, S->d_dict, (const int *)nullptr,
                             (int)S->n_cols - 1);
      } else if (S->use_vdict) {
        const bool vd_two = pa_vd_two(S);
        if (S->ctx->capturing) { const_cast<pa_csr *>(S)->vd_captured = true; if (vd_two) const_cast<pa_csr *>(S)->vd_captured_two = true; }
        switch (sel_) {
          case 5: if (vd_two) PA_LAUNCH_SPMV(true, 2, 2); else PA_LAUNCH_SPMV(true, 2, 1); break;
          case 4: if (vd_two) PA_LAUNCH_SPMV(false, 2, 2); else PA_LAUNCH_SPMV(false, 2, 1); break;
          case 3: if (vd_two) PA_LAUNCH_SPMV(true, 1, 2); else PA_LAUNCH_SPMV(true, 1, 1); break;
          case 2: if (vd_two) PA_LAUNCH_SPMV(false, 1, 2); else PA_LAUNCH_SPMV(false, 1, 1); break;
          case 1: if (vd_two) PA_LAUNCH_SPMV(true, 0, 2); else PA_LAUNCH_SPMV(true, 0, 1); break;
          default: if (vd_two) PA_LAUNCH_SPMV(false, 0, 2); else PA_LAUNCH_SPMV(false, 0, 1); break;
        }
      } else {
        switch (sel_) {
          case 5: PA_LAUNCH_SPMV(true, 2, 0); break;
          case 4: PA_LAUNCH_SPMV(false, 2, 0); break;
          case 3: PA_LAUNCH_SPMV(true, 1, 0); break;
          case 2: PA_LAUNCH_SPMV(false, 1, 0); break;
          case 1: PA_LAUNCH_SPMV(true, 0, 0); break;
          default: PA_LAUNCH_SPMV(false, 0, 0); break;
        }
      }
#undef PA_LAUNCH_SPMV
  }
}

extern "C" int pa_spmv(const pa_csr *A, const pa_vec *x, int xseg, pa_vec *y, int yseg, double alpha, double beta) {
  PA_REQUIRE(A && x && y, "bad arguments");
  return pa_spmv_on(A, x, xseg, y, yseg, alpha, beta, A->ctx->s[0]);
}

// the product on a stream of the caller's choice (pa_mul_all queues the parts' own x ghost products on the comm stream, beside
// the next part's own x own)
int pa_spmv_on(const pa_csr *A, const pa_vec *x, int xseg, pa_vec *y, int yseg, double alpha, double beta, hipStream_t st) {
  int64_t xoff, xlen, yoff, ylen;
  PA_TRY(seg_range(x, xseg, &xoff, &xlen));
  PA_TRY(seg_range(y, yseg, &yoff, &ylen));
  // @boundscheck of spmv! (src/sparse_utils.jl:618-621)
  PA_REQUIRE(ylen == A->t_rows, "length(b)=%lld != size(A,1)=%lld", (long long)ylen, (long long)A->t_rows);
  PA_REQUIRE(xlen == A->n_cols, "length(x)=%lld != size(A,2)=%lld", (long long)xlen, (long long)A->n_cols);
  PA_REQUIRE(x->d != y->d || xseg != yseg, "x and y alias");
  pa_ctx *c = A->ctx;
  PA_HIP(hipSetDevice(c->device));
  vdict_maintain(A);
  const double *xs_all = x->d + xoff;
  if (A->alpha_inside && alpha != 1.0) {
    // A block made from CSC storage: SparseArrays.mul!(y,A::SparseMatrixCSC,x,alpha,beta) forms axj = x[col]*alpha once per column
    // and adds nzval*axj -- a*(x*alpha), one rounding apart from the CSR method's (a*x)*alpha unless alpha is a power of two.
    // x .* alpha goes to a scratch vector (one pass over x: 16 B per column next to 12 B per stored entry) and the kernel runs with
    // alpha = 1: per output entry the same products, added in ascending column as the column-major scatter loop adds them.
    const int sx = st == c->s[1] ? 1 : 0;            // (a scratch per stream: pa_mul_all runs own x ghost on the comm stream beside own x own)
    if (xlen > c->n_xalpha[sx]) {
      PA_REQUIRE(!c->capturing, "the scaled copy of x needs its scratch before a capture opens (run the product once eagerly)");
      PA_HIP(hipStreamSynchronize(st));
      if (c->d_xalpha[sx]) pa_dev_free(c, c->d_xalpha[sx]);
      c->d_xalpha[sx] = nullptr; c->n_xalpha[sx] = 0;
      PA_TRY(pa_dev_alloc(c, (void **)&c->d_xalpha[sx], sizeof(double) * (size_t)(xlen + 2), PA_MEM_VECTOR));
      c->n_xalpha[sx] = xlen;
    }
    if (xlen) hipLaunchKernelGGL(k_axpby, dim3(grid_for(xlen, 256)), dim3(256), 0, st, c->d_xalpha[sx], xs_all, xlen, alpha, 0.0);
    xs_all = c->d_xalpha[sx];
    alpha = 1.0;
  }
  if (A->colsplit && A->d_chain && c->sw.chain_fused) {
    // a column-split chain whose pieces share their row runs: one launch, y written once (k_spmv_xring_chain).  A piece that has
    // gone over to the one-byte value stream in the meantime reads through k_spmv_rowsplit: then piece by piece as before.
    bool ring = true;
    for (const pa_csr *S = A; S; S = S->next) ring = ring && !S->use_vdict && S->n_xw_ring > 0;
    if (ring) {
      const int ng = (int)A->chain_groups, gpx = (ng + 7) / 8;
      hipLaunchKernelGGL((k_spmv_xring_chain<2, 4, SPMV_NT, 512>), dim3(gpx * 8), dim3(1024), 0, st, (const pa_chain_piece *)A->d_chain,
                         A->chain_pieces, xs_all, y->d + yoff, ng, gpx, (int)A->n_cols, alpha, beta);
      ++c->n_chain_fused;
      PA_HIP(hipGetLastError());
      return PA_OK;
    }
  }
  for (const pa_csr *S = A; S; S = S->next) {          // one slab unless the block has 2^31 stored entries or more
    double *ys = y->d + yoff + S->row0;
    double kbeta = S->accumulate ? 1.0 : beta;             // (a column piece behind the first adds onto what the pieces before it left)
    if (S->compact && kbeta != 1.0) {
      // rows without stored entries still get beta*y (rmul!/fill! of the reference); the kernel then accumulates
      if (S->n_rows) hipLaunchKernelGGL(k_scale, dim3(grid_for(S->n_rows, 256)), dim3(256), 0, st, ys, S->n_rows, beta);
      kbeta = 1.0;
    }
    spmv_launch_slab(S, xs_all, ys, alpha, kbeta, st);
  }
  PA_HIP(hipGetLastError());
  return PA_OK;
}

// ---- placement A/B of a product's write stream with the PRODUCT kernel itself (round 4) ----------------------------------
// The arena places y by rule after a 40 us stand-in probe of undocumented hardware behaviour (pa_arena.hip); whether the rule
// was right on THIS box is answered by timing y = A*x with y where it is, in every other memory class the held extents have
// room in (the matrix streams' own class included: the control that should be ~13 % slower) and in a plain hipMalloc --
// `rounds` interleaved passes of `reps` launches each, the minimum per place.  where[i]: 0..2 = arena class, 9 = plain
// allocation the pair check had verified, -1 = plain / outside the arena; entry 0 is y's current place.  When another place is
// more than 1.5 % faster, y's storage MOVES there (contents copied; do this before capturing graphs that hold y's address) and
// *chosen names it; otherwise *chosen = 0.  Events on the compute stream; returns after a synchronize.
extern "C" int pa_spmv_tune_output(const pa_csr *A, const pa_vec *x, int xseg, pa_vec *y, int reps, int rounds, int32_t capacity,
                                   int32_t *where, double *ms, int32_t *n_out, int32_t *chosen) {
  PA_REQUIRE(A && x && y && where && ms && n_out && chosen && capacity >= 1, "bad arguments");
  PA_REQUIRE(y->owned, "the vector's storage is the caller's (pa_vec_wrap): it cannot move");
  PA_REQUIRE(reps >= 1 && rounds >= 1, "reps and rounds must be positive");
  pa_ctx *c = A->ctx;
  PA_REQUIRE(!c->capturing, "not inside a graph capture");
  PA_HIP(hipSetDevice(c->device));
  const size_t bytes = sizeof(double) * (size_t)(y->n_own + y->n_ghost + 2);
  struct cand { int where; double *p; double best; };
  std::vector<cand> cs;
  const int cur = pa_mem_class(c, y->d);
  cs.push_back({cur, y->d, 1e30});
  for (int k = 0; k < 3 && (int)cs.size() < capacity; ++k) {
    if (k == cur) continue;
    void *q = nullptr;
    PA_TRY(pa_dev_alloc_at(c, &q, bytes, k));
    if (q) cs.push_back({k, (double *)q, 1e30});
  }
  if ((int)cs.size() < capacity) {
    void *q = nullptr;
    PA_TRY(pa_dev_alloc_at(c, &q, bytes, -1));
    if (q) cs.push_back({-1, (double *)q, 1e30});
  }
  auto drop_others = [&](size_t keep) {
    for (size_t i = 1; i < cs.size(); ++i) if (i != keep) pa_dev_free(c, cs[i].p);
  };
  hipEvent_t e0 = nullptr, e1 = nullptr;
  int st = PA_OK;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { pa_set_err("hipEventCreate failed"); st = PA_ERR_HIP; }
  for (size_t i = 1; i < cs.size() && st == PA_OK; ++i)
    if (hipMemsetAsync(cs[i].p, 0, bytes, c->s[0]) != hipSuccess) { pa_set_err("hipMemsetAsync failed"); st = PA_ERR_HIP; }
  for (int r = 0; r < rounds && st == PA_OK; ++r)
    for (size_t i = 0; i < cs.size() && st == PA_OK; ++i) {
      pa_vec t = *y;
      t.d = cs[i].p; t.owned = false;
      for (int k = 0; k < 2 && st == PA_OK; ++k) st = pa_spmv(A, x, xseg, &t, PA_SEG_OWN, 1.0, 0.0);
      if (st != PA_OK) break;
      (void)hipEventRecord(e0, c->s[0]);
      for (int k = 0; k < reps && st == PA_OK; ++k) st = pa_spmv(A, x, xseg, &t, PA_SEG_OWN, 1.0, 0.0);
      (void)hipEventRecord(e1, c->s[0]);
      if (hipEventSynchronize(e1) != hipSuccess) { pa_set_err("hipEventSynchronize failed"); st = PA_ERR_HIP; break; }
      float dt = 0;
      (void)hipEventElapsedTime(&dt, e0, e1);
      cs[i].best = std::min(cs[i].best, (double)dt / reps);
    }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st != PA_OK) { (void)hipStreamSynchronize(c->s[0]); drop_others(0); return st; }
  size_t best = 0;
  for (size_t i = 1; i < cs.size(); ++i) if (cs[i].best < cs[best].best) best = i;
  if (best != 0 && !(cs[best].best < 0.985 * cs[0].best)) best = 0;
  for (size_t i = 0; i < cs.size(); ++i) { where[i] = cs[i].where; ms[i] = cs[i].best; }
  *n_out = (int32_t)cs.size();
  *chosen = (int32_t)best;
  // the timed products overwrote the own segment of every candidate (y's included: y = A*x now); a move carries y's values over
  if (best != 0) {
    PA_HIP(hipMemcpyAsync(cs[best].p, y->d, bytes, hipMemcpyDeviceToDevice, c->s[0]));
    PA_HIP(hipStreamSynchronize(c->s[0]));
    PA_HIP(hipStreamSynchronize(c->s[1]));
    double *old = y->d;
    y->d = cs[best].p;
    pa_dev_free(c, old);
  }
  PA_HIP(hipStreamSynchronize(c->s[0]));
  drop_others(best);
  return PA_OK;
}

// PCI address of the context's device ("0000:75:00.0"): the key of its sysfs directory (/sys/bus/pci/devices/<id>/: clocks,
// power, partition modes) -- a box shows the sysfs entries of all its GPUs, whichever ones the process may use.
extern "C" int pa_ctx_pci_bus_id(pa_ctx *c, char *out, size_t len) {
  PA_REQUIRE(c && out && len >= 16, "bad arguments");
  PA_HIP(hipDeviceGetPCIBusId(out, (int)len, c->device));
  for (char *q = out; *q; ++q) if (*q >= 'A' && *q <= 'F') *q = (char)(*q - 'A' + 'a');
  return PA_OK;
}

// One multicolour Gauss-Seidel sweep written as SpMV with a fused update: colour k's rows are the (row-compacted)
// block blocks[k] (n_own x n_local, every stored entry of those rows); its launch gathers from x and updates x's own
// rows of that colour in place, x[row] += (b[row] - (A x)[row]) / diag[row].  Colours run in ascending order
// (backward != 0: descending), one launch each on the compute stream.
static int gs_color_check(pa_csr *const *blocks, int n_colors, pa_vec *x, const pa_vec *b, const pa_vec *diag) {
  PA_REQUIRE(blocks && x && b && diag && n_colors >= 0, "bad arguments");
  for (int k = 0; k < n_colors; ++k) {
    const pa_csr *A = blocks[k];
    PA_REQUIRE(A != nullptr, "colour block %d is NULL", k);
    PA_REQUIRE(A->t_rows == x->n_own && A->n_cols == x->n_own + x->n_ghost, "colour block %d is %lld x %lld, x has %lld own + %lld ghost",
               k, (long long)A->t_rows, (long long)A->n_cols, (long long)x->n_own, (long long)x->n_ghost);
    PA_REQUIRE(A->next == nullptr, "colour block %d is stored in several slabs (>= 2^31 entries): not supported by the fused sweep", k);
  }
  PA_REQUIRE(b->n_own == x->n_own && diag->n_own == x->n_own, "b / diag own sizes differ from x");
  PA_REQUIRE(x->d != b->d && x->d != diag->d, "x aliases b or diag");
  return PA_OK;
}

// one colour: x[row] += (b[row] - (A x)[row]) / diag[row] on the rows of the block, in place
static void gs_color_launch(pa_ctx *c, const pa_csr *A, pa_vec *x, const pa_vec *b, const pa_vec *diag) {
  if (A->n_chunks == 0) return;
  if (const int pm = pa_pell_mode(A)) {                // a pattern block: one lane per row (pa_pell.h), the same update, the same bits
    (void)pa_pell_launch(A, pm, 1, nullptr, nullptr, 1.0, 0.0, x->d, b->d, diag->d, c->s[0]);
    return;
  }
  const int cpx = (int)((A->n_chunks + 7) / 8);
#define PA_LAUNCH_GS(C16, PAT, VD)                                                                                       \
  hipLaunchKernelGGL((k_spmv_rowsplit<SPMV_BLK, SPMV_NPT, SPMV_NT, C16, PAT, 1, VD>), dim3(cpx * 8), dim3(SPMV_BLK), 0,    \
                     c->s[0], A->d_crp, A->d_col, A->d_col16, A->d_win, A->d_pdesc, A->d_pdelta, A->d_val,           \
                     (const double *)nullptr, (double *)nullptr, A->d_chunk_rp, A->d_row_ids, (int)A->n_chunks, cpx, \
                     1.0, 0.0, x->d, (const double *)b->d, (const double *)diag->d, A->d_code, A->d_dict,                  \
                     (const int *)nullptr, (int)A->n_cols - 1)
  const int sel_ = (A->use_pattern ? (A->compact ? 2 : 1) : 0) * 2 + (A->use_c16 ? 1 : 0);
  if (A->use_vdict) {
    const bool vd_two = pa_vd_two(A);
    if (c->capturing) { const_cast<pa_csr *>(A)->vd_captured = true; if (vd_two) const_cast<pa_csr *>(A)->vd_captured_two = true; }
    switch (sel_) {
      case 5: if (vd_two) PA_LAUNCH_GS(true, 2, 2); else PA_LAUNCH_GS(true, 2, 1); break;
      case 4: if (vd_two) PA_LAUNCH_GS(false, 2, 2); else PA_LAUNCH_GS(false, 2, 1); break;
      case 3: if (vd_two) PA_LAUNCH_GS(true, 1, 2); else PA_LAUNCH_GS(true, 1, 1); break;
      case 2: if (vd_two) PA_LAUNCH_GS(false, 1, 2); else PA_LAUNCH_GS(false, 1, 1); break;
      case 1: if (vd_two) PA_LAUNCH_GS(true, 0, 2); else PA_LAUNCH_GS(true, 0, 1); break;
      default: if (vd_two) PA_LAUNCH_GS(false, 0, 2); else PA_LAUNCH_GS(false, 0, 1); break;
    }
  } else {
    switch (sel_) {
      case 5: PA_LAUNCH_GS(true, 2, 0); break;
      case 4: PA_LAUNCH_GS(false, 2, 0); break;
      case 3: PA_LAUNCH_GS(true, 1, 0); break;
      case 2: PA_LAUNCH_GS(false, 1, 0); break;
      case 1: PA_LAUNCH_GS(true, 0, 0); break;
      default: PA_LAUNCH_GS(false, 0, 0); break;
    }
  }
#undef PA_LAUNCH_GS
}

extern "C" int pa_gs_color_sweep(pa_csr *const *blocks, int n_colors, pa_vec *x, const pa_vec *b, const pa_vec *diag,
                                 int backward) {
  PA_TRY(gs_color_check(blocks, n_colors, x, b, diag));
  pa_ctx *c = x->ctx;
  PA_HIP(hipSetDevice(c->device));
  for (int i = 0; i < n_colors; ++i) gs_color_launch(c, blocks[backward ? n_colors - 1 - i : i], x, b, diag);
  PA_HIP(hipGetLastError());
  return PA_OK;
}

// the first colour of a sweep over x == 0: its rows see (A x)[row] == 0, so the update is b / diag -- the colour launch's own
// expression with a zero row sum, without reading the block's entries (same bits: x[row] is +0.0, b - (+-0.0) is b)
__global__ void k_gs_first_color_zero(double *x, const double *__restrict__ b, const double *__restrict__ diag,
                                      const int32_t *__restrict__ crp, const int32_t *__restrict__ row_ids, int nc) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= nc || crp[c + 1] == crp[c]) return;
  const int row = row_ids ? row_ids[c] : c;
  x[row] = x[row] + (b[row] - 0.0) / diag[row];
}

// The symmetric sweep of the multicolour smoother in one call: colours 0 .. K-1, then K-2 .. 0.  The backward half starts
// at K-2: colour K-1 has just been relaxed and nothing it couples to has changed since, so relaxing it again adds
// (b - A x)[row] / diag[row] == 0 up to the rounding of the first update (rows of one colour are not coupled).
// zero_guess != 0: the caller guarantees x == 0 (own and ghost entries); colour 0 then takes the shortcut above.
extern "C" int pa_gs_color_symmetric_sweep(pa_csr *const *blocks, int n_colors, pa_vec *x, const pa_vec *b, const pa_vec *diag,
                                           int zero_guess) {
  PA_TRY(gs_color_check(blocks, n_colors, x, b, diag));
  pa_ctx *c = x->ctx;
  PA_HIP(hipSetDevice(c->device));
  for (int k = 0; k < n_colors; ++k) {
    const pa_csr *A = blocks[k];
    if (k == 0 && zero_guess) {
      if (A->n_crows > 0)
        hipLaunchKernelGGL(k_gs_first_color_zero, dim3((unsigned)((A->n_crows + 255) / 256)), dim3(256), 0, c->s[0], x->d,
                           (const double *)b->d, (const double *)diag->d, A->d_crp, A->d_row_ids, (int)A->n_crows);
    } else gs_color_launch(c, A, x, b, diag);
  }
  for (int k = n_colors - 2; k >= 0; --k) gs_color_launch(c, blocks[k], x, b, diag);
  PA_HIP(hipGetLastError());
  return PA_OK;
}

// The same on a zero guess with the blocks pa_csr_select_rows_lower cuts (lower[k]: colour k's rows, only their entries in
// columns of a colour < k): the forward half reads those instead -- every other entry of a colour's rows meets an x that is
// still zero, and adding +-0.0 products to a row sum changes none of its bits -- 48 % of the entries for the 27-point
// colouring.  lower[k] may be NULL (colour 0 always; any colour: the full block is used).  A lower block must hold an
// entry for every row of its colour (greedy colouring guarantees it: a row has colour k because it has neighbours of every
// lower colour) -- checked, because a row without entries would be skipped by the launch.
extern "C" int pa_gs_color_symmetric_sweep_zero(pa_csr *const *blocks, pa_csr *const *lower, int n_colors, pa_vec *x,
                                                const pa_vec *b, const pa_vec *diag) {
  PA_TRY(gs_color_check(blocks, n_colors, x, b, diag));
  PA_REQUIRE(lower != nullptr, "lower is NULL");
  for (int k = 1; k < n_colors; ++k)
    if (lower[k]) {
      PA_REQUIRE(lower[k]->t_rows == x->n_own && lower[k]->n_cols == x->n_own + x->n_ghost && !lower[k]->next, "lower block %d does not match x", k);
      PA_REQUIRE(lower[k]->n_nonempty == blocks[k]->n_nonempty, "lower block %d misses rows of its colour (%lld of %lld)", k,
                 (long long)lower[k]->n_nonempty, (long long)blocks[k]->n_nonempty);
    }
  pa_ctx *c = x->ctx;
  PA_HIP(hipSetDevice(c->device));
  for (int k = 0; k < n_colors; ++k) {
    const pa_csr *A = blocks[k];
    if (k == 0) {
      if (A->n_crows > 0)
        hipLaunchKernelGGL(k_gs_first_color_zero, dim3((unsigned)((A->n_crows + 255) / 256)), dim3(256), 0, c->s[0], x->d,
                           (const double *)b->d, (const double *)diag->d, A->d_crp, A->d_row_ids, (int)A->n_crows);
    } else gs_color_launch(c, lower[k] ? lower[k] : A, x, b, diag);
  }
  for (int k = n_colors - 2; k >= 0; --k) gs_color_launch(c, blocks[k], x, b, diag);
  PA_HIP(hipGetLastError());
  return PA_OK;
}


// ------------------------------------------------------------------------------------------------
// deterministic scatter-add maps (sparse_matrix!(A,V,K), src/sparse_utils.jl:454-466)
// ------------------------------------------------------------------------------------------------
extern "C" int pa_scatter_create(pa_ctx *c, int64_t n_dst, int64_t n_src, const int32_t *dest, int index_base, pa_scatter **out) {
  PA_REQUIRE(c && out && n_dst >= 0 && n_src >= 0 && (n_src == 0 || dest), "bad arguments");
  PA_REQUIRE(index_base == 0 || index_base == 1, "index_base must be 0 or 1");
  if (n_src >= (1 << 16) && !(getenv("PA_SETUP_DEVICE") && atoi(getenv("PA_SETUP_DEVICE")) == 0)) {
    // the stable grouping by destination as a radix sort on the device (pa_assemble.hip): same lists
    PA_HIP(hipSetDevice(c->device));
    int32_t *d = nullptr;
    PA_HIP(hipMalloc((void **)&d, sizeof(int32_t) * (size_t)n_src));
    std::vector<int32_t> z;
    const int32_t *src = dest;
    if (index_base) { z.resize(n_src); for (int64_t p = 0; p < n_src; ++p) z[p] = dest[p] - 1; src = z.data(); }
    int st = hipMemcpy(d, src, sizeof(int32_t) * (size_t)n_src, hipMemcpyHostToDevice) == hipSuccess ? PA_OK : PA_ERR_HIP;
    if (st == PA_OK) st = pa_scatter_from_device_dest(c, n_dst, n_src, d, out);
    (void)hipFree(d);
    return st;
  }
  std::vector<int32_t> order;
  order.reserve(n_src);
  for (int64_t p = 0; p < n_src; ++p) {
    const int64_t k = (int64_t)dest[p] - index_base;
    if (k < 0) continue;  // `if k < 1 continue` (src/sparse_utils.jl:461)
    PA_REQUIRE(k < n_dst, "destination %lld out of range at source %lld", (long long)k, (long long)p);
    order.push_back((int32_t)p);
  }
  std::stable_sort(order.begin(), order.end(), [&](int32_t a, int32_t b) { return dest[a] < dest[b]; });
  std::vector<int32_t> tgt, tptr;
  tptr.push_back(0);
  for (size_t k = 0; k < order.size(); ++k)
    if (k == 0 || dest[order[k]] != dest[order[k - 1]]) {
      if (k) tptr.push_back((int32_t)k);
      tgt.push_back(dest[order[k]] - index_base);
    }
  if (!order.empty()) tptr.push_back((int32_t)order.size());
  pa_scatter *s = new pa_scatter();
  s->ctx = c; s->n_dst = n_dst; s->n_src = n_src; s->n_tgt = (int64_t)tgt.size();
  PA_HIP(hipSetDevice(c->device));
  PA_TRY(upload_i32(tgt, &s->d_tgt));
  PA_TRY(upload_i32(tptr, &s->d_tptr));
  PA_TRY(upload_i32(order, &s->d_tp));
  *out = s;
  return PA_OK;
}

extern "C" int pa_scatter_destroy(pa_scatter *s) {
  if (!s) return PA_OK;
  (void)hipSetDevice(s->ctx->device);
  (void)hipStreamSynchronize(s->ctx->s[0]);
  (void)pa_raw_free(s->d_tgt);
  (void)pa_raw_free(s->d_tptr);
  (void)pa_raw_free(s->d_tp);
  delete s;
  return PA_OK;
}

extern "C" int pa_scatter_add(pa_scatter *s, pa_vec *dst, const pa_vec *src, int zero_first) {
  PA_REQUIRE(s && dst && src, "bad arguments");
  PA_REQUIRE(dst->n_own + dst->n_ghost == s->n_dst && src->n_own + src->n_ghost == s->n_src, "vector sizes do not match the map");
  pa_ctx *c = s->ctx;
  PA_HIP(hipSetDevice(c->device));
  if (zero_first && s->n_dst) PA_HIP(hipMemsetAsync(dst->d, 0, sizeof(double) * s->n_dst, c->s[0]));
  if (s->n_tgt)
    hipLaunchKernelGGL(k_unpack_add, dim3((s->n_tgt + 255) / 256), dim3(256), 0, c->s[0], dst->d, src->d, s->d_tgt, s->d_tptr,
                       s->d_tp, (int)s->n_tgt);
  PA_HIP(hipGetLastError());
  return PA_OK;
}
