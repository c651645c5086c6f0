// pa_internal.h -- private definitions shared by the translation units of libpa_hip.so.
#ifndef PA_INTERNAL_H
#define PA_INTERNAL_H

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <mutex>
#include <vector>

#include "pa_hip.h"
#include "pa_hip_experimental.h"

void pa_set_err(const char *fmt, ...);

#define PA_HIP(call)                                                                          \
  do {                                                                                        \
    hipError_t pa_e_ = (call);                                                                \
    if (pa_e_ != hipSuccess) {                                                                \
      pa_set_err("%s failed: %s (%s:%d)", #call, hipGetErrorString(pa_e_), __FILE__, __LINE__); \
      return PA_ERR_HIP;                                                                      \
    }                                                                                         \
  } while (0)

#define PA_REQUIRE(cond, ...)     \
  do {                            \
    if (!(cond)) {                \
      pa_set_err(__VA_ARGS__);    \
      return PA_ERR_ARG;          \
    }                             \
  } while (0)

#define PA_TRY(call)              \
  do {                            \
    int pa_s_ = (call);           \
    if (pa_s_ != PA_OK) return pa_s_; \
  } while (0)

struct pa_arena;
#define PA_FUSED_GAVE_UP 1001   /* internal (pa_mul_fused_rccl -> pa_mul5): an earlier product timed out, take the separate launches */

// Switches of the product path, read from the environment ONCE per context (pa_ctx_create) and again on request
// (pa_ctx_reload_env: the tests flip them inside one process) -- no getenv on the path of a product.
struct pa_switches {
  int push = 1;               // PA_PUSH: pa_mul_all packs and delivers all parts with one push launch (0: pack per part + copies)
  int graph_one_stream = 1;   // PA_GRAPH_ONE_STREAM: inside a capture pa_mul_all queues ONE chain on the compute stream
  int ghost_from_buffer = 1;  // PA_MUL_GHOST_FROM_BUFFER: own x ghost reads consistent!'s receive buffer (the renamed twin)
  int mul_fused = 1;          // PA_MUL_FUSED: mul!(c,a,b) of a part as one launch (pa_fused.hip)
  int mul_fused_rccl = 0;     // PA_MUL_FUSED_RCCL: also over RCCL (the launch's tail acquires a flag the comm stream raises behind the receives).
                              //   OFF by default since round 6: on one GPU over a 1-rank communicator one product in ~10 waited out its whole
                              //   time-out for a flag whose raising depends on the comm stream's progress THROUGH the host-side runtime (RCCL's
                              //   launch bookkeeping), and the one launch was slower than the separate launches anyway (notebook R6.2)
  int fused_tail_blocks = 1024;  // PA_FUSED_TAIL_BLOCKS: tail blocks of a fused launch that may SPIN on arrival flags (ranks sharing one GPU: keep it small)
  int spmv_alternate = 1;     // PA_SPMV_ALTERNATE: every other product of a block walks its chunks backwards
  int vd_select = 1;          // PA_SPMV_VDICT_SELECT: a dictionary of at most two values is decoded by a select, not through the lane dictionary
  int pell = 1;               // PA_SPMV_PELL: blocks that have pattern-ELL storage run on k_spmv_pell (0: the row-split kernel; also read at block creation)
  int pell_bytes = 1;         // PA_SPMV_PELL_BYTES: pattern blocks whose dictionary has 3 .. 64 values run on pattern-ELL's one-byte stream (0: the row-split kernel's; also read at block creation)
  int pell_lean = 1;          // PA_SPMV_PELL_LEAN: slabs of a class run the instruction-lean form (scalar base, lane ballots, scalar bits; pa_pell.h); 0: the masked form everywhere
  int test_skip_raise = 0;    // PA_TEST_FUSED_SKIP_RAISE=k (tests only): the k-th fused product over RCCL never gets its flag raised -> its tail times out
  int chain_fused = 1;        // PA_SPMV_CHAIN_FUSED: a column-split chain is built for, and run as, one launch (k_spmv_xring_chain)
};

struct pa_ctx {
  pa_switches sw;
  int64_t n_chain_fused = 0;                    // column-split chains run as one launch so far
  int64_t n_fused = 0, n_fused_exchange = 0;    // fused product launches so far / of those, with the exchange inside the launch
  // A fused product over RCCL whose tail gave up waiting for its flag (PA_IPC_TIMEOUT_S) leaves boundary rows unsummed.  The status
  // words of the context's plans are looked at by pa_ctx_sync -- the point where a host could read the result -- which reports the
  // time-out ONCE (PA_ERR_STATE); the handle itself goes on with separate launches (round 4's chain) from its next product on.
  std::vector<int *> fused_status;
  int64_t n_fused_timeouts = 0;
  bool keep_coo_slots = false;       // pa_coo_keep_input_slots: the assemblies remember where their input triplets went
  int device = 0;
  hipStream_t s[2] = {nullptr, nullptr};  // [0] compute, [1] comm
  hipEvent_t ev_compute = nullptr;
  int cus = 0, xcds = 8;
  size_t hbm = 0;
  char name[128] = {0};
  double *d_partials = nullptr;
  int n_partials = 0;
  double *d_scalar = nullptr;
  double *d_dotpart = nullptr;            // per-chunk partial sums of a fused product + dot (pa_mul_dot)
  int64_t n_dotpart = 0;
  void *d_vdict_scratch = nullptr;        // hash table, slot codes and counter of the value-dictionary build (vdict_build)
  double *d_xalpha[2] = {nullptr, nullptr};   // x .* alpha of a product on a CSC-made block (pa_spmv), one per stream, grown on demand
  int64_t n_xalpha[2] = {0, 0};
  bool capturing = false;                 // a pa_graph_begin is open on the compute stream
  bool keep_raw_columns = false;          // pa_ctx_keep_raw_columns: blocks created now keep their Int32 columns in HBM (pa_rowsel.hip)
  int comm_priority = 0;                  // priority the comm stream was created with (the device's greatest)
  pa_arena *arena = nullptr;              // contiguous HBM extents with their memory-class maps (pa_arena.hip), on demand
  bool arena_tried = false;
  std::mutex mem_mu;                      // guards the arena's maps: create / destroy may come from any host thread
};

// Device memory of a context (pa_arena.hip).  kind says what the buffer is FOR, which decides its memory class:
// matrix streams and vectors never share one (a product's write stream must not sit in its read stream's class).
#define PA_MEM_PLAIN 0    /* hipMalloc */
#define PA_MEM_MATRIX 1   /* streamed by the product kernels: values, columns, row pointers, descriptors */
#define PA_MEM_VECTOR 2   /* local values of a PVector */
int pa_dev_alloc(pa_ctx *c, void **p, size_t bytes, int kind);
void pa_dev_free(pa_ctx *c, void *p);
int pa_dev_alloc_at(pa_ctx *c, void **p, size_t bytes, int cls);   // memory class cls of the held extents, or plain (cls < 0); *p = NULL: no room
// hipMalloc / hipFree of the library's small buffers.  PA_DEBUG_GUARD=1 (a debugging aid, pa_arena.hip): EVERY device buffer
// then ends (within 16 bytes) at the end of its own mapping with unmapped address space behind it, so that a load or store
// past the end of a buffer faults instead of landing in a neighbour; =2 also fills every new buffer with 0xFF bytes, so that
// a buffer the library forgets to initialise reads as NaNs.  Nothing is freed in this mode.  The fuzzers of tests/fuzz run
// under it.
hipError_t pa_raw_malloc_impl(void **p, size_t bytes);
hipError_t pa_raw_free(void *p);
template <class T>
inline hipError_t pa_raw_malloc(T **p, size_t bytes) { return pa_raw_malloc_impl((void **)p, bytes); }
// Host-to-device copy of set-up data, complete ON THE DEVICE when it returns.  The library's streams are non-blocking: nothing
// orders a kernel on them behind a null-stream copy, and a synchronous hipMemcpy from pageable memory may return once the
// data sits in the staging buffer -- with every new buffer poisoned (PA_DEBUG_GUARD=2) the first product after a block's
// creation then read the poison in the tail of the value stream or in the chunk table.
inline hipError_t pa_h2d(void *dst, const void *src, size_t bytes) {
  hipError_t e = hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
  return e != hipSuccess ? e : hipStreamSynchronize(nullptr);
}
#define PA_MEM_CLASS_PLAIN_VERIFIED 9   /* a plain allocation the pair check found clear of the matrix streams' class */
int pa_mem_class(const pa_ctx *c, const void *p);    // 0..2, PA_MEM_CLASS_PLAIN_VERIFIED, or -1 (outside, unknown)
void pa_arena_destroy(pa_ctx *c);

struct pa_event {
  pa_ctx *ctx = nullptr;
  hipEvent_t ev = nullptr;
};

struct pa_vec {
  pa_ctx *ctx = nullptr;
  double *d = nullptr;
  int64_t n_own = 0, n_ghost = 0;
  bool owned = true;
};

struct pa_pell;      // pattern-ELL storage of a slab (pa_pell.hip), or none

struct pa_csr {
  pa_ctx *ctx = nullptr;
  pa_pell *pell = nullptr;         // second storage of a pattern block for the lane-per-row kernel k_spmv_pell (pa_pell.h); NULL: none
  int64_t n_rows = 0, n_cols = 0, nnz = 0;
  int64_t n_crows = 0, n_chunks = 0, n_nonempty = 0, n_long = 0;
  bool compact = false;
  bool alpha_inside = false;       // made from CSC storage (pa_csr_create_from_csc): mul!(y,A,x,alpha,beta) forms a*(x*alpha) as
                                   // SparseArrays' CSC method does, not (a*x)*alpha (SparseMatricesCSR's): see pa_spmv
  int32_t *d_crp = nullptr;        // (compacted) row pointer, 0-based
  int32_t *d_col = nullptr;        // 0-based columns, padded
  int32_t *d_raw_col = nullptr;    // every stored entry's column (kept on request while d_col holds a compacted stream:
                                   // pa_ctx_keep_raw_columns; what pa_csr_select_rows reads), or NULL
  double *d_val = nullptr;         // padded
  int32_t *d_chunk_row = nullptr;  // n_chunks+1 row boundaries of the row split
  int32_t *d_chunk_rp = nullptr;   // the same interleaved with the rows' pointers, {chunk_row[k], crp[chunk_row[k]]}: what k_spmv_rowsplit reads
  int32_t *d_row_ids = nullptr;    // compacted row -> row, or NULL
  bool pad_products = false;       // no row patterns and most rows hold a multiple of 8 entries: padded product slots (PADP)
  bool use_c16 = false;            // 16-bit windowed column stream present
  int64_t n_c16_fallback = 0;      // chunks that keep 32-bit columns
  int64_t n_c16_chunks = 0, n_c32_chunks = 0;   // chunks by column encoding (with n_pattern_chunks: all of them)
  int64_t n_pdelta = 0;
  int64_t nnz_c16 = 0, nnz_c32 = 0;             // stored entries whose chunk reads the 16-bit stream / 32-bit columns
  int64_t n_col32 = 0, n_col16 = 0;             // entries held in d_col / d_col16 (compacted when the block has row patterns)
  uint16_t *d_col16 = nullptr;     // (slot << 12) | (col & 4095), padded
  int32_t *d_win = nullptr;        // n_chunks * 16 window bases; [c*16] < 0 => 32-bit chunk
  bool use_pattern = false;        // row-pattern descriptors present
  int64_t n_pattern_chunks = 0;    // chunks whose columns are recomputed from a pattern
  int32_t *d_pdesc = nullptr;      // n_chunks * PA_PDESC_INTS descriptor ints; first of a chunk = #segments or 0
  int32_t *d_pdelta = nullptr;     // 32 deltas per pattern
  bool use_vdict = false;          // value dictionary present and current (dropped when the values are updated)
  bool vdict_stale = false;        // the values changed under the codes: fp64 stream until vdict_maintain renews them
  bool vdict_dead = false;         // updated values overflowed the dictionary: fp64 stream for good
  int vdict_products = 0;          // products served since the values changed
  bool vd_captured = false;        // a product of this slab on the one-byte stream has been recorded into a hipGraph: the codes
                                   // must follow every value update AT ONCE (the replay reads them), not eight products later
  bool vd_captured_two = false;    // ... and through the kernel that decodes a dictionary of at most TWO values by a select (VD = 2)
  uint64_t val_epoch = 0;          // (head) bumped by every value update: what derived blocks (pa_matrix::oh_rb) compare with
  int n_dict = 0;
  bool dict_finite = true;         // every dictionary value is finite (pattern-ELL's lean form on the one-byte stream needs it)
  uint8_t *d_code = nullptr;       // one byte per stored entry (padded)
  double *d_dict = nullptr;        // PA_VDICT_MAX values
  // x-window launch (pa_spmv_xwin.h): groups of consecutive 16-bit chunks whose x span is staged in LDS; the other chunks
  // of the block stay on k_spmv_rowsplit through d_xw_rest.  n_xw_groups = 0: the block does not use it.
  int64_t n_xw_groups = 0, n_xw_rest = 0, n_xw_chunks = 0, xw_staged = 0;   // n_xw_groups: both tiers
  int64_t n_xw_tier[3] = {0, 0, 0};       // d_xw_grp = [40 KiB-window groups..., 96 KiB..., 128 KiB...]
  int64_t n_xw_ring = 0;           // ring groups (k_spmv_xring), behind the three tiers in d_xw_grp
  uint64_t n_launched = 0;         // products launched on this slab (k_spmv_rowsplit walks its chunks backwards on every other one)
  int32_t *d_chunk_cmax = nullptr; // per chunk: its highest column (what a ring group's rounds load up to)
  int32_t *d_chunk_p = nullptr;    // n_chunks+1: crp[chunk_row[c]]
  void *d_xw_grp = nullptr;        // n_xw_groups x {first chunk, chunks, first column, columns}
  int32_t *d_xw_rest = nullptr;
  // A block with 2^31 stored entries or more is a chain of row slabs, each a complete pa_csr with Int32 offsets of its
  // own: this node holds rows [row0, row0 + n_rows) and the entries [nnz0, nnz0 + nnz) of the block.  The head also
  // carries the block's totals.
  pa_csr *next = nullptr;
  int64_t row0 = 0, nnz0 = 0;
  // Round 4: a COLUMN-SPLIT chain (csr_colsplit, pa_transpose.hip) -- every node holds ALL rows of the block and the entries whose
  // columns fall into its band of the diagonal; node j > 0 accumulates (beta = 1) onto what the nodes before it left in y.  Columns
  // ascend inside a row, so the pieces' entries are consecutive runs of the row's sum: the same additions in the same order.
  bool accumulate = false;         // this node adds onto y whatever beta the caller passed (a column piece behind the first)
  bool colsplit = false;           // (head) the chain is a column split, not row slabs
  int32_t *d_src = nullptr;        // column split: original entry index of every stored entry (value updates, downloads)
  int64_t xw_max_span = 0;         // widest column span of a 16-bit chunk (what decides a column split)
  void *d_chain = nullptr;         // (head of a column split) pa_chain_piece per piece: the pieces' ring groups cover the same rows, the
  int64_t chain_groups = 0;        // chain runs as ONE launch of chain_groups workgroups (k_spmv_xring_chain, pa_spmv_xwin.h)
  int chain_pieces = 0;
  int64_t t_rows = 0, t_nnz = 0;
};

// pa_pell.hip
int pa_pell_build(pa_csr *A);                       // at the end of a slab's creation; never an error (a block that does not qualify has none)
void pa_pell_free(pa_csr *A);
int pa_pell_after_update(pa_csr *A);                // behind a value update: the fp64 stream follows in place
int pa_pell_bits_refresh(pa_csr *A);                // behind a renewal of the value dictionary: one bit per entry again (<= 2 values)
int pa_pell_mode(const pa_csr *A);                  // 0: the row-split kernel serves; 1: pattern-ELL fp64 stream; 2: one bit per entry; 3: one byte per entry
int pa_pell_launch(const pa_csr *A, int mode, int epi, const double *x, double *y, double alpha, double beta, double *gs_x,
                   const double *gs_b, const double *gs_diag, hipStream_t st);
int64_t pa_pell_partials(const pa_csr *A);          // EPI 3 writes one partial sum per slab
int64_t pa_pell_stream_bytes(const pa_csr *A, int mode);

struct pa_push_table;   // device tables of the push transport for a group of plans in one process (pa_push.hip)
struct pa_ipc_link;     // the neighbours' receive buffers and flags mapped over hipIpc, one part per process (pa_push.hip)

// local values of a PVector{Vector{Float32}}, [own | ghost] (csrc/pa_f32.hip)
struct pa_vec32 {
  pa_ctx *ctx = nullptr;
  float *d = nullptr;
  int64_t n_own = 0, n_ghost = 0;
};

struct pa_plan {
  struct side {
    std::vector<int32_t> nbr;   // 0-based part ids
    std::vector<int32_t> ptrs;  // 0-based offsets, n+1
    std::vector<int32_t> idx;   // 0-based local ids
    int64_t n = 0;
    int32_t *d_idx = nullptr;
    double *d_buf = nullptr;    // JaggedArray.data of buffer_snd (snd side) / buffer_rcv (rcv side)
  };
  pa_ctx *ctx = nullptr;
  int32_t part = 0;  // 0-based
  uint64_t serial = 0;            // unique per plan ever made in this process (a cached push table is keyed on it, not on the address)
  int64_t n_local = 0;
  side snd, rcv;     // assembly orientation (src/p_vector.jl:418-426)
  int64_t n_tgt = 0;
  int32_t *d_tgt = nullptr, *d_tptr = nullptr, *d_tp = nullptr;
  hipEvent_t ev_packed = nullptr, ev_arrived = nullptr;
  int phase = 0;     // 0 idle, 1 packed, 2 arrived
  int elem = 8;      // bytes per value of the payload in the buffers: 8, or 4 between pa_exchange_pack32 / _pack_raw and their finish
  int raw_dtype = -1; // >= 0 between pa_exchange_pack_raw and pa_exchange_finish_raw: the payload's dtype
  bool own_comm_stream = false;   // this exchange's transport ran on this part's comm stream alone (RCCL: one part per process)
  int mode = 0;
  hipEvent_t ev_wait = nullptr;   // what wait(t) waits for: ev_arrived, or the event a group launch recorded once for all its parts
  pa_push_table *push[2] = {nullptr, nullptr};   // plans[0] of a group caches the group's tables here, per mode
  pa_ipc_link *ipc = nullptr;     // pa_plan_ipc_connect
  unsigned long long seq[2] = {0, 0};            // exchanges started so far, per mode (the push transport's sequence numbers)
  bool bufs_in_ipc_region = false; // snd.d_buf / rcv.d_buf point into the ipc link's region (freed with it)
  bool ipc_ack_due = false;       // this exchange arrived over the ipc link: pa_exchange_finish acknowledges it to the senders
  // the fused product over RCCL (pa_fused.hip): a flag word (+ a zero index) the comm stream raises behind the receives and the
  // launch's tail acquires, its sequence number, and a host-visible status word for the tail's time-out
  unsigned long long *d_rflag = nullptr;
  unsigned long long rseq = 0;
  int *h_rstatus = nullptr;
};

bool pa_plan_ipc_connected(const pa_plan *p);
int pa_exchange_start(pa_plan *p, pa_comm *comm, pa_vec *v, int mode);   // pack + transport of one part (RCCL / ipc / none)
int pa_exchange_finish_all_insert(pa_plan *const *plans, int32_t n_parts, pa_vec *const *v, int on_comm_stream);   // after pa_exchange_push_local(CONSISTENT)
int pa_exchange_push_local_one_stream(pa_plan *const *plans, int32_t n_parts, pa_vec *const *v, int mode);
int pa_exchange_join_all(pa_plan *const *plans, int32_t n_parts);
void pa_push_release(pa_plan *p);                // frees what pa_push.hip hung on a plan
int pa_ipc_ack(pa_plan *p, int mode);            // (compute stream) tell the senders of the exchange just consumed that the buffer is free

struct pa_scatter {
  pa_ctx *ctx = nullptr;
  int64_t n_dst = 0, n_src = 0, n_tgt = 0;
  int32_t *d_tgt = nullptr, *d_tptr = nullptr, *d_tp = nullptr;
};

struct pa_gs {
  pa_ctx *ctx = nullptr;
  int64_t n_own = 0, n_local = 0, nnz = 0, max_level_rows = 0;
  int32_t *d_rowptr = nullptr, *d_col = nullptr, *d_rows = nullptr;  // 0-based CSR; rows sorted by level
  double *d_val = nullptr, *d_diag = nullptr;
  std::vector<int32_t> lev_ptr;  // host: level l owns d_rows[lev_ptr[l] .. lev_ptr[l+1])
  struct graph_entry {             // one captured sweep (a chain of level kernels) per (x, b, direction, zero_guess)
    const void *x, *b;
    int backward, zero_guess;
    hipGraphExec_t exec;
  };
  std::vector<graph_entry> graphs;
};

struct pa_rowset {
  pa_ctx *ctx = nullptr;
  int64_t n = 0;
  int32_t *d_rows = nullptr;
};

struct pa_transfer {
  pa_ctx *ctx = nullptr;
  int64_t n_coarse = 0;
  int32_t *d_f2c = nullptr;  // 0-based fine row of every coarse row
  const pa_csr *rows = nullptr;  // optional: the stored entries of exactly those fine rows (fused residual + restrict)
};

struct pa_matrix {
  pa_ctx *ctx = nullptr;
  const pa_csr *oo = nullptr, *oh = nullptr;   // own_own, own_ghost (not owned)
  pa_plan *plan = nullptr;                     // exchange plan of the column partition (not owned)
  pa_csr *oh_rb = nullptr;                     // own_ghost with its columns renamed to positions of consistent!'s RECEIVE BUFFER (owned;
  bool rb_tried = false;                       //   built at the first product): own x ghost then needs no unpack before it
  uint64_t rb_epoch = 0;                       // oh->val_epoch the twin's values were taken at
  // the fused launch (pa_fused.hip): the boundary rows' block {own_own entries, own_ghost entries on buffer positions} and the bitmap
  // of the boundary rows, built at the first product, rebuilt when either block's values change
  pa_csr *bd = nullptr;
  unsigned *d_rowmask = nullptr;
  int64_t n_bd_rows = 0;
  uint64_t bd_epoch_oo = 0, bd_epoch_oh = 0;
  int32_t *d_bd_hrow = nullptr;                // bd's i-th row sits at this stored row of the twin (what an in-place value refresh reads)
  bool bd_captured = false;                    // a fused launch of this handle sits in a recorded hipGraph: bd and the twin keep their
                                               //   addresses and follow every value update AT the update (pa_csr_values_changed)
  bool fuse_tried = false;
  bool fused_off = false;                      // a fused product of this handle timed out waiting for its exchange: separate launches from now on
  bool transposed = false;                     // pa_matrix_create_transposed: oo = A_oo', oh = A_oh' (pa_csr_create_transpose), for pa_mul5_transpose
};

struct pa_graph {
  pa_ctx *ctx = nullptr;
  hipGraphExec_t exec = nullptr;
};

int pa_plan_mark_arrived(pa_plan *p);

// What the fused product launch does FOR THE EXCHANGE besides the product (pa_fused.hip; everything optional):
struct pa_push_seg;
struct pa_fused_comm {
  // its first blocks pack and push this part's send list over the ipc link before they take chunks (pa_push_ipc_block)
  int p_n = 0, p_nseg = 0;
  const int32_t *p_idx = nullptr;
  const pa_push_seg *p_segs = nullptr;
  unsigned *p_done = nullptr;
  // its tail acquires the arrival of b's ghost values: flags[wait_idx[i]] >= seq for every i, or gives up after `ticks`
  const unsigned long long *flags = nullptr;
  const int32_t *wait_idx = nullptr;
  int n_wait = 0;
  long long ticks = 0;
  int *status = nullptr;
  // its tail unpacks the receive buffer into b's ghost entries (the rest of consistent!, src/p_vector.jl:603-611) ...
  const int32_t *u_idx = nullptr;
  int u_n = 0;
  // ... and its last tail block tells the senders that the buffer is free again
  unsigned long long *const *ack_dst = nullptr;
  int n_ack = 0;
  unsigned *t_done = nullptr;
};

// pa_fused.hip
int pa_matrix_fused_build(pa_matrix *m);
int pa_matrix_fused_refresh(pa_matrix *m);       // bd's values again from the two blocks, in place (same pattern, same addresses)
// Handles whose derived blocks (the twin of own_ghost, the boundary rows' block) copy a block's VALUES are registered per block;
// a value update of the block (vdict_after_update, pa_csr.hip) makes them follow at once, in place -- ADVICE r05: a recorded graph
// replayed after an update summed boundary rows from the old values, and the next eager product freed what the graph still read.
void pa_watch_add(const pa_csr *A, pa_matrix *m);
void pa_watch_drop_matrix(pa_matrix *m);
void pa_watch_drop_csr(const pa_csr *A);
int pa_csr_values_changed(const pa_csr *A);
bool pa_matrix_fused_ready(const pa_matrix *m);
void pa_matrix_fused_release(pa_matrix *m);
// comm: DEVICE copy of what the launch does for the exchange (or NULL), seq: this exchange's sequence number, n_push_blocks: its
// first blocks push, max_tail_blocks > 0: at most so many tail blocks (they may spin: never fill the GPU with them)
int pa_mul_fused_launch(pa_matrix *m, pa_vec *c, pa_vec *b, double alpha, double beta, hipStream_t st, const pa_fused_comm *comm = nullptr,
                        unsigned long long seq = 0, int n_push_blocks = 0, int max_tail_blocks = 0);
int pa_mul_fused_ipc(pa_matrix *m, pa_vec *c, pa_vec *b, double alpha, double beta);   // pa_push.hip: one part per process, ONE launch
bool pa_fused_ipc_fits(const pa_matrix *m);
int pa_mul_fused_rccl(pa_matrix *m, pa_comm *comm, pa_vec *c, pa_vec *b, double alpha, double beta);   // pa_fused.hip
void pa_fused_plan_release(pa_plan *p);
void pa_csr_before_product(const pa_csr *A);     // pa_csr.hip: upkeep a product does first (the value dictionary's renewal)
// pa_push.hip: consistent!(v) of every part of this process COMPLETE with one launch on the compute stream -- the push kernel also
// stores every delivered value into the receiving part's ghost entry (the unpack of src/p_vector.jl:603-611)
int pa_exchange_push_unpack_one_stream(pa_plan *const *plans, int32_t n_parts, pa_vec *const *v);

// ---- shared between the device units (pa_device.hip / pa_csr.hip / pa_mg.hip / pa_plan.hip / pa_fused.hip) ----------------------
#define PA_SLOT_OK(s) ((s) >= 0 && (s) < PA_N_SLOTS)
#define PA_COEF_OK(s) ((s) >= -1 && (s) < PA_N_SLOTS)
extern thread_local int pa_tls_plain_encoding;   // pa_device.hip: > 0 while pa_matrix_fused_build makes its block (Int32 columns, nothing else)
// which decode of the one-byte value stream a launch on this block takes (VD = 2: at most two values, a select per entry)
static inline bool pa_vd_two(const pa_csr *S) { return S->use_vdict && S->n_dict <= 2 && S->ctx->sw.vd_select; }
// pa_csr.hip: the product on a stream of the caller's choice; the x-window launches of one slab (u != NULL: with the fused dot)
int pa_spmv_on(const pa_csr *A, const pa_vec *x, int xseg, pa_vec *y, int yseg, double alpha, double beta, hipStream_t st);
void pa_launch_xwin(const pa_csr *S, const double *xs, double *ys, double alpha, double kbeta, const double *u, double *partial,
                    hipStream_t st = nullptr);

#endif
