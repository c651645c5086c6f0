/*
 * pa_hip_experimental.h -- everything libpa_hip.so exports BESIDE the contract of pa_hip.h: measurement and introspection
 * (memory classes / arena, encodings, byte counts), the placement A/B, testing aids (debug arrays, host-side checkers),
 * the HPCG multigrid set-up and smoothers (a caller of the hot path), the SELL-C-sigma parity format, and
 * the native host twins of the reference's set-up loops (pa_host_*: CPU code, no GPU needed) that the Python host mirror and
 * the tests call.  Same conventions as pa_hip.h (statuses, 1-based host arrays where the reference stores them so).
 *
 * STATUS: experimental -- names and signatures may change between rounds; a binding that only needs
 * mul! / consistent! / assemble! / psparse needs nothing from this file.
 */
#ifndef PA_HIP_EXPERIMENTAL_H
#define PA_HIP_EXPERIMENTAL_H

#include "pa_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct pa_gs pa_gs;       /* the level-scheduled Gauss-Seidel smoother of one part (below)  */

/* ---- where the big arrays live: the context's HBM extents and their memory-class maps (csrc/pa_arena.hip) -----------
 * Measured on MI355X: device memory falls into three classes (about a third each, physically contiguous regions of tens
 * of GiB); a product whose 64-byte write stream (y) sits in the class its read stream (the values) comes from runs
 * 13-15 % slower than with y in either other class.  A context therefore serves every buffer >= 1 MiB from physically
 * contiguous EXTENTS acquired on demand (16 GiB each, PA_ARENA_EXTENT_GIB; the first when an allocation >=
 * PA_ARENA_MIN_MIB = 256 arrives; together at most PA_ARENA_FRACTION = 0.70 of the free memory or PA_ARENA_GIB;
 * PA_ARENA=0: none) and classified with a stand-in kernel when acquired (~20-50 ms each): matrix streams
 * (pa_csr_create*) go to the class the first one landed in, vectors (pa_vec_create) to a class without matrix streams --
 * found, when none is at hand, by walking over further extents that are handed back at once (PA_ARENA_PLAIN_VECTORS=1,
 * experimental: first a plain allocation that the pair check finds clear of the matrix streams' class, class code 9).
 * Every big vector handed out is pair-checked once against the matrix streams' class (~3 ms).  An extent nothing lives
 * in is released.  Nothing is timed at the caller's expense, nothing ever moves, any failure falls back to hipMalloc.
 * pa_ctx_arena_info: bytes held, classes met (<= 3), bytes per class in the held extents, bytes in use, time spent
 * acquiring + classifying, the class matrix streams go to (-1: none yet).
 * pa_ctx_arena_map: class of every 512 MiB cell, extent after extent (-1: a boundary runs through it, -2: between two
 * extents).  pa_ctx_arena_stats: extents held, bytes acquired / released so far, peak bytes in use, vectors whose
 * (matrix stream, vector) pair passed / failed the self-check, the budget.  pa_ctx_arena_build acquires a first extent now.
 * pa_csr_memory_class / pa_vec_memory_class: class of a block's value stream / a vector's storage (-1: outside). */
int pa_ctx_arena_build(pa_ctx *ctx);
/* vector_classes = 2: a solver's vectors alternate between two memory classes of their own -- kernels that read vectors and
 * write one run ~1 % faster (MG-PCG at 256^3); costs one more walk of <= 16 GiB, once (up to a second on memory other
 * processes have used).  1 = the default. */
int pa_ctx_arena_hint(pa_ctx *ctx, int vector_classes);
/* An arena nothing lives in keeps up to PA_ARENA_SPARE_GIB (24) of extents for what the caller builds next (memory that
 * has been used is wiped by the driver when allocated again: 0.9 s per 16 GiB); this hands them back now. */
int pa_ctx_arena_release(pa_ctx *ctx);
int pa_ctx_arena_info(pa_ctx *ctx, int64_t *bytes, int *n_classes, int64_t class_bytes[3], int64_t *used, double *map_ms,
                      int *matrix_class);
int pa_ctx_arena_map(pa_ctx *ctx, int64_t *cell_bytes, int8_t *classes, int64_t capacity, int64_t *n_cells);
int pa_ctx_arena_stats(pa_ctx *ctx, int64_t *n_extents, int64_t *bytes_acquired, int64_t *bytes_released, int64_t *peak_used,
                       int64_t *pairs_ok, int64_t *pairs_failed, int64_t *budget, int64_t *plain_vector_bytes);
/* Placement A/B of a product's write stream with the product kernel itself: times y = A*x (x's segment xseg) with y where it
 * is, in every other memory class the held extents have room in and in a plain allocation (`rounds` interleaved passes of
 * `reps` launches, the minimum per place) and MOVES y's storage when another place is more than 1.5 % faster.  where[i]
 * (i < *n <= capacity): 0..2 arena class, 9 verified plain allocation, -1 plain / outside; ms[i] per launch; entry 0 = where
 * y was; *chosen = the entry y lives in afterwards.  y = A*x on return.  Not inside a graph capture. */
int pa_spmv_tune_output(const pa_csr *A, const pa_vec *x, int xseg, pa_vec *y, int reps, int rounds, int32_t capacity,
                        int32_t *where, double *ms, int32_t *n, int32_t *chosen);
/* PCI address of the context's device, e.g. "0000:75:00.0" (the key of /sys/bus/pci/devices/: clocks, power, partitions) */
int pa_ctx_pci_bus_id(pa_ctx *ctx, char *out, size_t len);
/* ---- blocks made of some rows of a part's matrix, built on the device (csrc/pa_rowsel.hip) ----------------------------
 * What the multigrid set-up of the HPCG driver takes from a level's matrix: the colours of the multicolour Gauss-Seidel
 * smoother (PartitionedSolvers/src/smoothers.jl:98-176 as SpMV + update) and the fine rows the coarse grid keeps
 * (HPCG/src/mg_preconditioner.jl:224-251,314-329).  pa_ctx_keep_raw_columns(ctx, 1): blocks created from now on keep their
 * Int32 columns in HBM next to the compacted streams the product reads (4 B per stored entry, until
 * pa_csr_drop_raw_columns).  pa_csr_select_rows: out[k] = the n_rows x (own + ghost columns) block holding the rows r with
 * mask[r] == k (host array of n_rows entries in -1..n_sel-1; -1: in no block) of own_own | own_ghost (own_ghost may be NULL;
 * its columns are shifted by own_own's column count: the unsplit order HPCG stores), entries in stored order -- the blocks
 * pa_host_color_split + pa_csr_create give, without the host copy and the second trip over PCIe.  pa_csr_diagonal:
 * d[r] = the stored (r,r) entry of the block, 0.0 when there is none. */
int pa_ctx_keep_raw_columns(pa_ctx *ctx, int on);
int pa_csr_has_raw_columns(const pa_csr *A, int *yes);
int pa_csr_drop_raw_columns(pa_csr *A);
int pa_csr_select_rows(const pa_csr *own_own, const pa_csr *own_ghost, const int32_t *mask, int32_t n_sel, pa_csr **out);
/* out[k] = the rows with mask == k restricted to their entries in own columns j with 0 <= mask[j] < k (NULL when there are
 * none: k = 0 always); n_cols = the column count the blocks get (the part's own + ghost columns) */
int pa_csr_select_rows_lower(const pa_csr *own_own, int64_t n_cols, const int32_t *mask, int32_t n_sel, pa_csr **out);
int pa_csr_diagonal(const pa_csr *own_own, pa_vec *d);
/* pa_gs_create (below) for the sequential ordering from the part's blocks in HBM: the unsplit CSR, the diagonal and the
 * dependency levels of the sweep (PartitionedSolvers/src/smoothers.jl:144-160) computed on the device, verified against
 * their definition entry by entry.  PA_ERR_ARG when the own|own pattern is not structurally symmetric or a diagonal
 * entry is missing (pa_gs_create's own conditions). */
int pa_gs_create_from_blocks(const pa_csr *own_own, const pa_csr *own_ghost, int ordering, pa_gs **out);
/* pa_host_greedy_coloring (below) of the own_own block in HBM: color[r] (host, n_rows entries) = the smallest colour no own
 * neighbour j < r has, computed by rounds on the device and verified against that definition; PA_ERR_ARG when the pattern
 * is not structurally symmetric (colour on the host then). */
int pa_csr_greedy_coloring(const pa_csr *own_own, int32_t *color, int32_t *n_colors);
/* The same from the dependency levels a sequential smoother of this block already holds (pa_gs_create_from_blocks): one small launch
 * per level instead of the discovery by rounds. */
int pa_csr_greedy_coloring_by_levels(const pa_csr *own_own, const pa_gs *gs, int32_t *color, int32_t *n_colors);
/* affinity[k] = the mean number of stored entries of a colour-k row whose column is one of kept_rows (0-based own rows: the
 * fine rows a coarse grid keeps).  A multicolour smoother inside a multigrid cycle sweeps its colours in order of decreasing
 * affinity, so that the kept rows' own colour sits at the turn of the symmetric sweep, not at its end (where the residual
 * the restriction injects would be zero up to rounding). */
int pa_csr_color_affinity(const pa_csr *own_own, const int32_t *color, int32_t n_colors, const int32_t *kept_rows, int64_t n_kept,
                          double *affinity);
/* testing aid: a host copy of one of the arrays the product kernel reads (first slab of the block) -- 0 row pointers,
 * 1 32-bit columns, 2 16-bit codes, 3 windows, 4 pattern descriptors, 5 pattern table, 6 chunk table, 7 compacted row ids;
 * *bytes = the array's size, copied when capacity allows.  The set-up runs on the device (csrc/pa_setup.hip; PA_SETUP_DEVICE=0:
 * the host encoder): the tests compare the two builds of every array with this. */
int pa_csr_debug_array(const pa_csr *A, int which, void *host, int64_t capacity, int64_t *bytes);
int pa_csr_memory_class(const pa_csr *A, int *cls);
int pa_vec_memory_class(const pa_vec *v, int *cls);

/* ---- SELL-C-sigma storage with one lane per row (csrc/pa_sell.hip): a second, structurally different bit-exact SpMV ----
 * A wavefront owns a slab of 64 rows (sorted by length inside windows of `sigma` rows; sigma = 1: as they come) and every
 * lane walks ITS row's stored entries in the reference's order (spmv_csr! src/sparse_utils.jl:649-669;
 * SparseMatricesCSR.mul!(y,A,x,alpha,beta)) with the accumulator in a register: no LDS stage, no cross-lane sum, so the
 * result equals the reference's -- and pa_spmv's -- bit for bit.  Streams 12 bytes per stored entry plus the padding of
 * each slab (pa_sell_info), against 8 for a stencil block on row patterns: a parity / debugging mode and a format for short
 * irregular rows, not the product path.  Arguments as pa_csr_create / pa_spmv. */
typedef struct pa_sell pa_sell;
int pa_sell_create(pa_ctx *ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, const void *rowptr, const void *colval,
                   int index_bytes, int index_base, const double *nzval, int sigma, pa_sell **A);
int pa_sell_destroy(pa_sell *A);
int pa_sell_info(const pa_sell *A, int64_t *n_slabs, int64_t *padded_entries, int64_t *nnz);
int pa_sell_spmv(const pa_sell *A, const pa_vec *x, int x_segment, pa_vec *y, int y_segment, double alpha, double beta);

/* ---- laplacian_fem's triplets generated in HBM (csrc/pa_assemble.hip; round 6) ------------------------------------------------
 * laplacian_fem (src/gallery.jl:110-239) makes every part loop over ITS cells lo..hi (1-based cell coordinates of the (nodes + 1)^D
 * cell grid, column-major) and emit (node_i, node_j, Aref[i,j]) for the pairs of interior corners: the disassembled COO input of
 * psparse.  pa_fem_triplets_device writes those triplets, in that order, straight into HBM (Int64 ids, Float64 values: what
 * pa_host_laplacian_fem returns on the host); pa_coo_subassemble / pa_coo_assemble take device arrays as they take host arrays.
 * Aref: the 2^D x 2^D reference matrix, row-major.  The three arrays are freed with pa_triplets_free; pa_triplets_download copies n
 * 8-byte values to the host (tests, host routes). */
int pa_fem_triplets_device(pa_ctx *ctx, int32_t D, const int64_t *nodes, const int64_t *lo, const int64_t *hi, const double *Aref,
                           int64_t *count, void **dI, void **dJ, void **dV);
int pa_triplets_free(pa_ctx *ctx, void *dI, void *dJ, void *dV);
int pa_triplets_download(pa_ctx *ctx, const void *d, int64_t n, void *host);

/* ---- all parts in ONE process, one GPU each, over RCCL (csrc/pa_rccl.cpp; round 6) --------------------------------------------
 * SURVEY 8(b)'s sketch of the single-process multi-GPU mode: the parts of a DebugArray each in a context on a device of its own
 * (pa_ctx_create per device), their communicators made by ONE ncclCommInitAll (comms[p]: rank p of n, on ctxs[p]'s device; the devices
 * must be distinct), and the exchange of all parts ONE group of ncclSend / ncclRecv -- exchange! src/primitives.jl:1020-1042 /
 * exchange_impl! src/mpi_array.jl:575-614 with the parts' comm streams in the place of MPI requests.  Sequence per exchange:
 * pa_exchange_pack (or _pack32) on every part -> pa_exchange_rccl_all -> pa_exchange_finish (_finish32) on every part.  The peer copies
 * of pa_exchange_local serve the same layout without RCCL.  On a one-GPU box only n = 1 can run (tests): a run between distinct GPUs
 * has not been possible on the leases this library was built on. */
int pa_comm_create_all(pa_ctx *const *ctxs, int32_t n, pa_comm **comms);        /* comms: n slots; each freed with pa_comm_destroy */
int pa_exchange_rccl_all(pa_plan *const *plans, pa_comm *const *comms, int32_t n, int mode);

/* ---- pattern-ELL: the lane-per-row product kernel of pattern blocks (csrc/pa_pell.h, pa_pell.hip; round 6) ---------------
 * A block whose slabs of 64 consecutive stored rows each have at most 32 distinct (column - row) offsets -- stencil and structured
 * FEM operators, their row-compacted colour and restriction subsets -- gets, at creation, a second storage: per slab the ascending
 * union of offsets (a pattern table entry), per row a 32-bit mask of the offsets it has, and the values delta-major per slab (or,
 * when the block's value dictionary has at most two values, one bit per entry).  pa_spmv / pa_mul* then run it on k_spmv_pell: one
 * lane per row adds its products in stored order (spmv_csr! src/sparse_utils.jl:649-669: same bits as the row-split kernel), the
 * offsets are wave-uniform scalars, no LDS, no barrier.  PA_SPMV_PELL=0 (read per context / at block creation): the row-split kernel.
 * *mode: what a product of this block runs on NOW -- 0 row split, 1 pattern-ELL with the fp64 stream, 2 pattern-ELL with one bit per
 * entry, 3 pattern-ELL with one BYTE per entry (a value dictionary of 3 .. 64 values: the codes in pattern-ELL order, the dictionary in
 * 512 bytes of LDS per workgroup; PA_SPMV_PELL_BYTES=0: such blocks stay on the row-split kernel's one-byte stream); the other outputs
 * describe the storage (0 when the block has none).  Any output may be NULL. */
int pa_csr_pell_info(const pa_csr *A, int *mode, int64_t *n_slabs, int64_t *n_patterns, int64_t *value_slots, int *unroll);
/* Slab CLASSES and the lean form (round 6, second step; csrc/pa_pell.h pa_pell_slab_fast).  Slabs of one union whose offsets sit in
 * the same LANES and whose row ids have the same stride (1: consecutive rows; 2: every other row of a grid line -- a colour of the
 * smoother, the rows a restriction keeps) form a class; a structured grid has a few dozen.  For a slab of a class whose gathers all lie
 * inside x, a product with alpha = 1 and nothing to add to reads no row masks: which lanes have an offset is a 64-bit ballot of the
 * class (scalar), every gather starts from a scalar base (a run of three consecutive offsets is ONE gather plus wave shifts; stride 2:
 * one 16-byte gather and one shift), and on the one-bit stream the slab's bits are one scalar when all its rows carry the same word.  Same bits as the masked form
 * (PA_SPMV_PELL_LEAN=0) and as spmv_csr! src/sparse_utils.jl:649-669.  *n_classes: 0 when the block keeps plain patterns (more than
 * 4096 classes, PA_SPMV_PELL_CLASSES=0); *n_lean / *n_lean_bits: slabs the lean form serves on the fp64 stream / on the one-bit
 * stream.  Any output may be NULL. */
int pa_csr_pell_lean_info(const pa_csr *A, int64_t *n_classes, int64_t *n_lean, int64_t *n_lean_bits);

/* ---- Float32 blocks and vectors (csrc/pa_f32.hip; round 6: the first widening beyond the FP64 scope of the path) -----------
 * The reference's local loops are generic in the element type (spmv_csr! / spmv_csc! src/sparse_utils.jl:649-690) and its own test
 * runs them in Float32 (test/sparse_utils_tests.jl:72-79).  pa_vec32: local values of a PVector{Vector{Float32}}, [own | ghost];
 * pa_csr32: a SparseMatrixCSR{Bi,Float32,Ti} / SparseMatrixCSC{Float32,Ti} block; pa_spmv32: spmv!(y,A,x) / mul!(y,A,x,alpha,beta)
 * with every product and every sum rounded to Float32 in the reference's order (bit-identical to the oracle's float loops).  A block
 * whose rows follow patterns reuses the fp64 path's pattern-ELL structure with a 4-byte value stream, any other is stored SELL-64.
 * Arguments as pa_csr_create / pa_vec_* / pa_spmv.  Not yet: Float32 exchange payloads, epilogue forms, value updates. */
typedef struct pa_vec32 pa_vec32;
typedef struct pa_csr32 pa_csr32;
int pa_vec32_create(pa_ctx *ctx, int64_t n_own, int64_t n_ghost, pa_vec32 **v);
int pa_vec32_destroy(pa_vec32 *v);
int pa_vec32_upload(pa_vec32 *v, const float *host, int64_t offset, int64_t len);
int pa_vec32_download(const pa_vec32 *v, float *host, int64_t offset, int64_t len);
int pa_vec32_fill(pa_vec32 *v, int segment, float value);
int pa_vec32_data(pa_vec32 *v, void **device_ptr);      /* the [own | ghost] array in HBM (4-byte values: also the carrier of an Int32 payload) */
int pa_csr32_create(pa_ctx *ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, const void *rowptr, const void *colval, int index_bytes,
                    int index_base, const float *nzval, pa_csr32 **A);
int pa_csr32_create_from_csc(pa_ctx *ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, const void *colptr, const void *rowval,
                             int index_bytes, int index_base, const float *nzval, pa_csr32 **A);
int pa_csr32_destroy(pa_csr32 *A);
int pa_csr32_info(const pa_csr32 *A, int *on_pattern_ell, int64_t *n_slabs, int64_t *padded_entries);
int pa_spmv32(const pa_csr32 *A, const pa_vec32 *x, int x_segment, pa_vec32 *y, int y_segment, float alpha, float beta);
/* consistent! / assemble! of a PVector{Vector{Float32}} (src/p_vector.jl:747-755, 695-708; assemble_impl! :587-612 is generic in the
 * element type, exchange! in the payload, src/primitives.jl:1020-1042).  The plan is the one of the Float64 path (pa_plan_create: the
 * cache's index lists serve any element type); its buffers then hold floats at the same ELEMENT offsets (pa_plan_buffers: same
 * pointers, lengths in values of 4 bytes).  Sequence: pa_exchange_pack32 -> a pack-then-transport transport -- pa_exchange_local (all
 * parts in one process), pa_exchange_rccl (ncclFloat) or the caller's own copies between pa_plan_buffers -- -> pa_exchange_finish32
 * (insert, or + in ascending p with every sum rounded to Float32, then every ghost zeroed).  The push / ipc / fused transports carry
 * Float64 only.  Bit-identical to the reference's loops on Float32 (data movement; the assemble! sums in the reference's order). */
int pa_exchange_pack32(pa_plan *plan, const pa_vec32 *v, int mode);
int pa_exchange_finish32(pa_plan *plan, pa_vec32 *v, int mode);
/* The same for any local-values array in HBM, device layout [own | ghost]: Float64, Float32, Int32, Int64 (the reference exchanges
 * Int64 global ids and Int32 owners at set-up, src/p_range.jl:436-531, and assemble!(+) counts on integers alike).  `values`: n_local
 * values of the dtype; finish_raw: the first n_own are the own values (assemble! zeroes what is behind them).  consistent!: a copy of
 * the bits; assemble!: + in ascending p in the dtype's arithmetic, then every ghost := 0. */
#define PA_DTYPE_F64 0
#define PA_DTYPE_F32 1
#define PA_DTYPE_I32 2
#define PA_DTYPE_I64 3
int pa_exchange_pack_raw(pa_plan *plan, const void *values, int64_t n_local, int dtype, int mode);
int pa_exchange_finish_raw(pa_plan *plan, void *values, int64_t n_own, int64_t n_local, int dtype, int mode);

/* ---- Gauss-Seidel smoother and grid transfer of the HPCG multigrid preconditioner ----------- */
/* gauss_seidel_sweep! / gauss_seidel_sweep_zero! (PartitionedSolvers/src/smoothers.jl:144-160,236-259) on the
 * UNSPLIT local CSR of one part (n_own rows, n_local columns [own|ghost], as HPCG builds it: split_format=false).
 * The reference sweeps rows sequentially; here rows are grouped into dependency levels (level(i) = 1 + max level of
 * the own columns j < i of row i) and one kernel per level runs its rows in parallel.  With a structurally symmetric
 * own x own pattern (checked at creation) every row still sees exactly the values the sequential sweep gives it, and
 * each row's arithmetic is the reference's (s = b; s -= a*x[col] in stored order; s += d*x[row]; s /= d), so the
 * sweep is bit-identical to the CPU loop.  backward != 0 walks the levels in reverse (rows n:-1:1).
 *
 * ordering = PA_GS_MULTICOLOR replaces the dependency levels by the colours of a greedy colouring of the own x own
 * pattern (27-pt stencil: 8 colours instead of ~7n levels): a different -- much more parallel -- sweep order, NOT
 * the reference's arithmetic.  It is the "optimised" variant HPCG's opt_cg! hook is for (HPCG/src/opt_cg.jl,
 * HPCG/src/hpcg_benchmark.jl:60-78): it must reach the reference tolerance and is charged for its extra iterations. */
#define PA_GS_SEQUENTIAL 0
#define PA_GS_MULTICOLOR 1
int pa_gs_create(pa_ctx *ctx, int64_t n_own, int64_t n_local, int64_t nnz, const int32_t *rowptr,
                 const int32_t *colval, const double *nzval, int index_base, int ordering, pa_gs **gs);
int pa_gs_destroy(pa_gs *gs);
int pa_gs_info(const pa_gs *gs, int64_t *n_levels, int64_t *max_rows_per_level);
int pa_gs_sweep(pa_gs *gs, pa_vec *x, const pa_vec *b, int backward, int zero_guess);
/* Colours of the greedy colouring pa_gs_create(PA_GS_MULTICOLOR) uses (natural order, smallest free colour). */
int pa_host_greedy_coloring(int64_t n_own, const int32_t *rowptr, const int32_t *colval, int index_base,
                            int32_t *color, int32_t *n_colors);
/* a[i] = map[a[i]] in place (host threads): the colours renamed into the order they are swept in */
int pa_host_remap_int32(int32_t *a, int64_t n, const int32_t *map, int32_t n_map);
/* One colour of a multicolour Gauss-Seidel sweep written as SpMV + update (the optimised HPCG variant):
 *   x[row] = x[row] + (b[row] - t[row]) / diag[row];  t[row] = 0   for the listed rows, where t = A*x was accumulated
 * for these rows by pa_spmv(beta = 1) on the colour's sub-matrix into a zeroed t.  rows are local ids in `index_base`. */
typedef struct pa_rowset pa_rowset;
int pa_rowset_create(pa_ctx *ctx, int64_t n, const int32_t *rows, int index_base, pa_rowset **rs);
int pa_rowset_destroy(pa_rowset *rs);
int pa_gs_color_update(pa_rowset *rs, pa_vec *x, const pa_vec *b, pa_vec *t, const pa_vec *diag);
/* The same sweep (gauss_seidel_sweep!, PartitionedSolvers/src/smoothers.jl:144-160, rows taken colour by colour instead
 * of 1..n: HPCG's optimised variant, HPCG/src/opt_cg.jl) with the update fused into the SpMV kernel's epilogue and all
 * colours queued by one call:
 * blocks[k] holds every stored entry of colour k's own rows (n_own x n_local, e.g. from pa_csr_create on the rows of
 * that colour; empty rows are compacted away); for k ascending (backward != 0: descending)
 *   x[row] = x[row] + (b[row] - sum_p val[p]*x[col[p]]) / diag[row]      for the rows of colour k, in place.
 * Bit-identical to pa_spmv(beta = 1) into a zeroed t followed by pa_gs_color_update; a proper colouring (no stored
 * entry couples two rows of one colour, pa_host_greedy_coloring) makes the in-place update race-free. */
int pa_gs_color_sweep(pa_csr *const *blocks, int n_colors, pa_vec *x, const pa_vec *b, const pa_vec *diag,
                      int backward);
/* The symmetric sweep (forward, then backward: gauss_seidel_step, smoothers.jl:105-131) as colours 0..K-1, K-2..0: the
 * backward half does not relax colour K-1 a second time (nothing it couples to has changed; the update would add 0 up to
 * rounding).  zero_guess != 0: the caller guarantees x == 0 (own and ghost); colour 0 is then x = b / diag, the colour
 * launch's own expression with a zero row sum, without reading the block. */
int pa_gs_color_symmetric_sweep(pa_csr *const *blocks, int n_colors, pa_vec *x, const pa_vec *b, const pa_vec *diag,
                                int zero_guess);
/* The zero-guess sweep with lower[k] = colour k's rows restricted to their entries in columns of a colour < k
 * (pa_csr_select_rows_lower; NULL: the full block is used): the forward half reads only those -- every other entry meets
 * an x that is still zero, and +-0.0 products change no bit of a row sum.  Same bits as zero_guess = 1 above. */
int pa_gs_color_symmetric_sweep_zero(pa_csr *const *blocks, pa_csr *const *lower, int n_colors, pa_vec *x, const pa_vec *b,
                                     const pa_vec *diag);
/* restrict! / prolongate! (HPCG/src/mg_preconditioner.jl:224-251): f2c[i] = fine row of coarse row i.
 *   restrict  : r_c[i] = r_f[f2c[i]] - Axf[f2c[i]]          prolongate: x_f[f2c[i]] += x_c[i] */
typedef struct pa_transfer pa_transfer;
int pa_transfer_create(pa_ctx *ctx, int64_t n_coarse, const int32_t *f2c, int index_base, pa_transfer **t);
int pa_transfer_destroy(pa_transfer *t);
int pa_transfer_restrict(pa_transfer *t, pa_vec *r_c, const pa_vec *r_f, const pa_vec *Axf);
int pa_transfer_prolongate(pa_transfer *t, pa_vec *x_f, const pa_vec *x_c);
/* Fused residual + restriction: r_c[i] = r_f[f2c[i]] - (A x_f)[f2c[i]] without forming A x_f on the other fine rows.
 * `rows` is a block (n_own x n_local, pa_csr_create) holding the stored entries of exactly the fine rows f2c -- attach
 * checks that -- and must outlive the transfer's use of it.  Same row sums as pa_spmv + pa_transfer_restrict. */
int pa_transfer_attach_rows(pa_transfer *t, const pa_csr *rows);
int pa_transfer_restrict_fused(pa_transfer *t, pa_vec *r_c, const pa_vec *r_f, const pa_vec *x_f);

/* ---- host-side set-up helpers (native twins of the reference's set-up loops) ----------------- */
/* All ids 1-based Int64/Int32 exactly as the reference stores them. */
/* HPCG/src/sparse_matrix.jl:27-80 build_matrix: COO stream (row,col,val) + b + row ids. Returns nnz
 * through *nnz_out; pass NULL arrays to only count. */
int pa_host_hpcg_build_matrix(int64_t nx, int64_t ny, int64_t nz, int64_t gnx, int64_t gny, int64_t gnz,
                              int64_t gix0, int64_t giy0, int64_t giz0, int64_t *I, int64_t *J, double *V,
                              double *b, int64_t *row_b, int64_t *nnz_out);
/* src/gallery.jl:12-86 laplacian_fdm `setup` for one part's own box [lo,hi] per dimension (D<=3). */
int pa_host_laplacian_fdm(int32_t D, const int64_t *nodes_per_dir, const int64_t *lo, const int64_t *hi,
                          int64_t *I, int64_t *J, double *V, int64_t *nnz_out);
/* src/gallery.jl:110-239 laplacian_fem `setup` for one part's CELL box [lo,hi] of the (nodes+1)^D cell grid (D<=3): the disassembled
 * COO of the part's cells, cells column-major, local node i then j; Aref = the 2^D x 2^D reference matrix (row-major) the caller
 * computes as :123-162 does.  NULL arrays: only count. */
int pa_host_laplacian_fem(int32_t D, const int64_t *nodes_per_dir, const int64_t *cell_lo, const int64_t *cell_hi, const double *Aref,
                          int64_t *I, int64_t *J, double *V, int64_t *nnz_out);
/* src/p_range.jl:1609-1619 find_owner for block partitions: starts[d] has np[d]+1 entries. */
int pa_host_find_owner_block(int32_t D, const int64_t *n, const int32_t *np, const int64_t *const *starts,
                             const int64_t *gids, int64_t count, int32_t *owners);
/* src/p_range.jl:205-241 filter_ghost: unseen non-own gids in first-seen order. out arrays sized by
 * a first call with out_gids == NULL (returns the count in *n_new). */
int pa_host_filter_ghost(int32_t part, const int64_t *gids, const int32_t *owners, int64_t count,
                         const int64_t *known_ghost_gids, int64_t n_known, int64_t *out_gids,
                         int32_t *out_owners, int64_t *n_new);
/* map_global_to_local! for a block partition: own box + ghost list (src/p_range.jl:287,298,1729). */
int pa_host_global_to_local_block(int32_t D, const int64_t *n, const int64_t *lo, const int64_t *hi,
                                  const int64_t *ghost_gids, int64_t n_ghost, const int64_t *gids,
                                  int64_t count, int32_t *lids);
/* compresscoo(SparseMatrixCSR{1,Float64,Int32},I,J,V,m,n;combine=+,skip) src/sparse_utils.jl:313-350.
 * rowptr has m+1 entries; colval/nzval sized by a first call with colval == NULL (*nnz_out). */
int pa_host_compresscoo_csr(const int32_t *I, const int32_t *J, const double *V, int64_t count, int64_t m,
                            int64_t n, int skip, int32_t *rowptr, int32_t *colval, double *nzval,
                            int64_t *nnz_out);
/* split_format_locally for an assembled matrix whose local ids are [own|ghost] (perm = identity):
 * src/p_sparse_matrix.jl:823-899, own-row branches. Two-call protocol like above. */
int pa_host_split_csr(int64_t n_own_rows, int64_t n_own_cols, int64_t n_ghost_cols, const int32_t *rowptr,
                      const int32_t *colval, const double *nzval, int32_t *oo_rowptr, int32_t *oo_colval,
                      double *oo_nzval, int32_t *oh_rowptr, int32_t *oh_colval, double *oh_nzval,
                      int64_t *nnz_oo, int64_t *nnz_oh);

/* Host-only self-check of the SpMV row split and of the library-internal column encodings (row patterns, 16-bit
 * windows): encodes the given CSR pattern as pa_csr_create would and decodes every entry with the kernel's arithmetic.
 * Returns PA_ERR_ARG on any mismatch; the counters report how many chunks each encoding covers. */
int pa_host_check_spmv_encodings(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t *rowptr,
                                 const int32_t *colval, int index_base, int64_t *n_chunks, int64_t *n_pattern_chunks,
                                 int64_t *n_c16_chunks, int64_t *n_patterns);
/* Host-only self-check of the x-window groups of pa_csr_xwin_info (no GPU): every chunk in exactly one group or left to
 * the general kernel, every column of a group inside its window, windows within the kernel's LDS stage. */
int pa_host_check_xw_groups(int64_t n_rows, int64_t n_cols, int64_t nnz, const int32_t *rowptr, const int32_t *colval,
                            int index_base, int64_t *n_groups, int64_t *n_grouped_chunks, int64_t *staged_x_entries,
                            int64_t *grouped_entries, int64_t *n_big_groups);

/* Fused HPCG set-up for large parts: the same arrays as the chain above (build_matrix -> find_owner ->
 * union_ghost -> map_global_to_local! -> compresscoo -> split_format_locally, HPCG/src/sparse_matrix.jl:105-122)
 * without materialising the Int64 COO triplets.  Pass 1 returns the ghost gids in first-seen order
 * (src/p_range.jl:205-241) and the block sizes; pass 2 writes own_own / own_ghost (1-based Int32) and b. */
int pa_host_hpcg_ghosts(int64_t nx, int64_t ny, int64_t nz, int64_t gnx, int64_t gny, int64_t gnz, int64_t gix0,
                        int64_t giy0, int64_t giz0, int64_t *ghost_gids, int64_t *n_ghost, int64_t *nnz_oo,
                        int64_t *nnz_oh);
int pa_host_hpcg_split_csr(int64_t nx, int64_t ny, int64_t nz, int64_t gnx, int64_t gny, int64_t gnz, int64_t gix0,
                           int64_t giy0, int64_t giz0, const int64_t *ghost_gids, int64_t n_ghost, int32_t *oo_rowptr,
                           int32_t *oo_colval, double *oo_nzval, int32_t *oh_rowptr, int32_t *oh_colval,
                           double *oh_nzval, double *b);

/* the own_ghost block alone (the same oh_* arrays): for a part whose own_own block and b are generated in HBM
 * (pa_hpcg_own_block_create); only the rows on the part's surface are visited past the closed-form count */
int pa_host_hpcg_ghost_block(int64_t nx, int64_t ny, int64_t nz, int64_t gnx, int64_t gny, int64_t gnz, int64_t gix0,
                             int64_t giy0, int64_t giz0, const int64_t *ghost_gids, int64_t n_ghost, int32_t *oh_rowptr,
                             int32_t *oh_colval, double *oh_nzval);

/* Set-up of the multicolour smoother: the rows of a part (split blocks, 1-based Int32) dealt by colour into n_colors
 * blocks in the unsplit column order (own columns, then ghost columns + n_own_cols), plus the diagonal.  out_rowptr[k]
 * (n_own+1 entries, 1-based) is prefilled by the caller: a row of another colour has length 0 in block k. */
/* row pointers (1-based, n_own + 1 entries each) of the n_colors blocks pa_host_color_split fills; color[r] == -1: no block */
int pa_host_color_rowptrs(int64_t n_own, const int32_t *oo_rowptr, const int32_t *oh_rowptr, const int32_t *color,
                          int32_t n_colors, int32_t *const *out_rowptr);
int pa_host_color_split(int64_t n_own, int64_t n_own_cols, const int32_t *oo_rowptr, const int32_t *oo_colval,
                        const double *oo_nzval, const int32_t *oh_rowptr, const int32_t *oh_colval, const double *oh_nzval,
                        const int32_t *color, int32_t n_colors, const int32_t *const *out_rowptr,
                        int32_t *const *out_colval, double *const *out_nzval, double *diag);
/* The same with Int64 row pointers, for a part of 2^31 stored entries or more (columns stay Int32). */
int pa_host_hpcg_split_csr64(int64_t nx, int64_t ny, int64_t nz, int64_t gnx, int64_t gny, int64_t gnz, int64_t gix0,
                             int64_t giy0, int64_t giz0, const int64_t *ghost_gids, int64_t n_ghost, int64_t *oo_rowptr,
                             int32_t *oo_colval, double *oo_nzval, int64_t *oh_rowptr, int32_t *oh_colval,
                             double *oh_nzval, double *b);


/* ---- library-side renumbering for blocks without locality (csrc/pa_transpose.hip; round 4) ----------------------------------
 * pa_csr_locality_order: a reverse Cuthill-McKee order of a square block (own x own), computed on the device by level sets:
 * new_pos[i] (host, n_rows entries) = the position row / column i should take; band_before / band_after = max |row - col| in the
 * two numberings.  pa_csr_create_permuted: the same block with row i stored at row_pos[i] and column j renamed col_pos[j] (either
 * may be NULL); every row keeps its entries in their ORIGINAL order, so its sum adds the same products in the same order --
 * y_new[row_pos[i]] == y[i] bit for bit when x_new[col_pos[j]] = x[j].  The host mirror hides the order in the index partition's
 * local_to_device map (vectors, exchange plans): `renumber_for_locality(A)` of partitionedarrays.jl_amd/p_sparse_matrix.py. */
int pa_csr_locality_order(const pa_csr *A, int32_t *new_pos, int64_t *band_before, int64_t *band_after);
int pa_csr_create_permuted(const pa_csr *A, const int32_t *row_pos, const int32_t *col_pos, pa_csr **out);
/* pa_csr_create_transpose (pa_hip.h) for a block whose rows were renumbered: inside a row of A' the entries stand in ascending
 * row_rank[row] (host, n_rows entries: the ORIGINAL row of every stored row), the order the reference's transposed loop has on the
 * caller's numbering; NULL = ascending row, pa_csr_create_transpose itself. */
int pa_csr_create_transpose_ranked(const pa_csr *A, const int32_t *row_rank, pa_csr **out);

/* The block as a chain of `pieces` column pieces (2..8; entry (r, c) in piece floor((c - lower band edge at r) / width)): what the
 * library does by itself for unstructured rows whose band is wider than the sliding x window (PA_SPMV_COLSPLIT=0: never).  pa_spmv
 * runs the pieces in sequence, the later ones accumulating: the same additions in the same order as on the unsplit block. */
/* a scatter map back as destinations (0-based, -1 = skipped); checks that every slot adds its sources in ascending order (tests) */
int pa_scatter_download(const pa_scatter *s, int32_t *dest);
int pa_csr_create_colsplit(const pa_csr *A, int pieces, pa_csr **out);
/* A block of n_rows x n_cols without stored entries -- pa_csr_create with n_rows + 1 equal row pointers, without that array. */
int pa_csr_create_empty(pa_ctx *c, int64_t n_rows, int64_t n_cols, pa_csr **out);
/* A column-split chain: *pieces = its pieces (0: A is not one), *groups_one_launch = workgroups of the ONE launch pa_spmv runs it as
 * (the pieces' chunks and ring groups are cut at the same rows: a workgroup runs its rows through every piece and y reaches HBM
 * once; PA_SPMV_CHAIN_FUSED=0: never), 0 = a launch per piece, y written and re-read between them. */
int pa_csr_chain_info(const pa_csr *A, int32_t *pieces, int64_t *groups_one_launch);

/* ---- introspection of a CSR block (what the row-split kernel reads; none of it is needed to use the block) ---------- */
/* How the row-split chunks of A get their column indices (library-internal index compression; the values, the
 * results and pa_csr_update_values are unaffected): recomputed from row patterns / 16-bit windowed stream / 32-bit. */
int pa_csr_encoding(const pa_csr *A, int64_t *n_pattern_chunks, int64_t *n_c16_chunks, int64_t *n_c32_chunks);
/* Banded rows without a pattern: groups of consecutive chunks whose span of x is copied into LDS once and gathered from
 * there (csrc/pa_spmv_xwin.h; same products, same order -- spmv_csr!, src/sparse_utils.jl:649-669).  n_groups = 0: the
 * block does not use it (PA_SPMV_XWIN=0 turns it off, =2 forces it for every block that has groups). */
int pa_csr_xwin_info(const pa_csr *A, int64_t *n_groups, int64_t *n_chunks, int64_t *staged_x_entries,
                     int64_t *n_big_groups /* of n_groups: those on the 96 / 128 KiB windows (one workgroup per CU) */);
/* groups of the sliding x window (csrc/pa_spmv_xwin.h, k_spmv_xring; counted in pa_csr_xwin_info's groups too) */
int pa_csr_xring_info(const pa_csr *A, int64_t *n_ring_groups);
/* HBM bytes the block occupies (values, the column streams actually kept, row pointers, chunk table, descriptors).
 * A block whose chunks are described by row patterns keeps no columns for them: a stencil operator costs ~8 bytes per
 * stored entry, a block on the 16-bit stream ~14 (8 + 4 + 2). */
int pa_csr_device_bytes(const pa_csr *A, int64_t *bytes);
/* Bytes one product must READ from the block (each once): values, row pointers, chunk table and, per chunk, whatever
 * gives it its columns (pattern descriptor / window table + 16-bit stream / 32-bit columns).  Plus x once and y once
 * this is the compulsory HBM traffic of pa_spmv -- bench.py's `roofline.moved_bytes_per_launch` -- as opposed to the
 * reference's CSR bytes (12 per stored entry) `roofline.achieved` is quoted on. */
int pa_csr_stream_bytes(const pa_csr *A, int64_t *bytes);
/* Optional, lossless: with PA_SPMV_VALUE_DICT=1 in the environment at creation, a block whose stored values take at most
 * 64 distinct bit patterns (27-point HPCG: 2; Q1 stiffness on a uniform grid: about a dozen) also keeps one byte per
 * entry and the kernels stream that instead of the 8-byte values -- same values, same products, same order, same bits.
 * pa_csr_update_values* drop the dictionary (the block continues on the fp64 stream).  n_values = distinct values in
 * use, 0 when the block streams fp64 values.  Off by default: bench.py's headline never uses it. */
int pa_csr_value_dict(const pa_csr *A, int *n_values);
/* Round 4: at a handle's first product the library makes a twin of own_ghost whose columns are positions of consistent!'s RECEIVE
 * BUFFER (same entries, same order): own x ghost then gathers b's ghost values from buffer_rcv as soon as the messages are in,
 * and the unpack that makes b itself consistent (src/p_vector.jl:603-611) runs behind it -- one launch less on the critical path
 * of every mul!.  Same bits.  Not when a ghost column with stored entries receives no message, nor with
 * PA_MUL_GHOST_FROM_BUFFER=0.  pa_mul_all packs and delivers all parts with one push launch (PA_PUSH=0: pack per part + copies). */
int pa_matrix_ghost_from_buffer(const pa_matrix *m, int *yes);
int pa_csr_download_entries(const pa_csr *A, int32_t *rows, int32_t *cols);
/* Round 5: mul!(c,a,b) of a part as ONE launch (csrc/pa_fused.hip; src/p_sparse_matrix.jl:2090-2142, same additions in the same
 * order: same bits).  The part's boundary rows -- rows with stored entries in own_ghost -- are summed by the tail of the launch from
 * a block that holds their own_own entries followed by their own_ghost entries; the other rows by own x own's chunks in front.
 * pa_mul_all then queues P + 1 launches on one stream (the push launch, which also completes consistent!(b), and one per part).
 * Decided per handle at its first product; PA_MUL_FUSED=0 keeps the separate launches.  *yes = 1: fused; *n_boundary_rows: the
 * tail's rows. */
int pa_matrix_fused(const pa_matrix *m, int *yes, int64_t *n_boundary_rows);
/* products this context has run as one launch so far; of those, with the exchange inside the launch (one part per process over the
 * ipc link: push by the first blocks, arrival acquired by the tail, unpack and acknowledgement by the tail) */
int pa_ctx_fused_launches(const pa_ctx *c, int64_t *all, int64_t *with_exchange);
/* The switches of the product path (PA_PUSH, PA_GRAPH_ONE_STREAM, PA_MUL_GHOST_FROM_BUFFER, PA_MUL_FUSED, PA_SPMV_ALTERNATE) are
 * read from the environment when a context is created; this reads them again (tests flip them inside one process). */
int pa_ctx_reload_env(pa_ctx *c);

#ifdef __cplusplus
}
#endif
#endif /* PA_HIP_EXPERIMENTAL_H */
