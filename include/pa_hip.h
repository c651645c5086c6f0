/*
 * pa_hip.h -- C ABI of libpa_hip.so: the MI355X (gfx950) device path for
 * PartitionedArrays.jl's  mul!(c::PVector, A::PSparseMatrix, b::PVector)  and the ghost exchange
 * (consistent! / assemble! / exchange!) it depends on.
 *
 * The reference (100 % Julia) has no FFI; its extension points are multiple dispatch on the local
 * vector type, the local matrix type and the backend array type.  Each entry point
 * below names the reference method it replaces (paths relative to /root/reference).  The Julia glue
 * that `ccall`s these symbols is partitionedarrays.jl_amd/julia/PartitionedArraysHIP.jl; the
 * Python/ctypes host mirror used by the tests is partitionedarrays.jl_amd/.
 *
 * Conventions
 *  - plain pointers and sizes only; every function returns 0 on success, <0 on error
 *    (pa_last_error() gives the message of the calling thread's last failure);
 *    nothing throws or aborts.  The reference raises Julia exceptions at the same places
 *    (@assert / @boundscheck / error(...)); the glue turns a non-zero status into error(...).
 *  - host index arrays are handed over exactly as the reference stores them: 1-based Int32/Int64
 *    (`index_base` = 1); the library converts to 0-based Int32 in HBM.
 *  - all device work is asynchronous on the two streams a pa_ctx owns ("compute" and "comm").
 *    A pa_ctx is driven by one host thread at a time (the reference is single-threaded per rank).
 *  - the caller keeps ownership of every host array; the library copies what it needs.
 *  - fp64 values; the per-row summation order of pa_spmv is the reference's (ascending p, one
 *    rounding per multiply and per add, no FMA) for every row -- rows longer than one LDS chunk
 *    (PA_SPMV_CHUNK_NNZ) are walked window by window in the same order -- so results are
 *    bit-identical to spmv_csr! / SparseMatricesCSR.mul!.
 */
#ifndef PA_HIP_H
#define PA_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PA_OK 0
#define PA_ERR_HIP -1      /* a HIP runtime call failed */
#define PA_ERR_ARG -2      /* invalid argument (the reference's @assert / @boundscheck) */
#define PA_ERR_RCCL -3     /* an RCCL call failed or librccl could not be loaded */
#define PA_ERR_STATE -4    /* call sequence violated (e.g. exchange_end without exchange_begin) */

#define PA_SEG_OWN 0       /* own_values(v)   src/p_vector.jl:20-22  */
#define PA_SEG_GHOST 1     /* ghost_values(v) src/p_vector.jl:24-26  */
#define PA_SEG_LOCAL 2     /* local_values(v): the whole [own|ghost] array */

#define PA_CONSISTENT 0    /* consistent!(v): ghost <- owner, `insert`  src/p_vector.jl:747-755 */
#define PA_ASSEMBLE 1      /* assemble!(v):   owner += ghosts, then ghosts := 0  src/p_vector.jl:695-708 */

#define PA_STREAM_COMPUTE 0
#define PA_STREAM_COMM 1

#ifndef PA_SPMV_CHUNK_NNZ        /* (probe builds of the library override it) */
#define PA_SPMV_CHUNK_NNZ 1536   /* LDS-staged products per workgroup (12 KiB of fp64) */
#endif

typedef struct pa_ctx pa_ctx;     /* one device + its two streams                                  */
typedef struct pa_vec pa_vec;     /* local values of one part of a PVector, layout [own | ghost]   */
typedef struct pa_gs pa_gs;       /* the level-scheduled Gauss-Seidel smoother of one part (below)  */
typedef struct pa_csr pa_csr;     /* one CSR block of a SplitMatrix (own_own, own_ghost, ...)       */
typedef struct pa_plan pa_plan;   /* VectorAssemblyCache of one part (src/p_vector.jl:418-426)      */
typedef struct pa_comm pa_comm;   /* RCCL communicator: one rank per part (MPIArray analogue)       */
typedef struct pa_event pa_event; /* HIP event on one of the ctx streams                            */

/* ---- library ------------------------------------------------------------------------------- */
int pa_version(void);
const char *pa_last_error(void);
int pa_device_count(int *count);

/* ---- context (with_mpi / with_debug lifecycle: src/mpi_array.jl:64-83, src/debug_array.jl:7) - */
int pa_ctx_create(int device, pa_ctx **ctx);
int pa_ctx_destroy(pa_ctx *ctx);
int pa_ctx_sync(pa_ctx *ctx);                        /* both streams */
int pa_ctx_stream(pa_ctx *ctx, int which, void **hip_stream);
int pa_ctx_device_info(pa_ctx *ctx, int *cus, int *xcds, size_t *hbm_bytes, char *name, size_t name_len);
/* The comm stream is created with the device's greatest priority (numerically lowest), the compute stream with the
 * least: the pack / RCCL send-recv / unpack kernels of t = consistent!(b) must get CUs while own x own (~300 k
 * workgroups) is running, or the exchange of mul! (src/p_sparse_matrix.jl:2098-2100) is not hidden.  Any out-pointer
 * may be NULL. */
int pa_ctx_stream_priority(pa_ctx *ctx, int which, int *priority, int *least, int *greatest);

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
/* ---- psparse(I,J,V,rows,cols;assembled=true) of one part ON THE DEVICE (csrc/pa_assemble.hip) ------------------------
 * The reference's route for COO triplets in global ids -- union_ghost(rows, J, find_owner(rows, J)) (src/p_range.jl:205-259),
 * map_global_to_local! (src/p_sparse_matrix.jl:1253-1254), compresscoo(...; combine = +, skip) (src/sparse_utils.jl:313-350),
 * split_format_locally (src/p_sparse_matrix.jl:823-899) -- as kernels, scans and radix sorts over the uploaded triplets.
 * rows / columns: block partitions (own ids = the box [lo, hi] of an n_global grid, 1-based inclusive, column-major; the row
 * partition has no ghosts).  known_ghosts: ghost gids the column partition already has, in its order; discover_ghosts != 0
 * appends the new ones in first-seen order (union_ghost), 0 treats unknown columns as not local (id 0: the CSR skip rule turns
 * such entries into (1,1,0.0)).  Duplicates are added in input order.  Results: the ghost gids (known, then new), the two
 * blocks as pa_csr objects (created on the device: no host CSR exists unless pa_coo_assembly_download is called), sizes. */
typedef struct pa_coo_assembly pa_coo_assembly;
int pa_coo_assemble(pa_ctx *ctx, int64_t count, const int64_t *I, const int64_t *J, const double *V, int32_t D,
                    const int64_t *n_rows_global, const int64_t *row_lo, const int64_t *row_hi, const int64_t *n_cols_global,
                    const int64_t *col_lo, const int64_t *col_hi, int64_t n_known_ghosts, const int64_t *known_ghosts,
                    int discover_ghosts, pa_coo_assembly **out);
int pa_coo_assembly_info(const pa_coo_assembly *h, int64_t *n_own_rows, int64_t *n_own_cols, int64_t *n_ghost,
                         int64_t *nnz_own_own, int64_t *nnz_own_ghost, double *device_ms);
int pa_coo_assembly_ghosts(const pa_coo_assembly *h, int64_t *ghost_gids);
int pa_coo_assembly_blocks(pa_coo_assembly *h, pa_csr **own_own, pa_csr **own_ghost);
/* 1-based host copies of a block (which: 0 own_own, 1 own_ghost): rowptr[n_own_rows + 1], colval / nzval[nnz] */
int pa_coo_assembly_download(const pa_coo_assembly *h, int which, int32_t *rowptr, int32_t *colval, double *nzval);
int pa_coo_assembly_destroy(pa_coo_assembly *h);
/* The disassembled route -- psparse(I,J,V,rows,cols) with the default flags, then assemble (src/p_sparse_matrix.jl:1150-1219,
 * 1590-1756): a part's triplets may name rows other parts own.  pa_coo_subassemble: the sub-assembled local matrix (ghost rows
 * and ghost columns numbered in first-seen order, duplicates added in input order); the own rows stay in HBM, the ghost rows
 * -- gids, and entries sorted by (0-based ghost row, 0-based local column [own | ghost]) -- come back through
 * pa_coo_subassembly_ghost_rows (ghost columns: pa_coo_assembly_ghosts) for the caller to send to their owners.
 * pa_coo_assemble_finish: the own rows' entries in CSR order followed by the n_rcv triplets (global ids) that arrived, through
 * the assembled route: the blocks and final ghost columns of setup_own_triplets + union_ghost + finalize_values (:1656-1723),
 * read with pa_coo_assembly_info/_ghosts/_blocks/_download. */
int pa_coo_subassemble(pa_ctx *ctx, int64_t count, const int64_t *I, const int64_t *J, const double *V, int32_t D,
                       const int64_t *n_rows_global, const int64_t *row_lo, const int64_t *row_hi, const int64_t *n_cols_global,
                       const int64_t *col_lo, const int64_t *col_hi, pa_coo_assembly **out);
int pa_coo_subassembly_info(const pa_coo_assembly *h, int64_t *n_ghost_rows, int64_t *n_ghost_cols, int64_t *n_own_entries,
                            int64_t *n_ghost_row_entries);
int pa_coo_subassembly_ghost_rows(const pa_coo_assembly *h, int64_t *row_gids, int32_t *g_row, int32_t *g_col, double *g_val);
int pa_coo_assemble_finish(const pa_coo_assembly *sub, int64_t n_rcv, const int64_t *I, const int64_t *J, const double *V,
                           pa_coo_assembly **out);

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
/* affinity[k] = the mean number of stored entries of a colour-k row whose column is one of kept_rows (0-based own rows: the
 * fine rows a coarse grid keeps).  A multicolour smoother inside a multigrid cycle sweeps its colours in order of decreasing
 * affinity, so that the kept rows' own colour sits at the turn of the symmetric sweep, not at its end (where the residual
 * the restriction injects would be zero up to rounding). */
int pa_csr_color_affinity(const pa_csr *own_own, const int32_t *color, int32_t n_colors, const int32_t *kept_rows, int64_t n_kept,
                          double *affinity);
/* HPCG's 27-point operator of one part (HPCG/src/sparse_matrix.jl:28-122), own_own block and right-hand side, generated
 * in HBM: the arrays pa_host_hpcg_split_csr writes (oo_*, b), no host copy, no upload.  nx,ny,nz: the part's box; gnx,gny,gnz:
 * the global grid; gix0,giy0,giz0: global coordinates (1-based) of the part's first node.  b may be NULL. */
int pa_hpcg_own_block_create(pa_ctx *ctx, int64_t nx, int64_t ny, int64_t nz, int64_t gnx, int64_t gny, int64_t gnz,
                             int64_t gix0, int64_t giy0, int64_t giz0, pa_csr **own_own, pa_vec *b);
/* b alone (a vector created after the block is placed knowing the matrix streams' memory class) */
int pa_hpcg_rhs(pa_ctx *ctx, int64_t nx, int64_t ny, int64_t nz, int64_t gnx, int64_t gny, int64_t gnz, int64_t gix0,
                int64_t giy0, int64_t giz0, pa_vec *b);

/* testing aid: a host copy of one of the arrays the product kernel reads (first slab of the block) -- 0 row pointers,
 * 1 32-bit columns, 2 16-bit codes, 3 windows, 4 pattern descriptors, 5 pattern table, 6 chunk table, 7 compacted row ids;
 * *bytes = the array's size, copied when capacity allows.  The set-up runs on the device (csrc/pa_setup.hip; PA_SETUP_DEVICE=0:
 * the host encoder): the tests compare the two builds of every array with this. */
int pa_csr_debug_array(const pa_csr *A, int which, void *host, int64_t capacity, int64_t *bytes);
int pa_csr_memory_class(const pa_csr *A, int *cls);
int pa_vec_memory_class(const pa_vec *v, int *cls);

/* ---- events / timing (replaces PTimer, src/p_timer.jl; HIP events on the launching stream) --- */
int pa_event_create(pa_ctx *ctx, pa_event **ev);
int pa_event_destroy(pa_event *ev);
int pa_event_record(pa_event *ev, int which_stream);
int pa_event_elapsed_ms(pa_event *start, pa_event *stop, float *ms);   /* synchronises `stop` */

/* ---- device vectors: the local vector type V of PVector{V} (src/p_vector.jl:8-26,324-345) ---- */
int pa_vec_create(pa_ctx *ctx, int64_t n_own, int64_t n_ghost, pa_vec **v);   /* zero-filled */
int pa_vec_wrap(pa_ctx *ctx, void *device_ptr, int64_t n_own, int64_t n_ghost, pa_vec **v);
int pa_vec_destroy(pa_vec *v);
int pa_vec_sizes(const pa_vec *v, int64_t *n_own, int64_t *n_ghost);
int pa_vec_data(pa_vec *v, void **device_ptr);
int pa_vec_upload(pa_vec *v, const double *host, int64_t offset, int64_t len);     /* local offset */
int pa_vec_download(const pa_vec *v, double *host, int64_t offset, int64_t len);
int pa_vec_fill(pa_vec *v, int segment, double value);                  /* fill!(…_values(v), value) */
int pa_vec_copy(pa_vec *dst, const pa_vec *src, int segment);           /* copy!  */
/* own-values BLAS-1 (src/p_vector.jl:1189-1206, broadcast :1216-1277): y = a*x + b*y on a segment, the multiply and the
 * add rounded separately.  b == 0 is the assignment y = a*x: y is not read (what `dest .= a .* v` does); x may be y. */
int pa_vec_axpby(pa_vec *y, double a, const pa_vec *x, double b, int segment);
/* local dot over OWN values (the per-part term of dot(a,b), src/p_vector.jl:1190); deterministic
 * two-pass reduction; result left in a device scalar (pa_vec_dot_result) and, if host_out != NULL,
 * copied back (synchronises the compute stream). */
int pa_vec_dot(const pa_vec *x, const pa_vec *y, double *host_out);
int pa_vec_dot_result(pa_ctx *ctx, void **device_scalar);
/* Read the device scalar back (synchronises the compute stream): the value of the last pa_vec_dot, or of its
 * all-reduce over the parts after pa_comm_allreduce_sum(comm, device_scalar, 1, PA_STREAM_COMPUTE). */
int pa_ctx_read_scalar(pa_ctx *ctx, double *host_out);

/* Solver scalars that never visit the host.  The reference's CG (HPCG/src/ref_cg.jl:52-67) reads rho, u'c and
 * |r| back on every iteration, each a blocking reduction; here a context owns PA_N_SLOTS device doubles ("slots";
 * slot 0 is also where pa_vec_dot leaves its result) and a coefficient is written (c, num, den) = c*slot[num]/slot[den]
 * with PA_SLOT_ONE standing for 1.0, evaluated on the device in IEEE fp64 -- the same value the host would compute.
 * accumulate != 0 adds to the slot instead of overwriting it (the in-order sum over several parts of one process,
 * reduction(+,...) in src/debug_array.jl); across processes all-reduce pa_ctx_slot_ptr with pa_comm_allreduce_sum. */
#define PA_N_SLOTS 16
#define PA_SLOT_ONE (-1)
int pa_vec_dot_slot(const pa_vec *x, const pa_vec *y, int slot, int accumulate);
int pa_vec_axpby_slot(pa_vec *y, double ca, int a_num, int a_den, const pa_vec *x, double cb, int b_num, int b_den,
                      int segment);                                   /* y = (ca*s[a_num]/s[a_den])*x + (cb*...)*y */
/* x .+= alpha .* u ; r .-= alpha .* c ; slot[rr_slot] = dot(r,r), alpha = slot[num]/slot[den]: the three
 * statements HPCG/src/ref_cg.jl:64-67 in one pass over the own values (bit-identical to the unfused calls). */
int pa_cg_update(pa_vec *x, pa_vec *r, const pa_vec *u, const pa_vec *c, int num, int den, int rr_slot,
                 int accumulate);
/* The same three statements split so that the loop needs two passes less: r .-= alpha .* c with slot[rr_slot] = dot(r,r)
 * (pa_cg_r_update), and x .+= alpha .* u deferred to the moment u is about to change, fused with u .= z .+ beta .* u
 * (pa_cg_xu_update, alpha = slot[a_num]/slot[a_den] of the iteration before, beta = slot[b_num]/slot[b_den];
 * HPCG/src/ref_cg.jl:56,64-65).  x is not read inside the loop, so the iterates are the same numbers, bit for bit; the
 * caller flushes the last x update with pa_vec_axpby_slot when the loop ends. */
int pa_cg_r_update(pa_vec *r, const pa_vec *c, int num, int den, int rr_slot, int accumulate);
int pa_cg_xu_update(pa_vec *x, pa_vec *u, const pa_vec *z, int a_num, int a_den, int b_num, int b_den);
int pa_ctx_slot_ptr(pa_ctx *ctx, int slot, void **device_ptr);
int pa_ctx_write_slot(pa_ctx *ctx, int slot, double value);           /* asynchronous, compute stream */
int pa_ctx_read_slots(pa_ctx *ctx, int first, int n, double *host_out); /* synchronises the compute stream */

/* ---- CSR blocks: the local matrix type (src/sparse_utils.jl:609-669; SplitMatrix blocks
 *      src/p_sparse_matrix.jl:588-627,670-681) ------------------------------------------------- */
/* rowptr has n_rows+1 entries, colval/nzval nnz entries, columns sorted inside a row (what
 * compresscoo / sparsecsr produce).  index_bytes in {4,8}, index_base in {0,1}. */
int pa_csr_create(pa_ctx *ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, const void *rowptr,
                  const void *colval, int index_bytes, int index_base, const double *nzval, pa_csr **A);
/* The same with separate widths for the row pointers and the column indices.  Device offsets are Int32; a block with
 * 2^31 stored entries or more (up to the 288 GB of the GPU: rows and columns still < 2^31) needs 64-bit row pointers
 * and is stored as consecutive row slabs with Int32 offsets each -- pa_spmv, pa_csr_update_values* and pa_mul* see one
 * block; the fused Gauss-Seidel / restriction epilogues need a single slab.  0-based Int32 colval is used in place. */
int pa_csr_create_mixed(pa_ctx *ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, const void *rowptr, int rowptr_bytes,
                        const void *colval, int colval_bytes, int index_base, const double *nzval, pa_csr **A);
/* CSC input (the reference's default SparseMatrixCSC storage); converted to CSR at upload:
 * spmv_csc! and spmv_csr! give bit-identical results (same per-row add order). */
int pa_csr_create_from_csc(pa_ctx *ctx, int64_t n_rows, int64_t n_cols, int64_t nnz, const void *colptr,
                           const void *rowval, int index_bytes, int index_base, const double *nzval,
                           pa_csr **A);
/* mul!(y,A,x,alpha,beta) with alpha != 1 is third-party arithmetic in the reference, and its two local matrix types differ by one
 * rounding: SparseMatricesCSR adds (nz*x[col])*alpha, SparseArrays' CSC method forms axj = x[col]*alpha per column and adds nz*axj.
 * A block made by pa_csr_create_from_csc follows the CSC form (pa_spmv scales x into a scratch vector, then runs with alpha = 1),
 * every other block the CSR form; identical for alpha in {1, -1, 2^k}.  pa_csr_set_alpha_inside overrides per block
 * (PA_CSC_ALPHA_INSIDE=0: CSC-made blocks use the CSR form too). */
int pa_csr_set_alpha_inside(pa_csr *A, int on);
int pa_csr_update_values(pa_csr *A, const double *nzval);   /* same pattern, new nonzeros(A) */
/* nonzeros(A) .= src_local[offset : offset+nnz) -- device to device, on the compute stream (K7: the re-assembled
 * values of psparse!/assemble!(B,A,cache), src/p_sparse_matrix.jl:1291-1305,1762-1816, never leave HBM). */
int pa_csr_update_values_from(pa_csr *A, const pa_vec *src, int64_t offset);
int pa_csr_destroy(pa_csr *A);
int pa_csr_info(const pa_csr *A, int64_t *n_rows, int64_t *n_cols, int64_t *nnz, int64_t *n_chunks,
                int64_t *n_nonempty_rows, int64_t *n_long_rows);
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
/* y_seg = beta*y_seg + alpha*A*x_seg.
 *   spmv!(b,A,x)            (src/sparse_utils.jl:617-623,649-669)  <=> alpha=1, beta=0
 *   muladd!(b,A,x)          (src/p_sparse_matrix.jl:2088)            <=> alpha=1, beta=1
 *   mul!(b,A,x,alpha,beta)  (SparseMatricesCSR 0.6)                  <=> general
 * Launched on the compute stream.  x and y must not alias. */
int pa_spmv(const pa_csr *A, const pa_vec *x, int x_segment, pa_vec *y, int y_segment, double alpha,
            double beta);

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

/* ---- exchange plans: p_vector_cache_impl / VectorAssemblyCache (src/p_vector.jl:418-468) ------ */
/* The arrays are the cache fields in ASSEMBLY orientation, as the reference builds them
 * (src/p_range.jl:417-531):
 *   nbr_snd/ptrs_snd/idx_snd : owners of my ghosts (sorted), and my ghost local ids grouped by owner
 *   nbr_rcv/ptrs_rcv/idx_rcv : parts that ghost my owns, and the own local ids they hold
 * ptrs are JaggedArray.ptrs (n+1 entries).  consistent! uses the reversed cache (src/p_vector.jl:748).
 * `part` is this part's id in the same base as the neighbour ids. */
int pa_plan_create(pa_ctx *ctx, int32_t part, int64_t n_local, int32_t n_snd, const int32_t *nbr_snd,
                   const int32_t *ptrs_snd, const int32_t *idx_snd, int32_t n_rcv, const int32_t *nbr_rcv,
                   const int32_t *ptrs_rcv, const int32_t *idx_rcv, int index_base, pa_plan **plan);
int pa_plan_destroy(pa_plan *plan);
/* Device send/receive buffers of the given mode (JaggedArray.data of buffer_snd / buffer_rcv after
 * the reverse() of consistent!), for callers that move the bytes themselves. */
int pa_plan_buffers(pa_plan *plan, int mode, void **snd, int64_t *snd_len, void **rcv, int64_t *rcv_len);

/* Split-phase exchange, mirroring  t = exchange!(…)  …  wait(t)  (src/p_vector.jl:587-612):
 *   pa_exchange_pack   : comm stream waits for the compute stream, then
 *                        buffer_snd.data[p] = values[idx[p]]                      (:595-599)
 *   <transport>        : one of pa_exchange_local / pa_exchange_rccl / caller-driven copies
 *   pa_exchange_finish : compute stream waits for the comm stream, then
 *                        values[idx[p]] = f(values[idx[p]], buffer_rcv.data[p])   (:605-609)
 *                        f = insert (PA_CONSISTENT) or + in ascending p (PA_ASSEMBLE, deterministic,
 *                        bit-identical to the reference's loop) followed by ghost := 0 (:703-705). */
int pa_exchange_pack(pa_plan *plan, const pa_vec *v, int mode);
int pa_exchange_finish(pa_plan *plan, pa_vec *v, int mode);

/* Transport A: every part lives in this process (DebugArray analogue, src/debug_array.jl:250-255,
 * src/primitives.jl:1020-1042): device-to-device slice copies between the plans' buffers.
 * plans[i] must be the plan of part (i + index_base). Call after pa_exchange_pack on ALL parts. */
int pa_exchange_local(pa_plan *const *plans, int32_t n_parts, int mode);
/* Transport B: one process per part over RCCL (MPIArray analogue, src/mpi_array.jl:575-614):
 * one ncclGroup of ncclSend/ncclRecv per neighbour on the comm stream; rank = part - index_base. */
int pa_exchange_rccl(pa_plan *plan, pa_comm *comm, int mode);
/* Transport C, "push" (csrc/pa_push.hip): the pack kernel stores every send slice straight into the receive buffer of the part
 * it goes to -- pack and exchange! fused, no send buffer, no copies.
 *   pa_exchange_push_local: every part of this process, ONE launch per device; replaces pa_exchange_pack on every part +
 *     pa_exchange_local (v[i]: the vector of part i).  pa_exchange_finish per part is next, as after pa_exchange_local.
 *   One part per process: the neighbours' receive buffers are mapped over hipIpc.  pa_plan_ipc_blob gives the opaque bytes
 *     (ipc handles + slice tables) a neighbour needs; the host language carries every part's blob to its neighbours (or to
 *     everybody) as it carries the RCCL unique id; pa_plan_ipc_connect(plan, blobs...) opens them.  pa_exchange_push_ipc then
 *     replaces pa_exchange_pack + pa_exchange_rccl: the stores travel over xGMI, arrival is a sequence number written behind the
 *     payload, the receiver acknowledges after its unpack (flow control for the next exchange).  A wait that lasts longer than
 *     PA_IPC_TIMEOUT_S (30) raises the link's status (pa_plan_ipc_status: 0 ok, 1 arrival, 2 acknowledgement) and the next
 *     exchange over the link fails with PA_ERR_STATE -- the GPU is never left spinning.  Not inside a graph capture.
 * pa_mul5 / pa_mul_no_lat / pa_mul_dot / pa_mul5_transpose with comm == NULL use a connected plan's ipc link. */
int pa_exchange_push_local(pa_plan *const *plans, int32_t n_parts, pa_vec *const *v, int mode);
int pa_plan_ipc_blob_size(pa_plan *plan, int64_t *bytes);
int pa_plan_ipc_blob(pa_plan *plan, void *out, int64_t capacity);
int pa_plan_ipc_connect(pa_plan *plan, int32_t n_blobs, const void *const *blobs, const int64_t *sizes);
int pa_plan_ipc_status(pa_plan *plan, int *status);
int pa_exchange_push_ipc(pa_plan *plan, const pa_vec *v, int mode);

/* ---- operator level: the whole mul! of a part in one call ------------------------------------------------------------
 * pa_matrix = the operands of mul! that do not change between products: the own_own / own_ghost blocks of an ASSEMBLED
 * PSparseMatrix part (src/p_sparse_matrix.jl:588-627,1040-1092) and the exchange plan of its column partition (the
 * cache of the PVector it multiplies).  Blocks and plan stay the caller's (pa_matrix_destroy frees only the handle).
 *   pa_mul (c,a,b)            src/p_sparse_matrix.jl:2090-2103:  t = consistent!(b) [pack, neighbour exchange on the comm
 *                             stream]; c_own = A_oo b_own on the compute stream, overlapping; wait(t) [unpack]; c_own += A_oh b_ghost
 *   pa_mul5(c,a,b,alpha,beta) :2105-2142, assembled branch (a sub-assembled matrix needs its ghost-row blocks and
 *                             assemble!(c): compose pa_spmv / pa_exchange_* as the host mirror's mul5_ does)
 * One part per process: pass the RCCL communicator (NULL only for a single part without neighbours).
 *   pa_mul_all                every part of one process (DebugArray, src/debug_array.jl:110-117,250): parts r = 0..n-1 in
 *                             order, exchange by device-to-device copies; one call queues all the pipelines. */
typedef struct pa_matrix pa_matrix;
int pa_matrix_create(pa_ctx *ctx, const pa_csr *own_own, const pa_csr *own_ghost, pa_plan *col_plan, pa_matrix **m);
int pa_matrix_destroy(pa_matrix *m);
int pa_mul(pa_matrix *m, pa_comm *comm, pa_vec *c, pa_vec *b);
int pa_mul5(pa_matrix *m, pa_comm *comm, pa_vec *c, pa_vec *b, double alpha, double beta);
/* mul_no_lat!(c,a,b) (HPCG/src/hpcg_utils.jl:6-17): the exchange is completed BEFORE own x own -- HPCG's reference order and the
 * "overlap off" side of bench.py's on/off comparison.  Same kernels and bits as pa_mul. */
int pa_mul_no_lat(pa_matrix *m, pa_comm *comm, pa_vec *c, pa_vec *b);
int pa_mul_all(pa_matrix *const *m, int32_t n_parts, pa_vec *const *c, pa_vec *const *b, double alpha, double beta);
/* Round 4: at a handle's first product the library makes a twin of own_ghost whose columns are positions of consistent!'s RECEIVE
 * BUFFER (same entries, same order): own x ghost then gathers b's ghost values from buffer_rcv as soon as the messages are in,
 * and the unpack that makes b itself consistent (src/p_vector.jl:603-611) runs behind it -- one launch less on the critical path
 * of every mul!.  Same bits.  Not when a ghost column with stored entries receives no message, nor with
 * PA_MUL_GHOST_FROM_BUFFER=0.  pa_mul_all packs and delivers all parts with one push launch (PA_PUSH=0: pack per part + copies). */
int pa_matrix_ghost_from_buffer(const pa_matrix *m, int *yes);
/* ---- transpose(A) on the device (csrc/pa_transpose.hip): mul!(c,transpose(a),b,alpha,beta), src/p_sparse_matrix.jl:2144-2162;
 * spmtv!, src/sparse_utils.jl:613-647 --------------------------------------------------------------------------------------
 * pa_csr_create_transpose: A' of a block resident in HBM as a new block, built without a host copy (column encoding decoded,
 * one stable radix sort by column, the usual device-side block constructor).  Inside a row of A' the entries are ordered by
 * ascending row of A: y = beta*y + alpha*A'*x through pa_spmv then adds, per output entry, in the order of the reference's
 * scatter loop (spmv_csc! on the CSR arrays; SparseMatricesCSR's transposed mul!) -- bit-identical.  Not for chains of slabs.
 * pa_matrix_create_transposed: the handle of transpose(a) for an ASSEMBLED a from A_oo' and A_oh' (pa_csr_create_transpose of
 * a's own_own / own_ghost blocks) and the plan of the vector c lives on (a's column partition); blocks and plan stay the caller's.
 * pa_mul5_transpose: ghost(c) = alpha*A_oh'*own(b); assemble!(c) started; own(c) = beta*own(c) + alpha*A_oo'*own(b) overlapped
 * with the exchange; wait.  c lives on axes(a,2) (own + ghost columns), b on axes(a,1).  comm / _all as for pa_mul5 / pa_mul_all.
 * pa_csr_download_entries: 0-based (row, column) of every stored entry in storage order as the kernels decode them (tests). */
int pa_csr_create_transpose(const pa_csr *A, pa_csr **out);
int pa_matrix_create_transposed(pa_ctx *ctx, const pa_csr *own_own_t, const pa_csr *own_ghost_t, pa_plan *col_plan, pa_matrix **out);
int pa_mul5_transpose(pa_matrix *m, pa_comm *comm, pa_vec *c, pa_vec *b, double alpha, double beta);
int pa_mul5_transpose_all(pa_matrix *const *m, int32_t n_parts, pa_vec *const *c, pa_vec *const *b, double alpha, double beta);
int pa_csr_download_entries(const pa_csr *A, int32_t *rows, int32_t *cols);
/* mul!(c,a,b) that also leaves this part's share of dot(b,c) in a slot (accumulate != 0: added to it): the CG loop's
 * c = A*u and u'c (HPCG/src/ref_cg.jl:59-60) without a pass over u and c for the dot -- every workgroup of the product
 * kernels adds b_own[row] * (its rows' products) in a fixed order, two small launches reduce the per-chunk partial sums.
 * Deterministic, but not the association of pa_vec_dot_slot: u'c agrees with the separate dot to rounding (~1e-15
 * relative), not bit for bit.  c is bit-identical to pa_mul's.  Across processes all-reduce the slot as for pa_vec_dot_slot.
 * pa_mul_all_dot: every part of one process, the slot = the sum over the parts in part order. */
int pa_mul_dot(pa_matrix *m, pa_comm *comm, pa_vec *c, pa_vec *b, int slot, int accumulate);
int pa_mul_all_dot(pa_matrix *const *m, int32_t n_parts, pa_vec *const *c, pa_vec *const *b, int slot);

/* ---- hipGraph capture: a launch-bound loop body is queued once, captured, and replayed ------------------------------
 * Between pa_graph_begin and pa_graph_end nothing runs: every asynchronous entry point (pa_spmv, pa_exchange_pack /
 * _local / _finish, pa_vec_axpby*, pa_vec_dot_slot, pa_cg_update, pa_mul*, pa_gs_color_sweep, ...) is recorded from the
 * compute stream (the comm stream joins through the exchange's events); calls that synchronise or allocate
 * (pa_vec_download, pa_ctx_read_*, pa_*_create) are not allowed inside.  pa_graph_launch replays on the compute stream. */
typedef struct pa_graph pa_graph;
int pa_graph_begin(pa_ctx *ctx);
int pa_graph_end(pa_ctx *ctx, pa_graph **g);
int pa_graph_launch(pa_graph *g);
int pa_graph_destroy(pa_graph *g);

/* ---- deterministic scatter-add maps: sparse_matrix!(A,V,K) (src/sparse_utils.jl:454-466) -------------------- */
/* dst[dest[p]] += src[p] for p ascending (entries with dest[p] < index_base are skipped, as `k < 1` is there);
 * one lane per distinct destination adds its sources in ascending p: the reference's order, no atomics.
 * zero_first != 0 does fillstored!(dst,0) first.  dst/src are whole local vectors. */
typedef struct pa_scatter pa_scatter;
int pa_scatter_create(pa_ctx *ctx, int64_t n_dst, int64_t n_src, const int32_t *dest, int index_base, pa_scatter **s);
int pa_scatter_destroy(pa_scatter *s);
int pa_scatter_add(pa_scatter *s, pa_vec *dst, const pa_vec *src, int zero_first);

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

/* ---- RCCL communicator (MPI.Init / Comm_dup analogue, src/mpi_array.jl:42-53) ---------------- */
#define PA_UNIQUE_ID_BYTES 128
int pa_comm_unique_id(char id[PA_UNIQUE_ID_BYTES]);           /* on one rank; broadcast it yourself */
int pa_comm_create(pa_ctx *ctx, const char id[PA_UNIQUE_ID_BYTES], int rank, int nranks, pa_comm **comm);
int pa_comm_destroy(pa_comm *comm);
/* reduction(+,…;destination=:all) of device doubles (src/mpi_array.jl:494): in place, given stream */
int pa_comm_allreduce_sum(pa_comm *comm, void *device_ptr, int64_t count, int which_stream);
int pa_comm_barrier(pa_comm *comm);
/* What the communicator itself reports (ncclCommUserRank / ncclCommCount): MPI.Comm_rank / Comm_size,
 * src/mpi_array.jl:51-53.  bench.py prints nranks as `rccl_ranks_seen`. */
int pa_comm_info(pa_comm *comm, int *rank, int *nranks);

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

#ifdef __cplusplus
}
#endif
#endif /* PA_HIP_H */
