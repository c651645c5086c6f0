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
 * TIERS.  This header is the CONTRACT: context, device vectors and BLAS-1, CSR blocks and pa_spmv, exchange plans
 * and transports, the operator level (pa_mul*), psparse on the device, the RCCL communicator, graphs.  Everything else the
 * library exports -- arena and memory classes, introspection, testing aids, the HPCG multigrid pieces, SELL, the pa_host_* CPU
 * helpers -- is declared in pa_hip_experimental.h and may change between rounds.  Environment switches: INTEGRATION.md section 6.
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
/* y_seg = beta*y_seg + alpha*A*x_seg.
 *   spmv!(b,A,x)            (src/sparse_utils.jl:617-623,649-669)  <=> alpha=1, beta=0
 *   muladd!(b,A,x)          (src/p_sparse_matrix.jl:2088)            <=> alpha=1, beta=1
 *   mul!(b,A,x,alpha,beta)  (SparseMatricesCSR 0.6)                  <=> general
 * Launched on the compute stream.  x and y must not alias. */
int pa_spmv(const pa_csr *A, const pa_vec *x, int x_segment, pa_vec *y, int y_segment, double alpha,
            double beta);

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
/* pa_exchange_finish of every part behind pa_exchange_push_local in one call (consistent!: one unpack launch for all parts) */
int pa_exchange_finish_all(pa_plan *const *plans, int32_t n_parts, pa_vec *const *v, int mode);
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
/* The cache of psparse(I,J,V,rows,cols; reuse=true) (src/p_sparse_matrix.jl:1150-1219; psparse! :1291-1305) built on the device.
 * While pa_coo_keep_input_slots is on, pa_coo_subassemble / pa_coo_assemble_finish remember where every input triplet went (the K
 * of sparse_matrix!, composed with the split).  pa_coo_reuse_scatter composes the two into the scatter of a part's COO values
 * into W = [nonzeros(own_own) | nonzeros(own_ghost) || nonzeros(ghost_own) | nonzeros(ghost_ghost)]; ghost_slot[k] (host): the
 * 0-based position in the last two of ghost-row entry k as pa_coo_subassembly_ghost_rows lists them; k_rcv[q] (host, out): the
 * 1-based slot in W of the q-th triplet pa_coo_assemble_finish received = idx_rcv of the plan that assembles W. */
int pa_coo_keep_input_slots(pa_ctx *ctx, int on);
int pa_coo_reuse_scatter(const pa_coo_assembly *sub, const pa_coo_assembly *fin, const int32_t *ghost_slot, pa_scatter **out,
                         int64_t n_rcv, int32_t *k_rcv);

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

/* ---- HPCG set-up without the big upload (round 5: in the contract, the Julia glue's hpcg_blocks_hip calls it) ---- */
/* HPCG's 27-point operator of one part (HPCG/src/sparse_matrix.jl:28-122), own_own block and right-hand side, generated
 * in HBM: the arrays pa_host_hpcg_split_csr writes (oo_*, b), no host copy, no upload.  nx,ny,nz: the part's box; gnx,gny,gnz:
 * the global grid; gix0,giy0,giz0: global coordinates (1-based) of the part's first node.  b may be NULL. */
int pa_hpcg_own_block_create(pa_ctx *ctx, int64_t nx, int64_t ny, int64_t nz, int64_t gnx, int64_t gny, int64_t gnz,
                             int64_t gix0, int64_t giy0, int64_t giz0, pa_csr **own_own, pa_vec *b);
/* b alone (a vector created after the block is placed knowing the matrix streams' memory class) */
int pa_hpcg_rhs(pa_ctx *ctx, int64_t nx, int64_t ny, int64_t nz, int64_t gnx, int64_t gny, int64_t gnz, int64_t gix0,
                int64_t giy0, int64_t giz0, pa_vec *b);


#ifdef __cplusplus
}
#endif
#endif /* PA_HIP_H */
