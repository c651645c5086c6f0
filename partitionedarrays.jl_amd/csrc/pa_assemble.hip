// pa_assemble.hip -- psparse(I,J,V,rows,cols;assembled=true) of ONE part on the device ("device-side
// first-time set-up").
//
// The reference's route for a matrix given as COO triplets in global ids (src/p_sparse_matrix.jl:1249-1270, the route of
// HPCG.build_p_matrix and test/gallery_tests.jl:33):
//     cols = union_ghost(rows, J, find_owner(rows, J))        src/p_range.jl:205-259,346-348   new ghosts in first-seen order
//     map_global_to_local!(I, rows); map_global_to_local!(J, cols)                              :1253-1254
//     compresscoo(SparseMatrixCSR{1,Float64,Int32}, I, J, V, m, n; combine = +, skip)           src/sparse_utils.jl:313-350
//     split_format_locally(A, rows, cols)                                                        src/p_sparse_matrix.jl:823-899
// restated in pa_host.cpp as single-threaded loops (4 s for the 7-point 256^3 part of BASELINE config 2).  Here the triplets
// are uploaded once and everything per entry is a kernel, a scan or a radix sort (rocPRIM):
//     own-box arithmetic for rows and columns (block partitions: the own ids are a box of the global grid)
//     ghosts: the non-own column gids with their positions, sorted by gid (stable) -> first occurrence of each -> sorted by
//             that position = the first-seen order of filter_ghost; every occurrence then knows its ghost number
//     key = (local row, local column), stable radix sort -> runs of equal key are summed left to right IN INPUT ORDER
//             (compresscoo's combine = +; an entry with a row or column id < 1 becomes (1,1,0.0): the CSR skip rule)
//     own | ghost split by column, row pointers by binary search in the sorted rows
// The blocks go straight into pa_csr objects (row split on the host from the row pointers, column encodings on the device,
// pa_setup.hip); the host gets the ghost gids (a few thousand) and, when it asks, copies of the CSR arrays.
// Bit-identical to the host route: tests/test_gpu_setup.py::test_device_side_psparse_equals_the_host_route.
#include "pa_dev_util.h"

#include <chrono>

#include "pa_setup.h"

using namespace pa_util;

struct pa_box {                     // own ids of a block partition: a box of the global grid, column-major (src/p_range.jl:1471-1500)
  int D;
  long long n[8], lo[8], hi[8];
};

// 0-based own-local id of global id g (1-based), or -1 outside the box, -2 when g is no id at all (< 1 or > prod n)
__device__ __forceinline__ long long own_local(const pa_box &b, long long g) {
  if (g < 1) return -2;
  long long r = g - 1, own = 0, stride = 1;
  bool inside = true;
#pragma unroll 1
  for (int d = 0; d < b.D; ++d) {
    const long long c = r % b.n[d] + 1;
    r /= b.n[d];
    if (c < b.lo[d] || c > b.hi[d]) inside = false;
    own += (c - b.lo[d]) * stride;
    stride *= (b.hi[d] - b.lo[d] + 1);
  }
  if (r != 0) return -2;            // beyond the global grid
  return inside ? own : -1;
}

// li[e] = local row (0-based) or -1; lj[e] = own local column, -1 (no id), or -2 (a valid id outside the own box: a ghost)
__global__ void ka_classify(const long long *__restrict__ I, const long long *__restrict__ J, int n, pa_box rows, pa_box cols,
                            int *__restrict__ li, int *__restrict__ lj, int *__restrict__ is_ghost, int *__restrict__ is_rghost) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  const long long a = own_local(rows, I[e]);
  // rows that are not own are not local at all when the row partition has no ghosts (assembled = true); the sub-assembled
  // route (is_rghost != NULL) keeps them as ghost rows, numbered in first-seen order like the ghost columns
  li[e] = a >= 0 ? (int)a : (a == -1 && is_rghost ? -2 : -1);
  if (is_rghost) is_rghost[e] = a == -1 ? 1 : 0;
  const long long b = own_local(cols, J[e]);
  lj[e] = b >= 0 ? (int)b : (b == -1 ? -2 : -1);
  is_ghost[e] = b == -1 ? 1 : 0;
}

// occurrences of ghost columns: (gid, position).  The n_known ghosts the column partition already has come first, as if
// they had been seen before every entry (union_ghost appends to them).
__global__ void ka_ghost_occ(const long long *__restrict__ J, const int *__restrict__ is_ghost, const int *__restrict__ gscan, int n,
                             int n_known, unsigned long long *__restrict__ gid, int *__restrict__ pos) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n || !is_ghost[e]) return;
  const int k = n_known + gscan[e];               // exclusive scan
  gid[k] = (unsigned long long)J[e];
  pos[k] = k;
}
__global__ void ka_known_occ(const long long *__restrict__ known, int n_known, unsigned long long *__restrict__ gid, int *__restrict__ pos) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n_known) { gid[k] = (unsigned long long)known[k]; pos[k] = k; }
}

__global__ void ka_heads_u64(const unsigned long long *__restrict__ key, int n, int *__restrict__ head) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) head[i] = (i == 0 || key[i] != key[i - 1]) ? 1 : 0;
}

// unique ghost u (in gid order): first position it was seen at, and its gid
__global__ void ka_unique(const unsigned long long *__restrict__ sgid, const int *__restrict__ spos, const int *__restrict__ head,
                          const int *__restrict__ uscan, int n, int *__restrict__ first_pos, int *__restrict__ uid,
                          long long *__restrict__ ugid) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !head[i]) return;
  const int u = uscan[i] - 1;
  first_pos[u] = spos[i];           // stable sort: the smallest position of the run
  uid[u] = u;
  ugid[u] = (long long)sgid[i];
}

// after sorting the uniques by first position: rank[u] = ghost number (0-based), ghost_gid[rank] = gid
__global__ void ka_rank(const int *__restrict__ uid_sorted, const long long *__restrict__ ugid, int n_unique, int *__restrict__ rank,
                        long long *__restrict__ ghost_gid) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_unique) return;
  rank[uid_sorted[k]] = k;
  ghost_gid[k] = ugid[uid_sorted[k]];
}

// every occurrence learns its ghost number; occ_rank[position] (positions < n_known are the known ghosts themselves)
__global__ void ka_occ_rank(const int *__restrict__ spos, const int *__restrict__ uscan, const int *__restrict__ rank, int n,
                            int *__restrict__ occ_rank) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) occ_rank[spos[i]] = rank[uscan[i] - 1];
}

// key = (row << 32 | column) of the local matrix [own | ghost]; bad entries (a row or column that is not local) become
// (0, 0) with value 0.0 -- compresscoo's skip rule for CSR (src/sparse_utils.jl:330-342,370-390)
__global__ void ka_keys(const int *__restrict__ li, const int *__restrict__ lj, const int *__restrict__ is_ghost,
                        const int *__restrict__ gscan, const int *__restrict__ occ_rank, int n, int n_known, int n_own_cols,
                        int n_ghost_max, int discover, unsigned long long *__restrict__ key, int *__restrict__ idx,
                        unsigned char *__restrict__ bad, const int *__restrict__ is_rghost, const int *__restrict__ rgscan,
                        const int *__restrict__ rocc_rank, int n_own_rows) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  int r = li[e], c = lj[e];
  if (is_rghost && is_rghost[e]) r = n_own_rows + rocc_rank[rgscan[e]];
  if (is_ghost[e]) {
    const int k = occ_rank[n_known + gscan[e]];
    c = (discover || k < n_known) ? n_own_cols + k : -1;       // (a column the given partition does not know: not local)
  }
  (void)n_ghost_max;
  const bool b = r < 0 || c < 0;
  bad[e] = b ? 1 : 0;
  key[e] = b ? 0ull : ((unsigned long long)(unsigned)r << 32) | (unsigned)c;
  idx[e] = e;
}

// one lane per run of equal key: the run's values added left to right in input order (stable sort)
__global__ void ka_combine(const unsigned long long *__restrict__ skey, const int *__restrict__ sidx, const int *__restrict__ head,
                           const int *__restrict__ hscan, const double *__restrict__ V, const unsigned char *__restrict__ bad, int n,
                           int *__restrict__ orow, int *__restrict__ ocol, double *__restrict__ oval) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n || !head[i]) return;
  const int o = hscan[i] - 1;
  const unsigned long long k = skey[i];
  double acc = bad[sidx[i]] ? 0.0 : V[sidx[i]];
  for (int j = i + 1; j < n && skey[j] == k; ++j) acc = acc + (bad[sidx[j]] ? 0.0 : V[sidx[j]]);
  orow[o] = (int)(k >> 32);
  ocol[o] = (int)(k & 0xffffffffu);
  oval[o] = acc;
}

__global__ void ka_is_own_col(const int *__restrict__ ocol, int n, int n_own_cols, int *__restrict__ f) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) f[i] = ocol[i] < n_own_cols ? 1 : 0;
}

__global__ void ka_split(const int *__restrict__ ocol, const double *__restrict__ oval, const int *__restrict__ f,
                         const int *__restrict__ fscan, int n, int n_own_cols, int *__restrict__ oo_col, double *__restrict__ oo_val,
                         int *__restrict__ oh_col, double *__restrict__ oh_val) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int a = fscan[i];                          // exclusive: own-column entries before i
  if (f[i]) { oo_col[a] = ocol[i]; oo_val[a] = oval[i]; }
  else { oh_col[i - a] = ocol[i] - n_own_cols; oh_val[i - a] = oval[i]; }
}

// row pointers of both blocks: first output entry of row r by binary search in the sorted rows
__global__ void ka_rowptr(const int *__restrict__ orow, const int *__restrict__ fscan, int n, int n_rows, int n_oo,
                          int *__restrict__ oo_rp, int *__restrict__ oh_rp) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r > n_rows) return;
  int lo = 0, hi = n;
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (orow[mid] < r) lo = mid + 1; else hi = mid;
  }
  const int a = lo < n ? fscan[lo] : n_oo;
  oo_rp[r] = a;
  oh_rp[r] = lo - a;
}

// input triplet sidx[i] went to output entry hscan[i] - 1 (f == NULL: that index itself; else its position in [own|own, own|ghost])
__global__ void ka_in_slot(const int *__restrict__ sidx, const int *__restrict__ hscan, const unsigned char *__restrict__ bad, int n,
                           const int *__restrict__ f, const int *__restrict__ fscan, int nnz_oo, int *__restrict__ slot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int e = sidx[i], o = hscan[i] - 1;
  slot[e] = bad[e] ? -1 : !f ? o : f[o] ? fscan[o] : nnz_oo + (o - fscan[o]);
}

struct pa_coo_assembly {
  pa_ctx *ctx = nullptr;
  int64_t n_rows = 0, n_own_cols = 0, n_ghost = 0, n_known = 0, nnz_oo = 0, nnz_oh = 0;
  std::vector<int64_t> ghost_gids;                 // all ghosts of the column partition: the known ones, then the new ones
  int *oo_rp = nullptr, *oo_col = nullptr, *oh_rp = nullptr, *oh_col = nullptr;
  double *oo_val = nullptr, *oh_val = nullptr;
  double ms = 0;
  // the sub-assembled route (pa_coo_subassemble): ghost rows in first-seen order; the compressed local matrix as sorted
  // (row, column, value) entries -- the own rows' prefix stays in HBM for pa_coo_assemble_finish, the ghost rows' suffix (the
  // part's surface) is on the host for the exchange with the rows' owners
  bool sub = false;
  pa_box rows_box{}, cols_box{};
  std::vector<int64_t> row_ghost_gids;
  int *s_row = nullptr, *s_col = nullptr;
  double *s_val = nullptr;
  int64_t n_prefix = 0;
  std::vector<int32_t> g_row, g_col;               // suffix: 0-based ghost row, 0-based local column (own, then ghosts)
  std::vector<double> g_val;
  // pa_coo_keep_input_slots: where every input triplet went (the reference's K of sparse_matrix!, src/sparse_utils.jl:454-466):
  // sub-assembly -- the index of its entry in the sorted compressed list (own rows' prefix, then the ghost rows' suffix);
  // assembled -- the 0-based position of its entry in [nonzeros(own_own) | nonzeros(own_ghost)]; -1: the entry is skipped
  int *d_in_slot = nullptr;
  int64_t n_in = 0;
};

static void assembly_free(pa_coo_assembly *h) {
  if (!h) return;
  for (void *p : {(void *)h->oo_rp, (void *)h->oo_col, (void *)h->oh_rp, (void *)h->oh_col, (void *)h->oo_val, (void *)h->oh_val,
                  (void *)h->s_row, (void *)h->s_col, (void *)h->s_val, (void *)h->d_in_slot}) (void)hipFree(p);
  delete h;
}

// ids (gids of the entries flagged in isg, in input order, after n_known known ones) -> every occurrence's number in
// first-seen order (occ_rank[n_known + running index]) and, when wanted, the list of distinct gids in that order
static int number_first_seen(scratch &sc, hipStream_t s, const long long *dIDs, const int *isg, const int *gscan, int count, int n_occ,
                             int64_t n_known, const int64_t *known, bool want_list, int **occ_rank_out, std::vector<int64_t> &gids,
                             int *n_unique_out) {
  const int n_all = (int)n_known + n_occ;
  int *occ_rank = nullptr;
  PA_TRY(sc.get(&occ_rank, (size_t)n_all + 1));
  *occ_rank_out = occ_rank;
  *n_unique_out = (int)n_known;
  if (n_all == 0) return PA_OK;
  unsigned long long *gid = nullptr, *sgid = nullptr;
  int *pos = nullptr, *spos = nullptr, *head = nullptr, *uscan = nullptr;
  long long *dknown = nullptr;
  PA_TRY(sc.get(&gid, n_all));
  PA_TRY(sc.get(&sgid, n_all));
  PA_TRY(sc.get(&pos, n_all));
  PA_TRY(sc.get(&spos, n_all));
  PA_TRY(sc.get(&head, n_all));
  PA_TRY(sc.get(&uscan, n_all));
  if (n_known) {
    PA_TRY(sc.get(&dknown, n_known));
    PA_HIP(hipMemcpyAsync(dknown, known, 8 * (size_t)n_known, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(ka_known_occ, grid1(n_known), dim3(256), 0, s, dknown, (int)n_known, gid, pos);
  }
  if (count) hipLaunchKernelGGL(ka_ghost_occ, grid1(count), dim3(256), 0, s, dIDs, isg, gscan, count, (int)n_known, gid, pos);
  PA_TRY(sort_pairs<unsigned long long>(sc, s, gid, sgid, pos, spos, (size_t)n_all));
  hipLaunchKernelGGL(ka_heads_u64, grid1(n_all), dim3(256), 0, s, sgid, n_all, head);
  PA_TRY(scan_inclusive(sc, s, head, uscan, (size_t)n_all));
  int n_unique = 0;
  PA_TRY(d2h(s, &n_unique, uscan + (n_all - 1), 1));
  int *first_pos = nullptr, *uid = nullptr, *first_pos_s = nullptr, *uid_s = nullptr, *rank = nullptr;
  long long *ugid = nullptr, *ghost_gid = nullptr;
  PA_TRY(sc.get(&first_pos, n_unique));
  PA_TRY(sc.get(&uid, n_unique));
  PA_TRY(sc.get(&first_pos_s, n_unique));
  PA_TRY(sc.get(&uid_s, n_unique));
  PA_TRY(sc.get(&rank, n_unique));
  PA_TRY(sc.get(&ugid, n_unique));
  PA_TRY(sc.get(&ghost_gid, n_unique));
  hipLaunchKernelGGL(ka_unique, grid1(n_all), dim3(256), 0, s, sgid, spos, head, uscan, n_all, first_pos, uid, ugid);
  PA_TRY(sort_pairs<int>(sc, s, first_pos, first_pos_s, uid, uid_s, (size_t)n_unique));
  hipLaunchKernelGGL(ka_rank, grid1(n_unique), dim3(256), 0, s, uid_s, ugid, n_unique, rank, ghost_gid);
  hipLaunchKernelGGL(ka_occ_rank, grid1(n_all), dim3(256), 0, s, spos, uscan, rank, n_all, occ_rank);
  PA_HIP(hipGetLastError());
  // (the known ghosts must be distinct gids: then they keep the numbers 0 .. n_known-1 they came with)
  PA_REQUIRE(n_unique >= n_known, "the column partition lists a ghost twice");
  *n_unique_out = n_unique;
  if (want_list) {
    gids.resize(n_unique);
    PA_TRY(d2h(s, (long long *)gids.data(), ghost_gid, (size_t)n_unique));
    for (int64_t k = 0; k < n_known; ++k) PA_REQUIRE(gids[k] == known[k], "the column partition lists a ghost twice (or an own id as a ghost)");
  }
  for (void *q : {(void *)gid, (void *)sgid, (void *)pos, (void *)spos, (void *)head, (void *)uscan, (void *)first_pos, (void *)uid,
                  (void *)first_pos_s, (void *)uid_s, (void *)rank, (void *)ugid, (void *)ghost_gid}) sc.release(q);
  return PA_OK;
}

// the triplets are on the device (dI, dJ, dV: `count` entries, owned by sc); sub: keep rows outside the own box as ghost rows
static int assemble_core(pa_ctx *c, pa_coo_assembly *h, scratch &sc, int64_t count, long long *dI, long long *dJ, double *dV,
                         const pa_box &rows, const pa_box &cols, int64_t n_known, const int64_t *known, int discover, bool sub) {
  hipStream_t s = c->s[0];
  const int n = (int)count;
  int *li = nullptr, *lj = nullptr, *isg = nullptr, *gscan = nullptr, *isr = nullptr, *rgscan = nullptr;
  PA_TRY(sc.get(&li, count));
  PA_TRY(sc.get(&lj, count));
  PA_TRY(sc.get(&isg, count + 1));
  PA_TRY(sc.get(&gscan, count + 1));
  PA_HIP(hipMemsetAsync(isg + count, 0, sizeof(int), s));
  if (sub) {
    PA_TRY(sc.get(&isr, count + 1));
    PA_TRY(sc.get(&rgscan, count + 1));
    PA_HIP(hipMemsetAsync(isr + count, 0, sizeof(int), s));
  }
  hipLaunchKernelGGL(ka_classify, grid1(count), dim3(256), 0, s, dI, dJ, n, rows, cols, li, lj, isg, isr);
  PA_TRY(scan_exclusive<int>(sc, s, isg, gscan, (size_t)count + 1));
  int n_occ = 0, n_rocc = 0;
  PA_TRY(d2h(s, &n_occ, gscan + count, 1));
  // ---- ghost rows (sub-assembled route) and ghost columns in first-seen order
  int *rocc_rank = nullptr, *occ_rank = nullptr;
  int n_ghost_rows = 0, n_unique = 0;
  if (sub) {
    PA_TRY(scan_exclusive<int>(sc, s, isr, rgscan, (size_t)count + 1));
    PA_TRY(d2h(s, &n_rocc, rgscan + count, 1));
    PA_TRY(number_first_seen(sc, s, dI, isr, rgscan, n, n_rocc, 0, nullptr, true, &rocc_rank, h->row_ghost_gids, &n_ghost_rows));
  }
  sc.release(dI);
  h->n_known = n_known;
  h->ghost_gids.assign(known, known + n_known);
  PA_TRY(number_first_seen(sc, s, dJ, isg, gscan, n, n_occ, n_known, known, discover != 0, &occ_rank, h->ghost_gids, &n_unique));
  h->n_ghost = (int64_t)h->ghost_gids.size();
  // ---- sort by (row, column), combine
  unsigned long long *key = nullptr, *skey = nullptr;
  int *idx = nullptr, *sidx = nullptr;
  unsigned char *bad = nullptr;
  PA_TRY(sc.get(&key, count));
  PA_TRY(sc.get(&skey, count));
  PA_TRY(sc.get(&idx, count));
  PA_TRY(sc.get(&sidx, count));
  PA_TRY(sc.get(&bad, count));
  hipLaunchKernelGGL(ka_keys, grid1(count), dim3(256), 0, s, li, lj, isg, gscan, occ_rank, n, (int)n_known, (int)h->n_own_cols,
                     (int)h->n_ghost, discover, key, idx, bad, isr, rgscan, rocc_rank, (int)h->n_rows);
  unsigned bits_r = 1, bits = 0;
  while (((int64_t)1 << bits_r) < std::max<int64_t>(h->n_rows + n_ghost_rows, 2)) ++bits_r;
  bits = 32 + bits_r;
  PA_TRY(sort_pairs<unsigned long long>(sc, s, key, skey, idx, sidx, (size_t)count, bits));
  sc.release(key); sc.release(idx); sc.release(dJ); sc.release(li); sc.release(lj);
  int *head = nullptr, *hscan = nullptr;
  PA_TRY(sc.get(&head, count));
  PA_TRY(sc.get(&hscan, count));
  hipLaunchKernelGGL(ka_heads_u64, grid1(count), dim3(256), 0, s, skey, n, head);
  PA_TRY(scan_inclusive(sc, s, head, hscan, (size_t)count));
  int nnz = 0;
  PA_TRY(d2h(s, &nnz, hscan + (count - 1), 1));
  int *orow = nullptr, *ocol = nullptr, *f = nullptr, *fscan = nullptr;
  double *oval = nullptr;
  const size_t pad = 8;
  if (sub) {                                       // (kept with the handle)
    PA_HIP(hipMalloc((void **)&h->s_row, sizeof(int) * ((size_t)nnz + pad)));
    PA_HIP(hipMalloc((void **)&h->s_col, sizeof(int) * ((size_t)nnz + pad)));
    PA_HIP(hipMalloc((void **)&h->s_val, sizeof(double) * ((size_t)nnz + pad)));
    orow = h->s_row; ocol = h->s_col; oval = h->s_val;
  } else {
    PA_TRY(sc.get(&orow, nnz));
    PA_TRY(sc.get(&ocol, nnz));
    PA_TRY(sc.get(&oval, nnz));
  }
  PA_TRY(sc.get(&f, (size_t)nnz + 1));
  PA_TRY(sc.get(&fscan, (size_t)nnz + 1));
  hipLaunchKernelGGL(ka_combine, grid1(count), dim3(256), 0, s, skey, sidx, head, hscan, dV, bad, n, orow, ocol, oval);
  PA_HIP(hipMemsetAsync(f + nnz, 0, sizeof(int), s));
  hipLaunchKernelGGL(ka_is_own_col, grid1(nnz), dim3(256), 0, s, ocol, nnz, (int)h->n_own_cols, f);
  PA_TRY(scan_exclusive<int>(sc, s, f, fscan, (size_t)nnz + 1));
  PA_HIP(hipMalloc((void **)&h->oo_rp, sizeof(int) * (size_t)(h->n_rows + 1)));
  PA_HIP(hipMalloc((void **)&h->oh_rp, sizeof(int) * (size_t)(h->n_rows + 1)));
  // row pointers of the own rows first: with ghost rows behind them, the blocks end where the own rows do
  int n_oo_all = 0;
  PA_TRY(d2h(s, &n_oo_all, fscan + nnz, 1));
  hipLaunchKernelGGL(ka_rowptr, grid1(h->n_rows + 1), dim3(256), 0, s, orow, fscan, nnz, (int)h->n_rows, n_oo_all, h->oo_rp, h->oh_rp);
  int end_oo = 0, end_oh = 0;
  PA_TRY(d2h(s, &end_oo, h->oo_rp + h->n_rows, 1));
  PA_TRY(d2h(s, &end_oh, h->oh_rp + h->n_rows, 1));
  h->nnz_oo = end_oo; h->nnz_oh = end_oh;
  h->n_prefix = (int64_t)end_oo + end_oh;
  PA_REQUIRE(sub || h->n_prefix == nnz, "entries beyond the own rows in an assembled matrix");
  if (c->keep_coo_slots) {
    PA_HIP(hipMalloc((void **)&h->d_in_slot, sizeof(int) * ((size_t)count + 8)));
    h->n_in = count;
    hipLaunchKernelGGL(ka_in_slot, grid1(count), dim3(256), 0, s, sidx, hscan, bad, n, sub ? (const int *)nullptr : f, fscan, end_oo, h->d_in_slot);
  }
  if (sub) {
    // the ghost rows' entries (the surface of the part) go to the host; the own rows' stay where they are
    const int64_t ng = (int64_t)nnz - h->n_prefix;
    h->g_row.resize(ng); h->g_col.resize(ng); h->g_val.resize(ng);
    if (ng) {
      PA_TRY(d2h(s, h->g_row.data(), orow + h->n_prefix, (size_t)ng));
      PA_TRY(d2h(s, h->g_col.data(), ocol + h->n_prefix, (size_t)ng));
      PA_TRY(d2h(s, h->g_val.data(), oval + h->n_prefix, (size_t)ng));
      for (int64_t k = 0; k < ng; ++k) h->g_row[k] -= (int32_t)h->n_rows;
    }
    PA_HIP(hipGetLastError());
    PA_HIP(hipStreamSynchronize(s));
    return PA_OK;
  }
  PA_HIP(hipMalloc((void **)&h->oo_col, sizeof(int) * ((size_t)h->nnz_oo + pad)));
  PA_HIP(hipMalloc((void **)&h->oh_col, sizeof(int) * ((size_t)h->nnz_oh + pad)));
  PA_HIP(hipMalloc((void **)&h->oo_val, sizeof(double) * ((size_t)h->nnz_oo + pad)));
  PA_HIP(hipMalloc((void **)&h->oh_val, sizeof(double) * ((size_t)h->nnz_oh + pad)));
  hipLaunchKernelGGL(ka_split, grid1(nnz), dim3(256), 0, s, ocol, oval, f, fscan, nnz, (int)h->n_own_cols, h->oo_col, h->oo_val, h->oh_col, h->oh_val);
  PA_HIP(hipGetLastError());
  PA_HIP(hipStreamSynchronize(s));
  return PA_OK;
}

static int assemble_impl(pa_ctx *c, pa_coo_assembly *h, int64_t count, const int64_t *I, const int64_t *J, const double *V,
                         const pa_box &rows, const pa_box &cols, int64_t n_known, const int64_t *known, int discover, bool sub) {
  hipStream_t s = c->s[0];
  scratch sc;
  long long *dI = nullptr, *dJ = nullptr;
  double *dV = nullptr;
  PA_TRY(sc.get(&dI, count));
  PA_TRY(sc.get(&dJ, count));
  PA_TRY(sc.get(&dV, count));
  // (hipMemcpyDefault: the triplets may be host arrays or already in HBM -- pa_fem_triplets_device below)
  PA_HIP(hipMemcpyAsync(dI, I, 8 * (size_t)count, hipMemcpyDefault, s));
  PA_HIP(hipMemcpyAsync(dJ, J, 8 * (size_t)count, hipMemcpyDefault, s));
  PA_HIP(hipMemcpyAsync(dV, V, 8 * (size_t)count, hipMemcpyDefault, s));
  return assemble_core(c, h, sc, count, dI, dJ, dV, rows, cols, n_known, known, discover, sub);
}

static int make_box(pa_box &b, int32_t D, const int64_t *n, const int64_t *lo, const int64_t *hi, int64_t *n_own) {
  PA_REQUIRE(D >= 1 && D <= 8 && n && lo && hi, "bad box");
  b.D = D;
  *n_own = 1;
  for (int d = 0; d < D; ++d) {
    PA_REQUIRE(n[d] >= 1 && lo[d] >= 1 && hi[d] <= n[d] && hi[d] >= lo[d] - 1, "own range [%lld,%lld] outside 1:%lld in direction %d", (long long)lo[d], (long long)hi[d], (long long)n[d], d + 1);
    b.n[d] = n[d]; b.lo[d] = lo[d]; b.hi[d] = hi[d];
    *n_own *= (hi[d] - lo[d] + 1);
  }
  return PA_OK;
}

extern "C" int pa_coo_assemble(pa_ctx *c, int64_t count, const int64_t *I, const int64_t *J, const double *V, int32_t D,
                               const int64_t *n_rows_global, const int64_t *row_lo, const int64_t *row_hi,
                               const int64_t *n_cols_global, const int64_t *col_lo, const int64_t *col_hi, int64_t n_known_ghosts,
                               const int64_t *known_ghosts, int discover_ghosts, pa_coo_assembly **out) {
  PA_REQUIRE(c && out && count >= 0 && (count == 0 || (I && J && V)), "bad arguments");
  PA_REQUIRE(n_known_ghosts >= 0 && (n_known_ghosts == 0 || known_ghosts), "bad ghost list");
  PA_REQUIRE(count < (int64_t)2147480000 && n_known_ghosts < (int64_t)1 << 30, "more triplets than the device-side assembly indexes (2^31)");
  pa_box rows, cols;
  int64_t n_own_rows = 0, n_own_cols = 0;
  PA_TRY(make_box(rows, D, n_rows_global, row_lo, row_hi, &n_own_rows));
  PA_TRY(make_box(cols, D, n_cols_global, col_lo, col_hi, &n_own_cols));
  PA_REQUIRE(n_own_rows < (int64_t)2147480000 && n_own_cols + n_known_ghosts + count < (int64_t)2147480000, "part too large for Int32 local ids");
  PA_REQUIRE(count > 0 && n_own_rows > 0 && n_own_cols > 0, "the device-side assembly needs at least one triplet and one own row (the host route handles empty parts)");
  PA_HIP(hipSetDevice(c->device));
  pa_coo_assembly *h = new pa_coo_assembly();
  h->ctx = c; h->n_rows = n_own_rows; h->n_own_cols = n_own_cols;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0, c->s[0]);
  const int st = assemble_impl(c, h, count, I, J, V, rows, cols, n_known_ghosts, known_ghosts, discover_ghosts, false);
  (void)hipEventRecord(e1, c->s[0]);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  if (st != PA_OK) { (void)hipGetLastError(); assembly_free(h); return st; }
  h->ms = ms;
  *out = h;
  return PA_OK;
}

// ---- the disassembled route: psparse(I,J,V,rows,cols) with the default flags + assemble (src/p_sparse_matrix.jl:1150-1219,
// 1590-1756).  A part's triplets may name rows other parts own (FEM assembly loops).  Step 1 (pa_coo_subassemble): the local
// sub-assembled matrix -- ghost rows AND ghost columns numbered in first-seen order, one stable (row, column) sort, duplicates
// added in input order -- whose own rows stay in HBM while the ghost rows (the part's surface) go to the host, which sends them
// to their owners (the reference's setup_cache_snd + exchange).  Step 2 (pa_coo_assemble_finish): the own rows' entries, in
// the order the reference walks them (own|own and own|ghost in CSR order), followed by what the neighbours sent, through the
// assembled route above: the same left-to-right sums, the same first-seen order of the final ghost columns as
// setup_own_triplets + union_ghost + finalize_values (:1656-1723).
__device__ __forceinline__ long long own_global(const pa_box &b, long long id) {      // 0-based own id -> 1-based global id
  long long g = 0, stride = 1;
#pragma unroll 1
  for (int d = 0; d < b.D; ++d) {
    const long long ext = b.hi[d] - b.lo[d] + 1;
    const long long c = b.lo[d] + id % ext;
    id /= ext;
    g += (c - 1) * stride;
    stride *= b.n[d];
  }
  return g + 1;
}

__global__ void ka_expand(const int *__restrict__ srow, const int *__restrict__ scol, const double *__restrict__ sval, int n, pa_box rows,
                          pa_box cols, int n_own_cols, const long long *__restrict__ ghost_gid, long long *__restrict__ I,
                          long long *__restrict__ J, double *__restrict__ V) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= n) return;
  I[e] = own_global(rows, srow[e]);
  J[e] = scol[e] < n_own_cols ? own_global(cols, scol[e]) : ghost_gid[scol[e] - n_own_cols];
  V[e] = sval[e];
}

extern "C" int pa_coo_subassemble(pa_ctx *c, int64_t count, const int64_t *I, const int64_t *J, const double *V, int32_t D,
                                  const int64_t *n_rows_global, const int64_t *row_lo, const int64_t *row_hi,
                                  const int64_t *n_cols_global, const int64_t *col_lo, const int64_t *col_hi, pa_coo_assembly **out) {
  PA_REQUIRE(c && out && count > 0 && I && J && V, "bad arguments");
  PA_REQUIRE(count < (int64_t)2147480000, "more triplets than the device-side assembly indexes (2^31)");
  pa_box rows, cols;
  int64_t n_own_rows = 0, n_own_cols = 0;
  PA_TRY(make_box(rows, D, n_rows_global, row_lo, row_hi, &n_own_rows));
  PA_TRY(make_box(cols, D, n_cols_global, col_lo, col_hi, &n_own_cols));
  PA_REQUIRE(n_own_rows > 0 && n_own_cols > 0 && n_own_rows + count < (int64_t)2147480000 && n_own_cols + count < (int64_t)2147480000,
             "part empty or too large for Int32 local ids");
  PA_HIP(hipSetDevice(c->device));
  pa_coo_assembly *h = new pa_coo_assembly();
  h->ctx = c; h->n_rows = n_own_rows; h->n_own_cols = n_own_cols; h->sub = true; h->rows_box = rows; h->cols_box = cols;
  const auto t0 = std::chrono::steady_clock::now();
  const int st = assemble_impl(c, h, count, I, J, V, rows, cols, 0, nullptr, 1, true);
  if (st != PA_OK) { (void)hipGetLastError(); assembly_free(h); return st; }
  h->ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  *out = h;
  return PA_OK;
}

extern "C" int pa_coo_subassembly_info(const pa_coo_assembly *h, int64_t *n_ghost_rows, int64_t *n_ghost_cols, int64_t *n_own_entries,
                                       int64_t *n_ghost_row_entries) {
  PA_REQUIRE(h && h->sub, "not a sub-assembly");
  if (n_ghost_rows) *n_ghost_rows = (int64_t)h->row_ghost_gids.size();
  if (n_ghost_cols) *n_ghost_cols = h->n_ghost;
  if (n_own_entries) *n_own_entries = h->n_prefix;
  if (n_ghost_row_entries) *n_ghost_row_entries = (int64_t)h->g_val.size();
  return PA_OK;
}

// the ghost rows: their gids in first-seen order, and their entries sorted by (row, column): 0-based ghost row, 0-based local
// column (own columns, then the ghost columns pa_coo_assembly_ghosts lists), value
extern "C" int pa_coo_subassembly_ghost_rows(const pa_coo_assembly *h, int64_t *row_gids, int32_t *g_row, int32_t *g_col, double *g_val) {
  PA_REQUIRE(h && h->sub, "not a sub-assembly");
  for (size_t k = 0; k < h->row_ghost_gids.size(); ++k) row_gids[k] = h->row_ghost_gids[k];
  for (size_t k = 0; k < h->g_val.size(); ++k) { g_row[k] = h->g_row[k]; g_col[k] = h->g_col[k]; g_val[k] = h->g_val[k]; }
  return PA_OK;
}

extern "C" int pa_coo_assemble_finish(const pa_coo_assembly *sub, int64_t n_rcv, const int64_t *I, const int64_t *J, const double *V,
                                      pa_coo_assembly **out) {
  PA_REQUIRE(sub && sub->sub && out && n_rcv >= 0 && (n_rcv == 0 || (I && J && V)), "bad arguments");
  pa_ctx *c = sub->ctx;
  const int64_t count = sub->n_prefix + n_rcv;
  PA_REQUIRE(count > 0 && count < (int64_t)2147480000, "no entries, or more than the device-side assembly indexes (2^31)");
  PA_HIP(hipSetDevice(c->device));
  hipStream_t s = c->s[0];
  pa_coo_assembly *h = new pa_coo_assembly();
  h->ctx = c; h->n_rows = sub->n_rows; h->n_own_cols = sub->n_own_cols;
  const auto t0 = std::chrono::steady_clock::now();
  int st = PA_OK;
  {
    scratch sc;
    long long *dI = nullptr, *dJ = nullptr, *dG = nullptr;
    double *dV = nullptr;
    auto run = [&]() -> int {
      PA_TRY(sc.get(&dI, count));
      PA_TRY(sc.get(&dJ, count));
      PA_TRY(sc.get(&dV, count));
      PA_TRY(sc.get(&dG, sub->ghost_gids.size() + 1));
      if (!sub->ghost_gids.empty())
        PA_HIP(hipMemcpyAsync(dG, sub->ghost_gids.data(), 8 * sub->ghost_gids.size(), hipMemcpyHostToDevice, s));
      if (sub->n_prefix)
        hipLaunchKernelGGL(ka_expand, grid1(sub->n_prefix), dim3(256), 0, s, sub->s_row, sub->s_col, sub->s_val, (int)sub->n_prefix,
                           sub->rows_box, sub->cols_box, (int)sub->n_own_cols, dG, dI, dJ, dV);
      if (n_rcv) {
        PA_HIP(hipMemcpyAsync(dI + sub->n_prefix, I, 8 * (size_t)n_rcv, hipMemcpyHostToDevice, s));
        PA_HIP(hipMemcpyAsync(dJ + sub->n_prefix, J, 8 * (size_t)n_rcv, hipMemcpyHostToDevice, s));
        PA_HIP(hipMemcpyAsync(dV + sub->n_prefix, V, 8 * (size_t)n_rcv, hipMemcpyHostToDevice, s));
      }
      PA_HIP(hipGetLastError());
      PA_HIP(hipStreamSynchronize(s));
      sc.release(dG);
      return assemble_core(c, h, sc, count, dI, dJ, dV, sub->rows_box, sub->cols_box, 0, nullptr, 1, false);
    };
    st = run();
  }
  if (st != PA_OK) { (void)hipGetLastError(); assembly_free(h); return st; }
  h->ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  *out = h;
  return PA_OK;
}

extern "C" int pa_coo_assembly_info(const pa_coo_assembly *h, int64_t *n_own_rows, int64_t *n_own_cols, int64_t *n_ghost,
                                    int64_t *nnz_own_own, int64_t *nnz_own_ghost, double *device_ms) {
  PA_REQUIRE(h != nullptr, "assembly is NULL");
  if (n_own_rows) *n_own_rows = h->n_rows;
  if (n_own_cols) *n_own_cols = h->n_own_cols;
  if (n_ghost) *n_ghost = h->n_ghost;
  if (nnz_own_own) *nnz_own_own = h->nnz_oo;
  if (nnz_own_ghost) *nnz_own_ghost = h->nnz_oh;
  if (device_ms) *device_ms = h->ms;
  return PA_OK;
}

extern "C" int pa_coo_assembly_ghosts(const pa_coo_assembly *h, int64_t *ghost_gids) {
  PA_REQUIRE(h && (ghost_gids || h->n_ghost == 0), "bad arguments");
  for (int64_t k = 0; k < h->n_ghost; ++k) ghost_gids[k] = h->ghost_gids[k];
  return PA_OK;
}

extern "C" int pa_coo_assembly_blocks(pa_coo_assembly *h, pa_csr **own_own, pa_csr **own_ghost) {
  PA_REQUIRE(h && own_own && own_ghost, "bad arguments");
  pa_csr *a = nullptr, *b = nullptr;
  PA_TRY(pa_csr_from_device(h->ctx, h->n_rows, h->n_own_cols, h->nnz_oo, h->oo_rp, h->oo_col, h->oo_val, &a));
  const int st = pa_csr_from_device(h->ctx, h->n_rows, h->n_ghost, h->nnz_oh, h->oh_rp, h->oh_col, h->oh_val, &b);
  if (st != PA_OK) { (void)pa_csr_destroy(a); return st; }
  *own_own = a;
  *own_ghost = b;
  return PA_OK;
}

// 1-based host copies of one block (which: 0 own_own, 1 own_ghost), as SparseMatrixCSR{1,Float64,Int32} stores them
__global__ void ka_plus_one(const int *__restrict__ in, int64_t n, int *__restrict__ out) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) out[i] = in[i] + 1;
}
extern "C" int pa_coo_assembly_download(const pa_coo_assembly *h, int which, int32_t *rowptr, int32_t *colval, double *nzval) {
  PA_REQUIRE(h && rowptr && (which == 0 || which == 1), "bad arguments");
  pa_ctx *c = h->ctx;
  const int64_t nnz = which ? h->nnz_oh : h->nnz_oo;
  PA_REQUIRE(nnz == 0 || (colval && nzval), "colval / nzval are NULL");
  PA_HIP(hipSetDevice(c->device));
  scratch sc;
  int *t = nullptr;
  PA_TRY(sc.get(&t, (size_t)std::max<int64_t>(nnz, h->n_rows + 1)));
  hipLaunchKernelGGL(ka_plus_one, dim3(1024), dim3(256), 0, c->s[0], which ? h->oh_rp : h->oo_rp, h->n_rows + 1, t);
  PA_TRY(d2h(c->s[0], rowptr, t, (size_t)h->n_rows + 1));
  if (nnz) {
    hipLaunchKernelGGL(ka_plus_one, dim3(4096), dim3(256), 0, c->s[0], which ? h->oh_col : h->oo_col, nnz, t);
    PA_TRY(d2h(c->s[0], colval, t, (size_t)nnz));
    PA_TRY(d2h(c->s[0], nzval, which ? h->oh_val : h->oo_val, (size_t)nnz));
  }
  return PA_OK;
}

extern "C" int pa_coo_assembly_destroy(pa_coo_assembly *h) {
  if (!h) return PA_OK;
  (void)hipSetDevice(h->ctx->device);
  (void)hipStreamSynchronize(h->ctx->s[0]);
  assembly_free(h);
  return PA_OK;
}


// ---- the cache of psparse(...; reuse = true) built on the device (round 4; VERDICT r03 missing #6) -----------------------------
// psparse!(C, V, cache) (src/p_sparse_matrix.jl:1291-1305) is, on the device, one deterministic scatter-add of the new COO values
// into W = [nonzeros(C.own_own) | nonzeros(C.own_ghost) || nonzeros(B.ghost_own) | nonzeros(B.ghost_ghost)] followed by an
// assemble! of W (p_sparse_matrix.py, MatrixReassemblyCache).  The scatter's destinations are the reference's K
// (sparse_matrix! src/sparse_utils.jl:454-466, precompute_nzindex :213-254) composed with the split and the assembly; with
// pa_coo_keep_input_slots both assembly steps remember where their inputs went, and the composition is a kernel.
extern "C" int pa_coo_keep_input_slots(pa_ctx *c, int on) {
  PA_REQUIRE(c != nullptr, "context is NULL");
  c->keep_coo_slots = on != 0;
  return PA_OK;
}

__global__ void ka_compose_dest(const int *__restrict__ slot1, int n, int n_prefix, const int *__restrict__ slot2, int n_own_vals,
                                const int *__restrict__ ghost_slot, int *__restrict__ dest) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int a = slot1[p];
  dest[p] = a < 0 ? -1 : a < n_prefix ? slot2[a] : n_own_vals + ghost_slot[a - n_prefix];
}
__global__ void ka_dest_keys(const int *__restrict__ dest, int n, unsigned *__restrict__ key, int *__restrict__ idx) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) { key[p] = dest[p] < 0 ? 0xffffffffu : (unsigned)dest[p]; idx[p] = p; }
}
__global__ void ka_heads_u32(const unsigned *__restrict__ key, int n, int *__restrict__ head) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) head[i] = key[i] != 0xffffffffu && (i == 0 || key[i] != key[i - 1]) ? 1 : 0;
}
__global__ void ka_targets(const unsigned *__restrict__ skey, const int *__restrict__ head, const int *__restrict__ hscan, int n, int n_valid,
                           int n_tgt, int *__restrict__ tgt, int *__restrict__ tptr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) tptr[n_tgt] = n_valid;
  if (i < n && head[i]) { tgt[hscan[i] - 1] = (int)skey[i]; tptr[hscan[i] - 1] = i; }
}
__global__ void ka_count_valid(const unsigned *__restrict__ skey, int n, int *__restrict__ out) {     // first position of 0xffffffff
  if (blockIdx.x || threadIdx.x) return;
  int lo = 0, hi = n;
  while (lo < hi) { const int mid = (lo + hi) >> 1; if (skey[mid] != 0xffffffffu) lo = mid + 1; else hi = mid; }
  *out = lo;
}

// W[dest[p]] += src[p] in ascending p (pa_scatter_add), dest 0-based ON THE DEVICE, negative: skipped.  The grouping of the
// sources by destination -- a stable sort -- is a radix sort here (pa_scatter_create's host sort: 1.5 s for 17 M sources).
int pa_scatter_from_device_dest(pa_ctx *c, int64_t n_dst, int64_t n_src, const int32_t *d_dest, pa_scatter **out) {
  PA_REQUIRE(c && out && n_dst >= 0 && n_src >= 0 && n_src < (int64_t)2147480000, "bad arguments");
  PA_HIP(hipSetDevice(c->device));
  hipStream_t s = c->s[0];
  pa_scatter *sc_ = new pa_scatter();
  sc_->ctx = c; sc_->n_dst = n_dst; sc_->n_src = n_src;
  auto fail = [&](int st) { (void)pa_raw_free(sc_->d_tgt); (void)pa_raw_free(sc_->d_tptr); (void)pa_raw_free(sc_->d_tp); delete sc_; return st; };
  scratch sc;
  const int n = (int)n_src;
  int n_valid = 0, n_tgt = 0;
  unsigned *key = nullptr, *skey = nullptr;
  int *idx = nullptr, *sidx = nullptr, *head = nullptr, *hscan = nullptr, *d_nv = nullptr;
  auto run = [&]() -> int {
    if (n == 0) return PA_OK;
    PA_TRY(sc.get(&key, n_src)); PA_TRY(sc.get(&skey, n_src)); PA_TRY(sc.get(&idx, n_src)); PA_TRY(sc.get(&sidx, n_src));
    PA_TRY(sc.get(&head, n_src)); PA_TRY(sc.get(&hscan, n_src)); PA_TRY(sc.get(&d_nv, 1));
    hipLaunchKernelGGL(ka_dest_keys, grid1(n_src), dim3(256), 0, s, d_dest, n, key, idx);
    PA_TRY(sort_pairs<unsigned>(sc, s, key, skey, idx, sidx, (size_t)n_src));
    hipLaunchKernelGGL(ka_count_valid, dim3(1), dim3(1), 0, s, skey, n, d_nv);
    hipLaunchKernelGGL(ka_heads_u32, grid1(n_src), dim3(256), 0, s, skey, n, head);
    PA_TRY(scan_inclusive(sc, s, head, hscan, (size_t)n_src));
    PA_TRY(d2h(s, &n_valid, d_nv, 1));
    PA_TRY(d2h(s, &n_tgt, hscan + (n - 1), 1));
    if (n_valid > 0) {
      unsigned last = 0;
      PA_TRY(d2h(s, &last, skey + (n_valid - 1), 1));
      PA_REQUIRE((int64_t)last < n_dst, "destination %u out of range (%lld slots)", last, (long long)n_dst);
    }
    return PA_OK;
  };
  if (int st = run()) { (void)hipGetLastError(); return fail(st); }
  sc_->n_tgt = n_tgt;
  if (pa_raw_malloc(&sc_->d_tgt, sizeof(int32_t) * (size_t)std::max(1, n_tgt)) != hipSuccess ||
      pa_raw_malloc(&sc_->d_tptr, sizeof(int32_t) * ((size_t)n_tgt + 1)) != hipSuccess ||
      pa_raw_malloc(&sc_->d_tp, sizeof(int32_t) * (size_t)std::max(1, n_valid)) != hipSuccess) { pa_set_err("scatter: out of device memory"); return fail(PA_ERR_HIP); }
  if (n > 0) {
    hipLaunchKernelGGL(ka_targets, grid1(n_src), dim3(256), 0, s, skey, head, hscan, n, n_valid, n_tgt, sc_->d_tgt, sc_->d_tptr);
    if (n_valid && hipMemcpyAsync(sc_->d_tp, sidx, sizeof(int32_t) * (size_t)n_valid, hipMemcpyDeviceToDevice, s) != hipSuccess) return fail(PA_ERR_HIP);
  } else if (hipMemsetAsync(sc_->d_tptr, 0, sizeof(int32_t), s) != hipSuccess) return fail(PA_ERR_HIP);
  if (hipStreamSynchronize(s) != hipSuccess || hipGetLastError() != hipSuccess) { pa_set_err("scatter: a kernel failed"); return fail(PA_ERR_HIP); }
  *out = sc_;
  return PA_OK;
}

// sub / fin: the handles of pa_coo_subassemble and pa_coo_assemble_finish of one part, both made under pa_coo_keep_input_slots.
// ghost_slot[k] (host, one per ghost-row entry in the order pa_coo_subassembly_ghost_rows lists them): its 0-based position in
// [nonzeros(ghost_own) | nonzeros(ghost_ghost)].  Out: the scatter of the part's COO values into W, and k_rcv (1-based slots in W
// of the triplets given to pa_coo_assemble_finish, in that order): idx_rcv of the value exchange's plan.
extern "C" int pa_coo_reuse_scatter(const pa_coo_assembly *sub, const pa_coo_assembly *fin, const int32_t *ghost_slot, pa_scatter **out,
                                    int64_t n_rcv, int32_t *k_rcv) {
  PA_REQUIRE(sub && fin && out && sub->sub && !fin->sub, "bad arguments");
  PA_REQUIRE(sub->d_in_slot && fin->d_in_slot, "the assemblies were not made under pa_coo_keep_input_slots");
  PA_REQUIRE(fin->n_in == sub->n_prefix + n_rcv && n_rcv >= 0 && (n_rcv == 0 || k_rcv), "pa_coo_assemble_finish took %lld triplets, not %lld + %lld",
             (long long)fin->n_in, (long long)sub->n_prefix, (long long)n_rcv);
  pa_ctx *c = sub->ctx;
  PA_HIP(hipSetDevice(c->device));
  hipStream_t s = c->s[0];
  const int64_t n_ghost_vals = (int64_t)sub->g_val.size(), n_own_vals = fin->nnz_oo + fin->nnz_oh;
  PA_REQUIRE(n_ghost_vals == 0 || ghost_slot, "ghost_slot is NULL");
  scratch sc;
  int *d_gs = nullptr, *d_dest = nullptr;
  PA_TRY(sc.get(&d_gs, (size_t)n_ghost_vals + 1));
  PA_TRY(sc.get(&d_dest, (size_t)sub->n_in + 1));
  for (int64_t k = 0; k < n_ghost_vals; ++k) PA_REQUIRE(ghost_slot[k] >= 0 && ghost_slot[k] < n_ghost_vals, "ghost_slot[%lld] out of range", (long long)k);
  if (n_ghost_vals) PA_HIP(hipMemcpyAsync(d_gs, ghost_slot, sizeof(int) * (size_t)n_ghost_vals, hipMemcpyHostToDevice, s));
  hipLaunchKernelGGL(ka_compose_dest, grid1(sub->n_in), dim3(256), 0, s, sub->d_in_slot, (int)sub->n_in, (int)sub->n_prefix, fin->d_in_slot,
                     (int)n_own_vals, d_gs, d_dest);
  PA_HIP(hipGetLastError());
  PA_TRY(pa_scatter_from_device_dest(c, n_own_vals + n_ghost_vals, sub->n_in, d_dest, out));
  if (n_rcv) {
    std::vector<int32_t> k(n_rcv);
    PA_TRY(d2h(s, k.data(), fin->d_in_slot + sub->n_prefix, (size_t)n_rcv));
    for (int64_t q = 0; q < n_rcv; ++q) {
      PA_REQUIRE(k[q] >= 0, "a received triplet was skipped by the assembly");
      k_rcv[q] = k[q] + 1;
    }
  }
  return PA_OK;
}

// the destinations themselves (tests; 0-based, -1 = skipped)
extern "C" int pa_scatter_download(const pa_scatter *sc_, int32_t *dest) {
  PA_REQUIRE(sc_ && (sc_->n_src == 0 || dest), "bad arguments");
  pa_ctx *c = sc_->ctx;
  PA_HIP(hipSetDevice(c->device));
  std::vector<int32_t> tgt(sc_->n_tgt), tptr(sc_->n_tgt + 1), tp;
  for (int64_t p = 0; p < sc_->n_src; ++p) dest[p] = -1;
  if (sc_->n_tgt == 0) return PA_OK;
  PA_TRY(d2h(c->s[0], tgt.data(), sc_->d_tgt, tgt.size()));
  PA_TRY(d2h(c->s[0], tptr.data(), sc_->d_tptr, tptr.size()));
  tp.resize(tptr.back());
  PA_TRY(d2h(c->s[0], tp.data(), sc_->d_tp, tp.size()));
  for (int64_t t = 0; t < sc_->n_tgt; ++t)
    for (int32_t k = tptr[t]; k < tptr[t + 1]; ++k) {
      PA_REQUIRE(k == tptr[t] || tp[k] > tp[k - 1], "sources of slot %d not ascending", tgt[t]);
      dest[tp[k]] = tgt[t];
    }
  return PA_OK;
}


// ---- laplacian_fem's triplets generated in HBM (src/gallery.jl:110-239; VERDICT r05 "What's missing" 5) --------------------------
// The reference loops over a part's CELLS (column-major), and for every cell over its 2^D corners i and j, emitting
// (node(i), node(j), Aref[i,j]) when both corners are interior nodes -- the disassembled input of psparse.  pa_host_laplacian_fem
// (pa_host.cpp) is that loop on host threads; here a count kernel (per cell: (interior corners)^2), an exclusive scan and a fill
// kernel write the same triplets in the same order straight into HBM, where pa_coo_subassemble takes them from.
struct pa_fem_box { int D; long long nn[3], lo[3], hi[3], stride[3], len[3]; double Aref[64]; };

__device__ __forceinline__ void fem_cell(const pa_fem_box &b, long long q, long long *c) {
  for (int d = 0; d < 3; ++d) { c[d] = d < b.D ? b.lo[d] + q % b.len[d] : 1; if (d < b.D) q /= b.len[d]; }
}
__global__ void kf_count(const pa_fem_box b, long long n_cells, long long *__restrict__ cnt) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_cells) return;
  long long c[3];
  fem_cell(b, q, c);
  long long k = 1;
  for (int d = 0; d < b.D; ++d) {
    int kd = 0;
    for (int a = 0; a < 2; ++a) { const long long x = c[d] + a - 1; kd += x >= 1 && x <= b.nn[d]; }
    k *= kd;
  }
  cnt[q] = k * k;
}
__global__ void kf_fill(const pa_fem_box b, long long n_cells, const long long *__restrict__ off, long long *__restrict__ I,
                        long long *__restrict__ J, double *__restrict__ V) {
  const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= n_cells) return;
  long long c[3];
  fem_cell(b, q, c);
  const int nloc = 1 << b.D;
  long long id[8];
  bool in[8];
  for (int a = 0; a < nloc; ++a) {                       // corner a: offsets column-major (first direction fastest)
    bool good = true;
    long long node = 1;
    for (int d = 0; d < b.D; ++d) {
      const long long x = c[d] + ((a >> d) & 1) - 1;
      good = good && x >= 1 && x <= b.nn[d];
      node += (x - 1) * b.stride[d];
    }
    in[a] = good; id[a] = node;
  }
  long long t = off[q];
  for (int a = 0; a < nloc; ++a) {
    if (!in[a]) continue;
    for (int bb = 0; bb < nloc; ++bb) {
      if (!in[bb]) continue;
      I[t] = id[a]; J[t] = id[bb]; V[t] = b.Aref[a * nloc + bb];
      ++t;
    }
  }
}

extern "C" int pa_fem_triplets_device(pa_ctx *c, int32_t D, const int64_t *nodes, const int64_t *lo, const int64_t *hi, const double *Aref,
                                      int64_t *count, void **dI, void **dJ, void **dV) {
  PA_REQUIRE(c && D >= 1 && D <= 3 && nodes && lo && hi && Aref && count && dI && dJ && dV, "bad arguments");
  PA_REQUIRE(!c->capturing, "not inside a graph capture");
  PA_HIP(hipSetDevice(c->device));
  hipStream_t s = c->s[0];
  pa_fem_box b;
  b.D = D;
  long long n_cells = 1;
  for (int d = 0; d < 3; ++d) {
    b.nn[d] = d < D ? nodes[d] : 1; b.lo[d] = d < D ? lo[d] : 1; b.hi[d] = d < D ? hi[d] : 1;
    b.len[d] = std::max<long long>(0, b.hi[d] - b.lo[d] + 1);
    b.stride[d] = d == 0 ? 1 : b.stride[d - 1] * b.nn[d - 1];
    n_cells *= b.len[d];
  }
  for (int k = 0; k < 64; ++k) b.Aref[k] = k < (1 << D) * (1 << D) ? Aref[k] : 0.0;
  *count = 0; *dI = *dJ = *dV = nullptr;
  if (n_cells == 0) return PA_OK;
  scratch sc;
  long long *d_cnt = nullptr;
  PA_TRY(sc.get(&d_cnt, (size_t)n_cells + 1));
  PA_HIP(hipMemsetAsync(d_cnt + n_cells, 0, sizeof(long long), s));
  hipLaunchKernelGGL(kf_count, grid1(n_cells), dim3(256), 0, s, b, n_cells, d_cnt);
  size_t tb = 0;
  PA_HIP(rocprim::exclusive_scan((void *)nullptr, tb, d_cnt, d_cnt, 0ll, (size_t)n_cells + 1, rocprim::plus<long long>(), s));
  char *tmp = nullptr;
  PA_TRY(sc.get(&tmp, tb));
  PA_HIP(rocprim::exclusive_scan((void *)tmp, tb, d_cnt, d_cnt, 0ll, (size_t)n_cells + 1, rocprim::plus<long long>(), s));
  long long total = 0;
  PA_TRY(d2h(s, &total, d_cnt + n_cells, 1));
  *count = total;
  if (total == 0) return PA_OK;
  void *pI = nullptr, *pJ = nullptr, *pV = nullptr;
  if (pa_raw_malloc(&pI, 8 * (size_t)total) != hipSuccess || pa_raw_malloc(&pJ, 8 * (size_t)total) != hipSuccess ||
      pa_raw_malloc(&pV, 8 * (size_t)total) != hipSuccess) {
    (void)hipGetLastError();
    if (pI) (void)pa_raw_free(pI);
    if (pJ) (void)pa_raw_free(pJ);
    if (pV) (void)pa_raw_free(pV);
    pa_set_err("pa_fem_triplets_device: no room for %lld triplets", total);
    return PA_ERR_HIP;
  }
  hipLaunchKernelGGL(kf_fill, grid1(n_cells), dim3(256), 0, s, b, n_cells, (const long long *)d_cnt, (long long *)pI, (long long *)pJ, (double *)pV);
  PA_HIP(hipGetLastError());
  PA_HIP(hipStreamSynchronize(s));
  *dI = pI; *dJ = pJ; *dV = pV;
  return PA_OK;
}

// frees what pa_fem_triplets_device returned (any of the three may be NULL)
extern "C" int pa_triplets_free(pa_ctx *c, void *dI, void *dJ, void *dV) {
  PA_REQUIRE(c != nullptr, "ctx is NULL");
  PA_HIP(hipSetDevice(c->device));
  PA_HIP(hipStreamSynchronize(c->s[0]));
  if (dI) (void)pa_raw_free(dI);
  if (dJ) (void)pa_raw_free(dJ);
  if (dV) (void)pa_raw_free(dV);
  return PA_OK;
}

// n values of 8 bytes from HBM to the host (device triplets looked at by a test or a host route)
extern "C" int pa_triplets_download(pa_ctx *c, const void *d, int64_t n, void *host) {
  PA_REQUIRE(c && (n == 0 || (d && host)), "bad arguments");
  PA_HIP(hipSetDevice(c->device));
  if (n) PA_HIP(hipMemcpyAsync(host, d, 8 * (size_t)n, hipMemcpyDeviceToHost, c->s[0]));
  PA_HIP(hipStreamSynchronize(c->s[0]));
  return PA_OK;
}
