// pa_host.cpp -- native host-side set-up helpers of libpa_hip.so (no device code).
//
// The reference does its set-up in compiled Julia loops; these are their native twins so that a
// 256^3-per-part problem (4.5e8 stored entries) is set up in seconds.  They run once, on the host,
// and produce exactly the arrays the reference would hand to the device path (1-based ids).
// Each function cites the reference loop it restates (paths relative to /root/reference).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "pa_internal.h"

// HPCG/src/sparse_matrix.jl:27-80 build_matrix
extern "C" int pa_host_hpcg_build_matrix(int64_t nx, int64_t ny, int64_t nz, int64_t gnx, int64_t gny, int64_t gnz,
                                         int64_t gix0, int64_t giy0, int64_t giz0, int64_t *I, int64_t *J, double *V,
                                         double *b, int64_t *row_b, int64_t *nnz_out) {
  PA_REQUIRE(nx > 0 && ny > 0 && nz > 0, "row_count must be > 0");  // @assert row_count > 0 (:30)
  PA_REQUIRE(nnz_out != nullptr, "nnz_out is NULL");
  const bool fill = I && J && V;
  int64_t k = 0;
  for (int64_t iz = 1; iz <= nz; ++iz) {
    const int64_t giz = giz0 + iz - 1;
    for (int64_t iy = 1; iy <= ny; ++iy) {
      const int64_t giy = giy0 + iy - 1;
      for (int64_t ix = 1; ix <= nx; ++ix) {
        const int64_t gix = gix0 + ix - 1;
        const int64_t cur_row = (iz - 1) * nx * ny + (iy - 1) * nx + (ix - 1);  // 0-based slot
        const int64_t cur_g = (giz - 1) * gnx * gny + (giy - 1) * gnx + (gix - 1) + 1;
        int64_t cnt = 0;
        for (int sz = -1; sz <= 1; ++sz) {
          if (!(giz + sz > 0 && giz + sz < gnz + 1)) continue;
          for (int sy = -1; sy <= 1; ++sy) {
            if (!(giy + sy > 0 && giy + sy < gny + 1)) continue;
            for (int sx = -1; sx <= 1; ++sx) {
              if (!(gix + sx > 0 && gix + sx < gnx + 1)) continue;
              if (fill) {
                const int64_t col = cur_g + sz * gnx * gny + sy * gnx + sx;
                V[k] = (col == cur_g) ? 26.0 : -1.0;
                I[k] = cur_g;
                J[k] = col;
              }
              ++k;
              ++cnt;
            }
          }
        }
        if (row_b) row_b[cur_row] = cur_g;
        if (b) b[cur_row] = 27.0 - (double)cnt;
      }
    }
  }
  *nnz_out = k;
  return PA_OK;
}

// src/gallery.jl:40-84 `setup`: diag first (alpha*2D), then d = 1..D, i in (-1,+1): -alpha
// The triplets are written by host threads over slabs of the outermost direction; a slab's first slot is a closed form
// (rows + per direction the nodes that have a left / a right neighbour), so the stream is the sequential loop's.
#include <thread>
static int64_t fdm_count(int D, const int64_t *l, const int64_t *h, const int64_t *nn) {
  int64_t rows = 1;
  for (int d = 0; d < D; ++d) rows *= std::max<int64_t>(0, h[d] - l[d] + 1);
  if (rows == 0) return 0;
  int64_t t = rows;
  for (int d = 0; d < D; ++d) {
    const int64_t len = h[d] - l[d] + 1, others = rows / len;
    const int64_t with_left = len - (l[d] <= 1 ? 1 : 0), with_right = len - (h[d] >= nn[d] ? 1 : 0);
    t += others * (std::max<int64_t>(0, with_left) + std::max<int64_t>(0, with_right));
  }
  return t;
}
extern "C" int pa_host_laplacian_fdm(int32_t D, const int64_t *n, const int64_t *lo, const int64_t *hi, int64_t *I,
                                     int64_t *J, double *V, int64_t *nnz_out) {
  PA_REQUIRE(D >= 1 && D <= 3 && n && lo && hi && nnz_out, "bad arguments");
  double alpha = 1.0;
  for (int d = 0; d < D; ++d) alpha *= (double)(n[d] + 1);
  int64_t stride[3] = {1, 1, 1}, l[3] = {1, 1, 1}, h[3] = {1, 1, 1}, nn[3] = {1, 1, 1};
  for (int d = 0; d < D; ++d) { l[d] = lo[d]; h[d] = hi[d]; nn[d] = n[d]; }
  for (int d = 1; d < D; ++d) stride[d] = stride[d - 1] * n[d - 1];
  const bool fill = I && J && V;
  *nnz_out = fdm_count(D, l, h, nn);
  if (!fill || *nnz_out == 0) return PA_OK;
  const int od = D - 1;                                  // slabs of the outermost direction
  const int64_t olen = h[od] - l[od] + 1;
  unsigned hw = std::thread::hardware_concurrency();
  int T = hw ? (int)std::min<unsigned>(hw, 32) : 4;
  if (const char *e = getenv("PA_HOST_THREADS")) T = std::max(1, atoi(e));
  if (*nnz_out < ((int64_t)1 << 20)) T = 1;
  T = (int)std::min<int64_t>(T, olen);
  auto work = [&](int th) {
    int64_t sl[3] = {l[0], l[1], l[2]}, sh[3] = {h[0], h[1], h[2]}, before_h[3] = {h[0], h[1], h[2]};
    sl[od] = l[od] + olen * th / T;
    sh[od] = l[od] + olen * (th + 1) / T - 1;
    before_h[od] = sl[od] - 1;
    int64_t t = fdm_count(D, l, before_h, nn);          // slots of the slabs before this one
    int64_t c[3];
    for (c[2] = sl[2]; c[2] <= sh[2]; ++c[2])
      for (c[1] = sl[1]; c[1] <= sh[1]; ++c[1])
        for (c[0] = sl[0]; c[0] <= sh[0]; ++c[0]) {      // CartesianIndices(ranges): first index fastest
          const int64_t node_i = (c[0] - 1) + (c[1] - 1) * (D > 1 ? stride[1] : 0) + (c[2] - 1) * (D > 2 ? stride[2] : 0) + 1;
          I[t] = node_i; J[t] = node_i; V[t] = alpha * 2 * D;
          ++t;
          for (int d = 0; d < D; ++d)
            for (int i = -1; i <= 1; i += 2) {
              const int64_t cj = c[d] + i;
              if (cj < 1 || cj > nn[d]) continue;
              I[t] = node_i; J[t] = node_i + i * stride[d]; V[t] = -alpha;
              ++t;
            }
        }
  };
  std::vector<std::thread> pool;
  for (int th = 1; th < T; ++th) pool.emplace_back(work, th);
  work(0);
  for (auto &x : pool) x.join();
  return PA_OK;
}

// src/gallery.jl:110-239 laplacian_fem `setup` of one part: the part loops over ITS CELLS (the box [lo,hi] of the (nodes+1)^D
// cell grid, column-major), and for every cell over local node i, then local node j (2^D corners, column-major), emitting
// (node_i, node_j, Aref[i][j]) when both nodes are interior -- a cell corner c + offset - 1 outside 1..nodes[d] is a boundary
// node.  Aref: the 2^D x 2^D reference matrix (row-major), computed by the caller exactly as :123-162 does.  Two passes over the
// slabs of the outermost direction: entries per slab, then the fill (threaded).
extern "C" int pa_host_laplacian_fem(int32_t D, const int64_t *nodes, const int64_t *lo, const int64_t *hi, const double *Aref,
                                     int64_t *I, int64_t *J, double *V, int64_t *nnz_out) {
  PA_REQUIRE(D >= 1 && D <= 3 && nodes && lo && hi && Aref && nnz_out, "bad arguments");
  const int nloc = 1 << D;
  int64_t nn[3] = {1, 1, 1}, l[3] = {1, 1, 1}, h[3] = {1, 1, 1}, stride[3] = {1, 1, 1};
  for (int d = 0; d < D; ++d) { nn[d] = nodes[d]; l[d] = lo[d]; h[d] = hi[d]; }
  for (int d = 1; d < D; ++d) stride[d] = stride[d - 1] * nodes[d - 1];
  const int od = D - 1;
  const int64_t olen = h[od] - l[od] + 1;
  if (olen <= 0) { *nnz_out = 0; return PA_OK; }
  // interior corners of a cell along direction d: coordinate c + a - 1 for a in {0, 1}
  auto ok1 = [&](int d, int64_t c, int a) { const int64_t x = c + a - 1; return x >= 1 && x <= nn[d]; };
  // entries of the cells with outermost coordinate c_od: (interior corners)^2 summed over the inner box; the count factorises
  auto slab_count = [&](int64_t c_od) {
    int64_t k_od = (int64_t)ok1(od, c_od, 0) + (int64_t)ok1(od, c_od, 1);
    // sum over inner cells of (k_od * prod_d k_d)^2 = k_od^2 * prod_d sum_c k_d(c)^2
    int64_t total = k_od * k_od;
    for (int d = 0; d < od; ++d) {
      int64_t sd = 0;
      for (int64_t c = l[d]; c <= h[d]; ++c) { const int64_t k = (int64_t)ok1(d, c, 0) + (int64_t)ok1(d, c, 1); sd += k * k; }
      total *= sd;
    }
    return total;
  };
  std::vector<int64_t> first(olen + 1, 0);
  for (int64_t k = 0; k < olen; ++k) first[k + 1] = first[k] + slab_count(l[od] + k);
  *nnz_out = first[olen];
  if (!(I && J && V) || *nnz_out == 0) return PA_OK;
  unsigned hw = std::thread::hardware_concurrency();
  int T = hw ? (int)std::min<unsigned>(hw, 32) : 4;
  if (const char *e = getenv("PA_HOST_THREADS")) T = std::max(1, atoi(e));
  if (*nnz_out < ((int64_t)1 << 20)) T = 1;
  T = (int)std::min<int64_t>(T, olen);
  auto work = [&](int th) {
    const int64_t k0 = olen * th / T, k1 = olen * (th + 1) / T;
    int64_t t = first[k0];
    int64_t c[3] = {1, 1, 1};
    int64_t sl[3] = {l[0], l[1], l[2]}, sh[3] = {h[0], h[1], h[2]};
    sl[od] = l[od] + k0; sh[od] = l[od] + k1 - 1;
    int64_t id[8];
    bool in[8];
    for (c[2] = sl[2]; c[2] <= sh[2]; ++c[2])
      for (c[1] = sl[1]; c[1] <= sh[1]; ++c[1])
        for (c[0] = sl[0]; c[0] <= sh[0]; ++c[0]) {
          for (int a = 0; a < nloc; ++a) {                       // corner a: offsets column-major (first direction fastest)
            bool good = true;
            int64_t node = 1;
            for (int d = 0; d < D; ++d) {
              const int64_t x = c[d] + ((a >> d) & 1) - 1;
              good = good && x >= 1 && x <= nn[d];
              node += (x - 1) * stride[d];
            }
            in[a] = good; id[a] = node;
          }
          for (int a = 0; a < nloc; ++a) {
            if (!in[a]) continue;
            for (int b = 0; b < nloc; ++b) {
              if (!in[b]) continue;
              I[t] = id[a]; J[t] = id[b]; V[t] = Aref[a * nloc + b];
              ++t;
            }
          }
        }
  };
  std::vector<std::thread> pool;
  for (int th = 1; th < T; ++th) pool.emplace_back(work, th);
  work(0);
  for (auto &x : pool) x.join();
  return PA_OK;
}

// src/p_range.jl:1502-1513,1609-1619: owner = LinearIndices(np)[searchsortedlast(start_d, c_d) ...]
extern "C" int pa_host_find_owner_block(int32_t D, const int64_t *n, const int32_t *np, const int64_t *const *starts,
                                        const int64_t *gids, int64_t count, int32_t *owners) {
  PA_REQUIRE(D >= 1 && D <= 8 && n && np && starts && (count == 0 || (gids && owners)), "bad arguments");
  for (int64_t k = 0; k < count; ++k) {
    int64_t r = gids[k] - 1;
    int64_t owner = 0, stride = 1;
    for (int d = 0; d < D; ++d) {
      const int64_t c = r % n[d] + 1;
      r /= n[d];
      const int64_t *s = starts[d];
      const int64_t j = (std::upper_bound(s, s + np[d] + 1, c) - s);  // searchsortedlast, 1-based
      owner += (j - 1) * stride;
      stride *= np[d];
    }
    owners[k] = (int32_t)(owner + 1);
  }
  return PA_OK;
}

// src/p_range.jl:205-241 filter_ghost
extern "C" int pa_host_filter_ghost(int32_t part, const int64_t *gids, const int32_t *owners, int64_t count,
                                    const int64_t *known, int64_t n_known, int64_t *out_gids, int32_t *out_owners,
                                    int64_t *n_new) {
  PA_REQUIRE((count == 0 || (gids && owners)) && n_new, "bad arguments");
  std::unordered_set<int64_t> seen;
  if (known) seen.insert(known, known + n_known);
  int64_t m = 0;
  for (int64_t k = 0; k < count; ++k) {
    const int64_t g = gids[k];
    if (g < 1) continue;
    if (owners[k] == part) continue;
    if (seen.insert(g).second) {
      if (out_gids) out_gids[m] = g;
      if (out_owners) out_owners[m] = owners[k];
      ++m;
    }
  }
  *n_new = m;
  return PA_OK;
}

// global_to_local of a block partition with appended ghosts: own box by arithmetic
// (BlockPartitionGlobalToOwn, src/p_range.jl:1483-1500) else ghost lookup (+n_own), else 0.
extern "C" int pa_host_global_to_local_block(int32_t D, const int64_t *n, const int64_t *lo, const int64_t *hi,
                                             const int64_t *ghost_gids, int64_t n_ghost, const int64_t *gids, int64_t count,
                                             int32_t *lids) {
  PA_REQUIRE(D >= 1 && D <= 8 && n && lo && hi && (count == 0 || (gids && lids)), "bad arguments");
  int64_t n_own = 1;
  for (int d = 0; d < D; ++d) n_own *= (hi[d] - lo[d] + 1);
  std::unordered_map<int64_t, int32_t> g2g;
  g2g.reserve((size_t)n_ghost * 2 + 1);
  for (int64_t k = 0; k < n_ghost; ++k) g2g.emplace(ghost_gids[k], (int32_t)(k + 1));
  for (int64_t k = 0; k < count; ++k) {
    const int64_t g = gids[k];
    if (g < 1) { lids[k] = (int32_t)g; continue; }  // map_x_to_y! leaves ids < 1 untouched (src/p_range.jl:300-309)
    int64_t r = g - 1, own = 0, stride = 1;
    bool inside = true;
    for (int d = 0; d < D; ++d) {
      const int64_t c = r % n[d] + 1;
      r /= n[d];
      if (c < lo[d] || c > hi[d]) { inside = false; break; }
      own += (c - lo[d]) * stride;
      stride *= (hi[d] - lo[d] + 1);
    }
    if (inside) { lids[k] = (int32_t)(own + 1); continue; }
    auto it = g2g.find(g);
    lids[k] = it == g2g.end() ? 0 : (int32_t)(it->second + n_own);
  }
  return PA_OK;
}

// compresscoo(SparseMatrixCSR{1,Float64,Int32},...;combine=+,skip): src/sparse_utils.jl:313-350.
// colval/nzval must have room for `count` entries; *nnz_out is the number actually stored.
extern "C" int pa_host_compresscoo_csr(const int32_t *I, const int32_t *J, const double *V, int64_t count, int64_t m,
                                       int64_t n, int skip, int32_t *rowptr, int32_t *colval, double *nzval,
                                       int64_t *nnz_out) {
  PA_REQUIRE(rowptr && nnz_out && m >= 0 && n >= 0 && (count == 0 || (I && J && V && colval && nzval)), "bad arguments");
  if (skip && m * n == 0) count = 0;  // :334-337
  // counting sort by row (stable), ids < 1 rewritten to (1,1,0.0) when skip (FilteredCooVector :370-390)
  std::vector<int64_t> start(m + 2, 0);
  auto row_of = [&](int64_t k) -> int64_t { return (skip && (I[k] < 1 || J[k] < 1)) ? 1 : I[k]; };
  for (int64_t k = 0; k < count; ++k) {
    const int64_t r = row_of(k);
    PA_REQUIRE(r >= 1 && r <= m, "row id %lld outside 1:%lld at entry %lld", (long long)r, (long long)m, (long long)k);
    start[r + 1]++;
  }
  for (int64_t r = 1; r <= m + 1; ++r) start[r] += start[r - 1];   // start[r] = first slot of row r (1-based rows)
  std::vector<int32_t> cj(count);
  std::vector<double> cv(count);
  {
    std::vector<int64_t> pos(start.begin(), start.end());
    for (int64_t k = 0; k < count; ++k) {
      const bool bad = skip && (I[k] < 1 || J[k] < 1);
      const int64_t r = bad ? 1 : I[k];
      const int64_t q = pos[r]++;
      cj[q] = bad ? 1 : J[k];
      cv[q] = bad ? 0.0 : V[k];
      PA_REQUIRE(cj[q] >= 1 && cj[q] <= n, "column id %d outside 1:%lld at entry %lld", cj[q], (long long)n, (long long)k);
    }
  }
  // per row: stable sort by column, combine duplicates with + in input order
  int64_t out = 0;
  rowptr[0] = 1;
  std::vector<std::pair<int32_t, double>> tmp;
  for (int64_t r = 1; r <= m; ++r) {
    const int64_t a = start[r], e = start[r + 1];
    bool sorted = true;
    for (int64_t q = a + 1; q < e; ++q)
      if (cj[q] <= cj[q - 1]) { sorted = false; break; }
    if (sorted) {
      for (int64_t q = a; q < e; ++q) { colval[out] = cj[q]; nzval[out] = cv[q]; ++out; }
    } else {
      tmp.clear();
      for (int64_t q = a; q < e; ++q) tmp.emplace_back(cj[q], cv[q]);
      std::stable_sort(tmp.begin(), tmp.end(), [](const auto &x, const auto &y) { return x.first < y.first; });
      for (size_t q = 0; q < tmp.size(); ++q) {
        if (q > 0 && tmp[q].first == tmp[q - 1].first) {
          nzval[out - 1] = nzval[out - 1] + tmp[q].second;
        } else {
          colval[out] = tmp[q].first; nzval[out] = tmp[q].second; ++out;
        }
      }
    }
    rowptr[r] = (int32_t)(out + 1);
  }
  *nnz_out = out;
  return PA_OK;
}

// split_format_locally (src/p_sparse_matrix.jl:823-899), own-row branches, identity permutations:
// an entry (i,j) of the local CSR goes to own_own if j <= n_own_cols else to own_ghost (column j-n_own_cols).
extern "C" int pa_host_split_csr(int64_t n_own_rows, int64_t n_own_cols, int64_t n_ghost_cols, const int32_t *rowptr,
                                 const int32_t *colval, const double *nzval, int32_t *oo_rowptr, int32_t *oo_colval,
                                 double *oo_nzval, int32_t *oh_rowptr, int32_t *oh_colval, double *oh_nzval,
                                 int64_t *nnz_oo, int64_t *nnz_oh) {
  PA_REQUIRE(rowptr && nnz_oo && nnz_oh && n_own_rows >= 0, "bad arguments");
  const bool fill = oo_rowptr && oh_rowptr;
  int64_t a = 0, b = 0;
  if (fill) { oo_rowptr[0] = 1; oh_rowptr[0] = 1; }
  for (int64_t r = 0; r < n_own_rows; ++r) {
    for (int64_t p = rowptr[r] - 1; p < rowptr[r + 1] - 1; ++p) {
      const int32_t j = colval[p];
      if (j <= n_own_cols) {
        if (fill && oo_colval) { oo_colval[a] = j; oo_nzval[a] = nzval[p]; }
        ++a;
      } else {
        PA_REQUIRE(j - n_own_cols <= n_ghost_cols, "column %d beyond the ghost columns", j);
        if (fill && oh_colval) { oh_colval[b] = (int32_t)(j - n_own_cols); oh_nzval[b] = nzval[p]; }
        ++b;
      }
    }
    if (fill) { oo_rowptr[r + 1] = (int32_t)(a + 1); oh_rowptr[r + 1] = (int32_t)(b + 1); }
  }
  *nnz_oo = a;
  *nnz_oh = b;
  return PA_OK;
}

// ------------------------------------------------------------------------------------------------
// Fused HPCG set-up for large parts (256^3 per part = 4.5e8 stored entries).
//
// Same result, bit for bit, as the generic chain  build_matrix -> find_owner -> union_ghost ->
// map_global_to_local! -> compresscoo -> split_format_locally  (HPCG/src/sparse_matrix.jl:105-122),
// but without materialising the Int64 COO triplets (3 x 3.6 GB per part at 256^3): the ghost numbering
// only depends on the boundary rows of the COO stream, and each CSR row can be written directly.
// tests/test_host_setup.py checks fused == generic == oracle on small grids.
// ------------------------------------------------------------------------------------------------
#include <thread>

namespace {
struct HpcgGeom {
  int64_t nx, ny, nz, gnx, gny, gnz, x0, y0, z0;  // x0.. = global coords (1-based) of the first own node
  inline bool in_grid(int64_t gx, int64_t gy, int64_t gz) const {
    return gx > 0 && gx < gnx + 1 && gy > 0 && gy < gny + 1 && gz > 0 && gz < gnz + 1;
  }
  inline bool in_own(int64_t gx, int64_t gy, int64_t gz) const {
    return gx >= x0 && gx < x0 + nx && gy >= y0 && gy < y0 + ny && gz >= z0 && gz < z0 + nz;
  }
  inline int64_t gid(int64_t gx, int64_t gy, int64_t gz) const { return (gz - 1) * gnx * gny + (gy - 1) * gnx + (gx - 1) + 1; }
  inline int64_t own_id(int64_t gx, int64_t gy, int64_t gz) const {  // 1-based own id, column-major in the own box
    return (gx - x0) + (gy - y0) * nx + (gz - z0) * nx * ny + 1;
  }
};

// per dimension: how many of s in {-1,0,1} land inside the own box (a) / inside the grid but outside the box (b)
inline void dim_counts(int64_t g, int64_t lo, int64_t n_own, int64_t gn, int &a, int &b) {
  a = 0; b = 0;
  for (int s = -1; s <= 1; ++s) {
    const int64_t c = g + s;
    if (c < 1 || c > gn) continue;
    if (c >= lo && c < lo + n_own) ++a; else ++b;
  }
}

int n_threads_for(int64_t work) {
  unsigned hw = std::thread::hardware_concurrency();
  int t = hw ? (int)hw : 4;
  if (const char *e = getenv("PA_HOST_THREADS")) t = atoi(e);
  if (t < 1) t = 1;
  if (t > 32) t = 32;
  if (work < (int64_t)1 << 20) t = 1;
  return t;
}
}  // namespace

// Pass 1: ghost gids in first-seen order of the COO stream (iz,iy,ix ; sz,sy,sx) and the block sizes.
// ghost_gids may be NULL (count only); otherwise it must hold *n_ghost entries from a previous call.
extern "C" int pa_host_hpcg_ghosts(int64_t nx, int64_t ny, int64_t nz, int64_t gnx, int64_t gny, int64_t gnz, int64_t gix0,
                                   int64_t giy0, int64_t giz0, int64_t *ghost_gids, int64_t *n_ghost, int64_t *nnz_oo,
                                   int64_t *nnz_oh) {
  PA_REQUIRE(nx > 0 && ny > 0 && nz > 0 && n_ghost && nnz_oo && nnz_oh, "bad arguments");
  const HpcgGeom G{nx, ny, nz, gnx, gny, gnz, gix0, giy0, giz0};
  std::unordered_set<int64_t> seen;
  int64_t m = 0;
  // block sizes in closed form: the counts factor per direction
  auto sums = [&](int64_t g0, int64_t n_own, int64_t gn, int64_t &own, int64_t &tot) {
    own = tot = 0;
    for (int64_t i = 0; i < n_own; ++i) { int a, b; dim_counts(g0 + i, g0, n_own, gn, a, b); own += a; tot += a + b; }
  };
  int64_t ox, tx, oy, ty, oz, tz;
  sums(gix0, nx, gnx, ox, tx); sums(giy0, ny, gny, oy, ty); sums(giz0, nz, gnz, oz, tz);
  const int64_t oo = ox * oy * oz, oh = tx * ty * tz - oo;
  // the ghosts: only rows on the part's surface have any; they are visited in the order of the full sweep (iz, iy, ix), the
  // interior of every line skipped (a line whose y or z neighbours leave the box is surface from end to end)
  auto visit = [&](int64_t ix, int64_t iy, int64_t iz) {
    const int64_t gx = gix0 + ix, gy = giy0 + iy, gz = giz0 + iz;
    for (int sz = -1; sz <= 1; ++sz) for (int sy = -1; sy <= 1; ++sy) for (int sx = -1; sx <= 1; ++sx) {
      if (!G.in_grid(gx + sx, gy + sy, gz + sz) || G.in_own(gx + sx, gy + sy, gz + sz)) continue;
      const int64_t g = G.gid(gx + sx, gy + sy, gz + sz);
      if (seen.insert(g).second) { if (ghost_gids) ghost_gids[m] = g; ++m; }
    }
  };
  for (int64_t iz = 0; iz < nz; ++iz) {
    int az, bz; dim_counts(giz0 + iz, giz0, nz, gnz, az, bz);
    for (int64_t iy = 0; iy < ny; ++iy) {
      int ay, by; dim_counts(giy0 + iy, giy0, ny, gny, ay, by);
      if (bz > 0 || by > 0) {
        for (int64_t ix = 0; ix < nx; ++ix) visit(ix, iy, iz);
      } else {
        int a0, b0, a1, b1;
        dim_counts(gix0, gix0, nx, gnx, a0, b0);
        dim_counts(gix0 + nx - 1, gix0, nx, gnx, a1, b1);
        if (b0 > 0) visit(0, iy, iz);
        if (nx > 1 && b1 > 0) visit(nx - 1, iy, iz);
      }
    }
  }
  *n_ghost = m; *nnz_oo = oo; *nnz_oh = oh;
  return PA_OK;
}

// Pass 2: own_own / own_ghost CSR (1-based, sorted columns) and b, written row by row in parallel.  RP = Int32 row
// pointers (SparseMatrixCSR{1,Float64,Int32}) or Int64 for a part with 2^31 stored entries or more.
template <typename RP>
static int hpcg_split_csr_impl(int64_t nx, int64_t ny, int64_t nz, int64_t gnx, int64_t gny, int64_t gnz, int64_t gix0,
                               int64_t giy0, int64_t giz0, const int64_t *ghost_gids, int64_t n_ghost,
                               RP *oo_rowptr, int32_t *oo_colval, double *oo_nzval, RP *oh_rowptr,
                               int32_t *oh_colval, double *oh_nzval, double *b) {
  PA_REQUIRE(nx > 0 && ny > 0 && nz > 0 && oo_rowptr && oh_rowptr && oo_colval && oo_nzval && b, "bad arguments");
  PA_REQUIRE(n_ghost == 0 || (ghost_gids && oh_colval && oh_nzval), "ghost arrays are NULL");
  const HpcgGeom G{nx, ny, nz, gnx, gny, gnz, gix0, giy0, giz0};
  const int64_t nrows = nx * ny * nz;
  std::unordered_map<int64_t, int32_t> g2g;
  g2g.reserve((size_t)n_ghost * 2 + 1);
  for (int64_t k = 0; k < n_ghost; ++k) g2g.emplace(ghost_gids[k], (int32_t)(k + 1));
  // row pointers by closed-form counts
  {
    int64_t a = 1, c = 1, row = 0;
    oo_rowptr[0] = 1; oh_rowptr[0] = 1;
    for (int64_t iz = 0; iz < nz; ++iz) {
      int az, bz; dim_counts(giz0 + iz, giz0, nz, gnz, az, bz);
      for (int64_t iy = 0; iy < ny; ++iy) {
        int ay, by; dim_counts(giy0 + iy, giy0, ny, gny, ay, by);
        for (int64_t ix = 0; ix < nx; ++ix) {
          int ax, bx; dim_counts(gix0 + ix, gix0, nx, gnx, ax, bx);
          const int64_t tot = (int64_t)(ax + bx) * (ay + by) * (az + bz), own = (int64_t)ax * ay * az;
          a += own; c += tot - own; ++row;
          PA_REQUIRE(sizeof(RP) == 8 || (a < 2147483647 && c < 2147483647), "block too large for Int32 row pointers");
          oo_rowptr[row] = (RP)a; oh_rowptr[row] = (RP)c;
          b[row - 1] = 27.0 - (double)tot;
        }
      }
    }
  }
  const int T = n_threads_for(nrows * 27);
  bool bad = false;
  auto work = [&](int t) {
    const int64_t z_lo = nz * t / T, z_hi = nz * (t + 1) / T;
    for (int64_t iz = z_lo; iz < z_hi; ++iz)
      for (int64_t iy = 0; iy < ny; ++iy)
        for (int64_t ix = 0; ix < nx; ++ix) {
          const int64_t row = iz * nx * ny + iy * nx + ix;
          const int64_t gx = gix0 + ix, gy = giy0 + iy, gz = giz0 + iz;
          const int64_t cur = G.gid(gx, gy, gz);
          int64_t p = oo_rowptr[row] - 1, q0 = oh_rowptr[row] - 1, q = q0;
          for (int sz = -1; sz <= 1; ++sz) for (int sy = -1; sy <= 1; ++sy) for (int sx = -1; sx <= 1; ++sx) {
            const int64_t cx = gx + sx, cy = gy + sy, cz = gz + sz;
            if (!G.in_grid(cx, cy, cz)) continue;
            const double v = (G.gid(cx, cy, cz) == cur) ? 26.0 : -1.0;
            if (G.in_own(cx, cy, cz)) {
              oo_colval[p] = (int32_t)G.own_id(cx, cy, cz);  // ascending in (sz,sy,sx) order
              oo_nzval[p] = v; ++p;
            } else {
              auto it = g2g.find(G.gid(cx, cy, cz));
              if (it == g2g.end()) { bad = true; continue; }
              // insertion sort by ghost id inside the row (compresscoo sorts columns)
              int64_t k = q;
              while (k > q0 && oh_colval[k - 1] > it->second) { oh_colval[k] = oh_colval[k - 1]; oh_nzval[k] = oh_nzval[k - 1]; --k; }
              oh_colval[k] = it->second; oh_nzval[k] = v; ++q;
            }
          }
        }
  };
  if (T == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(work, t);
    for (auto &x : th) x.join();
  }
  PA_REQUIRE(!bad, "a ghost column is missing from ghost_gids (call pa_host_hpcg_ghosts first)");
  return PA_OK;
}

extern "C" int pa_host_hpcg_split_csr(int64_t nx, int64_t ny, int64_t nz, int64_t gnx, int64_t gny, int64_t gnz, int64_t gix0,
                                      int64_t giy0, int64_t giz0, const int64_t *ghost_gids, int64_t n_ghost,
                                      int32_t *oo_rowptr, int32_t *oo_colval, double *oo_nzval, int32_t *oh_rowptr,
                                      int32_t *oh_colval, double *oh_nzval, double *b) {
  return hpcg_split_csr_impl<int32_t>(nx, ny, nz, gnx, gny, gnz, gix0, giy0, giz0, ghost_gids, n_ghost, oo_rowptr, oo_colval,
                                      oo_nzval, oh_rowptr, oh_colval, oh_nzval, b);
}

extern "C" int pa_host_hpcg_split_csr64(int64_t nx, int64_t ny, int64_t nz, int64_t gnx, int64_t gny, int64_t gnz, int64_t gix0,
                                        int64_t giy0, int64_t giz0, const int64_t *ghost_gids, int64_t n_ghost,
                                        int64_t *oo_rowptr, int32_t *oo_colval, double *oo_nzval, int64_t *oh_rowptr,
                                        int32_t *oh_colval, double *oh_nzval, double *b) {
  return hpcg_split_csr_impl<int64_t>(nx, ny, nz, gnx, gny, gnz, gix0, giy0, giz0, ghost_gids, n_ghost, oo_rowptr, oo_colval,
                                      oo_nzval, oh_rowptr, oh_colval, oh_nzval, b);
}


// The own|ghost block alone (1-based, sorted columns), for a part whose own|own block and right-hand side are generated in
// HBM (pa_rowsel.hip, pa_hpcg_own_block_create): only the rows on the part's surface hold entries, the interior is skipped
// after the closed-form count.  Same arrays as hpcg_split_csr_impl's oh_* outputs.
extern "C" int pa_host_hpcg_ghost_block(int64_t nx, int64_t ny, int64_t nz, int64_t gnx, int64_t gny, int64_t gnz, int64_t gix0,
                                        int64_t giy0, int64_t giz0, const int64_t *ghost_gids, int64_t n_ghost,
                                        int32_t *oh_rowptr, int32_t *oh_colval, double *oh_nzval) {
  PA_REQUIRE(nx > 0 && ny > 0 && nz > 0 && oh_rowptr, "bad arguments");
  PA_REQUIRE(n_ghost == 0 || (ghost_gids && oh_colval && oh_nzval), "ghost arrays are NULL");
  const HpcgGeom G{nx, ny, nz, gnx, gny, gnz, gix0, giy0, giz0};
  std::unordered_map<int64_t, int32_t> g2g;
  g2g.reserve((size_t)n_ghost * 2 + 1);
  for (int64_t k = 0; k < n_ghost; ++k) g2g.emplace(ghost_gids[k], (int32_t)(k + 1));
  const int T = std::max(1, std::min<int>((int)nz, n_threads_for(nx * ny * nz * 4)));
  std::vector<int64_t> first(T + 1, 0);
  bool bad = false;
  auto ghost_entries = [&](int64_t ix, int64_t iy, int64_t iz) {
    int ax, bx, ay, by, az, bz;
    dim_counts(gix0 + ix, gix0, nx, gnx, ax, bx);
    dim_counts(giy0 + iy, giy0, ny, gny, ay, by);
    dim_counts(giz0 + iz, giz0, nz, gnz, az, bz);
    return (int64_t)(ax + bx) * (ay + by) * (az + bz) - (int64_t)ax * ay * az;
  };
  auto count = [&](int t) {
    int64_t k = 0;
    for (int64_t iz = nz * t / T; iz < nz * (t + 1) / T; ++iz)
      for (int64_t iy = 0; iy < ny; ++iy)
        for (int64_t ix = 0; ix < nx; ++ix) k += ghost_entries(ix, iy, iz);
    first[t + 1] = k;
  };
  auto fill = [&](int t) {
    int64_t q = first[t];
    for (int64_t iz = nz * t / T; iz < nz * (t + 1) / T; ++iz)
      for (int64_t iy = 0; iy < ny; ++iy)
        for (int64_t ix = 0; ix < nx; ++ix) {
          const int64_t row = iz * nx * ny + iy * nx + ix;
          oh_rowptr[row] = (int32_t)(q + 1);
          if (ghost_entries(ix, iy, iz) == 0) continue;
          const int64_t gx = gix0 + ix, gy = giy0 + iy, gz = giz0 + iz, q0 = q;
          for (int sz = -1; sz <= 1; ++sz) for (int sy = -1; sy <= 1; ++sy) for (int sx = -1; sx <= 1; ++sx) {
            const int64_t cx = gx + sx, cy = gy + sy, cz = gz + sz;
            if (!G.in_grid(cx, cy, cz) || G.in_own(cx, cy, cz)) continue;
            auto it = g2g.find(G.gid(cx, cy, cz));
            if (it == g2g.end()) { bad = true; continue; }
            int64_t k = q;                               // insertion sort by ghost id inside the row (compresscoo sorts columns)
            while (k > q0 && oh_colval[k - 1] > it->second) { oh_colval[k] = oh_colval[k - 1]; oh_nzval[k] = oh_nzval[k - 1]; --k; }
            oh_colval[k] = it->second; oh_nzval[k] = -1.0; ++q;
          }
        }
    if (t == T - 1) oh_rowptr[nx * ny * nz] = (int32_t)(q + 1);
  };
  auto run = [&](auto &f) {
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(f, t);
    f(0);
    for (auto &x : th) x.join();
  };
  run(count);
  for (int t = 0; t < T; ++t) first[t + 1] += first[t];
  PA_REQUIRE(first[T] < (int64_t)2147483000, "block too large for Int32 row pointers");
  run(fill);
  PA_REQUIRE(!bad, "a ghost column is missing from ghost_gids (call pa_host_hpcg_ghosts first)");
  return PA_OK;
}

// ------------------------------------------------------------------------------------------------
// Multicolour smoother set-up: split the rows of a part by colour into n_colors blocks (n_own x n_local, unsplit
// column order: own columns, then ghost columns shifted by n_own_cols) and extract the diagonal.  out_rowptr[k] is
// colour k's 1-based row pointer (n_own+1 entries, prefilled by the caller from the row lengths: a row of another colour
// has length 0); the entries are copied row by row in parallel.  color[r] == -1: row r goes to no block.
// ------------------------------------------------------------------------------------------------
// 1-based row pointers of the n_colors blocks pa_host_color_split fills: out_rowptr[k][r + 1] - out_rowptr[k][r] = the stored
// entries of row r (own + ghost columns) when color[r] == k, else 0.  Threads take row ranges; a first pass gives every
// range its first slot per colour.
extern "C" int pa_host_color_rowptrs(int64_t n_own, const int32_t *oo_rowptr, const int32_t *oh_rowptr, const int32_t *color,
                                     int32_t n_colors, int32_t *const *out_rowptr) {
  PA_REQUIRE(n_own >= 0 && oo_rowptr && oh_rowptr && color && out_rowptr && n_colors > 0 && n_colors <= 64, "bad arguments");
  const int T = n_threads_for(n_own * 8);
  std::vector<int64_t> sums((size_t)T * n_colors, 0);
  auto len = [&](int64_t r) { return (int64_t)(oo_rowptr[r + 1] - oo_rowptr[r]) + (oh_rowptr[r + 1] - oh_rowptr[r]); };
  bool bad = false;
  auto count = [&](int t) {
    for (int64_t r = n_own * t / T; r < n_own * (t + 1) / T; ++r) {
      const int k = color[r];
      if (k == -1) continue;
      if (k < 0 || k >= n_colors) { bad = true; continue; }
      sums[(size_t)t * n_colors + k] += len(r);
    }
  };
  auto fill = [&](int t) {
    std::vector<int64_t> at(n_colors);
    for (int k = 0; k < n_colors; ++k) {
      int64_t a = 1;
      for (int u = 0; u < t; ++u) a += sums[(size_t)u * n_colors + k];
      at[k] = a;
    }
    for (int64_t r = n_own * t / T; r < n_own * (t + 1) / T; ++r) {
      const int k = color[r];
      for (int j = 0; j < n_colors; ++j) out_rowptr[j][r] = (int32_t)at[j];
      if (k >= 0 && k < n_colors) at[k] += len(r);
    }
    if (t == T - 1) for (int j = 0; j < n_colors; ++j) out_rowptr[j][n_own] = (int32_t)at[j];
  };
  auto run = [&](auto &f) {
    std::vector<std::thread> th;
    for (int t = 1; t < T; ++t) th.emplace_back(f, t);
    f(0);
    for (auto &x : th) x.join();
  };
  run(count);
  PA_REQUIRE(!bad, "a colour outside 0..n_colors-1");
  for (int k = 0; k < n_colors; ++k) {
    int64_t tot = 0;
    for (int t = 0; t < T; ++t) tot += sums[(size_t)t * n_colors + k];
    PA_REQUIRE(tot < (int64_t)2147483000, "colour %d holds more entries than Int32 row pointers address", k);
  }
  run(fill);
  return PA_OK;
}

extern "C" int pa_host_color_split(int64_t n_own, int64_t n_own_cols, const int32_t *oo_rowptr, const int32_t *oo_colval,
                                   const double *oo_nzval, const int32_t *oh_rowptr, const int32_t *oh_colval,
                                   const double *oh_nzval, const int32_t *color, int32_t n_colors,
                                   const int32_t *const *out_rowptr, int32_t *const *out_colval,
                                   double *const *out_nzval, double *diag) {
  PA_REQUIRE(n_own >= 0 && oo_rowptr && oh_rowptr && color && out_rowptr && out_colval && out_nzval && diag && n_colors > 0,
             "bad arguments");
  bool bad = false;
  const int T = n_threads_for(n_own * 27);
  auto work = [&](int t) {
    for (int64_t r = n_own * t / T; r < n_own * (t + 1) / T; ++r) {
      const int k = color[r];
      if (k == -1) continue;                         // a row no block takes (the caller wants some of the rows only)
      if (k < 0 || k >= n_colors) { bad = true; continue; }
      int64_t dst = out_rowptr[k][r] - 1;
      const int64_t need = (oo_rowptr[r + 1] - oo_rowptr[r]) + (oh_rowptr[r + 1] - oh_rowptr[r]);
      if (out_rowptr[k][r + 1] - out_rowptr[k][r] != need) { bad = true; continue; }
      double d = 0.0;
      for (int64_t p = oo_rowptr[r] - 1; p < oo_rowptr[r + 1] - 1; ++p, ++dst) {
        out_colval[k][dst] = oo_colval[p];
        out_nzval[k][dst] = oo_nzval[p];
        if (oo_colval[p] - 1 == r) d = oo_nzval[p];
      }
      for (int64_t p = oh_rowptr[r] - 1; p < oh_rowptr[r + 1] - 1; ++p, ++dst) {
        out_colval[k][dst] = oh_colval[p] + (int32_t)n_own_cols;
        out_nzval[k][dst] = oh_nzval[p];
      }
      diag[r] = d;
    }
  };
  if (T == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(work, t);
    for (auto &x : th) x.join();
  }
  PA_REQUIRE(!bad, "colour out of range, or a colour's row pointer does not match the row lengths");
  return PA_OK;
}

// a[i] = map[a[i]] in place, a[i] in [0, n_map) (the colours of a multicolour smoother renamed into sweep order: 16.7 M entries at
// 256^3, 40 ms as a numpy gather, 3 ms here on the host's threads)
extern "C" int pa_host_remap_int32(int32_t *a, int64_t n, const int32_t *map, int32_t n_map) {
  PA_REQUIRE((a || n == 0) && map && n_map > 0 && n >= 0, "bad arguments");
  unsigned hw = std::thread::hardware_concurrency();
  int T = hw ? (int)std::min<unsigned>(hw, 16) : 4;
  if (n < ((int64_t)1 << 18)) T = 1;
  std::vector<int> bad((size_t)T, 0);
  auto work = [&](int t) {
    for (int64_t i = n * t / T, e = n * (t + 1) / T; i < e; ++i) {
      const int32_t v = a[i];
      if (v < 0 || v >= n_map) { bad[t] = 1; continue; }
      a[i] = map[v];
    }
  };
  if (T == 1) work(0);
  else {
    std::vector<std::thread> th;
    for (int t = 0; t < T; ++t) th.emplace_back(work, t);
    for (auto &x : th) x.join();
  }
  for (int b : bad) PA_REQUIRE(!b, "an entry outside [0, %d)", n_map);
  return PA_OK;
}

