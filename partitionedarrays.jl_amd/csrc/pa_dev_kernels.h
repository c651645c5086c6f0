// pa_dev_kernels.h -- the small kernels and host helpers the device translation units share (pa_device.hip: context, vectors;
// pa_csr.hip: CSR blocks and the product launch; pa_mg.hip: smoother and grid transfer; pa_plan.hip: exchange plans, mul!).
// Everything here has internal linkage: a unit gets its own copy of what it uses.
#ifndef PA_DEV_KERNELS_H
#define PA_DEV_KERNELS_H
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wunused-function"

// ------------------------------------------------------------------------------------------------
// kernels
// ------------------------------------------------------------------------------------------------
#include "pa_spmv_kernel.h"
#include "pa_spmv_xwin.h"

// shipped configuration of the row-split kernel (chosen with tools/probe/spmv_probe.hip on MI355X)
constexpr int SPMV_BLK = 256;
constexpr int SPMV_NPT = PA_SPMV_CHUNK_NNZ / SPMV_BLK;  // stored entries per lane (6)
constexpr bool SPMV_NT = true;

static int host_threads(int64_t work) {
  unsigned hw = std::thread::hardware_concurrency();
  int t = hw ? (int)hw : 4;
  if (const char *e = getenv("PA_HOST_THREADS")) t = atoi(e);
  if (t < 1) t = 1;
  if (t > 32) t = 32;
  if (work < ((int64_t)1 << 20)) t = 1;
  return t;
}

// f(t, lo, hi) on T host threads over [0, n) split into T consecutive ranges (T = host_threads(work): 1 for small inputs)
template <class F>
static void host_parallel(int64_t n, int64_t work, F f) {
  const int T = (int)std::min<int64_t>(host_threads(work), std::max<int64_t>(1, n));
  if (T <= 1) { f(0, (int64_t)0, n); return; }
  std::vector<std::thread> th;
  for (int t = 1; t < T; ++t) th.emplace_back(f, t, n * t / T, n * (t + 1) / T);
  f(0, (int64_t)0, n / T);
  for (auto &x : th) x.join();
}

static __global__ void k_scale(double *__restrict__ y, int64_t n, double beta) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = (beta == 0.0) ? 0.0 : y[i] * beta;
}

static __global__ void k_fill(double *__restrict__ y, int64_t n, double v) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = v;
}

static __global__ void k_gather_values(double *__restrict__ dst, const double *__restrict__ src, const int *__restrict__ idx, int64_t n) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) dst[p] = src[idx[p]];
}

static __global__ void k_pack(double *__restrict__ buf, const double *__restrict__ v, const int *__restrict__ idx,
                       int n) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) buf[p] = v[idx[p]];
}

static __global__ void k_unpack_insert(double *__restrict__ v, const double *__restrict__ buf,
                                const int *__restrict__ idx, int n) {
  const int p = blockIdx.x * blockDim.x + threadIdx.x;
  if (p < n) v[idx[p]] = buf[p];
}

// one lane per distinct target; its contributions are added in ascending p (the reference's order)
static __global__ void k_unpack_add(double *__restrict__ v, const double *__restrict__ buf, const int *__restrict__ tgt,
                             const int *__restrict__ tptr, const int *__restrict__ tp, int n_tgt) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n_tgt) {
    const int lid = tgt[k];
    double acc = v[lid];
    for (int j = tptr[k]; j < tptr[k + 1]; ++j) acc = acc + buf[tp[j]];
    v[lid] = acc;
  }
}


// y = a*x + b*y, one rounding per multiply and per add.  b == 0 is a pure assignment y = a*x: y is NOT read (NaN / Inf
// left in y do not survive as 0*NaN, and -0.0 products keep their sign: what `dest .= a .* v` gives in the reference's
// broadcast, src/p_vector.jl:1216-1277).  x may be y itself (a scaling in place): no restrict promise on the pair.
static __global__ void k_axpby(double *y, const double *x, int64_t n, double a, double b) {
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // x only streams through (non-temporal: see k_cg_r_update); y is the vector that is wanted next (u before a product)
  if (b == 0.0) {
    for (; i < n; i += stride) y[i] = a * __builtin_nontemporal_load(&x[i]);
  } else {
    for (; i < n; i += stride) y[i] = a * __builtin_nontemporal_load(&x[i]) + b * y[i];
  }
}

__device__ inline double block_sum_256(double s, double *sh) {
  // 64-wide wavefront shuffle tree, then 4 wave sums through LDS; fixed order => deterministic
  for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = s;
  __syncthreads();
  double t = 0.0;
  if (threadIdx.x == 0) t = ((sh[0] + sh[1]) + (sh[2] + sh[3]));
  return t;
}

static __global__ __launch_bounds__(256) void k_dot_partial(const double *__restrict__ x, const double *__restrict__ y,
                                                     int64_t n, double *__restrict__ partial) {
  __shared__ double sh[4];
  double s = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    s += __builtin_nontemporal_load(&x[i]) * __builtin_nontemporal_load(&y[i]);
  const double t = block_sum_256(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

static __global__ __launch_bounds__(256) void k_dot_final(const double *__restrict__ partial, int n, double *out) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  const double t = block_sum_256(s, sh);
  if (threadIdx.x == 0) *out = t;
}

// ---- solver scalars that stay on the device (slots): coefficient = c * slot[num] / slot[den], index < 0 => 1 ----
__device__ __forceinline__ double slot_coef(const double *__restrict__ slots, double c, int num, int den) {
  double v = c;
  if (num >= 0) v = v * slots[num];
  if (den >= 0) v = v / slots[den];
  return v;
}

static __global__ void k_axpby_slot(double *__restrict__ y, const double *__restrict__ x, int64_t n,
                             const double *__restrict__ slots, double ca, int an, int ad, double cb, int bn, int bd) {
  const double a = slot_coef(slots, ca, an, ad), b = slot_coef(slots, cb, bn, bd);
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) y[i] = a * __builtin_nontemporal_load(&x[i]) + b * y[i];
}

// x .+= alpha .* u ; r .-= alpha .* c ; partial sums of dot(r,r) -- the tail of a CG iteration
// (HPCG/src/ref_cg.jl:64-67) in one pass; same per-element arithmetic and the same reduction tree as
// k_axpby + k_axpby + k_dot_partial, so the results are bit-identical to the unfused sequence.
static __global__ __launch_bounds__(256) void k_cg_update(double *__restrict__ x, double *__restrict__ r,
                                                   const double *__restrict__ u, const double *__restrict__ c,
                                                   int64_t n, const double *__restrict__ slots, int num, int den,
                                                   double *__restrict__ partial) {
  __shared__ double sh[4];
  const double a = slot_coef(slots, 1.0, num, den), ma = slot_coef(slots, -1.0, num, den);
  double s = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    __builtin_nontemporal_store(a * u[i] + 1.0 * __builtin_nontemporal_load(&x[i]), &x[i]);
    const double rn = ma * __builtin_nontemporal_load(&c[i]) + 1.0 * __builtin_nontemporal_load(&r[i]);
    __builtin_nontemporal_store(rn, &r[i]);
    s += rn * rn;
  }
  const double t = block_sum_256(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// r .-= alpha .* c ; partial sums of dot(r,r): the second and third statement of HPCG/src/ref_cg.jl:64-67 (per element
// and per reduction step the arithmetic of k_cg_update, so |r|^2 keeps its bits)
static __global__ __launch_bounds__(256) void k_cg_r_update(double *__restrict__ r, const double *__restrict__ c, int64_t n,
                                                     const double *__restrict__ slots, int num, int den,
                                                     double *__restrict__ partial) {
  __shared__ double sh[4];
  const double ma = slot_coef(slots, -1.0, num, den);
  double s = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // non-temporal: these streams are not wanted again before the next product, whose row pointers, descriptors and
  // gathered vector are (a product right behind another product finds ~225 MB of them in the Infinity Cache and runs
  // 7 % faster than one behind a kernel that streamed its operands through that cache: tools/probe/spmv_context.py)
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const double rn = ma * __builtin_nontemporal_load(&c[i]) + 1.0 * __builtin_nontemporal_load(&r[i]);
    __builtin_nontemporal_store(rn, &r[i]);
    s += rn * rn;
  }
  const double t = block_sum_256(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// x .+= alpha .* u (the first statement of ref_cg.jl:64-67, left over from the iteration before) and then
// u .= z .+ beta .* u (:56) in one pass: x is not read inside the loop, so its update may wait until u is about to change.
// Per element the arithmetic of k_cg_update's x line and of k_axpby_slot: same bits.
static __global__ void k_cg_xu_update(double *__restrict__ x, double *__restrict__ u, const double *__restrict__ z, int64_t n,
                               const double *__restrict__ slots, int a_num, int a_den, int b_num, int b_den) {
  const double a = slot_coef(slots, 1.0, a_num, a_den), b = slot_coef(slots, 1.0, b_num, b_den);
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {       // (x and z stream past the caches; u is what the next product gathers: it stays)
    const double ui = u[i];
    __builtin_nontemporal_store(a * ui + 1.0 * __builtin_nontemporal_load(&x[i]), &x[i]);
    u[i] = 1.0 * __builtin_nontemporal_load(&z[i]) + b * ui;
  }
}

static __global__ __launch_bounds__(256) void k_sum_partial(const double *__restrict__ p, int64_t n, double *__restrict__ partial) {
  __shared__ double sh[4];
  double s = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) s += p[i];
  const double t = block_sum_256(s, sh);
  if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

static __global__ __launch_bounds__(256) void k_dot_final_slot(const double *__restrict__ partial, int n, double *out,
                                                        int accumulate) {
  __shared__ double sh[4];
  double s = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
  const double t = block_sum_256(s, sh);
  if (threadIdx.x == 0) *out = accumulate ? *out + t : t;
}

// Gauss-Seidel, one dependency level: one lane per row of the level; the reference's per-row arithmetic
// (PartitionedSolvers/src/smoothers.jl:144-160; zero-guess variant :236-259).
static __global__ void k_gs_level(double *__restrict__ x, const double *__restrict__ b, const int *__restrict__ rowptr,
                           const int *__restrict__ col, const double *__restrict__ val, const double *__restrict__ diag,
                           const int *__restrict__ rows, int n, int zero_guess) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const int row = rows[k];
  double s = b[row];
  const int p0 = rowptr[row], p1 = rowptr[row + 1];
  // groups of GS_GROUP entries: all index/value loads, then all x gathers, then the (ordered) subtract chain -- three
  // memory round trips per group instead of three per entry (the kernel is latency-bound: a level is a few thousand
  // rows).  9 measured best on MI355X (27, a whole stencil row, needs 128 VGPRs and is slower).
  constexpr int GS_GROUP = 9;
  for (int p = p0; p < p1; p += GS_GROUP) {
    int c[GS_GROUP];
    double a[GS_GROUP], xv[GS_GROUP];
#pragma unroll
    for (int j = 0; j < GS_GROUP; ++j) {
      const int q = min(p + j, p1 - 1);
      c[j] = col[q];
      a[j] = val[q];
    }
#pragma unroll
    for (int j = 0; j < GS_GROUP; ++j) xv[j] = x[c[j]];
#pragma unroll
    for (int j = 0; j < GS_GROUP; ++j)
      if (p + j < p1 && (!zero_guess || c[j] < row)) s = s - a[j] * xv[j];
  }
  const double d = diag[row];
  if (!zero_guess) s = s + d * x[row];
  x[row] = s / d;
}

static __global__ void k_gs_color_update(double *__restrict__ x, const double *__restrict__ b, double *__restrict__ t,
                                  const double *__restrict__ diag, const int *__restrict__ rows, int n) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < n) {
    const int r = rows[k];
    x[r] = x[r] + (b[r] - t[r]) / diag[r];
    t[r] = 0.0;  // t is an accumulator for the next colour's A*x (pa_spmv with beta = 1 touches only its rows)
  }
}

static __global__ void k_restrict(double *__restrict__ rc, const double *__restrict__ rf, const double *__restrict__ axf,
                           const int *__restrict__ f2c, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) rc[i] = rf[f2c[i]] - axf[f2c[i]];
}

static __global__ void k_prolongate(double *__restrict__ xf, const double *__restrict__ xc, const int *__restrict__ f2c, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) xf[f2c[i]] = xf[f2c[i]] + xc[i];
}


static inline int seg_range(const pa_vec *v, int seg, int64_t *off, int64_t *len) {
  switch (seg) {
    case PA_SEG_OWN: *off = 0; *len = v->n_own; return PA_OK;
    case PA_SEG_GHOST: *off = v->n_own; *len = v->n_ghost; return PA_OK;
    case PA_SEG_LOCAL: *off = 0; *len = v->n_own + v->n_ghost; return PA_OK;
  }
  pa_set_err("unknown segment %d", seg);
  return PA_ERR_ARG;
}

static inline int grid_for(int64_t n, int threads, int cap = 4096) {
  int64_t g = (n + threads - 1) / threads;
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

static inline int upload_i32(const std::vector<int32_t> &h, int32_t **d) {
  PA_HIP(pa_raw_malloc(d, sizeof(int32_t) * std::max<size_t>(1, h.size())));
  if (!h.empty()) PA_HIP(pa_h2d(*d, h.data(), sizeof(int32_t) * h.size()));
  return PA_OK;
}

#pragma clang diagnostic pop
#endif
