// ctrefine.hip -- MI355X (gfx950) cluster-refinement engine behind include/ctrefine.h.
//
// A cluster is fitted start to finish on the chip: window (reference masks.py:30-68),
// elliptical masks (refine.py:43-51), sum-of-Gaussians residual and Jacobian rows
// (fitfunc.py:14-118,436-487), normal equations, bounded / equality-constrained
// Levenberg-Marquardt step, re-window rounds and failure rules (refine.py:343-430).
//
// Kernels:
//   frame_max_kernel      per-frame maximum (refine.py:354); HBM streaming
//   refine_small_kernel   singles / pairs with the default parameter modes: 16 or 64 lanes per
//                         cluster, [J r]^T [J r] accumulated in registers, DPP row all-reduce,
//                         register-resident solve, clusters pulled from a work counter
//   refine_block_kernel   everything else: W wavefronts per cluster, Jacobian rows staged in LDS
//                         and contracted with v_mfma_f64_16x16x4_f64, partial accumulators meet
//                         in LDS, register column-Cholesky (<= 31 variables) or cooperative LDS
//                         Cholesky + range-space KKT step (constraints)
//   find_clusters_kernel  cluster labelling of a frame (find.py:72-93)
// Clusters are binned by problem size on the host (ctr_plan_create); the bins run
// concurrently on forked streams.  gfx950 only; no CUDA paths, no CPU fallback.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "ctrefine.h"

namespace {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;
constexpr int MAXC = 6;       // equality constraints per cluster
constexpr int MAXF = 64;      // features per cluster
constexpr int MAXNT = 8;      // 16*8 = 128 columns >= CTR_MAX_VARS + 1
constexpr int FP = 14;        // derived per-feature constants (see fill_fpar)

struct KArgs {
  ctr_problem prob;
  const void* frames;
  int32_t frame_dtype;
  int32_t n_bin;
  int64_t shape[3];
  int64_t frame_elems;
  const int32_t* frame_index;
  const int32_t* feat_offset;
  const double* params;
  const double* low;
  const double* high;
  double* params_out;
  double* cost;
  int32_t* status;
  int32_t* n_rounds;
  int32_t* n_iter;
  const double* fmax;
  const int32_t* order;  // cluster ids of this bin
};

__device__ __forceinline__ size_t dtype_size(int dtype) {
  return dtype == CTR_DTYPE_U8 ? 1 : (dtype == CTR_DTYPE_U16 || dtype == CTR_DTYPE_I16) ? 2
       : (dtype == CTR_DTYPE_I32 || dtype == CTR_DTYPE_F32) ? 4 : 8;
}

__device__ __forceinline__ double load_pixel(const void* base, int dtype, size_t i) {
  switch (dtype) {
    case CTR_DTYPE_U8: return (double)((const uint8_t*)base)[i];
    case CTR_DTYPE_U16: return (double)((const uint16_t*)base)[i];
    case CTR_DTYPE_I16: return (double)((const int16_t*)base)[i];
    case CTR_DTYPE_I32: return (double)((const int32_t*)base)[i];
    case CTR_DTYPE_F32: return (double)((const float*)base)[i];
    default: return ((const double*)base)[i];
  }
}

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x += __shfl_xor(x, o);
  return x;
}
__device__ __forceinline__ double wave_max(double x) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) x = fmax(x, __shfl_xor(x, o));
  return x;
}
__device__ __forceinline__ double bcast0(double x) { return __shfl(x, 0); }

__device__ __forceinline__ int tri(int i) { return (i * (i + 1)) >> 1; }

struct Layout {
  int n, nv;
  int var_of[CTR_MAX_PARAMS];
  int per_feat[CTR_MAX_PARAMS];
};

// vect_from_params layout with groups=None (fitfunc.py:207-263)
__device__ __forceinline__ void make_layout(const ctr_problem& p, int n, Layout& L) {
  int nv = 0;
  L.n = n;
#pragma unroll
  for (int k = 0; k < CTR_MAX_PARAMS; ++k) {
    int m = k < p.n_params ? p.modes[k] : CTR_MODE_CONST;
    if (m == CTR_MODE_CONST) { L.var_of[k] = -1; L.per_feat[k] = 0; }
    else if (m == CTR_MODE_VAR) { L.var_of[k] = nv; L.per_feat[k] = 1; nv += n; }
    else { L.var_of[k] = nv; L.per_feat[k] = 0; nv += 1; }
  }
  L.nv = nv;
}

// ---- masks (refine.py:43-44) --------------------------------------------------

template <int ND>
__device__ __forceinline__ bool in_mask_exact(const int (&idx)[ND], const double (&rel)[ND],
                                              const int (&radius)[ND]) {
#pragma clang fp contract(off)
  double s = 0.;
#pragma unroll
  for (int a = 0; a < ND; ++a) {
    double t = ((double)idx[a] - rel[a]) / (double)radius[a];
    double t2 = t * t;
    s = s + t2;
  }
  return s <= 1.;
}

// Cheap test first; the IEEE-division form only where the two could disagree.
template <int ND>
__device__ __forceinline__ bool in_mask(const int (&idx)[ND], const double (&rel)[ND],
                                        const double (&inv_r2)[ND], const int (&radius)[ND]) {
  double s = 0.;
#pragma unroll
  for (int a = 0; a < ND; ++a) {
    double d = (double)idx[a] - rel[a];
    s += d * d * inv_r2[a];
  }
  if (fabs(s - 1.) > 1e-9) return s < 1.;
  return in_mask_exact<ND>(idx, rel, radius);
}

__device__ __forceinline__ double Msym(const double* Mp, int i, int j) {
  return i >= j ? Mp[tri(i) + j] : Mp[tri(j) + i];
}

// ---- equality constraints (constraints.py:59-137) -------------------------------

__device__ __forceinline__ int n_constraints(const ctr_problem& p, int n) {
  switch (p.constraint_kind) {
    case CTR_CONS_DIMER: return n == 2 ? 1 : 0;
    case CTR_CONS_TRIMER: return n == 3 ? 3 : 0;
    case CTR_CONS_TETRAMER: return n == 4 ? (p.ndim == 2 ? 4 : 6) : 0;
    default: return 0;
  }
}


// ---- generic clusters: one workgroup of W wavefronts per cluster ------------------------
//
// Any number of features / any parameter modes / constraints.  The W waves split
// the 64-pixel tiles of the window among themselves; each builds Jacobian rows in
// its own LDS row tile and contracts them with v_mfma_f64_16x16x4_f64 into
// register accumulators of the augmented matrix [J r]^T [J r]; the partial
// accumulators meet in LDS, wave 0 runs the bounded / constrained LM step on the
// sum (cooperative Cholesky on LDS) and publishes the next trial vector.  Two
// workgroup barriers per solver iteration.  W = 4 (NT <= 3), 2 (NT <= 6), 1.

__device__ __forceinline__ void wsync() {
  // LDS ordering inside ONE wavefront (DS operations of a wave execute in order)
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}


// ---- register-resident Cholesky step for small systems (one matrix column per lane) ----
//
// Lane c < NR holds column c of the damped normal matrix with the active set
// folded in (fixed variables and unused rows are identity rows).  Right-looking
// Cholesky: at step j the pivot column is broadcast row by row with v_readlane,
// every later column updates itself from its own (symmetric) entry col[j]; the
// right-hand side rides along as one more row, so the forward substitution is
// free.  The back substitution broadcasts each solved component once.  No LDS
// round trips, no dynamic register indexing.  Returns false if not positive definite.
// 1/sqrt(x) to double precision: hardware estimate + two Newton steps (the
// correctly rounded sqrt and division cost ~60 dependent instructions per pivot)
__device__ __forceinline__ double fast_rsqrt(double x) {
  double r = __builtin_amdgcn_rsq(x);
  r = r * fma(-0.5 * x * r, r, 1.5);
  r = r * fma(-0.5 * x * r, r, 1.5);
  return r;
}

__device__ __forceinline__ double readlane_f64(double x, int srclane) {
  const long long b = __double_as_longlong(x);
  const int lo = __builtin_amdgcn_readlane((int)(b & 0xffffffffLL), srclane);
  const int hi = __builtin_amdgcn_readlane((int)(b >> 32), srclane);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

template <int NR>
__device__ __forceinline__ bool column_solve(const double* Mp, int nv, double mu, bool is_free,
                                             int lane, double& x_own) {
  const int c = lane;
  const bool colv = c < nv && is_free;
  const unsigned long long fmask = __ballot(colv);
  double col[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const bool rowv = ((fmask >> i) & 1ull) != 0ull;
    double x = 0.;
    if (rowv && colv) x = Msym(Mp, i, c);
    if (i == c) x = colv ? x + mu * (x > 1e-300 ? x : 1.) : 1.;
    col[i] = x;
  }
  double y = colv ? Mp[tri(nv) + c] : 0.;
  double mydinv = 1., yown = 0.;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    if (j < nv) {  // rows beyond the variables are identity: nothing to eliminate (uniform branch)
      const double dj = readlane_f64(col[j], j);
      if (!(dj > 0.) || !isfinite(dj)) ok = false;
      const double dinv = fast_rsqrt(dj);
      const double lkj = c > j ? col[j] * dinv : 0.;  // L[c][j] for the columns still open
      const double yj = readlane_f64(y, j) * dinv;
      if (c == j) { mydinv = dinv; yown = yj; }
#pragma unroll
      for (int i = j + 1; i < NR; ++i) {
        const double lij = readlane_f64(col[i], j) * dinv;
        col[i] -= lij * lkj;
      }
      y -= yj * lkj;
    }
  }
  double s = 0.;
  x_own = 0.;
#pragma unroll
  for (int j = NR - 1; j >= 0; --j) {
    if (j < nv) {
      const double xj = readlane_f64((yown - s) * mydinv, j);
      if (c == j) x_own = xj;
      s += (c < j ? col[j] * mydinv : 0.) * xj;
    }
  }
  return ok;
}

template <int NT, int W>
struct SmemB {
  static constexpr int NVP = 16 * NT;
  static constexpr int RS = NVP + 1;
  static constexpr int NTILE = NT * (NT + 1) / 2;
  static constexpr int NF = NVP < MAXF ? NVP : MAXF;
  static constexpr int NVC = NVP < 32 ? NVP : 32;
  static constexpr int ROWS = WAVE * RS;               // one wave's row tile
  static constexpr int o_rows = 0;                      // W row tiles; tile 0 doubles as packed H
  static constexpr int o_M = o_rows + W * ROWS;
  static constexpr int o_v = o_M + NVP * (NVP + 1) / 2;
  static constexpr int o_vt = o_v + NVP;
  static constexpr int o_v0 = o_vt + NVP;
  static constexpr int o_lo = o_v0 + NVP;
  static constexpr int o_hi = o_lo + NVP;
  static constexpr int o_dl = o_hi + NVP;
  static constexpr int o_w = o_dl + NVP;
  static constexpr int o_cur = o_w + NVP;
  static constexpr int o_mco = o_cur + NF * CTR_MAX_PARAMS;
  static constexpr int o_fpar = o_mco + NF * 3;
  static constexpr int o_Cj = o_fpar + NF * FP;
  static constexpr int o_Cjt = o_Cj + MAXC * NVC;
  static constexpr int o_Y = o_Cjt + MAXC * NVC;
  static constexpr int o_small = o_Y + MAXC * NVC;      // cv[6] cvt[6] mult[6] . Sc[36] flag
  static constexpr int o_fr = o_small + 64;
  static constexpr int o_part = o_fr + NVP / 2 + 2;     // per wave: S, P
  static constexpr int o_ctl = o_part + 2 * W;          // ints: phase, origin[3], wshape[3]
  static constexpr int total = o_ctl + 8;
  static constexpr size_t bytes = (size_t)total * sizeof(double);
  static_assert(W == 1 || NTILE * 256 <= ROWS, "partial accumulators must fit a row tile");
};

enum { BP_EVAL_INIT = 1, BP_EVAL_TRIAL = 2, BP_STEP_ONLY = 3, BP_FINISH = 4 };

// in-place Cholesky of a packed lower-triangular matrix in LDS by ONE wave.
// Right-looking: per column one reciprocal square root, then the trailing
// update spread over the lanes as an 8 x 8 grid of (row, column) entries, so a
// lane does ~(nf - j)^2 / 128 multiply-subtracts per column instead of nf - j.
// dinv[j] = 1 / L[j][j] is kept for the substitutions.
__device__ bool chol_factor_w(double* Hp, double* dinv, int nf, int lane) {
  const int ty = lane >> 3, tx = lane & 7;
  for (int j = 0; j < nf; ++j) {
    const double d = Hp[tri(j) + j];
    if (!(d > 0.) || !isfinite(d)) return false;
    const double inv = fast_rsqrt(d);
    wsync();
    for (int i = j + 1 + lane; i < nf; i += WAVE) Hp[tri(i) + j] *= inv;
    if (lane == 0) { Hp[tri(j) + j] = d * inv; dinv[j] = inv; }
    wsync();
    for (int i = j + 1 + ty; i < nf; i += 8) {
      const double lij = Hp[tri(i) + j];
      double* ri = Hp + tri(i);
      for (int kk = j + 1 + tx; kk <= i; kk += 8) ri[kk] -= lij * Hp[tri(kk) + j];
    }
    wsync();
  }
  return true;
}

// solve L L^T x = b in place for nrhs right-hand sides x[r*ldx + i]
__device__ void chol_solve_w(const double* Lp, const double* dinv, int nf, double* x, int nrhs,
                             int ldx, int lane) {
  for (int j = 0; j < nf; ++j) {
    const double dj = dinv[j];
    for (int r = 0; r < nrhs; ++r) {
      const double yj = x[r * ldx + j] * dj;
      for (int i = j + 1 + lane; i < nf; i += WAVE) x[r * ldx + i] -= Lp[tri(i) + j] * yj;
      if (lane == 0) x[r * ldx + j] = yj;
    }
    wsync();
  }
  for (int j = nf - 1; j >= 0; --j) {
    const double dj = dinv[j];
    for (int r = 0; r < nrhs; ++r) {
      const double xj = x[r * ldx + j] * dj;
      for (int i = lane; i < j; i += WAVE) x[r * ldx + i] -= Lp[tri(j) + i] * xj;
      if (lane == 0) x[r * ldx + j] = xj;
    }
    wsync();
  }
}

#ifdef CTR_STAMPS
__device__ unsigned long long g_stamps[16];
#define STAMP(slot) do { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    if (wave == 0 && lane == 0) atomicAdd(&g_stamps[slot], now_ - t_prev_); t_prev_ = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define STAMP(slot) do {} while (0)
#endif

template <int ND, bool ISO, int NT, int W>
__global__ void __launch_bounds__(WAVE * W) refine_block_kernel(const KArgs k) {
#ifdef CTR_STAMPS
  unsigned long long t_prev_ = __builtin_amdgcn_s_memtime();
#endif
  using SM = SmemB<NT, W>;
  constexpr int NP = 2 + ND + (ISO ? 1 : ND);
  constexpr int NSZ = ISO ? 1 : ND;
  constexpr int LDC = SM::NVC;
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cl = k.order[blockIdx.x];
  const int f0 = k.feat_offset[cl], n = k.feat_offset[cl + 1] - f0;
  const double* params = k.params + (size_t)f0 * NP;
  double* pout = k.params_out + (size_t)f0 * NP;

  double *v = smem + SM::o_v, *vt = smem + SM::o_vt, *v0 = smem + SM::o_v0, *lo = smem + SM::o_lo,
         *hi = smem + SM::o_hi, *dl = smem + SM::o_dl, *w = smem + SM::o_w, *Mp = smem + SM::o_M,
         *Hp = smem + SM::o_rows, *cur = smem + SM::o_cur, *mco = smem + SM::o_mco,
         *fpar = smem + SM::o_fpar, *part = smem + SM::o_part;
  double *Cj = smem + SM::o_Cj, *Cjt = smem + SM::o_Cjt, *Y = smem + SM::o_Y;
  double *cv = smem + SM::o_small, *cvt = cv + 6, *mult = cv + 12, *Sc = cv + 24, *flag = cv + 60;
  int* fr = (int*)(smem + SM::o_fr);
  int* ctl = (int*)(smem + SM::o_ctl);
  double* myrows = smem + SM::o_rows + wave * SM::ROWS;

  Layout L;
  make_layout(k.prob, n, L);
  const int nv = L.nv;
  const int m = n_constraints(k.prob, n);
  const void* frame = (const char*)k.frames + (size_t)k.frame_index[cl] * k.frame_elems * dtype_size(k.frame_dtype);
  const int maxiter = k.prob.solver_maxiter > 0 ? k.prob.solver_maxiter : 100;
  const double xtol = k.prob.xtol > 0 ? k.prob.xtol : 1e-9;
  const double ftol = k.prob.ftol > 0 ? k.prob.ftol : 1e-14;
  int radius[ND];
  double inv_r2[ND];
  long fshape[ND];
#pragma unroll
  for (int a = 0; a < ND; ++a) {
    radius[a] = k.prob.radius[a];
    inv_r2[a] = 1. / ((double)radius[a] * (double)radius[a]);
    fshape[a] = k.shape[a];
  }
  // parameter kk of feature i at vector vv (vect_to_params, fitfunc.py:266-315)
  auto par = [&](const double* vv, int i, int kk) -> double {
    const int b = L.var_of[kk];
    if (b < 0) return cur[i * CTR_MAX_PARAMS + kk];
    return vv[b + (L.per_feat[kk] ? i : 0)];
  };
  // derived constants of every feature at vv: [0] signal [1..3] centre
  // [4..6] 1/size^2 [7..9] 2/size^2 [10..12] -2/size^3   (wave 0)
  bool size_is_var = false;
#pragma unroll
  for (int kk = 2 + ND; kk < NP; ++kk) size_is_var = size_is_var || L.var_of[kk] >= 0;
  auto fill_fpar = [&](const double* vv, bool sizes) {
    for (int i = lane; i < n; i += WAVE) {
      double* f = fpar + i * FP;
      f[0] = par(vv, i, 1);
#pragma unroll
      for (int a = 0; a < ND; ++a) {
        f[1 + a] = par(vv, i, 2 + a);
        if (sizes) {  // three f64 divisions per axis: only when a size actually changed
          const double sz = par(vv, i, ISO ? 2 + ND : 2 + ND + a);
          const double s2 = sz * sz;
          f[4 + a] = 1. / s2;
          f[7 + a] = 2. / s2;
          f[10 + a] = -2. / (s2 * sz);
        }
      }
    }
  };
  // masks.py:42-68 on the mask centres; wave-uniform
  auto window_of = [&](int* origin, int* wshape) -> bool {
    long wlo[ND], whi[ND];
    bool any = false;
    for (int i = 0; i < n; ++i) {
      long ci[ND];
      bool ok = true;
#pragma unroll
      for (int a = 0; a < ND; ++a) {
        ci[a] = (long)rint(mco[i * 3 + a]);
        if (!(ci[a] >= -(long)radius[a] && ci[a] < fshape[a] + radius[a])) ok = false;
      }
      if (!ok) continue;
#pragma unroll
      for (int a = 0; a < ND; ++a) {
        wlo[a] = (!any || ci[a] < wlo[a]) ? ci[a] : wlo[a];
        whi[a] = (!any || ci[a] > whi[a]) ? ci[a] : whi[a];
      }
      any = true;
    }
    if (!any) return false;
#pragma unroll
    for (int a = 0; a < ND; ++a) {
      long l = wlo[a] - radius[a], u = whi[a] + radius[a] + 1;
      l = l < 0 ? 0 : l;
      u = u > fshape[a] ? fshape[a] : u;
      origin[a] = (int)l;
      wshape[a] = (int)(u - l);
    }
    return true;
  };
  // cv[m], Cj[m][LDC] at vv (constraints.py:59-137); wave 0
  auto eval_constraints = [&](const double* vv, double* cvo, double* Cjo) {
    if (m == 0) return;
    const int npairs = k.prob.constraint_kind == CTR_CONS_DIMER ? 1
                     : k.prob.constraint_kind == CTR_CONS_TRIMER ? 3 : 6;
    double d2[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) {
      d2[q] = 0.;
      const int i0 = (q == 0 || q == 2 || q == 4) ? 0 : (q == 5 ? 2 : 1);
      const int i1 = q == 0 ? 1 : (q <= 2 ? 2 : 3);
      if (q < npairs) {
#pragma unroll
        for (int a = 0; a < ND; ++a) {
          const double t = (par(vv, i0, 2 + a) - par(vv, i1, 2 + a)) / k.prob.constraint_dist[a];
          d2[q] += t * t;
        }
      }
    }
    for (int e = lane; e < m * LDC; e += WAVE) Cjo[e] = 0.;
    wsync();
    const int q = lane;
    if (q < npairs) {
      int rank = q;
      double mine = 0.;
#pragma unroll
      for (int p = 0; p < 6; ++p) if (p == q) mine = d2[p];
      if (k.prob.constraint_kind == CTR_CONS_TETRAMER && ND == 2) {
        rank = 0;  // stable rank among the 6 squared distances (constraints.py:112)
#pragma unroll
        for (int p = 0; p < 6; ++p) rank += (d2[p] < mine || (d2[p] == mine && p < q)) ? 1 : 0;
      }
      if (rank < m) {
        const int i0 = (q == 0 || q == 2 || q == 4) ? 0 : (q == 5 ? 2 : 1);
        const int i1 = q == 0 ? 1 : (q <= 2 ? 2 : 3);
        cvo[rank] = 1. - mine;
#pragma unroll
        for (int a = 0; a < ND; ++a) {
          const int kk = 2 + a, b = L.var_of[kk];
          if (b < 0) continue;
          const double da = k.prob.constraint_dist[a];
          const double t = -2. * (par(vv, i0, kk) - par(vv, i1, kk)) / (da * da);
          Cjo[rank * LDC + b + (L.per_feat[kk] ? i0 : 0)] += t;
          Cjo[rank * LDC + b + (L.per_feat[kk] ? i1 : 0)] -= t;
        }
      }
    }
    wsync();
  };

  // ---- set-up (all threads) ---------------------------------------------------------
  bool finite = true;
  for (int e = tid; e < n * NP; e += WAVE * W) {
    const double x = params[e];
    pout[e] = x;  // failures keep their input (refine.py:408-418)
    cur[(e / NP) * CTR_MAX_PARAMS + (e % NP)] = x;
    if (!isfinite(x)) finite = false;
  }
  for (int e = tid; e < W * SM::ROWS; e += WAVE * W) smem[SM::o_rows + e] = 0.;
  for (int e = tid; e < n * 3; e += WAVE * W) {
    const int i = e / 3, a = e % 3;
    mco[e] = a < ND ? params[i * NP + 2 + a] : 0.;
  }
  {
    const double* low = k.low + (size_t)f0 * NP;
    const double* high = k.high + (size_t)f0 * NP;
#pragma unroll
    for (int kk = 0; kk < NP; ++kk) {
      const int b = L.var_of[kk];
      if (b < 0) continue;
      if (L.per_feat[kk]) {
        for (int i = tid; i < n; i += WAVE * W) {
          v0[b + i] = params[i * NP + kk];
          lo[b + i] = low[i * NP + kk];
          hi[b + i] = high[i * NP + kk];
        }
      } else if (tid == 0) {
        double s = 0., l = INFINITY, h = -INFINITY;
        for (int i = 0; i < n; ++i) {
          s += params[i * NP + kk];
          l = fmin(l, low[i * NP + kk]);
          h = fmax(h, high[i * NP + kk]);
        }
        v0[b] = s / n;
        lo[b] = l;
        hi[b] = h;
      }
    }
  }
  const int nonfinite = __syncthreads_or(!finite ? 1 : 0);

  // state of the solver, meaningful in wave 0 (uniform there)
  int status = nonfinite ? CTR_STATUS_NONFINITE : (n <= 0 ? CTR_STATUS_OUT_OF_BOUNDS : CTR_STATUS_OK);
  int round = 0, it = 0, iters = 0, Pround = 0;
  double mu = 1e-3, nu = 2., sigma = 0., S = 0., pred = 0., cn = 0., rms = NAN;
  bool last_acc = true;
  const double fm = k.fmax[k.frame_index[cl]];
  const double norm = fm * fm / k.prob.residual_factor;  // refine.py:354
  const double ms2 = k.prob.max_shift * k.prob.max_shift;

  // start of a round (wave 0): window, clipped start vector, derived constants
  auto begin_round = [&]() -> int {
    int origin[ND], wshape[ND];
    if (!window_of(origin, wshape)) { status = CTR_STATUS_OUT_OF_BOUNDS; return BP_FINISH; }
    bool infeasible = false;
    for (int i = lane; i < nv; i += WAVE) {
      if (lo[i] > hi[i]) infeasible = true;
      const double x = v0[i];
      vt[i] = x < lo[i] ? lo[i] : (x > hi[i] ? hi[i] : x);
    }
    if (lane < MAXC) mult[lane] = 0.;
    if (__ballot(infeasible) != 0ull) { status = CTR_STATUS_NO_CONVERGENCE; return BP_FINISH; }
    if (lane == 0) {
#pragma unroll
      for (int a = 0; a < ND; ++a) { ctl[1 + a] = origin[a]; ctl[4 + a] = wshape[a]; }
    }
    wsync();
    fill_fpar(vt, size_is_var || round == 0);
    it = 0;
    return BP_EVAL_INIT;
  };

  if (wave == 0) {
    int ph = status == CTR_STATUS_OK ? begin_round() : BP_FINISH;
    if (lane == 0) ctl[0] = ph;
  }
  __syncthreads();

  v4d acc[SM::NTILE];
  while (true) {
    const int phase = ctl[0];
    if (phase == BP_FINISH) break;
    double Sloc = 0.;
    int P = 0;
    if (phase == BP_EVAL_INIT || phase == BP_EVAL_TRIAL) {
      // ---- all waves: their share of the window at vt ----------------------------------
      int origin[ND], wshape[ND];
#pragma unroll
      for (int a = 0; a < ND; ++a) { origin[a] = ctl[1 + a]; wshape[a] = ctl[4 + a]; }
      const int w1 = wshape[ND - 2], w2 = wshape[ND - 1];
      const int npix = (ND == 3 ? wshape[0] : 1) * w1 * w2;
      const int bgvar = L.var_of[0];
      const double bg = par(vt, 0, 0);
      const float inv_w2 = 1.f / (float)w2, inv_w1 = 1.f / (float)w1;
      const bool big_window = npix >= (1 << 21);
#pragma unroll
      for (int t = 0; t < SM::NTILE; ++t) acc[t] = v4d{0., 0., 0., 0.};
      double* row = myrows + lane * SM::RS;
      for (int base = wave * WAVE; base < npix; base += WAVE * W) {
        const int q = base + lane;
        const bool valid = q < npix;
        int idx[ND];
        size_t off;
        {
          // q / w2 through the float reciprocal: exact for q < 2^21 (q + 0.5 is never
          // within 0.5 / w2 of a multiple of w2, far above the float rounding error)
          const int t = big_window ? q / w2 : (int)(((float)q + 0.5f) * inv_w2);
          const int x = q - t * w2;
          if (ND == 3) {
            const int z = big_window ? t / w1 : (int)(((float)t + 0.5f) * inv_w1);
            const int y = t - z * w1;
            idx[0] = z; idx[1] = y; idx[ND - 1] = x;
            off = ((size_t)(z + origin[0]) * fshape[1] + (y + origin[1])) * fshape[ND - 1] + (x + origin[ND - 1]);
          } else {
            idx[0] = t; idx[ND - 1] = x;
            off = (size_t)(t + origin[0]) * fshape[ND - 1] + (x + origin[ND - 1]);
          }
        }
        bool any = false;
        double res = 0.;
        double shared[CTR_MAX_PARAMS];
#pragma unroll
        for (int kk = 0; kk < CTR_MAX_PARAMS; ++kk) shared[kk] = 0.;
        for (int i = 0; i < n; ++i) {
          double d[1 + ND + NSZ];
#pragma unroll
          for (int t = 0; t < 1 + ND + NSZ; ++t) d[t] = 0.;
          bool in = false;
          if (valid) {
            double rel[ND];
#pragma unroll
            for (int a = 0; a < ND; ++a) rel[a] = mco[i * 3 + a] - (double)origin[a];
            in = in_mask<ND>(idx, rel, inv_r2, radius);
          }
          if (in) {
            const double* f = fpar + i * FP;
            if (!any) {
              any = true;
              res = load_pixel(frame, k.frame_dtype, off) - bg;
            }
            double r2 = 0., dd[ND];
#pragma unroll
            for (int a = 0; a < ND; ++a) {
              dd[a] = (double)(idx[a] + origin[a]) - f[1 + a];
              r2 += dd[a] * dd[a] * f[4 + a];
            }
            const double gv = exp(-0.5 * ND * r2);  // fitfunc.py:112-118
            const double sig = f[0];
            const double sdg = sig * (0.5 * ND) * gv;  // -signal * dg/dr2
            res -= sig * gv;
            d[0] = -gv;
            double q2 = 0.;
#pragma unroll
            for (int a = 0; a < ND; ++a) {
              d[1 + a] = sdg * (-dd[a] * f[7 + a]);
              if (ISO) q2 += dd[a] * dd[a];
              else d[1 + ND + a] = sdg * (dd[a] * dd[a] * f[10 + a]);
            }
            if (ISO) d[1 + ND] = sdg * (q2 * f[10]);
          }
#pragma unroll
          for (int kk = 1; kk < NP; ++kk) {
            const int b = L.var_of[kk];
            if (b < 0) continue;
            if (L.per_feat[kk]) row[b + i] = d[kk - 1];
            else shared[kk] += d[kk - 1];
          }
        }
        const bool good = any && (res == res);  // nansum (fitfunc.py:449,483)
#pragma unroll
        for (int kk = 1; kk < NP; ++kk) {
          const int b = L.var_of[kk];
          if (b >= 0 && !L.per_feat[kk]) row[b] = shared[kk];
        }
        if (bgvar >= 0) row[bgvar] = good ? -1. : 0.;
        row[nv] = good ? res : 0.;
        if (any && !good) {
          for (int j = 0; j < nv; ++j) row[j] = 0.;
        }
        const unsigned long long bal = __ballot(any);
        P += __popcll(bal);
        if (good) Sloc += res * res;
        wsync();
        if (bal != 0ull) {
          const int kr = lane >> 4, cc = lane & 15;
#pragma unroll 4
          for (int s = 0; s < 16; ++s) {
            const double* rp = myrows + (4 * s + kr) * SM::RS + cc;
            double val[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) val[t] = rp[16 * t];
            int tt = 0;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
              for (int tj = 0; tj <= ti; ++tj) {
                acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(val[ti], val[tj], acc[tt], 0, 0, 0);
                ++tt;
              }
          }
        }
        wsync();
      }
      Sloc = wave_sum(Sloc);
      if (W > 1) {
        // partial accumulators meet in LDS: wave w parks its tiles in its own row tile
        if (wave != 0) {
#pragma unroll
          for (int t = 0; t < SM::NTILE; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) myrows[(t * 4 + r) * WAVE + lane] = acc[t][r];
        }
        if (lane == 0) { part[2 * wave] = Sloc; part[2 * wave + 1] = (double)P; }
      }
    }
    STAMP(0);
    if (W > 1) __syncthreads();
    STAMP(1);

    if (wave == 0) {
      // ---- wave 0: sum, accept / reject, next step ---------------------------------------
      int next = phase;
      bool failed = false;
      double St = Sloc;
      if (phase == BP_EVAL_INIT || phase == BP_EVAL_TRIAL) {
        if (W > 1) {
#pragma unroll
          for (int ww = 1; ww < W; ++ww) {
            const double* pr = smem + SM::o_rows + ww * SM::ROWS;
#pragma unroll
            for (int t = 0; t < SM::NTILE; ++t)
#pragma unroll
              for (int r = 0; r < 4; ++r) acc[t][r] += pr[(t * 4 + r) * WAVE + lane];
            St += part[2 * ww];
            P += (int)part[2 * ww + 1];
          }
          wsync();
          // the parked tiles overwrote columns of the row tiles: clear what rows never rewrite
          // (columns > nv are read by the MFMA but only feed entries nobody looks at)
        }
        eval_constraints(vt, cvt, Cjt);
      }
      bool accept = false;
      if (phase == BP_EVAL_INIT) {
        if (P == 0) { status = CTR_STATUS_OUT_OF_BOUNDS; failed = true; }
        else if (!isfinite(St)) { status = CTR_STATUS_NO_CONVERGENCE; failed = true; }
        mu = 1e-3; nu = 2.; sigma = 0.; last_acc = true;
        Pround = P;
        accept = !failed;
      } else if (phase == BP_EVAL_TRIAL) {
        double cnt = 0.;
        for (int r = 0; r < m; ++r) cnt += fabs(cvt[r]);
        double act = 0.5 * (S - St) + (m ? sigma * (cn - cnt) : 0.);
        act = bcast0(act);
        if (isfinite(St) && pred > 0. && act > 0.) {
          const double rho = act / pred, t = 2. * rho - 1.;
          const double f = 1. - t * t * t;
          mu *= f > 1. / 3. ? f : 1. / 3.;
          nu = 2.;
          accept = true;
          last_acc = true;
        } else {
          mu *= nu; nu *= 2.; last_acc = false;
          if (mu > 1e30) { status = CTR_STATUS_NO_CONVERGENCE; failed = true; }
        }
      }
      if (accept) {
        for (int i = lane; i < nv; i += WAVE) v[i] = vt[i];
        for (int e = lane; e < m * LDC; e += WAVE) Cj[e] = Cjt[e];
        if (lane < m) cv[lane] = cvt[lane];
        // acc -> packed lower triangle; D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
        {
          const int cc = lane & 15, r0 = lane >> 4;
          int tt = 0;
#pragma unroll
          for (int ti = 0; ti < NT; ++ti)
#pragma unroll
            for (int tj = 0; tj <= ti; ++tj) {
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int gi = 16 * ti + r0 + 4 * r, gj = 16 * tj + cc;
                if (gi >= gj) Mp[tri(gi) + gj] = acc[tt][r];
              }
              ++tt;
            }
        }
        S = St;
        wsync();
      }
      STAMP(2);
      if (!failed && it >= maxiter) { status = CTR_STATUS_NO_CONVERGENCE; failed = true; }
      bool converged = false;
      if (!failed) {
        ++it;
        ++iters;
        // active set: fixed if at a bound and the Lagrangian gradient pushes outward
        int nf = 0;
        for (int b0 = 0; b0 < nv; b0 += WAVE) {
          const int i = b0 + lane;
          bool fre = false;
          if (i < nv) {
            double gl = Mp[tri(nv) + i];
            for (int r = 0; r < m; ++r) gl += Cj[r * LDC + i] * mult[r];
            const bool fixed = (lo[i] == hi[i]) || (v[i] <= lo[i] && gl > 0.) || (v[i] >= hi[i] && gl < 0.);
            fre = !fixed;
          }
          const unsigned long long bal = __ballot(fre);
          if (fre) fr[nf + __popcll(bal & ((1ull << lane) - 1ull))] = i;
          nf += __popcll(bal);
        }
        wsync();
        STAMP(3);
        bool ok_step = true;
        bool have_dl = false;
        if (nf == 0) {
          converged = true;
        } else if (NT <= 2 && m == 0) {
         if constexpr (NT <= 2) {
          // small unconstrained system: one column per lane, in registers
          bool is_free = false;
          if (lane < nv) {
            const double gl = Mp[tri(nv) + lane];
            is_free = !((lo[lane] == hi[lane]) || (v[lane] <= lo[lane] && gl > 0.) || (v[lane] >= hi[lane] && gl < 0.));
          }
          double x_own;
          ok_step = column_solve<16 * NT>(Mp, nv, mu, is_free, lane, x_own);
          if (ok_step && lane < nv) dl[lane] = -x_own;
          wsync();
          have_dl = true;
         }
        } else {
          for (int e = lane; e < tri(nf); e += WAVE) {
            int a = (int)((sqrt(8. * e + 1.) - 1.) * 0.5);
            while (tri(a + 1) <= e) ++a;
            while (tri(a) > e) --a;
            const int b = e - tri(a);
            double h = Msym(Mp, fr[a], fr[b]);
            if (a == b) h += mu * (h > 1e-300 ? h : 1.);
            Hp[e] = h;
          }
          wsync();
          ok_step = chol_factor_w(Hp, dl, nf, lane);  // dl doubles as 1/diag until the step is built
          if (ok_step) {
            for (int a = lane; a < nf; a += WAVE) {
              w[a] = Mp[tri(nv) + fr[a]];
              for (int r = 0; r < m; ++r) Y[r * LDC + a] = Cj[r * LDC + fr[a]];
            }
            wsync();
            chol_solve_w(Hp, dl, nf, w, 1, 0, lane);
            if (m) {
              chol_solve_w(Hp, dl, nf, Y, m, LDC, lane);
              // (C H^-1 C^T) mult = c - C H^-1 g   (range-space form of the KKT step)
              if (lane < m * m) {
                const int r = lane / m, s = lane % m;
                double t = 0.;
                for (int a = 0; a < nf; ++a) t += Cj[r * LDC + fr[a]] * Y[s * LDC + a];
                Sc[r * MAXC + s] = t;
              }
              if (lane < m) {
                double t = cv[lane];
                for (int a = 0; a < nf; ++a) t -= Cj[lane * LDC + fr[a]] * w[a];
                mult[lane] = t;
              }
              wsync();
              if (lane == 0) {
                double tr = 0.;
                for (int r = 0; r < m; ++r) tr += Sc[r * MAXC + r];
                for (int r = 0; r < m; ++r) Sc[r * MAXC + r] += 1e-14 * tr + 1e-300;
                bool okc = true;
                for (int j = 0; j < m && okc; ++j) {
                  double d = Sc[j * MAXC + j];
                  for (int q = 0; q < j; ++q) d -= Sc[j * MAXC + q] * Sc[j * MAXC + q];
                  if (!(d > 0.) || !isfinite(d)) { okc = false; break; }
                  d = sqrt(d);
                  Sc[j * MAXC + j] = d;
                  for (int i = j + 1; i < m; ++i) {
                    double s = Sc[i * MAXC + j];
                    for (int q = 0; q < j; ++q) s -= Sc[i * MAXC + q] * Sc[j * MAXC + q];
                    Sc[i * MAXC + j] = s / d;
                  }
                }
                if (okc) {
                  for (int i = 0; i < m; ++i) {
                    double s = mult[i];
                    for (int q = 0; q < i; ++q) s -= Sc[i * MAXC + q] * mult[q];
                    mult[i] = s / Sc[i * MAXC + i];
                  }
                  for (int i = m - 1; i >= 0; --i) {
                    double s = mult[i];
                    for (int q = i + 1; q < m; ++q) s -= Sc[q * MAXC + i] * mult[q];
                    mult[i] = s / Sc[i * MAXC + i];
                  }
                } else {
                  for (int i = 0; i < m; ++i) mult[i] = 0.;
                }
                flag[0] = okc ? 1. : 0.;
              }
              wsync();
              ok_step = flag[0] != 0.;
            }
          }
        }
        STAMP(4);
        if (!converged) {
          if (!ok_step) {
            mu *= nu; nu *= 2.; last_acc = false;
            if (mu > 1e30) { status = CTR_STATUS_NO_CONVERGENCE; failed = true; }
            next = BP_STEP_ONLY;
          } else {
            if (!have_dl) {
              for (int i = lane; i < nv; i += WAVE) dl[i] = 0.;
              wsync();
              for (int a = lane; a < nf; a += WAVE) {
                double t = w[a];
                for (int r = 0; r < m; ++r) t += Y[r * LDC + a] * mult[r];
                dl[fr[a]] = -t;
              }
              wsync();
            }
            double stepmax = 0.;
            for (int i = lane; i < nv; i += WAVE) {
              double t = v[i] + dl[i];
              t = t < lo[i] ? lo[i] : (t > hi[i] ? hi[i] : t);
              vt[i] = t;
              const double d = t - v[i];
              dl[i] = d;
              stepmax = fmax(stepmax, fabs(d) / (fabs(v[i]) + 1.));
            }
            wsync();
            stepmax = wave_max(stepmax);
            double partial = 0.;
            for (int i = lane; i < nv; i += WAVE) {
              double t = 0.;
              for (int j = 0; j < nv; ++j) t += Msym(Mp, i, j) * dl[j];
              partial += dl[i] * (Mp[tri(nv) + i] + 0.5 * t);
            }
            pred = -wave_sum(partial);
            cn = 0.;
            if (m) {
              double cn_lin = 0., mmax = 0.;
              for (int r = 0; r < m; ++r) {
                double t = cv[r];
                for (int i = 0; i < nv; ++i) t += Cj[r * LDC + i] * dl[i];
                cn += fabs(cv[r]);
                cn_lin += fabs(t);
                mmax = fmax(mmax, fabs(mult[r]));
              }
              if (sigma < 2. * mmax) sigma = 2. * mmax;
              pred += sigma * (cn - cn_lin);
            }
            pred = bcast0(pred);
            stepmax = bcast0(stepmax);
            const bool feasible = (m == 0) || (cn <= 1e-10);
            converged = feasible && ((last_acc && stepmax <= xtol) || pred <= ftol * (0.5 * S) + 1e-300);
            next = BP_EVAL_TRIAL;
          }
        }
      }
      STAMP(5);
      if (failed) next = BP_FINISH;
      if (converged) {
        // end of a round: vect_to_params and the shift test (refine.py:379-388)
        rms = sqrt(((S / (double)Pround) / norm) / k.prob.residual_factor);
        bool moved = false;
        for (int i = lane; i < n; i += WAVE) {
          double d2 = 0.;
#pragma unroll
          for (int kk = 0; kk < NP; ++kk) {
            const int b = L.var_of[kk];
            if (b >= 0) cur[i * CTR_MAX_PARAMS + kk] = v[b + (L.per_feat[kk] ? i : 0)];
          }
#pragma unroll
          for (int a = 0; a < ND; ++a) {
            const double d = cur[i * CTR_MAX_PARAMS + 2 + a] - mco[i * 3 + a];
            d2 += d * d;
          }
          if (!(d2 < ms2)) moved = true;
        }
        const bool any_moved = __ballot(moved) != 0ull;
        wsync();
        ++round;
        if (!any_moved || round >= k.prob.max_iter) {
          if (rms > k.prob.max_rms_dev) status = CTR_STATUS_RMS_DEV;  // refine.py:391
          next = BP_FINISH;
        } else {
          for (int e = lane; e < n * 3; e += WAVE) {
            const int i = e / 3, a = e % 3;
            if (a < ND) mco[e] = cur[i * CTR_MAX_PARAMS + 2 + a];
          }
          wsync();
          next = begin_round();
        }
      } else if (next == BP_EVAL_TRIAL) {
        fill_fpar(vt, size_is_var);
      }
      STAMP(6);
      if (lane == 0) ctl[0] = next;
    }
    __syncthreads();
    STAMP(7);
  }

  if (wave == 0) {
    const bool ok = status == CTR_STATUS_OK;
    if (ok)
      for (int e = lane; e < n * NP; e += WAVE) pout[e] = cur[(e / NP) * CTR_MAX_PARAMS + (e % NP)];
    if (lane == 0) {
      k.status[cl] = status;
      k.cost[cl] = ok ? rms : NAN;
      k.n_rounds[cl] = status == CTR_STATUS_NONFINITE || n <= 0 ? 0
                     : (ok || status == CTR_STATUS_RMS_DEV ? round : round + 1);
      k.n_iter[cl] = iters;
    }
  }
}


// ---- small clusters: 16 lanes per cluster, everything in registers ------------------
//
// Singles and pairs with the default parameter modes (background per cluster,
// signal and positions per feature, sizes constant: fitfunc.py:356,379-387) are
// >95 % of the clusters of a typical frame.  For them the normal equations are
// tiny (4..9 variables), so four clusters share one wavefront: each 16-lane
// group runs its own LM state machine, accumulates its [J r]^T [J r] in
// registers, all-reduces it inside the 16-lane DPP row (no LDS, no MFMA padding)
// and solves it redundantly in registers.  Groups pull clusters from a global
// work counter, so a slow cluster only delays its own group.
//
// Variable order (fitfunc.py:207-263): [bg, s_0.., pos(axis 0)_0.., pos(axis 1)_0.., ...].


template <int CTRL>
__device__ __forceinline__ double dpp_f64(double x) {
  const long long b = __double_as_longlong(x);
  int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
  lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
  hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
// all-reduce inside each 16-lane DPP row
__device__ __forceinline__ double row_sum(double x) {
  x += dpp_f64<0xB1>(x);   // quad_perm [1,0,3,2]
  x += dpp_f64<0x4E>(x);   // quad_perm [2,3,0,1]
  x += dpp_f64<0x141>(x);  // row_half_mirror
  x += dpp_f64<0x140>(x);  // row_mirror
  return x;
}
// all-reduce inside a group of SG lanes (16 = one DPP row, 64 = the whole wave)
template <int SG>
__device__ __forceinline__ double group_sum(double x) {
  x = row_sum(x);
  if (SG == 64) {
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
  }
  return x;
}

enum { PH_FETCH = 0, PH_EVAL_INIT = 1, PH_EVAL_TRIAL = 2, PH_STEP_ONLY = 3, PH_DONE = 4 };

// SG = lanes per cluster: 16 (four clusters per wave; singles) or 64 (pairs: a
// quarter of the per-iteration latency, which is what bounds the slowest pair).
template <int ND, int NF, bool ISO, int SG>
__global__ void __launch_bounds__(WAVE) refine_small_kernel(const KArgs k, int* __restrict__ counter) {
  constexpr int NV = 1 + NF * (1 + ND);
  constexpr int NR = NV + 1;               // row length incl. the residual
  constexpr int NM = NR * (NR + 1) / 2;    // packed upper triangle
  constexpr int NP = 2 + ND + (ISO ? 1 : ND);
  const int lane = threadIdx.x, sub = lane & (SG - 1), grp = lane / SG;
  // per group: Mcur[NM] v0[NV] lo[NV] hi[NV]
  constexpr int GS = NM + 3 * NV;
  __shared__ double lds[(WAVE / SG) * GS];
  double* Mcur = lds + grp * GS;
  double* v0 = Mcur + NM;
  double* lo = v0 + NV;
  double* hi = lo + NV;

  const int maxiter = k.prob.solver_maxiter > 0 ? k.prob.solver_maxiter : 100;
  const double xtol = k.prob.xtol > 0 ? k.prob.xtol : 1e-9;
  const double ftol = k.prob.ftol > 0 ? k.prob.ftol : 1e-14;
  const double ms2 = k.prob.max_shift * k.prob.max_shift;
  double inv_r2[ND];
  int radius[ND];
#pragma unroll
  for (int a = 0; a < ND; ++a) {
    radius[a] = k.prob.radius[a];
    inv_r2[a] = 1. / ((double)radius[a] * (double)radius[a]);
  }
  long fshape[ND];
#pragma unroll
  for (int a = 0; a < ND; ++a) fshape[a] = k.shape[a];

  // per-group state (replicated in the group's 16 lanes)
  int phase = PH_FETCH, cl = -1, f0 = 0;
  int round = 0, it = 0, iters = 0, status = CTR_STATUS_OK, Pround = 0;
  int origin[ND], wshape[ND], npix = 0;
  double v[NV], vt[NV];
  double mco[NF][ND], isz2[NF][ND], cst[NF][CTR_MAX_PARAMS];  // mask centres, 1/size^2, p0 rows
  double mu = 1e-3, nu = 2., S = 0., pred = 0., norm = 1., rms = NAN;
  bool last_acc = true;
  const char* frame = nullptr;

  while (true) {
    // ---- 1. idle groups pull the next cluster -----------------------------------
    if (phase == PH_FETCH) {
      int id = 0;
      if (sub == 0) id = atomicAdd(counter, 1);
      id = __shfl(id, lane & ~(SG - 1));
      if (id >= k.n_bin) {
        phase = PH_DONE;
      } else {
        cl = k.order[id];
        f0 = k.feat_offset[cl];
        const double* params = k.params + (size_t)f0 * NP;
        const double* low = k.low + (size_t)f0 * NP;
        const double* high = k.high + (size_t)f0 * NP;
        bool finite = true;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
#pragma unroll
          for (int kk = 0; kk < CTR_MAX_PARAMS; ++kk) {
            cst[i][kk] = kk < NP ? params[i * NP + kk] : 0.;
            if (!isfinite(cst[i][kk])) finite = false;
          }
#pragma unroll
          for (int a = 0; a < ND; ++a) {
            mco[i][a] = cst[i][2 + a];
            const double sz = cst[i][ISO ? 2 + ND : 2 + ND + a];
            isz2[i][a] = 1. / (sz * sz);
          }
        }
        // start vector: mean background (refine.py:361), loosest background bound (fitfunc.py:554-557)
        if (sub == 0) {
          double sb = 0., lb = INFINITY, hb = -INFINITY;
#pragma unroll
          for (int i = 0; i < NF; ++i) {
            sb += cst[i][0];
            lb = fmin(lb, low[i * NP]);
            hb = fmax(hb, high[i * NP]);
            v0[1 + i] = cst[i][1];
            lo[1 + i] = low[i * NP + 1];
            hi[1 + i] = high[i * NP + 1];
#pragma unroll
            for (int a = 0; a < ND; ++a) {
              v0[1 + NF + a * NF + i] = cst[i][2 + a];
              lo[1 + NF + a * NF + i] = low[i * NP + 2 + a];
              hi[1 + NF + a * NF + i] = high[i * NP + 2 + a];
            }
          }
          v0[0] = sb / NF;
          lo[0] = lb;
          hi[0] = hb;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        frame = (const char*)k.frames + (size_t)k.frame_index[cl] * k.frame_elems * dtype_size(k.frame_dtype);
        const double fm = k.fmax[k.frame_index[cl]];
        norm = fm * fm / k.prob.residual_factor;
        round = 0; iters = 0; status = CTR_STATUS_OK; rms = NAN;
        phase = PH_EVAL_INIT;
        if (!finite) { status = CTR_STATUS_NONFINITE; phase = PH_FETCH + 100; }
      }
    }
    // (re)start a round: window from the mask centres, trial = clipped start vector
    if (phase == PH_EVAL_INIT) {
      long wlo[ND], whi[ND];
      bool any = false;
#pragma unroll
      for (int i = 0; i < NF; ++i) {
        long ci[ND];
        bool ok = true;
#pragma unroll
        for (int a = 0; a < ND; ++a) {
          ci[a] = (long)rint(mco[i][a]);
          if (!(ci[a] >= -(long)radius[a] && ci[a] < fshape[a] + radius[a])) ok = false;
        }
        if (ok) {
#pragma unroll
          for (int a = 0; a < ND; ++a) {
            wlo[a] = (!any || ci[a] < wlo[a]) ? ci[a] : wlo[a];
            whi[a] = (!any || ci[a] > whi[a]) ? ci[a] : whi[a];
          }
          any = true;
        }
      }
      if (!any) {
        status = CTR_STATUS_OUT_OF_BOUNDS;
        phase = PH_FETCH + 100;
      } else {
        npix = 1;
#pragma unroll
        for (int a = 0; a < ND; ++a) {
          long l = wlo[a] - radius[a], u = whi[a] + radius[a] + 1;
          l = l < 0 ? 0 : l;
          u = u > fshape[a] ? fshape[a] : u;
          origin[a] = (int)l;
          wshape[a] = (int)(u - l);
          npix *= wshape[a];
        }
        bool infeasible = false;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const double x = v0[j], l = lo[j], h = hi[j];
          if (l > h) infeasible = true;
          vt[j] = x < l ? l : (x > h ? h : x);
        }
        it = 0;
        if (infeasible) { status = CTR_STATUS_NO_CONVERGENCE; phase = PH_FETCH + 100; }
      }
    }
    if (__all(phase == PH_DONE)) break;

    // ---- 2. one pass over the window at vt: M = [J r]^T [J r], P ----------------
    double M[NM];
#pragma unroll
    for (int e = 0; e < NM; ++e) M[e] = 0.;
    int P = 0;
    const bool evaluating = (phase == PH_EVAL_INIT || phase == PH_EVAL_TRIAL);
    const int npix_here = evaluating ? npix : 0;
    {
      const int w_last = wshape[ND - 1];
      const float inv_w2 = 1.f / (float)w_last;
      const float inv_w1 = ND == 3 ? 1.f / (float)wshape[1] : 1.f;
      const bool big_window = npix >= (1 << 21);
      const double bg = vt[0];
      for (int base = 0; __any(base < npix_here); base += SG) {
        const int q = base + sub;
        if (q < npix_here) {
          int idx[ND];
          size_t off;
          {
            const int t = big_window ? q / w_last : (int)(((float)q + 0.5f) * inv_w2);
            const int x = q - t * w_last;
            if (ND == 3) {
              const int z = big_window ? t / wshape[1] : (int)(((float)t + 0.5f) * inv_w1);
              const int y = t - z * wshape[1];
              idx[0] = z; idx[1] = y; idx[ND - 1] = x;
              off = ((size_t)(z + origin[0]) * fshape[1] + (y + origin[1])) * fshape[ND - 1] + (x + origin[ND - 1]);
            } else {
              idx[0] = t; idx[ND - 1] = x;
              off = (size_t)(t + origin[0]) * fshape[ND - 1] + (x + origin[ND - 1]);
            }
          }
          double row[NR];
#pragma unroll
          for (int j = 0; j < NR; ++j) row[j] = 0.;
          bool any = false;
          double res = 0.;
#pragma unroll
          for (int i = 0; i < NF; ++i) {
            double rel[ND];
#pragma unroll
            for (int a = 0; a < ND; ++a) rel[a] = mco[i][a] - (double)origin[a];
            if (in_mask<ND>(idx, rel, inv_r2, radius)) {
              any = true;
              double r2 = 0., dd[ND];
#pragma unroll
              for (int a = 0; a < ND; ++a) {
                dd[a] = (double)(idx[a] + origin[a]) - vt[1 + NF + a * NF + i];
                r2 += dd[a] * dd[a] * isz2[i][a];
              }
              const double gv = exp(-0.5 * ND * r2);
              const double sig = vt[1 + i];
              res -= sig * gv;
              row[1 + i] = -gv;
              const double sng = -sig * (double)ND * gv;
#pragma unroll
              for (int a = 0; a < ND; ++a) row[1 + NF + a * NF + i] = sng * dd[a] * isz2[i][a];
            }
          }
          if (any) {
            res += load_pixel(frame, k.frame_dtype, off) - bg;
            ++P;
            if (res == res) {
              row[0] = -1.;
              row[NV] = res;
              int e = 0;
#pragma unroll
              for (int p = 0; p < NR; ++p)
#pragma unroll
                for (int c2 = p; c2 < NR; ++c2) { M[e] += row[p] * row[c2]; ++e; }
            }
          }
        }
      }
    }
    if (__any(evaluating)) {
#pragma unroll
      for (int e = 0; e < NM; ++e) M[e] = group_sum<SG>(M[e]);
      P = (int)group_sum<SG>((double)P);
    }

    // ---- 3. accept / reject, next step, convergence, rounds ---------------------
    if (phase == PH_EVAL_INIT || phase == PH_EVAL_TRIAL || phase == PH_STEP_ONLY) {
      const double St = M[NM - 1];
      bool failed = false;
      if (phase == PH_EVAL_INIT) {
        if (P == 0) { status = CTR_STATUS_OUT_OF_BOUNDS; failed = true; }
        else if (!isfinite(St)) { status = CTR_STATUS_NO_CONVERGENCE; failed = true; }
        mu = 1e-3; nu = 2.; last_acc = true;
        Pround = P;
      }
      bool accept = phase == PH_EVAL_INIT;
      if (phase == PH_EVAL_TRIAL) {
        const double act = 0.5 * (S - St);
        if (isfinite(St) && pred > 0. && act > 0.) {
          const double rho = act / pred, t = 2. * rho - 1.;
          const double f = 1. - t * t * t;
          mu *= f > 1. / 3. ? f : 1. / 3.;
          nu = 2.;
          accept = true;
          last_acc = true;
        } else {
          mu *= nu; nu *= 2.; last_acc = false;
          if (mu > 1e30) { status = CTR_STATUS_NO_CONVERGENCE; failed = true; }
        }
      }
      if (accept && !failed) {
#pragma unroll
        for (int j = 0; j < NV; ++j) v[j] = vt[j];
        S = St;
        if (sub == 0) {
#pragma unroll
          for (int e = 0; e < NM; ++e) Mcur[e] = M[e];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      if (!failed && it >= maxiter) { status = CTR_STATUS_NO_CONVERGENCE; failed = true; }
      bool converged = false;
      if (!failed) {
        ++it;
        ++iters;
        // Mcur: upper triangle packed row-major over [J r]; g = last column
        auto Mc = [&](int p, int c2) -> double {
          const int a = p < c2 ? p : c2, b = p < c2 ? c2 : p;
          return Mcur[a * NR - (a * (a - 1)) / 2 + (b - a)];
        };
        double g[NV];
#pragma unroll
        for (int p = 0; p < NV; ++p) g[p] = Mc(p, NV);
        // active set folded into the system: fixed variables get an identity row
        bool fixed[NV];
        int nfree = 0;
#pragma unroll
        for (int j = 0; j < NV; ++j) {
          const double l = lo[j], h = hi[j];
          fixed[j] = (l == h) || (v[j] <= l && g[j] > 0.) || (v[j] >= h && g[j] < 0.);
          nfree += fixed[j] ? 0 : 1;
        }
        if (nfree == 0) {
          converged = true;
        } else {
          double L[NV][NV], rhs[NV], dinv[NV];
          bool okc = true;
#pragma unroll
          for (int p = 0; p < NV; ++p) {
#pragma unroll
            for (int c2 = 0; c2 <= p; ++c2) {
              double h = (fixed[p] || fixed[c2]) ? 0. : Mc(p, c2);
              if (p == c2) h = fixed[p] ? 1. : h + mu * (h > 1e-300 ? h : 1.);
              L[p][c2] = h;
            }
            rhs[p] = fixed[p] ? 0. : g[p];
          }
#pragma unroll
          for (int j = 0; j < NV; ++j) {
            double d = L[j][j];
#pragma unroll
            for (int q2 = 0; q2 < j; ++q2) d -= L[j][q2] * L[j][q2];
            if (!(d > 0.) || !isfinite(d)) okc = false;
            const double di = 1. / sqrt(d);
            dinv[j] = di;
#pragma unroll
            for (int i = j + 1; i < NV; ++i) {
              double s = L[i][j];
#pragma unroll
              for (int q2 = 0; q2 < j; ++q2) s -= L[i][q2] * L[j][q2];
              L[i][j] = s * di;
            }
          }
          if (!okc) {
            mu *= nu; nu *= 2.; last_acc = false;
            if (mu > 1e30) { status = CTR_STATUS_NO_CONVERGENCE; failed = true; }
            phase = PH_STEP_ONLY;
          } else {
#pragma unroll
            for (int i = 0; i < NV; ++i) {
              double s = rhs[i];
#pragma unroll
              for (int q2 = 0; q2 < i; ++q2) s -= L[i][q2] * rhs[q2];
              rhs[i] = s * dinv[i];
            }
#pragma unroll
            for (int i = NV - 1; i >= 0; --i) {
              double s = rhs[i];
#pragma unroll
              for (int q2 = i + 1; q2 < NV; ++q2) s -= L[q2][i] * rhs[q2];
              rhs[i] = s * dinv[i];
            }
            double dl[NV], stepmax = 0.;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
              double t = v[j] - rhs[j];
              const double l = lo[j], h = hi[j];
              t = t < l ? l : (t > h ? h : t);
              vt[j] = t;
              dl[j] = t - v[j];
              stepmax = fmax(stepmax, fabs(dl[j]) / (fabs(v[j]) + 1.));
            }
            double acc = 0.;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
              double t = 0.;
#pragma unroll
              for (int j = 0; j < NV; ++j) t += Mc(i, j) * dl[j];
              acc += dl[i] * (g[i] + 0.5 * t);
            }
            pred = -acc;
            converged = (last_acc && stepmax <= xtol) || pred <= ftol * (0.5 * S) + 1e-300;
            phase = PH_EVAL_TRIAL;
          }
        }
      }
      if (failed) phase = PH_FETCH + 100;
      if (converged) {
        // end of a round (refine.py:376-388)
        rms = sqrt(((S / (double)Pround) / norm) / k.prob.residual_factor);
        bool moved = false;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
          double d2 = 0.;
#pragma unroll
          for (int a = 0; a < ND; ++a) {
            const double d = v[1 + NF + a * NF + i] - mco[i][a];
            d2 += d * d;
          }
          if (!(d2 < ms2)) moved = true;
        }
        ++round;
        if (!moved || round >= k.prob.max_iter) {
          if (rms > k.prob.max_rms_dev) status = CTR_STATUS_RMS_DEV;
          phase = PH_FETCH + 100;
        } else {
#pragma unroll
          for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int a = 0; a < ND; ++a) mco[i][a] = v[1 + NF + a * NF + i];
          phase = PH_EVAL_INIT;
        }
      }
    }
    // ---- 4. write the outputs of a finished cluster --------------------------------
    if (phase == PH_FETCH + 100) {
      double* pout = k.params_out + (size_t)f0 * NP;
      const bool ok = status == CTR_STATUS_OK;
      if (sub < NF) {
        const int i = sub;
#pragma unroll
        for (int ii = 0; ii < NF; ++ii)
          if (ii == i) {
#pragma unroll
            for (int kk = 0; kk < NP; ++kk) {
              double x = cst[ii][kk];
              if (ok) {
                if (kk == 0) x = v[0];
                else if (kk == 1) x = v[1 + ii];
                else if (kk < 2 + ND) {
#pragma unroll
                  for (int a = 0; a < ND; ++a)
                    if (kk == 2 + a) x = v[1 + NF + a * NF + ii];
                }
              }
              pout[ii * NP + kk] = x;
            }
          }
      }
      if (sub == 0) {
        k.status[cl] = status;
        k.cost[cl] = ok ? rms : NAN;
        k.n_rounds[cl] = status == CTR_STATUS_NONFINITE ? 0 : (ok || status == CTR_STATUS_RMS_DEV ? round : round + 1);
        k.n_iter[cl] = iters;
      }
      phase = PH_FETCH;
    }
  }
}

// clusters the engine cannot take (too many variables / features)
__global__ void mark_kernel(const KArgs k, int code) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= k.n_bin) return;
  const int cl = k.order[t];
  const int np = k.prob.n_params;
  for (int e = k.feat_offset[cl] * np; e < k.feat_offset[cl + 1] * np; ++e) k.params_out[e] = k.params[e];
  k.status[cl] = code;
  k.cost[cl] = NAN;
  k.n_rounds[cl] = 0;
  k.n_iter[cl] = 0;
}


// ---- cluster labelling (reference find.py:72-93): which features are fitted together ----
// Features of one frame closer than `separation` (per-axis scaled Euclidean distance <= 1,
// the criterion of cKDTree(pos / separation).query_pairs(1)) share a cluster.  One workgroup
// per frame; label propagation to the smallest row index of the cluster until nothing
// changes (bounded by the number of features of the frame).  The label is canonical (the
// reference's ids depend on Python set order); the PARTITION is the reference's.
constexpr int FC_THREADS = 256;

template <int ND>
__global__ void __launch_bounds__(FC_THREADS) find_clusters_kernel(const double* __restrict__ pos,
                                                                   const int32_t* __restrict__ frame_offset,
                                                                   double s0, double s1, double s2,
                                                                   double* __restrict__ spos, int32_t* label,
                                                                   int32_t* __restrict__ count,
                                                                   int32_t* __restrict__ size_out) {
  const int f = blockIdx.x;
  const int r0 = frame_offset[f], r1 = frame_offset[f + 1];
  const double sep[3] = {s0, s1, s2};
  for (int i = r0 + threadIdx.x; i < r1; i += FC_THREADS) {
#pragma unroll
    for (int a = 0; a < ND; ++a) spos[(size_t)i * ND + a] = pos[(size_t)i * ND + a] / sep[a];
    label[i] = i;
  }
  __syncthreads();
  for (int sweep = 0; sweep <= r1 - r0; ++sweep) {
    bool changed = false;
    for (int i = r0 + threadIdx.x; i < r1; i += FC_THREADS) {
      double p[ND];
#pragma unroll
      for (int a = 0; a < ND; ++a) p[a] = spos[(size_t)i * ND + a];
      const int mine = __hip_atomic_load(&label[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      int m = mine;
      for (int j = r0; j < r1; ++j) {
        double d2 = 0.;
#pragma unroll
        for (int a = 0; a < ND; ++a) {
          const double d = p[a] - spos[(size_t)j * ND + a];
          d2 += d * d;
        }
        if (d2 <= 1.) {
          const int lj = __hip_atomic_load(&label[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          m = lj < m ? lj : m;
        }
      }
      if (m < mine) {
        __hip_atomic_store(&label[i], m, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        changed = true;
      }
    }
    if (!__syncthreads_or(changed ? 1 : 0)) break;
  }
  // labels of this frame are final: root = smallest row index; count members, then sizes
  for (int i = r0 + threadIdx.x; i < r1; i += FC_THREADS) atomicAdd(&count[label[i]], 1);
  __syncthreads();
  for (int i = r0 + threadIdx.x; i < r1; i += FC_THREADS)
    size_out[i] = __hip_atomic_load(&count[label[i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- per-frame maximum (the norm of refine.py:354) --------------------------------
// Streams the frame block once: 16 B per lane per load, one ordered-u64 atomicMax
// per workgroup.  HBM-bound.

__device__ __forceinline__ unsigned long long enc_f64(double x) {
  unsigned long long b = (unsigned long long)__double_as_longlong(x);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ double dec_f64(unsigned long long e) {
  unsigned long long b = (e >> 63) ? (e & 0x7fffffffffffffffull) : ~e;
  return __longlong_as_double((long long)b);
}

template <typename T>
__device__ __forceinline__ double chunk_max(const T* p, size_t n, int tid, int nthreads) {
  constexpr int V = 16 / sizeof(T);
  double m = -INFINITY;
  const uintptr_t addr = (uintptr_t)p;
  size_t head = (16 - (addr & 15)) & 15;
  head /= sizeof(T);
  if (head > n) head = n;
  for (size_t i = tid; i < head; i += nthreads) {
    const double x = (double)p[i];
    m = (x > m || x != x) ? x : m;
  }
  const size_t nvec = (n - head) / V;
  const uint4* pv = (const uint4*)(p + head);
  for (size_t i = tid; i < nvec; i += nthreads) {
    uint4 raw = pv[i];
    T vals[V];
    __builtin_memcpy(vals, &raw, 16);
#pragma unroll
    for (int j = 0; j < V; ++j) {
      const double x = (double)vals[j];
      m = (x > m || x != x) ? x : m;
    }
  }
  for (size_t i = head + nvec * V + tid; i < n; i += nthreads) {
    const double x = (double)p[i];
    m = (x > m || x != x) ? x : m;
  }
  return m;
}

constexpr int FM_THREADS = 256;
constexpr size_t FM_CHUNK_BYTES = 64 * 1024;

__global__ void __launch_bounds__(FM_THREADS) frame_max_kernel(const void* frames, int dtype,
                                                               size_t frame_elems, int chunks_per_frame,
                                                               size_t chunk_elems,
                                                               unsigned long long* enc) {
  const int frame = blockIdx.x / chunks_per_frame, chunk = blockIdx.x % chunks_per_frame;
  const size_t begin = (size_t)chunk * chunk_elems;
  size_t n = frame_elems - begin;
  if (n > chunk_elems) n = chunk_elems;
  const size_t e0 = (size_t)frame * frame_elems + begin;
  double m;
  switch (dtype) {
    case CTR_DTYPE_U8: m = chunk_max((const uint8_t*)frames + e0, n, threadIdx.x, FM_THREADS); break;
    case CTR_DTYPE_U16: m = chunk_max((const uint16_t*)frames + e0, n, threadIdx.x, FM_THREADS); break;
    case CTR_DTYPE_I16: m = chunk_max((const int16_t*)frames + e0, n, threadIdx.x, FM_THREADS); break;
    case CTR_DTYPE_I32: m = chunk_max((const int32_t*)frames + e0, n, threadIdx.x, FM_THREADS); break;
    case CTR_DTYPE_F32: m = chunk_max((const float*)frames + e0, n, threadIdx.x, FM_THREADS); break;
    default: m = chunk_max((const double*)frames + e0, n, threadIdx.x, FM_THREADS); break;
  }
  unsigned long long e = enc_f64(m);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    unsigned long long other = __shfl_xor(e, o);
    e = other > e ? other : e;
  }
  __shared__ unsigned long long part[FM_THREADS / WAVE];
  if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = e;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int wv = 1; wv < FM_THREADS / WAVE; ++wv) e = part[wv] > e ? part[wv] : e;
    atomicMax(enc + frame, e);
  }
}

__global__ void frame_max_decode_kernel(const unsigned long long* enc, double* out, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = dec_f64(enc[i]);
}

// ---- host side ------------------------------------------------------------------------

typedef void (*kernel_fn)(const KArgs);
typedef void (*small_fn)(const KArgs, int*);

template <int NT> struct WavesFor { static constexpr int value = NT == 2 ? 8 : (NT <= 3 ? 4 : (NT <= 6 ? 2 : 1)); };

template <int ND, bool ISO, int NT>
void fill_one(kernel_fn* t, size_t* bytes, int* threads) {
  constexpr int W = WavesFor<NT>::value;
  static_assert(SmemB<NT, W>::bytes <= 160 * 1024, "LDS budget of one CU");
  t[NT - 1] = refine_block_kernel<ND, ISO, NT, W>;
  bytes[NT - 1] = SmemB<NT, W>::bytes;
  threads[NT - 1] = WAVE * W;
}

template <int ND, bool ISO>
void fill_table(kernel_fn* t, size_t* bytes, int* threads) {
  fill_one<ND, ISO, 1>(t, bytes, threads);
  fill_one<ND, ISO, 2>(t, bytes, threads);
  fill_one<ND, ISO, 3>(t, bytes, threads);
  fill_one<ND, ISO, 4>(t, bytes, threads);
  fill_one<ND, ISO, 5>(t, bytes, threads);
  fill_one<ND, ISO, 6>(t, bytes, threads);
  fill_one<ND, ISO, 7>(t, bytes, threads);
  fill_one<ND, ISO, 8>(t, bytes, threads);
}

std::string g_create_error;
std::mutex g_mutex;

}  // namespace

// bins: 0..MAXNT-1 generic kernel by NT-1; MAXNT = too large for the engine;
// MAXNT+1 / MAXNT+2 = singles / pairs with default modes (small kernel)
constexpr int BIN_TOO_LARGE = MAXNT, BIN_SMALL1 = MAXNT + 1, BIN_SMALL2 = MAXNT + 2, NBINS = MAXNT + 3;
constexpr int NSIDE = 3;  // side streams for concurrent bin launches

struct ctr_plan {
  ctr_problem prob;
  int64_t n_clusters = 0;
  int device = 0;
  int32_t* d_order = nullptr;  // all bins back to back
  int64_t bin_begin[NBINS + 1] = {0};
  int64_t bin_count[NBINS] = {0};
};

struct ctr_handle {
  int device = 0;
  hipStream_t stream = nullptr;
  hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
  bool ev_valid = false;
  std::string err;
  // grow-only device scratch
  void* d_buf = nullptr;
  size_t d_buf_bytes = 0;
  unsigned long long* d_enc = nullptr;
  double* d_fmax = nullptr;
  int64_t fmax_cap = 0;
  kernel_fn table[2][2][MAXNT];
  size_t smem_bytes[2][2][MAXNT];
  int block_threads[2][2][MAXNT];
  bool attr_set[2][2][MAXNT] = {};
  small_fn small_table[2][2][2];  // [ndim-2][iso][nf-1]; singles with 16 lanes per cluster
  small_fn small_wide1[2][2];     // singles with 64 lanes per cluster (large windows)
  hipStream_t side[NSIDE] = {nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[NSIDE] = {nullptr, nullptr, nullptr};
  int* d_counter = nullptr;       // work counters of the small-kernel launches
};

namespace {

int fail(ctr_handle* h, int code, const std::string& msg) {
  if (h) h->err = msg;
  else { std::lock_guard<std::mutex> g(g_mutex); g_create_error = msg; }
  return code;
}

#define HIP_TRY(h, call)                                                                   \
  do {                                                                                     \
    hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess)                                                                  \
      return fail(h, CTR_ERR_DEVICE, std::string(#call) + ": " + hipGetErrorString(e_));   \
  } while (0)

int validate(const ctr_problem* p, std::string& msg) {
  if (!p) { msg = "null problem"; return CTR_ERR_INVALID; }
  if (p->ndim != 2 && p->ndim != 3) { msg = "ndim must be 2 or 3"; return CTR_ERR_INVALID; }
  if (p->fit_function != CTR_FIT_GAUSS) {
    msg = "only the gauss fit function is implemented";
    return (p->fit_function >= 0 && p->fit_function <= CTR_FIT_INV_SERIES) ? CTR_ERR_UNSUPPORTED : CTR_ERR_INVALID;
  }
  const int np = 2 + p->ndim + (p->isotropic ? 1 : p->ndim);
  if (p->n_params != np) { msg = "n_params does not match ndim/isotropic"; return CTR_ERR_INVALID; }
  for (int k = 0; k < np; ++k) {
    const int m = p->modes[k];
    if (m == CTR_MODE_GLOBAL) { msg = "param mode 'global' couples all clusters and is not supported"; return CTR_ERR_UNSUPPORTED; }
    if (m != CTR_MODE_CONST && m != CTR_MODE_VAR && m != CTR_MODE_CLUSTER) { msg = "unknown param mode"; return CTR_ERR_INVALID; }
  }
  if (p->modes[0] == CTR_MODE_VAR) { msg = "background cannot vary per feature (fitfunc.py:389-392)"; return CTR_ERR_INVALID; }
  for (int a = 0; a < p->ndim; ++a)
    if (p->radius[a] < 1) { msg = "radius must be >= 1"; return CTR_ERR_INVALID; }
  if (p->max_iter < 1) { msg = "max_iter must be >= 1"; return CTR_ERR_INVALID; }
  if (p->constraint_kind < CTR_CONS_NONE || p->constraint_kind > CTR_CONS_TETRAMER) { msg = "unknown constraint kind"; return CTR_ERR_INVALID; }
  if (p->constraint_kind != CTR_CONS_NONE)
    for (int a = 0; a < p->ndim; ++a)
      if (!(p->constraint_dist[a] > 0.)) { msg = "constraint distance must be positive"; return CTR_ERR_INVALID; }
  if (!(p->residual_factor > 0.)) { msg = "residual_factor must be positive"; return CTR_ERR_INVALID; }
  return CTR_OK;
}

int n_vars(const ctr_problem* p, int n) {
  int nv = 0;
  for (int k = 0; k < p->n_params; ++k) {
    if (p->modes[k] == CTR_MODE_VAR) nv += n;
    else if (p->modes[k] != CTR_MODE_CONST) nv += 1;
  }
  return nv;
}

size_t host_dtype_size(int dtype) {
  switch (dtype) {
    case CTR_DTYPE_U8: return 1;
    case CTR_DTYPE_U16: case CTR_DTYPE_I16: return 2;
    case CTR_DTYPE_I32: case CTR_DTYPE_F32: return 4;
    case CTR_DTYPE_F64: return 8;
    default: return 0;
  }
}

int ensure_fmax(ctr_handle* h, int64_t n_frames) {
  if (n_frames <= h->fmax_cap) return CTR_OK;
  if (h->d_enc) { (void)hipFree(h->d_enc); h->d_enc = nullptr; }
  if (h->d_fmax) { (void)hipFree(h->d_fmax); h->d_fmax = nullptr; }
  h->fmax_cap = 0;
  HIP_TRY(h, hipMalloc((void**)&h->d_enc, sizeof(unsigned long long) * (size_t)n_frames));
  HIP_TRY(h, hipMalloc((void**)&h->d_fmax, sizeof(double) * (size_t)n_frames));
  h->fmax_cap = n_frames;
  return CTR_OK;
}

int launch_frame_max(ctr_handle* h, const void* frames, int dtype, int64_t n_frames,
                     int64_t frame_elems, double* out, hipStream_t s) {
  const size_t isz = host_dtype_size(dtype);
  if (!isz) return fail(h, CTR_ERR_INVALID, "unknown frame dtype");
  if (n_frames <= 0) return CTR_OK;
  int rc = ensure_fmax(h, n_frames);
  if (rc) return rc;
  const size_t chunk_elems = FM_CHUNK_BYTES / isz;
  const int chunks = (int)(((size_t)frame_elems + chunk_elems - 1) / chunk_elems);
  HIP_TRY(h, hipMemsetAsync(h->d_enc, 0, sizeof(unsigned long long) * (size_t)n_frames, s));
  const long long grid = (long long)n_frames * chunks;
  if (grid > 0x7fffffffLL) return fail(h, CTR_ERR_INVALID, "frame block too large for one launch");
  hipLaunchKernelGGL(frame_max_kernel, dim3((unsigned)grid), dim3(FM_THREADS), 0, s, frames, dtype,
                     (size_t)frame_elems, chunks, chunk_elems, h->d_enc);
  hipLaunchKernelGGL(frame_max_decode_kernel, dim3((unsigned)((n_frames + 255) / 256)), dim3(256), 0, s,
                     h->d_enc, out, n_frames);
  HIP_TRY(h, hipGetLastError());
  return CTR_OK;
}

}  // namespace

extern "C" {

int ctr_abi_version(void) { return CTR_ABI_VERSION; }

const char* ctr_last_error(const ctr_handle* h) {
  if (h) return h->err.c_str();
  return g_create_error.c_str();
}

int ctr_validate_problem(const ctr_problem* p, char* msg, int msg_len) {
  std::string m;
  const int rc = validate(p, m);
  if (msg && msg_len > 0) {
    std::snprintf(msg, (size_t)msg_len, "%s", m.c_str());
  }
  return rc;
}

int ctr_cluster_n_vars(const ctr_problem* p, int n_features) { return p ? n_vars(p, n_features) : -1; }

int ctr_create(ctr_handle** out, int device) {
  if (!out) return fail(nullptr, CTR_ERR_INVALID, "null out pointer");
  *out = nullptr;
  int count = 0;
  hipError_t e = hipGetDeviceCount(&count);
  if (e != hipSuccess || count <= 0)
    return fail(nullptr, CTR_ERR_DEVICE, std::string("no HIP device visible (") + hipGetErrorString(e) + "); there is no CPU fallback");
  if (device < 0 || device >= count) return fail(nullptr, CTR_ERR_INVALID, "device index out of range");
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(nullptr, CTR_ERR_DEVICE, "hipGetDeviceProperties failed");
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, CTR_ERR_DEVICE, std::string("device is ") + prop.gcnArchName + ", this engine is built for gfx950 (MI355X) only");
  ctr_handle* h = new ctr_handle();
  h->device = device;
  if (hipSetDevice(device) != hipSuccess || hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
    delete h;
    return fail(nullptr, CTR_ERR_DEVICE, "cannot create a stream on the device");
  }
  for (auto& ev : h->ev)
    if (hipEventCreate(&ev) != hipSuccess) { delete h; return fail(nullptr, CTR_ERR_DEVICE, "hipEventCreate failed"); }
  for (auto& st : h->side)
    if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) { delete h; return fail(nullptr, CTR_ERR_DEVICE, "cannot create side streams"); }
  bool evok = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess;
  for (auto& ev : h->ev_join) evok = evok && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
  if (!evok || hipMalloc((void**)&h->d_counter, sizeof(int) * 8) != hipSuccess) { delete h; return fail(nullptr, CTR_ERR_DEVICE, "cannot create events / counters"); }
  h->small_wide1[0][1] = refine_small_kernel<2, 1, true, 64>;
  h->small_wide1[0][0] = refine_small_kernel<2, 1, false, 64>;
  h->small_wide1[1][1] = refine_small_kernel<3, 1, true, 64>;
  h->small_wide1[1][0] = refine_small_kernel<3, 1, false, 64>;
  h->small_table[0][1][0] = refine_small_kernel<2, 1, true, 16>;
  h->small_table[0][1][1] = refine_small_kernel<2, 2, true, 64>;
  h->small_table[0][0][0] = refine_small_kernel<2, 1, false, 16>;
  h->small_table[0][0][1] = refine_small_kernel<2, 2, false, 64>;
  h->small_table[1][1][0] = refine_small_kernel<3, 1, true, 16>;
  h->small_table[1][1][1] = refine_small_kernel<3, 2, true, 64>;
  h->small_table[1][0][0] = refine_small_kernel<3, 1, false, 16>;
  h->small_table[1][0][1] = refine_small_kernel<3, 2, false, 64>;
  fill_table<2, true>(h->table[0][1], h->smem_bytes[0][1], h->block_threads[0][1]);
  fill_table<2, false>(h->table[0][0], h->smem_bytes[0][0], h->block_threads[0][0]);
  fill_table<3, true>(h->table[1][1], h->smem_bytes[1][1], h->block_threads[1][1]);
  fill_table<3, false>(h->table[1][0], h->smem_bytes[1][0], h->block_threads[1][0]);
  *out = h;
  return CTR_OK;
}

void ctr_destroy(ctr_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  if (h->stream) (void)hipStreamSynchronize(h->stream);
  if (h->d_buf) (void)hipFree(h->d_buf);
  if (h->d_enc) (void)hipFree(h->d_enc);
  if (h->d_fmax) (void)hipFree(h->d_fmax);
  for (auto& ev : h->ev) if (ev) (void)hipEventDestroy(ev);
  for (auto& st : h->side) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
  for (auto& ev : h->ev_join) if (ev) (void)hipEventDestroy(ev);
  if (h->d_counter) (void)hipFree(h->d_counter);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

int ctr_plan_create(ctr_handle* h, const ctr_problem* p, int64_t n_clusters,
                    const int32_t* feat_offset_host, ctr_plan** out) {
  if (!h || !out) return CTR_ERR_INVALID;
  *out = nullptr;
  std::string msg;
  int rc = validate(p, msg);
  if (rc) return fail(h, rc, msg);
  if (n_clusters < 0 || (n_clusters > 0 && !feat_offset_host)) return fail(h, CTR_ERR_INVALID, "bad cluster table");
  if (n_clusters > 0x7ffffff0LL) return fail(h, CTR_ERR_INVALID, "too many clusters for one batch");
  std::vector<int32_t> bin_of((size_t)n_clusters);
  ctr_plan* plan = new ctr_plan();
  plan->prob = *p;
  plan->n_clusters = n_clusters;
  plan->device = h->device;
  // singles / pairs with the default modes (fitfunc.py:356,379-387) go to the small kernel
  bool default_modes = p->modes[0] == CTR_MODE_CLUSTER && p->modes[1] == CTR_MODE_VAR;
  for (int a = 0; a < p->ndim; ++a) default_modes = default_modes && p->modes[2 + a] == CTR_MODE_VAR;
  for (int k2 = 2 + p->ndim; k2 < p->n_params; ++k2) default_modes = default_modes && p->modes[k2] == CTR_MODE_CONST;
  for (int64_t c = 0; c < n_clusters; ++c) {
    const int64_t n = (int64_t)feat_offset_host[c + 1] - feat_offset_host[c];
    if (n < 0) { delete plan; return fail(h, CTR_ERR_INVALID, "feat_offset must be non-decreasing"); }
    int bin;
    const bool constrained = (p->constraint_kind == CTR_CONS_DIMER && n == 2);
    if (n > MAXF) bin = BIN_TOO_LARGE;
    else if (default_modes && n == 1) bin = BIN_SMALL1;
    else if (default_modes && n == 2 && !constrained) bin = BIN_SMALL2;
    else {
      const int nv = n_vars(p, (int)n);
      const int nt = (nv + 1 + 15) / 16;
      bin = nt > MAXNT ? BIN_TOO_LARGE : (nt < 1 ? 0 : nt - 1);
      if (bin < MAXNT && n > (16 * (bin + 1) < MAXF ? 16 * (bin + 1) : MAXF)) bin = BIN_TOO_LARGE;
    }
    bin_of[(size_t)c] = bin;
    plan->bin_count[bin]++;
  }
  plan->bin_begin[0] = 0;
  for (int b = 0; b < NBINS; ++b) plan->bin_begin[b + 1] = plan->bin_begin[b] + plan->bin_count[b];
  std::vector<int32_t> order((size_t)n_clusters);
  {
    int64_t cursor[NBINS];
    for (int b = 0; b < NBINS; ++b) cursor[b] = plan->bin_begin[b];
    for (int64_t c = 0; c < n_clusters; ++c) order[(size_t)cursor[bin_of[(size_t)c]]++] = (int32_t)c;
    // inside a generic bin the clusters with the most features start first
    for (int b = 0; b < MAXNT; ++b)
      std::stable_sort(order.begin() + plan->bin_begin[b], order.begin() + plan->bin_begin[b + 1],
                       [&](int32_t x, int32_t y) {
                         return feat_offset_host[x + 1] - feat_offset_host[x] > feat_offset_host[y + 1] - feat_offset_host[y];
                       });
  }
  if (n_clusters > 0) {
    if (hipSetDevice(h->device) != hipSuccess ||
        hipMalloc((void**)&plan->d_order, sizeof(int32_t) * (size_t)n_clusters) != hipSuccess) {
      delete plan;
      return fail(h, CTR_ERR_NOMEM, "cannot allocate the plan on the device");
    }
    if (hipMemcpy(plan->d_order, order.data(), sizeof(int32_t) * (size_t)n_clusters, hipMemcpyHostToDevice) != hipSuccess) {
      (void)hipFree(plan->d_order);
      delete plan;
      return fail(h, CTR_ERR_DEVICE, "cannot upload the plan");
    }
  }
  *out = plan;
  return CTR_OK;
}

void ctr_plan_destroy(ctr_plan* plan) {
  if (!plan) return;
  if (plan->d_order) { (void)hipSetDevice(plan->device); (void)hipFree(plan->d_order); }
  delete plan;
}

int ctr_frame_max_device(ctr_handle* h, const void* frames, int32_t frame_dtype, int64_t n_frames,
                         int64_t frame_elems, double* out_max, void* hip_stream) {
  if (!h) return CTR_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
  return launch_frame_max(h, frames, frame_dtype, n_frames, frame_elems, out_max, s);
}

int ctr_refine_batch_device(ctr_handle* h, const ctr_plan* plan, const ctr_batch* b, void* hip_stream) {
  if (!h || !plan || !b) return CTR_ERR_INVALID;
  if (b->n_clusters != plan->n_clusters) return fail(h, CTR_ERR_INVALID, "batch does not match the plan");
  const ctr_problem& p = plan->prob;
  if (!host_dtype_size(b->frame_dtype)) return fail(h, CTR_ERR_INVALID, "unknown frame dtype");
  for (int a = 0; a < p.ndim; ++a)
    if (b->shape[a] < 1) return fail(h, CTR_ERR_INVALID, "frame shape must be positive");
  HIP_TRY(h, hipSetDevice(h->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
  int64_t frame_elems = 1;
  for (int a = 0; a < p.ndim; ++a) frame_elems *= b->shape[a];
  h->ev_valid = false;
  HIP_TRY(h, hipEventRecord(h->ev[0], s));
  int rc = ensure_fmax(h, b->n_frames > 0 ? b->n_frames : 1);
  if (rc) return rc;
  rc = launch_frame_max(h, b->frames, b->frame_dtype, b->n_frames, frame_elems, h->d_fmax, s);
  if (rc) return rc;
  HIP_TRY(h, hipEventRecord(h->ev[1], s));

  KArgs k;
  std::memset(&k, 0, sizeof k);
  k.prob = p;
  k.frames = b->frames;
  k.frame_dtype = b->frame_dtype;
  for (int a = 0; a < 3; ++a) k.shape[a] = a < p.ndim ? b->shape[a] : 1;
  k.frame_elems = frame_elems;
  k.frame_index = b->frame_index;
  k.feat_offset = b->feat_offset;
  k.params = b->params;
  k.low = b->low;
  k.high = b->high;
  k.params_out = b->params_out;
  k.cost = b->cost;
  k.status = b->status;
  k.n_rounds = b->n_rounds;
  k.n_iter = b->n_iter;
  k.fmax = h->d_fmax;
  const int di = p.ndim == 3 ? 1 : 0, ii = p.isotropic ? 1 : 0;
  // The bins are independent: the big bin of singles runs on the caller's
  // stream, the others on side streams forked from / joined to it by events, so
  // that a few slow many-feature clusters overlap with the bulk.
  HIP_TRY(h, hipEventRecord(h->ev_fork, s));
  bool used[NSIDE] = {false, false, false};
  int next_side = 0;
  auto pick_stream = [&](bool main_stream) -> hipStream_t {
    if (main_stream) return s;
    const int j = next_side++ % NSIDE;
    if (!used[j]) { used[j] = true; (void)hipStreamWaitEvent(h->side[j], h->ev_fork, 0); }
    return h->side[j];
  };
  // generic bins first (largest problems first) so that their tails start early
  for (int bin = MAXNT - 1; bin >= 0; --bin) {
    const int64_t cnt = plan->bin_count[bin];
    if (cnt == 0) continue;
    kernel_fn fn = h->table[di][ii][bin];
    const size_t bytes = h->smem_bytes[di][ii][bin];
    if (!h->attr_set[di][ii][bin]) {
      HIP_TRY(h, hipFuncSetAttribute((const void*)fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
      h->attr_set[di][ii][bin] = true;
    }
    k.order = plan->d_order + plan->bin_begin[bin];
    k.n_bin = (int32_t)cnt;
    hipLaunchKernelGGL(fn, dim3((unsigned)cnt), dim3((unsigned)h->block_threads[di][ii][bin]), bytes, pick_stream(false), k);
  }
  if (plan->bin_count[BIN_TOO_LARGE] > 0) {
    const int64_t cnt = plan->bin_count[BIN_TOO_LARGE];
    k.order = plan->d_order + plan->bin_begin[BIN_TOO_LARGE];
    k.n_bin = (int32_t)cnt;
    hipLaunchKernelGGL(mark_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, pick_stream(false), k,
                       (int)CTR_STATUS_TOO_LARGE);
  }
  for (int nf = 2; nf >= 1; --nf) {
    const int bin = nf == 1 ? BIN_SMALL1 : BIN_SMALL2;
    const int64_t cnt = plan->bin_count[bin];
    if (cnt == 0) continue;
    hipStream_t st = pick_stream(nf == 1);
    int* counter = h->d_counter + nf;
    HIP_TRY(h, hipMemsetAsync(counter, 0, sizeof(int), st));
    k.order = plan->d_order + plan->bin_begin[bin];
    k.n_bin = (int32_t)cnt;
    // lanes per cluster by the size of a single-feature window: 16 (four clusters per
    // wave) while a window is a few passes, 64 once it is thousands of pixels (3D)
    int64_t vol = 1;
    for (int a = 0; a < p.ndim; ++a) vol *= 2 * (int64_t)p.radius[a] + 1;
    const bool wide = nf == 2 || vol > 600;
    small_fn fn = nf == 2 ? h->small_table[di][ii][1] : (wide ? h->small_wide1[di][ii] : h->small_table[di][ii][0]);
    int64_t waves = wide ? cnt : (cnt + 3) / 4;
    if (waves > 8192) waves = 8192;
    hipLaunchKernelGGL(fn, dim3((unsigned)waves), dim3(WAVE), 0, st, k, counter);
  }
  for (int j = 0; j < NSIDE; ++j)
    if (used[j]) {
      HIP_TRY(h, hipEventRecord(h->ev_join[j], h->side[j]));
      HIP_TRY(h, hipStreamWaitEvent(s, h->ev_join[j], 0));
    }
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipEventRecord(h->ev[2], s));
  h->ev_valid = true;
  return CTR_OK;
}

#ifdef CTR_STAMPS
int ctr_debug_stamps(unsigned long long* out16, int reset) {
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z) != hipSuccess) return 1;
  }
  return 0;
}
#endif


int ctr_find_clusters(ctr_handle* h, int32_t ndim, const double* pos, const int32_t* frame_offset,
                      int64_t n_frames, const double* separation, int32_t* label_out, int32_t* size_out) {
  if (!h || !pos || !frame_offset || !separation || !label_out || !size_out) return CTR_ERR_INVALID;
  if (ndim != 2 && ndim != 3) return fail(h, CTR_ERR_INVALID, "ndim must be 2 or 3");
  if (n_frames < 0 || n_frames > 0x7fffffffLL) return fail(h, CTR_ERR_INVALID, "bad frame count");
  if (n_frames == 0) return CTR_OK;
  const int64_t N = frame_offset[n_frames];
  if (frame_offset[0] != 0 || N < 0) return fail(h, CTR_ERR_INVALID, "frame_offset must run from 0 to the row count");
  for (int64_t f = 0; f < n_frames; ++f)
    if (frame_offset[f + 1] < frame_offset[f]) return fail(h, CTR_ERR_INVALID, "frame_offset must be non-decreasing");
  for (int a = 0; a < ndim; ++a)
    if (!(separation[a] > 0.)) return fail(h, CTR_ERR_INVALID, "separation must be positive");
  if (N == 0) return CTR_OK;
  HIP_TRY(h, hipSetDevice(h->device));
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t sz_pos = al(sizeof(double) * (size_t)N * ndim), sz_fo = al(sizeof(int32_t) * (size_t)(n_frames + 1));
  const size_t sz_i = al(sizeof(int32_t) * (size_t)N);
  const size_t total = 2 * sz_pos + sz_fo + 3 * sz_i;
  if (total > h->d_buf_bytes) {
    if (h->d_buf) { (void)hipFree(h->d_buf); h->d_buf = nullptr; h->d_buf_bytes = 0; }
    if (hipMalloc(&h->d_buf, total) != hipSuccess) return fail(h, CTR_ERR_NOMEM, "cannot allocate device memory");
    h->d_buf_bytes = total;
  }
  char* q = (char*)h->d_buf;
  double* d_pos = (double*)q; q += sz_pos;
  double* d_spos = (double*)q; q += sz_pos;
  int32_t* d_fo = (int32_t*)q; q += sz_fo;
  int32_t* d_label = (int32_t*)q; q += sz_i;
  int32_t* d_count = (int32_t*)q; q += sz_i;
  int32_t* d_size = (int32_t*)q;
  hipStream_t s = h->stream;
  HIP_TRY(h, hipMemcpyAsync(d_pos, pos, sizeof(double) * (size_t)N * ndim, hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemcpyAsync(d_fo, frame_offset, sizeof(int32_t) * (size_t)(n_frames + 1), hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemsetAsync(d_count, 0, sizeof(int32_t) * (size_t)N, s));
  const double s2 = ndim == 3 ? separation[2] : 1.;
  if (ndim == 2)
    hipLaunchKernelGGL(find_clusters_kernel<2>, dim3((unsigned)n_frames), dim3(FC_THREADS), 0, s, d_pos, d_fo,
                       separation[0], separation[1], s2, d_spos, d_label, d_count, d_size);
  else
    hipLaunchKernelGGL(find_clusters_kernel<3>, dim3((unsigned)n_frames), dim3(FC_THREADS), 0, s, d_pos, d_fo,
                       separation[0], separation[1], s2, d_spos, d_label, d_count, d_size);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipMemcpyAsync(label_out, d_label, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipMemcpyAsync(size_out, d_size, sizeof(int32_t) * (size_t)N, hipMemcpyDeviceToHost, s));
  HIP_TRY(h, hipStreamSynchronize(s));
  return CTR_OK;
}

int ctr_synchronize(ctr_handle* h, void* hip_stream) {
  if (!h) return CTR_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(hip_stream ? (hipStream_t)hip_stream : h->stream));
  return CTR_OK;
}

int ctr_last_kernel_ms(ctr_handle* h, double* frame_max_ms, double* refine_ms) {
  if (!h) return CTR_ERR_INVALID;
  if (!h->ev_valid) return fail(h, CTR_ERR_INVALID, "no ctr_refine_batch_device call to time");
  HIP_TRY(h, hipEventSynchronize(h->ev[2]));
  float a = 0.f, c = 0.f;
  HIP_TRY(h, hipEventElapsedTime(&a, h->ev[0], h->ev[1]));
  HIP_TRY(h, hipEventElapsedTime(&c, h->ev[1], h->ev[2]));
  if (frame_max_ms) *frame_max_ms = a;
  if (refine_ms) *refine_ms = c;
  return CTR_OK;
}

int ctr_refine_batch(ctr_handle* h, const ctr_problem* p, const ctr_batch* b) {
  if (!h || !p || !b) return CTR_ERR_INVALID;
  std::string msg;
  int rc = validate(p, msg);
  if (rc) return fail(h, rc, msg);
  const size_t isz = host_dtype_size(b->frame_dtype);
  if (!isz) return fail(h, CTR_ERR_INVALID, "unknown frame dtype");
  if (b->n_clusters < 0 || b->n_features < 0 || b->n_frames < 0) return fail(h, CTR_ERR_INVALID, "negative counts");
  if (b->n_clusters == 0) return CTR_OK;
  if (!b->frames || !b->frame_index || !b->feat_offset || !b->params || !b->low || !b->high ||
      !b->params_out || !b->cost || !b->status || !b->n_rounds || !b->n_iter)
    return fail(h, CTR_ERR_INVALID, "null buffer in batch");
  int64_t frame_elems = 1;
  for (int a = 0; a < p->ndim; ++a) {
    if (b->shape[a] < 1) return fail(h, CTR_ERR_INVALID, "frame shape must be positive");
    frame_elems *= b->shape[a];
  }
  const int64_t C = b->n_clusters, N = b->n_features;
  if (b->feat_offset[0] != 0 || b->feat_offset[C] != N) return fail(h, CTR_ERR_INVALID, "feat_offset must run from 0 to n_features");
  for (int64_t c = 0; c < C; ++c) {
    if (b->feat_offset[c + 1] < b->feat_offset[c]) return fail(h, CTR_ERR_INVALID, "feat_offset must be non-decreasing");
    if (b->frame_index[c] < 0 || b->frame_index[c] >= b->n_frames) return fail(h, CTR_ERR_INVALID, "frame_index out of range");
  }
  HIP_TRY(h, hipSetDevice(h->device));

  const size_t np = (size_t)p->n_params;
  auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
  const size_t sz_frames = al((size_t)b->n_frames * (size_t)frame_elems * isz);
  const size_t sz_fi = al(sizeof(int32_t) * (size_t)C), sz_fo = al(sizeof(int32_t) * (size_t)(C + 1));
  const size_t sz_par = al(sizeof(double) * (size_t)N * np);
  const size_t sz_c = al(sizeof(double) * (size_t)C), sz_ci = al(sizeof(int32_t) * (size_t)C);
  const size_t total = sz_frames + sz_fi + sz_fo + 4 * sz_par + sz_c + 3 * sz_ci;
  if (total > h->d_buf_bytes) {
    if (h->d_buf) { (void)hipFree(h->d_buf); h->d_buf = nullptr; h->d_buf_bytes = 0; }
    if (hipMalloc(&h->d_buf, total) != hipSuccess) return fail(h, CTR_ERR_NOMEM, "cannot allocate device memory for the batch");
    h->d_buf_bytes = total;
  }
  char* q = (char*)h->d_buf;
  ctr_batch d = *b;
  auto take = [&](size_t n) { char* r = q; q += n; return r; };
  char* d_frames = take(sz_frames);
  int32_t* d_fi = (int32_t*)take(sz_fi);
  int32_t* d_fo = (int32_t*)take(sz_fo);
  double* d_par = (double*)take(sz_par);
  double* d_low = (double*)take(sz_par);
  double* d_high = (double*)take(sz_par);
  double* d_out = (double*)take(sz_par);
  double* d_cost = (double*)take(sz_c);
  int32_t* d_status = (int32_t*)take(sz_ci);
  int32_t* d_rounds = (int32_t*)take(sz_ci);
  int32_t* d_iter = (int32_t*)take(sz_ci);
  hipStream_t s = h->stream;
  HIP_TRY(h, hipMemcpyAsync(d_frames, b->frames, (size_t)b->n_frames * (size_t)frame_elems * isz, hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemcpyAsync(d_fi, b->frame_index, sizeof(int32_t) * (size_t)C, hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemcpyAsync(d_fo, b->feat_offset, sizeof(int32_t) * (size_t)(C + 1), hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemcpyAsync(d_par, b->params, sizeof(double) * (size_t)N * np, hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemcpyAsync(d_low, b->low, sizeof(double) * (size_t)N * np, hipMemcpyHostToDevice, s));
  HIP_TRY(h, hipMemcpyAsync(d_high, b->high, sizeof(double) * (size_t)N * np, hipMemcpyHostToDevice, s));
  d.frames = d_frames; d.frame_index = d_fi; d.feat_offset = d_fo;
  d.params = d_par; d.low = d_low; d.high = d_high; d.params_out = d_out;
  d.cost = d_cost; d.status = d_status; d.n_rounds = d_rounds; d.n_iter = d_iter;
  ctr_plan* plan = nullptr;
  rc = ctr_plan_create(h, p, C, b->feat_offset, &plan);
  if (rc) return rc;
  rc = ctr_refine_batch_device(h, plan, &d, s);
  if (rc == CTR_OK) {
    hipError_t e = hipMemcpyAsync(b->params_out, d_out, sizeof(double) * (size_t)N * np, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(b->cost, d_cost, sizeof(double) * (size_t)C, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(b->status, d_status, sizeof(int32_t) * (size_t)C, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(b->n_rounds, d_rounds, sizeof(int32_t) * (size_t)C, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipMemcpyAsync(b->n_iter, d_iter, sizeof(int32_t) * (size_t)C, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) rc = fail(h, CTR_ERR_DEVICE, std::string("copy back / kernel execution: ") + hipGetErrorString(e));
  } else {
    (void)hipStreamSynchronize(s);
  }
  ctr_plan_destroy(plan);
  return rc;
}

}  // extern "C"
