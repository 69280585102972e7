// block_kernel.h -- refine_block_kernel: W wavefronts per cluster, LDS row tiles + f64 MFMA
// Part of the MI355X cluster-refinement engine; included by ctrefine.hip inside its
// anonymous namespace (device code only, gfx950).
#ifndef CTREFINE_BLOCK_KERNEL_H
#define CTREFINE_BLOCK_KERNEL_H

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

// nwt: add the exact second-order part.  qk[kr] is the entry between this lane's variable
// and the variable of kind kr (0 signal, 1 + a position) of the same feature; ic = vinfo[c].
template <int NR, int NK>
__device__ __forceinline__ bool column_solve(const double* Mp, int nv, double mu, bool is_free,
                                             int lane, double& x_own, bool nwt, const int* vinfo,
                                             int ic, const double (&qk)[NK]) {
  const int c = lane;
  const bool colv = c < nv && is_free;
  const unsigned long long fmask = __ballot(colv);
  double col[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const bool rowv = ((fmask >> i) & 1ull) != 0ull;
    double x = 0., xq = 0.;
    if (rowv && colv) x = Msym(Mp, i + 1, c + 1);
    if (nwt) {
      const int ir = vinfo[i];   // same address in every lane
      const int kr = (ir & 7) - 1;
      double q = qk[0];
#pragma unroll
      for (int t = 1; t < NK; ++t) q = kr == t ? qk[t] : q;
      xq = (rowv && colv && ic >= 0 && ((ir ^ ic) >> 3) == 0 && kr >= 0) ? q : 0.;
    }
    // Marquardt scaling by the Gauss-Newton diagonal in both models
    if (i == c) x = colv ? x + xq + mu * (x > 1e-300 ? x : 1.) : 1.;
    else x += xq;
    col[i] = x;
  }
  double y = colv ? Mp[tri(c + 1)] : 0.;
  double mydinv = 1., yown = 0.;
  bool ok = true;
#pragma unroll
  for (int j = 0; j < NR; ++j) {
    if (j < nv && ok) {  // rows beyond the variables are identity: nothing to eliminate; a failed
                         // pivot ends the factorisation (both conditions are wave-uniform)
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
    if (j < nv && ok) {
      const double xj = readlane_f64((yown - s) * mydinv, j);
      if (c == j) x_own = xj;
      s += (c < j ? col[j] * mydinv : 0.) * xj;
    }
  }
  return ok;
}

// Internal column order of the block kernel's [r J] matrix: column 0 is the
// residual, then the shared variables (background, any 'cluster'-mode column), then
// the per-feature variables FEATURE-MAJOR, so that a feature occupies one (at most
// two) 16-column MFMA blocks and a pixel tile only pays for the blocks of its
// candidate features.  Variable c lives in column c + 1.  (The reference's
// parameter-major order, fitfunc.py:207-263, is an external convention only: the
// engine's inputs and outputs are parameter tables.)
struct LayoutB {
  int n, nv, nshared, npf;
  int slot[CTR_MAX_PARAMS];     // rank among the shared / per-feature variables, -1 if constant
  int per_feat[CTR_MAX_PARAMS];
  __device__ __forceinline__ int vidx(int kk, int i) const {
    return slot[kk] < 0 ? -1 : (per_feat[kk] ? nshared + i * npf + slot[kk] : slot[kk]);
  }
  // bit mask of the 16-column blocks that hold the columns of feature i
  __device__ __forceinline__ unsigned blocks(int i) const {
    if (npf == 0) return 0u;
    const int c0 = 1 + nshared + i * npf, c1 = c0 + npf - 1;
    return (1u << (c0 >> 4)) | (1u << (c1 >> 4));
  }
};

__device__ __forceinline__ void make_layout_b(const ctr_problem& p, int n, LayoutB& L) {
  int ns = 0, np = 0;
  L.n = n;
#pragma unroll
  for (int k = 0; k < CTR_MAX_PARAMS; ++k) {
    const int m = k < p.n_params ? p.modes[k] : CTR_MODE_CONST;
    if (m == CTR_MODE_CONST) { L.slot[k] = -1; L.per_feat[k] = 0; }
    else if (m == CTR_MODE_VAR) { L.slot[k] = np++; L.per_feat[k] = 1; }
    else { L.slot[k] = ns++; L.per_feat[k] = 0; }
  }
  L.nshared = ns;
  L.npf = np;
  L.nv = ns + n * np;
}

constexpr int QT = 9;  // second-order entries kept per feature: ND (signal, pos_a) + ND(ND+1)/2 (pos_a, pos_b)

template <int NT, int W, bool CONS = false>
struct SmemB {
  static constexpr int NVP = 16 * NT;
  static constexpr int RS = NVP + 1;
  static constexpr int NTILE = NT * (NT + 1) / 2;
  static constexpr int NF = NVP < MAXF ? NVP : MAXF;
  // constraint Jacobians: constrained clusters have 2..4 features = at most 29 variables (NT <= 2)
  static constexpr int NVC = NT <= 2 ? NVP : 1;
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
  // second-order entries of the current point, QT per feature (features with >= 3 variables each)
  static constexpr int NFQ = (NVP + 2) / 3 < MAXF ? (NVP + 2) / 3 : MAXF;
  static constexpr int o_uc = o_ctl + 8;
  static constexpr int o_vinfo = o_uc + NFQ * QT;       // ints: 8 * feature + kind + 1 of every variable
  static constexpr int o_cpair = o_vinfo + NVP / 2;     // ints: pair behind constraint r (current, trial)
  // constrained fits (cons_qp): 1/diag of the factor, gradient of the model, and the
  // second-order part in all variables as a packed triangle (non-default modes)
  static constexpr int o_qd = o_cpair + MAXC;
  static constexpr int o_gq = o_qd + (CONS ? NVP : 0);
  static constexpr int o_Qp = o_gq + (CONS ? NVP : 0);
  static constexpr int total = o_Qp + (CONS ? NVP * (NVP + 1) / 2 : 0);
  static constexpr size_t bytes = (size_t)total * sizeof(double);
  static_assert(W == 1 || NTILE * 256 + 6 * WAVE <= ROWS, "partial accumulators must fit a row tile");
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

// CONS: the instantiation that takes the clusters with equality constraints (dimers, trimers,
// tetramers: at most 29 variables, NT <= 2); in the others the constrained code is compiled out
// (it costs the unconstrained fits registers otherwise: 700 B of scratch per lane, measured).
// LP: the instantiation for problems with a lowpass of the window (ctr_problem.noise_size): every
// pixel value is the filtered one, computed from the raw frame where it is needed.
// FIT: the radial profile (CTR_FIT_GAUSS / RING / DISC / INV_SERIES; device_common.h:profile_dev,
// profile_inv_dev); ring and disc carry one more parameter column (thickness / disc_size),
// inv_series_<N> N + 1 (signal_mult, param_a, ...); all three iterate with the Gauss-Newton model.
// Minimum wavefronts per SIMD asked of the compiler (the register budget) for the instantiations
// of the throughput scheduling (2D, fewest wavefronts per cluster: W = 2 for NT <= 2, 1 above).
// NT = 1 (3-4 features): 3 -- 168 VGPRs + 272 B of scratch instead of 231 + 100 at 2 per SIMD:
// +6 % on cfg 2 with ten batches in flight (interleaved A/B, tools/ab_bench.sh: 48.4 -> 51.4 M fits/s;
// 4 per SIMD: 128 VGPRs + 484 B, -8 %).  The kernel waits on dependent FP64 chains; a third
// wavefront per SIMD hides more of them than the spills cost.
#ifndef CTR_OCC_NT1
#define CTR_OCC_NT1 3
#endif
#ifndef CTR_OCC_NT2
#define CTR_OCC_NT2 1
#endif
#ifndef CTR_OCC_NT34
#define CTR_OCC_NT34 1
#endif
constexpr int block_occ(int nt, int w, bool cons, bool lp, int fit) {
  return (cons || lp || fit != 0) ? 1 : (nt == 1 && w == 2) ? CTR_OCC_NT1 : (nt == 2 && w == 2) ? CTR_OCC_NT2
       : ((nt == 3 || nt == 4) && w == 1) ? CTR_OCC_NT34 : 1;
}
template <int ND, bool ISO, int NT, int W, bool CONS, bool LP = false, int FIT = 0>
__global__ void __launch_bounds__(WAVE * W, (block_occ(NT, W, CONS, LP, FIT)))
refine_block_kernel(const KArgs k) {
#ifdef CTR_STAMPS
  unsigned long long t_prev_ = __builtin_amdgcn_s_memtime();
#endif
  using SM = SmemB<NT, W, CONS>;
  // profile parameters after the sizes: one for ring / disc; inv_series_<N> has N + 1 of them, known
  // at run time: NP is then the problem's n_params, NPC (the bound of the unrolled loops over the
  // parameter columns) its maximum; columns beyond n_params are constants to the layout
  constexpr int NX = FIT == CTR_FIT_INV_SERIES ? INV_NX : (FIT != CTR_FIT_GAUSS ? 1 : 0);
  constexpr int NSZ = ISO ? 1 : ND;
  constexpr int NPC_ = 2 + ND + NSZ + NX;
  constexpr int NPC = NPC_ < CTR_MAX_PARAMS ? NPC_ : CTR_MAX_PARAMS;
  const int NP = FIT == CTR_FIT_INV_SERIES ? k.prob.n_params : NPC;
  const int nx_inv = NP - (2 + ND + NSZ);
  constexpr bool SAFE_R2 = FIT == CTR_FIT_RING || FIT == CTR_FIT_DISC;   // r2_*_safe (fitfunc.py:396-411)
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
  double* Uc = smem + SM::o_uc;
  // 8 * feature + kind + 1 of variable c (kind: 0 signal, 1 + a position axis a, -1 anything
  // else); negative for a shared variable
  int* vinfo = (int*)(smem + SM::o_vinfo);
  int* cpair = (int*)(smem + SM::o_cpair);   // [0..5] current point, [6..11] trial
  double* myrows = smem + SM::o_rows + wave * SM::ROWS;

  LayoutB L;
  make_layout_b(k.prob, n, L);
  const int nv = L.nv;
  const int m = CONS ? n_constraints(k.prob, n) : 0;
  // the exact second-order terms need signal and positions as per-feature variables (the
  // default modes); otherwise the model Hessian is J^T J throughout (same rule: oracle solve())
  // (those terms are the gaussian's: the other profiles iterate with J^T J)
  bool newton_on = FIT == CTR_FIT_GAUSS && L.slot[1] >= 0 && L.per_feat[1];
#pragma unroll
  for (int a = 0; a < ND; ++a) newton_on = newton_on && L.slot[2 + a] >= 0 && L.per_feat[2 + a];
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
    const int b = L.vidx(kk, i);
    if (b < 0) return cur[i * CTR_MAX_PARAMS + kk];
    return vv[b];
  };
  // derived constants of every feature at vv: [0] signal [1..3] centre
  // [4..6] 1/size^2 [7..9] 2/size^2 [10..12] -2/size^3   (wave 0)
  bool size_is_var = false;
#pragma unroll
  for (int kk = 2 + ND; kk < NPC; ++kk) size_is_var = size_is_var || L.slot[kk] >= 0;
  // constrained fits with other than the default modes: the second-order part in ALL variables
  // (second_order_pass after every accepted step) instead of the (signal, position) part that
  // the pixel pass sums (same rule: oracle solve(), fullq)
  const bool fullq = CONS && FIT == CTR_FIT_GAUSS && m != 0 && (!newton_on || size_is_var);
  const bool cheapq = newton_on && !fullq;
  newton_on = newton_on || fullq;
  double* Qp = smem + SM::o_Qp;
  auto fill_fpar = [&](const double* vv, bool sizes) {
    bool bad_size = false;
    for (int i = lane; i < n; i += WAVE) {
      double* f = fpar + i * FP;
      f[0] = par(vv, i, 1);
      if (SAFE_R2) f[13] = par(vv, i, NPC - 1);   // the profile parameter (ring, disc)
#pragma unroll
      for (int a = 0; a < ND; ++a) {
        f[1 + a] = par(vv, i, 2 + a);
        if (sizes) {  // three f64 divisions per axis: only when a size actually changed
          const double sz = par(vv, i, ISO ? 2 + ND : 2 + ND + a);
          const double s2 = sz * sz;
          bad_size = bad_size || !(sz > 0.);
          f[4 + a] = 1. / s2;
          f[7 + a] = 2. / s2;
          f[10 + a] = -2. / (s2 * sz);
        }
      }
    }
    if (sizes) {
      const bool anyb = __ballot(bad_size) != 0ull;
      if (lane == 0) ctl[7] = anyb ? 1 : 0;
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
  // cv[m], Cj[m][LDC] at vv (constraints.py:59-137); wave 0.  2D tetramer: the constrained
  // pairs are the 4 smallest of the 6 pair distances; with use_fixed the pairs fixedp[0..m) are
  // used as they are (a smooth branch of that function) instead of being ranked again.
  auto eval_constraints = [&](const double* vv, double* cvo, double* Cjo, int* cpo, bool use_fixed,
                              const int (&fixedp)[MAXC]) {
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
      if (use_fixed) {
        rank = MAXC;
#pragma unroll
        for (int r = 0; r < MAXC; ++r) rank = (r < m && fixedp[r] == q) ? r : rank;
      } else if (k.prob.constraint_kind == CTR_CONS_TETRAMER && ND == 2) {
        rank = 0;  // stable rank among the 6 squared distances (constraints.py:112)
#pragma unroll
        for (int p = 0; p < 6; ++p) rank += (d2[p] < mine || (d2[p] == mine && p < q)) ? 1 : 0;
      }
      if (rank < m) {
        const int i0 = (q == 0 || q == 2 || q == 4) ? 0 : (q == 5 ? 2 : 1);
        const int i1 = q == 0 ? 1 : (q <= 2 ? 2 : 3);
        cvo[rank] = 1. - mine;
        cpo[rank] = q;
#pragma unroll
        for (int a = 0; a < ND; ++a) {
          const int kk = 2 + a;
          if (L.slot[kk] < 0) continue;
          const double da = k.prob.constraint_dist[a];
          const double t = -2. * (par(vv, i0, kk) - par(vv, i1, kk)) / (da * da);
          Cjo[rank * LDC + L.vidx(kk, i0)] += t;
          Cjo[rank * LDC + L.vidx(kk, i1)] -= t;
        }
      }
    }
    wsync();
  };

  // m x m SPD solve in LDS by lane 0 (m <= 6): Gm[MAXC*MAXC] <- Cholesky factor, rhs solved in
  // place; flag[0] = 1 on success.  Ends with a wsync.
  auto small_spd_solve = [&](double* Gm, double* rhs) {
    if (lane == 0) {
      double tr = 0.;
      for (int r = 0; r < m; ++r) tr += Gm[r * MAXC + r];
      bool okc = tr > 0.;
      for (int r = 0; r < m; ++r) Gm[r * MAXC + r] += 1e-14 * tr + 1e-300;
      for (int j = 0; j < m && okc; ++j) {
        double d = Gm[j * MAXC + j];
        for (int q = 0; q < j; ++q) d -= Gm[j * MAXC + q] * Gm[j * MAXC + q];
        if (!(d > 0.) || !isfinite(d)) { okc = false; break; }
        d = sqrt(d);
        Gm[j * MAXC + j] = d;
        for (int i = j + 1; i < m; ++i) {
          double s2 = Gm[i * MAXC + j];
          for (int q = 0; q < j; ++q) s2 -= Gm[i * MAXC + q] * Gm[j * MAXC + q];
          Gm[i * MAXC + j] = s2 / d;
        }
      }
      if (okc) {
        for (int i = 0; i < m; ++i) {
          double s2 = rhs[i];
          for (int q = 0; q < i; ++q) s2 -= Gm[i * MAXC + q] * rhs[q];
          rhs[i] = s2 / Gm[i * MAXC + i];
        }
        for (int i = m - 1; i >= 0; --i) {
          double s2 = rhs[i];
          for (int q = i + 1; q < m; ++q) s2 -= Gm[q * MAXC + i] * rhs[q];
          rhs[i] = s2 / Gm[i * MAXC + i];
        }
      }
      flag[0] = okc ? 1. : 0.;
    }
    wsync();
  };
  // Retraction onto the constraint manifold c(x) = 0 (oracle retract()): minimum-norm Newton
  // steps x <- x - C^T (C C^T)^-1 c over the variables the box leaves free, clipped to the
  // box; a variable on a bound that the correction would push outward is left out.  Wave 0,
  // one variable per lane (constrained clusters have at most 29 variables).  On success cvo /
  // Cjo / cpo describe the constraints at the returned point.
  auto retract = [&](double* x, double* cvo, double* Cjo, int* cpo, unsigned long long heldmask) -> bool {
    double* ry = cv + 18;   // (6 free doubles of the small block)
    const bool ranked = k.prob.constraint_kind == CTR_CONS_TETRAMER && ND == 2;
    int pairs[MAXC];
#pragma unroll
    for (int r = 0; r < MAXC; ++r) pairs[r] = 0;
    eval_constraints(x, cvo, Cjo, cpo, false, pairs);
#pragma unroll
    for (int r = 0; r < MAXC; ++r) pairs[r] = r < m ? cpo[r] : 0;
    wsync();
    // 2D tetramer: Newton on the branch of the current pair set; if the ranking at the point
    // reached names another set, once more on that one
    for (int branch = 0; branch < 4; ++branch) {
      bool conv = false;
      for (int iter = 0; iter <= 30; ++iter) {
        eval_constraints(x, cvo, Cjo, cpo, true, pairs);
        double cmax = 0.;
        bool nan = false;
        for (int r = 0; r < m; ++r) { const double a = fabs(cvo[r]); nan = nan || !(a == a); cmax = fmax(cmax, a); }
        if (nan) return false;
        if (cmax <= 1e-13) { conv = true; break; }
        if (iter == 30) break;
        const bool isvar = lane < nv;
        const double xi = isvar ? x[lane] : 0., li = isvar ? lo[lane] : 0., ui = isvar ? hi[lane] : 0.;
        bool pinned = !isvar || !(li < ui) || ((heldmask >> lane) & 1ull) != 0ull;
        double ti = 0.;
        for (int pass = 0; pass < 2; ++pass) {
          const unsigned long long fm = __ballot(!pinned);
          if (lane < m * m) {
            const int r = lane / (m > 0 ? m : 1), s2 = lane % (m > 0 ? m : 1);
            double t = 0.;
            for (unsigned long long q = fm; q != 0ull; q &= q - 1ull) {
              const int i = __builtin_ctzll(q);
              t += Cjo[r * LDC + i] * Cjo[s2 * LDC + i];
            }
            Sc[r * MAXC + s2] = t;
          }
          if (lane < m) ry[lane] = cvo[lane];
          wsync();
          small_spd_solve(Sc, ry);
          if (flag[0] == 0.) return false;
          ti = 0.;
          if (!pinned)
            for (int r = 0; r < m; ++r) ti += Cjo[r * LDC + lane] * ry[r];
          if (pass == 1) break;
          const bool newpin = !pinned && ((xi <= li && ti > 0.) || (xi >= ui && ti < 0.));
          if (__ballot(newpin) == 0ull) break;
          pinned = pinned || newpin;
          wsync();
        }
        if (!pinned && ti != 0.) {
          const double t = xi - ti;
          x[lane] = t < li ? li : (t > ui ? ui : t);
        }
        wsync();
      }
      if (!conv) return false;
      if (!ranked) return true;
      eval_constraints(x, cvo, Cjo, cpo, false, pairs);
      unsigned sa = 0u, sb = 0u;
#pragma unroll
      for (int r = 0; r < MAXC; ++r)
        if (r < m) { sa |= 1u << cpo[r]; sb |= 1u << pairs[r]; }
      if (sa == sb) return true;
#pragma unroll
      for (int r = 0; r < MAXC; ++r) pairs[r] = r < m ? cpo[r] : 0;
      wsync();
    }
    return false;
  };

  // Sum_p res_p d2res_p/dv dv in ALL variables at v (oracle full_second_order), added to the
  // packed lower triangle dest[tri(nv)]; wave 0, fpar filled for v.  wsum: 28 doubles of scratch.
  // Used by compute_error (refine.py:400-406) and, for constrained fits with other than the
  // default modes, as the second-order part of the solver's model.
  auto second_order_pass = [&](double* dest, double* wsum) {
        // per feature the second derivatives of m = s g, g = exp(E), in parameter space
        // (signal, centres, sizes): m_s,t = g E_t, m_t,u = s g (E_t E_u + E_tu); entry (t <= u) of
        // the PW x PW block at PWI(t, u); lane e = PW t + u adds it to the packed H at the
        // variables the two parameters map to
        constexpr int PW = 1 + ND + NSZ, NW = PW * (PW + 1) / 2, NG = (NW + 3) / 4;
        int origin[ND], wshape[ND];
#pragma unroll
        for (int a = 0; a < ND; ++a) { origin[a] = ctl[1 + a]; wshape[a] = ctl[4 + a]; }
        const int w1 = wshape[ND - 2], w2 = wshape[ND - 1];
        const int npix = (ND == 3 ? wshape[0] : 1) * w1 * w2;
        const double bg = par(v, 0, 0);
        const int et = lane / PW, eu = lane - et * PW;
        const int widx = et <= eu ? et * PW - (et * (et - 1)) / 2 + (eu - et)
                                  : eu * PW - (eu * (eu - 1)) / 2 + (et - eu);
        for (int base = 0; base < npix; base += WAVE) {
          const int q = base + lane;
          const bool valid = q < npix;
          int idx[ND];
          size_t off;
          {
            const int t = q / w2, x = q - t * w2;
            if (ND == 3) {
              const int z = t / w1, y = t - z * w1;
              idx[0] = z; idx[1] = y; idx[ND - 1] = x;
              off = ((size_t)(z + origin[0]) * fshape[1] + (y + origin[1])) * fshape[ND - 1] + (x + origin[ND - 1]);
            } else {
              idx[0] = t; idx[ND - 1] = x;
              off = (size_t)(t + origin[0]) * fshape[ND - 1] + (x + origin[ND - 1]);
            }
          }
          unsigned long long mine = 0ull;
          double model = 0.;
          for (int i = 0; i < n; ++i) {
            double rel[ND];
#pragma unroll
            for (int a = 0; a < ND; ++a) rel[a] = mco[i * 3 + a] - (double)origin[a];
            if (valid && in_mask<ND>(idx, rel, inv_r2, radius)) {
              const double* f = fpar + i * FP;
              double r2 = 0.;
#pragma unroll
              for (int a = 0; a < ND; ++a) {
                const double d = (double)(idx[a] + origin[a]) - f[1 + a];
                r2 += d * d * f[4 + a];
              }
              model += f[0] * exp(-0.5 * ND * r2);
              mine |= 1ull << i;
            }
          }
          double resg = 0.;
          if (mine != 0ull) {
            double pixv;
            if constexpr (LP) pixv = lowpass_pixel<ND>(frame, k.frame_dtype, fshape, origin, wshape, idx, k.lp_w, k.lp_half, k.prob.threshold);
            else pixv = load_pixel(frame, k.frame_dtype, off);
            const double res = pixv - bg - model;
            resg = res == res ? res : 0.;  // nansum
          }
          for (int i = 0; i < n; ++i) {
            const bool in = ((mine >> i) & 1ull) != 0ull;
            if (__ballot(in) == 0ull) continue;
            const double* f = fpar + i * FP;
            double E1[PW], dd[ND], i2z[ND], q2 = 0., r2 = 0.;
#pragma unroll
            for (int a = 0; a < ND; ++a) {
              dd[a] = (double)(idx[a] + origin[a]) - f[1 + a];
              i2z[a] = -0.5 * f[10 + a];   // 1 / size^3
              q2 += dd[a] * dd[a];
              r2 += dd[a] * dd[a] * f[4 + a];
            }
            const double G = in ? exp(-0.5 * ND * r2) : 0.;
            const double sg = f[0] * G, mr = -resg;   // d2res = -d2m
            E1[0] = 0.;
#pragma unroll
            for (int a = 0; a < ND; ++a) {
              E1[1 + a] = (double)ND * dd[a] * f[4 + a];
              if (!ISO) E1[(1 + ND + a) % PW] = (double)ND * dd[a] * dd[a] * i2z[a];
            }
            if (ISO) E1[1 + ND] = (double)ND * q2 * i2z[0];
            auto hval = [&](int t, int u) -> double {   // t <= u, compile-time after unrolling
              if (u == 0) return 0.;
              if (t == 0) return G * E1[u];
              double h = sg * (E1[t] * E1[u]), e2 = 0.;
              const int a = t - 1;
              if (u <= ND) {
                if (t == u) e2 = -(double)ND * f[4 + a];
              } else if (t <= ND) {
                if (ISO) e2 = -2. * ND * dd[a] * i2z[0];
                else if (u - 1 - ND == a) e2 = -2. * ND * dd[a] * i2z[a];
              } else {
                if (ISO) e2 = -3. * ND * q2 * f[4] * f[4];
                else if (t == u) {
                  const int b = t - 1 - ND;
                  e2 = -3. * ND * dd[b] * dd[b] * f[4 + b] * f[4 + b];
                }
              }
              return h + sg * e2;
            };
            double hv[4 * NG];
            {
              int e = 0;
#pragma unroll
              for (int t = 0; t < PW; ++t)
#pragma unroll
                for (int u = t; u < PW; ++u) hv[e++] = mr * hval(t, u);
#pragma unroll
              for (; e < 4 * NG; ++e) hv[e] = 0.;
            }
#pragma unroll
            for (int gq = 0; gq < NG; ++gq) {
              const double t4 = wave_sum4(hv[4 * gq], hv[4 * gq + 1], hv[4 * gq + 2], hv[4 * gq + 3], lane);
              if ((lane & 15) == 0) wsum[4 * gq + (lane >> 4)] = t4;
            }
            wsync();
            if (lane < PW * PW) {
              const int ct = L.vidx(1 + et, i), cu = L.vidx(1 + eu, i);
              if (ct >= 0 && cu >= 0 && ct >= cu) dest[tri(ct) + cu] += wsum[widx];
            }
            wsync();
          }
        }
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
  for (int c = tid; c < SM::NVP; c += WAVE * W) {
    int fi = -1, kind = -1;
    if (c >= L.nshared && c < nv) {
      fi = (c - L.nshared) / L.npf;
      const int sl = c - L.nshared - fi * L.npf;
      if (L.per_feat[1] && sl == L.slot[1]) kind = 0;
#pragma unroll
      for (int a = 0; a < ND; ++a)
        if (L.per_feat[2 + a] && sl == L.slot[2 + a]) kind = 1 + a;
    }
    vinfo[c] = fi < 0 ? -1 : 8 * fi + kind + 1;
  }
  for (int e = tid; e < n * 3; e += WAVE * W) {
    const int i = e / 3, a = e % 3;
    mco[e] = a < ND ? params[i * NP + 2 + a] : 0.;
  }
  {
    const double* low = k.low + (size_t)f0 * NP;
    const double* high = k.high + (size_t)f0 * NP;
#pragma unroll
    for (int kk = 0; kk < NPC; ++kk) {
      if (L.slot[kk] < 0) continue;
      if (L.per_feat[kk]) {
        for (int i = tid; i < n; i += WAVE * W) {
          const int b = L.vidx(kk, i);
          v0[b] = params[i * NP + kk];
          lo[b] = low[i * NP + kk];
          hi[b] = high[i * NP + kk];
        }
      } else if (tid == 0) {
        const int b = L.vidx(kk, 0);
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
  double mu = 1e-3, nu = 2., S = 0., pred = 0., rms = NAN;
  bool retr_fail = false;   // the start vector could not be brought onto the constraint manifold
  double prev_step = INFINITY, trial_step = 0.;   // relative size of the last accepted / of the pending step
  bool last_acc = true;
  double gain = INFINITY;  // relative merit decrease of the last accepted step
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
    // the start vector need not satisfy the constraints: restore feasibility first
    retr_fail = m > 0 && !retract(vt, cvt, Cjt, cpair + MAXC, 0ull);
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
  // second-order sums U[a<=b] = sum_p res J_pos_a dE/dpos_b of feature `lane` (this wave's share)
  constexpr int NUF = ND * (ND + 1) / 2;
  double uacc[NUF];
#pragma unroll
  for (int t = 0; t < NUF; ++t) uacc[t] = 0.;
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
      const double bg = par(vt, 0, 0);
      const float inv_w2 = 1.f / (float)w2, inv_w1 = 1.f / (float)w1;
      const bool big_window = npix >= (1 << 21);
#pragma unroll
      for (int t = 0; t < SM::NTILE; ++t) acc[t] = v4d{0., 0., 0., 0.};
#pragma unroll
      for (int t = 0; t < NUF; ++t) uacc[t] = 0.;
      double* row = myrows + lane * SM::RS;
      int pf_base[CTR_MAX_PARAMS], pf_step[CTR_MAX_PARAMS];
#pragma unroll
      for (int kk = 0; kk < CTR_MAX_PARAMS; ++kk) {
        const bool pf = L.slot[kk] >= 0 && L.per_feat[kk];
        pf_base[kk] = pf ? 1 + L.nshared + L.slot[kk] : SM::NVP;
        pf_step[kk] = pf ? L.npf : 0;
      }
      // first tile of an evaluation: write every feature's columns (the row tile was used as
      // scratch by the solve / the parked accumulators in between)
      const unsigned long long all_feat = n >= 64 ? ~0ull : ((1ull << n) - 1ull);
      unsigned long long prev_cand = all_feat;
      for (int base = wave * WAVE; base < npix; base += WAVE * W) {
        // Which features can touch this tile at all?  Lane l tests the box of feature l
        // (mask centre +- radius, a superset of its mask) against the extent of the 64
        // consecutive pixels: one ballot gives the candidates.  Features written by the
        // previous tile of this wave but not candidates now still get their zeros.
        unsigned long long cand;
        {
          const int q0 = base, q1 = (base + WAVE - 1 < npix ? base + WAVE - 1 : npix - 1);
          int lo_i[ND], hi_i[ND];
          const int t0 = q0 / w2, t1 = q1 / w2;
          if (ND == 3) {
            const int z0 = t0 / w1, z1 = t1 / w1;
            lo_i[0] = z0; hi_i[0] = z1;
            const bool same_z = z0 == z1;
            lo_i[1] = same_z ? t0 - z0 * w1 : 0;
            hi_i[1] = same_z ? t1 - z1 * w1 : w1 - 1;
            const bool same_row = t0 == t1;
            lo_i[ND - 1] = same_row ? q0 - t0 * w2 : 0;
            hi_i[ND - 1] = same_row ? q1 - t1 * w2 : w2 - 1;
          } else {
            lo_i[0] = t0; hi_i[0] = t1;
            const bool same_row = t0 == t1;
            lo_i[ND - 1] = same_row ? q0 - t0 * w2 : 0;
            hi_i[ND - 1] = same_row ? q1 - t1 * w2 : w2 - 1;
          }
          bool hit = lane < n;
          if (hit) {
#pragma unroll
            for (int a = 0; a < ND; ++a) {
              const double ca = mco[lane * 3 + a] - (double)origin[a];
              hit = hit && ((double)hi_i[a] >= ca - (double)radius[a]) && ((double)lo_i[a] <= ca + (double)radius[a]);
            }
          }
          cand = __ballot(hit);
        }
        if ((cand | prev_cand) == 0ull) continue;  // nothing to compute, nothing to clear
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
        // the pixel is fetched up front (its latency hides behind the mask tests)
        double pix = 0.;
        if (valid && cand != 0ull) {
          if constexpr (LP) pix = lowpass_pixel<ND>(frame, k.frame_dtype, fshape, origin, wshape, idx, k.lp_w, k.lp_half, k.prob.threshold);
          else pix = load_pixel(frame, k.frame_dtype, off);
        }
        unsigned long long todo = cand | prev_cand;
        prev_cand = cand;
        double res = 0.;
        double shared[CTR_MAX_PARAMS];
#pragma unroll
        for (int kk = 0; kk < CTR_MAX_PARAMS; ++kk) shared[kk] = 0.;
        while (todo != 0ull) {
          const int i = __builtin_ctzll(todo);
          todo &= todo - 1ull;
          double d[NPC - 1];
#pragma unroll
          for (int t = 0; t < NPC - 1; ++t) d[t] = 0.;
          bool in = false;
          if (valid && ((cand >> i) & 1ull)) {
            double rel[ND];
#pragma unroll
            for (int a = 0; a < ND; ++a) rel[a] = mco[i * 3 + a] - (double)origin[a];
            in = in_mask<ND>(idx, rel, inv_r2, radius);
          }
          if (in) {
            const double* f = fpar + i * FP;
            if (!any) {
              any = true;
              res = pix - bg;
            }
            double r2 = 0., dd[ND];
#pragma unroll
            for (int a = 0; a < ND; ++a) {
              dd[a] = (double)(idx[a] + origin[a]) - f[1 + a];
              r2 += dd[a] * dd[a] * f[4 + a];
            }
            const double sig = f[0];
            double gv, sdg;
            if constexpr (FIT == CTR_FIT_GAUSS) {
              gv = exp(-0.5 * ND * r2);  // fitfunc.py:112-118
              sdg = sig * (0.5 * ND) * gv;  // -signal * dg/dr2
            } else if constexpr (FIT == CTR_FIT_INV_SERIES) {
              // the profile parameters of the trial vector (constants or variables: par())
              double ex[INV_NX], dge[INV_NX], dg;
#pragma unroll
              for (int t = 0; t < INV_NX; ++t) ex[t] = t < nx_inv ? par(vt, i, 2 + ND + NSZ + t) : 0.;
              profile_inv_dev(nx_inv, r2, ex, gv, dg, dge);
              sdg = -sig * dg;
#pragma unroll
              for (int t = 0; t < INV_NX; ++t)
                if (1 + ND + NSZ + t < NPC - 1) d[1 + ND + NSZ + t] = -sig * dge[t];   // fitfunc.py:480-481
            } else {
              double qraw = 0., dg, dge;
#pragma unroll
              for (int a = ND - 1; a >= 0; --a) qraw += dd[a] * dd[a];
              profile_dev<FIT, ND>(r2, f[13], gv, dg, dge);
              // r2_*_safe (fitfunc.py:20-26,...): no value within one pixel of the centre; the
              // pixel is skipped like a NaN pixel of the image (it still counts in P)
              // (the disc has the value 1 there: its function only overwrites where r2 > disc_size^2)
              if (qraw < 1.) { gv = (FIT == CTR_FIT_DISC && f[13] > 0.) ? 1. : NAN; dg = 0.; dge = 0.; }
              sdg = -sig * dg;
              d[1 + ND + NSZ] = -sig * dge;   // fitfunc.py:480-481
            }
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
          // branch-free scatter: columns of constant / shared parameters go to the pad
          // column of the row (index 16*NT, never read by the MFMA)
#pragma unroll
          for (int kk = 1; kk < NPC; ++kk) {
            row[pf_base[kk] + pf_step[kk] * i] = d[kk - 1];
            shared[kk] += d[kk - 1];
          }
        }
        const bool good = any && (res == res);  // nansum (fitfunc.py:449,483)
#pragma unroll
        for (int kk = 1; kk < NPC; ++kk)
          if (L.slot[kk] >= 0 && !L.per_feat[kk]) row[1 + L.slot[kk]] = good ? shared[kk] : 0.;
        if (L.slot[0] >= 0) row[1 + L.slot[0]] = good ? -1. : 0.;
        row[0] = good ? res : 0.;
        if (any && !good) {
          for (int j = 1 + L.nshared; j <= nv; ++j) row[j] = 0.;
        }
        if (cheapq && cand != 0ull) {
          // second pass over the candidates, now that the residual is complete: the lanes'
          // res * J_pos_a * dE/dpos_b are summed over the tile (four sums per exchange
          // sequence) and land in the accumulators of lane i = the feature
          const double resg = good ? res : 0.;
          for (unsigned long long cm = cand; cm != 0ull; cm &= cm - 1ull) {
            const int i = __builtin_ctzll(cm);
            const double* f = fpar + i * FP;
            double rj[ND], E[ND];
#pragma unroll
            for (int a = 0; a < ND; ++a) {
              rj[a] = resg * row[pf_base[2 + a] + pf_step[2 + a] * i];
              E[a] = (double)ND * (((double)(idx[a] + origin[a]) - f[1 + a]) * f[4 + a]);
            }
            if (ND == 2) {
              const double t = wave_sum4(rj[0] * E[0], rj[0] * E[ND - 1], rj[ND - 1] * E[ND - 1], 0., lane);
              const double s0 = readlane_f64(t, 0), s1 = readlane_f64(t, 16), s2 = readlane_f64(t, 32);
              if (lane == i) { uacc[0] += s0; uacc[1] += s1; uacc[NUF - 1] += s2; }
            } else {
              const double t = wave_sum4(rj[0] * E[0], rj[0] * E[1], rj[0] * E[ND - 1], rj[1] * E[1], lane);
              const double t2 = wave_sum4(rj[1] * E[ND - 1], rj[ND - 1] * E[ND - 1], 0., 0., lane);
              const double s0 = readlane_f64(t, 0), s1 = readlane_f64(t, 16), s2 = readlane_f64(t, 32),
                           s3 = readlane_f64(t, 48), s4 = readlane_f64(t2, 0), s5 = readlane_f64(t2, 16);
              if (lane == i) {
                uacc[0] += s0; uacc[1] += s1; uacc[2 % NUF] += s2;
                uacc[3 % NUF] += s3; uacc[4 % NUF] += s4; uacc[5 % NUF] += s5;
              }
            }
          }
        }
        const unsigned long long bal = __ballot(any);
        P += __popcll(bal);
        if (good) Sloc += res * res;
        wsync();
        if (bal != 0ull) {
          // column blocks worth multiplying: block 0 (residual + shared) and the blocks of
          // the candidate features; everything else in this tile is zero
          unsigned tilemask = 1u;
          for (unsigned long long cm = cand; cm != 0ull; cm &= cm - 1ull) tilemask |= L.blocks(__builtin_ctzll(cm));
          const int kr = lane >> 4, cc = lane & 15;
          const double* rbase = myrows + kr * SM::RS + cc;
          if (NT == 1) {
#pragma unroll 4
            for (int s = 0; s < 16; ++s) {
              const double x = rbase[4 * s * SM::RS];
              acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, acc[0], 0, 0, 0);
            }
          } else if (tilemask == (1u << NT) - 1u) {
            // every block is live: all pairs per k-step, one LDS read per block
#pragma unroll 4
            for (int s = 0; s < 16; ++s) {
              double val[NT];
#pragma unroll
              for (int t = 0; t < NT; ++t) val[t] = rbase[4 * s * SM::RS + 16 * t];
              int tt = 0;
#pragma unroll
              for (int ti = 0; ti < NT; ++ti)
#pragma unroll
                for (int tj = 0; tj <= ti; ++tj) {
                  acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(val[ti], val[tj], acc[tt], 0, 0, 0);
                  ++tt;
                }
            }
          } else {
            // one (block, block) pair at a time, so that the skip is one uniform branch per pair
            int tt = 0;
#pragma unroll
            for (int ti = 0; ti < NT; ++ti)
#pragma unroll
              for (int tj = 0; tj <= ti; ++tj) {
                if ((((tilemask >> ti) & (tilemask >> tj)) & 1u) != 0u) {
#pragma unroll 4
                  for (int s = 0; s < 16; ++s) {
                    const double xa = rbase[4 * s * SM::RS + 16 * ti];
                    const double xb = rbase[4 * s * SM::RS + 16 * tj];
                    acc[tt] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, xb, acc[tt], 0, 0, 0);
                  }
                }
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
#pragma unroll
          for (int t = 0; t < NUF; ++t) myrows[(SM::NTILE * 4 + t) * WAVE + lane] = uacc[t];
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
#pragma unroll
            for (int t = 0; t < NUF; ++t) uacc[t] += pr[(SM::NTILE * 4 + t) * WAVE + lane];
            St += part[2 * ww];
            P += (int)part[2 * ww + 1];
          }
          wsync();
          // the parked tiles overwrote columns of the row tiles: clear what rows never rewrite
          // (columns > nv are read by the MFMA but only feed entries nobody looks at)
        }
        if (ctl[7] != 0) St = NAN;   // a size of 0 (its lower bound): no valid model (oracle: model_nan)
      }
      bool accept = false;
      if (phase == BP_EVAL_INIT) {
        if (P == 0) { status = CTR_STATUS_OUT_OF_BOUNDS; failed = true; }
        else if (!isfinite(St) || retr_fail) { status = CTR_STATUS_NO_CONVERGENCE; failed = true; }
        // (a size among the variables: more damping at the start, see oracle solve())
        mu = size_is_var ? 1. : 1e-3; nu = 2.; last_acc = true; gain = INFINITY; prev_step = INFINITY;
        Pround = P;
        accept = !failed;
      } else if (phase == BP_EVAL_TRIAL) {
        // every iterate is feasible (retraction): a step is judged by the objective alone
        double act = 0.5 * (S - St);
        act = bcast0(act);
        if (isfinite(St) && pred > 0. && act > 0.) {
          const double rho = act / pred, t = 2. * rho - 1.;
          const double f = 1. - t * t * t;
          mu *= f > 1. / 3. ? f : 1. / 3.;
          nu = 2.;
          gain = act / (0.5 * S + 1e-300);
          accept = true;
          last_acc = true;
          prev_step = trial_step;
        } else {
          mu *= nu; nu *= 2.; last_acc = false;
          if (mu > 1e30) { status = CTR_STATUS_NO_CONVERGENCE; failed = true; }
        }
      }
      if (accept) {
        for (int i = lane; i < nv; i += WAVE) v[i] = vt[i];
        for (int e = lane; e < m * LDC; e += WAVE) Cj[e] = Cjt[e];
        if (lane < m) { cv[lane] = cvt[lane]; cpair[lane] = cpair[MAXC + lane]; }

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
        if constexpr (CONS) {
          if (fullq) {
            for (int e = lane; e < tri(nv); e += WAVE) Qp[e] = 0.;
            fill_fpar(v, true);
            wsync();
            second_order_pass(Qp, Y);
            wsync();
          }
        }
        if (cheapq && lane < n) {
          // second-order entries of feature `lane` at the new point: U from the pixel pass,
          // the rest from the gradient (oracle eval_cluster): d2res/ds dpos_a = g_pos_a / s,
          // d2res/dpos_a dpos_b = U_ab + delta_ab (-ND/size_a^2) s g_s
          const int cs = L.vidx(1, lane);
          const double sig = v[cs], gs = Mp[tri(cs + 1)];
          int e = 0;
#pragma unroll
          for (int a = 0; a < ND; ++a) {
            const double ga = Mp[tri(L.vidx(2 + a, lane) + 1)];
            Uc[lane * QT + a] = sig != 0. ? ga / sig : 0.;
#pragma unroll
            for (int b2 = a; b2 < ND; ++b2) {
              double u = uacc[e];
              if (b2 == a) {
                const double sz = par(v, lane, ISO ? 2 + ND : 2 + ND + a);
                u += -(double)ND / (sz * sz) * sig * gs;
              }
              Uc[lane * QT + ND + e] = u;
              ++e;
            }
          }
        }
        wsync();
      }
      STAMP(2);
      bool converged = false;
      if (!failed && it >= maxiter) {
        // iteration limit: a stationary point still counts as converged (see solve() of the oracle)
        if (gain <= CTR_STALL_TOL) converged = true;
        else { status = CTR_STATUS_NO_CONVERGENCE; failed = true; }
      }
      if (!failed && !converged) {
        ++it;
        ++iters;
        // least-squares multipliers lam = -(C C^T)^-1 C g over the variables the box does not
        // pin (oracle solve(): the ones that go with the minimum-norm retraction)
        double mult0[MAXC];
#pragma unroll
        for (int r = 0; r < MAXC; ++r) mult0[r] = 0.;
        unsigned long long freemask = 0ull;   // constrained fits: the free variables
        if (m) {
          // two passes (oracle solve()): the multipliers over all variables the box does not fix
          // give a first active set; the multipliers over the variables THAT leaves free give
          // the active set used
          double* ry = cv + 18;
          for (int pass = 0; pass < 2; ++pass) {
            const unsigned long long inset = pass == 0
                ? __ballot(lane < nv && lo[lane < nv ? lane : 0] < hi[lane < nv ? lane : 0]) : freemask;
            if (lane < m * m) {
              const int r = lane / (m > 0 ? m : 1), s2 = lane % (m > 0 ? m : 1);
              double t = 0.;
              for (int i = 0; i < nv; ++i)
                if ((inset >> i) & 1ull) t += Cj[r * LDC + i] * Cj[s2 * LDC + i];
              Sc[r * MAXC + s2] = t;
            }
            if (lane < m) {
              double t = 0.;
              for (int i = 0; i < nv; ++i)
                if ((inset >> i) & 1ull) t -= Cj[lane * LDC + i] * Mp[tri(i + 1)];
              ry[lane] = t;
            }
            wsync();
            small_spd_solve(Sc, ry);
            const bool okl = flag[0] != 0.;
#pragma unroll
            for (int r = 0; r < MAXC; ++r)
              if (pass == 0 || okl) mult0[r] = (okl && r < m) ? ry[r] : 0.;
            wsync();
            bool fre = false;
            if (lane < nv) {
              double gl = Mp[tri(lane + 1)];
#pragma unroll
              for (int r = 0; r < MAXC; ++r)
                if (r < m) gl += Cj[r * LDC + lane] * mult0[r];
              fre = !((lo[lane] == hi[lane]) || (v[lane] <= lo[lane] && gl > 0.) || (v[lane] >= hi[lane] && gl < 0.));
            }
            freemask = __ballot(fre);
          }
        }
        // active set: fixed if at a bound and the Lagrangian gradient pushes outward
        int nf = 0;
        bool is_free = false;   // of variable `lane` (all that the register solve needs)
        const bool reg_solve = NT <= 2 && m == 0;
        if (reg_solve) {
          if (lane < nv) {
            const double gl = Mp[tri(lane + 1)];
            is_free = !((lo[lane] == hi[lane]) || (v[lane] <= lo[lane] && gl > 0.) || (v[lane] >= hi[lane] && gl < 0.));
          }
          nf = __popcll(__ballot(is_free));
        } else if (m) {
          nf = __popcll(freemask);
          if ((freemask >> lane) & 1ull) fr[__popcll(freemask & ((1ull << lane) - 1ull))] = lane;
          wsync();
        } else {
          for (int b0 = 0; b0 < nv; b0 += WAVE) {
            const int i = b0 + lane;
            bool fre = false;
            if (i < nv) {
              const double gl = Mp[tri(i + 1)];
              const bool fixed = (lo[i] == hi[i]) || (v[i] <= lo[i] && gl > 0.) || (v[i] >= hi[i] && gl < 0.);
              fre = !fixed;
            }
            const unsigned long long bal = __ballot(fre);
            if (fre) fr[nf + __popcll(bal & ((1ull << lane) - 1ull))] = i;
            nf += __popcll(bal);
          }
          wsync();
        }
        STAMP(3);
        bool ok_step = false;
        const double tiny = ftol * (0.5 * S) + 1e-300;
        if (nf == 0) {
          converged = true;
        } else {
          // Model Hessian.  First choice: the exact one, J^T J + sum_p r_p d2r_p (signal and
          // positions of every feature: qvar) + sum_r lam_r d2c_r (ccurv).  Where that is not
          // positive definite on the free variables, or its projected step is not a descent
          // step of the model, the second-order part of the residuals is dropped, then (for
          // constrained fits) the curvature of the constraints as well: J^T J.  Same sequence
          // as solve() of oracle/ctr_oracle.c.
          double stepmax = 0.;
          // second-order part between two variables (0 unless both belong to one feature)
          auto qvar = [&](int gi, int gj) -> double {
            if constexpr (CONS) {
              if (fullq) return gi >= gj ? Qp[tri(gi) + gj] : Qp[tri(gj) + gi];
            }
            const int ia = vinfo[gi], ib = vinfo[gj];
            const int ka = (ia & 7) - 1, kb = (ib & 7) - 1;
            const bool hit = ia >= 0 && ((ia ^ ib) >> 3) == 0 && ka >= 0 && kb >= 0 && ka + kb > 0;
            const int k0 = ka < kb ? ka : kb, k1 = ka < kb ? kb : ka;   // 0 signal, 1 + a position
            int idx = k0 == 0 ? k1 - 1 : ND + ((k0 - 1) * ND - ((k0 - 1) * (k0 - 2)) / 2 + (k1 - k0));
            idx = hit ? (ia >> 3) * QT + idx : 0;
            const double q = Uc[idx];
            return hit ? q : 0.;
          };
          // curvature of the constraints between two position variables of the same axis
          auto ccurv = [&](int gi, int gj) -> double {
            const int ia = vinfo[gi], ib = vinfo[gj];
            const int ka = (ia & 7) - 1;
            if (ia < 0 || ib < 0 || ka < 1 || ka != (ib & 7) - 1) return 0.;
            const int fi = ia >> 3, fj = ib >> 3;
            const double da = k.prob.constraint_dist[ka - 1];
            double t = 0.;
#pragma unroll
            for (int r = 0; r < MAXC; ++r) {
              if (r < m) {
                const int q = cpair[r];
                const int i0 = (q == 0 || q == 2 || q == 4) ? 0 : (q == 5 ? 2 : 1);
                const int i1 = q == 0 ? 1 : (q <= 2 ? 2 : 3);
                const double tr = -2. * mult0[r] / (da * da);
                if (fi == fj) { if (fi == i0 || fi == i1) t += tr; }
                else if ((fi == i0 && fj == i1) || (fi == i1 && fj == i0)) t -= tr;
              }
            }
            return t;
          };
          // register solve: the second-order entries of this lane's variable, by kind of the
          // other variable of its feature
          const int ic = (reg_solve && newton_on && lane < nv) ? vinfo[lane] : -1;
          double qk[ND + 1];
          {
            const int kc = (ic & 7) - 1;
#pragma unroll
            for (int kr = 0; kr <= ND; ++kr) {
              const int k0 = kr < kc ? kr : kc, k1 = kr < kc ? kc : kr;
              const bool valid = ic >= 0 && kc >= 0 && kr + kc > 0;
              const int idx = k0 == 0 ? k1 - 1 : ND + ((k0 - 1) * ND - ((k0 - 1) * (k0 - 2)) / 2 + (k1 - k0));
              const double q = Uc[valid ? (ic >> 3) * QT + idx : 0];
              qk[kr] = valid ? q : 0.;
            }
          }
          STAMP(8);
#pragma nounroll
          for (int attempt = (newton_on ? 1 : 0) + (m ? 1 : 0); attempt >= 0 && !ok_step; --attempt) {
            const bool nwt = newton_on && attempt == (m ? 2 : 1);   // with the second-order part Q
            const bool use_cc = m != 0 && attempt >= 1;             // with the constraints' curvature
            bool ok_a = true;
            bool have_dl = false;
            unsigned long long heldmask = 0ull, lowmask = 0ull;   // constrained fits: the QP's working set
            if (reg_solve) {
             if constexpr (NT <= 2) {
              // small unconstrained system: one column per lane, in registers
              double x_own;
              // unrolled to the next multiple of four rows (the identity rows beyond the
              // variables would otherwise be eliminated too: work grows with the square)
              if (NT == 1) {
                if (nv <= 8) ok_a = column_solve<8>(Mp, nv, mu, is_free, lane, x_own, nwt, vinfo, ic, qk);
                else if (nv <= 12) ok_a = column_solve<12>(Mp, nv, mu, is_free, lane, x_own, nwt, vinfo, ic, qk);
                else ok_a = column_solve<16>(Mp, nv, mu, is_free, lane, x_own, nwt, vinfo, ic, qk);
              } else {
                if (nv <= 20) ok_a = column_solve<20>(Mp, nv, mu, is_free, lane, x_own, nwt, vinfo, ic, qk);
                else if (nv <= 24) ok_a = column_solve<24>(Mp, nv, mu, is_free, lane, x_own, nwt, vinfo, ic, qk);
                else if (nv <= 28) ok_a = column_solve<28>(Mp, nv, mu, is_free, lane, x_own, nwt, vinfo, ic, qk);
                else ok_a = column_solve<32>(Mp, nv, mu, is_free, lane, x_own, nwt, vinfo, ic, qk);
              }
              if (ok_a && lane < nv) dl[lane] = -x_own;
              wsync();
              have_dl = true;
             }
            } else if (CONS && m != 0) {
             if constexpr (CONS) {
              // Constrained fit: the bound-constrained QP on the tangent space (oracle cons_qp()):
              //   minimise g.d + 1/2 d.H.d   subject to   C d = 0,   lo - v <= d <= hi - v
              // by a primal active-set iteration; lane i = variable i (at most 29 of them).
              // held: -1 / +1 = kept on its lower / upper bound, 0 = free.  The sub-problems on
              // the free variables are solved in range-space form with H_FF + rho C_F^T C_F
              // (same minimiser on C p = 0, positive definite whenever the reduced Hessian is).
              double* dinvq = smem + SM::o_qd;
              double* gq = smem + SM::o_gq;
              const bool isv = lane < nv;
              const int li = isv ? lane : 0;
              auto Hm = [&](int i, int j) -> double {
                double h = Msym(Mp, i + 1, j + 1);
                const double hs = i == j ? mu * (h > 1e-300 ? h : 1.) : 0.;
                if (nwt) h += qvar(i, j);
                if (use_cc) h += ccurv(i, j);
                return h + hs;
              };
              int held = !isv ? 2 : (((freemask >> lane) & 1ull) ? 0 : (v[li] <= lo[li] ? -1 : 1));
              if (isv) dl[lane] = 0.;
              if (lane < MAXC) mult[lane] = 0.;
              wsync();
              int released = 0;
              const int qp_maxit = 3 * nv + 8;
              for (int qit = 0; qit < qp_maxit; ++qit) {
                const unsigned long long fb = __ballot(held == 0);
                const int nq = __popcll(fb);
                if (nq == 0) break;
                if (held == 0) fr[__popcll(fb & ((1ull << lane) - 1ull))] = lane;
                if (isv) {
                  double t = Mp[tri(lane + 1)];
                  for (int j = 0; j < nv; ++j) t += Hm(lane, j) * dl[j];
                  gq[lane] = t;
                }
                wsync();
                const double hmax = wave_max(held == 0 ? Hm(li, li) : 0.);
                double cm = 0.;
                if (lane < m)
                  for (int a = 0; a < nq; ++a) { const double c1 = Cj[lane * LDC + fr[a]]; cm += c1 * c1; }
                const double cmax = wave_max(cm);
                const double rho = cmax > 0. ? CTR_QP_RHO * hmax / cmax : 0.;
                for (int e = lane; e < tri(nq); e += WAVE) {
                  int a = (int)((sqrt(8. * e + 1.) - 1.) * 0.5);
                  while (tri(a + 1) <= e) ++a;
                  while (tri(a) > e) --a;
                  const int b = e - tri(a);
                  double t = Hm(fr[a], fr[b]);
                  for (int r = 0; r < m; ++r) t += rho * Cj[r * LDC + fr[a]] * Cj[r * LDC + fr[b]];
                  Hp[e] = t;
                }
                wsync();
                if (!chol_factor_w(Hp, dinvq, nq, lane)) { ok_a = false; break; }
                for (int a = lane; a < nq; a += WAVE) {
                  w[a] = gq[fr[a]];
                  for (int r = 0; r < m; ++r) Y[r * LDC + a] = Cj[r * LDC + fr[a]];
                }
                wsync();
                chol_solve_w(Hp, dinvq, nq, w, 1, 0, lane);
                chol_solve_w(Hp, dinvq, nq, Y, m, LDC, lane);
                if (lane < m * m) {
                  const int r = lane / m, s2 = lane % m;
                  double t = 0.;
                  for (int a = 0; a < nq; ++a) t += Cj[r * LDC + fr[a]] * Y[s2 * LDC + a];
                  Sc[r * MAXC + s2] = t;
                }
                if (lane < m) {
                  double t = 0.;
                  for (int a = 0; a < nq; ++a) t -= Cj[lane * LDC + fr[a]] * w[a];
                  mult[lane] = t;
                }
                wsync();
                small_spd_solve(Sc, mult);
                if (flag[0] == 0. && lane < MAXC) mult[lane] = 0.;
                wsync();
                // p = -H'^-1 (gq + C^T mult) of free variable fr[lane]; the ratio test along p
                double pa = 0., rel = 0., ratio = INFINITY;
                int vi = 0;
                if (lane < nq) {
                  vi = fr[lane];
                  double t = w[lane];
                  for (int r = 0; r < m; ++r) t += Y[r * LDC + lane] * mult[r];
                  pa = -t;
                  const double room = pa < 0. ? (lo[vi] - v[vi]) - dl[vi] : (hi[vi] - v[vi]) - dl[vi];
                  rel = fabs(pa) / (fabs(v[vi]) + 1.);
                  if (pa != 0.) ratio = room / pa;
                }
                const double pmax = wave_max(rel);
                const double rmin = -wave_max(-ratio);
                double alpha = rmin < 1. ? rmin : 1.;
                const unsigned long long blk = __ballot(lane < nq && rmin < 1. && ratio == rmin);
                const int blane = blk != 0ull ? __builtin_ctzll(blk) : -1;
                if (alpha < 0.) alpha = 0.;
                const int bvar = blane >= 0 ? __shfl(vi, blane) : -1;
                const int bside = blane >= 0 ? (__shfl(pa, blane) < 0. ? -1 : 1) : 0;
                wsync();
                if (pmax > 1e-15) {
                  if (lane < nq) dl[vi] += alpha * pa;
                  wsync();
                  if (bvar >= 0) {
                    if (lane == bvar) {
                      held = bside;
                      dl[lane] = (bside < 0 ? lo[lane] : hi[lane]) - v[lane];
                    }
                    wsync();
                    continue;
                  }
                  if (isv) {   // gradient of the model at the new d
                    double t = Mp[tri(lane + 1)];
                    for (int j = 0; j < nv; ++j) t += Hm(lane, j) * dl[j];
                    gq[lane] = t;
                  }
                  wsync();
                }
                // a minimiser on this working set: does a held variable want to leave its bound?
                double viol = 0.;
                if (isv && held != 0 && lo[li] < hi[li]) {
                  double s = gq[lane], sc = fabs(s);
                  for (int r = 0; r < m; ++r) {
                    const double t = Cj[r * LDC + lane] * mult[r];
                    s += t;
                    sc += fabs(t);
                  }
                  const double vv2 = held < 0 ? -s : s;   // > 0: the model falls when it moves inward
                  viol = vv2 > 1e-10 * sc + 1e-300 ? vv2 : 0.;
                }
                const double wv = wave_max(viol);
                if (!(wv > 0.) || released >= nv) break;
                const unsigned long long wb = __ballot(viol == wv);
                if (lane == __builtin_ctzll(wb)) held = 0;
                ++released;
              }
              heldmask = __ballot(held == -1 || held == 1);
              lowmask = __ballot(held == -1);
              wsync();
              have_dl = true;
             }
            } else {
              for (int e = lane; e < tri(nf); e += WAVE) {
                int a = (int)((sqrt(8. * e + 1.) - 1.) * 0.5);
                while (tri(a + 1) <= e) ++a;
                while (tri(a) > e) --a;
                const int b = e - tri(a);
                double h = Msym(Mp, fr[a] + 1, fr[b] + 1);
                const double hs = a == b ? mu * (h > 1e-300 ? h : 1.) : 0.;
                if (nwt) h += qvar(fr[a], fr[b]);
                if (use_cc) h += ccurv(fr[a], fr[b]);
                Hp[e] = h + hs;
              }
              wsync();
              STAMP(12);
              ok_a = chol_factor_w(Hp, dl, nf, lane);  // dl doubles as 1/diag until the step is built
              STAMP(13);
              if (ok_a) {
                for (int a = lane; a < nf; a += WAVE) {
                  w[a] = Mp[tri(fr[a] + 1)];
                  for (int r = 0; r < m; ++r) Y[r * LDC + a] = Cj[r * LDC + fr[a]];
                }
                wsync();
                chol_solve_w(Hp, dl, nf, w, 1, 0, lane);
                STAMP(14);
                if (m) {
                  chol_solve_w(Hp, dl, nf, Y, m, LDC, lane);
                  // tangent step, range-space form: (C H^-1 C^T) mult = -C H^-1 g, so that C d = 0
                  if (lane < m * m) {
                    const int r = lane / (m > 0 ? m : 1), s2 = lane % (m > 0 ? m : 1);
                    double t = 0.;
                    for (int a = 0; a < nf; ++a) t += Cj[r * LDC + fr[a]] * Y[s2 * LDC + a];
                    Sc[r * MAXC + s2] = t;
                  }
                  if (lane < m) {
                    double t = 0.;
                    for (int a = 0; a < nf; ++a) t -= Cj[lane * LDC + fr[a]] * w[a];
                    mult[lane] = t;
                  }
                  wsync();
                  small_spd_solve(Sc, mult);
                  if (flag[0] == 0. && lane < m) mult[lane] = 0.;
                  wsync();
                  ok_a = flag[0] != 0.;
                }
              }
            }
            STAMP(9);
            if (!ok_a) continue;
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
            // projected trial point, retracted onto the constraint manifold (first with the
            // variables the QP holds on their bounds pinned, so that they stay there)
            for (int i = lane; i < nv; i += WAVE) {
              const double t = v[i] + dl[i];
              double x = t < lo[i] ? lo[i] : (t > hi[i] ? hi[i] : t);
              if (((heldmask >> i) & 1ull) && lo[i] < hi[i]) x = ((lowmask >> i) & 1ull) ? lo[i] : hi[i];
              vt[i] = x;
              if (m) w[i] = x;
            }
            wsync();
            if (m) {
              bool okr = retract(vt, cvt, Cjt, cpair + MAXC, heldmask);
              if (!okr) {
                for (int i = lane; i < nv; i += WAVE) vt[i] = w[i];
                wsync();
                okr = retract(vt, cvt, Cjt, cpair + MAXC, 0ull);
              }
              if (!okr) continue;
            }
            stepmax = 0.;
            for (int i = lane; i < nv; i += WAVE) {
              const double d = vt[i] - v[i];
              dl[i] = d;
              stepmax = fmax(stepmax, fabs(d) / (fabs(v[i]) + 1.));
            }
            wsync();
            stepmax = wave_max(stepmax);
            STAMP(10);
            double partial = 0.;
            for (int i = lane; i < nv; i += WAVE) {
              double t = 0.;
              for (int j = 0; j < nv; ++j) t += Msym(Mp, i + 1, j + 1) * dl[j];
              if (nwt && reg_solve) {
                if (ic >= 0) {   // one variable per lane here: i == lane
                  const int j0 = L.nshared + (ic >> 3) * L.npf;
                  t += qk[0] * dl[j0 + L.slot[1]];
#pragma unroll
                  for (int a = 0; a < ND; ++a) t += qk[1 + a] * dl[j0 + L.slot[2 + a]];
                }
              } else if (nwt && fullq) {
                for (int j = 0; j < nv; ++j) t += qvar(i, j) * dl[j];
              } else if (nwt && vinfo[i] >= 0) {
                const int j0 = L.nshared + (vinfo[i] >> 3) * L.npf;
                // (with constraints the model of the OBJECTIVE along the retracted step: the
                // curvature of the manifold is in dl itself, not in the matrix)
                for (int j = j0; j < j0 + L.npf; ++j) t += qvar(i, j) * dl[j];
              }
              partial += dl[i] * (Mp[tri(i + 1)] + 0.5 * t);
            }
            pred = -wave_sum(partial);
            pred = bcast0(pred);
            stepmax = bcast0(stepmax);
            STAMP(11);
            if (attempt >= 1 && !(pred > -tiny)) continue;
            ok_step = true;
          }
          STAMP(4);
          if (!ok_step) {
            mu *= nu; nu *= 2.; last_acc = false;
            if (mu > 1e30) { status = CTR_STATUS_NO_CONVERGENCE; failed = true; }
            next = BP_STEP_ONLY;
          } else {
            converged = (last_acc && stepmax <= xtol) || fabs(pred) <= tiny;
            // converging faster than linearly and the step after this one would be below xtol:
            // finished by TAKING this step, without the pixel pass that would only confirm it
            // (oracle solve(): fast exit)
            if (!converged && last_acc && pred > 0. && stepmax < prev_step && isfinite(prev_step) &&
                stepmax * (stepmax / prev_step) <= xtol) {
              for (int i = lane; i < nv; i += WAVE) v[i] = vt[i];
              wsync();
              S = fmax(S - 2. * pred, 0.);
              converged = true;
            }
            trial_step = stepmax;
            next = BP_EVAL_TRIAL;
            if (!converged && !(pred > 0.)) {
              // the model itself predicts no decrease: rejected without a pixel pass
              mu *= nu; nu *= 2.; last_acc = false;
              if (mu > 1e30) { status = CTR_STATUS_NO_CONVERGENCE; failed = true; }
              next = BP_STEP_ONLY;
            }
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
          for (int kk = 0; kk < NPC; ++kk) {
            const int b = L.vidx(kk, i);
            if (b >= 0) cur[i * CTR_MAX_PARAMS + kk] = v[b];
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
    if (k.params_std != nullptr) {
      // refine.py:400-406: std = sqrt(2 diag(inv(Hessian of F))) at the solution, all variables
      // free: std_j = sqrt(P norm [(J^T J + Q)^-1]_jj).  Mp belongs to the accepted point v; Q =
      // sum_p res_p d2res_p/dv dv in ALL variables is summed here in one more pass over the
      // window (oracle: full_second_order), whatever the parameter modes.
      double* sd = vt;   // (free now)
      bool pd = ok && n > 0 && FIT == CTR_FIT_GAUSS;   // (the second derivatives are the gaussian's)
      if (pd) {
        for (int e = lane; e < tri(nv); e += WAVE) {
          int a = (int)((sqrt(8. * e + 1.) - 1.) * 0.5);
          while (tri(a + 1) <= e) ++a;
          while (tri(a) > e) --a;
          const int b = e - tri(a);
          Hp[e] = Msym(Mp, a + 1, b + 1);
        }
        fill_fpar(v, true);
        wsync();
        second_order_pass(Hp, cv);   // (28 doubles of the constraint scratch, free now)
        wsync();
        pd = chol_factor_w(Hp, dl, nv, lane);
      }
      if (pd) {
        const double scale = (double)Pround * norm;
        for (int j = 0; j < nv; ++j) {
          for (int i = lane; i < nv; i += WAVE) w[i] = i == j ? 1. : 0.;
          wsync();
          chol_solve_w(Hp, dl, nv, w, 1, 0, lane);
          if (lane == 0) sd[j] = sqrt(scale * w[j]);
          wsync();
        }
      }
      double* ps = k.params_std + (size_t)f0 * NP;
      for (int e = lane; e < n * NP; e += WAVE) {
        const int b = L.vidx(e % NP, e / NP);
        ps[e] = (pd && b >= 0) ? sd[b] : NAN;
      }
    }
    if (lane == 0) {
      k.status[cl] = status;
      k.cost[cl] = ok ? rms : NAN;
      k.n_rounds[cl] = status == CTR_STATUS_NONFINITE || n <= 0 ? 0
                     : (ok || status == CTR_STATUS_RMS_DEV ? round : round + 1);
      k.n_iter[cl] = iters;
    }
  }
}



#endif  // CTREFINE_BLOCK_KERNEL_H
