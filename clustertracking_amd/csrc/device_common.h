// device_common.h -- kernel arguments, pixel loads, wave reductions, masks, variable layout
// Part of the MI355X cluster-refinement engine; included by ctrefine.hip inside its
// anonymous namespace (device code only, gfx950).
#ifndef CTREFINE_DEVICE_COMMON_H
#define CTREFINE_DEVICE_COMMON_H


typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;
constexpr int MAXC = 6;       // equality constraints per cluster
constexpr double CTR_QP_RHO = 1e3;  // weight of rho C^T C in the sub-problems of a constrained step (oracle cons_qp)
constexpr int MAXF = 64;      // features per cluster
constexpr int MAXNT = 8;      // 16*8 = 128 columns >= CTR_MAX_VARS + 1
constexpr int FP = 14;        // derived per-feature constants (see fill_fpar)
// at the iteration limit a fit whose last accepted step lowered the objective by less than this
// (relative) is reported as converged; same rule as STALL_TOL in oracle/ctr_oracle.c
#define CTR_STALL_TOL 1e-9

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

// refine.py:37-40 / preprocessing.py:12-49: pixel idx of the lowpass-filtered WINDOW (origin,
// wshape), computed from the raw frame on the fly.  The reference correlates the window axis by
// axis (axis 0 first) with scipy.ndimage.correlate1d(mode='constant', cval=0), in place; written
// out for one pixel that is the nest below, each level with SciPy's order for a symmetric
// kernel: centre tap, then the pairs from the outermost inwards.  Values <= threshold -> 0.
template <int ND>
__device__ double lowpass_pixel(const void* frame, int dtype, const long* fshape, const int* origin,
                                             const int* wshape, const int* idx, const double* lp_w,
                                             const int* lp_half, double threshold) {
  auto raw = [&](int z, int y, int x) -> double {   // window coordinates; 0 beyond the window
    if (x < 0 || x >= wshape[ND - 1] || y < 0 || y >= wshape[ND - 2]) return 0.;
    if (ND == 3 && (z < 0 || z >= wshape[0])) return 0.;
    const size_t off = ND == 3
        ? ((size_t)(z + origin[0]) * fshape[1] + (y + origin[1])) * fshape[ND - 1] + (x + origin[ND - 1])
        : (size_t)(y + origin[0]) * fshape[ND - 1] + (x + origin[ND - 1]);
    return load_pixel(frame, dtype, off);
  };
  const int hz = ND == 3 ? lp_half[0] : 0, hy = lp_half[ND - 2], hx = lp_half[ND - 1];
  const double *wz = lp_w, *wy = lp_w + (ND - 2) * LP_STRIDE, *wx = lp_w + (ND - 1) * LP_STRIDE;
  const int z0 = ND == 3 ? idx[0] : 0, y0 = idx[ND - 2], x0 = idx[ND - 1];
  auto along_z = [&](int y, int x) -> double {
    if (ND != 3) return raw(0, y, x);
    double t = raw(z0, y, x) * wz[hz];
    for (int j = -hz; j < 0; ++j) t += (raw(z0 + j, y, x) + raw(z0 - j, y, x)) * wz[hz + j];
    return t;
  };
  auto along_y = [&](int x) -> double {
    if (x < 0 || x >= wshape[ND - 1]) return 0.;
    double t = along_z(y0, x) * wy[hy];
    for (int j = -hy; j < 0; ++j) {
      const double a = y0 + j >= 0 ? along_z(y0 + j, x) : 0.;
      const double b = y0 - j < wshape[ND - 2] ? along_z(y0 - j, x) : 0.;
      t += (a + b) * wy[hy + j];
    }
    return t;
  };
  double t = along_y(x0) * wx[hx];
  for (int j = -hx; j < 0; ++j) t += (along_y(x0 + j) + along_y(x0 - j)) * wx[hx + j];
  return t > threshold ? t : 0.;
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
// Four wave-wide sums for the price of one and a half: after two exchange steps every
// 16-lane row carries one of the four values, which the DPP row reduction finishes.
// Returns, in every lane of row r (lanes 16r..16r+15), the sum over the wave of v_r.
__device__ __forceinline__ double wave_sum4(double v0, double v1, double v2, double v3, int lane) {
  const bool up = (lane & 32) != 0;
  double ka = up ? v2 : v0, kb = up ? v3 : v1;
  const double sa = up ? v0 : v2, sb = up ? v1 : v3;
  ka += __shfl_xor(sa, 32);
  kb += __shfl_xor(sb, 32);
  const bool hi = (lane & 16) != 0;
  double kk = hi ? kb : ka;
  const double ss = hi ? ka : kb;
  kk += __shfl_xor(ss, 16);
  return row_sum(kk);
}

// Radial profiles (fitfunc.py:112-146; oracle/ctr_oracle.c:profile, same operations in the same
// order): value g(r2; e), dg/dr2 and dg/de with e the profile parameter (ring: 'thickness',
// disc: 'disc_size').  The disc's derivatives are the analytic ones of fitfunc.py:121-131 (the
// reference differentiates numerically).
template <int FIT, int ND>
__device__ __forceinline__ void profile_dev(double r2, double e, double& g, double& dg_dr2, double& dg_de) {
  if constexpr (FIT == CTR_FIT_RING) {
    const double r = sqrt(r2), num = r - 1. + e;
    const double f = exp(-0.5 * ND * ((num / e) * (num / e)));
    g = f;
    dg_dr2 = f * (-0.5 * ND / (r * (e * e))) * num;
    dg_de = f * ND * (num * num / (e * e * e) - num / (e * e));
  } else if constexpr (FIT == CTR_FIT_DISC) {
    if (e > 0.) {
      const bool clamped = e >= 1.;
      const double ds = clamped ? 0.999 : e;
      if (r2 > ds * ds) {
        const double r = sqrt(r2), w = 1. - ds, u = (r - ds) / w;
        const double f = exp(u * u * ND / -2.);
        g = f;
        dg_dr2 = f * (-(double)ND * u / w) / (2. * r);
        dg_de = clamped ? 0. : f * (-(double)ND * u) * ((r - 1.) / (w * w));
      } else {
        g = r2 == r2 ? 1. : NAN;
        dg_dr2 = 0.;
        dg_de = 0.;
      }
    } else {
      g = exp(-0.5 * ND * r2);
      dg_dr2 = -0.5 * ND * g;
      dg_de = 0.;
    }
  } else {
    g = exp(-0.5 * ND * r2);
    dg_dr2 = -0.5 * ND * g;
    dg_de = 0.;
  }
}

// inv_series_<N> (fitfunc.py:148-154): g = e[0] / y(r2), y = np.polyval([1, e[1], ..., e[N]], r2)
// = r2^N + e[1] r2^(N-1) + ... + e[N] in Horner's order; nx = N + 1 profile parameters, at most
// INV_NX.  The reference has no derivative of it (its SLSQP differentiates the objective
// numerically); these are the analytic ones: dg/dr2 = -e[0] y'/y^2, dg/de[0] = 1/y,
// dg/de[k] = -e[0] r2^(N-k) / y^2.  Same expressions as oracle/ctr_oracle.c:profile_inv.
constexpr int INV_NX = 7;
__device__ __forceinline__ void profile_inv_dev(int nx, double r2, const double* e, double& g, double& dg_dr2, double* dg_de) {
  double y = 1., dy = 0.;
#pragma unroll
  for (int t = 1; t < INV_NX; ++t)
    if (t < nx) { dy = dy * r2 + y; y = y * r2 + e[t]; }
  const double inv = 1. / y, c = -e[0] * (inv * inv);
  g = e[0] / y;
  dg_dr2 = c * dy;
  dg_de[0] = inv;
  double pw = 1.;
#pragma unroll
  for (int t = INV_NX - 1; t >= 1; --t) {
    dg_de[t] = 0.;
    if (t < nx) { dg_de[t] = c * pw; pw *= r2; }
  }
}

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



#endif  // CTREFINE_DEVICE_COMMON_H
