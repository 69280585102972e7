/* ctr_oracle.c -- CPU restatement of the refine hot path.  TEST INFRASTRUCTURE.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this; the product (clustertracking_amd + libctrefine.so) never does.
 *
 * What follows the reference line by line (and is pinned by the golden
 * fixtures generated from the reference, tests/golden/):
 *   window()            masks.py:30-68     (round-half-even, in-bounds filter, clip)
 *   in_mask()           refine.py:43-44    (ellipse on the UNROUNDED coords, <= 1)
 *   union / P           refine.py:47-58    (P = pixels in the union mask)
 *   layout / pack       fitfunc.py:207-315 (parameter-major vector, groups=None)
 *   model + derivatives fitfunc.py:14-118  (r2_*, dr2_*, gauss_func/gauss_dfunc)
 *   objective, gradient fitfunc.py:436-487 (each feature only inside its own mask)
 *   packed bounds       fitfunc.py:554-557 (shared parameter: min of lows, max of highs)
 *   constraints         constraints.py:59-137 (dimer, trimer, tetramer)
 *   round loop, rules   refine.py:343-430
 *
 * What does NOT follow the reference: the minimiser.  The reference hands the
 * scalar objective to scipy.optimize.minimize(method='SLSQP', tol=1e-6)
 * (refine.py:373-375; SciPy is a third-party dependency, not in the repo).
 * Here -- exactly as in the HIP engine this file checks -- the same objective
 * is minimised under the same bounds/constraints by a bounded Levenberg-
 * Marquardt iteration on the residual vector (active set for the box,
 * range-space SQP step for the equality constraints).  Parity of the fitted
 * parameters with the reference is therefore "same minimiser of the same
 * problem", pinned by the fixtures' converged reference output (oracle B,
 * tol=1e-14) and bounded by the reference's own default-tolerance scatter
 * (oracle A).  The faithful SLSQP restatement is oracle/ref_numpy.py.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ctrefine.h"

#define MAXV CTR_MAX_VARS
#define MAXC 6 /* equality constraints per cluster */

typedef struct {
  int n, nd, np, nv, nsz;
  int var_of[CTR_MAX_PARAMS];   /* first variable of the column, -1 if constant */
  int per_feat[CTR_MAX_PARAMS]; /* 1: one variable per feature (mode 'var') */
} layout_t;

typedef struct {
  const ctr_problem* p;
  const ctr_batch* b;
  layout_t L;
  const double* pconst; /* [n][np] parameters supplying the constants */
  const void* frame;    /* pixels of this cluster's frame */
  int origin[3], wshape[3];
  const double* mcoords; /* [n][nd] coordinates the masks are centred on */
  int n_cons;            /* active equality constraints (0 if size mismatch) */
  double* fwin;          /* lowpass-filtered window [wshape] (refine.py:37-40), NULL without noise_size */
} ctx_t;

/* ---- small helpers ------------------------------------------------------- */

static double pixel(const void* base, int dtype, size_t i) {
  switch (dtype) {
    case CTR_DTYPE_U8: return (double)((const uint8_t*)base)[i];
    case CTR_DTYPE_U16: return (double)((const uint16_t*)base)[i];
    case CTR_DTYPE_I16: return (double)((const int16_t*)base)[i];
    case CTR_DTYPE_I32: return (double)((const int32_t*)base)[i];
    case CTR_DTYPE_F32: return (double)((const float*)base)[i];
    default: return ((const double*)base)[i];
  }
}

static size_t dtype_size(int dtype) {
  switch (dtype) {
    case CTR_DTYPE_U8: return 1;
    case CTR_DTYPE_U16: case CTR_DTYPE_I16: return 2;
    case CTR_DTYPE_I32: case CTR_DTYPE_F32: return 4;
    default: return 8;
  }
}

static int make_layout(const ctr_problem* p, int n, layout_t* L) {
  int nv = 0;
  L->n = n; L->nd = p->ndim; L->np = p->n_params;
  L->nsz = p->isotropic ? 1 : p->ndim;
  for (int k = 0; k < p->n_params; ++k) {
    int m = p->modes[k];
    if (m == CTR_MODE_CONST) { L->var_of[k] = -1; L->per_feat[k] = 0; }
    else if (m == CTR_MODE_VAR) { L->var_of[k] = nv; L->per_feat[k] = 1; nv += n; }
    else { L->var_of[k] = nv; L->per_feat[k] = 0; nv += 1; } /* groups=None: one per cluster */
  }
  L->nv = nv;
  return nv;
}

int ctro_cluster_n_vars(const ctr_problem* p, int n) {
  layout_t L;
  return make_layout(p, n, &L);
}

/* ---- lowpass of the window (refine.py:37-40, preprocessing.py:12-49) ------- */

/* trackpy.masks.gaussian_kernel(sigma, truncate=4) -- trackpy (v0.3/0.4, masks.py) is absent
 * from this image: restated from its published source, PARITY UNPINNED for this function alone:
 *   lw = int(truncate * sigma + 0.5); x = arange(-lw, lw + 1)
 *   result = exp(x**2 / (-2 * sigma**2)); return result / sum(result)
 * w[2*lw+1]; returns lw.  The sum follows numpy's pairwise order for short arrays. */
int ctro_gaussian_kernel(double sigma, double* w) {
  const int lw = (int)(4.0 * sigma + 0.5), nw = 2 * lw + 1;
  double sum;
  for (int i = 0; i < nw; ++i) {
    const double x = (double)(i - lw);
    w[i] = exp((x * x) / (-2. * (sigma * sigma)));
  }
  if (nw < 8) {
    sum = 0.;
    for (int i = 0; i < nw; ++i) sum += w[i];
  } else {
    double r[8];
    int i;
    for (int j = 0; j < 8; ++j) r[j] = w[j];
    for (i = 8; i < nw - (nw % 8); i += 8)
      for (int j = 0; j < 8; ++j) r[j] += w[i + j];
    sum = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < nw; ++i) sum += w[i];
  }
  for (int i = 0; i < nw; ++i) w[i] /= sum;
  return lw;
}

/* preprocessing.py:41-49 on a window of doubles, in place: per axis with size > 0
 * scipy.ndimage.correlate1d(result, kernel, axis, output=result, mode='constant', cval=0)
 * -- SciPy's loop for a symmetric kernel: centre tap first, then the pairs from the outermost
 * inwards (ni_filters.c:NI_Correlate1D) -- then values <= threshold become 0 (NaN too). */
void ctro_lowpass(int nd, const int* wshape, const double* noise_size, double threshold, double* win) {
  size_t total = 1;
  int maxlen = 0;
  for (int a = 0; a < nd; ++a) {
    total *= (size_t)wshape[a];
    if (wshape[a] > maxlen) maxlen = wshape[a];
  }
  for (int a = 0; a < nd; ++a) {
    double w[2 * 16 + 1 + 8];
    if (!(noise_size[a] > 0.)) continue;
    const int lw = ctro_gaussian_kernel(noise_size[a] > CTR_MAX_NOISE_SIZE ? CTR_MAX_NOISE_SIZE : noise_size[a], w);
    const int len = wshape[a];
    size_t stride = 1;
    for (int b2 = a + 1; b2 < nd; ++b2) stride *= (size_t)wshape[b2];
    double* line = calloc((size_t)len + 2 * (size_t)lw, sizeof(double));
    for (size_t start = 0; start < total; ++start) {
      /* every line along axis a: start = index with coordinate 0 on that axis */
      if ((start / stride) % (size_t)len != 0) continue;
      for (int i = 0; i < len; ++i) line[lw + i] = win[start + (size_t)i * stride];
      for (int i = 0; i < len; ++i) {
        const double* il = line + lw + i;
        double t = il[0] * w[lw];
        for (int jj = -lw; jj < 0; ++jj) t += (il[jj] + il[-jj]) * w[lw + jj];
        win[start + (size_t)i * stride] = t;
      }
    }
    free(line);
  }
  for (size_t i = 0; i < total; ++i) win[i] = win[i] > threshold ? win[i] : 0.;
}

static int has_lowpass(const ctr_problem* p) {
  if (p->flags & CTR_FLAG_WINDOW_FILTER) return 1;   /* refine.py:37: noise_size is not None */
  for (int a = 0; a < p->ndim; ++a)
    if (p->noise_size[a] > 0.) return 1;
  return 0;
}

/* the filtered window of this round (c->origin / c->wshape set), or NULL */
static void make_fwin(ctx_t* c) {
  const int nd = c->p->ndim;
  const int64_t* fs = c->b->shape;
  free(c->fwin);
  c->fwin = NULL;
  if (!has_lowpass(c->p)) return;
  const int w0 = nd == 3 ? c->wshape[0] : 1, w1 = c->wshape[nd - 2], w2 = c->wshape[nd - 1];
  double* win = malloc(sizeof(double) * (size_t)w0 * w1 * w2);
  for (int z = 0; z < w0; ++z)
    for (int y = 0; y < w1; ++y)
      for (int x = 0; x < w2; ++x) {
        const size_t off = nd == 3
            ? ((size_t)(z + c->origin[0]) * fs[1] + (y + c->origin[1])) * fs[2] + (x + c->origin[2])
            : (size_t)(y + c->origin[0]) * fs[1] + (x + c->origin[1]);
        win[((size_t)z * w1 + y) * w2 + x] = pixel(c->frame, c->b->frame_dtype, off);
      }
  ctro_lowpass(nd, c->wshape, c->p->noise_size, c->p->threshold, win);
  c->fwin = win;
}

/* parameter k of feature i at the trial vector v */
static inline double par(const ctx_t* c, const double* v, int i, int k) {
  int b = c->L.var_of[k];
  if (b < 0) return c->pconst[i * c->L.np + k];
  return v[b + (c->L.per_feat[k] ? i : 0)];
}

/* masks.py:42-68.  Returns 0 when no coordinate is inside the frame. */
static int window(int nd, const int64_t* shape, const int32_t* radius,
                  const double* coords, int n, int* origin, int* wshape) {
  long lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
  int any = 0;
  for (int i = 0; i < n; ++i) {
    long ci[3];
    int ok = 1;
    for (int a = 0; a < nd; ++a) {
      double r = nearbyint(coords[i * nd + a]); /* default rounding mode: half to even (masks.py:54) */
      ci[a] = (long)r;
      if (!(ci[a] >= -(long)radius[a] && ci[a] < (long)shape[a] + radius[a])) ok = 0;
    }
    if (!ok) continue;
    for (int a = 0; a < nd; ++a) {
      if (!any || ci[a] < lo[a]) lo[a] = ci[a];
      if (!any || ci[a] > hi[a]) hi[a] = ci[a];
    }
    any = 1;
  }
  if (!any) return 0;
  for (int a = 0; a < nd; ++a) {
    long l = lo[a] - radius[a], u = hi[a] + radius[a] + 1;
    if (l < 0) l = 0;
    if (u > shape[a]) u = shape[a];
    origin[a] = (int)l;
    wshape[a] = (int)(u - l);
  }
  return 1;
}

int ctro_window(int nd, const int64_t* shape, const int32_t* radius,
                const double* coords, int n, int32_t* origin, int32_t* wshape) {
  int o[3], w[3];
  if (!window(nd, shape, radius, coords, n, o, w)) return 0;
  for (int a = 0; a < nd; ++a) { origin[a] = o[a]; wshape[a] = w[a]; }
  return 1;
}

/* refine.py:43: sum(((idx - (coord - origin)) / radius)**2) <= 1, no FMA */
#pragma GCC push_options
#pragma GCC optimize("fp-contract=off")
static inline int in_mask(int nd, const int* idx, const double* coord,
                          const int* origin, const int32_t* radius) {
  double s = 0.;
  for (int a = 0; a < nd; ++a) {
    double rel = coord[a] - (double)origin[a];
    double t = ((double)idx[a] - rel) / (double)radius[a];
    double t2 = t * t;
    s = s + t2;
  }
  return s <= 1.;
}
#pragma GCC pop_options

/* mask pixel counts (for known-answer tests): P and per-feature counts */
long ctro_mask_counts(int nd, const int64_t* shape, const int32_t* radius,
                      const double* coords, int n, int64_t* per_feature) {
  int origin[3], ws[3] = {1, 1, 1}, idx[3];
  long P = 0;
  if (!window(nd, shape, radius, coords, n, origin, ws)) return -1;
  for (int i = 0; i < n; ++i) per_feature[i] = 0;
  int w0 = nd == 3 ? ws[0] : 1, w1 = ws[nd - 2], w2 = ws[nd - 1];
  for (int z = 0; z < w0; ++z)
    for (int y = 0; y < w1; ++y)
      for (int x = 0; x < w2; ++x) {
        int any = 0;
        if (nd == 3) { idx[0] = z; idx[1] = y; idx[2] = x; } else { idx[0] = y; idx[1] = x; }
        for (int i = 0; i < n; ++i)
          if (in_mask(nd, idx, coords + i * nd, origin, radius)) { per_feature[i]++; any = 1; }
        P += any;
      }
  return P;
}

/* ---- constraints (constraints.py:59-137) --------------------------------- */

static const int PAIRS[6][2] = {{0, 1}, {1, 2}, {0, 2}, {1, 3}, {0, 3}, {2, 3}};

static int n_constraints(const ctr_problem* p, int n) {
  switch (p->constraint_kind) {
    case CTR_CONS_DIMER: return n == 2 ? 1 : 0;
    case CTR_CONS_TRIMER: return n == 3 ? 3 : 0;
    case CTR_CONS_TETRAMER: return n == 4 ? (p->ndim == 2 ? 4 : 6) : 0;
    default: return 0;
  }
}

/* c[m] and Jacobian Cj[m][nv] at v.  2D tetramer: the constrained pairs are the 4 smallest of
 * the 6 pair distances (constraints.py:102-114); with use_pairs != NULL those pairs are used
 * as they are (a smooth branch of that function) instead of being ranked again. */
static void eval_constraints(const ctx_t* c, const double* v, double* cv, double* Cj, int* pair_of,
                             const int* use_pairs) {
  const int nd = c->L.nd, nv = c->L.nv, m = c->n_cons;
  int npairs = c->p->constraint_kind == CTR_CONS_DIMER ? 1
             : c->p->constraint_kind == CTR_CONS_TRIMER ? 3 : 6;
  double d2[6];
  int order[6];
  if (m == 0) return;
  for (int q = 0; q < npairs; ++q) {
    double s = 0.;
    for (int a = 0; a < nd; ++a) {
      double t = (par(c, v, PAIRS[q][0], 2 + a) - par(c, v, PAIRS[q][1], 2 + a)) /
                 c->p->constraint_dist[a];
      s += t * t;
    }
    d2[q] = s;
    order[q] = q;
  }
  if (use_pairs) {
    for (int r = 0; r < m; ++r) order[r] = use_pairs[r];
  } else if (c->p->constraint_kind == CTR_CONS_TETRAMER && nd == 2) {
    /* the 4 smallest of the 6 pair distances (constraints.py:102-114) */
    for (int i = 1; i < 6; ++i) {
      int o = order[i], j = i - 1;
      while (j >= 0 && d2[order[j]] > d2[o]) { order[j + 1] = order[j]; --j; }
      order[j + 1] = o;
    }
  }
  memset(Cj, 0, sizeof(double) * (size_t)m * nv);
  for (int r = 0; r < m; ++r) {
    int q = order[r];
    int i0 = PAIRS[q][0], i1 = PAIRS[q][1];
    if (pair_of) pair_of[r] = q;
    cv[r] = 1. - d2[q];
    for (int a = 0; a < nd; ++a) {
      int k = 2 + a, b = c->L.var_of[k];
      if (b < 0) continue;
      double da = c->p->constraint_dist[a];
      double t = -2. * (par(c, v, i0, k) - par(c, v, i1, k)) / (da * da);
      Cj[r * nv + b + (c->L.per_feat[k] ? i0 : 0)] += t;
      Cj[r * nv + b + (c->L.per_feat[k] ? i1 : 0)] -= t;
    }
  }
}

/* ---- radial profiles (fitfunc.py:112-146) ----------------------------------------------------
 * value g(r2; e) with dg/dr2 and dg/de, e = the profile's extra parameter:
 *   gauss  exp(-nd/2 r2)                                                 (fitfunc.py:112-118)
 *   ring   exp(-nd/2 ((r - 1 + t)/t)^2), r = sqrt(r2), t = 'thickness'   (fitfunc.py:134-146;
 *          the reference's own derivatives)
 *   disc   1 for r <= ds, exp(-nd/2 ((r - ds)/(1 - ds))^2) beyond, ds = 'disc_size' (<= 0: the
 *          gaussian, >= 1: 0.999; fitfunc.py:121-131).  The reference has no derivative of it (its
 *          SLSQP differentiates the objective numerically, fitfunc.py:451-452); these are the
 *          analytic ones of the same function.
 * ring and disc are not "continuous" (fitfunc.py:195-204): their r2 is NaN within one pixel of
 * the centre (r2_*_safe, fitfunc.py:20-26,43-49,67-73,91-98): nansum skips those pixels for the
 * ring; the disc has the value 1 there (its function only overwrites where r2 > disc_size^2). */
/* inv_series_<N> (fitfunc.py:148-154): g = e[0] / y(r2) with y = np.polyval([1, e[1], ..., e[N]], r2)
 * = r2^N + e[1] r2^(N-1) + ... + e[N], evaluated in polyval's (Horner's) order; nx = N + 1 profile
 * parameters 'signal_mult', 'param_a', ... (fitfunc.py:334-343).  "continuous": the plain r2.  The
 * reference has no derivative of it (SLSQP differentiates the objective numerically); these are
 * the analytic ones: dg/dr2 = -e[0] y'/y^2, dg/de[0] = 1/y, dg/de[k] = -e[0] r2^(N-k)/y^2. */
#define INV_NX 7
static void profile_inv(int nx, double r2, const double* e, double* g, double* dg_dr2, double* dg_de) {
  double y = 1., dy = 0.;
  for (int t = 1; t < nx; ++t) { dy = dy * r2 + y; y = y * r2 + e[t]; }
  const double inv = 1. / y, c = -e[0] * (inv * inv);
  *g = e[0] / y;
  *dg_dr2 = c * dy;
  dg_de[0] = inv;
  double pw = 1.;
  for (int t = nx - 1; t >= 1; --t) { dg_de[t] = c * pw; pw *= r2; }
}

static void profile(int fit, int nd, double r2, double e, double* g, double* dg_dr2, double* dg_de) {
  if (fit == CTR_FIT_RING) {
    const double r = sqrt(r2), num = r - 1. + e;
    const double f = exp(-0.5 * nd * ((num / e) * (num / e)));
    *g = f;
    *dg_dr2 = f * (-0.5 * nd / (r * (e * e))) * num;
    *dg_de = f * nd * (num * num / (e * e * e) - num / (e * e));
    return;
  }
  if (fit == CTR_FIT_DISC && e > 0.) {
    const int clamped = e >= 1.;
    const double ds = clamped ? 0.999 : e;
    if (r2 > ds * ds) {
      const double r = sqrt(r2), w = 1. - ds, u = (r - ds) / w;
      const double f = exp(u * u * nd / -2.);
      *g = f;
      *dg_dr2 = f * (-(double)nd * u / w) / (2. * r);
      *dg_de = clamped ? 0. : f * (-(double)nd * u) * ((r - 1.) / (w * w));
    } else {
      *g = r2 == r2 ? 1. : NAN;
      *dg_dr2 = 0.;
      *dg_de = 0.;
    }
    return;
  }
  *g = exp(-0.5 * nd * r2);
  *dg_dr2 = -0.5 * nd * *g;
  *dg_de = 0.;
}

/* ---- objective: S = sum r^2, g = J^T r, A = J^T J ------------------------- */

static void eval_cluster(const ctx_t* c, const double* v, double* S_out, double* g,
                         double* A, double* Q, long* P_out) {
  const layout_t* L = &c->L;
  const int nd = L->nd, nv = L->nv, n = L->n, np = L->np;
  const int64_t* fshape = c->b->shape;
  const int dtype = c->b->frame_dtype;
  double prm[CTR_MAX_PARAMS];
  double* row = malloc(sizeof(double) * (size_t)(nv > 0 ? nv : 1));
  int* nz = malloc(sizeof(int) * (size_t)(nv > 0 ? nv : 1));
  double S = 0.;
  long P = 0;
  int model_nan = 0;
  const int bgvar = L->var_of[0];
  const double bg = par(c, v, 0, 0);
  const int w0 = nd == 3 ? c->wshape[0] : 1, w1 = c->wshape[nd - 2], w2 = c->wshape[nd - 1];
  if (g) memset(g, 0, sizeof(double) * nv);
  if (A) memset(A, 0, sizeof(double) * nv * nv);
  if (Q) memset(Q, 0, sizeof(double) * nv * nv);
  /* second-order part (see solve()): U[i][a][b] = sum_p res * J_pos_a * dE/dpos_b of feature i */
  double (*U)[3][3] = calloc((size_t)n, sizeof(double) * 9);
  struct hf_t { int i; double J[3], E[3]; }* hf = malloc(sizeof(struct hf_t) * (size_t)n);
  /* per feature: a box (window indices) that contains its mask -- features whose box misses
   * the pixel are skipped without the ellipse test (large clusters); results unchanged */
  int (*box)[2][3] = malloc(sizeof(int) * 6 * (size_t)n);
  for (int i = 0; i < n; ++i)
    for (int a = 0; a < nd; ++a) {
      double rel = c->mcoords[i * nd + a] - (double)c->origin[a];
      box[i][0][a] = (int)floor(rel - c->p->radius[a]) - 1;
      box[i][1][a] = (int)ceil(rel + c->p->radius[a]) + 1;
    }
  for (int z = 0; z < w0; ++z)
    for (int y = 0; y < w1; ++y)
      for (int x = 0; x < w2; ++x) {
        int idx[3];
        double mesh[3];
        size_t off;
        int any = 0, nnz = 0, nhf = 0, safe_nan = 0;
        double res = 0., pv = 0.;
        if (nd == 3) {
          idx[0] = z; idx[1] = y; idx[2] = x;
          off = ((size_t)(z + c->origin[0]) * fshape[1] + (y + c->origin[1])) * fshape[2] + (x + c->origin[2]);
        } else {
          idx[0] = y; idx[1] = x;
          off = (size_t)(y + c->origin[0]) * fshape[1] + (x + c->origin[1]);
        }
        for (int a = 0; a < nd; ++a) mesh[a] = (double)(idx[a] + c->origin[a]);
        for (int i = 0; i < n; ++i) {
          int inbox = 1;
          for (int a = 0; a < nd; ++a) inbox = inbox && idx[a] >= box[i][0][a] && idx[a] <= box[i][1][a];
          if (!inbox) continue;
          if (!in_mask(nd, idx, c->mcoords + i * nd, c->origin, c->p->radius)) continue;
          if (!any) {
            any = 1;
            pv = c->fwin ? c->fwin[((size_t)z * w1 + y) * w2 + x] : pixel(c->frame, dtype, off);
            res = pv - bg;
            if (bgvar >= 0) { row[bgvar] = -1.; nz[nnz++] = bgvar; }
          }
          for (int k = 1; k < np; ++k) prm[k] = par(c, v, i, k);
          const double sig = prm[1];
          double r2 = 0., dr2[6], qraw = 0.;
          if (c->p->isotropic) {
            const double size = prm[2 + nd];
            double q = 0.;
            if (!(size > 0.)) model_nan = 1; /* a size of 0 (its lower bound): no valid model */
            for (int a = nd - 1; a >= 0; --a) { /* x first (fitfunc.py:17,40) */
              double d = mesh[a] - prm[2 + a];
              q += d * d;
            }
            qraw = q;
            r2 = q / (size * size);
            for (int a = 0; a < nd; ++a) dr2[a] = (prm[2 + a] - mesh[a]) * (2. / (size * size));
            dr2[nd] = q * (-2. / (size * size * size));
          } else {
            for (int a = nd - 1; a >= 0; --a) {
              double d = mesh[a] - prm[2 + a], sz = prm[2 + nd + a];
              if (!(sz > 0.)) model_nan = 1;
              qraw += d * d;
              r2 += d * d / (sz * sz);
              dr2[a] = (prm[2 + a] - mesh[a]) * (2. / (sz * sz));
              dr2[nd + a] = d * d * (-2. / (sz * sz * sz));
            }
          }
          double gv, dg, dge, dgx[INV_NX];
          const int fit = c->p->fit_function;
          const int nx = np - (2 + nd + L->nsz);   /* profile parameters behind the sizes */
          /* r2_*_safe (fitfunc.py:20-26,...): within one pixel of the centre the profiles that are
           * not continuous have no value; nansum skips the pixel (it still counts in P) */
          const double extra = np > 2 + nd + L->nsz ? prm[np - 1] : 0.;
          if (fit == CTR_FIT_INV_SERIES) {
            profile_inv(nx, r2, prm + 2 + nd + L->nsz, &gv, &dg, dgx);
            dge = 0.;
          } else if (fit != CTR_FIT_GAUSS && qraw < 1.) {
            /* (disc_func starts from ones and only overwrites where r2 > disc_size^2, false for a
             *  NaN: the disc has the value 1 there, fitfunc.py:122-131; disc_size <= 0: gauss(NaN)) */
            if (fit == CTR_FIT_DISC && extra > 0.) gv = 1.;
            else { safe_nan = 1; gv = NAN; }
            dg = 0.; dge = 0.;
          } else profile(fit, nd, r2, extra, &gv, &dg, &dge);
          res -= sig * gv;
          if (Q) {
            hf[nhf].i = i;
            for (int a = 0; a < nd; ++a) {
              hf[nhf].J[a] = -sig * dg * dr2[a];       /* d res / d pos_a */
              hf[nhf].E[a] = -0.5 * nd * dr2[a];       /* d (-nd/2 r2) / d pos_a */
            }
            ++nhf;
          }
          if (g) {
            /* d res / d signal, positions, sizes (fitfunc.py:475-478, sign of the residual) */
            double d[CTR_MAX_PARAMS];
            d[0] = -gv;
            for (int t = 0; t < nd + L->nsz; ++t) d[1 + t] = -sig * dg * dr2[t];
            if (fit == CTR_FIT_INV_SERIES)
              for (int t = 0; t < nx; ++t) d[1 + nd + L->nsz + t] = -sig * dgx[t];
            else if (np > 2 + nd + L->nsz) d[1 + nd + L->nsz] = -sig * dge;   /* fitfunc.py:480-481 */
            for (int k = 1; k < np; ++k) {
              int b = L->var_of[k];
              if (b < 0) continue;
              int col = b + (L->per_feat[k] ? i : 0);
              int seen = 0;
              for (int t = 0; t < nnz; ++t) if (nz[t] == col) { seen = 1; break; }
              if (seen) row[col] += d[k - 1];
              else { row[col] = d[k - 1]; nz[nnz++] = col; }
            }
          }
        }
        if (!any) continue;
        ++P;
        if (res != res) { /* nansum (fitfunc.py:449,483): a NaN pixel of the IMAGE is skipped */
          if (pv == pv && !safe_nan) model_nan = 1; /* a NaN of the MODEL (a size of 0): no valid evaluation */
          continue;
        }
        S += res * res;
        if (g) {
          for (int s = 0; s < nnz; ++s) {
            int cs = nz[s];
            double rs = row[cs];
            g[cs] += rs * res;
            for (int t = 0; t < nnz; ++t) A[cs * nv + nz[t]] += rs * row[nz[t]];
          }
        }
        if (Q)
          for (int f = 0; f < nhf; ++f)
            for (int a = 0; a < nd; ++a)
              for (int b2 = a; b2 < nd; ++b2) U[hf[f].i][a][b2] += res * hf[f].J[a] * hf[f].E[b2];
      }
  if (Q) {
    /* sum_p res_p d2res_p/dv dv restricted to (signal, positions) of each feature; all but U
     * follows from the gradient:  d2res/ds dpos_a = (dres/dpos_a)/s,
     * d2res/dpos_a dpos_b = J_a E_b + delta_ab (-nd/size_a^2) s (dres/ds) */
    memset(Q, 0, sizeof(double) * nv * nv);
    for (int i = 0; i < n; ++i) {
      const int cs = L->var_of[1] + i;
      const double sig = v[cs];
      for (int a = 0; a < nd; ++a) {
        const int ca = L->var_of[2 + a] + i;
        const double sz = par(c, v, i, c->p->isotropic ? 2 + nd : 2 + nd + a);
        const double t = sig != 0. ? g[ca] / sig : 0.;
        Q[cs * nv + ca] = Q[ca * nv + cs] = t;
        for (int b2 = a; b2 < nd; ++b2) {
          const int cb = L->var_of[2 + b2] + i;
          double u = U[i][a][b2];
          if (a == b2) u += -(double)nd / (sz * sz) * sig * g[cs];
          Q[ca * nv + cb] = Q[cb * nv + ca] = u;
        }
      }
    }
  }
  *S_out = model_nan ? NAN : S;
  *P_out = P;
  free(row); free(nz); free(U); free(hf); free(box);
}

/* ---- dense helpers -------------------------------------------------------- */

/* in-place lower Cholesky of H[n][n] (row stride ld); 0 on failure */
static int cholesky(double* H, int n, int ld) {
  for (int j = 0; j < n; ++j) {
    double d = H[j * ld + j];
    for (int k = 0; k < j; ++k) d -= H[j * ld + k] * H[j * ld + k];
    if (!(d > 0.) || !isfinite(d)) return 0;
    d = sqrt(d);
    H[j * ld + j] = d;
    for (int i = j + 1; i < n; ++i) {
      double s = H[i * ld + j];
      for (int k = 0; k < j; ++k) s -= H[i * ld + k] * H[j * ld + k];
      H[i * ld + j] = s / d;
    }
  }
  return 1;
}

static void chol_solve(const double* Lm, int n, int ld, double* x) {
  for (int i = 0; i < n; ++i) {
    double s = x[i];
    for (int k = 0; k < i; ++k) s -= Lm[i * ld + k] * x[k];
    x[i] = s / Lm[i * ld + i];
  }
  for (int i = n - 1; i >= 0; --i) {
    double s = x[i];
    for (int k = i + 1; k < n; ++k) s -= Lm[k * ld + i] * x[k];
    x[i] = s / Lm[i * ld + i];
  }
}

/* ---- bounded (+ equality constrained) Levenberg-Marquardt ----------------- */

#define STALL_TOL 1e-9
/* diagnostic switch: 0 = Gauss-Newton model only (for A/B measurements of the oracle itself) */
static int use_newton = 1;
void ctro_set_newton(int on) { use_newton = on; }
/* diagnostic switch: print one line per solver iteration to stderr */
static int trace = 0;
static int fast_exit = 1;
void ctro_set_fast_exit(int on) { fast_exit = on; }
static double mu0_sizevar = 1.;
void ctro_set_mu0_sizevar(double x) { mu0_sizevar = x; }
void ctro_set_trace(int on) { trace = on; }

typedef struct { double S; long P; int iters; int ok; } solve_t;
static void full_second_order(const ctx_t* c, const double* v, double* Qf);

/* Retraction onto the constraint manifold c(x) = 0 (constraints.py:59-137): minimum-norm
 * Newton steps x <- x - C^T (C C^T)^-1 c over the position variables that the box leaves
 * free, clipped to the box.  The constraints are quadratics of the positions, so a point
 * near the manifold needs 3-5 steps.  On success cv / Cj / pair_of describe the constraints
 * at the returned point.  0: no convergence (degenerate geometry, or the box is in the way). */
#define RETRACT_MAXIT 30
#define RETRACT_TOL 1e-13
static int same_pair_set(const int* a, const int* b, int m) {
  unsigned sa = 0, sb = 0;
  for (int r = 0; r < m; ++r) { sa |= 1u << a[r]; sb |= 1u << b[r]; }
  return sa == sb;
}

static int retract(const ctx_t* c, double* x, const double* lo, const double* hi,
                   double* cv, double* Cj, int* pair_of, const int* held) {
  const int nv = c->L.nv, m = c->n_cons;
  const int ranked = c->p->constraint_kind == CTR_CONS_TETRAMER && c->L.nd == 2;
  double G[MAXC * MAXC], y[MAXC];
  int pinned[MAXV], pairs[MAXC];
  eval_constraints(c, x, cv, Cj, pairs, NULL);
  /* 2D tetramer: Newton on the branch of the current pair set; if the ranking at the point
   * reached names another set, once more on that one */
  for (int branch = 0; branch < 4; ++branch) {
    int converged = 0;
    for (int iter = 0; iter <= RETRACT_MAXIT && !converged; ++iter) {
      double cmax = 0.;
      eval_constraints(c, x, cv, Cj, pair_of, pairs);
      for (int r = 0; r < m; ++r) if (fabs(cv[r]) > cmax) cmax = fabs(cv[r]);
      if (!(cmax == cmax)) return 0;
      if (cmax <= RETRACT_TOL) { converged = 1; break; }
      if (iter == RETRACT_MAXIT) break;
      /* pass 0: all variables the box does not fix; a variable ON a bound that this correction
       * would push outward is pinned and the correction recomputed without it (pass 1) */
      for (int i = 0; i < nv; ++i) pinned[i] = !(lo[i] < hi[i]) || (held && held[i]);
      for (int pass = 0; pass < 2; ++pass) {
        double tr = 0.;
        int changed = 0;
        for (int r = 0; r < m; ++r)
          for (int s = 0; s <= r; ++s) {
            double t = 0.;
            for (int i = 0; i < nv; ++i)
              if (!pinned[i]) t += Cj[r * nv + i] * Cj[s * nv + i];
            G[r * m + s] = t;
            if (r == s) tr += t;
          }
        if (!(tr > 0.)) return 0;
        for (int r = 0; r < m; ++r) { G[r * m + r] += 1e-14 * tr + 1e-300; y[r] = cv[r]; }
        if (!cholesky(G, m, m)) return 0;
        chol_solve(G, m, m, y);
        if (pass == 1) break;
        for (int i = 0; i < nv; ++i) {
          if (pinned[i]) continue;
          double t = 0.;
          for (int r = 0; r < m; ++r) t += Cj[r * nv + i] * y[r];
          if ((x[i] <= lo[i] && t > 0.) || (x[i] >= hi[i] && t < 0.)) { pinned[i] = 1; changed = 1; }
        }
        if (!changed) break;
      }
      for (int i = 0; i < nv; ++i) {
        if (pinned[i]) continue;
        double t = 0.;
        for (int r = 0; r < m; ++r) t += Cj[r * nv + i] * y[r];
        if (t == 0.) continue;
        t = x[i] - t;
        x[i] = t < lo[i] ? lo[i] : (t > hi[i] ? hi[i] : t);
      }
    }
    if (!converged) return 0;
    if (!ranked) return 1;
    eval_constraints(c, x, cv, Cj, pair_of, NULL);
    if (same_pair_set(pair_of, pairs, m)) return 1;
    memcpy(pairs, pair_of, sizeof(int) * (size_t)m);
  }
  return 0;
}

/* ---- the step of a constrained fit: a bound-constrained QP on the tangent space -------------
 *   minimise  g.d + 1/2 d.H.d   subject to   C d = 0,   lo - v <= d <= hi - v
 * (H = model Hessian + Marquardt diagonal, C = Jacobian of the equality constraints at the
 * feasible iterate v) by a primal active-set iteration: what the reference's SLSQP solves in
 * every major iteration (scipy slsqp_optmz.f: LSQ with the linearised constraints and the box).
 * d = 0 is feasible; every step of the iteration lowers the model, so any iterate is a descent
 * step of it.  held[i]: -1 / +1 = variable i is kept on its lower / upper bound (in: the start
 * guess, out: the final working set), 0 = free.  The equality-constrained sub-problems on the free variables are solved in
 * range-space form with H_FF + rho C_F^T C_F in place of H_FF: the same minimiser on C p = 0,
 * and positive definite whenever the REDUCED Hessian is (the Lagrangian Hessian itself is
 * indefinite at most constrained minima: Finsler / Debreu).
 * Returns 0 when a sub-problem has no positive definite matrix (caller: next model). */
#define QP_MAXIT(nv) (3 * (nv) + 8)
static double qp_rho = 1e3;
static int cons_use_qp = 1, lam_passes = 2, full_q_mode = 1;
void ctro_set_full_q(int mode) { full_q_mode = mode; }
void ctro_set_cons_qp(int on, int passes) { cons_use_qp = on; lam_passes = passes; }
void ctro_set_qp_rho(double x) { qp_rho = x; }
static int cons_qp(int nv, int m, const double* Hm, const double* g, const double* Cj,
                   const double* v, const double* lo, const double* hi, int* held,
                   double* d, double* mult) {
  double* H = malloc(sizeof(double) * nv * nv);
  double* gq = malloc(sizeof(double) * nv), *w = malloc(sizeof(double) * nv);
  double* p = malloc(sizeof(double) * nv), *Y = malloc(sizeof(double) * nv * MAXC);
  int* fr = malloc(sizeof(int) * nv);
  double Sc[MAXC * MAXC];
  int ok = 1, released = 0;
  memset(d, 0, sizeof(double) * nv);
  memset(mult, 0, sizeof(double) * MAXC);
  for (int qit = 0; qit < QP_MAXIT(nv); ++qit) {
    int nf = 0;
    double hmax = 0., cmax = 0.;
    for (int i = 0; i < nv; ++i) {
      double t = g[i];
      for (int j = 0; j < nv; ++j) t += Hm[i * nv + j] * d[j];
      gq[i] = t;
      if (!held[i]) fr[nf++] = i;
    }
    if (nf == 0) break;
    for (int a = 0; a < nf; ++a) if (Hm[fr[a] * nv + fr[a]] > hmax) hmax = Hm[fr[a] * nv + fr[a]];
    for (int r = 0; r < m; ++r) {
      double t = 0.;
      for (int a = 0; a < nf; ++a) t += Cj[r * nv + fr[a]] * Cj[r * nv + fr[a]];
      if (t > cmax) cmax = t;
    }
    const double rho = cmax > 0. ? qp_rho * hmax / cmax : 0.;
    for (int a = 0; a < nf; ++a)
      for (int b = 0; b <= a; ++b) {
        double t = Hm[fr[a] * nv + fr[b]];
        for (int r = 0; r < m; ++r) t += rho * Cj[r * nv + fr[a]] * Cj[r * nv + fr[b]];
        H[a * nf + b] = t;
      }
    if (!cholesky(H, nf, nf)) { ok = 0; break; }
    for (int a = 0; a < nf; ++a) w[a] = gq[fr[a]];
    chol_solve(H, nf, nf, w);
    for (int r = 0; r < m; ++r) {
      for (int a = 0; a < nf; ++a) Y[r * nf + a] = Cj[r * nv + fr[a]];
      chol_solve(H, nf, nf, Y + r * nf);
    }
    {
      double tr = 0.;
      for (int r = 0; r < m; ++r) {
        for (int s = 0; s <= r; ++s) {
          double t = 0.;
          for (int a = 0; a < nf; ++a) t += Cj[r * nv + fr[a]] * Y[s * nf + a];
          Sc[r * m + s] = t;
        }
        double t = 0.;
        for (int a = 0; a < nf; ++a) t -= Cj[r * nv + fr[a]] * w[a];
        mult[r] = t;
        tr += Sc[r * m + r];
      }
      for (int r = 0; r < m; ++r) Sc[r * m + r] += 1e-14 * tr + 1e-300;
      if (!(tr > 0.) || !cholesky(Sc, m, m)) memset(mult, 0, sizeof(double) * MAXC);
      else chol_solve(Sc, m, m, mult);
    }
    /* p = -H'^-1 (gq + C^T mult); the ratio test along p: the first variable to reach a bound */
    double alpha = 1., pmax = 0.;
    int blocking = -1;
    for (int a = 0; a < nf; ++a) {
      const int i = fr[a];
      double t = w[a];
      for (int r = 0; r < m; ++r) t += Y[r * nf + a] * mult[r];
      p[a] = -t;
      const double room = p[a] < 0. ? (lo[i] - v[i]) - d[i] : (hi[i] - v[i]) - d[i];
      const double rel = fabs(p[a]) / (fabs(v[i]) + 1.);
      if (rel > pmax) pmax = rel;
      if (p[a] != 0.) {
        const double ratio = room / p[a];
        if (ratio < alpha) { alpha = ratio; blocking = a; }
      }
    }
    if (alpha < 0.) alpha = 0.;
    if (pmax > 1e-15) {
      for (int a = 0; a < nf; ++a) d[fr[a]] += alpha * p[a];
      if (blocking >= 0) {
        const int i = fr[blocking];
        held[i] = p[blocking] < 0. ? -1 : 1;
        d[i] = (held[i] < 0 ? lo[i] : hi[i]) - v[i];
        continue;
      }
      for (int i = 0; i < nv; ++i) {   /* gradient of the model at the new d */
        double t = g[i];
        for (int j = 0; j < nv; ++j) t += Hm[i * nv + j] * d[j];
        gq[i] = t;
      }
    }
    /* a minimiser on this working set: does a held variable want to leave its bound?  (mult: the
     * multipliers of the equality constraints at this point -- C p = 0, so the rho term is 0) */
    {
      int worst = -1;
      double wv = 0.;
      for (int i = 0; i < nv; ++i) {
        if (!held[i] || !(lo[i] < hi[i])) continue;
        double s = gq[i], sc = fabs(gq[i]);
        for (int r = 0; r < m; ++r) { s += Cj[r * nv + i] * mult[r]; sc += fabs(Cj[r * nv + i] * mult[r]); }
        const double viol = held[i] < 0 ? -s : s;   /* > 0: the model falls when the variable moves inward */
        if (viol > 1e-10 * sc + 1e-300 && viol > wv) { wv = viol; worst = i; }
      }
      if (worst < 0 || released >= nv) break;
      held[worst] = 0;
      ++released;
    }
  }
  free(H); free(gq); free(w); free(p); free(Y); free(fr);
  return ok;
}

/* One solver run (one re-window round).  Unconstrained clusters: bounded Levenberg-Marquardt
 * with an active set for the box.  Constrained clusters (dimer / trimer / tetramer): the same
 * iteration as a FEASIBLE-POINT method -- every iterate lies on the constraint manifold: the
 * step is the minimiser of the damped model in the tangent space (C d = 0, range-space form),
 * the trial point is retracted onto the manifold, and a step is judged by the objective alone.
 * No merit function and no penalty weight, hence no Maratos effect. */
static solve_t solve(const ctx_t* c, const double* v0, const double* lo, const double* hi,
                     double* v /* out */) {
  const int nv = c->L.nv, m = c->n_cons;
  const int maxiter = c->p->solver_maxiter > 0 ? c->p->solver_maxiter : 100;
  const double xtol = c->p->xtol > 0 ? c->p->xtol : 1e-9;
  const double ftol = c->p->ftol > 0 ? c->p->ftol : 1e-14;
  solve_t out = {NAN, 0, 0, 0};
  double *g = malloc(sizeof(double) * nv), *A = malloc(sizeof(double) * nv * nv);
  double *gt = malloc(sizeof(double) * nv), *At = malloc(sizeof(double) * nv * nv);
  double *Q = calloc((size_t)nv * nv, sizeof(double)), *Qt = calloc((size_t)nv * nv, sizeof(double));
  double *B = malloc(sizeof(double) * nv * nv); /* model Hessian: A + Q + sum mult d2c */
  int pair_of[MAXC], pair_of_t[MAXC];
  /* exact second-order terms are used when signal and positions are per-feature variables
   * (the default modes); otherwise the model Hessian is J^T J throughout */
  /* (the second-order terms are those of the gaussian: other profiles iterate with J^T J) */
  int newton = use_newton && c->p->fit_function == CTR_FIT_GAUSS && c->p->modes[1] == CTR_MODE_VAR;
  for (int a = 0; a < c->L.nd; ++a) newton = newton && c->p->modes[2 + a] == CTR_MODE_VAR;
  /* fullq: the second-order part in ALL variables (full_second_order) instead */
  int fullq = 0;
  if (use_newton && c->p->fit_function == CTR_FIT_GAUSS && full_q_mode && (full_q_mode == 2 || m)) {
    int sizevar = 0;
    for (int k2 = 2 + c->L.nd; k2 < c->L.np; ++k2) if (c->L.var_of[k2] >= 0) sizevar = 1;
    if (!newton || sizevar) { fullq = 1; newton = 1; }
  }
  double *H = malloc(sizeof(double) * nv * nv), *vt = malloc(sizeof(double) * nv);
  double *dl = malloc(sizeof(double) * nv), *w = malloc(sizeof(double) * nv);
  double *Y = malloc(sizeof(double) * nv * MAXC), *Cj = malloc(sizeof(double) * nv * MAXC);
  double *Cjt = malloc(sizeof(double) * nv * MAXC);
  int* fr = malloc(sizeof(int) * nv);
  int *inset = malloc(sizeof(int) * nv), *isfree = malloc(sizeof(int) * nv), *held = malloc(sizeof(int) * nv);
  double *Hm = m ? malloc(sizeof(double) * nv * nv) : NULL;
  double cv[MAXC], cvt[MAXC], mult[MAXC], lam[MAXC], lam_first[MAXC], Sc[MAXC * MAXC];
  double S, St, mu, nu = 2.;
  long P;
  int last_accepted = 1, it = 0;
  double prev_step = INFINITY; /* relative size of the last accepted step */
  double gain = INFINITY; /* relative decrease of the objective by the last accepted step */

  for (int i = 0; i < nv; ++i) {
    if (lo[i] > hi[i]) goto done; /* infeasible box (SciPy raises ValueError) */
    v[i] = v0[i] < lo[i] ? lo[i] : (v0[i] > hi[i] ? hi[i] : v0[i]);
  }
  memset(mult, 0, sizeof mult);
  memset(lam, 0, sizeof lam);
  memset(cv, 0, sizeof cv);
  /* the start vector need not satisfy the constraints: restore feasibility first */
  if (m && !retract(c, v, lo, hi, cv, Cj, pair_of, NULL)) {
    eval_cluster(c, v, &S, NULL, NULL, NULL, &P);
    out.P = P;
    goto done;
  }
  eval_cluster(c, v, &S, g, A, newton && !fullq ? Q : NULL, &P);
  if (fullq) full_second_order(c, v, Q);
  out.P = P;
  if (P == 0 || !isfinite(S)) goto done;
  /* multiplies the Marquardt diagonal below.  With a size among the variables the first steps
   * are damped more: a nearly undamped first step from a start 20 % off in size and half a size
   * off in position can land on a spike (size on its lower bound) that lowers the objective a
   * little and is never left again (reference tests/test_refine.py:622-631, 3D, found by
   * tests/test_gpu_accuracy_matrix.py) */
  mu = 1e-3;
  for (int k2 = 2 + c->L.nd; k2 < c->L.np; ++k2)
    if (c->L.var_of[k2] >= 0) mu = mu0_sizevar;

  for (it = 0; it < maxiter; ++it) {
    int nf = 0;
    out.iters = it + 1;
    /* least-squares multipliers lam = -(C C^T)^-1 C g over the variables the box does not pin:
     * the ones that go with the minimum-norm retraction (g . n = lam^T C n for its normal
     * correction n), hence the ones whose curvature term makes the tangent-space model agree
     * with the objective along the retracted step to second order */
    /* two passes: the multipliers over all variables the box does not fix give a first active
     * set; the multipliers over the variables THAT leaves free (the ones a retraction with the
     * held variables pinned goes with) give the active set used */
    for (int pass = 0; pass < (m ? lam_passes : 1); ++pass) {
      if (m) {
        double tr = 0.;
        for (int i = 0; i < nv; ++i) inset[i] = pass == 0 ? lo[i] < hi[i] : isfree[i];
        for (int r = 0; r < m; ++r) {
          for (int s = 0; s <= r; ++s) {
            double t = 0.;
            for (int i = 0; i < nv; ++i) if (inset[i]) t += Cj[r * nv + i] * Cj[s * nv + i];
            Sc[r * m + s] = t;
          }
          double t = 0.;
          for (int i = 0; i < nv; ++i) if (inset[i]) t -= Cj[r * nv + i] * g[i];
          lam[r] = t;
          tr += Sc[r * m + r];
        }
        for (int r = 0; r < m; ++r) Sc[r * m + r] += 1e-14 * tr + 1e-300;
        if (tr > 0. && cholesky(Sc, m, m)) chol_solve(Sc, m, m, lam);
        else if (pass == 0) memset(lam, 0, sizeof lam);
        else memcpy(lam, lam_first, sizeof lam);
        if (pass == 0) memcpy(lam_first, lam, sizeof lam);
      }
      /* active set: fixed if at a bound and the Lagrangian gradient pushes outward */
      nf = 0;
      for (int i = 0; i < nv; ++i) {
        double gl = g[i];
        for (int r = 0; r < m; ++r) gl += Cj[r * nv + i] * lam[r];
        int fixed = (lo[i] == hi[i]) || (v[i] <= lo[i] && gl > 0.) || (v[i] >= hi[i] && gl < 0.);
        isfree[i] = !fixed;
        if (!fixed) fr[nf++] = i;
      }
    }
    if (nf == 0) { out.ok = 1; break; }
    /* Model Hessian.  First choice: the exact one, J^T J + sum_p r_p d2r_p + sum_r mult_r d2c_r
     * (quadratic convergence also where the residual stays large: overlapping features,
     * constrained fits).  Where that is not positive definite on the free variables, or its
     * projected step is not a descent step of the model, the Gauss-Newton matrix J^T J is
     * used for this iteration instead. */
    double stepmax = 0., pred = 0.;
    int step_ok = 0;
    /* models tried in turn: with the second-order part Q of the residuals (and the curvature
     * of the constraints), without Q, and -- constrained fits -- plain J^T J */
    for (int attempt = (newton ? 1 : 0) + (m ? 1 : 0); attempt >= 0 && !step_ok; --attempt) {
      const int use_q = newton && attempt == (m ? 2 : 1), use_cc = m && attempt >= 1;
      memcpy(B, A, sizeof(double) * nv * nv);
      if (use_q)
        for (int i = 0; i < nv * nv; ++i) B[i] += Q[i];
      if (use_cc) {
        /* curvature of the constraints (c_r = 1 - |dp/dist|^2) */
        for (int r = 0; r < m; ++r) {
          int i0 = PAIRS[pair_of[r]][0], i1 = PAIRS[pair_of[r]][1];
          for (int a = 0; a < c->L.nd; ++a) {
            int k = 2 + a, b = c->L.var_of[k];
            if (b < 0 || !c->L.per_feat[k]) continue;
            double da = c->p->constraint_dist[a], t = -2. * lam[r] / (da * da);
            B[(b + i0) * nv + b + i0] += t;
            B[(b + i1) * nv + b + i1] += t;
            B[(b + i0) * nv + b + i1] -= t;
            B[(b + i1) * nv + b + i0] -= t;
          }
        }
      }
      int held_ok = 0;
      if (m && cons_use_qp) {
        /* constrained fit: the bound-constrained QP on the tangent space (cons_qp) */
        memcpy(Hm, B, sizeof(double) * nv * nv);
        for (int i = 0; i < nv; ++i) {
          double d = A[i * nv + i];
          Hm[i * nv + i] += mu * (d > 1e-300 ? d : 1.);
          held[i] = isfree[i] ? 0 : (v[i] <= lo[i] ? -1 : 1);
        }
        if (!cons_qp(nv, m, Hm, g, Cj, v, lo, hi, held, dl, mult)) { if (trace > 2) fprintf(stderr, "   it %d attempt %d: QP matrix not PD (mu %g)\n", it, attempt, mu); continue; }
        held_ok = 1;
        goto have_step;
      }
      for (int a = 0; a < nf; ++a) {
        for (int b = 0; b <= a; ++b) H[a * nf + b] = B[fr[a] * nv + fr[b]];
        double d = A[fr[a] * nv + fr[a]];
        H[a * nf + a] += mu * (d > 1e-300 ? d : 1.);
      }
      if (!cholesky(H, nf, nf)) { if (trace > 2) fprintf(stderr, "   it %d attempt %d: H not PD (mu %g)\n", it, attempt, mu); continue; }
      for (int a = 0; a < nf; ++a) w[a] = g[fr[a]];
      chol_solve(H, nf, nf, w); /* w = H^-1 g_F */
      memset(dl, 0, sizeof(double) * nv);
      if (m == 0) {
        for (int a = 0; a < nf; ++a) dl[fr[a]] = -w[a];
      } else {
        /* tangent step, range-space form: (C H^-1 C^T) mult = -C H^-1 g ;
         * d = -H^-1 (g + C^T mult), so that C d = 0 */
        for (int r = 0; r < m; ++r) {
          for (int a = 0; a < nf; ++a) Y[r * nf + a] = Cj[r * nv + fr[a]];
          chol_solve(H, nf, nf, Y + r * nf);
        }
        for (int r = 0; r < m; ++r) {
          for (int s = 0; s <= r; ++s) {
            double t = 0.;
            for (int a = 0; a < nf; ++a) t += Cj[r * nv + fr[a]] * Y[s * nf + a];
            Sc[r * m + s] = t;
          }
          double t = 0.;
          for (int a = 0; a < nf; ++a) t -= Cj[r * nv + fr[a]] * w[a];
          mult[r] = t;
        }
        {
          double tr = 0.;
          for (int r = 0; r < m; ++r) tr += Sc[r * m + r];
          for (int r = 0; r < m; ++r) Sc[r * m + r] += 1e-14 * tr + 1e-300;
          if (!(tr > 0.)) { memset(mult, 0, sizeof mult); continue; }
        }
        if (!cholesky(Sc, m, m)) { memset(mult, 0, sizeof mult); continue; }
        chol_solve(Sc, m, m, mult);
        for (int a = 0; a < nf; ++a) {
          double t = w[a];
          for (int r = 0; r < m; ++r) t += Y[r * nf + a] * mult[r];
          dl[fr[a]] = -t;
        }
      }
    have_step:
      /* projected trial point, retracted onto the constraint manifold (first with the variables
       * the QP holds on their bounds pinned, so that they stay there) */
      for (int i = 0; i < nv; ++i) {
        double t = v[i] + dl[i];
        vt[i] = t < lo[i] ? lo[i] : (t > hi[i] ? hi[i] : t);
        if (held_ok && held[i] && lo[i] < hi[i]) vt[i] = held[i] < 0 ? lo[i] : hi[i];
      }
      if (m && held_ok) {
        memcpy(w, vt, sizeof(double) * nv);
        if (retract(c, vt, lo, hi, cvt, Cjt, pair_of_t, held)) goto retracted;
        memcpy(vt, w, sizeof(double) * nv);
      }
      if (m && !retract(c, vt, lo, hi, cvt, Cjt, pair_of_t, NULL)) { if (trace > 2) fprintf(stderr, "   it %d attempt %d: retraction failed\n", it, attempt); continue; }
    retracted:;
      double gd = 0., dAd = 0.;
      stepmax = 0.;
      for (int i = 0; i < nv; ++i) {
        dl[i] = vt[i] - v[i];
        double rel = fabs(dl[i]) / (fabs(v[i]) + 1.);
        if (rel > stepmax) stepmax = rel;
      }
      /* predicted decrease of the OBJECTIVE along the actual (retracted) step; with
       * constraints its model is A (+ Q): the curvature of the manifold is in dl itself */
      for (int i = 0; i < nv; ++i) {
        double t = 0.;
        for (int j = 0; j < nv; ++j)
          t += (m ? A[i * nv + j] + (use_q ? Q[i * nv + j] : 0.) : B[i * nv + j]) * dl[j];
        dAd += dl[i] * t;
        gd += g[i] * dl[i];
      }
      pred = -(gd + 0.5 * dAd);
      if (attempt >= 1 && !(pred > -(ftol * (0.5 * S) + 1e-300))) { if (trace > 2) fprintf(stderr, "   it %d attempt %d: pred %g\n", it, attempt, pred); continue; }
      step_ok = 1;
    }
    if (!step_ok) { mu *= nu; nu *= 2.; last_accepted = 0; if (mu > 1e30) break; continue; }
    /* converged: negligible step after an accepted one, or negligible model change.
     * A clearly NEGATIVE predicted decrease (the projection onto the box turned the
     * step uphill) is not convergence: the trial below is then rejected and mu grows,
     * which turns the step towards the projected gradient. */
    if ((last_accepted && stepmax <= xtol) || fabs(pred) <= ftol * (0.5 * S) + 1e-300) {
      out.ok = 1;
      break;
    }
    /* Converging faster than linearly at rate r = step / previous step: the step after this
     * one would be <= r * step.  Where that is below xtol the fit is finished by TAKING this
     * step, without the pixel pass that would only confirm it (the point returned is then as
     * exact as the one the plain test returns an iteration later; S from the model). */
    if (fast_exit && last_accepted && pred > 0. && stepmax < prev_step && isfinite(prev_step) &&
        stepmax * (stepmax / prev_step) <= xtol) {
      memcpy(v, vt, sizeof(double) * nv);
      S -= 2. * pred;
      if (S < 0.) S = 0.;
      out.ok = 1;
      break;
    }
    if (trace > 2 && !(pred > 0.)) fprintf(stderr, "   it %d: pred %g, no trial\n", it, pred);
    if (!(pred > 0.)) {
      /* the model itself predicts no decrease (projection): rejected without looking at the
       * pixels -- the test below would reject it whatever the trial gave */
      mu *= nu;
      nu *= 2.;
      last_accepted = 0;
      if (mu > 1e30) break;
      continue;
    }
    eval_cluster(c, vt, &St, gt, At, newton && !fullq ? Qt : NULL, &P);
    if (fullq && isfinite(St)) full_second_order(c, vt, Qt);
    double act = 0.5 * (S - St);
    if (trace)
      fprintf(stderr, "it %3d nf %d S %.10g St %.10g mu %.2e pred %.3e act %.3e step %.2e mult0 %.3e\n",
              it, nf, S, St, mu, pred, act, stepmax, m ? mult[0] : 0.);
    if (trace > 1) {
      fprintf(stderr, "      vt:");
      for (int i = 0; i < nv; ++i) fprintf(stderr, " %.6g", vt[i]);
      fprintf(stderr, "\n      g :");
      for (int i = 0; i < nv; ++i) fprintf(stderr, " %.3g", g[i]);
      fprintf(stderr, "\n");
    }
    if (isfinite(St) && pred > 0. && act > 0.) {
      double rho = act / pred, t = 2. * rho - 1.;
      double f = 1. - t * t * t;
      mu *= f > 1. / 3. ? f : 1. / 3.;
      nu = 2.;
      gain = act / (0.5 * S + 1e-300);
      memcpy(v, vt, sizeof(double) * nv);
      memcpy(g, gt, sizeof(double) * nv);
      memcpy(A, At, sizeof(double) * nv * nv);
      memcpy(Q, Qt, sizeof(double) * nv * nv);
      memcpy(pair_of, pair_of_t, sizeof pair_of);
      memcpy(cv, cvt, sizeof(double) * (m ? m : 1));
      memcpy(Cj, Cjt, sizeof(double) * (size_t)(m ? m : 0) * nv);
      S = St;
      last_accepted = 1;
      prev_step = stepmax;
    } else {
      mu *= nu;
      nu *= 2.;
      last_accepted = 0;
      if (mu > 1e30) break;
    }
  }
  /* iteration limit with a stationary objective: the last accepted step lowered it by
   * less than STALL_TOL (relative).  The reference's SLSQP stops at |dF| < 1e-6 absolute
   * (refine.py:243,373-375), far looser, so it reports such fits as converged. */
  if (!out.ok && it == maxiter && gain <= STALL_TOL) out.ok = 1;
  out.S = S;
done:
  free(g); free(A); free(gt); free(At); free(H); free(vt); free(dl); free(w);
  free(Y); free(Cj); free(Cjt); free(fr); free(Q); free(Qt); free(B);
  free(inset); free(isfree); free(held); free(Hm);
  return out;
}

/* sum_p res_p d2res_p/dv dv over ALL variables (the part of the Hessian of S/2 that J^T J
 * lacks), for any parameter modes.  Per feature the second derivatives of its model
 * m = s g, g = exp(E), E = -(nd/2) sum_a (x_a - c_a)^2 / size_a^2 (fitfunc.py:112-118) are summed
 * in parameter space (signal, centres, sizes):
 *   m_s,t = g E_t      m_t,u = s g (E_t E_u + E_tu)      m_s,s = 0
 * and then added to the rows/columns of the variables these parameters map to
 * (vect_to_params, fitfunc.py:266-315: a shared variable collects every feature's block).
 * Qf[nv*nv].  Used for compute_error only (solution_std); the solver keeps its own Q. */
static void full_second_order(const ctx_t* c, const double* v, double* Qf) {
  const layout_t* L = &c->L;
  const int nd = L->nd, nv = L->nv, n = L->n, np = L->np;
  const int iso = c->p->isotropic;
  const int nsz = L->nsz, pw = 1 + nd + nsz;
  const int64_t* fshape = c->b->shape;
  const int dtype = c->b->frame_dtype;
  const double bg = par(c, v, 0, 0);
  const int w0 = nd == 3 ? c->wshape[0] : 1, w1 = c->wshape[nd - 2], w2 = c->wshape[nd - 1];
  double (*W)[7][7] = calloc((size_t)(n > 0 ? n : 1), sizeof(double) * 49);
  struct cf_t { int i; double h[7][7]; }* cf = malloc(sizeof(struct cf_t) * (size_t)(n > 0 ? n : 1));
  memset(Qf, 0, sizeof(double) * nv * nv);
  for (int z = 0; z < w0; ++z)
    for (int y = 0; y < w1; ++y)
      for (int x = 0; x < w2; ++x) {
        int idx[3], ncf = 0;
        double mesh[3], model = 0.;
        size_t off;
        if (nd == 3) {
          idx[0] = z; idx[1] = y; idx[2] = x;
          off = ((size_t)(z + c->origin[0]) * fshape[1] + (y + c->origin[1])) * fshape[2] + (x + c->origin[2]);
        } else {
          idx[0] = y; idx[1] = x;
          off = (size_t)(y + c->origin[0]) * fshape[1] + (x + c->origin[1]);
        }
        for (int a = 0; a < nd; ++a) mesh[a] = (double)(idx[a] + c->origin[a]);
        for (int i = 0; i < n; ++i) {
          double E1[7], dd[3], i2[3], isz[3], q = 0., r2 = 0.;
          double (*h)[7];
          if (!in_mask(nd, idx, c->mcoords + i * nd, c->origin, c->p->radius)) continue;
          const double sig = par(c, v, i, 1);
          for (int a = 0; a < nd; ++a) {
            const double sz = par(c, v, i, iso ? 2 + nd : 2 + nd + a);
            dd[a] = mesh[a] - par(c, v, i, 2 + a);
            isz[a] = 1. / sz;
            i2[a] = isz[a] * isz[a];
            q += dd[a] * dd[a];
            r2 += dd[a] * dd[a] * i2[a];
          }
          const double G = exp(-0.5 * nd * r2);
          model += sig * G;
          cf[ncf].i = i;
          h = cf[ncf].h;
          ++ncf;
          memset(h, 0, sizeof(double) * 49);
          /* first derivatives of E: [0] unused (signal), [1+a] centre a, [1+nd+..] sizes */
          E1[0] = 0.;
          for (int a = 0; a < nd; ++a) E1[1 + a] = nd * dd[a] * i2[a];
          if (iso) E1[1 + nd] = nd * q * i2[0] * isz[0];
          else for (int a = 0; a < nd; ++a) E1[1 + nd + a] = nd * dd[a] * dd[a] * i2[a] * isz[a];
          for (int t = 1; t < pw; ++t) {
            h[0][t] = h[t][0] = G * E1[t];
            for (int u = 1; u < pw; ++u) h[t][u] = sig * G * (E1[t] * E1[u]);
          }
          /* second derivatives of E */
          for (int a = 0; a < nd; ++a) {
            h[1 + a][1 + a] += sig * G * (-(double)nd * i2[a]);
            if (iso) {
              const double e = -2. * nd * dd[a] * i2[0] * isz[0];
              h[1 + a][1 + nd] += sig * G * e;
              h[1 + nd][1 + a] += sig * G * e;
            } else {
              const double e = -2. * nd * dd[a] * i2[a] * isz[a];
              h[1 + a][1 + nd + a] += sig * G * e;
              h[1 + nd + a][1 + a] += sig * G * e;
              h[1 + nd + a][1 + nd + a] += sig * G * (-3. * nd * dd[a] * dd[a] * i2[a] * i2[a]);
            }
          }
          if (iso) h[1 + nd][1 + nd] += sig * G * (-3. * nd * q * i2[0] * i2[0]);
        }
        if (!ncf) continue;
        const double res = (c->fwin ? c->fwin[((size_t)z * w1 + y) * w2 + x] : pixel(c->frame, dtype, off)) - bg - model;
        if (res != res) continue; /* nansum */
        for (int f = 0; f < ncf; ++f)
          for (int t = 0; t < pw; ++t)
            for (int u = 0; u < pw; ++u) W[cf[f].i][t][u] -= res * cf[f].h[t][u]; /* d2res = -d2m */
      }
  for (int i = 0; i < n; ++i)
    for (int t = 0; t < pw; ++t) {
      const int bt = L->var_of[1 + t];
      if (bt < 0) continue;
      const int ct = bt + (L->per_feat[1 + t] ? i : 0);
      for (int u = 0; u < pw; ++u) {
        const int bu = L->var_of[1 + u];
        if (bu < 0) continue;
        Qf[ct * nv + bu + (L->per_feat[1 + u] ? i : 0)] += W[i][t][u];
      }
    }
  (void)np;
  free(W); free(cf);
}

/* refine.py:400-406: std = sqrt(2 diag(inv(Hessian of F))) at the solution, all variables
 * free.  Hessian of F = 2 (J^T J + Q) / (P norm), so std_i = sqrt(P norm [(J^T J + Q)^-1]_ii).
 * Returns 0 (and NaNs) when the matrix is not positive definite. */
static int solution_std(const ctx_t* c, const double* v, double norm, double* std) {
  const int nv = c->L.nv;
  double *g = malloc(sizeof(double) * nv), *A = malloc(sizeof(double) * nv * nv);
  double *Q = calloc((size_t)nv * nv, sizeof(double)), *e = malloc(sizeof(double) * nv);
  double S;
  long P;
  int ok;
  eval_cluster(c, v, &S, g, A, NULL, &P);
  full_second_order(c, v, Q);
  for (int i = 0; i < nv * nv; ++i) A[i] += Q[i];
  ok = cholesky(A, nv, nv);
  for (int i = 0; i < nv; ++i) {
    std[i] = NAN;
    if (!ok) continue;
    memset(e, 0, sizeof(double) * nv);
    e[i] = 1.;
    chol_solve(A, nv, nv, e);
    std[i] = sqrt((double)P * norm * e[i]);
  }
  free(g); free(A); free(Q); free(e);
  return ok;
}

/* ---- one cluster (refine.py:343-430) -------------------------------------- */

static void pack_start(const ctx_t* c, const double* params, const double* low,
                       const double* high, double* v0, double* lo, double* hi) {
  const layout_t* L = &c->L;
  for (int k = 0; k < L->np; ++k) {
    int b = L->var_of[k];
    if (b < 0) continue;
    if (L->per_feat[k]) {
      for (int i = 0; i < L->n; ++i) {
        v0[b + i] = params[i * L->np + k];
        lo[b + i] = low[i * L->np + k];
        hi[b + i] = high[i * L->np + k];
      }
    } else {
      /* mean start (refine.py:361), loosest bound (fitfunc.py:554-557) */
      double s = 0., l = INFINITY, h = -INFINITY;
      for (int i = 0; i < L->n; ++i) {
        s += params[i * L->np + k];
        if (low[i * L->np + k] < l) l = low[i * L->np + k];
        if (high[i * L->np + k] > h) h = high[i * L->np + k];
      }
      v0[b] = s / L->n;
      lo[b] = l;
      hi[b] = h;
    }
  }
}

static void refine_one(const ctr_problem* p, const ctr_batch* b, int64_t cl,
                       const double* fmax) {
  const int32_t f0 = b->feat_offset[cl], f1 = b->feat_offset[cl + 1];
  const int n = f1 - f0, np = p->n_params, nd = p->ndim;
  const double* params = b->params + (size_t)f0 * np;
  double* pout = b->params_out + (size_t)f0 * np;
  ctx_t c;
  size_t felems = 1;
  memcpy(pout, params, sizeof(double) * (size_t)n * np);
  if (b->params_std)
    for (int i = 0; i < n * np; ++i) b->params_std[(size_t)f0 * np + i] = NAN;
  b->cost[cl] = NAN;
  b->n_rounds[cl] = 0;
  b->n_iter[cl] = 0;
  c.p = p; c.b = b; c.fwin = NULL;
  if (n <= 0) { b->status[cl] = CTR_STATUS_OUT_OF_BOUNDS; return; }
  make_layout(p, n, &c.L);   /* (the oracle has no size limit; constraints only for n <= 4) */
  for (int i = 0; i < n * np; ++i)
    if (!isfinite(params[i])) { b->status[cl] = CTR_STATUS_NONFINITE; return; } /* refine.py:356-357 */
  for (int a = 0; a < nd; ++a) felems *= (size_t)b->shape[a];
  c.frame = (const char*)b->frames + (size_t)b->frame_index[cl] * felems * dtype_size(b->frame_dtype);
  c.n_cons = n_constraints(p, n);
  {
    const int nv = c.L.nv;
    double* vecs = malloc(sizeof(double) * 5 * (size_t)(nv > 0 ? nv : 1));
    double *v0 = vecs, *lo = vecs + nv, *hi = vecs + 2 * nv, *v = vecs + 3 * nv, *vstd = vecs + 4 * nv;
    double* cur = malloc(sizeof(double) * (size_t)n * np);   /* params after the latest round */
    double* coords = malloc(sizeof(double) * (size_t)n * nd);
    const double fm = fmax[b->frame_index[cl]];
    const double norm = fm * fm / p->residual_factor;          /* refine.py:354 */
    double rms = NAN;
    int status = CTR_STATUS_OK;
    memcpy(cur, params, sizeof(double) * (size_t)n * np);
    for (int i = 0; i < n; ++i)
      for (int a = 0; a < nd; ++a) coords[i * nd + a] = params[i * np + 2 + a];
    c.pconst = cur;
    pack_start(&c, params, b->low + (size_t)f0 * np, b->high + (size_t)f0 * np, v0, lo, hi);
    for (int round = 0; round < p->max_iter; ++round) {
      b->n_rounds[cl] = round + 1;
      if (!window(nd, b->shape, p->radius, coords, n, c.origin, c.wshape)) {
        status = CTR_STATUS_OUT_OF_BOUNDS;                      /* refine.py:33-34 */
        break;
      }
      c.mcoords = coords;
      make_fwin(&c);                                            /* refine.py:37-40 */
      solve_t r = solve(&c, v0, lo, hi, v);
      b->n_iter[cl] += r.iters;
      if (r.P == 0) { status = CTR_STATUS_OUT_OF_BOUNDS; break; }
      if (!r.ok) { status = CTR_STATUS_NO_CONVERGENCE; break; } /* refine.py:376-377 */
      rms = sqrt(((r.S / (double)r.P) / norm) / p->residual_factor); /* refine.py:379 */
      if (b->params_std) { /* refine.py:400-406, this round's masks; gaussian only (full_second_order) */
        if (p->fit_function == CTR_FIT_GAUSS) solution_std(&c, v, norm, vstd);
        else for (int i = 0; i < nv; ++i) vstd[i] = NAN;
      }
      int moved = 0;
      for (int i = 0; i < n; ++i) {
        double d2 = 0.;
        for (int k = 0; k < np; ++k) cur[i * np + k] = par(&c, v, i, k); /* vect_to_params */
        for (int a = 0; a < nd; ++a) {
          double d = cur[i * np + 2 + a] - coords[i * nd + a];
          d2 += d * d;
        }
        if (!(d2 < p->max_shift * p->max_shift)) moved = 1;     /* refine.py:384 */
      }
      if (!moved) break;
      for (int i = 0; i < n; ++i)
        for (int a = 0; a < nd; ++a) coords[i * nd + a] = cur[i * np + 2 + a];
    }
    if (status == CTR_STATUS_OK && rms > p->max_rms_dev) status = CTR_STATUS_RMS_DEV; /* refine.py:391 */
    b->status[cl] = status;
    if (status == CTR_STATUS_OK) {
      memcpy(pout, cur, sizeof(double) * (size_t)n * np);
      b->cost[cl] = rms;
      if (b->params_std)   /* vect_to_params of the std vector (refine.py:403-406) */
        for (int i = 0; i < n; ++i)
          for (int k = 0; k < np; ++k) {
            int bb = c.L.var_of[k];
            if (bb >= 0) b->params_std[(size_t)(f0 + i) * np + k] = vstd[bb + (c.L.per_feat[k] ? i : 0)];
          }
    }
    free(vecs);
    free(cur);
    free(coords);
    free(c.fwin);
  }
}

static void frame_max(const ctr_batch* b, int nd, double* fmax) {
  size_t felems = 1;
  for (int a = 0; a < nd; ++a) felems *= (size_t)b->shape[a];
  for (int64_t f = 0; f < b->n_frames; ++f) {
    const char* base = (const char*)b->frames + (size_t)f * felems * dtype_size(b->frame_dtype);
    double m = -INFINITY;
    for (size_t i = 0; i < felems; ++i) {
      double x = pixel(base, b->frame_dtype, i);
      if (x > m || x != x) m = x; /* numpy max propagates NaN */
      if (m != m) break;
    }
    fmax[f] = m;
  }
}

int ctro_refine_batch(const ctr_problem* p, const ctr_batch* b, int n_threads) {
  double* fmax = malloc(sizeof(double) * (size_t)(b->n_frames > 0 ? b->n_frames : 1));
  if (p->ndim < 2 || p->ndim > 3 || p->max_iter < 1 || p->fit_function < CTR_FIT_GAUSS || p->fit_function > CTR_FIT_INV_SERIES || p->n_params > CTR_MAX_PARAMS) { free(fmax); return CTR_ERR_INVALID; }
  frame_max(b, p->ndim, fmax);
  if (n_threads < 1) n_threads = 1;
#pragma omp parallel for schedule(dynamic, 16) num_threads(n_threads)
  for (int64_t cl = 0; cl < b->n_clusters; ++cl) refine_one(p, b, cl, fmax);
  free(fmax);
  return CTR_OK;
}

/* Known answer: objective F and its gradient in the reference's normalisation
 * (fitfunc.py:436-487) at the packed start vector of cluster `cl`, first-round
 * window.  vect/grad: [nv]; bounds: [nv][2] (compute_bounds layout).  With
 * v_in != NULL the objective is evaluated at v_in instead (masks unchanged). */
int ctro_objective(const ctr_problem* p, const ctr_batch* b, int64_t cl, double* F,
                   double* vect, double* grad, double* bounds, int32_t* origin,
                   int32_t* wshape, int64_t* P_out, const double* v_in) {
  const int32_t f0 = b->feat_offset[cl], f1 = b->feat_offset[cl + 1];
  const int n = f1 - f0, np = p->n_params, nd = p->ndim;
  const double* params = b->params + (size_t)f0 * np;
  ctx_t c;
  size_t felems = 1;
  double lo[MAXV], hi[MAXV], S;
  long P;
  double* fmax = malloc(sizeof(double) * (size_t)b->n_frames);
  double* coords = malloc(sizeof(double) * (size_t)n * nd);
  double* A;
  c.p = p; c.b = b;
  if (make_layout(p, n, &c.L) > MAXV) { free(fmax); free(coords); return -1; }
  frame_max(b, nd, fmax);
  for (int a = 0; a < nd; ++a) felems *= (size_t)b->shape[a];
  c.frame = (const char*)b->frames + (size_t)b->frame_index[cl] * felems * dtype_size(b->frame_dtype);
  c.n_cons = 0;
  c.pconst = params;
  for (int i = 0; i < n; ++i)
    for (int a = 0; a < nd; ++a) coords[i * nd + a] = params[i * np + 2 + a];
  if (!window(nd, b->shape, p->radius, coords, n, c.origin, c.wshape)) { free(fmax); free(coords); return -2; }
  c.mcoords = coords;
  c.fwin = NULL;
  make_fwin(&c);
  pack_start(&c, params, b->low + (size_t)f0 * np, b->high + (size_t)f0 * np, vect, lo, hi);
  if (v_in) memcpy(vect, v_in, sizeof(double) * c.L.nv); /* evaluate elsewhere, masks stay at p0 */
  A = malloc(sizeof(double) * c.L.nv * c.L.nv);
  eval_cluster(&c, vect, &S, grad, A, NULL, &P);
  {
    const double fm = fmax[b->frame_index[cl]];
    const double norm = fm * fm / p->residual_factor;
    *F = (S / (double)P) / norm;
    for (int i = 0; i < c.L.nv; ++i) {
      grad[i] = 2. * grad[i] / (double)P / norm;
      bounds[2 * i] = lo[i];
      bounds[2 * i + 1] = hi[i];
    }
  }
  for (int a = 0; a < nd; ++a) { origin[a] = c.origin[a]; wshape[a] = c.wshape[a]; }
  *P_out = P;
  free(A); free(fmax); free(coords); free(c.fwin);
  return c.L.nv;
}

/* Model Hessian of F at v_in (masks at the start coordinates, as ctro_objective):
 * hess[nv*nv] = 2 (J^T J + Q) / (P norm), with Q the exact second-order part that solve()
 * adds for (signal, positions) of every feature when exact == 1, the second-order part in all
 * variables (full_second_order, any modes) when exact == 2.  Test hook: a finite
 * difference of ctro_objective's gradient must reproduce it.  Returns nv. */
int ctro_hessian(const ctr_problem* p, const ctr_batch* b, int64_t cl, const double* v_in,
                 int exact, double* hess) {
  const int32_t f0 = b->feat_offset[cl], f1 = b->feat_offset[cl + 1];
  const int n = f1 - f0, np = p->n_params, nd = p->ndim;
  const double* params = b->params + (size_t)f0 * np;
  ctx_t c;
  size_t felems = 1;
  double lo[MAXV], hi[MAXV], vect[MAXV], grad[MAXV], S;
  long P;
  double* fmax = malloc(sizeof(double) * (size_t)b->n_frames);
  double* coords = malloc(sizeof(double) * (size_t)n * nd);
  c.p = p; c.b = b;
  if (make_layout(p, n, &c.L) > MAXV) { free(fmax); free(coords); return -1; }
  frame_max(b, nd, fmax);
  for (int a = 0; a < nd; ++a) felems *= (size_t)b->shape[a];
  c.frame = (const char*)b->frames + (size_t)b->frame_index[cl] * felems * dtype_size(b->frame_dtype);
  c.n_cons = 0;
  c.pconst = params;
  for (int i = 0; i < n; ++i)
    for (int a = 0; a < nd; ++a) coords[i * nd + a] = params[i * np + 2 + a];
  if (!window(nd, b->shape, p->radius, coords, n, c.origin, c.wshape)) { free(fmax); free(coords); return -2; }
  c.mcoords = coords;
  c.fwin = NULL;
  make_fwin(&c);
  pack_start(&c, params, b->low + (size_t)f0 * np, b->high + (size_t)f0 * np, vect, lo, hi);
  if (v_in) memcpy(vect, v_in, sizeof(double) * c.L.nv);
  {
    const int nv = c.L.nv;
    double* A = malloc(sizeof(double) * nv * nv);
    double* Q = calloc((size_t)nv * nv, sizeof(double));
    eval_cluster(&c, vect, &S, grad, A, exact == 1 ? Q : NULL, &P);
    if (exact == 2) full_second_order(&c, vect, Q);
    const double fm = fmax[b->frame_index[cl]];
    const double norm = fm * fm / p->residual_factor;
    for (int i = 0; i < nv * nv; ++i) hess[i] = 2. * (A[i] + Q[i]) / (double)P / norm;
    free(A); free(Q);
  }
  free(fmax); free(coords); free(c.fwin);
  return c.L.nv;
}
