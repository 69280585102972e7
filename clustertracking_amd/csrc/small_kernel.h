// small_kernel.h -- refine_small_kernel: singles and pairs, 16/64 lanes per cluster, registers only
// Part of the MI355X cluster-refinement engine; included by ctrefine.hip inside its
// anonymous namespace (device code only, gfx950).
#ifndef CTREFINE_SMALL_KERNEL_H
#define CTREFINE_SMALL_KERNEL_H

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


// all-reduce inside a group of SG lanes (8 = half a DPP row, 16 = one row, 64 = the whole wave)
template <int SG>
__device__ __forceinline__ double group_sum(double x) {
  if (SG == 8) {
    x += dpp_f64<0xB1>(x);   // quad_perm [1,0,3,2]
    x += dpp_f64<0x4E>(x);   // quad_perm [2,3,0,1]
    x += dpp_f64<0x141>(x);  // row_half_mirror: the other quad of the same eight lanes
    return x;
  }
  x = row_sum(x);
  if (SG == 64) {
    x += __shfl_xor(x, 16);
    x += __shfl_xor(x, 32);
  }
  return x;
}

enum { PH_FETCH = 0, PH_EVAL_INIT = 1, PH_EVAL_TRIAL = 2, PH_STEP_ONLY = 3, PH_DONE = 4 };

// SG = lanes per cluster: 8 (eight singles per wave), 16 (four pairs per wave) or 64 (pairs: a
// quarter of the per-iteration latency, which is what bounds the slowest pair).
// (singles: at least 3 waves per SIMD = at most 168 VGPRs; measured best of 128 / 168 / 203+ with
//  two pixels per lane in flight, tools/ab_classes.py: 0.375 -> 0.305 ms for the 33 k singles of cfg 2)
template <int ND, int NF, bool ISO, int SG>
__global__ void __launch_bounds__(WAVE, NF == 1 ? 3 : 1) refine_small_kernel(const KArgs k, int* __restrict__ counter) {
  constexpr int NV = 1 + NF * (1 + ND);
  constexpr int NR = NV + 1;               // row length incl. the residual
  constexpr int NM = NR * (NR + 1) / 2;    // packed upper triangle
  constexpr int NUF = ND * (ND + 1) / 2;   // second-order sums per feature: U[a<=b] = sum res J_pos_a dE/dpos_b
  constexpr int NU = NF * NUF;
  constexpr int NA = NM + NU;              // everything a pass over the window accumulates
  constexpr int NP = 2 + ND + (ISO ? 1 : ND);
  const int lane = threadIdx.x, sub = lane & (SG - 1), grp = lane / SG;
  // per group: Mcur[NM] Ucur[NU] v0[NV] lo[NV] hi[NV]
  constexpr int GS = NA + 3 * NV;
  __shared__ double lds[(WAVE / SG) * GS];
  double* Mcur = lds + grp * GS;
  double* v0 = Mcur + NA;
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
  double mco[NF][ND], isz2[NF][ND];  // mask centres, 1/size^2 (p0 rows are re-read from HBM when needed)
  double mu = 1e-3, nu = 2., S = 0., pred = 0., norm = 1., rms = NAN;
  bool last_acc = true, bad_size = false;
  double prev_step = INFINITY, trial_step = 0.;  // relative size of the last accepted / of the pending step
  double gain = INFINITY;  // relative decrease of S by the last accepted step
  const char* frame = nullptr;

  while (true) {
    // ---- 1. idle groups pull the next cluster -----------------------------------
    if (phase == PH_FETCH) {
      int id = 0;
      if (sub == 0) id = atomicAdd(counter, 1);
      id = __shfl(id, lane & ~(SG - 1));
      int id_end = k.n_bin;
      if (k.split != nullptr) {
        const int first = *k.split;
        if (k.split_part == 1) id_end = first; else id += first;
      }
      if (id >= id_end) {
        phase = PH_DONE;
      } else {
        cl = k.order[id];
        f0 = k.feat_offset[cl];
        const double* params = k.params + (size_t)f0 * NP;
        const double* low = k.low + (size_t)f0 * NP;
        const double* high = k.high + (size_t)f0 * NP;
        bool finite = true;
        bad_size = false;
#pragma unroll
        for (int i = 0; i < NF; ++i) {
#pragma unroll
          for (int kk = 0; kk < NP; ++kk)
            if (!isfinite(params[i * NP + kk])) finite = false;
#pragma unroll
          for (int a = 0; a < ND; ++a) {
            mco[i][a] = params[i * NP + 2 + a];
            const double sz = params[i * NP + (ISO ? 2 + ND : 2 + ND + a)];
            isz2[i][a] = 1. / (sz * sz);
            bad_size = bad_size || !(sz > 0.);   // no valid model (oracle eval_cluster: model_nan)
          }
        }
        // start vector: mean background (refine.py:361), loosest background bound (fitfunc.py:554-557)
        if (sub == 0) {
          double sb = 0., lb = INFINITY, hb = -INFINITY;
#pragma unroll
          for (int i = 0; i < NF; ++i) {
            sb += params[i * NP];
            lb = fmin(lb, low[i * NP]);
            hb = fmax(hb, high[i * NP]);
            v0[1 + i] = params[i * NP + 1];
            lo[1 + i] = low[i * NP + 1];
            hi[1 + i] = high[i * NP + 1];
#pragma unroll
            for (int a = 0; a < ND; ++a) {
              v0[1 + NF + a * NF + i] = params[i * NP + 2 + a];
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
    double M[NA];
#pragma unroll
    for (int e = 0; e < NA; ++e) M[e] = 0.;
    int P = 0;
    const bool evaluating = (phase == PH_EVAL_INIT || phase == PH_EVAL_TRIAL);
    const int npix_here = evaluating ? npix : 0;
    {
      const int w_last = wshape[ND - 1];
      const float inv_w2 = 1.f / (float)w_last;
      const float inv_w1 = ND == 3 ? 1.f / (float)wshape[1] : 1.f;
      const bool big_window = npix >= (1 << 21);
      const double bg = vt[0];
#ifndef CTR_SMALL_PX
#define CTR_SMALL_PX 2
#endif
#ifndef CTR_SMALL_PX2
#define CTR_SMALL_PX2 1
#endif
#ifndef CTR_SMALL_BRANCHFREE2
#define CTR_SMALL_BRANCHFREE2 1
#endif
      if constexpr (NF == 1 || CTR_SMALL_BRANCHFREE2) {
        constexpr int PX = NF == 1 ? CTR_SMALL_PX : CTR_SMALL_PX2;
        // PX pixels per lane in flight and no branch around the model -- the chains of dependent
        // f64 operations of the pixels (and of the two features of a pair) interleave: the
        // kernel is latency bound at 2-3 waves per SIMD; a lane outside the window or a mask
        // contributes zeros.  Same sums as the loop below, other order.
        for (int base = 0; __any(base < npix_here); base += PX * SG) {
          double rw[PX][NR], Ek[PX][NF][ND];
#pragma unroll
          for (int u = 0; u < PX; ++u) {
            const int q = base + u * SG + sub;
            const bool valid = q < npix_here;
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
            const double pix = valid ? load_pixel(frame, k.frame_dtype, off) : 0.;
            bool in[NF], any = false;
            double gv[NF], tt[NF][ND];
            double res = pix - bg;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
              double rel[ND];
#pragma unroll
              for (int a = 0; a < ND; ++a) rel[a] = mco[i][a] - (double)origin[a];
              in[i] = valid && in_mask<ND>(idx, rel, inv_r2, radius);
              any = any || in[i];
              double r2 = 0.;
#pragma unroll
              for (int a = 0; a < ND; ++a) {
                const double dd = (double)(idx[a] + origin[a]) - vt[1 + NF + a * NF + i];
                tt[i][a] = dd * isz2[i][a];
                r2 += dd * tt[i][a];
              }
              gv[i] = exp(-0.5 * ND * r2);
              res -= in[i] ? vt[1 + i] * gv[i] : 0.;
            }
            const bool ok = any && (res == res);
            P += any ? 1 : 0;
            rw[u][0] = ok ? -1. : 0.;
            rw[u][NV] = ok ? res : 0.;
#pragma unroll
            for (int i = 0; i < NF; ++i) {
              const bool oi = ok && in[i];
              const double sng = -vt[1 + i] * (double)ND * gv[i];
              rw[u][1 + i] = oi ? -gv[i] : 0.;
#pragma unroll
              for (int a = 0; a < ND; ++a) {
                rw[u][1 + NF + a * NF + i] = oi ? sng * tt[i][a] : 0.;
                Ek[u][i][a] = oi ? (double)ND * tt[i][a] : 0.;
              }
            }
          }
          int e = 0;
#pragma unroll
          for (int p = 0; p < NR; ++p)
#pragma unroll
            for (int c2 = p; c2 < NR; ++c2) {
              double t = rw[0][p] * rw[0][c2];
#pragma unroll
              for (int u = 1; u < PX; ++u) t += rw[u][p] * rw[u][c2];
              M[e] += t;
              ++e;
            }
#pragma unroll
          for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int a = 0; a < ND; ++a)
#pragma unroll
              for (int b = a; b < ND; ++b) {
                double t = 0.;
#pragma unroll
                for (int u = 0; u < PX; ++u) t += (rw[u][NV] * rw[u][1 + NF + a * NF + i]) * Ek[u][i][b];
                M[e] += t;
                ++e;
              }
        }
      } else
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
          // fetched first: the load latency hides behind the mask tests and the model
          const double pix = load_pixel(frame, k.frame_dtype, off);
          double row[NR];
#pragma unroll
          for (int j = 0; j < NR; ++j) row[j] = 0.;
          bool any = false;
          double res = 0.;
          double Ed[NF][ND];  // d(-ND/2 r2)/d pos of the features covering this pixel, else 0
#pragma unroll
          for (int i = 0; i < NF; ++i)
#pragma unroll
            for (int a = 0; a < ND; ++a) Ed[i][a] = 0.;
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
              for (int a = 0; a < ND; ++a) {
                const double t = dd[a] * isz2[i][a];
                row[1 + NF + a * NF + i] = sng * t;
                Ed[i][a] = (double)ND * t;
              }
            }
          }
          if (any) {
            res += pix - bg;
            ++P;
            if (res == res) {
              row[0] = -1.;
              row[NV] = res;
              int e = 0;
#pragma unroll
              for (int p = 0; p < NR; ++p)
#pragma unroll
                for (int c2 = p; c2 < NR; ++c2) { M[e] += row[p] * row[c2]; ++e; }
              // second-order sums (block_kernel.h / oracle solve(): the exact model Hessian)
#pragma unroll
              for (int i = 0; i < NF; ++i)
#pragma unroll
                for (int a = 0; a < ND; ++a) {
                  const double rj = res * row[1 + NF + a * NF + i];
#pragma unroll
                  for (int b = a; b < ND; ++b) { M[e] += rj * Ed[i][b]; ++e; }
                }
            }
          }
        }
      }
    }
    if (__any(evaluating)) {
#pragma unroll
      for (int e = 0; e < NA; ++e) M[e] = group_sum<SG>(M[e]);
      P = (int)group_sum<SG>((double)P);
    }

    // ---- 3. accept / reject, next step, convergence, rounds ---------------------
    if (phase == PH_EVAL_INIT || phase == PH_EVAL_TRIAL || phase == PH_STEP_ONLY) {
      const double St = M[NM - 1];
      bool failed = false;
      if (phase == PH_EVAL_INIT) {
        if (P == 0) { status = CTR_STATUS_OUT_OF_BOUNDS; failed = true; }
        else if (!isfinite(St) || bad_size) { status = CTR_STATUS_NO_CONVERGENCE; failed = true; }
        mu = 1e-3; nu = 2.; last_acc = true; gain = INFINITY; prev_step = INFINITY;
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
          gain = act / (0.5 * S + 1e-300);
          accept = true;
          last_acc = true;
          prev_step = trial_step;
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
          for (int e = 0; e < NA; ++e) Mcur[e] = M[e];
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
      }
      bool converged = false;
      if (!failed && it >= maxiter) {
        // iteration limit: a stationary objective still counts as converged (see solve() of the oracle)
        if (gain <= CTR_STALL_TOL) converged = true;
        else { status = CTR_STATUS_NO_CONVERGENCE; failed = true; }
      }
      if (!failed && !converged) {
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
          // model Hessians (lower triangles): GN = J^T J; NW adds sum_p res_p d2res_p over
          // (signal, positions) of each feature -- U from the pixel pass, the rest from the
          // gradient: d2res/ds dpos_a = g_pos_a / s,  d2res/dpos_a^2 += (-ND/size_a^2) s g_s
          double Hn[NV][NV];
#pragma unroll
          for (int p = 0; p < NV; ++p)
#pragma unroll
            for (int c2 = 0; c2 <= p; ++c2) Hn[p][c2] = Mc(p, c2);
#pragma unroll
          for (int i = 0; i < NF; ++i) {
            const double sig = v[1 + i], gs = g[1 + i];
#pragma unroll
            for (int a = 0; a < ND; ++a) {
              const int ca = 1 + NF + a * NF + i;
              Hn[ca][1 + i] += sig != 0. ? g[ca] / sig : 0.;
#pragma unroll
              for (int b = 0; b <= a; ++b) {
                double u = Mcur[NM + i * NUF + (b * ND - (b * (b - 1)) / 2 + (a - b))];
                if (a == b) u -= (double)ND * isz2[i][a] * sig * gs;
                Hn[ca][1 + NF + b * NF + i] += u;
              }
            }
          }
          // one LM step on the free variables with model `newton ? Hn : GN`; false = not positive definite
          auto lm_step = [&](bool newton, double (&vnew)[NV], double& smax, double& pr) -> bool {
            double L[NV][NV], rhs[NV], dinv[NV];
            bool okc = true;
#pragma unroll
            for (int p = 0; p < NV; ++p) {
#pragma unroll
              for (int c2 = 0; c2 <= p; ++c2) {
                double h = (fixed[p] || fixed[c2]) ? 0. : (newton ? Hn[p][c2] : Mc(p, c2));
                if (p == c2) {
                  const double d = Mc(p, p);
                  h = fixed[p] ? 1. : h + mu * (d > 1e-300 ? d : 1.);
                }
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
                double t = L[i][j];
#pragma unroll
                for (int q2 = 0; q2 < j; ++q2) t -= L[i][q2] * L[j][q2];
                L[i][j] = t * di;
              }
            }
            if (!okc) return false;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
              double t = rhs[i];
#pragma unroll
              for (int q2 = 0; q2 < i; ++q2) t -= L[i][q2] * rhs[q2];
              rhs[i] = t * dinv[i];
            }
#pragma unroll
            for (int i = NV - 1; i >= 0; --i) {
              double t = rhs[i];
#pragma unroll
              for (int q2 = i + 1; q2 < NV; ++q2) t -= L[q2][i] * rhs[q2];
              rhs[i] = t * dinv[i];
            }
            double dl[NV];
            smax = 0.;
#pragma unroll
            for (int j = 0; j < NV; ++j) {
              double t = v[j] - rhs[j];
              const double l = lo[j], h = hi[j];
              t = t < l ? l : (t > h ? h : t);
              vnew[j] = t;
              dl[j] = t - v[j];
              smax = fmax(smax, fabs(dl[j]) / (fabs(v[j]) + 1.));
            }
            double acc = 0.;
#pragma unroll
            for (int i = 0; i < NV; ++i) {
              double t = 0.;
#pragma unroll
              for (int j = 0; j < NV; ++j) {
                const int hi_ = i > j ? i : j, lo_ = i > j ? j : i;
                t += (newton ? Hn[hi_][lo_] : Mc(hi_, lo_)) * dl[j];
              }
              acc += dl[i] * (g[i] + 0.5 * t);
            }
            pr = -acc;
            return true;
          };
          const double tiny = ftol * (0.5 * S) + 1e-300;
          double stepmax = 0.;
          // the exact Hessian first; Gauss-Newton where that is not positive definite on the
          // free variables or its projected step is not a descent step of the model
          bool ok_step = lm_step(true, vt, stepmax, pred) && pred > -tiny;
          if (__any(!ok_step)) {
            double vg[NV], sg = 0., pg = 0.;
            const bool okg = lm_step(false, vg, sg, pg);
            if (!ok_step && okg) {
#pragma unroll
              for (int j = 0; j < NV; ++j) vt[j] = vg[j];
              stepmax = sg; pred = pg; ok_step = true;
            }
          }
          if (!ok_step) {
            mu *= nu; nu *= 2.; last_acc = false;
            if (mu > 1e30) { status = CTR_STATUS_NO_CONVERGENCE; failed = true; }
            phase = PH_STEP_ONLY;
          } else {
            converged = (last_acc && stepmax <= xtol) || fabs(pred) <= tiny;
            // converging faster than linearly and the step after this one would be below xtol:
            // finished by TAKING this step, without the pixel pass that would only confirm it
            // (oracle solve(): fast exit)
            if (!converged && last_acc && pred > 0. && stepmax < prev_step && isfinite(prev_step) &&
                stepmax * (stepmax / prev_step) <= xtol) {
#pragma unroll
              for (int j = 0; j < NV; ++j) v[j] = vt[j];
              S = fmax(S - 2. * pred, 0.);
              converged = true;
            }
            trial_step = stepmax;
            phase = PH_EVAL_TRIAL;
            if (!converged && !(pred > 0.)) {
              // the model itself predicts no decrease: rejected without a pixel pass
              mu *= nu; nu *= 2.; last_acc = false;
              if (mu > 1e30) { status = CTR_STATUS_NO_CONVERGENCE; failed = true; }
              phase = PH_STEP_ONLY;
            }
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
              double x = k.params[(size_t)(f0 + ii) * NP + kk];
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
      if (k.params_std != nullptr) {
        // refine.py:400-406: std = sqrt(2 diag(inv(Hessian of F))) at the solution, all variables
        // free; Hessian of F = 2 (J^T J + Q) / (P norm)  =>  std_j = sqrt(P norm [(J^T J + Q)^-1]_jj).
        // Mcur holds [J r]^T [J r] and the second-order sums of the accepted point v.
        double sd[NV];
#pragma unroll
        for (int j = 0; j < NV; ++j) sd[j] = NAN;
        if (ok) {
          auto Mc = [&](int p2, int c2) -> double {
            const int a = p2 < c2 ? p2 : c2, b = p2 < c2 ? c2 : p2;
            return Mcur[a * NR - (a * (a - 1)) / 2 + (b - a)];
          };
          double L[NV][NV];
#pragma unroll
          for (int p2 = 0; p2 < NV; ++p2)
#pragma unroll
            for (int c2 = 0; c2 <= p2; ++c2) L[p2][c2] = Mc(p2, c2);
#pragma unroll
          for (int i = 0; i < NF; ++i) {
            const double sig = v[1 + i], gs = Mc(1 + i, NV);
#pragma unroll
            for (int a = 0; a < ND; ++a) {
              const int ca = 1 + NF + a * NF + i;
              L[ca][1 + i] += sig != 0. ? Mc(ca, NV) / sig : 0.;
#pragma unroll
              for (int b = 0; b <= a; ++b) {
                double u = Mcur[NM + i * NUF + (b * ND - (b * (b - 1)) / 2 + (a - b))];
                if (a == b) u -= (double)ND * isz2[i][a] * sig * gs;
                L[ca][1 + NF + b * NF + i] += u;
              }
            }
          }
          bool pd = true;
          double dinv[NV];
#pragma unroll
          for (int j = 0; j < NV; ++j) {
            double d = L[j][j];
#pragma unroll
            for (int q2 = 0; q2 < j; ++q2) d -= L[j][q2] * L[j][q2];
            if (!(d > 0.) || !isfinite(d)) pd = false;
            const double di = 1. / sqrt(d);
            dinv[j] = di;
#pragma unroll
            for (int i = j + 1; i < NV; ++i) {
              double t = L[i][j];
#pragma unroll
              for (int q2 = 0; q2 < j; ++q2) t -= L[i][q2] * L[j][q2];
              L[i][j] = t * di;
            }
          }
          if (pd) {
            // [(L L^T)^-1]_jj = sum_k (L^-1)_kj^2: column j of L^-1 by forward substitution
#pragma unroll
            for (int j = 0; j < NV; ++j) {
              double y[NV], acc = 0.;
#pragma unroll
              for (int kk = 0; kk < NV; ++kk) {
                double t = kk == j ? 1. : 0.;
#pragma unroll
                for (int q2 = 0; q2 < kk; ++q2)
                  if (q2 >= j) t -= L[kk][q2] * y[q2];
                y[kk] = kk < j ? 0. : t * dinv[kk];
                acc += y[kk] * y[kk];
              }
              sd[j] = sqrt((double)Pround * norm * acc);
            }
          }
        }
        double* ps = k.params_std + (size_t)f0 * NP;
        if (sub < NF) {
#pragma unroll
          for (int ii = 0; ii < NF; ++ii)
            if (ii == sub) {
#pragma unroll
              for (int kk = 0; kk < NP; ++kk) {
                double x = NAN;   // constant parameters (the sizes here) carry no error
                if (kk == 0) x = sd[0];
                else if (kk == 1) x = sd[1 + ii];
                else if (kk < 2 + ND) {
#pragma unroll
                  for (int a = 0; a < ND; ++a)
                    if (kk == 2 + a) x = sd[1 + NF + a * NF + ii];
                }
                ps[ii * NP + kk] = x;
              }
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


#endif  // CTREFINE_SMALL_KERNEL_H
