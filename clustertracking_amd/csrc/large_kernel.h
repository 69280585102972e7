// large_kernel.h -- refine_large_kernel: clusters beyond the block kernel (> 64 features or
// > 127 variables): a LEADER workgroup of 512 threads per cluster runs the solver, up to 7 HELPER
// workgroups share its pixel passes; the normal matrix is block sparse in HBM/L2.
// Part of the MI355X cluster-refinement engine; included by tu_large.hip inside its anonymous
// namespace (device code only, gfx950).
//
// The reference hands a cluster of any size to SLSQP (refine.py:343-375).  A cluster of n
// features has nv = NS + n NPF variables (NS shared: background and any 'cluster'-mode column;
// NPF per feature), and its Gauss-Newton matrix J^T J is block sparse: the residual is a sum over
// the features, so two features couple only where their masks overlap (plus the dense rows of
// the shared variables).  Stored per feature in the cluster's workspace (kargs.h:large_ws):
//   tile[i]   16 x 16: [r, shared.., own.., r_o, shared_o..]^T [same] summed over the pixels of
//             mask i -- own x own = diagonal block, own x r = gradient, own x shared = coupling;
//             the "_o" columns are multiplied by [i is the lowest feature covering the pixel],
//             so that their sums over i count every union pixel once: S, P, shared x shared
//   off[i][s] NPF x NPF: d_i d_j^T over mask i & mask j for the s-th neighbour j of i
// both from Jacobian rows staged in LDS and contracted with v_mfma_f64_4x4x4_4b_f64 on the live
// 4-column groups, one wavefront per feature over the round's lists of mask pixels (one visit per
// neighbour for its model and the pair's block, then the own tiles), no atomics (bitwise
// reproducible).  The bounded Levenberg-Marquardt iteration is that of the other kernels (oracle
// solve(), m = 0); the linear system is solved by conjugate gradients preconditioned with the
// inverted diagonal blocks of aggregates of strongly coupled features (the oracle factors the
// dense matrix: same minimiser, iteration counts may differ).
#ifndef CTREFINE_LARGE_KERNEL_H
#define CTREFINE_LARGE_KERNEL_H

#ifndef CTR_LARGE_WAVES
#define CTR_LARGE_WAVES 8
#endif
constexpr int LW = CTR_LARGE_WAVES; // wavefronts per workgroup (512 threads: 256 VGPRs per lane -- with 1024
                                   // the kernel spilled 0.8-1.3 KB per lane; two workgroups fit a CU's LDS)
constexpr int LT = LW * WAVE;      // threads
constexpr int LRS = 17;            // row stride of a wave's LDS tile (odd: conflict-free ds_write_b64)
constexpr int LRED = 12;           // values per wave in the reduction scratch
constexpr int LQT = 9;             // second-order entries per feature (block_kernel.h: QT)
constexpr int LNB = 19;            // doubles per neighbour in a wavefront's LDS table: 13 derived constants, (index, reverse slot), list
                                   // cursor, mask centre relative to the window, (offset, length) of the pair's list
                                   // (read with one address for all lanes)
constexpr int LREG = 1408;         // doubles of a wavefront's LDS region: that table + the neighbours' sums per pixel

struct SmemL {
  static constexpr int o_rows = 0;                       // LW row tiles of 64 x LRS; the CG's vectors during a solve
  static constexpr int n_rows = LW * WAVE * LRS > 8 * WAVE * LRS ? LW * WAVE * LRS : 8 * WAVE * LRS;
  static constexpr int o_red = o_rows + n_rows;          // reduction scratch [LW][LRED]
  static constexpr int o_tot = o_red + LW * LRED;        // 256 sums over the features of the tiles
  static constexpr int o_sh = o_tot + 256;               // shared-variable scratch: 8 x 8 + 6 x 8
  // per wavefront: the constants of the neighbours of the feature it is working on
  // (<= LARGE_MAXNB x LNB doubles), then the neighbours' model sums of a segment of its pixels
  static constexpr int o_nb = o_sh + 64 + 48;
  static constexpr int total = o_nb + LW * LREG;
  static_assert(LREG >= LARGE_MAXNB * LNB + 7 * 64 + 8, "room for one tile of pixels (model sum + 6 shared columns + flags) behind a full table");
  static constexpr size_t bytes = (size_t)total * sizeof(double);
};

// all-threads sum of up to LRED values (in place); two workgroup barriers
__device__ __forceinline__ void wg_sum(double* vals, int nvals, double* red, int lane, int wave) {
  for (int q = 0; q < nvals; ++q) {
    const double s = wave_sum(vals[q]);
    if (lane == 0) red[wave * LRED + q] = s;
  }
  __syncthreads();
  for (int q = 0; q < nvals; ++q) {
    double t = 0.;
#pragma unroll
    for (int w = 0; w < LW; ++w) t += red[w * LRED + q];
    vals[q] = t;
  }
  __syncthreads();
}
__device__ __forceinline__ double wg_max(double x, double* red, int lane, int wave) {
  x = wave_max(x);
  if (lane == 0) red[wave * LRED] = x;
  __syncthreads();
  double t = red[0];
#pragma unroll
  for (int w = 1; w < LW; ++w) t = fmax(t, red[w * LRED]);
  __syncthreads();
  return t;
}
__device__ __forceinline__ bool wg_any(bool x, double* red, int lane, int wave) {
  return __syncthreads_or(x ? 1 : 0) != 0;
}

// diagnostic counters of the CTR_STAMPS build only (tests/tools/run_cfg3.py via
// ctr_debug_large_counters): [0] linear solves, [1] conjugate-gradient iterations, [2] pixel
// passes, [3] ticks (100 MHz) in matrix-vector products, [4] in pixel passes, [5] in solves,
// [6] / [7] wave 0 of the leader in feature tiles / pair blocks
#ifdef CTR_STAMPS
__device__ unsigned long long g_large_dbg[48];   // [8..]: solves and CG iterations by (model, outcome), see below
#define LDBG_ADD(slot, val) atomicAdd(&g_large_dbg[slot], (unsigned long long)(val))
#define LDBG_NOW() __builtin_amdgcn_s_memrealtime()
#define LDBG_CYC() __builtin_amdgcn_s_memtime()
#else
#define LDBG_ADD(slot, val) do {} while (0)
#define LDBG_NOW() 0ull
#define LDBG_CYC() 0ull
#endif

// ---- leader / helpers ---------------------------------------------------------------------------
// The pixel pass is parallel over the features (one wavefront per feature); the solver is not.
// Grid: blocks [0, n_bin) are the leaders (block b: cluster order[b]), blocks beyond are helpers
// (block n_bin + h: cluster order[h % n_bin]).  The leader posts a pass as a JOB in the cluster's
// workspace; every wavefront of the leader and of whichever helpers are resident then claims
// features from one counter until none is left.  The leader waits only for features that WERE
// claimed -- by a wavefront that is running -- so a helper that never becomes resident costs
// nothing and nothing can deadlock; helpers leave when the leader posts EXIT (tagged with the
// launch's epoch: a stale word of an earlier launch is never mistaken for it).  Visibility between
// workgroups: agent-scope release before a flag / counter, agent-scope acquire behind it
// (MI355X_MICROARCH.md, inter-workgroup visibility); every spin is bounded.
constexpr int LSY_JOB = 0, LSY_EXIT = 1, LSY_CLAIM = 16, LSY_DONE = 17, LSY_P = 18, LSY_DESC = 32;
constexpr unsigned long long LARGE_SPIN_TICKS = 3000000000ull;   // 30 s at 100 MHz

__device__ __forceinline__ unsigned long long ld_agent(const unsigned long long* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent(unsigned long long* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void release_agent() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}
__device__ __forceinline__ void acquire_agent() {
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// ---- X^T X with v_mfma_f64_4x4x4_4b_f64 ------------------------------------------------------------
// The rows of a wavefront's LDS tile (64 pixels x <= 16 columns) hold few live columns: 8 for the
// default modes ([r, bg, signal, z, y, x, r_o, bg_o]), two 4-column groups of a pair's block.  The
// 16 x 16 x 4 instruction spends 64 cycles per 4 pixels on 256 outputs whatever the columns hold
// (1024 cycles per tile); the 4 x 4 x 4 form (4 blocks of 4 x 4 x 4, 16 cycles) multiplies one
// 4-column group by another for 16 pixels: its blocks serve as four more steps of the sum, added
// up once per feature.  Operand layout (probed on gfx950, tools/mfma_probe.hip):
// A[blk][i][k] in lane 16 k + 4 blk + i, B[blk][k][j] in lane 16 k + 4 blk + j, D[blk][i][j] in lane
// 16 i + 4 blk + j: lane l feeds column (l & 3) of its group for pixel l >> 2 of the 16.
__device__ __forceinline__ double mfma4_group(const double* rows, int step, int group, int lane) {
  return rows[(16 * step + (lane >> 2)) * LRS + 4 * group + (lane & 3)];
}
// sum of the four blocks: afterwards lane 16 i + j (and its three copies) holds entry (i, j)
__device__ __forceinline__ double mfma4_total(double v) {
  v += dpp_f64<0x128>(v);   // row_ror:8 -- lanes l and l ^ 8 of a row of 16
  v += dpp_f64<0x124>(v);   // row_ror:4
  return v;
}

// LP: with the lowpass of the window (ctr_problem.noise_size; device_common.h:lowpass_pixel)
template <int ND, bool ISO, bool LP = false>
__global__ void __launch_bounds__(LT) refine_large_kernel(const KArgs k, double* __restrict__ ws_base,
                                                          const long long* __restrict__ ws_off, const int epoch) {
  constexpr int NP = 2 + ND + (ISO ? 1 : ND);
  constexpr int NSZ = ISO ? 1 : ND;
  constexpr int MAXPF = 7;   // per-feature variables: signal + ND positions + NSZ sizes
  extern __shared__ double smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool helper = (int)blockIdx.x >= k.n_bin;
  const int cl = k.order[helper ? ((int)blockIdx.x - k.n_bin) % k.n_bin : (int)blockIdx.x];
  const int f0 = k.feat_offset[cl], n = k.feat_offset[cl + 1] - f0;
  const double* params = k.params + (size_t)f0 * NP;
  double* pout = k.params_out + (size_t)f0 * NP;
  double* red = smem + SmemL::o_red;
  double* tot = smem + SmemL::o_tot;
  double* shs = smem + SmemL::o_sh;       // [0..63] factor of the shared block, [64..] vectors
  double* myrows = smem + SmemL::o_rows + wave * (WAVE * LRS);

  LayoutB L;
  make_layout_b(k.prob, n, L);
  const int nv = L.nv, NS = L.nshared, NPF = L.npf;
  // columns of the 16-wide row: residual, shared, own, and the "owned pixel" copies
  const int c_own = 1 + NS, c_reso = 1 + NS + NPF, c_sho = 2 + NS + NPF;
  long long box_cap = 1;
#pragma unroll
  for (int a = 0; a < ND; ++a) box_cap *= 2 * (long long)k.prob.radius[a] + 1;
  const LargeWs W = large_ws(n, NPF, NS, box_cap);
  double* ws = ws_base + ws_off[cl];
  double *v = ws + W.o_vec, *vt = v + W.nvp, *v0 = vt + W.nvp, *lo = v0 + W.nvp, *hi = lo + W.nvp,
         *g = hi + W.nvp, *xs = g + W.nvp, *rs_g = xs + W.nvp, *zs_g = rs_g + W.nvp, *ps_g = zs_g + W.nvp,
         *Aps_g = ps_g + W.nvp, *dl = Aps_g + W.nvp, *Dm = dl + W.nvp, *fre = Dm + W.nvp;
  double *cur = ws + W.o_cur, *mco = ws + W.o_mco, *fpar = ws + W.o_fpar, *pre = ws + W.o_pre,
         *uq = ws + W.o_uq;
  double* tileA = ws + W.o_tile;                 // accepted point
  double* tileB = tileA + (size_t)n * 256;       // trial point
  double* offA = ws + W.o_off;
  double* offB = offA + (size_t)n * LARGE_MAXNB * 64;
  double* offC = ws + W.o_offc;                  // (scratch of the aggregation beyond its LDS tables)
  int* nbcnt = (int*)(ws + W.o_int);
  int* nbidx = nbcnt + W.nvp_i;
  int* rev = nbidx + (size_t)n * LARGE_MAXNB;
  unsigned long long* sy = (unsigned long long*)(ws + W.o_sync);   // leader / helper words (LSY_*)
  int* pix_cnt = (int*)(ws + W.o_pix);         // mask pixels of feature i: count, then the list
  int* pix_list = pix_cnt + W.nvp_i;
  double* pix_val = ws + W.o_pixv;        // ... and the pixels' values (after the lowpass, if any)
  // the pixels a feature shares with each of its neighbours: (position in its own list, packed box
  // coordinates), a pool of 2 * cap entries per feature
  int2* pair_pool = (int2*)(pix_list + (size_t)n * W.cap);
  int* pair_off = (int*)(pair_pool + (size_t)n * 2 * W.cap);
  int* pair_cnt = pair_off + (size_t)n * LARGE_MAXNB;
  // aggregates of the preconditioner: strongly coupled features share one diagonal block
  double* pre2 = ws + W.o_pre2;
  int* agg_of = (int*)(ws + W.o_agg);          // multi-feature aggregate of feature i, -1: on its own
  int* agg_par = agg_of + W.nvp_i;             // union-find scratch
  int* agg_sz = agg_par + W.nvp_i;
  int* agg_mem = agg_sz + W.nvp_i;             // [n_multi][LARGE_AGG] members (-1 beyond the size)
  int* agg_nm = agg_mem + LARGE_AGG * (n / 2 + 1);   // [0] number of multi-feature aggregates
  int* agg_slot = agg_nm + 8;                  // [n_multi][4][4]: member b's place in member a's neighbour list, -1: none

  // the exact second-order terms need signal and positions as per-feature variables
  bool newton_on = L.slot[1] >= 0 && L.per_feat[1];
#pragma unroll
  for (int a = 0; a < ND; ++a) newton_on = newton_on && L.slot[2 + a] >= 0 && L.per_feat[2 + a];
  // kind of per-feature slot s: 0 signal, 1 + a position axis a, -1 anything else
  int kind_of[MAXPF];
#pragma unroll
  for (int s2 = 0; s2 < MAXPF; ++s2) {
    int kd = -1;
    if (L.per_feat[1] && L.slot[1] == s2) kd = 0;
#pragma unroll
    for (int a = 0; a < ND; ++a)
      if (L.per_feat[2 + a] && L.slot[2 + a] == s2) kd = 1 + a;
    kind_of[s2] = kd;
  }
  bool size_is_var = false;
#pragma unroll
  for (int kk = 2 + ND; kk < NP; ++kk) size_is_var = size_is_var || L.slot[kk] >= 0;

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
  auto par = [&](const double* vv, int i, int kk) -> double {
    const int b = L.vidx(kk, i);
    if (b < 0) return cur[i * CTR_MAX_PARAMS + kk];
    return vv[b];
  };

  int origin[ND], wshape[ND];
#pragma unroll
  for (int a = 0; a < ND; ++a) { origin[a] = 0; wshape[a] = 1; }
  unsigned job_seq = 0;     // pixel passes posted in this launch (leader) / last one served (helper)

  // The features of one pass over the masks at the point whose derived constants are in fpar:
  // tiles and neighbour blocks -> (tile, off).  Every wavefront that takes part (the leader's and
  // the resident helpers') claims one feature at a time; when none is left the wavefront publishes
  // its share: the owned-pixel count and the number of features it completed.
  auto pass_features = [&](double* tile, double* off, double bgv, unsigned seq) {
    int Pown = 0, taken = 0;
    const bool bg_var = L.slot[0] >= 0;
    // shared variables that collect the features' derivative columns (any 'cluster'-mode column
    // besides the background)
    int nsh2 = 0;
#pragma unroll
    for (int kk = 1; kk < NP; ++kk) nsh2 += (L.slot[kk] >= 0 && !L.per_feat[kk]) ? 1 : 0;
    // 4-column groups of the own tile's live columns and of a pair's d_i / d_j
    const int G4 = (2 + 2 * NS + NPF + 3) >> 2, GP = (NPF + 3) >> 2;
    while (true) {
      // claim the next feature of THIS pass (the counter carries the pass number: a wavefront
      // that is late for a pass that has ended can never take a feature of the next one)
      int i = n;
      if (lane == 0) {
        unsigned long long cur_c = ld_agent(&sy[LSY_CLAIM]);
        while ((unsigned)(cur_c >> 32) == seq && (int)(cur_c & 0xffffffffull) < n) {
          const unsigned long long seen = atomicCAS(&sy[LSY_CLAIM], cur_c, cur_c + 1ull);
          if (seen == cur_c) { i = (int)(cur_c & 0xffffffffull); break; }
          cur_c = seen;
        }
      }
      i = __builtin_amdgcn_readfirstlane(i);
      if (i >= n) break;
      ++taken;
      // box of mask i inside the window (window indices)
      int blo[ND], bsz[ND];
      double rel_i[ND];
      int npx = 1;
#pragma unroll
      for (int a = 0; a < ND; ++a) {
        rel_i[a] = mco[i * 3 + a] - (double)origin[a];
        int l = (int)ceil(rel_i[a] - (double)radius[a]), u = (int)floor(rel_i[a] + (double)radius[a]);
        l = l < 0 ? 0 : l;
        u = u > wshape[a] - 1 ? wshape[a] - 1 : u;
        blo[a] = l;
        bsz[a] = u >= l ? u - l + 1 : 0;
        npx *= bsz[a];
      }
      const int cnt = nbcnt[i];
      const int* nb = nbidx + (size_t)i * LARGE_MAXNB;
      // This feature's constants in registers; its neighbours' are read from the workspace once
      // per feature into the wavefront's LDS table (dependent L2 round trips per tile and
      // neighbour made a tile cost 14 us): 13 derived constants, index, list cursor, mask centre.
      double fi[13];
#pragma unroll
      for (int q2 = 0; q2 < 13; ++q2) fi[q2] = fpar[(size_t)i * FP + q2];
      double* reg = smem + SmemL::o_nb + wave * LREG;
      if (lane < cnt) {
        const int njl = nb[lane];
#pragma unroll
        for (int q2 = 0; q2 < 13; ++q2) reg[lane * LNB + q2] = fpar[(size_t)njl * FP + q2];
        *(int2*)(reg + lane * LNB + 13) = make_int2(njl, rev[(size_t)i * LARGE_MAXNB + lane]);
        reg[lane * LNB + 14] = 0.;
#pragma unroll
        for (int a = 0; a < ND; ++a) reg[lane * LNB + 15 + a] = mco[njl * 3 + a] - (double)origin[a];
        *(int2*)(reg + lane * LNB + 18) = make_int2(pair_off[(size_t)i * LARGE_MAXNB + lane], pair_cnt[(size_t)i * LARGE_MAXNB + lane]);
      }
      // Behind the table, in what is left of the wavefront's region: per pixel of a SEGMENT of
      // the feature's list the sum of the neighbours' models (and of their derivative columns of
      // the shared variables, if there are any besides the background), and one byte "this
      // feature is the lowest that covers the pixel" (one BIT per pixel).  cfg 3: 1072 pixels,
      // one segment up to 16 neighbours.
      double* accb = reg + cnt * LNB;
      const int SEG = (((LREG - cnt * LNB) * 64) / (64 * (1 + nsh2) + 1)) & ~63;
      unsigned* ownf = (unsigned*)(accb + SEG * (1 + nsh2));
      wsync();
      const unsigned long long tf0 = LDBG_NOW();
      // accumulators of the own tile: one per pair (ga <= gb) of 4-column groups
      double accq[10];
#pragma unroll
      for (int t = 0; t < 10; ++t) accq[t] = 0.;
      constexpr int NUF = ND * (ND + 1) / 2;
      double uacc[NUF];
#pragma unroll
      for (int t = 0; t < NUF; ++t) uacc[t] = 0.;
      double* row = myrows + lane * LRS;
      // the mask pixels of feature i from its list (built once per re-window round, box order):
      // packed box coordinates and the pixel's value; 64 of them per tile, every lane at work --
      // 41 % of a 9 x 17 x 17 box is mask
      const int npl = pix_cnt[i];
      const int* plist = pix_list + (size_t)i * W.cap;
      const double* pvals = pix_val + (size_t)i * W.cap;
      const int2* pool = pair_pool + (size_t)i * (2 * W.cap);
      constexpr int PK_NAN = 1 << 30;   // (pair entry: the pixel of the image is NaN)
      for (int q0 = 0; q0 < npl; q0 += SEG) {
        const int q1 = q0 + SEG < npl ? q0 + SEG : npl;
        for (int e = lane; e < SEG * (1 + nsh2); e += WAVE) accb[e] = 0.;
        for (int e = lane; e < SEG / 32; e += WAVE) ownf[e] = 0xffffffffu;
        wsync();
        // ---- One visit per neighbour over the pixels it shares with feature i (the pair's list:
        // positions in i's list, ascending, and coordinates; every lane at work): its model into
        // the pixels' sums, and for j > i the block d_i d_j^T of the normal matrix.  The first
        // tile of the next neighbour's list is fetched while this one is worked on (a visit is
        // one L2 / HBM round trip otherwise: the wavefront has one companion on its SIMD).
        const unsigned long long tc_a = LDBG_CYC();
        int pk_n = 0, pk_n2 = 0;      // list entries of the first two own tiles: in flight during the visits
        double px_n = 0., px_n2 = 0.;
        if (q0 + lane < q1) { pk_n = plist[q0 + lane]; px_n = pvals[q0 + lane]; }
        if (q0 + WAVE + lane < q1) { pk_n2 = plist[q0 + WAVE + lane]; px_n2 = pvals[q0 + WAVE + lane]; }
        int2 en_next = make_int2(0x7fffffff, 0);
        {
          const double* t0 = reg;
          const int2 oc = cnt > 0 ? *(const int2*)(t0 + 18) : make_int2(0, 0);
          const int e0 = (int)t0[14] + lane;
          if (cnt > 0 && oc.y >= 0 && e0 < oc.y) en_next = pool[oc.x + e0];
        }
        for (int s2 = 0; s2 < cnt; ++s2) {
          double* tj = reg + s2 * LNB;
          const int2 jr = *(const int2*)(tj + 13);   // the neighbour and i's place in ITS list
          const int j = jr.x;
          const int2 oc = *(const int2*)(tj + 18);
          const int pc = oc.y;
          const int2* ppl = pool + oc.x;
          const bool pairblk = j > i;
          int2 en_cur = en_next;
          en_next = make_int2(0x7fffffff, 0);
          if (s2 + 1 < cnt) {
            const double* tn = tj + LNB;
            const int2 ocn = *(const int2*)(tn + 18);
            const int e0 = (int)tn[14] + lane;
            if (ocn.y >= 0 && e0 < ocn.y) en_next = pool[ocn.x + e0];
          }
          double accp[4];   // (group of d_i, group of d_j)
#pragma unroll
          for (int t = 0; t < 4; ++t) accp[t] = 0.;
          const unsigned long long tv0 = LDBG_CYC();
          unsigned long long tmf = 0ull;
          // (a pair whose list did not fit the pool: the segment of i's own list, mask test per pixel)
          int e = pc >= 0 ? (int)tj[14] : q0;
          const int eend = pc >= 0 ? pc : q1;
          bool first = true;
          while (e < eend) {
            const int ee = e + lane;
            int qi = 0x7fffffff, pk = 0;
            if (pc >= 0) {
              if (first) { qi = en_cur.x; pk = en_cur.y; }
              else if (ee < eend) { const int2 en = ppl[ee]; qi = en.x; pk = en.y; }
            } else if (ee < eend) {
              qi = ee;
              pk = plist[ee];
            }
            first = false;
            bool in = qi < q1;
            const int m = __popcll(__ballot(in));
            int idx[ND];
#pragma unroll
            for (int a = 0; a < ND; ++a) idx[a] = blo[a] + ((pk >> (10 * (ND - 1 - a))) & 1023);
            if (pc < 0 && in) {
              double rel[ND];
#pragma unroll
              for (int a = 0; a < ND; ++a) rel[a] = tj[15 + a];
              in = in_mask<ND>(idx, rel, inv_r2, radius);
            }
            if (pairblk) {
#pragma unroll
              for (int c2 = 0; c2 < 16; ++c2) row[c2] = 0.;
            }
            if (in) {
              double r2 = 0., dd[ND], d[1 + ND + NSZ];
#pragma unroll
              for (int a = 0; a < ND; ++a) {
                dd[a] = (double)(idx[a] + origin[a]) - tj[1 + a];
                r2 += dd[a] * dd[a] * tj[4 + a];
              }
              const double gv = exp(-0.5 * ND * r2);  // fitfunc.py:112-118
              const double sig = tj[0];
              accb[qi - q0] += sig * gv;
              if (j < i) atomicAnd(&ownf[(qi - q0) >> 5], ~(1u << ((qi - q0) & 31)));
              if (nsh2 > 0 || pairblk) {
                const double sdg = sig * (0.5 * ND) * gv;
                d[0] = -gv;
                double qq = 0.;
#pragma unroll
                for (int a = 0; a < ND; ++a) {
                  d[1 + a] = sdg * (-dd[a] * tj[7 + a]);
                  if (ISO) qq += dd[a] * dd[a];
                  else d[1 + ND + a] = sdg * (dd[a] * dd[a] * tj[10 + a]);
                }
                if (ISO) d[1 + ND] = sdg * (qq * tj[10]);
                if (nsh2 > 0) {
                  int t = 1;
#pragma unroll
                  for (int kk = 1; kk < NP; ++kk)
                    if (L.slot[kk] >= 0 && !L.per_feat[kk]) { accb[t * SEG + qi - q0] += d[kk - 1]; ++t; }
                }
                if (pairblk && (pk & PK_NAN) == 0) {   // (a NaN pixel of the image contributes nothing)
#pragma unroll
                  for (int kk = 1; kk < NP; ++kk)
                    if (L.slot[kk] >= 0 && L.per_feat[kk]) row[8 + L.slot[kk]] = d[kk - 1];
                  // ... and feature i's own derivatives at the pixel
                  double r2i = 0.;
#pragma unroll
                  for (int a = 0; a < ND; ++a) {
                    dd[a] = (double)(idx[a] + origin[a]) - fi[1 + a];
                    r2i += dd[a] * dd[a] * fi[4 + a];
                  }
                  const double gvi = exp(-0.5 * ND * r2i);
                  const double sdgi = fi[0] * (0.5 * ND) * gvi;
                  d[0] = -gvi;
                  qq = 0.;
#pragma unroll
                  for (int a = 0; a < ND; ++a) {
                    d[1 + a] = sdgi * (-dd[a] * fi[7 + a]);
                    if (ISO) qq += dd[a] * dd[a];
                    else d[1 + ND + a] = sdgi * (dd[a] * dd[a] * fi[10 + a]);
                  }
                  if (ISO) d[1 + ND] = sdgi * (qq * fi[10]);
#pragma unroll
                  for (int kk = 1; kk < NP; ++kk)
                    if (L.slot[kk] >= 0 && L.per_feat[kk]) row[L.slot[kk]] = d[kk - 1];
                }
              }
            }
            if (pairblk) {
              wsync();
              const unsigned long long tm0 = LDBG_CYC();
#pragma unroll
              for (int st = 0; st < 4; ++st) {
                double xi[2], xj[2];
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                  xi[g2] = g2 < GP ? mfma4_group(myrows, st, g2, lane) : 0.;
                  xj[g2] = g2 < GP ? mfma4_group(myrows, st, 2 + g2, lane) : 0.;
                }
#pragma unroll
                for (int gi = 0; gi < 2; ++gi)
#pragma unroll
                  for (int gj = 0; gj < 2; ++gj)
                    if (gi < GP && gj < GP) accp[2 * gi + gj] = __builtin_amdgcn_mfma_f64_4x4x4f64(xi[gi], xj[gj], accp[2 * gi + gj], 0, 0, 0);
              }
              wsync();
              tmf += LDBG_CYC() - tm0;
            }
            e += m;
            if (m < WAVE) break;
          }
          if (pc >= 0 && lane == 0) tj[14] = (double)e;
          const unsigned long long tv1 = LDBG_CYC();
          if (pairblk) {
            // columns 0..7 of the rows = d_i, 8..15 = d_j: lane = entry (a, b) of the 8 x 8 block, its
            // value in lane 16 (a & 3) + (b & 3) of the accumulator of groups (a >> 2, b >> 2); both
            // directions; a later segment of i's list adds to the first one's sums
            const int a8 = lane >> 3, b8 = lane & 7;
            double x = 0.;
#pragma unroll
            for (int gi = 0; gi < 2; ++gi)
#pragma unroll
              for (int gj = 0; gj < 2; ++gj)
                if (gi < GP && gj < GP) {
                  const double v = __shfl(mfma4_total(accp[2 * gi + gj]), 16 * (a8 & 3) + (b8 & 3));
                  if ((a8 >> 2) == gi && (b8 >> 2) == gj) x = v;
                }
            // (stored packed, NPF x NPF contiguous doubles per block: what the matrix-vector
            //  products of the solves read, one or two cache lines per block)
            if (a8 < NPF && b8 < NPF) {
              double* oij = off + ((size_t)i * LARGE_MAXNB + s2) * (NPF * NPF);
              double* oji = off + ((size_t)j * LARGE_MAXNB + jr.y) * (NPF * NPF);
              if (q0 != 0) x += oij[a8 * NPF + b8];
              oij[a8 * NPF + b8] = x;
              oji[b8 * NPF + a8] = x;
            }
          }
          wsync();   // (the next neighbour may cover the same pixels)
          if (tid == 0 && !helper) {
            if (pairblk) { LDBG_ADD(28, tv1 - tv0); LDBG_ADD(29, 1); LDBG_ADD(15, tmf); LDBG_ADD(11, LDBG_CYC() - tv1); }
            else { LDBG_ADD(30, tv1 - tv0); LDBG_ADD(31, 1); }
          }
          (void)tv0; (void)tv1; (void)tmf;
        }
        const unsigned long long tc_b = LDBG_CYC();
        if (tid == 0 && !helper) LDBG_ADD(24, tc_b - tc_a);
        (void)tc_a; (void)tc_b;
        // ---- the feature's own tiles of this segment (their list entries are fetched two tiles
        // ahead, the first two before the visits above)
        for (int base = q0; base < q1; base += WAVE) {
          const unsigned long long tc_c = LDBG_CYC();
          const int q = base + lane;
          const bool in_i = q < q1;
          const int pk = pk_n;
          const double pix = px_n;
          pk_n = pk_n2; px_n = px_n2;
          if (q + 2 * WAVE < q1) { pk_n2 = plist[q + 2 * WAVE]; px_n2 = pvals[q + 2 * WAVE]; }
          int idx[ND];
#pragma unroll
          for (int a = 0; a < ND; ++a) idx[a] = in_i ? blo[a] + ((pk >> (10 * (ND - 1 - a))) & 1023) : 0;
#pragma unroll
          for (int c2 = 0; c2 < 16; ++c2) row[c2] = 0.;
          if (in_i) {
            double r2 = 0., dd[ND], d[1 + ND + NSZ], Eown[ND];
#pragma unroll
            for (int a = 0; a < ND; ++a) {
              dd[a] = (double)(idx[a] + origin[a]) - fi[1 + a];
              r2 += dd[a] * dd[a] * fi[4 + a];
            }
            const double gv = exp(-0.5 * ND * r2);  // fitfunc.py:112-118
            const double sig = fi[0];
            const double sdg = sig * (0.5 * ND) * gv;
            // the features that cover this pixel: i itself, then its neighbours in list order
            const double res = ((pix - bgv) - sig * gv) - accb[q - q0];
            d[0] = -gv;
            double qq = 0.;
#pragma unroll
            for (int a = 0; a < ND; ++a) {
              d[1 + a] = sdg * (-dd[a] * fi[7 + a]);
              if (ISO) qq += dd[a] * dd[a];
              else d[1 + ND + a] = sdg * (dd[a] * dd[a] * fi[10 + a]);
              Eown[a] = (double)ND * (dd[a] * fi[4 + a]);
            }
            if (ISO) d[1 + ND] = sdg * (qq * fi[10]);
            const bool owner = ((ownf[(q - q0) >> 5] >> ((q - q0) & 31)) & 1u) != 0u;
            Pown += owner ? 1 : 0;
            if (res == res) {   // nansum (fitfunc.py:449,483): a NaN pixel counts in P only
              const double ow = owner ? 1. : 0.;
              row[0] = res;
              row[c_reso] = ow * res;
              if (bg_var) { row[1 + L.slot[0]] = -1.; row[c_sho + L.slot[0]] = -ow; }
              int t = 1;
#pragma unroll
              for (int kk = 1; kk < NP; ++kk) {
                if (L.slot[kk] < 0) continue;
                if (L.per_feat[kk]) row[c_own + L.slot[kk]] = d[kk - 1];
                else {
                  const double sh = d[kk - 1] + accb[t * SEG + q - q0];
                  ++t;
                  row[1 + L.slot[kk]] = sh;
                  row[c_sho + L.slot[kk]] = ow * sh;
                }
              }
              if (newton_on) {
                int e = 0;
#pragma unroll
                for (int a = 0; a < ND; ++a) {
                  const double rj = res * d[1 + a];
#pragma unroll
                  for (int b2 = a; b2 < ND; ++b2) { uacc[e] += rj * Eown[b2]; ++e; }
                }
              }
            }
          }
          const unsigned long long tc_d = LDBG_CYC();
          wsync();
#pragma unroll
          for (int st = 0; st < 4; ++st) {
            double xg[4];
#pragma unroll
            for (int g2 = 0; g2 < 4; ++g2) xg[g2] = g2 < G4 ? mfma4_group(myrows, st, g2, lane) : 0.;
#pragma unroll
            for (int gb = 0; gb < 4; ++gb)
#pragma unroll
              for (int ga = 0; ga <= gb; ++ga)
                if (gb < G4) accq[gb * (gb + 1) / 2 + ga] = __builtin_amdgcn_mfma_f64_4x4x4f64(xg[ga], xg[gb], accq[gb * (gb + 1) / 2 + ga], 0, 0, 0);
          }
          wsync();
          if (tid == 0 && !helper) { LDBG_ADD(25, tc_d - tc_c); LDBG_ADD(26, LDBG_CYC() - tc_d); LDBG_ADD(27, 1); }
          (void)tc_c; (void)tc_d;
        }
      }
      // entry (4 ga + i, 4 gb + j) from lane 16 i + j of the pair's accumulator, both triangles;
      // zeros beyond the live groups
      {
        double* t = tile + (size_t)i * 256;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int e = lane + 64 * r;
          if ((e >> 4) >= 4 * G4 || (e & 15) >= 4 * G4) t[e] = 0.;
        }
        const int ii = lane >> 4, jj = lane & 3;
#pragma unroll
        for (int gb = 0; gb < 4; ++gb)
#pragma unroll
          for (int ga = 0; ga <= gb; ++ga)
            if (gb < G4) {
              const double v = mfma4_total(accq[gb * (gb + 1) / 2 + ga]);
              if ((lane & 12) == 0) {
                t[(4 * ga + ii) * 16 + 4 * gb + jj] = v;
                if (ga != gb) t[(4 * gb + jj) * 16 + 4 * ga + ii] = v;
              }
            }
      }
      if (newton_on) {
#pragma unroll
        for (int t = 0; t < NUF; ++t) {
          const double s = wave_sum(uacc[t]);
          if (lane == 0) uq[(size_t)i * 16 + LQT + t] = s;   // raw sums of the TRIAL point (tabulated on accept)
        }
      }
      const unsigned long long tf1 = LDBG_NOW();
      if (tid == 0 && !helper) { LDBG_ADD(6, tf1 - tf0); LDBG_ADD(7, LDBG_NOW() - tf1); }
      (void)tf0; (void)tf1;
    }
    if (taken > 0) {
      const double pw = wave_sum((double)Pown);
      release_agent();     // this wavefront's tiles and blocks are visible before its count is
      if (lane == 0) {
        atomicAdd(&sy[LSY_P], (unsigned long long)pw);
        atomicAdd(&sy[LSY_DONE], (unsigned long long)taken);
      }
    }
  };

  if (helper) {
    // ---- a helper: serve the leader's passes until it posts EXIT -----------------------------
    for (int e = tid; e < LW * WAVE * LRS; e += LT) smem[SmemL::o_rows + e] = 0.;
    int* hctl = (int*)(smem + SmemL::o_tot);   // [0] 1 = a pass to serve, 2 = leave
    while (true) {
      __syncthreads();
      if (tid == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        int what = 2;
        while (__builtin_amdgcn_s_memrealtime() - t0 < LARGE_SPIN_TICKS) {
          if (ld_agent(&sy[LSY_EXIT]) == (unsigned long long)(unsigned)epoch) break;
          const unsigned long long w = ld_agent(&sy[LSY_JOB]);
          if ((unsigned)(w >> 32) == (unsigned)epoch && (unsigned)(w & 0xffffffffull) > job_seq) {
            hctl[1] = (int)(w & 0xffffffffull);
            what = 1;
            break;
          }
          __builtin_amdgcn_s_sleep(32);
        }
        if (what == 1) acquire_agent();
        hctl[0] = what;
      }
      __syncthreads();
      if (hctl[0] != 1) return;
      job_seq = (unsigned)hctl[1];
      // the pass: which buffers, the background, the window of the round
      const double* desc = (const double*)(sy + LSY_DESC);
      const bool to_b = desc[0] != 0.;
      const double bgv = desc[1];
#pragma unroll
      for (int a = 0; a < ND; ++a) { origin[a] = (int)desc[2 + a]; wshape[a] = (int)desc[5 + a]; }
      pass_features(to_b ? tileB : tileA, to_b ? offB : offA, bgv, job_seq);
    }
  }
  // ---- set-up ------------------------------------------------------------------------------
  bool finite = true;
  for (int e = tid; e < n * NP; e += LT) {
    const double x = params[e];
    pout[e] = x;  // failures keep their input (refine.py:408-418)
    cur[(e / NP) * CTR_MAX_PARAMS + (e % NP)] = x;
    if (!isfinite(x)) finite = false;
  }
  for (int e = tid; e < LW * WAVE * LRS; e += LT) smem[SmemL::o_rows + e] = 0.;
  for (int e = tid; e < n * 3; e += LT) {
    const int i = e / 3, a = e % 3;
    mco[e] = a < ND ? params[i * NP + 2 + a] : 0.;
  }
  if (k.params_std != nullptr)   // (no covariance output for clusters of this size: documented)
    for (int e = tid; e < n * NP; e += LT) k.params_std[(size_t)f0 * NP + e] = NAN;
  {
    const double* low = k.low + (size_t)f0 * NP;
    const double* high = k.high + (size_t)f0 * NP;
#pragma unroll
    for (int kk = 0; kk < NP; ++kk) {
      if (L.slot[kk] < 0) continue;
      if (L.per_feat[kk]) {
        for (int i = tid; i < n; i += LT) {
          const int b = L.vidx(kk, i);
          v0[b] = params[i * NP + kk];
          lo[b] = low[i * NP + kk];
          hi[b] = high[i * NP + kk];
        }
      } else {
        // shared: mean start (refine.py:361), loosest bound (fitfunc.py:554-557)
        double s = 0., l = INFINITY, h = -INFINITY;
        for (int i = tid; i < n; i += LT) {
          s += params[i * NP + kk];
          l = fmin(l, low[i * NP + kk]);
          h = fmax(h, high[i * NP + kk]);
        }
        double sv[1] = {s};
        wg_sum(sv, 1, red, lane, wave);
        l = -wg_max(-l, red, lane, wave);
        h = wg_max(h, red, lane, wave);
        if (tid == 0) {
          const int b = L.vidx(kk, 0);
          v0[b] = sv[0] / n;
          lo[b] = l;
          hi[b] = h;
        }
      }
    }
  }
  const bool nonfinite = wg_any(!finite, red, lane, wave);

  int status = nonfinite ? CTR_STATUS_NONFINITE : (n <= 0 ? CTR_STATUS_OUT_OF_BOUNDS : CTR_STATUS_OK);
  int round = 0, it = 0, iters = 0, Pround = 0;
  double mu = 1e-3, nu = 2., S = 0., pred = 0., rms = NAN, gain = INFINITY;
  bool last_acc = true, bad_size = false;
  double prev_step = INFINITY, trial_step = 0.;
  const double fm = k.fmax[k.frame_index[cl]];
  const double norm = fm * fm / k.prob.residual_factor;  // refine.py:354
  const double ms2 = k.prob.max_shift * k.prob.max_shift;
  // derived constants of every feature at vv: [0] signal [1..3] centre [4..6] 1/size^2
  // [7..9] 2/size^2 [10..12] -2/size^3
  auto fill_fpar = [&](const double* vv, bool sizes) {
    bool bad = false;
    for (int i = tid; i < n; i += LT) {
      double* f = fpar + (size_t)i * FP;
      f[0] = par(vv, i, 1);
#pragma unroll
      for (int a = 0; a < ND; ++a) {
        f[1 + a] = par(vv, i, 2 + a);
        if (sizes) {
          const double sz = par(vv, i, ISO ? 2 + ND : 2 + ND + a);
          const double s2 = sz * sz;
          bad = bad || !(sz > 0.);
          f[4 + a] = 1. / s2;
          f[7 + a] = 2. / s2;
          f[10 + a] = -2. / (s2 * sz);
        }
      }
    }
    if (sizes) bad_size = wg_any(bad, red, lane, wave);
    else __syncthreads();
  };

  // One pass (leader, all threads): post it, take part, wait for the features the helpers
  // claimed, then the sums over the features; returns P (union pixels).  bgv: background.
  bool sync_lost = false;
  auto evaluate = [&](double* tile, double* off, double bgv, int& Pout) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();           // everything the pass reads (fpar, the neighbour lists) is written
    ++job_seq;
    if (tid == 0) {
      st_agent(&sy[LSY_CLAIM], (unsigned long long)job_seq << 32);
      st_agent(&sy[LSY_DONE], 0ull);
      st_agent(&sy[LSY_P], 0ull);
      double* desc = (double*)(sy + LSY_DESC);
      desc[0] = tile == tileB ? 1. : 0.;
      desc[1] = bgv;
#pragma unroll
      for (int a = 0; a < ND; ++a) { desc[2 + a] = (double)origin[a]; desc[5 + a] = (double)wshape[a]; }
      release_agent();
      st_agent(&sy[LSY_JOB], ((unsigned long long)(unsigned)epoch << 32) | job_seq);
    }
    __syncthreads();
    pass_features(tile, off, bgv, job_seq);
    __syncthreads();
    if (tid == 0) {
      // features claimed by helpers are being computed by running wavefronts: this ends
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      bool got = false;
      while (__builtin_amdgcn_s_memrealtime() - t0 < LARGE_SPIN_TICKS) {
        if (ld_agent(&sy[LSY_DONE]) >= (unsigned long long)n) { got = true; break; }
        __builtin_amdgcn_s_sleep(8);
      }
      acquire_agent();
      red[0] = got ? 1. : 0.;
      red[1] = (double)ld_agent(&sy[LSY_P]);
    }
    __syncthreads();
    if (red[0] == 0.) sync_lost = true;
    const double Ptot = red[1];
    // sums over the features of every tile entry: the "_o" entries are the cluster totals
    __syncthreads();
    if (tid < 256) {
      double s = 0.;
      for (int i = 0; i < n; ++i) s += tile[(size_t)i * 256 + tid];
      tot[tid] = s;
    }
    __syncthreads();   // (orders tot[] for everybody)
    Pout = (int)Ptot;
  };

  // y = (B + mu diag(Dm)) x, rows of fixed variables zeroed when mk != nullptr (x is expected to
  // be zero there already); B = J^T J of (tile, off) plus, with use_q, the second-order entries
  // uq.  One matrix row per thread at a time (n NPF rows over the 512 threads).  Returns x^T y.
  auto matvec = [&](const double* x, double* y, const double* tile, const double* off, double muv,
                    bool use_q, const double* mk) -> double {
    __syncthreads();
    double part[1 + 8];
#pragma unroll
    for (int q = 0; q < 9; ++q) part[q] = 0.;
    double xsh[8];
#pragma unroll
    for (int kq = 0; kq < 8; ++kq) xsh[kq] = kq < NS ? x[kq] : 0.;
    const int nrows = n * NPF;
    for (int r = tid; r < nrows; r += LT) {
      const int i = r / NPF, a = r - i * NPF;
      const int b0 = NS + i * NPF;
      const double* t = tile + (size_t)i * 256 + (c_own + a) * 16;
      const double xa = x[b0 + a];
      double s = muv * Dm[b0 + a] * xa;
#pragma unroll
      for (int b = 0; b < MAXPF; ++b)
        if (b < NPF) s += t[c_own + b] * x[b0 + b];
#pragma unroll
      for (int kq = 0; kq < 8; ++kq)
        if (kq < NS) { s += t[1 + kq] * xsh[kq]; part[1 + kq] += t[1 + kq] * xa; }
      if (use_q) {
        // second-order part between (signal, positions) of this feature
        const double* u = uq + (size_t)i * 16;
        int ka = -1;
#pragma unroll
        for (int q = 0; q < MAXPF; ++q) ka = q == a ? kind_of[q] : ka;
        if (ka >= 0) {
#pragma unroll
          for (int b = 0; b < MAXPF; ++b) {
            if (b >= NPF) continue;
            const int kb = kind_of[b];
            if (kb < 0 || ka + kb == 0) continue;
            const int k0 = ka < kb ? ka : kb, k1 = ka < kb ? kb : ka;
            const int e = k0 == 0 ? k1 - 1 : ND + ((k0 - 1) * ND - ((k0 - 1) * (k0 - 2)) / 2 + (k1 - k0));
            s += u[e] * x[b0 + b];
          }
        }
      }
      const int cnt = nbcnt[i];
      const int* nb = nbidx + (size_t)i * LARGE_MAXNB;
      // (a block is NPF x NPF contiguous doubles, one or two cache lines -- the products of a solve
      //  read them ~200 times, through one CU's L2 port)
      const double* o = off + (size_t)i * LARGE_MAXNB * NPF * NPF + a * NPF;
      if (NPF == 4) {
        // four neighbours at a time, the 20 loads of a group in flight together: one block at
        // a time the product is bound by one memory round trip per neighbour (8 wavefronts x
        // 32 bytes per lane in flight: 27 GB/s).  Same sums in the same order.
        int s2 = 0;
        for (; s2 + 4 <= cnt; s2 += 4) {
          const int j0 = nb[s2], j1 = nb[s2 + 1], j2 = nb[s2 + 2], j3 = nb[s2 + 3];
          const double* ob = o + s2 * 16;
          const double a00 = ob[0], a01 = ob[1], a02 = ob[2], a03 = ob[3];
          const double a10 = ob[16], a11 = ob[17], a12 = ob[18], a13 = ob[19];
          const double a20 = ob[32], a21 = ob[33], a22 = ob[34], a23 = ob[35];
          const double a30 = ob[48], a31 = ob[49], a32 = ob[50], a33 = ob[51];
          const double* x0 = x + NS + j0 * 4;
          const double* x1 = x + NS + j1 * 4;
          const double* x2 = x + NS + j2 * 4;
          const double* x3 = x + NS + j3 * 4;
          s += a00 * x0[0]; s += a01 * x0[1]; s += a02 * x0[2]; s += a03 * x0[3];
          s += a10 * x1[0]; s += a11 * x1[1]; s += a12 * x1[2]; s += a13 * x1[3];
          s += a20 * x2[0]; s += a21 * x2[1]; s += a22 * x2[2]; s += a23 * x2[3];
          s += a30 * x3[0]; s += a31 * x3[1]; s += a32 * x3[2]; s += a33 * x3[3];
        }
        for (; s2 < cnt; ++s2) {
          const double* xj = x + NS + nb[s2] * 4;
          const double* ob = o + s2 * 16;
          s += ob[0] * xj[0]; s += ob[1] * xj[1]; s += ob[2] * xj[2]; s += ob[3] * xj[3];
        }
      } else {
        for (int s2 = 0; s2 < cnt; ++s2) {
          const double* xj = x + NS + nb[s2] * NPF;
          const double* ob = o + s2 * NPF * NPF;
#pragma unroll
          for (int b = 0; b < MAXPF; ++b)
            if (b < NPF) s += ob[b] * xj[b];
        }
      }
      s *= mk ? mk[b0 + a] : 1.;
      y[b0 + a] = s;
      part[0] += xa * s;
    }
    wg_sum(part, 1 + NS, red, lane, wave);
    double dot = part[0];
#pragma unroll
    for (int kq = 0; kq < 8; ++kq) {
      if (kq >= NS) continue;
      double s = part[1 + kq] + muv * Dm[kq] * xsh[kq];
#pragma unroll
      for (int k2 = 0; k2 < 8; ++k2)
        if (k2 < NS) s += tot[(c_sho + kq) * 16 + c_sho + k2] * xsh[k2];
      s *= mk ? mk[kq] : 1.;
      if (tid == 0) y[kq] = s;
      dot += xsh[kq] * s;
    }
    __syncthreads();
    return dot;
  };

  // ---- rounds ----------------------------------------------------------------------------------
  bool tiles_swapped = false;   // accepted data in tileB/offB
  while (status == CTR_STATUS_OK) {
    // window of this round (masks.py:42-68) from the mask centres
    {
      double mn[ND], mx[ND];
      bool any = false;
#pragma unroll
      for (int a = 0; a < ND; ++a) { mn[a] = INFINITY; mx[a] = -INFINITY; }
      for (int i = tid; i < n; i += LT) {
        double ci[ND];
        bool ok = true;
#pragma unroll
        for (int a = 0; a < ND; ++a) {
          ci[a] = rint(mco[i * 3 + a]);
          if (!(ci[a] >= -(double)radius[a] && ci[a] < (double)fshape[a] + radius[a])) ok = false;
        }
        if (ok) {
          any = true;
#pragma unroll
          for (int a = 0; a < ND; ++a) { mn[a] = fmin(mn[a], ci[a]); mx[a] = fmax(mx[a], ci[a]); }
        }
      }
      if (!wg_any(any, red, lane, wave)) { status = CTR_STATUS_OUT_OF_BOUNDS; break; }
#pragma unroll
      for (int a = 0; a < ND; ++a) {
        const double l0 = -wg_max(-mn[a], red, lane, wave), u0 = wg_max(mx[a], red, lane, wave);
        long l = (long)l0 - radius[a], u = (long)u0 + radius[a] + 1;
        l = l < 0 ? 0 : l;
        u = u > fshape[a] ? fshape[a] : u;
        origin[a] = (int)l;
        wshape[a] = (int)(u - l);
      }
    }
    const unsigned long long tr0 = LDBG_NOW();
    // neighbour lists: features whose mask ellipsoids (same semi-axes, the radius) overlap --
    // scaled centre distance <= 2; the masks are pixel subsets of the ellipsoids
    bool overflow = false;
    for (int i = tid; i < n; i += LT) {
      int cnt = 0;
      for (int j = 0; j < n; ++j) {
        if (j == i) continue;
        double s2 = 0.;
#pragma unroll
        for (int a = 0; a < ND; ++a) {
          const double t = (mco[i * 3 + a] - mco[j * 3 + a]) / (2. * radius[a]);
          s2 += t * t;
        }
        if (s2 <= 1. + 1e-9) {
          if (cnt < LARGE_MAXNB) nbidx[(size_t)i * LARGE_MAXNB + cnt] = j;
          ++cnt;
        }
      }
      if (cnt > LARGE_MAXNB) { overflow = true; cnt = LARGE_MAXNB; }
      nbcnt[i] = cnt;
    }
    if (wg_any(overflow, red, lane, wave)) { status = CTR_STATUS_TOO_LARGE; break; }
    for (int i = tid; i < n; i += LT) {
      const int cnt = nbcnt[i];
      for (int s2 = 0; s2 < cnt; ++s2) {
        const int j = nbidx[(size_t)i * LARGE_MAXNB + s2];
        int r = 0;
        for (int s3 = 0; s3 < nbcnt[j]; ++s3)
          if (nbidx[(size_t)j * LARGE_MAXNB + s3] == i) r = s3;
        rev[(size_t)i * LARGE_MAXNB + s2] = r;
      }
    }
    const unsigned long long tr1 = LDBG_NOW();
    // the mask pixels of every feature in this round's window, compacted: box coordinates in box
    // order, 10 bits per axis (one wavefront per feature; 41 cheap tiles for a 9 x 17 x 17 box,
    // once per round against ~25 passes over them)
    for (int i = wave; i < n; i += LW) {
      int blo[ND], bsz[ND], npx = 1;
      double rel_i[ND];
#pragma unroll
      for (int a = 0; a < ND; ++a) {
        rel_i[a] = mco[i * 3 + a] - (double)origin[a];
        int l = (int)ceil(rel_i[a] - (double)radius[a]), u = (int)floor(rel_i[a] + (double)radius[a]);
        l = l < 0 ? 0 : l;
        u = u > wshape[a] - 1 ? wshape[a] - 1 : u;
        blo[a] = l;
        bsz[a] = u >= l ? u - l + 1 : 0;
        npx *= bsz[a];
      }
      int* plist = pix_list + (size_t)i * W.cap;
      double* pvals = pix_val + (size_t)i * W.cap;
      int* lpk = (int*)(smem + SmemL::o_nb + wave * LREG);
      int filled = 0;
      for (int base = 0; base < npx; base += WAVE) {
        const int q = base + lane;
        bool in_i = false;
        int pk = 0;
        int idx[ND];
#pragma unroll
        for (int a = 0; a < ND; ++a) idx[a] = 0;
        if (q < npx) {
          int t = q;
#pragma unroll
          for (int a = ND - 1; a >= 0; --a) {
            const int w = bsz[a];
            const int c2 = t % w;
            t /= w;
            idx[a] = blo[a] + c2;
            pk |= c2 << (10 * (ND - 1 - a));
          }
          in_i = in_mask<ND>(idx, rel_i, inv_r2, radius);
        }
        const unsigned long long bal = __ballot(in_i);
        if (in_i) {
          const int at = filled + __popcll(bal & ((1ull << lane) - 1ull));
          plist[at] = pk;
          if (at < 2 * LREG) lpk[at] = pk;   // (a copy in LDS for the loops below)
        }
        filled += __popcll(bal);
      }
      if (lane == 0) pix_cnt[i] = filled;
      const bool lds_list = filled <= 2 * LREG;
      wsync();
      // the pixels' values go with them: read (and, with a lowpass, filtered) once per round; a loop
      // of its own over the compacted list, so that the loads of several tiles are in flight
#pragma unroll 4
      for (int base = 0; base < filled; base += WAVE) {
        const int q = base + lane;
        if (q < filled) {
          const int pk = lds_list ? lpk[q] : plist[q];
          int idx[ND];
#pragma unroll
          for (int a = 0; a < ND; ++a) idx[a] = blo[a] + ((pk >> (10 * (ND - 1 - a))) & 1023);
          double pix;
          if constexpr (LP) pix = lowpass_pixel<ND>(frame, k.frame_dtype, fshape, origin, wshape, idx, k.lp_w, k.lp_half, k.prob.threshold);
          else {
            const size_t offp = ND == 3
                ? ((size_t)(idx[0] + origin[0]) * fshape[1] + (idx[1] + origin[1])) * fshape[ND - 1] + (idx[ND - 1] + origin[ND - 1])
                : (size_t)(idx[0] + origin[0]) * fshape[ND - 1] + (idx[ND - 1] + origin[ND - 1]);
            pix = load_pixel(frame, k.frame_dtype, offp);
          }
          pvals[q] = pix;
          if (pix != pix) {   // (PK_NAN: a NaN pixel of the image; the pair lists inherit the flag)
            plist[q] = pk | (1 << 30);
            if (lds_list) lpk[q] = pk | (1 << 30);
          }
        }
      }
      wsync();
      // ... and the pixels it shares with every neighbour (out of its own list: position there
      // and packed coordinates, ascending)
      {
        const int cnt = nbcnt[i];
        int2* pool = pair_pool + (size_t)i * (2 * W.cap);
        int used = 0;
        for (int s2 = 0; s2 < cnt; ++s2) {
          const int j = nbidx[(size_t)i * LARGE_MAXNB + s2];
          int got = 0;
          double rel_j[ND];
#pragma unroll
          for (int a = 0; a < ND; ++a) rel_j[a] = mco[j * 3 + a] - (double)origin[a];
          for (int base = 0; base < filled && got >= 0; base += WAVE) {
            const int q = base + lane;
            bool both = false;
            int pk = q < filled ? (lds_list ? lpk[q] : plist[q]) : 0;
            {
              // can the box of j touch these 64 list pixels at all?  (box order: the slowest axis
              // runs from the first lane's to the last lane's value; an axis below one that changes
              // spans its whole range)
              const int last = base + WAVE - 1 < filled ? WAVE - 1 : filled - 1 - base;
              bool same = true, hit = true;
#pragma unroll
              for (int a = 0; a < ND; ++a) {
                const int c0 = __builtin_amdgcn_readfirstlane((pk >> (10 * (ND - 1 - a))) & 1023);
                const int c1 = __builtin_amdgcn_readlane((pk >> (10 * (ND - 1 - a))) & 1023, last);
                const int lo_a = blo[a] + (same ? c0 : 0), hi_a = blo[a] + (same ? c1 : bsz[a] - 1);
                hit = hit && ((double)hi_a >= rel_j[a] - (double)radius[a]) && ((double)lo_a <= rel_j[a] + (double)radius[a]);
                same = same && c0 == c1;
              }
              if (!hit) continue;
            }
            if (q < filled) {
              int idx[ND];
#pragma unroll
              for (int a = 0; a < ND; ++a) idx[a] = blo[a] + ((pk >> (10 * (ND - 1 - a))) & 1023);
              both = in_mask<ND>(idx, rel_j, inv_r2, radius);
            }
            const unsigned long long bal = __ballot(both);
            const int nb2 = __popcll(bal);
            if (used + got + nb2 > 2 * (int)W.cap) { got = -1; break; }   // (the pool is full: the passes test the mask)
            if (both) pool[used + got + __popcll(bal & ((1ull << lane) - 1ull))] = make_int2(q, pk);
            got += nb2;
          }
          if (lane == 0) { pair_off[(size_t)i * LARGE_MAXNB + s2] = used; pair_cnt[(size_t)i * LARGE_MAXNB + s2] = got; }
          used += got > 0 ? got : 0;
        }
      }
    }
    __syncthreads();
    // trial = clipped start vector
    bool infeasible = false;
    for (int i = tid; i < nv; i += LT) {
      if (lo[i] > hi[i]) infeasible = true;
      const double x = v0[i];
      vt[i] = x < lo[i] ? lo[i] : (x > hi[i] ? hi[i] : x);
    }
    if (wg_any(infeasible, red, lane, wave)) { status = CTR_STATUS_NO_CONVERGENCE; break; }
    fill_fpar(vt, size_is_var || round == 0);
    // Aggregates for the preconditioner of this round: features whose start positions are close
    // (in units of their sizes) have nearly dependent columns -- two features 0.25 sizes apart
    // leave eigenvalues of 1e-3 behind a per-feature block-Jacobi preconditioner, and the
    // conjugate gradients then take 130-200 iterations.  Greedy union of the closest pairs first
    // (three distance bands: <= 0.5, <= 1, <= 1.5 sizes), at most LARGE_AGG features per aggregate;
    // one thread, deterministic.  Measured on a 500-feature stack (numpy replica of the matrix at
    // the start vector): 137 -> 16 iterations to a relative residual of 1e-4.
    const unsigned long long tr2 = LDBG_NOW();
    // (The candidate pairs -- closer than 1.5 sizes, few -- are found by all threads; the one thread
    //  that unites them works on tables in LDS while they fit, n <= 1600: on the workspace its
    //  dependent loads cost 20 ms per round for 500 features.)
    {
      constexpr int CMAX = 8;                      // candidates kept per feature (j > i, nearest bands first)
      const bool in_lds = 14 * n + 8 <= 2 * LW * LREG;
      int* ibase = in_lds ? (int*)(smem + SmemL::o_nb) : (int*)offC;
      int* c_cnt = ibase;                          // [n]
      int* c_lst = c_cnt + n;                      // [n][CMAX]: 4 j + band
      int* l_par = in_lds ? c_lst + CMAX * n : agg_par;
      int* l_sz = in_lds ? l_par + n : agg_sz;
      int* l_of = in_lds ? l_sz + n : agg_of;
      int* l_mem = in_lds ? l_of + n : agg_mem;    // [n / 2 + 1][LARGE_AGG]
      for (int i = tid; i < n; i += LT) {
        const int cnt = nbcnt[i];
        int c = 0;
        for (int band = 0; band < 3; ++band) {
          const double lo2 = band == 0 ? -1. : (band == 1 ? 0.25 : 1.), hi2 = band == 0 ? 0.25 : (band == 1 ? 1. : 2.25);
          for (int s2 = 0; s2 < cnt; ++s2) {
            const int j = nbidx[(size_t)i * LARGE_MAXNB + s2];
            if (j <= i) continue;
            double q = 0.;
#pragma unroll
            for (int a = 0; a < ND; ++a) {
              const double d = mco[i * 3 + a] - mco[j * 3 + a];
              q += d * d * 0.5 * (fpar[(size_t)i * FP + 4 + a] + fpar[(size_t)j * FP + 4 + a]);
            }
            if (!(q > lo2 && q <= hi2)) continue;
            if (c < CMAX) c_lst[i * CMAX + c] = 4 * j + band;
            ++c;
          }
        }
        c_cnt[i] = c < CMAX ? c : CMAX;
        l_par[i] = i;
        l_sz[i] = 1;
      }
      __syncthreads();
      if (tid == 0) {
        auto find = [&](int i) { while (l_par[i] != i) { l_par[i] = l_par[l_par[i]]; i = l_par[i]; } return i; };
        for (int band = 0; band < 3; ++band)
          for (int i = 0; i < n; ++i) {
            const int c = c_cnt[i];
            for (int t = 0; t < c; ++t) {
              const int e = c_lst[i * CMAX + t];
              if ((e & 3) != band) continue;
              const int j = e >> 2;
              const int ri = find(i), rj = find(j);
              if (ri != rj && l_sz[ri] + l_sz[rj] <= LARGE_AGG) {
                const int lo_r = ri < rj ? ri : rj, hi_r = ri < rj ? rj : ri;
                l_par[hi_r] = lo_r;
                l_sz[lo_r] += l_sz[hi_r];
              }
            }
          }
        int nm = 0;
        for (int i = 0; i < n; ++i) {
          const int r = find(i);
          if (l_sz[r] < 2) { l_of[i] = -1; continue; }
          if (r == i) {   // (the root is the lowest member: met first)
            l_of[i] = nm;
            for (int q = 0; q < LARGE_AGG; ++q) l_mem[nm * LARGE_AGG + q] = -1;
            l_mem[nm * LARGE_AGG] = i;
            ++nm;
          } else {
            const int m2 = l_of[r];
            l_of[i] = m2;
            for (int q = 1; q < LARGE_AGG; ++q)
              if (l_mem[m2 * LARGE_AGG + q] < 0) { l_mem[m2 * LARGE_AGG + q] = i; break; }
          }
        }
        agg_nm[0] = nm;
      }
      __syncthreads();
      const int nm_all = agg_nm[0];
      if (in_lds) {
        for (int i = tid; i < n; i += LT) agg_of[i] = l_of[i];
        for (int e = tid; e < nm_all * LARGE_AGG; e += LT) agg_mem[e] = l_mem[e];
      }
      // where the block between two members of an aggregate sits (members of a chain need not be
      // neighbours): looked up once per round, not once per factorisation
      for (int e = tid; e < nm_all * LARGE_AGG * LARGE_AGG; e += LT) {
        const int m2 = e / (LARGE_AGG * LARGE_AGG), fa = (e / LARGE_AGG) % LARGE_AGG, fb = e % LARGE_AGG;
        const int ia = l_mem[m2 * LARGE_AGG + fa], ib = l_mem[m2 * LARGE_AGG + fb];
        int sl = -1;
        if (ia >= 0 && ib >= 0 && ia != ib) {
          const int cnt = nbcnt[ia];
          for (int s3 = 0; s3 < cnt; ++s3)
            if (nbidx[(size_t)ia * LARGE_MAXNB + s3] == ib) { sl = s3; break; }
        }
        agg_slot[e] = sl;
      }
    }
    __syncthreads();
    if (tid == 0) { LDBG_ADD(32, tr1 - tr0); LDBG_ADD(33, tr2 - tr1); LDBG_ADD(34, LDBG_NOW() - tr2); LDBG_ADD(38, 1); }
    (void)tr0; (void)tr1; (void)tr2;
    const int n_multi = agg_nm[0];
    it = 0;
    mu = size_is_var ? 1. : 1e-3; nu = 2.; last_acc = true; gain = INFINITY;   // (oracle solve())
    prev_step = INFINITY;

    // ---- one solver run ------------------------------------------------------------------
    bool need_eval = true, first = true, converged = false, failed = false;
    while (!converged && !failed) {
      double* tileT = tiles_swapped ? tileA : tileB;   // where a trial evaluation goes
      double* offT = tiles_swapped ? offA : offB;
      bool accept = false;
      if (need_eval) {
        int P = 0;
        const unsigned long long te0 = LDBG_NOW();
        evaluate(tileT, offT, par(vt, 0, 0), P);
        if (tid == 0) { LDBG_ADD(2, 1); LDBG_ADD(4, LDBG_NOW() - te0); }
        (void)te0;
        double St = tot[c_reso * 16 + c_reso];
        if (bad_size) St = NAN;
        if (sync_lost) { status = CTR_STATUS_NO_CONVERGENCE; failed = true; }   // (a bounded wait ran out)
        if (first) {
          if (P == 0) { status = CTR_STATUS_OUT_OF_BOUNDS; failed = true; }
          else if (!isfinite(St)) { status = CTR_STATUS_NO_CONVERGENCE; failed = true; }
          Pround = P;
          accept = !failed;
          first = false;
        } else {
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
        const unsigned long long ta0 = LDBG_NOW();
        if (accept) {
          S = St;
          tiles_swapped = !tiles_swapped;   // the trial data are the accepted data now
          const double* tl = tiles_swapped ? tileB : tileA;
          for (int i = tid; i < nv; i += LT) v[i] = vt[i];
          // gradient J^T r, Marquardt diagonal, second-order entries at the new point
          // (Marquardt diagonal: diag(J^T J), 1 where that vanishes -- a feature whose signal sits
          //  on its bound 0 has no position derivatives -- as in oracle solve() and block_kernel.h)
          if (tid < NS) {
            g[tid] = tot[(c_sho + tid) * 16 + c_reso];
            const double dd0 = tot[(c_sho + tid) * 16 + c_sho + tid];
            Dm[tid] = dd0 > 1e-300 ? dd0 : 1.;
          }
          for (int i = tid; i < n; i += LT) {
            const double* t = tl + (size_t)i * 256;
            const int b0 = NS + i * NPF;
            for (int a = 0; a < NPF; ++a) {
              g[b0 + a] = t[(c_own + a) * 16];
              const double dd0 = t[(c_own + a) * 16 + c_own + a];
              Dm[b0 + a] = dd0 > 1e-300 ? dd0 : 1.;
            }
            if (newton_on) {
              // block_kernel.h / oracle eval_cluster: d2res/ds dpos_a = g_pos_a / s,
              // d2res/dpos_a dpos_b = U_ab + delta_ab (-ND/size_a^2) s g_s
              double* u = uq + (size_t)i * 16;
              const double sig = vt[b0 + L.slot[1]], gs = t[(c_own + L.slot[1]) * 16];
              int e = 0;
#pragma unroll
              for (int a = 0; a < ND; ++a) {
                const double ga = t[(c_own + L.slot[2 + a]) * 16];
                u[a] = sig != 0. ? ga / sig : 0.;
#pragma unroll
                for (int b2 = a; b2 < ND; ++b2) {
                  double q = u[LQT + e];
                  if (b2 == a) {
                    const double sz = par(vt, i, ISO ? 2 + ND : 2 + ND + a);
                    q += -(double)ND / (sz * sz) * sig * gs;
                  }
                  u[ND + e] = q;
                  ++e;
                }
              }
            }
          }
          __syncthreads();
        }
        if (tid == 0) LDBG_ADD(35, LDBG_NOW() - ta0);
        (void)ta0;
      }
      need_eval = false;
      if (failed) break;
      if (it >= maxiter) {
        // iteration limit: a stationary point still counts as converged (oracle solve())
        if (gain <= CTR_STALL_TOL) { converged = true; break; }
        status = CTR_STATUS_NO_CONVERGENCE; failed = true; break;
      }
      ++it;
      ++iters;
      const double* tl = tiles_swapped ? tileB : tileA;
      const double* ol = tiles_swapped ? offB : offA;
      // active set: fixed if at a bound and the gradient pushes outward
      double nfv[1] = {0.};
      for (int i = tid; i < nv; i += LT) {
        const double gl = g[i];
        const bool fixed = (lo[i] == hi[i]) || (v[i] <= lo[i] && gl > 0.) || (v[i] >= hi[i] && gl < 0.);
        fre[i] = fixed ? 0. : 1.;
        nfv[0] += fixed ? 0. : 1.;
      }
      wg_sum(nfv, 1, red, lane, wave);
      if (nfv[0] == 0.) { converged = true; break; }
      const double tiny = ftol * (0.5 * S) + 1e-300;
      bool ok_step = false;
      double stepmax = 0.;
      for (int attempt = newton_on ? 1 : 0; attempt >= 0 && !ok_step; --attempt) {
        const bool nwt = attempt == 1;
        // ---- preconditioner: Cholesky factors of the diagonal blocks (fixed rows = identity)
        const unsigned long long tq0 = LDBG_NOW();
        bool notpd = false;
        for (int i = tid; i < n; i += LT) {
          const double* t = tl + (size_t)i * 256;
          const double* u = uq + (size_t)i * 16;
          const int b0 = NS + i * NPF;
          double Lm[MAXPF][MAXPF];
#pragma unroll
          for (int a = 0; a < MAXPF; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) {
              double h = 0.;
              if (a < NPF) {
                const bool fa = fre[b0 + a] != 0., fb = fre[b0 + b] != 0.;
                if (fa && fb) {
                  h = t[(c_own + a) * 16 + c_own + b];
                  if (nwt) {
                    const int ka = kind_of[a], kb = kind_of[b];
                    if (ka >= 0 && kb >= 0 && ka + kb > 0) {
                      const int k0 = ka < kb ? ka : kb, k1 = ka < kb ? kb : ka;
                      h += u[k0 == 0 ? k1 - 1 : ND + ((k0 - 1) * ND - ((k0 - 1) * (k0 - 2)) / 2 + (k1 - k0))];
                    }
                  }
                  if (a == b) h += mu * Dm[b0 + a];
                } else if (a == b) h = 1.;
              } else if (a == b) h = 1.;
              Lm[a][b] = h;
            }
          bool okc = true;
#pragma unroll
          for (int j = 0; j < MAXPF; ++j) {
            double d = Lm[j][j];
#pragma unroll
            for (int q = 0; q < j; ++q) d -= Lm[j][q] * Lm[j][q];
            if (!(d > 0.) || !isfinite(d)) okc = false;
            const double di = 1. / sqrt(d);
            Lm[j][j] = di;   // (the reciprocal of the pivot)
#pragma unroll
            for (int r = j + 1; r < MAXPF; ++r) {
              double s = Lm[r][j];
#pragma unroll
              for (int q = 0; q < j; ++q) s -= Lm[r][q] * Lm[j][q];
              Lm[r][j] = s * di;
            }
          }
          if (!okc) notpd = true;
          double* pf = pre + (size_t)i * 32;
          int e = 0;
#pragma unroll
          for (int a = 0; a < MAXPF; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) pf[e++] = Lm[a][b];
        }
        // shared block (NS <= 6): thread 0, in LDS
        if (tid == 0) {
          bool okc = true;
          for (int a = 0; a < NS; ++a)
            for (int b = 0; b <= a; ++b) {
              double h;
              if (fre[a] != 0. && fre[b] != 0.) {
                h = tot[(c_sho + a) * 16 + c_sho + b];
                if (a == b) h += mu * Dm[a];
              } else h = a == b ? 1. : 0.;
              shs[a * 8 + b] = h;
            }
          for (int j = 0; j < NS; ++j) {
            double d = shs[j * 8 + j];
            for (int q = 0; q < j; ++q) d -= shs[j * 8 + q] * shs[j * 8 + q];
            if (!(d > 0.) || !isfinite(d)) okc = false;
            const double di = 1. / sqrt(d);
            shs[j * 8 + j] = di;
            for (int r = j + 1; r < NS; ++r) {
              double s = shs[r * 8 + j];
              for (int q = 0; q < j; ++q) s -= shs[r * 8 + q] * shs[j * 8 + q];
              shs[r * 8 + j] = s * di;
            }
          }
          if (!okc) notpd = true;
        }
        const unsigned long long tq1 = LDBG_NOW();
        // multi-feature aggregates: their diagonal blocks, the neighbour blocks between their
        // members (the accepted packed blocks), the same damping, masks and second-order entries.
        // One WAVEFRONT per aggregate, in its LDS region: the packed matrix entry by entry over the
        // lanes, the cooperative Cholesky of the block kernel, then the explicit inverse -- lane c
        // solves for column c -- written behind nothing: only the inverse is kept (a symmetric
        // positive definite preconditioner needs no more accuracy than that; applying the factors
        // was a chain of ~dim^2 dependent loads for ONE thread per aggregate in every CG iteration,
        // factoring them in the workspace 0.23 ms of one thread per solve).
        {
          double* Hp = smem + SmemL::o_nb + wave * LREG;     // packed lower triangle, <= 406
          double* dinv = Hp + LARGE_AGG_TRI;                 // <= 28
          double* X = dinv + 32;                             // inverse, dim x dim <= 784
          static_assert(LARGE_AGG_TRI + 32 + LARGE_AGG * MAXPF * LARGE_AGG * MAXPF <= LREG, "aggregate scratch in a wavefront's region");
          for (int m2 = wave; m2 < n_multi; m2 += LW) {
            int mem[LARGE_AGG], msz = 0;
#pragma unroll
            for (int q = 0; q < LARGE_AGG; ++q) { mem[q] = agg_mem[m2 * LARGE_AGG + q]; msz += mem[q] >= 0 ? 1 : 0; }
            const int dim = msz * NPF, ne = dim * (dim + 1) / 2;
            for (int e = lane; e < ne; e += WAVE) {
              int ra = (int)((sqrtf(8.f * (float)e + 1.f) - 1.f) * 0.5f);
              while (ra * (ra + 1) / 2 > e) --ra;
              while ((ra + 1) * (ra + 2) / 2 <= e) ++ra;
              const int rb = e - ra * (ra + 1) / 2;
              const int fa = ra / NPF, a = ra - fa * NPF, fb = rb / NPF, b = rb - fb * NPF;
              int ia = 0, ib = 0;
#pragma unroll
              for (int q = 0; q < LARGE_AGG; ++q) { ia = q == fa ? mem[q] : ia; ib = q == fb ? mem[q] : ib; }
              const int ba = NS + ia * NPF, bb = NS + ib * NPF;
              const bool free_a = fre[ba + a] != 0., free_b = fre[bb + b] != 0.;
              double h = 0.;
              if (free_a && free_b) {
                if (ia == ib) {
                  const double* t = tl + (size_t)ia * 256;
                  h = t[(c_own + a) * 16 + c_own + b];
                  if (nwt) {
                    int ka = -1, kb = -1;
#pragma unroll
                    for (int q = 0; q < MAXPF; ++q) { ka = q == a ? kind_of[q] : ka; kb = q == b ? kind_of[q] : kb; }
                    if (ka >= 0 && kb >= 0 && ka + kb > 0) {
                      const int k0 = ka < kb ? ka : kb, k1 = ka < kb ? kb : ka;
                      h += uq[(size_t)ia * 16 + (k0 == 0 ? k1 - 1 : ND + ((k0 - 1) * ND - ((k0 - 1) * (k0 - 2)) / 2 + (k1 - k0)))];
                    }
                  }
                  if (a == b) h += mu * Dm[ba + a];
                } else {
                  const int s3 = agg_slot[(m2 * LARGE_AGG + fa) * LARGE_AGG + fb];
                  if (s3 >= 0) h = ol[((size_t)ia * LARGE_MAXNB + s3) * NPF * NPF + a * NPF + b];
                }
              } else if (ra == rb) h = 1.;
              Hp[e] = h;
            }
            wsync();
            const bool okc = chol_factor_w(Hp, dinv, dim, lane);
            if (!okc) { notpd = true; continue; }
            // column c of the inverse: L L^T x = e_c (rows above c vanish in the forward part)
            if (lane < dim) {
              const int c = lane;
              double* xc = X + c * dim;
              for (int r = 0; r < c; ++r) xc[r] = 0.;
              for (int r = c; r < dim; ++r) {
                double sacc = r == c ? 1. : 0.;
                for (int q = c; q < r; ++q) sacc -= Hp[tri(r) + q] * xc[q];
                xc[r] = sacc * dinv[r];
              }
              for (int r = dim - 1; r >= 0; --r) {
                double sacc = xc[r];
                for (int q = r + 1; q < dim; ++q) sacc -= Hp[tri(q) + r] * xc[q];
                xc[r] = sacc * dinv[r];
              }
            }
            wsync();
            double* Mi = pre2 + (size_t)m2 * LARGE_AGG_STRIDE + LARGE_AGG_TRI;
            for (int e = lane; e < dim * dim; e += WAVE) Mi[e] = X[e];
            wsync();
          }
        }
        if (wg_any(notpd, red, lane, wave)) continue;
        const unsigned long long tq2 = LDBG_NOW();
        if (n_multi > 0) __syncthreads();
        if (tid == 0) { LDBG_ADD(37, LDBG_NOW() - tq0); LDBG_ADD(39, tq1 - tq0); LDBG_ADD(40, tq2 - tq1); }
        (void)tq0; (void)tq1; (void)tq2;
        // z = P^-1 r on this thread's features (and, thread 0, the shared block); returns r.z
        auto precond = [&](const double* rr, double* zz) -> double {
          double acc2 = 0.;
          for (int i = tid; i < n; i += LT) {
            const int m2 = agg_of[i];
            if (m2 >= 0) {
              // a member of an aggregate: its NPF rows of the aggregate's inverse
              int mem[LARGE_AGG], msz = 0, rk = 0;
#pragma unroll
              for (int q = 0; q < LARGE_AGG; ++q) {
                mem[q] = agg_mem[m2 * LARGE_AGG + q];
                msz += mem[q] >= 0 ? 1 : 0;
                rk = mem[q] == i ? q : rk;
              }
              const int dim = msz * NPF;
              const double* Mi = pre2 + (size_t)m2 * LARGE_AGG_STRIDE + LARGE_AGG_TRI + (size_t)(rk * NPF) * dim;
              double zl[MAXPF];
#pragma unroll
              for (int a = 0; a < MAXPF; ++a) zl[a] = 0.;
#pragma unroll
              for (int fq = 0; fq < LARGE_AGG; ++fq) {
                if (fq >= msz) continue;
                const int bq = NS + mem[fq] * NPF;
                for (int aq = 0; aq < NPF; ++aq) {
                  const double rq = rr[bq + aq];
#pragma unroll
                  for (int a = 0; a < MAXPF; ++a)
                    if (a < NPF) zl[a] += Mi[a * dim + fq * NPF + aq] * rq;
                }
              }
              const int b0 = NS + i * NPF;
#pragma unroll
              for (int a = 0; a < MAXPF; ++a)
                if (a < NPF) { zz[b0 + a] = zl[a]; acc2 += rr[b0 + a] * zl[a]; }
              continue;
            }
            const double* pf = pre + (size_t)i * 32;
            const int b0 = NS + i * NPF;
            double y[MAXPF];
#pragma unroll
            for (int a = 0; a < MAXPF; ++a) {
              double s = a < NPF ? rr[b0 + a] : 0.;
#pragma unroll
              for (int q = 0; q < a; ++q) s -= pf[a * (a + 1) / 2 + q] * y[q];
              y[a] = s * pf[a * (a + 1) / 2 + a];
            }
#pragma unroll
            for (int a = MAXPF - 1; a >= 0; --a) {
              double s = y[a];
#pragma unroll
              for (int q = a + 1; q < MAXPF; ++q) s -= pf[q * (q + 1) / 2 + a] * y[q];
              y[a] = s * pf[a * (a + 1) / 2 + a];
            }
#pragma unroll
            for (int a = 0; a < MAXPF; ++a)
              if (a < NPF) { zz[b0 + a] = y[a]; acc2 += rr[b0 + a] * y[a]; }
          }
          if (tid == 0) {
            double y[8];
            for (int a = 0; a < NS; ++a) {
              double s = rr[a];
              for (int q = 0; q < a; ++q) s -= shs[a * 8 + q] * y[q];
              y[a] = s * shs[a * 8 + a];
            }
            for (int a = NS - 1; a >= 0; --a) {
              double s = y[a];
              for (int q = a + 1; q < NS; ++q) s -= shs[q * 8 + a] * y[q];
              y[a] = s * shs[a * 8 + a];
            }
            for (int a = 0; a < NS; ++a) { zz[a] = y[a]; acc2 += rr[a] * y[a]; }
          }
          return acc2;
        };
        // ---- conjugate gradients on the free variables: (B + mu D) x = g ----------------------
        // r, z, p, A p live in LDS while they fit (the row tiles are idle during a solve)
        const unsigned long long tc0 = LDBG_NOW();
        const bool cg_lds = 4 * W.nvp <= SmemL::n_rows;
        double* rs = cg_lds ? smem + SmemL::o_rows : rs_g;
        double* zs = cg_lds ? rs + W.nvp : zs_g;
        double* ps = cg_lds ? zs + W.nvp : ps_g;
        double* Aps = cg_lds ? ps + W.nvp : Aps_g;
        for (int i = tid; i < nv; i += LT) { xs[i] = 0.; rs[i] = g[i] * fre[i]; }
        // (a thread touches the entries of "its" features in every phase; the shared entries are
        //  thread 0's -- so only the matrix products and the sums need the barriers)
        __syncthreads();
        double rz[1] = {precond(rs, zs)};
        wg_sum(rz, 1, red, lane, wave);
        for (int i = tid; i < nv; i += LT) ps[i] = zs[i];
        const double rz0 = rz[0];
        bool cg_fail = !(rz0 >= 0.) || !isfinite(rz0);
        // Inexact steps: relative residual 1e-6 in the preconditioned norm while the iteration is far
        // from the minimum, 1e-11 once the accepted steps are below 1e-4 (the convergence test looks
        // at steps of 1e-9 relative size, and the fast exit at their ratio).  With the aggregated
        // preconditioner a solve takes ~15 iterations either way (1e-4: -12 % time, 6e-7 instead of
        // 5e-8 px from the oracle's dense solve on a 500-feature stack).
        const double cg_tol2 = (last_acc && prev_step < 1e-4) ? k.cg_tol2_near : k.cg_tol2_far;
        const int cg_max = nv < 400 ? nv + 20 : 420;
        int cg_it = 0;
        for (int ci = 0; ci < cg_max && !cg_fail && rz[0] > cg_tol2 * rz0 && rz[0] > 0.; ++ci) {
          ++cg_it;
          const unsigned long long tm0 = LDBG_NOW();
          const double pAp = matvec(ps, Aps, tl, ol, mu, nwt, fre);
          if (tid == 0) LDBG_ADD(3, LDBG_NOW() - tm0);
          (void)tm0;
          if (!(pAp > 0.) || !isfinite(pAp)) { cg_fail = true; break; }   // not positive definite
          const double alpha = rz[0] / pAp;
          for (int i = tid; i < n; i += LT)
            for (int a = 0; a < NPF; ++a) {
              const int e = NS + i * NPF + a;
              xs[e] += alpha * ps[e];
              rs[e] -= alpha * Aps[e];
            }
          if (tid == 0)
            for (int a = 0; a < NS; ++a) { xs[a] += alpha * ps[a]; rs[a] -= alpha * Aps[a]; }
          if (n_multi > 0) __syncthreads();   // (an aggregate's thread reads residual entries other threads own)
          double rzn[1] = {precond(rs, zs)};
          wg_sum(rzn, 1, red, lane, wave);
          const double beta = rzn[0] / rz[0];
          rz[0] = rzn[0];
          for (int i = tid; i < n; i += LT)
            for (int a = 0; a < NPF; ++a) {
              const int e = NS + i * NPF + a;
              ps[e] = zs[e] + beta * ps[e];
            }
          if (tid == 0)
            for (int a = 0; a < NS; ++a) ps[a] = zs[a] + beta * ps[a];
        }
        if (tid == 0) { LDBG_ADD(0, 1); LDBG_ADD(1, cg_it); LDBG_ADD(5, LDBG_NOW() - tc0); }
        // [8 + 4 k] solves, [9 + 4 k] their CG iterations, [10 + 4 k] of which at the iteration cap;
        // k = 0 exact Hessian converged, 1 exact Hessian 'not positive definite', 2 J^T J converged, 3 J^T J failed
        if (tid == 0) {
          const int kq = (nwt ? 0 : 2) + (cg_fail ? 1 : 0);
          LDBG_ADD(8 + 4 * kq, 1); LDBG_ADD(9 + 4 * kq, cg_it); LDBG_ADD(10 + 4 * kq, cg_it >= cg_max ? 1 : 0);
          (void)kq;
        }
        (void)tc0; (void)cg_it;
        if (cg_fail) continue;
        __syncthreads();
        // projected trial point
        double sm = 0.;
        for (int i = tid; i < nv; i += LT) {
          double t = v[i] - xs[i] * fre[i];
          t = t < lo[i] ? lo[i] : (t > hi[i] ? hi[i] : t);
          vt[i] = t;
          const double d = t - v[i];
          dl[i] = d;
          sm = fmax(sm, fabs(d) / (fabs(v[i]) + 1.));
        }
        stepmax = wg_max(sm, red, lane, wave);
        // predicted decrease of the model along the actual step: -(g.dl + 1/2 dl^T B dl)
        const double dBd = matvec(dl, Aps, tl, ol, 0., nwt, nullptr);
        double gd[1] = {0.};
        for (int i = tid; i < nv; i += LT) gd[0] += g[i] * dl[i];
        wg_sum(gd, 1, red, lane, wave);
        pred = -(gd[0] + 0.5 * dBd);
        if (nwt && !(pred > -tiny)) continue;
        ok_step = true;
      }
      if (!ok_step) {
        mu *= nu; nu *= 2.; last_acc = false;
        if (mu > 1e30) { status = CTR_STATUS_NO_CONVERGENCE; failed = true; }
        continue;
      }
      if ((last_acc && stepmax <= xtol) || fabs(pred) <= tiny) { converged = true; break; }
      if (last_acc && pred > 0. && stepmax < prev_step && isfinite(prev_step) &&
          stepmax * (stepmax / prev_step) <= xtol) {
        // (oracle solve(): fast exit -- the step is taken, the confirming pixel pass skipped)
        for (int i = tid; i < nv; i += LT) v[i] = vt[i];
        __syncthreads();
        S = fmax(S - 2. * pred, 0.);
        converged = true;
        break;
      }
      trial_step = stepmax;
      if (!(pred > 0.)) {
        // the model itself predicts no decrease: rejected without a pixel pass
        mu *= nu; nu *= 2.; last_acc = false;
        if (mu > 1e30) { status = CTR_STATUS_NO_CONVERGENCE; failed = true; }
        continue;
      }
      fill_fpar(vt, size_is_var);
      need_eval = true;
    }
    if (failed || status != CTR_STATUS_OK) break;
    // ---- end of a round: vect_to_params and the shift test (refine.py:379-388) ---------------
    rms = sqrt(((S / (double)Pround) / norm) / k.prob.residual_factor);
    bool moved = false;
    for (int i = tid; i < n; i += LT) {
      double d2 = 0.;
#pragma unroll
      for (int kk = 0; kk < NP; ++kk) {
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
    const bool any_moved = wg_any(moved, red, lane, wave);
    ++round;
    if (!any_moved || round >= k.prob.max_iter) {
      if (rms > k.prob.max_rms_dev) status = CTR_STATUS_RMS_DEV;  // refine.py:391
      break;
    }
    for (int e = tid; e < n * 3; e += LT) {
      const int i = e / 3, a = e % 3;
      if (a < ND) mco[e] = cur[i * CTR_MAX_PARAMS + 2 + a];
    }
    __syncthreads();
  }

  __syncthreads();
  if (tid == 0) st_agent(&sy[LSY_EXIT], (unsigned long long)(unsigned)epoch);   // the helpers may leave
  const bool ok = status == CTR_STATUS_OK;
  if (ok)
    for (int e = tid; e < n * NP; e += LT) pout[e] = cur[(e / NP) * CTR_MAX_PARAMS + (e % NP)];
  if (tid == 0) {
    k.status[cl] = status;
    k.cost[cl] = ok ? rms : NAN;
    k.n_rounds[cl] = status == CTR_STATUS_NONFINITE || n <= 0 ? 0
                   : (ok || status == CTR_STATUS_RMS_DEV ? round : round + 1);
    k.n_iter[cl] = iters;
  }
}

#endif  // CTREFINE_LARGE_KERNEL_H
