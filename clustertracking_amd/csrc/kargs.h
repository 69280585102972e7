// kargs.h -- what the translation units of the engine share: kernel arguments and the
// functions through which ctrefine.hip (host runtime) gets at the kernels compiled elsewhere.
// The engine is split into several .hip files only so that they compile in parallel.
#ifndef CTREFINE_KARGS_H
#define CTREFINE_KARGS_H

#include <hip/hip_runtime.h>

#include <cstddef>
#include <cstdint>

#include "ctrefine.h"

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
  double* params_std;    // [N, n_params] or nullptr (ctr_batch.params_std)
  const double* fmax;
  const int32_t* order;  // cluster ids of this bin
  // small kernel only: part of the bin a launch takes.  split == nullptr: all of it;
  // split_part 1: entries [0, *split) (the likely slow fits, front_load_kernel's count),
  // split_part 2: entries [*split, n_bin)
  const int32_t* split;
  int32_t split_part;
  // lowpass of the window (ctr_problem.noise_size): taps of axis a at lp_w[a * LP_STRIDE + 0 ..
  // 2 * lp_half[a]], normalised; an unfiltered axis has lp_half 0 and the single tap 1.
  // nullptr = no lowpass.  Only the LP instantiations of the block kernel read them.
  int32_t lp_half[3];
  const double* lp_w;
  // large-cluster kernel: squared relative residual (preconditioned norm) at which its conjugate
  // gradients stop, far from / near the minimum (large_kernel.h)
  double cg_tol2_far, cg_tol2_near;
};

constexpr int LP_STRIDE = 40;   // >= 2 * 16 + 1 taps (CTR_MAX_NOISE_SIZE = 4)

// a kernel of another translation unit: host-side handle for hipLaunchKernel /
// hipFuncSetAttribute, dynamic LDS bytes and workgroup size
struct KernelInfo {
  const void* fn;
  size_t smem;
  int threads;
};

// ---- workspace of one large cluster (large_kernel.h), in doubles ---------------------------
constexpr int LARGE_MAXNB = 48;   // neighbours (features with overlapping mask ellipsoids) per feature
constexpr int LARGE_AGG = 4;      // features per aggregate of the preconditioner
constexpr int LARGE_AGG_TRI = 416;      // >= (4 * 7) (4 * 7 + 1) / 2 = 406 packed entries of the factor
constexpr int LARGE_AGG_STRIDE = 1280;  // ... + (4 * 7)^2 = 784 of the explicit inverse behind it

struct LargeWs {
  long long nvp, nvp_i;   // variables / features rounded up to a multiple of 8
  long long o_vec, o_cur, o_mco, o_fpar, o_pre, o_uq, o_tile, o_off, o_offc, o_int, o_sync, o_pre2, o_agg, o_pix, o_pixv, cap, total;
};

// n features, npf per-feature and ns shared variables
// cap: pixels of a mask's bounding box, prod(2 radius + 1): room for a feature's list of mask pixels
__host__ __device__ inline LargeWs large_ws(int n, int npf, int ns, long long cap) {
  LargeWs W;
  const long long nn = n, nv = ns + nn * npf;
  W.nvp = (nv + 7) & ~7LL;
  W.nvp_i = (nn + 7) & ~7LL;
  long long o = 0;
  W.o_vec = o;  o += 14 * W.nvp;                 // v vt v0 lo hi g x r z p Ap dl Dm free
  W.o_cur = o;  o += W.nvp_i * CTR_MAX_PARAMS;   // parameter rows
  W.o_mco = o;  o += W.nvp_i * 3 + 8;            // mask centres
  W.o_fpar = o; o += W.nvp_i * 14;               // derived per-feature constants (FP)
  W.o_pre = o;  o += W.nvp_i * 32;               // factors of the diagonal blocks
  W.o_uq = o;   o += W.nvp_i * 16;               // second-order entries + raw sums
  W.o_tile = o; o += 2 * nn * 256;               // accepted / trial 16 x 16 tiles
  W.o_off = o;  o += 2 * nn * LARGE_MAXNB * 64;  // accepted / trial neighbour blocks
  W.o_offc = o; o += nn * LARGE_MAXNB * npf * npf;  // accepted blocks, packed (the CG's copy)
  W.o_int = o;  o += (W.nvp_i + 2 * nn * LARGE_MAXNB + 1) / 2 + 8;   // nbcnt, nbidx, rev (int32)
  // preconditioner aggregates (large_kernel.h): factors of up to n / 2 aggregates of <= 4 features
  // (LARGE_AGG_STRIDE doubles each), and their tables (int32: agg_of[n], parent[n], size[n],
  // members[4 (n / 2 + 1)], n_multi)
  W.o_pre2 = o; o += (nn / 2 + 1) * LARGE_AGG_STRIDE;
  W.o_agg = o;  o += (3 * W.nvp_i + 4 * (nn / 2 + 1) + 8 + 16 * (nn / 2 + 1) + 1) / 2 + 8;   // (+ [n_multi][4][4] neighbour slots)
  // the mask pixels of every feature, compacted (int32: box coordinates, 10 bits per axis), rebuilt
  // every re-window round: [n] counts, then n lists of `cap` entries; their values as doubles
  W.cap = (cap + 63) & ~63LL;
  W.o_pixv = o; o += nn * W.cap;
  // ... and, per feature, the pixels it shares with each neighbour (a pool of 2 `cap` entries of
  // two int32 -- position in the feature's list, packed coordinates -- per feature,
  // [n][LARGE_MAXNB] offsets and counts; a pair that does not fit: count -1)
  W.o_pix = o;  o += (W.nvp_i + 5 * nn * W.cap + 2 * nn * LARGE_MAXNB + 1) / 2 + 8;
  o = (o + 15) & ~15LL;
  W.o_sync = o; o += 64;                         // leader / helper words (large_kernel.h: LSY_*), zero at plan creation
  W.total = (o + 31) & ~31LL;
  return W;
}

// refine_large_kernel<ND, ISO>(KArgs, double* ws, const long long* ws_off_of_cluster, int epoch):
// grid = n_bin leaders + n_bin * (G - 1) helpers (large_kernel.h)
// refine_block_kernel<ND, ISO, NT, W, CONS>: nt = 1..8; throughput != 0: the fewest wavefronts;
// cons != 0: the instantiation for clusters with equality constraints (nt = 1, 2 only)
KernelInfo ctr_block_kernel_2d(int iso, int nt, int throughput, int cons);
KernelInfo ctr_block_kernel_3d(int iso, int nt, int throughput, int cons);
// the same with the lowpass of the window (LP = true); one instantiation per (ndim, iso, nt, cons)
KernelInfo ctr_block_kernel_lp(int ndim, int iso, int nt, int cons);
// ring / disc profiles (FIT = CTR_FIT_RING / CTR_FIT_DISC); cons: nt = 1..3
KernelInfo ctr_block_kernel_fit2d(int iso, int nt, int cons, int fit);
KernelInfo ctr_block_kernel_fit3d(int iso, int nt, int cons, int fit);
KernelInfo ctr_block_kernel_inv(int ndim, int iso, int nt, int cons);   // FIT = CTR_FIT_INV_SERIES
// refine_small_kernel<ND, NF, ISO, SG>(KArgs, int* counter); nullptr if not instantiated
const void* ctr_small_kernel(int ndim, int nf, int iso, int sg);
KernelInfo ctr_large_kernel(int ndim, int iso, int lp);   // lp: with the lowpass of the window

#endif  // CTREFINE_KARGS_H
