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
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <mutex>
#include <string>
#include <vector>

#include "kargs.h"

namespace {

#include "device_common.h"
#include "aux_kernels.h"
#include "synth_kernels.h"

// ---- host side ------------------------------------------------------------------------

typedef const void* kernel_fn;   // a kernel of another translation unit (kargs.h)
typedef const void* small_fn;

std::string g_create_error;
std::mutex g_mutex;

}  // namespace

// bins: 0..MAXNT-1 generic kernel by NT-1; MAXNT = beyond the engine (see CTR_STATUS_TOO_LARGE);
// MAXNT+1 / MAXNT+2 = singles / pairs with default modes (small kernel); MAXNT+3 = large
// clusters (more than MAXF features or 16 MAXNT columns: refine_large_kernel)
// MAXNT+4 / MAXNT+5 = clusters with equality constraints, NT = 1 / 2 (their own instantiation of
// the block kernel)
constexpr int BIN_TOO_LARGE = MAXNT, BIN_SMALL1 = MAXNT + 1, BIN_SMALL2 = MAXNT + 2, BIN_LARGE = MAXNT + 3,
              BIN_CONS1 = MAXNT + 4, BIN_CONS2 = MAXNT + 5, NBINS = MAXNT + 6;
static_assert(NBINS <= 16, "FrontArgs (aux_kernels.h) holds 16 bins");
// side streams for concurrent bin launches.  Three, not more: HIP deals streams to the hardware
// queues round robin, and with 1 + 3 streams per handle and a queue count that is a multiple of 4
// the streams that share a queue have the same role in different handles; eight handles in flight
// reach the same rate as with four side streams (48 M cluster-fits/s on cfg 2) over a wider range
// of GPU_MAX_HW_QUEUES (16..22 instead of 20..22)
constexpr int NSIDE = 3;
constexpr int GATE_US = 20;  // head start of the block kernels over the small kernels (delay_kernel)

struct ctr_plan {
  ctr_problem prob;
  int64_t n_clusters = 0;
  int device = 0;
  int32_t* d_order = nullptr;  // all bins back to back; second copy [n_clusters..2n): the order of one call
  int* d_front = nullptr;      // 2 * NBINS counters of front_load_kernel
  // large clusters: workspace in HBM (kargs.h:large_ws) and where each cluster's part begins.
  // The workspace belongs to the plan: calls that use one plan must not overlap in time.
  double* d_ws = nullptr;
  long long* d_ws_off = nullptr;   // [n_clusters] offset in doubles (0 for the other clusters)
  mutable int large_epoch = 0;     // launches of the large kernel on this plan (tags its leader / helper words)
  int64_t bin_begin[NBINS + 1] = {0};
  int64_t bin_count[NBINS] = {0};
  // lowpass of the window (ctr_problem.noise_size): taps per axis on the device (kargs.h: lp_w)
  bool lowpass = false;
  double* d_lp_w = nullptr;
  int lp_half[3] = {0, 0, 0};
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
  KernelInfo cons[2][2][2][2];       // [ndim-2][iso][throughput][nt-1]: constrained clusters
  bool cons_attr[2][2][2][2] = {};
  kernel_fn table_tp[2][2][MAXNT];   // the same for CTR_FLAG_THROUGHPUT (fewest wavefronts)
  size_t smem_bytes_tp[2][2][MAXNT];
  int block_threads_tp[2][2][MAXNT];
  bool attr_set_tp[2][2][MAXNT] = {};
  small_fn small_table[2][2][2];  // [ndim-2][iso][nf-1]; singles with 8 lanes per cluster (eight per wavefront)
  small_fn small_wide1[2][2];     // singles with 64 lanes per cluster (large windows)
  small_fn small_bulk2[2][2];     // pairs with 16 lanes per cluster (the bulk of a pairs bin)
  KernelInfo lp[2][2][MAXNT];     // block kernel with the lowpass of the window: [ndim-2][iso][nt-1]
  KernelInfo lp_cons[2][2][2];    // ... for constrained clusters, nt = 1, 2
  KernelInfo fit[3][2][2][MAXNT]; // ring / disc / inv_series profiles: [fit-1][ndim-2][iso][nt-1]
  KernelInfo fit_cons[3][2][2][2];
  bool fit_attr[3][2][2][MAXNT] = {};
  bool fit_cons_attr[3][2][2][2] = {};
  bool lp_attr[2][2][MAXNT] = {};
  bool lp_cons_attr[2][2][2] = {};
  KernelInfo large[2][2][2];      // refine_large_kernel<ndim, iso, lowpass>
  bool large_attr[2][2][2] = {};
  hipEvent_t ev_done = nullptr;   // end of the last ctr_refine_batch_device call of this handle
  hipStream_t side[NSIDE] = {};
  hipEvent_t ev_fork = nullptr, ev_gate = nullptr, ev_order = nullptr, ev_join[NSIDE] = {};
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
  if (p->fit_function < CTR_FIT_GAUSS || p->fit_function > CTR_FIT_INV_SERIES) { msg = "unknown fit function"; return CTR_ERR_INVALID; }
  const bool other_profile = p->fit_function != CTR_FIT_GAUSS;
  // ring: 'thickness', disc: 'disc_size' -- one more column after the sizes (fitfunc.py:195-204);
  // inv_series_<N>: 'signal_mult', 'param_a', ... N + 1 columns (fitfunc.py:334-343)
  const int base = 2 + p->ndim + (p->isotropic ? 1 : p->ndim);
  int np = base + (other_profile ? 1 : 0);
  if (p->fit_function == CTR_FIT_INV_SERIES) {
    if (p->n_params < base + 1) { msg = "inv_series needs at least the column 'signal_mult'"; return CTR_ERR_INVALID; }
    if (p->n_params > CTR_MAX_PARAMS) { msg = "inv_series: more than CTR_MAX_PARAMS parameter columns"; return CTR_ERR_UNSUPPORTED; }
    np = p->n_params;
  }
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
  for (int a = 0; a < p->ndim; ++a)
    if (!(p->noise_size[a] >= 0.) || p->noise_size[a] > CTR_MAX_NOISE_SIZE) {
      msg = "noise_size must be between 0 and CTR_MAX_NOISE_SIZE";
      return CTR_ERR_INVALID;
    }
  if (p->threshold != p->threshold) { msg = "threshold is NaN"; return CTR_ERR_INVALID; }
  if (other_profile) {
    bool lp = (p->flags & CTR_FLAG_WINDOW_FILTER) != 0;
    for (int a = 0; a < p->ndim; ++a) lp = lp || p->noise_size[a] > 0.;
    if (lp) { msg = "noise_size together with a profile other than gauss is not implemented"; return CTR_ERR_UNSUPPORTED; }
  }
  return CTR_OK;
}

// trackpy.masks.gaussian_kernel(sigma, truncate=4), the taps of the reference's lowpass
// (preprocessing.py:43).  trackpy is not part of the reference's tree: restated from its
// published source (lw = int(4 sigma + 0.5); exp(x^2 / (-2 sigma^2)) / sum, numpy's pairwise
// sum for short arrays); the same code as oracle/ctr_oracle.c:ctro_gaussian_kernel.
int gaussian_taps(double sigma, double* w) {
  const int lw = (int)(4.0 * sigma + 0.5), nw = 2 * lw + 1;
  double sum;
  for (int i = 0; i < nw; ++i) {
    const double x = (double)(i - lw);
    w[i] = std::exp((x * x) / (-2. * (sigma * sigma)));
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
  bool evok = hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&h->ev_gate, hipEventDisableTiming) == hipSuccess &&
              hipEventCreateWithFlags(&h->ev_order, hipEventDisableTiming) == hipSuccess;
  for (auto& ev : h->ev_join) evok = evok && hipEventCreateWithFlags(&ev, hipEventDisableTiming) == hipSuccess;
  evok = evok && hipEventCreateWithFlags(&h->ev_done, hipEventDisableTiming) == hipSuccess;
  if (!evok || hipMalloc((void**)&h->d_counter, sizeof(int) * 8) != hipSuccess) { delete h; return fail(nullptr, CTR_ERR_DEVICE, "cannot create events / counters"); }
  for (int di = 0; di < 2; ++di)
    for (int ii = 0; ii < 2; ++ii) {
      h->small_wide1[di][ii] = ctr_small_kernel(2 + di, 1, ii, 64);
      h->small_bulk2[di][ii] = ctr_small_kernel(2 + di, 2, ii, 16);
      h->small_table[di][ii][0] = ctr_small_kernel(2 + di, 1, ii, 8);
      h->small_table[di][ii][1] = ctr_small_kernel(2 + di, 2, ii, 64);
      h->large[di][ii][0] = ctr_large_kernel(2 + di, ii, 0);
      h->large[di][ii][1] = ctr_large_kernel(2 + di, ii, 1);
      for (int nt = 1; nt <= MAXNT; ++nt) {
        const KernelInfo a = di == 0 ? ctr_block_kernel_2d(ii, nt, 0, 0) : ctr_block_kernel_3d(ii, nt, 0, 0);
        const KernelInfo t = di == 0 ? ctr_block_kernel_2d(ii, nt, 1, 0) : ctr_block_kernel_3d(ii, nt, 1, 0);
        h->lp[di][ii][nt - 1] = ctr_block_kernel_lp(2 + di, ii, nt, 0);
        if (nt <= 2) h->lp_cons[di][ii][nt - 1] = ctr_block_kernel_lp(2 + di, ii, nt, 1);
        for (int fi = 0; fi < 2; ++fi) {
          h->fit[fi][di][ii][nt - 1] = di == 0 ? ctr_block_kernel_fit2d(ii, nt, 0, fi + 1) : ctr_block_kernel_fit3d(ii, nt, 0, fi + 1);
          if (nt <= 2) h->fit_cons[fi][di][ii][nt - 1] = di == 0 ? ctr_block_kernel_fit2d(ii, nt, 1, fi + 1) : ctr_block_kernel_fit3d(ii, nt, 1, fi + 1);
        }
        h->fit[2][di][ii][nt - 1] = ctr_block_kernel_inv(2 + di, ii, nt, 0);
        if (nt <= 2) h->fit_cons[2][di][ii][nt - 1] = ctr_block_kernel_inv(2 + di, ii, nt, 1);
        if (nt <= 2)
          for (int tp = 0; tp < 2; ++tp)
            h->cons[di][ii][tp][nt - 1] = di == 0 ? ctr_block_kernel_2d(ii, nt, tp, 1) : ctr_block_kernel_3d(ii, nt, tp, 1);
        h->table[di][ii][nt - 1] = a.fn; h->smem_bytes[di][ii][nt - 1] = a.smem; h->block_threads[di][ii][nt - 1] = a.threads;
        h->table_tp[di][ii][nt - 1] = t.fn; h->smem_bytes_tp[di][ii][nt - 1] = t.smem; h->block_threads_tp[di][ii][nt - 1] = t.threads;
      }
    }
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
  if (h->ev_gate) (void)hipEventDestroy(h->ev_gate);
  if (h->ev_order) (void)hipEventDestroy(h->ev_order);
  if (h->ev_done) (void)hipEventDestroy(h->ev_done);
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
  int npf = 0, nsh = 0;   // per-feature / shared variables
  for (int k2 = 0; k2 < p->n_params; ++k2) {
    if (p->modes[k2] == CTR_MODE_VAR) ++npf;
    else if (p->modes[k2] != CTR_MODE_CONST) ++nsh;
  }
  // (noise_size given with every sigma 0: the reference's lowpass is then its threshold alone)
  plan->lowpass = (p->flags & CTR_FLAG_WINDOW_FILTER) != 0;
  for (int a = 0; a < p->ndim; ++a) plan->lowpass = plan->lowpass || p->noise_size[a] > 0.;
  // (the lowpass lives in its own instantiations of the block kernel: every cluster goes there)
  if (plan->lowpass) default_modes = false;
  // (so do the ring / disc profiles; clusters beyond the block kernel get status 5 there)
  const bool other_profile = p->fit_function != CTR_FIT_GAUSS;
  if (other_profile) default_modes = false;
  std::vector<long long> ws_off((size_t)n_clusters, 0);
  long long ws_total = 0;
  long long box_cap = 1;     // pixels of a mask's bounding box (large_ws)
  for (int a = 0; a < p->ndim; ++a) box_cap *= 2 * (long long)p->radius[a] + 1;
  for (int64_t c = 0; c < n_clusters; ++c) {
    const int64_t n = (int64_t)feat_offset_host[c + 1] - feat_offset_host[c];
    if (n < 0) { delete plan; return fail(h, CTR_ERR_INVALID, "feat_offset must be non-decreasing"); }
    int bin;
    const bool constrained = (p->constraint_kind == CTR_CONS_DIMER && n == 2);
    const bool any_cons = (p->constraint_kind == CTR_CONS_DIMER && n == 2) ||
                          (p->constraint_kind == CTR_CONS_TRIMER && n == 3) ||
                          (p->constraint_kind == CTR_CONS_TETRAMER && n == 4);
    if (n > 0x3fffffLL) bin = BIN_TOO_LARGE;
    else if (n > MAXF) bin = BIN_LARGE;
    else if (default_modes && n == 1) bin = BIN_SMALL1;
    else if (default_modes && n == 2 && !constrained) bin = BIN_SMALL2;
    else {
      const int nv = n_vars(p, (int)n);
      const int nt = (nv + 1 + 15) / 16;
      bin = nt > MAXNT ? BIN_LARGE : (nt < 1 ? 0 : nt - 1);
      if (bin < MAXNT && n > (16 * (bin + 1) < MAXF ? 16 * (bin + 1) : MAXF)) bin = BIN_LARGE;
      if (any_cons && bin < 2) bin = bin == 0 ? BIN_CONS1 : BIN_CONS2;   // (at most 29 variables)
      else if (any_cons && bin < MAXNT) bin = BIN_TOO_LARGE;             // (ring / disc with every column free)
    }
    if (bin == BIN_LARGE) {
      // the large kernel's 16-column row: [r, shared.., own.., r_o, shared_o..]
      bool wide_mask = false;   // (its list of mask pixels packs box coordinates in 10 bits per axis)
      for (int a = 0; a < p->ndim; ++a) wide_mask = wide_mask || p->radius[a] > 500;
      if (npf < 1 || 2 + 2 * nsh + npf > 16 || other_profile || wide_mask) bin = BIN_TOO_LARGE;
      else {
        ws_off[(size_t)c] = ws_total;
        ws_total += large_ws((int)n, npf, nsh, box_cap).total;
      }
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
        hipMalloc((void**)&plan->d_order, sizeof(int32_t) * 2 * (size_t)n_clusters) != hipSuccess ||
        hipMalloc((void**)&plan->d_front, sizeof(int) * 2 * NBINS) != hipSuccess) {
      ctr_plan_destroy(plan);
      return fail(h, CTR_ERR_NOMEM, "cannot allocate the plan on the device");
    }
    if (hipMemcpy(plan->d_order, order.data(), sizeof(int32_t) * (size_t)n_clusters, hipMemcpyHostToDevice) != hipSuccess) {
      ctr_plan_destroy(plan);
      return fail(h, CTR_ERR_DEVICE, "cannot upload the plan");
    }
    if (plan->lowpass) {
      std::vector<double> taps(3 * LP_STRIDE, 0.);
      for (int a = 0; a < p->ndim; ++a) {
        if (p->noise_size[a] > 0.) plan->lp_half[a] = gaussian_taps(p->noise_size[a], taps.data() + a * LP_STRIDE);
        else taps[(size_t)a * LP_STRIDE] = 1.;   // this axis is not filtered
      }
      if (hipMalloc((void**)&plan->d_lp_w, sizeof(double) * taps.size()) != hipSuccess ||
          hipMemcpy(plan->d_lp_w, taps.data(), sizeof(double) * taps.size(), hipMemcpyHostToDevice) != hipSuccess) {
        ctr_plan_destroy(plan);
        return fail(h, CTR_ERR_NOMEM, "cannot upload the lowpass taps");
      }
    }
    if (ws_total > 0) {
      if (hipMalloc((void**)&plan->d_ws, sizeof(double) * (size_t)ws_total) != hipSuccess ||
          hipMalloc((void**)&plan->d_ws_off, sizeof(long long) * (size_t)n_clusters) != hipSuccess) {
        ctr_plan_destroy(plan);
        return fail(h, CTR_ERR_NOMEM, "cannot allocate the workspace of the large clusters on the device");
      }
      // (the leader / helper words of every cluster start at zero: no epoch matches)
      if (hipMemset(plan->d_ws, 0, sizeof(double) * (size_t)ws_total) != hipSuccess ||
          hipMemcpy(plan->d_ws_off, ws_off.data(), sizeof(long long) * (size_t)n_clusters, hipMemcpyHostToDevice) != hipSuccess) {
        ctr_plan_destroy(plan);
        return fail(h, CTR_ERR_DEVICE, "cannot upload the plan");
      }
    }
  }
  *out = plan;
  return CTR_OK;
}

void ctr_plan_destroy(ctr_plan* plan) {
  if (!plan) return;
  if (plan->d_order) { (void)hipSetDevice(plan->device); (void)hipFree(plan->d_order); }
  if (plan->d_front) (void)hipFree(plan->d_front);
  if (plan->d_ws) (void)hipFree(plan->d_ws);
  if (plan->d_ws_off) (void)hipFree(plan->d_ws_off);
  if (plan->d_lp_w) (void)hipFree(plan->d_lp_w);
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
  // The handle owns scratch that a call uses from start to end (frame maxima, work counters,
  // side streams): a call on another stream waits for the previous call of this handle.
  HIP_TRY(h, hipStreamWaitEvent(s, h->ev_done, 0));
  HIP_TRY(h, hipEventRecord(h->ev[0], s));
  int rc = ensure_fmax(h, b->n_frames > 0 ? b->n_frames : 1);
  if (rc) return rc;
  rc = launch_frame_max(h, b->frames, b->frame_dtype, b->n_frames, frame_elems, h->d_fmax, s);
  if (rc) return rc;
  HIP_TRY(h, hipEventRecord(h->ev[1], s));

  KArgs k;
  std::memset(&k, 0, sizeof k);   // (split = nullptr: a launch takes its whole bin)
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
  k.params_std = b->params_std;
  k.fmax = h->d_fmax;
  k.lp_w = plan->lowpass ? plan->d_lp_w : nullptr;
  k.cg_tol2_far = 1e-12;   // (with the aggregated preconditioner 1e-8 saves 12 % and costs parity: 6e-7 vs 5e-8 px)
  k.cg_tol2_near = 1e-22;
  if (const char* e = std::getenv("CTR_LARGE_CG_TOL2")) {   // (measurements: "far,near")
    double a = 0., b = 0.;
    if (std::sscanf(e, "%lf,%lf", &a, &b) == 2 && a > 0. && b > 0.) { k.cg_tol2_far = a; k.cg_tol2_near = b; }
  }
  for (int a = 0; a < 3; ++a) k.lp_half[a] = plan->lp_half[a];
  const int di = p.ndim == 3 ? 1 : 0, ii = p.isotropic ? 1 : 0;
  // The bins are independent: the big bin of singles runs on the caller's
  // stream, the others on side streams forked from / joined to it by events, so
  // that a few slow many-feature clusters overlap with the bulk.
  // this call's order inside the bins: likely stragglers first (front_load_kernel)
  int32_t* ord = plan->d_order + plan->n_clusters;
  {
    FrontArgs fa;
    fa.nbins = NBINS;
    fa.keep_bin = BIN_SMALL1;
    for (int bin = 0; bin < NBINS; ++bin) { fa.begin[bin] = (int)plan->bin_begin[bin]; fa.count[bin] = (int)plan->bin_count[bin]; }
    HIP_TRY(h, hipMemsetAsync(plan->d_front, 0, sizeof(int) * 2 * NBINS, s));
    // the work counters of the small-kernel launches, here rather than in their streams: a
    // fill kernel queued behind the gate would wait for a slot on a machine already flooded
    HIP_TRY(h, hipMemsetAsync(h->d_counter, 0, sizeof(int) * 8, s));
    hipLaunchKernelGGL(front_load_kernel, dim3((unsigned)((plan->n_clusters + 255) / 256)), dim3(256), 0, s, k, fa,
                       plan->d_order, ord, plan->d_front, (int)plan->n_clusters);
  }
  HIP_TRY(h, hipEventRecord(h->ev_fork, s));
  bool used[NSIDE] = {};
  int next_side = 0;
  // Which side stream a kernel takes.  Default: round robin in launch order.  With
  // CTR_FLAG_ISOLATE_TAIL the tier of the likely slow fits (the 64-lane pairs kernel, the
  // large-cluster kernel) has side stream 1 to itself and the others alternate between 0 and 2:
  // with 4 streams per handle on 2 hardware queues per handle, stream k of a handle shares its
  // queue with stream 3 - k of another handle (0 = main), so a long kernel on side 1 holds up
  // that handle's side-0 kernels -- not its main stream, on which its next batch begins (a slow
  // kernel in front of another handle's main stream delays that handle's whole next batch, slow
  // kernel included: the delays chain).
  const bool isolate = (p.flags & CTR_FLAG_ISOLATE_TAIL) != 0;
  bool slow_tier = false;
  auto pick_stream = [&](bool main_stream) -> hipStream_t {
    if (main_stream) return s;
    int j;
    if (isolate) j = slow_tier ? 1 : (next_side++ % 2 == 0 ? 0 : 2);
    else j = next_side++ % NSIDE;
    if (!used[j]) { used[j] = true; (void)hipStreamWaitEvent(h->side[j], h->ev_fork, 0); }
    return h->side[j];
  };
  // generic bins first (largest problems first) so that their tails start early
  for (int bin = MAXNT - 1; bin >= 0; --bin) {
    const int64_t cnt = plan->bin_count[bin];
    if (cnt == 0) continue;
    // (2D only: a 3D window has thousands of pixels, more wavefronts per cluster pay there)
    const bool tp = (p.flags & CTR_FLAG_THROUGHPUT) != 0 && p.ndim == 2;
    const bool lpk = plan->lowpass;
    const int fi = p.fit_function - 1;   // >= 0: ring / disc / inv_series
    kernel_fn fn = fi >= 0 ? h->fit[fi][di][ii][bin].fn : lpk ? h->lp[di][ii][bin].fn : (tp ? h->table_tp[di][ii][bin] : h->table[di][ii][bin]);
    const size_t bytes = fi >= 0 ? h->fit[fi][di][ii][bin].smem : lpk ? h->lp[di][ii][bin].smem : (tp ? h->smem_bytes_tp[di][ii][bin] : h->smem_bytes[di][ii][bin]);
    const int threads = fi >= 0 ? h->fit[fi][di][ii][bin].threads : lpk ? h->lp[di][ii][bin].threads : (tp ? h->block_threads_tp[di][ii][bin] : h->block_threads[di][ii][bin]);
    bool& attr = fi >= 0 ? h->fit_attr[fi][di][ii][bin] : lpk ? h->lp_attr[di][ii][bin] : (tp ? h->attr_set_tp[di][ii][bin] : h->attr_set[di][ii][bin]);
    if (!attr) {
      HIP_TRY(h, hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
      attr = true;
    }
    k.order = ord + plan->bin_begin[bin];
    k.n_bin = (int32_t)cnt;
    {
      void* kargs[] = {(void*)&k};
      HIP_TRY(h, hipLaunchKernel(fn, dim3((unsigned)cnt), dim3((unsigned)threads), kargs, bytes, pick_stream(false)));
    }
  }
  for (int cb = 1; cb >= 0; --cb) {
    const int bin = cb == 0 ? BIN_CONS1 : BIN_CONS2;
    const int64_t cnt = plan->bin_count[bin];
    if (cnt == 0) continue;
    const int tp = ((p.flags & CTR_FLAG_THROUGHPUT) != 0 && p.ndim == 2) ? 1 : 0;
    const int fi = p.fit_function - 1;   // >= 0: ring / disc / inv_series
    const KernelInfo& ki = fi >= 0 ? h->fit_cons[fi][di][ii][cb] : plan->lowpass ? h->lp_cons[di][ii][cb] : h->cons[di][ii][tp][cb];
    bool& cattr = fi >= 0 ? h->fit_cons_attr[fi][di][ii][cb] : plan->lowpass ? h->lp_cons_attr[di][ii][cb] : h->cons_attr[di][ii][tp][cb];
    if (!cattr) {
      HIP_TRY(h, hipFuncSetAttribute(ki.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ki.smem));
      cattr = true;
    }
    k.order = ord + plan->bin_begin[bin];
    k.n_bin = (int32_t)cnt;
    void* kargs[] = {(void*)&k};
    HIP_TRY(h, hipLaunchKernel(ki.fn, dim3((unsigned)cnt), dim3((unsigned)ki.threads), kargs, ki.smem, pick_stream(false)));
  }
  if (plan->bin_count[BIN_LARGE] > 0) {
    // a leader workgroup of 512 threads per cluster + helpers for its pixel passes
    // (large_kernel.h): as many as fill the machine, none when the clusters alone do
    const int64_t cnt = plan->bin_count[BIN_LARGE];
    const int lpi = plan->lowpass ? 1 : 0;
    const KernelInfo& ki = h->large[di][ii][lpi];
    if (!h->large_attr[di][ii][lpi]) {
      HIP_TRY(h, hipFuncSetAttribute(ki.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ki.smem));
      h->large_attr[di][ii][lpi] = true;
    }
    int64_t per_cluster = 512 / cnt;               // workgroups per cluster, leader included
    per_cluster = per_cluster < 1 ? 1 : (per_cluster > 8 ? 8 : per_cluster);
    // (several batches in flight fill the machine by themselves: a waiting helper would only hold
    //  the compute unit that another batch's leader needs)
    if ((p.flags & CTR_FLAG_THROUGHPUT) != 0) per_cluster = 1;
    if (const char* e = std::getenv("CTR_LARGE_WORKGROUPS")) {   // (measurements)
      const long v = std::strtol(e, nullptr, 10);
      if (v >= 1 && v <= 16) per_cluster = v;
    }
    k.order = ord + plan->bin_begin[BIN_LARGE];
    k.n_bin = (int32_t)cnt;
    double* wsp = plan->d_ws;
    const long long* wso = plan->d_ws_off;
    int epoch = ++plan->large_epoch;
    void* kargs[] = {(void*)&k, (void*)&wsp, (void*)&wso, (void*)&epoch};
    slow_tier = true;
    hipStream_t sl = pick_stream(false);
    slow_tier = false;
    HIP_TRY(h, hipLaunchKernel(ki.fn, dim3((unsigned)(cnt * per_cluster)), dim3((unsigned)ki.threads), kargs, ki.smem, sl));
  }
  if (plan->bin_count[BIN_TOO_LARGE] > 0) {
    const int64_t cnt = plan->bin_count[BIN_TOO_LARGE];
    k.order = ord + plan->bin_begin[BIN_TOO_LARGE];
    k.n_bin = (int32_t)cnt;
    hipLaunchKernelGGL(mark_kernel, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, pick_stream(false), k,
                       (int)CTR_STATUS_TOO_LARGE);
  }
  // the small kernels start a little later than the block kernels (see delay_kernel)
  bool gate = false;
  for (int bin = 0; bin < MAXNT; ++bin) gate = gate || plan->bin_count[bin] > 0;
  gate = gate || plan->bin_count[BIN_LARGE] > 0 || plan->bin_count[BIN_CONS1] > 0 || plan->bin_count[BIN_CONS2] > 0;
  if (gate) {
    hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(WAVE), 0, s, (unsigned long long)GATE_US * 100ull);
    HIP_TRY(h, hipEventRecord(h->ev_gate, s));
  }
  for (int nf = 2; nf >= 1; --nf) {
    const int bin = nf == 1 ? BIN_SMALL1 : BIN_SMALL2;
    const int64_t cnt = plan->bin_count[bin];
    if (cnt == 0) continue;
    slow_tier = nf == 2;
    hipStream_t st = pick_stream(nf == 1);
    slow_tier = false;
    if (gate && nf != 1) (void)hipStreamWaitEvent(st, h->ev_gate, 0);
    int* counter = h->d_counter + nf;
    k.order = ord + plan->bin_begin[bin];
    k.n_bin = (int32_t)cnt;
    // lanes per cluster by the size of a single-feature window: 8 (eight clusters per
    // wavefront: the solve, replicated in the lanes of a group, is shared by as many) while
    // a window is a few passes, 64 once it is thousands of pixels (3D)
    int64_t vol = 1;
    for (int a = 0; a < p.ndim; ++a) vol *= 2 * (int64_t)p.radius[a] + 1;
    const bool wide = nf == 2 || vol > 600;
    small_fn fn = nf == 2 ? h->small_table[di][ii][1] : (wide ? h->small_wide1[di][ii] : h->small_table[di][ii][0]);
    int64_t waves = wide ? cnt : (cnt + 7) / 8;
    if (nf == 2 && vol <= 600 && cnt >= 64 && (p.flags & CTR_FLAG_THROUGHPUT) != 0) {
      // Pairs in two tiers (CTR_FLAG_THROUGHPUT: +9 % with four batches in flight, but the
      // one-batch-at-a-time step gets 12 % longer).  One wavefront per pair gives the shortest iteration, which is
      // what the slow fits need; for the rest, four pairs per wavefront cost a third of the
      // machine time.  front_load_kernel has put the pairs closer than a quarter of the mask
      // radius first and counted them (on the device): the 64-lane kernel takes exactly
      // those, the 16-lane kernel, on a stream of its own, the others -- which kernel fits a
      // pair depends on the pair alone, not on the order of the batch.
      k.split = plan->d_front + 2 * bin;
      hipStream_t sb = pick_stream(false);
      if (gate) (void)hipStreamWaitEvent(sb, h->ev_gate, 0);
      // (a little later still than the first tier, whose few wavefronts must not queue up
      // behind the thousands of this one)
      hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(WAVE), 0, sb, (unsigned long long)GATE_US * 100ull);
      int* cbulk = h->d_counter + 3;
      k.split_part = 2;
      int64_t wb = (cnt + 3) / 4;
      if (wb > 8192) wb = 8192;
      {
        void* kargs[] = {(void*)&k, (void*)&cbulk};
        HIP_TRY(h, hipLaunchKernel(h->small_bulk2[di][ii], dim3((unsigned)wb), dim3(WAVE), kargs, 0, sb));
      }
      k.split_part = 1;
      if (waves > 2048) waves = 2048;   // persistent: a wavefront takes pair after pair
    }
    if (waves > 8192) waves = 8192;
    {
      void* kargs[] = {(void*)&k, (void*)&counter};
      HIP_TRY(h, hipLaunchKernel(fn, dim3((unsigned)waves), dim3(WAVE), kargs, 0, st));
    }
    k.split = nullptr;
    k.split_part = 0;
  }
  for (int j = 0; j < NSIDE; ++j)
    if (used[j]) {
      HIP_TRY(h, hipEventRecord(h->ev_join[j], h->side[j]));
      HIP_TRY(h, hipStreamWaitEvent(s, h->ev_join[j], 0));
    }
  if (b->result_rows != nullptr && b->n_features > 0)
    hipLaunchKernelGGL(result_rows_kernel, dim3((unsigned)((b->n_features + 255) / 256)), dim3(256), 0, s,
                       b->params_out, b->cost, b->feat_offset, (int)b->n_clusters, (int)b->n_features,
                       (int)p.n_params, b->result_rows);
  if (b->done_flag != nullptr)
    hipLaunchKernelGGL(done_flag_kernel, dim3(1), dim3(WAVE), 0, s, b->done_flag, b->done_value);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipEventRecord(h->ev[2], s));
  HIP_TRY(h, hipEventRecord(h->ev_done, s));
  h->ev_valid = true;
  return CTR_OK;
}



// the exported blob: the HIP IPC handle, then the PCI bus id of the device that owns the block
// (NUL-terminated) so that an importer can identify the owner BEFORE it maps or touches anything
constexpr size_t IPC_BUSID_OFF = sizeof(hipIpcMemHandle_t);
static_assert(IPC_BUSID_OFF + 32 <= CTR_IPC_HANDLE_BYTES, "IPC handle blob size");

int ctr_ipc_alloc(ctr_handle* h, int64_t bytes, void** dev_ptr, unsigned char* handle_out) {
  if (!h || !dev_ptr || !handle_out || bytes <= 0) return CTR_ERR_INVALID;
  *dev_ptr = nullptr;
  HIP_TRY(h, hipSetDevice(h->device));
  char busid[32] = {0};
  if (hipDeviceGetPCIBusId(busid, (int)sizeof busid, h->device) != hipSuccess || busid[0] == 0) {
    (void)hipGetLastError();
    return fail(h, CTR_ERR_DEVICE, "cannot read the PCI bus id of this device (needed to export an inbox)");
  }
  void* p = nullptr;
  if (hipMalloc(&p, (size_t)bytes) != hipSuccess) return fail(h, CTR_ERR_NOMEM, "cannot allocate the inbox");
  hipIpcMemHandle_t ih;
  hipError_t e = hipMemset(p, 0, (size_t)bytes);
  if (e == hipSuccess) e = hipDeviceSynchronize();
  if (e == hipSuccess) e = hipIpcGetMemHandle(&ih, p);
  if (e != hipSuccess) { (void)hipFree(p); return fail(h, CTR_ERR_DEVICE, std::string("hipIpcGetMemHandle: ") + hipGetErrorString(e)); }
  std::memset(handle_out, 0, CTR_IPC_HANDLE_BYTES);
  std::memcpy(handle_out, &ih, sizeof ih);
  std::memcpy(handle_out + IPC_BUSID_OFF, busid, sizeof busid - 1);
  *dev_ptr = p;
  return CTR_OK;
}

int ctr_ipc_open(ctr_handle* h, const unsigned char* handle, void** dev_ptr) {
  if (!h || !handle || !dev_ptr) return CTR_ERR_INVALID;
  *dev_ptr = nullptr;
  HIP_TRY(h, hipSetDevice(h->device));
  // Who owns the block?  A kernel of THIS device will store there, and a store that faults can
  // take the GPUs of the node down: the owner must be this device or one it has peer access to,
  // established before anything is mapped.  An owner this process cannot resolve (not visible,
  // malformed blob) counts as "no access": the caller falls back to the collective.
  char busid[32];
  std::memcpy(busid, handle + IPC_BUSID_OFF, sizeof busid);
  busid[sizeof busid - 1] = 0;
  int owner = -1;
  if (busid[0] == 0 || hipDeviceGetByPCIBusId(&owner, busid) != hipSuccess || owner < 0) {
    (void)hipGetLastError();
    return fail(h, CTR_ERR_DEVICE, "the device that owns the block is not visible to this process");
  }
  if (owner != h->device) {
    int can = 0;
    if (hipDeviceCanAccessPeer(&can, h->device, owner) != hipSuccess || !can) {
      (void)hipGetLastError();
      return fail(h, CTR_ERR_DEVICE, "no peer access from this device to the device that owns the block");
    }
  }
  hipIpcMemHandle_t ih;
  std::memcpy(&ih, handle, sizeof ih);
  void* p = nullptr;
  const hipError_t e = hipIpcOpenMemHandle(&p, ih, hipIpcMemLazyEnablePeerAccess);
  if (e != hipSuccess) return fail(h, CTR_ERR_DEVICE, std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e));
  // cross-check where the runtime can tell: the mapped pointer belongs to that owner (or is
  // reported under the device that mapped it -- runtimes differ), never to a third device
  hipPointerAttribute_t attr;
  if (hipPointerGetAttributes(&attr, p) == hipSuccess) {
    if (attr.device != owner && attr.device != h->device) {
      (void)hipIpcCloseMemHandle(p);
      return fail(h, CTR_ERR_DEVICE, "the mapped block does not belong to the device named in its handle");
    }
  } else {
    (void)hipGetLastError();
  }
  *dev_ptr = p;
  return CTR_OK;
}

int ctr_ipc_probe(ctr_handle* h, void* dst, int64_t value) {
  if (!h || !dst) return CTR_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  hipLaunchKernelGGL(done_flag_kernel, dim3(1), dim3(WAVE), 0, h->stream, (int64_t*)dst, value);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return CTR_OK;
}

int ctr_ipc_read(ctr_handle* h, void* dst_host, const void* src, int64_t bytes) {
  if (!h || !dst_host || !src || bytes < 0) return CTR_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipMemcpy(dst_host, src, (size_t)bytes, hipMemcpyDeviceToHost));
  return CTR_OK;
}

int ctr_ipc_close(ctr_handle* h, void* dev_ptr) {
  if (!h || !dev_ptr) return CTR_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipIpcCloseMemHandle(dev_ptr));
  return CTR_OK;
}

int ctr_ipc_free(ctr_handle* h, void* dev_ptr) {
  if (!h || !dev_ptr) return CTR_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipFree(dev_ptr));
  return CTR_OK;
}

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

int ctr_draw_frames_device(ctr_handle* h, const ctr_synth* sy, void* frames_out, void* hip_stream) {
  if (!h || !sy || !frames_out) return CTR_ERR_INVALID;
  if (sy->ndim != 2 && sy->ndim != 3) return fail(h, CTR_ERR_INVALID, "ndim must be 2 or 3");
  if (sy->frame_dtype != CTR_DTYPE_U8 && sy->frame_dtype != CTR_DTYPE_U16)
    return fail(h, CTR_ERR_UNSUPPORTED, "frames are drawn as uint8 or uint16 (integer wrap-around is part of the rule)");
  if (sy->n_frames < 0 || sy->n_features < 0) return fail(h, CTR_ERR_INVALID, "negative counts");
  if (!(sy->noise >= 0.)) return fail(h, CTR_ERR_INVALID, "noise must be >= 0");
  long long felems = 1;
  for (int a = 0; a < sy->ndim; ++a) {
    if (sy->shape[a] < 1) return fail(h, CTR_ERR_INVALID, "frame shape must be positive");
    felems *= sy->shape[a];
  }
  const long long total = felems * sy->n_frames;
  if (total == 0) return CTR_OK;
  if (sy->n_features > 0 && (!sy->frame_of || !sy->pos || !sy->size || !sy->max_value))
    return fail(h, CTR_ERR_INVALID, "null feature table");
  if (sy->n_features > 0x7fffffffLL) return fail(h, CTR_ERR_INVALID, "too many features for one launch");
  HIP_TRY(h, hipSetDevice(h->device));
  hipStream_t s = hip_stream ? (hipStream_t)hip_stream : h->stream;
  int* acc = nullptr;
  HIP_TRY(h, hipMallocAsync((void**)&acc, sizeof(int) * (size_t)total, s));
  HIP_TRY(h, hipMemsetAsync(acc, 0, sizeof(int) * (size_t)total, s));
  SynthArgs a;
  a.ndim = sy->ndim;
  for (int q = 0; q < 3; ++q) a.shape[q] = q < sy->ndim ? (long)sy->shape[q] : 1;
  a.frame_elems = (long)felems;
  a.n_features = (long)sy->n_features;
  a.frame_of = sy->frame_of;
  a.pos = sy->pos;
  a.size = sy->size;
  a.max_value = sy->max_value;
  a.acc = acc;
  if (sy->n_features > 0) {
    if (sy->ndim == 2) hipLaunchKernelGGL(draw_features_kernel<2>, dim3((unsigned)sy->n_features), dim3(SYN_THREADS), 0, s, a);
    else hipLaunchKernelGGL(draw_features_kernel<3>, dim3((unsigned)sy->n_features), dim3(SYN_THREADS), 0, s, a);
  }
  const unsigned blocks = (unsigned)((total + 255) / 256);
  if (sy->frame_dtype == CTR_DTYPE_U8)
    hipLaunchKernelGGL(finish_frames_kernel<uint8_t>, dim3(blocks), dim3(256), 0, s, acc, (uint8_t*)frames_out,
                       (long)total, sy->noise, (unsigned long long)sy->seed, 255L);
  else
    hipLaunchKernelGGL(finish_frames_kernel<uint16_t>, dim3(blocks), dim3(256), 0, s, acc, (uint16_t*)frames_out,
                       (long)total, sy->noise, (unsigned long long)sy->seed, 65535L);
  HIP_TRY(h, hipGetLastError());
  HIP_TRY(h, hipFreeAsync(acc, s));
  return CTR_OK;
}

int ctr_synchronize(ctr_handle* h, void* hip_stream) {
  if (!h) return CTR_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipStreamSynchronize(hip_stream ? (hipStream_t)hip_stream : h->stream));
  return CTR_OK;
}

int ctr_query_done(ctr_handle* h) {
  if (!h) return -1;
  if (hipSetDevice(h->device) != hipSuccess) return -1;
  const hipError_t e = hipEventQuery(h->ev_done);
  if (e == hipSuccess) return 1;
  if (e == hipErrorNotReady) return 0;
  h->err = std::string("hipEventQuery: ") + hipGetErrorString(e);
  return -1;
}

int ctr_engine_wait_stream(ctr_handle* h, void* hip_stream) {
  if (!h) return CTR_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipEventRecord(h->ev_order, (hipStream_t)hip_stream));
  HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_order, 0));
  return CTR_OK;
}

int ctr_stream_wait_engine(ctr_handle* h, void* hip_stream) {
  if (!h) return CTR_ERR_INVALID;
  HIP_TRY(h, hipSetDevice(h->device));
  HIP_TRY(h, hipEventRecord(h->ev_order, h->stream));
  HIP_TRY(h, hipStreamWaitEvent((hipStream_t)hip_stream, h->ev_order, 0));
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
  const size_t total = sz_frames + sz_fi + sz_fo + (b->params_std ? 5 : 4) * sz_par + sz_c + 3 * sz_ci;
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
  double* d_std = b->params_std ? (double*)take(sz_par) : nullptr;
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
  d.params_std = d_std;
  d.result_rows = nullptr;   // (the host-buffer call returns the tables themselves)
  d.done_flag = nullptr;
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
    if (e == hipSuccess && d_std) e = hipMemcpyAsync(b->params_std, d_std, sizeof(double) * (size_t)N * np, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) rc = fail(h, CTR_ERR_DEVICE, std::string("copy back / kernel execution: ") + hipGetErrorString(e));
  } else {
    (void)hipStreamSynchronize(s);
  }
  ctr_plan_destroy(plan);
  return rc;
}

}  // extern "C"
