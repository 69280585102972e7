/* ctrefine.h -- C-ABI of the MI355X cluster-refinement engine.
 *
 * One batched call replaces the per-cluster Python loop of the reference
 * (clustertracking/refine.py:343-430): for every (frame, cluster) group it
 * cuts the pixel window (masks.py:30-68), builds the per-feature elliptical
 * masks (refine.py:28-58), and minimises the sum-of-Gaussians least-squares
 * objective (fitfunc.py:421-489) under the box bounds (fitfunc.py:535-558) and
 * optional dimer/trimer/tetramer equality constraints (constraints.py:59-137),
 * including the re-window rounds (refine.py:365-388) and the failure rules
 * (refine.py:33-34,356-357,376-377,391-394,408-418).
 *
 * The reference has no FFI of its own: these entry points are
 * what a ctypes binding in its refine.py would call between the host-side
 * setup (refine.py:242-341) and the DataFrame write-back (refine.py:419-430).
 * INTEGRATION.md shows that binding.
 *
 * Plain C types only; every buffer is caller-owned; calls are synchronous
 * unless stated; per-cluster failures are DATA (status + NaN cost), never an
 * error return (mirrors refine.py:408-418).
 */
#ifndef CTREFINE_H
#define CTREFINE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define CTR_ABI_VERSION 7
#define CTR_MAX_NDIM 3
#define CTR_MAX_PARAMS 12 /* background, signal, <=3 positions, <=3 sizes, profile parameters (ring, disc: 1;
                              inv_series_<N>: N + 1, as many as fit: N <= 6 in 2D isotropic, <= 3 in 3D anisotropic) */
#define CTR_MAX_VARS 127 /* optimiser variables per cluster of the on-chip kernels; larger clusters
                            (or more than 64 features) take the large-cluster path: normal matrix
                            block-sparse in HBM, no limit on features or variables */
#define CTR_MAX_NEIGHBOURS 48
#define CTR_MAX_NOISE_SIZE 4.0   /* lowpass sigma: at most 2 * 16 + 1 taps per axis */ /* large-cluster path: features whose mask ellipsoids overlap one feature's */

/* error codes (return values) */
enum {
  CTR_OK = 0,
  CTR_ERR_INVALID = 1,     /* malformed descriptor (reference: ValueError / AssertionError, refine.py:256-262,283) */
  CTR_ERR_UNSUPPORTED = 2, /* recognised but not implemented (mode 'global', a lowpass with a profile other than gauss) */
  CTR_ERR_DEVICE = 3,      /* HIP runtime failure, no MI355X visible, ... */
  CTR_ERR_NOMEM = 4
};

/* pixel types of the frame block */
enum {
  CTR_DTYPE_U8 = 0,
  CTR_DTYPE_U16 = 1,
  CTR_DTYPE_I16 = 2,
  CTR_DTYPE_I32 = 3,
  CTR_DTYPE_F32 = 4,
  CTR_DTYPE_F64 = 5
};

/* radial profile (fitfunc.py:112-154,195-204,334-343): GAUSS, RING ('thickness' column), DISC
 * ('disc_size' column) and INV_SERIES: 'inv_series_<N>' = signal_mult / (r2^N + param_a r2^(N-1) + ... +
 * param_<N>) (what fitfunc.py:148-154 evaluates: np.polyval with the leading coefficient set to 1),
 * N + 1 columns 'signal_mult', 'param_a', ...; N = n_params - (2 + ndim + sizes) - 1 */
enum { CTR_FIT_GAUSS = 0, CTR_FIT_RING = 1, CTR_FIT_DISC = 2, CTR_FIT_INV_SERIES = 3 };

/* parameter modes (fitfunc.py:9-11); 2 ('global') is rejected: it couples all
 * clusters into one problem (refine.py:319-332) and does not shard */
enum { CTR_MODE_CONST = 0, CTR_MODE_VAR = 1, CTR_MODE_GLOBAL = 2, CTR_MODE_CLUSTER = 3 };

/* equality constraints (constraints.py:59-137); applied only to clusters of
 * exactly 2 / 3 / 4 features (constraints.py:32-34) */
enum { CTR_CONS_NONE = 0, CTR_CONS_DIMER = 1, CTR_CONS_TRIMER = 2, CTR_CONS_TETRAMER = 3 };

/* ctr_problem.flags.  The first two are scheduling only: the results do not depend on them.
 * CTR_FLAG_THROUGHPUT: the caller keeps several batches in flight on one device (one handle
 * each); favour machine time per cluster over the latency of one batch -- pairs that are not
 * likely to be slow fits share a wavefront four at a time, larger 2D clusters run on the
 * fewest wavefronts.
 * CTR_FLAG_ISOLATE_TAIL: for batches that hold a fit far slower than the rest (e.g. two start
 * positions on one real feature: hundreds of iterations) while several batches are in flight:
 * the kernel that takes the likely slow fits gets a stream to itself whose hardware queue it
 * does not share with the main stream of another handle (a queue is served in order: a long
 * kernel in front of a main stream delays that handle's whole next batch).  Costs 10-15 % on
 * batches without such a fit, gains 30 % on batches with one; a caller can time
 * both.  Assumes 2 hardware queues per handle (GPU_MAX_HW_QUEUES = 2 x handles in flight).
 * CTR_FLAG_WINDOW_FILTER (NOT a scheduling flag): `noise_size` was given (refine.py:37 `if
 *   noise_size is not None`): the window goes through the reference's lowpass even where every
 *   sigma is 0 -- then that is its threshold alone, values <= threshold become 0
 *   (preprocessing.py:41-49).  Implied by any noise_size > 0. */
enum { CTR_FLAG_THROUGHPUT = 1, CTR_FLAG_ISOLATE_TAIL = 2, CTR_FLAG_WINDOW_FILTER = 4 };

/* per-cluster status */
enum {
  CTR_STATUS_OK = 0,
  CTR_STATUS_OUT_OF_BOUNDS = 1, /* no coordinate inside the frame (refine.py:33-34) or empty mask */
  CTR_STATUS_NONFINITE = 2,     /* non-finite initial parameters (refine.py:356-357) */
  CTR_STATUS_NO_CONVERGENCE = 3,/* solver failed / iteration limit (refine.py:376-377) */
  CTR_STATUS_RMS_DEV = 4,       /* rms deviation above max_rms_dev (refine.py:391-394) */
  CTR_STATUS_TOO_LARGE = 5      /* beyond the engine: a cluster of the large-cluster path in which a feature
                                   has more than CTR_MAX_NEIGHBOURS overlapping neighbours, or whose
                                   parameter modes leave no per-feature variable (reported as data) */
};

/* What is fitted and how (one per call). */
typedef struct ctr_problem {
  int32_t ndim;                    /* 2 or 3 */
  int32_t isotropic;               /* 1: one 'size' column; 0: one per axis */
  int32_t fit_function;            /* CTR_FIT_* */
  int32_t n_params;                /* 2 + ndim + (isotropic ? 1 : ndim) + extras; column order
                                      [background, signal, (z,) y, x, size | size_(z,)y,x, extra]
                                      (fitfunc.py:353-354); extras: 0 for gauss, 1 for ring
                                      ('thickness') and disc ('disc_size'), N + 1 for inv_series_<N> */
  int32_t modes[CTR_MAX_PARAMS];   /* CTR_MODE_* per column (fitfunc.py:394) */
  int32_t radius[CTR_MAX_NDIM];    /* mask radius per axis = diameter // 2 (refine.py:286) */
  int32_t constraint_kind;         /* CTR_CONS_* */
  int32_t max_iter;                /* re-window rounds, >= 1 (refine.py:365; default 10) */
  int32_t solver_maxiter;          /* solver iterations per round (refine.py:243; default 100) */
  int32_t flags;                   /* CTR_FLAG_*; 0 = defaults */
  double constraint_dist[CTR_MAX_NDIM]; /* per-axis distance of the constraint */
  double max_shift;                /* refine.py:384 (default 1) */
  double max_rms_dev;              /* refine.py:391 (default 1) */
  double residual_factor;          /* refine.py:354,379 (default 1e5) */
  double xtol;                     /* relative step tolerance; <= 0 -> 1e-9 */
  double ftol;                     /* relative model-decrease tolerance; <= 0 -> 1e-14 */
  double threshold;                /* lowpass: filtered values <= threshold become 0
                                      (refine.py:38-40, preprocessing.py:47; default 0) */
  double noise_size[CTR_MAX_NDIM]; /* refine.py:37-40: the window of every re-window round is
                                      correlated, axis by axis, with a normalised Gaussian of this
                                      sigma truncated at 4 sigma (preprocessing.py:12-49), zero
                                      beyond the WINDOW edges, before it is fitted; 0 = that axis
                                      is not filtered, all 0 = no lowpass.  At most
                                      CTR_MAX_NOISE_SIZE. */
} ctr_problem;

/* The data of one batch.  All arrays C-contiguous.  For ctr_refine_batch the
 * pointers are HOST pointers; for ctr_refine_batch_device they are DEVICE
 * pointers (hipMalloc'ed memory of the handle's device). */
typedef struct ctr_batch {
  const void* frames;          /* [n_frames, shape...] pixels, dtype frame_dtype */
  int32_t frame_dtype;         /* CTR_DTYPE_* */
  int32_t reserved0;
  int64_t n_frames;
  int64_t shape[CTR_MAX_NDIM]; /* frame shape (z,) y, x; first ndim entries used */
  int64_t n_clusters;          /* C */
  int64_t n_features;          /* N */
  const int32_t* frame_index;  /* [C] index into frames of each cluster */
  const int32_t* feat_offset;  /* [C+1] CSR offsets into the feature table */
  const double* params;        /* [N, n_params] initial parameters (refine.py:345) */
  const double* low;           /* [N, n_params] per-feature lower bounds (fitfunc.py:541-546), -inf = none */
  const double* high;          /* [N, n_params] per-feature upper bounds (fitfunc.py:547-550), +inf = none */
  double* params_out;          /* [N, n_params] refined; equals params for failed clusters */
  double* cost;                /* [C] rms residual / frame max (refine.py:379); NaN on failure */
  int32_t* status;             /* [C] CTR_STATUS_* */
  int32_t* n_rounds;           /* [C] re-window rounds used */
  int32_t* n_iter;             /* [C] solver iterations, summed over rounds */
  double* params_std;          /* [N, n_params] or NULL.  refine.py:400-406 (compute_error):
                                  sqrt(2 diag(inv(H))), H = Hessian of the objective at the
                                  solution over ALL variables of the cluster (bounds and
                                  constraints ignored, as there); here from the exact second
                                  derivatives instead of finite differences.  NaN for constant
                                  parameters, failed clusters and a Hessian that is not positive
                                  definite.  Every parameter mode (second derivatives w.r.t.
                                  signal, centres and sizes).  NaN for clusters of the
                                  large-cluster path (> 64 features) and for the ring / disc /
                                  inv_series profiles (the second derivatives are the gaussian's). */
  double* result_rows;         /* [N, n_params + 1] or NULL: params_out and, last column, the cost of
                                  the row's cluster -- the rows of the result table (refine.py:426-427)
                                  in one block, written when the batch is done: what a pipeline sends
                                  on (the multi-GPU gather) without another pass over the outputs.
                                  May be memory of ANOTHER device of the node that this device can
                                  write (peer / IPC-mapped: rank 0's inbox) -- the rows then travel
                                  over xGMI as they are written, no collective per batch */
  int64_t* done_flag;          /* NULL, or where ctr_refine_batch_device stores done_value (one
                                  8-byte store from a kernel of its own, queued last on the call's
                                  stream: after every output above is written); device or peer
                                  memory like result_rows: a consumer polls it per batch.  There is
                                  no flow control: a caller that reuses result_rows / done_flag for a
                                  later batch must have its consumer's acknowledgement first (or one
                                  slot per batch in flight, as bench.py's inbox) */
  int64_t done_value;
} ctr_batch;

typedef struct ctr_handle ctr_handle;
typedef struct ctr_plan ctr_plan;

/* ABI version of the loaded library (== CTR_ABI_VERSION). */
int ctr_abi_version(void);

/* Create / destroy an engine bound to one HIP device.  Owns streams and
 * scratch buffers (frame maxima, work counters).  One caller thread at a time per handle; the
 * *_device calls of one handle are ordered on the device (a call waits for the previous one of
 * the same handle, whatever streams they were given): to overlap batches use one handle each. */
int ctr_create(ctr_handle** out, int device);
void ctr_destroy(ctr_handle* h);

/* Last error text of a failed call on this handle (or of ctr_create when h is NULL). */
const char* ctr_last_error(const ctr_handle* h);

/* Validate a problem descriptor without touching a device. */
int ctr_validate_problem(const ctr_problem* p, char* msg, int msg_len);

/* Number of optimiser variables of a cluster with n features
 * (vect_from_params layout, fitfunc.py:207-263, groups=None). */
int ctr_cluster_n_vars(const ctr_problem* p, int n_features);

/* Host-pointer entry: copies the batch to the device, runs it, copies the
 * results back.  Synchronous.  Replaces refine.py:343-430. */
int ctr_refine_batch(ctr_handle* h, const ctr_problem* p, const ctr_batch* b);

/* Device-resident path (inputs already in HBM, e.g. frames produced on the
 * GPU): a plan bins the clusters of a batch by problem size on the host once (and owns the
 * HBM workspace of its large clusters: use a plan with the handle it was created on);
 * the run is asynchronous on the given HIP stream (hipStream_t passed as
 * void*; NULL = the handle's own stream). */
int ctr_plan_create(ctr_handle* h, const ctr_problem* p, int64_t n_clusters,
                    const int32_t* feat_offset_host, ctr_plan** out);
void ctr_plan_destroy(ctr_plan* plan);
int ctr_refine_batch_device(ctr_handle* h, const ctr_plan* plan,
                            const ctr_batch* b_device, void* hip_stream);

/* Per-frame maximum on the device (the norm of refine.py:354); part of
 * ctr_refine_batch_device, exposed for measurement.  out_max: [n_frames] f64. */
int ctr_frame_max_device(ctr_handle* h, const void* frames, int32_t frame_dtype,
                         int64_t n_frames, int64_t frame_elems, double* out_max,
                         void* hip_stream);

/* Cluster labelling on the device (replaces reference find.py:72-93, the step right
 * before the hot path): rows [frame_offset[f], frame_offset[f+1]) of `pos` are the
 * features of frame f (table sorted by frame); two features of a frame closer than
 * `separation` (per-axis scaled distance <= 1, as cKDTree(pos/separation).query_pairs(1))
 * share a cluster.  label_out[i] = smallest row index of i's cluster (canonical: the
 * reference's own ids depend on Python set order; the partition is the same),
 * size_out[i] = number of features in it.  Host pointers, synchronous. */
int ctr_find_clusters(ctr_handle* h, int32_t ndim, const double* pos, const int32_t* frame_offset,
                      int64_t n_frames, const double* separation, int32_t* label_out, int32_t* size_out);

/* Synthetic frames on the device (the generator next to the hot path):
 * the drawing rule of reference artificial.draw_feature for the Gaussian (artificial.py:131-141:
 * patch [max(floor(c - 4 size), 0), min(ceil(c + 4 size + 1), lim)) per axis,
 * spot = max_value exp(-ndim/2 sum(((idx - c) / size)^2)), TRUNCATED to the pixel type and added
 * with integer wrap-around) and the noise rule of SimulatedImage.noisy_image (:368-378: Poisson
 * noise per pixel, clipped to the pixel type's range).  The noise-free bytes equal the
 * reference rule's; the noise comes from the engine's own counter-based generator (seed, pixel
 * index): same statistics as NumPy's, not the same bytes.  A feature whose centre lies outside
 * its frame is skipped (the reference raises ValueError: check on the host).
 * Device pointers; asynchronous on `hip_stream` (NULL = the handle's stream). */
typedef struct ctr_synth {
  int32_t ndim;                /* 2 or 3 */
  int32_t frame_dtype;         /* CTR_DTYPE_U8 or CTR_DTYPE_U16 */
  int64_t n_frames;
  int64_t shape[CTR_MAX_NDIM]; /* (z,) y, x */
  int64_t n_features;
  const int32_t* frame_of;     /* [N] frame of each feature */
  const double* pos;           /* [N, ndim] centres (z,) y, x */
  const double* size;          /* [N, ndim] radius of gyration per axis (fitfunc.py:112-113) */
  const double* max_value;     /* [N] peak value */
  double noise;                /* Poisson level per pixel, 0 = none */
  uint64_t seed;               /* of the noise */
} ctr_synth;
int ctr_draw_frames_device(ctr_handle* h, const ctr_synth* s, void* frames_out, void* hip_stream);

/* Has the last ctr_refine_batch_device call of this handle finished on the device?  1 yes (also
 * when there was none), 0 still running, -1 error.  Never blocks: lets a pipeline that keeps
 * several batches in flight hand finished batches on (e.g. to the result gather) from the host
 * without parking a stream in a device-side wait. */
int ctr_query_done(ctr_handle* h);

/* Block until the work queued by the *_device calls on `hip_stream` is done. */
int ctr_synchronize(ctr_handle* h, void* hip_stream);

/* The inbox of a multi-GPU pipeline (one process per GPU): the rank that collects the result rows
 * allocates a block on its device and exports it; every other rank maps it and passes addresses
 * inside it as ctr_batch.result_rows / done_flag, so that the rows travel as peer stores over xGMI
 * while they are written -- no collective per batch (bench.py:open_inbox is the worked example).
 *   ctr_ipc_alloc: hipMalloc of `bytes` zeroed bytes on the handle's device + a blob of
 *     CTR_IPC_HANDLE_BYTES bytes to be sent to the other processes by any means: the HIP IPC
 *     handle followed by the PCI bus id of the owning device (ABI 6).
 *   ctr_ipc_open: maps such a block into this process for the handle's device
 *     (hipIpcOpenMemHandle with lazy peer access, called with that device current).  The owner
 *     named in the blob must be this device or one it has peer access to (hipDeviceCanAccessPeer),
 *     checked BEFORE anything is mapped; an owner this process cannot resolve is refused
 *     (CTR_ERR_DEVICE): the caller then gathers with a collective instead.
 *   ctr_ipc_probe: stores `value` at `dst` (8 bytes) FROM A KERNEL of the handle's device and
 *     waits for it: proves at set-up time that this device can write the mapped block.
 *   ctr_ipc_read: copies `bytes` from device / mapped memory to the host (blocking).
 *   ctr_ipc_close / ctr_ipc_free: unmap (importer) / release (owner, after every importer closed). */
#define CTR_IPC_HANDLE_BYTES 128
int ctr_ipc_alloc(ctr_handle* h, int64_t bytes, void** dev_ptr, unsigned char* handle_out);
int ctr_ipc_open(ctr_handle* h, const unsigned char* handle, void** dev_ptr);
int ctr_ipc_probe(ctr_handle* h, void* dst, int64_t value);
int ctr_ipc_read(ctr_handle* h, void* dst_host, const void* src, int64_t bytes);
int ctr_ipc_close(ctr_handle* h, void* dev_ptr);
int ctr_ipc_free(ctr_handle* h, void* dev_ptr);

/* Ordering with a caller's stream when the *_device calls run on the handle's own stream
 * (hip_stream = NULL above), without blocking the host.  `hip_stream` here is a hipStream_t
 * passed as void*; NULL means the legacy default stream (what PyTorch uses unless told
 * otherwise).
 *   ctr_engine_wait_stream: work queued later on the handle's stream starts only when
 *     everything queued so far on `hip_stream` has finished (inputs produced there).
 *   ctr_stream_wait_engine: work queued later on `hip_stream` starts only when everything
 *     queued so far on the handle's stream has finished (results consumed there). */
int ctr_engine_wait_stream(ctr_handle* h, void* hip_stream);
int ctr_stream_wait_engine(ctr_handle* h, void* hip_stream);

/* Timing of the kernels of the last ctr_refine_batch_device call, measured
 * with HIP events on the launch stream (milliseconds); synchronises. */
int ctr_last_kernel_ms(ctr_handle* h, double* frame_max_ms, double* refine_ms);

#ifdef __cplusplus
}
#endif
#endif /* CTREFINE_H */
