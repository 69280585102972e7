// aux_kernels.h -- mark_kernel, find_clusters_kernel, frame_max_kernel
// Part of the MI355X cluster-refinement engine; included by ctrefine.hip inside its
// anonymous namespace (device code only, gfx950).
#ifndef CTREFINE_AUX_KERNELS_H
#define CTREFINE_AUX_KERNELS_H

// Order of the clusters inside every bin for one call: clusters with two features closer
// than a quarter of the mask radius (scaled distance, as find.py:72-93 scales by the
// separation) and clusters of more than 8 features go to the front of their bin.  Those
// are the fits that take tens of iterations and several re-window rounds; a bin lasts
// as long as its slowest cluster, so they must start first.  The results do not depend on
// the order.
struct FrontArgs { int begin[16]; int count[16]; int nbins; int keep_bin; /* bin copied as it is */ };

__global__ void front_load_kernel(const KArgs k, const FrontArgs fa, const int* __restrict__ order_in,
                                  int* __restrict__ order_out, int* __restrict__ counters, int total) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int lane = threadIdx.x & 63;
  int b = -1, cl = 0;
  bool close = false;
  if (t < total) {
    b = 0;
    for (int j = 1; j < fa.nbins; ++j)
      if (t >= fa.begin[j]) b = j;
    cl = order_in[t];
    if (b == fa.keep_bin) {   // the bulk of singles keeps its (frame) order
      order_out[t] = cl;
      b = -1;
    } else {
      const int f0 = k.feat_offset[cl], n = k.feat_offset[cl + 1] - f0;
      const int np = k.prob.n_params, nd = k.prob.ndim;
      close = n > 8;
      for (int i = 0; i < n && !close; ++i)
        for (int j = i + 1; j < n && !close; ++j) {
          double d2 = 0.;
          for (int a = 0; a < nd; ++a) {
            const double d = (k.params[(size_t)(f0 + i) * np + 2 + a] - k.params[(size_t)(f0 + j) * np + 2 + a]) /
                             (double)k.prob.radius[a];
            d2 += d * d;
          }
          if (d2 < 0.0625) close = true;
        }
    }
  }
  // one pair of atomics per wave and bin instead of one per cluster (all on two addresses)
  unsigned long long todo = __ballot(b >= 0);
  while (todo != 0ull) {
    const int leader = __builtin_ctzll(todo);
    const int lb = __shfl(b, leader);
    const unsigned long long same = __ballot(b == lb);
    const unsigned long long mc = __ballot(b == lb && close), mf = same & ~mc;
    int basec = 0, basef = 0;
    if (lane == leader) {
      basec = atomicAdd(&counters[2 * lb], __popcll(mc));
      basef = atomicAdd(&counters[2 * lb + 1], __popcll(mf));
    }
    basec = __shfl(basec, leader);
    basef = __shfl(basef, leader);
    if (b == lb) {
      const unsigned long long below = (1ull << lane) - 1ull;
      const int pos = close ? basec + __popcll(mc & below) : fa.count[lb] - 1 - (basef + __popcll(mf & below));
      order_out[fa.begin[lb] + pos] = cl;
    }
    todo &= ~same;
  }
}

// Head start for the workgroups that need a large contiguous piece of LDS (block kernel):
// the small kernels allocate ~1 KB of LDS per wave all over every CU, and once they have
// flooded the machine a 70..150 KB request finds no contiguous room until they drain.
// One wave that waits `ticks` of the 100 MHz clock, ahead of the small kernels in their streams.
__global__ void delay_kernel(unsigned long long ticks) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
}

// ctr_batch.result_rows: [N, n_params + 1] = params_out and the cost of the row's cluster
__global__ void result_rows_kernel(const double* __restrict__ params_out, const double* __restrict__ cost,
                                   const int32_t* __restrict__ feat_offset, int n_clusters, int n_features,
                                   int np, double* __restrict__ rows) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_features) return;
  int lo = 0, hi = n_clusters;   // the cluster c with feat_offset[c] <= i < feat_offset[c + 1]
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (feat_offset[mid] <= i) lo = mid; else hi = mid;
  }
  for (int q = 0; q < np; ++q) rows[(size_t)i * (np + 1) + q] = params_out[(size_t)i * np + q];
  rows[(size_t)i * (np + 1) + np] = cost[lo];
}

// clusters the engine cannot take (too many variables / features)
__global__ void mark_kernel(const KArgs k, int code) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= k.n_bin) return;
  const int cl = k.order[t];
  const int np = k.prob.n_params;
  for (int e = k.feat_offset[cl] * np; e < k.feat_offset[cl + 1] * np; ++e) {
    k.params_out[e] = k.params[e];
    if (k.params_std) k.params_std[e] = NAN;
  }
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


// ctr_batch.done_flag: one store, queued behind everything else of the call
__global__ void done_flag_kernel(int64_t* flag, int64_t value) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    __threadfence_system();
    __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  }
}

#endif  // CTREFINE_AUX_KERNELS_H
