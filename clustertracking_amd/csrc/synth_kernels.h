// synth_kernels.h -- synthetic frames on the device
// Part of the MI355X cluster-refinement engine; included by ctrefine.hip inside its
// anonymous namespace (device code only, gfx950).
//
// Restates the drawing rule of the reference's artificial.draw_feature for the Gaussian
// (artificial.py:131-141) and the noise rule of SimulatedImage.noisy_image (:368-378):
//   patch per axis [max(floor(c - 4 size), 0), min(ceil(c + 4 size + 1), lim))
//   r2 = sum_a ((idx_a - c_a) / size_a)^2,  spot = max_value exp(r2 ndim / -2)
//   spot TRUNCATED to the pixel type and ADDED WITH INTEGER WRAP-AROUND
//   + Poisson(noise) per pixel, clipped to the pixel type's range
// Sums modulo 2^bits commute, so the features are added with 32-bit integer atomics into an
// accumulation block (any order gives the reference's bytes) and reduced modulo 2^bits when the
// noise is added.  The noise itself is this engine's own counter-based generator, not NumPy's:
// the same statistics, not the same bytes (clustertracking_amd/artificial.py on the host is
// the bit-exact counterpart for the noise-free part).
#ifndef CTREFINE_SYNTH_KERNELS_H
#define CTREFINE_SYNTH_KERNELS_H

struct SynthArgs {
  int ndim;
  long shape[3];          // (z,) y, x
  long frame_elems;
  long n_features;
  const int32_t* frame_of;
  const double* pos;      // [N, ndim]
  const double* size;     // [N, ndim]
  const double* max_value;// [N]
  int* acc;               // [n_frames * frame_elems]
};

constexpr int SYN_THREADS = 256;

template <int ND>
__global__ void __launch_bounds__(SYN_THREADS) draw_features_kernel(const SynthArgs a) {
#pragma clang fp contract(off)
  const long f = blockIdx.x;
  if (f >= a.n_features) return;
  int lo[ND], ext[ND];
  double c[ND], s[ND];
  long npx = 1;
  bool inside = true;
#pragma unroll
  for (int ax = 0; ax < ND; ++ax) {
    c[ax] = a.pos[f * ND + ax];
    s[ax] = a.size[f * ND + ax];
    const long lim = a.shape[ax];
    if (!(c[ax] >= 0.) || !(c[ax] < (double)lim)) inside = false;   // the reference raises ValueError
    long l = (long)floor(c[ax] - 4. * s[ax]), h = (long)ceil(c[ax] + 4. * s[ax] + 1.);
    l = l < 0 ? 0 : l;
    h = h > lim ? lim : h;
    lo[ax] = (int)l;
    ext[ax] = h > l ? (int)(h - l) : 0;
    npx *= ext[ax];
  }
  if (!inside || npx <= 0) return;
  const double mv = a.max_value[f];
  const double fac = (double)ND / -2.;
  int* acc = a.acc + (long)a.frame_of[f] * a.frame_elems;
  for (long q = threadIdx.x; q < npx; q += SYN_THREADS) {
    long t = q;
    int idx[ND];
#pragma unroll
    for (int ax = ND - 1; ax >= 0; --ax) {
      idx[ax] = lo[ax] + (int)(t % ext[ax]);
      t /= ext[ax];
    }
    double r2 = 0.;
    long off = 0;
#pragma unroll
    for (int ax = 0; ax < ND; ++ax) {
      const double u = ((double)idx[ax] - c[ax]) / s[ax];
      r2 = r2 + u * u;
      off = off * a.shape[ax] + idx[ax];
    }
    const double spot = mv * exp(r2 * fac);
    const int add = (int)spot;   // astype(integer dtype): truncation
    if (add != 0) atomicAdd(acc + off, add);
  }
}

__device__ __forceinline__ unsigned long long splitmix64(unsigned long long& x) {
  unsigned long long z = (x += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ double uniform01(unsigned long long& st) {
  return (double)(splitmix64(st) >> 11) * (1. / 9007199254740992.);
}

// Poisson(lambda): multiplication method (Knuth) up to 64, normal approximation beyond
__device__ __forceinline__ long poisson_draw(double lambda, double explam, unsigned long long& st) {
  if (lambda <= 64.) {
    long kq = 0;
    double p = uniform01(st);
    while (p > explam) { ++kq; p *= uniform01(st); }
    return kq;
  }
  const double u1 = uniform01(st), u2 = uniform01(st);
  const double g = sqrt(-2. * log(u1 > 1e-300 ? u1 : 1e-300)) * cos(6.283185307179586 * u2);
  const double x = floor(lambda + sqrt(lambda) * g + 0.5);
  return x < 0. ? 0 : (long)x;
}

// acc -> pixels: value modulo 2^bits (the wrap-around of the sequential integer adds), plus
// noise, clipped to [0, saturation]
template <typename T>
__global__ void finish_frames_kernel(const int* __restrict__ acc, T* __restrict__ out, long n,
                                     double noise, unsigned long long seed, long saturation) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  long v = (long)(T)(unsigned int)acc[i];
  if (noise > 0.) {
    unsigned long long st = seed ^ ((unsigned long long)i * 0xD1342543DE82EF95ull + 0x2545F4914F6CDD1Dull);
    (void)splitmix64(st);
    v += poisson_draw(noise, exp(-noise), st);
    v = v > saturation ? saturation : v;
  }
  out[i] = (T)v;
}

#endif  // CTREFINE_SYNTH_KERNELS_H
