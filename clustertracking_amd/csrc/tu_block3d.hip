// tu_block3d.hip -- the 3D instantiations of refine_block_kernel (compiled on their own so
// that the engine builds in parallel); see block_kernel.h.
#include <cmath>

#include "kargs.h"

namespace {

#include "device_common.h"
#include "block_kernel.h"

constexpr size_t LDS_CU = 160 * 1024;  // LDS of one CU

template <int NT> struct WavesFor { static constexpr int value = NT <= 2 ? 8 : (NT <= 3 ? 4 : (NT <= 6 ? 2 : 1)); };
// CTR_FLAG_THROUGHPUT: fewest wavefronts per cluster (a quarter of the LDS and of the wave slots
// for 3-4 features; two or three workgroups per CU instead of one for 5-30 features)
template <int NT> struct WavesThroughput { static constexpr int value = NT <= 2 ? 2 : 1; };

template <bool ISO, int NT, bool TP, bool CONS = false>
KernelInfo one() {
  constexpr int W = TP ? WavesThroughput<NT>::value : WavesFor<NT>::value;
  static_assert(SmemB<NT, W, CONS>::bytes <= LDS_CU, "LDS budget of one CU");
  return KernelInfo{(const void*)refine_block_kernel<3, ISO, NT, W, CONS>, SmemB<NT, W, CONS>::bytes, WAVE * W};
}

template <bool ISO, bool TP>
KernelInfo by_nt(int nt, int cons) {
  if (cons) {   // constrained clusters have at most 4 features = 29 variables
    if (nt == 1) return one<ISO, 1, TP, true>();
    if (nt == 2) return one<ISO, 2, TP, true>();
    return KernelInfo{nullptr, 0, 0};
  }
  switch (nt) {
    case 1: return one<ISO, 1, TP>();
    case 2: return one<ISO, 2, TP>();
    case 3: return one<ISO, 3, TP>();
    case 4: return one<ISO, 4, TP>();
    case 5: return one<ISO, 5, TP>();
    case 6: return one<ISO, 6, TP>();
    case 7: return one<ISO, 7, TP>();
    case 8: return one<ISO, 8, TP>();
    default: return KernelInfo{nullptr, 0, 0};
  }
}

}  // namespace

KernelInfo ctr_block_kernel_3d(int iso, int nt, int throughput, int cons) {
  (void)throughput;   // (a 3D window has thousands of pixels: more wavefronts per cluster pay there)
  return iso ? by_nt<true, false>(nt, cons) : by_nt<false, false>(nt, cons);
}
