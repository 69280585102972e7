// tu_block_fit2d.hip -- the 2D instantiations of refine_block_kernel for the ring and disc
// profiles (FIT = CTR_FIT_RING / CTR_FIT_DISC; fitfunc.py:121-146), see block_kernel.h.  One per
// (isotropic, NT, constrained, profile): these problems take the wavefront counts of the default
// scheduling and have no lowpass variant.
#include <cmath>

#include "kargs.h"

namespace {

#include "device_common.h"
#include "block_kernel.h"

constexpr size_t LDS_CU = 160 * 1024;  // LDS of one CU

template <int NT> struct WavesFor { static constexpr int value = NT <= 2 ? 8 : (NT <= 3 ? 4 : (NT <= 6 ? 2 : 1)); };

template <bool ISO, int NT, int FIT, bool CONS = false>
KernelInfo one() {
  constexpr int W = WavesFor<NT>::value;
  static_assert(SmemB<NT, W, CONS>::bytes <= LDS_CU, "LDS budget of one CU");
  return KernelInfo{(const void*)refine_block_kernel<2, ISO, NT, W, CONS, false, FIT>, SmemB<NT, W, CONS>::bytes, WAVE * W};
}

template <bool ISO, int FIT>
KernelInfo by_nt(int nt, int cons) {
  if (cons) {   // constrained clusters of at most 31 variables (beyond: status 5, ctrefine.hip)
    if (nt == 1) return one<ISO, 1, FIT, true>();
    if (nt == 2) return one<ISO, 2, FIT, true>();
    return KernelInfo{nullptr, 0, 0};
  }
  switch (nt) {
    case 1: return one<ISO, 1, FIT>();
    case 2: return one<ISO, 2, FIT>();
    case 3: return one<ISO, 3, FIT>();
    case 4: return one<ISO, 4, FIT>();
    case 5: return one<ISO, 5, FIT>();
    case 6: return one<ISO, 6, FIT>();
    case 7: return one<ISO, 7, FIT>();
    case 8: return one<ISO, 8, FIT>();
    default: return KernelInfo{nullptr, 0, 0};
  }
}

}  // namespace

KernelInfo ctr_block_kernel_fit2d(int iso, int nt, int cons, int fit) {
  if (fit == CTR_FIT_RING) return iso ? by_nt<true, CTR_FIT_RING>(nt, cons) : by_nt<false, CTR_FIT_RING>(nt, cons);
  if (fit == CTR_FIT_DISC) return iso ? by_nt<true, CTR_FIT_DISC>(nt, cons) : by_nt<false, CTR_FIT_DISC>(nt, cons);
  return KernelInfo{nullptr, 0, 0};
}
