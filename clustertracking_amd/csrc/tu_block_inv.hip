// tu_block_inv.hip -- the instantiations of refine_block_kernel for the inv_series_<N> profiles
// (FIT = CTR_FIT_INV_SERIES; fitfunc.py:148-154,334-343), 2D and 3D; see block_kernel.h.  One per
// (ndim, isotropic, NT, constrained): these problems are rare, they all take the wavefront
// counts of the default scheduling.
#include <cmath>

#include "kargs.h"

namespace {

#include "device_common.h"
#include "block_kernel.h"

constexpr size_t LDS_CU = 160 * 1024;  // LDS of one CU

template <int NT> struct WavesFor { static constexpr int value = NT <= 2 ? 8 : (NT <= 3 ? 4 : (NT <= 6 ? 2 : 1)); };

template <int ND, bool ISO, int NT, bool CONS = false>
KernelInfo one() {
  constexpr int W = WavesFor<NT>::value;
  static_assert(SmemB<NT, W, CONS>::bytes <= LDS_CU, "LDS budget of one CU");
  return KernelInfo{(const void*)refine_block_kernel<ND, ISO, NT, W, CONS, false, CTR_FIT_INV_SERIES>, SmemB<NT, W, CONS>::bytes, WAVE * W};
}

template <int ND, bool ISO>
KernelInfo by_nt(int nt, int cons) {
  if (cons) {   // constrained clusters have at most 4 features = 29 variables
    if (nt == 1) return one<ND, ISO, 1, true>();
    if (nt == 2) return one<ND, ISO, 2, true>();
    return KernelInfo{nullptr, 0, 0};
  }
  switch (nt) {
    case 1: return one<ND, ISO, 1>();
    case 2: return one<ND, ISO, 2>();
    case 3: return one<ND, ISO, 3>();
    case 4: return one<ND, ISO, 4>();
    case 5: return one<ND, ISO, 5>();
    case 6: return one<ND, ISO, 6>();
    case 7: return one<ND, ISO, 7>();
    case 8: return one<ND, ISO, 8>();
    default: return KernelInfo{nullptr, 0, 0};
  }
}

}  // namespace

KernelInfo ctr_block_kernel_inv(int ndim, int iso, int nt, int cons) {
  if (ndim == 2) return iso ? by_nt<2, true>(nt, cons) : by_nt<2, false>(nt, cons);
  return iso ? by_nt<3, true>(nt, cons) : by_nt<3, false>(nt, cons);
}
