// tu_large.hip -- the instantiations of refine_large_kernel (clusters beyond the block kernel);
// see large_kernel.h.
#include <cmath>

#include "kargs.h"

namespace {

#include "device_common.h"
#include "block_kernel.h"   // LayoutB, wsync
#include "large_kernel.h"

template <int ND, bool ISO>
KernelInfo one() {
  static_assert(SmemL::bytes <= 160 * 1024, "LDS budget of one CU");
  return KernelInfo{(const void*)refine_large_kernel<ND, ISO>, SmemL::bytes, LT};
}

}  // namespace

KernelInfo ctr_large_kernel(int ndim, int iso) {
  if (ndim == 2) return iso ? one<2, true>() : one<2, false>();
  return iso ? one<3, true>() : one<3, false>();
}
