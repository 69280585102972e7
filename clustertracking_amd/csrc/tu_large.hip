// tu_large.hip -- the instantiations of refine_large_kernel (clusters beyond the block kernel);
// see large_kernel.h.
#include <cmath>

#include "kargs.h"

namespace {

#include "device_common.h"
#include "block_kernel.h"   // LayoutB, wsync
#include "large_kernel.h"

template <int ND, bool ISO, bool LP>
KernelInfo one() {
  static_assert(SmemL::bytes <= 160 * 1024, "LDS budget of one CU");
  return KernelInfo{(const void*)refine_large_kernel<ND, ISO, LP>, SmemL::bytes, LT};
}

}  // namespace

KernelInfo ctr_large_kernel(int ndim, int iso, int lp) {
  if (lp) {
    if (ndim == 2) return iso ? one<2, true, true>() : one<2, false, true>();
    return iso ? one<3, true, true>() : one<3, false, true>();
  }
  if (ndim == 2) return iso ? one<2, true, false>() : one<2, false, false>();
  return iso ? one<3, true, false>() : one<3, false, false>();
}

#ifdef CTR_STAMPS
// diagnostic build only (make stamps), not part of include/ctrefine.h: totals since the last reset
extern "C" int ctr_debug_large_counters(unsigned long long* out48, int reset) {
  if (hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_large_dbg), sizeof(unsigned long long) * 48) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[48] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_large_dbg), z, sizeof z) != hipSuccess) return 1;
  }
  return 0;
}
#endif
