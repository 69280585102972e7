// tu_small.hip -- the instantiations of refine_small_kernel (compiled on their own so that the
// engine builds in parallel); see small_kernel.h.
#include <cmath>

#include "kargs.h"

namespace {

#include "device_common.h"
#include "small_kernel.h"

template <int ND, bool ISO>
const void* pick(int nf, int sg) {
  if (nf == 1 && sg == 8) return (const void*)refine_small_kernel<ND, 1, ISO, 8>;
  if (nf == 1 && sg == 64) return (const void*)refine_small_kernel<ND, 1, ISO, 64>;
  if (nf == 2 && sg == 16) return (const void*)refine_small_kernel<ND, 2, ISO, 16>;
  if (nf == 2 && sg == 64) return (const void*)refine_small_kernel<ND, 2, ISO, 64>;
  return nullptr;
}

}  // namespace

const void* ctr_small_kernel(int ndim, int nf, int iso, int sg) {
  if (ndim == 2) return iso ? pick<2, true>(nf, sg) : pick<2, false>(nf, sg);
  return iso ? pick<3, true>(nf, sg) : pick<3, false>(nf, sg);
}
