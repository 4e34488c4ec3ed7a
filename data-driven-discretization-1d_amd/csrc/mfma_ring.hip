// The command-ring substep kernel of the per-equation models (rhs_mfma.h:
// substep_ring_kernel), compiled once per equation id (-DDDD_EQ=<0..5>) like mfma_spec.hip.
#include <hip/hip_runtime.h>

#include "launch.h"
#include "rhs_ring.h"

#ifndef DDD_EQ
#error "compile with -DDDD_EQ=<equation id 0..5>"
#endif

namespace ddd {
namespace launch {

template <>
void substep_ring_spec<DDD_EQ>(const DevParams& p, const RingArgs& r, int grid,
                               hipStream_t stream) {
  hipLaunchKernelGGL((mfma::substep_ring_kernel<64, 64, DDD_EQ>), dim3(grid), dim3(64), 0, stream,
                     p, r);
}

}  // namespace launch
}  // namespace ddd
