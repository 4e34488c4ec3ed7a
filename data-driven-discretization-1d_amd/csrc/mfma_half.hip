// The persistent integrator of the per-equation models on the block-diagonal tower of nets
// with up to 16 filters (rhs_mfma.h: HalfTower), compiled once per equation id
// (-DDDD_EQ=<0..5>) like mfma_spec.hip.
#include <hip/hip_runtime.h>

#include "launch.h"
#include "rhs_mfma.h"

#ifndef DDD_EQ
#error "compile with -DDDD_EQ=<equation id 0..5>"
#endif

namespace ddd {
namespace launch {

template <>
void integrate_half_spec<DDD_EQ>(const DevParams& p, const IntegrateArgs& a, int blocks,
                                 hipStream_t stream) {
  hipLaunchKernelGGL((mfma::integrate_kernel<64, 64, float, true, DDD_EQ, false, false, mfma::HalfTower>),
                     dim3(blocks), dim3(64), 0, stream, p, a);
}

}  // namespace launch
}  // namespace ddd
