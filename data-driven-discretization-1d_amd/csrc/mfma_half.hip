// The persistent integrators (fixed step in float32 / float64 state, adaptive RK23) of the
// per-equation models on the block-diagonal tower of nets with up to 16 filters (rhs_mfma.h: HalfTower), compiled once per equation id
// (-DDDD_EQ=<0..5>) like mfma_spec.hip.
#include <hip/hip_runtime.h>

#include "launch.h"
#include "rhs_adaptive.h"
#include "rhs_mfma.h"

#ifndef DDD_EQ
#error "compile with -DDDD_EQ=<equation id 0..5>"
#endif

// Which form of the tower the kernels of this unit carry: 1 = 16-channel tiles (Tile16Tower),
// 0 = the block-diagonal 32x32x2 form (HalfTower); A/B in profiles/r6_ablation.txt.
#ifndef DDD_HALF_T16
#define DDD_HALF_T16 1
#endif

namespace ddd {
namespace mfma {
#if DDD_HALF_T16
typedef Tile16Tower SmallTower;
#else
typedef HalfTower SmallTower;
#endif
}  // namespace mfma
namespace launch {

template <>
void integrate_half_spec<DDD_EQ>(const DevParams& p, const IntegrateArgs& a, int blocks,
                                 hipStream_t stream) {
  hipLaunchKernelGGL((mfma::integrate_kernel<64, 64, float, true, DDD_EQ, false, false, mfma::SmallTower>),
                     dim3(blocks), dim3(64), 0, stream, p, a);
}

// float64 state (SciPy holds y in float64, integrate.py:154) ...
template <>
void integrate_half_f64_spec<DDD_EQ>(const DevParams& p, const IntegrateArgs& a, int blocks,
                                     hipStream_t stream) {
  hipLaunchKernelGGL((mfma::integrate_kernel<64, 64, double, true, DDD_EQ, false, false, mfma::SmallTower>),
                     dim3(blocks), dim3(64), 0, stream, p, a);
}

// one launch per substep / per step (the walks of substep_multi_kernel / step_multi_kernel)
template <>
void substep_half_spec<DDD_EQ>(const DevParams& p, const SubstepArgs& a, int groups, int grid,
                               hipStream_t stream) {
  hipLaunchKernelGGL((mfma::substep_multi_kernel<64, 64, DDD_EQ, mfma::SmallTower>), dim3(grid), dim3(64), 0,
                     stream, p, a, groups);
}
template <>
void step_half_spec<DDD_EQ>(const DevParams& p, const StepArgs& a, int groups, int grid,
                            hipStream_t stream) {
  hipLaunchKernelGGL((mfma::step_multi_kernel<64, 64, DDD_EQ, mfma::SmallTower>), dim3(grid), dim3(64), 0,
                     stream, p, a, groups);
}

// ... and the production integrator: adaptive RK23, one controller per sample
template <>
void adaptive_half_spec<DDD_EQ>(const DevParams& p, const AdaptiveArgs& a, int blocks,
                                hipStream_t stream) {
  hipLaunchKernelGGL((mfma::adaptive_kernel<64, 64, true, DDD_EQ, false, mfma::SmallTower>), dim3(blocks),
                     dim3(64), 0, stream, p, a);
}

}  // namespace launch
}  // namespace ddd
