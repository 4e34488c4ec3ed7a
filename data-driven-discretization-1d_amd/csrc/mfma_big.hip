// Run-time-parameterised MFMA kernels for conv towers other than 5 taps x 32
// channels (rhs_mfma.h: Tower<kK, kCB>; layer weights streamed from L2); see launch.h.
// ONE kernel per translation unit (the fully unrolled 64-filter layers take minutes
// to compile: the units build side by side):
//   -DDDD_BIG_K=3|5|7 -DDDD_BIG_CB=1|2   the tower
//   -DDDD_BIG_ROWS=64|256                one-wave groups / four-wave groups
//   -DDDD_BIG_KIND=0  float32 integrator     1  fused substep (float32)
//                 2  float64 integrator     3  adaptive RK23 (float64 state)
#include <hip/hip_runtime.h>

#include "launch.h"
#include "rhs_adaptive.h"
#include "rhs_mfma.h"

#if !defined(DDD_BIG_K) || !defined(DDD_BIG_CB) || !defined(DDD_BIG_ROWS) || !defined(DDD_BIG_KIND)
#error "compile with -DDDD_BIG_K=.. -DDDD_BIG_CB=.. -DDDD_BIG_ROWS=.. -DDDD_BIG_KIND=.."
#endif

namespace ddd {
namespace launch {

typedef mfma::Tower<DDD_BIG_K, DDD_BIG_CB> BigTower;

#if DDD_BIG_KIND == 0 || DDD_BIG_KIND == 2
#if DDD_BIG_KIND == 2
typedef double BigState;
#else
typedef float BigState;
#endif
template <>
void integrate_big_unit<DDD_BIG_K, DDD_BIG_CB, DDD_BIG_ROWS, DDD_BIG_KIND == 2>(
    const DevParams& p, const IntegrateArgs& a, int blocks, hipStream_t stream) {
  hipLaunchKernelGGL(
      (mfma::integrate_kernel<DDD_BIG_ROWS, 64, BigState, false, -1, mfma::kTraceByDefault, false, BigTower>),
      dim3(blocks), dim3(DDD_BIG_ROWS), 0, stream, p, a);
}
#elif DDD_BIG_KIND == 1
template <>
void substep_big_unit<DDD_BIG_K, DDD_BIG_CB, DDD_BIG_ROWS>(const DevParams& p, const SubstepArgs& a,
                                                           int blocks, hipStream_t stream) {
  hipLaunchKernelGGL((mfma::substep_kernel<DDD_BIG_ROWS, 64, -1, false, BigTower>), dim3(blocks),
                     dim3(DDD_BIG_ROWS), 0, stream, p, a);
}
#else
template <>
void adaptive_big_unit<DDD_BIG_K, DDD_BIG_CB, DDD_BIG_ROWS>(const DevParams& p,
                                                            const AdaptiveArgs& a, int blocks,
                                                            hipStream_t stream) {
  hipLaunchKernelGGL((mfma::adaptive_kernel<DDD_BIG_ROWS, 64, false, -1, false, BigTower>),
                     dim3(blocks), dim3(DDD_BIG_ROWS), 0, stream, p, a);
}
#endif

}  // namespace launch
}  // namespace ddd
