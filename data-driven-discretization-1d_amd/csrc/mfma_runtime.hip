// Run-time-parameterised MFMA kernels (kEq = -1): every model the MFMA path
// accepts that has no per-equation specialisation; see launch.h.  Compiled
// once per (geometry, state type): -DDDD_RT_ROWS=64|256 -DDDD_RT_WR=64|32
// -DDDD_RT_F64=0|1; the float32 units also hold the geometry's substep kernel.
#include <hip/hip_runtime.h>

#include "launch.h"
#include "rhs_adaptive.h"
#include "rhs_mfma.h"

#if !defined(DDD_RT_ROWS) || !defined(DDD_RT_WR) || !defined(DDD_RT_F64)
#error "compile with -DDDD_RT_ROWS=.. -DDDD_RT_WR=.. -DDDD_RT_F64=.."
#endif
#ifndef DDD_RT_WIDE
#define DDD_RT_WIDE 0   // 1: the wide flavour (stencils <= 12, <= 24 output channels)
#endif

namespace ddd {
namespace launch {

#if DDD_RT_F64
typedef double RtState;
#else
typedef float RtState;
#endif

#if DDD_RT_WIDE
template <>
void integrate_wide_unit<DDD_RT_ROWS, DDD_RT_F64>(bool hoist, const DevParams& p,
                                                  const IntegrateArgs& a, int blocks,
                                                  hipStream_t stream) {
  const dim3 grid(blocks), block(DDD_RT_ROWS);
  if (hoist)
    hipLaunchKernelGGL((mfma::integrate_kernel<DDD_RT_ROWS, 64, RtState, true, -1, mfma::kTraceByDefault, true>),
                       grid, block, 0, stream, p, a);
  else
    hipLaunchKernelGGL((mfma::integrate_kernel<DDD_RT_ROWS, 64, RtState, false, -1, mfma::kTraceByDefault, true>),
                       grid, block, 0, stream, p, a);
}
#if !DDD_RT_F64
template <>
void substep_wide_unit<DDD_RT_ROWS>(const DevParams& p, const SubstepArgs& a, int blocks,
                                    hipStream_t stream) {
  hipLaunchKernelGGL((mfma::substep_kernel<DDD_RT_ROWS, 64, -1, true>), dim3(blocks),
                     dim3(DDD_RT_ROWS), 0, stream, p, a);
}
#else
template <>
void adaptive_wide_unit<DDD_RT_ROWS>(const DevParams& p, const AdaptiveArgs& a, int blocks,
                                     hipStream_t stream) {
  hipLaunchKernelGGL((mfma::adaptive_kernel<DDD_RT_ROWS, 64, false, -1, true>), dim3(blocks),
                     dim3(DDD_RT_ROWS), 0, stream, p, a);
}
#endif
#else   // !DDD_RT_WIDE
template <>
void integrate_runtime_unit<DDD_RT_ROWS, DDD_RT_WR, DDD_RT_F64>(bool hoist, const DevParams& p,
                                                                const IntegrateArgs& a,
                                                                int blocks, hipStream_t stream) {
  const dim3 grid(blocks), block(DDD_RT_ROWS / DDD_RT_WR * 64);
  if (hoist)
    hipLaunchKernelGGL((mfma::integrate_kernel<DDD_RT_ROWS, DDD_RT_WR, RtState, true>), grid,
                       block, 0, stream, p, a);
  else
    hipLaunchKernelGGL((mfma::integrate_kernel<DDD_RT_ROWS, DDD_RT_WR, RtState, false>), grid,
                       block, 0, stream, p, a);
}

#if !DDD_RT_F64
template <>
void substep_runtime_unit<DDD_RT_ROWS, DDD_RT_WR>(const DevParams& p, const SubstepArgs& a,
                                                  int blocks, hipStream_t stream) {
  hipLaunchKernelGGL((mfma::substep_kernel<DDD_RT_ROWS, DDD_RT_WR>), dim3(blocks),
                     dim3(DDD_RT_ROWS / DDD_RT_WR * 64), 0, stream, p, a);
}
#endif

#if DDD_RT_F64 && DDD_RT_WR == 64
// adaptive RK23 (float64 state): weights re-fetched per evaluation, any depth
template <>
void adaptive_runtime_unit<DDD_RT_ROWS>(const DevParams& p, const AdaptiveArgs& a, int blocks,
                                        hipStream_t stream) {
  hipLaunchKernelGGL((mfma::adaptive_kernel<DDD_RT_ROWS, 64, false, -1>), dim3(blocks),
                     dim3(DDD_RT_ROWS), 0, stream, p, a);
}
#endif

#endif   // DDD_RT_WIDE

}  // namespace launch
}  // namespace ddd
