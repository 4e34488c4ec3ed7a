// Launch entry points of the WENO5 + Godunov-flux exact solver's kernels (rhs_weno.h),
// compiled in weno_unit.hip; kept apart from launch.h, which every MFMA unit includes.
#pragma once
#include <hip/hip_runtime.h>

#include "dev_params.h"

namespace ddd {
namespace launch {

// models weno::supports() accepts (capi.hip: use_weno_kernel)
void weno_substep(const DevParams& p, const SubstepArgs& a, hipStream_t stream);
void weno_integrate(bool f64, const DevParams& p, const IntegrateArgs& a, hipStream_t stream);
void weno_adaptive(const DevParams& p, const AdaptiveArgs& a, hipStream_t stream);

}  // namespace launch
}  // namespace ddd
