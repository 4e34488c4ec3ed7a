// The WENO5 + Godunov-flux exact solver's kernels (rhs_weno.h): one instantiation per
// (grid points per lane, equation); see launch.h.
#include <hip/hip_runtime.h>

#include "launch_weno.h"
#include "rhs_weno.h"

namespace ddd {
namespace launch {

namespace {
inline dim3 weno_grid(int batch) { return dim3((batch + weno::kWaves - 1) / weno::kWaves); }
const dim3 kWenoBlock(64 * weno::kWaves);
}  // namespace

// (points per lane) x (equation): the launch expression X(kP, kEq) for the model's pair
#define DDD_WENO_DISPATCH(X)                                                  \
  do {                                                                        \
    const int pp = p.N / 64;                                                  \
    switch (p.equation) {                                                     \
      case EQ_BURGERS_GODUNOV:                                                \
        if (pp == 1) X(1, EQ_BURGERS_GODUNOV); else if (pp == 2) X(2, EQ_BURGERS_GODUNOV); \
        else if (pp == 4) X(4, EQ_BURGERS_GODUNOV); else X(8, EQ_BURGERS_GODUNOV);         \
        break;                                                                \
      case EQ_KDV_GODUNOV:                                                    \
        if (pp == 1) X(1, EQ_KDV_GODUNOV); else if (pp == 2) X(2, EQ_KDV_GODUNOV);         \
        else if (pp == 4) X(4, EQ_KDV_GODUNOV); else X(8, EQ_KDV_GODUNOV);                 \
        break;                                                                \
      default:                                                                \
        if (pp == 1) X(1, EQ_KS_GODUNOV); else if (pp == 2) X(2, EQ_KS_GODUNOV);           \
        else if (pp == 4) X(4, EQ_KS_GODUNOV); else X(8, EQ_KS_GODUNOV);                   \
        break;                                                                \
    }                                                                         \
  } while (0)

void weno_substep(const DevParams& p, const SubstepArgs& a, hipStream_t stream) {
#define X(P, EQ) hipLaunchKernelGGL((weno::substep_kernel<P, EQ>), weno_grid(a.batch), kWenoBlock, 0, stream, p, a)
  DDD_WENO_DISPATCH(X);
#undef X
}

void weno_integrate(bool f64, const DevParams& p, const IntegrateArgs& a, hipStream_t stream) {
#define X(P, EQ)                                                                              \
  do {                                                                                        \
    if (f64) hipLaunchKernelGGL((weno::integrate_kernel<P, EQ, double>), weno_grid(a.batch),  \
                                kWenoBlock, 0, stream, p, a);                                 \
    else hipLaunchKernelGGL((weno::integrate_kernel<P, EQ, float>), weno_grid(a.batch),       \
                            kWenoBlock, 0, stream, p, a);                                     \
  } while (0)
  DDD_WENO_DISPATCH(X);
#undef X
}

void weno_adaptive(const DevParams& p, const AdaptiveArgs& a, hipStream_t stream) {
#define X(P, EQ) hipLaunchKernelGGL((weno::adaptive_kernel<P, EQ>), weno_grid(a.batch), kWenoBlock, 0, stream, p, a)
  DDD_WENO_DISPATCH(X);
#undef X
}

}  // namespace launch
}  // namespace ddd
