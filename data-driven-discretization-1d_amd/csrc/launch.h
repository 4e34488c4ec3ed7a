// Launch entry points of the MFMA kernels, one translation unit per equation
// so that the library builds in parallel (each per-equation specialisation of
// the integrator is ~3 k lines of ISA; a single unit took minutes):
//   mfma_spec.hip   compiled once per equation id (-DDDD_EQ=0..5): the
//                   specialised persistent integrators (float32 state in both
//                   geometries, float64 state in the one-wave geometry, the
//                   traced instantiation) and the multi-group substep kernels
//   mfma_runtime.hip  the run-time-parameterised kernels (kEq = -1), once per
//                     (geometry, state type)
//   capi.hip          C ABI, packing, every other kernel
#pragma once
#include <hip/hip_runtime.h>

#include "dev_params.h"

namespace ddd {
struct RingArgs;   // ring_args.h
namespace launch {

// rows: 64 (one-wave groups) or 256; f64: float64 state (rows = 64 only);
// traced: the s_memtime-stamped instantiation (Burgers flux form, rows 64, f32).
template <int kEq>
void integrate_spec(int rows, bool f64, bool traced, const DevParams& p, const IntegrateArgs& a,
                    int blocks, hipStream_t stream);
// one sample on two 32-row wavefronts, output layer split by channel groups (float32 state)
template <int kEq>
void integrate_split_spec(const DevParams& p, const IntegrateArgs& a, int blocks,
                          hipStream_t stream);
// ... on four 16-row wavefronts, every layer on 16x16x4 MFMAs (rhs_mfma.h kQuad; float32 state)
template <int kEq>
void integrate_quad_spec(const DevParams& p, const IntegrateArgs& a, int blocks,
                         hipStream_t stream);
// nets of up to 16 filters on the block-diagonal tower (rhs_mfma.h HalfTower; one-wave groups,
// float32 state): mfma_half.hip, one unit per equation
template <int kEq>
void integrate_half_spec(const DevParams& p, const IntegrateArgs& a, int blocks, hipStream_t stream);
template <int kEq>
void integrate_half_f64_spec(const DevParams& p, const IntegrateArgs& a, int blocks, hipStream_t stream);
template <int kEq>
void adaptive_half_spec(const DevParams& p, const AdaptiveArgs& a, int blocks, hipStream_t stream);
template <int kEq>
void substep_half_spec(const DevParams& p, const SubstepArgs& a, int groups, int grid, hipStream_t stream);
template <int kEq>
void step_half_spec(const DevParams& p, const StepArgs& a, int groups, int grid, hipStream_t stream);
// ... and the adaptive RK23 on the same four-wavefront groups
template <int kEq>
void adaptive_quad_spec(const DevParams& p, const AdaptiveArgs& a, int blocks, hipStream_t stream);
// grid: workgroups to launch (<= groups); every workgroup walks over groups.
template <int kEq>
void substep_spec(int rows, const DevParams& p, const SubstepArgs& a, int groups, int grid,
                  hipStream_t stream);

// the same walk under the device-resident command ring (substep_ring_kernel; 64-row groups):
// mfma_ring.hip, one unit per equation
template <int kEq>
void substep_ring_spec(const DevParams& p, const RingArgs& r, int grid, hipStream_t stream);

// all stages of one step in one launch (step_multi_kernel)
template <int kEq>
void step_spec(int rows, const DevParams& p, const StepArgs& a, int groups, int grid,
               hipStream_t stream);
// adaptive RK23, one controller per sample (rhs_adaptive.h); rows: 64 or 256
template <int kEq>
void adaptive_spec(int rows, const DevParams& p, const AdaptiveArgs& a, int blocks,
                   hipStream_t stream);

#define DDD_DECLARE_SPEC(EQ)                                                               \
  template <> void integrate_spec<EQ>(int, bool, bool, const DevParams&, const IntegrateArgs&, \
                                      int, hipStream_t);                                       \
  template <> void substep_spec<EQ>(int, const DevParams&, const SubstepArgs&, int, int,       \
                                    hipStream_t);                                              \
  template <> void adaptive_spec<EQ>(int, const DevParams&, const AdaptiveArgs&, int,          \
                                     hipStream_t);                                             \
  template <> void step_spec<EQ>(int, const DevParams&, const StepArgs&, int, int, hipStream_t); \
  template <> void substep_ring_spec<EQ>(const DevParams&, const RingArgs&, int, hipStream_t); \
  template <> void integrate_half_spec<EQ>(const DevParams&, const IntegrateArgs&, int, hipStream_t); \
  template <> void integrate_half_f64_spec<EQ>(const DevParams&, const IntegrateArgs&, int, hipStream_t); \
  template <> void adaptive_half_spec<EQ>(const DevParams&, const AdaptiveArgs&, int, hipStream_t); \
  template <> void substep_half_spec<EQ>(const DevParams&, const SubstepArgs&, int, int, hipStream_t); \
  template <> void step_half_spec<EQ>(const DevParams&, const StepArgs&, int, int, hipStream_t); \
  template <> void integrate_split_spec<EQ>(const DevParams&, const IntegrateArgs&, int, hipStream_t); \
  template <> void integrate_quad_spec<EQ>(const DevParams&, const IntegrateArgs&, int, hipStream_t); \
  template <> void adaptive_quad_spec<EQ>(const DevParams&, const AdaptiveArgs&, int, hipStream_t);
DDD_DECLARE_SPEC(0) DDD_DECLARE_SPEC(1) DDD_DECLARE_SPEC(2)
DDD_DECLARE_SPEC(3) DDD_DECLARE_SPEC(4) DDD_DECLARE_SPEC(5)
#undef DDD_DECLARE_SPEC

// run-time-parameterised kernels, one unit per (rows, wave_rows, float64 state):
// (64, 64), (64, 32) -- the same 64 rows on two wavefronts -- and (256, 64)
template <int kRows, int kWR, int kF64>
void integrate_runtime_unit(bool hoist, const DevParams& p, const IntegrateArgs& a, int blocks,
                            hipStream_t stream);
template <int kRows, int kWR>
void substep_runtime_unit(const DevParams& p, const SubstepArgs& a, int blocks,
                          hipStream_t stream);

// adaptive RK23 on the run-time-parameterised kernels (float64 units of the
// 64-row-wavefront geometries)
template <int kRows>
void adaptive_runtime_unit(const DevParams& p, const AdaptiveArgs& a, int blocks,
                           hipStream_t stream);
template <> void adaptive_runtime_unit<64>(const DevParams&, const AdaptiveArgs&, int, hipStream_t);
template <> void adaptive_runtime_unit<256>(const DevParams&, const AdaptiveArgs&, int, hipStream_t);

#define DDD_DECLARE_RT(ROWS, WR)                                                              \
  template <> void integrate_runtime_unit<ROWS, WR, 0>(bool, const DevParams&,               \
                                                       const IntegrateArgs&, int, hipStream_t); \
  template <> void integrate_runtime_unit<ROWS, WR, 1>(bool, const DevParams&,               \
                                                       const IntegrateArgs&, int, hipStream_t); \
  template <> void substep_runtime_unit<ROWS, WR>(const DevParams&, const SubstepArgs&, int, \
                                                  hipStream_t);
DDD_DECLARE_RT(64, 64)
DDD_DECLARE_RT(64, 32)
DDD_DECLARE_RT(256, 64)
#undef DDD_DECLARE_RT

// "wide" flavour (stencils up to 12 points, up to 24 output channels; rhs_mfma.h
// kWide), 64-row wavefronts only: -DDDD_RT_WIDE=1 units
template <int kRows, int kF64>
void integrate_wide_unit(bool hoist, const DevParams& p, const IntegrateArgs& a, int blocks,
                         hipStream_t stream);
template <int kRows>
void substep_wide_unit(const DevParams& p, const SubstepArgs& a, int blocks, hipStream_t stream);
template <int kRows>
void adaptive_wide_unit(const DevParams& p, const AdaptiveArgs& a, int blocks,
                        hipStream_t stream);
#define DDD_DECLARE_WIDE(ROWS)                                                                \
  template <> void integrate_wide_unit<ROWS, 0>(bool, const DevParams&, const IntegrateArgs&, \
                                                int, hipStream_t);                            \
  template <> void integrate_wide_unit<ROWS, 1>(bool, const DevParams&, const IntegrateArgs&, \
                                                int, hipStream_t);                            \
  template <> void substep_wide_unit<ROWS>(const DevParams&, const SubstepArgs&, int,         \
                                           hipStream_t);                                      \
  template <> void adaptive_wide_unit<ROWS>(const DevParams&, const AdaptiveArgs&, int,       \
                                            hipStream_t);
DDD_DECLARE_WIDE(64)
DDD_DECLARE_WIDE(256)
#undef DDD_DECLARE_WIDE

// towers other than 5 taps x 32 channels (rhs_mfma.h: Tower<kK, kCB>), one kernel per
// unit: mfma_big.hip.  kRows: 64 or 256.
template <int kK, int kCB, int kRows, bool kF64>
void integrate_big_unit(const DevParams& p, const IntegrateArgs& a, int blocks, hipStream_t stream);
template <int kK, int kCB, int kRows>
void substep_big_unit(const DevParams& p, const SubstepArgs& a, int blocks, hipStream_t stream);
template <int kK, int kCB, int kRows>
void adaptive_big_unit(const DevParams& p, const AdaptiveArgs& a, int blocks, hipStream_t stream);
#define DDD_DECLARE_BIG_ROWS(K, CB, ROWS)                                                       \
  template <> void integrate_big_unit<K, CB, ROWS, false>(const DevParams&, const IntegrateArgs&, \
                                                          int, hipStream_t);                     \
  template <> void integrate_big_unit<K, CB, ROWS, true>(const DevParams&, const IntegrateArgs&,  \
                                                         int, hipStream_t);                      \
  template <> void substep_big_unit<K, CB, ROWS>(const DevParams&, const SubstepArgs&, int,      \
                                                 hipStream_t);                                   \
  template <> void adaptive_big_unit<K, CB, ROWS>(const DevParams&, const AdaptiveArgs&, int,    \
                                                  hipStream_t);
#define DDD_DECLARE_BIG(K, CB) DDD_DECLARE_BIG_ROWS(K, CB, 64) DDD_DECLARE_BIG_ROWS(K, CB, 256)
// the towers built (capi.hip: decide_mfma picks the smallest one that holds the net, embed_tower pads it;
// 7 taps x 64 filters runs its hidden layers as a loop over the taps -- rhs_mfma.h: Tower::kRolled --:
// fully unrolled they took > 20 minutes to compile, rolled 2-8 minutes per kernel)
#define DDD_FOR_EACH_BIG_TOWER(X) X(3, 1) X(7, 1) X(5, 2) X(7, 2)
DDD_FOR_EACH_BIG_TOWER(DDD_DECLARE_BIG)
#undef DDD_DECLARE_BIG
#undef DDD_DECLARE_BIG_ROWS

inline void integrate_runtime(int rows, int wave_rows, bool f64, bool hoist, const DevParams& p,
                              const IntegrateArgs& a, int blocks, hipStream_t stream) {
  if (rows == 64 && wave_rows == 64) {
    if (f64) integrate_runtime_unit<64, 64, 1>(hoist, p, a, blocks, stream);
    else integrate_runtime_unit<64, 64, 0>(hoist, p, a, blocks, stream);
  } else if (rows == 64) {
    if (f64) integrate_runtime_unit<64, 32, 1>(hoist, p, a, blocks, stream);
    else integrate_runtime_unit<64, 32, 0>(hoist, p, a, blocks, stream);
  } else {
    if (f64) integrate_runtime_unit<256, 64, 1>(hoist, p, a, blocks, stream);
    else integrate_runtime_unit<256, 64, 0>(hoist, p, a, blocks, stream);
  }
}

inline void substep_runtime(int rows, int wave_rows, const DevParams& p, const SubstepArgs& a,
                            int blocks, hipStream_t stream) {
  if (rows == 64 && wave_rows == 64) substep_runtime_unit<64, 64>(p, a, blocks, stream);
  else if (rows == 64) substep_runtime_unit<64, 32>(p, a, blocks, stream);
  else substep_runtime_unit<256, 64>(p, a, blocks, stream);
}

}  // namespace launch
}  // namespace ddd
