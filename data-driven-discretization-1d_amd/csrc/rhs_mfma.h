// Fused learned-stencil right-hand side + Runge-Kutta stepping on CDNA4 f32 MFMA.
//
// Work decomposition (DESIGN.md "MFMA kernel"):
//   * one workgroup = kRows "rows" (grid points) = floor(kRows / N) whole
//     samples; wavefront w owns rows [kWR w, kWR w + kWR), kWR = 64 or 32.
//     <64, 64>: one free-running wavefront per workgroup when samples fit a
//     wavefront (64 % N == 0); <64, 32>: the same 64 rows on two wavefronts,
//     for batches too small to give every SIMD two 64-row wavefronts;
//     <256, 64>: four wavefronts with block barriers between layers (any N <= 256);
//   * the conv tower runs on the matrix cores as implicit GEMMs
//         D[out-channel][position] += W[out-channel][k] * h[k][position],
//     reduction index k = (tap, in-channel):
//       - input layer   1 -> 32, K=5 : v_mfma_f32_32x32x2_f32, 3 steps (5 taps + bias)
//       - hidden layers 32 -> 32, K=5: v_mfma_f32_32x32x2_f32, 80 steps + 1 bias step;
//         the layer's 160x32 weight panel lives in 81 VGPRs per lane
//       - output layer  32 -> C_out<=16: v_mfma_f32_4x4x1_16b_f32 with the weight block
//         broadcast, 161 steps x ceil(C / 4) channel groups, lane == row (final_layer4)
//   * the VALU does what is left with lane == row: projection onto the
//     accuracy-constrained stencils, stencil apply, equation of motion, forcing
//     and the Runge-Kutta update;
//   * activations travel between layers through two LDS buffers [256][36] f32
//     (row stride 36 floats = 144 B keeps ds_read_b128 / ds_write_b128 of 16
//     consecutive rows on distinct 16-byte bank slots: 9 r mod 16 is a bijection);
//   * periodic halos are never materialised: each lane computes the row index
//     of (pos + tap - 2) mod N inside its own sample and reads that row.
//
// f32-input MFMA is bit-for-bit an fmaf chain in k order, so the arithmetic is
// IEEE float32 like the reference's TF graph; only the summation order differs.
// Two deliberate deviations, both documented where they live: tanh is fast_tanh (one
// v_exp + one v_rcp, |error| <= 2e-7 absolute and <= 3e-7 relative, not the library's
// 1-ulp tanhf), and relu is a clamp that maps NaN to 0 -- eval_rhs restores the NaNs the
// reference's relu would have propagated (see "NaN through relu" there).
#pragma once
#include <type_traits>
#include <utility>

#include "dev_params.h"

namespace ddd {
namespace mfma {

constexpr int kHS = 36;          // padded activation row stride (floats)
constexpr int kF = 32;           // hidden channels
constexpr int kKW = 5;           // conv taps
constexpr int kInSteps = 3;      // (5 taps + bias) / 2
constexpr int kHidSteps = 81;    // 5*32/2 MFMA steps + 1 bias step
constexpr int kFin4K = kKW * kF + 1;   // output layer on 4x4x1 MFMAs: 160 reduction steps + bias
constexpr int kTrigMax = 12;     // 2 * (distinct wavenumbers) kept per lane
// Flavours of the run-time-parameterised kernels (template parameter kWide):
// default: stencils <= 8 points, <= 16 output channels; wide: <= 12 points,
// <= 24 channels of the net, projection always folded into the output layer.
__host__ __device__ constexpr int flavour_stencil(bool wide) { return wide ? kGWide : kGMax; }
__host__ __device__ constexpr int flavour_channels(bool wide) { return wide ? kChWide : kChMax; }
// Output channels the kernels carry in registers.  The wide flavour's output layer is ALWAYS
// folded (round 5): it emits coefficient g of derivative d as channel kGWide d + g -- slots
// of twelve, three channel groups per derivative, D <= 3 --, so the epilogue's register
// indices are compile-time constants and there is no projection left to run there.
constexpr int kWideDerivs = 3;
__host__ __device__ constexpr int flavour_net_channels(bool wide) {
  return wide ? kWideDerivs * kGWide : kChMax;
}
// ... coefficient g of derivative d = channel wide_slot(G) d + g: slots of 8 for stencils
// of up to 8 points, of exactly G above (27 channels = 7 channel groups for 9 points and
// three derivatives, where slots of 12 would issue 9).
__host__ __device__ constexpr int wide_slot(int G) { return G <= kGMax ? kGMax : G; }
// projection tables in LDS: 4 bias rows + one null-space row per output channel (default
// flavour; the wide one has nothing to project and stages the bias rows -- the fixed
// stencils -- only: with the 24 null-space rows its 64-row workgroups took 20 808 bytes,
// seven to a CU instead of eight, one SIMD in four left with a single wavefront)
__host__ __device__ constexpr int tab_rows(bool wide) { return wide ? 4 : 4 + flavour_channels(false); }

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// Geometry of the conv tower a kernel instantiation carries: kK taps, kCB blocks of
// 32 hidden channels.  Tower<5, 1> is the reference's default net (training.py:134-136)
// and the only one with per-equation kernels, resident weights and the rest of the
// tuning above; the other towers (7 taps, 64 filters, 3 taps: hyper-parameters
// create_hparams admits, model.py:455-458 builds whatever they say) run on the
// run-time-parameterised kernels with the layer weights streamed from L2
// (input_layer_big / hidden_layer_stream below).  Nets in between are embedded with
// zero weights in the next tower up (capi.hip: embed_tower).
template <int kK_, int kCB_>
struct Tower {
  static constexpr int kK = kK_;                    // conv taps (odd)
  static constexpr int kCB = kCB_;                  // 32-channel blocks
  static constexpr int kC = 32 * kCB_;              // hidden channels
  static constexpr int kHS = kC + 4;                // padded activation row stride (floats):
                                                    // (kC / 4 + 1) r mod 16 is a bijection on 16 rows
  static constexpr int kInSteps = (kK_ + 2) / 2;    // (taps + bias) / 2, rounded up
  static constexpr int kHidK = kK_ * kC / 2;        // 32x32x2 steps per output block (without bias)
  static constexpr int kHidGroups = kHidK / 4;      // ... in groups of four (one float4 of weights)
  static constexpr int kFinC = kC;                  // channels the output layer reduces over per tap
  static constexpr int kFinK = kK_ * kC + 1;        // 4x4x1 reduction steps of the output layer (+ bias)
  static constexpr int kOperandGroups = kK_ * kC / 4;   // ds_read_b128 per lane in the output layer
  static constexpr bool kDefault = kK_ == 5 && kCB_ == 1;
  static constexpr bool kHalf = false;              // (HalfTower below)
  static constexpr bool kTile16 = false;            // (Tile16Tower below)
  // 7 taps x 64 filters: the hidden layer runs as a loop over the taps (hidden_layer_rolled);
  // fully unrolled its 896 MFMAs per layer and kernel took > 20 minutes to compile
  static constexpr bool kRolled = kK_ * kCB_ >= 14;
  static_assert(kK_ % 2 == 1 && kK_ >= 3 && kK_ <= 7 && kCB_ >= 1 && kCB_ <= 2, "tower geometry");
};
typedef Tower<kKW, 1> DefaultTower;
// Nets of up to 16 filters (filter_size <= 16, training.py:134-136 leaves it free) in the
// default tower's layout, BLOCK-DIAGONAL (round 6): embedded with zero weights they occupy a
// quarter of every 32 x 32 x 2 MFMA of the hidden layer (16 of 32 output rows x 16 of 32
// reduction channels: 25.7 % of the issued work is the net's).  Here one MFMA pass carries
// BOTH 32-position tiles of the wavefront: output rows 0..15 = the 16 channels at positions
// j, rows 16..31 = the same channels at positions 32 + j; reduction half 0 = the 16 input
// channels at the tap rows of position j, half 1 = those of position 32 + j, the weight
// panel block-diagonal.  Half the hidden-layer MFMAs (81 instead of 162), and the output
// layer reduces over 5 x 16 + 1 instead of 5 x 32 + 1 steps.  Every accumulation chain keeps
// the embedded evaluation's order (the products dropped are exact zeros): the same bits.
// LDS layout, input layer and everything outside the tower: the default tower's.
struct HalfTower {
  static constexpr int kK = kKW, kCB = 1, kC = 32, kHS = 36, kInSteps = 3;
  static constexpr int kHidK = kKW * 32 / 2, kHidGroups = kHidK / 4;
  static constexpr int kFinC = 16;
  static constexpr int kFinK = kKW * kFinC + 1;
  static constexpr int kOperandGroups = kKW * kFinC / 4;
  static constexpr bool kDefault = true, kRolled = false, kHalf = true, kTile16 = false;
};
// ... and the same nets on 16-CHANNEL TILES (round 6): every layer of the
// tower on v_mfma_f32_16x16x4_f32 -- D[16 channels][16 positions], four position tiles per
// wavefront, the reduction four (tap, channel) slots per step: no zero blocks at all
// (hidden layer: 4 x 21 steps of 32 cycles = 2 688 against the block-diagonal form's 5 184).
//   * LDS rows keep 36 floats, channel c at float 4 (c & 3) + (c >> 2): the hidden layer's B
//     operand of lane (position j, slot sg) is ONE aligned ds_read_b128 per tap -- channels
//     sg, 4 + sg, 8 + sg, 12 + sg = the lane's slot in steps 4 tap + 0 .. 3, so the chain of
//     every output runs over c = 0 .. 15 per tap: the embedded evaluation's order, the same
//     bits -- and D (lane: channels 4 sg + r) lands with four ds_write_b32 (store_tile16);
//   * input layer: step 0 = taps 0 .. 3, step 1 = tap 4, bias, 0, 0 (input_layer's k order),
//     operands by ds_bpermute from the lanes' u / std;
//   * output layer on the 4x4x1 MFMAs as before (lane == row), its operands four
//     ds_read_b128 per tap row, picked in natural channel order (final_layer4_t16).
struct Tile16Tower {
  static constexpr int kK = kKW, kCB = 1, kC = 32, kHS = 36, kInSteps = 3;
  static constexpr int kHidK = kKW * 32 / 2, kHidGroups = kHidK / 4;
  static constexpr int kFinC = 16;
  static constexpr int kFinK = kKW * kFinC + 1;
  static constexpr int kOperandGroups = kKW * kFinC / 4;
  static constexpr bool kDefault = true, kRolled = false, kHalf = false, kTile16 = true;
};
constexpr int kT16InSteps = 2, kT16HidSteps = 4 * kKW + 1;   // A-operand rows of DevParams::w_quad in this mode
// floats of one hidden layer in the streamed layout: [group][out block][lane] float4, then the
// bias rows [out block][lane]
template <class TW>
__host__ __device__ constexpr int stream_layer_floats() {
  return TW::kHidGroups * TW::kCB * 64 * 4 + TW::kCB * 64;
}
template <class TW>
__host__ __device__ constexpr int fin4_regs_t(int groups) { return (TW::kFinK * groups + 15) / 16; }

// kRows = rows (grid points) per workgroup: 256 (four wavefronts, block
// barriers between layers) or 64 (ONE wavefront owns whole samples, N <= 64:
// no cross-wave dependency, wavefronts free-run and eight workgroups share a
// CU).  Sizes are chosen so that 160 KiB of LDS hold 2 x 256-row or 8 x 64-row
// workgroups.
// A/B (profiles/r5_ablation.txt): the one-wave kernels of the 3-tap tower at THREE wavefronts
// per SIMD (single activation buffer: 11.9 KB of LDS, 168 VGPRs, two resident weight groups)
// measured below two wavefronts with the whole hidden layer resident (241 VGPRs): 62.9 vs
// 64.4 % at 4 096 samples, 39.7 vs 50.5 % with one launch per substep.
// Issue priority by phase (s_setprio; prio_mode(kEq) below).  0: none (rounds 1-5; the
// run-time-parameterised kernels).  1 (round 5): the matrix layers raised, the VALU phases at 0
// -- neutral.  2 (round 6): the other way round -- a wavefront's VALU phases (epilogue,
// forcing, Runge-Kutta update, the adaptive controller) raised, its matrix layers at 0: next
// to a SIMD partner that streams MFMAs a short VALU burst costs the partner nothing (the
// matrix pipe is busy with its last MFMA for 64 cycles anyway) and gets this wavefront back to
// its own MFMAs sooner.  3: as 2, and only the steady middle of the hidden / output layers at
// 0 -- their first and last operand groups, relu and the activation stores raised as well.
// Measured (profiles/r6_ablation.txt; fractions of 157.3 TFLOP/s, mode 0 / 2 / 3):
//   KdV N=64 84.3 / 84.7 / 88.0    adaptive KdV 76.0 / 76.4 / 78.6    KS N=256 82.8 / 83.5 / 83.7
//   Burgers (headline) 82.1 / 81.9 / 81.8    adaptive Burgers 76.4 / 77.0 / 76.7
// 4 (A/B): 3 with the forcing phases inside the matrix layers at 0 -- no change.  5: 3 with the
// output layer's middle at 1 instead of 0 (its 8-cycle MFMAs ahead of the partner's 64-cycle
// ones): KdV 88.4 -> 89.1, KS 84.2 -> 84.4, Burgers unchanged.
// -> per-equation kernels: 5 where the evaluation carries no forcing phases (KdV, KS), 2 in
//    the Burgers family.
#ifndef DDD_PRIO_PHASES
#define DDD_PRIO_PHASES (-1)   // -1: by equation (prio_mode)
#endif
#ifndef DDD_K3_WAVES
#define DDD_K3_WAVES 2
#endif
#ifndef DDD_K3_RESIDENT_ALL
#define DDD_K3_RESIDENT_ALL (DDD_K3_WAVES == 2)
#endif
template <int kRows, int kWR = 64, bool kWide = false, class TW = DefaultTower>
struct Shared {
  static constexpr int kPmMax = kRows;          // (sample, mode) pairs staged
  static constexpr int kFkMax = 3 * kRows / 4;  // samples * 12 harmonic sums (zero padded)
  // One-wave groups of the 64-channel towers keep ONE activation buffer: a wavefront issues
  // every operand read of a layer before it stores the layer's output (LDS operations of one
  // wavefront execute in order), so the layer can be written in place -- 20 KB instead of
  // 37 KB per group: eight groups per CU = two wavefronts per SIMD instead of one.  hB then
  // only holds the second half of the per-row cos / sin table (four floats per row).
  static constexpr bool kSingleBuffer = kRows == kWR && (TW::kCB == 2 || (TW::kK == 3 && DDD_K3_WAVES == 3));
  static constexpr int kHBStride = kSingleBuffer ? 4 : TW::kHS;
  static constexpr int kHBPad = kSingleBuffer ? 0 : TW::kC;      // float offset of a row's padding in hB
  float hA[kRows * TW::kHS];
  float hB[kRows * kHBStride];
  float u[kRows];
  float un[kRows == kWR ? 1 : kRows];   // u / standard_deviation (input-layer operand; one-wave
                                        // groups feed the input layer by lane permutes)
  float flux[kRows == kWR ? 1 : kRows];  // one-wave groups exchange flux by shuffle
  float2 pm[kPmMax + 8];          // per (sample, mode): a sin(psi), a cos(psi); + read-ahead padding
  float fk[kFkMax];               // per (sample, k): sums of pm over modes with that k
  // [0,4): bias rows [d][stencil]; then one null-space row per output channel
  float tab[tab_rows(kWide) * flavour_stencil(kWide)];
  // multi-wave groups: "some state of this group held a NaN" (eval_rhs: NaN through relu);
  // sticky for the launch -- a NaN state stays NaN --, cleared by setup_weights
  int nan_flag;
};
static_assert(sizeof(Shared<256>) <= 80 * 1024, "2 x 256-row workgroups per CU");
static_assert(sizeof(Shared<64>) <= 20 * 1024, "8 x 64-row workgroups per CU");
static_assert(8 * sizeof(Shared<64>) <= 158 * 1024, "2 x four-group workgroups per CU, with slack");
static_assert(sizeof(Shared<64, 32>) <= 40 * 1024, "4 x two-wave workgroups per CU");
static_assert(sizeof(Shared<64, 64, true>) <= 20 * 1024 && sizeof(Shared<256, 64, true>) <= 80 * 1024,
              "wide flavour: 8 x 64-row / 2 x 256-row workgroups per CU");
static_assert(sizeof(Shared<64, 64, false, Tower<5, 2>>) <= 20 * 1024 &&
              sizeof(Shared<256, 64, false, Tower<5, 2>>) <= 160 * 1024,
              "64-filter towers: 8 x 64-row (single activation buffer) / 1 x 256-row workgroups per CU");

// A one-wave row group (kRows == kWR) may be one of several INDEPENDENT groups
// sharing a workgroup (substep_quad_kernel): its thread index is the lane, and
// its "barrier" must not involve the other wavefronts -- within one wavefront
// LDS operations execute in order, so waiting for them to land is all a barrier
// means (and all `__syncthreads()` compiles to in a 64-thread workgroup).
template <int kRows, int kWR>
__device__ __forceinline__ int group_tid() {
  return kRows == kWR ? ((int)threadIdx.x & 63) : (int)threadIdx.x;
}
template <int kRows, int kWR>
__device__ __forceinline__ void group_barrier() {
  if (kRows == kWR) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  else __syncthreads();
}

// Value the optimiser must treat as unknown: stops loop-invariant code motion
// from hoisting per-evaluation index math and loads out of the time loop (where
// they would pin dozens of VGPRs for the whole kernel).
__device__ __forceinline__ int opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}
// Same for a wave-uniform value that must STAY scalar.  Without it the
// run-time-parameterised kernels hoist every predicate derived from the model
// parameters (g < G, d < D, the per-channel derivative selectors ...) out of the
// time loop as 64-bit lane masks -- ~60 SGPR pairs, spilled to VGPR lanes and
// read back with v_readlane (~100 VALU-issue slots per evaluation, which the f32
// MFMA stream pays for one to one); opaque, they are recomputed on the scalar
// unit inside each evaluation, which costs nothing.
__device__ __forceinline__ int opaque_scalar(int x) {
  asm volatile("" : "+s"(x));
  return x;
}
__device__ __forceinline__ unsigned long long opaque_scalar(unsigned long long x) {
  asm volatile("" : "+s"(x));
  return x;
}

// Compile-time specialisation of the evaluation on the equation (kEq >= 0):
// derivative count, stencil width and flux form become constants, the net is
// the default relu tower with the projection folded into the output layer
// where D <= 2.  kEq = -1 keeps every parameter a run-time (wave-uniform)
// value.  The host picks a specialised instantiation only when the model
// matches these assumptions (capi.hip: spec_equation).
__host__ __device__ constexpr int spec_derivs(int eq) {
  return (eq == EQ_KS || eq == EQ_KS_CONS) ? 3 : 2;
}
__host__ __device__ constexpr bool spec_flux_form(int eq) {
  return eq == EQ_BURGERS_CONS || eq == EQ_KDV_CONS || eq == EQ_KS_CONS;
}
__host__ __device__ constexpr int spec_stencil(int eq) { return spec_flux_form(eq) ? 6 : 7; }
// Only Burgers adds forcing(t) in finalize_time_derivative (equations.py:276-277):
// the other specialised kernels carry no forcing code at all.
__host__ __device__ constexpr bool spec_forced_family(int eq) {
  return eq == EQ_BURGERS || eq == EQ_BURGERS_CONS;
}
__host__ __device__ constexpr int prio_mode(int eq) {
  return DDD_PRIO_PHASES >= 0 ? DDD_PRIO_PHASES : eq < 0 ? 0 : spec_forced_family(eq) ? 2 : 5;
}
// Null-space sizes of the accuracy layers at the defaults the specialised
// kernels assume (polynomial_accuracy_order 1, coefficient_grid_min_size 6;
// checked by capi.hip: spec_equation): centred 7-point stencils: derivative
// order 1 -> 5, 2 -> 4, 3 -> 3, 4 -> 2; staggered 6-point: 0 -> 5, 1 -> 4,
// 2 -> 3, 3 -> 2.  Burgers (u_x, u_xx) 5 + 4, KdV (u_x, u_xxx) 5 + 3, KS 5 + 4 + 2,
// same for the flux forms.
__host__ __device__ constexpr int spec_in_size(int eq, int d) {
  return d == 0 ? 5
         : d == 1 ? ((eq == EQ_KDV || eq == EQ_KDV_CONS) ? 3 : 4)
         : d == 2 ? 2 : 0;
}
__host__ __device__ constexpr int spec_net_channels(int eq) {
  return spec_in_size(eq, 0) + spec_in_size(eq, 1) + (spec_derivs(eq) > 2 ? spec_in_size(eq, 2) : 0);
}
// derivative fed by output channel c of the (unfolded) conv tower
__host__ __device__ constexpr int spec_channel_deriv(int eq, int c) {
  return c < spec_in_size(eq, 0) ? 0 : c < spec_in_size(eq, 0) + spec_in_size(eq, 1) ? 1 : 2;
}
// Fold the projection (coeff = bias + net @ nullspace) into the output layer
// when that does not cost a channel group of four: the folded layer emits D x G
// coefficient channels, the plain one the null-space coordinates and leaves
// ~7 FMAs per channel to the epilogue.  Flux-form Burgers: 12 vs 9 channels = 3
// groups either way -> folded.  KdV: 12-14 vs 8 -> 2 groups unfolded.  Non-flux
// Burgers: 14 vs 9 -> 3 groups unfolded.  KS (D = 3): never folded.
__host__ __device__ constexpr bool spec_folded(int eq) {
  return spec_derivs(eq) <= 2 &&
         (spec_derivs(eq) * spec_stencil(eq) + 3) / 4 <= (spec_net_channels(eq) + 3) / 4;
}
// Output channels of the specialised kernels' last layer, in groups of four
// (final_layer4).
__host__ __device__ constexpr int spec_fin_channels(int eq) {
  return spec_folded(eq) ? spec_derivs(eq) * spec_stencil(eq) : spec_net_channels(eq);
}
__host__ __device__ constexpr int spec_fin_groups(int eq) { return (spec_fin_channels(eq) + 3) / 4; }
__host__ __device__ constexpr int fin4_regs(int groups) { return (kFin4K * groups + 15) / 16; }
// Run-time kernels issue their live channel groups as a head chunk followed by
// pairs: an even count has no head (0), a single group is its own head (1), any
// other odd count starts with three interleaved groups.
__host__ __device__ constexpr int rt_head_groups(int groups) {
  return groups % 2 == 0 ? 0 : groups == 1 ? 1 : 3;
}

// Entry `s` of a per-stage constant array that sits in the kernel-argument segment
// (StageConsts): scalar compares + selects on SGPRs instead of an indexed scalar LOAD
// (and its wait) inside the time loop.
// (written as scalar compare + select instructions: as C++ selects the compiler turns the
// four kernel-argument loads back into ONE load at a selected offset, or -- through
// opaque copies -- into branches or a scratch array, all of which stall the wavefront.)
__device__ __forceinline__ float pick4(int s, float v0, float v1, float v2, float v3) {
  float r;
  asm("s_cmp_eq_u32 %1, 1\n\ts_cselect_b32 %0, %3, %2\n\t"
      "s_cmp_eq_u32 %1, 2\n\ts_cselect_b32 %0, %4, %0\n\t"
      "s_cmp_eq_u32 %1, 3\n\ts_cselect_b32 %0, %5, %0"
      : "=&s"(r) : "s"(s), "s"(v0), "s"(v1), "s"(v2), "s"(v3) : "scc");
  return r;
}
__device__ __forceinline__ double pick4(int s, double v0, double v1, double v2, double v3) {
  double r;
  asm("s_cmp_eq_u32 %1, 1\n\ts_cselect_b64 %0, %3, %2\n\t"
      "s_cmp_eq_u32 %1, 2\n\ts_cselect_b64 %0, %4, %0\n\t"
      "s_cmp_eq_u32 %1, 3\n\ts_cselect_b64 %0, %5, %0"
      : "=&s"(r) : "s"(s), "s"(v0), "s"(v1), "s"(v2), "s"(v3) : "scc");
  return r;
}
// kSgpr = false: a plain indexed read (one scalar load per use) -- the run-time-parameterised
// kernels have no 16 SGPRs to spare for the whole time loop (their model scalars already
// spill: with the selects they went from 42 to 74 SGPR spills, each a v_readlane per use).
template <typename T, bool kSgpr = true>
struct StagePick {
  const T (&v)[kMaxStages];
  __device__ __forceinline__ explicit StagePick(const T (&v_)[kMaxStages]) : v(v_) {}
  __device__ __forceinline__ T at(int s) const {
    if constexpr (kSgpr) return pick4(s, v[0], v[1], v[2], v[3]);
    else return v[s];
  }
};
// a[s] h and b[s] h in the state's type
template <typename ST, bool kSgpr>
__device__ __forceinline__ StagePick<ST, kSgpr> stage_ah(const StageConsts& sc) {
  if constexpr (sizeof(ST) == 4) return StagePick<ST, kSgpr>(sc.ah);
  else return StagePick<ST, kSgpr>(sc.ahd);
}
template <typename ST, bool kSgpr>
__device__ __forceinline__ StagePick<ST, kSgpr> stage_bh(const StageConsts& sc) {
  if constexpr (sizeof(ST) == 4) return StagePick<ST, kSgpr>(sc.bh);
  else return StagePick<ST, kSgpr>(sc.bhd);
}

struct Lane {
  int row;       // row inside the workgroup this lane owns in VALU phases
  int base;      // first row of the row's sample
  int pos;       // grid index inside the sample
  int sl;        // sample index inside the workgroup
  int valid;     // row maps to a real sample of the batch
  int active;    // valid and this lane owns the row's stores
  long gidx;     // sample * N + pos (global element index), valid if active
  int wave, lane;
  int owner;     // this lane stores the row's results (kWR = 32: lanes 0..31 only)
  int rows_used; // samples_per_group * N
  float inv_n;
};

// Exact for rows < 256, N >= 8: (row + 0.5) / N is never within 2e-3 of an
// integer, far above float rounding.
__device__ __forceinline__ int row_sample(int row, float inv_n) {
  return (int)(((float)row + 0.5f) * inv_n);
}

template <int kRows, int kWR>
__device__ __forceinline__ Lane make_lane(const DevParams& p, int batch, int tid, int block) {
  Lane ln;
  ln.wave = tid >> 6;
  ln.lane = tid & 63;
  // kWR = 32: both half-waves carry the same 32 rows through the VALU phases
  // (identical values), the lower half owns the stores
  ln.row = ln.wave * kWR + (ln.lane & (kWR - 1));
  ln.owner = ln.lane < kWR;
  const int spg = kRows / p.N;
  ln.rows_used = spg * p.N;
  ln.inv_n = 1.0f / (float)p.N;
  if (kRows == 64 || (p.N & (p.N - 1)) == 0) {   // N | 64 is a power of two; else wave-uniform
    ln.pos = ln.row & (p.N - 1);
    ln.base = ln.row - ln.pos;
    ln.sl = ln.row >> (31 - __builtin_clz(p.N));
  } else {
    ln.sl = row_sample(ln.row, ln.inv_n);
    ln.base = ln.sl * p.N;
    ln.pos = ln.row - ln.base;
  }
  if (ln.row >= ln.rows_used) {
    // Spare rows (256 is not a multiple of N): read like row 0 of sample 0 so
    // every LDS index stays in range; results are never stored.
    ln.sl = 0; ln.base = 0; ln.pos = 0;
  }
  const long sample = (long)block * spg + ln.sl;
  ln.valid = (ln.row < ln.rows_used) && (sample < batch);
  ln.active = ln.valid && ln.owner;
  ln.gidx = sample * p.N + ln.pos;
  return ln;
}

// Row of grid point (pos + off) mod N of the sample starting at `base`.
// |off| < N is guaranteed by the host (N >= 8).
__device__ __forceinline__ int wrap_row(int base, int pos, int off, int n) {
  int q = pos + off;
  q = q < 0 ? q + n : q;
  q = q >= n ? q - n : q;
  return base + q;
}

// Same for an arbitrary row of the workgroup (tile rows differ from ln.row).
__device__ __forceinline__ int tile_src_row(const Lane& ln, int trow, int off, int n) {
  const int base = row_sample(trow, ln.inv_n) * n;
  const int src = wrap_row(base, trow - base, off, n);
  return trow < ln.rows_used ? src : trow;
}

// Rows of grid points (pos - 2 .. pos + 2) mod N for tile row `trow`: the five
// conv taps.  One sample lookup per tile row, then one compare/select per tap
// (the sign of each offset is known at compile time).
template <bool kPow2>
__device__ __forceinline__ void tap_rows(const Lane& ln, int trow, int n,
                                         int (&rows)[kKW]) {
  if (kPow2 || (n & (n - 1)) == 0) {
    // N a power of two (divides the group's rows, no spare rows): samples
    // start at multiples of N, so the wrap is an AND and the base an OR:
    // one v_add + one v_and_or per tap.  (wave-uniform branch)
    const int mask = n - 1;
    const int base = trow & ~mask;
    rows[0] = ((trow - 2) & mask) | base;
    rows[1] = ((trow - 1) & mask) | base;
    rows[2] = trow;
    rows[3] = ((trow + 1) & mask) | base;
    rows[4] = ((trow + 2) & mask) | base;
    return;
  }
  const int base = row_sample(trow, ln.inv_n) * n;
  const int pos = trow - base;
  int qm2 = pos - 2; qm2 = qm2 < 0 ? qm2 + n : qm2;
  int qm1 = pos - 1; qm1 = qm1 < 0 ? qm1 + n : qm1;
  int qp1 = pos + 1; qp1 = qp1 >= n ? qp1 - n : qp1;
  int qp2 = pos + 2; qp2 = qp2 >= n ? qp2 - n : qp2;
  const bool live = trow < ln.rows_used;   // spare rows read themselves
  rows[0] = live ? base + qm2 : trow;
  rows[1] = live ? base + qm1 : trow;
  rows[2] = trow;
  rows[3] = live ? base + qp1 : trow;
  rows[4] = live ? base + qp2 : trow;
}

// The MFMA-packed weight arrays (w_input, w_hidden, w_final4) hold one value per
// (row, lane); in memory, four rows to a float4 per lane -- row s of lane l at
// ((s >> 2) * 64 + l) * 4 + (s & 3), rows zero-padded to a multiple of four --,
// so a wavefront fetches four rows with ONE 16-byte load per lane.  A wavefront
// keeps at most 63 loads in flight: as 120 one-dword loads the resident weights
// were two full memory round trips at the start of every launch
// (profiles/tools/substep_wave_trace.py), as 31 quad loads they are one.
template <int NR>
__device__ __forceinline__ void load_rows4(const float* __restrict__ base, int lane,
                                           float (&w)[NR]) {
  const float4* __restrict__ q = reinterpret_cast<const float4*>(base) + lane;
#pragma unroll
  for (int k = 0; k < (NR + 3) / 4; ++k) {
    const float4 v = q[k * 64];
    w[4 * k] = v.x;
    if (4 * k + 1 < NR) w[4 * k + 1] = v.y;
    if (4 * k + 2 < NR) w[4 * k + 2] = v.z;
    if (4 * k + 3 < NR) w[4 * k + 3] = v.w;
  }
}
constexpr int padded_rows4(int rows) { return (rows + 3) / 4 * 4; }

__device__ __forceinline__ void load_hidden(const DevParams& p, int hidden_index,
                                            int lane, float (&w)[kHidSteps]) {
  load_rows4<kHidSteps>(p.w_hidden + (size_t)hidden_index * padded_rows4(kHidSteps) * 64, lane, w);
}

#define DDD_MFMA32(A, B, C) __builtin_amdgcn_mfma_f32_32x32x2f32((A), (B), (C), 0, 0, 0)

__device__ __forceinline__ float fast_tanh(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_exp2f(ax * -2.885390081777927f);   // exp(-2 |x|)
  const float q = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
  // |x| < 1/4: 1 - t cancels (the 2e-7 ABSOLUTE bound of the quotient is 2e-7 / |x|
  // relative: 2e-5 at 1e-2, and results quantised to 6e-8 below 1e-6; tf.tanh keeps
  // relative accuracy).  There the odd Taylor polynomial
  // |x| (1 - x^2 / 3 + 2 x^4 / 15 - 17 x^6 / 315 + 62 x^8 / 2835) is within 1e-8 relative
  // (next term 1382 x^10 / 155925); the quotient is within 2e-7 relative from 1/4 up.
  const float x2 = x * x;
  const float poly =
      ax * fmaf(x2, fmaf(x2, fmaf(x2, fmaf(x2, 62.0f / 2835.0f, -17.0f / 315.0f), 2.0f / 15.0f),
                         -1.0f / 3.0f), 1.0f);
  return copysignf(ax < 0.25f ? poly : q, x);
}

__device__ __forceinline__ void activate16(f32x16& acc, int act) {
  if (act == ACT_RELU) {
#if DDD_RELU_CLAMP
    // TWO elements per instruction: the [0, 1] output clamp of a packed add of zero, on
    // activations that carry the factor 2^-kReluShift (dev_params.h; the host packs the
    // weights accordingly) -- gfx950 has no packed f32 max, and every VALU instruction
    // of a layer boundary is time the matrix pipe stands still.  clamp(NaN) = 0 (DX10
    // clamp mode, the HSA default), like v_max_f32 0, NaN.  The inline constant 0 is 0 in
    // both halves whatever the operand's op_sel semantics.
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      f32x2 v{acc[r], acc[r + 1]}, y;
      asm("v_pk_add_f32 %0, %1, 0 clamp" : "=v"(y) : "v"(v));
      acc[r] = y[0];
      acc[r + 1] = y[1];
    }
#else
    // ONE v_max_f32 per element.  fmaxf / fmed3 builtins make the compiler
    // prepend a canonicalising v_max x, x (it cannot prove an MFMA result is
    // not a signalling NaN), doubling the VALU work of every layer boundary,
    // where the matrix pipe is idle; max(0, NaN) = 0 either way.
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float y;
      asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(acc[r]));
      acc[r] = y;
    }
#endif
  } else if (act == ACT_RELU6) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = fminf(fmaxf(acc[r], 0.0f), 6.0f);
  } else if (act == ACT_TANH) {
    // tanh(x) = sign(x) (1 - t) / (1 + t), t = exp(-2 |x|): one v_exp_f32 and one v_rcp_f32
    // (+ a five-term polynomial selected below |x| = 1/4: fifteen instructions) instead of
    // the library's ~30 per element -- 192 elements per lane and evaluation in a four-layer
    // net.  Absolute error <= 2e-7 (argument rounding 1.7e-7 x t <= 0.37, one ulp each of exp2
    // and rcp on values <= 1, two roundings) AND relative error <= 3e-7 (the polynomial branch),
    // measured against float64 tanh in tests/test_cpu_mfma_emulation.py; NaN propagates,
    // +-Inf -> +-1.
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = fast_tanh(acc[r]);
  } else if (act == ACT_SOFTPLUS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = apply_activation(acc[r], ACT_SOFTPLUS);
  } else if (act == ACT_ELU) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = apply_activation(acc[r], ACT_ELU);
  }
}

// D of a 32x32 tile -> LDS rows.  Lane l holds position l & 31 and, in register
// r, out-channel (r & 3) + 8 (r >> 2) + 4 (l >> 5): four ds_write_b128.
__device__ __forceinline__ void store_tile32(float* out, int trow, int half,
                                             const f32x16& acc) {
  float* orow = out + trow * kHS + 4 * half;
#pragma unroll
  for (int qd = 0; qd < 4; ++qd)
    *reinterpret_cast<float4*>(orow + 8 * qd) = make_float4(
        acc[4 * qd + 0], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]);
}

// The same with the row's LDS byte offset ((row * kHS + 4 half) * 4) already formed
// (Resident::st_off: lane == row never changes inside a launch).
__device__ __forceinline__ void store_tile32_at_bytes(float* out, int byte_off, const f32x16& acc) {
  char* orow = reinterpret_cast<char*>(out) + byte_off;
#pragma unroll
  for (int qd = 0; qd < 4; ++qd)
    *reinterpret_cast<float4*>(orow + 32 * qd) = make_float4(
        acc[4 * qd + 0], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]);
}

// x on lanes 0..31, 1.0 on lanes 32..63 (the bias row of the input layer's last MFMA
// step): ONE v_cndmask against a constant lane mask in an SGPR pair instead of
// and + compare + select on the lane index.
// (the inverse-ballot builtin, not inline asm: the select feeds an MFMA operand, and the
// compiler's hazard recognizer does not see VALU instructions hidden in asm statements --
// an asm v_cndmask here left the MFMA behind it reading a stale register.)
__device__ __forceinline__ float upper_half_one(float x) {
  return __builtin_amdgcn_inverse_ballot_w64(0xffffffff00000000ull) ? 1.0f : x;
}

// Input layer 1 -> 32 for this wave's two 32-row tiles (3 MFMA steps each).
//   A: lane l supplies W1[out = l & 31][k = 2 s + (l >> 5)]  (k = tap; k = 5: bias)
//   B: lane l supplies un[(pos(l & 31) + k - 2) mod N], un = u / std   (k = 5: 1.0)
// kShfl (one-wave groups: row == lane): operands come from the lanes' `un`
// registers by ds_bpermute instead of an LDS write + read round trip.
// kAddr (specialised one-wave integrators): `bperm` holds the six ds_bpermute
// byte addresses, resident for the launch; all six permutes are issued before
// the first MFMA (one LDS-crossbar latency instead of six).
template <int kWR, bool kShfl, bool kAddr = false>
__device__ __forceinline__ void input_layer(const DevParams& p, const Lane& ln,
                                            const float* __restrict__ us, float un,
                                            float* __restrict__ out,
                                            const float (&w)[kInSteps],
                                            const int (&rows)[2][kKW], int act,
                                            const int (*bperm)[3] = nullptr, int st_off = 0) {
  constexpr int kT = kWR / 32;
  const int j = ln.lane & 31;
  const int half = ln.lane >> 5;
  f32x16 acc[kT];
  float b0[kT], b1[kT], b2[kT];
#pragma unroll
  for (int t = 0; t < kT; ++t) {
    if (kAddr && kShfl) {
      const int uni = __float_as_int(un);
      b0[t] = __int_as_float(__builtin_amdgcn_ds_bpermute(bperm[t][0], uni));
      b1[t] = __int_as_float(__builtin_amdgcn_ds_bpermute(bperm[t][1], uni));
      b2[t] = __int_as_float(__builtin_amdgcn_ds_bpermute(bperm[t][2], uni));
    } else if (kAddr) {
      // four-wave groups: the same resident byte addresses, into Shared::un
      const char* __restrict__ ub = reinterpret_cast<const char*>(us);
      b0[t] = *reinterpret_cast<const float*>(ub + bperm[t][0]);
      b1[t] = *reinterpret_cast<const float*>(ub + bperm[t][1]);
      b2[t] = *reinterpret_cast<const float*>(ub + bperm[t][2]);
    } else {
      const int r0 = half ? rows[t][1] : rows[t][0];              // taps 0 / 1
      const int r1 = half ? rows[t][3] : rows[t][2];              // taps 2 / 3
      const int r2 = rows[t][4];                                  // tap 4 / bias row
      b0[t] = kShfl ? __shfl(un, r0, 64) : us[r0];
      b1[t] = kShfl ? __shfl(un, r1, 64) : us[r1];
      b2[t] = kShfl ? __shfl(un, r2, 64) : us[r2];
    }
    b2[t] = kAddr ? upper_half_one(b2[t]) : (half ? 1.0f : b2[t]);
  }
  if (kAddr && kShfl) __builtin_amdgcn_sched_group_barrier(0x080, 3 * kT, 0);   // the permutes first
#pragma unroll
  for (int t = 0; t < kT; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
  }
#pragma unroll
  for (int t = 0; t < kT; ++t) acc[t] = DDD_MFMA32(w[0], b0[t], acc[t]);
#pragma unroll
  for (int t = 0; t < kT; ++t) acc[t] = DDD_MFMA32(w[1], b1[t], acc[t]);
#pragma unroll
  for (int t = 0; t < kT; ++t) acc[t] = DDD_MFMA32(w[2], b2[t], acc[t]);
#pragma unroll
  for (int t = 0; t < kT; ++t) {
    activate16(acc[t], act);
    if (kAddr) store_tile32_at_bytes(out, st_off + t * 32 * kHS * 4, acc[t]);
    else store_tile32(out, ln.wave * kWR + t * 32 + j, half, acc[t]);
  }
}

// One hidden layer for this wave's 32-row tiles (two, or one when kWR = 32).
//   A operand (weights): lane l supplies W[out = l & 31][k = 2 s + (l >> 5)]
//   B operand (acts)   : lane l supplies h[k = 2 s + (l >> 5)][position = l & 31]
//   reduction index    : step s = 16 tap + jj, half = l >> 5  <->  (tap, cin = 16 half + jj)
// The 16 floats a lane needs per tap are four ds_read_b128; the read of group
// g + 1 is issued before the four MFMAs of group g (software prefetch).
// kByteOffsets: `rows` already holds the LDS byte offsets of the operand rows
// (row * 144 + 64 * half; kept resident by the specialised one-wave integrators).
// kHalfNet (HalfTower: called with kWR = 32, i.e. ONE tile): the tile's output rows 16..31 are
// the channels of position 32 + j -- registers 8..15 go to that row.
template <int kWR, bool kByteOffsets = false, bool kHalfNet = false, int kPrio = 0>
__device__ __forceinline__ void hidden_layer(const DevParams& p, const Lane& ln,
                                             const float* __restrict__ in,
                                             float* __restrict__ out,
                                             const float (&w)[kHidSteps],
                                             const int (&rows)[2][kKW], int act, int st_off = 0) {
  constexpr int kT = kWR / 32;   // 32-row tiles of this wave, advanced together
  const int j = ln.lane & 31;
  const int half = ln.lane >> 5;
  // 32-bit byte offsets (not pointers: pointer arrays make the compiler carry
  // 64-bit address arithmetic for what ends up as an LDS offset)
  int rowo[kT][kKW];
#pragma unroll
  for (int t = 0; t < kT; ++t)
#pragma unroll
    for (int tap = 0; tap < kKW; ++tap)
      rowo[t][tap] = kByteOffsets ? rows[t][tap]
                                  : (int)__umul24((unsigned)rows[t][tap], (unsigned)(kHS * 4)) +
                                        64 * half;   // bytes; rows < 256
  const auto operand = [&](int t, int tap, int q) {   // one v_mad_u32_u24 per row, q in the DS offset field
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(in) + rowo[t][tap] + 16 * q);
  };
  f32x16 acc[kT];
  float4 cur[kT], nxt[kT];
#pragma unroll
  for (int t = 0; t < kT; ++t) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    cur[t] = operand(t, 0, 0);
  }
  __builtin_amdgcn_sched_group_barrier(0x100, kT, 0);     // the reads of group 0
#pragma unroll
  for (int g = 0; g < 20; ++g) {
    if (kPrio >= 3 && g == 1) __builtin_amdgcn_s_setprio(0);
    if (kPrio >= 3 && g == 19) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int t = 0; t < kT; ++t) {
      nxt[t] = cur[t];
      if (g + 1 < 20) nxt[t] = operand(t, (g + 1) >> 2, (g + 1) & 3);
    }
#pragma unroll
    for (int t = 0; t < kT; ++t) acc[t] = DDD_MFMA32(w[4 * g + 0], cur[t].x, acc[t]);
#pragma unroll
    for (int t = 0; t < kT; ++t) acc[t] = DDD_MFMA32(w[4 * g + 1], cur[t].y, acc[t]);
#pragma unroll
    for (int t = 0; t < kT; ++t) acc[t] = DDD_MFMA32(w[4 * g + 2], cur[t].z, acc[t]);
#pragma unroll
    for (int t = 0; t < kT; ++t) acc[t] = DDD_MFMA32(w[4 * g + 3], cur[t].w, acc[t]);
#pragma unroll
    for (int t = 0; t < kT; ++t) cur[t] = nxt[t];
    // schedule: read group g+1 of every tile, then the MFMAs of group g
    __builtin_amdgcn_sched_group_barrier(0x100, kT, 0);       // DS reads
    __builtin_amdgcn_sched_group_barrier(0x008, 4 * kT, 0);   // MFMAs
  }
#pragma unroll
  for (int t = 0; t < kT; ++t) {
    acc[t] = DDD_MFMA32(w[80], 1.0f, acc[t]);   // bias row: k = 160 carries b[out]
  }
#pragma unroll
  for (int t = 0; t < kT; ++t) {
    activate16(acc[t], act);
    if constexpr (kHalfNet) {
      static_assert(kByteOffsets && kT == 1, "block-diagonal nets: one tile, resident offsets");
      char* orow = reinterpret_cast<char*>(out) + st_off;   // row j, + 16 half
#pragma unroll
      for (int qd = 0; qd < 4; ++qd)   // registers 4 qd ..: channels (qd & 1) 8 + 4 half + 0..3 of row j + 32 (qd >> 1)
        *reinterpret_cast<float4*>(orow + (qd >> 1) * 32 * kHS * 4 + 32 * (qd & 1)) = make_float4(
            acc[t][4 * qd + 0], acc[t][4 * qd + 1], acc[t][4 * qd + 2], acc[t][4 * qd + 3]);
    } else if (kByteOffsets) store_tile32_at_bytes(out, st_off + t * 32 * kHS * 4, acc[t]);
    else store_tile32(out, ln.wave * kWR + t * 32 + j, half, acc[t]);
  }
}

// ---------------------------------------------------------------------------
// FOUR wavefronts per 64-row group (kRows = 64, kWR = 16; per-equation integrators): one
// sample of a small ensemble on all four SIMDs of a CU.  The reference's callers integrate
// tens to hundreds of samples (scripts/run_evaluation.py:212-221,
// create_baseline_data.py:119-130); with one 64-row wavefront per sample three quarters of
// the SIMDs idle at 256 samples, with two 32-row wavefronts (kSplit) half of them.
//   * wavefront w = ph + 2 ch carries the 32 positions [32 ph, 32 ph + 32) x the 16 output
//     channels [16 ch, 16 ch + 16) of the input and hidden layers on v_mfma_f32_16x16x4_f32
//     (the same 32 FMA / cycle / SIMD as 32x32x2): 2 position tiles x 41 steps per hidden
//     layer instead of 162 steps of 32x32x2;
//   * the output layer also on 16x16x4: wavefront w computes all (<= 16) channels of ITS
//     OWN 16 rows [16 w, 16 w + 16) -- the rows it carries through the VALU phases --, so
//     the results return to lane == row through LDS inside the wavefront (no barrier);
//   * every accumulation chain runs in the order of the one-wavefront kernel -- hidden
//     layer: per tap c = 0, 16, 1, 17, ... (hidden_layer's step s = 16 tap + jj with
//     half = l >> 5), i.e. reduction slots [2 i, 16 + 2 i, 2 i + 1, 17 + 2 i] of step
//     8 tap + i; output layer: natural k = 32 tap + c --, so the bits are the
//     one-wavefront kernel's (tests/test_gpu_integrate.py: assert_array_equal);
//   * activations in LDS rows of 36 floats with channel c at float
//     4 (4 (c >> 4) + (c & 3)) + ((c & 15) >> 2): the B operands of both layers are two
//     aligned ds_read_b128 per tap (hidden layer, slot sg = l >> 4: blocks (sg & 1, sg >> 1)
//     and (sg & 1, (sg >> 1) + 2), elements alternating; output layer: blocks (0, sg), (1, sg)).
// A operands (capi.hip: pack_quad_weights): DevParams::w_quad = [2 ch][2] input rows,
// [2 ch][41] hidden rows, [41] output rows of 64 lanes, lane l = W[out = l & 15][slot l >> 4].
// ---------------------------------------------------------------------------
#define DDD_MFMA16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)
constexpr int kQuadHidSteps = 41, kQuadFinSteps = 41, kQuadInSteps = 2;
constexpr int kQuadRows = 2 * kQuadInSteps + 2 * kQuadHidSteps + kQuadFinSteps;   // rows of w_quad (4 + 82 + 41)
__host__ __device__ constexpr int quad_channel_float(int c) {   // position of channel c in an LDS row
  return 4 * (4 * (c >> 4) + (c & 3)) + ((c & 15) >> 2);
}

// relu etc. on one 16x16 tile (4 accumulator registers per lane)
__device__ __forceinline__ void activate4(f32x4& acc, int act) {
  if (act == ACT_RELU) {
#if DDD_RELU_CLAMP
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      f32x2 v{acc[r], acc[r + 1]}, y;
      asm("v_pk_add_f32 %0, %1, 0 clamp" : "=v"(y) : "v"(v));
      acc[r] = y[0];
      acc[r + 1] = y[1];
    }
#else
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float y;
      asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(acc[r]));
      acc[r] = y;
    }
#endif
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = apply_activation(acc[r], act);
  }
}

// D of a 16x16 tile -> LDS: lane l holds channels 16 ch + 4 (l >> 4) + r of position l & 15;
// st_off = byte offset of (row, channel 16 ch + 4 (l >> 4)): r advances the block (16 bytes)
__device__ __forceinline__ void store_tile16(float* out, int st_off, const f32x4& acc) {
  char* o = reinterpret_cast<char*>(out) + st_off;
#pragma unroll
  for (int r = 0; r < 4; ++r) *reinterpret_cast<float*>(o + 16 * r) = acc[r];
}

// input layer 1 -> 16 channels of this wavefront, two 16-position tiles
//   step 0: slots = taps 0..3; step 1: tap 4, bias (against 1.0), 0, 0
// in_off[t][0]: byte offset into Shared::un of tap (l >> 4)'s row, [t][1]: of tap 4's row
__device__ __forceinline__ void input_layer_quad(const float* __restrict__ un, float* __restrict__ out,
                                                 const float (&w)[kQuadInSteps],
                                                 const int (&in_off)[2][2], const int (&st_off)[2],
                                                 int act, int lane) {
  const char* ub = reinterpret_cast<const char*>(un);
  float b0[2], b1[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    b0[t] = *reinterpret_cast<const float*>(ub + in_off[t][0]);
    const float tap4 = *reinterpret_cast<const float*>(ub + in_off[t][1]);
    b1[t] = lane < 16 ? tap4 : 1.0f;   // (slots 2, 3 carry zero weights)
  }
  f32x4 acc[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int t = 0; t < 2; ++t) acc[t] = DDD_MFMA16(w[0], b0[t], acc[t]);
#pragma unroll
  for (int t = 0; t < 2; ++t) acc[t] = DDD_MFMA16(w[1], b1[t], acc[t]);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    activate4(acc[t], act);
    store_tile16(out, st_off[t], acc[t]);
  }
}

// hidden layer: this wavefront's 16 output channels x two 16-position tiles.
// hid_off[t][tap]: byte offset of block (sg & 1, sg >> 1) of the tap's row (the second
// block, (sg & 1, (sg >> 1) + 2), is 32 bytes on)
template <int kPrio = 0>
__device__ __forceinline__ void hidden_layer_quad(const float* __restrict__ in, float* __restrict__ out,
                                                  const float (&w)[kQuadHidSteps],
                                                  const int (&hid_off)[2][kKW], const int (&st_off)[2],
                                                  int act) {
  const char* ib = reinterpret_cast<const char*>(in);
  f32x4 acc[2];
  float4 ca[2], cb[2], na[2], nb[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    ca[t] = *reinterpret_cast<const float4*>(ib + hid_off[t][0]);
    cb[t] = *reinterpret_cast<const float4*>(ib + hid_off[t][0] + 32);
  }
  __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);     // the reads of tap 0
#pragma unroll
  for (int tap = 0; tap < kKW; ++tap) {
    if (kPrio >= 3 && tap == 1) __builtin_amdgcn_s_setprio(0);
    if (kPrio >= 3 && tap == kKW - 1) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      na[t] = ca[t]; nb[t] = cb[t];
      if (tap + 1 < kKW) {
        na[t] = *reinterpret_cast<const float4*>(ib + hid_off[t][tap + 1]);
        nb[t] = *reinterpret_cast<const float4*>(ib + hid_off[t][tap + 1] + 32);
      }
    }
    // steps 8 tap + i, i = 0..7: B = a.x, b.x, a.y, b.y, a.z, b.z, a.w, b.w; the two tiles
    // alternate (two independent accumulators: the 16x16x4 issues every 32 cycles, a
    // dependent one every 40)
#define DDD_QSTEP(I, V)                                                          \
    _Pragma("unroll") for (int t = 0; t < 2; ++t)                                \
      acc[t] = DDD_MFMA16(w[8 * tap + (I)], (V), acc[t]);
    DDD_QSTEP(0, ca[t].x) DDD_QSTEP(1, cb[t].x) DDD_QSTEP(2, ca[t].y) DDD_QSTEP(3, cb[t].y)
    DDD_QSTEP(4, ca[t].z) DDD_QSTEP(5, cb[t].z) DDD_QSTEP(6, ca[t].w) DDD_QSTEP(7, cb[t].w)
#undef DDD_QSTEP
#pragma unroll
    for (int t = 0; t < 2; ++t) { ca[t] = na[t]; cb[t] = nb[t]; }
    if (tap + 1 < kKW) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // reads of tap + 1 first
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);                      // the 16 MFMAs of this tap
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) acc[t] = DDD_MFMA16(w[40], 1.0f, acc[t]);   // bias: slot 0 carries b[out]
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    activate4(acc[t], act);
    store_tile16(out, st_off[t], acc[t]);
  }
}

// output layer for this wavefront's own 16 rows: D[channel 4 (l >> 4) + r][row l & 15],
// natural reduction order k = 32 tap + c, 4 k per step (slot sg: c = sg + 4 i).
// fin_off[tap]: byte offset of block (0, sg) of the tap's row (block (1, sg): 64 bytes on)
__device__ __forceinline__ f32x4 final_layer_quad(const float* __restrict__ in,
                                                  const float (&w)[kQuadFinSteps],
                                                  const int (&fin_off)[kKW]) {
  const char* ib = reinterpret_cast<const char*>(in);
  f32x4 acc{0.0f, 0.0f, 0.0f, 0.0f};
  float4 ca, cb, na, nb;
  ca = *reinterpret_cast<const float4*>(ib + fin_off[0]);
  cb = *reinterpret_cast<const float4*>(ib + fin_off[0] + 64);
  __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
  for (int tap = 0; tap < kKW; ++tap) {
    na = ca; nb = cb;
    if (tap + 1 < kKW) {
      na = *reinterpret_cast<const float4*>(ib + fin_off[tap + 1]);
      nb = *reinterpret_cast<const float4*>(ib + fin_off[tap + 1] + 64);
    }
    acc = DDD_MFMA16(w[8 * tap + 0], ca.x, acc);
    acc = DDD_MFMA16(w[8 * tap + 1], ca.y, acc);
    acc = DDD_MFMA16(w[8 * tap + 2], ca.z, acc);
    acc = DDD_MFMA16(w[8 * tap + 3], ca.w, acc);
    acc = DDD_MFMA16(w[8 * tap + 4], cb.x, acc);
    acc = DDD_MFMA16(w[8 * tap + 5], cb.y, acc);
    acc = DDD_MFMA16(w[8 * tap + 6], cb.z, acc);
    acc = DDD_MFMA16(w[8 * tap + 7], cb.w, acc);
    ca = na; cb = nb;
    if (tap + 1 < kKW) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
  }
  return DDD_MFMA16(w[40], 1.0f, acc);   // bias row: k = 160 against a constant 1
}

// ---- Tile16Tower: a net of <= 16 filters on 16x16x4 tiles, one wavefront per 64-row group ----
// perm[t][0]: ds_bpermute address of tap (l >> 4)'s row of tile t's position, [t][1]: of tap 4's
__device__ __forceinline__ void input_layer_t16(float un, float* __restrict__ out,
                                                const float (&w)[kInSteps], const int (&perm)[4][2],
                                                const int (&st_off)[4], int act, int lane) {
  const int uni = __float_as_int(un);
  float b0[4], b1[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    b0[t] = __int_as_float(__builtin_amdgcn_ds_bpermute(perm[t][0], uni));
    const float tap4 = __int_as_float(__builtin_amdgcn_ds_bpermute(perm[t][1], uni));
    b1[t] = lane < 16 ? tap4 : 1.0f;   // (slot 1: the bias row; slots 2, 3 carry zero weights)
  }
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = DDD_MFMA16(w[0], b0[t], acc[t]);
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = DDD_MFMA16(w[1], b1[t], acc[t]);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    activate4(acc[t], act);
    store_tile16(out, st_off[t], acc[t]);
  }
}

// hid_off[t][tap]: byte offset of floats 4 sg .. 4 sg + 3 of the tap's row of tile t's position
template <int kPrio = 0>
__device__ __forceinline__ void hidden_layer_t16(const float* __restrict__ in, float* __restrict__ out,
                                                 const float (&w)[kHidSteps], const int (&hid_off)[4][kKW],
                                                 const int (&st_off)[4], int act) {
  const char* ib = reinterpret_cast<const char*>(in);
  f32x4 acc[4];
  float4 cur[4], nxt[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    cur[t] = *reinterpret_cast<const float4*>(ib + hid_off[t][0]);
  }
  __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);     // the reads of tap 0
#pragma unroll
  for (int tap = 0; tap < kKW; ++tap) {
    if (kPrio >= 3 && tap == 1) __builtin_amdgcn_s_setprio(0);
    if (kPrio >= 3 && tap == kKW - 1) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      nxt[t] = cur[t];
      if (tap + 1 < kKW) nxt[t] = *reinterpret_cast<const float4*>(ib + hid_off[t][tap + 1]);
    }
    // steps 4 tap + e: slot sg carries channel 4 e + sg; four independent accumulators
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = DDD_MFMA16(w[4 * tap + 0], cur[t].x, acc[t]);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = DDD_MFMA16(w[4 * tap + 1], cur[t].y, acc[t]);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = DDD_MFMA16(w[4 * tap + 2], cur[t].z, acc[t]);
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = DDD_MFMA16(w[4 * tap + 3], cur[t].w, acc[t]);
#pragma unroll
    for (int t = 0; t < 4; ++t) cur[t] = nxt[t];
    if (tap + 1 < kKW) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);   // reads of tap + 1 first
    __builtin_amdgcn_sched_group_barrier(0x008, 16, 0);                      // the 16 MFMAs of this tap
  }
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = DDD_MFMA16(w[4 * kKW], 1.0f, acc[t]);   // bias: slot 0 carries b[out]
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    activate4(acc[t], act);
    store_tile16(out, st_off[t], acc[t]);
  }
}

// ---------------------------------------------------------------------------
// Towers other than 5 taps x 32 channels (Tower<7, 1>, <5, 2>, <7, 2>, <3, 1>): the
// same implicit GEMMs, generic in the tap count and in the number of 32-channel
// blocks, with every layer's weights STREAMED from L2 in the order the MFMAs
// consume them instead of living in registers (a 7-tap layer is 113 operand rows, a
// 64-filter layer 320 per lane).  Activations: LDS rows of kC + 4 floats.
// ---------------------------------------------------------------------------

// Rows of grid points (pos - K/2 .. pos + K/2) mod N for tile row `trow`.
template <class TW>
__device__ __forceinline__ void tap_rows_n(const Lane& ln, int trow, int n, bool pow2,
                                           int (&rows)[TW::kK]) {
  constexpr int kLeft = TW::kK / 2;
  if (pow2) {   // wave-uniform: samples start at multiples of N, no spare rows
    const int mask = n - 1, base = trow & ~mask;
#pragma unroll
    for (int k = 0; k < TW::kK; ++k) rows[k] = ((trow + k - kLeft) & mask) | base;
    return;
  }
  const int base = row_sample(trow, ln.inv_n) * n;
  const int pos = trow - base;
  const bool live = trow < ln.rows_used;   // spare rows read themselves
#pragma unroll
  for (int k = 0; k < TW::kK; ++k) {
    int q = pos + k - kLeft;               // |k - kLeft| <= 3 < N (N >= 8)
    q = q < 0 ? q + n : q;
    q = q >= n ? q - n : q;
    rows[k] = live ? base + q : trow;
  }
}

// D of a 32x32 tile of output block `blk` -> LDS rows of stride HS floats.
template <int HS>
__device__ __forceinline__ void store_tile32_at(float* out, int trow, int blk, int half,
                                                const f32x16& acc) {
  float* orow = out + trow * HS + 32 * blk + 4 * half;
#pragma unroll
  for (int qd = 0; qd < 4; ++qd)
    *reinterpret_cast<float4*>(orow + 8 * qd) = make_float4(
        acc[4 * qd + 0], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]);
}

// Input layer 1 -> kC: (K + 1) / 2 steps per output block (K taps + the bias row).
//   A (DevParams::w_input, [block][step][lane]): W1[out = 32 blk + (l & 31)][k = 2 s + (l >> 5)],
//      k < K: tap k, k = K: bias
//   B: un[(pos(l & 31) + k - K/2) mod N] for k < K, 1 for k = K
// kShfl (one-wave groups): the operands come from the lanes' registers.
template <class TW, int kWR, bool kShfl>
__device__ __forceinline__ void input_layer_big(const DevParams& p, const Lane& ln,
                                                const float* __restrict__ us, float un,
                                                float* __restrict__ out,
                                                const int (&rows)[2][TW::kK], int act) {
  constexpr int kT = kWR / 32, kS = TW::kInSteps, kLast = TW::kK - 1;
  const int j = ln.lane & 31, half = ln.lane >> 5;
  float w[TW::kCB][kS];
  const float* __restrict__ wsrc = p.w_input + opaque(ln.lane);
#pragma unroll
  for (int h = 0; h < TW::kCB; ++h)
#pragma unroll
    for (int s = 0; s < kS; ++s) w[h][s] = wsrc[(h * kS + s) * 64];
  float b[kT][kS];
#pragma unroll
  for (int t = 0; t < kT; ++t) {
#pragma unroll
    for (int s = 0; s < kS; ++s) {
      // taps 2 s / 2 s + 1 by half-wave; the last step pairs tap K - 1 with the bias row
      const int r = (2 * s < kLast) ? (half ? rows[t][2 * s + 1 < TW::kK ? 2 * s + 1 : kLast]
                                            : rows[t][2 * s])
                                    : rows[t][kLast];
      b[t][s] = kShfl ? __shfl(un, r, 64) : us[r];
    }
    b[t][kS - 1] = half ? 1.0f : b[t][kS - 1];
  }
#pragma unroll
  for (int h = 0; h < TW::kCB; ++h) {
    f32x16 acc[kT];
#pragma unroll
    for (int t = 0; t < kT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
#pragma unroll
    for (int s = 0; s < kS; ++s)
#pragma unroll
      for (int t = 0; t < kT; ++t) acc[t] = DDD_MFMA32(w[h][s], b[t][s], acc[t]);
#pragma unroll
    for (int t = 0; t < kT; ++t) {
      activate16(acc[t], act);
      store_tile32_at<TW::kHS>(out, ln.wave * kWR + t * 32 + j, h, half, acc[t]);
    }
  }
}

// One hidden layer kC -> kC for this wave's two 32-row tiles.  Reduction step
// s = (tap kCB + cb) 16 + jj pairs input channels 32 cb + jj (lower half-wave) and
// 32 cb + 16 + jj (upper); four steps = one float4 of weights per output block
// (DevParams::w_hidden: [layer][group][block][lane] float4, then the bias rows) and one
// ds_read_b128 of activations per tile.  Weights are requested kStreamAhead groups
// (8-16 MFMAs = 512-1024 cycles each) before their MFMAs, activations one group ahead.
constexpr int kStreamAhead = 2;
// The FIRST hidden layer's leading weight groups stay in registers for the whole launch
// (Resident::hw): the stream's first requests were a full L2 round trip that every evaluation
// waited for at the input -> hidden boundary (phase trace, round 5: 3 400 cycles exposed per
// evaluation on the 3-tap tower against 1 700 on the default tower with its resident
// weights).  3 taps x 32 filters: the whole layer (12 float4 + bias = 49 registers, no
// stream left); the bigger towers: as many groups as the stream runs ahead.
template <class TW>
__host__ __device__ constexpr int resident_groups() {
  return TW::kDefault ? 0 : (TW::kHidGroups * TW::kCB <= 12 && DDD_K3_RESIDENT_ALL ? TW::kHidGroups : kStreamAhead);
}
constexpr int kResidentQuads = 12;   // float4 registers of Resident::hw

template <class TW, int kT>
struct StreamState {
  f32x16 acc[TW::kCB][kT];
  float4 wbuf[kStreamAhead + 1][TW::kCB];
  float4 bbuf[2][kT];
  int rowo[kT][TW::kK];              // LDS byte offsets of the operand rows (+ 64 half)
  const float4* __restrict__ wq;     // this lane's weight stream
  const char* in;
};

// operand group g: tap g / (4 kCB), input block (g / 4) % kCB, quad g % 4
template <class TW, int kT, int G>
__device__ __forceinline__ float4 stream_operand(const StreamState<TW, kT>& st, int t) {
  return *reinterpret_cast<const float4*>(st.in + st.rowo[t][G / (4 * TW::kCB)] +
                                          128 * ((G / 4) % TW::kCB) + 16 * (G % 4));
}

// kRes: groups 0 .. kRes - 1 come from `hw` (Resident::hw), the stream starts at group kRes
// (its first kStreamAhead groups requested by hidden_layer_stream before group 0 is issued)
template <class TW, int kT, int kRes, int G>
__device__ __forceinline__ void stream_group(StreamState<TW, kT>& st, const float4* hw) {
  constexpr int kCB = TW::kCB, kG = TW::kHidGroups;
  if constexpr (G + kStreamAhead < kG && G + kStreamAhead >= kRes + kStreamAhead) {
#pragma unroll
    for (int h = 0; h < kCB; ++h)
      st.wbuf[(G + kStreamAhead) % (kStreamAhead + 1)][h] = st.wq[((G + kStreamAhead) * kCB + h) * 64];
  }
  if constexpr (G + 1 < kG) {
#pragma unroll
    for (int t = 0; t < kT; ++t) st.bbuf[(G + 1) & 1][t] = stream_operand<TW, kT, G + 1>(st, t);
  }
  const float4* wg = G < kRes ? hw + G * kCB : st.wbuf[G % (kStreamAhead + 1)];
  const float4* bg = st.bbuf[G & 1];
#pragma unroll
  for (int h = 0; h < kCB; ++h)
#pragma unroll
    for (int t = 0; t < kT; ++t) st.acc[h][t] = DDD_MFMA32(wg[h].x, bg[t].x, st.acc[h][t]);
#pragma unroll
  for (int h = 0; h < kCB; ++h)
#pragma unroll
    for (int t = 0; t < kT; ++t) st.acc[h][t] = DDD_MFMA32(wg[h].y, bg[t].y, st.acc[h][t]);
#pragma unroll
  for (int h = 0; h < kCB; ++h)
#pragma unroll
    for (int t = 0; t < kT; ++t) st.acc[h][t] = DDD_MFMA32(wg[h].z, bg[t].z, st.acc[h][t]);
#pragma unroll
  for (int h = 0; h < kCB; ++h)
#pragma unroll
    for (int t = 0; t < kT; ++t) st.acc[h][t] = DDD_MFMA32(wg[h].w, bg[t].w, st.acc[h][t]);
  // schedule: the requests of later groups first, then this group's MFMAs
  if constexpr (G + kStreamAhead < kG && G >= kRes)
    __builtin_amdgcn_sched_group_barrier(0x020, kCB, 0);   // VMEM reads
  if constexpr (G + 1 < kG) __builtin_amdgcn_sched_group_barrier(0x100, kT, 0);              // DS reads
  __builtin_amdgcn_sched_group_barrier(0x008, 4 * kCB * kT, 0);                              // MFMAs
}

template <class TW, int kT, int kRes, int... G>
__device__ __forceinline__ void stream_groups(StreamState<TW, kT>& st, const float4* hw,
                                              std::integer_sequence<int, G...>) {
  (stream_group<TW, kT, kRes, G>(st, hw), ...);
}

// kRes > 0: hidden layer 0 with its leading groups and bias rows resident (hw, hwb)
template <class TW, int kWR, int kRes = 0>
__device__ __forceinline__ void hidden_layer_stream(const DevParams& p, const Lane& ln,
                                                    int hidden_index, const float* in,
                                                    float* out,   // (may be `in`: see Shared::kSingleBuffer)
                                                    const int (&rows)[2][TW::kK], int act,
                                                    const float4* hw = nullptr,
                                                    const float* hwb = nullptr) {
  constexpr int kT = kWR / 32, kCB = TW::kCB, kG = TW::kHidGroups;
  const int j = ln.lane & 31, half = ln.lane >> 5;
  StreamState<TW, kT> st;
#pragma unroll
  for (int t = 0; t < kT; ++t)
#pragma unroll
    for (int k = 0; k < TW::kK; ++k)
      st.rowo[t][k] = (int)__umul24((unsigned)rows[t][k], (unsigned)(TW::kHS * 4)) + 64 * half;
  st.in = reinterpret_cast<const char*>(in);
  const float* __restrict__ layer =
      p.w_hidden + (size_t)hidden_index * stream_layer_floats<TW>();
  st.wq = reinterpret_cast<const float4*>(layer) + opaque(ln.lane);
  const float* __restrict__ wbias = layer + kG * kCB * 64 * 4 + opaque(ln.lane);
#pragma unroll
  for (int h = 0; h < kCB; ++h)
#pragma unroll
    for (int t = 0; t < kT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) st.acc[h][t][r] = 0.0f;
  float wb[kCB];
#pragma unroll
  for (int h = 0; h < kCB; ++h) wb[h] = kRes > 0 ? hwb[h] : wbias[h * 64];
#pragma unroll
  for (int g = kRes; g < kRes + kStreamAhead; ++g)
    if (g < kG) {
#pragma unroll
      for (int h = 0; h < kCB; ++h) st.wbuf[g % (kStreamAhead + 1)][h] = st.wq[(g * kCB + h) * 64];
    }
#pragma unroll
  for (int t = 0; t < kT; ++t) st.bbuf[0][t] = stream_operand<TW, kT, 0>(st, t);
  // The requests above stay above: without this fence they share the scheduling region of
  // the groups below, whose "kCB VMEM reads, then the MFMAs" slots are filled in program
  // order -- the FIRST slot took the first request of the prologue, every later one the
  // request meant for kStreamAhead groups earlier, and each group's weights arrived right
  // before its own MFMAs: `global_load_dwordx4 ; s_waitcnt vmcnt(0) ; v_mfma` in every group
  // of every streamed layer of rounds 3-4 (found in round 5's ISA census).
  __builtin_amdgcn_sched_barrier(0);
  stream_groups<TW, kT, kRes>(st, hw, std::make_integer_sequence<int, kG>{});
#pragma unroll
  for (int h = 0; h < kCB; ++h)
#pragma unroll
    for (int t = 0; t < kT; ++t) st.acc[h][t] = DDD_MFMA32(wb[h], 1.0f, st.acc[h][t]);   // bias row
#pragma unroll
  for (int h = 0; h < kCB; ++h)
#pragma unroll
    for (int t = 0; t < kT; ++t) {
      activate16(st.acc[h][t], act);
      store_tile32_at<TW::kHS>(out, ln.wave * kWR + t * 32 + j, h, half, st.acc[h][t]);
    }
}

// The same layer as a loop over the taps (Tower::kRolled).  One tap = 4 kCB operand groups;
// the weight ring is four groups deep (three requested ahead) so that its slot indices repeat
// with the tap, operands are double-buffered as above.  The operand rows of a tap are
// recomputed inside the loop (a run-time index into a register array would be a scratch
// array); the last tap's read-ahead re-reads its own first groups instead of running past
// the layer.
template <class TW>
__device__ __forceinline__ int tap_row_one(const Lane& ln, int trow, int n, bool pow2, int k) {
  constexpr int kLeft = TW::kK / 2;
  if (pow2) {
    const int mask = n - 1;
    return ((trow + k - kLeft) & mask) | (trow & ~mask);
  }
  const int base = row_sample(trow, ln.inv_n) * n;
  int q = trow - base + k - kLeft;
  q = q < 0 ? q + n : q;
  q = q >= n ? q - n : q;
  return trow < ln.rows_used ? base + q : trow;
}

template <class TW, int kWR>
__device__ __forceinline__ void hidden_layer_rolled(const DevParams& p, const Lane& ln,
                                                    int hidden_index, const float* in,
                                                    float* out, int n, bool pow2, int act) {
  constexpr int kT = kWR / 32, kCB = TW::kCB, kGT = 4 * kCB, kRing = 4, kAhead = kRing - 1;
  static_assert(kGT % kRing == 0, "ring slots repeat with the tap");
  const int j = ln.lane & 31, half = ln.lane >> 5;
  const float* __restrict__ layer = p.w_hidden + (size_t)hidden_index * stream_layer_floats<TW>();
  const float4* __restrict__ wq = reinterpret_cast<const float4*>(layer) + opaque(ln.lane);
  const float* __restrict__ wbias = layer + TW::kHidGroups * kCB * 64 * 4 + opaque(ln.lane);
  const char* inb = reinterpret_cast<const char*>(in);
  f32x16 acc[kCB][kT];
#pragma unroll
  for (int h = 0; h < kCB; ++h)
#pragma unroll
    for (int t = 0; t < kT; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[h][t][r] = 0.0f;
  float wb[kCB];
#pragma unroll
  for (int h = 0; h < kCB; ++h) wb[h] = wbias[h * 64];
  float4 wbuf[kRing][kCB], bbuf[2][kT];
#pragma unroll
  for (int g = 0; g < kAhead; ++g)
#pragma unroll
    for (int h = 0; h < kCB; ++h) wbuf[g][h] = wq[(g * kCB + h) * 64];
  int trow[kT], ro[kT];
#pragma unroll
  for (int t = 0; t < kT; ++t) {
    trow[t] = ln.wave * kWR + t * 32 + j;
    ro[t] = (int)__umul24((unsigned)tap_row_one<TW>(ln, trow[t], n, pow2, 0), (unsigned)(TW::kHS * 4)) + 64 * half;
    bbuf[0][t] = *reinterpret_cast<const float4*>(inb + ro[t]);
  }
#pragma unroll 1
  for (int tap = 0; tap < TW::kK; ++tap) {
    const bool more = tap + 1 < TW::kK;   // wave-uniform
    int ro_next[kT];
#pragma unroll
    for (int t = 0; t < kT; ++t)
      ro_next[t] = (int)__umul24((unsigned)tap_row_one<TW>(ln, trow[t], n, pow2, more ? tap + 1 : tap),
                                 (unsigned)(TW::kHS * 4)) + 64 * half;
    const float4* __restrict__ wt = wq + (size_t)tap * (kGT * kCB * 64);
    const float4* __restrict__ wt_ahead = more ? wt : wt - kGT * kCB * 64;   // (groups past the tap)
#pragma unroll
    for (int gi = 0; gi < kGT; ++gi) {
#pragma unroll
      for (int h = 0; h < kCB; ++h)
        wbuf[(gi + kAhead) % kRing][h] =
            (gi + kAhead < kGT ? wt : wt_ahead)[((gi + kAhead) * kCB + h) * 64];
#pragma unroll
      for (int t = 0; t < kT; ++t)
        bbuf[(gi + 1) & 1][t] = *reinterpret_cast<const float4*>(
            inb + (gi + 1 < kGT ? ro[t] + 128 * (((gi + 1) / 4) % kCB) + 16 * ((gi + 1) % 4) : ro_next[t]));
      const float4* wg = wbuf[gi % kRing];
      const float4* bg = bbuf[gi & 1];
#pragma unroll
      for (int h = 0; h < kCB; ++h)
#pragma unroll
        for (int t = 0; t < kT; ++t) acc[h][t] = DDD_MFMA32(wg[h].x, bg[t].x, acc[h][t]);
#pragma unroll
      for (int h = 0; h < kCB; ++h)
#pragma unroll
        for (int t = 0; t < kT; ++t) acc[h][t] = DDD_MFMA32(wg[h].y, bg[t].y, acc[h][t]);
#pragma unroll
      for (int h = 0; h < kCB; ++h)
#pragma unroll
        for (int t = 0; t < kT; ++t) acc[h][t] = DDD_MFMA32(wg[h].z, bg[t].z, acc[h][t]);
#pragma unroll
      for (int h = 0; h < kCB; ++h)
#pragma unroll
        for (int t = 0; t < kT; ++t) acc[h][t] = DDD_MFMA32(wg[h].w, bg[t].w, acc[h][t]);
      __builtin_amdgcn_sched_group_barrier(0x020, kCB, 0);           // VMEM reads
      __builtin_amdgcn_sched_group_barrier(0x100, kT, 0);            // DS reads
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * kCB * kT, 0);  // MFMAs
    }
#pragma unroll
    for (int t = 0; t < kT; ++t) ro[t] = ro_next[t];
  }
#pragma unroll
  for (int h = 0; h < kCB; ++h)
#pragma unroll
    for (int t = 0; t < kT; ++t) acc[h][t] = DDD_MFMA32(wb[h], 1.0f, acc[h][t]);   // bias row
#pragma unroll
  for (int h = 0; h < kCB; ++h)
#pragma unroll
    for (int t = 0; t < kT; ++t) {
      activate16(acc[h][t], act);
      store_tile32_at<TW::kHS>(out, ln.wave * kWR + t * 32 + j, h, half, acc[h][t]);
    }
}

// Output layer (32 -> C_out <= 16, linear) on v_mfma_f32_4x4x1_16b_f32.  A
// 16x16x4 formulation pads the 11-14 live output channels to 16 and leaves the
// result in a (position, channel-quad) layout that has to travel through LDS to
// reach the lane == row epilogue (round 1: 5248 pipe cycles + a store, a
// barrier and a read-back).  The 16-block 4x4x1 form with the A block broadcast
// (cbsz = 4, abid = b: lanes 4 b .. 4 b + 3 of the A register feed ALL blocks)
// computes, per instruction, four output channels x one reduction step for 64
// positions with B = one value per lane:
//   D[r] of lane l += W[k][4 grp + r] * h[k][row of lane l]
// so (a) only ceil(C / 4) channel groups are issued (C = 12: 3 groups, 483
// instructions x 8 cycles = 3864 pipe cycles instead of 5248), (b) one weight
// register holds 16 (k, group) slots -- 31 registers for 483 instructions --
// and (c) every lane ends up with its OWN row's channels: no LDS round trip.
//   instruction q = (((tap * 8 + c4) * 4 + e) * NG + grp), k = tap * 32 + 4 c4 + e,
//   weight register q / 16, abid q % 16 (capi.hip: pack_final4); q = 160 NG + grp:
//   bias row against B = 1.
// B operands: the lane's own five tap rows, 8 x ds_read_b128 each (row stride
// 144 B keeps 16 consecutive rows on distinct 16-byte bank slots), prefetched
// two operand groups (24-32 MFMAs) ahead.
template <int kAbid>
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 4, kAbid, 0);
}

// (kAcc accumulator chains: NG, one per channel group -- or TWO for a single group, whose lone
// dependent chain runs at 13.2 cycles per MFMA instead of 8.5: the direct heads, model.py:551-615,
// sum the two halves of the reduction at the end)
template <int NG, int Q0, int NW, int kAcc, int... J>
__device__ __forceinline__ void fin4_mfmas(const float (&w)[NW], const f32x4& b,
                                           f32x4 (&acc)[kAcc], std::integer_sequence<int, J...>) {
  ((acc[J % kAcc] = mfma4<(Q0 + J) % 16>(w[(Q0 + J) / 16], b[J / NG], acc[J % kAcc])), ...);
}

template <int NG>
__device__ __forceinline__ void load_final4(const DevParams& p, int lane,
                                            float (&w)[fin4_regs(NG)]) {
  load_rows4<fin4_regs(NG)>(p.w_final4, lane, w);
}

#ifndef DDD_FIN4_AHEAD
#define DDD_FIN4_AHEAD 2   // A/B (profiles/r6_ablation.txt)
#endif
constexpr int kFin4Ahead = DDD_FIN4_AHEAD;   // operand groups in flight ahead of the MFMAs

// (TW: the tower -- K taps x C channels: K C / 4 operand groups of four channels,
// C / 4 per tap row)
template <int NG, int OG, class TW, int kAcc>
__device__ __forceinline__ void fin4_step(const char* __restrict__ in, const int (&off)[TW::kK],
                                          const float (&w)[fin4_regs_t<TW>(NG)],
                                          f32x4 (&buf)[kFin4Ahead + 1], f32x4 (&acc)[kAcc]) {
  constexpr int kNext = OG + kFin4Ahead;
  constexpr int kPerTap = TW::kFinC / 4;
  if constexpr (kNext < TW::kOperandGroups)
    buf[kNext % (kFin4Ahead + 1)] =
        *reinterpret_cast<const f32x4*>(in + off[kNext / kPerTap] + 16 * (kNext % kPerTap));
  fin4_mfmas<NG, OG * 4 * NG, fin4_regs_t<TW>(NG), kAcc>(w, buf[OG % (kFin4Ahead + 1)], acc,
                                                         std::make_integer_sequence<int, 4 * NG>{});
  if constexpr (kNext < TW::kOperandGroups) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // DS read
  __builtin_amdgcn_sched_group_barrier(0x008, 4 * NG, 0);                        // MFMAs
}

template <int NG, class TW, int kAcc, int... OG>
__device__ __forceinline__ void fin4_run(const char* __restrict__ in, const int (&off)[TW::kK],
                                         const float (&w)[fin4_regs_t<TW>(NG)],
                                         f32x4 (&buf)[kFin4Ahead + 1], f32x4 (&acc)[kAcc],
                                         std::integer_sequence<int, OG...>) {
  (fin4_step<NG, OG, TW, kAcc>(in, off, w, buf, acc), ...);
}

// `off`: LDS byte offsets (row * kHS * 4) of the lane's tap rows.
template <int NG, class TW = DefaultTower, int kAcc = NG, int kPrio = 0>
__device__ __forceinline__ void final_layer4(const float* __restrict__ in_f,
                                             const float (&w)[fin4_regs_t<TW>(NG)],
                                             const int (&off)[TW::kK], f32x4 (&acc)[kAcc]) {
  const char* __restrict__ in = reinterpret_cast<const char*>(in_f);
  constexpr int kPerTap = TW::kFinC / 4;
  f32x4 buf[kFin4Ahead + 1];
#pragma unroll
  for (int g = 0; g < kAcc; ++g) acc[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int og = 0; og < kFin4Ahead; ++og)
    buf[og] = *reinterpret_cast<const f32x4*>(in + off[og / kPerTap] + 16 * (og % kPerTap));
  __builtin_amdgcn_sched_group_barrier(0x100, kFin4Ahead, 0);
  if (kPrio >= 3) __builtin_amdgcn_s_setprio(kPrio == 5 ? 1 : 0);   // (5, A/B: the 8-cycle MFMA stream one level up)
  fin4_run<NG, TW, kAcc>(in, off, w, buf, acc, std::make_integer_sequence<int, TW::kOperandGroups>{});
  if (kPrio >= 3) __builtin_amdgcn_s_setprio(3);
  // bias row: k = K C against a constant 1
  fin4_mfmas<NG, (TW::kFinK - 1) * NG, fin4_regs_t<TW>(NG), kAcc>(
      w, f32x4{1.0f, 1.0f, 1.0f, 1.0f}, acc, std::make_integer_sequence<int, NG>{});
}

// The output layer of Tile16Tower: lane == row as final_layer4, the row's 16 channels stored at
// float 4 (c & 3) + (c >> 2) -- four ds_read_b128 per tap row, picked in natural channel order.
// Weights: the 5 x 16 + 1 reduction rows of HalfTower's packing.
template <int NG, int kPrio = 0>
__device__ __forceinline__ void final_layer4_t16(const float* __restrict__ in_f,
                                                 const float (&w)[fin4_regs_t<Tile16Tower>(NG)],
                                                 const int (&off)[kKW], f32x4 (&acc)[NG]) {
  const char* __restrict__ in = reinterpret_cast<const char*>(in_f);
  f32x4 cur[4], nxt[4];
#pragma unroll
  for (int g = 0; g < NG; ++g) acc[g] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int q = 0; q < 4; ++q) cur[q] = *reinterpret_cast<const f32x4*>(in + off[0] + 16 * q);
  __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
  if (kPrio >= 3) __builtin_amdgcn_s_setprio(kPrio == 5 ? 1 : 0);
  constexpr int kNW = fin4_regs_t<Tile16Tower>(NG);
  const auto tap_steps = [&](auto tap_c) {
    constexpr int tap = decltype(tap_c)::value;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      nxt[q] = cur[q];
      if (tap + 1 < kKW) nxt[q] = *reinterpret_cast<const f32x4*>(in + off[tap + 1 < kKW ? tap + 1 : tap] + 16 * q);
    }
    // channels 4 m .. 4 m + 3 = element m of the four blocks
    fin4_mfmas<NG, (4 * tap + 0) * 4 * NG, kNW, NG>(w, f32x4{cur[0][0], cur[1][0], cur[2][0], cur[3][0]}, acc,
                                                 std::make_integer_sequence<int, 4 * NG>{});
    fin4_mfmas<NG, (4 * tap + 1) * 4 * NG, kNW, NG>(w, f32x4{cur[0][1], cur[1][1], cur[2][1], cur[3][1]}, acc,
                                                 std::make_integer_sequence<int, 4 * NG>{});
    fin4_mfmas<NG, (4 * tap + 2) * 4 * NG, kNW, NG>(w, f32x4{cur[0][2], cur[1][2], cur[2][2], cur[3][2]}, acc,
                                                 std::make_integer_sequence<int, 4 * NG>{});
    fin4_mfmas<NG, (4 * tap + 3) * 4 * NG, kNW, NG>(w, f32x4{cur[0][3], cur[1][3], cur[2][3], cur[3][3]}, acc,
                                                 std::make_integer_sequence<int, 4 * NG>{});
#pragma unroll
    for (int q = 0; q < 4; ++q) cur[q] = nxt[q];
    if (tap + 1 < kKW) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
    __builtin_amdgcn_sched_group_barrier(0x008, 16 * NG, 0);
  };
  tap_steps(std::integral_constant<int, 0>{});
  tap_steps(std::integral_constant<int, 1>{});
  tap_steps(std::integral_constant<int, 2>{});
  tap_steps(std::integral_constant<int, 3>{});
  tap_steps(std::integral_constant<int, 4>{});
  if (kPrio >= 3) __builtin_amdgcn_s_setprio(3);
  // bias row: k = 80 against a constant 1
  fin4_mfmas<NG, (Tile16Tower::kFinK - 1) * NG, kNW, NG>(
      w, f32x4{1.0f, 1.0f, 1.0f, 1.0f}, acc, std::make_integer_sequence<int, NG>{});
}

// Wavefronts per SIMD the kernels are compiled for: two, except the four-wave groups of the
// 64-channel towers (158 KB of LDS: one workgroup per CU anyway).
#ifndef DDD_MIN_WAVES_AB
#define DDD_MIN_WAVES_AB 2   // A/B (profiles/r4_ablation.txt): 1 = compile every kernel for ONE wavefront per SIMD
#endif
template <int kRows, int kWR, class TW, bool kAdaptive = false>
constexpr int min_waves() {
  return (TW::kCB == 2 && kRows != kWR) ? 1
         : (TW::kK == 3 && TW::kCB == 1 && kRows == kWR && !kAdaptive) ? DDD_K3_WAVES
                                                                      : DDD_MIN_WAVES_AB;
}

// Kernel-lifetime registers of one lane: hoisted once per launch.
struct Resident {
  float w_fin4[fin4_regs(4)];   // output layer weights (specialised one-wave integrators)
  int fin4_off[kKW];        // LDS byte offsets of this lane's five tap rows (same kernels)
  int hid_off[2][kKW];      // LDS byte offsets of the hidden layer's operand rows (same kernels)
  int in_perm[2][3];        // ds_bpermute addresses of the input layer's operands (same kernels)
  float4 trig[kTrigMax / 4];   // this grid point's cos / sin of the spatial phases (same kernels)
  int pch_idx[kGMax];       // indices into Shared::u of this row's stencil patch (same kernels)
  int st_off;               // LDS byte offset of this lane's tile-0 activation row (+ 16 half; same kernels)
  int fk_off;               // byte offset of this row's sample in Shared::fk (same kernels)
  float w_in[kInSteps];     // input-layer weights (MFMA A operand)
  float hid[kHidSteps];     // the hidden layer's weights when there is exactly one
  float4 hw[kResidentQuads];   // streamed towers: leading weight groups [group][block] of hidden layer 0
  float hwb[2];                // ... and its bias rows (resident_groups)
  bool hw_valid = false;       // (a compile-time constant after inlining: set by setup_weights)
  // four wavefronts per group (kWR = 16): A operands and LDS byte offsets of the 16x16x4 layers
  float q_in[kQuadInSteps], q_hid[kQuadHidSteps], q_fin[kQuadFinSteps];
  int q_in_off[2][2], q_hid_off[2][kKW], q_fin_off[kKW], q_st_off[2];
  int q_xch_off;            // byte offset (row, channel 4 (l >> 4)) of this lane's output-layer results
  // Tile16Tower: LDS byte offsets / ds_bpermute addresses of the four 16-position tiles
  int t16_hid_off[4][kKW], t16_in_perm[4][2], t16_st_off[4];
  float frc_a, frc_omega, frc_phi;   // this lane's (sample, mode) forcing parameters
  float frc_mask[8];        // 1 where entry i of this lane's first 8-mode trip belongs to its run
  float fk_next;            // this lane's harmonic sum for the NEXT evaluation's time
  int frc_run;              // its run of modes: (float offset into Shared::pm) | count << 16
                            // | (index of the sum in Shared::fk) << 24; 0: lane carries none
  int frc_pairs;            // lanes of forcing phase 1: (samples | batches) x P (wave-uniform)
  int frc_slot;             // index of this lane's sum in Shared::fk (fixed by the lane, not by
                            // the sample), -1: the lane carries no (sample, k, sin|cos) slot
#ifdef DDD_PROBES
  unsigned long long* probe_stamp = nullptr;   // setup_weights: "index math done" (substep walk trace)
#endif
};

// Forcing, phases 1 + 2, for time t:
//   sum_j a_j sin(omega_j t + theta_j(x) + phi_j)
//     = sum_j [a_j sin(psi_j)] cos(theta_j(x)) + [a_j cos(psi_j)] sin(theta_j(x)),
//   psi_j = omega_j t + phi_j,  theta_j(x) = 2 pi k_j x / L  (<= 6 distinct k).
// Phase 1: one (sample, mode) pair per lane -> Shared::pm.  A barrier.  Phase
// 2: lanes carrying a (sample, k, sin|cos) slot sum the modes with that k in
// mode order (modes are stored sorted by k, ddd_set_forcing, so the run is
// contiguous; Resident::frc_run).  The caller publishes the sum to Shared::fk
// at the start of the evaluation that uses it.  Inside an evaluation the two
// phases sit at the input->hidden and hidden->output layer boundaries, where
// the wavefront otherwise only waits for its activations to land in LDS.
// (t: per lane -- forcing batches give every batch of lanes its own evaluation time;
// res.frc_pairs: (samples or batches of the group) x P)
template <int kRows, int kWR, bool kWide, class TW>
__device__ __forceinline__ void forcing_phase1(const DevParams& p, Shared<kRows, kWR, kWide, TW>& sm,
                                               const Resident& res, float t, int tid) {
  if (tid < res.frc_pairs) {
    float sn, cs;
    sincos_branchless(res.frc_omega * t + res.frc_phi, &sn, &cs);
    sm.pm[tid] = make_float2(res.frc_a * sn, res.frc_a * cs);
  }
}

// kMasked (kernels that keep Resident::frc_mask): the first trip multiplies by the
// lane's 0 / 1 masks instead of compare + select per entry.  fma(v, 1, acc) is
// acc + v and fma(v, 0, acc) is acc for every finite v (the staged values are
// a sin / a cos of finite angles, the padding is zeroed at setup), so both
// forms give the same bits.
template <int kRows, int kWR, bool kMasked = false, bool kWide = false, class TW = DefaultTower>
__device__ __forceinline__ float forcing_phase2(Shared<kRows, kWR, kWide, TW>& sm, const Resident& res) {
  const int cnt = (res.frc_run >> 16) & 0xff;
  const float* __restrict__ pm = reinterpret_cast<const float*>(sm.pm) + (res.frc_run & 0xffff);
  float acc = 0.0f;
  // eight independent LDS reads per trip (runs average P / n_k = 7 modes);
  // entries past the run add an exact 0, so the sum keeps mode order
  // (reads past the run stay inside Shared::pm: it carries 8 entries of padding)
  int first = 0;
  if (kMasked) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = pm[2 * i];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = fmaf(v[i], res.frc_mask[i], acc);
    first = 8;
  }
  for (int m = first; m < cnt; m += 8) {
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = pm[2 * (m + i)];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = acc + (m + i < cnt ? v[i] : 0.0f);
  }
  return acc;
}

// (kMasked: forcing_phase2's masked first trip -- the same bits, a third of the
// instructions; needs Resident::frc_mask, i.e. apply_samples.)
template <int kRows, int kWR, bool kMasked = false, bool kWide = false, class TW = DefaultTower>
__device__ __forceinline__ float forcing_sums(const DevParams& p, Shared<kRows, kWR, kWide, TW>& sm,
                                              const Resident& res, float t, int tid) {
  forcing_phase1<kRows, kWR>(p, sm, res, t, tid);
  group_barrier<kRows, kWR>();
  return forcing_phase2<kRows, kWR, kMasked>(sm, res);
}

// One evaluation of finalize_time_derivative(t, predict_time_derivative(u))
// for the workgroup's rows.  Must be called by all kRows threads.
//   model.predict_coefficients  model.py:420-513   (conv tower + projection)
//   model.apply_coefficients    model.py:536-548   (stencil apply)
//   Equation.equation_of_motion equations.py       (dev_params.h)
//   finalize_time_derivative    equations.py:276-277 (forcing)
// kHoist: res.hid already holds the (single) hidden layer's weights.
// prepare_next: also compute the harmonic forcing sums of time t_next (the
// evaluation after this one) at the layer boundaries.
// ablate / trace: profiling hooks of libddd1d_probe.so (-DDDD_PROBES); every
// product call site passes the defaults, so they fold away.
// kLean (adaptive integrators): the output layer's weights and the cos / sin table
// are NOT kept resident (fetched from L2 / the LDS row padding per evaluation like
// the run-time kernels do): 31-43 VGPRs the controller state needs.
// (kLean = 2: only the cos / sin table leaves the registers -- 12 VGPRs for two ds_read_b128
// per evaluation; the round-5 adaptive kernels)
template <int kRows, int kWR, bool kHoist, int kEq, bool kTrace, bool kWide, int kLean = 0,
          class TW = DefaultTower>
__device__ __forceinline__ float eval_rhs(const DevParams& p, Shared<kRows, kWR, kWide, TW>& sm, int batch,
                                          float u, float t, float t_next, Resident& res,
                                          bool fast_forcing, float* derivs_out,
                                          float* coeffs_out, bool prepare_next = true,
                                          int group = -1, int ablate = 0,
                                          unsigned long long* trace = nullptr) {
#define DDD_STAMP(i) do { if (kTrace && trace != nullptr && group_tid<kRows, kWR>() == 0) trace[i] = __builtin_amdgcn_s_memtime(); } while (0)
  DDD_STAMP(0);
  // run-time parameters, or compile-time constants when specialised (kEq >= 0)
  constexpr bool kSpec = kEq >= 0;
  const int eqn = kSpec ? kEq : p.equation;
  const int nD = kSpec ? spec_derivs(kEq) : opaque_scalar(p.D);
  const int nG = kSpec ? spec_stencil(kEq) : opaque_scalar(p.G);
  const unsigned dsel_valid = kSpec ? 0u : (unsigned)opaque_scalar((int)p.dsel_valid);
  const unsigned long long dsel_bits = kSpec ? 0ull : opaque_scalar(p.dsel_bits);
  const int rt_groups = kSpec ? 0 : opaque_scalar(p.rt_groups);
  const bool flux_form = kSpec ? spec_flux_form(kEq) : (p.conservative != 0);
  const bool fixed = kSpec ? false : (p.fixed != 0 || p.linear_taps != 0);   // no conv tower to run
  const bool folded = kSpec ? spec_folded(kSpec ? kEq : 0) : (p.folded != 0);
  // what the tower predicts (model.py:579-640): stencil coefficients (default),
  // the spatial derivatives, the time derivative or the flux themselves
  const int target = kSpec ? (int)TARGET_COEFFICIENTS : p.target;
  const int act = kSpec ? (int)ACT_RELU : p.act;
  // specialised kernels: forcing only in the Burgers family, and only in its
  // harmonic-sum form (capi.hip routes anything else to the run-time kernels)
  const bool forced = kSpec ? (spec_forced_family(kSpec ? kEq : 0) && p.forced != 0)
                            : (p.forced != 0);
  if (kSpec) fast_forcing = true;
  const int nL = kHoist ? 3 : p.L;
  const bool pow2 = kRows == 64 || (p.N & (p.N - 1)) == 0;   // N | 64: always; else wave-uniform
  constexpr bool kOneWave = kRows == kWR;   // no other wavefront touches this group's LDS
  constexpr int kPrio = prio_mode(kEq);
#ifndef DDD_QUAD_PRIO
#define DDD_QUAD_PRIO 0   // A/B: 3 = the four-wavefront groups' hidden layer lowers its steady middle
#endif
  constexpr int kQuadPrio = DDD_QUAD_PRIO;
  // output-channel groups of four (final_layer4): the specialised kernels issue
  // their live ones as one compile-time interleaved stream (channels renumbered
  // contiguously, DevParams::w_final4); the run-time-parameterised kernels
  // issue DevParams::rt_groups live groups two by two (w_final4_rt)
  constexpr int kNG = kSpec ? spec_fin_groups(kSpec ? kEq : 0) : 2;
  constexpr int kGW = flavour_stencil(kWide);    // stencil columns carried
  constexpr int kCh = flavour_net_channels(kWide);   // output channels carried
  static_assert(!(kSpec && kWide), "the per-equation kernels have no wide flavour");
  static_assert(TW::kDefault || (!kSpec && !kWide && !kHoist && kWR == 64),
                "towers other than 5 taps x 32 channels: run-time-parameterised kernels only");
  // 64-row-wavefront integrators with one hidden layer keep the loop-invariant
  // LDS offsets / permute addresses / patch indices in registers (kKeepOffsets);
  // the per-equation ones also the output layer's weights and the grid point's
  // cos / sin table (kKeepRows)
  // kSplit (per-equation integrators, <64, 32>): ONE sample on TWO 32-row wavefronts
  // for ensembles that leave SIMDs with a single 64-row wavefront -- each wavefront
  // carries one 32-row tile through the input and hidden layers (half the MFMAs
  // each); the output layer, whose 4x4x1 form covers 64 rows per instruction, is
  // split by CHANNEL GROUPS instead of duplicated: wavefront 0 issues groups
  // [0, ceil(NG / 2)) for all 64 rows, wavefront 1 the rest, and the channels meet
  // in the free activation buffer.  Every accumulation chain keeps its order: the
  // bits equal the one-wavefront kernel's.
  constexpr bool kSplit = kSpec && kRows == 64 && kWR == 32 && kHoist;
  // kQuad (per-equation integrators, <64, 16>): the same sample on FOUR wavefronts, every
  // layer on v_mfma_f32_16x16x4_f32 (input_layer_quad .. final_layer_quad above)
  constexpr bool kQuad = kSpec && kRows == 64 && kWR == 16 && kHoist;
  static_assert(kWR != 16 || kQuad, "16-row wavefronts: per-equation integrators only");
  constexpr bool kKeepOffsets = (kWR == 64 || kSplit || kQuad) && kHoist;
  constexpr bool kKeepRows = kKeepOffsets && kEq >= 0 && kLean != 1;
  constexpr bool kKeepPatch = kKeepOffsets && !kWide;   // Resident::pch_idx holds 8 columns
  const int tid = opaque(group_tid<kRows, kWR>());
  const Lane ln = make_lane<kRows, kWR>(p, batch, tid, group < 0 ? (int)blockIdx.x : group);
  if (ln.owner) sm.u[ln.row] = u;
  // ---- NaN through relu.  np.maximum / Eigen's relu pass a NaN on, so in the reference
  // a NaN at grid point i makes the net's whole receptive field NaN: coefficients, hence
  // derivatives, at x with i in [x - L (K / 2), x + L (K - 1 - K / 2)].  Here relu is a clamp
  // (v_pk_add clamp / v_max / fmin(fmax)) that maps NaN to 0, and only the stencil's own
  // reach would see the NaN.  The reference's mask is restored explicitly: ONE v_cmp per
  // evaluation finds whether any state of the group is NaN (a ballot; multi-wave groups
  // share a sticky LDS flag), and only then -- a diverged sample -- every lane looks for a
  // NaN in its receptive field and replaces the net's outputs by NaN, from where it flows
  // through projection, stencil apply, equation of motion and flux difference exactly as
  // in the reference (tests/test_gpu_rhs.py::test_nan_mask_equals_the_oracles).
  const bool clamps_nan = !fixed && (act == ACT_RELU || act == ACT_RELU6);   // wave-uniform
  if (clamps_nan && !kOneWave) {
    if (__builtin_amdgcn_ballot_w64(u != u) != 0 && ln.lane == 0) sm.nan_flag = 1;
  }
  const auto nan_in_receptive_field = [&]() -> bool {   // (the rare path)
    const int n = p.N, lk = p.K >> 1;                  // layers.pad_periodic(center): K / 2 on the left
    int span = p.L * (p.K - 1);
    span = span < n - 1 ? span : n - 1;
    int q = (ln.pos - p.L * lk) % n;
    q = q < 0 ? q + n : q;
    bool hit = false;
    for (int i = 0; i <= span; ++i) {
      const float v = sm.u[ln.base + q];
      hit = hit || (v != v);
      q = q + 1 == n ? 0 : q + 1;
    }
    return hit;
  };
  // conv-tap source rows of this wave's two 32-row tiles (input + hidden
  // layers): index math placed here, in the shadow of the LDS round trip below
  int hid_rows[2][TW::kK];
  if (!fixed && !kQuad) {
    if constexpr (!TW::kDefault) {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
        tap_rows_n<TW>(ln, ln.wave * kWR + t2 * 32 + (ln.lane & 31), p.N, pow2, hid_rows[t2]);
    } else if constexpr (kKeepOffsets) {
#pragma unroll
      for (int t2 = 0; t2 < 2; ++t2)
#pragma unroll
        for (int k = 0; k < kKW; ++k) hid_rows[t2][k] = res.hid_off[t2][k];
    } else {
#pragma unroll
      for (int t2 = 0; t2 < kWR / 32; ++t2)
        tap_rows<kRows == 64>(ln, ln.wave * kWR + t2 * 32 + (ln.lane & 31), p.N, hid_rows[t2]);
    }
  }
  // model.py:450-451: net = u / std.  Three FMA-class instructions instead of
  // the 12 of the IEEE division sequence: q = RN(u r), r = RN(1 / std), then one
  // Newton step on the residual, q' = fma(fma(-q, std, u), r, q).  Whether q' is
  // the correctly rounded quotient for EVERY float32 u depends on std: the host
  // checks all 2^23 significands of one binade (the identity is invariant under
  // scaling u by powers of two) for the model's own standard deviation when the
  // model is created (capi.hip: division_shortcut_is_exact) and sets
  // DevParams::exact_div when a single case differs -- then the true division
  // runs.  Outside the normal range: NaN propagates; a quotient that overflows
  // (a diverged state) becomes NaN instead of Inf, a subnormal quotient may
  // differ in its last bit.  Every VALU instruction here is matrix-pipe time.
  const float q_un = u * p.inv_stddev;
  float un_reg = fmaf(fmaf(-q_un, p.stddev, u), p.inv_stddev, q_un);
  if (p.exact_div) {   // wave-uniform, and a BRANCH: as a select, the 12-instruction IEEE
    asm volatile("; exact division");   // sequence ran in every evaluation (round 4: 14 of 182)
    un_reg = u / p.stddev;
  }
  // (a one-wave group feeds the input layer by lane permutes, not through LDS)
  if (!fixed && !kOneWave && ln.owner) sm.un[ln.row] = un_reg;
  // harmonic forcing sums of THIS evaluation's time: computed during the
  // previous evaluation (or the launch prologue), published here
  // (after the barrier: slower wavefronts may still be reading sm.fk in the
  // epilogue of the previous evaluation; the next barrier orders the readers)
  group_barrier<kRows, kWR>();
  const bool trig_lds = p.n_k <= 4;   // cos/sin table staged in the LDS row padding
  if (forced && fast_forcing && res.frc_slot >= 0)   // (an empty run publishes its 0)
    sm.fk[res.frc_slot] = res.fk_next;

  // patches[i] = u[(x + i - G/2) mod N]   (model.extract_patches, model.py:516-533)
  // Four-wave groups must read them now (other waves rewrite sm.u as soon as
  // they enter the next evaluation); a one-wave group reads them in the
  // epilogue instead and saves 8 registers across the conv tower.
  // multi-wave groups decide now: a faster wavefront rewrites sm.u as soon as it enters the
  // next evaluation (one-wave groups decide where the net's outputs are consumed: nothing
  // is carried across the tower)
  bool nan_any = false, poisoned = false;
  if (clamps_nan && !kOneWave) {
    nan_any = __builtin_amdgcn_readfirstlane(sm.nan_flag) != 0;
    if (nan_any) poisoned = nan_in_receptive_field();
  }
  float pch[kGW];
  const int gl = nG >> 1;
  if (!kOneWave) {
#pragma unroll
    for (int g = 0; g < kGW; ++g)
      pch[g] = (g < nG) ? sm.u[kKeepPatch ? res.pch_idx[g < kGMax ? g : 0]
                               : pow2 ? (((ln.pos + g - gl) & (p.N - 1)) | ln.base)
                                      : wrap_row(ln.base, ln.pos, g - gl, p.N)] : 0.0f;
  }

  float net[kCh];
#pragma unroll
  for (int c = 0; c < kCh; ++c) net[c] = 0.0f;
  if (!fixed) {
    DDD_STAMP(1);
    // A/B (DDD_PRIO_PHASES): the matrix phases of an evaluation at raised issue priority, the
    // VALU phases (epilogue, forcing, Runge-Kutta update) at the lowest
    // (DDD_PRIO_PHASES = 2, round 6: the other way round -- the VALU phases raised, so that a
    // wavefront's short bookkeeping is not stretched by its SIMD partner's MFMA stream)
    if (kPrio) __builtin_amdgcn_s_setprio(kPrio == 2 ? 0 : 3);   // (3: stays raised)
    // kernels that do not keep the hidden layer resident (four-wave adaptive integrators,
    // run-time kernels): the first hidden layer's 81 operand rows are REQUESTED here, before
    // the input layer, its store and the barrier -- one L2 round trip per evaluation that
    // used to start right in front of the layer's first MFMA
    if constexpr (TW::kDefault && !kQuad) {
      if (!kHoist && nL > 2) load_hidden(p, 0, ln.lane, res.hid);
    }
    if constexpr (kQuad) {
      input_layer_quad(sm.un, sm.hA, res.q_in, res.q_in_off, res.q_st_off, act, ln.lane);
    } else if constexpr (TW::kTile16) {
      static_assert(kKeepOffsets && kWR == 64 && kOneWave, "16-channel tiles: resident one-wave integrators");
      input_layer_t16(un_reg, sm.hA, res.w_in, res.t16_in_perm, res.t16_st_off, act, ln.lane);
    } else if constexpr (!TW::kDefault) {
      input_layer_big<TW, kWR, kOneWave>(p, ln, sm.un, un_reg, sm.hA, hid_rows, act);
    } else if (!(ablate & 16)) {
      input_layer<kWR, kOneWave, kKeepOffsets>(p, ln, sm.un, un_reg, sm.hA, res.w_in, hid_rows,
                                               act, res.in_perm, res.st_off);
    }
    const bool frc_next = forced && fast_forcing && prepare_next && !(ablate & 1);
    if (frc_next) {
      if (kPrio == 4) __builtin_amdgcn_s_setprio(0);
      forcing_phase1<kRows, kWR>(p, sm, res, t_next, tid);
      if (kPrio == 4) __builtin_amdgcn_s_setprio(3);
    }
    float* in = sm.hA;
    float* out = Shared<kRows, kWR, kWide, TW>::kSingleBuffer ? sm.hA : sm.hB;
    for (int l = 1; l < nL - 1; ++l) {
      if constexpr (kQuad) {
        group_barrier<kRows, kWR>();
        hidden_layer_quad<kQuadPrio>(in, out, res.q_hid, res.q_hid_off, res.q_st_off, act);
      } else if constexpr (!TW::kDefault) {
        group_barrier<kRows, kWR>();
        if constexpr (TW::kRolled)
          hidden_layer_rolled<TW, kWR>(p, ln, l - 1, in, out, p.N, pow2, act);
        else if (l == 1 && res.hw_valid)
          hidden_layer_stream<TW, kWR, resident_groups<TW>()>(p, ln, 0, in, out, hid_rows, act,
                                                              res.hw, res.hwb);
        else
          hidden_layer_stream<TW, kWR>(p, ln, l - 1, in, out, hid_rows, act);
      } else {
        if (!kHoist && l > 1) load_hidden(p, l - 1, ln.lane, res.hid);   // (layer 1: requested above)
        group_barrier<kRows, kWR>();
        if constexpr (TW::kTile16) {
          hidden_layer_t16<kPrio>(in, out, res.hid, res.t16_hid_off, res.t16_st_off, act);
        } else if constexpr (TW::kHalf) {
          static_assert(kKeepOffsets && kWR == 64 && kOneWave, "block-diagonal nets: resident one-wave integrators");
          hidden_layer<32, true, true, kPrio>(p, ln, in, out, res.hid, hid_rows, act, res.st_off);
        } else {
          hidden_layer<kWR, kKeepOffsets, false, kPrio>(p, ln, in, out, res.hid, hid_rows, act, res.st_off);
        }
      }
      float* tmp = in; in = out; out = tmp;
    }
    DDD_STAMP(2);
    {
      // output layer: weights resident (specialised one-wave integrators) or
      // fetched from L2 here, in flight across the forcing sums below
      // (run-time kernels: the first chunk, up to three groups)
      if constexpr (kQuad) {
        constexpr bool kMaskedSums = spec_folded(kSpec ? kEq : 0);
        if (frc_next) res.fk_next = forcing_phase2<kRows, kWR, kMaskedSums>(sm, res);
        group_barrier<kRows, kWR>();   // every tile of the last hidden layer is in LDS
        // this wavefront's own 16 rows, all channels: D[channel 4 (l >> 4) + r][row l & 15]
        const f32x4 d4 = final_layer_quad(in, res.q_fin, res.q_fin_off);
        // back to lane == row through the free activation buffer (last read by the hidden
        // layer: every wavefront has passed the barrier above): rows of this wavefront only,
        // so waiting for its own stores to land is all the synchronisation there is
        *reinterpret_cast<float4*>(reinterpret_cast<char*>(out) + res.q_xch_off) =
            make_float4(d4[0], d4[1], d4[2], d4[3]);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        const float* mine = out + ln.row * kHS;   // natural channel order here
#pragma unroll
        for (int g4 = 0; g4 < kNG; ++g4) {
          const float4 v = *reinterpret_cast<const float4*>(mine + 4 * g4);
          net[4 * g4] = v.x; net[4 * g4 + 1] = v.y; net[4 * g4 + 2] = v.z; net[4 * g4 + 3] = v.w;
        }
      } else if constexpr (kSplit) {
        constexpr int kNA = (kNG + 1) / 2, kNB = kNG - kNA;   // channel groups of wavefront 0 / 1
        constexpr bool kMaskedSums = spec_folded(kSpec ? kEq : 0);
        if (frc_next) res.fk_next = forcing_phase2<kRows, kWR, kMaskedSums>(sm, res);
        group_barrier<kRows, kWR>();   // both tiles of the last hidden layer are in LDS
        int off4[kKW];
#pragma unroll
        for (int k = 0; k < kKW; ++k) off4[k] = res.fin4_off[k];   // rows of lane == row (0 .. 63)
        float* xch = out + ln.lane * kHS;   // this row's channels, in the free buffer
        if (ln.wave == 0) {   // wave-uniform
          float wa[fin4_regs(kNA)];
#pragma unroll
          for (int s2 = 0; s2 < fin4_regs(kNA); ++s2) wa[s2] = res.w_fin4[s2];
          f32x4 acc[kNA];
          final_layer4<kNA>(in, wa, off4, acc);
#pragma unroll
          for (int g4 = 0; g4 < kNA; ++g4)
            *reinterpret_cast<float4*>(xch + 4 * g4) =
                make_float4(acc[g4][0], acc[g4][1], acc[g4][2], acc[g4][3]);
        } else if constexpr (kNB > 0) {
          float wb[fin4_regs(kNB)];
#pragma unroll
          for (int s2 = 0; s2 < fin4_regs(kNB); ++s2) wb[s2] = res.w_fin4[s2];
          f32x4 acc[kNB];
          final_layer4<kNB>(in, wb, off4, acc);
#pragma unroll
          for (int g4 = 0; g4 < kNB; ++g4)
            *reinterpret_cast<float4*>(xch + 4 * (kNA + g4)) =
                make_float4(acc[g4][0], acc[g4][1], acc[g4][2], acc[g4][3]);
        }
        group_barrier<kRows, kWR>();
        const float* mine = out + ln.row * kHS;   // the VALU phases' row of this lane
#pragma unroll
        for (int g4 = 0; g4 < kNG; ++g4) {
          const float4 v = *reinterpret_cast<const float4*>(mine + 4 * g4);
          net[4 * g4] = v.x; net[4 * g4 + 1] = v.y; net[4 * g4 + 2] = v.z; net[4 * g4 + 3] = v.w;
        }
      } else {
      constexpr int kFirstRows = kSpec ? fin4_regs_t<TW>(kNG) : fin4_regs_t<TW>(3);
      float wf4[kFirstRows];
      if (!kKeepRows) {
        if constexpr (kSpec) {
          load_rows4<kFirstRows>(p.w_final4, opaque(ln.lane), wf4);
        } else {
          const float* __restrict__ wsrc = p.w_final4_rt + opaque(ln.lane);
#pragma unroll
          for (int s2 = 0; s2 < kFirstRows; ++s2) wf4[s2] = wsrc[s2 * 64];
        }
      }
      int off4[TW::kK];
      if constexpr (kKeepRows) {
#pragma unroll
        for (int s2 = 0; s2 < kFirstRows; ++s2) wf4[s2] = res.w_fin4[s2];
      }
      if constexpr (!TW::kDefault) {
        int rows_k[TW::kK];
        tap_rows_n<TW>(ln, ln.row, p.N, pow2, rows_k);
#pragma unroll
        for (int k = 0; k < TW::kK; ++k)
          off4[k] = (int)__umul24((unsigned)rows_k[k], (unsigned)(TW::kHS * 4));
      } else if constexpr (kKeepOffsets) {
#pragma unroll
        for (int k = 0; k < kKW; ++k) off4[k] = res.fin4_off[k];
      } else {
        int rows5[kKW];
        tap_rows<kRows == 64>(ln, ln.row, p.N, rows5);
#pragma unroll
        for (int k = 0; k < kKW; ++k)
          off4[k] = (int)__umul24((unsigned)rows5[k], (unsigned)(kHS * 4));
      }
      // the hidden layer's last activations are on their way to LDS: fill the
      // wait with the forcing sums the next evaluation needs
      if (nL == 2) group_barrier<kRows, kWR>();   // no hidden layer: phase 1 -> phase 2 ordering
      // (masked sums where registers allow: the folded kernels; the unfolded
      // non-flux Burgers kernel needs them for the projection)
      constexpr bool kMaskedSums = kKeepRows && spec_folded(kSpec ? kEq : 0);
      if (frc_next) {
        if (kPrio == 4) __builtin_amdgcn_s_setprio(0);
        res.fk_next = forcing_phase2<kRows, kWR, kMaskedSums>(sm, res);
        if (kPrio == 4) __builtin_amdgcn_s_setprio(3);
      }
      group_barrier<kRows, kWR>();
      if constexpr (kSpec) {
        f32x4 acc4[kNG];
        if (!(ablate & 4)) {
          if constexpr (TW::kTile16) final_layer4_t16<kNG, kPrio>(in, wf4, off4, acc4);
          else final_layer4<kNG, TW, kNG, kPrio>(in, wf4, off4, acc4);
        } else {
#pragma unroll
          for (int g4 = 0; g4 < kNG; ++g4) acc4[g4] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        }
        // every lane holds its own row's channels: no LDS round trip, no barrier
#pragma unroll
        for (int g4 = 0; g4 < kNG; ++g4)
#pragma unroll
          for (int r4 = 0; r4 < 4; ++r4) net[4 * g4 + r4] = acc4[g4][r4];
      } else if constexpr (!kSpec) {
        // Only the live groups are issued, as interleaved accumulator chains (two
        // or more chains run at 8.1 cycles per MFMA, a lone one at 13.2): a head
        // chunk of 3 groups when the count is odd (1 when there is only one),
        // then pairs -- a net with 8 channels issues half the matrix work of one
        // with 16 instead of the same.  Wave-uniform branches; every chunk's
        // weights are fetched while the previous chunk's MFMAs run.
        constexpr int kPairRows = fin4_regs_t<TW>(2);
        const int head = rt_head_groups(rt_groups);
        int done = 0;   // groups issued so far
        if (!(ablate & 4)) {
          float wnext[kPairRows];
          const auto fetch_pair = [&](int first_group) {
            const float* __restrict__ wn = p.w_final4_rt +
                (size_t)(fin4_regs_t<TW>(head) + (first_group - head) / 2 * kPairRows) * 64 +
                opaque(ln.lane);
#pragma unroll
            for (int s2 = 0; s2 < kPairRows; ++s2) wnext[s2] = wn[s2 * 64];
          };
          if (head > 0 && head < rt_groups) fetch_pair(head);   // (head 0: pair 0 is in wf4)
          if (head == 3) {
            f32x4 acc3[3];
            final_layer4<3, TW>(in, wf4, off4, acc3);
#pragma unroll
            for (int g4 = 0; g4 < 3; ++g4)
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4) net[4 * g4 + r4] = acc3[g4][r4];
          } else if (head == 1) {
            float w1[fin4_regs_t<TW>(1)];
#pragma unroll
            for (int s2 = 0; s2 < fin4_regs_t<TW>(1); ++s2) w1[s2] = wf4[s2];
            f32x4 acc1[2];   // two accumulator chains over alternate reduction steps, summed here
            final_layer4<1, TW, 2>(in, w1, off4, acc1);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) net[r4] = acc1[0][r4] + acc1[1][r4];
          }
          done = head;
          // pairs: group index 2 j (head 0) or 2 j + 1 (odd head): compile-time slots
#pragma unroll
          for (int j = 0; j < kCh / 8; ++j) {
            if (done >= rt_groups) break;
            float w2[kPairRows];
#pragma unroll
            for (int s2 = 0; s2 < kPairRows; ++s2) w2[s2] = (head == 0 && j == 0) ? wf4[s2] : wnext[s2];
            if (done + 2 < rt_groups) fetch_pair(done + 2);
            f32x4 acc2[2];
            final_layer4<2, TW>(in, w2, off4, acc2);
            if (head == 0) {
#pragma unroll
              for (int r4 = 0; r4 < 4; ++r4) {
                net[8 * j + r4] = acc2[0][r4];
                net[8 * j + 4 + r4] = acc2[1][r4];
              }
            } else if (8 * j + 12 <= kCh) {   // odd head: groups 2 j + head, 2 j + head + 1
              if (head == 1) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                  net[8 * j + 4 + r4] = acc2[0][r4];
                  net[8 * j + 8 + r4] = acc2[1][r4];
                }
              } else if (8 * j + 20 <= kCh) {
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                  net[8 * j + 12 + r4] = acc2[0][r4];
                  net[8 * j + 16 + r4] = acc2[1][r4];
                }
              }
            }
            done += 2;
          }
        }
      }
      }   // !kSplit
      DDD_STAMP(3);
      if (kPrio) __builtin_amdgcn_s_setprio(kPrio >= 2 ? 3 : 0);
    }
  } else {
    if (forced && fast_forcing && prepare_next && !(ablate & 1))
      res.fk_next = forcing_sums<kRows, kWR>(p, sm, res, t_next, tid);
    group_barrier<kRows, kWR>();   // all patch reads done before the next evaluation rewrites sm.u
  }

  if (clamps_nan) {   // NaN through relu (top of this function)
    if (kOneWave) {
      nan_any = __builtin_amdgcn_ballot_w64(u != u) != 0;
      if (nan_any) poisoned = nan_in_receptive_field();
    }
    if (nan_any) {   // wave-uniform, rare
      const float qnan = __int_as_float(0x7fc00000);
#pragma unroll
      for (int c = 0; c < kCh; ++c) net[c] = poisoned ? qnan : net[c];
    }
  }
  // ---- projection onto the accuracy-constrained stencils + stencil apply -----
  // coeff = bias + net[start:stop] @ nullspace   (polynomials.py:275-277)
  // deriv = sum_i coeff[i] * patch[i]            (model.py:548)
  // Output channel c feeds derivative dsel(c) (wave-uniform kernel argument)
  // with the null-space row staged in LDS at tab[4 + c]: per channel one
  // uniform branch, two broadcast ds_read_b128 at compile-time offsets and 8
  // FMAs -- no dependent address chain.
  // this grid point's cos / sin of the spatial phases: issued at the top of
  // the epilogue, consumed by its last statement
  float4 trig4[kTrigMax / 4];
  if (forced && fast_forcing) {
    if (kKeepOffsets && kLean == 0) {
      // resident for the launch: lane == grid point never changes
#pragma unroll
      for (int i = 0; i < kTrigMax / 4; ++i) trig4[i] = res.trig[i];
    } else if (trig_lds) {
      // <= 4 wavenumbers: the table sits in the four padding floats of this
      // row in each activation buffer (launch_setup), never overwritten
      trig4[0] = *reinterpret_cast<const float4*>(sm.hA + ln.row * TW::kHS + TW::kC);
      trig4[1] = *reinterpret_cast<const float4*>(
          sm.hB + ln.row * Shared<kRows, kWR, kWide, TW>::kHBStride + Shared<kRows, kWR, kWide, TW>::kHBPad);
      trig4[2] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    } else {
      const float4* __restrict__ tr =
          reinterpret_cast<const float4*>(p.trig) + (size_t)opaque(ln.pos) * (kTrigMax / 4);
#pragma unroll
      for (int i = 0; i < kTrigMax / 4; ++i) trig4[i] = tr[i];
    }
  }
  if (kOneWave) {
#pragma unroll
    for (int g = 0; g < kGW; ++g)
      pch[g] = (g < nG) ? sm.u[kKeepPatch ? res.pch_idx[g < kGMax ? g : 0]
                               : pow2 ? (((ln.pos + g - gl) & (p.N - 1)) | ln.base)
                                      : wrap_row(ln.base, ln.pos, g - gl, p.N)] : 0.0f;
  }
  // coefficients as register PAIRS: the projection and the bias add run on packed FMAs /
  // adds (two stencil columns per instruction, each column's fma chain unchanged -- the
  // same bits); left to the SLP vectoriser the unfolded kernels spent more v_mov than FMA
  // instructions shuffling pairs together (round 4: 84 v_mov_b32 per evaluation in the
  // non-flux Burgers kernel).  CF(d, g) is the scalar view.
  f32x2 cf2[kMaxDerivs][kGW / 2];
#define CF(d, g) cf2[d][(g) >> 1][(g) & 1]
#pragma unroll
  for (int d = 0; d < kMaxDerivs; ++d)
#pragma unroll
    for (int q = 0; q < kGW / 2; ++q) cf2[d][q] = f32x2{0.0f, 0.0f};
  if (kWide) {
    // wide flavour: the output layer emits the coefficients themselves, wide_slot(G)
    // channels per derivative (capi.hip: pack_mfma_weights folds the projection and the
    // accuracy bias for every wide model).  One wave-uniform branch over the slot width
    // makes every register index a compile-time constant; nothing is left to project.
    if (!fixed && target == TARGET_COEFFICIENTS) {
      const auto take = [&](auto slot_c) {
        constexpr int kSlot = decltype(slot_c)::value;
#pragma unroll
        for (int d = 0; d < kWideDerivs; ++d)
#pragma unroll
          for (int g = 0; g < kSlot; ++g)
            if (kSlot * d + g < kCh && g < kGW) CF(d, g) = net[(kSlot * d + g) % kCh];
      };
      if (nG <= kGMax) take(std::integral_constant<int, kGMax>{});
      else if (nG == 9) take(std::integral_constant<int, 9>{});
      else if (nG == 10) take(std::integral_constant<int, 10>{});
      else if (nG == 11) take(std::integral_constant<int, 11>{});
      else take(std::integral_constant<int, 12>{});
    }
  } else if (!fixed && folded) {
    // the output layer already applied the projection (or the net emits the
    // coefficients themselves, polynomial_accuracy_order 0): channel G d + g,
    // D <= 2.  Register indices must be compile-time: the run-time kernels
    // branch (wave-uniformly) over the stencil widths the host folds for, 6..8.
    if (kSpec) {
#pragma unroll
      for (int g = 0; g < kGW; ++g)
        if (g < nG) { CF(0, g) = net[g]; CF(1, g) = net[nG + g]; }
    } else if (nG == 6) {
#pragma unroll
      for (int g = 0; g < 6; ++g) { CF(0, g) = net[g]; CF(1, g) = net[6 + g]; }
    } else if (nG == 7) {
#pragma unroll
      for (int g = 0; g < 7; ++g) { CF(0, g) = net[g]; CF(1, g) = net[7 + g]; }
    } else {
#pragma unroll
      for (int g = 0; g < 8; ++g) { CF(0, g) = net[g]; CF(1, g) = net[8 + g]; }
    }
  } else if (!fixed && !(ablate & 2)) {
#pragma unroll
    for (int c = 0; c < kCh; ++c) {
      // specialised kernels: the null-space split is known (spec_in_size, checked
      // by capi.hip: spec_equation), so the channel -> derivative map is a
      // compile-time constant
      if (kSpec ? c >= spec_net_channels(kSpec ? kEq : 0) : !((dsel_valid >> c) & 1u)) continue;
      const unsigned d = kSpec ? (unsigned)spec_channel_deriv(kSpec ? kEq : 0, c)
                               : (unsigned)(dsel_bits >> (2 * c)) & 3u;
      const f32x2 nv2{net[c], net[c]};
      f32x2 nsr[kGW / 2];   // the channel's null-space row (zero beyond the stencil: fma(nv, 0, 0) = 0)
#pragma unroll
      for (int q4 = 0; q4 < kGW / 4; ++q4) {
        const float4 nq = *reinterpret_cast<const float4*>(sm.tab + (4 + c) * kGW + 4 * q4);
        nsr[2 * q4] = f32x2{nq.x, nq.y};
        nsr[2 * q4 + 1] = f32x2{nq.z, nq.w};
      }
      // (kSpec: pairs past the stencil are skipped at compile time)
      if (d == 0) {
#pragma unroll
        for (int q = 0; q < kGW / 2; ++q)
          if (!kSpec || 2 * q < nG) cf2[0][q] = __builtin_elementwise_fma(nv2, nsr[q], cf2[0][q]);
      } else if (d == 1) {
#pragma unroll
        for (int q = 0; q < kGW / 2; ++q)
          if (!kSpec || 2 * q < nG) cf2[1][q] = __builtin_elementwise_fma(nv2, nsr[q], cf2[1][q]);
      } else if (d == 2) {
#pragma unroll
        for (int q = 0; q < kGW / 2; ++q)
          if (!kSpec || 2 * q < nG) cf2[2][q] = __builtin_elementwise_fma(nv2, nsr[q], cf2[2][q]);
      } else {
#pragma unroll
        for (int q = 0; q < kGW / 2; ++q) cf2[3][q] = __builtin_elementwise_fma(nv2, nsr[q], cf2[3][q]);
      }
    }
  }
  if constexpr (!kSpec && !kWide) {
    if (p.linear_taps != 0) {
      // one-layer net: coeff += sum_k M[k][d] (u / std)[x + k - K/2]  (DevParams::linear_taps);
      // the bias part B is added with the table rows below
      const int taps = opaque_scalar(p.linear_taps), left = taps >> 1;
      for (int k = 0; k < taps; ++k) {   // wave-uniform trip count
        const float uk = sm.u[pow2 ? (((ln.pos + k - left) & (p.N - 1)) | ln.base)
                                   : wrap_row(ln.base, ln.pos, k - left, p.N)];
        const float qk = uk * p.inv_stddev;
        float unk = fmaf(fmaf(-qk, p.stddev, uk), p.inv_stddev, qk);   // u / std, as in the tower
        if (p.exact_div) { asm volatile("; exact division"); unk = uk / p.stddev; }
#pragma unroll
        for (int d = 0; d < kMaxDerivs - 1; ++d) {
          if (d >= nD) continue;
          const float* __restrict__ row = sm.tab + (4 + k * nD + d) * kGW;
          const float4 m0 = *reinterpret_cast<const float4*>(row);
          const float4 m1 = *reinterpret_cast<const float4*>(row + 4);
          const f32x2 u2{unk, unk};
          cf2[d][0] = __builtin_elementwise_fma(u2, f32x2{m0.x, m0.y}, cf2[d][0]);
          cf2[d][1] = __builtin_elementwise_fma(u2, f32x2{m0.z, m0.w}, cf2[d][1]);
          cf2[d][2] = __builtin_elementwise_fma(u2, f32x2{m1.x, m1.y}, cf2[d][2]);
          cf2[d][3] = __builtin_elementwise_fma(u2, f32x2{m1.z, m1.w}, cf2[d][3]);
        }
      }
      // four-wave groups: a faster wavefront rewrites sm.u as soon as it enters the next
      // evaluation -- not before every wavefront has read its neighbours here
      if (!kOneWave) group_barrier<kRows, kWR>();
    }
  }
  if (!kSpec && !fixed && p.pao <= 0 && p.unbiased) {
    // ensure_unbiased_coefficients (model.py:471-475): subtract the mean over
    // the stencil (same order of operations as the generic kernel)
#pragma unroll
    for (int d = 0; d < kMaxDerivs; ++d) {
      if (d >= nD) continue;
      float mean = 0.0f;
#pragma unroll
      for (int g = 0; g < kGW; ++g) if (g < nG) mean += CF(d, g);
      mean = mean / (float)nG;
#pragma unroll
      for (int g = 0; g < kGW; ++g) if (g < nG) CF(d, g) = CF(d, g) - mean;
    }
  }
  float dv[kMaxDerivs];
#pragma unroll
  for (int d = 0; d < kMaxDerivs; ++d) {
    dv[d] = 0.0f;
    if (d < nD) {
      if (!kSpec && target == TARGET_SPACE_DERIVATIVES) {   // predicted directly
        dv[d] = net[d];
        continue;
      }
      if (fixed || (!folded && !kWide)) {   // (folded / wide nets: the bias rides in the output layer's bias row)
#pragma unroll
        for (int q = 0; q < kGW / 2; ++q) {
          const float2 bq = *reinterpret_cast<const float2*>(sm.tab + d * kGW + 2 * q);
          cf2[d][q] = f32x2{bq.x, bq.y} + cf2[d][q];
        }
      }
      if (coeffs_out != nullptr && ln.active) {
        float* dst = coeffs_out + ((size_t)ln.gidx * nD + d) * nG;
#pragma unroll
        for (int g = 0; g < kGW; ++g) if (g < nG) dst[g] = CF(d, g);
      }
      // (one scalar chain per derivative, in stencil order -- the order of the streaming
      // kernel and the oracle.  The empty asm after every link keeps the SLP vectoriser
      // from pairing the derivatives' chains into packed FMAs: it paid two v_mov per pair
      // to build their operands.  It emits nothing, so the FMAs stay visible to the hazard
      // recognizer -- their operands come straight from the output layer's MFMAs.)
      float s = 0.0f;
#pragma unroll
      for (int g = 0; g < kGW; ++g)
        if (!kSpec || g < nG) {   // padded columns: cf = 0
          s = fmaf(CF(d, g), pch[g], s);
          asm("" : "+v"(s));
        }
      dv[d] = s;
    }
  }
  if (derivs_out != nullptr && ln.active) {
#pragma unroll
    for (int d = 0; d < kMaxDerivs; ++d)
      if (d < nD) derivs_out[(size_t)ln.gidx * nD + d] = dv[d];
  }

  // ---- equation of motion ------------------------------------------------------
  float r = equation_rhs_or_flux(eqn, u, dv, p.eta);
  // direct heads: the tower's single channel IS u_t (no flux difference, whatever
  // the equation's form) or the flux (model.predict_flux_directly returns
  // +staggered_first_derivative(flux), model.py:609-615)
  const bool direct_time = !kSpec && !fixed && target == TARGET_TIME_DERIVATIVE;
  const bool direct_flux = !kSpec && !fixed && target == TARGET_FLUX;
  if (direct_time || direct_flux) r = net[0];
  if ((flux_form && !direct_time) || direct_flux) {
    float fnext;
    if (kOneWave) {
      // whole samples live in this wavefront: the right neighbour's flux comes
      // straight from its lane
      if (p.N == 64 && p.dpp_rol)   // one sample = one wavefront: a DPP rotate
        fnext = ops::wave_rotate_left1(r);
      else
        fnext = __shfl(r, pow2 ? (((ln.pos + 1) & (p.N - 1)) | ln.base)
                                    : wrap_row(ln.base, ln.pos, 1, p.N), 64);
    } else {
      if (ln.owner) sm.flux[ln.row] = r;
      group_barrier<kRows, kWR>();
      fnext = sm.flux[wrap_row(ln.base, ln.pos, 1, p.N)];
    }
    r = p.inv_dx * (fnext - r);      // equations.staggered_first_derivative
    if (!direct_flux) r = -r;        // flux forms: u_t = -d(flux)/dx
  }
  if (forced && !(ablate & 1)) {
    if (kSpec || fast_forcing) {
      // phase 3: combine with this grid point's cos / sin table (both zero
      // padded to 12 entries: no branches)
      const float4* __restrict__ fk4 =
          kKeepOffsets ? reinterpret_cast<const float4*>(reinterpret_cast<const char*>(sm.fk) + res.fk_off)
                       : reinterpret_cast<const float4*>(sm.fk + ln.sl * kTrigMax);
      float total = 0.0f;
#pragma unroll
      for (int i = 0; i < kTrigMax / 4; ++i) {
        if (i == 2 && trig_lds && !(kKeepOffsets && kLean == 0)) break;   // entries 8..11 are zero padding
        const float4 f = fk4[i];
        total = fmaf(f.x, trig4[i].x, total);
        total = fmaf(f.y, trig4[i].y, total);
        total = fmaf(f.z, trig4[i].z, total);
        total = fmaf(f.w, trig4[i].w, total);
      }
      r = r + total;
    } else if (ln.valid) {
      r = r + forcing_at(p, p.frc + (size_t)(ln.gidx / p.N) * p.P, ln.pos, t);
    }
  }
  DDD_STAMP(4);
#undef DDD_STAMP
#undef CF
  return r;
}

template <int kRows, int kWR, bool kWide, class TW>
__device__ __forceinline__ void setup_samples(const DevParams& p, Shared<kRows, kWR, kWide, TW>& sm,
                                              int block, int batch, Resident& res, bool fast,
                                              int nb = 1);

// The forcing of this launch can take the harmonic-sum path (else: per-point sinf).
template <int kRows, int kWR>
__device__ __forceinline__ bool forcing_is_fast(const DevParams& p) {
  const int spg = kRows / p.N;
  return p.forced && spg * p.P <= Shared<kRows, kWR>::kPmMax && p.n_k <= 6 &&
         spg * kTrigMax <= Shared<kRows, kWR>::kFkMax && p.P < 256;
}

// Forcing BATCHES (persistent one-wave integrators, one sample per wavefront): phases 1 and 2
// of the harmonic sums occupy P = 20 and 2 n_k <= 12 lanes of 64, and every VALU instruction
// costs the matrix pipe its issue time whatever its lane mask.  With the evaluation times of a
// fixed-step scheme known in advance, lanes [b P, (b + 1) P) take the time of the (b + 1)-th
// next evaluation: one pass of phases 1 + 2 in every `nb`-th evaluation serves the next nb
// (43 instructions per pass: ~14 per evaluation instead of 43).  Every element is computed
// by the same instructions from the same operands as before -- the same bits as the
// one-launch-per-substep kernels.  Layout: batch b is "sample slot" b of the staging
// buffers (Shared::pm set b at entry b P, Shared::fk set b at entry b kTrigMax), exactly
// where the samples of a multi-sample group sit; phase 3 reads set e mod nb.
// MEASURED NEUTRAL (profiles/r5_ablation.txt: 82.5 % with and without; 61 GPU tests incl. the
// bit-for-bit launch-mode comparisons passed with it): phases 1 and 2 already sit in the LDS
// waits at the layer boundaries, their instructions were not what the matrix pipe waited
// for.  Off by default; -DDDD_FORCING_BATCH=1 builds it.
#ifndef DDD_FORCING_BATCH
#define DDD_FORCING_BATCH 0
#endif
constexpr int kForcingBatchMax = 3;
template <int kRows, int kWR>
__device__ __forceinline__ int forcing_batches(const DevParams& p) {
  if (!DDD_FORCING_BATCH || kRows != 64 || kWR != 64 || !forcing_is_fast<kRows, kWR>(p)) return 1;
  if (2 * p.N <= kRows) return 1;   // more than one sample per wavefront: no spare lanes
  const int nb = p.P > 0 ? 64 / p.P : 1;
  return nb > kForcingBatchMax ? kForcingBatchMax : (nb < 1 ? 1 : nb);
}

// The per-lane loop invariants of the kernels that keep them resident
// (Resident::fin4_off .. pch_idx): functions of the lane, N and G alone.
// (Loading them from a per-model table instead -- 115 VALU instructions fewer per
// launch -- changed nothing measurable: profiles/r3_ablation.txt.)
template <int kRows, int kWR, bool kHalfNet = false, bool kTile16 = false>
__device__ __forceinline__ void lane_offsets(const DevParams& p, const Lane& ln, Resident& res) {
  int rows[kKW];
  if constexpr (kTile16) {
    static_assert(kRows == 64 && kWR == 64, "16-channel tiles: one-wave groups");
    const int sg = ln.lane >> 4, j16 = ln.lane & 15;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int trow = 16 * t + j16;
      tap_rows<true>(ln, trow, p.N, rows);
#pragma unroll
      for (int k = 0; k < kKW; ++k)
        res.t16_hid_off[t][k] = opaque((int)__umul24((unsigned)rows[k], (unsigned)(kHS * 4)) + 16 * sg);
      res.t16_in_perm[t][0] = opaque(4 * (sg == 0 ? rows[0] : sg == 1 ? rows[1] : sg == 2 ? rows[2] : rows[3]));
      res.t16_in_perm[t][1] = opaque(4 * rows[4]);
      res.t16_st_off[t] = opaque((int)__umul24((unsigned)trow, (unsigned)(kHS * 4)) + 4 * sg);
    }
  }
  if constexpr (kWR == 16) {
    // four wavefronts per group: tiles of 16 positions, reduction slot sg = lane >> 4
    const int sg = ln.lane >> 4, j16 = ln.lane & 15;
    const int ph = ln.wave & 1, chh = ln.wave >> 1;
#pragma unroll
    for (int t2 = 0; t2 < 2; ++t2) {
      const int trow = 32 * ph + 16 * t2 + j16;
      tap_rows<kRows == 64>(ln, trow, p.N, rows);
#pragma unroll
      for (int k = 0; k < kKW; ++k)   // block (sg & 1, sg >> 1) of the tap's row
        res.q_hid_off[t2][k] = opaque((int)__umul24((unsigned)rows[k], (unsigned)(kHS * 4)) +
                                      16 * (4 * (sg & 1) + (sg >> 1)));
      // input layer: step 0 = tap sg (taps 0..3), step 1 = tap 4 (slot 0)
      res.q_in_off[t2][0] = opaque(4 * (sg == 0 ? rows[0] : sg == 1 ? rows[1] : sg == 2 ? rows[2] : rows[3]));
      res.q_in_off[t2][1] = opaque(4 * rows[4]);
      // D of the tile: channels 16 chh + 4 sg + r -> block (chh, r), element sg
      res.q_st_off[t2] = opaque((int)__umul24((unsigned)trow, (unsigned)(kHS * 4)) + 64 * chh + 4 * sg);
    }
    tap_rows<kRows == 64>(ln, ln.row, p.N, rows);   // the output layer: this wavefront's own rows
#pragma unroll
    for (int k = 0; k < kKW; ++k)
      res.q_fin_off[k] = opaque((int)__umul24((unsigned)rows[k], (unsigned)(kHS * 4)) + 16 * sg);
    res.q_xch_off = opaque((int)__umul24((unsigned)ln.row, (unsigned)(kHS * 4)) + 16 * sg);
  }
  // (two 32-row wavefronts per sample: the output layer runs lane == row over all 64 rows)
  tap_rows<kRows == 64>(ln, kWR == 32 ? ln.lane : ln.row, p.N, rows);
#pragma unroll
  for (int k = 0; k < kKW; ++k)   // opaque: keep it in a register, do not recompute
    res.fin4_off[k] = opaque((int)__umul24((unsigned)rows[k], (unsigned)(kHS * 4)));
  const int half = ln.lane >> 5;
#pragma unroll
  for (int t2 = 0; t2 < 2; ++t2) {
    tap_rows<kRows == 64>(ln, ln.wave * kWR + t2 * 32 + (ln.lane & 31), p.N, rows);
#pragma unroll
    for (int k = 0; k < kKW; ++k)
      res.hid_off[t2][k] = opaque((int)__umul24((unsigned)rows[k], (unsigned)(kHS * 4)) +
                                  64 * half);
    if constexpr (kHalfNet) {
      // block-diagonal nets: ONE tile; reduction half h reads channels 0..15 of the tap rows
      // of position 32 h + j
      if (t2 == 0) {
        int rows_h[kKW];
        tap_rows<kRows == 64>(ln, ln.wave * kWR + ln.lane, p.N, rows_h);   // (lane = 32 half + j)
#pragma unroll
        for (int k = 0; k < kKW; ++k)
          res.hid_off[0][k] = opaque((int)__umul24((unsigned)rows_h[k], (unsigned)(kHS * 4)));
      }
    }
    // input layer: taps 0 / 1, taps 2 / 3, tap 4, as byte addresses (one-wave
    // groups: ds_bpermute lane addresses, rows < 64; else into Shared::un)
    res.in_perm[t2][0] = opaque(4 * (half ? rows[1] : rows[0]));
    res.in_perm[t2][1] = opaque(4 * (half ? rows[3] : rows[2]));
    res.in_perm[t2][2] = opaque(4 * rows[4]);
  }
  res.st_off = opaque((int)__umul24((unsigned)(ln.wave * kWR + (ln.lane & 31)), (unsigned)(kHS * 4)) +
                      16 * half);
  const int gl = p.G >> 1;
  const bool pow2 = kRows == 64 || (p.N & (p.N - 1)) == 0;
#pragma unroll
  for (int g = 0; g < kGMax; ++g)
    res.pch_idx[g] = opaque(pow2 ? (((ln.pos + g - gl) & (p.N - 1)) | ln.base)
                                 : wrap_row(ln.base, ln.pos, g - gl, p.N));
}

// Per-launch setup, part 1: resident registers and the tables in LDS.
template <int kRows, int kWR, bool kHoist, bool kKeepTower = true, bool kWide, class TW>
__device__ __forceinline__ bool setup_weights(const DevParams& p, Shared<kRows, kWR, kWide, TW>& sm,
                                              const Lane& ln, Resident& res, int nb = 1) {
  constexpr int kThreads = kRows / kWR * 64;
  const int tid = group_tid<kRows, kWR>();
  constexpr int kGW = flavour_stencil(kWide);
  const bool fast = forcing_is_fast<kRows, kWR>(p);
  // ---- every load of the setup is REQUESTED before anything is consumed: the
  // tables, then the weights in the order the evaluation needs them.  (Written
  // as load + LDS write pairs, each pair waited for its own memory round trip
  // before the next load went out.)
  constexpr int kTabN = tab_rows(kWide) * kGW;
  constexpr int kTabTrips = (kTabN + kThreads - 1) / kThreads;
  float tabv[kTabTrips];
#pragma unroll
  for (int k = 0; k < kTabTrips; ++k) {
    const int i = min(tid + k * kThreads, kTabN - 1);
    const int rowi = i / kGW, g = i % kGW;
    tabv[k] = rowi < 4 ? p.bias8[rowi][g] : p.ns8[rowi - 4][g];
  }
#pragma unroll
  for (int s = 0; s < kInSteps; ++s) res.w_in[s] = 0.0f;
  if constexpr (!TW::kDefault && !TW::kRolled && kKeepTower) {
    // streamed towers: the leading groups of hidden layer 0 (resident_groups)
    constexpr int kRes = resident_groups<TW>();
    static_assert(kRes * TW::kCB <= kResidentQuads, "Resident::hw");
    if (!p.fixed && !p.linear_taps && p.L > 2) {
      const float4* __restrict__ wq = reinterpret_cast<const float4*>(p.w_hidden) + opaque(ln.lane);
#pragma unroll
      for (int i = 0; i < kRes * TW::kCB; ++i) res.hw[i] = wq[i * 64];
      const float* __restrict__ wbias = p.w_hidden + TW::kHidGroups * TW::kCB * 64 * 4 + opaque(ln.lane);
#pragma unroll
      for (int h = 0; h < TW::kCB; ++h) res.hwb[h] = wbias[h * 64];
    }
    res.hw_valid = true;
  }
  if constexpr (kWR == 16) {
    // four wavefronts per group: the 16x16x4 A operands of this wavefront's channel half
    // (DevParams::w_quad: [2][2] input rows, [2][41] hidden rows, [41] output rows x 64 lanes)
    if (!p.fixed && !p.linear_taps && kHoist && p.w_quad != nullptr) {
      const float* __restrict__ wq = p.w_quad + opaque(ln.lane);
      const int chh = ln.wave >> 1;
#pragma unroll
      for (int s2 = 0; s2 < kQuadInSteps; ++s2) res.q_in[s2] = wq[(chh * kQuadInSteps + s2) * 64];
#pragma unroll
      for (int s2 = 0; s2 < kQuadHidSteps; ++s2)
        res.q_hid[s2] = wq[(2 * kQuadInSteps + chh * kQuadHidSteps + s2) * 64];
#pragma unroll
      for (int s2 = 0; s2 < kQuadFinSteps; ++s2)
        res.q_fin[s2] = wq[(2 * kQuadInSteps + 2 * kQuadHidSteps + s2) * 64];
    }
  } else
  if constexpr (TW::kTile16) {
    // A operands of the 16x16x4 layers (DevParams::w_quad in this mode: [2] input rows, [21]
    // hidden rows of 64 lanes), the output layer's as HalfTower (DevParams::w_final4)
    const float* __restrict__ wq = p.w_quad + opaque(ln.lane);
#pragma unroll
    for (int s2 = 0; s2 < kT16InSteps; ++s2) res.w_in[s2] = wq[s2 * 64];
#pragma unroll
    for (int s2 = 0; s2 < kT16HidSteps; ++s2) res.hid[s2] = wq[(kT16InSteps + s2) * 64];
    load_rows4<fin4_regs(4)>(p.w_final4, ln.lane, res.w_fin4);
  } else
  if (!p.fixed && !p.linear_taps && TW::kDefault) {   // (other towers stream every layer's weights)
    load_rows4<kInSteps>(p.w_input, ln.lane, res.w_in);
    if (kHoist) load_hidden(p, 0, ln.lane, res.hid);
    // loop invariants the specialised one-wave integrators keep resident
    // (kHoist: the persistent kernels; a single fused substep has no loop)
    if (kHoist && kWR == 64) {
      if (p.w_final4 != nullptr)   // (dead in the run-time kernels, null for wide models)
        load_rows4<fin4_regs(4)>(p.w_final4, ln.lane, res.w_fin4);   // (zero padded to 4 groups)
    }
    if (kHoist && kWR == 32 && p.w_final4_split != nullptr) {
      // split integrators: this wavefront's chunk of channel groups (two chunks of
      // padded_rows4(fin4_regs(2)) rows each; dead code in the run-time kernels)
      float chunk[fin4_regs(2)];
      load_rows4<fin4_regs(2)>(p.w_final4_split + (size_t)ln.wave * padded_rows4(fin4_regs(2)) * 64,
                               ln.lane, chunk);
#pragma unroll
      for (int s2 = 0; s2 < fin4_regs(2); ++s2) res.w_fin4[s2] = chunk[s2];
    }
  }
  // cos / sin of this grid point's spatial phases (requested LAST: the register
  // allocator copies one of these values right after the load, and that wait
  // must not sit in front of the other requests)
  // (defined on both paths: declared without a value and loaded under `fast`, the array
  // lived in a 64-byte SCRATCH frame -- a store and a load per thread and launch, which
  // the one-launch-per-substep kernels paid in HBM writes, profiles/r5_spill_table.txt)
  float4 trg[kTrigMax / 4];
#pragma unroll
  for (int i = 0; i < kTrigMax / 4; ++i) trg[i] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (fast) {   // wave-uniform
    const float4* __restrict__ tr =
        reinterpret_cast<const float4*>(p.trig) + (size_t)ln.pos * (kTrigMax / 4);
#pragma unroll
    for (int i = 0; i < kTrigMax / 4; ++i) trg[i] = tr[i];
  }
  // ---- index math (no memory) ----
  {
    // (forcing batches, nb > 1: one sample per wavefront, batch b in "sample slot" b)
    const int spg = nb > 1 ? nb : kRows / p.N;
    const int sl = row_sample(tid >> 1, p.inv_nk);   // exact
    res.frc_slot = (fast && tid < spg * p.n_k * 2)
                       ? sl * kTrigMax + 2 * ((tid >> 1) - sl * p.n_k) + (tid & 1) : -1;
    res.frc_pairs = spg * p.P;
  }
  res.fk_off = opaque(ln.sl * kTrigMax * 4);   // (fixed-stencil models with forcing read it too)
  res.st_off = 0;
  if (!p.fixed && !p.linear_taps && kHoist && (kWR == 64 || kWR == 16 || p.w_final4_split != nullptr))
    lane_offsets<kRows, kWR, TW::kHalf, TW::kTile16>(p, ln, res);
  // staged (sample, mode) values: zero once, so that reads past a run are finite
  for (int i = tid; i < Shared<kRows, kWR>::kPmMax + 8; i += kThreads)
    sm.pm[i] = make_float2(0.0f, 0.0f);
  if (tid == 0) sm.nan_flag = 0;
#ifdef DDD_PROBES
  if (res.probe_stamp != nullptr && tid == 0)
    *res.probe_stamp = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- the tables land ----
#pragma unroll
  for (int k = 0; k < kTabTrips; ++k) {
    const int i = tid + k * kThreads;
    if (i < kTabN) sm.tab[i] = tabv[k];
  }
#pragma unroll
  for (int i = 0; i < kTrigMax / 4; ++i)
    res.trig[i] = (fast && kHoist) ? trg[i] : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  if (fast && p.n_k <= 4 && ln.owner) {
    // ... and into the row padding of the activation buffers
    *reinterpret_cast<float4*>(sm.hA + ln.row * TW::kHS + TW::kC) = trg[0];
    *reinterpret_cast<float4*>(sm.hB + ln.row * Shared<kRows, kWR, kWide, TW>::kHBStride +
                               Shared<kRows, kWR, kWide, TW>::kHBPad) = trg[1];
  }
  return fast;
}

// Per-launch setup of the persistent integrators and the one-group substep kernel.
template <int kRows, int kWR, bool kHoist, bool kKeepTower = true, bool kWide, class TW>
__device__ __forceinline__ bool launch_setup(const DevParams& p, Shared<kRows, kWR, kWide, TW>& sm,
                                             const Lane& ln, int batch, Resident& res,
                                             int nb = 1) {
  const bool fast = setup_weights<kRows, kWR, kHoist, kKeepTower>(p, sm, ln, res, nb);
  setup_samples<kRows, kWR>(p, sm, (int)blockIdx.x, batch, res, fast, nb);
  return fast;
}

// The part of the setup that belongs to the group's SAMPLES (their forcing
// parameters and mode runs): once per launch in the persistent integrators,
// once per group in the multi-group substep kernel, which fetches the next
// group's values (fetch_samples) while the current group is being evaluated.
// fetch_samples only REQUESTS (nothing in it depends on a loaded value, so no
// wait is placed there); apply_samples is where the values are first touched.
struct SampleSetup {
  float a, omega, phi;   // this lane's (sample, mode) forcing parameters
  int runs_raw;          // first mode of this lane's run | (first mode of the next run) << 8
  int run_base;          // Resident::frc_run without the loaded part
  int has_slot;          // 0: the lane carries no (sample, k, sin|cos) slot
};

template <int kRows, int kWR>
__device__ __forceinline__ SampleSetup fetch_samples(const DevParams& p, int block, int batch,
                                                     bool fast, int nb = 1) {
  const int tid = group_tid<kRows, kWR>();
  const int spg = kRows / p.N;
  // forcing batches (nb > 1, spg == 1): "sample slot" b of the staging buffers holds batch b
  // of the group's ONE sample -- the same index math with nb slots, all reading sample 0
  const int slots = nb > 1 ? nb : spg;
  SampleSetup s{0.0f, 0.0f, 0.0f, 0, 0, 0};
  if (fast) {   // wave-uniform
    // Both loads are UNCONDITIONAL (lanes without a pair / slot, and samples past
    // the batch, read entry 0: finite values that reach no stored row): a load
    // under a lane mask is waited for where the mask closes, and a wavefront
    // that stalls one memory round trip per load before it has even requested
    // its weights is what made the start of a launch cost more than an
    // evaluation (profiles/tools/substep_wave_trace.py).
    const int fsl = row_sample(tid, p.inv_P);   // tid / P, exact
    const long sample = (long)block * spg + (nb > 1 ? 0 : fsl);
    const bool has_pair = tid < slots * p.P && sample < batch;
    const float4* __restrict__ row = p.frc + (has_pair ? sample * p.P + (tid - fsl * p.P) : 0);
    s.a = row->x; s.omega = row->y; s.phi = row->z;
    // this lane's run of modes [m0, m1): runs[sample][kk] = first (sorted) mode of
    // the sample whose k index is >= kk, precomputed by ddd_set_forcing
    const int sl = row_sample(tid >> 1, p.inv_nk);   // exact
    const int kk = (tid >> 1) - sl * p.n_k;
    const long sample2 = (long)block * spg + (nb > 1 ? 0 : sl);
    s.has_slot = tid < slots * p.n_k * 2 && sample2 < batch;
    const unsigned char* __restrict__ rr = p.runs + (s.has_slot ? sample2 * 8 + kk : 0);
    s.runs_raw = (int)rr[0] | ((int)rr[1] << 8);
    s.run_base = (2 * (sl * p.P) + (tid & 1)) | ((sl * kTrigMax + 2 * kk + (tid & 1)) << 24);
  }
  return s;
}

// kReset: also forget the pending harmonic sum and clear Shared::fk (a fresh
// launch); without it only the parameters the NEXT sums are computed from change.
template <int kRows, int kWR, bool kReset = true, bool kWide = false, class TW = DefaultTower>
__device__ __forceinline__ void apply_samples(Shared<kRows, kWR, kWide, TW>& sm, Resident& res,
                                              const SampleSetup& s) {
  constexpr int kThreads = kRows / kWR * 64;
  res.frc_a = s.a; res.frc_omega = s.omega; res.frc_phi = s.phi;
  const int m0 = s.runs_raw & 0xff, m1 = s.runs_raw >> 8;
  // (the low half stays below 2^16: 2 (samples x modes) + 1, as before)
  res.frc_run = s.has_slot ? (s.run_base + 2 * m0) | ((m1 - m0) << 16) : 0;
  const int cnt = (res.frc_run >> 16) & 0xff;
#pragma unroll
  for (int i = 0; i < 8; ++i) res.frc_mask[i] = i < cnt ? 1.0f : 0.0f;
  if (kReset) {
    res.fk_next = 0.0f;
    for (int i = group_tid<kRows, kWR>(); i < Shared<kRows, kWR>::kFkMax; i += kThreads) sm.fk[i] = 0.0f;
  }
}

template <int kRows, int kWR, bool kWide, class TW>
__device__ __forceinline__ void setup_samples(const DevParams& p, Shared<kRows, kWR, kWide, TW>& sm,
                                              int block, int batch, Resident& res,
                                              bool fast, int nb) {
  apply_samples<kRows, kWR>(sm, res, fetch_samples<kRows, kWR>(p, block, batch, fast, nb));
}

// ---------------------------------------------------------------------------
// Kernel 1: one fused RK substep (also: plain time derivative, derivative and
// coefficient views).  State crosses HBM once in and once out.
// ---------------------------------------------------------------------------
template <int kRows, int kWR, int kEq = -1, bool kWide = false, class TW = DefaultTower>
__global__ __launch_bounds__(kRows / kWR * 64, (min_waves<kRows, kWR, TW>())) void substep_kernel(
    DevParams p, SubstepArgs a) {
  __shared__ Shared<kRows, kWR, kWide, TW> sm;
  const Lane ln = make_lane<kRows, kWR>(p, a.batch, threadIdx.x, blockIdx.x);
  Resident res;
  const bool fast_frc = launch_setup<kRows, kWR, false>(p, sm, ln, a.batch, res);
  const float u = ln.valid ? a.y_in[ln.gidx] : 0.0f;   // both half-waves carry the state
  if (fast_frc) res.fk_next = forcing_sums<kRows, kWR>(p, sm, res, (float)a.t, threadIdx.x);
  const float f = eval_rhs<kRows, kWR, false, kEq, false>(p, sm, a.batch, u, (float)a.t, (float)a.t, res,
                                              fast_frc, a.derivs_out, a.coeffs_out, false);   // (kWide, TW: from sm)
  if (!ln.active) return;
  if (a.y_out != nullptr) {
    const float cf = a.c1 * f;
    a.y_out[ln.gidx] = a.y_base != nullptr ? a.y_base[ln.gidx] + cf : cf;
  }
  if (a.acc_out != nullptr) {
    const float cf = a.c2 * f;
    a.acc_out[ln.gidx] = a.acc_in != nullptr ? a.acc_in[ln.gidx] + cf : cf;
  }
}

// Kernel 1b: the same fused substep for the per-equation specialised models,
// built like the persistent integrator: the grid is sized to the machine (two
// wavefronts per SIMD), every group keeps the conv weights and operand offsets
// resident and walks over `groups / gridDim.x` row groups.  A launch per group
// (kernel 1) makes every wavefront fetch its 29 KB of weights again -- 120 MB of
// L2 traffic per substep at batch 4096 -- and pays a second dispatch round.
// The walk of one row-group slot: `first` = its first row group, `stride` = the
// number of slots of the launch.
template <int kRows, int kWR, int kEq, class TW = DefaultTower>
__device__ __forceinline__ void substep_walk(const DevParams& p, const SubstepArgs& a,
                                             Shared<kRows, kWR, false, TW>& sm, int groups, int first,
                                             int stride) {
  const int tid = group_tid<kRows, kWR>();
#ifdef DDD_PROBES
  // wave lifetime: [0] first instruction, [1] weights / first sums ready, [2..5] end
  // of each row group, [6] hardware id, [7] last instruction
  unsigned long long* stamp =
      a.trace != nullptr ? a.trace + ((size_t)a.trace_row0 + first) * 8 : nullptr;
  int stamp_at = 2;
  if (stamp != nullptr && tid == 0) {
    stamp[0] = __builtin_amdgcn_s_memrealtime();
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);     // HW_ID
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    stamp[6] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
  }
#endif
  Lane ln = make_lane<kRows, kWR>(p, a.batch, tid, first);
  Resident res;
#ifdef DDD_PROBES
  res.probe_stamp = stamp != nullptr ? stamp + 4 : nullptr;   // index math of the setup done
#endif
  // the first group's state and forcing rows are requested BEFORE the 115 weight
  // registers: its input layer and forcing sums run while the hidden layer's
  // weights are still arriving (loads return in order)
  float u = ln.valid ? a.y_in[ln.gidx] : 0.0f;
  const bool fast_frc = forcing_is_fast<kRows, kWR>(p);
  const SampleSetup s_first = fetch_samples<kRows, kWR>(p, first, a.batch, fast_frc);
  setup_weights<kRows, kWR, true>(p, sm, ln, res);
#ifdef DDD_PROBES
  if (stamp != nullptr && tid == 0) stamp[5] = __builtin_amdgcn_s_memrealtime();   // loads issued
#endif
  apply_samples<kRows, kWR>(sm, res, s_first);
  if (fast_frc) res.fk_next = forcing_sums<kRows, kWR, true>(p, sm, res, (float)a.t, tid);
#ifdef DDD_PROBES
  if (stamp != nullptr && tid == 0) stamp[1] = __builtin_amdgcn_s_memrealtime();
#endif
  for (int grp = first; grp < groups; grp += stride) {
    // the next group's state and forcing rows: in flight during this evaluation
    const int nxt = grp + stride;
    const bool more = nxt < groups;
    Lane ln_next = ln;
    float u_next = 0.0f;
    SampleSetup s_next{0.0f, 0.0f, 0.0f, 0, 0, 0};
    if (more) {
      ln_next = make_lane<kRows, kWR>(p, a.batch, tid, nxt);
      u_next = ln_next.valid ? a.y_in[ln_next.gidx] : 0.0f;
      s_next = fetch_samples<kRows, kWR>(p, nxt, a.batch, fast_frc);
    }
    // the update's other operands: requested now, consumed after the evaluation
    const float base = (ln.active && a.y_out != nullptr && a.y_base != nullptr)
                           ? a.y_base[ln.gidx] : 0.0f;
    const float acc_in = (ln.active && a.acc_out != nullptr && a.acc_in != nullptr)
                             ? a.acc_in[ln.gidx] : 0.0f;
    // from here on the harmonic sums computed are the NEXT group's: this
    // group's are already in res.fk_next and are published by eval_rhs
    // (after the requests above: one wait for all of them)
    if (more) apply_samples<kRows, kWR, false>(sm, res, s_next);
    // `more`: the evaluation also prepares the sums of the next group (same time,
    // other samples) at its layer boundaries, as the persistent integrator does
    // for its next stage; the last group has no successor
    const float f = eval_rhs<kRows, kWR, true, kEq, false>(p, sm, a.batch, u, (float)a.t,
                                                           (float)a.t, res, fast_frc, nullptr,
                                                           nullptr, more, grp);
    if (ln.active) {
      // (x + c f with x = 0 when there is no base array: the same bits as c f)
      if (a.y_out != nullptr) a.y_out[ln.gidx] = base + a.c1 * f;
      if (a.acc_out != nullptr) a.acc_out[ln.gidx] = acc_in + a.c2 * f;
    }
#ifdef DDD_PROBES
    if (stamp != nullptr && tid == 0 && stamp_at < 4)
      stamp[stamp_at++] = __builtin_amdgcn_s_memrealtime();
#endif
    if (more) {
      group_barrier<kRows, kWR>();   // this group's epilogue has read sm.fk / sm.u
      ln = ln_next;
      u = u_next;
    }
  }
#ifdef DDD_PROBES
  if (stamp != nullptr && tid == 0) stamp[7] = __builtin_amdgcn_s_memrealtime();
#endif
}

template <int kRows, int kWR, int kEq, class TW = DefaultTower>
__global__ __launch_bounds__(kRows / kWR * 64, 2) void substep_multi_kernel(DevParams p,
                                                                            SubstepArgs a,
                                                                            int groups) {
  __shared__ Shared<kRows, kWR, false, TW> sm;
  substep_walk<kRows, kWR, kEq, TW>(p, a, sm, groups, (int)blockIdx.x, (int)gridDim.x);
}

// (Several independent one-wave groups per workgroup were measured slower, twice.
// Four per 256-thread workgroup with a run-time LDS base: 35.1 vs 33.0 us per
// substep at B = 4096.  Two per 128-thread workgroup with statically addressed LDS
// blocks (two inlined copies of the walk): the dispatcher then puts BOTH wavefronts
// of a launch on one SIMD for half the SIMDs, and two wavefronts in the same
// phases of the same launch take 23 us for their first evaluation -- 64.8 % against
// 71.6 %.  One-wave workgroups of two chains (whatever their start offset) land exactly one
// wavefront of each chain on every SIMD.  profiles/r3_ablation.txt.)

// Kernel 1c: ALL stages of one Runge-Kutta step in one launch, for callers that
// do not need the substeps (DDD_LAUNCH_PER_STEP): the same walk over row groups
// with the stage loop of the persistent integrator inside -- the state crosses
// HBM once per step and the launch boundary is paid once per step.
template <int kRows, int kWR, int kEq, class TW = DefaultTower>
__device__ __forceinline__ void step_walk(const DevParams& p, const StepArgs& a,
                                          Shared<kRows, kWR, false, TW>& sm, int groups, int first,
                                          int stride) {
  const int tid = group_tid<kRows, kWR>();
  Lane ln = make_lane<kRows, kWR>(p, a.batch, tid, first);
  Resident res;
  float u = ln.valid ? a.y_in[ln.gidx] : 0.0f;
  const bool fast_frc = forcing_is_fast<kRows, kWR>(p);
  const SampleSetup s_first = fetch_samples<kRows, kWR>(p, first, a.batch, fast_frc);
  setup_weights<kRows, kWR, true>(p, sm, ln, res);
  apply_samples<kRows, kWR>(sm, res, s_first);
  const StagePick<float> ah(a.sc.ah), bh(a.sc.bh);
  const StagePick<double> ct(a.sc.ct);
  const float t_first = (float)(a.t + ct.at(0));
  if (fast_frc) res.fk_next = forcing_sums<kRows, kWR, true>(p, sm, res, t_first, tid);
  for (int grp = first; grp < groups; grp += stride) {
    const int nxt = grp + stride;
    const bool more = nxt < groups;
    Lane ln_next = ln;
    float u_next = 0.0f;
    SampleSetup s_next{0.0f, 0.0f, 0.0f, 0, 0, 0};
    if (more) {   // in flight during this group's stages
      ln_next = make_lane<kRows, kWR>(p, a.batch, tid, nxt);
      u_next = ln_next.valid ? a.y_in[ln_next.gidx] : 0.0f;
      s_next = fetch_samples<kRows, kWR>(p, nxt, a.batch, fast_frc);
    }
    const float y = u;
    float ynew = y, kprev = 0.0f;
    for (int s = 0; s < a.tab.stages; ++s) {
      const float us = s > 0 ? y + kprev * ah.at(s) : y;
      const bool last = s + 1 == a.tab.stages;
      // sums prepared inside this evaluation: this group's next stage, or -- in
      // its last stage -- the NEXT group's first stage (other samples, same step)
      if (last && more) apply_samples<kRows, kWR, false>(sm, res, s_next);
      const float tn = last ? t_first : (float)(a.t + ct.at(s + 1));
      const float f = eval_rhs<kRows, kWR, true, kEq, false>(
          p, sm, a.batch, us, (float)(a.t + ct.at(s)), tn, res, fast_frc, nullptr,
          nullptr, !last || more, grp);
      if ((a.sc.b_nonzero >> s) & 1) ynew = ynew + bh.at(s) * f;
      kprev = f;
    }
    if (ln.active) a.y_out[ln.gidx] = ynew;
    if (more) {
      group_barrier<kRows, kWR>();
      ln = ln_next;
      u = u_next;
    }
  }
}

template <int kRows, int kWR, int kEq, class TW = DefaultTower>
__global__ __launch_bounds__(kRows / kWR * 64, 2) void step_multi_kernel(DevParams p, StepArgs a,
                                                                         int groups) {
  __shared__ Shared<kRows, kWR, false, TW> sm;
  step_walk<kRows, kWR, kEq, TW>(p, a, sm, groups, (int)blockIdx.x, (int)gridDim.x);
}

// ---------------------------------------------------------------------------
// Kernel 2: persistent integrator.  The whole time loop runs inside one launch;
// each lane keeps its grid point's state in registers, HBM sees y0 once and the
// requested snapshots.
// ---------------------------------------------------------------------------
// kTrace (libddd1d_probe.so only): s_memtime phase stamps (debug option
// "trace_ptr", profiles/tools/trace_phases.py), compiled into the
// run-time-parameterised instantiation and into one dedicated specialised
// instantiation: their branches cost ~2 % otherwise.
#ifdef DDD_PROBES
constexpr bool kTraceByDefault = true;
#else
constexpr bool kTraceByDefault = false;
#endif
template <int kRows, int kWR, typename ST, bool kHoist, int kEq = -1,
          bool kTrace = (kEq < 0) && kTraceByDefault, bool kWide = false, class TW = DefaultTower>
__global__ __launch_bounds__(kRows / kWR * 64, (min_waves<kRows, kWR, TW>())) void integrate_kernel(
    DevParams p, IntegrateArgs a) {
  __shared__ Shared<kRows, kWR, kWide, TW> sm;
  const Lane ln = make_lane<kRows, kWR>(p, a.batch, (int)threadIdx.x, (int)blockIdx.x);
  Resident res;
  // forcing batches (per-equation kernels: registers to spare; see forcing_batches)
  const int nb = (kEq >= 0 && kHoist) ? forcing_batches<kRows, kWR>(p) : 1;
  const bool fast_frc = launch_setup<kRows, kWR, kHoist>(p, sm, ln, a.batch, res, nb);
  // (Two wavefronts share each SIMD and run the same phases.  Static priorities
  // by hardware wave slot and start staggering were measured and change nothing
  // (profiles/r2_ablation.txt); the switches survive in the probe build only.)
  int ablate = 0;
  unsigned long long* trace_base = nullptr;
#ifdef DDD_PROBES
  // the ablate mask (debug option "ablate") is honoured by the run-time-parameterised
  // instantiation only (debug option "no_spec" selects it for the default models)
  ablate = kEq >= 0 ? 0 : (a.ablate & 0xff);
  if (kEq < 0 && (a.ablate >> 8) != 0 &&
      ((__builtin_amdgcn_s_getreg((3 << 11) | 4) & 1u) != 0))
    ablate = (a.ablate >> 8) & 0xff;   // high byte: mask for odd wave slots
  if (a.prio_split) {   // A/B experiments (debug options "prio_split" / "stagger")
    const unsigned wave_slot = __builtin_amdgcn_s_getreg((3 << 11) | 4) & 0xfu;   // HW_ID.wave_id
    const bool odd = (a.prio_split & 2) ? ((blockIdx.x >> 10) & 1u) != 0 : (wave_slot & 1u) != 0;
    if (odd) {
      if (a.prio_split & 4) __builtin_amdgcn_s_setprio(3);
      for (int i = 0; i < a.stagger; ++i) __builtin_amdgcn_s_sleep(127);
    }
  }
  trace_base = a.trace;
#endif
  // traced instantiation: shader-clock ticks (s_memtime) against the constant
  // 100 MHz counter (s_memrealtime) over the whole launch -> effective clock
  const unsigned long long clk0 = kTrace ? __builtin_amdgcn_s_memtime() : 0ull;
  const unsigned long long real0 = kTrace ? __builtin_amdgcn_s_memrealtime() : 0ull;
  const ST* y0 = static_cast<const ST*>(a.y0);
  ST* y_out = static_cast<ST*>(a.y_out);
  ST y = ln.valid ? y0[ln.gidx] : (ST)0;   // both half-waves carry the state
  const ST h = (ST)a.dt;
  const size_t snap_stride = (size_t)a.batch * p.N;
  int until_save = a.save_every;
  size_t snap = 0;
  int evals = 0;
  // time of the evaluation `ahead` evaluations after stage s of step `step` (ahead >= 0),
  // in the arithmetic of the loop below; this lane's batch of forcing phase 1
  const int frc_b = nb > 1 ? row_sample((int)threadIdx.x, p.inv_P) : 0;
  // (the stage time by an indexed read: one pass in nb evaluations)
  const auto time_ahead = [&](int step, int s, int ahead) {
    for (int i = 0; i < ahead; ++i) { if (++s == a.tab.stages) { s = 0; ++step; } }
    return (float)((a.t0 + (double)step * a.dt) + a.sc.ct[s]);
  };
  if (fast_frc && !(ablate & 1)) {
    float t_lane = (float)(a.t0 + a.tab.c[0] * a.dt);
    if (nb > 1) {   // batch b: evaluation b of the launch
      const float t1 = time_ahead(0, 0, 1), t2 = time_ahead(0, 0, 2);
      t_lane = frc_b == 0 ? t_lane : frc_b == 1 ? t1 : t2;
    }
    res.fk_next = forcing_sums<kRows, kWR>(p, sm, res, t_lane, (int)threadIdx.x);
  }
  int frc_phase = 0;   // evaluation index mod nb: the set of Shared::fk this evaluation reads
  // (the per-stage products a[s] h, b[s] h, c[s] dt: StageConsts, formed on the host in this
  // arithmetic -- (ST)a[s] * (ST)dt etc. -- and picked by scalar selects)
  (void)h;
  constexpr bool kSgprStages = kEq >= 0;   // (per-equation kernels: SGPRs to spare)
  const StagePick<ST, kSgprStages> ah = stage_ah<ST, kSgprStages>(a.sc), bh = stage_bh<ST, kSgprStages>(a.sc);
  const StagePick<double, kSgprStages> ct(a.sc.ct);
  for (int step = 0; step < a.n_steps; ++step) {
    const double t = a.t0 + (double)step * a.dt;
    const double t_after = a.t0 + (double)(step + 1) * a.dt;
    ST ynew = y;
    float kprev = 0.0f;
    for (int s = 0; s < a.tab.stages; ++s) {
      ST us = y;
      if (s > 0) us = y + (ST)kprev * ah.at(s);
      unsigned long long* tr = nullptr;
      if (kTrace && trace_base != nullptr && evals * 5 + 5 <= kTraceSlots)
        tr = trace_base + (size_t)(int)blockIdx.x * kTraceSlots + evals * 5;
      ++evals;
      // time of the evaluation after this one (next stage, or stage 0 of the
      // next step): its forcing sums are prepared inside this evaluation
      const bool last = s + 1 == a.tab.stages;
      const double tn = (last ? t_after : t) + ct.at(last ? 0 : s + 1);
      float tn_lane = (float)tn;
      bool prepare = true;
      if (nb > 1) {
        // batches: this evaluation reads set frc_phase; the last one of a batch prepares
        // the next nb evaluations (lanes of batch b: the (b + 1)-th next evaluation)
        res.fk_off = frc_phase * (kTrigMax * 4);
        prepare = frc_phase == nb - 1;
        if (prepare) {
          const float t2 = time_ahead(step, s, 2), t3 = time_ahead(step, s, 3);
          tn_lane = frc_b == 0 ? tn_lane : frc_b == 1 ? t2 : t3;
        }
        frc_phase = prepare ? 0 : frc_phase + 1;
      }
      const float f = eval_rhs<kRows, kWR, kHoist, kEq, kTrace>(
          p, sm, a.batch, (float)us, (float)(t + ct.at(s)), tn_lane, res, fast_frc,
          nullptr, nullptr, prepare, -1, ablate, tr);
      if ((a.sc.b_nonzero >> s) & 1) ynew = ynew + bh.at(s) * (ST)f;
      kprev = f;
    }
    y = ynew;
    if (--until_save == 0) {
      until_save = a.save_every;
      if (ln.active) y_out[snap * snap_stride + ln.gidx] = y;
      ++snap;
    }
  }
  if (kTrace && trace_base != nullptr && threadIdx.x == 0) {
    unsigned long long* tr = trace_base + (size_t)blockIdx.x * kTraceSlots;
    tr[kTraceSlots - 2] = __builtin_amdgcn_s_memtime() - clk0;
    tr[kTraceSlots - 1] = __builtin_amdgcn_s_memrealtime() - real0;
  }
}

}  // namespace mfma
}  // namespace ddd
