// Device-side parameter blocks shared by the kernels and the C-ABI layer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ddd {

constexpr int kMaxDerivs = 4;
constexpr int kMaxLayers = 8;
constexpr int kMaxStages = 4;
constexpr int kGMax = 8;      // widest stencil of the default MFMA kernels (and the stream kernel)
// "wide" run-time-parameterised MFMA kernels (rhs_mfma.h, kWide): stencils up
// to 12 points and up to 24 output channels -- coefficient_grid_min_size = 9 and
// polynomial_accuracy_order = 0 with three derivatives (training_test.py:56-57)
constexpr int kGWide = 12;
constexpr int kChMax = 16, kChWide = 24;
constexpr int kTraceSlots = 256;
constexpr int kWalkTraceRows = 2048;   // rows per launch of the substep-walk trace (probe library)
constexpr int kInMax = 8;     // widest per-derivative null space (G - rank)

// relu on the MFMA path = the VALU's [0, 1] output clamp on a PACKED add (v_pk_add_f32 x, 0
// clamp: two accumulator registers per instruction; gfx950 has no packed f32 max), on
// activations the host scaled by 2^-kReluShift: the input layer's weights and every
// bias row of the tower carry the factor, the output layer's weights carry its inverse
// (capi.hip: pack_mfma_weights).  Powers of two commute with every rounding of the fma
// chains, so the finite results are the bits of max(x, 0) for activations in
// [2^(-126 + kReluShift), 2^kReluShift] -- beyond 1.8e19 a state has diverged, below
// 2e-19 an activation contributes nothing float32 can see.  NaN -> 0 like v_max (DX10 clamp):
// rhs_mfma.h::eval_rhs re-creates the NaNs a propagating relu would have passed on (one
// v_cmp per evaluation; the rest only when a state holds a NaN).
// 64 relu instructions per wave-evaluation become 32 (profiles/r5_valu_census.txt).
#ifndef DDD_RELU_CLAMP
#define DDD_RELU_CLAMP 1   // A/B (profiles/r5_ablation.txt): 0 = one v_max_f32 per element, unscaled weights
#endif
constexpr int kReluShift = DDD_RELU_CLAMP ? 64 : 0;

// Equation ids: include/ddd1d.h enum ddd_equation.
enum : int {
  EQ_BURGERS = 0, EQ_BURGERS_CONS = 1, EQ_KDV = 2, EQ_KDV_CONS = 3,
  EQ_KS = 4, EQ_KS_CONS = 5, EQ_BURGERS_GODUNOV = 6, EQ_KDV_GODUNOV = 7,
  EQ_KS_GODUNOV = 8
};
enum : int { ACT_NONE = -1, ACT_RELU = 0, ACT_RELU6 = 1, ACT_TANH = 2,
             ACT_SOFTPLUS = 3, ACT_ELU = 4 };
enum : int { TARGET_COEFFICIENTS = 0, TARGET_SPACE_DERIVATIVES = 1,
             TARGET_TIME_DERIVATIVE = 2, TARGET_FLUX = 3 };

struct DevParams {
  // equation + grid
  int equation, N, D, G;
  float eta, stddev, inv_dx;
  float inv_stddev;    // RN(1 / stddev): seed of the three-instruction division in rhs_mfma.h
  int exact_div;       // 1: that shortcut is NOT bit-equal to u / stddev for this stddev
                       // (checked exhaustively on the host at model creation): divide
  int conservative;    // flux form: needs the staggered difference
  // model
  int fixed;           // 1: fixed stencils in `bias` ([D][G]); no conv net
  int weno;            // fixed only: derivatives 0 / 1 are WENO5 reconstructions
  // One-layer nets (num_layers = 1, a hyper-parameter training.create_hparams admits;
  // integrate_test.py:48 names it in `model_kwargs` but never passes it -- the reference's
  // tests train the default three-layer net): the coefficients are AFFINE in the K
  // neighbouring values,
  //   coeff[d][g] = B[d][g] + sum_k M[k][d][g] (u / std)[x + k - K/2],
  // with B / M folded on the host (float64) from the conv layer, the null space and the
  // accuracy bias.  linear_taps = K > 0: the MFMA-path kernels skip the tower like a fixed
  // model and add the M terms on the VALU; M[k][d] is row 4 + k D + d of the LDS table
  // (ns8), B is bias8.  The generic kernel ignores these fields and runs the true net.
  int linear_taps;
  int dpp_rol;         // 1: `v_mov_b32_dpp wave_rol:1` verified on this device (capi.hip)
  int target, L, F, K, act, C_out, pao, unbiased;
  int in_start[kMaxDerivs], in_size[kMaxDerivs], ns_off[kMaxDerivs];
  int w_off[kMaxLayers], b_off[kMaxLayers], cin[kMaxLayers], cout[kMaxLayers];
  const float* weights;    // natural layout, layer-major (kernel then bias)
  const float* weights4;   // the same with output channels zero-padded to fours: per layer
  int w4_off[kMaxLayers];  //   [K][cin][c4] then bias [c4] (the generic kernel's LDS image)
  const float* nullspace;  // per derivative [in_size][G]
  const float* bias;       // [D][G]  (accuracy-layer bias, or fixed stencils)
  // MFMA-path projection tables, carried in the kernel-argument segment so
  // that rows arrive as scalar (SGPR) operands: for output channel c of the
  // conv tower, ns8[c] is its null-space row zero-padded to 8 stencil columns
  // and dsel the derivative it feeds; bias8[d] likewise.
  // (sized for the wide kernels; the default ones use the first 16 rows / 8 columns)
  float ns8[kChWide][kGWide];
  float bias8[kMaxDerivs][kGWide];
  // dsel packed: 2 bits per channel (derivative index) + validity mask, so the
  // hot loop tests one scalar register instead of indexing an SGPR tuple.
  unsigned long long dsel_bits;
  unsigned dsel_valid;
  // folded = 1: w_final4_rt already contains the null-space projection
  // (W3 @ nullspace) -- or the net emits the coefficients themselves
  // (polynomial_accuracy_order 0) --, so its output channel G d + g IS
  // coefficient g of derivative d (D <= 2, G in 6..8; run-time-parameterised
  // kernels.  The specialised kernels know at compile time whether w_final4 is
  // folded: rhs_mfma.h spec_folded).
  int folded;
  // MFMA-packed weights: rows x 64 lanes, stored four rows to a float4 per lane with the
  // rows zero-padded to a multiple of four (rhs_mfma.h load_rows4)
  const float* w_input;    // input layer, 3 rows
  const float* w_hidden;   // hidden layers, (L-2) x 81 rows (each layer padded to 84)
  // output layer packed for the 4x4x1 broadcast MFMA (rhs_mfma.h: final_layer4), 41 rows:
  const float* w_final4;      // live channels renumbered contiguously, ceil(channels / 4) groups
  // the same layer for the split integrators (two 32-row wavefronts per sample, rhs_mfma.h
  // kSplit): chunk 0 = groups [0, ceil(NG / 2)), chunk 1 = the rest, each packed on its own
  // (q = k * groups_of_chunk + g) in 24 quad-stored rows; null: no split kernels for this model
  const float* w_final4_split;
  // four wavefronts per 64-row group (rhs_mfma.h kQuad): the A operands of the 16x16x4 layers,
  // [2 channel halves][2] input rows, [2][41] hidden rows, [41] output rows of 64 lanes
  // (lane l: W[out = l & 15][reduction slot l >> 4]); null: no such kernels for this model
  const float* w_quad;
  // run-time-parameterised kernels: the live channel groups packed two by two
  // (pair gp: fin4_regs(2) rows for groups 2 gp, 2 gp + 1; an odd last group:
  // fin4_regs(1) rows), channels in natural / G d + g (folded) numbering; plain [rows][64]
  const float* w_final4_rt;
  int rt_groups;              // live channel groups of w_final4_rt
  int fin4_groups;            // groups of w_final4
  // per-sample forcing
  int forced, P, n_k, forcing_batch;
  float inv_P, inv_nk;     // RN(1 / P), RN(1 / max(n_k, 1)): what the kernels' 1.0f / (float)P gave,
                           // formed once on the host instead of once per wavefront
  const float4* frc;       // [batch][P] = (amplitude, omega, phase, k_index)
  const float* sp;         // [n_k][N]   spatial phase table
  const float* trig;       // [N][12] cos / sin of the spatial phases, zero padded
  const unsigned char* runs;   // [batch][8]: first mode of each sample with k index >= kk
};

// Explicit RK tableau in "previous-stage only" form:
//   u_s = y + a[s] * h * k_{s-1},  t_s = t + c[s] * h,  y' = y + sum b[s] h k_s
struct Tableau {
  int stages;
  float a[kMaxStages];
  float b[kMaxStages];
  double c[kMaxStages];
};

// Per-stage constants of one fixed-step launch, formed ONCE on the host in the kernels' own
// arithmetic (capi.hip: make_stage_consts) and carried in the kernel-argument segment:
// indexed by the stage inside the time loop they were three scalar loads with their
// waits and four float / float64 multiplies per evaluation; here they are SGPRs picked
// by scalar selects.
struct StageConsts {
  float ah[kMaxStages], bh[kMaxStages];      // float state:   a[s] * (float)dt, b[s] * (float)dt
  double ahd[kMaxStages], bhd[kMaxStages];   // float64 state: (double)a[s] * dt, (double)b[s] * dt
  double ct[kMaxStages];                     // c[s] * dt
  int b_nonzero;                             // bit s: b[s] != 0 (stage s enters the update)
};

struct IntegrateArgs {
  double t0, dt;
  int n_steps, save_every;
  Tableau tab;
  StageConsts sc;
  const void* y0;   // [batch][N] StateT
  void* y_out;      // [n_saved][batch][N] StateT
  int batch;
#ifdef DDD_PROBES   // libddd1d_probe.so only (profiles/tools): never in the product library
  int prio_split;   // 1: odd hardware wave slots run at raised priority
  int ablate;       // debug option "ablate": bit mask of phases to skip (WRONG results)
  int stagger;      // debug option "stagger": initial s_sleep count for odd waves
  unsigned long long* trace;   // [blocks][kTraceSlots] s_memtime stamps
#endif
};

// ddd_integrate_adaptive_f64 (rhs_adaptive.h)
struct AdaptiveArgs {
  const double* times;   // [n_times] device, strictly increasing; times[0] = t0
  int n_times;
  double rtol, atol, max_step;
  const double* y0;      // [batch][N]
  double* y_out;         // [n_times][batch][N]; rows a failed sample did not reach: NaN
  int* nfev;             // [batch]
  int* status;           // [batch]: 0 finished, -1 step size too small, -2 attempt limit
  int batch;
  long long max_attempts;   // safety net against runaway samples (SciPy has none)
};

// One whole Runge-Kutta step in one launch (DDD_LAUNCH_PER_STEP; rhs_mfma.h: step_multi_kernel)
struct StepArgs {
  double t, dt;
  Tableau tab;
  StageConsts sc;
  const float* y_in;
  float* y_out;
  int batch;
};

struct SubstepArgs {
  double t;
  const float* y_in;
  const float* y_base;   // may be null
  float c1;
  float* y_out;          // may be null
  const float* acc_in;   // may be null
  float c2;
  float* acc_out;        // may be null
  float* derivs_out;     // [batch][N][D] or null
  float* coeffs_out;     // [batch][N][D][G] or null
  int batch;
#ifdef DDD_PROBES   // libddd1d_probe.so only: wave-lifetime stamps of the multi-group walk
  unsigned long long* trace;   // [launches][kWalkTraceRows][8] s_memrealtime (100 MHz) stamps
  int trace_row0;              // first row of this launch
#endif
};

__device__ __forceinline__ float apply_activation(float x, int act) {
  switch (act) {
    // (compare + select, not fmaxf: maxNum(NaN, 0) = 0 would stop a NaN that np.maximum /
    // Eigen's relu -- and therefore the reference -- pass on; divergence must reach the caller
    // as NaN at the grid points the reference marks, integrate.py:161-167)
    case ACT_RELU: return x < 0.0f ? 0.0f : x;
    case ACT_RELU6: return x < 0.0f ? 0.0f : (x > 6.0f ? 6.0f : x);
    case ACT_TANH: return tanhf(x);
    case ACT_SOFTPLUS: return (x > 0.0f ? x : 0.0f) + log1pf(expf(-fabsf(x)));
    case ACT_ELU: return x > 0.0f ? x : expm1f(x);
    default: return x;
  }
}

// Branch-free sin and cos of a float32 angle (|x| < ~1e5): three-term
// Cody-Waite reduction by pi/2 with FMAs, then the classic single-precision
// minimax polynomials on [-pi/4, pi/4].  Max error ~1e-7 absolute: below the
// float32 rounding of the phase itself (ulp(25)/2 = 1e-6).
__device__ __forceinline__ void sincos_branchless(float x, float* s, float* c) {
  const float j = rintf(x * 0.63661977236758134f);   // x * 2/pi
  float r = fmaf(-j, 1.5703125f, x);
  r = fmaf(-j, 4.837512969970703125e-4f, r);
  r = fmaf(-j, 7.54978995489188216e-8f, r);
  const float r2 = r * r;
  float ps = fmaf(r2, -1.9515295891e-4f, 8.3321608736e-3f);
  ps = fmaf(ps, r2, -1.6666654611e-1f);
  const float sr = fmaf(r * r2, ps, r);
  float pc = fmaf(r2, 2.443315711809948e-5f, -1.388731625493765e-3f);
  pc = fmaf(pc, r2, 4.166664568298827e-2f);
  const float cr = fmaf(r2 * r2, pc, fmaf(r2, -0.5f, 1.0f));
  const int q = (int)j;
  const float s0 = (q & 1) ? cr : sr;
  const float c0 = (q & 1) ? sr : cr;
  *s = (q & 2) ? -s0 : s0;
  *c = ((q + 1) & 2) ? -c0 : c0;
}

// Fifth-order upwind-biased WENO reconstructions at the LEFT cell edge of grid
// point `pos` (weno.py:43-123, rolled by one cell as integrate.py:137-138 and
// model.py:82-88 do): *um = reconstruct_left(u)[pos - 1],
// *up = reconstruct_right(u)[pos - 1].  `u` holds one periodic sample of n
// points.  Arithmetic follows the reference's expression order in float32.
__device__ __forceinline__ void weno_indicators(float m2, float m1, float c, float p1,
                                                float p2, float (&is)[3]) {
  // weno.calculate_smoothness_indicators, Equation (7) of Tang (2005)
  const float a0 = (m2 - 4.0f * m1) + 3.0f * c, b0 = (m2 - 2.0f * m1) + c;
  const float a1 = m1 - p1, b1 = (m1 - 2.0f * c) + p1;
  const float a2 = (3.0f * c - 4.0f * p1) + p2, b2 = (c - 2.0f * p1) + p2;
  const float q = 0.25f, r = (float)(13.0 / 12.0);
  is[0] = q * (a0 * a0) + r * (b0 * b0);
  is[1] = q * (a1 * a1) + r * (b1 * b1);
  is[2] = q * (a2 * a2) + r * (b2 * b2);
}

__device__ __forceinline__ void weno_omega(const float (&is)[3], float w0, float w1,
                                           float w2, float (&om)[3]) {
  // weno.calculate_omega: alpha = w / (eps + IS)^2, omega = alpha / sum(alpha)
  const float eps = 1e-6f;
  const float d0 = eps + is[0], d1 = eps + is[1], d2 = eps + is[2];
  const float al0 = w0 / (d0 * d0), al1 = w1 / (d1 * d1), al2 = w2 / (d2 * d2);
  const float total = (al0 + al1) + al2;
  om[0] = al0 / total; om[1] = al1 / total; om[2] = al2 / total;
}

__device__ __forceinline__ void weno_minus_plus(const float* u, int pos, int n,
                                                float* um, float* up) {
  float w[6];   // u[pos - 3 .. pos + 2]
#pragma unroll
  for (int j = 0; j < 6; ++j) {
    int q = pos - 3 + j;
    q = q < 0 ? q + n : q;
    q = q >= n ? q - n : q;
    w[j] = u[q];
  }
  float is[3], om[3];
  // reconstruct_left at cell pos - 1: window pos-3 .. pos+1
  weno_indicators(w[0], w[1], w[2], w[3], w[4], is);
  weno_omega(is, 0.1f, 0.6f, 0.3f, om);
  {
    const float c0 = om[0] / 3.0f;
    const float c1 = -(7.0f * om[0] + om[1]) / 6.0f;
    const float c2 = ((11.0f * om[0] + 5.0f * om[1]) + 2.0f * om[2]) / 6.0f;
    const float c3 = (2.0f * om[1] + 5.0f * om[2]) / 6.0f;
    const float c4 = -om[2] / 6.0f;
    *um = (((c0 * w[0] + c1 * w[1]) + c2 * w[2]) + c3 * w[3]) + c4 * w[4];
  }
  // reconstruct_right at cell pos - 1: indicators of cell pos (weights
  // reversed, omega rolled by -1), window pos-2 .. pos+2
  weno_indicators(w[1], w[2], w[3], w[4], w[5], is);
  weno_omega(is, 0.3f, 0.6f, 0.1f, om);
  {
    const float o2 = om[0], o1 = om[1], o0 = om[2];
    const float c0 = -o2 / 6.0f;
    const float c1 = (5.0f * o2 + 2.0f * o1) / 6.0f;
    const float c2 = ((2.0f * o2 + 5.0f * o1) + 11.0f * o0) / 6.0f;
    const float c3 = -(o1 + 7.0f * o0) / 6.0f;
    const float c4 = o0 / 3.0f;
    *up = (((c0 * w[1] + c1 * w[2]) + c2 * w[3]) + c3 * w[4]) + c4 * w[5];
  }
}

// Godunov flux for u^2/2 (equations.py:341-349).
__device__ __forceinline__ float godunov_flux(float um, float up) {
  const float a = um * um, b = up * up;
  return 0.5f * (um <= up ? fminf(a, b) : fmaxf(a, b));
}

// u_t for the non-flux forms, or the flux for the flux forms.  `d` holds the
// spatial derivatives in DERIVATIVE_NAMES order.  Arithmetic mirrors the
// expression order of the reference's equation_of_motion methods.
__device__ __forceinline__ float equation_rhs_or_flux(int eq, float y,
                                                      const float (&d)[kMaxDerivs],
                                                      float eta) {
  switch (eq) {
    case EQ_BURGERS: return eta * d[1] - y * d[0];
    case EQ_BURGERS_CONS: return 0.5f * (d[0] * d[0]) - eta * d[1];
    case EQ_BURGERS_GODUNOV: return godunov_flux(d[0], d[1]) - eta * d[2];
    case EQ_KDV: return (-6.0f * y) * d[0] - d[1];
    case EQ_KDV_CONS: return 3.0f * (d[0] * d[0]) + d[1];
    case EQ_KDV_GODUNOV: return 6.0f * godunov_flux(d[0], d[1]) + d[2];
    case EQ_KS: return (-y * d[0] - d[2]) - d[1];
    case EQ_KS_CONS: return (0.5f * (d[0] * d[0]) + d[2]) + d[1];
    case EQ_KS_GODUNOV: return (d[3] + d[2]) + godunov_flux(d[0], d[1]);
    default: return 0.0f;
  }
}

namespace ops {
// DPP wavefront rotate: out[l] = in[(l + 1) % 64] if `wave_rol:1` does what the
// flux exchange of the one-wave kernel assumes (checked on the device at first
// use: ops.h dpp_rotate_probe_kernel, capi.hip dpp_wave_rol_ok).
__device__ __forceinline__ float wave_rotate_left1(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x134 /* wave_rol:1 */,
                                                    0xf, 0xf, false));
}
}  // namespace ops

// Forcing of one sample at one grid point: sum_j a sin((omega t + sp) + phi),
// float32 in the TF graph's order (equations.py:214-219).
__device__ __forceinline__ float forcing_at(const DevParams& p, const float4* frc,
                                            int pos, float t) {
  float total = 0.0f;
  for (int m = 0; m < p.P; ++m) {
    const float4 q = frc[m];
    const int kidx = __float_as_int(q.w);
    const float sp = p.sp[kidx * p.N + pos];
    const float phase = __fadd_rn(__fadd_rn(__fmul_rn(q.y, t), sp), q.z);
    total = __fadd_rn(total, __fmul_rn(q.x, sinf(phase)));
  }
  return total;
}

}  // namespace ddd
