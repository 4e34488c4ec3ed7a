// The fine-grid "exact" solver of the Godunov-flux equations on a kernel of its own:
// fifth-order WENO reconstructions of u_minus / u_plus (weno.py:43-123, rolled one cell
// as integrate.py:137-138 and model.py:82-88 do), fixed polynomial stencils for the
// remaining derivatives (model.baseline_space_derivatives, model.py:59-112), Godunov
// flux + staggered flux difference (equations.py:341-370, 460-478, 570-587), Burgers'
// forcing(t) (equations.py:214-219, 276-277) -- integrate.WENODifferentiator,
// integrate.py:124-140 -- as
//   * one fused launch per evaluation       (ddd_time_derivative, ddd_rk_substep, the views),
//   * a fixed-step persistent launch        (ddd_integrate_fixed / _f64),
//   * the batched integrate.odeint          (ddd_integrate_adaptive_f64: SciPy's RK23 of
//                                            rk23.h, one controller per sample),
// which is what scripts/create_exact_data.py:96-133 maps over seeds.
//
// Rounds 3-5 ran this on the generic kernel's skeleton (one 256-thread workgroup per
// sample, every array in LDS, a block barrier after every loop, 20 sinf per grid point
// and evaluation for the forcing): 4.8e9 grid-point-evaluations/s on a ~300-instruction
// right-hand side.  Here:
//   * ONE WAVEFRONT PER SAMPLE, lane l owns the kP = N / 64 consecutive grid points
//     kP l .. kP l + kP - 1 (N = 64, 128, 256, 512); state, stage derivatives and the
//     controller live in registers for the whole launch; no barrier inside the time loop
//     (a wavefront's LDS operations execute in order);
//   * the periodic window u[x - 4 .. x + kP + 3] of a lane comes from a per-wavefront LDS
//     row with a 4-point halo either side, as aligned 16-byte reads; the sliding WENO /
//     stencil windows of the lane's points are register indices;
//   * the smoothness indicators of a cell are shared by the left reconstruction of one
//     point and the right reconstruction of its neighbour (common subexpressions of the
//     unrolled code);
//   * forcing as harmonic sums (rhs_mfma.h: forcing_phase1 / 2): one sincos per (sample,
//     mode) and evaluation on P lanes, the per-wavenumber sums on 2 n_k lanes, 12 FMAs per
//     grid point against the point's cos / sin row, which the four wavefronts of a
//     workgroup share in LDS (stored lane-major so that the reads are conflict-free);
//   * the flux of the right neighbour's first point by one ds_bpermute.
// The three kernels share `eval` (every expression in the order of dev_params.h's
// weno_minus_plus / equation_rhs_or_flux, i.e. the reference's, in float32), so a
// trajectory of the adaptive kernel equals SciPy driving ddd_time_derivative one sample
// at a time (tests/test_gpu_exact_solvers.py: equal nfev, 1e-9).
// Everything else (N not a multiple of 64 or > 512, more than 64 forcing modes or 6
// wavenumbers) stays on rhs_generic.h.
#pragma once
#include "dev_params.h"
#include "rk23.h"

namespace ddd {
namespace weno {

constexpr int kWaves = 4;        // wavefronts (samples) per workgroup
constexpr int kMaxN = 512;
constexpr int kHalo = 4;         // stencil offsets -4 .. +3 (G <= 8), WENO -3 .. +2
constexpr int kTrig = 12;        // cos / sin of <= 6 wavenumbers per grid point (DevParams::trig)
constexpr int kMaxModes = 64;    // forcing modes per sample (one lane each)

// Models these kernels carry (host side: capi.hip).
inline bool supports(const DevParams& p) {
  if (!p.fixed || !p.weno) return false;
  if (p.N % 64 != 0 || p.N > kMaxN) return false;
  const int pp = p.N / 64;
  if (pp != 1 && pp != 2 && pp != 4 && pp != 8) return false;
  if (p.equation < EQ_BURGERS_GODUNOV || p.equation > EQ_KS_GODUNOV) return false;
  if (p.D != (p.equation == EQ_KS_GODUNOV ? 4 : 3) || p.G < 1 || p.G > kGMax) return false;
  if (!p.conservative) return false;
  if (p.forced && (p.P < 1 || p.P > kMaxModes || p.n_k < 1 || p.n_k > 6)) return false;
  return true;
}

struct WaveShared {
  float u[kHalo + kMaxN + kHalo];   // the sample's state with periodic halos
  float2 pm[kMaxModes + 8];         // per mode: a sin(psi), a cos(psi); 8 entries of zero padding
  float fk[kTrig];                  // the 12 harmonic sums of this evaluation
  float pad[4];
};
struct Shared {
  float4 trig[3 * kMaxN];           // [quad q][point slot j 64 + lane]: cos / sin of the spatial phases
  WaveShared w[kWaves];
};

// Per-lane constants of one launch.
struct Lane {
  int lane;
  // forcing (Burgers): this lane's mode (lanes < P) and its (wavenumber, sin | cos) slot
  float frc_a, frc_omega, frc_phi;
  bool has_mode, has_sum;
  int sum_first, sum_cnt;
  // the fixed stencils of derivatives 2, 3 re-indexed by OFFSET: cst[d - 2][i] multiplies
  // u[x + i - 4] (zero outside the stencil's G columns: fma(0, u, s) = s)
  float cst[kMaxDerivs - 2][2 * kHalo];
  bool lo_in, hi_in;   // offsets -4 / +3 belong to the (common, zero-padded) stencil
};

template <int kP>
__device__ __forceinline__ void stage_trig(const DevParams& p, Shared& sm) {
  // DevParams::trig is [N][12]; LDS slot of point x = kP l + j is j 64 + l
  if (!p.forced) return;
  const float4* __restrict__ src = reinterpret_cast<const float4*>(p.trig);
  for (int i = (int)threadIdx.x; i < 3 * p.N; i += 64 * kWaves) {
    const int x = i / 3, q = i - 3 * x;
    const int l = x / kP, j = x - l * kP;
    sm.trig[q * p.N + j * 64 + l] = src[i];
  }
}

__device__ __forceinline__ Lane make_lane(const DevParams& p, WaveShared& ws, long sample) {
  Lane ln;
  ln.lane = (int)threadIdx.x & 63;
  ln.frc_a = ln.frc_omega = ln.frc_phi = 0.0f;
  ln.has_mode = ln.has_sum = false;
  ln.sum_first = ln.sum_cnt = 0;
  const int gl0 = p.G >> 1;                      // patches[i] = u[(x + i - G/2) mod N]
  ln.lo_in = gl0 >= kHalo;                       // g = -4 + gl0 >= 0
  ln.hi_in = kHalo - 1 + gl0 < p.G;              // g = +3 + gl0 < G
#pragma unroll
  for (int d = 0; d < kMaxDerivs - 2; ++d)
#pragma unroll
    for (int i = 0; i < 2 * kHalo; ++i) {
      const int g = i - kHalo + gl0;
      // (in VECTOR registers: as the wave-uniform values they are they would sit in SGPRs
      // for the whole launch, and the adaptive kernel's controller needs those)
      float c = (g >= 0 && g < p.G && d + 2 < p.D) ? p.bias8[d + 2][g] : 0.0f;
      asm("" : "+v"(c));
      ln.cst[d][i] = c;
    }
  if (p.forced) {
    ln.has_mode = ln.lane < p.P;
    const float4 row = p.frc[(size_t)sample * p.P + (ln.has_mode ? ln.lane : 0)];
    ln.frc_a = row.x; ln.frc_omega = row.y; ln.frc_phi = row.z;
    // harmonic sum (k, sin | cos) on lane 2 k + sc: the modes with wavenumber index k are a
    // contiguous run (ddd_set_forcing sorts by k): runs[sample][k] = first such mode
    const int kk = ln.lane >> 1;
    ln.has_sum = ln.lane < 2 * p.n_k;
    const unsigned char* rr = p.runs + (size_t)sample * 8 + (ln.has_sum ? kk : 0);
    const int m0 = rr[0], m1 = rr[1];
    ln.sum_first = ln.has_sum ? 2 * m0 + (ln.lane & 1) : 0;   // float index into pm
    ln.sum_cnt = ln.has_sum ? m1 - m0 : 0;
    for (int i = ln.lane; i < kMaxModes + 8; i += 64) ws.pm[i] = make_float2(0.0f, 0.0f);
    if (ln.lane < kTrig) ws.fk[ln.lane] = 0.0f;
  }
  return ln;
}

// x / 3 and x / 6 as the reference's float32 division gives them (correctly rounded) in three
// FMA-class instructions instead of the ~10 of the IEEE sequence: q = RN(x r), r = RN(1 / d),
// one Newton step on the exact residual.  Checked exhaustively against x / d over all 2^23
// significands (the identity is invariant under scaling x by powers of two): 0 mismatches
// for d = 3 and d = 6.  (Inf becomes NaN, a subnormal quotient may differ in its last bit:
// neither occurs for the weights, which lie in [0, 1].)  Ten of the 22 divisions per point.
__device__ __forceinline__ float div3(float x) {
  const float r = (float)(1.0 / 3.0), q = x * r;
  return fmaf(fmaf(-q, 3.0f, x), r, q);
}
__device__ __forceinline__ float div6(float x) {
  const float r = (float)(1.0 / 6.0), q = x * r;
  return fmaf(fmaf(-q, 6.0f, x), r, q);
}

// weno_minus_plus of dev_params.h on a register window w[0..5] = u[pos - 3 .. pos + 2].
__device__ __forceinline__ void minus_plus6(const float (&w)[6], float* um, float* up) {
  float is[3], om[3];
  weno_indicators(w[0], w[1], w[2], w[3], w[4], is);
  weno_omega(is, 0.1f, 0.6f, 0.3f, om);
  {
    const float c0 = div3(om[0]);
    const float c1 = div6(-(7.0f * om[0] + om[1]));
    const float c2 = div6((11.0f * om[0] + 5.0f * om[1]) + 2.0f * om[2]);
    const float c3 = div6(2.0f * om[1] + 5.0f * om[2]);
    const float c4 = div6(-om[2]);
    *um = (((c0 * w[0] + c1 * w[1]) + c2 * w[2]) + c3 * w[3]) + c4 * w[4];
  }
  weno_indicators(w[1], w[2], w[3], w[4], w[5], is);
  weno_omega(is, 0.3f, 0.6f, 0.1f, om);
  {
    const float o2 = om[0], o1 = om[1], o0 = om[2];
    const float c0 = div6(-o2);
    const float c1 = div6(5.0f * o2 + 2.0f * o1);
    const float c2 = div6((2.0f * o2 + 5.0f * o1) + 11.0f * o0);
    const float c3 = div6(-(o1 + 7.0f * o0));
    const float c4 = div3(o0);
    *up = (((c0 * w[1] + c1 * w[2]) + c2 * w[3]) + c3 * w[4]) + c4 * w[5];
  }
}

// One evaluation of finalize_time_derivative(t, WENODifferentiator(u)) for this
// wavefront's sample: u[j] = the lane's kP points in, f[j] = their time derivatives out.
// derivs_out (ddd_space_derivatives): [N][D] of this sample, or null.
template <int kP, int kEq>
__device__ __forceinline__ void eval(const DevParams& p, Shared& sm, WaveShared& ws, const Lane& ln,
                                     const float (&u)[kP], float t, float (&f)[kP],
                                     float* derivs_out) {
  constexpr int kD = kEq == EQ_KS_GODUNOV ? 4 : 3;
  const int lane = ln.lane, n = p.N;
  // ---- the state into the halo'd LDS row (wrapped copies from the first / last lanes) ----
  float* row = ws.u + kHalo + kP * lane;
#pragma unroll
  for (int j = 0; j < kP; ++j) row[j] = u[j];
#pragma unroll
  for (int j = 0; j < kP; ++j) {
    if (j < kHalo && kP * lane + j < kHalo) row[n + j] = u[j];               // also the right halo
    if (kP - 1 - j < kHalo && kP * lane + j >= n - kHalo) row[j - n] = u[j];   // ... the left halo
  }
  // ---- forcing, phase 1: one (mode) per lane ----
  const bool forced = kEq == EQ_BURGERS_GODUNOV && p.forced != 0;
  if (forced && ln.has_mode) {
    float sn, cs;
    sincos_branchless(ln.frc_omega * t + ln.frc_phi, &sn, &cs);
    ws.pm[lane] = make_float2(ln.frc_a * sn, ln.frc_a * cs);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (one wavefront: LDS in order; writes landed)
  // ---- the window u[x0 - 4 .. x0 + kP + 3], x0 = kP lane ----
  float w[kP + 2 * kHalo];
  {
    const float* win = ws.u + kP * lane;
    if constexpr (kP % 4 == 0) {
#pragma unroll
      for (int q = 0; q < (kP + 2 * kHalo) / 4; ++q) {
        const float4 v = *reinterpret_cast<const float4*>(win + 4 * q);
        w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
      }
    } else if constexpr (kP == 2) {
#pragma unroll
      for (int q = 0; q < (kP + 2 * kHalo) / 2; ++q) {
        const float2 v = *reinterpret_cast<const float2*>(win + 2 * q);
        w[2 * q] = v.x; w[2 * q + 1] = v.y;
      }
    } else {
#pragma unroll
      for (int q = 0; q < kP + 2 * kHalo; ++q) w[q] = win[q];
    }
  }
  // ---- forcing, phase 2: per (wavenumber, sin | cos) the sum over its run of modes, in mode
  //      order (rhs_mfma.h: forcing_phase2) ----
  if (forced) {
    const float* pmf = reinterpret_cast<const float*>(ws.pm) + ln.sum_first;
    float acc = 0.0f;
    for (int m = 0; m < ln.sum_cnt; m += 8) {
      float v[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = pmf[2 * (m + i)];
#pragma unroll
      for (int i = 0; i < 8; ++i) acc = acc + (m + i < ln.sum_cnt ? v[i] : 0.0f);
    }
    if (ln.has_sum) ws.fk[lane] = acc;
  }
  // ---- per point: WENO reconstructions, fixed stencils, Godunov flux ----
  float flux[kP + 1];
#pragma unroll
  for (int j = 0; j < kP; ++j) {
    float dv[kMaxDerivs] = {0.0f, 0.0f, 0.0f, 0.0f};
    const float w6[6] = {w[j + 1], w[j + 2], w[j + 3], w[j + 4], w[j + 5], w[j + 6]};
    minus_plus6(w6, &dv[0], &dv[1]);
#pragma unroll
    for (int d = 2; d < kD; ++d) {
      // one chain in stencil order (rhs_generic.h); offsets outside the stencil carry a 0.
      // Offsets -3 .. +2 lie inside the WENO window (a NaN there marks this point in the
      // reference too); -4 and +3 may lie outside every stencil: they then read the point
      // itself, so that 0 x NaN never marks a point the reference does not
      float s = 0.0f;
#pragma unroll
      for (int i = 0; i < 2 * kHalo; ++i) {
        float wv = w[j + i];
        if (i == 0) wv = ln.lo_in ? wv : w[j + kHalo];
        if (i == 2 * kHalo - 1) wv = ln.hi_in ? wv : w[j + kHalo];
        s = fmaf(ln.cst[d - 2][i], wv, s);
      }
      dv[d] = s;
    }
    if (derivs_out != nullptr) {
#pragma unroll
      for (int d = 0; d < kD; ++d) derivs_out[(size_t)(kP * lane + j) * kD + d] = dv[d];
    }
    flux[j] = equation_rhs_or_flux(kEq, w[j + kHalo], dv, p.eta);
  }
  // the right neighbour's first flux (periodic: lane 63 -> lane 0)
  flux[kP] = __shfl(flux[0], (lane + 1) & 63, 64);
  float fk[kTrig];
  if (forced) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const float4* fk4 = reinterpret_cast<const float4*>(ws.fk);
#pragma unroll
    for (int i = 0; i < kTrig / 4; ++i) {
      const float4 v = fk4[i];
      fk[4 * i] = v.x; fk[4 * i + 1] = v.y; fk[4 * i + 2] = v.z; fk[4 * i + 3] = v.w;
    }
  }
#pragma unroll
  for (int j = 0; j < kP; ++j) {
    const float diff = p.inv_dx * (flux[j + 1] - flux[j]);   // equations.staggered_first_derivative
    float r = -diff;
    if (forced) {
      float total = 0.0f;
#pragma unroll
      for (int i = 0; i < kTrig / 4; ++i) {
        const float4 tg = sm.trig[i * n + j * 64 + lane];
        total = fmaf(fk[4 * i], tg.x, total);
        total = fmaf(fk[4 * i + 1], tg.y, total);
        total = fmaf(fk[4 * i + 2], tg.z, total);
        total = fmaf(fk[4 * i + 3], tg.w, total);
      }
      r = r + total;
    }
    f[j] = r;
  }
  asm volatile("" ::: "memory");   // (this evaluation's LDS reads precede the next one's writes)
}

// a double summed over the wavefront, bitwise identical on every lane (xor butterfly:
// both partners of a step form the same sum)
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  return v;
}
__device__ __forceinline__ double uniform(double v) {   // the value as a wave-uniform (scalar) one
  const long long b = __double_as_longlong(v);
  const int lo = __builtin_amdgcn_readfirstlane((int)b);
  const int hi = __builtin_amdgcn_readfirstlane((int)(b >> 32));
  return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}

// ---- one fused launch per evaluation: ddd_time_derivative / ddd_rk_substep / the views ----
template <int kP, int kEq>
__global__ __launch_bounds__(64 * kWaves) void substep_kernel(DevParams p, SubstepArgs a) {
  __shared__ Shared sm;
  stage_trig<kP>(p, sm);
  __syncthreads();
  const int wave = (int)threadIdx.x >> 6;
  const long sample = (long)blockIdx.x * kWaves + wave;
  if (sample >= a.batch) return;
  WaveShared& ws = sm.w[wave];
  const Lane ln = make_lane(p, ws, sample);
  constexpr int kD = kEq == EQ_KS_GODUNOV ? 4 : 3;
  const size_t off = (size_t)sample * p.N + (size_t)kP * ln.lane;
  float u[kP], f[kP];
#pragma unroll
  for (int j = 0; j < kP; ++j) u[j] = a.y_in[off + j];
  eval<kP, kEq>(p, sm, ws, ln, u, (float)a.t, f,
                a.derivs_out != nullptr ? a.derivs_out + (size_t)sample * p.N * kD : nullptr);
#pragma unroll
  for (int j = 0; j < kP; ++j) {
    if (a.y_out != nullptr) {
      const float cf = a.c1 * f[j];
      a.y_out[off + j] = a.y_base != nullptr ? a.y_base[off + j] + cf : cf;
    }
    if (a.acc_out != nullptr) {
      const float cf = a.c2 * f[j];
      a.acc_out[off + j] = a.acc_in != nullptr ? a.acc_in[off + j] + cf : cf;
    }
  }
}

// ---- fixed-step persistent launch (ddd_integrate_fixed / _f64): rhs_generic.h's
//      integrate_kernel statement by statement, state in registers ----
template <int kP, int kEq, typename ST>
__global__ __launch_bounds__(64 * kWaves) void integrate_kernel(DevParams p, IntegrateArgs a) {
  __shared__ Shared sm;
  stage_trig<kP>(p, sm);
  __syncthreads();
  const int wave = (int)threadIdx.x >> 6;
  const long sample = (long)blockIdx.x * kWaves + wave;
  if (sample >= a.batch) return;
  WaveShared& ws = sm.w[wave];
  const Lane ln = make_lane(p, ws, sample);
  const size_t off = (size_t)sample * p.N + (size_t)kP * ln.lane;
  const ST* y0 = static_cast<const ST*>(a.y0);
  ST* y_out = static_cast<ST*>(a.y_out);
  ST y[kP], ynew[kP];
  float kprev[kP], u[kP], f[kP];
#pragma unroll
  for (int j = 0; j < kP; ++j) { y[j] = y0[off + j]; kprev[j] = 0.0f; }
  const ST h = (ST)a.dt;
  const size_t snap_stride = (size_t)a.batch * p.N;
  size_t snap = 0;
  int until_save = a.save_every;
  for (int step = 0; step < a.n_steps; ++step) {
    const double t = a.t0 + (double)step * a.dt;
    for (int s = 0; s < a.tab.stages; ++s) {
#pragma unroll
      for (int j = 0; j < kP; ++j) {
        ST us = y[j];
        if (s == 0) ynew[j] = us;
        else us = us + (ST)kprev[j] * ((ST)a.tab.a[s] * h);
        u[j] = (float)us;
      }
      eval<kP, kEq>(p, sm, ws, ln, u, (float)(t + a.tab.c[s] * a.dt), f, nullptr);
#pragma unroll
      for (int j = 0; j < kP; ++j) {
        if (a.tab.b[s] != 0.0f) ynew[j] = ynew[j] + ((ST)a.tab.b[s] * h) * (ST)f[j];
        kprev[j] = f[j];
      }
    }
#pragma unroll
    for (int j = 0; j < kP; ++j) y[j] = ynew[j];
    if (--until_save == 0) {
      until_save = a.save_every;
#pragma unroll
      for (int j = 0; j < kP; ++j) y_out[snap * snap_stride + off + j] = ynew[j];
      ++snap;
    }
  }
}

// ---- integrate.odeint (integrate.py:143-169) for a batch: SciPy's RK23 (rk23.h), one
//      controller per sample = per wavefront, float64 state and controller, float32
//      right-hand side; rhs_generic.h's adaptive_kernel statement by statement ----
template <int kP, int kEq>
__global__ __launch_bounds__(64 * kWaves) void adaptive_kernel(DevParams p, AdaptiveArgs a) {
  __shared__ Shared sm;
  stage_trig<kP>(p, sm);
  __syncthreads();
  const int wave = (int)threadIdx.x >> 6;
  const long sample = (long)blockIdx.x * kWaves + wave;
  if (sample >= a.batch) return;
  WaveShared& ws = sm.w[wave];
  const Lane ln = make_lane(p, ws, sample);
  const int n = p.N;
  const size_t off = (size_t)sample * n + (size_t)kP * ln.lane;
  const size_t row_stride = (size_t)a.batch * n;
  double y[kP], y_new[kP];
  float k0[kP], k1[kP], k2[kP], u[kP], f[kP];
#pragma unroll
  for (int j = 0; j < kP; ++j) { y[j] = a.y0[off + j]; y_new[j] = y[j]; k0[j] = k1[j] = k2[j] = 0.0f; }

  const double t0 = a.times[0];
  const double t_bound = a.times[a.n_times - 1];
  const double interval = fabs(t_bound - t0);
  const double rtol = a.rtol, atol = a.atol, max_step = a.max_step;
  const double sqrt_n = sqrt((double)n);
  rk23::Control ctl;
  ctl.init(t0, true);
  double h0 = 0.0, d1 = 0.0;
  long long attempts = 0;
  int phase = 0;
  while (ctl.status == rk23::RUNNING) {   // uniform over the wavefront
    double tt;
    if (phase == 0) tt = ctl.t;
    else if (phase == 1) tt = ctl.t + h0;
    else if (phase == 2) tt = ctl.t + 0.5 * ctl.h;
    else if (phase == 3) tt = ctl.t + 0.75 * ctl.h;
    else tt = ctl.t + ctl.h;
    if (phase == 0) {
#pragma unroll
      for (int j = 0; j < kP; ++j) u[j] = (float)y[j];
    } else if (phase == 1) {
#pragma unroll
      for (int j = 0; j < kP; ++j) u[j] = (float)(y[j] + h0 * (double)k0[j]);
    } else if (phase == 2) {
#pragma unroll
      for (int j = 0; j < kP; ++j) u[j] = (float)rk23::stage2_input(y[j], k0[j], ctl.h);
    } else if (phase == 3) {
#pragma unroll
      for (int j = 0; j < kP; ++j) u[j] = (float)rk23::stage3_input(y[j], k0[j], k1[j], ctl.h);
    } else {
#pragma unroll
      for (int j = 0; j < kP; ++j) u[j] = (float)y_new[j];
    }
    eval<kP, kEq>(p, sm, ws, ln, u, (float)tt, f, nullptr);
    ++ctl.nfev;
    double part = 0.0;
    if (phase == 0) {
#pragma unroll
      for (int j = 0; j < kP; ++j) k0[j] = f[j];
      if (a.n_times == 1) {
#pragma unroll
        for (int j = 0; j < kP; ++j) a.y_out[off + j] = y[j];
        ctl.ti = 1;
        ctl.status = rk23::FINISHED;
      } else {
#pragma unroll
        for (int j = 0; j < kP; ++j) {
          const double q = y[j] / (atol + fabs(y[j]) * rtol);
          part += q * q;
        }
        const double d0 = sqrt(uniform(wave_sum(part))) / sqrt_n;
        part = 0.0;
#pragma unroll
        for (int j = 0; j < kP; ++j) {
          const double q = (double)f[j] / (atol + fabs(y[j]) * rtol);
          part += q * q;
        }
        d1 = sqrt(uniform(wave_sum(part))) / sqrt_n;
        h0 = rk23::Control::first_guess(d0, d1, interval);
      }
      phase = 1;
    } else if (phase == 1) {
#pragma unroll
      for (int j = 0; j < kP; ++j) {
        const double q = (double)(f[j] - k0[j]) / (atol + fabs(y[j]) * rtol);   // float32 difference
        part += q * q;
      }
      const double d2 = sqrt(uniform(wave_sum(part))) / sqrt_n / h0;
      ctl.initial_step(h0, d1, d2, interval, max_step);
      ctl.begin_step(max_step);
      ctl.begin_attempt(t_bound);
      phase = 2;
    } else if (phase == 2) {
#pragma unroll
      for (int j = 0; j < kP; ++j) k1[j] = f[j];
      phase = 3;
    } else if (phase == 3) {
#pragma unroll
      for (int j = 0; j < kP; ++j) {
        k2[j] = f[j];
        y_new[j] = rk23::new_state(y[j], k0[j], k1[j], k2[j], ctl.h);
      }
      phase = 4;
    } else {
#pragma unroll
      for (int j = 0; j < kP; ++j) {
        const double q = rk23::scaled_error(y[j], y_new[j], k0[j], k1[j], k2[j], f[j], ctl.h,
                                            rtol, atol);
        part += q * q;
      }
      const double error_norm = sqrt(uniform(wave_sum(part))) / sqrt_n;
      if (ctl.error_test(error_norm)) {
        while (ctl.ti < a.n_times) {
          const double te = a.times[ctl.ti];
          if (!(te <= ctl.t_new)) break;
          const double x = (te - ctl.t) / ctl.h;
#pragma unroll
          for (int j = 0; j < kP; ++j)
            a.y_out[(size_t)ctl.ti * row_stride + off + j] =
                rk23::dense_output(y[j], k0[j], k1[j], k2[j], f[j], x, ctl.h);
          ++ctl.ti;
        }
#pragma unroll
        for (int j = 0; j < kP; ++j) { y[j] = y_new[j]; k0[j] = f[j]; }
        ctl.advance(t_bound, max_step);
      }
      ++attempts;
      if (ctl.status == rk23::RUNNING && attempts >= a.max_attempts)
        ctl.status = rk23::ATTEMPT_LIMIT;
      ctl.begin_attempt(t_bound);
      phase = 2;
    }
  }
  if (ctl.status != rk23::FINISHED) {
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    for (int r = ctl.ti; r < a.n_times; ++r)
#pragma unroll
      for (int j = 0; j < kP; ++j) a.y_out[(size_t)r * row_stride + off + j] = nan;
  }
  if (ln.lane == 0) {
    a.nfev[sample] = ctl.nfev;
    a.status[sample] = ctl.status;
  }
}

}  // namespace weno
}  // namespace ddd
