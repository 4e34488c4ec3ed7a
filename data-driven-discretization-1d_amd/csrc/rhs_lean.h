// The VALU-bound members of the model family on a kernel of their own: fixed stencils
// (ddd_baseline_create: model.baseline_space_derivatives, model.py:59-112;
// integrate.PolynomialDifferentiator, integrate.py:74-105) and one-layer nets
// (num_layers = 1: coefficients affine in the K neighbouring values,
// DevParams::linear_taps; model.predict_coefficients, model.py:420-513 without a hidden
// layer) in the PERSISTENT launch mode.
//
// Rounds 1-4 ran them on the MFMA kernels with the conv tower skipped: 13.8 k shader
// cycles per evaluation for ~110 FMAs per grid point -- the MFMA kernel's skeleton
// (242 VGPRs = two wavefronts per SIMD, LDS round trips sized for the tower, forcing
// sums placed to hide under MFMAs that are not there) is what was being timed.  Here:
//   * lane == grid point, one wavefront == 64 / N whole samples (N | 64), one
//     wavefront per workgroup: no barrier anywhere, wavefronts free-run;
//   * the state lives in registers for the whole time loop; the stencil / tap window is
//     read from a per-wavefront LDS row with a 4-point periodic halo on either side, so
//     every window read is base + immediate offset (no index math);
//   * the affine map's K x D x G coefficients and the D x G bias sit in registers
//     (uniform values), accumulated as packed FMAs on stencil-column pairs;
//   * forcing(t) (Burgers, equations.py:214-219, 276-277) as harmonic sums: the
//     sin / cos of ALL stages of a step are evaluated in one pass over
//     (stage, sample, mode) lanes wherever those fit the wavefront -- once per step
//     instead of once per evaluation;
//   * <= 128 VGPRs (four or more wavefronts per SIMD), no scratch.
// The arithmetic (order of every fma chain, the u / std shortcut, the forcing sums in
// mode order) is that of rhs_mfma.h::eval_rhs and rhs_stream.h, so the three launch
// modes of a model stay bit-identical (tests/test_gpu_integrate.py,
// tests/test_gpu_rhs.py::test_one_layer_nets_on_the_valu_route).
#pragma once
#include "dev_params.h"
#include "rhs_mfma.h"   // StagePick / pick4, sincos_branchless (dev_params.h), kTrigMax

namespace ddd {
namespace lean {

constexpr int kHalo = 4;                 // stencil reach: offsets -4 .. +3 (G <= 8), taps -3 .. +3
constexpr int kRowMax = 64 + 8 * 2 * kHalo;   // 64 points + a halo per sample, N >= 8
constexpr int kSlots = 4;                // stages whose harmonic sums can be staged at once
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct Shared {
  float u[kRowMax];                  // state of the wavefront's samples, halo'd rows of N + 8
  float un[kRowMax];                 // u / standard_deviation (one-layer nets)
  float2 pm[kSlots * 64 + 8];        // per (stage slot, sample, mode): a sin(psi), a cos(psi)
  float fk[kSlots * 8 * mfma::kTrigMax];   // per (stage slot, sample): 12 harmonic sums
  float trig[64 * mfma::kTrigMax];   // per lane: cos / sin of its grid point's spatial phases
};
static_assert(sizeof(Shared) <= 8 * 1024, "LDS per wavefront: 20 wavefronts per CU fit 160 KB");

// stencil-column pairs carried (kGP) for a stencil of G points
inline int column_pairs(int G) { return G <= 6 ? 3 : 4; }

// Models this kernel carries (checked on the host: capi.hip launch_integrate).
inline bool supports(const DevParams& p) {
  if (!(p.fixed || p.linear_taps > 0) || p.weno) return false;
  if (p.N < 8 || p.N > 64 || 64 % p.N != 0) return false;
  if (p.G < 1 || p.G > kGMax || p.G > p.N || p.D < 1 || p.D > kMaxDerivs) return false;
  // the kernel reads 2 column_pairs(G) window entries at offsets g - G / 2: they must stay
  // inside the 4-point halo (G = 1: offsets 0 .. 5 would leave the sample's row -- 0 x the
  // neighbour's halo, or never-written LDS, is NaN if that happens to hold one)
  if (2 * column_pairs(p.G) - 1 - (p.G >> 1) > kHalo || (p.G >> 1) > kHalo) return false;
  if (p.linear_taps > 0 && (p.linear_taps > 7 || p.D > 3 || p.target != TARGET_COEFFICIENTS))
    return false;
  if (p.fixed && p.target != TARGET_COEFFICIENTS) return false;
  if (p.forced) {   // harmonic-sum forcing only (the MFMA path's `fast` test)
    const int spw = 64 / p.N;
    if (spw * p.P > 64 || p.P >= 256 || p.n_k > 6 || p.n_k < 1 || spw * p.n_k * 2 > 64) return false;
  }
  return true;
}

// Wavefronts per SIMD an instantiation is compiled for: the model's registers
// (2 kGP kD (kK + 1)) leave room for four (<= 128 VGPRs: an ensemble of 4 096 N = 64 samples
// is four wavefronts per SIMD, all resident at once) up to 5 taps x 2 derivatives x 6 columns.
constexpr int model_registers(int k, int d, int gp) { return 2 * gp * d * (k + 1); }
constexpr int waves_per_simd(int k, int d, int gp) {
  return model_registers(k, d, gp) <= 72 ? 4 : model_registers(k, d, gp) <= 112 ? 3 : 2;
}

// kK: taps of the affine map (0: fixed stencils), kD: derivatives, kGP: stencil-column pairs.
template <int kK, int kD, int kGP>
__global__ __launch_bounds__(64, (waves_per_simd(kK, kD, kGP))) void integrate_kernel(DevParams p,
                                                                                    IntegrateArgs a) {
  __shared__ Shared sm;
  const int lane = (int)threadIdx.x;
  const int n = p.N;                                    // power of two, divides 64
  const int shift = 31 - __builtin_clz(n);
  const int spw = 64 >> shift;                          // samples per wavefront
  const int sl = lane >> shift, pos = lane & (n - 1);
  const long sample = (long)blockIdx.x * spw + sl;
  const bool valid = sample < a.batch;
  const long gidx = sample * n + pos;
  // LDS row of this lane's sample: [4 halo][N points][4 halo]; window entry j (offset
  // j - 4 from the point) is win[j]
  const int center = sl * (n + 2 * kHalo) + pos + kHalo;
  const float* win_u = sm.u + center - kHalo;
  const float* win_un = sm.un + center - kHalo;
  const bool halo_left = pos >= n - kHalo;              // this point is also a left-halo entry
  const bool halo_right = pos < kHalo;
  const int gl = p.G >> 1;                              // patches[i] = u[(x + i - G/2) mod N]
  const int tap_left = kK >> 1;                         // taps k - K/2 (layers.pad_periodic, center)
  const int next_lane = (lane & ~(n - 1)) | ((pos + 1) & (n - 1));   // right neighbour (flux forms)

  // ---- the model in registers -------------------------------------------------------
  // bias[d] (accuracy-layer bias, or the fixed stencil) and M[k][d] (one-layer nets),
  // zero-padded to 2 kGP columns (DevParams::bias8 / ns8 are)
  // (in VECTOR registers, through an empty asm: as the wave-uniform kernel arguments they
  // are, the compiler keeps them in SGPRs, runs out at ~100 and reads the spilled ones
  // back with one v_readlane per use -- VALU instructions, the resource this kernel is
  // bound by)
  const auto vreg = [](float x) { asm("" : "+v"(x)); return x; };
  f32x2 bias2[kD][kGP];
  f32x2 m2[kK > 0 ? kK : 1][kD][kGP];
#pragma unroll
  for (int d = 0; d < kD; ++d)
#pragma unroll
    for (int q = 0; q < kGP; ++q) {
      // (the D x G bias stays in SGPRs -- one scalar pair operand of the v_pk_add below --:
      // twelve VGPRs the 5-tap x 2-derivative instantiation needs to stay spill-free at 128)
      bias2[d][q] = f32x2{p.bias8[d][2 * q], p.bias8[d][2 * q + 1]};
#pragma unroll
      for (int k = 0; k < kK; ++k)
        m2[k][d][q] = f32x2{vreg(p.ns8[k * kD + d][2 * q]), vreg(p.ns8[k * kD + d][2 * q + 1])};
    }

  // ---- forcing: this lane's (stage slot, sample, mode) pair and its harmonic-sum slot ----
  const bool forced = p.forced != 0;
  const int stages = a.tab.stages;
  int slots = 1;                                         // stages evaluated per sin / cos pass
  float frc_a = 0.0f, frc_omega = 0.0f, frc_phi = 0.0f;
  int my_slot = 0;                                       // stage slot of this lane's pair
  bool has_pair = false, has_sum = false;
  int sum_run = 0, sum_cnt = 0, sum_out = 0, sum_slot = 0;
  // this lane's cos / sin table row: in LDS (12 registers fewer: the one-layer kernels
  // must stay within 128 VGPRs = four wavefronts per SIMD)
  float4* my_trig = reinterpret_cast<float4*>(sm.trig + lane * mfma::kTrigMax);
  if (forced) {   // wave-uniform
    const int pairs = spw * p.P, sums = spw * p.n_k * 2;
    while (slots < stages && slots < kSlots && (slots + 1) * pairs <= 64 && (slots + 1) * sums <= 64)
      ++slots;
    my_slot = lane / pairs;
    const int pair = lane - my_slot * pairs;
    const int psl = pair / p.P, mode = pair - psl * p.P;
    const long psample = (long)blockIdx.x * spw + psl;
    has_pair = my_slot < slots && psample < a.batch;
    const float4 row = p.frc[has_pair ? psample * p.P + mode : 0];
    frc_a = row.x; frc_omega = row.y; frc_phi = row.z;
    // harmonic sum (slot, sample, k, sin | cos): the modes with wavenumber index k are a
    // contiguous run (ddd_set_forcing sorts by k): runs[sample][kk] = first such mode
    sum_slot = lane / sums;
    const int sidx = lane - sum_slot * sums;
    const int ssl = (sidx >> 1) / p.n_k, kk = (sidx >> 1) - ssl * p.n_k, sc = sidx & 1;
    const long ssample = (long)blockIdx.x * spw + ssl;
    has_sum = sum_slot < slots && ssample < a.batch;
    const unsigned char* rr = p.runs + (has_sum ? ssample * 8 + kk : 0);
    const int m0 = rr[0], m1 = rr[1];
    sum_run = has_sum ? 2 * (sum_slot * pairs + ssl * p.P + m0) + sc : 0;   // float index into pm ([slot][pair])
    sum_cnt = has_sum ? m1 - m0 : 0;
    sum_out = (sum_slot * 8 + ssl) * mfma::kTrigMax + 2 * kk + sc;
    const float4* tr = reinterpret_cast<const float4*>(p.trig) + (size_t)pos * (mfma::kTrigMax / 4);
#pragma unroll
    for (int i = 0; i < mfma::kTrigMax / 4; ++i) my_trig[i] = tr[i];
    for (int i = lane; i < kSlots * 64 + 8; i += 64) sm.pm[i] = make_float2(0.0f, 0.0f);
    for (int i = lane; i < kSlots * 8 * mfma::kTrigMax; i += 64) sm.fk[i] = 0.0f;
  }
  const float* my_fk = sm.fk + sl * mfma::kTrigMax;

  const float* y0 = static_cast<const float*>(a.y0);
  float* y_out = static_cast<float*>(a.y_out);
  float y = valid ? y0[gidx] : 0.0f;
  const size_t snap_stride = (size_t)a.batch * n;
  int until_save = a.save_every;
  size_t snap = 0;
  const mfma::StagePick<float> ah(a.sc.ah), bh(a.sc.bh);
  const int eqn = p.equation;
  const bool flux_form = p.conservative != 0;

  for (int step = 0; step < a.n_steps; ++step) {
    const double t = a.t0 + (double)step * a.dt;
    float ynew = y, kprev = 0.0f;
    for (int s = 0; s < stages; ++s) {
      const float u = s > 0 ? y + kprev * ah.at(s) : y;
      // ---- the window: state (and u / std) into the halo'd LDS row ----
      sm.u[center] = u;
      if (halo_left) sm.u[center - n] = u;
      if (halo_right) sm.u[center + n] = u;
      if (kK > 0) {
        // model.py:450-451: net = u / std, as rhs_mfma.h::eval_rhs forms it (three
        // FMA-class instructions, exact for this std -- or the true division)
        const float q_un = u * p.inv_stddev;
        float un = fmaf(fmaf(-q_un, p.stddev, u), p.inv_stddev, q_un);
        if (p.exact_div) { asm volatile("; exact division"); un = u / p.stddev; }
        sm.un[center] = un;
        if (halo_left) sm.un[center - n] = un;
        if (halo_right) sm.un[center + n] = un;
      }
      // ---- forcing: harmonic sums of the next `slots` stages, one pass ----
      if (forced && s % slots == 0) {   // wave-uniform
        if (has_pair) {
          // the stage this lane's pair belongs to, and its time (float32(t + c dt): the
          // TF placeholder's value, integrate.py:57-60)
          const int st = s + my_slot;
          double ct = a.sc.ct[0];
          ct = st == 1 ? a.sc.ct[1] : ct;
          ct = st == 2 ? a.sc.ct[2] : ct;
          ct = st == 3 ? a.sc.ct[3] : ct;
          const float ts = (float)(t + ct);
          float sn, cs;
          sincos_branchless(frc_omega * ts + frc_phi, &sn, &cs);
          sm.pm[lane] = make_float2(frc_a * sn, frc_a * cs);   // (lane = slot * pairs + pair)
        }
      }
      // (one wavefront per workgroup: its LDS operations execute in order; waiting for
      // the writes to land is all a barrier means here)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (forced && s % slots == 0) {   // wave-uniform
        // per (slot, sample, k, sin | cos): the sum over the run of modes, in mode order
        // (fma(v, 1, acc) = acc + v, fma(v, 0, acc) = acc for the finite staged values:
        // the bits of rhs_mfma.h::forcing_phase2)
        const float* pmf = reinterpret_cast<const float*>(sm.pm) + sum_run;
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = pmf[2 * i];
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < 8; ++i) acc = fmaf(v[i], i < sum_cnt ? 1.0f : 0.0f, acc);
        for (int m = 8; m < sum_cnt; ++m) acc = acc + pmf[2 * m];   // (runs longer than 8 modes: rare)
        if (has_sum) sm.fk[sum_out] = acc;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }

      // ---- coefficients: bias + sum_k M[k] (u / std)[x + k - K/2]  (fixed: the bias) ----
      f32x2 cf[kD][kGP];
#pragma unroll
      for (int d = 0; d < kD; ++d)
#pragma unroll
        for (int q = 0; q < kGP; ++q) cf[d][q] = f32x2{0.0f, 0.0f};
#pragma unroll
      for (int k = 0; k < kK; ++k) {
        const float unk = win_un[kHalo + k - tap_left];
        const f32x2 u2{unk, unk};
#pragma unroll
        for (int d = 0; d < kD; ++d)
#pragma unroll
          for (int q = 0; q < kGP; ++q) cf[d][q] = __builtin_elementwise_fma(u2, m2[k][d][q], cf[d][q]);
      }
#pragma unroll
      for (int d = 0; d < kD; ++d)
#pragma unroll
        for (int q = 0; q < kGP; ++q) cf[d][q] = bias2[d][q] + cf[d][q];
      // ---- stencil apply (model.py:536-548), one chain per derivative in stencil order ----
      float dv[kMaxDerivs];
#pragma unroll
      for (int d = 0; d < kMaxDerivs; ++d) dv[d] = 0.0f;
      float pch[2 * kGP];
#pragma unroll
      // (columns g >= G carry a zero coefficient; they read the grid point itself -- offset
      // 0 is inside every stencil -- so that 0 x NaN never marks a point the reference's
      // stencil does not touch: integrate.py:161-167 signals divergence by NaN rows)
      for (int g = 0; g < 2 * kGP; ++g) pch[g] = win_u[kHalo + (g < p.G ? g - gl : 0)];
#pragma unroll
      for (int d = 0; d < kD; ++d) {
        float acc = 0.0f;
#pragma unroll
        for (int g = 0; g < 2 * kGP; ++g) {
          acc = fmaf(cf[d][g >> 1][g & 1], pch[g], acc);
          asm("" : "+v"(acc));   // (keeps the SLP vectoriser off the chains: rhs_mfma.h)
        }
        dv[d] = acc;
      }
      // ---- equation of motion, staggered flux difference, forcing ----
      float r = equation_rhs_or_flux(eqn, u, dv, p.eta);
      if (flux_form) {
        const float fnext = __shfl(r, next_lane, 64);
        r = p.inv_dx * (fnext - r);      // equations.staggered_first_derivative
        r = -r;
      }
      if (forced) {
        const float4* fk4 = reinterpret_cast<const float4*>(my_fk + (s % slots) * 8 * mfma::kTrigMax);
        float total = 0.0f;
#pragma unroll
        for (int i = 0; i < mfma::kTrigMax / 4; ++i) {
          const float4 f = fk4[i], tg = my_trig[i];
          total = fmaf(f.x, tg.x, total);
          total = fmaf(f.y, tg.y, total);
          total = fmaf(f.z, tg.z, total);
          total = fmaf(f.w, tg.w, total);
        }
        r = r + total;
      }
      if ((a.sc.b_nonzero >> s) & 1) ynew = ynew + bh.at(s) * r;
      kprev = r;
      asm volatile("" ::: "memory");   // (LDS in order: this stage's window reads precede the next stage's writes)
    }
    y = ynew;
    if (--until_save == 0) {
      until_save = a.save_every;
      if (valid) y_out[snap * snap_stride + gidx] = y;
      ++snap;
    }
  }
}

}  // namespace lean
}  // namespace ddd
