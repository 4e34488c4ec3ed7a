// Fused learned-stencil right-hand side + Runge-Kutta stepping on CDNA4 f32 MFMA.
//
// Work decomposition (DESIGN.md "MFMA kernel"):
//   * one workgroup = 256 "rows" (grid points) = floor(256 / N) whole samples,
//     4 wavefronts, wavefront w owns rows [64 w, 64 w + 64);
//   * the conv tower runs on the matrix cores as implicit GEMMs
//         D[out-channel][position] += W[out-channel][k] * h[k][position],
//     reduction index k = (tap, in-channel):
//       - input layer   1 -> 32, K=5 : v_mfma_f32_32x32x2_f32, 3 steps (5 taps + bias)
//       - hidden layers 32 -> 32, K=5: v_mfma_f32_32x32x2_f32, 80 steps + 1 bias step;
//         the layer's 160x32 weight panel lives in 81 VGPRs per lane
//       - output layer  32 -> C_out<=16: v_mfma_f32_16x16x4_f32, 40 steps + 1 bias step
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
#pragma once
#include "dev_params.h"

namespace ddd {
namespace mfma {

constexpr int kRows = 256;       // rows per workgroup
constexpr int kHS = 36;          // padded activation row stride (floats)
constexpr int kF = 32;           // hidden channels
constexpr int kKW = 5;           // conv taps
constexpr int kInSteps = 3;      // (5 taps + bias) / 2
constexpr int kHidSteps = 81;    // 5*32/2 MFMA steps + 1 bias step
constexpr int kFinSteps = 41;    // 5*32/4 MFMA steps + 1 bias step
constexpr int kFrcLds = 256;     // forcing entries (samples*modes) staged in LDS
constexpr int kTabRows = 4 + 16; // bias8 rows + nullspace8 rows

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct Shared {
  float hA[kRows * kHS];
  float hB[kRows * kHS];
  float u[kRows];
  float un[kRows];                // u / standard_deviation
  float flux[kRows];
  float4 frc[kFrcLds];
  float tab[kTabRows * kGMax];    // [0,4): bias8[d][8]; [4,20): nullspace8 rows
};
static_assert(sizeof(Shared) <= 80 * 1024, "two workgroups must fit one CU's LDS");

// Value the optimiser must treat as unknown: stops loop-invariant code motion
// from hoisting per-evaluation index math and loads out of the time loop (where
// they would pin dozens of VGPRs for the whole kernel).
__device__ __forceinline__ int opaque(int x) {
  asm volatile("" : "+v"(x));
  return x;
}

struct Lane {
  int row;       // row inside the workgroup this lane owns in VALU phases
  int base;      // first row of the row's sample
  int pos;       // grid index inside the sample
  int sl;        // sample index inside the workgroup
  int active;    // row maps to a real sample of the batch
  long gidx;     // sample * N + pos (global element index), valid if active
  int wave, lane;
  int rows_used; // samples_per_group * N
  float inv_n;
};

// Exact for rows < 256, N >= 8: (row + 0.5) / N is never within 2e-3 of an
// integer, far above float rounding.
__device__ __forceinline__ int row_sample(int row, float inv_n) {
  return (int)(((float)row + 0.5f) * inv_n);
}

__device__ __forceinline__ Lane make_lane(const DevParams& p, int batch, int tid) {
  Lane ln;
  ln.row = tid;
  ln.wave = tid >> 6;
  ln.lane = tid & 63;
  const int spg = kRows / p.N;
  ln.rows_used = spg * p.N;
  ln.inv_n = 1.0f / (float)p.N;
  ln.sl = row_sample(ln.row, ln.inv_n);
  ln.base = ln.sl * p.N;
  ln.pos = ln.row - ln.base;
  if (ln.row >= ln.rows_used) {
    // Spare rows (256 is not a multiple of N): read like row 0 of sample 0 so
    // every LDS index stays in range; results are never stored.
    ln.sl = 0; ln.base = 0; ln.pos = 0;
  }
  const long sample = (long)blockIdx.x * spg + ln.sl;
  ln.active = (ln.row < ln.rows_used) && (sample < batch);
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

__device__ __forceinline__ void load_hidden(const DevParams& p, int hidden_index,
                                            int lane, float (&w)[kHidSteps]) {
  const float* src = p.w_hidden + (size_t)hidden_index * kHidSteps * 64 + lane;
#pragma unroll
  for (int s = 0; s < kHidSteps; ++s) w[s] = src[s * 64];
}

#define DDD_MFMA32(A, B, C) __builtin_amdgcn_mfma_f32_32x32x2f32((A), (B), (C), 0, 0, 0)
#define DDD_MFMA16(A, B, C) __builtin_amdgcn_mfma_f32_16x16x4f32((A), (B), (C), 0, 0, 0)

__device__ __forceinline__ void activate16(f32x16& acc, int act) {
  if (act == ACT_RELU) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = fmaxf(acc[r], 0.0f);
  } else if (act == ACT_RELU6) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = fminf(fmaxf(acc[r], 0.0f), 6.0f);
  } else if (act == ACT_TANH) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = tanhf(acc[r]);
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

// Input layer 1 -> 32 for this wave's two 32-row tiles (3 MFMA steps each).
//   A: lane l supplies W1[out = l & 31][k = 2 s + (l >> 5)]  (k = tap; k = 5: bias)
//   B: lane l supplies un[(pos(l & 31) + k - 2) mod N]        (k = 5: 1.0)
__device__ __forceinline__ void input_layer(const DevParams& p, const Lane& ln,
                                            const float* __restrict__ un,
                                            float* __restrict__ out) {
  const int j = ln.lane & 31;
  const int half = ln.lane >> 5;
  float w[kInSteps];
#pragma unroll
  for (int s = 0; s < kInSteps; ++s) w[s] = p.w_input[s * 64 + ln.lane];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int trow = ln.wave * 64 + t * 32 + j;
    const float b0 = un[tile_src_row(ln, trow, half - 2, p.N)];        // taps 0 / 1
    const float b1 = un[tile_src_row(ln, trow, half, p.N)];            // taps 2 / 3
    const float b2 = half ? 1.0f : un[tile_src_row(ln, trow, 2, p.N)]; // tap 4 / bias
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    acc = DDD_MFMA32(w[0], b0, acc);
    acc = DDD_MFMA32(w[1], b1, acc);
    acc = DDD_MFMA32(w[2], b2, acc);
    activate16(acc, p.act);
    store_tile32(out, trow, half, acc);
  }
}

// One hidden layer for this wave's two 32-row tiles.
//   A operand (weights): lane l supplies W[out = l & 31][k = 2 s + (l >> 5)]
//   B operand (acts)   : lane l supplies h[k = 2 s + (l >> 5)][position = l & 31]
//   reduction index    : step s = 16 tap + jj, half = l >> 5  <->  (tap, cin = 16 half + jj)
// The 16 floats a lane needs per tap are four ds_read_b128; the read of group
// g + 1 is issued before the four MFMAs of group g (software prefetch).
__device__ __forceinline__ void hidden_layer(const DevParams& p, const Lane& ln,
                                             const float* __restrict__ in,
                                             float* __restrict__ out,
                                             const float (&w)[kHidSteps]) {
  const int j = ln.lane & 31;
  const int half = ln.lane >> 5;
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int trow = ln.wave * 64 + t * 32 + j;
    const float4* rowp[kKW];
#pragma unroll
    for (int tap = 0; tap < kKW; ++tap)
      rowp[tap] = reinterpret_cast<const float4*>(
          in + tile_src_row(ln, trow, tap - 2, p.N) * kHS + 16 * half);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float4 cur = rowp[0][0];
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);     // the read of group 0
#pragma unroll
    for (int g = 0; g < 20; ++g) {
      float4 nxt = cur;
      if (g + 1 < 20) nxt = rowp[(g + 1) >> 2][(g + 1) & 3];
      acc = DDD_MFMA32(w[4 * g + 0], cur.x, acc);
      acc = DDD_MFMA32(w[4 * g + 1], cur.y, acc);
      acc = DDD_MFMA32(w[4 * g + 2], cur.z, acc);
      acc = DDD_MFMA32(w[4 * g + 3], cur.w, acc);
      cur = nxt;
      // keep "read group g+1, then the 4 MFMAs of group g" in the schedule
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // 4 MFMA
    }
    acc = DDD_MFMA32(w[80], 1.0f, acc);   // bias row: k = 160 carries b[out]
    activate16(acc, p.act);
    store_tile32(out, trow, half, acc);
  }
}

// Output layer (32 -> C_out <= 16, linear) for this wave's four 16-row tiles.
//   A: lane l supplies W[out = l & 15][k = 4 s + (l >> 4)]
//   B: lane l supplies h[k = 4 s + (l >> 4)][position = l & 15]
//   step s = 8 tap + jj, quarter = l >> 4  <->  (tap, cin = 8 quarter + jj)
//   D: lane l holds position l & 15, out-channel 4 (l >> 4) + r.
// The 41 weight values are streamed from L2 tap by tap (8 live registers), not
// kept resident: with the hidden layer's 81 resident registers that keeps the
// kernel inside 256 VGPRs (2 waves / SIMD).
__device__ __forceinline__ void final_layer(const DevParams& p, const Lane& ln,
                                            const float* __restrict__ in,
                                            float* __restrict__ out) {
  const int j = ln.lane & 15;
  const int quarter = ln.lane >> 4;
  const float* __restrict__ wsrc = p.w_final + ln.lane;
  f32x4 acc[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
  for (int tap = 0; tap < kKW; ++tap) {
    float w[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) w[i] = wsrc[(tap * 8 + i) * 64];
    float4 v0[4], v1[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int trow = ln.wave * 64 + t * 16 + j;
      const float4* q = reinterpret_cast<const float4*>(
          in + tile_src_row(ln, trow, tap - 2, p.N) * kHS + 8 * quarter);
      v0[t] = q[0];
      v1[t] = q[1];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t] = DDD_MFMA16(w[0], v0[t].x, acc[t]);
      acc[t] = DDD_MFMA16(w[1], v0[t].y, acc[t]);
      acc[t] = DDD_MFMA16(w[2], v0[t].z, acc[t]);
      acc[t] = DDD_MFMA16(w[3], v0[t].w, acc[t]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      acc[t] = DDD_MFMA16(w[4], v1[t].x, acc[t]);
      acc[t] = DDD_MFMA16(w[5], v1[t].y, acc[t]);
      acc[t] = DDD_MFMA16(w[6], v1[t].z, acc[t]);
      acc[t] = DDD_MFMA16(w[7], v1[t].w, acc[t]);
    }
  }
  const float wb = wsrc[40 * 64];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    acc[t] = DDD_MFMA16(wb, 1.0f, acc[t]);   // bias row
    const int trow = ln.wave * 64 + t * 16 + j;
    *reinterpret_cast<float4*>(out + trow * kHS + 4 * quarter) =
        make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
  }
}

// One evaluation of finalize_time_derivative(t, predict_time_derivative(u))
// for the workgroup's rows.  Must be called by all 256 threads.
//   model.predict_coefficients  model.py:420-513   (conv tower + projection)
//   model.apply_coefficients    model.py:536-548   (stencil apply)
//   Equation.equation_of_motion equations.py       (dev_params.h)
//   finalize_time_derivative    equations.py:276-277 (forcing)
// kHoist: wts_hid already holds the (single) hidden layer's weights.
template <bool kHoist>
__device__ __forceinline__ float eval_rhs(const DevParams& p, Shared& sm, int batch,
                                          float u, float t,
                                          float (&wts_hid)[kHidSteps], bool frc_in_lds,
                                          float* derivs_out, float* coeffs_out) {
  const Lane ln = make_lane(p, batch, opaque((int)threadIdx.x));
  sm.u[ln.row] = u;
  if (!p.fixed) sm.un[ln.row] = u / p.stddev;   // model.py:450-451
  __syncthreads();

  // patches[i] = u[(x + i - G/2) mod N]   (model.extract_patches, model.py:516-533)
  float pch[kGMax];
  const int gl = p.G >> 1;
#pragma unroll
  for (int g = 0; g < kGMax; ++g)
    pch[g] = (g < p.G) ? sm.u[wrap_row(ln.base, ln.pos, g - gl, p.N)] : 0.0f;

  const float* net = nullptr;
  if (!p.fixed) {
    input_layer(p, ln, sm.un, sm.hA);
    float* in = sm.hA;
    float* out = sm.hB;
    for (int l = 1; l < p.L - 1; ++l) {
      if (!kHoist) load_hidden(p, l - 1, ln.lane, wts_hid);
      __syncthreads();
      hidden_layer(p, ln, in, out, wts_hid);
      float* tmp = in; in = out; out = tmp;
    }
    __syncthreads();
    final_layer(p, ln, in, out);
    __syncthreads();
    net = out + ln.row * kHS;
  } else {
    __syncthreads();   // all patch reads done before the next evaluation rewrites sm.u
  }

  // ---- projection onto the accuracy-constrained stencils + stencil apply -----
  // coeff = bias + net[start:stop] @ nullspace   (polynomials.py:275-277)
  // deriv = sum_i coeff[i] * patch[i]            (model.py:548)
  float dv[kMaxDerivs];
#pragma unroll
  for (int d = 0; d < kMaxDerivs; ++d) {
    dv[d] = 0.0f;
    if (d < p.D) {
      float coeff[kGMax];
#pragma unroll
      for (int g = 0; g < kGMax; ++g) coeff[g] = 0.0f;
      if (!p.fixed) {
        const float* __restrict__ ns = sm.tab + (4 + p.in_start[d]) * kGMax;
        for (int jx = 0; jx < p.in_size[d]; ++jx) {
          const float nv = net[p.in_start[d] + jx];
          const float4 n0 = *reinterpret_cast<const float4*>(ns + jx * kGMax);
          const float4 n1 = *reinterpret_cast<const float4*>(ns + jx * kGMax + 4);
          coeff[0] = fmaf(nv, n0.x, coeff[0]); coeff[1] = fmaf(nv, n0.y, coeff[1]);
          coeff[2] = fmaf(nv, n0.z, coeff[2]); coeff[3] = fmaf(nv, n0.w, coeff[3]);
          coeff[4] = fmaf(nv, n1.x, coeff[4]); coeff[5] = fmaf(nv, n1.y, coeff[5]);
          coeff[6] = fmaf(nv, n1.z, coeff[6]); coeff[7] = fmaf(nv, n1.w, coeff[7]);
        }
      }
      const float* __restrict__ b8 = sm.tab + d * kGMax;
#pragma unroll
      for (int g = 0; g < kGMax; ++g) coeff[g] = b8[g] + coeff[g];
      if (coeffs_out != nullptr && ln.active) {
        float* dst = coeffs_out + ((size_t)ln.gidx * p.D + d) * p.G;
#pragma unroll
        for (int g = 0; g < kGMax; ++g) if (g < p.G) dst[g] = coeff[g];
      }
      float s = 0.0f;
#pragma unroll
      for (int g = 0; g < kGMax; ++g) s = fmaf(coeff[g], pch[g], s);
      dv[d] = s;
    }
  }
  if (derivs_out != nullptr && ln.active) {
#pragma unroll
    for (int d = 0; d < kMaxDerivs; ++d)
      if (d < p.D) derivs_out[(size_t)ln.gidx * p.D + d] = dv[d];
  }

  // ---- equation of motion ------------------------------------------------------
  float r = equation_rhs_or_flux(p.equation, u, dv, p.eta);
  if (p.conservative) {
    sm.flux[ln.row] = r;
    __syncthreads();
    const float fnext = sm.flux[wrap_row(ln.base, ln.pos, 1, p.N)];
    r = -(p.inv_dx * (fnext - r));   // equations.staggered_first_derivative
  }
  if (p.forced) {
    if (frc_in_lds) {
      float total = 0.0f;
      const float4* frc = sm.frc + ln.sl * p.P;
      for (int m = 0; m < p.P; ++m) {
        const float4 q = frc[m];
        const float sp = p.sp[__float_as_int(q.w) * p.N + ln.pos];
        const float phase = (q.y * t + sp) + q.z;
        total = total + q.x * sinf(phase);
      }
      r = r + total;
    } else if (ln.active) {
      r = r + forcing_at(p, p.frc + (size_t)(ln.gidx / p.N) * p.P, ln.pos, t);
    }
  }
  return r;
}

// Per-launch staging of the small read-only tables into LDS.
__device__ __forceinline__ bool stage_tables(const DevParams& p, Shared& sm, int batch) {
  const int tid = threadIdx.x;
  if (tid < kTabRows * kGMax) {
    float v = 0.0f;
    const int rowi = tid / kGMax, g = tid % kGMax;
    if (rowi < 4) {
      if (rowi < p.D) v = p.bias8[rowi * kGMax + g];
    } else if (!p.fixed && rowi - 4 < p.C_out) {
      v = p.nullspace8[(rowi - 4) * kGMax + g];
    }
    sm.tab[tid] = v;
  }
  const int spg = kRows / p.N;
  const bool fits = p.forced && (spg * p.P <= kFrcLds);
  if (fits) {
    for (int i = tid; i < spg * p.P; i += blockDim.x) {
      const long sample = (long)blockIdx.x * spg + i / p.P;
      sm.frc[i] = sample < batch ? p.frc[sample * p.P + (i % p.P)]
                                 : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
  return fits;   // visibility: the first __syncthreads() of eval_rhs
}

// ---------------------------------------------------------------------------
// Kernel 1: one fused RK substep (also: plain time derivative, derivative and
// coefficient views).  State crosses HBM once in and once out.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void substep_kernel(DevParams p, SubstepArgs a) {
  __shared__ Shared sm;
  const Lane ln = make_lane(p, a.batch, threadIdx.x);
  const bool frc_lds = stage_tables(p, sm, a.batch);
  float wts_hid[kHidSteps];
  const float u = ln.active ? a.y_in[ln.gidx] : 0.0f;
  const float f = eval_rhs<false>(p, sm, a.batch, u, (float)a.t, wts_hid, frc_lds,
                                  a.derivs_out, a.coeffs_out);
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

// ---------------------------------------------------------------------------
// Kernel 2: persistent integrator.  The whole time loop runs inside one launch;
// each lane keeps its grid point's state in registers, HBM sees y0 once and the
// requested snapshots.
// ---------------------------------------------------------------------------
template <typename ST, bool kHoist>
__global__ __launch_bounds__(256, 2) void integrate_kernel(DevParams p, IntegrateArgs a) {
  __shared__ Shared sm;
  const Lane ln = make_lane(p, a.batch, threadIdx.x);
  const bool frc_lds = stage_tables(p, sm, a.batch);
  float wts_hid[kHidSteps];
  if (kHoist) load_hidden(p, 0, ln.lane, wts_hid);
  const ST* y0 = static_cast<const ST*>(a.y0);
  ST* y_out = static_cast<ST*>(a.y_out);
  ST y = ln.active ? y0[ln.gidx] : (ST)0;
  const ST h = (ST)a.dt;
  const size_t snap_stride = (size_t)a.batch * p.N;
  int until_save = a.save_every;
  size_t snap = 0;
  for (int step = 0; step < a.n_steps; ++step) {
    const double t = a.t0 + (double)step * a.dt;
    ST ynew = y;
    float kprev = 0.0f;
    for (int s = 0; s < a.tab.stages; ++s) {
      ST us = y;
      if (s > 0) us = y + (ST)kprev * ((ST)a.tab.a[s] * h);
      const float f = eval_rhs<kHoist>(p, sm, a.batch, (float)us,
                                       (float)(t + a.tab.c[s] * a.dt), wts_hid,
                                       frc_lds, nullptr, nullptr);
      if (a.tab.b[s] != 0.0f) ynew = ynew + ((ST)a.tab.b[s] * h) * (ST)f;
      kprev = f;
    }
    y = ynew;
    if (--until_save == 0) {
      until_save = a.save_every;
      if (ln.active) y_out[snap * snap_stride + ln.gidx] = y;
      ++snap;
    }
  }
}

}  // namespace mfma
}  // namespace ddd
