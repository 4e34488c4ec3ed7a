// Streaming fused RK substep for FIXED-stencil models (ddd_baseline_create):
//   model.baseline_space_derivatives  model.py:59-112   (explicit accuracy_order)
//   model.apply_space_derivatives     model.py:115-135
//   integrate.PolynomialDifferentiator integrate.py:74-105
// With one launch per substep this is the only HBM-shaped kernel of the path
// (~20 FMA against 8-12 B per grid point), so it is written as a stream: each
// thread owns eight consecutive grid points (two float4 per array; four where
// N is not a multiple of eight), a block stages 2048 (1024) points -- whole
// samples -- in LDS for the periodic stencil reads, and the grid is
// batch * N / 2048 blocks -- no per-sample workgroup setup.
// Unforced equations only (KdV, KS, unforced Burgers); forced or odd-sized
// cases keep the per-sample kernels (rhs_mfma.h / rhs_generic.h).
#pragma once
#include "dev_params.h"
#include "rhs_mfma.h"   // StagePick: per-stage constants as SGPR selects

namespace ddd {
namespace stream {

constexpr int kThreads = 256;
// kQuads: float4 rows per thread and array (2 where N % 8 == 0, else 1)
__host__ __device__ inline int quads_for(int n) { return n % 8 == 0 ? 2 : 1; }
__host__ __device__ inline int samples_per_block(int n) { return kThreads * 4 * quads_for(n) / n; }

// The configurations this kernel covers (checked on the host before launch).
inline bool supports(const DevParams& p) {
  return p.fixed && !p.weno && !p.forced && p.N >= 8 && p.N <= kThreads * 4 && p.N % 4 == 0 &&
         p.G <= kGMax;
}

template <int kQuads>
__global__ __launch_bounds__(kThreads) void fixed_substep_kernel(DevParams p, SubstepArgs a) {
  constexpr int kPer = 4 * kQuads;              // consecutive grid points per thread
  constexpr int kTilePoints = kThreads * kPer;  // grid points per block
  constexpr int kWin = kPer + 1 + kGMax - 1;    // stencil window of one thread (flux form: +1)
  __shared__ float tile[kTilePoints];
  const int n = p.N;
  const int pts = samples_per_block(n) * n;                 // multiple of kPer, <= kTilePoints
  const long base = (long)blockIdx.x * pts;
  const long total = (long)a.batch * n;
  const long rest = total - base;
  const int live = rest < (long)pts ? (int)rest : pts;      // whole samples
  const int i0 = threadIdx.x * kPer;
  // every global load of the thread is issued up front (the update's other
  // operands are not needed before the stencil is done: their latency hides
  // behind the LDS exchange and the arithmetic)
  const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  float4 base_in[kQuads], acc_in4[kQuads];
#pragma unroll
  for (int k = 0; k < kQuads; ++k) { base_in[k] = zero4; acc_in4[k] = zero4; }
  if (i0 < live) {
    const long gi0 = base + i0;
#pragma unroll
    for (int k = 0; k < kQuads; ++k)
      *reinterpret_cast<float4*>(tile + i0 + 4 * k) =
          *reinterpret_cast<const float4*>(a.y_in + gi0 + 4 * k);
    if (a.y_out != nullptr && a.y_base != nullptr && a.y_base != a.y_in) {
#pragma unroll
      for (int k = 0; k < kQuads; ++k)
        base_in[k] = *reinterpret_cast<const float4*>(a.y_base + gi0 + 4 * k);
    }
    if (a.acc_out != nullptr && a.acc_in != nullptr && a.acc_in != a.y_in) {
#pragma unroll
      for (int k = 0; k < kQuads; ++k)
        acc_in4[k] = *reinterpret_cast<const float4*>(a.acc_in + gi0 + 4 * k);
    }
  }
  __syncthreads();
  if (i0 >= live) return;

  const int s0 = (i0 / n) * n;     // first point of this thread's sample inside the tile
  const int pos0 = i0 - s0;        // its points are pos0 .. pos0 + kPer - 1 (N % kPer == 0)
  const int gl = p.G >> 1;         // patches[i] = u[(x + i - G/2) mod N]  (model.py:516-533)
  float w[kWin];
#pragma unroll
  for (int j = 0; j < kWin; ++j) {
    int q = pos0 - gl + j;         // in (-N, 2N): one conditional wrap each way
    q = q < 0 ? q + n : q;
    q = q >= n ? q - n : q;
    w[j] = tile[s0 + q];
  }
  // the points themselves (window entry gl + q, read directly: gl is a run-time value)
  float4 own[kQuads];
#pragma unroll
  for (int k = 0; k < kQuads; ++k) own[k] = *reinterpret_cast<const float4*>(tile + i0 + 4 * k);
  const int qn = pos0 + kPer >= n ? pos0 + kPer - n : pos0 + kPer;
  float uc[kPer + 1];
#pragma unroll
  for (int k = 0; k < kQuads; ++k) {
    uc[4 * k] = own[k].x; uc[4 * k + 1] = own[k].y; uc[4 * k + 2] = own[k].z; uc[4 * k + 3] = own[k].w;
  }
  uc[kPer] = tile[s0 + qn];
  // u_t (plain forms) or the flux (flux forms; one extra point for the
  // staggered difference, equations.staggered_first_derivative)
  float f[kPer + 1];
#pragma unroll
  for (int q = 0; q < kPer + 1; ++q) {
    f[q] = 0.0f;
    if (q < kPer || p.conservative) {
      float dv[kMaxDerivs];
#pragma unroll
      for (int d = 0; d < kMaxDerivs; ++d) {
        float s = 0.0f;
        if (d < p.D) {
#pragma unroll
          for (int g = 0; g < kGMax; ++g) s = fmaf(p.bias8[d][g], w[q + g], s);
        }
        dv[d] = s;
      }
      f[q] = equation_rhs_or_flux(p.equation, uc[q], dv, p.eta);
    }
  }
  float r[kPer];
#pragma unroll
  for (int q = 0; q < kPer; ++q)
    r[q] = p.conservative ? -(p.inv_dx * (f[q + 1] - f[q])) : f[q];

  const long gi = base + i0;
  if (a.y_out != nullptr) {
#pragma unroll
    for (int k = 0; k < kQuads; ++k) {
      float4 o = make_float4(a.c1 * r[4 * k], a.c1 * r[4 * k + 1], a.c1 * r[4 * k + 2],
                             a.c1 * r[4 * k + 3]);
      if (a.y_base != nullptr) {
        // stage 0 reads y as both input and base: reuse the staged tile
        const float4 b = a.y_base == a.y_in ? own[k] : base_in[k];
        o = make_float4(b.x + o.x, b.y + o.y, b.z + o.z, b.w + o.w);
      }
      *reinterpret_cast<float4*>(a.y_out + gi + 4 * k) = o;
    }
  }
  if (a.acc_out != nullptr) {
#pragma unroll
    for (int k = 0; k < kQuads; ++k) {
      float4 o = make_float4(a.c2 * r[4 * k], a.c2 * r[4 * k + 1], a.c2 * r[4 * k + 2],
                             a.c2 * r[4 * k + 3]);
      if (a.acc_in != nullptr) {
        const float4 b = a.acc_in == a.y_in ? own[k] : acc_in4[k];
        o = make_float4(b.x + o.x, b.y + o.y, b.z + o.z, b.w + o.w);
      }
      *reinterpret_cast<float4*>(a.acc_out + gi + 4 * k) = o;
    }
  }
}

// ALL stages of one Runge-Kutta step in one launch (DDD_LAUNCH_PER_STEP): a block
// already stages whole samples in LDS, so the stage inputs y + a h k are formed in
// the tile and never leave the CU: 8 B per grid point and STEP (y in, y out)
// instead of 20 B with one launch per midpoint substep.  Same arithmetic in the
// same order as the substep chain (y + (a h) k, acc + (b h) k): bit-identical.
// Two things keep it a stream although a tile now carries 2-4 stencil passes and
// barriers between them: (a) blocks are persistent -- a block walks over tiles
// gridDim.x apart with the NEXT tile's state already requested while the current
// one is computed; (b) the stencil window is read as aligned 16-byte quads (a
// thread's points start at a multiple of four, N % 4 == 0: a quad never straddles the
// periodic wrap) instead of one dword per window entry; (c) four points per thread:
// small threads, many wavefronts -- the kernel is latency-bound between barriers.
#ifndef DDD_STEP_QUADS
#define DDD_STEP_QUADS 1   // measured: 1 -> 6.50e11, 2 -> 4.31e11, 4 -> 3.48e11 grid-point-steps/s (occupancy:
#endif                     // the kernel is latency-bound between its barriers, profiles/r4_ablation.txt)
constexpr int kStepQuads = DDD_STEP_QUADS;          // float4 rows per thread
constexpr int kStepPer = 4 * kStepQuads;            // grid points per thread
constexpr int kStepTile = kThreads * kStepPer;      // grid points per tile
constexpr int kStepWin = kStepPer + 12;             // window floats: quads [pos0 - 4, pos0 + kStepPer + 8)
inline bool step_supports(const DevParams& p) {
  return supports(p) && p.N % kStepPer == 0 && p.G >= 3;
}

// Derivative count and form of the equation, as compile-time facts (the kernel is
// instantiated per equation: with the equation a run-time switch per grid point
// the fused step was instruction-bound at 2.5 TB/s -- profiles/r4_ablation.txt).
__host__ __device__ constexpr int eq_derivs(int eq) {
  return (eq == EQ_KS || eq == EQ_KS_CONS || eq == EQ_BURGERS_GODUNOV || eq == EQ_KDV_GODUNOV) ? 3
         : eq == EQ_KS_GODUNOV ? 4 : 2;
}
__host__ __device__ constexpr bool eq_flux_form(int eq) {
  return eq == EQ_BURGERS_CONS || eq == EQ_KDV_CONS || eq == EQ_KS_CONS || eq >= EQ_BURGERS_GODUNOV;
}

// kG: stencil points (no FMAs on zero-padded columns); G / 2 = offset of the stencil's first
// point; window = quads [pos0 - 4, pos0 + 16)
template <int kEq, int kG>
__device__ __forceinline__ void step_stage(const DevParams& p, const float* tile, int s0,
                                           int pos0, int n, float (&r)[kStepPer]) {
  static_assert(kG >= 3 && kG <= kGMax, "stencils of 3 to 8 points");
  constexpr int kGl = kG / 2;
  constexpr int kD = eq_derivs(kEq);
  constexpr bool kFlux = eq_flux_form(kEq);
  float blk[kStepWin];   // tile[s0 + (pos0 - 4 + i) mod N]
#pragma unroll
  for (int b = 0; b < kStepWin / 4; ++b) {
    int q = pos0 - 4 + 4 * b;
    q = q < 0 ? q + n : q;
    q = q >= n ? q - n : q;
    const float4 v = *reinterpret_cast<const float4*>(tile + s0 + q);
    blk[4 * b] = v.x; blk[4 * b + 1] = v.y; blk[4 * b + 2] = v.z; blk[4 * b + 3] = v.w;
  }
  float f[kStepPer + 1];
#pragma unroll
  for (int q = 0; q < kStepPer + 1; ++q) {
    f[q] = 0.0f;
    if (q < kStepPer || kFlux) {
      float dv[kMaxDerivs];
#pragma unroll
      for (int d = 0; d < kMaxDerivs; ++d) {
        float acc = 0.0f;
        if (d < kD) {
#pragma unroll
          for (int g = 0; g < kG; ++g)   // patches[g] = u[x + g - G/2]
            acc = fmaf(p.bias8[d][g], blk[4 - kGl + q + g], acc);
        }
        dv[d] = acc;
      }
      f[q] = equation_rhs_or_flux(kEq, blk[4 + q], dv, p.eta);
    }
  }
#pragma unroll
  for (int q = 0; q < kStepPer; ++q)
    r[q] = kFlux ? -(p.inv_dx * (f[q + 1] - f[q])) : f[q];
}

template <int kEq, int kG>
__global__ __launch_bounds__(kThreads) void fixed_step_kernel(DevParams p, StepArgs a, int tiles) {
  __shared__ float tile[kStepTile];
  const int n = p.N;
  const int pts = (kStepTile / n) * n;                     // whole samples per tile
  const long total = (long)a.batch * n;
  const int i0 = threadIdx.x * kStepPer;
  const int s0 = (i0 / n) * n;
  const int pos0 = i0 - s0;
  // a[s] h / b[s] h from StageConsts (host-formed, the same float products): indexing the
  // by-value tableau with the stage made the compiler copy it to SCRATCH -- a 32-byte frame
  // per thread, written at every launch: the 72 MiB of WRITE_SIZE against 64 MiB of state
  // in round 4's counters (profiles/r4_rocprof_summary.txt "stream_step")
  const mfma::StagePick<float> ah(a.sc.ah), bh(a.sc.bh);
  const float4 zero4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
  const auto fetch = [&](int t, float4 (&v)[kStepQuads]) {
    const long base = (long)t * pts;
    const long rest = total - base;
    const int live = rest < (long)pts ? (int)rest : pts;
    // (if / else, not `cond ? *ptr : zero4`: both arms of that conditional are lvalues, so
    // it selected between a global address and the ADDRESS of zero4 -- which put zero4 in a
    // scratch slot, written by every thread of every launch (the 8 MiB of WRITE_SIZE above
    // the 64 MiB of state in round 4's counters), and turned the load into a flat one)
#pragma unroll
    for (int k = 0; k < kStepQuads; ++k) {
      v[k] = zero4;
      if (t < tiles && i0 < live) v[k] = *reinterpret_cast<const float4*>(a.y_in + base + i0 + 4 * k);
    }
  };
  float4 next[kStepQuads];
  fetch((int)blockIdx.x, next);
  for (int t = (int)blockIdx.x; t < tiles; t += (int)gridDim.x) {
    const long base = (long)t * pts;
    const long rest = total - base;
    const int live = rest < (long)pts ? (int)rest : pts;
    const bool mine = i0 < live;
    float y[kStepPer], ynew[kStepPer];
#pragma unroll
    for (int k = 0; k < kStepQuads; ++k) {
      y[4 * k] = next[k].x; y[4 * k + 1] = next[k].y; y[4 * k + 2] = next[k].z; y[4 * k + 3] = next[k].w;
    }
    __syncthreads();   // the previous tile's last stage has read the LDS tile
#pragma unroll
    for (int k = 0; k < kStepQuads; ++k) *reinterpret_cast<float4*>(tile + i0 + 4 * k) = next[k];
    fetch(t + (int)gridDim.x, next);   // in flight during this tile's stages
#pragma unroll
    for (int q = 0; q < kStepPer; ++q) ynew[q] = y[q];
    for (int s = 0; s < a.tab.stages; ++s) {
      __syncthreads();   // the tile holds this stage's input
      float r[kStepPer];
      if (mine) {
        step_stage<kEq, kG>(p, tile, s0, pos0, n, r);
      } else {
#pragma unroll
        for (int q = 0; q < kStepPer; ++q) r[q] = 0.0f;
      }
      const bool last = s + 1 == a.tab.stages;
      if (((a.sc.b_nonzero >> s) & 1) || last) {
        const float c2 = bh.at(s);
#pragma unroll
        for (int q = 0; q < kStepPer; ++q) ynew[q] = ynew[q] + c2 * r[q];
      }
      if (!last) {
        __syncthreads();   // every stencil read of this stage is done
        const float c1 = ah.at(s + 1);
#pragma unroll
        for (int k = 0; k < kStepQuads; ++k)
          *reinterpret_cast<float4*>(tile + i0 + 4 * k) =
              make_float4(y[4 * k] + c1 * r[4 * k], y[4 * k + 1] + c1 * r[4 * k + 1],
                          y[4 * k + 2] + c1 * r[4 * k + 2], y[4 * k + 3] + c1 * r[4 * k + 3]);
      }
    }
    if (mine) {
#pragma unroll
      for (int k = 0; k < kStepQuads; ++k)
        *reinterpret_cast<float4*>(a.y_out + base + i0 + 4 * k) =
            make_float4(ynew[4 * k], ynew[4 * k + 1], ynew[4 * k + 2], ynew[4 * k + 3]);
    }
  }
}

// Host side: the instantiation for (equation, stencil width).
template <int kEq>
inline void launch_fixed_step_eq(int g, dim3 grid, hipStream_t stream, const DevParams& p,
                                 const StepArgs& a, int tiles) {
#define DDD_STEP_G(G) \
  case G: hipLaunchKernelGGL((fixed_step_kernel<kEq, G>), grid, dim3(kThreads), 0, stream, p, a, tiles); break;
  switch (g) {
    DDD_STEP_G(3) DDD_STEP_G(4) DDD_STEP_G(5) DDD_STEP_G(6) DDD_STEP_G(7)
    default: hipLaunchKernelGGL((fixed_step_kernel<kEq, 8>), grid, dim3(kThreads), 0, stream, p, a, tiles); break;
  }
#undef DDD_STEP_G
}
inline void launch_fixed_step(dim3 grid, hipStream_t stream, const DevParams& p, const StepArgs& a,
                              int tiles) {
  const int gl = p.G;   // (step_supports: 3 <= G <= 8)
  switch (p.equation) {
    case EQ_BURGERS: launch_fixed_step_eq<EQ_BURGERS>(gl, grid, stream, p, a, tiles); break;
    case EQ_BURGERS_CONS: launch_fixed_step_eq<EQ_BURGERS_CONS>(gl, grid, stream, p, a, tiles); break;
    case EQ_KDV: launch_fixed_step_eq<EQ_KDV>(gl, grid, stream, p, a, tiles); break;
    case EQ_KDV_CONS: launch_fixed_step_eq<EQ_KDV_CONS>(gl, grid, stream, p, a, tiles); break;
    case EQ_KS: launch_fixed_step_eq<EQ_KS>(gl, grid, stream, p, a, tiles); break;
    case EQ_KS_CONS: launch_fixed_step_eq<EQ_KS_CONS>(gl, grid, stream, p, a, tiles); break;
    case EQ_BURGERS_GODUNOV: launch_fixed_step_eq<EQ_BURGERS_GODUNOV>(gl, grid, stream, p, a, tiles); break;
    case EQ_KDV_GODUNOV: launch_fixed_step_eq<EQ_KDV_GODUNOV>(gl, grid, stream, p, a, tiles); break;
    default: launch_fixed_step_eq<EQ_KS_GODUNOV>(gl, grid, stream, p, a, tiles); break;
  }
}

}  // namespace stream
}  // namespace ddd
