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
template <int kQuads>
__global__ __launch_bounds__(kThreads) void fixed_step_kernel(DevParams p, StepArgs a) {
  constexpr int kPer = 4 * kQuads;
  constexpr int kTilePoints = kThreads * kPer;
  constexpr int kWin = kPer + 1 + kGMax - 1;
  __shared__ float tile[kTilePoints];
  const int n = p.N;
  const int pts = samples_per_block(n) * n;
  const long base = (long)blockIdx.x * pts;
  const long total = (long)a.batch * n;
  const long rest = total - base;
  const int live = rest < (long)pts ? (int)rest : pts;
  const int i0 = threadIdx.x * kPer;
  const bool mine = i0 < live;
  float y[kPer], ynew[kPer];
#pragma unroll
  for (int q = 0; q < kPer; ++q) y[q] = 0.0f;
  if (mine) {
#pragma unroll
    for (int k = 0; k < kQuads; ++k) {
      const float4 v = *reinterpret_cast<const float4*>(a.y_in + base + i0 + 4 * k);
      y[4 * k] = v.x; y[4 * k + 1] = v.y; y[4 * k + 2] = v.z; y[4 * k + 3] = v.w;
      *reinterpret_cast<float4*>(tile + i0 + 4 * k) = v;
    }
  }
#pragma unroll
  for (int q = 0; q < kPer; ++q) ynew[q] = y[q];
  const int s0 = (i0 / n) * n;
  const int pos0 = i0 - s0;
  const int gl = p.G >> 1;
  const int qn = pos0 + kPer >= n ? pos0 + kPer - n : pos0 + kPer;
  const float h = (float)a.dt;
  for (int s = 0; s < a.tab.stages; ++s) {
    __syncthreads();   // the tile holds this stage's input
    float r[kPer];
#pragma unroll
    for (int q = 0; q < kPer; ++q) r[q] = 0.0f;
    if (mine) {
      float w[kWin];
#pragma unroll
      for (int j = 0; j < kWin; ++j) {
        int q = pos0 - gl + j;
        q = q < 0 ? q + n : q;
        q = q >= n ? q - n : q;
        w[j] = tile[s0 + q];
      }
      float uc[kPer + 1];
#pragma unroll
      for (int k = 0; k < kQuads; ++k) {
        const float4 v = *reinterpret_cast<const float4*>(tile + i0 + 4 * k);
        uc[4 * k] = v.x; uc[4 * k + 1] = v.y; uc[4 * k + 2] = v.z; uc[4 * k + 3] = v.w;
      }
      uc[kPer] = tile[s0 + qn];
      float f[kPer + 1];
#pragma unroll
      for (int q = 0; q < kPer + 1; ++q) {
        f[q] = 0.0f;
        if (q < kPer || p.conservative) {
          float dv[kMaxDerivs];
#pragma unroll
          for (int d = 0; d < kMaxDerivs; ++d) {
            float acc = 0.0f;
            if (d < p.D) {
#pragma unroll
              for (int g = 0; g < kGMax; ++g) acc = fmaf(p.bias8[d][g], w[q + g], acc);
            }
            dv[d] = acc;
          }
          f[q] = equation_rhs_or_flux(p.equation, uc[q], dv, p.eta);
        }
      }
#pragma unroll
      for (int q = 0; q < kPer; ++q)
        r[q] = p.conservative ? -(p.inv_dx * (f[q + 1] - f[q])) : f[q];
    }
    const bool last = s + 1 == a.tab.stages;
    if (a.tab.b[s] != 0.0f || last) {
      const float c2 = a.tab.b[s] * h;
#pragma unroll
      for (int q = 0; q < kPer; ++q) ynew[q] = ynew[q] + c2 * r[q];
    }
    if (!last) {
      __syncthreads();   // every stencil read of this stage is done
      const float c1 = a.tab.a[s + 1] * h;
      if (mine) {
#pragma unroll
        for (int k = 0; k < kQuads; ++k)
          *reinterpret_cast<float4*>(tile + i0 + 4 * k) =
              make_float4(y[4 * k] + c1 * r[4 * k], y[4 * k + 1] + c1 * r[4 * k + 1],
                          y[4 * k + 2] + c1 * r[4 * k + 2], y[4 * k + 3] + c1 * r[4 * k + 3]);
      }
    }
  }
  if (mine) {
#pragma unroll
    for (int k = 0; k < kQuads; ++k)
      *reinterpret_cast<float4*>(a.y_out + base + i0 + 4 * k) =
          make_float4(ynew[4 * k], ynew[4 * k + 1], ynew[4 * k + 2], ynew[4 * k + 3]);
  }
}

}  // namespace stream
}  // namespace ddd
