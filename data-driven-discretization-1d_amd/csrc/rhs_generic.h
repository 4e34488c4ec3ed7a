// Generic (any configuration) learned-stencil right-hand side on the VALU.
//
// One workgroup per sample, activations ping-pong through dynamic LDS
// [N][Cmax] float32, every (grid point, output channel) pair is one fmaf
// chain.  Handles what the MFMA path does not: arbitrary N, kernel size,
// filter size, layer count, activation, stencil width, the accuracy-order-0
// head, and the space_derivatives / time_derivative / flux model targets
// (model.py:551-615).  It is an order of magnitude slower than the MFMA path
// and exists for coverage, not for speed.
#pragma once
#include "dev_params.h"
#include "rk23.h"

namespace ddd {
namespace generic {

constexpr int kThreads = 256;

struct Carve {
  float* act_a;
  float* act_b;
  float* u;      // [N]
  float* flux;   // [N]
  float* dy;     // [N]
  float* wl;     // LDS image of the weights ([K][Cin][Cout4] + bias [Cout4] per layer): of
                 // every layer (all_layers; staged once per launch), of one layer at a time
                 // (staged per layer and evaluation), or null (does not fit)
  bool all_layers;
};

// Floats of the per-layer weight stage: the largest layer with its output
// channels padded to a multiple of four (the conv loop reads four per ds_read_b128).
__host__ __device__ inline size_t weight_stage_floats(const DevParams& p) {
  size_t most = 0;
  for (int l = 0; l < p.L; ++l) {
    const size_t c4 = (size_t)((p.cout[l] + 3) & ~3);
    const size_t need = (size_t)p.K * p.cin[l] * c4 + c4;
    most = need > most ? need : most;
  }
  return most;
}
// ... and of all layers together (DevParams::weights4 as it lies in memory)
__host__ __device__ inline size_t weight_image_floats(const DevParams& p) {
  size_t all = 0;
  for (int l = 0; l < p.L; ++l) {
    const size_t c4 = (size_t)((p.cout[l] + 3) & ~3);
    all += (size_t)p.K * p.cin[l] * c4 + c4;
  }
  return all;
}

__host__ __device__ inline int max_channels(const DevParams& p) {
  int c = 1;
  for (int l = 0; l < p.L; ++l) c = p.cout[l] > c ? p.cout[l] : c;
  return c;
}

// Bytes of dynamic LDS for `state_bytes`-wide integration state (0 for the
// substep kernel).
// stage: 2 = every layer's weights resident in LDS, 1 = one layer at a time,
// 0 = none (weights read from L2): weight_stage picks.
__host__ __device__ inline size_t lds_bytes_with(const DevParams& p, int state_bytes,
                                                int stage) {
  const size_t n = (size_t)p.N;
  size_t floats = 3 * n;
  if (!p.fixed) floats += 2 * n * (size_t)max_channels(p);
  floats = (floats + 3) & ~(size_t)3;
  if (!p.fixed && stage == 1) floats += weight_stage_floats(p);
  if (!p.fixed && stage == 2) floats += weight_image_floats(p);
  size_t bytes = floats * sizeof(float);
  bytes = (bytes + 15) & ~(size_t)15;
  if (state_bytes) bytes += 2 * n * (size_t)state_bytes + n * sizeof(float);
  return bytes;
}
// adaptive_kernel's extra state behind the front part: 2 float64 + 3 float32 per point
__host__ __device__ inline size_t adaptive_extra_bytes(const DevParams& p) {
  return (size_t)p.N * (2 * sizeof(double) + 3 * sizeof(float));
}
__host__ __device__ inline int weight_stage(const DevParams& p, int state_bytes,
                                            size_t extra_bytes = 0) {
  if (p.fixed) return 0;
  const size_t cap = 160 * 1024;
  const size_t b1 = lds_bytes_with(p, state_bytes, 1) + extra_bytes;
  const size_t b2 = lds_bytes_with(p, state_bytes, 2) + extra_bytes;
  // every layer resident only where that does not cost a workgroup per CU: the
  // kernel is latency-bound (default net: 4 -> 3 workgroups per CU, 15.6 -> 10.2 %)
  if (b2 <= cap && (b1 > cap || cap / b2 >= cap / b1)) return 2;
  return b1 <= cap ? 1 : 0;
}
__host__ __device__ inline size_t lds_bytes(const DevParams& p, int state_bytes,
                                            size_t extra_bytes = 0) {
  return lds_bytes_with(p, state_bytes, weight_stage(p, state_bytes, extra_bytes));
}

__device__ __forceinline__ Carve carve(const DevParams& p, float* base, int stage) {
  Carve c;
  const int cm = p.fixed ? 0 : max_channels(p);
  c.act_a = base;
  c.act_b = base + (size_t)p.N * cm;
  c.u = base + 2 * (size_t)p.N * cm;
  c.flux = c.u + p.N;
  c.dy = c.flux + p.N;
  const size_t front = ((size_t)(3 * p.N) + 2 * (size_t)p.N * cm + 3) & ~(size_t)3;
  c.wl = (!p.fixed && stage > 0) ? base + front : nullptr;
  c.all_layers = stage == 2;
  return c;
}

// stage 2: the whole weight image, once per launch (all threads; ends with a barrier)
__device__ __forceinline__ void stage_all_layers(const DevParams& p, const Carve& c) {
  if (c.wl == nullptr || !c.all_layers) return;
  const float4* __restrict__ src = reinterpret_cast<const float4*>(p.weights4);
  float4* __restrict__ dst = reinterpret_cast<float4*>(c.wl);
  const int quads_total = (int)(weight_image_floats(p) / 4);
  for (int i = threadIdx.x; i < quads_total; i += kThreads) dst[i] = src[i];
  __syncthreads();
}

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 fma2(float x, f32x2 w, f32x2 a) {
  const f32x2 xx = {x, x};
  return __builtin_elementwise_fma(xx, w, a);   // fmaf per element
}

__device__ __forceinline__ int wrap(int i, int n) {
  i %= n;
  return i < 0 ? i + n : i;
}

// dy[0..N) = finalize_time_derivative(t, predict_time_derivative(u[0..N)))
// for the sample this workgroup owns.  u and dy live in LDS (c.u, c.dy).
__device__ inline void eval_rhs(const DevParams& p, const Carve& c, long sample,
                                float t, float* derivs_out, float* coeffs_out) {
  const int tid = threadIdx.x;
  const int n = p.N;
  const float* net = nullptr;   // [N][C_out]
  if (!p.fixed) {
    for (int i = tid; i < n; i += kThreads) c.act_a[i] = c.u[i] / p.stddev;
    __syncthreads();
    float* cur = c.act_a;
    float* nxt = c.act_b;
    const int cm_floats = (int)(c.act_b - c.act_a);   // both activation buffers 16-byte aligned?
    for (int l = 0; l < p.L; ++l) {
      const int cin = p.cin[l], cout = p.cout[l];
      const float* __restrict__ w = p.weights + p.w_off[l];
      const float* __restrict__ b = p.weights + p.b_off[l];
      const int act = (l < p.L - 1) ? p.act : ACT_NONE;
      const int left = p.K / 2;   // ceil((K-1)/2): layers.pad_periodic(center=True)
      if (c.wl != nullptr) {
        // Fast form: the layer's weights staged in LDS with the output channels
        // padded to fours; a thread owns 2 positions x 4 output channels and reads,
        // per (tap, input channel), one ds_read_b128 of weights and two
        // activations for eight FMAs.  Same accumulation order as the plain
        // form below (tap-major, then input channel): same bits.
        const int c4 = (cout + 3) & ~3;
        const float* __restrict__ wl = c.wl + (c.all_layers ? p.w4_off[l] : 0);
        if (!c.all_layers) {
          // the layer's LDS image is kept in global memory (DevParams::weights4): float4 copies
          const float4* __restrict__ src =
              reinterpret_cast<const float4*>(p.weights4 + p.w4_off[l]);
          float4* __restrict__ dst = reinterpret_cast<float4*>(c.wl);
          const int quads_total = (p.K * cin + 1) * (c4 / 4);
          for (int i = tid; i < quads_total; i += kThreads) dst[i] = src[i];
          __syncthreads();
        }
        const float* bl = wl + (size_t)p.K * cin * c4;
        const int quads = c4 / 4, pairs = (n + 1) / 2;
        for (int item = tid; item < pairs * quads; item += kThreads) {
          const int q = item % quads, pp = item / quads;
          const int pos0 = 2 * pp, pos1 = (2 * pp + 1 < n) ? 2 * pp + 1 : 2 * pp;
          float a0[4] = {0.0f, 0.0f, 0.0f, 0.0f}, a1[4] = {0.0f, 0.0f, 0.0f, 0.0f};
          // four input channels per trip where the rows allow 16-byte reads: two
          // ds_read_b128 of activations + four of weights for 32 FMAs, issued as
          // packed pairs (v_pk_fma_f32: the scalar FMA rate is half the f32 peak)
          const bool by4 = (cin & 3) == 0 && (cm_floats & 3) == 0;
          for (int k = 0; k < p.K; ++k) {
            const float* __restrict__ r0 = cur + (size_t)wrap(pos0 + k - left, n) * cin;
            const float* __restrict__ r1 = cur + (size_t)wrap(pos1 + k - left, n) * cin;
            const float4* __restrict__ wk =
                reinterpret_cast<const float4*>(wl + (size_t)k * cin * c4) + q;
            if (by4) {
              f32x2 p0a = {a0[0], a0[1]}, p0b = {a0[2], a0[3]};
              f32x2 p1a = {a1[0], a1[1]}, p1b = {a1[2], a1[3]};
              for (int ci = 0; ci < cin; ci += 4) {
                const float4 x0 = *reinterpret_cast<const float4*>(r0 + ci);
                const float4 x1 = *reinterpret_cast<const float4*>(r1 + ci);
                const float xs0[4] = {x0.x, x0.y, x0.z, x0.w}, xs1[4] = {x1.x, x1.y, x1.z, x1.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {   // (channel order ci, ci + 1, ...: the same sums)
                  const float4 w4 = wk[(size_t)(ci + j) * quads];
                  const f32x2 wa = {w4.x, w4.y}, wb = {w4.z, w4.w};
                  p0a = fma2(xs0[j], wa, p0a); p0b = fma2(xs0[j], wb, p0b);
                  p1a = fma2(xs1[j], wa, p1a); p1b = fma2(xs1[j], wb, p1b);
                }
              }
              a0[0] = p0a.x; a0[1] = p0a.y; a0[2] = p0b.x; a0[3] = p0b.y;
              a1[0] = p1a.x; a1[1] = p1a.y; a1[2] = p1b.x; a1[3] = p1b.y;
              continue;
            }
            for (int ci = 0; ci < cin; ++ci) {
              const float4 w4 = wk[(size_t)ci * quads];
              const float x0 = r0[ci], x1 = r1[ci];
              a0[0] = fmaf(x0, w4.x, a0[0]); a0[1] = fmaf(x0, w4.y, a0[1]);
              a0[2] = fmaf(x0, w4.z, a0[2]); a0[3] = fmaf(x0, w4.w, a0[3]);
              a1[0] = fmaf(x1, w4.x, a1[0]); a1[1] = fmaf(x1, w4.y, a1[1]);
              a1[2] = fmaf(x1, w4.z, a1[2]); a1[3] = fmaf(x1, w4.w, a1[3]);
            }
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int co = 4 * q + r;
            if (co >= cout) continue;
            nxt[(size_t)pos0 * cout + co] = apply_activation(a0[r] + bl[co], act);
            if (pos1 != pos0) nxt[(size_t)pos1 * cout + co] = apply_activation(a1[r] + bl[co], act);
          }
        }
      } else
      for (int idx = tid; idx < n * cout; idx += kThreads) {
        const int pos = idx / cout;
        const int co = idx - pos * cout;
        float acc = 0.0f;
        for (int k = 0; k < p.K; ++k) {
          const float* __restrict__ row = cur + (size_t)wrap(pos + k - left, n) * cin;
          const float* __restrict__ wk = w + (size_t)k * cin * cout + co;
          for (int ci = 0; ci < cin; ++ci) acc = fmaf(row[ci], wk[(size_t)ci * cout], acc);
        }
        nxt[idx] = apply_activation(acc + b[co], act);
      }
      __syncthreads();
      float* tmp = cur; cur = nxt; nxt = tmp;
    }
    net = cur;
  }

  const int gl = p.G / 2;
  // the staggered flux difference applies to flux-form equations of motion and
  // to the 'flux' model target, never to a directly predicted time derivative
  const bool direct_time = !p.fixed && p.target == TARGET_TIME_DERIVATIVE;
  const bool needs_flux_diff =
      !direct_time && (p.conservative || (!p.fixed && p.target == TARGET_FLUX));
  for (int pos = tid; pos < n; pos += kThreads) {
    const float y = c.u[pos];
    float r;
    if (!p.fixed && p.target == TARGET_TIME_DERIVATIVE) {
      r = net[pos];
    } else if (!p.fixed && p.target == TARGET_FLUX) {
      r = net[pos];
    } else {
      float dv[kMaxDerivs] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
      for (int d = 0; d < kMaxDerivs; ++d) {
        if (d >= p.D) continue;
        float s = 0.0f;
        if (!p.fixed && p.target == TARGET_SPACE_DERIVATIVES) {
          s = net[(size_t)pos * p.C_out + d];
        } else {
          float mean = 0.0f;
          if (!p.fixed && p.pao == 0 && p.unbiased) {
            for (int g = 0; g < p.G; ++g) mean += net[(size_t)pos * p.C_out + d * p.G + g];
            mean = mean / (float)p.G;
          }
          for (int g = 0; g < p.G; ++g) {
            float coeff;
            if (p.fixed) {
              coeff = p.bias[d * p.G + g];
            } else if (p.pao == 0) {
              coeff = net[(size_t)pos * p.C_out + d * p.G + g] - mean;
            } else {
              const float* __restrict__ ns = p.nullspace + p.ns_off[d];
              const float* __restrict__ nv = net + (size_t)pos * p.C_out + p.in_start[d];
              float proj = 0.0f;
              for (int jx = 0; jx < p.in_size[d]; ++jx) proj = fmaf(nv[jx], ns[jx * p.G + g], proj);
              coeff = p.bias[d * p.G + g] + proj;
            }
            if (coeffs_out != nullptr)
              coeffs_out[(((size_t)sample * n + pos) * p.D + d) * p.G + g] = coeff;
            s = fmaf(coeff, c.u[wrap(pos + g - gl, n)], s);
          }
        }
        dv[d] = s;
        if (derivs_out != nullptr)
          derivs_out[((size_t)sample * n + pos) * p.D + d] = s;
      }
      if (p.fixed && p.weno) {
        // WENODifferentiator / best WENO baseline: u_minus, u_plus replaced
        weno_minus_plus(c.u, pos, n, &dv[0], &dv[1]);
        if (derivs_out != nullptr) {
          derivs_out[((size_t)sample * n + pos) * p.D + 0] = dv[0];
          derivs_out[((size_t)sample * n + pos) * p.D + 1] = dv[1];
        }
      }
      r = equation_rhs_or_flux(p.equation, y, dv, p.eta);
    }
    if (needs_flux_diff) c.flux[pos] = r; else c.dy[pos] = r;
  }
  __syncthreads();
  for (int pos = tid; pos < n; pos += kThreads) {
    float r;
    if (needs_flux_diff) {
      const float here = c.flux[pos];
      const float next = c.flux[pos + 1 == n ? 0 : pos + 1];
      const float diff = p.inv_dx * (next - here);
      // model.predict_flux_directly returns +staggered_first_derivative(flux)
      // (model.py:609-615); the flux-form equations return its negative.
      r = (!p.fixed && p.target == TARGET_FLUX) ? diff : -diff;
    } else {
      r = c.dy[pos];
    }
    if (p.forced) r = r + forcing_at(p, p.frc + (size_t)sample * p.P, pos, t);
    c.dy[pos] = r;
  }
  __syncthreads();
}

__global__ __launch_bounds__(kThreads) void substep_kernel(DevParams p, SubstepArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Carve c = carve(p, smem, weight_stage(p, 0));
  stage_all_layers(p, c);
  const long sample = blockIdx.x;
  const size_t off = (size_t)sample * p.N;
  for (int i = threadIdx.x; i < p.N; i += kThreads) c.u[i] = a.y_in[off + i];
  __syncthreads();
  eval_rhs(p, c, sample, (float)a.t, a.derivs_out, a.coeffs_out);
  for (int i = threadIdx.x; i < p.N; i += kThreads) {
    const float f = c.dy[i];
    if (a.y_out != nullptr) {
      const float cf = a.c1 * f;
      a.y_out[off + i] = a.y_base != nullptr ? a.y_base[off + i] + cf : cf;
    }
    if (a.acc_out != nullptr) {
      const float cf = a.c2 * f;
      a.acc_out[off + i] = a.acc_in != nullptr ? a.acc_in[off + i] + cf : cf;
    }
  }
}

template <typename ST>
__global__ __launch_bounds__(kThreads) void integrate_kernel(DevParams p, IntegrateArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int staged = weight_stage(p, (int)sizeof(ST));
  const Carve c = carve(p, smem, staged);
  stage_all_layers(p, c);
  const size_t front = (lds_bytes_with(p, 0, staged) + 15) & ~(size_t)15;
  ST* y = reinterpret_cast<ST*>(reinterpret_cast<char*>(smem) + front);
  ST* ynew = y + p.N;
  float* kprev = reinterpret_cast<float*>(ynew + p.N);
  const long sample = blockIdx.x;
  const size_t off = (size_t)sample * p.N;
  const ST* y0 = static_cast<const ST*>(a.y0);
  ST* y_out = static_cast<ST*>(a.y_out);
  for (int i = threadIdx.x; i < p.N; i += kThreads) y[i] = y0[off + i];
  __syncthreads();
  const ST h = (ST)a.dt;
  const size_t snap_stride = (size_t)a.batch * p.N;
  size_t snap = 0;
  int until_save = a.save_every;
  for (int step = 0; step < a.n_steps; ++step) {
    const double t = a.t0 + (double)step * a.dt;
    for (int s = 0; s < a.tab.stages; ++s) {
      for (int i = threadIdx.x; i < p.N; i += kThreads) {
        ST us = y[i];
        if (s == 0) ynew[i] = us;
        else us = us + (ST)kprev[i] * ((ST)a.tab.a[s] * h);
        c.u[i] = (float)us;
      }
      __syncthreads();
      eval_rhs(p, c, sample, (float)(t + a.tab.c[s] * a.dt), nullptr, nullptr);
      for (int i = threadIdx.x; i < p.N; i += kThreads) {
        const float f = c.dy[i];
        if (a.tab.b[s] != 0.0f) ynew[i] = ynew[i] + ((ST)a.tab.b[s] * h) * (ST)f;
        kprev[i] = f;
      }
      __syncthreads();
    }
    for (int i = threadIdx.x; i < p.N; i += kThreads) y[i] = ynew[i];
    if (--until_save == 0) {
      until_save = a.save_every;
      for (int i = threadIdx.x; i < p.N; i += kThreads)
        y_out[snap * snap_stride + off + i] = ynew[i];
      ++snap;
    }
    __syncthreads();
  }
}

// Dynamic LDS of adaptive_kernel: the front part + y, y_new (float64) and three
// stage derivatives (float32) per grid point.
__host__ __device__ inline size_t adaptive_lds_bytes(const DevParams& p) {
  const int staged = weight_stage(p, 0, adaptive_extra_bytes(p) + 16);
  return ((lds_bytes_with(p, 0, staged) + 15) & ~(size_t)15) + adaptive_extra_bytes(p);
}

// integrate.odeint (integrate.py:143-169) for a batch of samples on the generic
// right-hand side: SciPy's RK23 (rk23.h) with one workgroup and one controller
// per sample, float64 state and controller, float32 right-hand side with the
// sample's own forcing.  Serves what the MFMA path does not carry -- first of
// all the WENO5 + Godunov-flux "exact" Burgers solver (WENODifferentiator,
// integrate.py:124-140) on its fine grid.
__global__ __launch_bounds__(kThreads) void adaptive_kernel(DevParams p, AdaptiveArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __shared__ double red[4];
  const int staged = weight_stage(p, 0, adaptive_extra_bytes(p) + 16);
  const Carve c = carve(p, smem, staged);
  stage_all_layers(p, c);
  const size_t front = (lds_bytes_with(p, 0, staged) + 15) & ~(size_t)15;
  double* y = reinterpret_cast<double*>(reinterpret_cast<char*>(smem) + front);
  double* y_new = y + p.N;
  float* k0 = reinterpret_cast<float*>(y_new + p.N);
  float* k1 = k0 + p.N;
  float* k2 = k1 + p.N;
  const int n = p.N, tid = (int)threadIdx.x;
  const long sample = blockIdx.x;
  const size_t off = (size_t)sample * n;
  const size_t row_stride = (size_t)a.batch * n;
  for (int i = tid; i < n; i += kThreads) { y[i] = a.y0[off + i]; y_new[i] = y[i]; }
  __syncthreads();

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
  while (ctl.status == rk23::RUNNING) {   // uniform over the workgroup
    double tt;
    if (phase == 0) tt = ctl.t;
    else if (phase == 1) tt = ctl.t + h0;
    else if (phase == 2) tt = ctl.t + 0.5 * ctl.h;
    else if (phase == 3) tt = ctl.t + 0.75 * ctl.h;
    else tt = ctl.t + ctl.h;
    for (int i = tid; i < n; i += kThreads) {
      double yy;
      if (phase == 0) yy = y[i];
      else if (phase == 1) yy = y[i] + h0 * (double)k0[i];
      else if (phase == 2) yy = rk23::stage2_input(y[i], k0[i], ctl.h);
      else if (phase == 3) yy = rk23::stage3_input(y[i], k0[i], k1[i], ctl.h);
      else yy = y_new[i];
      c.u[i] = (float)yy;
    }
    __syncthreads();
    eval_rhs(p, c, sample, (float)tt, nullptr, nullptr);   // -> c.dy, ends with a barrier
    ++ctl.nfev;
    double part = 0.0;
    if (phase == 0) {
      for (int i = tid; i < n; i += kThreads) k0[i] = c.dy[i];
      if (a.n_times == 1) {
        for (int i = tid; i < n; i += kThreads) a.y_out[off + i] = y[i];
        ctl.ti = 1;
        ctl.status = rk23::FINISHED;
      } else {
        for (int i = tid; i < n; i += kThreads) {
          const double q = y[i] / (atol + fabs(y[i]) * rtol);
          part += q * q;
        }
        const double d0 = sqrt(rk23::block_sum256(part, red)) / sqrt_n;
        part = 0.0;
        for (int i = tid; i < n; i += kThreads) {
          const double q = (double)c.dy[i] / (atol + fabs(y[i]) * rtol);
          part += q * q;
        }
        d1 = sqrt(rk23::block_sum256(part, red)) / sqrt_n;
        h0 = rk23::Control::first_guess(d0, d1, interval);
      }
      phase = 1;
    } else if (phase == 1) {
      for (int i = tid; i < n; i += kThreads) {
        const double q = (double)(c.dy[i] - k0[i]) / (atol + fabs(y[i]) * rtol);   // float32 difference
        part += q * q;
      }
      const double d2 = sqrt(rk23::block_sum256(part, red)) / sqrt_n / h0;
      ctl.initial_step(h0, d1, d2, interval, max_step);
      ctl.begin_step(max_step);
      ctl.begin_attempt(t_bound);
      phase = 2;
    } else if (phase == 2) {
      for (int i = tid; i < n; i += kThreads) k1[i] = c.dy[i];
      phase = 3;
    } else if (phase == 3) {
      for (int i = tid; i < n; i += kThreads) {
        k2[i] = c.dy[i];
        y_new[i] = rk23::new_state(y[i], k0[i], k1[i], k2[i], ctl.h);
      }
      phase = 4;
    } else {
      for (int i = tid; i < n; i += kThreads) {
        const double q = rk23::scaled_error(y[i], y_new[i], k0[i], k1[i], k2[i], c.dy[i], ctl.h,
                                            rtol, atol);
        part += q * q;
      }
      const double error_norm = sqrt(rk23::block_sum256(part, red)) / sqrt_n;
      if (ctl.error_test(error_norm)) {
        while (ctl.ti < a.n_times) {
          const double te = a.times[ctl.ti];
          if (!(te <= ctl.t_new)) break;
          const double x = (te - ctl.t) / ctl.h;
          for (int i = tid; i < n; i += kThreads)
            a.y_out[(size_t)ctl.ti * row_stride + off + i] =
                rk23::dense_output(y[i], k0[i], k1[i], k2[i], c.dy[i], x, ctl.h);
          ++ctl.ti;
        }
        for (int i = tid; i < n; i += kThreads) { y[i] = y_new[i]; k0[i] = c.dy[i]; }
        ctl.advance(t_bound, max_step);
      }
      ++attempts;
      if (ctl.status == rk23::RUNNING && attempts >= a.max_attempts)
        ctl.status = rk23::ATTEMPT_LIMIT;
      ctl.begin_attempt(t_bound);
      phase = 2;
    }
    __syncthreads();   // c.dy / k / y are rewritten by the next iteration
  }
  if (ctl.status != rk23::FINISHED) {
    const double nan = __longlong_as_double(0x7ff8000000000000ll);
    for (int r = ctl.ti; r < a.n_times; ++r)
      for (int i = tid; i < n; i += kThreads) a.y_out[(size_t)r * row_stride + off + i] = nan;
  }
  if (tid == 0) {
    a.nfev[sample] = ctl.nfev;
    a.status[sample] = ctl.status;
  }
}

}  // namespace generic
}  // namespace ddd
