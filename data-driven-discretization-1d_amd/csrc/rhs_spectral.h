// Spectral "exact" right-hand side in float64:
//   integrate.SpectralDifferentiator   integrate.py:108-121
//   duckarray.spectral_derivative      duckarray.py:105-113 (model.py:78-80)
// Both reference forms are linear, translation-invariant operators on a
// periodic grid, i.e. circulant matrices: deriv_d[x] = sum_j c_d[(x - j) mod N] y[j]
// with c_d = the operator applied to a unit impulse (computed on the host with
// the very SciPy / NumPy call the reference uses, so Nyquist conventions carry
// over).  One workgroup per sample; y and the D kernels sit in LDS as float64;
// each thread owns grid points and runs D dot products of length N (N <= 2048).
// The reference evaluates this path in float64 NumPy; so does this kernel.
//
// Also here, both float64 and both for the exact KdV / KS solver that produces
// training data and evaluation baselines (integrate_exact, integrate.py:282-293):
//   adaptive_kernel          SciPy RK23 with one controller per sample, one
//                            workgroup per sample (rk23.h), whole solve in one launch
//   circulant_apply_kernel   duckarray.smoothing_filter (duckarray.py:116-128): the
//                            low-pass filter of odeint_with_periodic_filtering
//                            (integrate.py:172-212) is a circulant operator too
#pragma once
#include "dev_params.h"
#include "rk23.h"

namespace ddd {
namespace spectral {

constexpr int kThreads = 256;
constexpr int kMaxPoints = 2048;

struct Params {
  int equation, N, D;
  double eta;
  const double* kernels;   // [D][N] circulant first columns
  // FFT mode (N a power of two >= kFftMinPoints; round 5): the same D operators as
  // diagonal multipliers in Fourier space, mult[d][k] = DFT(kernels[d])[k] (formed on the
  // host in extended precision, capi.hip: ddd_spectral_create), and the twiddle table
  // exp(-2 pi i m / N), m < N / 2.  fft_log2n = 0: circulant mode.
  const double2* fft_mult;
  const double2* fft_twiddle;
  int fft_log2n;
};
constexpr int kFftMinPoints = 512;   // below: the O(N^2) circulant form wins (profiles/r3_spectral_exact.txt)

struct SubstepArgs64 {
  const double* y_in;
  const double* y_base;   // may be null
  double c1;
  double* y_out;          // may be null
  const double* acc_in;   // may be null
  double c2;
  double* acc_out;        // may be null
  int batch;
};

// circulant mode: the stage input + the D kernels; FFT mode: the stage input, three
// complex work buffers (spectrum, product, ping-pong partner) and the twiddle table
__host__ __device__ inline size_t lds_bytes(const Params& p) {
  if (p.fft_log2n > 0) return (size_t)8 * p.N * sizeof(double);
  return (size_t)(1 + p.D) * p.N * sizeof(double);
}

// equation_of_motion of the non-flux forms (equations.py:269-274, 410-415,
// 519-526) in the reference's expression order.
__device__ __forceinline__ double equation_rhs(int eq, double y, const double (&d)[kMaxDerivs],
                                               double eta) {
  switch (eq) {
    case EQ_BURGERS: return eta * d[1] - y * d[0];
    case EQ_KDV: return (-6.0 * y) * d[0] - d[1];
    default: return (-y * d[0] - d[2]) - d[1];   // EQ_KS: -u u_x - u_xxxx - u_xx
  }
}

// ---------------------------------------------------------------------------
// FFT mode (round 5): for N >= 512 the O(N^2) circulant products lose to an FFT
// (156 us per right-hand side at N = 512 and 4.3 ms at N = 2048 against 90 / 170 us
// for a rocFFT rfft -> multiply -> irfft chain of several launches,
// profiles/r3_spectral_exact.txt).  One workgroup per sample as before; the sample's
// spectrum never leaves LDS:
//   Y = FFT(u);  Z = Y m_0 + i Y m_1;  IFFT(Z) = deriv_0 + i deriv_1
// (two real derivatives per complex inverse: Y is Hermitian and the multipliers are
// the DFTs of REAL circulant kernels), a second inverse for a third derivative (KS).
// Radix-2 Stockham autosort (natural order in and out, no bit reversal), float64,
// N / 2 butterflies per stage over 256 threads, twiddles from an LDS table.
// ---------------------------------------------------------------------------
__device__ __forceinline__ double2 cmul(double2 a, double2 b) {
  return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}

// All threads of the workgroup.  x: input (complete, synchronised), y: partner buffer.
// Returns the buffer holding the transform (x or y).  kInverse: conjugate twiddles, no
// 1 / N (the caller scales).
template <bool kInverse>
__device__ __forceinline__ double2* fft_pow2(double2* x, double2* y, const double2* __restrict__ tw,
                                             int n, int log2n) {
  const int half = n >> 1;
  for (int s = 0; s < log2n; ++s) {
    const int ns = 1 << s;                    // half-size of the sub-transforms being merged
    const int tstride = half >> s;            // twiddle exp(-2 pi i k / (2 ns)) = tw[k * tstride]
    for (int j = (int)threadIdx.x; j < half; j += kThreads) {
      const int k = j & (ns - 1);
      double2 w = tw[k * tstride];
      if (kInverse) w.y = -w.y;
      const double2 a = x[j];
      const double2 b = cmul(w, x[j + half]);
      const int j0 = ((j - k) << 1) + k;
      y[j0] = make_double2(a.x + b.x, a.y + b.y);
      y[j0 + ns] = make_double2(a.x - b.x, a.y - b.y);
    }
    __syncthreads();
    double2* t = x; x = y; y = t;
  }
  return x;
}

// The D derivatives of the sample whose stage input sits in u[0 .. N) (complete,
// synchronised), at the points this thread owns (pos = tid + i kThreads), and the
// equation of motion.  work: 3 N double2 + N / 2 double2 of twiddles (lds_bytes).
template <int kPts>
__device__ __forceinline__ void eval_points_fft(const Params& p, const double* __restrict__ u,
                                                double2* work, double (&f)[kPts]) {
  const int n = p.N, tid = (int)threadIdx.x;
  double2* spec = work;            // the spectrum Y (kept across both inverses)
  double2* b0 = work + n;
  double2* b1 = work + 2 * n;
  const double2* tw = work + 3 * n;
  for (int i = tid; i < n; i += kThreads) b0[i] = make_double2(u[i], 0.0);
  __syncthreads();
  {
    double2* y = fft_pow2<false>(b0, b1, tw, n, p.fft_log2n);
    for (int i = tid; i < n; i += kThreads) spec[i] = y[i];
    __syncthreads();
  }
  const double inv_n = 1.0 / (double)n;
  double dv[kPts][kMaxDerivs];
#pragma unroll
  for (int i = 0; i < kPts; ++i)
#pragma unroll
    for (int d = 0; d < kMaxDerivs; ++d) dv[i][d] = 0.0;
  for (int d0 = 0; d0 < p.D; d0 += 2) {       // derivatives (d0, d0 + 1) per inverse transform
    const bool pair = d0 + 1 < p.D;
    for (int k = tid; k < n; k += kThreads) {
      const double2 yk = spec[k];
      double2 z = cmul(yk, p.fft_mult[(size_t)d0 * n + k]);
      if (pair) {
        const double2 z1 = cmul(yk, p.fft_mult[(size_t)(d0 + 1) * n + k]);
        z.x -= z1.y;                           // z + i z1
        z.y += z1.x;
      }
      b0[k] = z;
    }
    __syncthreads();
    const double2* r = fft_pow2<true>(b0, b1, tw, n, p.fft_log2n);
#pragma unroll
    for (int i = 0; i < kPts; ++i) {
      const int pos = tid + i * kThreads;
      if (pos >= n) continue;
      const double2 v = r[pos];
#pragma unroll
      for (int d = 0; d < kMaxDerivs; ++d) {   // (compile-time register indices)
        if (d == d0) dv[i][d] = v.x * inv_n;
        if (pair && d == d0 + 1) dv[i][d] = v.y * inv_n;
      }
    }
    __syncthreads();                           // r is rewritten by the next product
  }
#pragma unroll
  for (int i = 0; i < kPts; ++i) {
    const int pos = tid + i * kThreads;
    f[i] = pos < n ? equation_rhs(p.equation, u[pos], dv[i], p.eta) : 0.0;
  }
}

__device__ __forceinline__ void load_twiddles(const Params& p, double2* work) {
  double2* tw = work + 3 * p.N;
  for (int i = (int)threadIdx.x; i < p.N / 2; i += kThreads) tw[i] = p.fft_twiddle[i];
}

__global__ __launch_bounds__(kThreads) void substep_kernel(Params p, SubstepArgs64 a) {
  extern __shared__ __attribute__((aligned(16))) double smem64[];
  double* y = smem64;
  double* c = smem64 + p.N;
  const int n = p.N;
  const size_t off = (size_t)blockIdx.x * n;
  for (int i = threadIdx.x; i < n; i += kThreads) y[i] = a.y_in[off + i];
  if (p.fft_log2n > 0) load_twiddles(p, reinterpret_cast<double2*>(c));
  else for (int i = threadIdx.x; i < p.D * n; i += kThreads) c[i] = p.kernels[i];
  __syncthreads();
  double fft_f[kMaxPoints / kThreads];
  if (p.fft_log2n > 0)   // (uniform over the workgroup)
    eval_points_fft<kMaxPoints / kThreads>(p, y, reinterpret_cast<double2*>(c), fft_f);
  for (int pos = threadIdx.x, it = 0; pos < n; pos += kThreads, ++it) {
    double dv[kMaxDerivs] = {0.0, 0.0, 0.0, 0.0};
    int idx = pos;   // (pos - j) mod N
    for (int j = 0; j < n && p.fft_log2n == 0; ++j) {
      const double yj = y[j];
#pragma unroll
      for (int d = 0; d < kMaxDerivs; ++d)
        if (d < p.D) dv[d] = fma(c[d * n + idx], yj, dv[d]);
      idx = idx == 0 ? n - 1 : idx - 1;
    }
    double f = equation_rhs(p.equation, y[pos], dv, p.eta);
    if (p.fft_log2n > 0) {
      // (compile-time register index: the thread's `it`-th point)
#pragma unroll
      for (int q = 0; q < kMaxPoints / kThreads; ++q) if (q == it) f = fft_f[q];
    }
    const size_t gi = off + pos;
    if (a.y_out != nullptr) {
      const double cf = a.c1 * f;
      a.y_out[gi] = a.y_base != nullptr ? a.y_base[gi] + cf : cf;
    }
    if (a.acc_out != nullptr) {
      const double cf = a.c2 * f;
      a.acc_out[gi] = a.acc_in != nullptr ? a.acc_in[gi] + cf : cf;
    }
  }
}

// Right-hand side of the workgroup's sample at the points this thread owns
// (pos = tid + i kThreads): u = stage input in LDS (complete), c = the D kernels.
template <int kPts>
__device__ __forceinline__ void eval_points(const Params& p, const double* __restrict__ u,
                                            const double* __restrict__ c, double (&f)[kPts]) {
  const int n = p.N;
#pragma unroll
  for (int i = 0; i < kPts; ++i) {
    const int pos = (int)threadIdx.x + i * kThreads;
    f[i] = 0.0;
    if (pos >= n) continue;
    double dv[kMaxDerivs] = {0.0, 0.0, 0.0, 0.0};
    int idx = pos;   // (pos - j) mod N
    for (int j = 0; j < n; ++j) {
      const double uj = u[j];
#pragma unroll
      for (int d = 0; d < kMaxDerivs; ++d)
        if (d < p.D) dv[d] = fma(c[d * n + idx], uj, dv[d]);
      idx = idx == 0 ? n - 1 : idx - 1;
    }
    f[i] = equation_rhs(p.equation, u[pos], dv, p.eta);
  }
}

// integrate.odeint over SpectralDifferentiator (integrate.py:108-121, 143-169)
// for a batch of samples: SciPy's RK23 (rk23.h) with float64 right-hand side,
// one workgroup and one controller per sample, kPts grid points per thread.
// The equations of motion are autonomous and carry no forcing here (KdV / KS:
// finalize_time_derivative is the identity, equations.py:417-419, 528-530).
template <int kPts>
__global__ __launch_bounds__(kThreads) void adaptive_kernel(Params p, AdaptiveArgs a) {
  extern __shared__ __attribute__((aligned(16))) double smem64[];
  double* u = smem64;
  double* c = smem64 + p.N;
  __shared__ double red[4];
  const int n = p.N;
  const int tid = (int)threadIdx.x;
  const size_t off = (size_t)blockIdx.x * n;
  const size_t row_stride = (size_t)a.batch * n;
  const bool use_fft = p.fft_log2n > 0;   // (uniform)
  if (use_fft) load_twiddles(p, reinterpret_cast<double2*>(c));
  else for (int i = tid; i < p.D * n; i += kThreads) c[i] = p.kernels[i];

  const double t0 = a.times[0];
  const double t_bound = a.times[a.n_times - 1];
  const double interval = fabs(t_bound - t0);
  const double rtol = a.rtol, atol = a.atol, max_step = a.max_step;
  const double sqrt_n = sqrt((double)n);
  double y[kPts], y_new[kPts], k0[kPts], k1[kPts], k2[kPts], f[kPts];
#pragma unroll
  for (int i = 0; i < kPts; ++i) {
    const int pos = tid + i * kThreads;
    y[i] = pos < n ? a.y0[off + pos] : 0.0;
    y_new[i] = y[i]; k0[i] = k1[i] = k2[i] = 0.0;
  }
  const auto rms = [&](const double (&q)[kPts]) {
    double part = 0.0;
#pragma unroll
    for (int i = 0; i < kPts; ++i)
      if (tid + i * kThreads < n) part += q[i] * q[i];
    return sqrt(rk23::block_sum256(part, red)) / sqrt_n;
  };

  rk23::Control ctl;
  ctl.init(t0, true);
  double h0 = 0.0, d1 = 0.0;
  long long attempts = 0;
  int phase = 0;
  while (ctl.status == rk23::RUNNING) {   // the controller is uniform over the workgroup
#pragma unroll
    for (int i = 0; i < kPts; ++i) {
      const int pos = tid + i * kThreads;
      if (pos >= n) continue;
      double yy;
      if (phase == 0) yy = y[i];
      else if (phase == 1) yy = y[i] + h0 * k0[i];
      else if (phase == 2) yy = rk23::stage2_input(y[i], k0[i], ctl.h);
      else if (phase == 3) yy = rk23::stage3_input(y[i], k0[i], k1[i], ctl.h);
      else yy = y_new[i];
      u[pos] = yy;
    }
    __syncthreads();
    if (use_fft) eval_points_fft<kPts>(p, u, reinterpret_cast<double2*>(c), f);
    else eval_points<kPts>(p, u, c, f);
    __syncthreads();
    ++ctl.nfev;
    double q[kPts];
    if (phase == 0) {
#pragma unroll
      for (int i = 0; i < kPts; ++i) k0[i] = f[i];
      if (a.n_times == 1) {
#pragma unroll
        for (int i = 0; i < kPts; ++i)
          if (tid + i * kThreads < n) a.y_out[off + tid + i * kThreads] = y[i];
        ctl.ti = 1;
        ctl.status = rk23::FINISHED;
      } else {
#pragma unroll
        for (int i = 0; i < kPts; ++i) q[i] = y[i] / (atol + fabs(y[i]) * rtol);
        const double d0 = rms(q);
#pragma unroll
        for (int i = 0; i < kPts; ++i) q[i] = k0[i] / (atol + fabs(y[i]) * rtol);
        d1 = rms(q);
        h0 = rk23::Control::first_guess(d0, d1, interval);
      }
      phase = 1;
    } else if (phase == 1) {
#pragma unroll
      for (int i = 0; i < kPts; ++i) q[i] = (f[i] - k0[i]) / (atol + fabs(y[i]) * rtol);
      const double d2 = rms(q) / h0;
      ctl.initial_step(h0, d1, d2, interval, max_step);
      ctl.begin_step(max_step);
      ctl.begin_attempt(t_bound);
      phase = 2;
    } else if (phase == 2) {
#pragma unroll
      for (int i = 0; i < kPts; ++i) k1[i] = f[i];
      phase = 3;
    } else if (phase == 3) {
#pragma unroll
      for (int i = 0; i < kPts; ++i) {
        k2[i] = f[i];
        y_new[i] = rk23::new_state(y[i], k0[i], k1[i], k2[i], ctl.h);
      }
      phase = 4;
    } else {
#pragma unroll
      for (int i = 0; i < kPts; ++i)
        q[i] = rk23::scaled_error(y[i], y_new[i], k0[i], k1[i], k2[i], f[i], ctl.h, rtol, atol);
      const double error_norm = rms(q);
      if (ctl.error_test(error_norm)) {
        while (ctl.ti < a.n_times) {
          const double te = a.times[ctl.ti];
          if (!(te <= ctl.t_new)) break;
          const double x = (te - ctl.t) / ctl.h;
#pragma unroll
          for (int i = 0; i < kPts; ++i)
            if (tid + i * kThreads < n)
              a.y_out[(size_t)ctl.ti * row_stride + off + tid + i * kThreads] =
                  rk23::dense_output(y[i], k0[i], k1[i], k2[i], f[i], x, ctl.h);
          ++ctl.ti;
        }
#pragma unroll
        for (int i = 0; i < kPts; ++i) { y[i] = y_new[i]; k0[i] = f[i]; }
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
      for (int i = 0; i < kPts; ++i)
        if (tid + i * kThreads < n) a.y_out[(size_t)r * row_stride + off + tid + i * kThreads] = nan;
  }
  if (tid == 0) {
    a.nfev[blockIdx.x] = ctl.nfev;
    a.status[blockIdx.x] = ctl.status;
  }
}

// out[r][x] = sum_j kernel[(x - j) mod N] in[r][j]: duckarray.smoothing_filter
// (irfft(sigma * rfft(x)), duckarray.py:116-128) with kernel = that call applied
// to a unit impulse.  One workgroup per row; in-place (out == in) is allowed.
__global__ __launch_bounds__(kThreads) void circulant_apply_kernel(
    const double* __restrict__ kernel, const double* in, double* out, int n) {
  extern __shared__ __attribute__((aligned(16))) double smem64[];
  double* x = smem64;
  double* c = smem64 + n;
  const size_t off = (size_t)blockIdx.x * n;
  for (int i = threadIdx.x; i < n; i += kThreads) { x[i] = in[off + i]; c[i] = kernel[i]; }
  __syncthreads();
  for (int pos = threadIdx.x; pos < n; pos += kThreads) {
    double acc = 0.0;
    int idx = pos;
    for (int j = 0; j < n; ++j) {
      acc = fma(c[idx], x[j], acc);
      idx = idx == 0 ? n - 1 : idx - 1;
    }
    out[off + pos] = acc;
  }
}

}  // namespace spectral
}  // namespace ddd
