// SciPy's RK23 step-size controller (scipy/integrate/_ivp: RungeKutta.__init__,
// select_initial_step, RungeKutta._step_impl, rk_step, RkDenseOutput), the part
// that is per SAMPLE and scalar, shared by the three on-device adaptive
// integrators:
//   rhs_adaptive.h            learned / fixed stencils on the MFMA kernels, float32
//                             right-hand side (SavedModelDifferentiator, integrate.py:48-71)
//   rhs_generic.h             the generic kernel: WENO5 + Godunov "exact" Burgers solver
//                             (WENODifferentiator, integrate.py:124-140), any other net
//   rhs_spectral.h            spectral "exact" solver, float64 right-hand side
//                             (integrate.SpectralDifferentiator, integrate.py:108-121)
// Call site in the reference: integrate.odeint, integrate.py:154-155
//   solve_ivp(differentiator, (t0, t1), y0, t_eval=times, max_step=0.01, method='RK23').
// oracle/oracle.py::rk23_adaptive is the same restatement in NumPy, pinned
// against the installed SciPy (tests/test_cpu_oracle.py).  Python's min / max
// are restated with their NaN behaviour (min(a, b) is a unless b < a).
#pragma once
#ifdef DDD_RK23_HOST
// The same statements compiled by g++ for the CPU test tier
// (oracle/rk23_host.cpp, tests/test_cpu_rk23_source.py): this very header,
// driven over a Python right-hand side, against the installed SciPy.
#include <cmath>
#include <cstring>
#define __device__
#define __forceinline__ inline
#else
#include <hip/hip_runtime.h>
#endif

namespace ddd {
namespace rk23 {

enum : int { RUNNING = 1, FINISHED = 0, STEP_TOO_SMALL = -1, ATTEMPT_LIMIT = -2 };

// RungeKutta._step_impl: min_step = 10 |nextafter(t, inf) - t|
__device__ __forceinline__ long long bits_of(double x) {
#ifdef DDD_RK23_HOST
  long long b;
  std::memcpy(&b, &x, sizeof(b));
  return b;
#else
  return __double_as_longlong(x);
#endif
}
__device__ __forceinline__ double double_of(long long b) {
#ifdef DDD_RK23_HOST
  double x;
  std::memcpy(&x, &b, sizeof(x));
  return x;
#else
  return __longlong_as_double(b);
#endif
}
__device__ __forceinline__ double min_step_at(double t) {
  // t > 0 with a normal spacing (every time this path sees after t0 = 0): the
  // spacing above t is 2^(e - 52), e the biased exponent of t -- built from the
  // bits, exact, instead of the library nextafter's ~20 instructions (every VALU
  // instruction of the controller is paid for by the matrix pipe it shares the
  // lanes with).  Everything else (t <= 0, tiny, Inf, NaN) keeps nextafter.
  const long long e = bits_of(t) >> 52;   // sign bit clear when t > 0
  if (t > 0.0 && e > 52 && e < 2047) return 10.0 * double_of((e - 52) << 52);
  return 10.0 * fabs(nextafter(t, (double)INFINITY) - t);
}

struct Control {
  double t, t_new, h, h_abs;   // t_new / h: the attempt in flight
  int status, nfev, ti;        // ti: next entry of t_eval to emit
  bool rejected;               // an attempt of the current step was rejected

  __device__ __forceinline__ void init(double t0, bool valid) {
    t = t0; t_new = t0; h = 0.0; h_abs = 0.0;
    status = valid ? RUNNING : FINISHED;
    nfev = 0; ti = 0; rejected = false;
  }

  // select_initial_step, before its probe evaluation: h0
  __device__ __forceinline__ static double first_guess(double d0, double d1, double interval) {
    double h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
    if (interval < h0) h0 = interval;
    return h0;
  }

  // select_initial_step, after it: h_abs = min(100 h0, h1, interval, max_step)
  __device__ __forceinline__ void initial_step(double h0, double d1, double d2, double interval,
                                               double max_step) {
    double h1;
    if (d1 <= 1e-15 && d2 <= 1e-15) {
      h1 = h0 * 1e-3 > 1e-6 ? h0 * 1e-3 : 1e-6;
    } else {
      // (0.01 / max(d1, d2)) ** (1 / (error_estimator_order + 1)), order 2: the cube root
      // (as safety_factor below; the device library's pow is a 17-coefficient polynomial
      // whose constants the compiler parked in scratch inside the adaptive kernels' loop,
      // profiles/r5_spill_table.txt; differs from pow(x, 0.3333333333333333) by a few ulp)
      h1 = cbrt(0.01 / (d2 > d1 ? d2 : d1));
    }
    h_abs = 100.0 * h0;
    if (h1 < h_abs) h_abs = h1;
    if (interval < h_abs) h_abs = interval;
    if (max_step < h_abs) h_abs = max_step;
  }

  // _step_impl, head: the limits of the step about to be attempted
  __device__ __forceinline__ void begin_step(double max_step) {
    const double min_step = min_step_at(t);
    if (h_abs > max_step) h_abs = max_step;
    else if (h_abs < min_step) h_abs = min_step;
    rejected = false;
  }

  // ... and the head of its attempt loop
  __device__ __forceinline__ void begin_attempt(double t_bound) {
    if (status != RUNNING) return;
    if (h_abs < min_step_at(t) || !(h_abs == h_abs)) {   // TOO_SMALL_STEP (a NaN step would
      status = STEP_TOO_SMALL;                           // make SciPy spin forever)
      return;
    }
    t_new = t + h_abs;
    if (t_new - t_bound > 0.0) t_new = t_bound;
    h = t_new - t;
    h_abs = fabs(h);
  }

  // SAFETY * error_norm ** error_exponent, error_exponent = -1 / 3: the reciprocal
  // cube root (a third of pow's instruction count; differs from
  // pow(x, -0.3333333333333333) by a few units in the last place -- a relative
  // 1e-16 in the next step size, against 1e-9 asserted on whole trajectories)
  __device__ __forceinline__ static double safety_factor(double error_norm) {
    return 0.9 / cbrt(error_norm);
  }

  // the error test: true = step accepted (the caller then emits dense output
  // and calls advance); h_abs is rescaled either way
  __device__ __forceinline__ bool error_test(double error_norm) {
    if (error_norm < 1.0) {
      double factor;
      if (error_norm == 0.0) {
        factor = 10.0;                                      // MAX_FACTOR
      } else {
        factor = safety_factor(error_norm);
        if (!(factor < 10.0)) factor = 10.0;
      }
      if (rejected && !(factor < 1.0)) factor = 1.0;
      h_abs *= factor;
      return true;
    }
    const double factor = safety_factor(error_norm);
    h_abs *= factor > 0.2 ? factor : 0.2;                   // MIN_FACTOR
    rejected = true;
    return false;
  }

  __device__ __forceinline__ void advance(double t_bound, double max_step) {
    t = t_new;
    if (t - t_bound >= 0.0) status = FINISHED;
    else begin_step(max_step);
  }
};

#ifndef DDD_RK23_HOST
// Sum over a 256-thread workgroup, identical (bitwise) on every thread: wave
// butterflies, then the four wave totals in a fixed order.  `red`: 4 doubles of LDS.
__device__ __forceinline__ double block_sum256(double v, double* red) {
  for (int m = 1; m < 64; m <<= 1) v += __shfl_xor(v, m, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  const double s = ((red[0] + red[1]) + red[2]) + red[3];
  __syncthreads();
  return s;
}
#endif

// ---- per grid point: rk_step, the error estimate, RkDenseOutput --------------
// KT: type of the right-hand side's values (float32 for the TF-graph models,
// float64 for the spectral solver); all combinations are formed in float64 in
// the order NumPy forms them (np.dot over the stages, then * h).
template <typename KT>
__device__ __forceinline__ double stage2_input(double y, KT k0, double h) {
  return y + ((double)k0 * 0.5) * h;
}
template <typename KT>
__device__ __forceinline__ double stage3_input(double y, KT k0, KT k1, double h) {
  return y + ((double)k0 * 0.0 + (double)k1 * 0.75) * h;
}
template <typename KT>
__device__ __forceinline__ double new_state(double y, KT k0, KT k1, KT k2, double h) {
  return y + h * (((double)k0 * (2.0 / 9.0) + (double)k1 * (1.0 / 3.0)) +
                  (double)k2 * (4.0 / 9.0));
}
// err / scale, scale = atol + max(|y|, |y_new|) rtol  (np.maximum: NaN if either is)
template <typename KT>
__device__ __forceinline__ double scaled_error(double y, double y_new, KT k0, KT k1, KT k2,
                                               KT k3, double h, double rtol, double atol) {
  const double ay = fabs(y), an = fabs(y_new);
  double amax = ay > an ? ay : an;
  if (an != an) amax = an;
  if (ay != ay) amax = ay;
  const double scale = atol + amax * rtol;
  const double err = ((((double)k0 * (5.0 / 72.0) + (double)k1 * (-1.0 / 12.0)) +
                       (double)k2 * (-1.0 / 9.0)) + (double)k3 * (1.0 / 8.0)) * h;
  return err / scale;
}
// RkDenseOutput at x = (t_eval - t_old) / h: Q = K^T P, y_old + h Q [x, x^2, x^3]
template <typename KT>
__device__ __forceinline__ double dense_output(double y_old, KT k0, KT k1, KT k2, KT k3,
                                               double x, double h) {
  const double q1 = (((double)k0 * (-4.0 / 3.0) + (double)k1) + (double)k2 * (4.0 / 3.0)) -
                    (double)k3;
  const double q2 = (((double)k0 * (5.0 / 9.0) + (double)k1 * (-2.0 / 3.0)) +
                     (double)k2 * (-8.0 / 9.0)) + (double)k3;
  const double x2 = x * x, x3 = x2 * x;
  return h * (((double)k0 * x + q1 * x2) + q2 * x3) + y_old;
}

}  // namespace rk23
}  // namespace ddd
