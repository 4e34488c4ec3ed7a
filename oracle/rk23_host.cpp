// TEST INFRASTRUCTURE ONLY (CPU test tier; never part of the product).
//
// Drives the PRODUCT header csrc/rk23.h -- the per-sample RK23 controller and the
// per-grid-point formulas the three on-device adaptive integrators share --
// on the CPU, over a right-hand side supplied by the caller (a Python callback
// in tests/test_cpu_rk23_source.py), with the phase structure of the kernels
// (rhs_adaptive.h / rhs_generic.h / rhs_spectral.h: f0, the initial-step probe,
// stages 2 / 3 / FSAL).  What it proves without a GPU: the device source itself,
// not a twin of it, reproduces scipy.integrate.solve_ivp(method='RK23')
// (the reference's integrator, pde_superresolution/integrate.py:154-155).
#define DDD_RK23_HOST 1
#include "../data-driven-discretization-1d_amd/csrc/rk23.h"

#include <cmath>
#include <vector>

using ddd::rk23::Control;
namespace rk = ddd::rk23;

// KT: float (TF-graph models, generic kernel) or double (spectral solver)
template <typename KT>
static int solve(void (*fun)(double, const double*, KT*, void*), void* user, int n,
                 const double* times, int n_times, double rtol, double atol, double max_step,
                 long long max_attempts, const double* y0, double* y_out, int* nfev_out) {
  const double t0 = times[0], t_bound = times[n_times - 1];
  const double interval = std::fabs(t_bound - t0);
  const double sqrt_n = std::sqrt((double)n);
  std::vector<double> y(y0, y0 + n), y_new(y0, y0 + n), u(n);
  std::vector<KT> k0(n), k1(n), k2(n), f(n);
  const double nan = std::nan("");
  for (long i = 0; i < (long)n_times * n; ++i) y_out[i] = nan;
  const auto rms = [&](const std::vector<double>& q) {
    double s = 0.0;
    for (int i = 0; i < n; ++i) s += q[i] * q[i];
    return std::sqrt(s) / sqrt_n;
  };
  std::vector<double> q(n);
  Control c;
  c.init(t0, true);
  double h0 = 0.0, d1 = 0.0;
  long long attempts = 0;
  int phase = 0;
  while (c.status == rk::RUNNING) {
    double tt;
    for (int i = 0; i < n; ++i) {
      if (phase == 0) u[i] = y[i];
      else if (phase == 1) u[i] = y[i] + h0 * (double)k0[i];
      else if (phase == 2) u[i] = rk::stage2_input(y[i], k0[i], c.h);
      else if (phase == 3) u[i] = rk::stage3_input(y[i], k0[i], k1[i], c.h);
      else u[i] = y_new[i];
    }
    if (phase == 0) tt = c.t;
    else if (phase == 1) tt = c.t + h0;
    else if (phase == 2) tt = c.t + 0.5 * c.h;
    else if (phase == 3) tt = c.t + 0.75 * c.h;
    else tt = c.t + c.h;
    fun(tt, u.data(), f.data(), user);
    ++c.nfev;
    if (phase == 0) {
      k0 = f;
      if (n_times == 1) {
        for (int i = 0; i < n; ++i) y_out[i] = y[i];
        c.ti = 1;
        c.status = rk::FINISHED;
      } else {
        for (int i = 0; i < n; ++i) q[i] = y[i] / (atol + std::fabs(y[i]) * rtol);
        const double d0 = rms(q);
        for (int i = 0; i < n; ++i) q[i] = (double)k0[i] / (atol + std::fabs(y[i]) * rtol);
        d1 = rms(q);
        h0 = Control::first_guess(d0, d1, interval);
      }
      phase = 1;
    } else if (phase == 1) {
      for (int i = 0; i < n; ++i)   // the difference in the right-hand side's own type
        q[i] = (double)(KT)(f[i] - k0[i]) / (atol + std::fabs(y[i]) * rtol);
      const double d2 = rms(q) / h0;
      c.initial_step(h0, d1, d2, interval, max_step);
      c.begin_step(max_step);
      c.begin_attempt(t_bound);
      phase = 2;
    } else if (phase == 2) {
      k1 = f;
      phase = 3;
    } else if (phase == 3) {
      k2 = f;
      for (int i = 0; i < n; ++i) y_new[i] = rk::new_state(y[i], k0[i], k1[i], k2[i], c.h);
      phase = 4;
    } else {
      for (int i = 0; i < n; ++i)
        q[i] = rk::scaled_error(y[i], y_new[i], k0[i], k1[i], k2[i], f[i], c.h, rtol, atol);
      if (c.error_test(rms(q))) {
        while (c.ti < n_times && times[c.ti] <= c.t_new) {
          const double x = (times[c.ti] - c.t) / c.h;
          for (int i = 0; i < n; ++i)
            y_out[(long)c.ti * n + i] = rk::dense_output(y[i], k0[i], k1[i], k2[i], f[i], x, c.h);
          ++c.ti;
        }
        y = y_new;
        k0 = f;
        c.advance(t_bound, max_step);
      }
      ++attempts;
      if (c.status == rk::RUNNING && max_attempts > 0 && attempts >= max_attempts)
        c.status = rk::ATTEMPT_LIMIT;
      c.begin_attempt(t_bound);
      phase = 2;
    }
  }
  *nfev_out = c.nfev;
  return c.status;
}

extern "C" {
int rk23_host_solve_f32(void (*fun)(double, const double*, float*, void*), void* user, int n,
                        const double* times, int n_times, double rtol, double atol,
                        double max_step, long long max_attempts, const double* y0,
                        double* y_out, int* nfev) {
  return solve<float>(fun, user, n, times, n_times, rtol, atol, max_step, max_attempts, y0, y_out,
                      nfev);
}
int rk23_host_solve_f64(void (*fun)(double, const double*, double*, void*), void* user, int n,
                        const double* times, int n_times, double rtol, double atol,
                        double max_step, long long max_attempts, const double* y0,
                        double* y_out, int* nfev) {
  return solve<double>(fun, user, n, times, n_times, rtol, atol, max_step, max_attempts, y0,
                       y_out, nfev);
}
}
