// Batched adaptive Bogacki-Shampine integration on the device: SciPy's RK23
// (the reference's production integrator, integrate.py:143-169:
// solve_ivp(..., t_eval=times, max_step=0.01, method='RK23')) with ONE step-size
// controller per sample, inside the persistent MFMA kernel.
//
// What is restated (scipy/integrate/_ivp, third-party, unpinned by the
// reference; oracle/oracle.py::rk23_adaptive is the same restatement in NumPy,
// pinned against the installed SciPy):
//   RungeKutta.__init__      f0 = fun(t0, y0); select_initial_step (order 2)
//   RungeKutta._step_impl    min_step, clamp to [min_step, max_step], attempt loop,
//                            error norm = RMS(err / (atol + rtol max(|y|, |y_new|))),
//                            factor = SAFETY norm^(-1/3) in [0.2, 10], no growth
//                            after a rejection, TOO_SMALL_STEP failure
//   rk_step                  FSAL Bogacki-Shampine stages, K in float64
//   solve_ivp + RkDenseOutput  cubic dense output at every t_eval <= t
// The state, the stage combinations, the controller and the dense output are
// float64 (SciPy holds y in float64); the right-hand side is the float32
// evaluation of rhs_mfma.h, fed float32(y) and float32(t) exactly like the
// reference's TF placeholders (integrate.py:57-60).
//
// Shape: the work decomposition of integrate_kernel (rhs_mfma.h): a workgroup
// owns whole samples, lane == grid point.  Every sample carries its own t, h,
// status and evaluation count; the samples of one workgroup share the
// evaluations (all are at the same stage of *some* attempt), finished samples
// idle until the last sample of their workgroup is done.  One call site of
// eval_rhs (a phase counter drives what its input and result mean): the
// evaluation is ~3 k instructions and must not be replicated five times.
#pragma once
#include "rhs_mfma.h"

namespace ddd {
namespace mfma {

// Sum of `v` over the lanes of this lane's sample, identical (bitwise) on all of them.
template <int kRows, int kWR>
__device__ __forceinline__ double sample_sum(const DevParams& p, const Lane& ln, double v,
                                             double* red) {
  if (kRows == kWR) {
    // one wavefront, N | 64 (a power of two): xor butterfly inside aligned
    // groups of N lanes; a + b == b + a, so every lane ends with the same bits
    for (int m = 1; m < p.N; m <<= 1) v += __shfl_xor(v, m, 64);
    return v;
  } else {
    red[ln.row] = v;
    __syncthreads();
    double s = 0.0;
    const double* src = red + ln.base;   // spare rows: base 0, result unused
    for (int i = 0; i < p.N; ++i) s += src[i];
    __syncthreads();
    return s;
  }
}

template <int kRows, int kWR, bool kHoist, int kEq>
__global__ __launch_bounds__(kRows / kWR * 64, 2) void adaptive_kernel(DevParams p,
                                                                       AdaptiveArgs a) {
  __shared__ Shared<kRows, kWR> sm;
  __shared__ double red[kRows == kWR ? 2 : kRows];
  __shared__ float tev[kRows / 8];   // evaluation time of each sample of the group (N >= 8)
  const int tid = (int)threadIdx.x;
  const Lane ln = make_lane<kRows, kWR>(p, a.batch, tid, (int)blockIdx.x);
  Resident res;
  const bool fast_frc = launch_setup<kRows, kWR, kHoist>(p, sm, ln, a.batch, res);
  // sample whose (sample, mode) pair this lane evaluates in forcing phase 1
  const int frc_sl = (fast_frc && tid < (kRows / p.N) * p.P)
                         ? row_sample(tid, 1.0f / (float)p.P) : 0;
  const bool row_live = ln.row < ln.rows_used;

  const double t0 = a.times[0];
  const double t_bound = a.times[a.n_times - 1];
  const double rtol = a.rtol, atol = a.atol, max_step = a.max_step;
  const double sqrt_n = sqrt((double)p.N);   // _ivp.common.norm: ||x|| / x.size ** 0.5
  const size_t row_stride = (size_t)a.batch * p.N;

  double t = t0;
  double y = ln.valid ? a.y0[ln.gidx] : 0.0;
  double y_new = y, h = 0.0, h_abs = 0.0, t_new = t0;
  // select_initial_step's h0 and d1 live only until the first attempt starts:
  // they share the registers of h and t_new (set by begin_attempt after their
  // last use), which keeps the loop-carried state inside the register budget
  double& h0 = h;
  double& d1 = t_new;
  float k0 = 0.0f, k1 = 0.0f, k2 = 0.0f;
  int status = ln.valid ? 1 : 0;   // 1 running, 0 finished, -1 / -2 failed
  int nfev = 0, ti = 0;
  bool rejected = false;
  long long attempts = 0;

  // RungeKutta._step_impl, head: limits of the step about to be attempted
  const auto min_step_at = [](double tc) {
    return 10.0 * fabs(nextafter(tc, (double)INFINITY) - tc);
  };
  const auto begin_step = [&]() {
    const double min_step = min_step_at(t);
    if (h_abs > max_step) h_abs = max_step;
    else if (h_abs < min_step) h_abs = min_step;
    rejected = false;
  };
  // ... and the head of its attempt loop
  const auto begin_attempt = [&]() {
    if (status != 1) return;
    if (h_abs < min_step_at(t) || !(h_abs == h_abs)) {   // TOO_SMALL_STEP (a NaN step would spin forever)
      status = -1;
      return;
    }
    t_new = t + h_abs;
    if (t_new - t_bound > 0.0) t_new = t_bound;
    h = t_new - t;
    h_abs = fabs(h);
  };

  // phase 0: f(t0, y0); 1: the probe of select_initial_step; 2..4: stages 2, 3
  // and the FSAL stage of one attempt.  Uniform over the workgroup.
  int phase = 0;
  for (;;) {
    double tt, yy;
    if (phase == 0) { tt = t; yy = y; }
    else if (phase == 1) { tt = t + h0; yy = y + h0 * (double)k0; }
    else if (phase == 2) { tt = t + 0.5 * h; yy = y + ((double)k0 * 0.5) * h; }
    else if (phase == 3) { tt = t + 0.75 * h; yy = y + ((double)k0 * 0.0 + (double)k1 * 0.75) * h; }
    else { tt = t + h; yy = y_new; }
    if (status != 1) { tt = t; yy = y; }   // idle samples: keep the arithmetic finite

    if (fast_frc) {
      // harmonic forcing sums at THIS sample's time (no look-ahead: the next
      // evaluation's time is not known before the error test)
      if (row_live && ln.pos == 0) tev[ln.sl] = (float)tt;
      __syncthreads();
      res.fk_next = forcing_sums<kRows, kWR>(p, sm, res, tev[frc_sl], tid);
    }
    const float f = eval_rhs<kRows, kWR, kHoist, kEq, false>(
        p, sm, a.batch, (float)yy, (float)tt, (float)tt, res, fast_frc, nullptr, nullptr, false);
    if (status == 1) ++nfev;

    if (phase == 0) {
      k0 = f;
      if (a.n_times == 1) {   // t0 == t_bound: nothing to integrate
        if (status == 1) { a.y_out[ln.gidx] = y; ti = 1; status = 0; }
      } else {
        // select_initial_step, first half
        const double scale = atol + fabs(y) * rtol;
        const double q0 = y / scale, q1 = (double)k0 / scale;
        const double d0 = sqrt(sample_sum<kRows, kWR>(p, ln, q0 * q0, red)) / sqrt_n;
        d1 = sqrt(sample_sum<kRows, kWR>(p, ln, q1 * q1, red)) / sqrt_n;
        h0 = (d0 < 1e-5 || d1 < 1e-5) ? 1e-6 : 0.01 * d0 / d1;
        const double interval = fabs(t_bound - t0);
        if (interval < h0) h0 = interval;
      }
      phase = 1;
    } else if (phase == 1) {
      const double scale = atol + fabs(y) * rtol;
      const double q2 = (double)(f - k0) / scale;   // float32 difference, as SciPy forms it
      const double d2 = sqrt(sample_sum<kRows, kWR>(p, ln, q2 * q2, red)) / sqrt_n / h0;
      double h1;
      if (d1 <= 1e-15 && d2 <= 1e-15) {
        h1 = h0 * 1e-3 > 1e-6 ? h0 * 1e-3 : 1e-6;
      } else {
        h1 = pow(0.01 / (d2 > d1 ? d2 : d1), 1.0 / 3.0);
      }
      const double interval = fabs(t_bound - t0);
      h_abs = 100.0 * h0;
      if (h1 < h_abs) h_abs = h1;
      if (interval < h_abs) h_abs = interval;
      if (max_step < h_abs) h_abs = max_step;
      begin_step();
      begin_attempt();
      phase = 2;
    } else if (phase == 2) {
      k1 = f;
      phase = 3;
    } else if (phase == 3) {
      k2 = f;
      y_new = y + h * (((double)k0 * (2.0 / 9.0) + (double)k1 * (1.0 / 3.0)) +
                       (double)k2 * (4.0 / 9.0));
      phase = 4;
    } else {
      const float k3 = f;
      // np.maximum(|y|, |y_new|): NaN if either is
      const double ay = fabs(y), an = fabs(y_new);
      double amax = ay > an ? ay : an;
      if (an != an) amax = an;
      if (ay != ay) amax = ay;
      const double scale = atol + amax * rtol;
      const double err = ((((double)k0 * (5.0 / 72.0) + (double)k1 * (-1.0 / 12.0)) +
                           (double)k2 * (-1.0 / 9.0)) + (double)k3 * (1.0 / 8.0)) * h;
      const double q = err / scale;
      const double error_norm = sqrt(sample_sum<kRows, kWR>(p, ln, q * q, red)) / sqrt_n;
      if (status == 1) {
        if (error_norm < 1.0) {
          double factor;
          if (error_norm == 0.0) {
            factor = 10.0;
          } else {
            factor = 0.9 * pow(error_norm, -1.0 / 3.0);
            if (!(factor < 10.0)) factor = 10.0;
          }
          if (rejected && !(factor < 1.0)) factor = 1.0;
          h_abs *= factor;
          // solve_ivp: RkDenseOutput at every t_eval in (t_old, t_new]
          const double q1 = (((double)k0 * (-4.0 / 3.0) + (double)k1) + (double)k2 * (4.0 / 3.0)) -
                            (double)k3;
          const double q2 = (((double)k0 * (5.0 / 9.0) + (double)k1 * (-2.0 / 3.0)) +
                             (double)k2 * (-8.0 / 9.0)) + (double)k3;
          while (ti < a.n_times) {
            const double te = a.times[ti];
            if (!(te <= t_new)) break;
            const double x = (te - t) / h;
            const double x2 = x * x, x3 = x2 * x;
            a.y_out[(size_t)ti * row_stride + ln.gidx] =
                h * (((double)k0 * x + q1 * x2) + q2 * x3) + y;
            ++ti;
          }
          t = t_new;
          y = y_new;
          k0 = k3;
          if (t - t_bound >= 0.0) status = 0;
          else begin_step();
        } else {
          const double factor = 0.9 * pow(error_norm, -1.0 / 3.0);
          h_abs *= factor > 0.2 ? factor : 0.2;
          rejected = true;
        }
      }
      ++attempts;
      if (status == 1 && attempts >= a.max_attempts) status = -2;
      begin_attempt();
      phase = 2;
    }
    // the workgroup is done when none of its samples is running
    const int running = status == 1;
    if (kRows == kWR) {
      if (!__any(running)) break;
    } else {
      if (!__syncthreads_or(running)) break;
    }
  }

  if (ln.active) {
    if (status != 0) {
      const double nan = __longlong_as_double(0x7ff8000000000000ll);
      for (int i = ti; i < a.n_times; ++i) a.y_out[(size_t)i * row_stride + ln.gidx] = nan;
    }
    if (ln.pos == 0) {
      const long sample = ln.gidx / p.N;
      a.nfev[sample] = nfev;
      a.status[sample] = status;
    }
  }
}

}  // namespace mfma
}  // namespace ddd
