"""The product's RK23 controller SOURCE (csrc/rk23.h: the per-sample controller
and per-grid-point formulas shared by rhs_adaptive.h, rhs_generic.h and
rhs_spectral.h) compiled for the CPU (oracle/rk23_host.cpp, g++) and driven over
Python right-hand sides, against the installed SciPy -- the reference's
integrator (integrate.py:154-155: solve_ivp(..., max_step=0.01, method='RK23')).
No GPU: this pins the statements the kernels execute, not a twin of them."""
import ctypes
import os
import subprocess

import numpy as np
import pytest
import scipy.integrate

from helpers import ROOT, make_model, random_phase_ic, batch_forcing, oracle

_PATH = os.path.join(ROOT, 'oracle', 'librk23host.so')
_FUN32 = ctypes.CFUNCTYPE(None, ctypes.c_double, ctypes.POINTER(ctypes.c_double),
                          ctypes.POINTER(ctypes.c_float), ctypes.c_void_p)
_FUN64 = ctypes.CFUNCTYPE(None, ctypes.c_double, ctypes.POINTER(ctypes.c_double),
                          ctypes.POINTER(ctypes.c_double), ctypes.c_void_p)


@pytest.fixture(scope='module')
def host():
  if not os.path.exists(_PATH):
    subprocess.run(['make', '-C', os.path.join(ROOT, 'oracle')], check=True, capture_output=True)
  return ctypes.CDLL(_PATH)


def _solve(host, fun, y0, times, max_step=0.01, rtol=1e-3, atol=1e-6, f64=False,
           max_attempts=0):
  n = len(y0)
  times = np.ascontiguousarray(times, np.float64)
  y0 = np.ascontiguousarray(y0, np.float64)
  out = np.empty((len(times), n))
  nfev = ctypes.c_int(0)
  dtype = np.float64 if f64 else np.float32

  def callback(t, y_ptr, out_ptr, _):
    y = np.ctypeslib.as_array(y_ptr, shape=(n,))
    np.ctypeslib.as_array(out_ptr, shape=(n,))[:] = np.asarray(fun(t, y.copy()), dtype)

  cb = (_FUN64 if f64 else _FUN32)(callback)
  entry = host.rk23_host_solve_f64 if f64 else host.rk23_host_solve_f32
  entry.restype = ctypes.c_int
  status = entry(cb, None, n, times.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), len(times),
                 ctypes.c_double(rtol), ctypes.c_double(atol), ctypes.c_double(max_step),
                 ctypes.c_longlong(max_attempts),
                 y0.ctypes.data_as(ctypes.POINTER(ctypes.c_double)),
                 out.ctypes.data_as(ctypes.POINTER(ctypes.c_double)), ctypes.byref(nfev))
  return out, nfev.value, status


def _toy(n, seed, stiffness, dtype):
  rs = np.random.RandomState(seed)
  q, _ = np.linalg.qr(rs.randn(n, n))
  a = ((q * np.linspace(1.0, stiffness, n)) @ q.T).astype(dtype)

  def fun(t, y):
    yd = np.asarray(y, dtype)
    return (-(a @ yd) - dtype(0.5) * yd ** 3 + dtype(np.sin(7 * t))).astype(dtype)
  return fun


@pytest.mark.parametrize('f64', [False, True])
@pytest.mark.parametrize('stiffness,max_step', [(3.0, 0.01), (3.0, np.inf), (300.0, np.inf),
                                               (3000.0, 0.01)])
def test_device_controller_source_equals_scipy(host, stiffness, max_step, f64):
  """Saturated, controller-limited and stability-limited (rejecting) runs, float32
  and float64 right-hand sides: nfev identical, dense output to rounding."""
  times = np.array([0.0, 0.0371, 0.2, 0.55, 1.0]) * (0.25 if stiffness > 1000 else 1.0)
  dtype = np.float64 if f64 else np.float32
  for seed in range(2):
    fun = _toy(16, seed, stiffness, dtype)
    y0 = np.random.RandomState(seed + 10).randn(16).astype(np.float32).astype(np.float64)
    sol = scipy.integrate.solve_ivp(fun, (times[0], times[-1]), y0, t_eval=times,
                                    max_step=max_step, method='RK23')
    y, nfev, status = _solve(host, fun, y0, times, max_step=max_step, f64=f64)
    assert status == 0 and sol.status == 0
    assert nfev == sol.nfev, (seed, nfev, sol.nfev)
    np.testing.assert_allclose(y, sol.y.T, rtol=1e-9, atol=1e-10)


def test_device_controller_source_failure_and_limits(host):
  def blow_up(t, y):
    y32 = np.asarray(y, np.float32)
    return y32 * y32
  times = np.linspace(0.0, 1.0, 6)
  y0 = np.array([2.5, 3.0])
  sol = scipy.integrate.solve_ivp(blow_up, (0.0, 1.0), y0, t_eval=times, max_step=0.01,
                                  method='RK23')
  y, nfev, status = _solve(host, blow_up, y0, times)
  assert sol.status == -1 and status == -1 and nfev == sol.nfev
  reached = sol.y.shape[1]
  np.testing.assert_allclose(y[:reached], sol.y.T, rtol=1e-9)
  assert np.isnan(y[reached:]).all()
  # attempt limit (a safety net SciPy does not have) and the single-time corner case
  y, nfev, status = _solve(host, blow_up, y0, times, max_attempts=4)
  assert status == -2 and nfev == 2 + 3 * 4
  y, nfev, status = _solve(host, blow_up, y0, [0.3])
  assert status == 0 and nfev == 1 and np.array_equal(y[0], y0)


def test_device_controller_source_on_the_learned_stencil_rhs(host):
  """The reference's actual use: SciPy RK23 over the (oracle's) learned-stencil
  right-hand side with forcing, one sample; same evaluations, same trajectory."""
  model = make_model('burgers', True, num_points=32, resample_factor=4)
  spec = model.spec()
  y0 = 0.5 * random_phase_ic(model.equation, 2)
  forcing = batch_forcing(2)
  times = np.array([0.0, 0.03, 0.11, 0.2])
  for b in range(2):
    one = {k: v[b] for k, v in forcing.items()}
    want, want_nfev = oracle.odeint_rk23(spec, y0[b], times, one)
    frc = {k: np.asarray(v)[None] for k, v in one.items()}
    fun = lambda t, y: oracle.time_derivative(spec, t, y[None, :], frc)[0]
    got, nfev, status = _solve(host, fun, y0[b].astype(np.float64), times)
    assert status == 0 and nfev == want_nfev
    np.testing.assert_allclose(got, want, rtol=1e-9, atol=1e-10)
