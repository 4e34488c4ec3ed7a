"""The reference's own integration tests (integrate_test.py:47-203) re-run
against the HIP solvers: same equations, grid sizes, time grids, tolerances
and assertions -- all thirteen parameterisations of its exact / baseline /
model triple (integrate_test.py:55-69), the three Kuramoto-Sivashinsky ones
included.  The trained checkpoint is replaced by a synthetic model of the
architecture those tests train: the DEFAULT three-layer net (integrate_test.py:48
defines `model_kwargs = dict(num_layers=1, filter_size=32)` but never passes it to
create_hparams; its training on noise is out of scope here).  The two
single-solver KS cases with warmup=50 use a shorter warm-up (the explicit RK23
solver needs ~3e4 evaluations per time unit on that equation)."""
import json

import numpy as np
import pytest

import ddd1d_amd
from ddd1d_amd import duckarray, equations, integrate, model as model_lib

pytestmark = pytest.mark.gpu

NUM_X_POINTS = 256
RANDOM_SEED = 0
TIMES = np.linspace(0, 1, num=11)


def _y(ds, name='y'):
  v = ds.data_vars[name]
  return np.asarray(v[1] if isinstance(v, tuple) else v)


def _assert_mean_zero(y):
  np.testing.assert_allclose(y.mean(axis=-1), 0, atol=1e-3)


@pytest.mark.parametrize('hparam_values,warmup,conservative,filter_interval', [
    (dict(equation='burgers'), 0, False, None),
    (dict(equation='kdv'), 0, False, None),
    (dict(equation='burgers'), 0, True, None),
    (dict(equation='kdv'), 0, True, None),
    (dict(equation='burgers', numerical_flux=True), 0, True, None),
    (dict(equation='kdv', numerical_flux=True), 0, True, None),
    (dict(equation='ks'), 0, False, None),
    (dict(equation='ks'), 0, True, None),
    (dict(equation='ks', numerical_flux=True), 0, True, None),
    (dict(equation='burgers'), 1, False, None),
    (dict(equation='burgers'), 1, True, None),
    (dict(equation='kdv'), 1, True, None),
    (dict(equation='kdv'), 1, True, 1),
])
def test_integrate_exact_baseline_and_model(hparam_values, warmup, conservative,
                                            filter_interval, resample_factor=4):
  """integrate_test.py:72-127."""
  hparams = ddd1d_amd.create_hparams(
      equation_kwargs=json.dumps({'num_points': NUM_X_POINTS}),
      conservative=conservative, resample_factor=resample_factor, **hparam_values)
  _, eq_coarse = equations.from_hparams(hparams, random_seed=RANDOM_SEED)
  model = model_lib.LearnedStencilModel(eq_coarse, hparams, init_seed=0, output_scale=0.02)
  results = integrate.integrate_exact_baseline_and_model(
      None, hparams=hparams, random_seed=RANDOM_SEED, times=TIMES, warmup=warmup,
      exact_filter_interval=filter_interval, model=model)
  y_exact, y_base, y_model = (_y(results, k) for k in ('y_exact', 'y_baseline', 'y_model'))
  assert y_exact.shape == (11, NUM_X_POINTS)
  assert y_base.shape == y_model.shape == (11, NUM_X_POINTS // resample_factor)
  _assert_mean_zero(y_exact)                                   # 'average should be zero'
  resample = duckarray.resample_mean if conservative else duckarray.subsample
  first = resample(y_exact[0], resample_factor)                # 'matching initial conditions'
  np.testing.assert_allclose(first, y_base[0])
  np.testing.assert_allclose(first, y_model[0])
  equation_type = equations.equation_type_from_hparams(hparams)  # 'matches integrate_baseline'
  assert equation_type.CONSERVATIVE == conservative
  equation = equation_type(NUM_X_POINTS // resample_factor, resample_factor=resample_factor,
                           random_seed=RANDOM_SEED)
  results2 = integrate.integrate_baseline(equation, times=TIMES, warmup=warmup,
                                          exact_filter_interval=filter_interval)
  np.testing.assert_allclose(y_base, _y(results2), atol=1e-5)
  assert np.isfinite(y_model).all()


@pytest.mark.parametrize('equation,kwargs', [
    (equations.BurgersEquation(200), {}),
    (equations.KdVEquation(200), {}),
    (equations.KSEquation(200), dict(warmup=0.5)),     # reference: warmup=50.0
])
def test_integrate_exact(equation, kwargs):
  """integrate_test.py:129-143."""
  results = integrate.integrate_exact(equation, times=TIMES, **kwargs)
  y = _y(results)
  assert y.shape == (11, 200)
  _assert_mean_zero(y)


def test_burgers_exact_weno():
  """integrate_test.py:145-154."""
  exact = integrate.integrate_exact(equations.BurgersEquation(200), times=TIMES)
  weno = integrate.integrate_weno(equations.GodunovBurgersEquation(200), times=TIMES)
  np.testing.assert_allclose(_y(exact), _y(weno), atol=1e-10)


@pytest.mark.parametrize('equation', [equations.KdVEquation(200), equations.KSEquation(200)])
def test_spectral_exact(equation):
  """integrate_test.py:156-166."""
  times = TIMES if isinstance(equation, equations.KdVEquation) else np.linspace(0, 0.2, 11)
  exact = integrate.integrate_exact(equation, times=times)
  spectral = integrate.integrate_spectral(equation, times=times)
  np.testing.assert_allclose(_y(exact), _y(spectral), atol=1e-10)


@pytest.mark.parametrize('equation,kwargs', [
    (equations.BurgersEquation(200), {}),
    (equations.ConservativeBurgersEquation(200), {}),
    (equations.KdVEquation(200), {}),
    (equations.KSEquation(200), dict(warmup=0.5)),     # reference: warmup=50.0
])
def test_integrate_baseline(equation, kwargs):
  """integrate_test.py:168-184."""
  times = TIMES if not isinstance(equation, equations.KSEquation) else np.linspace(0, 0.2, 11)
  results = integrate.integrate_baseline(equation, times=times, **kwargs)
  y = _y(results)
  assert y.shape == (11, 200)
  _assert_mean_zero(y)


@pytest.mark.parametrize('equation,tol', [
    (equations.GodunovBurgersEquation(200), 1e-3),
    (equations.GodunovKdVEquation(200), 5e-3),
    (equations.GodunovKSEquation(200), 1e-3),
])
def test_integrate_baseline_and_weno_consistency(equation, tol):
  """integrate_test.py:186-199: first-order Godunov baseline vs WENO5."""
  times = TIMES if not isinstance(equation, equations.GodunovKSEquation) else np.linspace(0, 0.2, 11)
  base = _y(integrate.integrate_baseline(equation, times=times))
  weno = _y(integrate.integrate_weno(equation, times=times))
  np.testing.assert_allclose(base, weno, rtol=tol, atol=tol)
