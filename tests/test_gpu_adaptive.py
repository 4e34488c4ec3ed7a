"""On-device batched adaptive RK23 (`ddd_integrate_adaptive_f64`) against the
reference's per-sample SciPy loop (integrate.odeint, integrate.py:143-169;
scripts/run_evaluation.py:152-174): per-sample nfev EQUAL to solve_ivp's and
trajectories within 1e-5, for the three equation families, one-wave and
four-wave geometries, saturated and controller-limited step sizes, and the
failure path (NaN rows + status -1).

The checker is `oracle.odeint_rk23` = the installed SciPy over the NumPy
right-hand side -- SciPy is the reference's own third-party integrator."""
import numpy as np
import pytest

from helpers import batch_forcing, make_model, oracle, random_phase_ic, rel_err
from ddd1d_amd import integrate, model as model_lib

pytestmark = pytest.mark.gpu

TOL = 1e-5   # north_star: fp32 trajectories within 1e-5 rel of the reference


def _one(forcing, b):
  return None if forcing is None else {k: np.asarray(v)[b] for k, v in forcing.items()}


def _check(model, y0, times, forcing=None, max_step=0.01, tol=TOL, samples=None,
           **solver):
  """Device controller vs SciPy per sample.  Returns (nfev, status, mismatches)."""
  if forcing is not None:
    model.set_forcing(forcing)
  y, nfev, status = model.integrate_adaptive(y0, times, max_step=max_step, **solver)
  y = y.cpu().numpy()
  nfev = nfev.cpu().numpy()
  status = status.cpu().numpy()
  assert y.shape == (len(times),) + y0.shape and y.dtype == np.float64
  spec = model.spec()
  bad = []
  worst = 0.0
  for b in (range(y0.shape[0]) if samples is None else samples):
    want, want_nfev = _scipy(spec, y0[b], times, _one(forcing, b), max_step, **solver)
    if want_nfev != nfev[b]:
      bad.append((b, int(nfev[b]), want_nfev))
      continue
    finite = np.isfinite(want).all(axis=1)
    np.testing.assert_array_equal(np.isfinite(y[:, b]).all(axis=1), finite)
    assert (status[b] == 0) == bool(finite.all())
    err = rel_err(y[finite, b], want[finite])
    worst = max(worst, err)
    assert err < tol, (b, err)
  return nfev, status, bad, worst


def _scipy(spec, y0, times, forcing, max_step, rtol=1e-3, atol=1e-6):
  import scipy.integrate
  one = None if forcing is None else {k: np.asarray(v)[None] for k, v in forcing.items()}

  def fun(t, y):
    return oracle.time_derivative(spec, t, y[None, :], one)[0]
  sol = scipy.integrate.solve_ivp(fun, (times[0], times[-1]), np.asarray(y0, np.float64),
                                  t_eval=times, max_step=max_step, method='RK23',
                                  rtol=rtol, atol=atol)
  y = sol.y.T
  if len(times) - y.shape[0]:
    y = np.pad(y, ((0, len(times) - y.shape[0]), (0, 0)), mode='constant',
               constant_values=np.nan)
  return y, sol.nfev


@pytest.mark.parametrize('equation,conservative', [
    ('burgers', True), ('burgers', False), ('kdv', True), ('kdv', False),
    ('ks', True), ('ks', False)])
def test_reference_settings_n64(equation, conservative):
  """max_step = 0.01, rtol 1e-3, atol 1e-6 (integrate.py:154): B = 64 samples in
  one launch, each with its own controller, forcing and nfev."""
  batch = 64
  model = make_model(equation, conservative, num_points=64, resample_factor=4)
  assert model.kernel_name == 'mfma_f32_r64'
  scale = 0.3 if equation == 'burgers' else 1.0
  y0 = (scale * random_phase_ic(model.equation, batch)).astype(np.float64)
  forcing = batch_forcing(batch) if equation == 'burgers' else None
  times = np.linspace(0.0, 0.2, 5)
  nfev, status, bad, worst = _check(model, y0, times, forcing)
  print(equation, conservative, 'nfev', nfev.min(), nfev.max(), 'worst rel err {:.1e}'.format(worst))
  assert not bad, bad
  assert (status == 0).all()


@pytest.mark.parametrize('equation,num_points,max_step', [
    ('burgers', 64, np.inf), ('burgers', 32, np.inf), ('burgers', 16, 0.05),
    ('kdv', 64, np.inf), ('ks', 64, np.inf)])
def test_controller_limited_steps(equation, num_points, max_step):
  """No max_step ceiling: every step size comes from the error controller
  (initial step selection, growth, rejections), several samples per wavefront
  for N < 64."""
  batch = 40   # not a multiple of the samples per workgroup for N = 16
  model = make_model(equation, True, num_points=num_points, resample_factor=4)
  scale = 0.5 if equation == 'burgers' else 1.0
  y0 = (scale * random_phase_ic(model.equation, batch)).astype(np.float64)
  forcing = batch_forcing(batch) if equation == 'burgers' else None
  times = np.array([0.0, 0.013, 0.1, 0.25, 0.4])   # not aligned with any step
  nfev, status, bad, worst = _check(model, y0, times, forcing, max_step=max_step)
  print(equation, num_points, 'nfev', nfev.min(), nfev.max(), 'worst {:.1e}'.format(worst),
        'mismatched', bad)
  assert len(np.unique(nfev)) > 1, 'samples should need different numbers of steps'
  assert len(bad) <= batch // 20, bad   # see test docstring of test_ks256
  assert (status == 0).all()


def test_ks256_four_wave_groups():
  """BASELINE configs[3] geometry (KS N = 256, one sample per 256-row
  workgroup): the step is stability-limited (dt ~ 1e-4 << max_step), so the
  controller rejects and regrows continuously."""
  batch = 8
  model = make_model('ks', True, num_points=256, resample_factor=2)
  assert model.kernel_name == 'mfma_f32_r256'
  y0 = random_phase_ic(model.equation, batch).astype(np.float64)
  times = np.linspace(0.0, 0.02, 5)
  nfev, status, bad, worst = _check(model, y0, times)
  print('ks256 nfev', nfev, 'worst {:.1e}'.format(worst), 'mismatched', bad)
  assert nfev.min() > 100
  assert not bad, bad


def test_n128_two_samples_per_group_and_non_power_of_two():
  for num_points in (128, 96):
    model = make_model('kdv', True, num_points=num_points, resample_factor=2)
    assert model.kernel_name == 'mfma_f32_r256'
    y0 = random_phase_ic(model.equation, 5).astype(np.float64)
    times = np.linspace(0.0, 0.05, 3)
    nfev, status, bad, worst = _check(model, y0, times)
    print('kdv', num_points, nfev, 'worst {:.1e}'.format(worst))
    assert not bad, bad


def test_fixed_stencil_baseline_and_runtime_kernels():
  """PolynomialDifferentiator models (integrate_baseline, integrate.py:296-308)
  and a non-default net run on the run-time-parameterised adaptive kernel."""
  eq = make_model('burgers', True, num_points=32, resample_factor=4).equation
  base = model_lib.BaselineModel(eq, accuracy_order=1)
  y0 = (0.5 * random_phase_ic(eq, 6)).astype(np.float64)
  times = np.linspace(0.0, 0.3, 4)
  forcing = batch_forcing(6)
  base.set_forcing(forcing)
  y, nfev, status = base.integrate_adaptive(y0, times)
  spec = base.spec()
  for b in range(6):
    want, want_nfev = _scipy(spec, y0[b], times, _one(forcing, b), 0.01)
    assert want_nfev == int(nfev[b])
    assert rel_err(y[:, b].cpu().numpy(), want) < TOL
  deep = make_model('kdv', False, num_points=64, resample_factor=4, num_layers=4,
                    nonlinearity='tanh')
  y0 = random_phase_ic(deep.equation, 4).astype(np.float64)
  nfev, status, bad, worst = _check(deep, y0, np.linspace(0, 0.1, 3))
  assert not bad, bad


def test_failure_path_nan_rows_and_status():
  """A state that blows up: SciPy shrinks the step until it is below 10 ulp(t),
  returns status -1 and integrate.odeint NaN-pads the rows not reached
  (integrate.py:161-167).  Other samples of the same launch are unaffected."""
  model = make_model('burgers', False, num_points=64, resample_factor=4)
  y0 = (0.3 * random_phase_ic(model.equation, 4)).astype(np.float64)
  y0[2] *= 400.0   # anti-diffusive learned stencils at this amplitude: finite-time blow-up
  times = np.linspace(0.0, 0.5, 6)
  forcing = batch_forcing(4)
  model.set_forcing(forcing)
  y, nfev, status = model.integrate_adaptive(y0, times)
  y, nfev, status = y.cpu().numpy(), nfev.cpu().numpy(), status.cpu().numpy()
  want, want_nfev = _scipy(model.spec(), y0[2], times, _one(forcing, 2), 0.01)
  print('blow-up sample: nfev', nfev[2], 'scipy', want_nfev, 'status', status)
  if np.isnan(want).any():
    assert status[2] == -1
    np.testing.assert_array_equal(np.isnan(y[:, 2]).all(axis=1), np.isnan(want).all(axis=1))
    assert abs(int(nfev[2]) - want_nfev) <= 0.02 * want_nfev   # chaotic tail before the failure
  for b in (0, 1, 3):
    want_b, nfev_b = _scipy(model.spec(), y0[b], times, _one(forcing, b), 0.01)
    assert status[b] == 0 and nfev[b] == nfev_b and rel_err(y[:, b], want_b) < TOL


def test_attempt_limit_and_argument_checks():
  from ddd1d_amd import _lib
  model = make_model('kdv', True, num_points=64, resample_factor=4)
  y0 = random_phase_ic(model.equation, 3).astype(np.float64)
  y, nfev, status = model.integrate_adaptive(y0, [0.0, 1.0], max_attempts=5)
  assert (status.cpu().numpy() == -2).all() and (nfev.cpu().numpy() == 2 + 3 * 5).all()
  assert np.isnan(y[1].cpu().numpy()).all() and np.array_equal(y[0].cpu().numpy(), y0)
  with pytest.raises(_lib.DDDError):
    model.integrate_adaptive(y0, [0.0, 0.0])
  with pytest.raises(_lib.DDDError):
    model.integrate_adaptive(y0, [0.0, 1.0], rtol=0.0)
  # a single output time: nothing to integrate, one evaluation (RungeKutta.__init__)
  y, nfev, status = model.integrate_adaptive(y0, [0.25])
  assert np.array_equal(y[0].cpu().numpy(), y0) and (nfev.cpu().numpy() == 1).all()
  generic = make_model('kdv', True, num_points=64, resample_factor=4, kernel_size=3)
  assert generic.kernel_name == 'generic'
  with pytest.raises(_lib.DDDError):
    generic.integrate_adaptive(y0, [0.0, 1.0])


def test_integrate_batch_adaptive_dataset():
  model = make_model('burgers', True, num_points=64, resample_factor=4)
  y0 = 0.3 * random_phase_ic(model.equation, 5)
  times = np.linspace(0, 0.1, 3)
  ds = integrate.integrate_batch(model, y0, times, forcing=batch_forcing(5), adaptive=True)
  y = integrate._dataset_array(ds, 'y')
  assert y.shape == (5, 3, 64) and y.dtype == np.float64
  np.testing.assert_array_equal(y[:, 0], y0.astype(np.float64))
  evals = np.asarray(integrate._dataset_coord(ds, 'num_evals'))
  assert evals.shape == (5,) and (evals == 2 + 3 * 10).all()
