"""Fine-grid "exact" solvers on the GPU: WENO5 + Godunov flux (float32 kernel,
integrate.py:124-140) and the spectral method (float64 circulant kernel,
integrate.py:108-121), against the oracle and against trajectories produced
from the reference's own functions (tests/golden/make_golden_exact.py)."""
import os

import numpy as np
import pytest

from helpers import (oracle, baseline_spec, baseline_rhs_f64, measured_bound, random_phase_ic,
                     rel_err)
from ddd1d_amd import equations, integrate, model as model_lib

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))

TOL = 1e-5      # float32 kernels, as everywhere else
TOL64 = 1e-9    # float64 spectral kernel vs the reference's FFT evaluation


@pytest.fixture(scope='module')
def exact():
  return np.load(os.path.join(HERE, 'golden', 'reference_exact_solvers.npz'))


@pytest.mark.parametrize('cls_name,n', [('GodunovBurgersEquation', 64),
                                        ('GodunovBurgersEquation', 512),
                                        ('GodunovKdVEquation', 64),
                                        ('GodunovKSEquation', 96)])
def test_weno_rhs_vs_oracle(cls_name, n):
  eq = getattr(equations, cls_name)(n, random_seed=2)
  model = model_lib.BaselineModel(eq, 3, weno=True)
  spec = model.spec()
  y = random_phase_ic(eq, 5)
  y[1] = np.where(np.arange(n) < n // 2, 1.0, -0.5)     # a shock: nonlinear weights switch
  got = model.time_derivative(y, 0.0).cpu().numpy()
  want = oracle.time_derivative(spec, 0.0, y)
  err = rel_err(got, want)
  print(cls_name, n, 'rel err {:.2e}'.format(err))
  # 1e-5, or 4 x the float32 oracle's measured distance from the all-float64
  # evaluation of the same formulas (KS: u_xxx stencils cancel ~1e3-fold)
  assert err < measured_bound(want, baseline_rhs_f64(spec, y), TOL, cls_name + ' WENO rhs:',
                              got=got)
  derivs = model.space_derivatives(y).cpu().numpy()
  np.testing.assert_allclose(derivs[..., 0], np.roll(oracle.weno_reconstruct_left(y), 1, axis=-1),
                             rtol=0, atol=TOL * np.abs(y).max())
  np.testing.assert_allclose(derivs[..., 1], np.roll(oracle.weno_reconstruct_right(y), 1, axis=-1),
                             rtol=0, atol=TOL * np.abs(y).max())


@pytest.mark.parametrize('cls_name,n,seed', [('GodunovBurgersEquation', 64, 3),
                                             ('GodunovBurgersEquation', 128, 5),
                                             ('GodunovKdVEquation', 64, 1)])
def test_weno_differentiator_vs_reference_trajectories(exact, cls_name, n, seed):
  """integrate_weno (SciPy RK23 + HIP WENO RHS, float32) against float64
  trajectories assembled from reference functions."""
  base = 'weno_odeint/%s/n%d/s%d' % (cls_name, n, seed)
  eq = getattr(equations, cls_name)(n, random_seed=seed)
  diff = integrate.WENODifferentiator(eq)
  probe = exact[base + '/probe']
  want_rhs = exact[base + '/rhs_t0.2_probe']
  got_rhs = diff(0.2, probe)
  # float32 kernel vs the reference's float64 evaluation: 1e-5, or 4 x the float32
  # oracle's own distance from that reference value
  spec = diff.model.spec()
  frc = ({k: v[0] for k, v in model_lib.forcing_from_equations([eq]).items()}
         if eq.has_time_dependent_forcing else None)
  f32_rhs = oracle.time_derivative(spec, 0.2, probe[None], None if frc is None else
                                   {k: v[None] for k, v in frc.items()})[0]
  assert rel_err(got_rhs, want_rhs) < measured_bound(f32_rhs, want_rhs, TOL, base + ' rhs:',
                                                         got=got_rhs)
  ds = integrate.integrate_weno(eq, times=exact[base + '/times'])
  got = np.asarray(ds.data_vars['y'][1] if isinstance(ds.data_vars['y'], tuple)
                   else ds['y'].data)
  want = exact[base + '/y']
  print(base, 'trajectory rel err {:.2e}'.format(rel_err(got, want)),
        'nfev', int(np.asarray(ds.coords['num_evals'])), int(exact[base + '/nfev']))
  # trajectory: the float32 right-hand side under SciPy's adaptive RK23 against the
  # float64 reference run; floor = the float32 ORACLE under the same SciPy run
  f32_run, f32_nfev = oracle.odeint_rk23(spec, eq.initial_value(), exact[base + '/times'], frc)
  assert rel_err(got, want) < measured_bound(f32_run, want, TOL, base + ' trajectory:')
  # mean conservation of the flux form (integrate_test.py:101-104)
  np.testing.assert_allclose(got.mean(axis=1), got[0].mean(), atol=1e-3)


def test_best_weno_baseline_and_exact_differentiator():
  eq = equations.GodunovBurgersEquation(128, random_seed=4)
  best = model_lib.BaselineModel(eq, accuracy_order=None)
  assert best.kernel_name == 'valu_f32_weno'   # (rhs_weno.h; 'generic' on grids it does not carry)
  explicit = model_lib.BaselineModel(eq, 3, weno=True)
  y = random_phase_ic(eq, 3)
  np.testing.assert_array_equal(best.time_derivative(y, 0.0).cpu().numpy(),
                                explicit.time_derivative(y, 0.0).cpu().numpy())
  diff = integrate.exact_differentiator(eq)
  assert isinstance(diff, integrate.WENODifferentiator)
  assert isinstance(integrate.exact_differentiator(equations.KdVEquation(64)),
                    integrate.SpectralDifferentiator)
  with pytest.raises(TypeError, match='exact equation'):
    integrate.exact_differentiator(equations.ConservativeBurgersEquation(64))


@pytest.mark.parametrize('cls_name,n', [('KdVEquation', 64), ('KSEquation', 128),
                                        ('BurgersEquation', 64)])
def test_spectral_differentiator_vs_reference(exact, cls_name, n):
  eq = getattr(equations, cls_name)(n, random_seed=3)
  diff = integrate.SpectralDifferentiator(eq)
  y = exact['spectral_rhs/%s/n%d/y' % (cls_name, n)]
  want = exact['spectral_rhs/%s/n%d/out_t0.3' % (cls_name, n)]
  got = diff(0.3, y)
  assert got.dtype == np.float64
  err = rel_err(got, want)
  print(cls_name, n, 'spectral rhs rel err {:.2e}'.format(err))
  assert err < TOL64


def test_spectral_batched_and_rfft_convention():
  eq = equations.KSEquation(256, random_seed=0)
  y = random_phase_ic(eq, 33).astype(np.float64)
  spec = eq.kernel_spec()
  for convention in ('fftpack', 'rfft'):
    model = model_lib.SpectralModel(eq, convention=convention)
    got = model.time_derivative(y).cpu().numpy()
    if convention == 'fftpack':
      want = oracle.spectral_time_derivative(spec['equation'], y, spec['derivative_orders'],
                                             spec['period'], spec['eta'], spec['dx'])
    else:
      derivs = np.stack([oracle.spectral_derivative(y, order, spec['period'])
                         for order in spec['derivative_orders']], axis=-1)
      want = oracle.equation_of_motion(spec['equation'], y, derivs, spec['eta'], spec['dx'])
    assert rel_err(got, want) < TOL64
  # float32 entry points refuse a spectral model, and vice versa
  from ddd1d_amd import _lib
  with pytest.raises(_lib.DDDError, match='f64'):
    model_lib._DeviceModel.time_derivative(model, y.astype(np.float32))
  base = model_lib.BaselineModel(equations.KdVEquation(64), 1)
  with pytest.raises(_lib.DDDError, match='spectral model'):
    model_lib.SpectralModel.time_derivative(base, random_phase_ic(base.equation, 2).astype(np.float64))


@pytest.mark.parametrize('cls_name,n,batch', [('KdVEquation', 512, 9), ('KSEquation', 512, 5),
                                              ('KSEquation', 1024, 3), ('KdVEquation', 2048, 2),
                                              ('BurgersEquation', 512, 4)])
def test_spectral_fft_mode_vs_reference_operator(cls_name, n, batch):
  """N >= 512: the spectral right-hand side runs as an in-LDS float64 FFT
  (rhs_spectral.h: eval_points_fft; multipliers = DFT of the very circulant kernels the
  smaller grids apply directly) -- against scipy.fftpack.diff / numpy.fft on the host,
  both conventions.  Bound: float64 rounding x the largest multiplier of the equation's
  highest derivative (the reference's own FFT has that noise too), far below 1e-9."""
  eq = getattr(equations, cls_name)(n, random_seed=3)
  y = random_phase_ic(eq, batch).astype(np.float64)
  spec = eq.kernel_spec()
  kmax = 2 * np.pi * (n // 2) / spec['period']
  bound = max(TOL64, 200 * np.finfo(np.float64).eps * kmax ** max(spec['derivative_orders']))
  for convention in ('fftpack', 'rfft'):
    model = model_lib.SpectralModel(eq, convention=convention)
    got = model.time_derivative(y).cpu().numpy()
    if convention == 'fftpack':
      want = oracle.spectral_time_derivative(spec['equation'], y, spec['derivative_orders'],
                                             spec['period'], spec['eta'], spec['dx'])
    else:
      derivs = np.stack([oracle.spectral_derivative(y, order, spec['period'])
                         for order in spec['derivative_orders']], axis=-1)
      want = oracle.equation_of_motion(spec['equation'], y, derivs, spec['eta'], spec['dx'])
    err = rel_err(got, want)
    print(cls_name, n, convention, 'fft-mode rhs rel err {:.2e} (bound {:.1e})'.format(err, bound))
    assert err < bound


def test_spectral_fft_mode_adaptive_exact_solver():
  """integrate_exact on a 512-point KdV grid (the exact grid of integrate_test-style runs
  at resample factor 8): the batched device solver in FFT mode against SciPy's RK23 over
  the host restatement of SpectralDifferentiator -- equal evaluation counts, 1e-9.  With a
  SATURATED controller (max_step below the stability limit of u_xxx on dx = 1/16): at the
  reference's max_step this grid sits on the stability boundary, where accept / reject
  decisions amplify rounding noise (2 552 vs 2 576 evaluations between two float64
  evaluations of the same operator, gpurun_out/r5d) -- as for KS N = 256."""
  import scipy.integrate
  eq = equations.KdVEquation(512, random_seed=2)
  model = model_lib.SpectralModel(eq)
  spec = eq.kernel_spec()
  y0 = np.stack([eq.initial_value(), equations.KdVEquation(512, random_seed=5).initial_value()])
  times = np.linspace(0.0, 5e-4, 3)
  max_step = 5e-6    # (k_max^3 = 1.3e5: the explicit RK23 stability limit is ~1.3e-5)
  y, nfev, status = model.integrate_adaptive(y0, times, max_step=max_step)
  y = y.cpu().numpy()
  assert (status.cpu().numpy() == 0).all()
  rhs = lambda t, v: oracle.spectral_time_derivative(
      spec['equation'], v, spec['derivative_orders'], spec['period'], spec['eta'], spec['dx'])
  for b in range(2):
    sol = scipy.integrate.solve_ivp(rhs, (times[0], times[-1]), y0[b], t_eval=times,
                                    max_step=max_step, method='RK23')
    assert sol.nfev == int(nfev[b]), (b, sol.nfev, int(nfev[b]))
    assert rel_err(y[:, b], sol.y.T) < 1e-9


def test_spectral_fixed_step_f64_vs_oracle():
  """ddd_integrate_fixed_f64 on a spectral model: BS3 in float64 end to end."""
  eq = equations.KdVEquation(64, random_seed=1)
  model = model_lib.SpectralModel(eq)
  spec = eq.kernel_spec()
  y0 = random_phase_ic(eq, 4).astype(np.float64)
  dt, steps = 1e-4, 50
  got = model.integrate_fixed(y0, steps, dt=dt, scheme='bs3', save_every=25).cpu().numpy()
  rhs = lambda y: oracle.spectral_time_derivative(
      spec['equation'], y, spec['derivative_orders'], spec['period'], spec['eta'], spec['dx'])
  y = y0.copy()
  want = []
  for step in range(steps):
    k1 = rhs(y); k2 = rhs(y + 0.5 * dt * k1); k3 = rhs(y + 0.75 * dt * k2)
    y = y + dt * (2 / 9 * k1 + 1 / 3 * k2 + 4 / 9 * k3)
    if (step + 1) % 25 == 0:
      want.append(y.copy())
  assert got.dtype == np.float64
  assert rel_err(got, np.stack(want)) < TOL64


def test_integrate_exact_with_warmup_and_filtering(exact):
  """integrate.integrate_exact end to end (warm-up on the exact equation,
  periodic smoothing filter) against the reference's own runs."""
  eq = equations.KdVEquation(64, random_seed=1)
  ds = integrate.integrate_exact(eq, times=np.linspace(0, 0.1, 3), warmup=0.05)
  got = _y(ds)
  np.testing.assert_allclose(np.asarray(_coord(ds, 'time')), exact['exact/kdv64_warmup/times'])
  assert rel_err(got, exact['exact/kdv64_warmup/y']) < 1e-6
  eq = equations.KSEquation(64, random_seed=2)
  ds = integrate.integrate_exact(eq, times=np.linspace(0, 0.04, 5), warmup=0.02,
                                 filter_interval=0.01)
  assert rel_err(_y(ds), exact['exact/ks64_filtered/y']) < 1e-6


def test_integrate_with_warmup_for_a_coarse_baseline():
  """integrate.integrate(warmup > 0): WENO warm-up on the fine Godunov grid,
  resampled to the coarse conservative grid, then the coarse run
  (integrate.py:256-268); t = warmup row equals the resampled warm-up state."""
  fine, coarse = equations.from_hparams(_burgers_hparams(), random_seed=5)
  ds = integrate.integrate_baseline(coarse, times=np.linspace(0, 0.1, 3), warmup=0.2)
  y = _y(ds)
  assert y.shape == (3, coarse.grid.solution_num_points) and np.isfinite(y).all()
  exact_eq = coarse.to_exact()
  warm, _ = integrate.odeint(exact_eq.initial_value(), integrate.exact_differentiator(exact_eq),
                             np.array([0, 0.2]))
  np.testing.assert_allclose(y[0], coarse.grid.resample(warm[-1]), rtol=0, atol=1e-6)
  np.testing.assert_allclose(np.asarray(_coord(ds, 'time')), 0.2 + np.linspace(0, 0.1, 3))


def _burgers_hparams():
  import json
  import ddd1d_amd
  return ddd1d_amd.create_hparams('burgers', conservative=True, resample_factor=4,
                                  equation_kwargs=json.dumps({'num_points': 128}))


def _y(ds):
  v = ds.data_vars['y']
  return np.asarray(v[1] if isinstance(v, tuple) else v)


def _coord(ds, name):
  v = ds.coords[name]
  return v[1] if isinstance(v, tuple) else v


def test_best_baseline_for_spectral_equations():
  """PolynomialDifferentiator(accuracy_order=None) on a spectral-exact equation:
  duckarray.spectral_derivative (rfft form, model.py:78-80) through the
  float64 kernel."""
  eq = equations.KdVEquation(64, random_seed=6)
  diff = integrate.PolynomialDifferentiator(eq, accuracy_order=None)
  y = random_phase_ic(eq, 1)[0].astype(np.float64)
  spec = eq.kernel_spec()
  derivs = np.stack([oracle.spectral_derivative(y, order, spec['period'])
                     for order in spec['derivative_orders']], axis=-1)
  want = oracle.equation_of_motion(spec['equation'], y, derivs, spec['eta'], spec['dx'])
  assert rel_err(diff(0.0, y), want) < TOL64
  named = diff.calculate_space_derivatives(y)
  assert sorted(named) == sorted(eq.DERIVATIVE_NAMES)
  np.testing.assert_allclose(named['u_x'], derivs[..., 0], rtol=0, atol=1e-12)


def test_integrate_exact_baseline_and_model():
  """integrate.py:344-396 end to end: WENO exact solution on the fine grid,
  then baseline and learned model from its resampled first row."""
  from helpers import make_model
  model = make_model('burgers', True, num_points=32, resample_factor=4)
  times = np.linspace(0, 0.2, 3)
  ds = integrate.integrate_exact_baseline_and_model(None, hparams=model.hparams,
                                                    random_seed=3, times=times,
                                                    warmup=0.1, model=model)
  exact = _y_named(ds, 'y_exact'); base = _y_named(ds, 'y_baseline'); learned = _y_named(ds, 'y_model')
  assert exact.shape == (3, 128) and base.shape == (3, 32) and learned.shape == (3, 32)
  assert np.isfinite(exact).all() and np.isfinite(base).all() and np.isfinite(learned).all()
  # both coarse runs start from the block-averaged exact state
  np.testing.assert_allclose(base[0], exact[0].reshape(32, 4).mean(axis=1), atol=1e-12)
  np.testing.assert_array_equal(base[0], learned[0])
  # the baseline tracks the resampled exact solution closely over this horizon
  assert np.abs(base[-1] - exact[-1].reshape(32, 4).mean(axis=1)).max() < 0.05
  np.testing.assert_allclose(np.asarray(_coord(ds, 'time')), 0.1 + times)


def _y_named(ds, name):
  v = ds.data_vars[name]
  return np.asarray(v[1] if isinstance(v, tuple) else v)


# ---------------------------------------------------------------------------
# The exact spectral solver entirely on the device: batched adaptive RK23 over
# the float64 right-hand side + the smoothing filter as a circulant kernel
# ---------------------------------------------------------------------------
def test_smoothing_filter_on_device_vs_reference(exact):
  """duckarray.smoothing_filter (duckarray.py:116-128) as ddd_circulant_apply_f64
  against the reference's own outputs."""
  from ddd1d_amd import _lib, duckarray
  x = exact['spectral_derivative/x']
  impulse = np.zeros(x.shape[-1])
  impulse[0] = 1.0
  for order in (2, 3, 4):
    kernel = duckarray.smoothing_filter(impulse, order=order)
    got = _lib.circulant_apply(kernel, x).cpu().numpy()
    want = exact['smoothing_filter/order%d' % order]
    assert got.shape == want.shape
    assert rel_err(got, want) < 1e-12, order
  # batches of trajectories [time, sample, x], in one launch
  stacked = np.stack([x, 2 * x, -x])
  got = _lib.circulant_apply(duckarray.smoothing_filter(impulse, order=2), stacked).cpu().numpy()
  assert rel_err(got, duckarray.smoothing_filter(stacked, order=2)) < 1e-12


def test_integrate_exact_batch_vs_reference_and_per_sample(exact, monkeypatch):
  """integrate_exact_batch: every sample with its own RK23 controller over the
  float64 spectral right-hand side, warm-up and periodic filtering on the device.
  The sample the reference fixture was generated for reproduces the reference's
  trajectory AND its evaluation count; every sample equals the one-sample host
  SciPy run over the same kernel (nfev equal, 1e-9)."""
  monkeypatch.setattr(integrate, 'DEVICE_ODEINT', False)   # per-sample runs: SciPy on the host
  times = np.linspace(0, 0.1, 3)
  eqs = [equations.KdVEquation(64, random_seed=s) for s in (1, 5, 9, 12)]
  ds = integrate.integrate_exact_batch(eqs, times=times, warmup=0.05)
  y = _y(ds)
  nfev = np.asarray(_coord(ds, 'num_evals'))
  assert y.shape == (4, 3, 64) and y.dtype == np.float64
  np.testing.assert_allclose(np.asarray(_coord(ds, 'time')), exact['exact/kdv64_warmup/times'])
  assert rel_err(y[0], exact['exact/kdv64_warmup/y']) < 1e-6
  assert int(nfev[0]) == int(exact['exact/kdv64_warmup/nfev'])
  for b in (1, 3):
    one = integrate.integrate_exact(eqs[b], times=times, warmup=0.05)
    assert int(np.asarray(_coord(one, 'num_evals'))) == int(nfev[b])
    assert rel_err(y[b], _y(one)) < 1e-9
  # periodic filtering: segments + circulant filter, nothing leaves the device
  times = np.linspace(0, 0.04, 5)
  eqs = [equations.KSEquation(64, random_seed=s) for s in (2, 3)]
  ds = integrate.integrate_exact_batch(eqs, times=times, warmup=0.02, filter_interval=0.01)
  y = _y(ds)
  nfev = np.asarray(_coord(ds, 'num_evals'))
  assert rel_err(y[0], exact['exact/ks64_filtered/y']) < 1e-6
  assert int(nfev[0]) == int(exact['exact/ks64_filtered/nfev'])
  one = integrate.integrate_exact(eqs[1], times=times, warmup=0.02, filter_interval=0.01)
  assert int(np.asarray(_coord(one, 'num_evals'))) == int(nfev[1])
  assert rel_err(y[1], _y(one)) < 1e-9


def test_integrate_exact_batch_weno_burgers(monkeypatch):
  """The exact Burgers solver (WENO5 + Godunov flux, per-seed forcing) for a batch
  on the device: every sample equals its one-sample integrate_exact run over the
  same kernel (equal nfev, 1e-9), which the reference fixtures pin."""
  monkeypatch.setattr(integrate, 'DEVICE_ODEINT', False)   # per-sample runs: SciPy on the host
  times = np.linspace(0, 0.3, 4)
  eqs = [equations.BurgersEquation(128, random_seed=s) for s in (3, 8, 11)]
  ds = integrate.integrate_exact_batch(eqs, times=times, warmup=0.1)
  y = _y(ds)
  nfev = np.asarray(_coord(ds, 'num_evals'))
  assert y.shape == (3, 4, 128) and np.isfinite(y).all() and np.abs(y).max() > 1e-3
  for b in range(3):
    one = integrate.integrate_exact(eqs[b], times=times, warmup=0.1)
    assert int(np.asarray(_coord(one, 'num_evals'))) == int(nfev[b])
    assert rel_err(y[b], _y(one)) < 1e-9


def test_spectral_adaptive_large_grid_and_failure(monkeypatch):
  """N = 512 (two grid points per thread) and N = 2048 (eight); a sample that
  blows up stops with status -1 and NaN rows while its neighbours finish."""
  monkeypatch.setattr(integrate, 'DEVICE_ODEINT', False)
  for n, horizon in ((512, 2e-3), (2048, 2e-5)):
    eq = equations.KdVEquation(n, random_seed=3)
    model = model_lib.SpectralModel(eq)
    y0 = np.stack([eq.initial_value(), 0.5 * eq.initial_value()])
    times = np.linspace(0, horizon, 3)
    y, nfev, status = model.integrate_adaptive(y0, times)
    diff = integrate.SpectralDifferentiator(eq)
    for b in range(2):
      want, want_nfev = integrate.odeint(y0[b], diff, times)
      assert int(nfev[b]) == want_nfev and int(status[b]) == 0, (n, b, int(nfev[b]), want_nfev)
      assert rel_err(y[:, b].cpu().numpy(), want) < 1e-9


def test_spectral_adaptive_refuses_the_forced_family():
  """The spectral adaptive kernel has no forcing term: a Burgers spectral model
  (finalize_time_derivative adds forcing(t), equations.py:276-277) is refused by
  ddd_integrate_adaptive_f64 instead of silently integrating the unforced
  equation; KdV / KS (no forcing) are accepted."""
  from ddd1d_amd import _lib
  eq = equations.BurgersEquation(64, random_seed=1)
  model = model_lib.SpectralModel(eq, convention='fftpack')
  y0 = eq.initial_value()[None].astype(np.float64)
  with pytest.raises(_lib.DDDError, match='no forcing term'):
    model.integrate_adaptive(y0, np.linspace(0.0, 0.01, 3))
  kdv = equations.KdVEquation(64, random_seed=1)
  y, nfev, status = model_lib.SpectralModel(kdv, convention='fftpack').integrate_adaptive(
      kdv.initial_value()[None].astype(np.float64), np.linspace(0.0, 1e-4, 3))
  assert int(status[0]) == 0 and np.isfinite(y.cpu().numpy()).all()
