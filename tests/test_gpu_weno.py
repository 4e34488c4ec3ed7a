"""The WENO5 + Godunov-flux exact solver on its own kernels (csrc/rhs_weno.h: one
wavefront per sample, lane = N / 64 consecutive grid points) against
  * the generic kernel it replaces (same expressions in the same order: bit-equal where
    no forcing is involved; forcing as harmonic sums instead of one sine per point and
    mode: float32 rounding apart),
  * the oracle (weno.py:43-123 + integrate.py:124-140 restated in NumPy),
  * SciPy's RK23 driving the same right-hand side one sample at a time (the reference's
    execution shape, integrate.py:143-169): equal evaluation counts, 1e-9.
"""
import numpy as np
import pytest

from helpers import (oracle, baseline_rhs_f64, batch_forcing, measured_bound, random_phase_ic,
                     rel_err)
from ddd1d_amd import equations, integrate, model as model_lib

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _models(cls_name, n, seed=2):
  eq = getattr(equations, cls_name)(n, random_seed=seed)
  lean = model_lib.BaselineModel(eq, 3, weno=True)
  generic = model_lib.BaselineModel(eq, 3, weno=True)
  generic.set_kernel('generic')
  return eq, lean, generic


def _state(eq, batch, n):
  y = random_phase_ic(eq, batch)
  y[1 % batch] = np.where(np.arange(n) < n // 2, 1.0, -0.5)   # a shock: the nonlinear weights switch
  return y


def test_kernel_selection():
  for n, want in ((64, 'valu_f32_weno'), (128, 'valu_f32_weno'), (256, 'valu_f32_weno'),
                  (512, 'valu_f32_weno'), (96, 'generic'), (32, 'generic'), (1024, 'generic')):
    model = model_lib.BaselineModel(equations.GodunovBurgersEquation(n), 3, weno=True)
    assert model.kernel_name == want, (n, model.kernel_name)
  # an explicit choice keeps the generic kernel; 'auto' gives the WENO kernel back
  model = model_lib.BaselineModel(equations.GodunovBurgersEquation(128), 3, weno=True)
  model.set_kernel('generic')
  assert model.kernel_name == 'generic'
  model.set_kernel('auto')
  assert model.kernel_name == 'valu_f32_weno'
  # fixed stencils without the WENO reconstruction are not this kernel's
  plain = model_lib.BaselineModel(equations.GodunovBurgersEquation(128), 3)
  assert plain.kernel_name != 'valu_f32_weno'


@pytest.mark.parametrize('cls_name,n,batch', [
    ('GodunovBurgersEquation', 64, 5), ('GodunovBurgersEquation', 128, 4),
    ('GodunovBurgersEquation', 256, 3), ('GodunovBurgersEquation', 512, 9),
    ('GodunovKdVEquation', 64, 1), ('GodunovKdVEquation', 256, 6),
    ('GodunovKSEquation', 128, 7), ('GodunovKSEquation', 512, 2)])
def test_rhs_unforced_equals_the_generic_kernel_and_the_oracle(cls_name, n, batch):
  eq, lean, generic = _models(cls_name, n)
  y = _state(eq, batch, n)
  got = lean.time_derivative(y, 0.3).cpu().numpy()
  assert lean.kernel_name == 'valu_f32_weno'
  # same expressions, same order, no forcing table set: the same bits
  np.testing.assert_array_equal(got, generic.time_derivative(y, 0.3).cpu().numpy())
  np.testing.assert_array_equal(lean.space_derivatives(y).cpu().numpy(),
                                generic.space_derivatives(y).cpu().numpy())
  spec = lean.spec()
  want = oracle.time_derivative(spec, 0.3, y)
  assert rel_err(got, want) < measured_bound(want, baseline_rhs_f64(spec, y), TOL,
                                             '%s n=%d WENO rhs:' % (cls_name, n), got=got)
  derivs = lean.space_derivatives(y).cpu().numpy()
  np.testing.assert_allclose(derivs[..., 0], np.roll(oracle.weno_reconstruct_left(y), 1, axis=-1),
                             rtol=0, atol=TOL * np.abs(y).max())
  np.testing.assert_allclose(derivs[..., 1], np.roll(oracle.weno_reconstruct_right(y), 1, axis=-1),
                             rtol=0, atol=TOL * np.abs(y).max())


@pytest.mark.parametrize('n,batch', [(64, 6), (128, 3), (512, 5)])
def test_forced_burgers_rhs(n, batch):
  """forcing(t) as harmonic sums: against the generic kernel's one-sine-per-(point, mode)
  form and against the oracle, per-sample forcing."""
  eq, lean, generic = _models('GodunovBurgersEquation', n)
  frc = batch_forcing(batch, seed0=11)
  lean.set_forcing(frc)
  generic.set_forcing(frc)
  y = _state(eq, batch, n)
  spec = lean.spec()
  for t in (0.0, 1.7, 9.99):
    got = lean.time_derivative(y, t).cpu().numpy()
    other = generic.time_derivative(y, t).cpu().numpy()
    want = oracle.time_derivative(spec, t, y, frc)
    assert rel_err(got, other) < TOL, (n, t)
    bound = measured_bound(want, baseline_rhs_f64(spec, y, t, frc), TOL, 'forced WENO rhs n=%d:' % n,
                           got=got)
    assert rel_err(got, want) < bound, (n, t)
  # rk_substep: y_out = y_base + c1 f, acc_out = acc_in + c2 f in one launch
  import torch
  yd = torch.as_tensor(y, device='cuda')
  base = torch.as_tensor(random_phase_ic(eq, batch, seed0=77), device='cuda')
  out = torch.empty_like(yd)
  acc = torch.empty_like(yd)
  lean.rk_substep(0.4, yd, y_base=base, c1=0.25, y_out=out, acc_in=base, c2=-0.5, acc_out=acc)
  f = lean.time_derivative(y, 0.4)
  np.testing.assert_array_equal(out.cpu().numpy(), (base + np.float32(0.25) * f).cpu().numpy())
  np.testing.assert_array_equal(acc.cpu().numpy(), (base + np.float32(-0.5) * f).cpu().numpy())


@pytest.mark.parametrize('cls_name,n,state_dtype', [
    ('GodunovBurgersEquation', 128, 'float32'), ('GodunovBurgersEquation', 512, 'float64'),
    ('GodunovKdVEquation', 64, 'float32'), ('GodunovKSEquation', 256, 'float64')])
def test_fixed_step_integration(cls_name, n, state_dtype):
  eq, lean, generic = _models(cls_name, n)
  batch = 5
  frc = None
  if eq.has_time_dependent_forcing:
    frc = batch_forcing(batch, seed0=3)
    lean.set_forcing(frc)
    generic.set_forcing(frc)
  y0 = random_phase_ic(eq, batch)
  dt = 0.2 * eq.time_step if cls_name != 'GodunovBurgersEquation' else 1e-3
  for scheme in ('midpoint', 'bs3'):
    got = lean.integrate_fixed(y0, 12, dt=dt, scheme=scheme, save_every=4,
                               state_dtype=state_dtype).cpu().numpy()
    other = generic.integrate_fixed(y0, 12, dt=dt, scheme=scheme, save_every=4,
                                    state_dtype=state_dtype).cpu().numpy()
    assert got.shape == (3, batch, n) and np.isfinite(got).all()
    if frc is None:
      np.testing.assert_array_equal(got, other)
    else:
      assert rel_err(got, other) < TOL
    want = oracle.integrate_fixed(lean.spec(), {'midpoint': oracle.SCHEME_MIDPOINT, 'bs3': oracle.SCHEME_BS3}[scheme], 0.0, dt, 12, 4, y0,
                                  forcing=frc, state_dtype={'float32': np.float32, 'float64': np.float64}[state_dtype])
    assert rel_err(got, want) < TOL, (scheme, rel_err(got, want))
    # one launch per substep walks the same kernel family: the same bits (float32 state)
    if state_dtype == 'float32':
      per = lean.integrate_fixed(y0, 12, dt=dt, scheme=scheme, save_every=4,
                                 launch_mode='per_substep').cpu().numpy()
      np.testing.assert_array_equal(got, per)


@pytest.mark.parametrize('cls_name,n', [('GodunovBurgersEquation', 64), ('GodunovBurgersEquation', 512),
                                        ('GodunovKdVEquation', 128)])
def test_adaptive_against_scipy_over_the_same_rhs(cls_name, n, monkeypatch):
  """ddd_integrate_adaptive_f64 on the WENO kernel: every sample equals SciPy's RK23
  driving ddd_time_derivative for that sample alone (equal nfev, 1e-9), and the generic
  adaptive kernel (the controller of rk23.h over the other right-hand side)."""
  monkeypatch.setattr(integrate, 'DEVICE_ODEINT', False)   # per-sample runs: SciPy on the host
  seeds = (3, 8, 11, 12, 20)
  eqs = [getattr(equations, cls_name)(n, random_seed=s) for s in seeds]
  model = model_lib.BaselineModel(eqs[0], 3, weno=True)
  forced = eqs[0].has_time_dependent_forcing
  if forced:
    model.set_forcing(model_lib.forcing_from_equations(eqs))
  y0 = np.stack([random_phase_ic(eq, 1, seed0=50 + s)[0] for eq, s in zip(eqs, seeds)]).astype(np.float64)
  horizon = 0.3 if cls_name == 'GodunovBurgersEquation' else 2e-3
  times = np.linspace(0.0, horizon, 4)
  y, nfev, status = model.integrate_adaptive(y0, times)
  assert model.kernel_name == 'valu_f32_weno'
  y = y.cpu().numpy()
  assert np.isfinite(y).all() and (status.cpu().numpy() == 0).all()
  for b, eq in enumerate(eqs):
    diff = integrate.WENODifferentiator(eq)
    want, want_nfev = integrate.odeint(y0[b], diff, times)
    assert int(nfev[b]) == want_nfev, (b, int(nfev[b]), want_nfev)
    assert rel_err(y[:, b], want) < 1e-9, (b, rel_err(y[:, b], want))
  generic = model_lib.BaselineModel(eqs[0], 3, weno=True)
  generic.set_kernel('generic')
  if forced:
    generic.set_forcing(model_lib.forcing_from_equations(eqs))
  y2, nfev2, status2 = generic.integrate_adaptive(y0, times)
  if forced:
    # the two kernels evaluate forcing(t) differently (harmonic sums / one sine per point and
    # mode: float32 rounding apart), and forced Burgers on the fine grid is stability-limited
    # (rejections): accept / reject decisions amplify that rounding -- the sharp check is the
    # one above, against SciPy over the SAME right-hand side
    np.testing.assert_allclose(nfev.cpu().numpy(), nfev2.cpu().numpy(), rtol=0.05)
    assert rel_err(y, y2.cpu().numpy()) < 5e-2   # (different accept / reject histories: percent level)
  else:
    np.testing.assert_array_equal(nfev.cpu().numpy(), nfev2.cpu().numpy())
    assert rel_err(y, y2.cpu().numpy()) < 1e-9


def test_adaptive_failure_next_to_healthy_samples_and_single_time():
  """A sample that blows up stops with status -1 and NaN rows; its neighbours -- other
  wavefronts of the same workgroup -- finish.  n_times = 1: y0 back, one evaluation."""
  eq = equations.GodunovKdVEquation(64, random_seed=1)
  model = model_lib.BaselineModel(eq, 3, weno=True)
  y0 = random_phase_ic(eq, 6).astype(np.float64)
  y0[2] *= 1e18        # overflows float32 products: NaN right-hand side
  times = np.linspace(0.0, 1e-3, 3)
  y, nfev, status = model.integrate_adaptive(y0, times)
  y, status = y.cpu().numpy(), status.cpu().numpy()
  assert status[2] == -1 and np.isnan(y[1:, 2]).all()
  ok = [0, 1, 3, 4, 5]
  assert (status[ok] == 0).all() and np.isfinite(y[:, ok]).all()
  generic = model_lib.BaselineModel(eq, 3, weno=True)
  generic.set_kernel('generic')
  y2, nfev2, status2 = generic.integrate_adaptive(y0, times)
  np.testing.assert_array_equal(nfev.cpu().numpy(), nfev2.cpu().numpy())
  np.testing.assert_array_equal(status, status2.cpu().numpy())
  y1, nfev1, status1 = model.integrate_adaptive(y0[:3], times[:1])
  np.testing.assert_array_equal(y1.cpu().numpy()[0], y0[:3])
  assert (nfev1.cpu().numpy() == 1).all() and (status1.cpu().numpy() == 0).all()


def test_batch_independence_and_determinism():
  eq = equations.GodunovBurgersEquation(256, random_seed=0)
  model = model_lib.BaselineModel(eq, 3, weno=True)
  batch = 37
  frc = batch_forcing(batch, seed0=5)
  model.set_forcing(frc)
  y0 = random_phase_ic(eq, batch)
  a = model.time_derivative(y0, 0.5).cpu().numpy()
  np.testing.assert_array_equal(a, model.time_derivative(y0, 0.5).cpu().numpy())
  # a sample's result does not depend on the ensemble around it
  model.set_forcing({k: v[10:11] for k, v in frc.items()})
  np.testing.assert_array_equal(a[10:11], model.time_derivative(y0[10:11], 0.5).cpu().numpy())
