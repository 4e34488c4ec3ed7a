"""Oracle restatements and host mirrors of the fine-grid "exact" solver pieces
(weno.py, duckarray spectral helpers, SpectralDifferentiator) against fixtures
produced by the reference itself (tests/golden/make_golden_exact.py)."""
import os

import numpy as np
import pytest

from helpers import oracle, baseline_spec
from ddd1d_amd import duckarray, equations, model as model_lib, weno

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope='module')
def exact():
  return np.load(os.path.join(HERE, 'golden', 'reference_exact_solvers.npz'))


def test_weno_reconstructions_match_reference(exact):
  u = exact['weno/u']
  # oracle restatement: bit-identical in float64
  np.testing.assert_array_equal(oracle.weno_reconstruct_left(u), exact['weno/left'])
  np.testing.assert_array_equal(oracle.weno_reconstruct_right(u), exact['weno/right'])
  np.testing.assert_array_equal(oracle.weno_omega(u), exact['weno/omega'])
  # host mirror (different evaluation order): float64 rounding
  np.testing.assert_allclose(weno.reconstruct_left(u), exact['weno/left'], rtol=0, atol=1e-14)
  np.testing.assert_allclose(weno.reconstruct_right(u), exact['weno/right'], rtol=0, atol=1e-14)
  np.testing.assert_allclose(weno.calculate_omega(u), exact['weno/omega'], rtol=0, atol=1e-15)


def test_weno_is_fifth_order_on_smooth_data_and_bounded_at_shocks():
  """weno_test.py-style properties: design order for smooth data; no new
  extrema across a discontinuity (essentially non-oscillatory)."""
  errs = []
  for n in (32, 64):
    x = 2 * np.pi * np.arange(n) / n
    dx = x[1] - x[0]
    # cell averages of sin: exact edge value at x + dx/2 is sin(x + dx/2)
    cell = (np.cos(x - dx / 2) - np.cos(x + dx / 2)) / dx
    errs.append(np.abs(weno.reconstruct_left(cell) - np.sin(x + dx / 2)).max())
  assert errs[0] / errs[1] > 2 ** 4.5   # ~ order 5
  step = np.where(np.arange(64) < 32, 1.0, 0.0)
  for rec in (weno.reconstruct_left(step), weno.reconstruct_right(step)):
    assert rec.min() > -1e-3 and rec.max() < 1 + 1e-3


def test_spectral_helpers_match_reference(exact):
  x = exact['spectral_derivative/x']
  for order in (1, 2, 3, 4):
    want = exact['spectral_derivative/order%d_period7' % order]
    np.testing.assert_array_equal(oracle.spectral_derivative(x, order, 7.0), want)
    np.testing.assert_array_equal(duckarray.spectral_derivative(x, order, 7.0), want)
  for order in (2, 3, 4):
    want = exact['smoothing_filter/order%d' % order]
    np.testing.assert_array_equal(oracle.smoothing_filter(x, order=order), want)
    np.testing.assert_array_equal(duckarray.smoothing_filter(x, order=order), want)
  with pytest.raises(ValueError, match='even length'):
    duckarray.spectral_derivative(np.zeros(7))
  with pytest.raises(ValueError, match='even length'):
    duckarray.smoothing_filter(np.zeros(7))


@pytest.mark.parametrize('cls_name,n', [('KdVEquation', 64), ('KSEquation', 128),
                                        ('BurgersEquation', 64)])
def test_spectral_rhs_oracle_and_circulant_form(exact, cls_name, n):
  """SpectralDifferentiator (integrate.py:113-121): the oracle reproduces the
  reference's output, and the circulant kernels the GPU uses are the same
  operator (checked here in NumPy, no GPU needed)."""
  eq = getattr(equations, cls_name)(n, random_seed=3)
  y = exact['spectral_rhs/%s/n%d/y' % (cls_name, n)]
  want = exact['spectral_rhs/%s/n%d/out_t0.3' % (cls_name, n)]
  spec = eq.kernel_spec()
  y_t = oracle.spectral_time_derivative(spec['equation'], y, spec['derivative_orders'],
                                        spec['period'], spec['eta'], spec['dx'])
  got = eq.finalize_time_derivative(0.3, y_t)
  np.testing.assert_allclose(got, want, rtol=0, atol=1e-12 * np.abs(want).max())
  model = model_lib.SpectralModel(eq)            # host part only: no handle yet
  idx = (np.arange(n)[:, None] - np.arange(n)[None, :]) % n
  derivs = np.stack([model.kernels[d][idx] @ y for d in range(len(model.kernels))], axis=-1)
  circ = oracle.equation_of_motion(spec['equation'], y, derivs, spec['eta'], spec['dx'])
  np.testing.assert_allclose(eq.finalize_time_derivative(0.3, circ), want, rtol=0,
                             atol=1e-9 * np.abs(want).max())


def test_weno_oracle_rhs_matches_reference_pieces(exact):
  """The float64 oracle path of WENODifferentiator against the RHS assembled
  from reference functions only."""
  for cls_name, n, seed in (('GodunovBurgersEquation', 64, 3),
                            ('GodunovBurgersEquation', 128, 5),
                            ('GodunovKdVEquation', 64, 1)):
    base = 'weno_odeint/%s/n%d/s%d' % (cls_name, n, seed)
    eq = getattr(equations, cls_name)(n, random_seed=seed)
    spec = baseline_spec(eq, accuracy_order=3)
    y = exact[base + '/probe']
    derivs = np.stack([
        sum(c * np.roll(y, len(taps) // 2 - i) for i, c in enumerate(taps))
        for taps in spec['baseline_coefficients']], axis=-1)
    derivs[..., 0] = np.roll(oracle.weno_reconstruct_left(y), 1)
    derivs[..., 1] = np.roll(oracle.weno_reconstruct_right(y), 1)
    y_t = oracle.equation_of_motion(spec['equation'], y, derivs, spec['eta'], spec['dx'])
    got = eq.finalize_time_derivative(0.2, y_t)
    want = exact[base + '/rhs_t0.2_probe']
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12 * np.abs(want).max())


def test_best_baseline_model_tables():
  """BaselineModel(accuracy_order=None) and weno=True build the stencil tables
  model.py:69-95 / integrate.py:124-131 describe (host side only)."""
  eq = equations.GodunovBurgersEquation(64)
  best = model_lib.BaselineModel(eq, accuracy_order=None)
  assert best.weno and best.stencil_size == 4
  assert not best.table[:2].any()                       # u_minus / u_plus rows: WENO
  np.testing.assert_allclose(best.table[2] * eq.grid.solution_dx,
                             [1 / 12, -5 / 4, 5 / 4, -1 / 12], atol=1e-6)   # finite-volume form
  wd = model_lib.BaselineModel(equations.GodunovKdVEquation(64), 3, weno=True)
  assert wd.weno and len(wd.stencils) == 3
  with pytest.raises(ValueError, match='Godunov'):
    model_lib.BaselineModel(equations.KdVEquation(64), 1, weno=True)
  with pytest.raises(AssertionError, match='exact equation type'):
    model_lib.BaselineModel(equations.ConservativeBurgersEquation(64), None)
  with pytest.raises(ValueError, match='invalid equation'):
    model_lib.SpectralModel(equations.ConservativeKdVEquation(64))


# ---- known-answer tables of the reference's weno_test.py:30-96 -------------
def test_weno_known_answers_smooth_weights():
  u = np.zeros(5)
  np.testing.assert_allclose(weno.calculate_omega(u), np.stack(5 * [[0.1, 0.6, 0.3]], axis=1))
  np.testing.assert_allclose(weno.left_coefficients(u),
                             np.stack(5 * [[2 / 60, -13 / 60, 47 / 60, 27 / 60, -3 / 60]]))
  np.testing.assert_allclose(weno.right_coefficients(u),
                             np.stack(5 * [[-3 / 60, 27 / 60, 47 / 60, -13 / 60, 2 / 60]]))


def test_weno_known_answers_discontinuity():
  u = np.array([0, 1, 2, 3, 4, -4, -3, -2, -1])
  for mod in (weno, ):
    np.testing.assert_allclose(mod.reconstruct_left(u),
                               [0.5, 1.5, 2.5, 3.5, 4.5, -3.5, -2.5, -1.5, -0.5], atol=0.005)
    np.testing.assert_allclose(mod.reconstruct_right(u),
                               [0.5, 1.5, 2.5, 3.5, -4.5, -3.5, -2.5, -1.5, -0.5], atol=0.005)
  uf = u.astype(np.float64)
  np.testing.assert_allclose(oracle.weno_reconstruct_left(uf), weno.reconstruct_left(uf), atol=1e-13)
  np.testing.assert_allclose(oracle.weno_reconstruct_right(uf), weno.reconstruct_right(uf), atol=1e-13)


@pytest.mark.parametrize('u', [
    [0, 0, 0, 0, 1, 0, 0, 0, 0, 0], [1, 1, 1, 1, 1, 0, 0, 0, 0, 0],
    [1, 2, 3, 4, 5, 0, 0, 0, 0, 0], [0, 0, 1, 2, 3, 0, 0, 0, 0, 0],
    [0, 0, 0, 1, 2, 0, 0, 0, 0, 0], list(2 * np.random.RandomState(0).rand(10))])
def test_weno_reflection_symmetry(u):
  u = np.array(u, dtype=float)
  flip_staggered = lambda x: np.roll(x, +1)[::-1]
  np.testing.assert_allclose(weno.reconstruct_left(u),
                             flip_staggered(weno.reconstruct_right(u[::-1])), atol=1e-6)
  np.testing.assert_allclose(weno.reconstruct_right(u),
                             flip_staggered(weno.reconstruct_left(u[::-1])), atol=1e-6)


def test_weno_batched():
  ub = np.array([[0, 0, 0, 1, 2, 3, 4], [0, 0, 1, 2, 3, 4, 5]], dtype=float)
  np.testing.assert_allclose(weno.reconstruct_left(ub),
                             np.stack([weno.reconstruct_left(r) for r in ub]))
  np.testing.assert_allclose(weno.reconstruct_right(ub),
                             np.stack([weno.reconstruct_right(r) for r in ub]))


@pytest.mark.parametrize('y,period', [
    (np.sin(2 * np.pi * np.arange(8) / 8), 1), (np.sin(2 * np.pi * np.arange(8) / 8), 8),
    (np.linspace(-1, 1, num=12) ** 2, 2)])
def test_spectral_derivative_vs_fftpack(y, period):
  """The reference's duckarray_test.py:56-66: orders 0-2 agree with
  scipy.fftpack.diff (they differ only in the Nyquist mode of odd orders)."""
  import scipy.fftpack
  for order in range(3):
    expected = scipy.fftpack.diff(y, order=order, period=period)
    np.testing.assert_allclose(duckarray.spectral_derivative(y, order, period), expected,
                               atol=1e-12)
    np.testing.assert_allclose(oracle.spectral_derivative(y, order, period), expected,
                               atol=1e-12)
