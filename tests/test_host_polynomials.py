"""Host stencil algebra vs the reference's known-answer tables and golden dumps.

Tables restated from pde_superresolution/polynomials_test.py:36-157; golden
arrays produced by the reference itself (tests/golden/make_golden.py).
"""
import numpy as np
import pytest

from ddd1d_amd import polynomials

FD = polynomials.Method.FINITE_DIFFERENCES
FV = polynomials.Method.FINITE_VOLUMES
CENTERED = polynomials.GridOffset.CENTERED
STAGGERED = polynomials.GridOffset.STAGGERED


@pytest.mark.parametrize('grid,order,expected', [
    ([-1, 0, 1], 1, [-1 / 2, 0, 1 / 2]),
    ([-1, 0, 1], 2, [1, -2, 1]),
    ([-2, -1, 0, 1, 2], 2, [-1 / 12, 4 / 3, -5 / 2, 4 / 3, -1 / 12]),
    ([0, 1], 1, [-1, 1]),
    ([0, 2], 1, [-0.5, 0.5]),
    ([0, 0.5], 1, [-2, 2]),
    ([0, 1, 2, 3, 4], 4, [1, -4, 6, -4, 1]),
])
def test_finite_difference_coefficients(grid, order, expected):
  np.testing.assert_allclose(
      polynomials.coefficients(np.array(grid), FD, order), expected)


@pytest.mark.parametrize('grid,order,expected', [
    ([-0.5, 0.5], 0, [1 / 2, 1 / 2]),
    ([-1, 1], 0, [1 / 2, 1 / 2]),
    ([-1.5, -0.5], 0, [-1 / 2, 3 / 2]),
    ([-0.5, 0.5, 1.5], 0, [1 / 3, 5 / 6, -1 / 6]),
    ([-0.25, 0.25, 0.75], 0, [1 / 3, 5 / 6, -1 / 6]),
    ([2.5, 1.5, 0.5, -0.5, -1.5], 0,
     [2 / 60, -13 / 60, 47 / 60, 27 / 60, -3 / 60]),
    ([-0.5, 0.5], 1, [-1, 1]),
    ([-1, 1], 1, [-1 / 2, 1 / 2]),
    ([0.5, 1.5, 2.5], 1, [-2, 3, -1]),
    ([-1.5, -0.5, 0.5, 1.5], 1, [1 / 12, -5 / 4, 5 / 4, -1 / 12]),
    ([-.75, -0.25, 0.25, 0.75], 1, [1 / 6, -5 / 2, 5 / 2, -1 / 6]),
])
def test_finite_volume_coefficients(grid, order, expected):
  np.testing.assert_allclose(
      polynomials.coefficients(np.array(grid), FV, order), expected)


@pytest.mark.parametrize('offset,order,accuracy,expected', [
    (CENTERED, 0, 1, [0]),
    (CENTERED, 1, 1, [-1, 0, 1]),
    (CENTERED, 2, 1, [-1, 0, 1]),
    (CENTERED, 3, 1, [-2, -1, 0, 1, 2]),
    (CENTERED, 4, 1, [-2, -1, 0, 1, 2]),
    (STAGGERED, 0, 1, [-0.5, 0.5]),
    (STAGGERED, 1, 1, [-0.5, 0.5]),
    (STAGGERED, 2, 1, [-1.5, -0.5, 0.5, 1.5]),
    (STAGGERED, 3, 1, [-1.5, -0.5, 0.5, 1.5]),
    (CENTERED, 0, 6, [-3, -2, -1, 0, 1, 2, 3]),
    (STAGGERED, 0, 6, [-2.5, -1.5, -0.5, 0.5, 1.5, 2.5]),
])
def test_regular_grid(offset, order, accuracy, expected):
  np.testing.assert_allclose(
      polynomials.regular_grid(offset, order, accuracy), expected)


@pytest.mark.parametrize('grid,method,order', [
    ([-2, -1, 0, 1, 2], FD, 1),
    ([-2, -1, 0, 1, 2], FD, 2),
    ([-1.5, -0.5, 0.5, 1.5], FD, 1),
    ([-1.5, -0.5, 0.5, 1.5], FV, 1),
])
def test_polynomial_accuracy_layer_consistency(grid, method, order):
  args = (np.array(grid), method, order, 2)
  A, b = polynomials.constraints(*args)
  layer = polynomials.PolynomialAccuracyLayer(*args)
  inputs = np.random.RandomState(0).randn(10, layer.input_size)
  outputs = layer.bias + np.einsum('bi,ij->bj', inputs, layer.nullspace)
  residual = np.einsum('ij,bj->bi', A, outputs) - b
  np.testing.assert_allclose(residual, 0, atol=1e-7)


def test_bias_zero_padding():
  layer = polynomials.PolynomialAccuracyLayer(
      np.array([-1.5, -0.5, 0.5, 1.5]), FD, derivative_order=0,
      bias_zero_padding=(0, 1))
  expected = np.concatenate(
      [polynomials.coefficients(np.array([-1.5, -0.5, 0.5]), FD, 0), [0.0]])
  np.testing.assert_allclose(layer.bias, expected)


def test_error_cases():
  with pytest.raises(ValueError, match='not a regular grid'):
    polynomials.constraints(np.array([0, 1, 3.0]), FD, 1)
  with pytest.raises(ValueError, match='no valid'):
    polynomials.constraints(np.array([0, 1.0]), FD, 1, accuracy_order=4)
  with pytest.raises(ValueError, match='non-positive'):
    polynomials.constraints(np.array([0, 1.0]), FD, 3)
  with pytest.raises(ValueError, match='only one valid solution'):
    polynomials.PolynomialAccuracyLayer(np.array([-1, 0, 1.0]), FD, 1, 2)
  with pytest.raises(ValueError, match='not in nullspace'):
    polynomials.PolynomialAccuracyLayer(
        np.array([-2, -1, 0, 1, 2.0]), FD, 1, 1, bias=np.ones(5))


# -- golden: arrays dumped from the reference ------------------------------
def test_regular_grid_golden(golden):
  for key in golden.index['regular_grid']:
    _, offset, d, a, dx = key.split('/')
    got = polynomials.regular_grid(polynomials.GridOffset[offset], int(d[1:]),
                                   int(a[1:]), float(dx[2:]))
    np.testing.assert_array_equal(got, golden[key])


def _parse_layer_key(key):
  _, offset, method, g, dx, d, a, s = key.split('/')
  return (polynomials.GridOffset[offset],
          FD if method == 'FD' else FV, int(g[1:]), float(dx[2:]),
          int(d[1:]), int(a[1:]), float(s[1:]))


def test_polynomial_accuracy_layers_golden(golden):
  """A, b, bias bit-identical; null space identical as a subspace AND as the
  same LAPACK basis on this machine (the basis is data)."""
  keys = golden.index['polynomial_accuracy_layers']
  assert len(keys) > 100
  for key in keys:
    offset, method, g, dx_rounded, d, acc, scale = _parse_layer_key(key)
    # keys carry dx rounded to 6 decimals; recover the exact value
    dx = min((2 * np.pi / 64, 0.25), key=lambda v: abs(v - dx_rounded))
    grid = polynomials.regular_grid(offset, 0, g, dx)
    A, b = polynomials.constraints(grid, method, d, acc)
    np.testing.assert_array_equal(A, golden[key + '/A'])
    np.testing.assert_array_equal(b, golden[key + '/b'])
    layer = polynomials.PolynomialAccuracyLayer(grid, method, d, acc,
                                                out_scale=scale)
    assert layer.input_size == int(golden[key + '/input_size'])
    np.testing.assert_array_equal(layer.bias, golden[key + '/bias'])
    np.testing.assert_allclose(layer.nullspace, golden[key + '/nullspace'],
                               rtol=1e-12, atol=1e-12)
    ckey = 'coef/' + '/'.join(key.split('/')[1:6])
    np.testing.assert_array_equal(
        polynomials.coefficients(grid, method, d), golden[ckey])


def test_zero_padded_golden(golden):
  got = polynomials.zero_padded_coefficients(
      np.array([-1.5, -0.5, 0.5, 1.5]), FD, 0, (0, 1))
  np.testing.assert_array_equal(got, golden['zero_padded/example'])
