"""GPU operators vs the reference's known-answer tables and the oracle.

Tables restated from pde_superresolution/layers_test.py:49-86.
"""
import numpy as np
import pytest

from helpers import oracle, rel_err
from ddd1d_amd import equations, layers, polynomials, _lib, model as model_lib

pytestmark = pytest.mark.gpu


def _pad_1d(values, padding, center):
  x = np.asarray(values, dtype=np.float32)[None, :, None]
  return layers.pad_periodic(x, padding, center).cpu().numpy()[0, :, 0]


@pytest.mark.parametrize('padding,center,expected', [
    (0, True, [0, 1, 2]),
    (1, True, [2, 0, 1, 2]),
    (2, True, [2, 0, 1, 2, 0]),
    (3, True, [1, 2, 0, 1, 2, 0]),
    (4, True, [1, 2, 0, 1, 2, 0, 1]),
    (6, True, [0, 1, 2, 0, 1, 2, 0, 1, 2]),
    (7, True, [2, 0, 1, 2, 0, 1, 2, 0, 1, 2]),
    (0, False, [0, 1, 2]),
    (1, False, [0, 1, 2, 0]),
    (2, False, [0, 1, 2, 0, 1]),
    (3, False, [0, 1, 2, 0, 1, 2]),
    (5, False, [0, 1, 2, 0, 1, 2, 0, 1]),
])
def test_pad_periodic_table(padding, center, expected):
  np.testing.assert_array_equal(_pad_1d(range(3), padding, center), expected)
  # and the oracle restatement agrees with the same table
  want = oracle.pad_periodic(np.arange(3.0)[None, :, None], padding, center)
  np.testing.assert_array_equal(want[0, :, 0], expected)


def test_pad_periodic_rejects_2d():
  with pytest.raises(ValueError, match='3D'):
    layers.pad_periodic(np.zeros((2, 3), np.float32), 2)


def test_nn_conv1d_periodic_table():
  inputs = np.arange(5.0, dtype=np.float32)[None, :, None]
  for filt, expected in [
      ([0.0, 1.0, 0.0], inputs[0, :, 0]),
      ([0.0, 1.0], inputs[0, :, 0]),
      ([0.5, 0.5], [2.0, 0.5, 1.5, 2.5, 3.5]),
  ]:
    filters = np.asarray(filt, np.float32)[:, None, None]
    got = layers.nn_conv1d_periodic(inputs, filters, center=True).cpu().numpy()
    np.testing.assert_allclose(got[0, :, 0], expected)
    want = oracle.nn_conv1d_periodic(inputs, filters, center=True)
    np.testing.assert_allclose(want[0, :, 0], expected)


@pytest.mark.parametrize('n,cin,cout,k,center', [
    (64, 1, 32, 5, True), (64, 32, 32, 5, True), (37, 3, 7, 4, True),
    (16, 8, 5, 3, False), (8, 2, 2, 9, True), (5, 1, 1, 13, True),
])
def test_conv1d_periodic_random(n, cin, cout, k, center):
  rs = np.random.RandomState(n + k)
  x = rs.randn(3, n, cin).astype(np.float32)
  w = rs.randn(k, cin, cout).astype(np.float32)
  b = rs.randn(cout).astype(np.float32)
  got = layers.conv1d_periodic_layer(x, w, b, activation='relu',
                                     center=center).cpu().numpy()
  want = oracle.conv1d_periodic_layer(x, w, b, 'relu', center=center)
  assert rel_err(got, want) < 2e-6


@pytest.mark.parametrize('n,cin,cout,k,strides,dilation,center', [
    (64, 3, 5, 5, 1, 2, True), (64, 3, 5, 5, 1, 3, False), (37, 2, 4, 3, 1, 4, True),
    (64, 3, 5, 5, 2, 1, True), (37, 2, 4, 4, 3, 1, True), (16, 1, 1, 2, 4, 1, False),
])
def test_conv1d_periodic_strides_and_dilation(n, cin, cout, k, strides, dilation, center):
  """layers.py:103-137: strides / dilation_rate of conv1d_periodic_layer (never used by the
  reference's models, part of its layer API): [batch, ceil(N / strides), filters]."""
  rs = np.random.RandomState(n + k + strides + dilation)
  x = rs.randn(2, n, cin).astype(np.float32)
  w = rs.randn(k, cin, cout).astype(np.float32)
  b = rs.randn(cout).astype(np.float32)
  got = layers.conv1d_periodic_layer(x, w, b, activation='tanh', strides=strides,
                                     dilation_rate=dilation, center=center).cpu().numpy()
  want = oracle.conv1d_periodic_layer(x, w, b, 'tanh', center=center, strides=strides,
                                      dilation_rate=dilation)
  assert got.shape == want.shape == (2, -(-n // strides), cout)
  assert rel_err(got, want) < 2e-6
  if dilation == 1:
    plain = layers.nn_conv1d_periodic(x, w, stride=strides, center=center).cpu().numpy()
    assert plain.shape == (2, -(-n // strides), cout)
  with pytest.raises(ValueError, match='conjunction'):
    layers.conv1d_periodic_layer(x, w, b, strides=2, dilation_rate=2)


@pytest.mark.parametrize('activation', ['relu', 'relu6', 'tanh', 'softplus', 'elu'])
def test_activations(activation):
  rs = np.random.RandomState(7)
  x = (4 * rs.randn(2, 32, 4)).astype(np.float32)
  w = rs.randn(3, 4, 6).astype(np.float32)
  got = layers.conv1d_periodic_layer(x, w, None, activation=activation,
                                     center=True).cpu().numpy()
  want = oracle.conv1d_periodic_layer(x, w, np.zeros(6, np.float32), activation)
  assert rel_err(got, want) < 2e-6


@pytest.mark.parametrize('grid,method,order', [
    ([-2, -1, 0, 1, 2], polynomials.Method.FINITE_DIFFERENCES, 1),
    ([-2, -1, 0, 1, 2], polynomials.Method.FINITE_DIFFERENCES, 2),
    ([-1.5, -0.5, 0.5, 1.5], polynomials.Method.FINITE_DIFFERENCES, 1),
    ([-1.5, -0.5, 0.5, 1.5], polynomials.Method.FINITE_VOLUMES, 1),
])
def test_polynomial_accuracy_layer_apply(grid, method, order):
  """polynomials_test.py:88-104 through the GPU apply()."""
  args = (np.array(grid), method, order, 2)
  A, b = polynomials.constraints(*args)
  layer = polynomials.PolynomialAccuracyLayer(*args)
  inputs = np.random.RandomState(0).randn(10, layer.input_size).astype(np.float32)
  outputs = layer.apply(inputs).cpu().numpy()
  residual = np.einsum('ij,bj->bi', A, outputs.astype(np.float64)) - b
  np.testing.assert_allclose(residual, 0, atol=2e-6)
  want = layer.bias.astype(np.float32) + inputs @ layer.nullspace.astype(np.float32)
  np.testing.assert_allclose(outputs, want, rtol=1e-6, atol=1e-6)


def test_reconstruct_matches_oracle():
  rs = np.random.RandomState(3)
  u = rs.randn(4, 48).astype(np.float32)
  grid = polynomials.regular_grid(polynomials.GridOffset.CENTERED, 2, 3, 0.1)
  got = polynomials.reconstruct(u, grid, polynomials.Method.FINITE_DIFFERENCES,
                                2).cpu().numpy()
  taps = polynomials.coefficients(grid, polynomials.Method.FINITE_DIFFERENCES, 2)
  want = oracle.nn_conv1d_periodic(u[..., None],
                                   taps.astype(np.float32)[:, None, None],
                                   center=True)[..., 0]
  assert rel_err(got, want) < 2e-6


def test_mfma_layout_selftest():
  _lib.selftest_mfma_layout()


def test_extract_patches_alignment_and_apply_coefficients():
  """model.extract_patches / apply_coefficients (model.py:516-548): patch i of
  point x is u[x + i - size // 2]; for even sizes the quantity sits at the left
  cell edge (offsets -3..+2 for size 6)."""
  u = np.arange(8, dtype=np.float32)[None, :] * 10
  got6 = model_lib.extract_patches(u, 6).cpu().numpy()
  np.testing.assert_array_equal(got6[0, 0], [50, 60, 70, 0, 10, 20])
  np.testing.assert_array_equal(got6[0, 4], [10, 20, 30, 40, 50, 60])
  got7 = model_lib.extract_patches(u, 7).cpu().numpy()
  np.testing.assert_array_equal(got7[0, 0], [50, 60, 70, 0, 10, 20, 30])
  rs = np.random.RandomState(0)
  y = rs.randn(5, 48).astype(np.float32)
  for size in (3, 6, 7):
    np.testing.assert_array_equal(model_lib.extract_patches(y, size).cpu().numpy(),
                                  oracle.extract_patches(y, size))
    coeff = rs.randn(5, 48, 3, size).astype(np.float32)
    got = model_lib.apply_coefficients(coeff, y).cpu().numpy()
    want = oracle.apply_coefficients(coeff, y)
    assert rel_err(got, want) < 1e-6
  with pytest.raises(ValueError):
    model_lib.apply_coefficients(np.zeros((2, 8, 2, 6), np.float32), np.zeros((2, 9), np.float32))


@pytest.mark.parametrize('cls', [
    equations.BurgersEquation, equations.ConservativeBurgersEquation,
    equations.GodunovBurgersEquation, equations.KdVEquation,
    equations.ConservativeKdVEquation, equations.GodunovKdVEquation,
    equations.KSEquation, equations.ConservativeKSEquation, equations.GodunovKSEquation])
def test_apply_space_derivatives_all_equations(cls):
  """model.apply_space_derivatives (model.py:115-135) for all nine equations."""
  eq = cls(64, random_seed=0)
  rs = np.random.RandomState(1)
  y = rs.randn(3, 64).astype(np.float32)
  derivs = rs.randn(3, 64, len(eq.DERIVATIVE_NAMES)).astype(np.float32)
  got = model_lib.apply_space_derivatives(derivs, y, eq).cpu().numpy()
  spec = eq.kernel_spec()
  want = oracle.equation_of_motion(spec['equation'], y, derivs, spec['eta'], spec['dx'])
  assert rel_err(got, want) < 1e-6
  # and the host mirror of the reference's own method (bit-identical to it)
  named = {n: derivs[..., i].astype(np.float64) for i, n in enumerate(eq.DERIVATIVE_NAMES)}
  ref = eq.equation_of_motion(y.astype(np.float64), named)
  assert rel_err(got, ref) < 1e-5
  with pytest.raises(ValueError, match='unexpected size'):
    model_lib.apply_space_derivatives(derivs, y[:, :32], eq)
