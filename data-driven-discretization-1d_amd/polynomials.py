"""Finite-difference / finite-volume stencil algebra (host side, NumPy float64).

Everything in this module runs ONCE on the host and produces the small constant
tables (stencil grids, standard coefficients, null-space bases) that the HIP
kernels consume.  It mirrors the public surface of the reference module
``pde_superresolution/polynomials.py``:

  * ``GridOffset`` / ``Method``            -> polynomials.py:31-40
  * ``regular_grid``                       -> polynomials.py:43-71
  * ``constraints``                        -> polynomials.py:74-149
  * ``coefficients``                       -> polynomials.py:152-167
  * ``zero_padded_coefficients``           -> polynomials.py:170-195
  * ``PolynomialAccuracyLayer``            -> polynomials.py:198-277
  * ``reconstruct``                        -> polynomials.py:280-303

The reference evaluates ``PolynomialAccuracyLayer.apply`` and ``reconstruct``
as TensorFlow graph ops; here they dispatch to the HIP library (the projection
is fused into the learned-stencil kernel, ``reconstruct`` is the fixed-stencil
periodic convolution kernel).
"""
import enum
import math
from typing import Optional, Tuple

import numpy as np


class GridOffset(enum.Enum):
  """Where the output sample sits relative to the input samples."""
  CENTERED = 1    # output on an input grid point
  STAGGERED = 2   # output half-way between two input grid points


class Method(enum.Enum):
  """What the input samples represent."""
  FINITE_DIFFERENCES = 1   # point values
  FINITE_VOLUMES = 2       # cell averages


def regular_grid(grid_offset: GridOffset,
                 derivative_order: int,
                 accuracy_order: int = 1,
                 dx: float = 1) -> np.ndarray:
  """Smallest symmetric stencil able to deliver the requested accuracy.

  A stencil needs ``derivative_order + accuracy_order`` degrees of freedom.
  Centered stencils have an odd number of points, staggered ones an even
  number, so the count is rounded up to the next admissible size.
  """
  need = derivative_order + accuracy_order
  if grid_offset is GridOffset.CENTERED:
    half = need // 2
    offsets = np.arange(-half, half + 1)
    return offsets * dx
  if grid_offset is GridOffset.STAGGERED:
    half = (need + 1) // 2
    offsets = np.arange(-half, half) + 0.5
    return offsets * dx
  raise ValueError('unexpected grid_offset: {}'.format(grid_offset))


def _moment_row(grid: np.ndarray, method: Method, power: int,
                spacing: float) -> np.ndarray:
  """Action of the stencil points on the monomial x**power."""
  if method is Method.FINITE_DIFFERENCES:
    return grid ** power
  if method is Method.FINITE_VOLUMES:
    # exact cell average of x**power over [x - h/2, x + h/2]
    hi = (grid + spacing / 2) ** (power + 1)
    lo = (grid - spacing / 2) ** (power + 1)
    return 1 / spacing * (hi - lo) / (power + 1)
  raise ValueError('unexpected method: {}'.format(method))


def constraints(grid: np.ndarray,
                method: Method,
                derivative_order: int,
                accuracy_order: Optional[int] = None
                ) -> Tuple[np.ndarray, np.ndarray]:
  """Linear system ``A @ c = b`` characterising admissible coefficients.

  Row ``m`` demands that the stencil differentiates the monomial ``x**m``
  exactly (at x = 0): zero for ``m != derivative_order`` and
  ``derivative_order!`` for ``m == derivative_order``.  As in the reference
  (polynomials.py:117-147) the homogeneous rows are de-duplicated through a
  set and emitted in sorted order, with the inhomogeneous row last; that
  ordering fixes the SVD basis returned by PolynomialAccuracyLayer.
  """
  grid = np.asarray(grid)
  if accuracy_order is None:
    accuracy_order = grid.size - derivative_order
  if accuracy_order < 1:
    raise ValueError('cannot compute constriants with non-positive '
                     'accuracy_order: {}'.format(accuracy_order))

  steps = np.unique(np.diff(grid))
  if (abs(steps - steps[0]) > 1e-8).any():
    raise ValueError('not a regular grid: {}'.format(steps))
  spacing = steps[0]

  homogeneous = set()
  inhomogeneous = None
  for power in range(accuracy_order + derivative_order):
    row = _moment_row(grid, method, power, spacing)
    if power == derivative_order:
      inhomogeneous = row
    else:
      homogeneous.add(tuple(row))
  assert inhomogeneous is not None

  if len(homogeneous) + 1 > grid.size:
    raise ValueError('no valid {} stencil exists for derivative_order={} and '
                     'accuracy_order={} with grid={}'
                     .format(method, derivative_order, accuracy_order, grid))

  A = np.array(sorted(homogeneous) + [inhomogeneous])
  b = np.zeros(A.shape[0])
  b[-1] = math.factorial(derivative_order)
  return A, b


def coefficients(grid: np.ndarray, method: Method,
                 derivative_order: int) -> np.ndarray:
  """Standard (maximum accuracy) coefficients on ``grid``."""
  A, b = constraints(np.asarray(grid), method, derivative_order)
  return np.linalg.solve(A, b)


def zero_padded_coefficients(grid: np.ndarray, method: Method,
                             derivative_order: int,
                             padding: Tuple[int, int]) -> np.ndarray:
  """Standard coefficients on a trimmed grid, zero-filled back to full size."""
  left, right = padding
  grid = np.asarray(grid)
  inner = grid[left:grid.size - right]
  return np.pad(coefficients(inner, method, derivative_order), padding,
                mode='constant')


class PolynomialAccuracyLayer(object):
  """Affine map from free parameters onto the accuracy-constrained stencils.

  ``coeff = bias + inputs @ nullspace`` satisfies ``A @ coeff = b`` for every
  ``inputs``, because ``bias`` is a particular solution and the rows of
  ``nullspace`` span ker(A).  Reference: polynomials.py:209-264.

  NOTE: the basis comes from ``np.linalg.svd`` and is only defined up to an
  orthogonal transformation; it is therefore *model data* and is stored next
  to the conv weights (see ``model.LearnedStencilModel``) rather than being
  recomputed where a checkpoint is consumed.

  Attributes:
    input_size: number of free parameters (= grid_size - rank(A)).
    grid_size: stencil width.
    bias: float64 [grid_size].
    nullspace: float64 [input_size, grid_size], pre-scaled by
      ``out_scale / dx**derivative_order``.
  """

  def __init__(self,
               grid: np.ndarray,
               method: Method,
               derivative_order: int,
               accuracy_order: int = 2,
               bias: Optional[np.ndarray] = None,
               bias_zero_padding: Tuple[int, int] = (0, 0),
               out_scale: float = 1.0):
    grid = np.asarray(grid, dtype=float)
    A, b = constraints(grid, method, derivative_order, accuracy_order)

    if bias is None:
      bias = zero_padded_coefficients(grid, method, derivative_order,
                                      bias_zero_padding)
    if np.linalg.norm(A.dot(bias) - b) > 1e-8:
      raise ValueError('invalid bias, not in nullspace')

    input_size = A.shape[1] - A.shape[0]
    if not input_size:
      raise ValueError(
          'there is only one valid solution accurate to this order')
    _, _, vt = np.linalg.svd(A)
    kernel_basis = vt[-input_size:]

    dx = grid[1] - grid[0]
    self.input_size = input_size
    self.grid_size = grid.size
    self.derivative_order = derivative_order
    self.nullspace = kernel_basis * (out_scale / dx ** derivative_order)
    self.bias = bias

  def apply(self, inputs):
    """[batch, x, input_size] -> [batch, x, grid_size] (float32).

    The reference runs this as a TF einsum (polynomials.py:266-277).  The
    production path never materialises the result: the projection is fused
    into the learned-stencil kernel.  This entry point exists for parity
    tests and dispatches to the HIP library.
    """
    from . import _lib
    return _lib.polynomial_accuracy_apply(
        inputs, self.nullspace.astype(np.float32),
        self.bias.astype(np.float32))


def reconstruct(inputs, grid: np.ndarray, method: Method,
                derivative_order: int):
  """Fixed-stencil derivative of ``inputs`` [batch, x] with periodic wrap.

  Reference: polynomials.py:280-303 (coefficients as a [G,1,1] filter through
  ``layers.nn_conv1d_periodic(center=True)``).  Runs on the GPU.
  """
  from . import layers
  taps = coefficients(grid, method, derivative_order).astype(np.float32)
  out = layers.nn_conv1d_periodic(inputs[..., None], taps[:, None, None],
                                  center=True)
  return out[..., 0]
