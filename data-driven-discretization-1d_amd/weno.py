"""Fifth-order upwind-biased WENO reconstruction, host (NumPy) form.

Same public names as the reference's ``pde_superresolution/weno.py`` (43-130);
used for host-side checks and by callers that hold NumPy arrays.  Batches on
the device go through the HIP kernels (``BaselineModel(weno=True)``,
csrc/dev_params.h ``weno_minus_plus``), which evaluate the same formulas.

Notation: for cell j with neighbours u[j-2..j+2] the three candidate stencils
have smoothness indicators IS_k (Tang 2005, Eq. 7), nonlinear weights
omega_k = alpha_k / sum(alpha), alpha_k = d_k / (eps + IS_k)^p, and the
reconstruction at the j+1/2 edge is a convex combination of the three
third-order candidates (Shu 1998, Procedure 2.2).
"""
import numpy as np

OPTIMAL_SMOOTH_WEIGHTS = (0.1, 0.6, 0.3)

# candidate-stencil combination matrices: row i gives the weight of window
# entry i as a linear form in (omega_0, omega_1, omega_2), times 1/6
_LEFT_FORM = np.array([[2, 0, 0], [-7, -1, 0], [11, 5, 2], [0, 2, 5], [0, 0, -1]]) / 6.0
_RIGHT_FORM = np.array([[-1, 0, 0], [5, 2, 0], [2, 5, 11], [0, -1, -7], [0, 0, 2]]) / 6.0


def _window(u, offsets):
  """[..., len(offsets), x]: entry i holds u[x + offsets[i]] (periodic)."""
  return np.stack([np.roll(u, -o, axis=-1) for o in offsets], axis=-2)


def calculate_smoothness_indicators(u):
  """[..., 3, x] smoothness indicators of the three candidate stencils."""
  m2, m1, c, p1, p2 = np.moveaxis(_window(np.asarray(u), (-2, -1, 0, 1, 2)), -2, 0)
  return np.stack([
      1 / 4 * (m2 - 4 * m1 + 3 * c) ** 2 + 13 / 12 * (m2 - 2 * m1 + c) ** 2,
      1 / 4 * (m1 - p1) ** 2 + 13 / 12 * (m1 - 2 * c + p1) ** 2,
      1 / 4 * (3 * c - 4 * p1 + p2) ** 2 + 13 / 12 * (c - 2 * p1 + p2) ** 2,
  ], axis=-2)


def calculate_omega(u, optimal_linear_weights=OPTIMAL_SMOOTH_WEIGHTS,
                    epsilon=1e-6, p=2):
  """[..., 3, x] nonlinear weights."""
  indicators = calculate_smoothness_indicators(u)
  alpha = np.asarray(optimal_linear_weights)[:, np.newaxis] / (epsilon + indicators) ** p
  return alpha / alpha.sum(axis=-2, keepdims=True)


def left_coefficients(u):
  """[..., x, 5] coefficients of u[j-2..j+2] for the left-biased value at j+1/2."""
  omega = calculate_omega(u)
  return np.einsum('ik,...kx->...xi', _LEFT_FORM, omega)


def reconstruct_left(u):
  u = np.asarray(u)
  window = np.moveaxis(_window(u, (-2, -1, 0, 1, 2)), -2, -1)
  return np.sum(left_coefficients(u) * window, axis=-1)


def right_coefficients(u):
  """[..., x, 5] coefficients of u[j-1..j+3] for the right-biased value at j+1/2."""
  omega = np.roll(calculate_omega(u, OPTIMAL_SMOOTH_WEIGHTS[::-1]), -1, axis=-1)
  return np.einsum('ik,...kx->...xi', _RIGHT_FORM, omega)


def reconstruct_right(u):
  u = np.asarray(u)
  window = np.moveaxis(_window(u, (-1, 0, 1, 2, 3)), -2, -1)
  return np.sum(right_coefficients(u) * window, axis=-1)
