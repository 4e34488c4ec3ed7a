"""Golden fixtures for the fine-grid "exact" solvers (WENO5 / spectral).

Same rules as make_golden.py (run in the authoring container only; the
reference is imported unmodified under in-memory stubs; nothing is copied):

    python tests/golden/make_golden_exact.py

Recorded, all from the reference's own NumPy/SciPy code:
  weno        : reconstruct_left / reconstruct_right on random inputs
  duckarray   : spectral_derivative (orders 1-4), smoothing_filter (orders 2-4)
  integrate   : SpectralDifferentiator RHS for KdV / KS,
                integrate_exact (spectral) with warm-up and periodic filtering,
                odeint driven by a WENO differentiator assembled from reference
                functions only (weno.reconstruct_*, polynomials.coefficients at
                accuracy 3 applied with np.roll, GodunovBurgers
                equation_of_motion + finalize_time_derivative) -- the
                reference's WENODifferentiator needs a TF session for the
                polynomial part.
Output: tests/golden/reference_exact_solvers.npz
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden  # noqa: E402  (stub machinery)


def main():
  duckarray, polynomials, equations, integrate = make_golden.import_reference()
  from pde_superresolution import weno
  out = {}
  rs = np.random.RandomState(7)

  # -- weno.py --------------------------------------------------------------
  x = np.linspace(0, 2 * np.pi, 64, endpoint=False)
  smooth = np.stack([np.sin(x) + 0.3 * np.cos(3 * x + 1.0),
                     np.where(x < 3, 1.0, -0.5) + 0.1 * np.sin(2 * x),   # a shock
                     rs.randn(64)])
  out['weno/u'] = smooth
  out['weno/left'] = weno.reconstruct_left(smooth)
  out['weno/right'] = weno.reconstruct_right(smooth)
  out['weno/omega'] = weno.calculate_omega(smooth)

  # -- duckarray spectral helpers ------------------------------------------
  y = rs.randn(2, 32)
  out['spectral_derivative/x'] = y
  for order in (1, 2, 3, 4):
    out['spectral_derivative/order%d_period7' % order] = (
        duckarray.spectral_derivative(y, order, 7.0))
  for order in (2, 3, 4):
    out['smoothing_filter/order%d' % order] = duckarray.smoothing_filter(y, order=order)

  # -- SpectralDifferentiator RHS -------------------------------------------
  for cls_name, n in (('KdVEquation', 64), ('KSEquation', 128), ('BurgersEquation', 64)):
    eq = getattr(equations, cls_name)(n, random_seed=3)
    diff = integrate.SpectralDifferentiator(eq)
    xs = eq.grid.solution_x
    state = 0.5 * np.sin(2 * np.pi * xs / eq.grid.period) + 0.2 * np.cos(
        6 * np.pi * xs / eq.grid.period + 0.4)
    out['spectral_rhs/%s/n%d/y' % (cls_name, n)] = state
    out['spectral_rhs/%s/n%d/out_t0.3' % (cls_name, n)] = diff(0.3, state)

  # -- integrate_exact (spectral exact solvers), warm-up and filtering ------
  eq = equations.KdVEquation(64, random_seed=1)
  times = np.linspace(0, 0.1, 3)
  ds = integrate.integrate_exact(eq, times=times, warmup=0.05)
  out['exact/kdv64_warmup/times'] = np.asarray(ds.coords['time'])
  out['exact/kdv64_warmup/y'] = ds.data_vars['y'][1]
  out['exact/kdv64_warmup/nfev'] = np.array(ds.coords['num_evals'])
  eq = equations.KSEquation(64, random_seed=2)
  times = np.linspace(0, 0.04, 5)
  ds = integrate.integrate_exact(eq, times=times, warmup=0.02, filter_interval=0.01)
  out['exact/ks64_filtered/times'] = np.asarray(ds.coords['time'])
  out['exact/ks64_filtered/y'] = ds.data_vars['y'][1]
  out['exact/ks64_filtered/nfev'] = np.array(ds.coords['num_evals'])

  # -- WENO differentiator from reference pieces ----------------------------
  FV = polynomials.Method.FINITE_VOLUMES

  class WenoRollDifferentiator(integrate.Differentiator):
    """integrate.WENODifferentiator (integrate.py:124-140) with the polynomial
    part (PolynomialDifferentiator, accuracy 3) applied by np.roll."""

    def __init__(self, equation, accuracy_order=3):
      self.equation = equation
      self.stencils = []
      for d in equation.DERIVATIVE_ORDERS:
        grid = polynomials.regular_grid(equation.GRID_OFFSET, d, accuracy_order,
                                        equation.grid.solution_dx)
        taps = polynomials.coefficients(grid, FV, d)
        left = -(-(len(taps) - 1) // 2)
        self.stencils.append((taps, left))

    def __call__(self, t, y):
      derivs = {}
      for name, (taps, left) in zip(self.equation.DERIVATIVE_NAMES, self.stencils):
        derivs[name] = sum(c * np.roll(y, -(i - left)) for i, c in enumerate(taps))
      derivs['u_minus'] = np.roll(weno.reconstruct_left(y), 1)
      derivs['u_plus'] = np.roll(weno.reconstruct_right(y), 1)
      y_t = self.equation.equation_of_motion(y, derivs)
      return self.equation.finalize_time_derivative(t, y_t)

  for cls_name, n, seed, times in (
      ('GodunovBurgersEquation', 64, 3, np.linspace(0, 0.5, 6)),
      ('GodunovBurgersEquation', 128, 5, np.linspace(0, 0.3, 4)),
      ('GodunovKdVEquation', 64, 1, np.linspace(0, 0.02, 3)),
  ):
    eq = getattr(equations, cls_name)(n, random_seed=seed)
    diff = WenoRollDifferentiator(eq)
    y0 = eq.initial_value()
    if not np.any(y0):   # Burgers starts from rest; give the RHS check a state
      probe = np.sin(eq.grid.solution_x) + 0.5
    else:
      probe = y0
    sol, nfev = integrate.odeint(y0, diff, times, method='RK23')
    base = 'weno_odeint/%s/n%d/s%d' % (cls_name, n, seed)
    out[base + '/times'] = times
    out[base + '/y0'] = y0
    out[base + '/y'] = sol
    out[base + '/nfev'] = np.array(nfev)
    out[base + '/probe'] = probe
    out[base + '/rhs_t0.2_probe'] = diff(0.2, probe)

  path = os.path.join(HERE, 'reference_exact_solvers.npz')
  np.savez_compressed(path, **out)
  print('wrote', path, os.path.getsize(path), 'bytes;', len(out), 'arrays')


if __name__ == '__main__':
  main()
