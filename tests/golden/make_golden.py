"""Generate golden fixtures by running the REFERENCE's own NumPy code paths.

Run in the authoring container only (needs /root/reference, which does not
exist on the GPU box):

    python tests/golden/make_golden.py

The reference package imports TensorFlow 1.x, absl and xarray at module import
time; none is installed and there is no network.  This script therefore
installs permissive in-memory import stubs for those names (nothing is written
to disk and nothing from the reference is copied), imports the reference
modules unmodified, and records inputs/outputs of the functions whose
implementation is NumPy/SciPy:

  polynomials : regular_grid, constraints, coefficients,
                zero_padded_coefficients, PolynomialAccuracyLayer (A, b, bias,
                nullspace, input_size)
  equations   : Grid, RandomForcing draws and forcing(t), initial_value,
                equation_of_motion of all nine equations on random inputs,
                staggered_first_derivative, godunov_convective_flux, params()
  duckarray   : resample_mean, subsample
  integrate   : odeint (SciPy RK23, max_step=0.01) driven by a Differentiator
                whose RHS is the reference's own equation_of_motion +
                finalize_time_derivative over FIXED polynomial stencils applied
                with np.roll (so the golden trajectory depends only on
                reference code + SciPy), and integrate_spectral end to end.

Output: tests/golden/reference_numpy_paths.npz (+ .json index).  The TF graph
ops (conv1d / extract_image_patches / einsum / odeint_fixed) cannot run here;
see oracle/oracle.py header for what that leaves unpinned.
"""
import importlib.machinery
import json
import os
import sys
import types

import numpy as np

REFERENCE_ROOT = '/root/reference'
HERE = os.path.dirname(os.path.abspath(__file__))


# ---------------------------------------------------------------------------
# import stubs
# ---------------------------------------------------------------------------
class _PermissiveMeta(type):
  def __getattr__(cls, name):
    if name.startswith('__'):
      raise AttributeError(name)
    return _make_stub(cls.__name__ + '.' + name)

  def __call__(cls, *args, **kwargs):
    # Calling a stub "function" returns another stub class; instantiating
    # tf.Tensor etc. never happens on the NumPy paths.
    return _make_stub(cls.__name__ + '()')

  def __instancecheck__(cls, instance):
    return False

  def __getitem__(cls, item):
    return cls


def _make_stub(name):
  return _PermissiveMeta(name, (), {})


class _StubModule(types.ModuleType):
  def __getattr__(self, name):
    if name.startswith('__'):
      raise AttributeError(name)
    value = _make_stub(self.__name__ + '.' + name)
    setattr(self, name, value)
    return value


_STUB_PREFIXES = ('tensorflow', 'absl', 'xarray', 'h5py', 'apache_beam',
                  'google')


class _StubFinder(object):
  """sys.meta_path hook: any import below a stubbed prefix yields a stub."""

  def find_spec(self, fullname, path=None, target=None):
    if fullname.split('.')[0] in _STUB_PREFIXES:
      return importlib.machinery.ModuleSpec(fullname, self, is_package=True)
    return None

  def create_module(self, spec):
    module = _StubModule(spec.name)
    module.__path__ = []
    return module

  def exec_module(self, module):
    pass


def install_stubs():
  if not any(isinstance(f, _StubFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _StubFinder())
  import absl.logging
  absl.logging.info = lambda *a, **k: None   # must be callable and silent
  import xarray

  # integrate.integrate() wraps its result in xarray.Dataset: tiny stand-in
  class _Dataset(dict):
    def __init__(self, data_vars=None, coords=None):
      super().__init__()
      self.data_vars = data_vars or {}
      self.coords = coords or {}
  xarray.Dataset = _Dataset


def import_reference():
  install_stubs()
  if REFERENCE_ROOT not in sys.path:
    sys.path.insert(0, REFERENCE_ROOT)
  from pde_superresolution import duckarray, polynomials, equations, integrate
  return duckarray, polynomials, equations, integrate


# ---------------------------------------------------------------------------
# fixture generation
# ---------------------------------------------------------------------------
EQUATION_CLASS_NAMES = [
    'BurgersEquation', 'ConservativeBurgersEquation', 'GodunovBurgersEquation',
    'KdVEquation', 'ConservativeKdVEquation', 'GodunovKdVEquation',
    'KSEquation', 'ConservativeKSEquation', 'GodunovKSEquation',
]


def main():
  duckarray, polynomials, equations, integrate = import_reference()
  out = {}
  index = {}

  FD = polynomials.Method.FINITE_DIFFERENCES
  FV = polynomials.Method.FINITE_VOLUMES
  CENTERED = polynomials.GridOffset.CENTERED
  STAGGERED = polynomials.GridOffset.STAGGERED

  # -- polynomials ---------------------------------------------------------
  grids = []
  for offset_name, offset in (('CENTERED', CENTERED), ('STAGGERED', STAGGERED)):
    for d in range(0, 5):
      for acc in range(1, 8):
        for dx in (0.25,):
          key = 'grid/{}/d{}/a{}/dx{}'.format(offset_name, d, acc, dx)
          out[key] = polynomials.regular_grid(offset, d, acc, dx)
          grids.append(key)
  index['regular_grid'] = grids

  layers_index = []
  for offset_name, offset in (('CENTERED', CENTERED), ('STAGGERED', STAGGERED)):
    for method_name, method in (('FD', FD), ('FV', FV)):
      for min_size in (6, 9):
        for dx in (2 * np.pi / 64, 0.25):
          grid = polynomials.regular_grid(offset, 0, min_size, dx)
          for d in range(0, 5):
            # standard coefficients at max accuracy
            ckey = 'coef/{}/{}/g{}/dx{:.6f}/d{}'.format(
                offset_name, method_name, min_size, dx, d)
            out[ckey] = polynomials.coefficients(grid, method, d)
            for acc in (1, 2, 3):
              for scale in ((1.0, 0.5) if (acc == 1 and min_size == 6) else (1.0,)):
                try:
                  layer = polynomials.PolynomialAccuracyLayer(
                      grid, method, d, accuracy_order=acc, out_scale=scale)
                except ValueError:
                  continue
                A, b = polynomials.constraints(grid, method, d, acc)
                base = 'pal/{}/{}/g{}/dx{:.6f}/d{}/a{}/s{}'.format(
                    offset_name, method_name, min_size, dx, d, acc, scale)
                out[base + '/A'] = A
                out[base + '/b'] = b
                out[base + '/bias'] = layer.bias
                out[base + '/nullspace'] = layer.nullspace
                out[base + '/input_size'] = np.array(layer.input_size)
                layers_index.append(base)
  index['polynomial_accuracy_layers'] = layers_index

  out['zero_padded/example'] = polynomials.zero_padded_coefficients(
      np.array([-1.5, -0.5, 0.5, 1.5]), FD, 0, (0, 1))

  # -- duckarray -----------------------------------------------------------
  rs = np.random.RandomState(123)
  x = rs.randn(3, 24)
  out['duck/x'] = x
  out['duck/resample_mean_4'] = duckarray.resample_mean(x, 4)
  out['duck/subsample_4'] = duckarray.subsample(x, 4)
  out['duck/resample_mean_axis0'] = duckarray.resample_mean(x.T, 3, axis=0)

  # -- equations -----------------------------------------------------------
  eq_index = []
  for cls_name in EQUATION_CLASS_NAMES:
    cls = getattr(equations, cls_name)
    for (n, rf, seed) in ((32, 1, 0), (64, 8, 3), (16, 4, 11)):
      eq = cls(n, resample_factor=rf, random_seed=seed)
      base = 'eq/{}/n{}/rf{}/s{}'.format(cls_name, n, rf, seed)
      eq_index.append(base)
      out[base + '/solution_x'] = eq.grid.solution_x
      out[base + '/reference_x'] = eq.grid.reference_x
      out[base + '/scalars'] = np.array([
          eq.grid.solution_dx, eq.grid.reference_dx, eq.grid.period,
          eq.time_step, eq.standard_deviation, getattr(eq, 'eta', np.nan)])
      out[base + '/derivative_orders'] = np.array(eq.DERIVATIVE_ORDERS)
      out[base + '/conservative'] = np.array(bool(eq.CONSERVATIVE))
      out[base + '/staggered'] = np.array(eq.GRID_OFFSET is STAGGERED)
      f = eq.forcing
      out[base + '/forcing_a'] = f.a
      out[base + '/forcing_omega'] = f.omega
      out[base + '/forcing_k'] = f.k
      out[base + '/forcing_phi'] = f.phi
      ts = np.array([0.0, 0.3, 7.25, 50.0])
      out[base + '/forcing_t'] = ts
      out[base + '/forcing_values'] = np.stack([f(t) for t in ts])
      out[base + '/initial_value'] = eq.initial_value()
      # equation of motion on random inputs
      rs = np.random.RandomState(seed + 100)
      y = rs.randn(2, n)
      derivs = {name: rs.randn(2, n) for name in eq.DERIVATIVE_NAMES}
      out[base + '/eom_y'] = y
      out[base + '/eom_derivs'] = np.stack(
          [derivs[name] for name in eq.DERIVATIVE_NAMES], axis=-1)
      y_t = eq.equation_of_motion(y, derivs)
      out[base + '/eom_out'] = y_t
      out[base + '/finalize_t0.7'] = eq.finalize_time_derivative(0.7, y_t)
      params = eq.params()
      out[base + '/params_json'] = np.array(json.dumps(params, sort_keys=True))
      fine = eq.to_fine()
      out[base + '/fine_num_points'] = np.array(fine.grid.solution_num_points)
      out[base + '/exact_type'] = np.array(type(eq.to_exact()).__name__)
      out[base + '/conservative_type'] = np.array(
          type(eq.to_conservative()).__name__)
  index['equations'] = eq_index

  rs = np.random.RandomState(0)
  y = rs.randn(10)
  out['staggered/y'] = y
  out['staggered/out_dx0.5'] = equations.staggered_first_derivative(y, 0.5)
  um, up = rs.randn(50), rs.randn(50)
  out['godunov/u_minus'] = um
  out['godunov/u_plus'] = up
  out['godunov/flux'] = equations.godunov_convective_flux(um, up)

  # -- integrate.odeint with reference physics + fixed polynomial stencils --
  # Differentiator assembled ONLY from reference functions: standard
  # coefficients from polynomials.coefficients on polynomials.regular_grid,
  # applied with np.roll in float64, then the reference equation_of_motion and
  # finalize_time_derivative.  (The reference's own PolynomialDifferentiator
  # needs a TF session.)
  class RollDifferentiator(integrate.Differentiator):
    def __init__(self, equation, accuracy_order):
      self.equation = equation
      self.stencils = []
      method = FV if equation.CONSERVATIVE else FD
      for d in equation.DERIVATIVE_ORDERS:
        grid = polynomials.regular_grid(equation.GRID_OFFSET, d,
                                        accuracy_order,
                                        equation.grid.solution_dx)
        taps = polynomials.coefficients(grid, method, d)
        # pad_periodic(center=True) + VALID correlation: tap i multiplies
        # u[x + i - ceil((G-1)/2)]
        left = -(-(len(taps) - 1) // 2)
        self.stencils.append((taps, left))

    def __call__(self, t, y):
      derivs = {}
      for name, (taps, left) in zip(self.equation.DERIVATIVE_NAMES,
                                    self.stencils):
        derivs[name] = sum(c * np.roll(y, -(i - left))
                           for i, c in enumerate(taps))
      y_t = self.equation.equation_of_motion(y, derivs)
      return self.equation.finalize_time_derivative(t, y_t)

  ode_index = []
  cases = [
      ('BurgersEquation', 32, 1, 0, 1, np.linspace(0, 1, 11)),
      ('ConservativeBurgersEquation', 64, 4, 2, 1, np.linspace(0, 0.5, 6)),
      ('KdVEquation', 64, 1, 1, 1, np.linspace(0, 0.05, 6)),
      ('ConservativeKdVEquation', 64, 4, 5, 1, np.linspace(0, 0.05, 6)),
      ('KSEquation', 64, 1, 4, 1, np.linspace(0, 0.02, 5)),
      ('ConservativeKSEquation', 64, 2, 7, 1, np.linspace(0, 0.02, 5)),
      ('BurgersEquation', 32, 1, 9, 3, np.linspace(0, 0.3, 4)),
  ]
  for cls_name, n, rf, seed, acc, times in cases:
    eq = getattr(equations, cls_name)(n, resample_factor=rf, random_seed=seed)
    diff = RollDifferentiator(eq, acc)
    y0 = eq.initial_value()
    y, nfev = integrate.odeint(y0, diff, times, method='RK23')
    base = 'odeint/{}/n{}/rf{}/s{}/a{}'.format(cls_name, n, rf, seed, acc)
    out[base + '/times'] = times
    out[base + '/y0'] = y0
    out[base + '/y'] = y
    out[base + '/nfev'] = np.array(nfev)
    out[base + '/rhs_t0.1_y0'] = diff(0.1, y0 + 0.1 * np.sin(eq.grid.solution_x))
    ode_index.append(base)
    # integrate.integrate wrapper (warmup=0): must match odeint
    ds = integrate.integrate(eq, diff, times=times)
    np.testing.assert_array_equal(ds.data_vars['y'][1], y)
    assert ds.coords['num_evals'] == nfev
  index['odeint'] = ode_index

  # -- spectral end-to-end (pure NumPy/SciPy in the reference) --------------
  eq = equations.KdVEquation(64, random_seed=0)
  times = np.linspace(0, 0.2, 5)
  ds = integrate.integrate_spectral(eq, times=times)
  out['spectral/kdv64/times'] = times
  out['spectral/kdv64/y'] = ds.data_vars['y'][1]
  out['spectral/kdv64/nfev'] = np.array(ds.coords['num_evals'])

  path = os.path.join(HERE, 'reference_numpy_paths.npz')
  np.savez_compressed(path, **out)
  with open(os.path.join(HERE, 'reference_numpy_paths.json'), 'w') as f:
    json.dump(index, f, indent=1)
  print('wrote', path, os.path.getsize(path), 'bytes;', len(out), 'arrays')


if __name__ == '__main__':
  main()
