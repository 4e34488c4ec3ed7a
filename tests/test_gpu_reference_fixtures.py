"""Reference-generated fixtures pushed STRAIGHT through the GPU entry points
(no oracle in between): the arrays tests/golden/make_golden.py dumped from the
imported reference (pde_superresolution/equations.py run on NumPy inputs).

  eom_y, eom_derivs -> ddd_apply_space_derivatives == eom_out
      (Equation.equation_of_motion of all nine equations, equations.py:269-587)
  forcing_{a,omega,k,phi}, forcing_t -> a forced fixed-stencil
      ddd_time_derivative minus the same model unforced == forcing_values
      (RandomForcing.__call__ inside finalize_time_derivative,
      equations.py:214-219, 276-277), in both forcing code paths of the kernels
      (harmonic sums on the MFMA-path kernels, per-point sines on the generic one)
"""
import numpy as np
import pytest

from helpers import rel_err
from ddd1d_amd import equations, model as model_lib

pytestmark = pytest.mark.gpu


def _build(key):
  _, cls_name, n, rf, seed = key.split('/')
  cls = getattr(equations, cls_name)
  return cls_name, cls(int(n[1:]), resample_factor=int(rf[2:]), random_seed=int(seed[1:]))


def test_equation_of_motion_fixtures_through_the_gpu(golden):
  """float32 kernel vs the reference's float64 output: 2e-6 of the largest
  value (the bound the float32 NumPy restatement meets, test_cpu_oracle.py)."""
  seen = set()
  for key in golden.index['equations']:
    cls_name, eq = _build(key)
    y = golden[key + '/eom_y']
    derivs = golden[key + '/eom_derivs']          # [batch, x, derivative]
    want = golden[key + '/eom_out']
    got = model_lib.apply_space_derivatives(
        derivs.astype(np.float32), y.astype(np.float32), eq).cpu().numpy()
    err = rel_err(got, want)
    assert got.shape == want.shape and err < 2e-6, (key, err)
    seen.add(cls_name)
  assert len(seen) == 9, seen   # every equation of equations.py:590-606


@pytest.mark.parametrize('kernel', ['auto', 'generic'])
def test_forcing_fixtures_through_the_gpu(golden, kernel):
  """forcing(t) isolated as f(t, y) - f_unforced(y) on a fixed-stencil model.
  3e-5 absolute: the float32 rounding of phases up to ~50 rad (ulp/2 = 2e-6 per
  mode, 20 modes of amplitude <= 0.5), the bound derived for the float32 NumPy
  restatement in test_cpu_oracle.py."""
  checked = 0
  for key in golden.index['equations']:
    cls_name, eq = _build(key)
    if not eq.has_time_dependent_forcing:
      continue
    model = model_lib.BaselineModel(eq, accuracy_order=1)
    model.set_kernel(kernel)
    rs = np.random.RandomState(len(key))
    y = rs.uniform(-1, 1, size=(1, eq.grid.solution_num_points)).astype(np.float32)
    plain = model.time_derivative(y, 0.0).cpu().numpy().astype(np.float64)
    model.set_forcing_from_equation(batch=1)
    worst = 0.0
    for t, want in zip(golden[key + '/forcing_t'], golden[key + '/forcing_values']):
      forced = model.time_derivative(y, float(t)).cpu().numpy().astype(np.float64)
      # the subtraction costs one float32 rounding of the (larger) forced value
      slack = 2.0 ** -23 * np.abs(forced).max()
      err = np.abs((forced - plain)[0] - want).max()
      worst = max(worst, err)
      assert err < 3e-5 + slack, (key, kernel, t, err)
    print(key, model.kernel_name, 'worst abs forcing error {:.1e}'.format(worst))
    checked += 1
  assert checked >= 6
