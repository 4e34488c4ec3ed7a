"""Host equation definitions vs golden dumps of the reference's NumPy paths.

Also restates pde_superresolution/equations_test.py:30-40 (Grid) and
duckarray_test.py:32-54 (resampling tables).
"""
import json

import numpy as np
import pytest

from ddd1d_amd import duckarray, equations, polynomials, create_hparams


def test_grid_table():
  grid = equations.Grid(3, resample_factor=2, period=60)
  assert grid.solution_num_points == 3
  np.testing.assert_equal(grid.solution_x, [0, 20, 40])
  np.testing.assert_equal(grid.solution_dx, 20)
  assert grid.reference_num_points == 6
  np.testing.assert_equal(grid.reference_x, [0, 10, 20, 30, 40, 50])
  np.testing.assert_equal(grid.reference_dx, 10)


def test_resample_tables():
  np.testing.assert_allclose(duckarray.resample_mean(np.arange(6.0), 2),
                             [0.5, 2.5, 4.5])
  np.testing.assert_allclose(duckarray.subsample(np.arange(6), 2), [0, 2, 4])
  with pytest.raises(ValueError, match='must divide'):
    duckarray.resample_mean(np.arange(5.0), 2)
  with pytest.raises(ValueError, match='invalid axis'):
    duckarray.subsample(np.arange(6), 2, axis=3)


def test_resample_golden(golden):
  x = golden['duck/x']
  np.testing.assert_array_equal(duckarray.resample_mean(x, 4),
                                golden['duck/resample_mean_4'])
  np.testing.assert_array_equal(duckarray.subsample(x, 4),
                                golden['duck/subsample_4'])
  np.testing.assert_array_equal(duckarray.resample_mean(x.T, 3, axis=0),
                                golden['duck/resample_mean_axis0'])


def _build(key):
  _, cls_name, n, rf, seed = key.split('/')
  cls = getattr(equations, cls_name)
  return cls(int(n[1:]), resample_factor=int(rf[2:]), random_seed=int(seed[1:]))


def test_equations_golden(golden):
  keys = golden.index['equations']
  assert len(keys) == 27
  for key in keys:
    eq = _build(key)
    np.testing.assert_array_equal(eq.grid.solution_x, golden[key + '/solution_x'])
    np.testing.assert_array_equal(eq.grid.reference_x,
                                  golden[key + '/reference_x'])
    scalars = golden[key + '/scalars']
    got = np.array([eq.grid.solution_dx, eq.grid.reference_dx, eq.grid.period,
                    eq.time_step, eq.standard_deviation,
                    getattr(eq, 'eta', np.nan)])
    np.testing.assert_array_equal(got, scalars)
    np.testing.assert_array_equal(np.array(eq.DERIVATIVE_ORDERS),
                                  golden[key + '/derivative_orders'])
    assert bool(eq.CONSERVATIVE) == bool(golden[key + '/conservative'])
    assert (eq.GRID_OFFSET is polynomials.GridOffset.STAGGERED) == bool(
        golden[key + '/staggered'])
    # forcing draws and values: bit-identical
    for name in ('a', 'omega', 'k', 'phi'):
      np.testing.assert_array_equal(getattr(eq.forcing, name),
                                    golden[key + '/forcing_' + name])
    for t, want in zip(golden[key + '/forcing_t'],
                       golden[key + '/forcing_values']):
      np.testing.assert_array_equal(eq.forcing(t), want)
    np.testing.assert_array_equal(eq.initial_value(),
                                  golden[key + '/initial_value'])
    # equation of motion
    y = golden[key + '/eom_y']
    stacked = golden[key + '/eom_derivs']
    derivs = {name: stacked[..., i]
              for i, name in enumerate(eq.DERIVATIVE_NAMES)}
    y_t = eq.equation_of_motion(y, derivs)
    np.testing.assert_array_equal(y_t, golden[key + '/eom_out'])
    np.testing.assert_array_equal(eq.finalize_time_derivative(0.7, y_t),
                                  golden[key + '/finalize_t0.7'])
    assert json.dumps(eq.params(), sort_keys=True) == str(
        golden[key + '/params_json'])
    assert eq.to_fine().grid.solution_num_points == int(
        golden[key + '/fine_num_points'])
    assert type(eq.to_exact()).__name__ == str(golden[key + '/exact_type'])
    assert type(eq.to_conservative()).__name__ == str(
        golden[key + '/conservative_type'])


def test_staggered_and_godunov_golden(golden):
  np.testing.assert_array_equal(
      equations.staggered_first_derivative(golden['staggered/y'], 0.5),
      golden['staggered/out_dx0.5'])
  np.testing.assert_array_equal(
      equations.godunov_convective_flux(golden['godunov/u_minus'],
                                        golden['godunov/u_plus']),
      golden['godunov/flux'])


@pytest.mark.parametrize('equation', ['burgers', 'kdv', 'ks'])
@pytest.mark.parametrize('conservative,numerical_flux',
                         [(False, False), (True, False), (True, True)])
def test_from_hparams(equation, conservative, numerical_flux):
  hparams = create_hparams(equation, conservative=conservative,
                           numerical_flux=numerical_flux,
                           equation_kwargs=json.dumps({'num_points': 256}),
                           resample_factor=4)
  fine, coarse = equations.from_hparams(hparams, random_seed=5)
  assert coarse.grid.solution_num_points == 64
  assert coarse.grid.resample_factor == 4
  assert fine.grid.solution_num_points == 256
  assert fine.grid.resample_factor == 1
  assert type(fine) is type(coarse)
  table = (equations.EQUATION_TYPES if not conservative else
           equations.FLUX_EQUATION_TYPES if numerical_flux else
           equations.CONSERVATIVE_EQUATION_TYPES)
  assert type(coarse) is table[equation]
  assert coarse.random_seed == 5
  np.testing.assert_array_equal(fine.forcing.a, coarse.forcing.a)


def test_from_hparams_bad_factor():
  hparams = create_hparams('burgers', resample_factor=7,
                           equation_kwargs=json.dumps({'num_points': 256}))
  with pytest.raises(ValueError, match='does not divide'):
    equations.from_hparams(hparams)


def test_hparams_defaults_and_parse(tmp_path):
  from ddd1d_amd import hparams as hp
  h = create_hparams('burgers')
  # training.py:125-141
  assert (h.conservative, h.numerical_flux, h.resample_factor) == (True, False, 4)
  assert (h.model_target, h.num_layers, h.filter_size, h.kernel_size) == (
      'coefficients', 3, 32, 5)
  assert (h.nonlinearity, h.polynomial_accuracy_order,
          h.polynomial_accuracy_scale, h.coefficient_grid_min_size) == (
              'relu', 1, 1.0, 6)
  h.parse('num_layers=2,filter_size=16,learning_rates=[0.1,0.01],'
          'conservative=false,nonlinearity=tanh')
  assert h.num_layers == 2 and h.filter_size == 16
  assert h.learning_rates == [0.1, 0.01]
  assert h.conservative is False and h.nonlinearity == 'tanh'
  with pytest.raises(ValueError, match='unknown hyperparameter'):
    h.parse('not_a_key=1')
  hp.save_hparams(h, str(tmp_path))
  loaded = hp.load_hparams(str(tmp_path))
  lhs, rhs = h.values(), loaded.values()
  for k in lhs:
    if isinstance(lhs[k], list) and lhs[k] and lhs[k][0] != lhs[k][0]:
      assert rhs[k][0] != rhs[k][0]   # NaN round-trips as NaN
    else:
      assert lhs[k] == rhs[k], k
