"""run_evaluation on the GPU: all samples advanced together vs the reference's
one-sample-at-a-time SciPy loop, and the scores built from them."""
import json

import numpy as np
import pytest

import ddd1d_amd
from helpers import make_model, random_phase_ic, rel_err
from ddd1d_amd import evaluation

pytestmark = pytest.mark.gpu


def _setup(equation, samples, rf=4):
  hp = ddd1d_amd.create_hparams(equation, conservative=True, resample_factor=rf,
                                equation_kwargs=json.dumps({'num_points': 64 * rf}))
  model = make_model(equation, True, num_points=64, resample_factor=rf)
  return hp, model


def test_batched_run_matches_per_sample_rk23():
  """Fixed-step BS3 at max_step over the batch == SciPy RK23 per sample while
  the controller is saturated (smooth Burgers states, run_evaluation.py:152-174)."""
  hp, model = _setup('burgers', 6)
  times = np.arange(0, 0.5 + 1e-9, 0.1)
  y0 = 0.3 * random_phase_ic(model.equation, 6)
  batch = evaluation.run_integrate_batch(model, hp, y0, times)
  assert batch['y'].shape == (6, 6, 64)
  np.testing.assert_array_equal(batch['y'][:, 0], y0)
  for seed in (0, 3, 5):
    one = evaluation.run_integrate((seed, y0[seed]), model, hp, times)
    assert one['sample'] == seed and one['y'].shape == (6, 64)
    err = rel_err(batch['y'][seed], one['y'])
    print('seed', seed, 'batched vs RK23 rel err {:.2e}'.format(err), 'nfev', one['num_evals'])
    assert err < 1e-3            # two different (both 3rd order) step sequences
    assert abs(one['num_evals'] - batch['num_evals'][seed]) <= 0.1 * one['num_evals']


def test_evaluate_scores_a_perfect_and_a_broken_model():
  hp, model = _setup('kdv', 5)
  times = np.arange(0, 0.02 + 1e-9, 0.005)
  y0 = random_phase_ic(model.equation, 5)
  run = evaluation.run_integrate_batch(model, hp, y0, times, max_step=2.5e-5, scheme='midpoint')
  # "exact" data = the model's own trajectories on a 4x finer grid
  y_exact = np.repeat(run['y'], 4, axis=-1).astype(np.float64)
  result = evaluation.evaluate(model, hp, y_exact, times, stop_times=(0.01, 0.02),
                               quantiles=(0.8, 0.9), max_step=2.5e-5, scheme='midpoint')
  assert result['mae'].shape == (2, 5) and result['survival'].shape == (2, 5)
  np.testing.assert_allclose(result['mae'], 0, atol=1e-6)
  np.testing.assert_array_equal(result['survival'], times.max())
  # break the reference after t = 0.01: survival stops there, MAE grows
  y_bad = y_exact.copy()
  y_bad[:, 3:] += 5.0
  result = evaluation.evaluate(model, hp, y_bad, times, stop_times=(0.01, 0.02),
                               quantiles=(0.8,), max_step=2.5e-5, scheme='midpoint')
  np.testing.assert_allclose(result['mae'][0], 0, atol=1e-6)
  np.testing.assert_allclose(result['mae'][1], 5.0 * 2 / 5, atol=1e-5)
  np.testing.assert_array_equal(result['survival'][0], times[3])
