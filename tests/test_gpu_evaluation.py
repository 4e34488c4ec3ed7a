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
  """All samples in one launch, one RK23 controller per sample on the device ==
  the reference's per-sample SciPy RK23 loop over the same HIP right-hand side
  (run_evaluation.py:152-174): equal nfev, same trajectory."""
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
    assert err < 1e-5
    assert one['num_evals'] == batch['num_evals'][seed]


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


# ---------------------------------------------------------------------------
# parity of the evaluation harness with the oracle (scripts/run_evaluation.py:
# 152-210, analysis.py:39-90): trajectories AND the scores built from them
# ---------------------------------------------------------------------------
def _oracle_forcing(hp, seeds):
  from ddd1d_amd import equations, model as model_lib
  eqs = [equations.from_hparams(hp, random_seed=s)[1] for s in seeds]
  return model_lib.forcing_from_equations(eqs)


def test_run_integrate_batch_matches_oracle_per_sample():
  """run_integrate_batch (HIP, all samples at once, per-seed forcing) against
  the oracle stepping every sample with the same fixed-step BS3 scheme."""
  import oracle
  samples = 6
  hp, model = _setup('burgers', samples)
  times = np.arange(0, 0.3 + 1e-9, 0.1)
  y0 = 0.3 * random_phase_ic(model.equation, samples)
  got = evaluation.run_integrate_batch(model, hp, y0, times, adaptive=False)
  forcing = _oracle_forcing(hp, range(samples))
  want = oracle.integrate_fixed(model.spec(), oracle.SCHEME_BS3, 0.0, 0.01, 30, 10, y0,
                                forcing=forcing)           # [time, sample, x]
  want = np.concatenate([y0[None], want], axis=0).transpose(1, 0, 2)
  assert got['y'].shape == want.shape == (samples, 4, 64)
  for s in range(samples):
    err = rel_err(got['y'][s], want[s])
    assert err < 1e-5, (s, err)
  np.testing.assert_array_equal(got['sample'], np.arange(samples))
  np.testing.assert_array_equal(got['num_evals'], 3 * 30)


def test_run_integrate_matches_oracle_rk23():
  """run_integrate: one sample, SciPy RK23 over the HIP right-hand side, vs the
  same controller over the NumPy right-hand side: same evaluations, same path."""
  import oracle
  hp, model = _setup('burgers', 3)
  times = np.arange(0, 0.3 + 1e-9, 0.1)
  y0 = 0.3 * random_phase_ic(model.equation, 3)
  for seed in (0, 2):
    one = evaluation.run_integrate((seed, y0[seed]), model, hp, times)
    forcing = {k: v[0] for k, v in _oracle_forcing(hp, [seed]).items()}
    want, nfev = oracle.odeint_rk23(model.spec(), y0[seed], times, forcing)
    assert one['num_evals'] == nfev
    assert rel_err(one['y'], want) < 1e-5


def test_evaluate_scores_match_scores_of_oracle_trajectories():
  """MAE at stop times and mostly-good survival from the HIP trajectories equal
  the ones computed from oracle trajectories of the same model."""
  import oracle
  samples, rf = 5, 4
  hp, model = _setup('kdv', samples, rf=rf)
  times = np.arange(0, 0.02 + 1e-9, 0.005)
  dt = 2.5e-5
  y0 = random_phase_ic(model.equation, samples)
  # "exact" data on the fine grid: a smooth, slowly drifting perturbation of the
  # initial condition, so errors cross the survival threshold part-way
  rs = np.random.RandomState(0)
  y_fine0 = np.repeat(y0, rf, axis=-1).astype(np.float64)
  drift = 200.0 * times[None, :, None] * rs.uniform(0.5, 1.5, size=(samples, 1, 1))
  y_exact = y_fine0[:, None, :] * (1.0 + drift)
  result = evaluation.evaluate(model, hp, y_exact, times, stop_times=(0.01, 0.02),
                               quantiles=(0.5, 0.8), max_step=dt, scheme='midpoint')
  y0_low = evaluation.load_initial_conditions(y_exact, rf)
  steps = int(round(times[1] / dt))
  traj = oracle.integrate_fixed(model.spec(), oracle.SCHEME_MIDPOINT, 0.0, dt,
                                steps * (len(times) - 1), steps,
                                y0_low.astype(np.float32))
  y_oracle = np.concatenate([y0_low.astype(np.float32)[None], traj], 0).transpose(1, 0, 2)
  assert rel_err(result['samples']['y'], y_oracle) < 1e-5
  models = {'y_model': y_oracle}
  mae = evaluation.mean_absolute_error(models, y_exact, times, (0.01, 0.02))['y_model']
  np.testing.assert_allclose(result['mae'], mae, rtol=1e-4, atol=1e-7)
  assert np.all(result['mae'] > 0)
  for qi, q in enumerate((0.5, 0.8)):
    surv = evaluation.mostly_good_survival(models, y_exact, times, q)['y_model']
    np.testing.assert_array_equal(result['survival'][qi], surv)
  # the drift makes samples fail at different times: the score is not trivial
  assert len(np.unique(result['survival'][0])) > 1 and result['survival'].max() < times.max()
