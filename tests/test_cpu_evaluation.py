"""analysis.py / run_evaluation.py scoring functions (host NumPy)."""
import numpy as np
import pytest

from ddd1d_amd import evaluation


@pytest.mark.parametrize('data,expected', [
    (np.arange(100) < 65, 6.5), (np.ones(100), 9.9), (np.zeros(100), 0),
    (np.concatenate([np.ones(10), np.zeros(1), np.ones(9), np.zeros(80)]), 1)])
def test_calculate_survival_known_answers(data, expected):
  """The table of the reference's analysis_test.py:31-43."""
  times = np.arange(100) / 10
  assert evaluation.calculate_survival(data, times) == expected


def test_is_good_and_mostly_good():
  exact = np.zeros((2, 3, 10))
  model = np.zeros((2, 3, 10))
  model[0, 1, :3] = 1.0            # 30 % of the points off at one time
  model[1, 2, :1] = 0.4            # inside the threshold
  assert evaluation.is_good(model, exact).sum() == 60 - 3
  good = evaluation.mostly_good(model, exact, max_error=0.5, frac_good=0.8)
  np.testing.assert_array_equal(good, [[True, False, True], [True, True, True]])


def test_mae_and_survival_on_synthetic_runs():
  times = np.array([0.0, 1.0, 2.0, 3.0])
  rs = np.random.RandomState(0)
  y_exact = rs.randn(3, 4, 32)                                 # [sample, time, x_high]
  low = evaluation.unify_x_coords(np.zeros((3, 4, 8)), y_exact)
  np.testing.assert_allclose(low, y_exact.reshape(3, 4, 8, 4).mean(-1))
  y_model = low.copy()
  y_model[1, 2:] += 10.0                                       # sample 1 fails from t = 2
  y_model[2, 3, 0] = np.nan                                    # sample 2 diverges at t = 3
  mae = evaluation.mean_absolute_error({'y_model': y_model}, y_exact, times, [1.0, 3.0])['y_model']
  assert mae.shape == (2, 3)
  np.testing.assert_allclose(mae[:, 0], 0, atol=1e-15)
  np.testing.assert_allclose(mae[0, 1], 0, atol=1e-15)
  np.testing.assert_allclose(mae[1, 1], 10.0 * 2 / 4)
  assert mae[0, 2] == 0 and np.isnan(mae[1, 2])                # skipna=False
  surv = evaluation.mostly_good_survival({'y_model': y_model}, y_exact, times, 0.8)['y_model']
  np.testing.assert_array_equal(surv[:2], [3.0, 2.0])
  y0 = evaluation.load_initial_conditions(y_exact, 4, num_samples=3)
  np.testing.assert_allclose(y0, low[:, 0])
  with pytest.raises(ValueError, match='number of samples'):
    evaluation.load_initial_conditions(y_exact, 4, num_samples=5)
  bad = y_exact.copy(); bad[0, 0, 0] = np.nan
  with pytest.raises(ValueError, match='NaNs'):
    evaluation.load_initial_conditions(bad, 4)
