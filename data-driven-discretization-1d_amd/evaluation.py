"""Batched evaluation of a learned-stencil model against exact solutions.

What ``scripts/run_evaluation.py`` (136-221) and ``analysis.py`` (39-90)
compute, without Beam / xarray / netCDF: every sample of an exact data set is
integrated by the coarse model from its resampled initial condition, then the
mean absolute error up to each stop time and the "mostly good" survival time
are reported.

The reference integrates one sample per Beam worker with SciPy RK23 through a
TF session.  Here ``run_integrate`` keeps that execution shape (one sample, one
adaptive solve, HIP right-hand side) and ``run_integrate_batch`` advances all
samples together in one launch with the same adaptive RK23 -- one
SciPy-identical step-size controller per sample on the device
(``ddd_integrate_adaptive_f64``), so every sample gets the trajectory and the
``num_evals`` of its own ``solve_ivp`` call whatever its stiffness.
``adaptive=False`` selects the fixed-step Bogacki-Shampine scheme at
``max_step`` instead (equal only while the controller is saturated, as in
notebooks/time-integration.ipynb).  With more than one rank the samples are
sharded and gathered (``distributed``), the analogue of
``beam.CombineGlobally(ConcatCombineFn('sample'))`` (run_evaluation.py:218).

Arrays are plain NumPy: ``y_model`` [sample, time, x_low], ``y_exact``
[sample, time, x_high], ``times`` [time].
"""
from typing import Dict, Optional, Sequence

import numpy as np

from . import distributed
from . import duckarray
from . import equations as equations_lib
from . import integrate
from . import model as model_lib


# ---------------------------------------------------------------------------
# analysis.py
# ---------------------------------------------------------------------------
def unify_x_coords(y_low: np.ndarray, y_high: np.ndarray) -> np.ndarray:
  """High-resolution data block-averaged onto the low-resolution grid
  (analysis.unify_x_coords, analysis.py:39-53)."""
  factor = y_high.shape[-1] // y_low.shape[-1]
  return duckarray.resample_mean(y_high, factor)


def is_good(model, exact, max_error: float = 0.5):
  """Pointwise accuracy within ``max_error`` (analysis.py:56-62)."""
  return np.abs(model - exact) <= max_error


def mostly_good(model, exact, max_error: float = 0.5, frac_good: float = 0.8):
  """Per time: at least ``frac_good`` of the points accurate (analysis.py:65-72)."""
  return is_good(model, exact, max_error=max_error).mean(axis=-1) >= frac_good


def calculate_survival(good: np.ndarray, times: np.ndarray) -> np.ndarray:
  """"Lifetime" of a boolean [..., time] array: the first time it is False,
  the last time if it never is (analysis.py:75-79)."""
  good = np.asarray(good).astype(bool)
  times = np.asarray(times)
  first_bad = np.argmin(good, axis=-1)
  return np.where(good.all(axis=-1), times.max(), times[first_bad])


def mostly_good_survival(y_models: Dict[str, np.ndarray], y_exact: np.ndarray,
                         times: np.ndarray, quantile: float = 0.8
                         ) -> Dict[str, np.ndarray]:
  """Survival time per sample of every model variable (analysis.py:82-90).

  The error threshold is the (1 - quantile) quantile of |y_exact| at full
  resolution; the comparison happens on the low-resolution grid.
  """
  max_error = float(np.quantile(np.abs(y_exact), 1 - quantile))
  out = {}
  for name, y_model in y_models.items():
    exact_low = unify_x_coords(y_model, y_exact)
    good = mostly_good(y_model, exact_low, max_error=max_error, frac_good=quantile)
    out[name] = calculate_survival(good, times)
  return out


def mean_absolute_error(y_models: Dict[str, np.ndarray], y_exact: np.ndarray,
                        times: np.ndarray, stop_times: Sequence[float]
                        ) -> Dict[str, np.ndarray]:
  """MAE over x and over times <= each stop time, per sample; NaNs propagate
  (run_evaluation.py:196-204).  Returns name -> [time_max, sample]."""
  times = np.asarray(times)
  out = {}
  for name, y_model in y_models.items():
    exact_low = unify_x_coords(y_model, y_exact)
    rows = []
    for time_max in stop_times:
      keep = times <= time_max
      rows.append(np.abs(y_model[:, keep] - exact_low[:, keep]).mean(axis=(1, 2)))
    out[name] = np.stack(rows)
  return out


# ---------------------------------------------------------------------------
# run_evaluation.py
# ---------------------------------------------------------------------------
def load_initial_conditions(y_exact: np.ndarray, resample_factor: int,
                            num_samples: Optional[int] = None) -> np.ndarray:
  """t = 0 of the exact data block-averaged to the model grid
  (run_evaluation.py:136-150)."""
  initial_conditions = duckarray.resample_mean(y_exact[:, 0, :], resample_factor)
  if np.isnan(initial_conditions).any():
    raise ValueError('initial conditions cannot have NaNs')
  if num_samples is not None and y_exact.shape[0] != num_samples:
    raise ValueError('invalid number of samples in exact dataset')
  return initial_conditions


def run_integrate(seed_and_initial_condition, model: model_lib.LearnedStencilModel,
                  hparams, times: np.ndarray, warmup: float = 0,
                  integrate_method: str = 'RK23'):
  """One sample, SciPy adaptive stepping, HIP right-hand side
  (run_evaluation.py:152-174)."""
  random_seed, y0 = seed_and_initial_condition
  _, equation_coarse = equations_lib.from_hparams(hparams, random_seed=random_seed)
  differentiator = integrate.SavedModelDifferentiator(None, equation_coarse,
                                                      hparams, model=model)
  solution, num_evals = integrate.odeint(y0, differentiator, warmup + times,
                                         method=integrate_method)
  return dict(y=solution, time=warmup + times,
              x=equation_coarse.grid.solution_x, num_evals=num_evals,
              sample=random_seed)


def run_integrate_batch(model: model_lib.LearnedStencilModel, hparams,
                        initial_conditions: np.ndarray, times: np.ndarray,
                        warmup: float = 0, max_step: float = 0.01,
                        scheme: str = 'bs3', first_seed: int = 0,
                        adaptive: Optional[bool] = None):
  """All samples of this rank together: adaptive RK23 with one controller per
  sample (``adaptive=True``; the default for scheme='bs3'), or the fixed step
  ``max_step`` with ``scheme``.

  Sample i uses random_seed = first_seed + i for its forcing, like the
  reference's per-seed equations.  With torch.distributed initialised the
  samples are split across ranks and the trajectories gathered on every rank.
  Returns dict(y [sample, time, x], time, x, num_evals, sample).
  """
  times = np.asarray(times, dtype=np.float64)
  total = initial_conditions.shape[0]
  rank, _, world = distributed.world_info()
  lo, hi = distributed.shard_bounds(total, rank, world)
  seeds = range(first_seed + lo, first_seed + hi)
  forcing = None
  if model.equation.has_time_dependent_forcing and hi > lo:
    eqs = [equations_lib.from_hparams(hparams, random_seed=s)[1] for s in seeds]
    forcing = model_lib.forcing_from_equations(eqs)
  if adaptive is None:   # RK23's own tableau: the reference's integrator, on the device
    adaptive = scheme == 'bs3'
  if hi > lo:
    ds = integrate.integrate_batch(model, initial_conditions[lo:hi], warmup + times,
                                   dt=max_step, scheme=scheme, forcing=forcing,
                                   adaptive=adaptive)
    y_local = _data(ds, 'y')
    evals_local = np.broadcast_to(np.asarray(_coord(ds, 'num_evals'), dtype=np.int64),
                                  (hi - lo,)).copy()
  else:
    dtype = np.float64 if adaptive else np.float32
    y_local = np.zeros((0, len(times), initial_conditions.shape[1]), dtype)
    evals_local = np.zeros(0, np.int64)
  y = y_local
  num_evals = evals_local
  if world > 1:
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
      raise RuntimeError('WORLD_SIZE > 1 but torch.distributed is not initialised')
    device = 'cuda' if dist.get_backend() == 'nccl' else 'cpu'
    y = distributed.gather_states(torch.from_numpy(np.ascontiguousarray(y_local)).to(device),
                                  total).cpu().numpy()
    num_evals = distributed.gather_states(
        torch.from_numpy(evals_local).to(device), total).cpu().numpy()
  return dict(y=y, time=warmup + times, x=model.equation.grid.solution_x,
              num_evals=num_evals, sample=first_seed + np.arange(total))


def evaluate(model: model_lib.LearnedStencilModel, hparams, y_exact: np.ndarray,
             times: np.ndarray, stop_times: Sequence[float] = (5, 10, 20, 40),
             quantiles: Sequence[float] = (0.8, 0.9, 0.95), warmup: float = 0,
             batched: bool = True, **kwargs):
  """Integrate every sample and score it (run_evaluation.py:181-216).

  ``y_exact`` [sample, time, x_high] holds the exact solution at ``times``.
  Returns dict(samples=..., mae [time_max, sample], survival [quantile, sample]).
  """
  y0 = load_initial_conditions(y_exact, hparams.resample_factor)
  if batched:
    samples = run_integrate_batch(model, hparams, y0, times, warmup=warmup, **kwargs)
  else:
    rows = [run_integrate((seed, y0[seed]), model, hparams, times, warmup=warmup, **kwargs)
            for seed in range(y0.shape[0])]
    samples = dict(y=np.stack([r['y'] for r in rows]), time=rows[0]['time'],
                   x=rows[0]['x'], num_evals=np.array([r['num_evals'] for r in rows]),
                   sample=np.array([r['sample'] for r in rows]))
  models = {'y_model': samples['y']}
  mae = mean_absolute_error(models, y_exact, samples['time'], stop_times)['y_model']
  survival = np.stack([
      mostly_good_survival(models, y_exact, samples['time'], q)['y_model']
      for q in quantiles])
  return dict(samples=samples, mae=mae, stop_times=np.asarray(stop_times),
              survival=survival, quantiles=np.asarray(quantiles))


def _data(ds, name):
  v = ds.data_vars[name]
  return np.asarray(v[1] if isinstance(v, tuple) else v)


def _coord(ds, name):
  v = ds.coords[name]
  return v[1] if isinstance(v, tuple) else v
