"""Time integration drivers with the reference's API, backed by HIP kernels.

Mirror of ``pde_superresolution/integrate.py`` for the learned-stencil path:

  Differentiator                 integrate.py:40-45
  SavedModelDifferentiator       integrate.py:48-71   (HIP model instead of a TF session)
  PolynomialDifferentiator       integrate.py:74-105
  odeint                         integrate.py:143-169 (SciPy RK23, max_step 0.01; over a HIP
                                 differentiator the same controller runs on the device)
  integrate_exact_batch          integrate_exact for a batch, entirely on the device
  SpectralDifferentiator         integrate.py:108-121 (float64 circulant kernel)
  WENODifferentiator             integrate.py:124-140
  odeint_with_periodic_filtering integrate.py:172-212
  exact_differentiator           integrate.py:215-235
  integrate                      integrate.py:238-279 (warm-up + filtering)
  integrate_exact / _weno / _spectral  integrate.py:282-341
  integrate_baseline             integrate.py:296-308
  integrate_exact_baseline_and_model integrate.py:344-396
  integrate_model_from_warm_start integrate.py:399-427

plus ``integrate_batch``: the whole batch of independent initial conditions
advanced on the GPU with a fixed step -- what ``scripts/run_evaluation.py``'s
per-sample loop (run_evaluation.py:152-174) becomes on one MI355X.

Results are ``xarray.Dataset`` objects when xarray is importable, otherwise a
``Dataset`` stand-in exposing the same ``data_vars`` / ``coords`` mapping.
"""
import logging
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from . import duckarray
from . import equations as equations_lib
from . import hparams as hparams_lib
from . import model as model_lib

_DEFAULT_TIMES = np.linspace(0, 10, num=201)

try:  # pragma: no cover - xarray is optional
  import xarray
  _HAVE_XARRAY = True
except ImportError:  # pragma: no cover
  xarray = None
  _HAVE_XARRAY = False


class Dataset(dict):
  """Minimal stand-in for xarray.Dataset: ds['y'] -> ndarray, ds.coords[...]."""

  def __init__(self, data_vars, coords):
    super(Dataset, self).__init__()
    self.dims = {}
    for name, (dims, values) in data_vars.items():
      values = np.asarray(values)
      self[name] = values
      self.dims.update(dict(zip(dims, values.shape)))
    self.data_vars = {k: v for k, v in self.items()}
    self.coords = dict(coords)


def _make_dataset(data_vars, coords):
  if _HAVE_XARRAY:
    return xarray.Dataset(data_vars=data_vars, coords=coords)
  return Dataset(data_vars, coords)


class Differentiator(object):
  """Base class: ``__call__(t, y[x]) -> dy/dt[x]``."""

  def __call__(self, t: float, y: np.ndarray) -> np.ndarray:
    raise NotImplementedError


class _HipDifferentiator(Differentiator):
  """One sample at a time through the batched kernel (batch = 1).

  Like the reference's TF differentiators the input is cast to float32, the
  derivative is computed on the device and returned to the host (here as
  float32 NumPy, which SciPy promotes to float64).
  """

  def __init__(self, device_model):
    self.model = device_model
    self._torch = _lib.require_gpu()
    self._pinned = None   # (input, output) page-locked host rows the kernel reads / writes

  @property
  def device_model(self):
    """The model whose on-device RK23 equals solve_ivp over this differentiator."""
    return self.model

  def __call__(self, t: float, y: np.ndarray) -> np.ndarray:
    # One sample per call is all latency: the state goes into a page-locked host
    # row the kernel reads directly (and writes its result next to), so a call is
    # one launch and one stream synchronisation -- no staging copies, no
    # allocations (256 B each way over the host link).
    torch = self._torch
    if self._pinned is None:
      n = self.model.num_points
      rows = torch.empty((2, 1, n), dtype=torch.float32).pin_memory()
      self._pinned = (rows[0], rows[1], rows[0].numpy(), rows[1].numpy())
    y_in, dydt, y_np, dydt_np = self._pinned
    y = np.asarray(y)
    if y.shape != (y_np.shape[1],):
      raise ValueError('solution has unexpected size for equation: {} vs {}'
                       .format(y.shape, y_np.shape[1]))
    y_np[0, :] = y   # cast to float32, as the reference feeds its float32 placeholder
    self.model.time_derivative_rows(y_in, dydt, t)
    torch.cuda.current_stream().synchronize()
    return dydt_np[0].copy()


class SavedModelDifferentiator(_HipDifferentiator):
  """Derivatives from a saved learned-stencil model (integrate.py:48-71).

  ``checkpoint_dir`` holds ``hparams.json`` + ``model.npz`` written by
  ``LearnedStencilModel.save``; alternatively pass an in-memory ``model``.
  The equation's own RandomForcing is applied in finalize_time_derivative,
  exactly as the reference does for the Burgers family.
  """

  def __init__(self, checkpoint_dir: Optional[str], equation, hparams=None,
               model: Optional[model_lib.LearnedStencilModel] = None):
    if model is None:
      model = model_lib.LearnedStencilModel.load(checkpoint_dir, equation,
                                                 hparams)
    elif model.equation is not equation:
      model = model_lib.LearnedStencilModel(
          equation, model.hparams, model.conv_kernels, model.conv_biases,
          model.nullspaces, model.biases, model.constant_coefficients)
    if equation.has_time_dependent_forcing:
      model.set_forcing_from_equation(batch=1)
    super(SavedModelDifferentiator, self).__init__(model)


class PolynomialDifferentiator(_HipDifferentiator):
  """Standard finite difference / volume stencils (integrate.py:74-105).

  ``accuracy_order=None`` is the reference's "best baseline"
  (model.py:69-95): the 6-point stencil, WENO5, or -- for the equations whose
  exact method is spectral -- ``duckarray.spectral_derivative``, which here
  runs on the float64 circulant kernel (``model.SpectralModel('rfft')``; the
  reference evaluates that branch in a float32 TF graph).
  """

  def __init__(self, equation, accuracy_order: Optional[int] = 1):
    self.equation = equation
    self._spectral = (
        accuracy_order is None and
        equation.EXACT_METHOD is equations_lib.ExactMethod.SPECTRAL)
    if self._spectral:
      if equation.exact_type() is not type(equation):
        raise AssertionError('the best baseline needs an exact equation type '
                             '(model.py:70)')
      self.model = model_lib.SpectralModel(equation, convention='rfft')
      self._torch = _lib.require_gpu()
      return
    model = model_lib.BaselineModel(equation, accuracy_order)
    if equation.has_time_dependent_forcing:
      model.set_forcing_from_equation(batch=1)
    super(PolynomialDifferentiator, self).__init__(model)

  @property
  def device_model(self):
    if self._spectral and self.equation.has_time_dependent_forcing:
      return None   # finalize_time_derivative runs on the host
    return self.model

  def __call__(self, t: float, y: np.ndarray) -> np.ndarray:
    if not self._spectral:
      return super(PolynomialDifferentiator, self).__call__(t, y)
    y64 = np.ascontiguousarray(np.asarray(y, dtype=np.float64)[np.newaxis, :])
    y_t = self.model.time_derivative(y64, t)[0].cpu().numpy()
    return self.equation.finalize_time_derivative(t, y_t)

  def calculate_space_derivatives(self, y):
    if self._spectral:
      return {name: duckarray.spectral_derivative(np.asarray(y, np.float64), order,
                                                  self.equation.grid.period)
              for name, order in zip(self.equation.DERIVATIVE_NAMES,
                                     self.equation.DERIVATIVE_ORDERS)}
    y32 = np.ascontiguousarray(np.asarray(y, dtype=np.float32)[np.newaxis, :])
    derivs = self.model.space_derivatives(y32)[0].cpu().numpy()
    return {name: derivs[:, i]
            for i, name in enumerate(self.equation.DERIVATIVE_NAMES)}


class SpectralDifferentiator(Differentiator):
  """Derivatives from a spectral method, float64 (integrate.py:108-121).

  Space derivatives and the equation of motion run on the GPU
  (``model.SpectralModel``: the circulant form of ``scipy.fftpack.diff``);
  ``finalize_time_derivative`` (the Burgers forcing) is applied on the host in
  float64 exactly as the reference does.
  """

  def __init__(self, equation):
    self.equation = equation
    self.model = model_lib.SpectralModel(equation, convention='fftpack')
    self._torch = _lib.require_gpu()

  @property
  def device_model(self):
    if self.equation.has_time_dependent_forcing:
      return None   # finalize_time_derivative (the forcing) runs on the host in float64
    return self.model

  def __call__(self, t: float, y: np.ndarray) -> np.ndarray:
    y64 = np.ascontiguousarray(np.asarray(y, dtype=np.float64)[np.newaxis, :])
    time_derivative = self.model.time_derivative(y64, t)[0].cpu().numpy()
    return self.equation.finalize_time_derivative(t, time_derivative)


class WENODifferentiator(_HipDifferentiator):
  """Fifth-order WENO for the Godunov-flux equations (integrate.py:124-140).

  ``u_minus`` / ``u_plus`` are WENO5 reconstructions, every other derivative a
  polynomial stencil of ``non_weno_accuracy_order``; Godunov flux, staggered
  difference and forcing follow -- all in one kernel (rhs_generic.h).
  """

  def __init__(self, equation, non_weno_accuracy_order: int = 3):
    if not ('u_minus' in equation.DERIVATIVE_NAMES and
            'u_plus' in equation.DERIVATIVE_NAMES):
      raise AssertionError('WENO needs u_minus and u_plus (integrate.py:136)')
    model = model_lib.BaselineModel(equation, non_weno_accuracy_order,
                                    weno=True)
    if equation.has_time_dependent_forcing:
      model.set_forcing_from_equation(batch=1)
    super(WENODifferentiator, self).__init__(model)
    self.equation = equation


def exact_differentiator(equation) -> Differentiator:
  """The "exact" differentiator of an exact equation type (integrate.py:215-235)."""
  if type(equation.to_exact()) is not type(equation):
    raise TypeError('an exact equation must be provided')
  method = equation.EXACT_METHOD
  if method is equations_lib.ExactMethod.POLYNOMIAL:
    return PolynomialDifferentiator(equation, accuracy_order=None)
  if method is equations_lib.ExactMethod.SPECTRAL:
    return SpectralDifferentiator(equation)
  if method is equations_lib.ExactMethod.WENO:
    return WENODifferentiator(equation)
  raise TypeError('unexpected equation: {}'.format(equation))


# ---------------------------------------------------------------------------
# Solvers.  The reference has one execution shape -- SciPy on the host driving a
# Differentiator, one sample at a time.  Here the same drivers (plain solve,
# segmented solve with the low-pass filter between segments, warm-up) are
# written once over a small solver interface with two implementations: the
# host one (SciPy, any Differentiator, one sample) and the device one (the
# batched on-device RK23 of ddd_integrate_adaptive_f64 + the circulant filter
# kernel, whole batches, nothing leaves the GPU between segments).
# ---------------------------------------------------------------------------
# One-sample solves over a HIP differentiator run SciPy's RK23 on the device
# (ddd_integrate_adaptive_f64 with batch 1: same evaluations, same trajectory,
# one launch instead of one host round trip per evaluation).  False restores
# the literal reference shape: SciPy on the host calling the differentiator.
DEVICE_ODEINT = True


class _HostSolver(object):
  """scipy.integrate.solve_ivp(max_step=0.01) over a Differentiator: one sample,
  state [x], trajectories [time, x] (integrate.py:143-169)."""

  def __init__(self, differentiator: Differentiator, method: str = 'RK23'):
    self.differentiator = differentiator
    self.method = method

  def solve(self, y0, times):
    device_model = getattr(self.differentiator, 'device_model', None)
    if DEVICE_ODEINT and self.method == 'RK23' and device_model is not None:
      y, nfev, _ = device_model.integrate_adaptive(
          np.asarray(y0, dtype=np.float64)[np.newaxis], np.asarray(times, dtype=np.float64),
          max_step=0.01)
      return y[:, 0].cpu().numpy(), int(nfev[0])
    import scipy.integrate
    logging.info('solve_ivp from %s to %s', times[0], times[-1])
    sol = scipy.integrate.solve_ivp(self.differentiator, (times[0], times[-1]), y0,
                                    t_eval=times, max_step=0.01, method=self.method)
    logging.info('nfev: %r, njev: %r, nlu: %r; status: %r, message: %s', sol.nfev,
                 sol.njev, sol.nlu, sol.status, sol.message)
    y = sol.y.T   # (time, x)
    if y.shape[0] < len(times):   # the solver gave up: NaN rows, not an exception
      logging.info('padding with %s values', len(times) - y.shape[0])
      y = np.concatenate([y, np.full((len(times) - y.shape[0], y.shape[1]), np.nan)])
    return y, sol.nfev

  @staticmethod
  def smooth(y, order):
    return duckarray.smoothing_filter(y, order=order)

  @staticmethod
  def last(y):
    return y[-1]

  @staticmethod
  def join(pieces):
    return np.concatenate(pieces, axis=0)


class _DeviceSolver(object):
  """The same for a batch on the GPU: state [batch, x] and trajectories
  [time, batch, x] are device tensors, nfev is a per-sample device tensor."""

  def __init__(self, device_model, max_step: float = 0.01):
    self.model = device_model
    self.max_step = max_step
    self._filters = {}

  def solve(self, y0, times):
    y, nfev, _ = self.model.integrate_adaptive(y0, times, max_step=self.max_step)
    return y, nfev.to(_lib._torch().int64)

  def smooth(self, y, order):
    n = y.shape[-1]
    if order not in self._filters:   # the filter applied to a unit impulse, uploaded once
      impulse = np.zeros(n)
      impulse[0] = 1.0
      self._filters[order] = _lib.as_device(duckarray.smoothing_filter(impulse, order=order))
    return _lib.circulant_apply(self._filters[order], y)

  @staticmethod
  def last(y):
    return y[-1]

  @staticmethod
  def join(pieces):
    return _lib._torch().cat(pieces, dim=0)


def _filter_segments(times: np.ndarray, filter_interval: float):
  """Index ranges [start - 1, stop) of ``times`` between consecutive filter
  times (integrate.py:191-199); every filter time must be an output time."""
  boundaries = np.arange(times[0], times[-1] + 1e-8, filter_interval)
  if not np.isin(boundaries, times).all():
    raise ValueError('all times in filter_interval must be sampled')
  cuts = np.searchsorted(times, boundaries, side='right')
  return [(int(a) - 1, int(b)) for a, b in zip(cuts[:-1], cuts[1:])]


def _solve_with_filtering(solver, y0, times, filter_interval, filter_order):
  """integrate.py:172-212 over either solver: integrate segment by segment,
  pass the state through the smoothing filter between segments (spectral methods
  for hyperbolic problems alias) and filter the saved trajectory once at the end
  -- filtering every saved step during integration would add noise."""
  pieces = [y0[None]]
  num_evals = 0
  for first, stop in _filter_segments(times, filter_interval):
    segment, evals = solver.solve(y0, times[first:stop])
    pieces.append(segment[1:])   # its first row repeats y0
    y0 = solver.smooth(solver.last(segment), filter_order)
    num_evals = num_evals + evals
  y = solver.join(pieces)
  assert y.shape[0] == times.size
  return solver.smooth(y, filter_order), num_evals


def _solve(solver, equation, initial_state, times, warmup, filter_interval,
           filter_all_times, exact_solver=None, resample=None):
  """integrate.py:238-279 over either solver: optional warm-up with the exact
  solver (filtered every ``filter_interval`` if given), then the run itself."""
  filter_order = (max(equation.to_exact().DERIVATIVE_ORDERS)
                  if filter_interval is not None else None)

  def run(which, y0, run_times, filtered):
    if filtered:
      return _solve_with_filtering(which, y0, run_times, filter_interval, filter_order)
    return which.solve(y0, run_times)

  if warmup:
    warmup_times = (np.arange(0, warmup + 1e-8, filter_interval)
                    if filter_interval is not None else np.array([0, warmup]))
    warm, _ = run(exact_solver, initial_state, warmup_times, filter_interval is not None)
    # the state after warm-up, on this equation's grid, starts the run
    y0 = resample(exact_solver.last(warm))
  else:
    y0 = initial_state
  return run(solver, y0, warmup + times, filter_all_times and filter_interval is not None)


def odeint(y0: np.ndarray, differentiator: Differentiator, times: np.ndarray,
           method: str = 'RK23') -> Tuple[np.ndarray, int]:
  """integrate.py:143-169: SciPy solve_ivp, max_step 0.01, NaN rows on failure."""
  return _HostSolver(differentiator, method).solve(y0, times)


def odeint_with_periodic_filtering(y0: np.ndarray,
                                   differentiator: Differentiator,
                                   times: np.ndarray, filter_interval: float,
                                   filter_order: int, method: str = 'RK23'):
  """integrate.py:172-212: segments of ``odeint`` with ``duckarray.smoothing_filter``
  between them and over the saved trajectory."""
  return _solve_with_filtering(_HostSolver(differentiator, method), y0, times,
                               filter_interval, filter_order)


def integrate(equation, differentiator: Differentiator,
              times: np.ndarray = _DEFAULT_TIMES, warmup: float = 0,
              integrate_method: str = 'RK23', filter_interval: float = None,
              filter_all_times: bool = False):
  """Integrate with optional exact warm-up and periodic filtering
  (integrate.py:238-279)."""
  exact_solver = None
  initial_state = None
  if warmup:
    equation_exact = equation.to_exact()
    exact_solver = _HostSolver(exact_differentiator(equation_exact), integrate_method)
    initial_state = equation_exact.initial_value()
  else:
    initial_state = equation.initial_value()
  solution, num_evals = _solve(
      _HostSolver(differentiator, integrate_method), equation, initial_state, times,
      warmup, filter_interval, filter_all_times, exact_solver, equation.grid.resample)
  return _make_dataset(
      data_vars={'y': (('time', 'x'), solution)},
      coords={'time': warmup + times, 'x': equation.grid.solution_x,
              'num_evals': num_evals})


def integrate_exact_batch(equations, times: np.ndarray = _DEFAULT_TIMES,
                          warmup: float = 0, filter_interval: float = None):
  """``integrate_exact`` (integrate.py:282-293) for many samples at once, on the
  device from start to end: what ``scripts/create_exact_data.py`` maps over
  random seeds.

  ``equations``: same type, same grid, one per sample (they differ by
  ``random_seed``, i.e. by initial condition and forcing).  Every sample is
  advanced by its own SciPy-identical RK23 controller
  (``ddd_integrate_adaptive_f64``) over the equation's exact right-hand side:
  the float64 spectral kernel for KdV / KS (``SpectralDifferentiator``), the
  float32 WENO5 + Godunov-flux kernel with per-sample forcing for Burgers
  (``WENODifferentiator``).  With ``filter_interval`` the state passes through
  the smoothing filter between segments and the saved trajectory once at the
  end, as a circulant kernel on the device (``ddd_circulant_apply_f64``).
  Returns a Dataset with y [sample, time, x] float64 and per-sample num_evals.
  """
  equations = list(equations)
  first = equations[0].to_exact()
  for eq in equations[1:]:
    if type(eq) is not type(equations[0]) or (eq.grid.solution_num_points, eq.grid.period) != (
        equations[0].grid.solution_num_points, equations[0].grid.period):
      raise ValueError('all equations must share their type and grid')
  # to_exact() keeps an equation's parameters (grid, random_seed -> the same forcing draws,
  # equations.py:184-185): the per-sample forcing tables come from the equations as given.
  # Only the initial value can depend on the exact TYPE (forcing(0) resampled the exact
  # grid's way); Burgers starts from zeros whatever the type (equations.py:256-257), so its
  # samples are not rebuilt -- each rebuild re-seeds a RandomState, ~0.1 ms per sample, which
  # was most of this function's time for the WENO solver (profiles/r6_weno_exact.txt)
  zero_start = type(first).initial_value is equations_lib.BurgersEquation.initial_value
  exact = equations if zero_start else [eq.to_exact() for eq in equations]
  method = first.EXACT_METHOD
  if method is equations_lib.ExactMethod.SPECTRAL:
    device_model = model_lib.SpectralModel(first, convention='fftpack')
  elif method is equations_lib.ExactMethod.WENO:
    device_model = model_lib.BaselineModel(first, 3, weno=True)   # WENODifferentiator's default
    if first.has_time_dependent_forcing:
      device_model.set_forcing(model_lib.forcing_from_equations(equations))
  else:
    raise ValueError('integrate_exact_batch covers the spectral and WENO exact solvers; '
                     'use integrate_exact per sample for {}'.format(type(first).__name__))
  solver = _DeviceSolver(device_model)
  if zero_start:
    y0 = _lib._torch().zeros((len(equations), first.grid.solution_num_points),
                             dtype=_lib._torch().float64, device='cuda')
  else:
    y0 = _lib.as_device(np.stack([eq.initial_value() for eq in exact]),
                        _lib._torch().float64)
  solution, num_evals = _solve(solver, first, y0, np.asarray(times, dtype=np.float64),
                               warmup, filter_interval, False, solver,
                               lambda state: state)   # exact grid == its own grid
  return _make_dataset(
      data_vars={'y': (('sample', 'time', 'x'),
                       solution.permute(1, 0, 2).contiguous().cpu().numpy())},
      coords={'time': warmup + np.asarray(times), 'x': first.grid.solution_x,
              'sample': np.arange(len(exact)),
              'num_evals': ('sample', num_evals.cpu().numpy())})


def integrate_exact(equation, times: np.ndarray = _DEFAULT_TIMES,
                    warmup: float = 0, integrate_method: str = 'RK23',
                    filter_interval: float = None):
  """Integrate only the exact model (integrate.py:282-293)."""
  equation = equation.to_exact()
  return integrate(equation, exact_differentiator(equation), times, warmup,
                   integrate_method=integrate_method,
                   filter_interval=filter_interval)


def integrate_weno(equation, times: np.ndarray = _DEFAULT_TIMES,
                   warmup: float = 0, integrate_method: str = 'RK23',
                   exact_filter_interval: float = None, **kwargs):
  """integrate.py:311-325."""
  if type(equation) not in equations_lib.FLUX_EQUATION_TYPES.values():
    raise ValueError('invalid equation: {}'.format(equation))
  return integrate(equation, WENODifferentiator(equation, **kwargs), times,
                   warmup, integrate_method=integrate_method,
                   filter_interval=exact_filter_interval)


def integrate_spectral(equation, times: np.ndarray = _DEFAULT_TIMES,
                       warmup: float = 0, integrate_method: str = 'RK23',
                       exact_filter_interval: float = None):
  """integrate.py:328-341."""
  if type(equation) not in equations_lib.EQUATION_TYPES.values():
    raise ValueError('invalid equation: {}'.format(equation))
  return integrate(equation, SpectralDifferentiator(equation), times, warmup,
                   integrate_method=integrate_method,
                   filter_interval=exact_filter_interval)


def integrate_baseline(equation, times: np.ndarray = _DEFAULT_TIMES,
                       warmup: float = 0, accuracy_order: int = 1,
                       integrate_method: str = 'RK23',
                       exact_filter_interval: float = None):
  """integrate.py:296-308."""
  differentiator = PolynomialDifferentiator(equation, accuracy_order)
  return integrate(equation, differentiator, times, warmup,
                   integrate_method=integrate_method,
                   filter_interval=exact_filter_interval)


def integrate_baseline_batch(equations, accuracy_orders: Sequence[int] = (1, 3, 5),
                             times: np.ndarray = _DEFAULT_TIMES, warmup: float = 0,
                             exact_filter_interval: float = None):
  """The computational content of ``scripts/create_baseline_data.py:96-130`` on the
  device: ``integrate_baseline`` (integrate.py:296-308) for every (sample,
  accuracy order) pair, which the script maps over seeds x ``--accuracy_orders`` one
  SciPy solve at a time and concatenates along ('sample', 'accuracy_order').

  ``equations``: the coarse (conservative, in the script) equations, same type and
  grid, one per sample -- they differ by ``random_seed`` (initial condition, forcing).
  The exact warm-up (``integrate.integrate``: exact solver on the fine grid, optional
  periodic filtering, then ``grid.resample``) runs ONCE per sample and is shared by
  all accuracy orders; every order is then one launch of the batched RK23 integrator
  (one SciPy-identical controller per sample) over its fixed-stencil model, with
  per-sample forcing for Burgers.  Returns a Dataset with
  y [sample, accuracy_order, time, x] float32 (the script's ``astype``) and
  num_evals [sample, accuracy_order].
  """
  torch = _lib._torch()
  equations = list(equations)
  first = equations[0]
  for eq in equations[1:]:
    if type(eq) is not type(first) or (
        eq.grid.solution_num_points, eq.grid.period, eq.grid.resample_factor) != (
            first.grid.solution_num_points, first.grid.period, first.grid.resample_factor):
      raise ValueError('all equations must share their type and grid')
  times = np.asarray(times, dtype=np.float64)
  if warmup:
    warm = integrate_exact_batch(
        equations, times=np.array([0.0]), warmup=warmup, filter_interval=exact_filter_interval)
    fine = _dataset_array(warm, 'y')[:, -1]                     # [sample, x_fine] at t = warmup
    y0 = np.stack([eq.grid.resample(row) for eq, row in zip(equations, fine)])
  else:
    y0 = np.stack([eq.initial_value() for eq in equations])
  y0 = _lib.as_device(y0, torch.float64)
  forcing = (model_lib.forcing_from_equations(equations)
             if first.has_time_dependent_forcing else None)
  ys, evals = [], []
  for accuracy_order in accuracy_orders:
    device_model = model_lib.BaselineModel(first, accuracy_order)
    if forcing is not None:
      device_model.set_forcing(forcing)
    y, nfev = _DeviceSolver(device_model).solve(y0, warmup + times)   # [time, sample, x]
    ys.append(y.permute(1, 0, 2))
    evals.append(nfev)
  solution = torch.stack(ys, dim=1).to(torch.float32).cpu().numpy()
  return _make_dataset(
      data_vars={'y': (('sample', 'accuracy_order', 'time', 'x'), solution)},
      coords={'time': warmup + times, 'x': first.grid.solution_x,
              'sample': np.arange(len(equations)),
              'accuracy_order': np.asarray(list(accuracy_orders)),
              'num_evals': (('sample', 'accuracy_order'),
                            torch.stack(evals, dim=1).cpu().numpy())})


def integrate_exact_baseline_and_model(checkpoint_dir: Optional[str],
                                       hparams=None, random_seed: int = 0,
                                       times: np.ndarray = _DEFAULT_TIMES,
                                       warmup: float = 0,
                                       integrate_method: str = 'RK23',
                                       exact_filter_interval: float = None,
                                       model=None):
  """Exact (fine grid), baseline and learned-model solutions of one sample
  (integrate.py:344-396): the fine equation through its exact solver, then --
  from the exact solution's first row resampled to the coarse grid -- standard
  finite differences and the neural-network stencils."""
  if hparams is None:
    hparams = (model.hparams if model is not None
               else hparams_lib.load_hparams(checkpoint_dir))
  logging.info('integrating %s with seed=%s', hparams.equation, random_seed)
  equation_fine, equation_coarse = equations_lib.from_hparams(
      hparams, random_seed=random_seed)
  logging.info('solving the "exact" model at high resolution')
  ds_exact = integrate_exact(equation_fine, times, warmup,
                             integrate_method=integrate_method,
                             filter_interval=exact_filter_interval)
  solution_exact = _dataset_array(ds_exact, 'y')
  num_evals_exact = int(np.asarray(_dataset_coord(ds_exact, 'num_evals')))
  y0 = equation_coarse.grid.resample(solution_exact[0, :])
  if np.isnan(y0).any():
    raise ValueError('solution contains NaNs')
  logging.info('solving baseline finite differences at low resolution')
  solution_baseline, num_evals_baseline = odeint(
      y0, PolynomialDifferentiator(equation_coarse), warmup + times,
      method=integrate_method)
  logging.info('solving neural network model at low resolution')
  differentiator = SavedModelDifferentiator(checkpoint_dir, equation_coarse,
                                            hparams, model=model)
  solution_model, num_evals_model = odeint(y0, differentiator, warmup + times,
                                           method=integrate_method)
  return _make_dataset(
      data_vars={'y_exact': (('time', 'x_high'), solution_exact),
                 'y_baseline': (('time', 'x_low'), solution_baseline),
                 'y_model': (('time', 'x_low'), solution_model)},
      coords={'time': warmup + times,
              'x_low': equation_coarse.grid.solution_x,
              'x_high': equation_fine.to_exact().grid.solution_x,
              'num_evals_exact': num_evals_exact,
              'num_evals_baseline': num_evals_baseline,
              'num_evals_model': num_evals_model})


def _dataset_array(ds, name):
  value = ds.data_vars[name] if not _HAVE_XARRAY else ds[name].data
  return np.asarray(value[1] if isinstance(value, tuple) else value)


def _dataset_coord(ds, name):
  value = ds.coords[name]
  if _HAVE_XARRAY:
    return value.values
  return value[1] if isinstance(value, tuple) else value


def integrate_model_from_warm_start(checkpoint_dir: Optional[str],
                                    y0: np.ndarray, hparams=None,
                                    random_seed: int = 0,
                                    times: np.ndarray = _DEFAULT_TIMES,
                                    warmup: float = 0,
                                    integrate_method: str = 'RK23',
                                    model=None):
  """integrate.py:399-427."""
  if hparams is None:
    hparams = (model.hparams if model is not None
               else hparams_lib.load_hparams(checkpoint_dir))
  _, equation_coarse = equations_lib.from_hparams(hparams,
                                                  random_seed=random_seed)
  differentiator = SavedModelDifferentiator(checkpoint_dir, equation_coarse,
                                            hparams, model=model)
  solution, num_evals = odeint(y0, differentiator, warmup + times,
                               method=integrate_method)
  return _make_dataset(
      data_vars={'y': (('time', 'x'), solution)},
      coords={'time': warmup + times, 'x': equation_coarse.grid.solution_x,
              'num_evals': num_evals})


def integrate_batch(device_model, y0, times: np.ndarray, dt: float = 0.01,
                    scheme: str = 'bs3', forcing: Optional[dict] = None,
                    launch_mode: str = 'persistent',
                    state_dtype: str = 'float32', adaptive: bool = False,
                    rtol: float = 1e-3, atol: float = 1e-6):
  """All samples at once on the GPU; sample b starts from y0[b].

  ``adaptive=True``: the batched form of ``odeint`` -- SciPy's RK23 with
  ``max_step = dt`` and one step-size controller per sample on the device
  (``ddd_integrate_adaptive_f64``): each sample's trajectory and ``num_evals``
  are what the reference's per-sample ``solve_ivp`` call produces
  (integrate.py:143-169, run_evaluation.py:152-174); ``times`` may be any
  increasing sequence, rows a diverged sample did not reach are NaN.

  ``adaptive=False``: fixed step ``dt`` (``times`` uniformly spaced multiples
  of it).  With scheme='bs3' this equals the adaptive result only while RK23's
  controller sits at max_step (smooth Burgers runs); it is the training-time
  ``model.integrate_ode`` shape and the throughput path.

  Returns a Dataset with y [sample, time, x] and num_evals (per sample when
  adaptive).
  """
  times = np.asarray(times, dtype=np.float64)
  if forcing is not None:
    device_model.set_forcing(forcing)
  if adaptive:
    traj, nfev, status = device_model.integrate_adaptive(
        y0, times, rtol=rtol, atol=atol, max_step=dt)
    y = traj.permute(1, 0, 2).contiguous().cpu().numpy()
    status = status.cpu().numpy()
    if (status != 0).any():
      logging.info('%d trajectories stopped early (NaN rows)', int((status != 0).sum()))
    return _make_dataset(
        data_vars={'y': (('sample', 'time', 'x'), y)},
        coords={'time': times, 'x': device_model.equation.grid.solution_x,
                'sample': np.arange(y.shape[0]),
                'num_evals': ('sample', nfev.cpu().numpy().astype(np.int64)),
                'status': ('sample', status)})
  spacing = np.diff(times)
  if len(times) < 2 or not np.allclose(spacing, spacing[0]):
    raise ValueError('times must be uniformly spaced')
  save_every = int(round(spacing[0] / dt))
  if save_every < 1 or abs(save_every * dt - spacing[0]) > 1e-9 * max(1, spacing[0]):
    raise ValueError('output spacing {} is not a multiple of dt {}'
                     .format(spacing[0], dt))
  num_steps = save_every * (len(times) - 1)
  traj = device_model.integrate_fixed(
      y0, num_steps, dt=dt, t0=float(times[0]), scheme=scheme,
      save_every=save_every, launch_mode=launch_mode, state_dtype=state_dtype)
  y0_dev = _lib.as_device(y0, traj.dtype)
  full = _lib._torch().cat([y0_dev[None], traj], dim=0)   # [time, sample, x]
  lib = _lib.load_library()
  stages = lib.ddd_scheme_stages(_lib.SCHEMES[scheme])
  y = full.permute(1, 0, 2).contiguous().cpu().numpy()
  if np.isnan(y).any():
    # divergence is reported as NaN rows, not an exception (integrate.py:161-167)
    logging.info('some trajectories diverged (NaN rows)')
  return _make_dataset(
      data_vars={'y': (('sample', 'time', 'x'), y)},
      coords={'time': times, 'x': device_model.equation.grid.solution_x,
              'sample': np.arange(y.shape[0]),
              'num_evals': stages * num_steps})
