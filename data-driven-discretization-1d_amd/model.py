"""Learned-stencil model: weights container + the GPU views of the network.

Host mirror of the inference half of ``pde_superresolution/model.py``.  The
reference assembles a TF graph per call; here a model object owns a handle of
the HIP library (``ddd_model``) and the functions below are thin launches of
the fused kernel:

  predict_coefficients        model.py:420-513  -> LearnedStencilModel.coefficients
  apply_coefficients          model.py:536-548  (fused; no separate launch)
  predict_space_derivatives   model.py:579-600  -> .space_derivatives
  predict_time_derivative     model.py:618-640  -> .time_derivative
  baseline_space_derivatives  model.py:59-112   -> BaselineModel.space_derivatives
  integrate_ode               model.py:138-159  -> .integrate_fixed / integrate_ode
  predict_time_evolution      model.py:643-661  -> predict_time_evolution

Weights are plain arrays (``conv_kernels[l]`` in the tf.layers.conv1d variable
layout [K, Cin, Cout], ``conv_biases[l]`` [Cout]); the polynomial-accuracy
null-space bases and biases are stored with them because the SVD basis is not
unique across LAPACK builds (polynomials.py:246-254).
"""
import ctypes
import contextlib
import json
import os
from typing import List, Optional, Sequence

import numpy as np

from . import _lib
from . import duckarray
from . import equations as equations_lib
from . import hparams as hparams_lib
from . import polynomials

FINITE_DIFF = polynomials.Method.FINITE_DIFFERENCES
FINITE_VOL = polynomials.Method.FINITE_VOLUMES

WEIGHTS_FILENAME = 'model.npz'


# ---------------------------------------------------------------------------
# forcing tables
# ---------------------------------------------------------------------------
def batched_forcing_parameters(seeds: Sequence[int], nparams: int = 20,
                               amplitude: float = 1, k_min: int = 1,
                               k_max: int = 3) -> dict:
  """RandomForcing draws (equations.py:207-212) for many seeds at once.

  Returns float64 arrays a, omega, phi and int array k, each
  [len(seeds), nparams]; row i equals RandomForcing(seed=seeds[i]).
  """
  count = len(seeds)
  out = {name: np.empty((count, nparams)) for name in ('a', 'omega', 'phi')}
  out['k'] = np.empty((count, nparams), dtype=np.int64)
  wavenumbers = np.arange(k_min, k_max + 1)
  choices = np.concatenate([-wavenumbers, wavenumbers])
  for i, seed in enumerate(seeds):
    rs = np.random.RandomState(int(seed))
    out['a'][i] = 0.5 * amplitude * rs.uniform(-1, 1, size=nparams)
    out['omega'][i] = rs.uniform(-0.4, 0.4, size=nparams)
    out['k'][i] = rs.choice(choices, size=nparams)
    out['phi'][i] = rs.uniform(0, 2 * np.pi, size=nparams)
  return out


def forcing_from_equations(eqs: Sequence[equations_lib.Equation]) -> dict:
  """Stack the RandomForcing parameters of per-sample Equation objects."""
  return dict(
      a=np.stack([e.forcing.a[:, 0] for e in eqs]),
      omega=np.stack([e.forcing.omega[:, 0] for e in eqs]),
      k=np.stack([e.forcing.k[:, 0] for e in eqs]).astype(np.int64),
      phi=np.stack([e.forcing.phi[:, 0] for e in eqs]))


def forcing_kernel_tables(forcing: dict, grid: equations_lib.Grid) -> dict:
  """Turn (a, omega, k, phi) into the tables ddd_set_forcing takes.

  The reference evaluates the waves on the *reference* grid and resamples to
  the solution grid (equations.py:214-219).  Sub-sampling keeps reference
  points i*rf, so the solution-grid phase is 2 pi k x_i / L.  Mean-resampling
  averages rf consecutive reference points; for a sine that is exactly
      mean_r sin(theta + r delta) = D sin(theta + (rf-1) delta / 2),
      delta = 2 pi k / (N rf),  D = sin(rf delta/2) / (rf sin(delta/2)),
  so the block mean folds into the amplitude (x D) and phase (+ shift), and the
  kernel evaluates one sine per mode whatever the resample factor.
  """
  a = np.asarray(forcing['a'], dtype=np.float64)
  omega = np.asarray(forcing['omega'], dtype=np.float64)
  k = np.asarray(forcing['k']).astype(np.int64)
  phi = np.asarray(forcing['phi'], dtype=np.float64)
  rf = grid.resample_factor
  n = grid.solution_num_points
  if grid.resample_method == 'mean' and rf > 1:
    delta = 2 * np.pi * k / (n * rf)
    # k = 0 (equation_kwargs k_min=0): a constant mode, block mean = itself
    safe = np.where(k == 0, 1.0, delta)
    dirichlet = np.where(k == 0, 1.0, np.sin(rf * safe / 2) / (rf * np.sin(safe / 2)))
    amplitude = a * dirichlet
    phase = phi + (rf - 1) * delta / 2
  else:
    amplitude, phase = a, phi
  k_values = np.unique(k)
  k_index = np.searchsorted(k_values, k).astype(np.int32)
  # same expression/rounding as RandomForcing: 2 pi k x / period, then float32
  spatial = (2 * np.pi * k_values[:, None] * grid.solution_x[None, :]
             / grid.period).astype(np.float32)
  return dict(amplitude=amplitude.astype(np.float32),
              omega=omega.astype(np.float32),
              phase=phase.astype(np.float32),
              k_index=k_index, spatial_phase=spatial)


# ---------------------------------------------------------------------------
# model handles
# ---------------------------------------------------------------------------
class _DeviceModel(object):
  """Owns a ``ddd_model*`` and exposes the batched kernel entry points."""

  def __init__(self):
    self._handle_ = None
    self.equation = None
    self._forcing = None   # dict of float64 arrays (for spec())
    self._kernel_kind = 'auto'

  # The device handle is created on first use, so weights can be built, saved
  # and loaded on a host without a GPU; every kernel entry point needs one.
  @property
  def _handle(self):
    if self._handle_ is None:
      self._create_handle()
      if self._kernel_kind != 'auto':
        _lib.check(_lib.load_library().ddd_set_kernel(
            self._handle_, _lib.KERNELS[self._kernel_kind]))
      if self._forcing is not None:
        self._upload_forcing(self._forcing)
    return self._handle_

  @_handle.setter
  def _handle(self, value):
    self._handle_ = value

  # -- lifecycle -------------------------------------------------------------
  def close(self):
    if self._handle_ is not None and _lib._lib is not None:
      _lib._lib.ddd_model_destroy(self._handle_)
    self._handle_ = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  def _base_config(self, equation, stencil_size: int) -> _lib.DDDConfig:
    spec = equation.kernel_spec()
    cfg = _lib.DDDConfig()
    cfg.struct_size = ctypes.sizeof(_lib.DDDConfig)
    cfg.equation = spec['equation']
    cfg.num_points = spec['num_points']
    cfg.num_derivatives = len(spec['derivative_orders'])
    for i, order in enumerate(spec['derivative_orders']):
      cfg.derivative_orders[i] = order
    cfg.dx = spec['dx']
    cfg.period = spec['period']
    cfg.eta = spec['eta']
    cfg.standard_deviation = spec['standard_deviation']
    cfg.stencil_size = stencil_size
    return cfg

  # -- introspection ---------------------------------------------------------
  @property
  def kernel_name(self) -> str:
    return _lib.load_library().ddd_kernel_name(self._handle).decode()

  def set_kernel(self, kind: str):
    _lib.check(_lib.load_library().ddd_set_kernel(self._handle,
                                                  _lib.KERNELS[kind]))
    self._kernel_kind = kind

  @property
  def fma_per_point(self) -> int:
    return int(_lib.load_library().ddd_fma_per_point(self._handle))

  @property
  def num_points(self) -> int:
    return self.equation.grid.solution_num_points

  # -- forcing ---------------------------------------------------------------
  def set_forcing(self, forcing: Optional[dict]):
    """Per-sample forcing parameters: dict a/omega/k/phi, each [batch, P].

    Only Burgers-family equations apply forcing(t) in
    finalize_time_derivative (equations.py:276-277); for the others the tables
    are accepted and ignored, as in the reference.
    """
    if forcing is None:
      self._forcing = None
      if self._handle_ is not None:
        _lib.check(_lib.load_library().ddd_clear_forcing(self._handle_))
      return
    forcing = {k: np.asarray(v) for k, v in forcing.items()}
    shapes = {np.shape(forcing[k]) for k in ('a', 'omega', 'k', 'phi')}
    if len(shapes) != 1 or len(next(iter(shapes))) != 2:
      raise ValueError('forcing arrays must share one [batch, nparams] shape')
    self._forcing = forcing
    if self._handle_ is not None:
      self._upload_forcing(forcing)

  def _upload_forcing(self, forcing):
    lib = _lib.load_library()
    tables = forcing_kernel_tables(forcing, self.equation.grid)
    batch, nparams = tables['amplitude'].shape
    amp = np.ascontiguousarray(tables['amplitude'])
    omega = np.ascontiguousarray(tables['omega'])
    phase = np.ascontiguousarray(tables['phase'])
    kidx = np.ascontiguousarray(tables['k_index'])
    spatial = np.ascontiguousarray(tables['spatial_phase'])
    _lib.check(lib.ddd_set_forcing(
        self._handle_, batch, nparams, _lib.fptr(amp), _lib.fptr(omega),
        _lib.fptr(phase), kidx.ctypes.data_as(_lib._I), _lib.fptr(spatial),
        spatial.shape[0]))

  def set_forcing_from_equation(self, batch: int = 1):
    """Use this model's own equation.forcing for every sample (B copies)."""
    f = self.equation.forcing
    self.set_forcing(dict(a=np.repeat(f.a.T, batch, 0),
                          omega=np.repeat(f.omega.T, batch, 0),
                          k=np.repeat(f.k.T, batch, 0),
                          phi=np.repeat(f.phi.T, batch, 0)))

  # -- kernels ---------------------------------------------------------------
  def _check_state(self, y, dtype):
    torch = _lib.require_gpu()
    y = _lib.as_device(y, dtype)
    if y.dim() != 2 or y.shape[1] != self.num_points:
      raise ValueError('solution has unexpected size for equation: {} vs {}'
                       .format(tuple(y.shape), self.num_points))
    return torch, y

  def time_derivative(self, y, t: float = 0.0):
    """finalize_time_derivative(t, predict_time_derivative(y)); y [batch, x]."""
    lib = _lib.load_library()
    torch, y = self._check_state(y, _lib._torch().float32)
    out = torch.empty_like(y)
    _lib.check(lib.ddd_time_derivative(self._handle, float(t), y.data_ptr(),
                                       out.data_ptr(), y.shape[0],
                                       _lib.current_stream()))
    return out

  def time_derivative_rows(self, y, out, t: float = 0.0):
    """time_derivative on caller-owned float32 rows [batch, x] the device can
    address (device tensors or page-locked host tensors); enqueued on the
    current stream, nothing is copied, allocated or synchronised."""
    lib = _lib.load_library()
    _lib.check(lib.ddd_time_derivative(self._handle, float(t), y.data_ptr(),
                                       out.data_ptr(), y.shape[0],
                                       _lib.current_stream()))

  def space_derivatives(self, y):
    """[batch, x] -> [batch, x, derivative]."""
    lib = _lib.load_library()
    torch, y = self._check_state(y, _lib._torch().float32)
    d = len(self.equation.DERIVATIVE_ORDERS)
    out = torch.empty(y.shape + (d,), dtype=torch.float32, device=y.device)
    _lib.check(lib.ddd_space_derivatives(self._handle, y.data_ptr(),
                                         out.data_ptr(), y.shape[0],
                                         _lib.current_stream()))
    return out

  def coefficients(self, y):
    """[batch, x] -> [batch, x, derivative, stencil]."""
    lib = _lib.load_library()
    torch, y = self._check_state(y, _lib._torch().float32)
    d = len(self.equation.DERIVATIVE_ORDERS)
    out = torch.empty(y.shape + (d, self.stencil_size), dtype=torch.float32,
                      device=y.device)
    _lib.check(lib.ddd_coefficients(self._handle, y.data_ptr(), out.data_ptr(),
                                    y.shape[0], _lib.current_stream()))
    return out

  def rk_substep(self, t, y_in, y_base=None, c1=1.0, y_out=None, acc_in=None,
                 c2=0.0, acc_out=None):
    """One fused launch: f = rhs(t, y_in); y_out = y_base + c1 f; acc_out = acc_in + c2 f."""
    lib = _lib.load_library()
    torch, y_in = self._check_state(y_in, _lib._torch().float32)
    ptr = lambda x: None if x is None else x.data_ptr()
    _lib.check(lib.ddd_rk_substep(
        self._handle, float(t), y_in.data_ptr(), ptr(y_base), float(c1),
        ptr(y_out), ptr(acc_in), float(c2), ptr(acc_out), y_in.shape[0],
        _lib.current_stream()))

  @contextlib.contextmanager
  def chained_substeps(self):
    """`with model.chained_substeps(): ...` around a caller-owned Runge-Kutta loop
    of rk_substep / time_derivative_rows calls on the current stream
    (ddd_stream_fork .. ddd_stream_join, include/ddd1d.h): large ensembles then
    advance as two half-ensemble chains that stay alive across the calls -- or,
    after `set_region_mode('ring')`, for the per-equation kernels on one-wave groups,
    as commands to ONE persistent kernel (the command ring, round 6).  Inside
    the block the arrays handed to those calls must not be touched by anything
    else; after it the current stream is ordered behind every substep."""
    lib = _lib.load_library()
    stream = _lib.current_stream()
    _lib.check(lib.ddd_stream_fork(self._handle, stream))
    try:
      yield self
    finally:
      _lib.check(lib.ddd_stream_join(self._handle, stream))

  def set_region_mode(self, mode: str = 'auto'):
    """How `chained_substeps` regions run: 'auto' / 'chains' (launches) or 'ring' (the
    command ring where the model has one).  ddd_set_region_mode."""
    lib = _lib.load_library()
    _lib.check(lib.ddd_set_region_mode(
        self._handle, {'auto': 0, 'chains': 1, 'ring': 2}[mode]))

  def region_stats(self):
    """(persistent-kernel launches, commands) of the command ring so far."""
    import ctypes
    lib = _lib.load_library()
    launches, commands = ctypes.c_int64(0), ctypes.c_int64(0)
    _lib.check(lib.ddd_region_stats(self._handle, ctypes.byref(launches),
                                    ctypes.byref(commands)))
    return launches.value, commands.value

  def integrate_fixed(self, y0, num_steps: int, dt: Optional[float] = None,
                      t0: float = 0.0, scheme: str = 'midpoint',
                      save_every: int = 1, launch_mode: str = 'persistent',
                      state_dtype: str = 'float32', out=None):
    """Fixed-step explicit RK over the whole batch.

    Returns [num_steps // save_every, batch, x] (state after steps save_every,
    2*save_every, ...).  ``dt`` defaults to ``equation.time_step``.
    """
    lib = _lib.load_library()
    torch = _lib.require_gpu()
    dt = self.equation.time_step if dt is None else dt
    dtype = {'float32': torch.float32, 'float64': torch.float64}[state_dtype]
    _, y0 = self._check_state(y0, dtype)
    num_saved = num_steps // save_every
    shape = (num_saved,) + tuple(y0.shape)
    if out is None:
      out = torch.empty(shape, dtype=dtype, device=y0.device)
    elif tuple(out.shape) != shape or out.dtype != dtype or not out.is_contiguous():
      raise ValueError('out must be a contiguous {} tensor of shape {}'
                       .format(dtype, shape))
    if state_dtype == 'float64':
      if launch_mode != 'persistent':
        raise ValueError('float64 state requires launch_mode="persistent"')
      _lib.check(lib.ddd_integrate_fixed_f64(
          self._handle, _lib.SCHEMES[scheme], float(t0), float(dt),
          int(num_steps), int(save_every), y0.data_ptr(), out.data_ptr(),
          y0.shape[0], _lib.current_stream()))
    else:
      _lib.check(lib.ddd_integrate_fixed(
          self._handle, _lib.SCHEMES[scheme], _lib.LAUNCH_MODES[launch_mode],
          float(t0), float(dt), int(num_steps), int(save_every),
          y0.data_ptr(), out.data_ptr(), y0.shape[0], _lib.current_stream()))
    return out


  def integrate_adaptive(self, y0, times, rtol: float = 1e-3, atol: float = 1e-6,
                         max_step: float = 0.01, max_attempts: int = 0):
    """SciPy RK23 with one step-size controller per sample, on the device.

    The batched form of ``integrate.odeint`` (integrate.py:143-169): every
    sample b is advanced from y0[b] exactly as
    ``solve_ivp(rhs_b, (times[0], times[-1]), y0[b], t_eval=times,
    max_step=max_step, method='RK23')`` would, in one launch.  Returns
    ``(y [time, batch, x] float64, nfev [batch] int32, status [batch] int32)``
    as device tensors; status 0 = finished, -1 = step size too small (the
    sample's remaining rows are NaN), -2 = attempt limit.
    """
    lib = _lib.load_library()
    torch, y0 = self._check_state(y0, _lib._torch().float64)
    times = np.ascontiguousarray(np.asarray(times, dtype=np.float64))
    if times.ndim != 1 or times.size < 1:
      raise ValueError('times must be a non-empty 1-D array')
    batch = y0.shape[0]
    out = torch.empty((times.size,) + tuple(y0.shape), dtype=torch.float64,
                      device=y0.device)
    nfev = torch.zeros(batch, dtype=torch.int32, device=y0.device)
    status = torch.zeros(batch, dtype=torch.int32, device=y0.device)
    _lib.check(lib.ddd_integrate_adaptive_f64(
        self._handle, times.ctypes.data_as(_lib._D), int(times.size),
        float(rtol), float(atol), float(max_step), int(max_attempts),
        y0.data_ptr(), out.data_ptr(), nfev.data_ptr(), status.data_ptr(),
        batch, _lib.current_stream()))
    return out, nfev, status


class LearnedStencilModel(_DeviceModel):
  """Conv-net coefficient predictor for one equation + hparams.

  Args:
    equation: coarse-grid Equation the model integrates.
    hparams: hyper-parameters (create_hparams); only the network keys are read.
    conv_kernels / conv_biases: per-layer weights; ``None`` draws synthetic
      Glorot-uniform weights (seed ``init_seed``), output layer scaled by
      ``output_scale`` so stencils stay near the polynomial bias.
    nullspaces / biases: stored polynomial-accuracy tables; ``None`` computes
      them here with this machine's LAPACK.
  """

  def __init__(self, equation, hparams, conv_kernels=None, conv_biases=None,
               nullspaces=None, biases=None, constant_coefficients=None,
               init_seed: int = 0, output_scale: float = 0.1):
    super(LearnedStencilModel, self).__init__()
    self.equation = equation
    self.hparams = hparams
    num_derivatives = len(equation.DERIVATIVE_ORDERS)
    if hparams.model_target not in _lib.MODEL_TARGETS:
      raise NotImplementedError(
          'unrecognized model_target: {}'.format(hparams.model_target))
    if hparams.nonlinearity not in _lib.ACTIVATIONS:
      raise KeyError(hparams.nonlinearity)

    # model.py:443-446: the coefficient grid
    self.grid = polynomials.regular_grid(
        equation.GRID_OFFSET, derivative_order=0,
        accuracy_order=hparams.coefficient_grid_min_size,
        dx=equation.grid.solution_dx)
    self.stencil_size = self.grid.size

    projected = (hparams.model_target == 'coefficients'
                 and bool(hparams.polynomial_accuracy_order))
    self.input_sizes = []
    if projected:
      if nullspaces is None or biases is None:
        method = FINITE_VOL if equation.CONSERVATIVE else FINITE_DIFF
        layers = [polynomials.PolynomialAccuracyLayer(
            self.grid, method, order,
            accuracy_order=hparams.polynomial_accuracy_order,
            out_scale=hparams.polynomial_accuracy_scale)
                  for order in equation.DERIVATIVE_ORDERS]
        nullspaces = [layer.nullspace for layer in layers]
        biases = [layer.bias for layer in layers]
      self.input_sizes = [int(np.shape(ns)[0]) for ns in nullspaces]
    self.nullspaces = None if nullspaces is None else [
        np.asarray(ns, dtype=np.float64) for ns in nullspaces]
    self.biases = None if biases is None else [
        np.asarray(b, dtype=np.float64) for b in biases]

    if hparams.model_target == 'coefficients':
      if projected:
        c_out = sum(self.input_sizes)
      else:
        if hparams.num_layers == 0:
          raise NotImplementedError
        if (hparams.ensure_unbiased_coefficients
            and 0 in equation.DERIVATIVE_ORDERS):
          raise ValueError('ensure_unbiased not yet supported for 0th order '
                           'spatial derivatives')
        c_out = num_derivatives * self.stencil_size
    elif hparams.model_target == 'space_derivatives':
      c_out = num_derivatives
    else:
      c_out = 1
    if hparams.num_layers == 0 and not projected:
      raise NotImplementedError('not implemented yet')
    self.num_outputs = c_out

    layer_shapes = []
    for l in range(hparams.num_layers):
      cin = 1 if l == 0 else hparams.filter_size
      cout = c_out if l == hparams.num_layers - 1 else hparams.filter_size
      layer_shapes.append((hparams.kernel_size, cin, cout))
    if conv_kernels is None:
      conv_kernels, conv_biases = self.synthetic_weights(
          layer_shapes, init_seed, output_scale)
    self.conv_kernels = [np.asarray(w, dtype=np.float32) for w in conv_kernels]
    self.conv_biases = [np.asarray(b, dtype=np.float32) for b in conv_biases]
    if [w.shape for w in self.conv_kernels] != layer_shapes:
      raise ValueError('conv kernel shapes {} do not match hparams {}'.format(
          [w.shape for w in self.conv_kernels], layer_shapes))
    self.constant_coefficients = None
    if hparams.num_layers == 0:
      self.constant_coefficients = (
          np.zeros(c_out, np.float32) if constant_coefficients is None
          else np.asarray(constant_coefficients, dtype=np.float32))

  @staticmethod
  def synthetic_weights(layer_shapes, seed=0, output_scale=0.1):
    """Glorot-uniform kernels (tf.layers.conv1d default), zero biases."""
    rs = np.random.RandomState(seed)
    kernels, biases = [], []
    for i, (k, cin, cout) in enumerate(layer_shapes):
      limit = np.sqrt(6.0 / (k * cin + k * cout))
      w = rs.uniform(-limit, limit, size=(k, cin, cout))
      if i == len(layer_shapes) - 1:
        w = w * output_scale
      kernels.append(w.astype(np.float32))
      biases.append(np.zeros(cout, np.float32))
    return kernels, biases

  def _create_handle(self):
    lib = _lib.load_library()
    _lib.require_gpu()
    hp = self.hparams
    handle = ctypes.c_void_p()
    if hp.num_layers == 0:
      # model.py:496-502: coefficients = bias + const @ nullspace, the same for
      # every grid point -> fold into fixed stencils (float32, reference order)
      stencils = []
      start = 0
      for ns, b in zip(self.nullspaces, self.biases):
        stop = start + ns.shape[0]
        stencils.append(b.astype(np.float32) + np.einsum(
            'i,ij->j', self.constant_coefficients[start:stop],
            ns.astype(np.float32)).astype(np.float32))
        start = stop
      table = _lib.host_f32(np.stack(stencils))
      cfg = self._base_config(self.equation, self.stencil_size)
      _lib.check(lib.ddd_baseline_create(ctypes.byref(cfg), _lib.fptr(table),
                                         table.size, ctypes.byref(handle)))
      self._handle = handle
      return
    cfg = self._base_config(self.equation, self.stencil_size)
    cfg.model_target = _lib.MODEL_TARGETS[hp.model_target]
    cfg.num_layers = hp.num_layers
    cfg.filter_size = hp.filter_size
    cfg.kernel_size = hp.kernel_size
    cfg.activation = _lib.ACTIVATIONS[hp.nonlinearity]
    cfg.polynomial_accuracy_order = int(hp.polynomial_accuracy_order or 0)
    cfg.ensure_unbiased_coefficients = int(bool(hp.ensure_unbiased_coefficients))
    for i, size in enumerate(self.input_sizes):
      cfg.input_sizes[i] = size
    flat = np.concatenate(
        [np.concatenate([w.ravel(), b.ravel()])
         for w, b in zip(self.conv_kernels, self.conv_biases)]).astype(np.float32)
    flat = np.ascontiguousarray(flat)
    if self.input_sizes:
      ns = _lib.host_f32(np.concatenate([n.ravel() for n in self.nullspaces]))
      bs = _lib.host_f32(np.concatenate([b.ravel() for b in self.biases]))
      ns_ptr, bs_ptr, n_ns, n_bs = _lib.fptr(ns), _lib.fptr(bs), ns.size, bs.size
    else:
      ns_ptr = bs_ptr = None
      n_ns = n_bs = 0
    _lib.check(lib.ddd_model_create(ctypes.byref(cfg), _lib.fptr(flat),
                                    flat.size, ns_ptr, n_ns, bs_ptr, n_bs,
                                    ctypes.byref(handle)))
    self._handle = handle

  # -- (de)serialisation -------------------------------------------------------
  def save(self, checkpoint_dir: str) -> str:
    """Write hparams.json + model.npz (the analogue of hparams.pbtxt + ckpt)."""
    hparams_lib.save_hparams(self.hparams, checkpoint_dir)
    arrays = {}
    for i, (w, b) in enumerate(zip(self.conv_kernels, self.conv_biases)):
      arrays['conv{}_kernel'.format(i)] = w
      arrays['conv{}_bias'.format(i)] = b
    for i, (ns, b) in enumerate(zip(self.nullspaces or [], self.biases or [])):
      arrays['nullspace{}'.format(i)] = ns
      arrays['bias{}'.format(i)] = b
    if self.constant_coefficients is not None:
      arrays['constant_coefficients'] = self.constant_coefficients
    path = os.path.join(checkpoint_dir, WEIGHTS_FILENAME)
    np.savez(path, **arrays)
    return path

  @classmethod
  def load(cls, checkpoint_dir: str, equation=None, hparams=None,
           random_seed: int = 0):
    if hparams is None:
      hparams = hparams_lib.load_hparams(checkpoint_dir)
    if equation is None:
      _, equation = equations_lib.from_hparams(hparams, random_seed=random_seed)
    from . import checkpoint
    if (not os.path.exists(os.path.join(checkpoint_dir, WEIGHTS_FILENAME)) and
        checkpoint.has_tf_checkpoint(checkpoint_dir)):
      # the reference's own artefact: hparams.pbtxt + model.ckpt.  Only the conv
      # tower is stored; the null-space tables are rebuilt from the hparams
      # exactly as model.predict_coefficients does at restore time
      # (model.py:480-489) -- and, as there, depend on the local LAPACK's SVD.
      if hparams.num_layers == 0:
        # the learned constant vector (model.py:496-499); KeyError if absent
        const = checkpoint.load_constant_coefficients(checkpoint_dir)
        return cls(equation, hparams, [], [], constant_coefficients=const)
      kernels, conv_biases = checkpoint.load_conv_weights(
          checkpoint_dir, hparams.num_layers, hparams.model_target)
      return cls(equation, hparams, kernels, conv_biases)
    with np.load(os.path.join(checkpoint_dir, WEIGHTS_FILENAME)) as data:
      kernels = [data['conv{}_kernel'.format(i)]
                 for i in range(hparams.num_layers)]
      conv_biases = [data['conv{}_bias'.format(i)]
                     for i in range(hparams.num_layers)]
      count = len([k for k in data.files if k.startswith('nullspace')])
      nullspaces = [data['nullspace{}'.format(i)] for i in range(count)] or None
      biases = [data['bias{}'.format(i)] for i in range(count)] or None
      const = (data['constant_coefficients']
               if 'constant_coefficients' in data.files else None)
    return cls(equation, hparams, kernels, conv_biases, nullspaces, biases,
               constant_coefficients=const)

  # -- the description the CPU oracle consumes (tests / bench baseline) --------
  def spec(self) -> dict:
    spec = dict(self.equation.kernel_spec())
    hp = self.hparams
    spec.update(
        resample_factor=self.equation.grid.resample_factor,
        stencil_size=self.stencil_size,
        model_target=hp.model_target,
        num_layers=hp.num_layers,
        nonlinearity=hp.nonlinearity,
        polynomial_accuracy_order=int(hp.polynomial_accuracy_order or 0),
        ensure_unbiased_coefficients=bool(hp.ensure_unbiased_coefficients),
        conv_kernels=self.conv_kernels,
        conv_biases=self.conv_biases,
        nullspaces=self.nullspaces,
        biases=self.biases,
        constant_coefficients=self.constant_coefficients,
        baseline_coefficients=None)
    return spec


class BaselineModel(_DeviceModel):
  """Standard polynomial stencils (model.baseline_space_derivatives).

  Reference: model.py:59-112.  With an explicit ``accuracy_order``: per
  derivative ``regular_grid(GRID_OFFSET, d, accuracy_order, dx)`` and
  ``polynomials.coefficients`` (FV for conservative equations, FD otherwise).
  ``weno=True`` additionally replaces ``u_minus`` / ``u_plus`` of a
  Godunov-flux equation by WENO5 reconstructions on the GPU, which is
  ``integrate.WENODifferentiator`` (integrate.py:124-140).

  ``accuracy_order=None`` selects the reference's best baseline
  (model.py:69-95) for equations that are their own exact type: the 6-point
  stencil for ExactMethod.POLYNOMIAL, WENO5 + third-order ``u_x`` for
  ExactMethod.WENO (Godunov Burgers).  ExactMethod.SPECTRAL is
  ``SpectralModel``.
  """

  def __init__(self, equation, accuracy_order: Optional[int] = 1,
               weno: bool = False):
    super(BaselineModel, self).__init__()
    method = FINITE_VOL if equation.CONSERVATIVE else FINITE_DIFF
    is_flux = type(equation) in equations_lib.FLUX_EQUATION_TYPES.values()
    self.equation = equation
    self.accuracy_order = accuracy_order
    self.weno = bool(weno)
    self.stencils = []
    if accuracy_order is None:
      if equation.exact_type() is not type(equation):
        raise AssertionError('the best baseline needs an exact equation type '
                             '(model.py:70)')
      exact = equation.EXACT_METHOD
      if exact is equations_lib.ExactMethod.POLYNOMIAL:
        for order in equation.DERIVATIVE_ORDERS:
          grid = (0.5 + np.arange(-3, 3)) * equation.grid.solution_dx
          self.stencils.append(polynomials.coefficients(grid, method, order))
      elif exact is equations_lib.ExactMethod.WENO:
        self.weno = True
        for name, order in zip(equation.DERIVATIVE_NAMES,
                               equation.DERIVATIVE_ORDERS):
          if name in ('u_minus', 'u_plus'):
            # placeholder rows: the kernel overwrites these two derivatives
            self.stencils.append(np.zeros(2))
            continue
          if name != 'u_x':
            raise AssertionError('best WENO baseline only knows u_x '
                                 '(model.py:89)')
          grid = polynomials.regular_grid(equation.GRID_OFFSET, order, 3,
                                          equation.grid.solution_dx)
          self.stencils.append(polynomials.coefficients(grid, method, order))
      else:
        raise ValueError('spectral best baseline: use SpectralModel')
    else:
      for order in equation.DERIVATIVE_ORDERS:
        grid = polynomials.regular_grid(equation.GRID_OFFSET, order,
                                        accuracy_order,
                                        equation.grid.solution_dx)
        self.stencils.append(polynomials.coefficients(grid, method, order))
    if self.weno and not is_flux:
      raise ValueError('WENO reconstruction needs a Godunov-flux equation')
    width = max(len(s) for s in self.stencils)
    # Centre every stencil in a common window so that tap i multiplies
    # u[x + i - width // 2] -- the alignment pad_periodic(center=True) gives
    # each individual filter (layers.py:76-79).
    table = np.zeros((len(self.stencils), width), np.float32)
    for d, taps in enumerate(self.stencils):
      shift = width // 2 - len(taps) // 2
      table[d, shift:shift + len(taps)] = np.asarray(taps).astype(np.float32)
    self.stencil_size = width
    self.table = table

  def _create_handle(self):
    lib = _lib.load_library()
    _lib.require_gpu()
    cfg = self._base_config(self.equation, self.stencil_size)
    cfg.weno_reconstruction = int(self.weno)
    handle = ctypes.c_void_p()
    flat = _lib.host_f32(self.table)
    _lib.check(lib.ddd_baseline_create(ctypes.byref(cfg), _lib.fptr(flat),
                                       flat.size, ctypes.byref(handle)))
    self._handle = handle

  def spec(self) -> dict:
    spec = dict(self.equation.kernel_spec())
    spec.update(resample_factor=self.equation.grid.resample_factor,
                stencil_size=self.stencil_size,
                baseline_coefficients=self.stencils, weno=self.weno)
    return spec


class SpectralModel(_DeviceModel):
  """Spectral space derivatives + equation of motion in float64 on the GPU.

  The fine-grid "exact" method of KdV and KS (ExactMethod.SPECTRAL):
  ``integrate.SpectralDifferentiator`` (integrate.py:108-121, derivatives from
  ``scipy.fftpack.diff``) and the spectral branch of
  ``model.baseline_space_derivatives`` (model.py:78-80,
  ``duckarray.spectral_derivative``).  Both are circulant linear operators;
  their first columns are obtained by applying the reference's own call to a
  unit impulse (``convention`` picks which), and the kernel evaluates
  ``deriv[x] = sum_j c[(x - j) mod N] y[j]`` in float64.

  No forcing on the device: ``finalize_time_derivative`` is applied by the
  host-side differentiator in float64, as in the reference.
  """

  def __init__(self, equation, convention: str = 'fftpack'):
    super(SpectralModel, self).__init__()
    if type(equation) not in equations_lib.EQUATION_TYPES.values():
      raise ValueError('invalid equation: {}'.format(equation))
    if convention not in ('fftpack', 'rfft'):
      raise ValueError('convention must be "fftpack" or "rfft"')
    self.equation = equation
    self.convention = convention
    n = equation.grid.solution_num_points
    period = equation.grid.period
    impulse = np.zeros(n)
    impulse[0] = 1.0
    rows = []
    for order in equation.DERIVATIVE_ORDERS:
      if convention == 'fftpack':
        import scipy.fftpack
        rows.append(scipy.fftpack.diff(impulse, order, period))
      else:
        rows.append(duckarray.spectral_derivative(impulse, order, period))
    self.kernels = np.ascontiguousarray(np.stack(rows), dtype=np.float64)
    self.stencil_size = n

  def _create_handle(self):
    lib = _lib.load_library()
    _lib.require_gpu()
    cfg = self._base_config(self.equation, 1)
    handle = ctypes.c_void_p()
    _lib.check(lib.ddd_spectral_create(
        ctypes.byref(cfg), self.kernels.ctypes.data_as(_lib._D),
        self.kernels.size, ctypes.byref(handle)))
    self._handle = handle

  def set_forcing(self, forcing):
    raise ValueError('spectral models carry no device forcing; '
                     'finalize_time_derivative runs on the host')

  def time_derivative(self, y, t: float = 0.0):
    """equation_of_motion(y, spectral derivatives); y [batch, x] float64."""
    lib = _lib.load_library()
    torch, y = self._check_state(y, _lib._torch().float64)
    out = torch.empty_like(y)
    _lib.check(lib.ddd_time_derivative_f64(self._handle, float(t), y.data_ptr(),
                                           out.data_ptr(), y.shape[0],
                                           _lib.current_stream()))
    return out

  def space_derivatives(self, y):
    raise NotImplementedError('spectral models expose time derivatives only')

  coefficients = space_derivatives

  def rk_substep(self, t, y_in, y_base=None, c1=1.0, y_out=None, acc_in=None,
                 c2=0.0, acc_out=None):
    lib = _lib.load_library()
    torch, y_in = self._check_state(y_in, _lib._torch().float64)
    ptr = lambda x: None if x is None else x.data_ptr()
    _lib.check(lib.ddd_rk_substep_f64(
        self._handle, float(t), y_in.data_ptr(), ptr(y_base), float(c1),
        ptr(y_out), ptr(acc_in), float(c2), ptr(acc_out), y_in.shape[0],
        _lib.current_stream()))

  def integrate_fixed(self, y0, num_steps: int, dt: Optional[float] = None,
                      t0: float = 0.0, scheme: str = 'midpoint',
                      save_every: int = 1, launch_mode: str = 'per_substep',
                      state_dtype: str = 'float64', out=None):
    """Fixed-step explicit RK, state and right-hand side in float64."""
    if self.equation.has_time_dependent_forcing:
      raise ValueError('forced equations need the host finalize step: '
                       'use integrate.SpectralDifferentiator with odeint')
    if state_dtype != 'float64' or launch_mode != 'per_substep':
      raise ValueError('spectral models step in float64, one launch per substep')
    lib = _lib.load_library()
    torch = _lib.require_gpu()
    dt = self.equation.time_step if dt is None else dt
    _, y0 = self._check_state(y0, torch.float64)
    shape = (num_steps // save_every,) + tuple(y0.shape)
    if out is None:
      out = torch.empty(shape, dtype=torch.float64, device=y0.device)
    elif tuple(out.shape) != shape or out.dtype != torch.float64 or not out.is_contiguous():
      raise ValueError('out must be a contiguous float64 tensor of shape {}'.format(shape))
    _lib.check(lib.ddd_integrate_fixed_f64(
        self._handle, _lib.SCHEMES[scheme], float(t0), float(dt), int(num_steps),
        int(save_every), y0.data_ptr(), out.data_ptr(), y0.shape[0],
        _lib.current_stream()))
    return out

  def spec(self) -> dict:
    spec = dict(self.equation.kernel_spec())
    spec.update(spectral=True, convention=self.convention)
    return spec


# ---------------------------------------------------------------------------
# functional API with the reference's names
# ---------------------------------------------------------------------------
def assert_consistent_solution(equation, solution):
  """model.py:42-56: the last axis must match the equation's grid."""
  if equation.grid.solution_num_points != np.shape(solution)[-1]:
    raise ValueError('solution has unexpected size for equation: {} vs {}'.format(
        np.shape(solution)[-1], equation.grid.solution_num_points))


def extract_patches(inputs, size: int):
  """model.py:516-533: [batch, x] -> [batch, x, size] periodic patches."""
  return _lib.extract_patches(inputs, size)


def apply_coefficients(coefficients, inputs):
  """model.py:536-548: combine stencil coefficients with patches of ``inputs``."""
  return _lib.apply_coefficients(coefficients, inputs)


def apply_space_derivatives(derivatives, inputs, equation):
  """model.py:115-135: the equation of motion on given space derivatives
  [batch, x, derivative] (DERIVATIVE_NAMES order); no finalize step."""
  assert_consistent_solution(equation, inputs)
  spec = equation.kernel_spec()
  return _lib.apply_space_derivatives(spec['equation'], derivatives, inputs,
                                      spec['eta'], spec['dx'])


def predict_coefficients(inputs, model: LearnedStencilModel):
  return model.coefficients(inputs)


def predict_space_derivatives(inputs, model: LearnedStencilModel):
  return model.space_derivatives(inputs)


def _forcing_changes_rhs(model) -> bool:
  """A forcing table only enters the device right-hand side for the equations
  whose finalize_time_derivative adds forcing(t) (the Burgers family,
  equations.py:276-277; capi.hip: is_forced_family).  KdV / KS models may carry
  one -- ddd_set_forcing is a documented no-op for them."""
  return model._forcing is not None and model.equation.has_time_dependent_forcing


def predict_time_derivative(inputs, model: LearnedStencilModel):
  """model.py:618-640: equation of motion only, no finalize (no forcing)."""
  if _forcing_changes_rhs(model):
    raise ValueError('predict_time_derivative excludes forcing; call '
                     'model.time_derivative(y, t) for the finalized value')
  return model.time_derivative(inputs, 0.0)


def baseline_space_derivatives(inputs, equation, accuracy_order: int = 1):
  return BaselineModel(equation, accuracy_order).space_derivatives(inputs)


def integrate_ode(model: _DeviceModel, inputs, num_time_steps: int,
                  time_step: float):
  """model.py:138-159: midpoint rule, result [batch, x, num_time_steps].

  The reference takes ``func(y, t)``; here the model is the function: its
  right-hand side includes ``finalize_time_derivative`` (forcing at time t)
  when forcing has been set on the model, and not otherwise."""
  out = model.integrate_fixed(inputs, num_time_steps, dt=time_step,
                              scheme='midpoint')
  return out.permute(1, 2, 0)


def baseline_time_evolution(inputs, num_time_steps: int, equation):
  """model.py:162-183: time evolution with the baseline model -- fixed polynomial
  stencils at accuracy order 1, WITHOUT finalize_time_derivative (the reference's
  ``func`` drops ``t``: no forcing) -- by the midpoint rule with
  ``equation.time_step``.  Returns [batch, x, num_time_steps] (``integrate_ode`` drops
  the initial state, model.py:158-159; the reference's docstring says + 1).  One
  persistent launch of the fixed-stencil kernel for the whole batch."""
  return integrate_ode(BaselineModel(equation, accuracy_order=1), inputs, num_time_steps,
                       equation.time_step)


def predict_time_evolution(inputs, model: LearnedStencilModel):
  """model.py:643-661 (uses hparams.num_time_steps and equation.time_step).

  The reference's function drops ``t`` and integrates
  ``predict_time_derivative`` -- the equation of motion WITHOUT
  ``finalize_time_derivative`` (no forcing, model.py:655-657).  A model that
  has device forcing set would integrate a different right-hand side, so, like
  ``predict_time_derivative``, this refuses to run then; use
  ``integrate_ode(model, ...)`` / ``model.integrate_fixed`` for the forced
  trajectory.
  """
  if _forcing_changes_rhs(model):
    raise ValueError('predict_time_evolution excludes forcing (model.py:655-657); '
                     'call model.set_forcing(None) first, or use integrate_ode / '
                     'model.integrate_fixed for the forced trajectory')
  return integrate_ode(model, inputs, model.hparams.num_time_steps,
                       model.equation.time_step)
