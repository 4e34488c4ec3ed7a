"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Not part of the product path.

A NumPy restatement of the reference's per-timestep learned-stencil
integration path (google/data-driven-discretization-1d, package
``pde_superresolution``).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import this module, and only as the
checker / the timed CPU baseline.  The product (``data-driven-discretization-
1d_amd``) never imports it and fails loudly if its HIP library is missing.

Pinning status (see DESIGN.md "Oracle"):
  * PINNED against the reference itself (imported in the authoring container
    under a TensorFlow import stub, tests/golden/make_golden.py) for the parts
    whose reference implementation is NumPy: stencil grids / constraints /
    coefficients / null-space layers, Grid + resampling, RandomForcing,
    equation_of_motion of all nine equations, staggered derivative, and the
    SciPy RK23 driver ``integrate.odeint``; likewise (tests/golden/
    make_golden_exact.py) the WENO5 reconstructions, the spectral derivative
    and smoothing filter, SpectralDifferentiator and integrate_exact.
  * PINNED against the reference's own known-answer tables for periodic
    padding and convolution alignment (layers_test.py:49-86) and for stencil
    coefficients (polynomials_test.py:36-76, 116-157).
  * PARITY UNPINNED for the TensorFlow graph ops that cannot be executed here
    (TensorFlow 1.x is not installed and there is no network):
    ``tf.layers.conv1d`` with >1 channels, ``tf.extract_image_patches``,
    ``tf.einsum`` and ``tf.contrib.integrate.odeint_fixed``.  Their semantics
    are restated from the call sites cited below and from the TF-1.x
    documentation; the multi-channel convolution is additionally cross-checked
    against ``torch.nn.functional.conv1d`` on circularly padded input.

All arithmetic that the reference performs inside the TF graph is done in
float32 here, in the reference's operation order where that order is
observable from the Python source.  Functions cite the reference lines they
follow.
"""
import numpy as np

F32 = np.float32

# equation ids (same numbering as include/ddd1d.h)
EQ_BURGERS = 0
EQ_BURGERS_CONSERVATIVE = 1
EQ_KDV = 2
EQ_KDV_CONSERVATIVE = 3
EQ_KS = 4
EQ_KS_CONSERVATIVE = 5
EQ_BURGERS_GODUNOV = 6
EQ_KDV_GODUNOV = 7
EQ_KS_GODUNOV = 8


# ---------------------------------------------------------------------------
# layers.py
# ---------------------------------------------------------------------------
def pad_periodic(inputs, padding, center=False):
  """layers.py:39-83.  inputs [batch, x, channel]."""
  inputs = np.asarray(inputs)
  if inputs.ndim != 3:
    raise ValueError('inputs must be 3D for periodic padding')
  if padding == 0:
    return inputs
  n = inputs.shape[1]
  if center:
    repeats = (padding // 2) // n
  else:
    repeats = padding // n
  tiled = np.tile(inputs, (1, 1 + repeats, 1))
  if center:
    # -padding//2 is floor(-padding/2): the LEFT halo has ceil(padding/2)
    # points, the right halo floor(padding/2)  (layers.py:76-79).
    left = tiled[:, -padding // 2:, :]
    right = tiled[:, :padding // 2, :]
    return np.concatenate([left, inputs, right], axis=1)
  return np.concatenate([inputs, tiled[:, :padding, :]], axis=1)


def conv1d_valid(padded, filters, bias=None):
  """tf.nn.conv1d / tf.layers.conv1d, padding='VALID', stride 1.

  Cross-correlation: out[b,x,f] = sum_{k,c} padded[b,x+k,c] * filters[k,c,f].
  Accumulated tap by tap in float32.
  """
  padded = np.asarray(padded, dtype=F32)
  filters = np.asarray(filters, dtype=F32)
  k_size = filters.shape[0]
  n_out = padded.shape[1] - k_size + 1
  out = np.zeros(padded.shape[:1] + (n_out, filters.shape[2]), dtype=F32)
  for k in range(k_size):
    out += np.einsum('bxc,cf->bxf', padded[:, k:k + n_out, :], filters[k],
                     dtype=F32)
  if bias is not None:
    out = out + np.asarray(bias, dtype=F32)
  return out


def nn_conv1d_periodic(inputs, filters, center=False):
  """layers.py:95-100."""
  filters = np.asarray(filters)
  return conv1d_valid(pad_periodic(inputs, filters.shape[0] - 1, center),
                      filters)


_NONLINEARITIES = {
    'relu': lambda x: np.maximum(x, F32(0)),
    'relu6': lambda x: np.minimum(np.maximum(x, F32(0)), F32(6)),
    'tanh': np.tanh,
    'softplus': lambda x: np.logaddexp(x, F32(0)).astype(F32),
    'elu': lambda x: np.where(x > 0, x, np.expm1(np.minimum(x, F32(0)))
                              ).astype(F32),
}


def conv1d_periodic_layer(inputs, kernel, bias, activation=None, center=True, strides=1,
                          dilation_rate=1):
  """layers.py:103-137: pad (K - 1) * dilation_rate periodically, then
  tf.layers.conv1d(padding='valid', strides, dilation_rate):
    out[b, i, f] = b_f + sum_{k, c} W[k, c, f] padded[b, i * strides + k * dilation_rate, c],
    i < ceil(N / strides)."""
  kernel = np.asarray(kernel)
  k_size = kernel.shape[0]
  padded = pad_periodic(inputs, (k_size - 1) * dilation_rate, center)
  if strides == 1 and dilation_rate == 1:
    out = conv1d_valid(padded, kernel, bias)
  else:
    n = np.asarray(inputs).shape[1]
    n_out = -(-n // strides)
    out = np.zeros((padded.shape[0], n_out, kernel.shape[2]), F32)
    for k in range(k_size):
      rows = padded[:, k * dilation_rate:k * dilation_rate + (n_out - 1) * strides + 1:strides, :]
      out = out + np.einsum('bxc,cf->bxf', rows.astype(F32), kernel[k].astype(F32)).astype(F32)
    if bias is not None:
      out = (out + np.asarray(bias, F32)).astype(F32)
  if activation is not None:
    out = _NONLINEARITIES[activation](out).astype(F32)
  return out


# ---------------------------------------------------------------------------
# model.py
# ---------------------------------------------------------------------------
def extract_patches(inputs, size):
  """model.py:516-533.  [batch, x] -> [batch, x, size].

  patches[b, x, i] = inputs[b, (x + i - ceil((size-1)/2)) mod N].
  """
  inputs = np.asarray(inputs)
  padded = pad_periodic(inputs[..., None], size - 1, center=True)[..., 0]
  n = inputs.shape[1]
  return np.stack([padded[:, i:i + n] for i in range(size)], axis=-1)


def conv_stack(inputs, spec):
  """The conv tower of predict_coefficients / _multilayer_conv1d.

  model.py:449-458 and :492-495 (or :460-466 / :551-576): input scaling by the
  standard deviation, (num_layers-1) hidden layers with the nonlinearity, one
  linear output layer.  Returns [batch, x, C_out] float32.
  """
  net = np.asarray(inputs, dtype=F32)[:, :, None]
  net = net / F32(spec['standard_deviation'])
  kernels, biases = spec['conv_kernels'], spec['conv_biases']
  assert len(kernels) == spec['num_layers']
  for kernel, bias in zip(kernels[:-1], biases[:-1]):
    net = conv1d_periodic_layer(net, kernel, bias, spec['nonlinearity'])
  return conv1d_periodic_layer(net, kernels[-1], biases[-1], None)


def predict_coefficients(inputs, spec):
  """model.py:420-513 -> [batch, x, derivative, stencil]."""
  inputs = np.asarray(inputs, dtype=F32)
  num_derivatives = len(spec['derivative_orders'])
  grid_size = spec['stencil_size']

  if not spec['polynomial_accuracy_order']:
    # model.py:460-475: the net emits the coefficients directly.
    net = conv_stack(inputs, spec)
    out = net.reshape(inputs.shape + (num_derivatives, grid_size))
    if spec.get('ensure_unbiased_coefficients', False):
      out = out - out.mean(axis=-1, keepdims=True, dtype=F32)
    return out.astype(F32)

  if spec['num_layers'] > 0:
    net = conv_stack(inputs, spec)
  else:
    # model.py:496-502: one learned constant vector, tiled.
    const = np.asarray(spec['constant_coefficients'], dtype=F32)
    net = np.broadcast_to(const, inputs.shape + const.shape)

  out = []
  start = 0
  for nullspace, bias in zip(spec['nullspaces'], spec['biases']):
    nullspace = np.asarray(nullspace, dtype=F32)   # polynomials.py:275-276
    bias = np.asarray(bias, dtype=F32)
    stop = start + nullspace.shape[0]
    out.append(bias + np.einsum('bxi,ij->bxj', net[..., start:stop],
                                nullspace, dtype=F32))
    start = stop
  return np.stack(out, axis=-2).astype(F32)


def apply_coefficients(coefficients, inputs):
  """model.py:536-548."""
  patches = extract_patches(np.asarray(inputs, dtype=F32),
                            coefficients.shape[3])
  return np.einsum('bxdi,bxi->bxd', coefficients, patches, dtype=F32)


def predict_space_derivatives(inputs, spec):
  """model.py:579-600."""
  target = spec.get('model_target', 'coefficients')
  if target == 'coefficients':
    return apply_coefficients(predict_coefficients(inputs, spec), inputs)
  if target == 'space_derivatives':
    return conv_stack(inputs, spec)
  raise NotImplementedError('unrecognized model_target: {}'.format(target))


def baseline_space_derivatives(inputs, spec):
  """model.py:59-112 (explicit accuracy_order branch) via polynomials.py:280-303.

  ``spec['baseline_coefficients']`` holds one float64 filter per derivative
  (from polynomials.coefficients on regular_grid(GRID_OFFSET, d, acc, dx)).
  """
  inputs = np.asarray(inputs, dtype=F32)
  cols = []
  for taps in spec['baseline_coefficients']:
    filt = np.asarray(taps).astype(F32)[:, None, None]
    cols.append(nn_conv1d_periodic(inputs[..., None], filt, center=True)[..., 0])
  return np.stack(cols, axis=-1)


# ---------------------------------------------------------------------------
# weno.py -- in the dtype of ``u`` (float64 in the reference's NumPy callers,
# float32 when checking the GPU kernel)
# ---------------------------------------------------------------------------
WENO_OPTIMAL_WEIGHTS = (0.1, 0.6, 0.3)


def weno_smoothness_indicators(u):
  """weno.py:43-58, Equation (7) of Tang (2005); returns [..., 3, x]."""
  u = np.asarray(u)
  f = u.dtype.type
  m2, m1 = np.roll(u, 2, axis=-1), np.roll(u, 1, axis=-1)
  p1, p2 = np.roll(u, -1, axis=-1), np.roll(u, -2, axis=-1)
  q, r = f(1 / 4), f(13 / 12)
  return np.stack([
      q * (m2 - f(4) * m1 + f(3) * u) ** 2 + r * (m2 - f(2) * m1 + u) ** 2,
      q * (m1 - p1) ** 2 + r * (m1 - f(2) * u + p1) ** 2,
      q * (f(3) * u - f(4) * p1 + p2) ** 2 + r * (u - f(2) * p1 + p2) ** 2,
  ], axis=-2)


def weno_omega(u, weights=WENO_OPTIMAL_WEIGHTS, epsilon=1e-6, p=2):
  """weno.py:61-75: alpha = w / (eps + IS)^p, omega = alpha / sum(alpha)."""
  u = np.asarray(u)
  f = u.dtype.type
  indicator = weno_smoothness_indicators(u)
  alpha = np.array(weights, dtype=u.dtype)[:, np.newaxis] / (f(epsilon) + indicator) ** p
  return alpha / np.sum(alpha, axis=-2, keepdims=True)


def weno_reconstruct_left(u):
  """weno.py:78-101: u at the +1/2 edges from the left-biased stencil."""
  u = np.asarray(u)
  f = u.dtype.type
  om = weno_omega(u)
  o0, o1, o2 = om[..., 0, :], om[..., 1, :], om[..., 2, :]
  coeff = [o0 / f(3), -(f(7) * o0 + o1) / f(6),
           (f(11) * o0 + f(5) * o1 + f(2) * o2) / f(6),
           (f(2) * o1 + f(5) * o2) / f(6), -o2 / f(6)]
  total = 0
  for c, shift in zip(coeff, (2, 1, 0, -1, -2)):
    total = total + c * np.roll(u, shift, axis=-1)
  return total


def weno_reconstruct_right(u):
  """weno.py:104-130 (weights reversed, omega rolled by -1)."""
  u = np.asarray(u)
  f = u.dtype.type
  om = np.roll(weno_omega(u, WENO_OPTIMAL_WEIGHTS[::-1]), -1, axis=-1)
  o2, o1, o0 = om[..., 0, :], om[..., 1, :], om[..., 2, :]
  coeff = [-o2 / f(6), (f(5) * o2 + f(2) * o1) / f(6),
           (f(2) * o2 + f(5) * o1 + f(11) * o0) / f(6),
           -(o1 + f(7) * o0) / f(6), o0 / f(3)]
  total = 0
  for c, shift in zip(coeff, (1, 0, -1, -2, -3)):
    total = total + c * np.roll(u, shift, axis=-1)
  return total


# ---------------------------------------------------------------------------
# duckarray.py spectral helpers and integrate.SpectralDifferentiator (float64)
# ---------------------------------------------------------------------------
def spectral_derivative(x, order=1, period=2 * np.pi):
  """duckarray.py:105-113 (rfft form; keeps the Nyquist mode for odd orders)."""
  x = np.asarray(x)
  length = x.shape[-1]
  if length % 2:
    raise ValueError('spectral derivative only works for even length data')
  c = 2 * np.pi * 1j / period
  k = np.fft.rfftfreq(length, d=1 / length)
  return np.fft.irfft((c * k) ** order * np.fft.rfft(x))


def smoothing_filter(x, alpha=-np.log(1e-15), order=2):
  """duckarray.py:116-128 (Gottlieb & Hesthaven exponential filter)."""
  x = np.asarray(x)
  length = x.shape[-1]
  if length % 2:
    raise ValueError('smoothing filter only works for even length data')
  count = length // 2
  eta = np.arange(count + 1) / count
  sigma = np.exp(-alpha * eta ** (2 * order))
  return np.fft.irfft(sigma * np.fft.rfft(x))


def spectral_time_derivative(equation, y, derivative_orders, period, eta, dx):
  """integrate.SpectralDifferentiator.__call__ (integrate.py:113-121) without
  finalize: scipy.fftpack.diff per derivative (third-party: SciPy, unpinned in
  setup.py:25, called exactly as the reference calls it), then the equation
  of motion, all float64."""
  import scipy.fftpack
  y = np.asarray(y, dtype=np.float64)
  rows = y.reshape(-1, y.shape[-1])          # fftpack.diff is one-dimensional
  derivs = np.stack([
      np.stack([scipy.fftpack.diff(row, order, period) for row in rows]).reshape(y.shape)
      for order in derivative_orders], axis=-1)
  return equation_of_motion(equation, y, derivs, eta, dx)


# ---------------------------------------------------------------------------
# equations.py
# ---------------------------------------------------------------------------
def staggered_first_derivative(y, dx):
  """equations.py:305-320 in the dtype of ``y``."""
  y = np.asarray(y)
  forward = np.concatenate([y[..., 1:], y[..., :1]], axis=-1)
  return y.dtype.type(1 / dx) * (forward - y)


def godunov_convective_flux(u_minus, u_plus):
  """equations.py:341-349."""
  lo, hi = u_minus ** 2, u_plus ** 2
  half = u_minus.dtype.type(0.5)
  return half * np.where(u_minus <= u_plus, np.minimum(lo, hi),
                         np.maximum(lo, hi))


def equation_of_motion(equation, y, derivs, eta, dx):
  """equations.py:269-274, 331-338, 359-370, 410-415, 450-457, 468-478,
  518-524, 559-567, 576-587.  ``derivs`` [..., D] in DERIVATIVE_NAMES order."""
  y = np.asarray(y)
  c = y.dtype.type
  d = [derivs[..., i] for i in range(derivs.shape[-1])]
  if equation == EQ_BURGERS:
    return c(eta) * d[1] - y * d[0]
  if equation == EQ_BURGERS_CONSERVATIVE:
    flux = c(0.5) * d[0] ** 2 - c(eta) * d[1]
  elif equation == EQ_BURGERS_GODUNOV:
    flux = godunov_convective_flux(d[0], d[1]) - c(eta) * d[2]
  elif equation == EQ_KDV:
    return c(-6) * y * d[0] - d[1]
  elif equation == EQ_KDV_CONSERVATIVE:
    flux = c(3) * d[0] ** 2 + d[1]
  elif equation == EQ_KDV_GODUNOV:
    flux = c(6) * godunov_convective_flux(d[0], d[1]) + d[2]
  elif equation == EQ_KS:
    return -y * d[0] - d[2] - d[1]
  elif equation == EQ_KS_CONSERVATIVE:
    flux = c(0.5) * d[0] ** 2 + d[2] + d[1]
  elif equation == EQ_KS_GODUNOV:
    flux = d[3] + d[2] + godunov_convective_flux(d[0], d[1])
  else:
    raise ValueError('unknown equation id {}'.format(equation))
  return -staggered_first_derivative(flux, dx)


def forcing_f32(t, forcing, num_points, resample_factor, period, conservative):
  """RandomForcing.__call__ (equations.py:214-219) as the TF graph runs it.

  ``t`` is a float32 placeholder in the reference (integrate.py:54), so the
  phase ``omega*t + spatial_phase + phi`` is accumulated in float32 in that
  order; the float64 NumPy constants are cast on first contact.  ``forcing``
  is a dict of float64 arrays a/omega/k/phi with shape [batch, nparams]
  (one row per sample).  Returns float32 [batch, num_points].
  """
  a = np.asarray(forcing['a'], dtype=np.float64)
  omega = np.asarray(forcing['omega'], dtype=np.float64)
  k = np.asarray(forcing['k'], dtype=np.float64)
  phi = np.asarray(forcing['phi'], dtype=np.float64)
  n_ref = num_points * resample_factor
  reference_x = (period / n_ref) * np.arange(n_ref)
  spatial = 2 * np.pi * k[..., None] * reference_x / period   # f64 [B,P,Nref]
  phase = (omega.astype(F32) * F32(t))[..., None]
  phase = phase + spatial.astype(F32)
  phase = phase + phi.astype(F32)[..., None]
  waves = np.sin(phase, dtype=F32)
  total = np.sum(a.astype(F32)[..., None] * waves, axis=-2, dtype=F32)
  if conservative:   # Grid.resample: 'mean' for conservative equations
    return total.reshape(total.shape[:-1] + (num_points, resample_factor)
                         ).mean(axis=-1, dtype=F32)
  return total[..., ::resample_factor]


def forcing_f64(t, forcing, num_points, resample_factor, period, conservative):
  """Same as forcing_f32 but in float64 (what NumPy-side callers get)."""
  a, omega, k, phi = (np.asarray(forcing[key], dtype=np.float64)
                      for key in ('a', 'omega', 'k', 'phi'))
  n_ref = num_points * resample_factor
  reference_x = (period / n_ref) * np.arange(n_ref)
  spatial = 2 * np.pi * k[..., None] * reference_x / period
  waves = np.sin(omega[..., None] * t + spatial + phi[..., None])
  total = np.sum(a[..., None] * waves, axis=-2)
  if conservative:
    return total.reshape(total.shape[:-1] + (num_points, resample_factor)
                         ).mean(axis=-1)
  return total[..., ::resample_factor]


# ---------------------------------------------------------------------------
# The differentiator: integrate.py:48-105
# ---------------------------------------------------------------------------
def time_derivative(spec, t, y, forcing=None):
  """finalize_time_derivative(t, predict_time_derivative(y)) in float32.

  integrate.py:59-64 (SavedModelDifferentiator) when spec has a conv tower,
  integrate.py:85-92 (PolynomialDifferentiator) when
  ``spec['baseline_coefficients']`` is set.  ``y`` [batch, x] (any float
  dtype; cast to float32 like the placeholder feed).
  """
  y32 = np.asarray(y, dtype=F32)
  target = spec.get('model_target', 'coefficients')
  if spec.get('baseline_coefficients') is not None:
    derivs = baseline_space_derivatives(y32, spec)
    if spec.get('weno'):
      # integrate.py:134-138 / model.py:82-88: u_minus, u_plus replaced by the
      # WENO reconstructions, rolled one cell to the left edge
      derivs[..., 0] = np.roll(weno_reconstruct_left(y32), 1, axis=-1)
      derivs[..., 1] = np.roll(weno_reconstruct_right(y32), 1, axis=-1)
    y_t = equation_of_motion(spec['equation'], y32, derivs, spec['eta'],
                             spec['dx'])
  elif target == 'time_derivative':          # model.py:603-606
    y_t = conv_stack(y32, spec)[..., 0]
  elif target == 'flux':                     # model.py:609-615
    flux = conv_stack(y32, spec)[..., 0]
    y_t = staggered_first_derivative(flux, spec['dx'])
  else:
    derivs = predict_space_derivatives(y32, spec)
    y_t = equation_of_motion(spec['equation'], y32, derivs, spec['eta'],
                             spec['dx'])
  if spec.get('forced', False) and forcing is not None:
    y_t = y_t + forcing_f32(t, forcing, spec['num_points'],
                            spec['resample_factor'], spec['period'],
                            spec['conservative'])
  return y_t.astype(F32)


# ---------------------------------------------------------------------------
# Steppers
# ---------------------------------------------------------------------------
def odeint_rk23(spec, y0, times, forcing=None, method='RK23'):
  """integrate.odeint (integrate.py:143-169) for ONE sample.

  SciPy's adaptive solve_ivp with max_step=0.01 in float64, calling the
  float32 right-hand side once per stage; NaN-pads rows the solver did not
  reach.  Returns (y [time, x] float64, nfev).
  """
  import scipy.integrate
  one = None if forcing is None else {k: np.asarray(v)[None]
                                      for k, v in forcing.items()}

  def fun(t, y):
    return time_derivative(spec, t, y[None, :], one)[0]

  times = np.asarray(times, dtype=np.float64)
  sol = scipy.integrate.solve_ivp(fun, (times[0], times[-1]),
                                  np.asarray(y0, dtype=np.float64),
                                  t_eval=times, max_step=0.01, method=method)
  y = sol.y.T
  missing = len(times) - y.shape[0]
  if missing:
    y = np.pad(y, ((0, missing), (0, 0)), mode='constant',
               constant_values=np.nan)
  return y, sol.nfev


# SciPy's RK23 (scipy.integrate._ivp.rk.RK23 / RungeKutta / rk_step,
# _ivp.common.select_initial_step, _ivp.ivp.solve_ivp) restated statement by
# statement: the algorithm `ddd_integrate_adaptive_f64` runs on the device, one
# controller per sample.  SciPy is a third-party dependency of the reference
# (setup.py:25, unpinned; call site integrate.py:154-155); this restatement is
# pinned against the installed SciPy itself (tests/test_cpu_oracle.py:
# identical nfev, trajectories equal to float64 rounding).
RK23_C = (0.0, 0.5, 0.75)
RK23_A = ((0.0, 0.0, 0.0), (0.5, 0.0, 0.0), (0.0, 0.75, 0.0))
RK23_B = (2 / 9, 1 / 3, 4 / 9)
RK23_E = (5 / 72, -1 / 12, -1 / 9, 1 / 8)
RK23_P = ((1, -4 / 3, 5 / 9), (0, 1, -2 / 3), (0, 4 / 3, -8 / 9), (0, -1, 1))
RK_SAFETY, RK_MIN_FACTOR, RK_MAX_FACTOR = 0.9, 0.2, 10.0


def _rms(x):
  """_ivp.common.norm: np.linalg.norm(x) / x.size ** 0.5."""
  return np.sqrt(np.sum(x * x)) / x.size ** 0.5


def rk23_adaptive(fun, y0, times, rtol=1e-3, atol=1e-6, max_step=0.01):
  """solve_ivp(fun, (times[0], times[-1]), y0, t_eval=times, method='RK23',
  max_step=max_step) for ONE sample; ``fun(t, y)`` returns float32.

  Returns (y [time, x] float64 with NaN rows past a failure, nfev, status)
  where status 0 = reached times[-1], -1 = step size fell below the spacing
  of floating point numbers (SciPy's TOO_SMALL_STEP).
  """
  times = np.asarray(times, dtype=np.float64)
  t, t_bound = float(times[0]), float(times[-1])
  y = np.asarray(y0, dtype=np.float64).copy()
  out = np.full((len(times), y.size), np.nan)
  # RungeKutta.__init__: f = fun(t0, y0); select_initial_step (order 2)
  f = np.asarray(fun(t, y))
  nfev = 1
  interval = abs(t_bound - t)
  if interval == 0.0:
    out[0] = y
    return out, nfev, 0
  scale = atol + np.abs(y) * rtol
  d0 = _rms(y / scale)
  d1 = _rms(f / scale)
  h0 = 1e-6 if (d0 < 1e-5 or d1 < 1e-5) else 0.01 * d0 / d1
  h0 = h0 if not interval < h0 else interval
  f1 = np.asarray(fun(t + h0, y + h0 * f))
  nfev += 1
  d2 = _rms((f1 - f) / scale) / h0          # float32 difference, as SciPy forms it
  if d1 <= 1e-15 and d2 <= 1e-15:
    h1 = 1e-6 if not h0 * 1e-3 > 1e-6 else h0 * 1e-3
  else:
    h1 = (0.01 / (d1 if not d2 > d1 else d2)) ** (1 / 3)
  h_abs = 100 * h0
  for cand in (h1, interval, max_step):
    if cand < h_abs:
      h_abs = cand
  ti = 0
  k = [None] * 4
  while True:
    # RungeKutta._step_impl
    min_step = 10 * abs(np.nextafter(t, np.inf) - t)
    if h_abs > max_step:
      h_abs = max_step
    elif h_abs < min_step:
      h_abs = min_step
    rejected = False
    while True:
      if h_abs < min_step:
        return out, nfev, -1
      t_new = t + h_abs
      if t_new - t_bound > 0:
        t_new = t_bound
      h = t_new - t
      h_abs = abs(h)
      # rk_step
      k[0] = f.astype(np.float64)
      k[1] = np.asarray(fun(t + RK23_C[1] * h, y + (k[0] * RK23_A[1][0]) * h),
                        dtype=np.float64)
      k[2] = np.asarray(fun(t + RK23_C[2] * h,
                            y + (k[0] * RK23_A[2][0] + k[1] * RK23_A[2][1]) * h),
                        dtype=np.float64)
      y_new = y + h * ((k[0] * RK23_B[0] + k[1] * RK23_B[1]) + k[2] * RK23_B[2])
      f_new = np.asarray(fun(t + h, y_new))
      k[3] = f_new.astype(np.float64)
      nfev += 3
      scale = atol + np.maximum(np.abs(y), np.abs(y_new)) * rtol
      err = (((k[0] * RK23_E[0] + k[1] * RK23_E[1]) + k[2] * RK23_E[2])
             + k[3] * RK23_E[3]) * h
      error_norm = _rms(err / scale)
      if error_norm < 1:
        if error_norm == 0:
          factor = RK_MAX_FACTOR
        else:
          factor = RK_SAFETY * error_norm ** (-1 / 3)
          factor = RK_MAX_FACTOR if not factor < RK_MAX_FACTOR else factor
        if rejected:
          factor = 1 if not factor < 1 else factor
        h_abs *= factor
        break
      factor = RK_SAFETY * error_norm ** (-1 / 3)
      h_abs *= RK_MIN_FACTOR if not factor > RK_MIN_FACTOR else factor
      rejected = True
    t_old, y_old = t, y
    t, y, f = t_new, y_new, f_new
    # solve_ivp: dense output (RkDenseOutput) at every t_eval <= t
    while ti < len(times) and times[ti] <= t:
      x = (times[ti] - t_old) / h
      p1 = x
      p2 = p1 * x
      p3 = p2 * x
      q = [sum(k[s] * RK23_P[s][j] for s in range(4)) for j in range(3)]
      out[ti] = h * ((q[0] * p1 + q[1] * p2) + q[2] * p3) + y_old
      ti += 1
    if t - t_bound >= 0:
      return out, nfev, 0


SCHEME_EULER = 0
SCHEME_MIDPOINT = 1
SCHEME_BS3 = 2
SCHEME_RK4 = 3

EVALS_PER_STEP = {SCHEME_EULER: 1, SCHEME_MIDPOINT: 2, SCHEME_BS3: 3,
                  SCHEME_RK4: 4}


# Explicit RK tableaus in "previous stage only" form (all four schemes have
# u_s = y + a_s h k_{s-1}):  name -> (a, b, c)
TABLEAUS = {
    SCHEME_EULER: ((0.0,), (1.0,), (0.0,)),
    SCHEME_MIDPOINT: ((0.0, 0.5), (0.0, 1.0), (0.0, 0.5)),
    SCHEME_BS3: ((0.0, 0.5, 0.75), (2 / 9, 1 / 3, 4 / 9), (0.0, 0.5, 0.75)),
    SCHEME_RK4: ((0.0, 0.5, 0.5, 1.0), (1 / 6, 1 / 3, 1 / 3, 1 / 6),
                 (0.0, 0.5, 0.5, 1.0)),
}


def integrate_fixed(spec, scheme, t0, dt, num_steps, save_every, y0,
                    forcing=None, state_dtype=F32, apply_forcing=True):
  """Fixed-step explicit Runge-Kutta over the whole batch.

  SCHEME_MIDPOINT follows model.integrate_ode (model.py:138-159) =
  tf.contrib.integrate.odeint_fixed(method='midpoint') [tensorflow<2, not in
  /root/reference; restated from its published definition]:
      k1 = f(y, t); k2 = f(y + k1*dt/2, t + dt/2); y <- y + dt*k2
  with dt cast to the state dtype (multiplying by 1/2 is exact, so
  k1*(dt/2) == (k1*dt)/2 bit for bit).  The reference's training-time caller
  drops the forcing (model.py:655-657); pass apply_forcing=False for that.

  SCHEME_BS3 is the Bogacki-Shampine tableau SciPy's RK23 uses
  (integrate.py:154-155 pins max_step=0.01; with the controller saturated the
  accepted steps are exactly these) without error control: 3 evaluations per
  step.  State update: y' = ((y + b1 h k1) + b2 h k2) + b3 h k3 with the
  float32 products (b_s h) formed first, in ``state_dtype``.

  Returns y [num_saved, batch, x] in ``state_dtype`` where
  num_saved = num_steps // save_every (state after steps save_every, 2*..).
  """
  dtype = np.dtype(state_dtype).type
  y = np.asarray(y0).astype(dtype)
  h = dtype(dt)
  frc = forcing if apply_forcing else None
  a, b, c = TABLEAUS[scheme]

  saved = []
  for step in range(num_steps):
    t = t0 + step * dt
    ynew = y
    kprev = None
    for s in range(len(a)):
      us = y if s == 0 else y + kprev.astype(dtype) * (dtype(F32(a[s])) * h)
      f = time_derivative(spec, t + c[s] * dt, us, frc)
      if b[s] != 0.0:
        ynew = ynew + (dtype(F32(b[s])) * h) * f.astype(dtype)
      kprev = f
    y = ynew
    if (step + 1) % save_every == 0:
      saved.append(y.copy())
  if not saved:
    return np.zeros((0,) + y.shape, dtype=dtype)
  return np.stack(saved, axis=0)
