"""The fused right-hand-side kernel vs the CPU oracle, through the C ABI.

Tolerance: north_star asks for float32 results within 1e-5 (relative) of the
reference.  Single evaluations are compared in the max norm relative to the
largest reference value.  The MFMA / generic kernels and the oracle all use
IEEE float32 but sum in different orders, so the irreducible difference is a
few float32 ulps times the cancellation in the stencil apply (coefficients of
size 1/dx^d multiply O(1) values that cancel); TOL_RHS below is that bound and
is asserted, the measured value is printed.
"""
import numpy as np
import pytest

from helpers import (oracle, make_model, random_phase_ic, batch_forcing, rel_err, FLOOR_CEILING,
                     assert_near_truth)

pytestmark = pytest.mark.gpu

TOL = 1e-5          # north_star's float32 tolerance
# KS: 4th-derivative stencils (|c| ~ 6/dx^4) on smooth data cancel ~1e4-fold, so
# two float32 evaluations of the same formulas differ by more than 1e-5 whatever
# the hardware.  No fixed relaxation: every KS assertion is bounded by 4 x the
# MEASURED distance between the float32 oracle and a float64 evaluation of the
# same formulas on the same inputs (_measured_tol, printed).

ALL_EQUATIONS = [
    ('burgers', False, False), ('burgers', True, False), ('burgers', True, True),
    ('kdv', False, False), ('kdv', True, False), ('kdv', True, True),
    ('ks', False, False), ('ks', True, False), ('ks', True, True),
]


_F64_ACTIVATIONS = {
    'relu': lambda x: np.maximum(x, 0.0), 'relu6': lambda x: np.clip(x, 0.0, 6.0),
    'tanh': np.tanh, 'softplus': lambda x: np.logaddexp(x, 0.0),
    'elu': lambda x: np.where(x > 0, x, np.expm1(x)),
}


def _f64_coefficients(spec, y0):
  """predict_coefficients (model.py:420-513) evaluated ENTIRELY in float64 from
  the float32 weights and tables: the conv tower (layers.py:39-137 alignment:
  tap k reads x + k - ceil((K - 1) / 2)), the projection / direct head, the
  optional mean subtraction.  The distance of the float32 oracle from this is
  the float32 rounding noise of the formulas themselves."""
  net = (np.asarray(y0, np.float64) / np.float64(np.float32(spec['standard_deviation'])))[..., None]
  act = _F64_ACTIVATIONS[spec['nonlinearity']]
  layers = list(zip(spec['conv_kernels'], spec['conv_biases']))
  for index, (kernel, bias) in enumerate(layers):
    taps = kernel.shape[0]
    left = -(-(taps - 1) // 2)
    out = np.zeros(net.shape[:2] + (kernel.shape[2],))
    for k in range(taps):
      out += np.einsum('bxc,cf->bxf', np.roll(net, left - k, axis=1),
                       kernel[k].astype(np.float64))
    out += bias.astype(np.float64)
    net = act(out) if index < len(layers) - 1 else out
  num_d, size = len(spec['derivative_orders']), spec['stencil_size']
  if not spec['polynomial_accuracy_order']:
    coeff = net.reshape(net.shape[:2] + (num_d, size))
    if spec.get('ensure_unbiased_coefficients', False):
      coeff = coeff - coeff.mean(axis=-1, keepdims=True)
    return coeff
  out, start = [], 0
  for nullspace, bias in zip(spec['nullspaces'], spec['biases']):
    stop = start + nullspace.shape[0]
    out.append(np.float32(bias).astype(np.float64) + net[..., start:stop] @
               np.float32(nullspace).astype(np.float64))
    start = stop
  return np.stack(out, axis=-2)


def _f64_derivatives(spec, y0):
  """Stencil apply in float64 from the float32 coefficients; None for heads
  that do not predict coefficients."""
  if spec.get('model_target', 'coefficients') != 'coefficients' or spec['num_layers'] == 0:
    return None
  coeff = _f64_coefficients(spec, y0)
  patches = oracle.extract_patches(y0.astype(np.float64), coeff.shape[3])
  return np.einsum('bxdi,bxi->bxd', coeff, patches)


def _measured_tol(spec, y0, t, forcing, want):
  """max(1e-5, 4 x float32 noise floor of the oracle itself on these inputs)."""
  derivs = _f64_derivatives(spec, y0)
  if derivs is None:
    return TOL, 0.0, None
  truth = oracle.equation_of_motion(spec['equation'], y0.astype(np.float64), derivs,
                                    spec['eta'], spec['dx'])
  if spec.get('forced', False) and forcing is not None:
    truth = truth + oracle.forcing_f64(t, forcing, spec['num_points'],
                                       spec['resample_factor'], spec['period'],
                                       spec['conservative'])
  floor = rel_err(want, truth)
  assert floor < FLOOR_CEILING, ('float32 oracle vs float64 evaluation', floor)   # (never relax silently)
  return max(TOL, 4 * floor), floor, truth


def _check_all_views(model, y0, t, forcing, tol=None):
  spec = model.spec()
  got = model.time_derivative(y0, t).cpu().numpy()
  want = oracle.time_derivative(spec, t, y0, forcing)
  err = rel_err(got, want)
  if tol is None:
    tol, floor, truth = _measured_tol(spec, y0, t, forcing, want)
    if tol > TOL:
      print('float32 noise floor of the oracle {:.1e} -> bound {:.1e}; measured {:.1e}'
            .format(floor, tol, err))
      # the floor route: the device result is also as close to the float64 truth as the
      # float32 oracle is (x TRUTH_RATIO), not merely inside the 4 x triangle bound
      assert_near_truth(got, truth, floor, 'time_derivative ' + model.kernel_name)
  assert np.isfinite(got).all()
  assert err < tol, ('time_derivative', model.kernel_name, err)
  if spec.get('model_target', 'coefficients') in ('coefficients', 'space_derivatives'):
    d_got = model.space_derivatives(y0).cpu().numpy()
    d_want = oracle.predict_space_derivatives(y0, spec)
    d_truth = _f64_derivatives(spec, y0)
    for d in range(d_want.shape[-1]):
      e = rel_err(d_got[..., d], d_want[..., d])
      # per derivative: 1e-5, or 4 x the float32 oracle's own noise on THIS derivative
      # (a single high-order derivative cancels more than their combination u_t)
      d_floor = 0.0 if d_truth is None else rel_err(d_want[..., d], d_truth[..., d])
      bound = max(TOL, 4 * d_floor)
      assert e < bound, ('space_derivatives', d, model.kernel_name, e, bound)
      if bound > TOL:
        assert_near_truth(d_got[..., d], d_truth[..., d], d_floor,
                          'space_derivatives[%d] %s' % (d, model.kernel_name))
  if spec.get('model_target', 'coefficients') == 'coefficients':
    c_got = model.coefficients(y0).cpu().numpy()
    c_want = oracle.predict_coefficients(y0, spec)
    assert c_got.shape == c_want.shape
    assert rel_err(c_got, c_want) < 1e-5, ('coefficients', model.kernel_name)
  return err


@pytest.mark.parametrize('kernel', ['mfma64', 'mfma64w32', 'mfma256', 'generic'])
@pytest.mark.parametrize('equation,conservative,numerical_flux', ALL_EQUATIONS)
def test_time_derivative_all_equations(equation, conservative, numerical_flux,
                                       kernel):
  model = make_model(equation, conservative, numerical_flux, num_points=64,
                     resample_factor=4, seed=3)
  model.set_kernel(kernel)
  assert model.kernel_name.startswith('mfma_f32' if kernel != 'generic' else 'generic')
  batch = 7    # not a multiple of the 4 samples per workgroup
  y0 = random_phase_ic(model.equation, batch)
  forcing = batch_forcing(batch, seed0=50)
  model.set_forcing(forcing)
  err = _check_all_views(model, y0, 0.37, forcing, None if equation == 'ks' else TOL)
  print('{} cons={} flux={} {}: rel err {:.2e}'.format(
      equation, conservative, numerical_flux, kernel, err))


@pytest.mark.parametrize('num_points,batch', [(32, 9), (48, 6), (128, 3),
                                              (256, 2), (200, 2), (16, 33)])
def test_grid_sizes_mfma(num_points, batch):
  """Rows per workgroup = floor(256 / N) samples, incl. non-power-of-two N."""
  model = make_model('burgers', True, num_points=num_points, resample_factor=2)
  assert model.kernel_name.startswith('mfma_f32')
  y0 = random_phase_ic(model.equation, batch)
  forcing = batch_forcing(batch)
  model.set_forcing(forcing)
  _check_all_views(model, y0, 1.25, forcing, TOL)


@pytest.mark.parametrize('overrides', [
    dict(num_layers=2), dict(num_layers=4), dict(num_layers=5),
    dict(nonlinearity='relu6'), dict(nonlinearity='tanh'),
    dict(nonlinearity='softplus'), dict(nonlinearity='elu'),
    dict(polynomial_accuracy_order=2), dict(polynomial_accuracy_order=3),
    dict(polynomial_accuracy_scale=0.5),
])
def test_network_variants_mfma(overrides):
  model = make_model('burgers', False, num_points=64, **overrides)
  assert model.kernel_name.startswith('mfma_f32'), overrides
  y0 = random_phase_ic(model.equation, 5)
  forcing = batch_forcing(5)
  model.set_forcing(forcing)
  tol = TOL
  mfma_err = _check_all_views(model, y0, 0.1, forcing, tol)
  model.set_kernel('generic')
  generic_err = _check_all_views(model, y0, 0.1, forcing, tol)
  print(overrides, mfma_err, generic_err)


@pytest.mark.parametrize('equation', ['burgers', 'kdv'])
@pytest.mark.parametrize('overrides', [
    dict(polynomial_accuracy_order=0),
    dict(polynomial_accuracy_order=0, ensure_unbiased_coefficients=True),
    dict(model_target='space_derivatives'),
    dict(model_target='time_derivative'), dict(model_target='flux'),
])
def test_direct_heads_on_mfma(equation, overrides):
  """training_test.py:54-83 variants whose conv tower is the default one: the
  D x G coefficients without the accuracy projection (model.py:460-475, with
  and without mean subtraction) and the heads that predict the derivatives, u_t
  or the flux directly (model.py:551-615) run on the MFMA tower too; both kernel
  families against the oracle, all views."""
  conservative = not overrides.get('ensure_unbiased_coefficients', False)
  model = make_model(equation, conservative, num_points=64, **overrides)
  assert model.kernel_name.startswith('mfma_f32'), overrides
  y0 = random_phase_ic(model.equation, 5)
  forcing = batch_forcing(5)
  model.set_forcing(forcing)
  # polynomial_accuracy_order = 0: unconstrained coefficients do not sum to zero, so
  # the stencil apply cancels less cleanly; bound = max(1e-5, 4 x the float32 oracle's
  # measured distance from the all-float64 evaluation) (tol=None), floor printed
  tol = None if overrides.get('polynomial_accuracy_order', 1) == 0 else TOL
  mfma_err = _check_all_views(model, y0, 0.2, forcing, tol)
  model.set_kernel('mfma256')
  _check_all_views(model, y0, 0.2, forcing, tol)
  model.set_kernel('generic')
  generic_err = _check_all_views(model, y0, 0.2, forcing, tol)
  print(equation, overrides, mfma_err, generic_err)
  if tol is None:
    spec = model.spec()
    tol, _, _ = _measured_tol(spec, y0, 0.2, forcing, oracle.time_derivative(spec, 0.2, y0, forcing))
  # and over a few steps of the persistent integrator
  model.set_kernel('auto')
  dt = model.equation.time_step
  got = model.integrate_fixed(y0, 10, dt=dt, scheme='midpoint', save_every=10).cpu().numpy()
  want = oracle.integrate_fixed(model.spec(), oracle.SCHEME_MIDPOINT, 0.0, dt, 10, 10, y0,
                                forcing=forcing if equation == 'burgers' else None)
  assert rel_err(got, want) < tol


def test_accuracy_order_zero_with_three_derivatives_runs_on_the_wide_kernels():
  """3 x 6 = 18 direct coefficient channels (training_test.py:57): beyond the
  default MFMA kernels' 16, carried by the wide flavour; same numbers as the
  generic kernel."""
  model = make_model('ks', True, num_points=64, polynomial_accuracy_order=0)
  assert model.kernel_name == 'mfma_f32_r64'
  y0 = random_phase_ic(model.equation, 3)
  a = model.time_derivative(y0, 0.0).cpu().numpy()
  model.set_kernel('generic')
  b = model.time_derivative(y0, 0.0).cpu().numpy()
  assert model.kernel_name == 'generic' and rel_err(a, b) < TOL


@pytest.mark.parametrize('overrides', [
    dict(filter_size=16), dict(kernel_size=3), dict(kernel_size=4), dict(kernel_size=1),
    dict(filter_size=8, kernel_size=2, num_layers=4, nonlinearity='softplus'),
    dict(filter_size=24, kernel_size=3, coefficient_grid_min_size=9),
])
def test_smaller_towers_embedded_in_the_mfma_layers(overrides):
  """kernel_size < 5 and filter_size < 32 ride the 5-tap x 32-channel MFMA layers
  with zero-padded weights (capi.hip: ddd_model_create): exact embedding, so the
  oracle evaluating the TRUE net is matched at the usual tolerance."""
  for equation, conservative in (('burgers', True), ('ks', False)):
    model = make_model(equation, conservative, num_points=64, **overrides)
    assert model.kernel_name == 'mfma_f32_r64', (overrides, model.kernel_name)
    y0 = random_phase_ic(model.equation, 5)
    forcing = batch_forcing(5)
    model.set_forcing(forcing)
    _check_all_views(model, y0, 0.2, forcing, None)
    dt = 1e-5
    got = model.integrate_fixed(y0, 10, dt=dt, save_every=10).cpu().numpy()
    want = oracle.integrate_fixed(model.spec(), oracle.SCHEME_MIDPOINT, 0.0, dt, 10, 10, y0,
                                  forcing=forcing if equation == 'burgers' else None)
    assert rel_err(got, want) < TOL, (equation, overrides)


@pytest.mark.parametrize('overrides', [
    dict(kernel_size=7), dict(filter_size=64),
    dict(kernel_size=6, filter_size=20),                       # embedded in 7 taps x 32
    dict(filter_size=40, kernel_size=4, num_layers=4, nonlinearity='tanh'),   # in 5 x 64
    dict(kernel_size=7, model_target='time_derivative'),
    dict(filter_size=64, model_target='space_derivatives'),
    dict(kernel_size=3, num_layers=2),                         # the 3-tap tower, no hidden layer
    dict(kernel_size=7, filter_size=64),                       # hidden layers as a loop over the taps
    dict(kernel_size=6, filter_size=48, num_layers=4, nonlinearity='tanh'),   # embedded in 7 x 64
    dict(kernel_size=3, filter_size=64),                       # embedded in 5 x 64
    dict(kernel_size=7, filter_size=64, model_target='time_derivative'),
])
def test_other_towers_on_mfma(overrides):
  """training.py:134-136 leaves kernel_size and filter_size free, model.py:455-458
  builds whatever they say: 7 taps, 64 filters, both, and 3 taps have MFMA towers of
  their own (rhs_mfma.h Tower<7, 1>, <5, 2>, <7, 2>, <3, 1>: weights streamed from L2),
  nets in between are embedded with zero weights.  One-wave (N = 64) and
  four-wave (N = 96) groups, all views and 10 midpoint steps against the oracle
  evaluating the TRUE net; the generic kernel agrees; launch modes agree bit
  for bit."""
  for equation, conservative, num_points in (('burgers', True, 64), ('ks', False, 64),
                                             ('kdv', True, 96)):
    model = make_model(equation, conservative, num_points=num_points, resample_factor=2,
                       **overrides)
    want_kernel = 'mfma_f32_r64' if num_points == 64 else 'mfma_f32_r256'
    assert model.kernel_name == want_kernel, (overrides, model.kernel_name)
    batch = 7
    y0 = random_phase_ic(model.equation, batch)
    forcing = batch_forcing(batch)
    model.set_forcing(forcing)
    mfma_err = _check_all_views(model, y0, 0.2, forcing, None)
    dt = 1e-5
    got = model.integrate_fixed(y0, 10, dt=dt, save_every=10).cpu().numpy()
    want = oracle.integrate_fixed(model.spec(), oracle.SCHEME_MIDPOINT, 0.0, dt, 10, 10, y0,
                                  forcing=forcing if equation == 'burgers' else None)
    assert rel_err(got, want) < TOL, (equation, overrides)
    per_substep = model.integrate_fixed(y0, 10, dt=dt, save_every=10,
                                        launch_mode='per_substep').cpu().numpy()
    np.testing.assert_array_equal(got, per_substep)
    f64 = model.integrate_fixed(y0, 10, dt=dt, save_every=10, state_dtype='float64').cpu().numpy()
    assert rel_err(f64, want) < TOL
    model.set_kernel('generic')
    generic_err = _check_all_views(model, y0, 0.2, forcing, None)
    print(equation, num_points, overrides, mfma_err, generic_err)


@pytest.mark.parametrize('overrides', [
    dict(num_layers=1), dict(num_layers=1, kernel_size=3), dict(num_layers=1, kernel_size=7),
    dict(num_layers=1, polynomial_accuracy_order=0),
    dict(num_layers=1, filter_size=8, nonlinearity='tanh'),    # (neither is used by a 1-layer net)
])
def test_one_layer_nets_on_the_valu_route(overrides):
  """num_layers = 1 (training.create_hparams admits it; integrate_test.py:48 names it in
  `model_kwargs` but never passes it, so the reference's own tests train three layers) has no
  hidden activations: its coefficients are affine in the K neighbouring values, folded on
  the host (conv layer x null space + accuracy bias) and evaluated on the VALU route of the
  MFMA-path kernels (DevParams::linear_taps) instead of the generic kernel.  Every equation
  form, both geometries, all views, 10 steps, launch modes bit-equal, generic kernel agrees."""
  for equation, conservative, num_points in (('burgers', True, 64), ('burgers', False, 32),
                                             ('kdv', True, 96), ('ks', False, 64),
                                             ('ks', True, 256)):
    model = make_model(equation, conservative, num_points=num_points, resample_factor=2,
                       **overrides)
    want_kernel = 'mfma_f32_r64' if 64 % num_points == 0 else 'mfma_f32_r256'
    if overrides.get('kernel_size', 5) * len(model.equation.DERIVATIVE_ORDERS) > 16:
      want_kernel = 'generic'   # (7 taps x 3 derivatives: more table rows than the LDS table holds)
    assert model.kernel_name == want_kernel, (overrides, model.kernel_name)
    batch = 7
    y0 = random_phase_ic(model.equation, batch)
    forcing = batch_forcing(batch)
    model.set_forcing(forcing)
    valu_err = _check_all_views(model, y0, 0.2, forcing, None)
    dt = 1e-5
    got = model.integrate_fixed(y0, 10, dt=dt, save_every=10).cpu().numpy()
    want = oracle.integrate_fixed(model.spec(), oracle.SCHEME_MIDPOINT, 0.0, dt, 10, 10, y0,
                                  forcing=forcing if equation == 'burgers' else None)
    assert rel_err(got, want) < TOL, (equation, overrides)
    per_substep = model.integrate_fixed(y0, 10, dt=dt, save_every=10,
                                        launch_mode='per_substep').cpu().numpy()
    np.testing.assert_array_equal(got, per_substep)
    model.set_kernel('generic')
    generic_err = _check_all_views(model, y0, 0.2, forcing, None)
    print(equation, num_points, overrides, valu_err, generic_err)


@pytest.mark.parametrize('overrides', [
    dict(num_layers=1, polynomial_accuracy_order=0, ensure_unbiased_coefficients=True),
    dict(filter_size=96), dict(kernel_size=9),
    dict(coefficient_grid_min_size=13), dict(num_layers=0),
])
def test_generic_only_variants(overrides):
  """Configurations the MFMA path does not cover (more than 64 filters or 7 taps,
  one-layer nets with mean-subtracted coefficients, stencils wider than 12) run on
  the generic kernel (never on the CPU)."""
  conservative = not overrides.get('ensure_unbiased_coefficients', False)
  model = make_model('burgers', conservative, num_points=64, **overrides)
  if overrides.get('num_layers', 3) != 0:
    assert model.kernel_name == 'generic', overrides
    with pytest.raises(Exception, match='MFMA path unavailable'):
      model.set_kernel('mfma')
  y0 = random_phase_ic(model.equation, 3)
  forcing = batch_forcing(3)
  model.set_forcing(forcing)
  _check_all_views(model, y0, 0.2, forcing, TOL)


@pytest.mark.parametrize('equation,conservative,num_points,resample_factor', [
    ('burgers', True, 4, 64),    # training_test.py:63: resample_factor = 64 on 256 points
    ('burgers', False, 6, 32), ('ks', True, 4, 16), ('kdv', False, 6, 16),
])
def test_grids_narrower_than_the_stencil(equation, conservative, num_points, resample_factor):
  """N = 4 / 6 with 6- and 7-point stencils and 5-tap convolutions: the periodic
  padding wraps more than once around the grid (layers.py:70-75 tiles the input),
  so stencil points and conv taps alias.  The MFMA path starts at N = 8; these run
  on the generic kernel: all views vs the oracle, then 20 midpoint steps."""
  model = make_model(equation, conservative, num_points=num_points,
                     resample_factor=resample_factor)
  assert model.stencil_size > num_points
  assert model.kernel_name == 'generic'
  batch = 5
  y0 = random_phase_ic(model.equation, batch)
  forcing = batch_forcing(batch) if equation == 'burgers' else None
  if forcing is not None:
    model.set_forcing(forcing)
  _check_all_views(model, y0, 0.1, forcing, None)
  dt = model.equation.time_step
  got = model.integrate_fixed(y0, 20, dt=dt, scheme='midpoint', save_every=10).cpu().numpy()
  want = oracle.integrate_fixed(model.spec(), oracle.SCHEME_MIDPOINT, 0.0, dt, 20, 10, y0,
                                forcing=forcing)
  assert np.isfinite(got).all() and rel_err(got, want) < TOL
  per_substep = model.integrate_fixed(y0, 20, dt=dt, scheme='midpoint', save_every=10,
                                      launch_mode='per_substep').cpu().numpy()
  np.testing.assert_array_equal(got, per_substep)


def test_unforced_and_zero_state():
  model = make_model('burgers', True, num_points=64)
  y0 = np.zeros((4, 64), np.float32)        # the reference's Burgers IC
  got = model.time_derivative(y0, 0.0).cpu().numpy()
  want = oracle.time_derivative(model.spec(), 0.0, y0, None)
  assert rel_err(got, want) < TOL or np.abs(want).max() < 1e-6
  forcing = batch_forcing(4)
  model.set_forcing(forcing)
  got = model.time_derivative(y0, 2.0).cpu().numpy()
  want = oracle.time_derivative(model.spec(), 2.0, y0, forcing)
  assert rel_err(got, want) < TOL
  model.set_forcing(None)
  got = model.time_derivative(y0, 2.0).cpu().numpy()
  assert np.abs(got).max() < 1e-5


def test_forcing_resample_factors():
  """Block-mean forcing folded into amplitude/phase vs the reference-grid sum."""
  for rf in (1, 2, 8, 16):
    for conservative in (False, True):
      model = make_model('burgers', conservative, num_points=32,
                         resample_factor=rf)
      forcing = batch_forcing(6, seed0=rf)
      model.set_forcing(forcing)
      y0 = np.zeros((6, 32), np.float32)
      for t in (0.0, 3.3, 47.0):
        got = model.time_derivative(y0, t).cpu().numpy()
        want = oracle.time_derivative(model.spec(), t, y0, forcing)
        # |forcing| ~ 1; float32 phase rounding at t = 47 is ~4e-6 per mode
        assert np.abs(got - want).max() < 2e-5, (rf, conservative, t)


def test_empty_batch_and_bad_shapes():
  model = make_model('burgers', True, num_points=64)
  out = model.time_derivative(np.zeros((0, 64), np.float32), 0.0)
  assert tuple(out.shape) == (0, 64)
  with pytest.raises(ValueError, match='unexpected size'):
    model.time_derivative(np.zeros((2, 32), np.float32), 0.0)
  model.set_forcing(batch_forcing(2))
  with pytest.raises(Exception, match='exceeds'):
    model.time_derivative(np.zeros((3, 64), np.float32), 0.0)


NAN_CASES = [
    # (kernel, equation, conservative, num_points, resample_factor, hparam overrides)
    ('mfma64', 'burgers', True, 64, 4, {}),                       # per-equation kernel, flux form
    ('mfma64', 'kdv', False, 32, 4, {}),                          # two samples per wavefront
    ('mfma64w32', 'burgers', True, 64, 4, {}),                    # one sample on two 32-row wavefronts
    ('mfma64w16', 'burgers', True, 64, 4, {}),                    # ... on four 16-row wavefronts
    ('mfma64w16', 'ks', False, 32, 4, {}),
    ('mfma256', 'ks', True, 128, 2, {}),                          # four-wave groups, sample = 2 wavefronts
    ('mfma256', 'burgers', False, 96, 2, {}),                     # ... N not a power of two
    ('mfma64', 'burgers', True, 64, 4, {'nonlinearity': 'relu6', 'num_layers': 4}),   # run-time kernel
    ('mfma64', 'burgers', True, 64, 4, {'nonlinearity': 'tanh'}),                     # (propagates by itself)
    ('mfma64', 'kdv', True, 64, 4, {'kernel_size': 7}),           # streamed tower
    ('mfma64', 'ks', True, 64, 4, {'coefficient_grid_min_size': 9}),   # wide flavour
    ('mfma64', 'burgers', True, 64, 4, {'model_target': 'time_derivative'}),   # direct head: the net IS u_t
    ('generic', 'burgers', True, 64, 4, {}),
    ('generic', 'kdv', False, 48, 2, {'kernel_size': 4}),         # even taps: asymmetric reach
    ('auto', 'burgers', True, 64, 4, {'num_layers': 1}),          # lean kernel: no activation at all
]


@pytest.mark.parametrize('kernel,equation,conservative,n,rf,overrides', NAN_CASES)
def test_nan_mask_equals_the_oracles(kernel, equation, conservative, n, rf, overrides):
  """Divergence is signalled by NaN, never clamped (integrate.py:161-167) -- and at exactly
  the grid points where the reference's arithmetic puts it.  np.maximum / Eigen's relu pass
  NaN on, so one NaN in the state poisons the net's receptive field (+ the stencil, + the
  flux difference); the MFMA kernels' relu is a clamp that maps NaN to 0 and they restore
  the mask explicitly (rhs_mfma.h::eval_rhs, "NaN through relu").  np.isnan of the device
  result must EQUAL the oracle's: one evaluation, its derivative view, and three midpoint
  steps (six evaluations: the mask widens by one reach per evaluation)."""
  model = make_model(equation, conservative, num_points=n, resample_factor=rf, **overrides)
  if kernel != 'auto':
    model.set_kernel(kernel)
  batch = 5
  forcing = batch_forcing(batch) if equation == 'burgers' else None
  model.set_forcing(forcing)
  spec = model.spec()
  y0 = random_phase_ic(model.equation, batch)
  y0[2, 10] = np.nan
  y0[4, n - 1] = np.nan          # reach wraps around the periodic boundary
  y0[4, 3] = np.nan
  got = model.time_derivative(y0, 0.1).cpu().numpy()
  want = oracle.time_derivative(spec, 0.1, y0, forcing)
  assert np.isnan(want[2]).any()
  np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
  assert np.isfinite(got[~np.isnan(want)]).all()
  if spec.get('model_target', 'coefficients') in ('coefficients', 'space_derivatives'):
    np.testing.assert_array_equal(np.isnan(model.space_derivatives(y0).cpu().numpy()),
                                  np.isnan(oracle.predict_space_derivatives(y0, spec)))
  got = model.integrate_fixed(y0, 3, dt=1e-4, scheme='midpoint').cpu().numpy()
  want = oracle.integrate_fixed(spec, oracle.SCHEME_MIDPOINT, 0.0, 1e-4, 3, 1, y0, forcing=forcing)
  np.testing.assert_array_equal(np.isnan(got), np.isnan(want))
  # samples without a NaN never see one
  assert np.isfinite(got[:, [0, 1, 3]]).all()
  if kernel.startswith('mfma'):
    per = model.integrate_fixed(y0, 3, dt=1e-4, scheme='midpoint', launch_mode='per_substep').cpu().numpy()
    np.testing.assert_array_equal(np.isnan(per), np.isnan(want))


@pytest.mark.parametrize('cls_name,n,order,weno,exact', [
    ('BurgersEquation', 64, 1, False, True),              # lean kernel, 3-point stencils padded to 6 columns
    ('ConservativeKdVEquation', 32, 3, False, False),     # lean kernel, flux form, stencils of 4 and 6 points
    ('KSEquation', 128, 1, False, False),                 # MFMA-path kernel, tower skipped; 3 / 3 / 5 points
    ('GodunovBurgersEquation', 128, 3, True, True),       # rhs_weno.h
    ('GodunovKSEquation', 64, 3, True, True),
    ('GodunovKdVEquation', 96, 3, True, True),            # generic kernel
])
def test_nan_mask_of_fixed_stencil_models(cls_name, n, order, weno, exact):
  """Fixed stencils / WENO5: no activation anywhere, the NaN flows by itself.  Where every
  derivative's stencil has the common width (or lies inside the WENO window) the mask
  EQUALS the oracle's -- columns a kernel pads with zeros read the grid point itself, never
  a neighbour the reference does not touch (0 x NaN = NaN).  Derivatives of different
  widths share one zero-padded table ([D][G], the ABI of ddd_baseline_create): there the
  narrower stencil multiplies the wider one's outer points by 0, and the device mask is a
  SUPERSET of the oracle's, by at most the zero-padded window the kernel reads (the
  streaming kernel's aligned 8-point window: <= 5 points) -- asserted as such (DESIGN.md
  section 5)."""
  from ddd1d_amd import equations, model as model_lib
  eq = getattr(equations, cls_name)(n, random_seed=1)
  model = model_lib.BaselineModel(eq, order, weno=weno)
  batch = 4
  forcing = batch_forcing(batch) if eq.has_time_dependent_forcing else None
  model.set_forcing(forcing)
  spec = model.spec()
  y0 = random_phase_ic(eq, batch)
  y0[1, 7] = np.nan
  y0[3, 0] = np.nan

  def check(got, want):
    got, want = np.isnan(got), np.isnan(want)
    if exact:
      np.testing.assert_array_equal(got, want)
      return
    assert (got | ~want).all()                   # every NaN of the reference is there
    allowed = want.copy()
    for shift in range(-5, 6):
      allowed |= np.roll(want, shift, axis=-1)
    assert (allowed | ~got).all()                # ... and nothing beyond the padded window around it
  check(model.time_derivative(y0, 0.1).cpu().numpy(), oracle.time_derivative(spec, 0.1, y0, forcing))
  dt = 1e-4 * eq.time_step
  got = model.integrate_fixed(y0, 1, dt=dt, scheme='euler').cpu().numpy()
  check(got, oracle.integrate_fixed(spec, oracle.SCHEME_EULER, 0.0, dt, 1, 1, y0, forcing=forcing))
  assert np.isfinite(got[:, [0, 2]]).all()     # samples without a NaN never see one


def test_batch_independence_and_determinism():
  """Samples never interact: permuting the batch permutes the result bit-exactly."""
  model = make_model('kdv', True, num_points=64)
  y0 = random_phase_ic(model.equation, 37)
  a = model.time_derivative(y0, 0.0).cpu().numpy()
  b = model.time_derivative(y0, 0.0).cpu().numpy()
  np.testing.assert_array_equal(a, b)
  perm = np.random.RandomState(0).permutation(37)
  c = model.time_derivative(y0[perm], 0.0).cpu().numpy()
  np.testing.assert_array_equal(a[perm], c)


def _f64_truth(spec, y0):
  """Same formulas in float64 from the float32 coefficients (no forcing)."""
  coeff = oracle.predict_coefficients(y0, spec).astype(np.float64)
  patches = oracle.extract_patches(y0.astype(np.float64), coeff.shape[3])
  derivs = np.einsum('bxdi,bxi->bxd', coeff, patches)
  return oracle.equation_of_motion(spec['equation'], y0.astype(np.float64),
                                   derivs, spec['eta'], spec['dx'])


def test_mfma_matches_generic_closely():
  """BASELINE config 4 shape (KS N=256): both kernel families sit inside the
  float32 noise band around the float64 evaluation of the same formulas."""
  model = make_model('ks', True, num_points=256, resample_factor=1)
  spec = model.spec()
  y0 = random_phase_ic(model.equation, 3)
  truth = _f64_truth(spec, y0)
  floor = rel_err(oracle.time_derivative(spec, 0.0, y0, None), truth)
  a = model.time_derivative(y0, 0.0).cpu().numpy()
  model.set_kernel('generic')
  b = model.time_derivative(y0, 0.0).cpu().numpy()
  print('KS N=256: oracle-f32 floor {:.2e}, mfma {:.2e}, generic {:.2e}'.format(
      floor, rel_err(a, truth), rel_err(b, truth)))
  assert rel_err(a, truth) < max(4 * floor, TOL)
  assert rel_err(b, truth) < max(4 * floor, TOL)


def test_f32_noise_floor_ks():
  """Documents why KS gets a looser bound: the float32 oracle itself differs
  from a float64 evaluation of the same formulas by more than 1e-5."""
  model = make_model('ks', False, num_points=64)
  spec = model.spec()
  y0 = random_phase_ic(model.equation, 4)
  f32 = oracle.time_derivative(spec, 0.0, y0, None)
  f64 = _f64_truth(spec, y0)
  floor = rel_err(f32, f64)
  got = model.time_derivative(y0, 0.0).cpu().numpy()
  print('KS float32 noise floor {:.2e}; HIP vs f64 {:.2e}'.format(
      floor, rel_err(got, f64)))
  assert rel_err(got, f64) < max(4 * floor, TOL)


def test_conservation_large_batch():
  """Flux-form equations conserve the mean: sum_x u_t = 0 up to rounding
  (integrate_test.py:101-104, 183-185 check this on trajectories)."""
  model = make_model('burgers', True, num_points=64)
  batch = 4096
  y0 = random_phase_ic(model.equation, batch)
  got = model.time_derivative(y0, 0.0).cpu().numpy().astype(np.float64)
  assert np.abs(got.sum(axis=1)).max() < 1e-3 * np.abs(got).max()


def test_rows_per_workgroup_selection():
  """Geometry: one free-running 64-row wavefront per workgroup when N divides
  64 (two 32-row wavefronts on request), else 256 rows."""
  for n, name in [(64, 'mfma_f32_r64'), (32, 'mfma_f32_r64'),
                  (16, 'mfma_f32_r64'), (48, 'mfma_f32_r256'),
                  (128, 'mfma_f32_r256'), (256, 'mfma_f32_r256')]:
    model = make_model('kdv', False, num_points=n, resample_factor=1)
    model.time_derivative(random_phase_ic(model.equation, 3), 0.0)   # small batch
    assert model.kernel_name == name, (n, model.kernel_name)
  model = make_model('kdv', False, num_points=64, resample_factor=1)
  model.set_kernel('mfma64w32')
  assert model.kernel_name == 'mfma_f32_r64w32'
  model = make_model('kdv', False, num_points=48, resample_factor=1)
  with pytest.raises(Exception, match='divide 64'):
    model.set_kernel('mfma64')
  model = make_model('burgers', True, num_points=32)
  y0 = random_phase_ic(model.equation, 11)
  forcing = batch_forcing(11)
  model.set_forcing(forcing)
  results = {}
  for kind in ('mfma64', 'mfma64w32', 'mfma256'):
    model.set_kernel(kind)
    results[kind] = model.time_derivative(y0, 0.4).cpu().numpy()
  assert model.kernel_name == 'mfma_f32_r256'
  # same arithmetic, different tiling
  np.testing.assert_array_equal(results['mfma64'], results['mfma64w32'])
  np.testing.assert_array_equal(results['mfma64'], results['mfma256'])


def test_standard_deviation_without_exact_division_shortcut():
  """rhs_mfma.h scales the input with a three-instruction reciprocal + one
  correction; ddd_model_create checks exhaustively (2^23 significands) that this
  equals u / std bit for bit for the model's standard deviation and falls back
  to the true division otherwise.  float32(1.9999999) (all-ones significand) is
  such a value; the reference constants 0.7917 / 0.594 / 0.299 are not."""
  model = make_model('kdv', True, num_points=64)
  model.equation._STANDARD_DEVIATION = float(np.float32(1.9999999))   # this instance only
  if True:
    spec = model.spec()
    assert spec['standard_deviation'] == float(np.float32(1.9999999))
    y0 = random_phase_ic(model.equation, 5)
    got = model.time_derivative(y0, 0.0).cpu().numpy()
    assert model.kernel_name.startswith('mfma_f32')
    assert rel_err(got, oracle.time_derivative(spec, 0.0, y0, None)) < TOL
    traj = model.integrate_fixed(y0, 20, dt=2.5e-5, save_every=20).cpu().numpy()
    want = oracle.integrate_fixed(spec, oracle.SCHEME_MIDPOINT, 0.0, 2.5e-5, 20, 20, y0)
    assert rel_err(traj, want) < TOL


# ---------------------------------------------------------------------------
# The wide flavour of the run-time-parameterised MFMA kernels: stencils up to 12
# points, up to 24 output channels (training_test.py:56-57: ks with
# coefficient_grid_min_size = 9; ks with polynomial_accuracy_order = 0)
# ---------------------------------------------------------------------------
WIDE_MODELS = [
    ('ks', True, dict(coefficient_grid_min_size=9)),            # G = 10, D = 3, 23 channels
    ('ks', False, dict(coefficient_grid_min_size=9)),           # G = 9 or 11 (centred)
    ('ks', True, dict(polynomial_accuracy_order=0)),            # 3 x 6 = 18 direct channels
    ('ks', False, dict(polynomial_accuracy_order=0, ensure_unbiased_coefficients=True)),
    ('burgers', True, dict(coefficient_grid_min_size=9)),       # G = 10, 17 channels, forced
    ('kdv', False, dict(coefficient_grid_min_size=11, nonlinearity='tanh', num_layers=4)),
]


@pytest.mark.parametrize('equation,conservative,overrides', WIDE_MODELS)
@pytest.mark.parametrize('num_points', [64, 32, 256, 100])
def test_wide_models_on_mfma(equation, conservative, overrides, num_points):
  model = make_model(equation, conservative, num_points=num_points,
                     resample_factor=2, **overrides)
  assert model.kernel_name.startswith('mfma_f32'), model.kernel_name
  batch = 5
  y0 = random_phase_ic(model.equation, batch)
  forcing = batch_forcing(batch, seed0=7)
  model.set_forcing(forcing)
  err = _check_all_views(model, y0, 0.6, forcing, None)
  print(equation, conservative, overrides, num_points, model.kernel_name, 'G',
        model.stencil_size, 'rel err {:.1e}'.format(err))


def test_wide_models_trajectories_all_launch_shapes():
  """Persistent, one launch per substep, float64 state and the adaptive
  controller on a wide model, against the oracle."""
  model = make_model('ks', True, num_points=64, resample_factor=2,
                     coefficient_grid_min_size=9)
  spec = model.spec()
  y0 = random_phase_ic(model.equation, 6)
  dt = 2.5e-5
  want = oracle.integrate_fixed(spec, oracle.SCHEME_MIDPOINT, 0.0, dt, 40, 20, y0)
  got = model.integrate_fixed(y0, 40, dt=dt, save_every=20).cpu().numpy()
  sub = model.integrate_fixed(y0, 40, dt=dt, save_every=20,
                              launch_mode='per_substep').cpu().numpy()
  f64 = model.integrate_fixed(y0, 40, dt=dt, save_every=20,
                              state_dtype='float64').cpu().numpy()
  assert rel_err(got, want) < TOL and rel_err(f64, want) < TOL
  np.testing.assert_array_equal(got, sub)
  times = np.linspace(0, 0.05, 3)
  y, nfev, status = model.integrate_adaptive(y0.astype(np.float64), times)
  for b in (0, 5):
    ref, ref_nfev = oracle.odeint_rk23(spec, y0[b], times)
    assert int(nfev[b]) == ref_nfev and rel_err(y[:, b].cpu().numpy(), ref) < TOL


def test_output_channel_groups_issued_in_pairs():
  """Run-time kernels issue only the live channel groups (two by two, a lone
  last one): 8 channels (KdV, unfolded), 9 (Burgers, pair + lone), 1 (the
  time-derivative head), 16 (Godunov KS, two pairs) -- all against the oracle."""
  cases = [('kdv', False, False, dict(nonlinearity='relu6')),
           ('burgers', True, False, dict(nonlinearity='relu6')),
           ('burgers', False, False, dict(model_target='time_derivative', nonlinearity='elu')),
           ('ks', True, True, dict(nonlinearity='relu6'))]
  for equation, conservative, flux, overrides in cases:
    model = make_model(equation, conservative, flux, num_points=64, **overrides)
    assert model.kernel_name == 'mfma_f32_r64'
    y0 = random_phase_ic(model.equation, 4)
    forcing = batch_forcing(4)
    model.set_forcing(forcing)
    _check_all_views(model, y0, 0.2, forcing, None)
    traj = model.integrate_fixed(y0, 10, dt=1e-5, save_every=10).cpu().numpy()
    want = oracle.integrate_fixed(model.spec(), oracle.SCHEME_MIDPOINT, 0.0, 1e-5, 10, 10, y0,
                                  forcing=forcing)
    assert rel_err(traj, want) < TOL, (equation, overrides)
