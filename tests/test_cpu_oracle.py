"""Pin the CPU oracle: reference goldens, reference known-answer tables, torch.

Runs without a GPU.  What cannot be pinned here (TF graph ops) is listed in
oracle/oracle.py's header.
"""
import numpy as np
import pytest

from helpers import oracle, baseline_spec, rel_err
from ddd1d_amd import equations, model as model_lib

EQ_IDS = {
    'BurgersEquation': oracle.EQ_BURGERS,
    'ConservativeBurgersEquation': oracle.EQ_BURGERS_CONSERVATIVE,
    'GodunovBurgersEquation': oracle.EQ_BURGERS_GODUNOV,
    'KdVEquation': oracle.EQ_KDV,
    'ConservativeKdVEquation': oracle.EQ_KDV_CONSERVATIVE,
    'GodunovKdVEquation': oracle.EQ_KDV_GODUNOV,
    'KSEquation': oracle.EQ_KS,
    'ConservativeKSEquation': oracle.EQ_KS_CONSERVATIVE,
    'GodunovKSEquation': oracle.EQ_KS_GODUNOV,
}


def _build(key):
  _, cls_name, n, rf, seed = key.split('/')
  cls = getattr(equations, cls_name)
  return cls_name, cls(int(n[1:]), resample_factor=int(rf[2:]),
                       random_seed=int(seed[1:]))


def test_kernel_ids_match_oracle_ids():
  for name, ident in EQ_IDS.items():
    assert getattr(equations, name).KERNEL_ID == ident


def test_equation_of_motion_vs_reference(golden):
  for key in golden.index['equations']:
    cls_name, eq = _build(key)
    y = golden[key + '/eom_y']
    derivs = golden[key + '/eom_derivs']
    got = oracle.equation_of_motion(EQ_IDS[cls_name], y, derivs,
                                    getattr(eq, 'eta', 0.0), eq.grid.solution_dx)
    np.testing.assert_allclose(got, golden[key + '/eom_out'], rtol=1e-13,
                               atol=1e-13)
    got32 = oracle.equation_of_motion(
        EQ_IDS[cls_name], y.astype(np.float32), derivs.astype(np.float32),
        getattr(eq, 'eta', 0.0), eq.grid.solution_dx)
    assert got32.dtype == np.float32
    assert rel_err(got32, golden[key + '/eom_out']) < 2e-6


def test_forcing_vs_reference(golden):
  for key in golden.index['equations']:
    _, eq = _build(key)
    forcing = {k: v[0] for k, v in
               model_lib.forcing_from_equations([eq]).items()}
    args = (eq.grid.solution_num_points, eq.grid.resample_factor,
            eq.grid.period, bool(eq.CONSERVATIVE))
    for t, want in zip(golden[key + '/forcing_t'],
                       golden[key + '/forcing_values']):
      got = oracle.forcing_f64(t, forcing, *args)
      np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-13)
      got32 = oracle.forcing_f32(t, forcing, *args)
      assert got32.dtype == np.float32
      # float32 phases of magnitude <= 50 round to ~4e-6
      assert np.abs(got32 - want).max() < 3e-5


def test_forcing_kernel_tables_reproduce_reference(golden):
  """The amplitude/phase folding handed to the GPU (block-mean -> one sine)
  reproduces RandomForcing.__call__ to float64 accuracy."""
  for key in golden.index['equations']:
    _, eq = _build(key)
    forcing = model_lib.forcing_from_equations([eq])
    tab = model_lib.forcing_kernel_tables(forcing, eq.grid)
    # evaluate the kernel's formula in float64 from the float32 tables' sources
    a = np.asarray(forcing['a'], float)
    k = forcing['k']
    rf, n = eq.grid.resample_factor, eq.grid.solution_num_points
    for t, want in zip(golden[key + '/forcing_t'],
                       golden[key + '/forcing_values']):
      sp = tab['spatial_phase'].astype(np.float64)[tab['k_index'][0]]   # [P, N]
      val = np.sum(tab['amplitude'][0].astype(np.float64)[:, None] * np.sin(
          tab['omega'][0].astype(np.float64)[:, None] * t + sp
          + tab['phase'][0].astype(np.float64)[:, None]), axis=0)
      assert np.abs(val - want).max() < 2e-5   # float32 table rounding only


ODEINT_CASES = [
    ('BurgersEquation', 32, 1, 0, 1), ('ConservativeBurgersEquation', 64, 4, 2, 1),
    ('KdVEquation', 64, 1, 1, 1), ('ConservativeKdVEquation', 64, 4, 5, 1),
    ('KSEquation', 64, 1, 4, 1), ('ConservativeKSEquation', 64, 2, 7, 1),
    ('BurgersEquation', 32, 1, 9, 3),
]


@pytest.mark.parametrize('cls_name,n,rf,seed,acc', ODEINT_CASES)
def test_oracle_odeint_vs_reference_trajectory(golden, cls_name, n, rf, seed, acc):
  """oracle.odeint_rk23 over the fixed-stencil RHS vs integrate.odeint run by
  the reference (float64 RHS there, float32 RHS here)."""
  key = 'odeint/{}/n{}/rf{}/s{}/a{}'.format(cls_name, n, rf, seed, acc)
  eq = getattr(equations, cls_name)(n, resample_factor=rf, random_seed=seed)
  spec = baseline_spec(eq, acc)
  forcing = {k: v[0] for k, v in model_lib.forcing_from_equations([eq]).items()}
  y, nfev = oracle.odeint_rk23(spec, golden[key + '/y0'], golden[key + '/times'],
                               forcing)
  assert nfev == int(golden[key + '/nfev'])
  assert rel_err(y, golden[key + '/y']) < 5e-5
  y_probe = eq.initial_value() + 0.1 * np.sin(eq.grid.solution_x)
  rhs = oracle.time_derivative(spec, 0.1, y_probe[None], {
      k: v[None] for k, v in forcing.items()})[0]
  tol = 2e-3 if 'KS' in cls_name else 2e-5
  assert rel_err(rhs, golden[key + '/rhs_t0.1_y0']) < tol


@pytest.mark.parametrize('padding,center,expected', [
    (0, True, [0, 1, 2]), (1, True, [2, 0, 1, 2]), (2, True, [2, 0, 1, 2, 0]),
    (3, True, [1, 2, 0, 1, 2, 0]), (4, True, [1, 2, 0, 1, 2, 0, 1]),
    (6, True, [0, 1, 2, 0, 1, 2, 0, 1, 2]),
    (7, True, [2, 0, 1, 2, 0, 1, 2, 0, 1, 2]),
    (0, False, [0, 1, 2]), (1, False, [0, 1, 2, 0]), (2, False, [0, 1, 2, 0, 1]),
    (3, False, [0, 1, 2, 0, 1, 2]), (5, False, [0, 1, 2, 0, 1, 2, 0, 1]),
])
def test_pad_periodic_table(padding, center, expected):
  """layers_test.py:49-67."""
  got = oracle.pad_periodic(np.arange(3)[None, :, None], padding, center)
  np.testing.assert_array_equal(got[0, :, 0], expected)


def test_nn_conv1d_periodic_table():
  """layers_test.py:69-86."""
  inputs = np.arange(5.0)[None, :, None]
  for filt, expected in [([0.0, 1.0, 0.0], np.arange(5.0)),
                         ([0.0, 1.0], np.arange(5.0)),
                         ([0.5, 0.5], [2.0, 0.5, 1.5, 2.5, 3.5])]:
    got = oracle.nn_conv1d_periodic(
        inputs, np.asarray(filt)[:, None, None], center=True)
    np.testing.assert_allclose(got[0, :, 0], expected)


@pytest.mark.parametrize('k', [2, 3, 4, 5])
def test_conv_vs_torch_circular(k):
  """Independent cross-check of the multi-channel periodic convolution."""
  import torch
  import torch.nn.functional as F
  rs = np.random.RandomState(k)
  x = rs.randn(3, 20, 6).astype(np.float32)
  w = rs.randn(k, 6, 4).astype(np.float32)
  b = rs.randn(4).astype(np.float32)
  got = oracle.conv1d_periodic_layer(x, w, b, None, center=True)
  left = k // 2                      # ceil((k-1)/2)
  right = (k - 1) - left
  xt = torch.from_numpy(x).permute(0, 2, 1)                      # [B, C, N]
  xt = torch.cat([xt[..., xt.shape[-1] - left:], xt, xt[..., :right]], dim=-1)
  wt = torch.from_numpy(w).permute(2, 1, 0).contiguous()         # [Cout, Cin, K]
  want = F.conv1d(xt, wt, torch.from_numpy(b)).permute(0, 2, 1).numpy()
  np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize('size', [3, 6, 7])
def test_extract_patches_alignment(size):
  """patches[x, i] = u[x + i - ceil((size-1)/2)]: centred stencils see offsets
  -3..3 (size 7), staggered ones -3..2 (size 6)."""
  u = np.arange(10.0)[None]
  p = oracle.extract_patches(u, size)
  left = size // 2
  for i in range(size):
    np.testing.assert_array_equal(p[0, :, i], np.roll(u[0], -(i - left)))


def test_midpoint_is_second_order_on_linear_problem():
  """Sanity of the restated fixed-step schemes: observed order of accuracy on
  unforced Burgers against a 64-step reference."""
  eq = equations.BurgersEquation(32)
  spec = baseline_spec(eq, 1)
  y0 = (0.5 * np.sin(eq.grid.solution_x))[None]
  errs = []
  for scheme, order in ((oracle.SCHEME_EULER, 1), (oracle.SCHEME_MIDPOINT, 2),
                        (oracle.SCHEME_BS3, 3), (oracle.SCHEME_RK4, 4)):
    def run(steps):
      return oracle.integrate_fixed(spec, scheme, 0.0, 0.4 / steps, steps,
                                    steps, y0, state_dtype=np.float64,
                                    apply_forcing=False)[0]
    ref = run(64)
    e1 = np.abs(run(2) - ref).max()
    e2 = np.abs(run(4) - ref).max()
    errs.append((order, e1 / e2))
  # halving the step divides the error by ~2^order (float32 RHS noise limits
  # the high-order ones, so only require monotone improvement there)
  assert errs[0][1] > 1.7 and errs[1][1] > 3.0


# ---------------------------------------------------------------------------
# The adaptive controller the device runs (rhs_adaptive.h) = oracle.rk23_adaptive,
# pinned against the installed SciPy (the reference's integrator, integrate.py:154)
# ---------------------------------------------------------------------------
def _toy_rhs(n, seed, stiffness):
  """Contracting (so that one-ulp differences do not amplify), stiff (so that the
  controller is stability-limited and rejects) float32 right-hand side."""
  rs = np.random.RandomState(seed)
  q, _ = np.linalg.qr(rs.randn(n, n))
  lam = np.linspace(1.0, stiffness, n)
  a = ((q * lam) @ q.T).astype(np.float32)

  def fun(t, y):
    y32 = np.asarray(y, np.float32)
    return (-(a @ y32) - np.float32(0.5) * y32 ** 3 + np.float32(np.sin(7 * t))).astype(np.float32)
  return fun


@pytest.mark.parametrize('stiffness,max_step', [(3.0, 0.01), (3.0, np.inf), (300.0, np.inf),
                                               (3000.0, 0.01)])
def test_rk23_restatement_equals_scipy(stiffness, max_step):
  """Saturated steps, controller-limited steps, rejections: identical nfev and
  float64-rounding-equal dense output at times that are not step boundaries."""
  import scipy.integrate
  times = np.array([0.0, 0.0371, 0.2, 0.55, 1.0])
  rejected_somewhere = False
  for seed in range(4):
    fun = _toy_rhs(16, seed, stiffness)
    y0 = np.random.RandomState(seed + 10).randn(16).astype(np.float32).astype(np.float64)
    sol = scipy.integrate.solve_ivp(fun, (times[0], times[-1]), y0, t_eval=times,
                                    max_step=max_step, method='RK23')
    y, nfev, status = oracle.rk23_adaptive(fun, y0, times, max_step=max_step)
    assert status == 0 and sol.status == 0
    assert nfev == sol.nfev
    # (not bitwise: BLAS sums the error norm in another order, and one ulp of a
    # step size moves a float32 rounding of the right-hand side's input)
    np.testing.assert_allclose(y, sol.y.T, rtol=1e-9, atol=1e-10)
    rejected_somewhere |= stiffness < 100 or (nfev - 2) // 3 > 150
  assert rejected_somewhere   # the stiff cases are limited by stability, not by max_step


def test_rk23_restatement_failure_path():
  """Finite-time blow-up: SciPy gives up when the step drops under 10 ulp(t)
  (status -1); the rows not reached are NaN (integrate.py:161-167)."""
  import scipy.integrate

  def fun(t, y):
    y32 = np.asarray(y, np.float32)
    return (y32 * y32).astype(np.float32)   # y' = y^2 blows up at t = 1 / y0
  times = np.linspace(0.0, 1.0, 6)
  y0 = np.array([2.5, 3.0])
  sol = scipy.integrate.solve_ivp(fun, (0.0, 1.0), y0, t_eval=times, max_step=0.01,
                                  method='RK23')
  y, nfev, status = oracle.rk23_adaptive(fun, y0, times)
  assert sol.status == -1 and status == -1
  assert nfev == sol.nfev
  reached = sol.y.shape[1]
  assert 0 < reached < len(times)
  np.testing.assert_allclose(y[:reached], sol.y.T, rtol=1e-9)
  assert np.isnan(y[reached:]).all()
