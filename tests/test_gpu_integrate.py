"""Time stepping on the GPU vs the oracle and vs reference-generated goldens."""
import numpy as np
import pytest

from helpers import (oracle, make_model, random_phase_ic, batch_forcing, rel_err)
import ddd1d_amd
from ddd1d_amd import equations, integrate, model as model_lib

pytestmark = pytest.mark.gpu

TOL = 1e-5   # north_star: float32 trajectories within 1e-5 rel of the reference

SCHEMES = {'euler': oracle.SCHEME_EULER, 'midpoint': oracle.SCHEME_MIDPOINT,
           'bs3': oracle.SCHEME_BS3, 'rk4': oracle.SCHEME_RK4}


@pytest.mark.parametrize('scheme', ['euler', 'midpoint', 'bs3', 'rk4'])
@pytest.mark.parametrize('kernel', ['mfma64', 'mfma64w32', 'mfma64w16', 'mfma256', 'generic'])
def test_fixed_step_schemes_vs_oracle(scheme, kernel):
  model = make_model('burgers', True, num_points=64, resample_factor=8)
  model.set_kernel(kernel)
  batch = 6
  forcing = batch_forcing(batch)
  model.set_forcing(forcing)
  y0 = random_phase_ic(model.equation, batch)
  dt = 1e-3 if scheme in ('euler', 'midpoint') else 1e-2
  got = model.integrate_fixed(y0, 40, dt=dt, t0=0.5, scheme=scheme,
                              save_every=10).cpu().numpy()
  want = oracle.integrate_fixed(model.spec(), SCHEMES[scheme], 0.5, dt, 40, 10,
                                y0, forcing=forcing)
  assert got.shape == want.shape == (4, batch, 64)
  err = rel_err(got, want)
  print(scheme, kernel, 'trajectory rel err {:.2e}'.format(err))
  assert err < TOL


@pytest.mark.parametrize('equation,conservative,dt,steps', [
    ('kdv', False, 2.5e-5, 60), ('kdv', True, 2.5e-5, 60),
    ('ks', False, 2.5e-5, 60), ('ks', True, 2.5e-5, 60),
    ('burgers', False, 1e-3, 60),
])
def test_midpoint_other_equations(equation, conservative, dt, steps):
  """model.integrate_ode semantics (model.py:138-159) with equation.time_step."""
  n = 256 if equation == 'ks' else 64
  model = make_model(equation, conservative, num_points=n, resample_factor=1)
  assert model.equation.time_step == dt
  y0 = random_phase_ic(model.equation, 3)
  got = model_lib.integrate_ode(model, y0, steps, dt).cpu().numpy()
  want = oracle.integrate_fixed(model.spec(), oracle.SCHEME_MIDPOINT, 0.0, dt,
                                steps, 1, y0)
  assert got.shape == (3, n, steps)            # [batch, x, time] like the reference
  err = rel_err(got, np.transpose(want, (1, 2, 0)))
  print(equation, conservative, 'rel err {:.2e}'.format(err))
  assert err < TOL


def test_launch_modes_agree():
  """One launch per substep (state in HBM) == persistent launch (state in VGPRs)."""
  model = make_model('burgers', True, num_points=64)
  forcing = batch_forcing(9)
  model.set_forcing(forcing)
  y0 = random_phase_ic(model.equation, 9)
  for scheme in ('midpoint', 'bs3', 'rk4', 'euler'):
    a = model.integrate_fixed(y0, 12, dt=1e-3, scheme=scheme, save_every=4,
                              launch_mode='persistent').cpu().numpy()
    b = model.integrate_fixed(y0, 12, dt=1e-3, scheme=scheme, save_every=4,
                              launch_mode='per_substep').cpu().numpy()
    np.testing.assert_array_equal(a, b)
    # all stages of a step in one launch (DDD_LAUNCH_PER_STEP)
    c = model.integrate_fixed(y0, 12, dt=1e-3, scheme=scheme, save_every=4,
                              launch_mode='per_step').cpu().numpy()
    np.testing.assert_array_equal(a, c)


@pytest.mark.parametrize('equation,num_points,batch', [
    ('burgers', 64, 5003),    # one-wave groups: the grid is capped at two per SIMD, so
    ('burgers', 32, 5003),    # every wavefront walks over 2-3 groups (ragged tail)
    ('kdv', 64, 4100),        # (>= 4096 groups: two half-ensembles on two streams)
    ('burgers', 64, 8200),
    ('burgers', 128, 1501),   # 256-row groups, two samples each, last group half empty
])
def test_launch_modes_agree_when_groups_outnumber_the_grid(equation, num_points, batch):
  """The per-substep kernel of the specialised models keeps a machine-sized
  grid and walks over the row groups (substep_multi_kernel: resident weights,
  next group's state / forcing prefetched): still bit-equal to the persistent
  launch, and the sub-sample that sits in second / third passes matches the
  oracle."""
  model = make_model(equation, True, num_points=num_points, resample_factor=2)
  forcing = batch_forcing(batch) if equation == 'burgers' else None
  if forcing is not None:
    model.set_forcing(forcing)
  y0 = random_phase_ic(model.equation, batch)
  dt = model.equation.time_step
  a = model.integrate_fixed(y0, 6, dt=dt, scheme='midpoint', save_every=3,
                            launch_mode='persistent').cpu().numpy()
  b = model.integrate_fixed(y0, 6, dt=dt, scheme='midpoint', save_every=3,
                            launch_mode='per_substep').cpu().numpy()
  np.testing.assert_array_equal(a, b)
  c = model.integrate_fixed(y0, 6, dt=dt, scheme='midpoint', save_every=3,
                            launch_mode='per_step').cpu().numpy()
  np.testing.assert_array_equal(a, c)
  assert np.isfinite(b).all()
  rows = np.array([0, batch // 2, batch - 2, batch - 1])     # first and later passes
  sub_forcing = None if forcing is None else {k: v[rows] for k, v in forcing.items()}
  want = oracle.integrate_fixed(model.spec(), oracle.SCHEME_MIDPOINT, 0.0, dt, 6, 3,
                                y0[rows], forcing=sub_forcing)
  assert rel_err(b[:, rows], want) < TOL


@pytest.mark.parametrize('equation,num_points,batch', [
    ('burgers', 48, 11),      # 256-row groups, 5 samples each, 16 spare rows
    ('burgers', 200, 3),      # one sample per group, 56 spare rows
    ('kdv', 100, 5),
    ('ks', 96, 4),
])
def test_trajectories_on_grids_that_are_not_powers_of_two(equation, num_points, batch):
  """The 256-row specialised integrators keep their operand-row offsets, patch
  indices and input addresses resident; for N that is not a power of two they
  come from the general wrap (tap_rows / wrap_row), spare rows read themselves.
  Persistent and per-substep launches agree bit for bit and match the oracle."""
  model = make_model(equation, True, num_points=num_points, resample_factor=2)
  assert model.kernel_name == 'mfma_f32_r256'
  forcing = batch_forcing(batch) if equation == 'burgers' else None
  if forcing is not None:
    model.set_forcing(forcing)
  y0 = random_phase_ic(model.equation, batch)
  dt = model.equation.time_step
  steps = 20 if equation == 'burgers' else 40
  a = model.integrate_fixed(y0, steps, dt=dt, scheme='midpoint', save_every=steps // 2,
                            launch_mode='persistent').cpu().numpy()
  b = model.integrate_fixed(y0, steps, dt=dt, scheme='midpoint', save_every=steps // 2,
                            launch_mode='per_substep').cpu().numpy()
  np.testing.assert_array_equal(a, b)
  c = model.integrate_fixed(y0, steps, dt=dt, scheme='bs3', save_every=steps // 2,
                            launch_mode='per_step').cpu().numpy()
  np.testing.assert_array_equal(c, model.integrate_fixed(
      y0, steps, dt=dt, scheme='bs3', save_every=steps // 2).cpu().numpy())
  want = oracle.integrate_fixed(model.spec(), oracle.SCHEME_MIDPOINT, 0.0, dt, steps,
                                steps // 2, y0, forcing=forcing)
  err = rel_err(a, want)
  print(equation, num_points, 'rel err {:.2e}'.format(err))
  assert err < TOL


def test_rk_substep_composes_midpoint():
  """ddd_rk_substep x2 == one midpoint step of ddd_integrate_fixed."""
  import torch
  model = make_model('kdv', False, num_points=64)
  y0 = torch.from_numpy(random_phase_ic(model.equation, 5)).cuda()
  dt = 2.5e-5
  ymid = torch.empty_like(y0)
  y1 = torch.empty_like(y0)
  model.rk_substep(0.0, y0, y_base=y0, c1=dt / 2, y_out=ymid)
  model.rk_substep(dt / 2, ymid, y_base=y0, c1=dt, y_out=y1)
  want = model.integrate_fixed(y0, 1, dt=dt, scheme='midpoint')[0]
  np.testing.assert_array_equal(y1.cpu().numpy(), want.cpu().numpy())
  # accumulator form
  acc = torch.empty_like(y0)
  model.rk_substep(dt / 2, ymid, acc_in=y0, c2=dt, acc_out=acc)
  np.testing.assert_array_equal(acc.cpu().numpy(), want.cpu().numpy())


@pytest.mark.parametrize('equation,num_points,batch', [
    ('burgers', 64, 4100),    # >= 4096 groups: two half-ensemble chains across the calls
    ('burgers', 64, 8200),
    ('kdv', 32, 8300),        # two samples per group
    ('burgers', 64, 300),     # too small to split: the bracket changes nothing
])
def test_external_rk_loop_inside_fork_join(equation, num_points, batch):
  """A caller that owns the RK loop (the shape of integrate.odeint, integrate.py:143-169):
  ddd_rk_substep twice per midpoint step inside ddd_stream_fork .. ddd_stream_join, written
  in Python and in C (examples/rk_driver.c).  Bit-identical to the unbracketed loop, to
  ddd_integrate_fixed in every launch mode, and a sub-sample against the oracle."""
  import ctypes
  import torch
  import bench
  driver = bench.load_rk_driver()
  model = make_model(equation, True, num_points=num_points, resample_factor=2)
  forcing = batch_forcing(batch) if equation == 'burgers' else None
  if forcing is not None:
    model.set_forcing(forcing)
  y0_host = random_phase_ic(model.equation, batch)
  y0 = torch.from_numpy(y0_host).cuda()
  dt = model.equation.time_step
  h = np.float32(dt)
  steps = 7

  def python_loop(chained):
    y, ystage, ynew = y0.clone(), torch.empty_like(y0), torch.empty_like(y0)
    ctx = model.chained_substeps() if chained else None
    if ctx is not None:
      ctx.__enter__()
    for step in range(steps):
      t = step * dt
      model.rk_substep(t, y, y_base=y, c1=0.5 * h, y_out=ystage)
      model.rk_substep(t + 0.5 * dt, ystage, acc_in=y, c2=h, acc_out=ynew)
      y, ynew = ynew, y
    if ctx is not None:
      ctx.__exit__(None, None, None)
    return y.cpu().numpy()   # (a copy on the caller's stream: ordered behind the join)

  def c_loop(chained):
    y, ystage, ynew = y0.clone(), torch.empty_like(y0), torch.empty_like(y0)
    final = ctypes.c_void_p()
    rc = driver.rk_driver_midpoint(model._handle, steps, 0.0, dt, y.data_ptr(),
                                   ystage.data_ptr(), ynew.data_ptr(), batch,
                                   ddd1d_amd._lib.current_stream(), int(chained),
                                   ctypes.byref(final))
    assert rc == 0
    return (y if final.value == y.data_ptr() else ynew).cpu().numpy()

  want = model.integrate_fixed(y0, steps, dt=dt, scheme='midpoint', save_every=steps,
                               launch_mode='persistent')[0].cpu().numpy()
  for got in (python_loop(False), python_loop(True), c_loop(True), c_loop(False)):
    np.testing.assert_array_equal(got, want)
  # the bracket survives entry points that join on their own, and an unmatched join
  lib = ddd1d_amd._lib.load_library()
  stream = ddd1d_amd._lib.current_stream()
  assert lib.ddd_stream_join(model._handle, stream) == 0
  assert lib.ddd_stream_fork(model._handle, stream) == 0
  ymid = torch.empty_like(y0)
  model.rk_substep(0.0, y0, y_base=y0, c1=0.5 * h, y_out=ymid)
  again = model.integrate_fixed(y0, steps, dt=dt, scheme='midpoint', save_every=steps,
                                launch_mode='per_substep')[0].cpu().numpy()   # joins first
  np.testing.assert_array_equal(again, want)
  ymid_plain = torch.empty_like(y0)   # the region is closed now: a plain launch
  model.rk_substep(0.0, y0, y_base=y0, c1=0.5 * h, y_out=ymid_plain)
  np.testing.assert_array_equal(ymid.cpu().numpy(), ymid_plain.cpu().numpy())
  rows = np.array([0, batch // 2, batch - 1])
  sub_forcing = None if forcing is None else {k: v[rows] for k, v in forcing.items()}
  ref = oracle.integrate_fixed(model.spec(), oracle.SCHEME_MIDPOINT, 0.0, dt, steps, steps,
                               y0_host[rows], forcing=sub_forcing)
  assert rel_err(want[rows], ref[0]) < TOL


@pytest.mark.parametrize('cls_name,n', [('KdVEquation', 64), ('ConservativeKdVEquation', 64),
                                        ('KSEquation', 256), ('ConservativeKSEquation', 100),
                                        ('ConservativeKdVEquation', 1024)])
def test_fixed_stencil_streaming_kernels_agree(cls_name, n):
  """Fixed stencils: the streaming kernel with one launch per substep, the same
  with ALL stages of a step in one launch (stage inputs stay in the block's LDS
  tile) and the persistent per-sample kernel: bit-identical, every scheme; a
  sub-sample against the oracle."""
  from helpers import baseline_spec
  eq = getattr(equations, cls_name)(n, random_seed=3)
  model = model_lib.BaselineModel(eq, accuracy_order=1)
  batch = 300 if n <= 256 else 9
  y0 = random_phase_ic(eq, batch)
  dt = eq.time_step
  for scheme in ('midpoint', 'bs3', 'rk4', 'euler'):
    a = model.integrate_fixed(y0, 9, dt=dt, scheme=scheme, save_every=3,
                              launch_mode='per_substep').cpu().numpy()
    assert model.kernel_name == 'stream_fixed'
    b = model.integrate_fixed(y0, 9, dt=dt, scheme=scheme, save_every=3,
                              launch_mode='per_step').cpu().numpy()
    assert model.kernel_name == 'stream_fixed'
    np.testing.assert_array_equal(a, b)
    if n <= 256:
      c = model.integrate_fixed(y0, 9, dt=dt, scheme=scheme, save_every=3,
                                launch_mode='persistent').cpu().numpy()
      np.testing.assert_array_equal(a, c)
    if 'KS' not in cls_name:   # (KS: bit-identity with the persistent kernel above, whose own
      #                            oracle parity is test_integrate_baseline_vs_reference_golden)
      want = oracle.integrate_fixed(baseline_spec(eq, 1), SCHEMES[scheme], 0.0, dt, 9, 3, y0[:3])
      assert rel_err(b[:, :3], want) < TOL, (scheme, cls_name)


@pytest.mark.parametrize('equation,conservative,num_points', [
    ('burgers', True, 64), ('burgers', False, 32), ('kdv', True, 64), ('ks', False, 16)])
def test_small_ensembles_run_two_wavefronts_per_sample(equation, conservative, num_points):
  """Small ensembles spread each 64-row group over several wavefronts so that the SIMDs do
  not idle (the reference's callers integrate tens to hundreds of samples,
  scripts/run_evaluation.py:212-221): FOUR 16-row wavefronts with every layer on 16x16x4
  MFMAs (rhs_mfma.h kQuad; the automatic choice while that leaves at most two wavefronts per
  SIMD) or TWO 32-row wavefronts with the output layer's channel groups divided between
  them (kSplit).  Every accumulation chain keeps its order, so a sample's bits do not depend
  on the ensemble around it: equal to the one-wavefront kernel (large ensemble, and the
  forced mfma64 geometry) and to one launch per substep."""
  model = make_model(equation, conservative, num_points=num_points, resample_factor=2)
  big, small = 1500 * (64 // num_points), 37
  forcing = batch_forcing(big) if equation == 'burgers' else None
  y0 = random_phase_ic(model.equation, big)
  dt = model.equation.time_step
  if forcing is not None:
    model.set_forcing(forcing)
  a = model.integrate_fixed(y0, 12, dt=dt, scheme='bs3', save_every=6).cpu().numpy()
  assert model.kernel_name == 'mfma_f32_r64'
  if forcing is not None:
    model.set_forcing({k: v[:small] for k, v in forcing.items()})
  b = model.integrate_fixed(y0[:small], 12, dt=dt, scheme='bs3', save_every=6).cpu().numpy()
  assert model.kernel_name == 'mfma_f32_r64w16'
  np.testing.assert_array_equal(a[:, :small], b)
  c = model.integrate_fixed(y0[:small], 12, dt=dt, scheme='bs3', save_every=6,
                            launch_mode='per_substep').cpu().numpy()
  np.testing.assert_array_equal(b, c)
  for kernel, name in (('mfma64', 'mfma_f32_r64'), ('mfma64w32', 'mfma_f32_r64w32'),
                       ('mfma64w16', 'mfma_f32_r64w16')):
    model.set_kernel(kernel)
    d = model.integrate_fixed(y0[:small], 12, dt=dt, scheme='bs3', save_every=6).cpu().numpy()
    assert model.kernel_name == name
    np.testing.assert_array_equal(b, d)
  # float64 state has no such kernels: the one-wavefront geometry, silently
  e = model.integrate_fixed(y0[:small], 12, dt=dt, scheme='bs3', save_every=6,
                            state_dtype='float64').cpu().numpy()
  assert model.kernel_name == 'mfma_f32_r64' and rel_err(e, b) < 1e-6


def test_four_wavefronts_per_sample_long_run_and_all_equations():
  """The kQuad integrators over a longer horizon, all six per-equation kernels, forced and
  unforced, N = 64 / 32 / 16 / 8 (8, 4, 2, 1 samples per group): bit-equal to the
  one-wavefront kernel, snapshots included."""
  for equation, conservative, n in (('burgers', True, 64), ('burgers', False, 64), ('kdv', False, 32),
                                    ('kdv', True, 8), ('ks', False, 64), ('ks', True, 16)):
    model = make_model(equation, conservative, num_points=n, resample_factor=2)
    batch = 19
    forcing = batch_forcing(batch, seed0=7) if equation == 'burgers' else None
    model.set_forcing(forcing)
    y0 = random_phase_ic(model.equation, batch)
    dt = model.equation.time_step
    out = {}
    for kernel in ('mfma64', 'mfma64w16'):
      model.set_kernel(kernel)
      out[kernel] = model.integrate_fixed(y0, 100, dt=dt, scheme='midpoint', save_every=25).cpu().numpy()
    assert model.kernel_name == 'mfma_f32_r64w16'
    assert np.isfinite(out['mfma64']).all()
    np.testing.assert_array_equal(out['mfma64'], out['mfma64w16'])
    want = oracle.integrate_fixed(model.spec(), oracle.SCHEME_MIDPOINT, 0.0, dt, 100, 25, y0,
                                  forcing=forcing)
    assert rel_err(out['mfma64w16'][:2], want[:2]) < (TOL if equation != 'ks' else 1e-3)
  # nets without such kernels refuse the explicit choice
  other = make_model('burgers', True, num_points=64, num_layers=4)
  with pytest.raises(Exception, match='four 16-row wavefronts'):
    other.set_kernel('mfma64w16')


def test_float64_state():
  model = make_model('burgers', False, num_points=64)
  forcing = batch_forcing(4)
  model.set_forcing(forcing)
  y0 = random_phase_ic(model.equation, 4).astype(np.float64)
  got = model.integrate_fixed(y0, 30, dt=1e-2, scheme='bs3', save_every=30,
                              state_dtype='float64').cpu().numpy()
  want = oracle.integrate_fixed(model.spec(), oracle.SCHEME_BS3, 0.0, 1e-2, 30,
                                30, y0, forcing=forcing, state_dtype=np.float64)
  assert got.dtype == np.float64
  assert rel_err(got, want) < TOL


@pytest.mark.parametrize('equation,conservative,n', [('ks', True, 256), ('burgers', True, 128),
                                                     ('kdv', False, 96)])
def test_float64_state_four_wave_groups(equation, conservative, n):
  """float64 state on 256-row groups: per-equation integrators since round 6 (they ran on
  the run-time-parameterised kernel).  Against the oracle with a float64 state, against the
  forced run-time ... and the float32-state run (same right-hand side: float32 rounding of
  the state apart)."""
  model = make_model(equation, conservative, num_points=n, resample_factor=1)
  batch = 5
  forcing = batch_forcing(batch) if equation == 'burgers' else None
  model.set_forcing(forcing)
  y0 = random_phase_ic(model.equation, batch)
  dt = model.equation.time_step
  got = model.integrate_fixed(y0.astype(np.float64), 20, dt=dt, scheme='bs3', save_every=10,
                              state_dtype='float64').cpu().numpy()
  assert model.kernel_name == 'mfma_f32_r256' and got.dtype == np.float64
  want = oracle.integrate_fixed(model.spec(), oracle.SCHEME_BS3, 0.0, dt, 20, 10, y0,
                                forcing=forcing, state_dtype=np.float64)
  assert rel_err(got, want) < (TOL if equation != 'ks' else 1e-4)
  f32 = model.integrate_fixed(y0, 20, dt=dt, scheme='bs3', save_every=10).cpu().numpy()
  assert rel_err(got, f32) < 1e-5


def test_save_every_and_zero_steps():
  model = make_model('burgers', True, num_points=64)
  y0 = random_phase_ic(model.equation, 3)
  full = model.integrate_fixed(y0, 12, dt=1e-3, save_every=1).cpu().numpy()
  sparse = model.integrate_fixed(y0, 12, dt=1e-3, save_every=5).cpu().numpy()
  assert sparse.shape[0] == 2
  np.testing.assert_array_equal(sparse[0], full[4])
  np.testing.assert_array_equal(sparse[1], full[9])
  none = model.integrate_fixed(y0, 0, dt=1e-3)
  assert tuple(none.shape) == (0, 3, 64)


def test_scipy_rk23_with_hip_differentiator_matches_oracle():
  """integrate.odeint (SciPy RK23, float64 state) driving the HIP RHS, B = 1:
  the reference's production execution shape (integrate.py:143-169)."""
  model = make_model('burgers', True, num_points=32, resample_factor=16, seed=4)
  eq = model.equation
  diff = integrate.SavedModelDifferentiator(None, eq, model=model)
  times = np.linspace(0, 0.5, 6)
  y0 = eq.initial_value()
  got, nfev = integrate.odeint(y0, diff, times)
  forcing = model_lib.forcing_from_equations([eq])
  want, nfev_want = oracle.odeint_rk23(model.spec(), y0, times,
                                       {k: v[0] for k, v in forcing.items()})
  assert nfev == nfev_want
  assert rel_err(got, want) < TOL


GOLDEN_ODEINT = [
    ('BurgersEquation', 32, 1, 0, 1), ('ConservativeBurgersEquation', 64, 4, 2, 1),
    ('KdVEquation', 64, 1, 1, 1), ('ConservativeKdVEquation', 64, 4, 5, 1),
    ('KSEquation', 64, 1, 4, 1), ('ConservativeKSEquation', 64, 2, 7, 1),
    ('BurgersEquation', 32, 1, 9, 3),
]


@pytest.mark.parametrize('cls_name,n,rf,seed,acc', GOLDEN_ODEINT)
def test_integrate_baseline_vs_reference_golden(golden, cls_name, n, rf, seed, acc):
  """integrate_baseline through the HIP kernel against trajectories produced by
  the reference's own odeint + equation_of_motion + finalize (float64 RHS).
  Config 1 of BASELINE.json is the first case (Burgers N=32, 100 RK23 steps)."""
  key = 'odeint/{}/n{}/rf{}/s{}/a{}'.format(cls_name, n, rf, seed, acc)
  eq = getattr(equations, cls_name)(n, resample_factor=rf, random_seed=seed)
  times = golden[key + '/times']
  ds = integrate.integrate_baseline(eq, times=times, accuracy_order=acc)
  want = golden[key + '/y']
  got = np.asarray(ds['y'])
  assert got.shape == want.shape
  # the reference evaluates the RHS in float32 inside TF; the golden RHS is
  # float64.  Bound: 1e-5, or 4 x the distance of the float32 NumPy restatement
  # (same SciPy run) from the golden trajectory where that float32 noise is larger
  from helpers import baseline_spec
  spec = baseline_spec(eq, acc)
  frc = ({k: v[0] for k, v in model_lib.forcing_from_equations([eq]).items()}
         if eq.has_time_dependent_forcing else None)
  f32_run, _ = oracle.odeint_rk23(spec, eq.initial_value(), times, frc)
  floor = rel_err(f32_run, want)
  err = rel_err(got, want)
  print(key, 'HIP vs reference {:.1e}; float32 restatement vs reference {:.1e}'
        .format(err, floor))
  assert err < max(TOL, 4 * floor)
  assert int(ds.coords['num_evals']) == int(golden[key + '/nfev'])
  # single RHS evaluation as well
  diff = integrate.PolynomialDifferentiator(eq, acc)
  y_probe = eq.initial_value() + 0.1 * np.sin(eq.grid.solution_x)
  rhs = diff(0.1, y_probe)
  # the golden RHS is the reference's float64 evaluation, the kernel's is float32:
  # bound = max(1e-5, 4 x the float32 oracle's own distance from the golden) -- KS's
  # 4th-derivative stencils (|c| ~ 6/dx^4) cancel ~1e4-fold on smooth data
  want_rhs = golden[key + '/rhs_t0.1_y0']
  f32_rhs = oracle.time_derivative(spec, 0.1, y_probe[None], None if frc is None else
                                   {k: v[None] for k, v in frc.items()})[0]
  from helpers import measured_bound
  assert rel_err(rhs, want_rhs) < measured_bound(f32_rhs, want_rhs, TOL, key + ' rhs:', got=rhs)


def test_integrate_batch_matches_per_sample_scipy():
  """integrate_batch(adaptive=True) = the reference's per-sample RK23 runs (equal
  nfev, 1e-5).  The fixed-step BS3 form at dt = max_step is a DIFFERENT step
  sequence (no small first steps): close while the controller sits at max_step
  (smooth Burgers), asserted at 2e-4 as a statement about the two algorithms,
  not as a parity tolerance."""
  hp_model = make_model('burgers', True, num_points=32, resample_factor=16)
  batch = 3
  times = np.linspace(0, 0.2, 3)
  forcing = batch_forcing(batch, seed0=11)
  y0 = random_phase_ic(hp_model.equation, batch)
  ds = integrate.integrate_batch(hp_model, y0, times, dt=0.01, scheme='bs3',
                                 forcing=forcing, state_dtype='float64')
  y = np.asarray(ds['y'])
  assert y.shape == (batch, 3, 32)
  ada = integrate.integrate_batch(hp_model, y0, times, dt=0.01, forcing=forcing,
                                  adaptive=True)
  y_ada = np.asarray(ada['y'])
  nfev = np.asarray(integrate._dataset_coord(ada, 'num_evals'))
  for b in range(batch):
    one = {k: v[b] for k, v in forcing.items()}
    want, want_nfev = oracle.odeint_rk23(hp_model.spec(), y0[b], times, one)
    assert nfev[b] == want_nfev and rel_err(y_ada[b], want) < TOL
    assert rel_err(y[b], want) < 2e-4


def test_mean_conservation_long_run():
  """integrate_test.py:101-104: conservative forms keep the spatial mean."""
  model = make_model('kdv', True, num_points=64, resample_factor=1)
  y0 = random_phase_ic(model.equation, 64)
  out = model.integrate_fixed(y0, 400, dt=2.5e-5, save_every=400).cpu().numpy()
  drift = np.abs(out[0].mean(axis=1) - y0.mean(axis=1)).max()
  assert np.isfinite(out).all()
  assert drift < 1e-5


def test_odeint_on_the_device_equals_scipy_on_the_host(monkeypatch):
  """integrate.odeint over a HIP differentiator runs SciPy's RK23 on the device
  (one launch per solve); with DEVICE_ODEINT = False it is SciPy on the host
  calling the differentiator once per evaluation (the reference's shape).  Same
  evaluation count, same trajectory, for every differentiator kind."""
  cases = []
  model = make_model('burgers', True, num_points=64)
  cases.append((integrate.SavedModelDifferentiator(None, model.equation, model=model),
                model.equation.initial_value() + 0.3 * np.sin(model.equation.grid.solution_x)))
  eq = equations.ConservativeKdVEquation(64, resample_factor=4, random_seed=2)
  cases.append((integrate.PolynomialDifferentiator(eq, 1), eq.initial_value()))
  eq = equations.GodunovBurgersEquation(96, random_seed=4)
  cases.append((integrate.WENODifferentiator(eq), 0.5 * np.sin(eq.grid.solution_x)))
  eq = equations.KdVEquation(64, random_seed=6)
  cases.append((integrate.SpectralDifferentiator(eq), eq.initial_value()))
  times = np.array([0.0, 0.013, 0.1, 0.2])
  for diff, y0 in cases:
    monkeypatch.setattr(integrate, 'DEVICE_ODEINT', True)
    dev, dev_nfev = integrate.odeint(y0, diff, times)
    monkeypatch.setattr(integrate, 'DEVICE_ODEINT', False)
    host, host_nfev = integrate.odeint(y0, diff, times)
    assert dev_nfev == host_nfev, (type(diff).__name__, dev_nfev, host_nfev)
    assert dev.shape == host.shape and rel_err(dev, host) < 1e-9, type(diff).__name__
  # forced spectral Burgers keeps the host loop (its forcing is applied on the host)
  eq = equations.BurgersEquation(64, random_seed=1)
  assert integrate.SpectralDifferentiator(eq).device_model is None
