"""Nets of up to 16 filters on 16-channel tiles (csrc/rhs_mfma.h Tile16Tower -- or, built with
-DDDD_HALF_T16=0, the block-diagonal HalfTower --, round 6; training.py:134-136 leaves
filter_size free, model.py:455-458 builds whatever it says).  Same bits as the zero-padded
embedding in 32 filters (the launch modes and geometries that still use it), oracle parity at
1e-5, NaN mask of the reference."""
import numpy as np
import pytest

from helpers import batch_forcing, make_model, oracle, random_phase_ic, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


@pytest.mark.parametrize('equation,conservative,num_points,overrides', [
    ('burgers', True, 64, dict(filter_size=16)),
    ('burgers', False, 64, dict(filter_size=12, kernel_size=3)),
    ('kdv', True, 64, dict(filter_size=16)),
    ('kdv', False, 64, dict(filter_size=8)),
    ('ks', True, 64, dict(filter_size=16, kernel_size=4)),
    ('ks', False, 64, dict(filter_size=5)),
    ('burgers', True, 32, dict(filter_size=16)),      # two samples per 64-row group
    ('kdv', True, 16, dict(filter_size=10)),          # four
    ('ks', False, 32, dict(filter_size=16, kernel_size=3)),
])
def test_block_diagonal_tower(equation, conservative, num_points, overrides):
  import torch
  model = make_model(equation, conservative, num_points=num_points, resample_factor=2, **overrides)
  # every SIMD holds 64-row wavefronts (small ensembles take the four-wave kernels); the last
  # group ragged
  batch = 2100 * (64 // num_points) + 1
  forcing = batch_forcing(batch) if equation == 'burgers' else None
  if forcing is not None:
    model.set_forcing(forcing)
  y0_host = random_phase_ic(model.equation, batch)
  y0 = torch.from_numpy(y0_host).cuda()
  dt = model.equation.time_step
  steps = 12
  got = model.integrate_fixed(y0, steps, dt=dt, scheme='midpoint', save_every=steps)[0].cpu().numpy()
  assert model.kernel_name == 'mfma_f32_r64h16', model.kernel_name
  # one launch per step / per substep: the same tiles, the same bits ...
  for mode in ('per_step', 'per_substep'):
    other = model.integrate_fixed(y0, steps, dt=dt, scheme='midpoint', save_every=steps,
                                  launch_mode=mode)[0].cpu().numpy()
    assert model.kernel_name == 'mfma_f32_r64h16', (mode, model.kernel_name)
    np.testing.assert_array_equal(got, other)
  rk4 = model.integrate_fixed(y0, 5, dt=dt, scheme='rk4', save_every=5)[0].cpu().numpy()
  # ... and the four-wave geometry, which still embeds the net in 32 filters
  model.set_kernel('mfma256')
  for mode in ('persistent', 'per_step'):
    other = model.integrate_fixed(y0, steps, dt=dt, scheme='midpoint', save_every=steps,
                                  launch_mode=mode)[0].cpu().numpy()
    assert model.kernel_name == 'mfma_f32_r256', (mode, model.kernel_name)
    np.testing.assert_array_equal(got, other)
  rk4_ref = model.integrate_fixed(y0, 5, dt=dt, scheme='rk4', save_every=5)[0].cpu().numpy()
  model.set_kernel('auto')
  np.testing.assert_array_equal(rk4, rk4_ref)
  rows = np.array([0, 1, batch // 2, batch - 1])
  sub = None if forcing is None else {k: v[rows] for k, v in forcing.items()}
  ref = oracle.integrate_fixed(model.spec(), oracle.SCHEME_MIDPOINT, 0.0, dt, steps, steps,
                               y0_host[rows], forcing=sub)
  assert rel_err(got[rows], ref[0]) < TOL
  # float64 state and the production integrator (adaptive RK23, integrate.py:143-169) as well;
  # the four-wave geometry still embeds the net in 32 filters: the same bits, the same nfev
  y64 = y0.double()
  f64 = model.integrate_fixed(y64, 6, dt=dt, scheme='bs3', save_every=6, state_dtype='float64')[0]
  assert model.kernel_name == 'mfma_f32_r64h16', model.kernel_name
  span = 12 * dt if equation != 'burgers' else 0.03
  times = np.linspace(0.0, span, 4)
  sub = 700 * (64 // num_points)
  traj, nfev, status = model.integrate_adaptive(y64[:sub], times)
  assert model.kernel_name == 'mfma_f32_r64h16', model.kernel_name
  model.set_kernel('mfma256')
  f64_ref = model.integrate_fixed(y64, 6, dt=dt, scheme='bs3', save_every=6, state_dtype='float64')[0]
  assert model.kernel_name == 'mfma_f32_r256', model.kernel_name
  traj_ref, nfev_ref, status_ref = model.integrate_adaptive(y64[:sub], times)
  model.set_kernel('auto')
  assert torch.equal(f64, f64_ref)
  assert torch.equal(nfev, nfev_ref) and torch.equal(status, status_ref) and int(status.abs().max()) == 0
  if num_points == 64:
    assert torch.equal(traj, traj_ref)
  else:
    # several samples per group: the four-wave geometry sums the error norm's N terms in another
    # order than the one-wave butterfly (rhs_adaptive.h: sample_sum) -- step sizes differ in
    # their last bits, the evaluations do not (the fixed-step comparison above is exact)
    assert torch.allclose(traj, traj_ref, rtol=1e-9, atol=1e-11)
  one = None if forcing is None else {k: v[3] for k, v in forcing.items()}
  ref_traj, ref_nfev = oracle.odeint_rk23(model.spec(), y0_host[3], times, one)
  assert int(nfev[3]) == ref_nfev
  assert rel_err(traj[:, 3].cpu().numpy(), ref_traj) < TOL


def test_block_diagonal_tower_nan_mask():
  """A NaN in one sample: the rows the reference's relu would poison (integrate.py:161-167
  signals divergence by NaN rows), nothing else -- as on the embedded route."""
  import torch
  model = make_model('burgers', True, num_points=64, resample_factor=2, filter_size=16)
  batch = 2100
  model.set_forcing(batch_forcing(batch))
  y0_host = random_phase_ic(model.equation, batch)
  y0_host[7, 20] = np.nan
  y0 = torch.from_numpy(y0_host).cuda()
  dt = model.equation.time_step
  got = model.integrate_fixed(y0, 1, dt=dt, scheme='euler', save_every=1)[0].cpu().numpy()
  assert model.kernel_name == 'mfma_f32_r64h16'
  model.set_kernel('mfma256')   # (embedded in 32 filters)
  other = model.integrate_fixed(y0, 1, dt=dt, scheme='euler', save_every=1)[0].cpu().numpy()
  model.set_kernel('auto')
  np.testing.assert_array_equal(np.isnan(got), np.isnan(other))
  assert np.isnan(got[7]).any() and not np.isnan(np.delete(got, 7, axis=0)).any()
  np.testing.assert_array_equal(got[~np.isnan(got)], other[~np.isnan(other)])
