"""The device-resident command ring behind ddd_stream_fork .. ddd_stream_join (round 6;
include/ddd1d.h, csrc/rhs_ring.h): a caller-owned Runge-Kutta loop (the shape of
integrate.odeint, integrate.py:143-169) served by ONE persistent kernel.  Bit-identical to
one launch per call; the park thread, the back-pressure and the fall-backs behave as the
header says."""
import time

import numpy as np
import pytest

from helpers import batch_forcing, make_model, oracle, random_phase_ic, rel_err

pytestmark = pytest.mark.gpu
TOL = 1e-5


def _midpoint_loop(model, y0, steps, dt, chained, t0=0.0, hook=None):
  import torch
  h = np.float32(dt)
  y, ystage, ynew = y0.clone(), torch.empty_like(y0), torch.empty_like(y0)
  ctx = model.chained_substeps() if chained else None
  if ctx is not None:
    ctx.__enter__()
  try:
    for step in range(steps):
      t = t0 + step * dt
      model.rk_substep(t, y, y_base=y, c1=0.5 * h, y_out=ystage)
      model.rk_substep(t + 0.5 * dt, ystage, acc_in=y, c2=h, acc_out=ynew)
      y, ynew = ynew, y
      if hook is not None:
        hook(step)
  finally:
    if ctx is not None:
      ctx.__exit__(None, None, None)
  return y.cpu().numpy()


def _setup(equation, num_points, batch, conservative=True):
  import torch
  model = make_model(equation, conservative, num_points=num_points, resample_factor=2)
  forcing = batch_forcing(batch) if equation == 'burgers' else None
  if forcing is not None:
    model.set_forcing(forcing)
  y0_host = random_phase_ic(model.equation, batch)
  return model, forcing, y0_host, torch.from_numpy(y0_host).cuda()


@pytest.mark.parametrize('equation,conservative,num_points,batch', [
    ('burgers', True, 64, 4100),    # two row groups per wavefront, the last one ragged
    ('burgers', False, 64, 700),    # fewer groups than the machine holds
    ('kdv', True, 32, 8301),        # two samples per group, odd batch: half a group at the end
    ('ks', True, 64, 2048),
    ('kdv', False, 16, 37),         # four samples per group
])
def test_ring_equals_launches(equation, conservative, num_points, batch):
  model, forcing, y0_host, y0 = _setup(equation, num_points, batch, conservative)
  dt = model.equation.time_step
  steps = 9
  want = _midpoint_loop(model, y0, steps, dt, chained=True)   # the default: launches
  assert model.region_stats() == (0, 0)
  model.set_region_mode('chains')
  want = _midpoint_loop(model, y0, steps, dt, chained=True)
  assert model.region_stats() == (0, 0)
  plain = _midpoint_loop(model, y0, steps, dt, chained=False)
  np.testing.assert_array_equal(plain, want)
  model.set_region_mode('ring')
  got = _midpoint_loop(model, y0, steps, dt, chained=True)
  launches, commands = model.region_stats()
  assert commands == 2 * steps and 1 <= launches <= 2, (launches, commands)
  np.testing.assert_array_equal(got, want)
  persistent = model.integrate_fixed(y0, steps, dt=dt, scheme='midpoint', save_every=steps,
                                     launch_mode='persistent')[0].cpu().numpy()
  np.testing.assert_array_equal(got, persistent)
  rows = np.array([0, batch // 2, batch - 1])
  sub_forcing = None if forcing is None else {k: v[rows] for k, v in forcing.items()}
  ref = oracle.integrate_fixed(model.spec(), oracle.SCHEME_MIDPOINT, 0.0, dt, steps, steps,
                               y0_host[rows], forcing=sub_forcing)
  assert rel_err(got[rows], ref[0]) < TOL


def test_ring_wraps_and_back_pressure():
  """More calls than the ring has slots (256): the host waits for room, nothing is lost."""
  model, _, _, y0 = _setup('burgers', 64, 2048)
  model.set_region_mode('ring')
  dt = model.equation.time_step
  steps = 330   # 660 commands
  got = _midpoint_loop(model, y0, steps, dt, chained=True)
  launches, commands = model.region_stats()
  assert commands == 2 * steps
  want = model.integrate_fixed(y0, steps, dt=dt, scheme='midpoint', save_every=steps,
                               launch_mode='persistent')[0].cpu().numpy()
  np.testing.assert_array_equal(got, want)


def test_ring_parks_when_the_region_idles():
  """A device synchronisation inside an open region: the park thread ends the persistent
  kernel after the idle time, the next call starts it again; results unchanged."""
  import torch
  model, _, _, y0 = _setup('burgers', 64, 512)
  model.set_region_mode('ring')
  dt = model.equation.time_step
  steps = 6
  waits = []

  def hook(step):
    if step in (1, 3):
      t0 = time.perf_counter()
      torch.cuda.synchronize()   # returns once the kernel has been parked
      waits.append(time.perf_counter() - t0)

  got = _midpoint_loop(model, y0, steps, dt, chained=True, hook=hook)
  launches, commands = model.region_stats()
  assert commands == 2 * steps and launches == 3, (launches, commands)
  assert max(waits) < 2.0, waits
  want = model.integrate_fixed(y0, steps, dt=dt, scheme='midpoint', save_every=steps,
                               launch_mode='persistent')[0].cpu().numpy()
  np.testing.assert_array_equal(got, want)


def test_ring_mixed_calls_in_one_region():
  """Batches that change inside a region, ddd_time_derivative next to ddd_rk_substep, and a
  call the ring does not take (derivative views: a launch, ordered behind the commands)."""
  import torch
  model, _, _, y0 = _setup('burgers', 64, 3000)
  dt = model.equation.time_step
  h = np.float32(dt)

  def run(chained):
    outs = []
    ctx = model.chained_substeps() if chained else None
    if ctx is not None:
      ctx.__enter__()
    a = torch.empty_like(y0)
    model.rk_substep(0.0, y0, y_base=y0, c1=0.5 * h, y_out=a)
    small = y0[:100].clone()
    b = torch.empty_like(small)
    model.rk_substep(0.25, small, y_base=small, c1=h, y_out=b)          # another batch
    c = torch.empty_like(a)
    model.time_derivative_rows(a, c, t=0.5)
    d = torch.empty_like(a)
    model.rk_substep(0.5, a, acc_in=y0, c2=h, acc_out=d)
    derivs = model.space_derivatives(d)                                 # closes the region
    if ctx is not None:
      ctx.__exit__(None, None, None)
    outs = [a, b, c, d, derivs]
    return [o.cpu().numpy() for o in outs]

  model.set_region_mode('chains')
  want = run(True)
  model.set_region_mode('ring')
  got = run(True)
  assert model.region_stats()[1] >= 3
  for g, w in zip(got, want):
    np.testing.assert_array_equal(g, w)


def test_ring_regions_back_to_back():
  """Two long regions enqueued without a synchronisation between them: the second region's
  kernel waits behind the first on the stream while the host is already posting its commands
  -- they must not overwrite what the first kernel has not read yet."""
  import torch
  model, _, _, y0 = _setup('burgers', 64, 4096)
  model.set_region_mode('ring')
  dt = model.equation.time_step
  steps = 200   # 400 commands per region, 256 slots
  h = np.float32(dt)
  bufs = [[y0.clone(), torch.empty_like(y0), torch.empty_like(y0)] for _ in range(2)]
  finals = []
  for region in range(2):
    y, ystage, ynew = bufs[region]
    with model.chained_substeps():
      for step in range(steps):
        t = step * dt
        model.rk_substep(t, y, y_base=y, c1=0.5 * h, y_out=ystage)
        model.rk_substep(t + 0.5 * dt, ystage, acc_in=y, c2=h, acc_out=ynew)
        y, ynew = ynew, y
    finals.append(y)
  launches, commands = model.region_stats()
  assert commands == 4 * steps and launches >= 2
  want = model.integrate_fixed(y0, steps, dt=dt, scheme='midpoint', save_every=steps,
                               launch_mode='persistent')[0].cpu().numpy()
  for y in finals:
    np.testing.assert_array_equal(y.cpu().numpy(), want)


def test_ring_c_driver():
  """examples/rk_driver.c inside the region: the C caller of the bench leg."""
  import ctypes
  import torch
  import bench
  import ddd1d_amd
  driver = bench.load_rk_driver()
  model, _, _, y0 = _setup('burgers', 64, 4096)
  model.set_region_mode('ring')
  dt = model.equation.time_step
  steps = 50
  y, ystage, ynew = y0.clone(), torch.empty_like(y0), torch.empty_like(y0)
  final = ctypes.c_void_p()
  rc = driver.rk_driver_midpoint(model._handle, steps, 0.0, dt, y.data_ptr(), ystage.data_ptr(),
                                 ynew.data_ptr(), 4096, ddd1d_amd._lib.current_stream(), 1,
                                 ctypes.byref(final))
  assert rc == 0
  got = (y if final.value == y.data_ptr() else ynew).cpu().numpy()
  assert model.region_stats() == (1, 2 * steps)
  want = model.integrate_fixed(y0, steps, dt=dt, scheme='midpoint', save_every=steps,
                               launch_mode='persistent')[0].cpu().numpy()
  np.testing.assert_array_equal(got, want)
