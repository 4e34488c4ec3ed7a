"""The lane == grid point kernel of csrc/rhs_lean.h (fixed stencils and one-layer nets in
the persistent launch mode) against the oracle and, bit for bit, against the MFMA-path
kernels with the conv tower skipped that carried these models through round 4; and the
batched drivers built on it (create_baseline_data.py's content, baseline_time_evolution)."""
import numpy as np
import pytest

from helpers import (oracle, make_model, random_phase_ic, batch_forcing, rel_err, baseline_spec)
from ddd1d_amd import equations, integrate, model as model_lib

pytestmark = pytest.mark.gpu

TOL = 1e-5
SCHEMES = {'euler': oracle.SCHEME_EULER, 'midpoint': oracle.SCHEME_MIDPOINT,
           'bs3': oracle.SCHEME_BS3, 'rk4': oracle.SCHEME_RK4}


def _both_kernels(model, y0, steps, dt, scheme):
  """(lean kernel, MFMA-path kernel) trajectories of the same persistent job."""
  model.set_kernel('auto')
  lean = model.integrate_fixed(y0, steps, dt=dt, scheme=scheme, save_every=steps // 2).cpu().numpy()
  assert model.kernel_name == 'valu_f32_lean', model.kernel_name
  model.set_kernel('mfma64')          # an explicit kernel choice switches the lean kernel off
  tower = model.integrate_fixed(y0, steps, dt=dt, scheme=scheme, save_every=steps // 2).cpu().numpy()
  assert model.kernel_name == 'mfma_f32_r64', model.kernel_name
  model.set_kernel('auto')
  return lean, tower


@pytest.mark.parametrize('cls_name,n,batch,accuracy_order', [
    ('KdVEquation', 64, 7, 1), ('ConservativeKdVEquation', 64, 70, 1),
    ('KSEquation', 64, 5, 1), ('ConservativeKSEquation', 32, 9, 3),
    ('BurgersEquation', 32, 11, 3), ('ConservativeKdVEquation', 16, 13, 1),
    ('KdVEquation', 8, 17, 1), ('GodunovBurgersEquation', 64, 4, 1),
])
def test_fixed_stencils_on_the_lean_kernel(cls_name, n, batch, accuracy_order):
  """Every fixed-stencil equation family, several samples per wavefront (N < 64), ragged
  last wavefront, forced (Burgers: harmonic sums, all stages of a step in one pass where
  they fit; 20 modes x 64 / N samples must fit the wavefront, so N >= 32) and unforced,
  every scheme: bit-identical to the MFMA-path kernel, oracle at 1e-5."""
  eq = getattr(equations, cls_name)(n, resample_factor=2, random_seed=3)
  model = model_lib.BaselineModel(eq, accuracy_order=accuracy_order)
  y0 = random_phase_ic(eq, batch)
  forcing = batch_forcing(batch) if 'Burgers' in cls_name else None
  if forcing is not None:
    model.set_forcing(forcing)
  dt = eq.time_step
  for scheme in ('midpoint', 'bs3', 'rk4', 'euler'):
    lean, tower = _both_kernels(model, y0, 10, dt, scheme)
    np.testing.assert_array_equal(lean, tower)
    if 'KS' not in cls_name and 'Godunov' not in cls_name:
      want = oracle.integrate_fixed(baseline_spec(eq, accuracy_order), SCHEMES[scheme], 0.0, dt, 10, 5,
                                    y0[:3], forcing=None if forcing is None else
                                    {k: v[:3] for k, v in forcing.items()})
      assert rel_err(lean[:, :3], want) < TOL, (cls_name, scheme)


@pytest.mark.parametrize('equation,conservative,num_points,overrides', [
    ('burgers', True, 64, dict(num_layers=1)),
    ('burgers', False, 32, dict(num_layers=1, kernel_size=3)),
    ('burgers', True, 32, dict(num_layers=1, kernel_size=7)),
    ('kdv', True, 16, dict(num_layers=1, kernel_size=7)),
    ('kdv', True, 64, dict(num_layers=1)),
    ('ks', False, 64, dict(num_layers=1)),
    ('ks', True, 32, dict(num_layers=1, kernel_size=3)),
    ('kdv', False, 8, dict(num_layers=1, polynomial_accuracy_order=0)),
])
def test_one_layer_nets_on_the_lean_kernel(equation, conservative, num_points, overrides):
  model = make_model(equation, conservative, num_points=num_points, resample_factor=2, **overrides)
  batch = 9
  y0 = random_phase_ic(model.equation, batch)
  forcing = batch_forcing(batch)
  model.set_forcing(forcing)
  dt = 1e-5
  for scheme in ('midpoint', 'bs3', 'rk4'):
    lean, tower = _both_kernels(model, y0, 10, dt, scheme)
    np.testing.assert_array_equal(lean, tower)
    want = oracle.integrate_fixed(model.spec(), SCHEMES[scheme], 0.0, dt, 10, 5, y0,
                                  forcing=forcing if equation == 'burgers' else None)
    assert rel_err(lean, want) < TOL, (equation, overrides, scheme)


def test_lean_kernel_full_size_properties():
  """4 096 samples x 1 000 midpoint steps of the one-layer Burgers model (the bench's
  `one_layer_b4096`): determinism, independence of a sample from the batch around it."""
  import torch
  model = make_model('burgers', True, num_points=64, resample_factor=8, num_layers=1)
  batch = 4096
  forcing = batch_forcing(batch)
  model.set_forcing(forcing)
  y0 = random_phase_ic(model.equation, batch)
  out = model.integrate_fixed(y0, 1000, dt=1e-3, save_every=1000)
  assert model.kernel_name == 'valu_f32_lean'
  again = model.integrate_fixed(y0, 1000, dt=1e-3, save_every=1000)
  assert torch.equal(out, again) and bool(torch.isfinite(out).all())
  sub = np.array([0, 1, 777, 4095])
  model.set_forcing({k: v[sub] for k, v in forcing.items()})
  alone = model.integrate_fixed(y0[sub], 1000, dt=1e-3, save_every=1000).cpu().numpy()
  np.testing.assert_array_equal(alone[0], out[0].cpu().numpy()[sub])
  want = oracle.integrate_fixed(model.spec(), oracle.SCHEME_MIDPOINT, 0.0, 1e-3, 1000, 1000,
                                y0[sub], forcing={k: v[sub] for k, v in forcing.items()})
  assert rel_err(alone, want) < TOL


def test_baseline_time_evolution():
  """model.py:162-183: midpoint rule over the accuracy-order-1 baseline, [batch, x, time]."""
  eq = equations.ConservativeKdVEquation(64, resample_factor=1, random_seed=0)
  y0 = random_phase_ic(eq, 5)
  got = model_lib.baseline_time_evolution(y0, 12, eq).cpu().numpy()
  want = oracle.integrate_fixed(baseline_spec(eq, 1), oracle.SCHEME_MIDPOINT, 0.0, eq.time_step,
                                12, 1, y0)
  assert got.shape == (5, 64, 12)
  assert rel_err(got, np.transpose(want, (1, 2, 0))) < TOL


@pytest.mark.parametrize('cls_name,warmup,filter_interval', [
    ('ConservativeBurgersEquation', 0, None),
    ('ConservativeKdVEquation', 0.2, None),
    ('ConservativeKdVEquation', 0.2, 0.1),
])
def test_integrate_baseline_batch_matches_per_sample_runs(cls_name, warmup, filter_interval):
  """create_baseline_data.py:96-130 as one batched job: every (sample, accuracy order)
  trajectory equals the per-sample integrate_baseline call the script makes."""
  seeds, orders = (0, 1, 2), (1, 3)
  times = np.linspace(0, 0.3, 4)
  eqs = [getattr(equations, cls_name)(32, resample_factor=4, random_seed=s) for s in seeds]
  batch = integrate.integrate_baseline_batch(eqs, orders, times, warmup=warmup,
                                             exact_filter_interval=filter_interval)
  y = np.asarray(batch.data_vars['y'][1] if isinstance(batch.data_vars['y'], tuple)
                 else batch.data_vars['y'])
  assert y.shape == (3, 2, 4, 32) and y.dtype == np.float32
  for si, eq in enumerate(eqs):
    for oi, order in enumerate(orders):
      one = integrate.integrate_baseline(eq, times, warmup=warmup, accuracy_order=order,
                                         exact_filter_interval=filter_interval)
      want = np.asarray(one.data_vars['y'][1] if isinstance(one.data_vars['y'], tuple)
                        else one.data_vars['y'])
      np.testing.assert_allclose(y[si, oi], want, rtol=0, atol=1e-6 * max(1.0, np.abs(want).max()))
