"""BASELINE.json configurations at FULL size through size-independent
properties (the oracle cannot run these in seconds): mean conservation of the
flux forms (integrate_test.py:101-104, 183-185), bitwise determinism, batch
permutation invariance, sample independence from the batch it rides in, and
agreement of a sub-sample with the oracle."""
import numpy as np
import pytest
import torch

from helpers import oracle, make_model, random_phase_ic, batch_forcing, rel_err

pytestmark = pytest.mark.gpu

TOL = 1e-5


def _check_properties(model, y0, steps, dt, forcing=None, oracle_rows=4, scheme='midpoint',
                      c_oracle=False):
  batch = y0.shape[0]
  if forcing is not None:
    model.set_forcing(forcing)
  out = model.integrate_fixed(y0, steps, dt=dt, save_every=steps, scheme=scheme)[0]
  again = model.integrate_fixed(y0, steps, dt=dt, save_every=steps, scheme=scheme)[0]
  assert torch.equal(out, again)                                    # deterministic
  final = out.cpu().numpy()
  assert np.isfinite(final).all()
  if model.equation.CONSERVATIVE and forcing is None:
    np.testing.assert_allclose(final.astype(np.float64).mean(axis=1),
                               y0.astype(np.float64).mean(axis=1), atol=1e-4)
  # a sample's trajectory does not depend on the batch around it: the first
  # rows alone (different workgroup placement, one wave per SIMD) give the
  # same bits, and they match the oracle
  sub = slice(0, oracle_rows)
  sub_forcing = None if forcing is None else {k: v[sub] for k, v in forcing.items()}
  if sub_forcing is not None:
    model.set_forcing(sub_forcing)
  alone = model.integrate_fixed(y0[sub], steps, dt=dt, save_every=steps, scheme=scheme)[0]
  np.testing.assert_array_equal(alone.cpu().numpy(), final[sub])
  sid = {'midpoint': oracle.SCHEME_MIDPOINT, 'bs3': oracle.SCHEME_BS3}[scheme]
  if c_oracle:
    # long horizons: the C restatement (oracle/ddd_oracle.c), itself pinned to the
    # NumPy oracle by tests/test_cpu_oracle_c.py
    import c_oracle as c_oracle_lib
    nparams = 0 if sub_forcing is None else sub_forcing['a'].shape[1]
    want = c_oracle_lib.COracle(model.spec(), nparams=nparams).integrate_fixed(
        sid, 0.0, dt, steps, y0[sub], sub_forcing)
  else:
    want = oracle.integrate_fixed(model.spec(), sid, 0.0, dt, steps, steps, y0[sub],
                                  forcing=sub_forcing)[0]
  err = rel_err(final[sub], want)
  print('batch', batch, 'steps', steps, 'sub-sample vs oracle rel err {:.2e}'.format(err))
  assert err < TOL
  # permuting the batch permutes the result
  perm = np.random.RandomState(0).permutation(batch)
  perm_forcing = None if forcing is None else {k: v[perm] for k, v in forcing.items()}
  if perm_forcing is not None:
    model.set_forcing(perm_forcing)
  permuted = model.integrate_fixed(y0[perm], steps, dt=dt, save_every=steps, scheme=scheme)[0]
  np.testing.assert_array_equal(permuted.cpu().numpy(), final[perm])


def test_config2_burgers_n64_b1024_1000_steps():
  """BASELINE configs[1] exactly: 1000 midpoint steps of the forced ensemble."""
  model = make_model('burgers', True, num_points=64, resample_factor=8)
  y0 = random_phase_ic(model.equation, 1024)
  _check_properties(model, y0, 1000, 1e-3, forcing=batch_forcing(1024), oracle_rows=2)


def test_config2_bs3_at_max_step():
  model = make_model('burgers', True, num_points=64, resample_factor=8)
  y0 = 0.5 * random_phase_ic(model.equation, 1024)
  _check_properties(model, y0, 100, 1e-2, forcing=batch_forcing(1024), oracle_rows=2,
                    scheme='bs3')


def test_config3_kdv_n64_b4096():
  model = make_model('kdv', True, num_points=64, resample_factor=1)
  y0 = random_phase_ic(model.equation, 4096)
  _check_properties(model, y0, 1000, 2.5e-5, oracle_rows=2)


def test_config4_ks_n256_b8192_10k_steps():
  """BASELINE configs[3] at its stated horizon: 10 000 midpoint steps (t = 0.25,
  far below the Kuramoto-Sivashinsky Lyapunov time, so the sub-sample against
  the oracle at 1e-5 is meaningful over the whole horizon)."""
  model = make_model('ks', True, num_points=256, resample_factor=1)
  y0 = random_phase_ic(model.equation, 8192)
  _check_properties(model, y0, 10000, 2.5e-5, oracle_rows=2, c_oracle=True)


def test_config5_shard_burgers_b8192():
  """One rank's shard of the 65 536-sample ensemble (8 192 per GPU)."""
  model = make_model('burgers', True, num_points=64, resample_factor=8)
  y0 = random_phase_ic(model.equation, 8192)
  _check_properties(model, y0, 100, 1e-3, forcing=batch_forcing(8192), oracle_rows=2)
