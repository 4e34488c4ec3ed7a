"""Reference-layout checkpoint directory -> LearnedStencilModel.load -> HIP
kernels, checked against the oracle fed the SAME arrays (training.py:586-592,
639-647 + integrate.py:66-68: restore -> differentiator), and
integrate_model_from_warm_start (integrate.py:399-427)."""
import os

import numpy as np
import pytest

import ddd1d_amd
from helpers import batch_forcing, random_phase_ic, rel_err
from ddd1d_amd import checkpoint, equations, integrate, model as model_lib

pytestmark = pytest.mark.gpu

FIXTURE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tf_checkpoint')


def _oracle_spec_from_arrays(model):
  """The oracle's description built from the fixture's own arrays (not from
  what the loader put into the model)."""
  expected = np.load(os.path.join(FIXTURE_DIR, 'expected.npz'))
  spec = dict(model.spec())
  spec['conv_kernels'] = [expected['predict_coefficients__conv1d{}__kernel'.format(s)]
                          for s in ('', '_1', '_2')]
  spec['conv_biases'] = [expected['predict_coefficients__conv1d{}__bias'.format(s)]
                         for s in ('', '_1', '_2')]
  return spec


def test_hand_assembled_checkpoint_runs_on_the_kernels():
  import oracle
  model = model_lib.LearnedStencilModel.load(FIXTURE_DIR)
  assert model.kernel_name.startswith('mfma_f32')
  spec = _oracle_spec_from_arrays(model)
  batch = 9
  y0 = random_phase_ic(model.equation, batch)
  forcing = batch_forcing(batch)
  model.set_forcing(forcing)
  got = model.time_derivative(y0, 0.3).cpu().numpy()
  want = oracle.time_derivative(spec, 0.3, y0, forcing)
  assert rel_err(got, want) < 1e-5
  traj = model.integrate_fixed(y0, 50, dt=1e-3, scheme='midpoint', save_every=25).cpu().numpy()
  ref = oracle.integrate_fixed(spec, oracle.SCHEME_MIDPOINT, 0.0, 1e-3, 50, 25, y0,
                               forcing=forcing)
  assert np.isfinite(traj).all() and rel_err(traj, ref) < 1e-5


def test_written_reference_layout_dir_round_trips_onto_the_kernels(tmp_path):
  """hparams.pbtxt + model.ckpt.{index,data-00000-of-00001} written here with
  the reference's variable names, loaded back and run: KdV, 50 steps."""
  import oracle
  hp = ddd1d_amd.create_hparams('kdv', conservative=True, resample_factor=4,
                                equation_kwargs='{"num_points": 256}')
  _, eq = equations.from_hparams(hp)
  source = model_lib.LearnedStencilModel(eq, hp, init_seed=11)
  tensors = {}
  for (kname, bname), w, b in zip(checkpoint.conv_variable_names(3), source.conv_kernels,
                                  source.conv_biases):
    tensors[kname], tensors[bname] = w, b
  (tmp_path / 'hparams.pbtxt').write_text(checkpoint.format_hparams_pbtxt(hp.values()))
  checkpoint.write_checkpoint(str(tmp_path / 'model.ckpt'), tensors)
  model = model_lib.LearnedStencilModel.load(str(tmp_path))
  y0 = random_phase_ic(model.equation, 7)
  got = model.time_derivative(y0, 0.0).cpu().numpy()
  want = oracle.time_derivative(source.spec(), 0.0, y0, None)
  assert rel_err(got, want) < 1e-5
  dt = model.equation.time_step
  traj = model.integrate_fixed(y0, 50, dt=dt, scheme='midpoint', save_every=50).cpu().numpy()
  ref = oracle.integrate_fixed(source.spec(), oracle.SCHEME_MIDPOINT, 0.0, dt, 50, 50, y0)
  assert rel_err(traj, ref) < 1e-5


def test_integrate_model_from_warm_start_matches_oracle():
  """integrate.py:399-427: restore the model from the checkpoint dir and run
  SciPy RK23 from a given state; the oracle runs the same controller over the
  NumPy right-hand side with the same seed's forcing."""
  import oracle
  times = np.linspace(0, 0.2, 5)
  hp = ddd1d_amd.load_hparams(FIXTURE_DIR)
  _, eq = equations.from_hparams(hp, random_seed=3)
  y0 = 0.5 * random_phase_ic(eq, 1)[0]
  ds = integrate.integrate_model_from_warm_start(FIXTURE_DIR, y0, random_seed=3, times=times,
                                                 warmup=0.1)
  y = integrate._dataset_array(ds, 'y')
  nfev = int(np.asarray(integrate._dataset_coord(ds, 'num_evals')))
  np.testing.assert_allclose(np.asarray(integrate._dataset_coord(ds, 'time')), 0.1 + times)
  model = model_lib.LearnedStencilModel.load(FIXTURE_DIR)
  spec = _oracle_spec_from_arrays(model)
  forcing = {k: v[0] for k, v in model_lib.forcing_from_equations([eq]).items()}
  want, want_nfev = oracle.odeint_rk23(spec, y0, 0.1 + times, forcing)
  assert y.shape == (5, 64) and nfev == want_nfev
  np.testing.assert_array_equal(y[0], y0)
  assert rel_err(y, want) < 1e-5
