"""The C restatement (oracle/ddd_oracle.c, bench.py's CPU baseline) agrees with
the NumPy oracle, which is pinned against the reference's goldens."""
import os
import sys

import numpy as np
import pytest

from helpers import (oracle, make_model, random_phase_ic, batch_forcing,
                     baseline_spec, rel_err, ROOT)
from ddd1d_amd import equations

sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import c_oracle  # noqa: E402


@pytest.mark.parametrize('equation,conservative,numerical_flux', [
    ('burgers', False, False), ('burgers', True, False), ('burgers', True, True),
    ('kdv', False, False), ('kdv', True, False), ('ks', False, False),
    ('ks', True, False), ('ks', True, True),
])
def test_c_time_derivative_matches_numpy(equation, conservative, numerical_flux):
  model = make_model(equation, conservative, numerical_flux, num_points=64,
                     resample_factor=4)
  spec = model.spec()
  y0 = random_phase_ic(model.equation, 5)
  forcing = batch_forcing(5)
  co = c_oracle.COracle(spec, nparams=20)
  got = co.time_derivative(0.6, y0, forcing)
  want = oracle.time_derivative(spec, 0.6, y0, forcing)
  tol = 5e-4 if equation == 'ks' else 2e-5
  assert rel_err(got, want) < tol


def test_c_integrate_matches_numpy():
  model = make_model('burgers', True, num_points=64, resample_factor=8)
  spec = model.spec()
  y0 = random_phase_ic(model.equation, 4)
  forcing = batch_forcing(4)
  co = c_oracle.COracle(spec, nparams=20)
  for scheme in (oracle.SCHEME_MIDPOINT, oracle.SCHEME_BS3):
    got = co.integrate_fixed(scheme, 0.0, 1e-3, 20, y0, forcing)
    want = oracle.integrate_fixed(spec, scheme, 0.0, 1e-3, 20, 20, y0,
                                  forcing=forcing)[0]
    assert rel_err(got, want) < 1e-5
  assert co.num_threads >= 1


def test_c_baseline_matches_numpy():
  eq = equations.KdVEquation(64, random_seed=2)
  spec = baseline_spec(eq, 3)
  co = c_oracle.COracle(spec)
  y0 = random_phase_ic(eq, 3)
  got = co.time_derivative(0.0, y0)
  want = oracle.time_derivative(spec, 0.0, y0)
  assert rel_err(got, want) < 2e-5
