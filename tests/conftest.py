"""pytest configuration: markers, import paths, shared fixtures."""
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)
GOLDEN_DIR = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
  config.addinivalue_line('markers', 'gpu: needs a real MI355X (gfx950) GPU')


class Golden(object):
  """Lazy accessor for tests/golden/reference_numpy_paths.npz."""

  def __init__(self):
    self._npz = np.load(os.path.join(GOLDEN_DIR, 'reference_numpy_paths.npz'))
    with open(os.path.join(GOLDEN_DIR, 'reference_numpy_paths.json')) as f:
      self.index = json.load(f)

  def __getitem__(self, key):
    return self._npz[key]

  def keys(self):
    return self._npz.files


@pytest.fixture(scope='session')
def golden():
  return Golden()
