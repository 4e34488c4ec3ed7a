"""Host logic and the C-ABI surface, without a GPU."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

from helpers import make_model, make_hparams, ROOT
import ddd1d_amd
from ddd1d_amd import _lib, distributed, equations, integrate, model as model_lib

HEADER = os.path.join(ROOT, 'include', 'ddd1d.h')


def _declared_symbols():
  text = open(HEADER).read()
  text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
  return sorted(set(re.findall(r'\b(ddd_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
  """The shared library loads (no GPU needed) and exports exactly the entry
  points include/ddd1d.h declares; the ctypes table covers all of them."""
  declared = _declared_symbols()
  assert len(declared) >= 20
  lib = _lib.load_library()
  for name in declared:
    assert hasattr(lib, name), name
  assert sorted(_lib.SIGNATURES) == declared
  out = subprocess.run(['nm', '-D', '--defined-only', _lib.LIBRARY_PATH],
                       capture_output=True, text=True, check=True).stdout
  # the ENTIRE defined dynamic symbol table (any type letter), not only ` T ddd_*`:
  # product ABI == the header.  No profiling / debug entry points (those live in
  # libddd1d_probe.so, __graft_entry__.build_probe), no C++ host launchers
  # (ddd::launch::*), no __device_stub__ symbols (-fvisibility=hidden + DDD_API).
  exported = set()
  for row in out.splitlines():
    parts = row.split()
    if parts:
      exported.add(parts[-1])
  # (linker-defined section markers some toolchains add to every shared object)
  exported -= {'_init', '_fini', '_edata', '_end', '__bss_start'}
  assert exported == set(declared), sorted(exported ^ set(declared))
  assert not hasattr(lib, 'ddd_debug_set_option')
  with pytest.raises(_lib.DDDError, match='probe'):
    _lib.debug_set_option('no_spec', 1)
  assert lib.ddd_abi_version() == 1
  assert [lib.ddd_scheme_stages(s) for s in range(4)] == [1, 2, 3, 4]
  assert lib.ddd_scheme_stages(9) == -1


def test_config_struct_matches_header():
  assert ctypes.sizeof(_lib.DDDConfig) == 128
  assert _lib.DDDConfig.dx.offset == 32
  assert _lib.DDDConfig.stencil_size.offset == 64


def test_argument_errors_without_device():
  """Validation happens before any device work, with reference-style messages."""
  lib = _lib.load_library()
  cfg = _lib.DDDConfig()
  handle = ctypes.c_void_p()
  rc = lib.ddd_model_create(ctypes.byref(cfg), None, 0, None, 0, None, 0,
                            ctypes.byref(handle))
  assert rc == -1
  assert b'struct_size' in lib.ddd_last_error()
  cfg.struct_size = ctypes.sizeof(_lib.DDDConfig)
  cfg.equation = 42
  rc = lib.ddd_model_create(ctypes.byref(cfg), None, 0, None, 0, None, 0,
                            ctypes.byref(handle))
  assert rc == -1 and b'unknown equation' in lib.ddd_last_error()
  assert lib.ddd_time_derivative(None, 0.0, None, None, 1, None) == -1


def test_product_path_has_no_cpu_fallback():
  import torch
  if torch.cuda.is_available():
    pytest.skip('GPU present')
  model = make_model('burgers', True)
  with pytest.raises(RuntimeError, match='no CPU fallback'):
    model.time_derivative(np.zeros((1, 64), np.float32))
  with pytest.raises(RuntimeError, match='no CPU fallback'):
    integrate.integrate_baseline(equations.BurgersEquation(32),
                                 times=np.linspace(0, 1, 3))


def test_model_save_load_roundtrip(tmp_path):
  model = make_model('kdv', True, num_points=64, resample_factor=4,
                     num_layers=4, nonlinearity='tanh')
  model.save(str(tmp_path))
  loaded = model_lib.LearnedStencilModel.load(str(tmp_path))
  assert loaded.hparams.values().keys() == model.hparams.values().keys()
  assert loaded.hparams.num_layers == 4 and loaded.hparams.nonlinearity == 'tanh'
  assert type(loaded.equation) is type(model.equation)
  for a, b in zip(model.conv_kernels + model.conv_biases,
                  loaded.conv_kernels + loaded.conv_biases):
    np.testing.assert_array_equal(a, b)
  for a, b in zip(model.nullspaces + model.biases,
                  loaded.nullspaces + loaded.biases):
    np.testing.assert_array_equal(a, b)


def test_model_shapes_and_errors():
  model = make_model('ks', True, num_points=64)
  assert [w.shape for w in model.conv_kernels] == [(5, 1, 32), (5, 32, 32),
                                                   (5, 32, 11)]
  assert model.input_sizes == [5, 4, 2] and model.stencil_size == 6
  plain = make_model('burgers', False, num_points=64)
  assert plain.input_sizes == [5, 4] and plain.stencil_size == 7
  hp = make_hparams('burgers', True, model_target='nonsense')
  _, eq = equations.from_hparams(hp)
  with pytest.raises(NotImplementedError, match='unrecognized model_target'):
    model_lib.LearnedStencilModel(eq, hp)
  hp = make_hparams('burgers', True, polynomial_accuracy_order=0,
                    ensure_unbiased_coefficients=True)
  _, eq = equations.from_hparams(hp)
  with pytest.raises(ValueError, match='0th order'):
    model_lib.LearnedStencilModel(eq, hp)
  hp = make_hparams('burgers', True, polynomial_accuracy_order=7)
  _, eq = equations.from_hparams(hp)
  with pytest.raises(ValueError, match='no valid|only one valid'):
    model_lib.LearnedStencilModel(eq, hp)
  with pytest.raises(ValueError, match='conv kernel shapes'):
    model_lib.LearnedStencilModel(plain.equation, plain.hparams,
                                  plain.conv_kernels[:2], plain.conv_biases[:2])


def test_forcing_tables_shapes():
  eq = equations.ConservativeBurgersEquation(32, resample_factor=4)
  forcing = model_lib.batched_forcing_parameters(range(5), nparams=20)
  for i in range(5):   # identical to RandomForcing(seed=i)
    ref = equations.RandomForcing(eq.grid, nparams=20, seed=i)
    np.testing.assert_array_equal(forcing['a'][i], ref.a[:, 0])
    np.testing.assert_array_equal(forcing['k'][i], ref.k[:, 0])
    np.testing.assert_array_equal(forcing['phi'][i], ref.phi[:, 0])
  tab = model_lib.forcing_kernel_tables(forcing, eq.grid)
  assert tab['amplitude'].shape == (5, 20) and tab['amplitude'].dtype == np.float32
  assert tab['spatial_phase'].shape[1] == 32
  assert tab['k_index'].max() < tab['spatial_phase'].shape[0]
  model = make_model('burgers', True, num_points=32)
  with pytest.raises(ValueError, match='share one'):
    model.set_forcing(dict(a=np.zeros((2, 3)), omega=np.zeros((2, 3)),
                           k=np.zeros((2, 4)), phi=np.zeros((2, 3))))


def test_forcing_tables_with_zero_wavenumber():
  """equation_kwargs k_min = 0 draws constant modes (k = 0): the Dirichlet
  factor of the block mean is 1 there, not 0/0.  The folded tables
  must reproduce the reference-grid evaluation (oracle.forcing_f32)."""
  import oracle
  eq = equations.ConservativeBurgersEquation(32, resample_factor=4, k_min=0, k_max=2,
                                             random_seed=4)
  forcing = model_lib.forcing_from_equations([eq])
  assert (forcing['k'] == 0).any()
  tab = model_lib.forcing_kernel_tables(forcing, eq.grid)
  for key in ('amplitude', 'phase', 'spatial_phase'):
    assert np.isfinite(tab[key]).all(), key
  t = 0.37
  phase = (tab['omega'][..., None] * np.float32(t)
           + tab['spatial_phase'][tab['k_index']] + tab['phase'][..., None])
  got = np.sum(tab['amplitude'][..., None] * np.sin(phase.astype(np.float64)), axis=1)
  want = oracle.forcing_f32(t, forcing, 32, 4, eq.grid.period, True)
  np.testing.assert_allclose(got, want, atol=2e-6)
  # and it is what the host equation computes (RandomForcing + Grid.resample)
  np.testing.assert_allclose(got[0], eq.forcing(t), atol=2e-6)


def test_shard_bounds():
  for total, world in [(65536, 8), (10, 3), (7, 8), (0, 4)]:
    covered = []
    for rank in range(world):
      lo, hi = distributed.shard_bounds(total, rank, world)
      assert 0 <= lo <= hi <= total
      covered.extend(range(lo, hi))
    assert covered == list(range(total))
    sizes = [np.diff(distributed.shard_bounds(total, r, world))[0]
             for r in range(world)]
    assert max(sizes) - min(sizes) <= 1
  with pytest.raises(ValueError):
    distributed.shard_bounds(10, 4, 4)
  assert list(distributed.weak_shard_ids(4, 2)) == [8, 9, 10, 11]


def test_integrate_batch_argument_checks():
  model = make_model('burgers', True)
  with pytest.raises(ValueError, match='uniformly spaced'):
    integrate.integrate_batch(model, np.zeros((1, 64)), np.array([0, 0.1, 0.3]))
  with pytest.raises(ValueError, match='not a multiple'):
    integrate.integrate_batch(model, np.zeros((1, 64)), np.array([0, 0.015, 0.03]),
                              dt=0.01)
  # warm-up runs the fine-grid exact solver on the GPU: no silent CPU fallback
  import torch
  if not torch.cuda.is_available():
    with pytest.raises(RuntimeError, match='no CPU fallback'):
      integrate.integrate(model.equation, integrate.Differentiator(), warmup=1.0)
