"""TF-free readers for the reference's training artefacts (hparams.pbtxt and
the model.ckpt tensor bundle).  No checkpoint ships with the reference, so the
formats are exercised through this package's own writers plus hand-built
byte strings from the public format definitions (parity with a real TF writer
is unpinned)."""
import os
import struct

import numpy as np
import pytest

import ddd1d_amd
from ddd1d_amd import checkpoint, equations, model as model_lib


HPARAMS_TEXT = r'''
hparam {
  key: "equation"
  value {
    bytes_value: "kdv"
  }
}
hparam { key: "conservative" value { bool_value: true } }
hparam { key: "num_layers" value { int64_value: 3 } }
hparam { key: "polynomial_accuracy_scale" value { float_value: 1.5 } }
hparam {
  key: "equation_kwargs"
  value { bytes_value: "{\"num_points\": 256}" }   # escaped quotes
}
hparam {
  key: "learning_rates"
  value { float_list { value: 0.001 value: 1e-04 } }
}
hparam { key: "learning_stops" value { int64_list { value: 20000 value: 40000 } } }
hparam { key: "nonlinearity" value { bytes_value: "r\145lu" } }
'''


def test_parse_hparams_pbtxt_text_format():
  values = checkpoint.parse_hparams_pbtxt(HPARAMS_TEXT)
  assert values == {
      'equation': 'kdv', 'conservative': True, 'num_layers': 3,
      'polynomial_accuracy_scale': 1.5, 'equation_kwargs': '{"num_points": 256}',
      'learning_rates': [0.001, 1e-4], 'learning_stops': [20000, 40000],
      'nonlinearity': 'relu'}                       # octal escape \145 = 'e'


def test_hparams_pbtxt_round_trip_and_load(tmp_path):
  hp = ddd1d_amd.create_hparams('burgers', conservative=False, resample_factor=8,
                                equation_kwargs='{"num_points": 512}', filter_size=32)
  text = checkpoint.format_hparams_pbtxt(hp.values())
  np.testing.assert_equal(checkpoint.parse_hparams_pbtxt(text),      # NaN-aware
                          {k: v for k, v in hp.values().items() if v is not None})
  (tmp_path / 'hparams.pbtxt').write_text(text)
  loaded = ddd1d_amd.load_hparams(str(tmp_path))
  np.testing.assert_equal(loaded.values(), hp.values())
  # a file written by an older version lacks new keys: defaults fill in
  (tmp_path / 'hparams.pbtxt').write_text(checkpoint.format_hparams_pbtxt(
      {'equation': 'ks', 'num_layers': 2}))
  partial = ddd1d_amd.load_hparams(str(tmp_path))
  assert partial.equation == 'ks' and partial.num_layers == 2
  assert partial.kernel_size == ddd1d_amd.create_hparams('ks').kernel_size


def test_crc32c_and_varints():
  assert checkpoint.crc32c(b'123456789') == 0xe3069283          # the standard check value
  assert checkpoint.crc32c(b'') == 0
  for value in (0, 1, 127, 128, 300, 2 ** 32 + 5):
    assert checkpoint._read_varint(checkpoint._write_varint(value), 0)[0] == value


def test_sstable_reader_handles_prefix_compression_and_snappy():
  """A table built by hand the way LevelDB does it: shared key prefixes with a
  restart interval, two data blocks, one of them snappy-compressed."""
  def entry(shared, key_suffix, value):
    return (checkpoint._write_varint(shared) + checkpoint._write_varint(len(key_suffix)) +
            checkpoint._write_varint(len(value)) + key_suffix + value)
  block1 = entry(0, b'conv1d/bias', b'AAAA') + entry(7, b'kernel', b'BBBBBB')
  block1 += struct.pack('<II', 0, 1)                             # one restart, at 0
  block2 = entry(0, b'conv1d_1/bias', b'CC') + struct.pack('<II', 0, 1)
  snappy2 = (checkpoint._write_varint(len(block2)) +
             bytes([(len(block2) - 1) << 2]) + block2)           # one literal run
  out = bytearray()
  def emit(payload, kind, raw=None):
    offset = len(out)
    out.extend(payload); out.append(kind)
    out.extend(struct.pack('<I', checkpoint._mask_crc(checkpoint.crc32c(payload + bytes([kind])))))
    return checkpoint._write_varint(offset) + checkpoint._write_varint(len(payload))
  h1 = emit(block1, 0)
  h2 = emit(snappy2, 1)
  meta = emit(checkpoint._build_block([]), 0)
  index = emit(checkpoint._build_block([(b'conv1d/kernel', h1), (b'conv1d_1/bias', h2)]), 0)
  footer = meta + index
  out.extend(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', 0xdb4775248b80fb57))
  table = checkpoint.read_table(bytes(out))
  assert table == {b'conv1d/bias': b'AAAA', b'conv1d/kernel': b'BBBBBB', b'conv1d_1/bias': b'CC'}
  corrupted = bytearray(out); corrupted[3] ^= 1
  with pytest.raises(ValueError, match='checksum'):
    checkpoint.read_table(bytes(corrupted))
  with pytest.raises(ValueError, match='table magic'):
    checkpoint.read_table(b'\x00' * 64)


def test_tensor_bundle_round_trip(tmp_path):
  rs = np.random.RandomState(0)
  tensors = {
      'predict_coefficients/conv1d/kernel': rs.randn(5, 1, 32).astype(np.float32),
      'predict_coefficients/conv1d/bias': rs.randn(32).astype(np.float32),
      'predict_coefficients/conv1d_1/kernel': rs.randn(5, 32, 32).astype(np.float32),
      'predict_coefficients/conv1d_1/bias': rs.randn(32).astype(np.float32),
      'predict_coefficients/conv1d_1/kernel/Adam': rs.randn(5, 32, 32).astype(np.float32),
      'global_step': np.array(40000, dtype=np.int64),
      'beta1_power': np.array(0.5, dtype=np.float32),
      'doubles': rs.randn(3, 2),
  }
  prefix = str(tmp_path / 'model.ckpt')
  checkpoint.write_checkpoint(prefix, tensors)
  assert os.path.exists(prefix + '.index') and os.path.exists(prefix + '.data-00000-of-00001')
  loaded = checkpoint.read_checkpoint(prefix)
  assert sorted(loaded) == sorted(tensors)
  for name, want in tensors.items():
    assert loaded[name].dtype == want.dtype and loaded[name].shape == want.shape
    np.testing.assert_array_equal(loaded[name], want)
  # flipped data byte: the per-tensor crc32c catches it
  path = prefix + '.data-00000-of-00001'
  raw = bytearray(open(path, 'rb').read()); raw[10] ^= 0xff
  open(path, 'wb').write(bytes(raw))
  with pytest.raises(ValueError, match='checksum mismatch'):
    checkpoint.read_checkpoint(prefix)
  assert checkpoint.read_checkpoint(prefix, verify=False) is not None


def test_model_loads_from_reference_style_checkpoint_dir(tmp_path):
  """hparams.pbtxt + model.ckpt -> LearnedStencilModel with the same weights
  and locally rebuilt null-space tables (model.py:480-489)."""
  hp = ddd1d_amd.create_hparams('burgers', conservative=True, resample_factor=8,
                                equation_kwargs='{"num_points": 512}')
  _, eq = equations.from_hparams(hp)
  model = model_lib.LearnedStencilModel(eq, hp, init_seed=3)
  (tmp_path / 'hparams.pbtxt').write_text(checkpoint.format_hparams_pbtxt(hp.values()))
  tensors = {}
  for (kname, bname), w, b in zip(checkpoint.conv_variable_names(hp.num_layers),
                                  model.conv_kernels, model.conv_biases):
    tensors[kname], tensors[bname] = w, b
    tensors[kname + '/Adam'] = np.zeros_like(w)                 # optimizer slots are ignored
  tensors['global_step'] = np.array(7, np.int64)
  checkpoint.write_checkpoint(str(tmp_path / 'model.ckpt'), tensors)
  assert checkpoint.conv_variable_names(3)[2] == ('predict_coefficients/conv1d_2/kernel',
                                                  'predict_coefficients/conv1d_2/bias')
  restored = model_lib.LearnedStencilModel.load(str(tmp_path))
  np.testing.assert_equal(restored.hparams.values(), hp.values())
  for a, b in zip(restored.conv_kernels, model.conv_kernels):
    np.testing.assert_array_equal(a, b)
  for a, b in zip(restored.nullspaces, model.nullspaces):
    np.testing.assert_array_equal(a, b)
  # a missing layer is reported with the variables that do exist
  del tensors['predict_coefficients/conv1d_2/kernel']
  checkpoint.write_checkpoint(str(tmp_path / 'model.ckpt'), tensors)
  with pytest.raises(KeyError, match='conv1d_2/kernel'):
    model_lib.LearnedStencilModel.load(str(tmp_path))


# ---------------------------------------------------------------------------
# bytes this package's writer did not produce (tests/golden/tf_checkpoint,
# assembled from the public format definitions by make_tf_checkpoint_fixture.py)
# ---------------------------------------------------------------------------
FIXTURE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'tf_checkpoint')


def test_reader_on_hand_assembled_bundle():
  tensors = checkpoint.read_checkpoint(os.path.join(FIXTURE_DIR, 'model.ckpt'))
  expected = np.load(os.path.join(FIXTURE_DIR, 'expected.npz'))
  assert len(expected.files) == 6
  for key in expected.files:
    name = key.replace('__', '/')
    assert tensors[name].dtype == np.float32
    np.testing.assert_array_equal(tensors[name], expected[key])
  # training leftovers are read faithfully and ignored by the model loader
  assert tensors['global_step'] == 40000 and tensors['global_step'].dtype == np.int64
  assert abs(float(tensors['beta1_power']) - 0.729) < 1e-6
  assert tensors['predict_coefficients/conv1d_1/kernel/Adam_1'].shape == (5, 32, 32)
  hp = ddd1d_amd.load_hparams(FIXTURE_DIR)
  assert (hp.equation, hp.conservative, hp.num_layers, hp.resample_factor) == (
      'burgers', True, 3, 8)
  assert hp.learning_rates == [0.001, 0.0001] and hp.learning_stops == [20000, 40000]
  model = model_lib.LearnedStencilModel.load(FIXTURE_DIR)
  assert type(model.equation).__name__ == 'ConservativeBurgersEquation'
  assert model.equation.grid.solution_num_points == 64
  for layer, suffix in enumerate(('', '_1', '_2')):
    np.testing.assert_array_equal(
        model.conv_kernels[layer],
        expected['predict_coefficients__conv1d{}__kernel'.format(suffix)])
    np.testing.assert_array_equal(
        model.conv_biases[layer],
        expected['predict_coefficients__conv1d{}__bias'.format(suffix)])


@pytest.mark.parametrize('target', ['coefficients', 'space_derivatives',
                                    'time_derivative', 'flux'])
def test_variable_scope_follows_model_target(tmp_path, target):
  """Only predict_coefficients opens the 'predict_coefficients' variable scope
  (model.py:442); the direct heads are built by _multilayer_conv1d without one
  (model.py:551-569) -> plain conv1d/kernel ...  (ADVICE r1)."""
  conservative = target != 'time_derivative'
  hp = ddd1d_amd.create_hparams('burgers', conservative=conservative, resample_factor=8,
                                equation_kwargs='{"num_points": 512}', model_target=target)
  _, eq = equations.from_hparams(hp)
  if target == 'flux' and not eq.CONSERVATIVE:
    pytest.skip('flux head needs a conservative equation')
  model = model_lib.LearnedStencilModel(eq, hp, init_seed=5)
  names = checkpoint.conv_variable_names(hp.num_layers, target)
  prefix = 'predict_coefficients/' if target == 'coefficients' else ''
  assert names[0] == (prefix + 'conv1d/kernel', prefix + 'conv1d/bias')
  assert names[2] == (prefix + 'conv1d_2/kernel', prefix + 'conv1d_2/bias')
  tensors = {}
  for (kname, bname), w, b in zip(names, model.conv_kernels, model.conv_biases):
    tensors[kname], tensors[bname] = w, b
  (tmp_path / 'hparams.pbtxt').write_text(checkpoint.format_hparams_pbtxt(hp.values()))
  checkpoint.write_checkpoint(str(tmp_path / 'model.ckpt'), tensors)
  restored = model_lib.LearnedStencilModel.load(str(tmp_path))
  assert restored.hparams.model_target == target
  for a, b in zip(restored.conv_kernels, model.conv_kernels):
    np.testing.assert_array_equal(a, b)
  # the other scope is NOT accepted silently
  wrong = {('predict_coefficients/' + k if not prefix else k[len(prefix):]): v
           for k, v in tensors.items()}
  checkpoint.write_checkpoint(str(tmp_path / 'model.ckpt'), wrong)
  with pytest.raises(KeyError, match='conv1d/kernel'):
    model_lib.LearnedStencilModel.load(str(tmp_path))


def test_num_layers_zero_checkpoint_restores_the_learned_constants(tmp_path):
  """num_layers = 0: the trained vector is predict_coefficients/coefficients
  (model.py:496-499); it must not be dropped."""
  hp = ddd1d_amd.create_hparams('burgers', conservative=True, resample_factor=8,
                                equation_kwargs='{"num_points": 512}', num_layers=0)
  _, eq = equations.from_hparams(hp)
  probe = model_lib.LearnedStencilModel(eq, hp)
  count = probe.num_outputs
  learned = np.arange(1, count + 1, dtype=np.float32)
  (tmp_path / 'hparams.pbtxt').write_text(checkpoint.format_hparams_pbtxt(hp.values()))
  checkpoint.write_checkpoint(str(tmp_path / 'model.ckpt'),
                              {'predict_coefficients/coefficients': learned,
                               'global_step': np.array(1, np.int64)})
  restored = model_lib.LearnedStencilModel.load(str(tmp_path))
  np.testing.assert_array_equal(restored.constant_coefficients, learned)
  checkpoint.write_checkpoint(str(tmp_path / 'model.ckpt'),
                              {'global_step': np.array(1, np.int64)})
  with pytest.raises(KeyError, match='predict_coefficients/coefficients'):
    model_lib.LearnedStencilModel.load(str(tmp_path))
