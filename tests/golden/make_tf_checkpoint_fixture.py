#!/usr/bin/env python
"""Hand-assembles a reference-layout checkpoint directory from the PUBLIC format
definitions, without importing anything from this package:

    tests/golden/tf_checkpoint/hparams.pbtxt
    tests/golden/tf_checkpoint/model.ckpt.index
    tests/golden/tf_checkpoint/model.ckpt.data-00000-of-00001
    tests/golden/tf_checkpoint/expected.npz      (the arrays that went in)

so that `checkpoint.py`'s READER is checked against bytes its own WRITER did not
produce.  No trained checkpoint ships with the reference
and TensorFlow is not installed, so this is still not a TF-written file; it is
the layout TF's BundleWriter documents:

  * tensorflow/core/protobuf/tensor_bundle.proto -- BundleHeaderProto (key ""):
      1 num_shards (varint), 2 endianness (LITTLE = 0), 3 version {1 producer};
    BundleEntryProto: 1 dtype, 2 shape {2 dim {1 size}}, 3 shard_id, 4 offset,
      5 size, 6 crc32c (fixed32, masked)
  * tensorflow/core/lib/io/table_format.txt (LevelDB table): data blocks of
      (shared, non_shared, value_len, key suffix, value) entries + restart
      array + restart count; per block a 1-byte compression type and a masked
      crc32c of (block + type); index block; 48-byte footer = metaindex handle,
      index handle, zero padding, magic 0xdb4775248b80fb57
  * crc32c (Castagnoli, reflected 0x82F63B78); mask = rotr15(crc) + 0xa282ead8
  * variable names: training.py / notebooks/time-integration.ipynb:597-602
      predict_coefficients/conv1d{,_1,_2}/{kernel,bias}, plus optimizer slots
      and global_step the reader must skip
  * hparams.pbtxt: text-format HParamDef (contrib/training/hparam.proto) as
      training.py:590-592 writes it

Deliberate differences from this package's writer, so the two do not share
code paths: several data blocks with prefix-compressed keys and a restart
interval of 2 (TF: 16), block boundaries every ~3 entries, tensors stored in
the data shard in sorted-name order with 8 bytes of padding between them.

Run from the repo root:  python tests/golden/make_tf_checkpoint_fixture.py
"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, 'tf_checkpoint')
MAGIC = 0xdb4775248b80fb57


# ---- primitives (local implementations) ------------------------------------
def varint(value):
  out = bytearray()
  while True:
    byte = value & 0x7f
    value >>= 7
    if value:
      out.append(byte | 0x80)
    else:
      out.append(byte)
      return bytes(out)


def crc32c(data):
  crc = 0xffffffff
  for byte in data:
    crc ^= byte
    for _ in range(8):
      crc = (crc >> 1) ^ (0x82f63b78 if crc & 1 else 0)
  return crc ^ 0xffffffff


def masked(crc):
  return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


def key_field(number, wire_type):
  return varint((number << 3) | wire_type)


def pb_varint(number, value):
  return key_field(number, 0) + varint(value)


def pb_bytes(number, payload):
  return key_field(number, 2) + varint(len(payload)) + payload


def pb_fixed32(number, value):
  return key_field(number, 5) + struct.pack('<I', value)


# ---- tensor bundle -----------------------------------------------------------
DTYPE_ENUM = {'float32': 1, 'float64': 2, 'int32': 3, 'int64': 9}


def header_proto():
  version = pb_varint(1, 1)                       # VersionDef.producer = 1
  # num_shards = 1; endianness LITTLE (0) is the proto default and is omitted,
  # as a real serializer does
  return pb_varint(1, 1) + pb_bytes(3, version)


def entry_proto(array, offset):
  raw = array.astype(array.dtype.newbyteorder('<')).tobytes()
  shape = b''.join(pb_bytes(2, pb_varint(1, int(d))) for d in array.shape)
  out = pb_varint(1, DTYPE_ENUM[array.dtype.name])
  if array.ndim:                                  # scalars: empty shape message omitted
    out += pb_bytes(2, shape)
  # shard_id = 0 omitted (default); offset omitted when 0 (default)
  if offset:
    out += pb_varint(4, offset)
  out += pb_varint(5, len(raw))
  out += pb_fixed32(6, masked(crc32c(raw)))
  return out, raw


def build_block(entries, restart_interval):
  """entries: sorted [(key, value)]; keys prefix-compressed between restarts."""
  out = bytearray()
  restarts = []
  previous = b''
  for index, (key, value) in enumerate(entries):
    if index % restart_interval == 0:
      restarts.append(len(out))
      shared = 0
    else:
      shared = 0
      while shared < min(len(previous), len(key)) and previous[shared] == key[shared]:
        shared += 1
    suffix = key[shared:]
    out += varint(shared) + varint(len(suffix)) + varint(len(value)) + suffix + value
    previous = key
  for r in restarts:
    out += struct.pack('<I', r)
  out += struct.pack('<I', len(restarts))
  return bytes(out)


def write_table(path, pairs, entries_per_block=3):
  pairs = sorted(pairs)
  out = bytearray()

  def emit(block):
    offset = len(out)
    out.extend(block)
    out.append(0)                                 # kNoCompression
    out.extend(struct.pack('<I', masked(crc32c(block + b'\x00'))))
    return varint(offset) + varint(len(block))

  index_entries = []
  for start in range(0, len(pairs), entries_per_block):
    chunk = pairs[start:start + entries_per_block]
    handle = emit(build_block(chunk, restart_interval=2))
    # index key: any key >= the block's last key and < the next block's first;
    # use the last key itself (what LevelDB does for the final block)
    index_entries.append((chunk[-1][0], handle))
  metaindex = emit(build_block([], restart_interval=1))
  index = emit(build_block(index_entries, restart_interval=1))
  footer = metaindex + index
  out.extend(footer + b'\x00' * (40 - len(footer)) + struct.pack('<Q', MAGIC))
  with open(path, 'wb') as f:
    f.write(bytes(out))


def main():
  os.makedirs(OUT, exist_ok=True)
  rs = np.random.RandomState(20260927)
  # the network of notebooks/time-integration.ipynb:597-602 (Burgers, C_out = 9)
  def glorot(shape):
    fan_in, fan_out = shape[0] * shape[1], shape[0] * shape[2]
    lim = np.sqrt(6.0 / (fan_in + fan_out))
    return rs.uniform(-lim, lim, size=shape).astype(np.float32)
  scope = 'predict_coefficients/'
  tensors = {
      scope + 'conv1d/kernel': glorot((5, 1, 32)),
      scope + 'conv1d/bias': rs.uniform(-0.05, 0.05, 32).astype(np.float32),
      scope + 'conv1d_1/kernel': glorot((5, 32, 32)),
      scope + 'conv1d_1/bias': rs.uniform(-0.05, 0.05, 32).astype(np.float32),
      scope + 'conv1d_2/kernel': (0.1 * glorot((5, 32, 9))).astype(np.float32),
      scope + 'conv1d_2/bias': rs.uniform(-0.005, 0.005, 9).astype(np.float32),
  }
  expected = dict(tensors)
  # what a training run leaves next to them (the reader must skip these)
  for name in list(tensors):
    tensors[name + '/Adam'] = np.zeros_like(tensors[name])
    tensors[name + '/Adam_1'] = np.full_like(tensors[name], 1e-8)
  tensors['beta1_power'] = np.array(0.9 ** 3, dtype=np.float32)
  tensors['beta2_power'] = np.array(0.999 ** 3, dtype=np.float32)
  tensors['global_step'] = np.array(40000, dtype=np.int64)

  data = bytearray()
  pairs = [(b'', header_proto())]
  for name in sorted(tensors):
    if data:
      data += b'\xee' * 8                         # padding between tensors
    proto, raw = entry_proto(tensors[name], len(data))
    data += raw
    pairs.append((name.encode('utf-8'), proto))
  with open(os.path.join(OUT, 'model.ckpt.data-00000-of-00001'), 'wb') as f:
    f.write(bytes(data))
  write_table(os.path.join(OUT, 'model.ckpt.index'), pairs)

  # hparams.pbtxt as training.py:590-592 writes it (text-format HParamDef): keys
  # of training.create_hparams (training.py:125-163)
  hparams = [
      ('equation', 'bytes_value', '"burgers"'),
      ('conservative', 'bool_value', 'true'),
      ('numerical_flux', 'bool_value', 'false'),
      ('equation_kwargs', 'bytes_value', r'"{\"num_points\": 512}"'),
      ('resample_factor', 'int64_value', '8'),
      ('model_target', 'bytes_value', '"coefficients"'),
      ('num_layers', 'int64_value', '3'),
      ('filter_size', 'int64_value', '32'),
      ('kernel_size', 'int64_value', '5'),
      ('nonlinearity', 'bytes_value', '"relu"'),
      ('polynomial_accuracy_order', 'int64_value', '1'),
      ('polynomial_accuracy_scale', 'float_value', '1.0'),
      ('coefficient_grid_min_size', 'int64_value', '6'),
      ('ensure_unbiased_coefficients', 'bool_value', 'false'),
      ('num_time_steps', 'int64_value', '0'),
      ('base_batch_size', 'int64_value', '128'),
      ('frac_training', 'float_value', '0.8'),
      ('noise_type', 'bytes_value', '"white"'),
      ('ground_truth_order', 'int64_value', '-1'),
  ]
  with open(os.path.join(OUT, 'hparams.pbtxt'), 'w') as f:
    for key, kind, value in hparams:
      f.write('hparam {\n  key: "%s"\n  value {\n    %s: %s\n  }\n}\n' % (key, kind, value))
    f.write('hparam {\n  key: "learning_rates"\n  value {\n    float_list {\n'
            '      value: 0.001\n      value: 0.0001\n    }\n  }\n}\n')
    f.write('hparam {\n  key: "learning_stops"\n  value {\n    int64_list {\n'
            '      value: 20000\n      value: 40000\n    }\n  }\n}\n')
  np.savez(os.path.join(OUT, 'expected.npz'),
           **{k.replace('/', '__'): v for k, v in expected.items()})
  print('wrote', OUT, 'index', os.path.getsize(os.path.join(OUT, 'model.ckpt.index')),
        'B, data', len(data), 'B')


if __name__ == '__main__':
  main()
