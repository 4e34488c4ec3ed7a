"""Reading the reference's training artefacts without TensorFlow.

A reference checkpoint directory (training.py:586-592, 515-524) holds
  hparams.pbtxt                      text-format HParamDef proto
  model.ckpt.index                   tensor-bundle index: an SSTable (LevelDB
                                     table format) mapping variable name ->
                                     serialized BundleEntryProto
  model.ckpt.data-00000-of-00001     the tensors' little-endian bytes
with the conv tower stored as ``predict_coefficients/conv1d{,_1,_2}/{kernel,
bias}`` (tf.layers.conv1d variables created in model.py:455-495; names as
listed in notebooks/time-integration.ipynb).  This module parses all three
formats from their public definitions (protobuf wire/text format, LevelDB
table_format.md, tensorflow/core/protobuf/tensor_bundle.proto) and also
writes them, which is how the tests exercise the reader -- no checkpoint ships
with the reference repository, so byte-level parity with a real TF writer is
UNPINNED.

Only what the integration path needs is implemented: float / double / int32 /
int64 tensors, un-sliced entries, single-shard bundles, raw or
snappy-compressed index blocks.
"""
import json
import os
import struct
from typing import Dict, Iterable, List, Tuple

import numpy as np

HPARAMS_PBTXT = 'hparams.pbtxt'
CHECKPOINT_PREFIX = 'model.ckpt'          # training.checkpoint_dir_to_path
_TABLE_MAGIC = 0xdb4775248b80fb57
_DTYPES = {1: np.dtype('<f4'), 2: np.dtype('<f8'), 3: np.dtype('<i4'), 9: np.dtype('<i8')}
_DTYPE_IDS = {np.dtype(v).newbyteorder('=').name: k for k, v in _DTYPES.items()}


# ---------------------------------------------------------------------------
# hparams.pbtxt  (tensorflow/contrib/training/python/training/hparam.proto)
# ---------------------------------------------------------------------------
def _tokenize(text: str) -> List[str]:
  tokens, i, n = [], 0, len(text)
  while i < n:
    ch = text[i]
    if ch.isspace():
      i += 1
    elif ch == '#':
      while i < n and text[i] != '\n':
        i += 1
    elif ch in '{}:':
      tokens.append(ch)
      i += 1
    elif ch in '"\'':
      j = i + 1
      while text[j] != ch:
        j += 2 if text[j] == '\\' else 1
      tokens.append(text[i:j + 1])
      i = j + 1
    else:
      j = i
      while j < n and not text[j].isspace() and text[j] not in '{}:':
        j += 1
      tokens.append(text[i:j])
      i = j
  return tokens


def _unescape(literal: str) -> bytes:
  """Text-format string literal (C escapes, octal, hex) -> bytes."""
  body, out, i = literal[1:-1], bytearray(), 0
  simple = {'n': 10, 't': 9, 'r': 13, '\\': 92, '"': 34, "'": 39, 'a': 7, 'b': 8,
            'f': 12, 'v': 11}
  while i < len(body):
    ch = body[i]
    if ch != '\\':
      out.extend(ch.encode('utf-8'))
      i += 1
      continue
    nxt = body[i + 1]
    if nxt in simple:
      out.append(simple[nxt]); i += 2
    elif nxt in 'xX':
      j = i + 2
      while j < len(body) and j < i + 4 and body[j] in '0123456789abcdefABCDEF':
        j += 1
      out.append(int(body[i + 2:j], 16)); i = j
    else:
      j = i + 1
      while j < len(body) and j < i + 4 and body[j] in '01234567':
        j += 1
      out.append(int(body[i + 1:j], 8)); i = j
  return bytes(out)


def _parse_message(tokens: List[str], pos: int) -> Tuple[list, int]:
  """[(field, value)] where value is a scalar token or a nested message list."""
  fields = []
  while pos < len(tokens) and tokens[pos] != '}':
    name = tokens[pos]
    pos += 1
    if tokens[pos] == ':':
      pos += 1
    if tokens[pos] == '{':
      value, pos = _parse_message(tokens, pos + 1)
      pos += 1                      # the closing brace
    else:
      value = tokens[pos]
      pos += 1
    fields.append((name, value))
  return fields, pos


def _scalar(kind: str, token: str):
  if kind.startswith('int64'):
    return int(token)
  if kind.startswith('float'):
    return float(token)
  if kind.startswith('bool'):
    return token == 'true'
  return _unescape(token).decode('utf-8')   # bytes_value: hparam strings are text


def parse_hparams_pbtxt(text: str) -> Dict[str, object]:
  """HParamDef text proto -> {name: value}; lists become Python lists."""
  message, _ = _parse_message(_tokenize(text), 0)
  values = {}
  for name, entry in message:
    if name != 'hparam':
      continue
    key, value = None, None
    for field, content in entry:
      if field == 'key':
        key = _unescape(content).decode('utf-8')
      elif field == 'value':
        for kind, payload in content:
          if kind.endswith('_list'):
            value = [_scalar(kind, tok) for f, tok in payload if f == 'value']
          else:
            value = _scalar(kind, payload)
    values[key] = value
  return values


def _escape(text: str) -> str:
  out = []
  for byte in text.encode('utf-8'):
    ch = chr(byte)
    if ch in '"\\':
      out.append('\\' + ch)
    elif 32 <= byte < 127:
      out.append(ch)
    else:
      out.append('\\%03o' % byte)
  return '"' + ''.join(out) + '"'


def format_hparams_pbtxt(values: Dict[str, object]) -> str:
  """The inverse of parse_hparams_pbtxt (what ``str(hparams.to_proto())`` emits)."""
  def scalar(v):
    if isinstance(v, bool):
      return 'bool', 'true' if v else 'false'
    if isinstance(v, (int, np.integer)):
      return 'int64', str(int(v))
    if isinstance(v, (float, np.floating)):
      return 'float', repr(float(v))
    return 'bytes', _escape(str(v))
  lines = []
  for key in sorted(values):
    value = values[key]
    if value is None:
      continue
    lines.append('hparam {\n  key: %s\n  value {' % _escape(key))
    if isinstance(value, (list, tuple)):
      kind = scalar(value[0])[0] if value else 'float'
      lines.append('    %s_list {' % kind)
      lines.extend('      value: %s' % scalar(v)[1] for v in value)
      lines.append('    }')
    else:
      kind, token = scalar(value)
      lines.append('    %s_value: %s' % (kind, token))
    lines.append('  }\n}')
  return '\n'.join(lines) + '\n'


# ---------------------------------------------------------------------------
# protobuf wire format, crc32c, snappy: the few primitives the bundle needs
# ---------------------------------------------------------------------------
def _read_varint(buf: bytes, pos: int) -> Tuple[int, int]:
  result, shift = 0, 0
  while True:
    byte = buf[pos]
    pos += 1
    result |= (byte & 0x7f) << shift
    if not byte & 0x80:
      return result, pos
    shift += 7


def _write_varint(value: int) -> bytes:
  out = bytearray()
  while True:
    byte = value & 0x7f
    value >>= 7
    out.append(byte | (0x80 if value else 0))
    if not value:
      return bytes(out)


def _parse_proto(buf: bytes) -> Dict[int, list]:
  """field number -> [values]; varints as int, length-delimited as bytes,
  fixed32/64 as int."""
  fields, pos = {}, 0
  while pos < len(buf):
    tag, pos = _read_varint(buf, pos)
    number, wire = tag >> 3, tag & 7
    if wire == 0:
      value, pos = _read_varint(buf, pos)
    elif wire == 1:
      value = struct.unpack_from('<Q', buf, pos)[0]; pos += 8
    elif wire == 2:
      size, pos = _read_varint(buf, pos)
      value = bytes(buf[pos:pos + size]); pos += size
    elif wire == 5:
      value = struct.unpack_from('<I', buf, pos)[0]; pos += 4
    else:
      raise ValueError('unsupported protobuf wire type {}'.format(wire))
    fields.setdefault(number, []).append(value)
  return fields


_CRC_TABLE = None


def crc32c(data: bytes, crc: int = 0) -> int:
  """CRC-32C (Castagnoli), the checksum LevelDB tables and bundles use."""
  global _CRC_TABLE
  if _CRC_TABLE is None:
    table = []
    for i in range(256):
      c = i
      for _ in range(8):
        c = (c >> 1) ^ 0x82f63b78 if c & 1 else c >> 1
      table.append(c)
    _CRC_TABLE = table
  crc ^= 0xffffffff
  for byte in data:
    crc = _CRC_TABLE[(crc ^ byte) & 0xff] ^ (crc >> 8)
  return crc ^ 0xffffffff


def _mask_crc(crc: int) -> int:
  return (((crc >> 15) | (crc << 17)) + 0xa282ead8) & 0xffffffff


def _snappy_decompress(data: bytes) -> bytes:
  length, pos = _read_varint(data, 0)
  out = bytearray()
  while pos < len(data):
    tag = data[pos]; pos += 1
    kind = tag & 3
    if kind == 0:                                   # literal
      size = tag >> 2
      if size >= 60:
        extra = size - 59
        size = int.from_bytes(data[pos:pos + extra], 'little'); pos += extra
      size += 1
      out += data[pos:pos + size]; pos += size
      continue
    if kind == 1:
      size = ((tag >> 2) & 7) + 4
      offset = ((tag >> 5) << 8) | data[pos]; pos += 1
    elif kind == 2:
      size = (tag >> 2) + 1
      offset = int.from_bytes(data[pos:pos + 2], 'little'); pos += 2
    else:
      size = (tag >> 2) + 1
      offset = int.from_bytes(data[pos:pos + 4], 'little'); pos += 4
    for _ in range(size):                           # may overlap: byte by byte
      out.append(out[-offset])
  if len(out) != length:
    raise ValueError('corrupt snappy block')
  return bytes(out)


# ---------------------------------------------------------------------------
# SSTable (LevelDB table_format.md) -- reader and a minimal writer
# ---------------------------------------------------------------------------
def _read_block(data: bytes, offset: int, size: int, verify: bool) -> bytes:
  raw = data[offset:offset + size]
  kind = data[offset + size]
  if verify:
    stored = struct.unpack_from('<I', data, offset + size + 1)[0]
    if stored != _mask_crc(crc32c(data[offset:offset + size + 1])):
      raise ValueError('checkpoint index block checksum mismatch')
  if kind == 0:
    return raw
  if kind == 1:
    return _snappy_decompress(raw)
  raise ValueError('unknown table block compression {}'.format(kind))


def _block_entries(block: bytes) -> Iterable[Tuple[bytes, bytes]]:
  num_restarts = struct.unpack_from('<I', block, len(block) - 4)[0]
  limit = len(block) - 4 - 4 * num_restarts
  pos, key = 0, b''
  while pos < limit:
    shared, pos = _read_varint(block, pos)
    unshared, pos = _read_varint(block, pos)
    value_len, pos = _read_varint(block, pos)
    key = key[:shared] + block[pos:pos + unshared]
    pos += unshared
    yield key, block[pos:pos + value_len]
    pos += value_len


def read_table(data: bytes, verify: bool = True) -> Dict[bytes, bytes]:
  """All key/value pairs of an SSTable held in ``data``."""
  if len(data) < 48 or struct.unpack_from('<Q', data, len(data) - 8)[0] != _TABLE_MAGIC:
    raise ValueError('not a TensorFlow checkpoint index (bad table magic)')
  footer = data[-48:]
  _, pos = _read_varint(footer, 0)           # metaindex handle: offset
  _, pos = _read_varint(footer, pos)         #                   size
  index_offset, pos = _read_varint(footer, pos)
  index_size, pos = _read_varint(footer, pos)
  entries = {}
  for _, handle in _block_entries(_read_block(data, index_offset, index_size, verify)):
    offset, p = _read_varint(handle, 0)
    size, _ = _read_varint(handle, p)
    for key, value in _block_entries(_read_block(data, offset, size, verify)):
      entries[key] = value
  return entries


def _build_block(pairs: List[Tuple[bytes, bytes]]) -> bytes:
  """One block, every entry a restart point (no prefix sharing)."""
  body, restarts = bytearray(), []
  for key, value in pairs:
    restarts.append(len(body))
    body += _write_varint(0) + _write_varint(len(key)) + _write_varint(len(value))
    body += key + value
  if not restarts:
    restarts = [0]
  for r in restarts:
    body += struct.pack('<I', r)
  body += struct.pack('<I', len(restarts))
  return bytes(body)


def write_table(pairs: List[Tuple[bytes, bytes]]) -> bytes:
  """SSTable with one data block, uncompressed, keys in sorted order."""
  out = bytearray()

  def emit(block: bytes) -> bytes:
    offset = len(out)
    out.extend(block)
    out.append(0)                                             # kNoCompression
    out.extend(struct.pack('<I', _mask_crc(crc32c(block + b'\x00'))))
    return _write_varint(offset) + _write_varint(len(block))

  pairs = sorted(pairs)
  data_handle = emit(_build_block(pairs))
  meta_handle = emit(_build_block([]))
  last_key = pairs[-1][0] if pairs else b''
  index_handle = emit(_build_block([(last_key, data_handle)]))
  footer = meta_handle + index_handle
  footer += b'\x00' * (40 - len(footer)) + struct.pack('<Q', _TABLE_MAGIC)
  out.extend(footer)
  return bytes(out)


# ---------------------------------------------------------------------------
# tensor bundle (tensorflow/core/protobuf/tensor_bundle.proto)
# ---------------------------------------------------------------------------
def _shape_of(entry: Dict[int, list]) -> Tuple[int, ...]:
  if 2 not in entry:
    return ()
  dims = []
  for dim in _parse_proto(entry[2][0]).get(2, []):      # TensorShapeProto.dim
    dims.append(_parse_proto(dim).get(1, [0])[0])       # Dim.size
  return tuple(dims)


def read_checkpoint(prefix: str, verify: bool = True) -> Dict[str, np.ndarray]:
  """All tensors of the TF checkpoint ``prefix`` (e.g. .../model.ckpt)."""
  with open(prefix + '.index', 'rb') as f:
    entries = read_table(f.read(), verify)
  header = _parse_proto(entries.pop(b'', b''))
  num_shards = header.get(1, [1])[0]
  if header.get(2, [0])[0] != 0:
    raise ValueError('big-endian checkpoints are not supported')
  shards = {}
  tensors = {}
  for key, blob in entries.items():
    entry = _parse_proto(blob)
    if 7 in entry:
      raise ValueError('sliced (partitioned) variable {!r} is not supported'.format(key))
    dtype_id = entry.get(1, [0])[0]
    if dtype_id not in _DTYPES:
      continue                                            # strings etc.: not on this path
    shard = entry.get(3, [0])[0]
    if shard not in shards:
      path = '{}.data-{:05d}-of-{:05d}'.format(prefix, shard, num_shards)
      with open(path, 'rb') as f:
        shards[shard] = f.read()
    offset, size = entry.get(4, [0])[0], entry.get(5, [0])[0]
    raw = shards[shard][offset:offset + size]
    if verify and 6 in entry and entry[6][0] != _mask_crc(crc32c(raw)):
      raise ValueError('tensor {!r} checksum mismatch'.format(key))
    tensors[key.decode('utf-8')] = np.frombuffer(raw, dtype=_DTYPES[dtype_id]).reshape(
        _shape_of(entry)).copy()
  return tensors


def write_checkpoint(prefix: str, tensors: Dict[str, np.ndarray]) -> None:
  """Single-shard tensor bundle with the layout TF's BundleWriter produces."""
  def field(number, wire, payload):
    return _write_varint((number << 3) | wire) + payload

  data = bytearray()
  pairs = [(b'', field(1, 0, _write_varint(1)) +                     # num_shards
            field(3, 2, _write_varint(2) + field(1, 0, _write_varint(1))))]  # version.producer
  for name in sorted(tensors):
    array = np.asarray(tensors[name])   # tobytes() below is C-ordered; keeps 0-d scalars
    dtype_id = _DTYPE_IDS[array.dtype.name]
    raw = array.astype(_DTYPES[dtype_id]).tobytes()
    shape = b''.join(field(2, 2, _len_prefixed(field(1, 0, _write_varint(d))))
                     for d in array.shape)
    entry = field(1, 0, _write_varint(dtype_id)) + field(2, 2, _len_prefixed(shape))
    if len(data):
      entry += field(4, 0, _write_varint(len(data)))
    entry += field(5, 0, _write_varint(len(raw)))
    entry += field(6, 5, struct.pack('<I', _mask_crc(crc32c(raw))))
    pairs.append((name.encode('utf-8'), entry))
    data += raw
  with open(prefix + '.index', 'wb') as f:
    f.write(write_table(pairs))
  with open(prefix + '.data-00000-of-00001', 'wb') as f:
    f.write(bytes(data))


def _len_prefixed(payload: bytes) -> bytes:
  return _write_varint(len(payload)) + payload


# ---------------------------------------------------------------------------
# the reference's variable naming
# ---------------------------------------------------------------------------
def conv_variable_names(num_layers: int,
                        model_target: str = 'coefficients') -> List[Tuple[str, str]]:
  """(kernel, bias) variable names in graph order.

  tf.layers.conv1d uniquifies as conv1d, conv1d_1, ...  Only
  ``model.predict_coefficients`` opens ``tf.variable_scope('predict_coefficients')``
  (model.py:442); the heads for model_target in {space_derivatives,
  time_derivative, flux} are built by ``_multilayer_conv1d`` with no scope
  (model.py:551-569), so their variables are plain ``conv1d/kernel`` etc."""
  prefix = 'predict_coefficients/' if model_target == 'coefficients' else ''
  names = []
  for layer in range(num_layers):
    scope = prefix + 'conv1d' + ('_%d' % layer if layer else '')
    names.append((scope + '/kernel', scope + '/bias'))
  return names


CONSTANT_COEFFICIENTS_NAME = 'predict_coefficients/coefficients'   # model.py:496-499


def load_conv_weights(checkpoint_dir: str, num_layers: int,
                      model_target: str = 'coefficients'):
  """(kernels [K, Cin, Cout], biases [Cout]) float32 lists from model.ckpt."""
  tensors = read_checkpoint(os.path.join(checkpoint_dir, CHECKPOINT_PREFIX))
  kernels, biases = [], []
  for kernel_name, bias_name in conv_variable_names(num_layers, model_target):
    if kernel_name not in tensors or bias_name not in tensors:
      raise KeyError('checkpoint has no variable {!r}; found {}'.format(
          kernel_name, sorted(k for k in tensors if 'Adam' not in k)))
    kernels.append(tensors[kernel_name].astype(np.float32))
    biases.append(tensors[bias_name].astype(np.float32))
  return kernels, biases


def load_constant_coefficients(checkpoint_dir: str) -> np.ndarray:
  """The learned constant vector of a ``num_layers = 0`` model
  (``predict_coefficients/coefficients``, model.py:496-499)."""
  tensors = read_checkpoint(os.path.join(checkpoint_dir, CHECKPOINT_PREFIX))
  if CONSTANT_COEFFICIENTS_NAME not in tensors:
    raise KeyError('checkpoint has no variable {!r}; found {}'.format(
        CONSTANT_COEFFICIENTS_NAME, sorted(k for k in tensors if 'Adam' not in k)))
  return tensors[CONSTANT_COEFFICIENTS_NAME].astype(np.float32)


def has_tf_checkpoint(checkpoint_dir: str) -> bool:
  return os.path.exists(os.path.join(checkpoint_dir, CHECKPOINT_PREFIX + '.index'))
