"""Hyper-parameter container with the reference's keys and defaults.

The reference stores its configuration in ``tf.contrib.training.HParams``
built by ``training.create_hparams`` (training.py:40-165) and persists it as
``hparams.pbtxt`` (training.py:590-592, 639-647).  The integration path reads
only a handful of keys (equation, conservative, numerical_flux,
equation_kwargs, resample_factor, model_target, num_layers, filter_size,
kernel_size, nonlinearity, polynomial_accuracy_order,
polynomial_accuracy_scale, ensure_unbiased_coefficients,
coefficient_grid_min_size); the training/loss keys are carried along with the
reference's default values so that a full hparams dump round-trips, but
nothing on this path interprets them.

There is no TensorFlow here: ``HParams`` is a plain attribute bag with the
subset of the tf.contrib API the reference's callers use (attribute access,
``values()``, ``override_from_dict``, ``parse('a=1,b=[2,3]')``, ``to_json`` /
``parse_json``).
"""
import json
import math
import os
import re
from typing import Any, Dict

_NAN = float('nan')

# training.py:125-163
_DEFAULTS = (
    # dataset
    ('equation', None),
    ('conservative', True),
    ('numerical_flux', False),
    ('equation_kwargs', '{}'),
    ('resample_factor', 4),
    # network
    ('model_target', 'coefficients'),
    ('num_layers', 3),
    ('filter_size', 32),
    ('kernel_size', 5),
    ('nonlinearity', 'relu'),
    ('polynomial_accuracy_order', 1),
    ('polynomial_accuracy_scale', 1.0),
    ('ensure_unbiased_coefficients', False),
    ('coefficient_grid_min_size', 6),
    # training (unused on the integration path)
    ('base_batch_size', 128),
    ('learning_rates', [1e-3, 1e-4]),
    ('learning_stops', [20000, 40000]),
    ('frac_training', 0.8),
    ('eval_interval', 250),
    ('noise_probability', 0.0),
    ('noise_amplitude', 0.0),
    ('noise_type', 'white'),
    # loss (unused on the integration path)
    ('ground_truth_order', -1),
    ('num_time_steps', 0),
    ('error_floor_quantile', 0.1),
    ('error_scale', [_NAN]),
    ('error_floor', [_NAN]),
    ('error_max', 0.0),
    ('absolute_error_weight', 1.0),
    ('relative_error_weight', 0.0),
    ('space_derivatives_weight', 0.0),
    ('time_derivative_weight', 1.0),
    ('integrated_solution_weight', 0.0),
)

HPARAMS_FILENAME = 'hparams.json'


def _coerce(text: str, like: Any):
  """Parse ``text`` into the type of the existing value ``like``."""
  if isinstance(like, bool):
    lowered = text.strip().lower()
    if lowered in ('true', '1'):
      return True
    if lowered in ('false', '0'):
      return False
    raise ValueError('could not parse {!r} as bool'.format(text))
  if isinstance(like, int):
    return int(text)
  if isinstance(like, float):
    return float(text)
  return text


class HParams(object):
  """Attribute bag of named hyper-parameters."""

  def __init__(self, **kwargs):
    object.__setattr__(self, '_names', [])
    for name, value in kwargs.items():
      self.add_hparam(name, value)

  def add_hparam(self, name: str, value: Any):
    if name in self._names:
      raise ValueError('hyperparameter {} already exists'.format(name))
    self._names.append(name)
    object.__setattr__(self, name, value)

  def __setattr__(self, name, value):
    if name not in self._names:
      raise AttributeError('unknown hyperparameter: {}'.format(name))
    object.__setattr__(self, name, value)

  def __contains__(self, name):
    return name in self._names

  def values(self) -> Dict[str, Any]:
    return {name: getattr(self, name) for name in self._names}

  def override_from_dict(self, values: Dict[str, Any]) -> 'HParams':
    for name, value in values.items():
      if name not in self._names:
        raise ValueError('unknown hyperparameter: {}'.format(name))
      object.__setattr__(self, name, value)
    return self

  def parse(self, spec: str) -> 'HParams':
    """Apply comma-separated ``name=value`` overrides (lists as ``[a,b]``)."""
    pattern = re.compile(r'\s*(\w+)\s*=\s*(\[[^\]]*\]|[^,\[]*)\s*(?:,|$)')
    position = 0
    while position < len(spec):
      match = pattern.match(spec, position)
      if not match:
        raise ValueError('malformed hparams string at {!r}'
                         .format(spec[position:]))
      name, raw = match.group(1), match.group(2).strip()
      if name not in self._names:
        raise ValueError('unknown hyperparameter: {}'.format(name))
      current = getattr(self, name)
      if isinstance(current, list):
        items = raw[1:-1] if raw.startswith('[') else raw
        like = current[0] if current else ''
        value = [_coerce(item, like) for item in items.split(',') if item.strip()]
      else:
        if current is None:
          value = raw
        else:
          value = _coerce(raw, current)
      object.__setattr__(self, name, value)
      position = match.end()
    return self

  def to_json(self) -> str:
    def _encode(value):
      if isinstance(value, float) and math.isnan(value):
        return 'NaN'
      if isinstance(value, list):
        return [_encode(v) for v in value]
      return value
    return json.dumps({k: _encode(v) for k, v in self.values().items()},
                      indent=1, sort_keys=True)

  def parse_json(self, text: str) -> 'HParams':
    def _decode(value):
      if value == 'NaN':
        return _NAN
      if isinstance(value, list):
        return [_decode(v) for v in value]
      return value
    loaded = {k: _decode(v) for k, v in json.loads(text).items()}
    # tolerate keys written by a newer/older version, like load_hparams in
    # the reference (training.py:639-647) keeps defaults for absent keys.
    known = {k: v for k, v in loaded.items() if k in self._names}
    return self.override_from_dict(known)

  def __repr__(self):
    return 'HParams({})'.format(
        ', '.join('{}={!r}'.format(k, v) for k, v in self.values().items()))


def create_hparams(equation: str, **kwargs: Any) -> HParams:
  """Reference defaults (training.py:125-163) with ``kwargs`` overriding."""
  hparams = HParams(**{k: (list(v) if isinstance(v, list) else v)
                       for k, v in _DEFAULTS})
  hparams.equation = equation
  hparams.override_from_dict(kwargs)
  return hparams


def save_hparams(hparams: HParams, checkpoint_dir: str) -> str:
  os.makedirs(checkpoint_dir, exist_ok=True)
  path = os.path.join(checkpoint_dir, HPARAMS_FILENAME)
  with open(path, 'w') as f:
    f.write(hparams.to_json())
  return path


def load_hparams(checkpoint_dir: str) -> HParams:
  """Load hparams saved next to a model checkpoint (training.py:639-647).

  Reads this package's ``hparams.json`` or, when only the reference's
  ``hparams.pbtxt`` (text-format HParamDef, training.py:590-592) is present,
  that; values missing from the file take the defaults, as in the reference.
  """
  path = os.path.join(checkpoint_dir, HPARAMS_FILENAME)
  if not os.path.exists(path):
    from . import checkpoint
    pbtxt = os.path.join(checkpoint_dir, checkpoint.HPARAMS_PBTXT)
    if os.path.exists(pbtxt):
      with open(pbtxt) as f:
        values = checkpoint.parse_hparams_pbtxt(f.read())
      return create_hparams(**values)
  with open(path) as f:
    text = f.read()
  equation = json.loads(text)['equation']
  return create_hparams(equation).parse_json(text)


def checkpoint_dir_to_path(checkpoint_dir: str) -> str:
  """Prefix of the reference's TF checkpoint files (training.py:523-524)."""
  return os.path.join(checkpoint_dir, 'model.ckpt')

