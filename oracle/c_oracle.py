"""ctypes access to oracle/liboracle.so (C restatement; TEST INFRASTRUCTURE).

Only tests/ and bench.py's cpu_baseline leg may import this.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, 'liboracle.so')


class OracleSpec(ctypes.Structure):
  _fields_ = [
      ('equation', ctypes.c_int), ('n', ctypes.c_int), ('d', ctypes.c_int),
      ('g', ctypes.c_int), ('layers', ctypes.c_int), ('filters', ctypes.c_int),
      ('ksize', ctypes.c_int), ('act', ctypes.c_int), ('c_out', ctypes.c_int),
      ('fixed', ctypes.c_int), ('conservative', ctypes.c_int),
      ('forced', ctypes.c_int), ('nparams', ctypes.c_int),
      ('resample_factor', ctypes.c_int),
      ('eta', ctypes.c_float), ('stddev', ctypes.c_float),
      ('inv_dx', ctypes.c_float), ('period', ctypes.c_double),
      ('in_start', ctypes.c_int * 4), ('in_size', ctypes.c_int * 4),
      ('weights', ctypes.POINTER(ctypes.c_float)),
      ('nullspace', ctypes.POINTER(ctypes.c_float)),
      ('bias', ctypes.POINTER(ctypes.c_float)),
  ]


_ACT = {'relu': 0, 'relu6': 1, 'tanh': 2, 'softplus': 3, 'elu': 4}
_lib = None


def load(build=True):
  global _lib
  if _lib is not None:
    return _lib
  if build and not os.path.exists(_PATH):
    subprocess.run(['make', '-C', _HERE], check=True, capture_output=True)
  lib = ctypes.CDLL(_PATH)
  lib.oracle_num_threads.restype = ctypes.c_int
  lib.oracle_integrate_fixed.restype = ctypes.c_int
  _lib = lib
  return lib


def _fp(a):
  return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _dp(a):
  return None if a is None else a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


class COracle(object):
  """Holds the C spec (and the arrays it points to) for one model spec dict."""

  def __init__(self, spec, nparams=0):
    self.lib = load()
    s = OracleSpec()
    s.equation = spec['equation']
    s.n = spec['num_points']
    s.d = len(spec['derivative_orders'])
    s.conservative = int(spec['conservative'])
    s.forced = int(spec.get('forced', False))
    s.nparams = nparams
    s.resample_factor = spec['resample_factor']
    s.eta = spec['eta']
    s.inv_dx = 1.0 / spec['dx']
    s.period = spec['period']
    self._keep = []
    if spec.get('baseline_coefficients') is not None:
      stencils = spec['baseline_coefficients']
      width = max(len(t) for t in stencils)
      table = np.zeros((len(stencils), width), np.float32)
      for d, taps in enumerate(stencils):
        shift = width // 2 - len(taps) // 2
        table[d, shift:shift + len(taps)] = np.asarray(taps, np.float32)
      s.fixed = 1
      s.g = width
      s.stddev = 1.0
      bias = np.ascontiguousarray(table)
      s.bias = _fp(bias)
      self._keep.append(bias)
    else:
      if spec.get('model_target', 'coefficients') != 'coefficients' or not spec[
          'polynomial_accuracy_order'] or spec['num_layers'] < 1:
        raise NotImplementedError('C oracle covers the coefficients target with '
                                  'polynomial accuracy layers only')
      s.fixed = 0
      s.g = spec['stencil_size']
      s.stddev = spec['standard_deviation']
      s.layers = spec['num_layers']
      s.ksize = spec['conv_kernels'][0].shape[0]
      s.filters = spec['conv_kernels'][0].shape[2] if s.layers > 1 else 1
      s.act = _ACT[spec['nonlinearity']]
      s.c_out = spec['conv_kernels'][-1].shape[2]
      start = 0
      for i, ns in enumerate(spec['nullspaces']):
        s.in_start[i] = start
        s.in_size[i] = np.shape(ns)[0]
        start += np.shape(ns)[0]
      weights = np.ascontiguousarray(np.concatenate(
          [np.concatenate([np.asarray(w, np.float32).ravel(),
                           np.asarray(b, np.float32).ravel()])
           for w, b in zip(spec['conv_kernels'], spec['conv_biases'])]))
      nullspace = np.ascontiguousarray(np.concatenate(
          [np.asarray(ns, np.float32).ravel() for ns in spec['nullspaces']]))
      bias = np.ascontiguousarray(np.concatenate(
          [np.asarray(b, np.float32).ravel() for b in spec['biases']]))
      s.weights, s.nullspace, s.bias = _fp(weights), _fp(nullspace), _fp(bias)
      self._keep += [weights, nullspace, bias]
    self.spec = s

  @property
  def num_threads(self):
    return self.lib.oracle_num_threads()

  def _forcing(self, forcing):
    if forcing is None:
      return None, None, None, None
    arrs = [np.ascontiguousarray(np.asarray(forcing[k], dtype=np.float64))
            for k in ('a', 'omega', 'k', 'phi')]
    self._frc = arrs
    return arrs

  def time_derivative(self, t, y, forcing=None):
    y = np.ascontiguousarray(np.asarray(y, np.float32))
    out = np.empty_like(y)
    a, om, k, ph = self._forcing(forcing)
    self.lib.oracle_time_derivative(ctypes.byref(self.spec), ctypes.c_double(t),
                                    _fp(y), _dp(a), _dp(om), _dp(k), _dp(ph),
                                    _fp(out), y.shape[0])
    return out

  def integrate_fixed(self, scheme, t0, dt, num_steps, y0, forcing=None):
    y = np.ascontiguousarray(np.array(y0, np.float32))
    a, om, k, ph = self._forcing(forcing)
    rc = self.lib.oracle_integrate_fixed(
        ctypes.byref(self.spec), int(scheme), ctypes.c_double(t0),
        ctypes.c_double(dt), int(num_steps), _fp(y), _dp(a), _dp(om), _dp(k),
        _dp(ph), y.shape[0])
    if rc != 0:
      raise ValueError('unknown scheme')
    return y
