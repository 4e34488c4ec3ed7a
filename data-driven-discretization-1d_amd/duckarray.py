"""Small array helpers used by the equation definitions (host side).

The reference's ``pde_superresolution/duckarray.py`` duck-types between NumPy
and TF tensors.  There is no graph tracer here, so these operate on NumPy
arrays (and anything NumPy can coerce); device-resident batches go through the
HIP kernels instead.  Only the members the integration path touches are
provided: duckarray.py:33-60 (concatenate/stack/sin/sum/mean),
:139-189 (resample_mean / subsample), :191-219 (roll), :222-225.
"""
from typing import Sequence, Union

import numpy as np


def concatenate(arrays, axis):
  return np.concatenate(arrays, axis=axis)


def stack(arrays, axis):
  return np.stack(arrays, axis=axis)


def sin(x):
  return np.sin(x)


def sum(x, axis=None, **kwargs):  # pylint: disable=redefined-builtin
  return np.sum(x, axis=axis, **kwargs)


def mean(x, axis=None, **kwargs):
  return np.mean(x, axis=axis, **kwargs)


def maximum(x, y):
  return np.maximum(x, y)


def minimum(x, y):
  return np.minimum(x, y)


def where(cond, x, y):
  return np.where(cond, x, y)


def _positive_axis(axis: int, ndim: int) -> int:
  if not -ndim <= axis < ndim:
    raise ValueError('invalid axis {} for ndim {}'.format(axis, ndim))
  return axis % ndim


def _check_divides(size: int, factor: int):
  if size % factor:
    raise ValueError('resample factor {} must divide size {}'
                     .format(factor, size))


def resample_mean(inputs, factor: int, axis: int = -1):
  """Average groups of ``factor`` consecutive samples along ``axis``."""
  inputs = np.asarray(inputs)
  axis = _positive_axis(axis, inputs.ndim)
  _check_divides(inputs.shape[axis], factor)
  blocked = inputs.reshape(inputs.shape[:axis]
                           + (inputs.shape[axis] // factor, factor)
                           + inputs.shape[axis + 1:])
  return blocked.mean(axis=axis + 1)


def subsample(inputs, factor: int, axis: int = -1):
  """Keep every ``factor``-th sample along ``axis`` (starting at index 0)."""
  inputs = np.asarray(inputs)
  axis = _positive_axis(axis, inputs.ndim)
  _check_divides(inputs.shape[axis], factor)
  return np.take(inputs, np.arange(0, inputs.shape[axis], factor), axis=axis)


def roll(tensor, shift: Union[int, Sequence[int]],
         axis: Union[int, Sequence[int]]):
  return np.roll(tensor, shift, axis)


RESAMPLE_FUNCS = {
    'mean': resample_mean,
    'subsample': subsample,
}


def spectral_derivative(x, order: int = 1, period: float = 2 * np.pi):
  """Differentiate along the last axis with a Fourier transform.

  duckarray.py:105-113 (NumPy branch); the rfft form keeps the Nyquist mode
  for odd orders, unlike scipy.fftpack.diff.
  """
  x = np.asarray(x)
  length = x.shape[-1]
  if length % 2:
    raise ValueError('spectral derivative only works for even length data')
  c = 2 * np.pi * 1j / period
  k = np.fft.rfftfreq(length, d=1 / length)
  return np.fft.irfft((c * k) ** order * np.fft.rfft(x))


def smoothing_filter(x, alpha: float = -np.log(1e-15), order: int = 2):
  """Low-pass exponential filter (duckarray.py:116-128)."""
  x = np.asarray(x)
  length = x.shape[-1]
  if length % 2:
    raise ValueError('smoothing filter only works for even length data')
  count = length // 2
  eta = np.arange(count + 1) / count
  sigma = np.exp(-alpha * eta ** (2 * order))
  return np.fft.irfft(sigma * np.fft.rfft(x))

