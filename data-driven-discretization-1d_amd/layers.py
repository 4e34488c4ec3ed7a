"""Periodic 1-D layers on [batch, x, channel] tensors (GPU operators).

Mirror of ``pde_superresolution/layers.py``.  Inside the learned-stencil
kernel the periodic halo is never materialised (each lane computes the wrapped
source index); these standalone operators exist so the reference's unit-test
surface (layers_test.py:49-86) runs against the HIP implementation, and for
``polynomials.reconstruct``.
"""
from . import _lib


def pad_periodic(inputs, padding: int, center: bool = False):
  """layers.py:39-83.  Centred: ceil(p/2) points on the left, floor on the right."""
  if len(inputs.shape) != 3:
    raise ValueError('inputs must be 3D for periodic padding')
  return _lib.pad_periodic(inputs, padding, center)


def _dilate(kernel, dilation_rate: int):
  """[K, cin, cout] -> [(K - 1) d + 1, cin, cout] with zero taps between the K real ones:
  a dilated VALID convolution over the periodically padded input is the plain one with
  this kernel (padding (K - 1) d either way, layers.py:128-129).  The default
  (dilation 1) hands the caller's tensor through untouched -- no host round trip --;
  the zero-stuffed kernel is built on the tensor's own device."""
  if dilation_rate == 1:
    return kernel
  if hasattr(kernel, 'new_zeros'):   # torch tensor, any device
    kernel = kernel.detach()
    k = kernel.shape[0]
    out = kernel.new_zeros(((k - 1) * dilation_rate + 1,) + tuple(kernel.shape[1:]))
    out[::dilation_rate] = kernel
    return out
  import numpy as np
  kernel = np.asarray(kernel)
  k = kernel.shape[0]
  out = np.zeros(((k - 1) * dilation_rate + 1,) + kernel.shape[1:], kernel.dtype)
  out[::dilation_rate] = kernel
  return out


def nn_conv1d_periodic(inputs, filters, stride: int = 1, center: bool = False):
  """layers.py:95-100: VALID cross-correlation after periodic padding; a stride keeps
  every ``stride``-th position of the stride-1 result (tf.nn.conv1d 'VALID':
  ceil(N / stride) outputs, the first at position 0)."""
  if stride < 1:
    raise ValueError('stride must be >= 1')
  out = _lib.conv1d_periodic(inputs, filters, None, center=center)
  return out if stride == 1 else out[:, ::stride].contiguous()


def conv1d_periodic_layer(inputs, kernel, bias=None, activation=None,
                          strides: int = 1, dilation_rate: int = 1,
                          center: bool = False):
  """layers.py:103-137 with explicit weights instead of TF variables.

  ``kernel`` is [kernel_size, in_channels, filters]; ``activation`` one of
  None/'relu'/'relu6'/'tanh'/'softplus'/'elu' (model.py:411-417).
  ``strides`` / ``dilation_rate``: padding = (kernel_size - 1) * dilation_rate, then
  tf.layers.conv1d(padding='valid', strides, dilation_rate): the dilated kernel is the
  zero-stuffed one on the same GPU operator, the stride a subsample of its output
  ([batch, ceil(length / strides), filters]).  Like tf.layers.conv1d, both > 1 at once
  is an error.  (The reference's models never use either: model.py:455-495.)
  """
  if strides < 1 or dilation_rate < 1:
    raise ValueError('strides and dilation_rate must be >= 1')
  if strides > 1 and dilation_rate > 1:
    raise ValueError('strides > 1 not supported in conjunction with dilation_rate > 1')
  out = _lib.conv1d_periodic(inputs, _dilate(kernel, dilation_rate), bias, center=center,
                             activation=activation)
  return out if strides == 1 else out[:, ::strides].contiguous()
