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


def nn_conv1d_periodic(inputs, filters, stride: int = 1, center: bool = False):
  """layers.py:95-100: VALID cross-correlation after periodic padding."""
  if stride != 1:
    raise NotImplementedError('only stride 1 is used on the integration path')
  return _lib.conv1d_periodic(inputs, filters, None, center=center)


def conv1d_periodic_layer(inputs, kernel, bias=None, activation=None,
                          strides: int = 1, dilation_rate: int = 1,
                          center: bool = False):
  """layers.py:103-137 with explicit weights instead of TF variables.

  ``kernel`` is [kernel_size, in_channels, filters]; ``activation`` one of
  None/'relu'/'relu6'/'tanh'/'softplus'/'elu' (model.py:411-417).
  """
  if strides != 1 or dilation_rate != 1:
    raise NotImplementedError('strides/dilation other than 1 are never used by '
                              'the reference models (model.py:455-495)')
  return _lib.conv1d_periodic(inputs, kernel, bias, center=center,
                              activation=activation)
