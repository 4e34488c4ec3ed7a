"""ctypes binding of csrc/libddd1d.so (C ABI: include/ddd1d.h).

PyTorch is used only as the device-memory / stream provider: tensors are
allocated with torch, their ``data_ptr()`` and torch's current HIP stream are
handed to the library as plain pointers.  There is NO CPU fallback: if the
shared library or a GPU is missing, calls raise.
"""
import ctypes
import os
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIBRARY_PATH = os.path.join(_HERE, 'csrc', 'libddd1d.so')
PROBE_LIBRARY_PATH = os.path.join(_HERE, 'csrc', 'libddd1d_probe.so')

MAX_DERIVATIVES = 4

# enums (include/ddd1d.h)
ACTIVATIONS = {'relu': 0, 'relu6': 1, 'tanh': 2, 'softplus': 3, 'elu': 4}
MODEL_TARGETS = {'coefficients': 0, 'space_derivatives': 1,
                 'time_derivative': 2, 'flux': 3}
SCHEMES = {'euler': 0, 'midpoint': 1, 'bs3': 2, 'rk23': 2, 'rk4': 3}
KERNELS = {'auto': 0, 'generic': 1, 'mfma': 2, 'mfma64': 3, 'mfma256': 4,
           'mfma64w32': 5, 'mfma64w16': 6}
LAUNCH_MODES = {'persistent': 0, 'per_substep': 1, 'per_step': 2}


class DDDConfig(ctypes.Structure):
  """struct ddd_config."""
  _fields_ = [
      ('struct_size', ctypes.c_int32),
      ('equation', ctypes.c_int32),
      ('num_points', ctypes.c_int32),
      ('num_derivatives', ctypes.c_int32),
      ('derivative_orders', ctypes.c_int32 * MAX_DERIVATIVES),
      ('dx', ctypes.c_double),
      ('period', ctypes.c_double),
      ('eta', ctypes.c_double),
      ('standard_deviation', ctypes.c_double),
      ('stencil_size', ctypes.c_int32),
      ('model_target', ctypes.c_int32),
      ('num_layers', ctypes.c_int32),
      ('filter_size', ctypes.c_int32),
      ('kernel_size', ctypes.c_int32),
      ('activation', ctypes.c_int32),
      ('polynomial_accuracy_order', ctypes.c_int32),
      ('ensure_unbiased_coefficients', ctypes.c_int32),
      ('input_sizes', ctypes.c_int32 * MAX_DERIVATIVES),
      ('weno_reconstruction', ctypes.c_int32),
      ('reserved', ctypes.c_int32 * 3),
  ]


class DDDError(RuntimeError):
  """A libddd1d call returned a non-zero status."""


_lib = None

_F = ctypes.POINTER(ctypes.c_float)
_D = ctypes.POINTER(ctypes.c_double)
_I = ctypes.POINTER(ctypes.c_int32)
_V = ctypes.c_void_p

# name -> (restype, argtypes); every symbol include/ddd1d.h declares.
SIGNATURES = {
    'ddd_model_create': (ctypes.c_int, [ctypes.POINTER(DDDConfig), _F,
                                        ctypes.c_size_t, _F, ctypes.c_size_t,
                                        _F, ctypes.c_size_t,
                                        ctypes.POINTER(_V)]),
    'ddd_baseline_create': (ctypes.c_int, [ctypes.POINTER(DDDConfig), _F,
                                           ctypes.c_size_t,
                                           ctypes.POINTER(_V)]),
    'ddd_spectral_create': (ctypes.c_int, [ctypes.POINTER(DDDConfig), _D,
                                           ctypes.c_size_t,
                                           ctypes.POINTER(_V)]),
    'ddd_model_destroy': (ctypes.c_int, [_V]),
    'ddd_set_forcing': (ctypes.c_int, [_V, ctypes.c_int, ctypes.c_int, _F, _F,
                                       _F, _I, _F, ctypes.c_int]),
    'ddd_clear_forcing': (ctypes.c_int, [_V]),
    'ddd_time_derivative': (ctypes.c_int, [_V, ctypes.c_double, _V, _V,
                                           ctypes.c_int, _V]),
    'ddd_rk_substep': (ctypes.c_int, [_V, ctypes.c_double, _V, _V,
                                      ctypes.c_float, _V, _V, ctypes.c_float,
                                      _V, ctypes.c_int, _V]),
    'ddd_stream_fork': (ctypes.c_int, [_V, _V]),
    'ddd_stream_join': (ctypes.c_int, [_V, _V]),
    'ddd_set_region_mode': (ctypes.c_int, [_V, ctypes.c_int]),
    'ddd_region_stats': (ctypes.c_int, [_V, ctypes.POINTER(ctypes.c_int64),
                                        ctypes.POINTER(ctypes.c_int64)]),
    'ddd_integrate_fixed': (ctypes.c_int, [_V, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_double, ctypes.c_double,
                                           ctypes.c_int, ctypes.c_int, _V, _V,
                                           ctypes.c_int, _V]),
    'ddd_integrate_fixed_f64': (ctypes.c_int, [_V, ctypes.c_int,
                                               ctypes.c_double,
                                               ctypes.c_double, ctypes.c_int,
                                               ctypes.c_int, _V, _V,
                                               ctypes.c_int, _V]),
    'ddd_integrate_adaptive_f64': (ctypes.c_int, [_V, _D, ctypes.c_int,
                                                  ctypes.c_double,
                                                  ctypes.c_double,
                                                  ctypes.c_double,
                                                  ctypes.c_longlong, _V, _V,
                                                  _V, _V, ctypes.c_int, _V]),
    'ddd_circulant_apply_f64': (ctypes.c_int, [_V, _V, _V, ctypes.c_int,
                                               ctypes.c_int, _V]),
    'ddd_time_derivative_f64': (ctypes.c_int, [_V, ctypes.c_double, _V, _V,
                                               ctypes.c_int, _V]),
    'ddd_rk_substep_f64': (ctypes.c_int, [_V, ctypes.c_double, _V, _V,
                                          ctypes.c_double, _V, _V,
                                          ctypes.c_double, _V, ctypes.c_int,
                                          _V]),
    'ddd_space_derivatives': (ctypes.c_int, [_V, _V, _V, ctypes.c_int, _V]),
    'ddd_coefficients': (ctypes.c_int, [_V, _V, _V, ctypes.c_int, _V]),
    'ddd_conv1d_periodic': (ctypes.c_int, [_V, _V, _V, _V, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, _V]),
    'ddd_pad_periodic': (ctypes.c_int, [_V, _V, ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, ctypes.c_int,
                                        ctypes.c_int, _V]),
    'ddd_extract_patches': (ctypes.c_int, [_V, _V, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, _V]),
    'ddd_apply_coefficients': (ctypes.c_int, [_V, _V, _V, ctypes.c_int,
                                              ctypes.c_int, ctypes.c_int,
                                              ctypes.c_int, _V]),
    'ddd_apply_space_derivatives': (ctypes.c_int, [ctypes.c_int, _V, _V, _V,
                                                   ctypes.c_int, ctypes.c_int,
                                                   ctypes.c_int, ctypes.c_double,
                                                   ctypes.c_double, _V]),
    'ddd_polynomial_accuracy_apply': (ctypes.c_int, [_V, _V, _V, _V,
                                                     ctypes.c_int64,
                                                     ctypes.c_int,
                                                     ctypes.c_int, _V]),
    'ddd_set_kernel': (ctypes.c_int, [_V, ctypes.c_int]),
    'ddd_kernel_name': (ctypes.c_char_p, [_V]),
    'ddd_fma_per_point': (ctypes.c_int64, [_V]),
    'ddd_scheme_stages': (ctypes.c_int, [ctypes.c_int]),
    'ddd_selftest_mfma_layout': (ctypes.c_int, []),
    'ddd_abi_version': (ctypes.c_int, []),
    'ddd_last_error': (ctypes.c_char_p, []),
}


def load_library(path: Optional[str] = None):
  """dlopen libddd1d.so and attach prototypes.  Raises if it is missing."""
  global _lib
  if _lib is not None and path is None:
    return _lib
  path = path or LIBRARY_PATH
  # PyTorch-ROCm bundles its own HIP runtime (torch/lib/libamdhip64.so, soname
  # libamdhip64.so.7).  Import torch FIRST so that libddd1d.so binds to that
  # already-loaded runtime: torch owns the device context, allocations and
  # streams this library is handed.  Loading libddd1d.so first would pull in
  # /opt/rocm's copy as a second, separate runtime in the same process.
  import torch  # noqa: F401  pylint: disable=unused-import
  if not os.path.exists(path):
    raise ImportError(
        'HIP library {} not found: build it with '
        '`python -c "import __graft_entry__ as g; g.build()"` (needs hipcc). '
        'There is no CPU fallback for the product path.'.format(path))
  lib = ctypes.CDLL(path)
  for name, (restype, argtypes) in SIGNATURES.items():
    fn = getattr(lib, name)   # AttributeError if a declared symbol is missing
    fn.restype = restype
    fn.argtypes = argtypes
  if lib.ddd_abi_version() != 1:
    raise ImportError('libddd1d ABI version mismatch')
  if ctypes.sizeof(DDDConfig) <= 0:
    raise ImportError('bad DDDConfig')
  _lib = lib
  return lib


def check(status: int):
  if status != 0:
    message = _lib.ddd_last_error().decode('utf-8', 'replace')
    raise DDDError('libddd1d error {}: {}'.format(status, message))


def _torch():
  import torch
  return torch


def require_gpu():
  torch = _torch()
  if not torch.cuda.is_available():
    raise RuntimeError('no HIP device visible: the ddd1d_amd product path needs '
                       'an AMD GPU (MI355X / gfx950); there is no CPU fallback')
  return torch


def current_stream() -> int:
  return _torch().cuda.current_stream().cuda_stream


def as_device(array, dtype=None):
  """NumPy array or torch tensor -> contiguous CUDA tensor (no copy if ok)."""
  torch = require_gpu()
  if isinstance(array, torch.Tensor):
    tensor = array
  else:
    tensor = torch.from_numpy(np.ascontiguousarray(array))
  if dtype is not None and tensor.dtype != dtype:
    tensor = tensor.to(dtype)
  if tensor.device.type != 'cuda':
    tensor = tensor.cuda()
  return tensor.contiguous()


def host_f32(array) -> np.ndarray:
  return np.ascontiguousarray(np.asarray(array, dtype=np.float32))


def fptr(array: np.ndarray):
  return array.ctypes.data_as(_F)


# ---------------------------------------------------------------------------
# standalone operators
# ---------------------------------------------------------------------------
def conv1d_periodic(inputs, filters, bias=None, center=False, activation=None):
  """layers.nn_conv1d_periodic on the GPU; see layers.py."""
  lib = load_library()
  torch = require_gpu()
  x = as_device(inputs, torch.float32)
  w = as_device(filters, torch.float32)
  if x.dim() != 3 or w.dim() != 3 or w.shape[1] != x.shape[2]:
    raise ValueError('expected inputs [batch, x, cin] and filters [k, cin, cout]')
  b = None if bias is None else as_device(bias, torch.float32)
  out = torch.empty((x.shape[0], x.shape[1], w.shape[2]), dtype=torch.float32,
                    device=x.device)
  act = -1 if activation is None else ACTIVATIONS[activation]
  check(lib.ddd_conv1d_periodic(
      x.data_ptr(), w.data_ptr(), None if b is None else b.data_ptr(),
      out.data_ptr(), x.shape[0], x.shape[1], x.shape[2], w.shape[2],
      w.shape[0], int(bool(center)), act, current_stream()))
  return out


def pad_periodic(inputs, padding: int, center: bool = False):
  lib = load_library()
  torch = require_gpu()
  x = as_device(inputs, torch.float32)
  if x.dim() != 3:
    raise ValueError('inputs must be 3D for periodic padding')
  out = torch.empty((x.shape[0], x.shape[1] + padding, x.shape[2]),
                    dtype=torch.float32, device=x.device)
  check(lib.ddd_pad_periodic(x.data_ptr(), out.data_ptr(), x.shape[0],
                             x.shape[1], x.shape[2], int(padding),
                             int(bool(center)), current_stream()))
  return out


def polynomial_accuracy_apply(inputs, nullspace, bias):
  lib = load_library()
  torch = require_gpu()
  x = as_device(inputs, torch.float32)
  ns = as_device(nullspace, torch.float32)
  b = as_device(bias, torch.float32)
  input_size, g = ns.shape
  if x.shape[-1] != input_size:
    raise ValueError('inputs last dimension must equal input_size')
  rows = x.numel() // input_size
  out = torch.empty(x.shape[:-1] + (g,), dtype=torch.float32, device=x.device)
  check(lib.ddd_polynomial_accuracy_apply(
      x.data_ptr(), ns.data_ptr(), b.data_ptr(), out.data_ptr(), rows,
      input_size, g, current_stream()))
  return out


def extract_patches(inputs, size: int):
  """model.extract_patches on the GPU: [batch, x] -> [batch, x, size]."""
  lib = load_library()
  torch = require_gpu()
  x = as_device(inputs, torch.float32)
  if x.dim() != 2:
    raise ValueError('inputs must be [batch, x]')
  out = torch.empty(tuple(x.shape) + (int(size),), dtype=torch.float32, device=x.device)
  check(lib.ddd_extract_patches(x.data_ptr(), out.data_ptr(), x.shape[0], x.shape[1],
                                int(size), current_stream()))
  return out


def apply_coefficients(coefficients, inputs):
  """model.apply_coefficients on the GPU: einsum('bxdi,bxi->bxd') with patches."""
  lib = load_library()
  torch = require_gpu()
  c = as_device(coefficients, torch.float32)
  x = as_device(inputs, torch.float32)
  if c.dim() != 4 or x.dim() != 2 or tuple(c.shape[:2]) != tuple(x.shape):
    raise ValueError('expected coefficients [batch, x, derivative, stencil] and inputs [batch, x]')
  out = torch.empty(tuple(c.shape[:3]), dtype=torch.float32, device=x.device)
  check(lib.ddd_apply_coefficients(c.data_ptr(), x.data_ptr(), out.data_ptr(), x.shape[0],
                                   x.shape[1], c.shape[2], c.shape[3], current_stream()))
  return out


def apply_space_derivatives(equation_id: int, derivatives, inputs, eta: float, dx: float):
  lib = load_library()
  torch = require_gpu()
  d = as_device(derivatives, torch.float32)
  x = as_device(inputs, torch.float32)
  if d.dim() != 3 or x.dim() != 2 or tuple(d.shape[:2]) != tuple(x.shape):
    raise ValueError('expected derivatives [batch, x, derivative] and inputs [batch, x]')
  out = torch.empty_like(x)
  check(lib.ddd_apply_space_derivatives(int(equation_id), d.data_ptr(), x.data_ptr(),
                                        out.data_ptr(), x.shape[0], x.shape[1], d.shape[2],
                                        float(eta), float(dx), current_stream()))
  return out


def circulant_apply(kernel, inputs):
  """out[..., x] = sum_j kernel[(x - j) mod n] inputs[..., j] in float64 on the
  device (ddd_circulant_apply_f64): duckarray.smoothing_filter as a kernel."""
  lib = load_library()
  torch = require_gpu()
  k = as_device(kernel, torch.float64)
  x = as_device(inputs, torch.float64)
  if k.dim() != 1 or x.shape[-1] != k.shape[0]:
    raise ValueError('kernel [n] and inputs [..., n] expected')
  out = torch.empty_like(x)
  rows = x.numel() // k.shape[0]
  check(lib.ddd_circulant_apply_f64(k.data_ptr(), x.data_ptr(), out.data_ptr(), rows,
                                    k.shape[0], current_stream()))
  return out


def load_probe_library():
  """dlopen libddd1d_probe.so (the -DDDD_PROBES flavour: ddd_debug_* entry points,
  phase tracing, hardware probes) INSTEAD of the product library.  Only
  profiles/tools/ and `bench.py --debug-option` do this, before any other call."""
  if not os.path.exists(PROBE_LIBRARY_PATH):
    raise ImportError(
        '{} not found: build it with `python -c "import __graft_entry__ as g; '
        'g.build_probe()"`'.format(PROBE_LIBRARY_PATH))
  return load_library(PROBE_LIBRARY_PATH)


def debug_set_option(name: str, value: int):
  """Profiling / A-B switches (capi.hip: ddd_debug_set_option); every change is
  logged to stderr by the library.  They exist in libddd1d_probe.so only: the
  product library has no such entry point (load_probe_library first)."""
  lib = load_library()
  if not hasattr(lib, 'ddd_debug_set_option'):
    raise DDDError('the product library has no debug switches: call '
                   '_lib.load_probe_library() (libddd1d_probe.so) before anything else')
  fn = lib.ddd_debug_set_option
  fn.restype = ctypes.c_int
  fn.argtypes = [ctypes.c_char_p, ctypes.c_longlong]
  check(fn(name.encode('utf-8'), int(value)))


def selftest_mfma_layout():
  lib = load_library()
  require_gpu()
  check(lib.ddd_selftest_mfma_layout())
