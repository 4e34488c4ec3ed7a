"""Exact (spectral, float64) solver on the device: throughput of the batched
adaptive RK23 solve and the cost of ONE right-hand-side evaluation of the
spectral kernel (O(N^2) circulant products below 512 points, the in-LDS FFT of
round 5 from 512 up) against an FFT evaluation through rocFFT
(torch.fft.rfft / irfft on the same device, same batch), N = 64 ... 2048.

  python profiles/tools/spectral_exact_bench.py > profiles/r5_spectral_exact.txt
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ddd1d_amd import equations, integrate, model as model_lib   # noqa: E402


def timed(fn, reps):
  fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    fn()
  e1.record()
  torch.cuda.synchronize()
  return e0.elapsed_time(e1) / reps   # ms


def fft_rhs(cls, y, period, eta=0.04):
  """The same right-hand side through rocFFT: derivatives by rfft -> (ik)^order -> irfft."""
  n = y.shape[-1]
  k = 2j * np.pi / period * torch.arange(n // 2 + 1, device=y.device, dtype=torch.float64)
  yh = torch.fft.rfft(y)
  d = lambda order: torch.fft.irfft(k ** order * yh, n=n)
  if cls is equations.KdVEquation:
    return (-6.0 * y) * d(1) - d(3)
  if cls is equations.KSEquation:
    return (-y * d(1) - d(4)) - d(2)
  return eta * d(2) - y * d(1)


print('# one right-hand-side evaluation, float64, batch 1024: spectral kernel (mode) vs rocFFT')
print('# equation N mode  ours_us  rocfft_us  ratio(ours/rocfft)  max_rel_diff')
for cls in (equations.KdVEquation, equations.KSEquation):
  for n in (64, 128, 256, 512, 1024, 2048):
    eq = cls(n, random_seed=0)
    model = model_lib.SpectralModel(eq)
    y = torch.from_numpy(np.tile(eq.initial_value(), (1024, 1))).cuda()
    ours = timed(lambda: model.time_derivative(y), 20) * 1e3
    fft = timed(lambda: fft_rhs(cls, y, eq.grid.period), 20) * 1e3
    diff = (model.time_derivative(y) - fft_rhs(cls, y, eq.grid.period)).abs().max().item()
    scale = model.time_derivative(y).abs().max().item()
    mode = 'fft' if n >= 512 and n & (n - 1) == 0 else 'circulant'
    print('{} {:5d} {:9s} {:10.1f} {:10.1f} {:6.2f} {:.1e}'.format(
        cls.__name__, n, mode, ours, fft, ours / fft, diff / scale))

print('# integrate_exact_batch: SciPy-RK23 semantics per sample, one launch per segment')
print('# config samples nfev_min nfev_max wall_s grid_point_evals_per_s')
for cls, n, samples, t_end in ((equations.KSEquation, 256, 256, 0.05),
                               (equations.KdVEquation, 512, 256, 0.05),
                               (equations.KSEquation, 256, 2048, 0.05)):
  eqs = [cls(n, random_seed=s) for s in range(samples)]
  times = np.linspace(0, t_end, 3)
  integrate.integrate_exact_batch(eqs[:8], times=times)   # build + warm
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  ds = integrate.integrate_exact_batch(eqs, times=times)
  torch.cuda.synchronize()
  wall = time.perf_counter() - t0
  nfev = np.asarray(ds.coords['num_evals'][1] if isinstance(ds.coords['num_evals'], tuple)
                    else ds.coords['num_evals'])
  print('{} N={} {} {} {} {:.3f} {:.3e}'.format(cls.__name__, n, samples, nfev.min(), nfev.max(),
                                              wall, nfev.sum() * n / wall))
# the WENO5 + Godunov exact Burgers solver (float32 generic kernel, per-seed forcing)
for n, samples, t_end in ((512, 256, 0.5), (512, 2048, 0.5)):
  eqs = [equations.BurgersEquation(n, random_seed=s) for s in range(samples)]
  times = np.linspace(0, t_end, 3)
  integrate.integrate_exact_batch(eqs[:8], times=times)
  torch.cuda.synchronize()
  t0 = time.perf_counter()
  ds = integrate.integrate_exact_batch(eqs, times=times)
  torch.cuda.synchronize()
  wall = time.perf_counter() - t0
  nfev = np.asarray(ds.coords['num_evals'][1] if isinstance(ds.coords['num_evals'], tuple)
                    else ds.coords['num_evals'])
  print('BurgersEquation(WENO) N={} {} {} {} {:.3f} {:.3e}'.format(n, samples, nfev.min(), nfev.max(),
                                                            wall, nfev.sum() * n / wall))
# the reference's own execution shape for one of these samples: host SciPy + HIP RHS
eq = equations.KSEquation(256, random_seed=0)
t0 = time.perf_counter()
one = integrate.integrate_exact(eq, times=np.linspace(0, 0.05, 3))
wall = time.perf_counter() - t0
nf = int(np.asarray(one.coords['num_evals']))
print('# per-sample host SciPy loop over the same kernel (reference execution shape): '
      'KS N=256 nfev {} in {:.2f} s = {:.3e} grid-point-evals/s'.format(nf, wall, nf * 256 / wall))
