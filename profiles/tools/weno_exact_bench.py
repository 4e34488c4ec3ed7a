"""The WENO5 + Godunov-flux exact Burgers solver (integrate.WENODifferentiator,
integrate.py:124-140; what scripts/create_exact_data.py maps over seeds) on the device:
grid-point-evaluations/s of the batched SciPy-RK23 solve, rhs_weno.h (one wavefront per
sample) against the generic kernel it replaces (one workgroup per sample).

  python profiles/tools/weno_exact_bench.py > profiles/r6_weno_exact.txt
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from ddd1d_amd import equations, integrate, model as model_lib   # noqa: E402


def kernel_rate(model, y0, times, reps=3):
  """nfev * N / kernel time of ddd_integrate_adaptive_f64 alone (HIP events on the launch stream)."""
  model.integrate_adaptive(y0, times)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  e0.record()
  for _ in range(reps):
    y, nfev, status = model.integrate_adaptive(y0, times)
  e1.record()
  torch.cuda.synchronize()
  ms = e0.elapsed_time(e1) / reps
  nfev = nfev.cpu().numpy()
  assert (status.cpu().numpy() == 0).all() and torch.isfinite(y).all()
  return nfev, ms


print('# ddd_integrate_adaptive_f64, forced Burgers (WENO5 + Godunov flux), t in [0, 0.5], per-seed forcing')
print('# N samples kernel nfev_min nfev_max kernel_ms grid_point_evals_per_s')
for n in (512, 256, 128):
  for samples in (256, 2048, 8192):
    eqs = [equations.GodunovBurgersEquation(n, random_seed=s) for s in range(samples)]
    frc = model_lib.forcing_from_equations(eqs)
    rs = np.random.RandomState(0)
    x = eqs[0].grid.solution_x
    y0 = np.stack([0.5 * np.sin(x + rs.uniform(0, 6.28)) + 0.2 * np.sin(2 * x + rs.uniform(0, 6.28))
                   for _ in range(samples)])
    times = np.linspace(0, 0.5, 3)
    for kernel in ('auto', 'generic'):
      if kernel == 'generic' and samples == 8192:
        continue
      model = model_lib.BaselineModel(eqs[0], 3, weno=True)
      model.set_kernel(kernel)
      model.set_forcing(frc)
      nfev, ms = kernel_rate(model, y0, times)
      print('{} {} {} {} {} {:.3f} {:.3e}'.format(n, samples, model.kernel_name, nfev.min(), nfev.max(),
                                                  ms, nfev.sum() * n / (ms * 1e-3)))
      sys.stdout.flush()

print('# integrate_exact_batch (host driver + launches, wall clock; round 5 measured 3.38e9 / 4.77e9 here)')
print('# config samples nfev_min nfev_max wall_s grid_point_evals_per_s')
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
