"""Where the adaptive RK23 integrator's distance to the fixed-step kernel comes from
(VERDICT r5 item 1a): the same model and ensemble through
  (1) the fixed-step persistent kernel, midpoint, float32 state   (the headline shape),
  (2) fixed-step Bogacki-Shampine (3 evaluations per step: the adaptive integrator's stages), float32,
  (3) the same with float64 state                                  (SciPy's state type),
  (4) ddd_integrate_adaptive_f64                                    (+ controller, error norm, dense output),
all as fractions of the f32 peak in useful evaluations (2 x fma_per_point FLOP each).
(2) -> (3): float64 state and stage combinations; (3) -> (4): the controller and, for ensembles
of few workgroup rounds, the spread of the per-sample evaluation counts.

  python profiles/tools/adaptive_gap.py > profiles/r6_adaptive_gap.txt
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench   # noqa: E402  (build_workload)

PEAK = 157.3


def timed(fn, min_ms=400.0):
  fn()
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  total, reps, out = 0.0, 0, None
  while total < min_ms:
    e0.record()
    out = fn()
    e1.record()
    torch.cuda.synchronize()
    total += e0.elapsed_time(e1)
    reps += 1
  return total / reps, out


print('# equation N batch | fraction of 157.3 TFLOP/s in useful evaluations')
print('# config fixed_midpoint_f32 fixed_bs3_f32 fixed_bs3_f64 adaptive_rk23 (nfev min/mean/max) adaptive_issued')
for equation, n, batch, unique, t_end, steps in (('burgers', 64, 4096, None, 1.0, 300),
                                                ('burgers', 64, 1024, None, 1.0, 300),
                                                ('kdv', 64, 4096, 1024, 0.2, 300),
                                                ('ks', 256, 8192, 256, 0.02, 100),
                                                ('ks', 256, 1024, 256, 0.02, 100)):
  args = argparse.Namespace(equation=equation, num_points=n, non_conservative=False, baseline_stencils=False,
                            kernel='auto', hparams='{}')
  eq, model, _, y0 = bench.build_workload(args, 0, batch, unique=unique)
  fma = model.fma_per_point
  dt = eq.time_step
  row = []
  for scheme, dtype, stages in (('midpoint', 'float32', 2), ('bs3', 'float32', 3), ('bs3', 'float64', 3)):
    y0d = torch.from_numpy(y0.astype(np.float32 if dtype == 'float32' else np.float64)).cuda()
    ms, _ = timed(lambda: model.integrate_fixed(y0d, steps, dt=dt, scheme=scheme, save_every=steps,
                                                state_dtype=dtype))
    row.append(2.0 * fma * batch * n * stages * steps / (ms * 1e-3) / 1e12 / PEAK)
  y0d = torch.from_numpy(y0.astype(np.float64)).cuda()
  times = np.linspace(0.0, t_end, 11)
  ms, (y, nfev, status) = timed(lambda: model.integrate_adaptive(y0d, times))
  nfev = nfev.cpu().numpy().astype(np.int64)
  frac = 2.0 * fma * float(nfev.sum()) * n / (ms * 1e-3) / 1e12 / PEAK
  spg = max(1, (64 if n <= 64 else 256) // n)
  issued = float(np.pad(nfev, (0, (-len(nfev)) % spg)).reshape(-1, spg).max(axis=1).sum()) * spg
  print('{} N={} B={} {:.3f} {:.3f} {:.3f} {:.3f} ({}/{:.0f}/{}) {:.3f}  [{}]'.format(
      equation, n, batch, row[0], row[1], row[2], frac, nfev.min(), nfev.mean(), nfev.max(),
      frac * issued / float(nfev.sum()), model.kernel_name))
  sys.stdout.flush()
  model.close()
