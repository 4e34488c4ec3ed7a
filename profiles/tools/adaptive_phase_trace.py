"""Phase trace of mfma::adaptive_kernel: where an attempt of the
production integrator spends its time -- inputs / first-stage forcing sums, the evaluation,
the controller arithmetic (error norm, error test, dense output), publish + workgroup vote --
by phase of the attempt (0 f(t0), 1 the probe of select_initial_step, 2 / 3 / 4 = stages
2, 3 and the FSAL stage), for the headline model's one-wavefront kernel and for KS N = 256 on
four-wave groups.  libddd1d_probe.so only (csrc/rhs_adaptive_trace.h: s_memtime stamps of
thread 0 of every workgroup, first 48 loop iterations).

  python -c "import __graft_entry__ as g; g.build_probe()"
  python profiles/tools/adaptive_phase_trace.py > profiles/r6_adaptive_phase_trace.txt
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ddd1d_amd   # noqa: E402
ddd1d_amd._lib.load_probe_library()
import bench   # noqa: E402

SLOTS = 256
NAMES = ('inputs + first-stage sums (0-1)', 'evaluation (1-2)', 'controller arithmetic (2-3)',
         'publish + vote (3-4)', 'loop back (4-0)')


def trace(model, y0, times, rows):
  lib = ddd1d_amd._lib.load_library()
  batch, n = y0.shape
  spg = max(1, rows // n)
  blocks = (batch + spg - 1) // spg
  y0d = torch.from_numpy(y0.astype(np.float64)).cuda()
  times = np.ascontiguousarray(times, dtype=np.float64)
  out = torch.empty((times.size, batch, n), dtype=torch.float64, device='cuda')
  head = (batch + 1) & ~1
  nfev = torch.zeros(head + 2 * blocks * SLOTS, dtype=torch.int32, device='cuda')
  status = torch.zeros(batch, dtype=torch.int32, device='cuda')
  for rep in range(2):   # second launch: warm
    nfev.zero_()
    ddd1d_amd._lib.check(lib.ddd_integrate_adaptive_f64(
        model._handle, times.ctypes.data_as(ddd1d_amd._lib._D), int(times.size), 1e-3, 1e-6, 0.01,
        ctypes.c_longlong((1 << 62) | 10 ** 9), y0d.data_ptr(), out.data_ptr(), nfev.data_ptr(),
        status.data_ptr(), batch, ddd1d_amd._lib.current_stream()))
    torch.cuda.synchronize()
  raw = nfev[head:].cpu().numpy().view(np.int64).reshape(blocks, SLOTS)[:, :240].reshape(blocks, 48, 5)
  return raw >> 3, raw[:, :, 0] & 7, nfev[:batch].cpu().numpy()


for equation, n, batch, unique, t_end, rows in (('burgers', 64, 4096, None, 1.0, 64),
                                               ('burgers', 64, 1024, None, 1.0, 64),
                                               ('ks', 256, 1024, 256, 0.02, 256),
                                               ('ks', 256, 8192, 256, 0.02, 256)):
  args = argparse.Namespace(equation=equation, num_points=n, non_conservative=False, baseline_stencils=False,
                            kernel='mfma64' if rows == 64 else 'auto', hparams='{}')
  eq, model, _, y0 = bench.build_workload(args, 0, batch, unique=unique)
  stamps, phase, nfev = trace(model, y0, np.linspace(0.0, t_end, 11), rows)
  good = (stamps[:, :, 0] > 0) & (stamps[:, :, 4] > 0)
  good[:, 0] = False          # (the first iteration follows the launch setup)
  d = np.diff(stamps, axis=2).astype(np.float64)                      # [blocks, 48, 4]
  back = np.zeros(stamps.shape[:2]); back[:, :-1] = stamps[:, 1:, 0] - stamps[:, :-1, 4]
  allp = np.concatenate([d, back[..., None]], axis=2)
  print('## {} N={} batch {} ({}-row groups, kernel {}; nfev {}..{}): s_memtime ticks (100 MHz x shader-clock ratio; '
        'one evaluation alone = the "evaluation" column), mean over workgroups'.format(
            equation, n, batch, rows, model.kernel_name, nfev.min(), nfev.max()))
  print('%-8s %8s' % ('phase', 'count') + ''.join(' %34s' % name for name in NAMES) + ' %10s' % 'iteration')
  for ph in range(5):
    sel = good & (phase == ph)
    sel[:, -1] = False
    if not sel.any():
      continue
    row = [allp[..., k][sel].mean() for k in range(5)]
    print('%-8d %8d' % (ph, sel.sum()) + ''.join(' %34.0f' % v for v in row) + ' %10.0f' % sum(row))
  sel = good & (phase >= 2); sel[:, -1] = False
  per_attempt = 3 * allp[sel].sum(axis=1).mean()
  evals = 3 * allp[..., 1][sel].mean()
  print('one attempt (stages 2 + 3 + 4): %.0f ticks, of which evaluations %.0f (%.1f %%), everything else %.0f (%.1f %%)\n' % (
      per_attempt, evals, 100 * evals / per_attempt, per_attempt - evals, 100 * (1 - evals / per_attempt)))
  sys.stdout.flush()
  model.close()
