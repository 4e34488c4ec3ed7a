"""Small ensembles of the headline model (Burgers N=64, conv-net stencils, midpoint): one
64-row wavefront per sample (mfma64), two 32-row wavefronts (mfma64w32, rhs_mfma.h kSplit)
and four 16-row wavefronts on 16x16x4 MFMAs (mfma64w16, kQuad), by ensemble size.

  python profiles/tools/small_ensemble_ab.py > profiles/r6_small_ensembles.txt
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import ddd1d_amd   # noqa: E402
from ddd1d_amd import equations, model as model_lib   # noqa: E402

PEAK = 157.3
hp = ddd1d_amd.create_hparams('burgers', conservative=True, resample_factor=8,
                              equation_kwargs=json.dumps({'num_points': 512}))
_, eq = equations.from_hparams(hp, random_seed=0)
model = model_lib.LearnedStencilModel(eq, hp, init_seed=0, output_scale=0.1)
steps = 1000
print('# Burgers N=64 learned stencils, midpoint, {} steps per launch, persistent launch'.format(steps))
print('# batch kernel(requested) kernel(ran) ms_per_launch grid_point_steps_per_s frac_of_157.3TF')
for batch in (64, 128, 256, 384, 512, 768, 1024, 2048):
  forcing = model_lib.batched_forcing_parameters(range(batch), nparams=20)
  model.set_forcing(forcing)
  ic = model_lib.batched_forcing_parameters(range(1 << 20, (1 << 20) + batch), nparams=10)
  x = eq.grid.reference_x
  y0 = eq.grid.resample(np.sum(ic['a'][..., None] * np.sin(
      2 * np.pi * ic['k'][..., None] * x / eq.grid.period + ic['phi'][..., None]), axis=1)).astype(np.float32)
  y0 = torch.from_numpy(y0).cuda()
  ref = None
  for kernel in ('mfma64', 'mfma64w32', 'mfma64w16', 'auto'):
    model.set_kernel(kernel)
    out = model.integrate_fixed(y0, steps, dt=1e-3, save_every=steps)
    torch.cuda.synchronize()
    reps = 5
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
      out = model.integrate_fixed(y0, steps, dt=1e-3, save_every=steps)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    rate = batch * 64 * steps / (ms * 1e-3)
    frac = rate * 2 * 2 * model.fma_per_point / 1e12 / PEAK
    same = '' if ref is None else (' bits==mfma64' if torch.equal(out, ref) else ' BITS DIFFER')
    if ref is None:
      ref = out.clone()
    print('{} {} {} {:.3f} {:.3e} {:.3f}{}'.format(batch, kernel, model.kernel_name, ms, rate, frac, same))
    sys.stdout.flush()
