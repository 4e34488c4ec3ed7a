#!/usr/bin/env python
"""Headline benchmark: grid-point-steps/s of the learned-stencil integration path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): Burgers N=64, learned conv-net stencils
(3 layers x 32 filters x kernel 5, conservative form, polynomial accuracy
order 1), batch = 1024 random-phase initial conditions PER GPU with per-sample
random forcing, fixed-step midpoint rule at the equation's time step (the
reference's batched integrator, model.integrate_ode, model.py:138-159).
A "step" is one full Runge-Kutta step of the whole batch; one grid-point-step =
one grid point advanced by one step.  Weights are
synthetic (Glorot, seeded); inputs are resident in HBM before timing starts.

With N > 1 every rank integrates its own shard of the ensemble (weak scaling,
no communication during stepping) and the final states are all-gathered over
RCCL inside the timed region (BASELINE.json config 5's "final gather").

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for the roofline and
cpu_baseline definitions.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32 MFMA = f32 vector peak
PEAK_HBM_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E spec peak


def parse_args():
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=1000)
  ap.add_argument('--warmup', type=int, default=100)
  ap.add_argument('--batch', type=int, default=1024, help='samples per GPU')
  ap.add_argument('--num-points', type=int, default=64)
  ap.add_argument('--equation', default='burgers')
  ap.add_argument('--scheme', default='midpoint',
                  choices=['euler', 'midpoint', 'bs3', 'rk4'])
  ap.add_argument('--launch-mode', default='persistent',
                  choices=['persistent', 'per_substep'])
  ap.add_argument('--non-conservative', action='store_true')
  ap.add_argument('--baseline-stencils', action='store_true',
                  help='fixed polynomial stencils instead of the conv net')
  ap.add_argument('--kernel', default='auto', choices=['auto', 'mfma', 'mfma64', 'mfma64w32', 'mfma256', 'generic'])
  ap.add_argument('--cpu-seconds', type=float, default=12.0,
                  help='budget for the CPU baseline sample (0 disables)')
  return ap.parse_args()


def build_workload(args, rank):
  import ddd1d_amd
  from ddd1d_amd import equations, model as model_lib
  rf = 8
  hp = ddd1d_amd.create_hparams(
      args.equation, conservative=not args.non_conservative,
      resample_factor=rf,
      equation_kwargs=json.dumps({'num_points': args.num_points * rf}))
  _, eq = equations.from_hparams(hp, random_seed=0)
  if args.baseline_stencils:
    model = model_lib.BaselineModel(eq, accuracy_order=1)
  else:
    model = model_lib.LearnedStencilModel(eq, hp, init_seed=0, output_scale=0.1)
  model.set_kernel(args.kernel)
  # sample ids are global: rank r owns ids [r*batch, (r+1)*batch)
  from ddd1d_amd import distributed
  seeds = distributed.weak_shard_ids(args.batch, rank)
  forcing = model_lib.batched_forcing_parameters(seeds, nparams=20)
  model.set_forcing(forcing)
  ic = model_lib.batched_forcing_parameters(
      [s + (1 << 20) for s in seeds], nparams=10)
  x = eq.grid.reference_x
  waves = np.sum(ic['a'][..., None] * np.sin(
      2 * np.pi * ic['k'][..., None] * x / eq.grid.period + ic['phi'][..., None]),
                 axis=1)
  y0 = eq.grid.resample(waves).astype(np.float32)
  return eq, model, forcing, y0


def cpu_baseline(model, forcing, y0, scheme, dt, budget_s):
  """Time the CPU port (oracle/ddd_oracle.c, OpenMP over samples) on a bounded
  sample of the same workload: same model, scheme and per-sample forcing."""
  sys.path.insert(0, os.path.join(ROOT, 'oracle'))
  import c_oracle   # the timed baseline leg; never the product
  import oracle
  scheme_id = {'euler': oracle.SCHEME_EULER, 'midpoint': oracle.SCHEME_MIDPOINT,
               'bs3': oracle.SCHEME_BS3, 'rk4': oracle.SCHEME_RK4}[scheme]
  co = c_oracle.COracle(model.spec(), nparams=forcing['a'].shape[1])
  threads = co.num_threads
  sample = min(y0.shape[0], max(threads, 64))
  sample -= sample % threads if sample >= threads else 0
  frc = {k: v[:sample] for k, v in forcing.items()}
  ys = y0[:sample]
  co.integrate_fixed(scheme_id, 0.0, dt, 2, ys, frc)    # thread pool warm-up
  steps, elapsed, chunk, state = 0, 0.0, 8, ys
  while elapsed < budget_s and steps < 200000:
    t0 = time.perf_counter()
    state = co.integrate_fixed(scheme_id, steps * dt, dt, chunk, state, frc)
    elapsed += time.perf_counter() - t0
    steps += chunk
    chunk = min(chunk * 2, 4096)
  points = sample * ys.shape[1] * steps
  # reference-style execution shape: ONE sample,
  # SciPy RK23 (max_step 0.01) driving the NumPy restatement of the RHS, one
  # core -- how the reference itself runs (integrate.py:143-169)
  one_forcing = {k: v[0] for k, v in forcing.items()}
  t_end = 0.25
  t0 = time.perf_counter()
  _, nfev = oracle.odeint_rk23(model.spec(), y0[0], np.array([0.0, t_end]), one_forcing)
  ref_elapsed = time.perf_counter() - t0
  ref_steps = max((nfev - 2) // 3, 1)          # RK23: 3 evaluations per step (FSAL)
  reference_style = {
      'value': ys.shape[1] * ref_steps / ref_elapsed, 'unit': 'grid-point-steps/s',
      'cores': 1,
      'sample': 'one sample, scipy.integrate.solve_ivp RK23 max_step 0.01 to t = {} '
                '({} evaluations in {:.2f} s) over the NumPy restatement of the '
                'right-hand side: the reference\'s own execution shape'
                .format(t_end, nfev, ref_elapsed)}
  return {
      'value': points / elapsed, 'unit': 'grid-point-steps/s', 'cores': threads,
      'kind': 'port',
      'sample': 'C/OpenMP float32 restatement of the reference path '
                '(oracle/ddd_oracle.c: conv tower, projection, stencil apply, '
                'forcing on the reference grid, same RK scheme), batch {} x {} '
                'steps in {:.1f} s on {} threads; host has {} logical cores'
                .format(sample, steps, elapsed, threads, os.cpu_count()),
      'reference_style': reference_style,
  }


def main():
  args = parse_args()
  import torch
  import torch.distributed as dist

  world = int(os.environ.get('WORLD_SIZE', '1'))
  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  if world != args.gpus:
    if world == 1 and args.gpus > 1:
      raise SystemExit('launch with torch.distributed.run --nproc-per-node {}'
                       .format(args.gpus))
    raise SystemExit('--gpus {} but WORLD_SIZE {}'.format(args.gpus, world))
  torch.cuda.set_device(local_rank)
  if world > 1:
    dist.init_process_group('nccl', device_id=torch.device('cuda', local_rank))

  import ddd1d_amd
  ddd1d_amd._lib.load_library()   # raises if the HIP extension is missing
  eq, model, forcing, y0_host = build_workload(args, rank)
  lib = ddd1d_amd._lib.load_library()
  stages = lib.ddd_scheme_stages(ddd1d_amd._lib.SCHEMES[args.scheme])
  dt = eq.time_step
  n = eq.grid.solution_num_points
  batch = args.batch

  y0 = torch.from_numpy(y0_host).cuda()
  final = torch.empty((1, batch, n), dtype=torch.float32, device='cuda')
  gathered = (torch.empty((world, batch, n), dtype=torch.float32, device='cuda')
              if world > 1 else None)

  def run(num_steps, t0):
    model.integrate_fixed(y0, num_steps, dt=dt, t0=t0, scheme=args.scheme,
                          save_every=num_steps, launch_mode=args.launch_mode,
                          out=final)
    if world > 1:
      dist.all_gather_into_tensor(gathered, final[0])

  def barrier():
    if world > 1:
      dist.barrier()
    torch.cuda.synchronize()

  if args.warmup > 0:
    run(args.warmup, 0.0)
  barrier()
  start_evt = torch.cuda.Event(enable_timing=True)
  stop_evt = torch.cuda.Event(enable_timing=True)
  wall0 = time.perf_counter()
  start_evt.record()
  model.integrate_fixed(y0, args.steps, dt=dt, t0=0.0, scheme=args.scheme,
                        save_every=args.steps, launch_mode=args.launch_mode,
                        out=final)
  stop_evt.record()
  if world > 1:
    dist.all_gather_into_tensor(gathered, final[0])
  barrier()
  wall = time.perf_counter() - wall0
  kernel_ms = start_evt.elapsed_time(stop_evt)   # events on the launch stream

  if world > 1:
    t = torch.tensor([wall, kernel_ms], dtype=torch.float64, device='cuda')
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    wall, kernel_ms = float(t[0]), float(t[1])

  finite = bool(torch.isfinite(final).all())
  if rank == 0:
    points_per_gpu = batch * n * args.steps
    total_points = points_per_gpu * world
    fma = model.fma_per_point
    launches = 1 if args.launch_mode == 'persistent' else args.steps * stages
    flops_per_launch = 2.0 * fma * batch * n * stages * args.steps / launches
    bytes_per_launch = (8.0 * batch * n if args.launch_mode == 'persistent' else
                        (20.0 if args.scheme == 'midpoint' else 8.0 * stages)
                        * batch * n / stages)
    launch_s = kernel_ms * 1e-3 / launches
    achieved_tflops = flops_per_launch / launch_s / 1e12
    achieved_gbps = bytes_per_launch / launch_s / 1e9
    compute_bound = not args.baseline_stencils
    traffic = measured_traffic(type(eq).__name__, n, batch, args.launch_mode,
                               args.baseline_stencils)
    result = {
        'metric': 'grid-point-steps/s',
        'value': total_points / wall,
        'unit': 'grid-point-steps/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'ms_per_step': wall * 1e3 / args.steps,
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': 'f32',
        'data': 'synthetic',
        'config': {
            'workload': '{} N={} {} learned-stencil ensemble, batch {}/GPU, '
                        '{} steps, {} dt={:g}'.format(
                            args.equation, n,
                            'fixed-stencil' if args.baseline_stencils else 'conv-net',
                            batch, args.steps, args.scheme, dt),
            'equation': type(eq).__name__, 'num_points': n,
            'batch_per_gpu': batch, 'global_batch': batch * world,
            'scheme': args.scheme, 'stages': stages, 'dt': dt,
            'launch_mode': args.launch_mode, 'kernel': model.kernel_name,
            'fma_per_point_eval': fma, 'parallelism': 'ensemble-shard x{}'.format(world),
            'finite': finite,
        },
        'roofline': {
            'bound': 'mfma' if compute_bound else 'hbm',
            'achieved': achieved_tflops if compute_bound else achieved_gbps,
            'peak': PEAK_FP32_TFLOPS if compute_bound else PEAK_HBM_GBPS,
            'unit': 'TFLOP/s' if compute_bound else 'GB/s',
            'frac': (achieved_tflops / PEAK_FP32_TFLOPS if compute_bound
                     else achieved_gbps / PEAK_HBM_GBPS),
            'traffic': traffic,
            'kernel_ms_per_launch': kernel_ms / launches,
            'launches': launches,
            'hbm_gbps': achieved_gbps,
            'hbm_frac': achieved_gbps / PEAK_HBM_GBPS,
            'fp32_tflops': achieved_tflops,
            'fp32_frac': achieved_tflops / PEAK_FP32_TFLOPS,
        },
    }
    if world == 1 and args.cpu_seconds > 0:
      result['cpu_baseline'] = cpu_baseline(model, forcing, y0_host, args.scheme,
                                            dt, args.cpu_seconds)
    else:
      result['cpu_baseline'] = None
    print(json.dumps(result))
  if world > 1:
    dist.destroy_process_group()


def measured_traffic(equation, num_points, batch, launch_mode, fixed):
  """HBM bytes per launch of the dominant kernel from the committed rocprofv3
  PMC passes (FETCH_SIZE / WRITE_SIZE collected separately, gfx950 correction
  applied; profiles/r1_hbm_traffic.json), or None when this configuration was
  not profiled.  bench.py cannot collect counters itself."""
  path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles',
                      'r1_hbm_traffic.json')
  try:
    with open(path) as f:
      table = json.load(f)
  except (OSError, ValueError):
    return None
  want = dict(equation=equation, num_points=num_points, batch_per_gpu=batch,
              launch_mode=launch_mode, fixed=bool(fixed))
  for entry in table.get('entries', []):
    if entry.get('match') == want:
      return entry['traffic_bytes_per_launch']
  return None


if __name__ == '__main__':
  main()
