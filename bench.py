#!/usr/bin/env python
"""Headline benchmark: grid-point-steps/s of the learned-stencil integration path.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N \
        --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W
    (`python bench.py --gpus N` without a torchrun environment re-launches
    itself under torch.distributed.run on 127.0.0.1)

Workload = BASELINE.json's north_star target configuration: Burgers N=64,
learned conv-net stencils (3 layers x 32 filters x kernel 5, conservative
form, polynomial accuracy order 1), batch = 4096 random-phase initial
conditions PER GPU with per-sample random forcing, fixed-step midpoint rule at
the equation's time step (the reference's batched integrator,
model.integrate_ode, model.py:138-159).  BASELINE.json configs[1] (the same
model at batch 1024) is measured in the same run and reported under
"secondary".  A "step" is one full Runge-Kutta step of the whole batch; one
grid-point-step = one grid point advanced by one step.
Weights are synthetic (Glorot, seeded); inputs are resident in HBM before
timing starts.

Timing protocol (every rank):
  1. pre-heat: the SAME kernel runs back to back for >= --preheat-ms (default
     300 ms), independent of --warmup, so that a short timed region does not
     run at ramping clocks / with a cold code object;
  2. --warmup W untimed steps (one launch);
  3. barrier + synchronize; the timed region is R back-to-back jobs of exactly
     --steps K steps each (every job: one persistent launch from the same y0 +,
     for N > 1, the RCCL all-gather of the final states, which runs on RCCL's
     stream and overlaps the next job's kernel; every gather is complete before
     the closing barrier); R = 1 when one job
     lasts >= --min-timed-ms (default 40 ms), otherwise the smallest R that
     fills it.  R is reported as "reps"; "ms_per_step" and "value" are means
     over the R*K timed steps, "roofline.kernel_ms_per_launch" the HIP-event
     time over the R launches (one event pair on the launch stream around all of
     them) divided by their number; "roofline.frac_wall" the same fraction from
     the wall clock;
  4. synchronize + barrier; wall = max over ranks.

With N > 1 every rank integrates its own shard of the ensemble (weak scaling,
no communication during stepping) and the final states are all-gathered over
RCCL inside the timed region (BASELINE.json config 5's "final gather").

At N = 1 the same run also measures, each over its own >= --config-timed-ms
timed region and each with its roofline fraction, the rest of the contract
(reported under "configs"): BASELINE configs[2] (KdV N=64 B=4096) and
configs[3] (KS N=256 B=8192), the headline workload with ONE FUSED LAUNCH PER
RK SUBSTEP (north_star's literal structure) and per RK step, the HBM-bound fixed-stencil
streaming kernel (GB/s against the HBM peak), the batch-1
`SavedModelDifferentiator.__call__` latency a SciPy caller sees, and the
on-device adaptive RK23 (the reference's production integrator, one controller
per sample).  With N > 1 the default shard is 8192 samples per GPU
(BASELINE configs[4]: 65 536 / 8); rank 0 also reports every rank's kernel
time and the isolated cost of the final gather.

Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for the roofline and
cpu_baseline definitions.
"""
import argparse
import glob
import hashlib
import json
import math
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_TFLOPS = 157.3   # MI355X_MICROARCH.md: f32 MFMA = f32 vector peak
PEAK_HBM_GBPS = 8000.0     # MI355X_MICROARCH.md: HBM3E spec peak
MEASURED_COPY_GBPS = 6290.0   # MI355X_MICROARCH.md: float4 copy, the achievable HBM rate
TRAFFIC_TABLES = ('r5_hbm_traffic.json', 'r4_hbm_traffic.json', 'r3_hbm_traffic.json', 'r2_hbm_traffic.json',
                  'r1_hbm_traffic.json')   # newest first


CONFIG_NAMES = ('kdv_n64_b4096', 'ks_n256_b8192', 'burgers_per_substep', 'burgers_per_step',
                'rk_substep_external', 'stream_fixed', 'stream_fixed_per_step',
                'differentiator_b1', 'adaptive_rk23',
                'adaptive_kdv_n64_b4096', 'adaptive_ks_n256_b1024',
                'tower_k7_b4096', 'tower_f64_b4096', 'tower_k7f64_b4096', 'tower_k3_b4096',
                'wide_ks_g9_b4096', 'burgers_b256', 'one_layer_b4096',
                # round 6: like-for-like partners of the adaptive / fixed-step KS legs, the
                # production integrator on a small ensemble, the WENO5 exact solver
                'adaptive_ks_n256_b8192', 'ks_n256_b1024', 'adaptive_rk23_b256', 'burgers_b512',
                'weno_exact_n512_b2048', 'tower_f16_b4096')


def parse_args(argv=None):
  ap = argparse.ArgumentParser()
  ap.add_argument('--gpus', type=int, default=1)
  ap.add_argument('--steps', type=int, default=1000)
  ap.add_argument('--warmup', type=int, default=100)
  ap.add_argument('--batch', type=int, default=None,
                  help='samples per GPU; default 4096 at --gpus 1 (north_star target), '
                       '8192 at --gpus > 1 (BASELINE configs[4]: 65 536 / 8)')
  ap.add_argument('--secondary-batch', type=int, default=1024,
                  help='second batch size measured in the same run at N=1 '
                       '(BASELINE.json configs[1]); 0 disables')
  ap.add_argument('--num-points', type=int, default=64)
  ap.add_argument('--equation', default='burgers')
  ap.add_argument('--scheme', default='midpoint',
                  choices=['euler', 'midpoint', 'bs3', 'rk4'])
  ap.add_argument('--launch-mode', default='persistent',
                  choices=['persistent', 'per_substep', 'per_step'])
  ap.add_argument('--state-dtype', default='float32', choices=['float32', 'float64'],
                  help='float64: state and RK update in f64, right-hand side in f32 '
                       '(the reference SciPy path, integrate.py:154)')
  ap.add_argument('--non-conservative', action='store_true')
  ap.add_argument('--baseline-stencils', action='store_true',
                  help='fixed polynomial stencils instead of the conv net')
  ap.add_argument('--hparams', default='{}',
                  help='JSON overrides of the model hyper-parameters (create_hparams), e.g. '
                       '\'{"coefficient_grid_min_size": 9}\' or \'{"nonlinearity": "tanh"}\'')
  ap.add_argument('--kernel', default='auto',
                  choices=['auto', 'mfma', 'mfma64', 'mfma64w32', 'mfma256', 'generic'])
  ap.add_argument('--preheat-ms', type=float, default=300.0,
                  help='run the timed kernel this long before --warmup (0 disables)')
  ap.add_argument('--min-timed-ms', type=float, default=1000.0,
                  help='repeat the K-step job until the timed region lasts this long')
  ap.add_argument('--configs', default='all',
                  help="comma list of the extra measurements at --gpus 1 ('all', 'none', or of "
                       + ', '.join(CONFIG_NAMES) + ')')
  ap.add_argument('--config-timed-ms', type=float, default=600.0,
                  help='minimum timed region of each extra measurement')
  ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                  help='collective backend for --gpus > 1: nccl = RCCL over xGMI (the '
                       'measured configuration); gloo stages the gather through host '
                       'memory and lets several ranks share one GPU (tests of the N > 1 '
                       'code path on a one-GPU box; never a headline number)')
  ap.add_argument('--debug-option', action='append', default=[], metavar='NAME=VALUE',
                  help='library A/B switch (ddd_debug_set_option of libddd1d_probe.so, '
                       '__graft_entry__.build_probe), e.g. no_spec=1; logged to stderr, never '
                       'set in a headline run')
  ap.add_argument('--library', default=None, metavar='NAME',
                  help='A/B build of the library to load instead of the product one '
                       '(csrc/libddd1d_<NAME>.so, __graft_entry__.build_hip(variant=NAME)); named '
                       'in config.library, never a headline run')
  ap.add_argument('--cpu-seconds', type=float, default=12.0,
                  help='budget for the CPU baseline sample (0 disables)')
  args = ap.parse_args(argv)
  if args.batch is None:
    args.batch = 4096 if args.gpus == 1 else 8192
  return args


def build_workload(args, rank, batch, unique=None):
  """Model + per-sample forcing + random-phase initial conditions for `batch`
  samples of this rank.  `unique`: draw only that many distinct samples and
  tile them (the 262 144-sample streaming config; sample content does not
  change the work)."""
  import ddd1d_amd
  from ddd1d_amd import equations, model as model_lib
  rf = 8
  hp = ddd1d_amd.create_hparams(
      args.equation, conservative=not args.non_conservative,
      resample_factor=rf,
      equation_kwargs=json.dumps({'num_points': args.num_points * rf}),
      **json.loads(getattr(args, 'hparams', '{}') or '{}'))
  _, eq = equations.from_hparams(hp, random_seed=0)
  if args.baseline_stencils:
    model = model_lib.BaselineModel(eq, accuracy_order=1)
  else:
    model = model_lib.LearnedStencilModel(eq, hp, init_seed=0, output_scale=0.1)
  model.set_kernel(args.kernel)
  # sample ids are global: rank r owns ids [r*batch, (r+1)*batch)
  from ddd1d_amd import distributed
  drawn = batch if unique is None else min(unique, batch)
  seeds = list(distributed.weak_shard_ids(batch, rank))[:drawn]
  forcing = model_lib.batched_forcing_parameters(seeds, nparams=20)
  ic = model_lib.batched_forcing_parameters(
      [s + (1 << 20) for s in seeds], nparams=10)
  x = eq.grid.reference_x
  waves = np.sum(ic['a'][..., None] * np.sin(
      2 * np.pi * ic['k'][..., None] * x / eq.grid.period + ic['phi'][..., None]),
                 axis=1)
  y0 = eq.grid.resample(waves).astype(np.float32)
  if drawn < batch:
    reps = -(-batch // drawn)
    y0 = np.tile(y0, (reps, 1))[:batch]
    forcing = {k: np.tile(v, (reps, 1))[:batch] for k, v in forcing.items()}
  model.set_forcing(forcing)
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


class ClockSampler(object):
  """Samples the shader clock and socket power of the GPU this rank runs on
  from sysfs (amdgpu hwmon: freq1_input [Hz], power1_average / power1_input
  [uW]) in a background thread while the kernel runs: evidence for the
  sustained-clock figure in DESIGN.md.  The card is found by the PCI address
  torch reports for the device; when that is unavailable every amdgpu card is
  sampled and the busiest one (highest mean clock in the window) is reported.
  Silent (returns None) where the files do not exist."""

  def __init__(self, device_index=0, period_s=0.004):
    self.period = period_s
    self.samples = []
    self._stop = threading.Event()
    self._thread = None
    self.cards = self._find(device_index)   # [(freq_path, power_path)]

  @staticmethod
  def _pci_address(device_index):
    try:
      import torch
      props = torch.cuda.get_device_properties(device_index)
      return '{:04x}:{:02x}:{:02x}.0'.format(props.pci_domain_id, props.pci_bus_id,
                                             props.pci_device_id)
    except Exception:   # attribute missing on this torch build
      return None

  @classmethod
  def _find(cls, device_index):
    def hwmon_of(device_dir):
      out = []
      for base in sorted(glob.glob(os.path.join(device_dir, 'hwmon', 'hwmon*'))):
        if not os.path.exists(os.path.join(base, 'freq1_input')):
          continue
        power = None
        for name in ('power1_average', 'power1_input'):
          if os.path.exists(os.path.join(base, name)):
            power = os.path.join(base, name)
            break
        out.append((os.path.join(base, 'freq1_input'), power))
      return out
    address = cls._pci_address(device_index)
    if address is not None:
      cards = hwmon_of(os.path.join('/sys/bus/pci/devices', address))
      if cards:
        return cards[:1]
    cards = []
    for dev in sorted(glob.glob('/sys/class/drm/card[0-9]*/device')):
      cards.extend(hwmon_of(dev))
    return cards

  @staticmethod
  def _read(path):
    try:
      with open(path) as f:
        return float(f.read().strip())
    except (OSError, ValueError):
      return None

  def _loop(self):
    while not self._stop.is_set():
      now = time.perf_counter()
      for index, (freq_path, power_path) in enumerate(self.cards):
        freq = self._read(freq_path)
        power = self._read(power_path) if power_path else None
        self.samples.append((now, index, freq, power))
      time.sleep(self.period)

  def start(self):
    if not self.cards:
      return self
    self._thread = threading.Thread(target=self._loop, daemon=True)
    self._thread.start()
    return self

  def stop(self, t_begin=None, t_end=None):
    if self._thread is None:
      return None
    self._stop.set()
    self._thread.join()
    best = None
    for index, (freq_path, _) in enumerate(self.cards):
      rows = [s for s in self.samples if s[1] == index
              and (t_begin is None or s[0] >= t_begin) and (t_end is None or s[0] <= t_end)]
      freqs = [s[2] / 1e6 for s in rows if s[2]]
      powers = [s[3] / 1e6 for s in rows if s[3]]
      if not freqs:
        continue
      out = {'source': freq_path, 'samples': len(freqs),
             'matched_by': 'pci address' if len(self.cards) == 1 else 'busiest card',
             'sclk_mhz_mean': sum(freqs) / len(freqs), 'sclk_mhz_min': min(freqs),
             'sclk_mhz_max': max(freqs)}
      if powers:
        out.update(power_w_mean=sum(powers) / len(powers), power_w_max=max(powers))
      if best is None or out['sclk_mhz_mean'] > best['sclk_mhz_mean']:
        best = out
    return best


def measure(args, model, y0_host, world, n, batch, sample_clocks=False):
  """Pre-heat, warm-up and the timed region for one (model, batch).  Returns
  wall seconds, HIP-event milliseconds, repetitions and bookkeeping; all ranks
  must call it together."""
  import torch
  import torch.distributed as dist
  dt = model.equation.time_step
  dtype = torch.float64 if args.state_dtype == 'float64' else torch.float32
  y0 = torch.from_numpy(y0_host).cuda().to(dtype)
  # two result / gather buffers: with N > 1 the RCCL gather of job r (its own
  # stream) overlaps the kernel of job r + 1 -- independent jobs stream through
  finals = [torch.empty((1, batch, n), dtype=dtype, device='cuda') for _ in range(2)]
  final = finals[0]
  host = args.backend == 'gloo'       # gloo: collectives on host copies
  gathers = ([torch.empty((world * batch, n), dtype=dtype,   # rank slabs concatenated
                          device='cpu' if host else 'cuda') for _ in range(2)]
             if world > 1 else None)
  pending = [None, None]

  def gather_final(slot):
    source = finals[slot][0].cpu() if host else finals[slot][0]
    pending[slot] = dist.all_gather_into_tensor(gathers[slot], source, async_op=True)

  def wait_slot(slot):
    if pending[slot] is not None:
      pending[slot].wait()          # the compute stream waits for that gather
      pending[slot] = None

  def job(num_steps, slot=0):
    wait_slot(slot)                 # its buffers are free again
    model.integrate_fixed(y0, num_steps, dt=dt, t0=0.0, scheme=args.scheme,
                          save_every=num_steps, launch_mode=args.launch_mode,
                          state_dtype=args.state_dtype, out=finals[slot])
    if world > 1:
      gather_final(slot)

  def barrier():
    wait_slot(0)
    wait_slot(1)
    if world > 1:
      torch.cuda.synchronize()
      dist.barrier()
    torch.cuda.synchronize()

  start_evt = torch.cuda.Event(enable_timing=True)
  stop_evt = torch.cuda.Event(enable_timing=True)
  sampler = ClockSampler(torch.cuda.current_device()).start() if sample_clocks else None

  # 1. pre-heat with the timed kernel itself (clocks, code object, L2)
  heat_steps = max(args.steps, 200)
  step_ms = None
  heated = 0.0
  heat_launches = 0
  while heated < args.preheat_ms or step_ms is None:
    start_evt.record()
    model.integrate_fixed(y0, heat_steps, dt=dt, t0=0.0, scheme=args.scheme,
                          save_every=heat_steps, launch_mode=args.launch_mode,
                          state_dtype=args.state_dtype, out=final)
    stop_evt.record()
    torch.cuda.synchronize()
    ms = start_evt.elapsed_time(stop_evt)
    heated += ms
    heat_launches += 1
    step_ms = ms / heat_steps
    if args.preheat_ms <= 0 or heat_launches >= 10000:
      break
  # repetitions: agreed across ranks (rank 0's estimate)
  reps = max(1, int(math.ceil(args.min_timed_ms / max(step_ms * args.steps, 1e-6))))
  reps = min(reps, 100000)
  if world > 1:
    r = torch.tensor([reps], dtype=torch.int64, device='cpu' if host else 'cuda')
    dist.broadcast(r, 0)
    reps = int(r[0])

  # 1b. N > 1: the SAME per-rank workload on rank 0 ALONE (the other ranks idle at the
  # barrier), so that one invocation yields the 1 -> N point of the weak-scaling curve
  # even when no separate --gpus 1 run of this shard size exists ("scaling_detail")
  solo = None
  if world > 1:
    barrier()
    if dist.get_rank() == 0:
      model.integrate_fixed(y0, args.steps, dt=dt, t0=0.0, scheme=args.scheme,
                            save_every=args.steps, launch_mode=args.launch_mode,
                            state_dtype=args.state_dtype, out=final)
      torch.cuda.synchronize()
      solo0 = time.perf_counter()
      start_evt.record()
      for rep in range(reps):
        model.integrate_fixed(y0, args.steps, dt=dt, t0=0.0, scheme=args.scheme,
                              save_every=args.steps, launch_mode=args.launch_mode,
                              state_dtype=args.state_dtype, out=finals[rep & 1])
      stop_evt.record()
      torch.cuda.synchronize()
      solo = {'wall': time.perf_counter() - solo0, 'kernel_ms': start_evt.elapsed_time(stop_evt)}
    barrier()

  # 2. the contract's warm-up steps
  if args.warmup > 0:
    job(args.warmup)
  # 3. timed region
  barrier()
  wall0 = time.perf_counter()
  # ONE event pair around the R back-to-back launches, on the stream they are
  # launched on: kernel_ms / launches = the average launch duration including the
  # dependent-launch gap (a pair per launch put two marker packets between every two
  # kernels: ~40 us of bubbles per 1.1 ms launch in round 3's line, which is what
  # made `value` and `roofline.frac` disagree by 3.7 %)
  start_evt.record()
  for rep in range(reps):
    slot = rep & 1
    wait_slot(slot)
    model.integrate_fixed(y0, args.steps, dt=dt, t0=0.0, scheme=args.scheme,
                          save_every=args.steps, launch_mode=args.launch_mode,
                          state_dtype=args.state_dtype, out=finals[slot])
    if world > 1:
      gather_final(slot)
  stop_evt.record()
  barrier()
  wall1 = time.perf_counter()
  wall = wall1 - wall0
  kernel_ms = start_evt.elapsed_time(stop_evt)
  clocks = sampler.stop(wall0, wall1) if sampler is not None else None

  per_rank = None
  if world > 1:
    # one line must explain a < N x result: every rank's own wall / kernel time,
    # and the final gather timed on its own (outside the timed region, after it)
    gather_ms = []
    for _ in range(5):
      barrier()
      g0 = time.perf_counter()
      gather_final(0)
      wait_slot(0)
      torch.cuda.synchronize()
      gather_ms.append((time.perf_counter() - g0) * 1e3)
    mine = torch.tensor([wall * 1e3, kernel_ms, min(gather_ms)], dtype=torch.float64,
                        device='cpu' if host else 'cuda')
    everyone = torch.empty(world * 3, dtype=torch.float64, device=mine.device)
    dist.all_gather_into_tensor(everyone, mine)
    table = everyone.cpu().reshape(world, 3)
    per_rank = {'wall_ms': [float(v) for v in table[:, 0]],
                'kernel_ms': [float(v) for v in table[:, 1]],
                'gather_ms_isolated': [float(v) for v in table[:, 2]],
                'gather_bytes_per_rank': int(finals[0][0].numel() * finals[0].element_size()),
                # the gathered ensemble after `steps` steps from y0 (rank slabs in rank
                # order): equals the single-process run of the same global sample ids
                'gathered_sha1': hashlib.sha1(
                    gathers[0].cpu().numpy().tobytes()).hexdigest(),
                'gathered_shape': list(gathers[0].shape)}
    wall, kernel_ms = float(table[:, 0].max()) * 1e-3, float(table[:, 1].max())
  finite = bool(torch.isfinite(finals[0]).all())
  if not finite:
    sys.stderr.write('bench.py: WARNING: the state is not finite after the timed run '
                     '(batch {}): its throughput is reported with "finite": false and '
                     'must not be quoted\n'.format(batch))
  return dict(wall=wall, kernel_ms=kernel_ms, reps=reps, finite=finite,
              preheat_ms=heated, preheat_launches=heat_launches, clocks=clocks,
              per_rank=per_rank, solo=solo)


def summarize(args, eq, model, m, world, n, batch, stages):
  """Throughput + roofline numbers for one measurement."""
  steps_timed = args.steps * m['reps']
  total_points = batch * n * steps_timed * world
  fma = model.fma_per_point
  # (launches per sample: large ensembles run as two half-ensemble launches side by side)
  launches_per_job = (1 if args.launch_mode == 'persistent' else
                      args.steps if args.launch_mode == 'per_step' else args.steps * stages)
  launches = launches_per_job * m['reps']
  flops_per_launch = 2.0 * fma * batch * n * stages * args.steps / launches_per_job
  state_bytes = 8.0 if args.state_dtype == 'float32' else 16.0   # in + out per point
  bytes_per_launch = (state_bytes * batch * n if args.launch_mode != 'per_substep' else
                      (20.0 if args.scheme == 'midpoint' else 8.0 * stages)
                      * batch * n / stages)
  launch_s = m['kernel_ms'] * 1e-3 / launches
  achieved_tflops = flops_per_launch / launch_s / 1e12
  achieved_gbps = bytes_per_launch / launch_s / 1e9
  compute_bound = not args.baseline_stencils
  traffic, traffic_source, traffic_command = measured_traffic(
      type(eq).__name__, n, batch, args.launch_mode, args.baseline_stencils,
      state_dtype=args.state_dtype, hparams=json.loads(getattr(args, 'hparams', '{}') or '{}'))
  roofline = {
      'bound': 'mfma' if compute_bound else 'hbm',
      'achieved': achieved_tflops if compute_bound else achieved_gbps,
      'peak': PEAK_FP32_TFLOPS if compute_bound else PEAK_HBM_GBPS,
      'unit': 'TFLOP/s' if compute_bound else 'GB/s',
      'frac': (achieved_tflops / PEAK_FP32_TFLOPS if compute_bound
               else achieved_gbps / PEAK_HBM_GBPS),
      'traffic': traffic,
      'traffic_source': traffic_source,     # committed rocprofv3 --pmc passes of this
      'traffic_command': traffic_command,   # configuration (bench.py cannot read counters)
      'kernel_ms_per_launch': m['kernel_ms'] / launches,
      'launches': launches,
      'hbm_gbps': achieved_gbps,
      'hbm_frac': achieved_gbps / PEAK_HBM_GBPS,
      'fp32_tflops': achieved_tflops,
      'fp32_frac': achieved_tflops / PEAK_FP32_TFLOPS,
  }
  # the same fraction from the WALL clock of the timed region (what `value` is made
  # of): below `frac` by the host gaps between the launches of a job list
  wall_per_launch = m['wall'] / launches
  roofline['frac_wall'] = ((flops_per_launch / wall_per_launch / 1e12 / PEAK_FP32_TFLOPS)
                           if compute_bound else
                           (bytes_per_launch / wall_per_launch / 1e9 / PEAK_HBM_GBPS))
  if not compute_bound and args.launch_mode == 'per_substep':
    # `frac` prices the bytes the launches actually move (midpoint: 8 B/point in stage
    # 1, 12 in stage 2, where y_n is read again: 10 on average, counter-verified);
    # SURVEY section 8(d)'s algorithmic figure is 8 B per grid-point-substep
    algorithmic = 8.0 * batch * n / launch_s / 1e9
    roofline['frac_algorithmic'] = algorithmic / PEAK_HBM_GBPS
    roofline['algorithmic_gbps'] = algorithmic
    roofline['frac_of_copy_rate'] = achieved_gbps / MEASURED_COPY_GBPS
  return {
      'value': total_points / m['wall'],
      'ms_per_step': m['wall'] * 1e3 / steps_timed,
      'roofline': roofline,
  }


def scaling_detail(args, m, world, n, batch, value):
  """N > 1: rank 0's throughput on the same per-rank workload with the other ranks idle
  (measured in the same invocation, right before the timed region), the whole job's
  value against N times that, and the process group as torch.distributed reports it
  (backend "nccl" = RCCL on ROCm).  The contract's `scaling` key stays "weak"; the
  driver computes its own efficiency from separate runs -- this is the self-check."""
  if world <= 1 or m.get('solo') is None:
    return None
  import torch.distributed as dist
  n1_value = batch * n * args.steps * m['reps'] / m['solo']['wall']
  return {
      'n1_value': n1_value,
      'n1_ms_per_step': m['solo']['wall'] * 1e3 / (args.steps * m['reps']),
      'n1_kernel_ms': m['solo']['kernel_ms'],
      'n1_batch': batch,
      'efficiency': value / (world * n1_value),
      'nranks': dist.get_world_size(),
      'backend': dist.get_backend(),
      'note': 'rank 0 alone on its shard (others idle at a barrier), same steps x reps; '
              'efficiency = value / (n_gpus x n1_value)',
  }


def _variant(args, **overrides):
  out = argparse.Namespace(**vars(args))
  for k, v in overrides.items():
    setattr(out, k, v)
  return out


def _fixed_step_config(args, lib, world, name, note, batch, unique=None, **overrides):
  """One extra fixed-step measurement through the same measure()/summarize()."""
  import ddd1d_amd
  a = _variant(args, min_timed_ms=args.config_timed_ms, preheat_ms=min(args.preheat_ms, 150.0),
               warmup=min(args.warmup, 20), **overrides)
  eq, model, _, y0 = build_workload(a, 0, batch, unique=unique)
  n = eq.grid.solution_num_points
  stages = lib.ddd_scheme_stages(ddd1d_amd._lib.SCHEMES[a.scheme])
  m = measure(a, model, y0, world, n, batch)
  s = summarize(a, eq, model, m, world, n, batch, stages)
  r = s['roofline']
  out = {
      'workload': note, 'equation': type(eq).__name__, 'num_points': n, 'batch': batch,
      'steps': a.steps, 'reps': m['reps'], 'scheme': a.scheme, 'launch_mode': a.launch_mode,
      'kernel': model.kernel_name, 'value': s['value'], 'unit': 'grid-point-steps/s',
      'ms_per_step': s['ms_per_step'], 'timed_wall_ms': m['wall'] * 1e3,
      'bound': r['bound'], 'achieved': r['achieved'], 'peak': r['peak'],
      'roofline_unit': r['unit'], 'frac': r['frac'],
      'kernel_ms_per_launch': r['kernel_ms_per_launch'], 'launches': r['launches'],
      'traffic': r['traffic'], 'traffic_source': r['traffic_source'], 'finite': m['finite'],
      'frac_wall': r['frac_wall'],
  }
  for key in ('frac_algorithmic', 'algorithmic_gbps', 'frac_of_copy_rate'):
    if key in r:
      out[key] = r[key]
  model.close()
  return name, out


def load_rk_driver():
  """examples/librk_driver.so (gcc, against include/ddd1d.h only): the caller-owned
  midpoint loop over ddd_rk_substep, in C."""
  import ctypes
  import ddd1d_amd
  ddd1d_amd._lib.load_library()
  lib = ctypes.CDLL(os.path.join(ROOT, 'examples', 'librk_driver.so'))
  V = ctypes.c_void_p
  lib.rk_driver_midpoint.restype = ctypes.c_int
  lib.rk_driver_midpoint.argtypes = [V, ctypes.c_int, ctypes.c_double, ctypes.c_double, V, V, V,
                                     ctypes.c_int, V, ctypes.c_int, ctypes.POINTER(V)]
  return lib


def _external_driver_config(args):
  """The section-8(b) seam as an EXTERNAL driver sees it: a host loop that owns the
  Runge-Kutta stages (integrate.py:143-169 calls the right-hand side per stage) and
  calls ddd_rk_substep twice per midpoint step.  Three drivers per batch size: the
  loop in Python inside ddd_stream_fork .. ddd_stream_join (two half-ensemble
  chains alive across the calls), the same loop in C (examples/rk_driver.c; takes
  Python out), and the Python loop WITHOUT the bracket (one launch per call on the
  caller's stream: what the seam gave before)."""
  import ctypes
  import torch
  import ddd1d_amd
  driver = load_rk_driver()
  lib = ddd1d_amd._lib.load_library()
  steps = 200
  out = {}
  for batch in (4096, 8192):
    a = _variant(args)
    eq, model, _, y0_host = build_workload(a, 0, batch, unique=4096)
    n = eq.grid.solution_num_points
    dt = eq.time_step
    h = np.float32(dt)
    y0 = torch.from_numpy(y0_host).cuda()
    bufs = [torch.empty_like(y0) for _ in range(3)]
    stream = ddd1d_amd._lib.current_stream()

    def python_loop(chained):
      y, ystage, ynew = bufs
      y.copy_(y0)
      ctx = model.chained_substeps() if chained else None
      if ctx is not None:
        ctx.__enter__()
      try:
        for step in range(steps):
          t = step * dt
          model.rk_substep(t, y, y_base=y, c1=0.5 * h, y_out=ystage)
          model.rk_substep(t + 0.5 * dt, ystage, acc_in=y, c2=h, acc_out=ynew)
          y, ynew = ynew, y
      finally:
        if ctx is not None:
          ctx.__exit__(None, None, None)
      return y

    def c_loop(chained):
      y, ystage, ynew = bufs
      y.copy_(y0)
      final = ctypes.c_void_p()
      rc = driver.rk_driver_midpoint(model._handle, steps, 0.0, dt, y.data_ptr(),
                                     ystage.data_ptr(), ynew.data_ptr(), batch, stream,
                                     1 if chained else 0, ctypes.byref(final))
      if rc != 0:
        raise RuntimeError(lib.ddd_last_error().decode())
      return y if final.value == y.data_ptr() else ynew

    def timed(fn, chained):
      for _ in range(2):
        fn(chained)
      torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      ms, jobs = 0.0, 0
      wall0 = time.perf_counter()
      while ms < args.config_timed_ms / 2:
        e0.record()
        last = fn(chained)
        e1.record()
        torch.cuda.synchronize()
        ms += e0.elapsed_time(e1)
        jobs += 1
      wall = time.perf_counter() - wall0
      flops = 2.0 * model.fma_per_point * batch * n * 2 * steps * jobs
      return {'value': batch * n * steps * jobs / (ms * 1e-3), 'us_per_substep_call':
              ms * 1e3 / (2 * steps * jobs), 'fp32_tflops': flops / (ms * 1e-3) / 1e12,
              'frac': flops / (ms * 1e-3) / 1e12 / PEAK_FP32_TFLOPS, 'jobs': jobs,
              'timed_wall_ms': wall * 1e3, 'finite': bool(torch.isfinite(last).all())}, last

    # the region on launches: two half-ensemble chains (round 4; the default) ...
    c_chained, y_c = timed(c_loop, True)
    y_c = y_c.clone()
    py_chained, y_py = timed(python_loop, True)
    same = bool(torch.equal(y_c, y_py))
    py_plain, y_plain = timed(python_loop, False)
    same = same and bool(torch.equal(y_c, y_plain))
    # ... and on the command ring (round 6, opt-in: ONE persistent kernel takes the calls as
    # 128-byte commands; DDD_REGION_RING)
    model.set_region_mode('ring')
    before = model.region_stats()
    c_ring, y_ring = timed(c_loop, True)
    after = model.region_stats()
    same = same and bool(torch.equal(y_c, y_ring))
    model.set_region_mode('auto')
    ref = model.integrate_fixed(y0, steps, dt=dt, scheme='midpoint', save_every=steps,
                                launch_mode='per_substep')[0]
    out['b{}'.format(batch)] = {
        'c_loop_chained': c_chained, 'python_loop_chained': py_chained,
        'c_loop_command_ring': c_ring,
        'python_loop_unchained': py_plain,
        'ring': {'persistent_kernel_launches': after[0] - before[0],
                 'commands': after[1] - before[1]},
        'drivers_bit_identical': same,
        'equals_ddd_integrate_fixed_per_substep': bool(torch.equal(y_c, ref)),
    }
    model.close()
  best = out['b4096']['c_loop_chained']
  result = {
      'workload': 'headline model (Burgers N=64 conv-net stencils), midpoint, a HOST loop that owns '
                  'the RK stages: 2 ddd_rk_substep calls per step, {} steps per job, batch 4096 '
                  'and 8192 (tiled from 4096 distinct samples); value = the C loop inside '
                  'ddd_stream_fork .. ddd_stream_join at batch 4096 (two chains of launches; '
                  'c_loop_command_ring = the same region on the opt-in device-resident command '
                  'ring: one persistent kernel, every call a 128-byte command)'.format(steps),
      'value': best['value'], 'unit': 'grid-point-steps/s', 'bound': 'mfma',
      'achieved': best['fp32_tflops'], 'peak': PEAK_FP32_TFLOPS, 'roofline_unit': 'TFLOP/s',
      'frac': best['frac'], 'finite': best['finite'], 'kernel': 'mfma_f32_r64',
  }
  result.update(out)
  return 'rk_substep_external', result


def _differentiator_config(args):
  """Batch-1 `SavedModelDifferentiator.__call__(t, y)` (integrate.py:48-71) as
  SciPy calls it: host float64 array in, host array out, one launch per call.
  The reference logs 2.0-4.3 ms per evaluation of this seam."""
  import torch
  from ddd1d_amd import integrate
  a = _variant(args)
  eq, model, _, y0 = build_workload(a, 0, 1)
  diff = integrate.SavedModelDifferentiator(None, eq, model.hparams, model=model)
  y = y0[0].astype(np.float64)
  for i in range(200):
    diff(1e-3 * i, y)
  torch.cuda.synchronize()
  calls, elapsed = 0, 0.0
  while elapsed * 1e3 < args.config_timed_ms:
    t0 = time.perf_counter()
    for i in range(500):
      out = diff(1e-3 * i, y)
    elapsed += time.perf_counter() - t0
    calls += 500
  us = elapsed / calls * 1e6
  # one whole one-sample solve as integrate.odeint does it: SciPy's RK23 for t in
  # [0, 1] (302 evaluations) on the device in one launch, and as the reference's
  # host loop over the same differentiator
  times = np.linspace(0.0, 1.0, 11)
  integrate.odeint(y, diff, times)
  t0 = time.perf_counter()
  _, nfev = integrate.odeint(y, diff, times)
  odeint_device_ms = (time.perf_counter() - t0) * 1e3
  integrate.DEVICE_ODEINT = False
  try:
    t0 = time.perf_counter()
    integrate.odeint(y, diff, times)
    odeint_host_ms = (time.perf_counter() - t0) * 1e3
  finally:
    integrate.DEVICE_ODEINT = True
  n = eq.grid.solution_num_points
  flops = 2.0 * diff.model.fma_per_point * n
  result = {
      'workload': 'Differentiator.__call__(t, y[{}]) -> dy/dt, batch 1, host arrays in and '
                  'out (page-locked rows read / written by the kernel itself: one fused launch + one stream sync per call), {} calls'.format(n, calls),
      'value': us, 'unit': 'us/evaluation', 'higher_is_better': False,
      'evaluations_per_s': 1e6 / us, 'kernel': diff.model.kernel_name,
      'bound': 'latency', 'fp32_tflops': flops / (us * 1e-6) / 1e12,
      'frac': flops / (us * 1e-6) / 1e12 / PEAK_FP32_TFLOPS,
      'reference_ms_per_evaluation': [2.0, 4.3], 'finite': bool(np.isfinite(out).all()),
      'odeint_one_sample': {
          'workload': 'integrate.odeint(y0, differentiator, linspace(0, 1, 11)): RK23, '
                      '{} evaluations'.format(nfev),
          'device_ms': odeint_device_ms, 'host_loop_ms': odeint_host_ms},
  }
  diff.model.close()
  return 'differentiator_b1', result


def _adaptive_config(args, name='adaptive_rk23', batch=4096, t_end=1.0, unique=None):
  """`ddd_integrate_adaptive_f64`: the reference's production integrator
  (solve_ivp RK23, max_step 0.01, integrate.py:143-169) for the whole batch in
  one launch, one controller per sample; float64 state, float32 right-hand side.
  Burgers sits at max_step (saturated controller), KdV and KS are stability-limited
  (continuous rejections; KS N=256 is one sample per four-wave group)."""
  import torch
  a = _variant(args)
  eq, model, _, y0 = build_workload(a, 0, batch, unique=unique)
  n = eq.grid.solution_num_points
  times = np.linspace(0.0, t_end, 11)
  y0d = torch.from_numpy(y0.astype(np.float64)).cuda()
  model.integrate_adaptive(y0d, times)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  kernel_ms, launches = 0.0, 0
  wall0 = time.perf_counter()
  while kernel_ms < args.config_timed_ms:
    e0.record()
    y, nfev, status = model.integrate_adaptive(y0d, times)
    e1.record()
    torch.cuda.synchronize()
    kernel_ms += e0.elapsed_time(e1)
    launches += 1
  wall = time.perf_counter() - wall0
  nfev = nfev.cpu().numpy().astype(np.int64)
  evals = float(nfev.sum()) * n              # grid-point-evaluations per launch
  steps = float(((nfev - 2) // 3).sum()) * n   # grid-point-steps (attempted RK23 steps)
  # samples of one workgroup share evaluations: a group runs as long as its slowest
  # sample, so the matrix work ISSUED is the per-group maximum (useful = the sum)
  spg = max(1, (64 if n <= 64 and 64 % n == 0 else 256) // n)
  pad = (-len(nfev)) % spg
  issued = float(np.pad(nfev, (0, pad)).reshape(-1, spg).max(axis=1).sum()) * spg * n
  tflops = 2.0 * model.fma_per_point * evals * launches / (kernel_ms * 1e-3) / 1e12
  result = {
      'workload': '{} N={} conv-net stencils, batch {}{}, solve_ivp-RK23 semantics per '
                  'sample (rtol 1e-3, atol 1e-6, max_step 0.01), t in [0, {:g}], 11 output '
                  'times, {} launches'.format(
                      a.equation, n, batch,
                      '' if unique is None else ' TILED from {} distinct samples'.format(unique),
                      t_end, launches),
      'value': evals * launches / (kernel_ms * 1e-3), 'unit': 'grid-point-evaluations/s',
      'grid_point_steps_per_s': steps * launches / (kernel_ms * 1e-3),
      'nfev_min': int(nfev.min()), 'nfev_max': int(nfev.max()),
      'nfev_mean': float(nfev.mean()),
      'samples_finished': int((status.cpu().numpy() == 0).sum()),
      'kernel_ms_per_launch': kernel_ms / launches, 'timed_wall_ms': wall * 1e3,
      'kernel': model.kernel_name, 'state_dtype': 'float64',
      'bound': 'mfma', 'achieved': tflops, 'peak': PEAK_FP32_TFLOPS,
      'roofline_unit': 'TFLOP/s', 'frac': tflops / PEAK_FP32_TFLOPS,
      'frac_issued': tflops / PEAK_FP32_TFLOPS * issued / max(evals, 1.0),
      'finite': bool(torch.isfinite(y).all()),
  }
  model.close()
  return name, result


def _weno_exact_config(args, n=512, batch=2048, t_end=0.5):
  """The fine-grid exact Burgers solver (integrate.WENODifferentiator, integrate.py:124-140;
  what scripts/create_exact_data.py maps over seeds): WENO5 + Godunov flux, per-seed forcing,
  SciPy-RK23 semantics per sample in one launch of csrc/rhs_weno.h (one wavefront per
  sample).  VALU-bound: no matrix work; ~450 VALU instructions per grid-point-evaluation,
  half of them the 21 correctly rounded float32 divisions of the nonlinear weights."""
  import torch
  from ddd1d_amd import equations, model as model_lib
  eqs = [equations.GodunovBurgersEquation(n, random_seed=s) for s in range(batch)]
  model = model_lib.BaselineModel(eqs[0], 3, weno=True)
  model.set_forcing(model_lib.forcing_from_equations(eqs))
  ic = model_lib.batched_forcing_parameters([s + (1 << 20) for s in range(batch)], nparams=10)
  x = eqs[0].grid.solution_x
  y0 = np.sum(ic['a'][..., None] * np.sin(
      2 * np.pi * ic['k'][..., None] * x / eqs[0].grid.period + ic['phi'][..., None]), axis=1)
  y0d = torch.from_numpy(y0.astype(np.float64)).cuda()
  times = np.linspace(0.0, t_end, 3)
  model.integrate_adaptive(y0d, times)
  torch.cuda.synchronize()
  e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
  kernel_ms, launches = 0.0, 0
  while kernel_ms < args.config_timed_ms:
    e0.record()
    y, nfev, status = model.integrate_adaptive(y0d, times)
    e1.record()
    torch.cuda.synchronize()
    kernel_ms += e0.elapsed_time(e1)
    launches += 1
  nfev = nfev.cpu().numpy().astype(np.int64)
  evals = float(nfev.sum()) * n
  result = {
      'workload': 'GodunovBurgers N={} (the reference\'s exact grid), WENO5 + Godunov flux, per-seed '
                  'forcing, batch {}, solve_ivp-RK23 semantics per sample, t in [0, {:g}], {} '
                  'launches'.format(n, batch, t_end, launches),
      'value': evals * launches / (kernel_ms * 1e-3), 'unit': 'grid-point-evaluations/s',
      'nfev_min': int(nfev.min()), 'nfev_max': int(nfev.max()),
      'samples_finished': int((status.cpu().numpy() == 0).sum()),
      'kernel_ms_per_launch': kernel_ms / launches, 'kernel': model.kernel_name,
      'state_dtype': 'float64', 'bound': 'valu', 'finite': bool(torch.isfinite(y).all()),
      'round5_generic_kernel': 4.77e9,
  }
  model.close()
  return 'weno_exact_n512_b2048', result


def extra_configs(args, lib, world):
  """The rest of the contract, measured in the same run at N = 1."""
  wanted = (CONFIG_NAMES if args.configs == 'all' else
            () if args.configs == 'none' else tuple(args.configs.split(',')))
  unknown = [w for w in wanted if w not in CONFIG_NAMES]
  if unknown:
    raise SystemExit('unknown --configs entries: {}'.format(unknown))
  base = dict(equation='burgers', num_points=64, non_conservative=False,
              baseline_stencils=False, kernel='auto', scheme='midpoint',
              launch_mode='persistent', state_dtype='float32')
  out = {}
  for name in wanted:
    try:
      if name == 'kdv_n64_b4096':
        key, val = _fixed_step_config(
            args, lib, world, name, 'BASELINE.json configs[2]: KdV N=64, conv-net stencils, '
            'batch 4096, midpoint at the equation time step', 4096,
            **dict(base, equation='kdv', steps=1000))
      elif name == 'ks_n256_b8192':
        key, val = _fixed_step_config(
            args, lib, world, name, 'BASELINE.json configs[3]: KS N=256, conv-net stencils, '
            'batch 8192 TILED from 1024 distinct samples, midpoint at the equation time step '
            '(400-step jobs; the 10k-step horizon is one longer launch of the same kernel, '
            'tests/test_gpu_full_size.py runs it once)', 8192, unique=1024,
            **dict(base, equation='ks', num_points=256, steps=400))
      elif name == 'burgers_per_substep':
        key, val = _fixed_step_config(
            args, lib, world, name, 'the headline workload with ONE FUSED LAUNCH PER RK '
            'SUBSTEP for every sample (north_star structure; state through HBM every substep; '
            'the ensemble advances as two half-ensembles on two streams, launches side by side)',
            4096,
            **dict(base, launch_mode='per_substep', steps=200))
      elif name == 'burgers_per_step':
        key, val = _fixed_step_config(
            args, lib, world, name, 'the headline workload with one launch per RK STEP (all '
            'stages fused, state through HBM once per step): for callers that need the host '
            'between steps but not between substeps', 4096,
            **dict(base, launch_mode='per_step', steps=200))
      elif name == 'rk_substep_external':
        key, val = _external_driver_config(_variant(args, **base))
      elif name == 'stream_fixed':
        key, val = _fixed_step_config(
            args, lib, world, name, 'fixed polynomial stencils (PolynomialDifferentiator), '
            'KdV N=64 batch 262144 TILED from 4096 distinct samples, one launch per substep: '
            'the HBM-bound kernel of the path', 262144, unique=4096,
            **dict(base, equation='kdv', baseline_stencils=True, launch_mode='per_substep',
                   steps=200))
      elif name == 'stream_fixed_per_step':
        key, val = _fixed_step_config(
            args, lib, world, name, 'the same fixed-stencil ensemble (KdV N=64 batch 262144 TILED '
            'from 4096 distinct samples) with ALL stages of a midpoint step in one launch: the '
            'stage input stays in the block\'s LDS tile, 8 B per grid point and step',
            262144, unique=4096,
            **dict(base, equation='kdv', baseline_stencils=True, launch_mode='per_step',
                   steps=200))
      elif name == 'wide_ks_g9_b4096':
        hp = {'coefficient_grid_min_size': 9}
        key, val = _fixed_step_config(
            args, lib, world, name, 'KS N=64 with 9-point stencils (coefficient_grid_min_size = 9, '
            'training_test.py:56): the wide flavour of the run-time MFMA kernels, output layer '
            'folded to the 27 coefficients; fractions in TRUE-net FLOPs', 4096,
            **dict(base, equation='ks', hparams=json.dumps(hp), steps=200))
      elif name in ('tower_k7_b4096', 'tower_f64_b4096', 'tower_k7f64_b4096', 'tower_k3_b4096'):
        hp = {'tower_k7_b4096': {'kernel_size': 7}, 'tower_f64_b4096': {'filter_size': 64},
              'tower_k7f64_b4096': {'kernel_size': 7, 'filter_size': 64},
              'tower_k3_b4096': {'kernel_size': 3}}[name]
        key, val = _fixed_step_config(
            args, lib, world, name, 'the headline workload with hyper-parameters {} '
            '(training.py:134-136 leaves them free): the MFMA towers with streamed weights; '
            'fractions in TRUE-net FLOPs'.format(json.dumps(hp)), 4096,
            **dict(base, hparams=json.dumps(hp), steps=200))
      elif name == 'tower_f16_b4096':
        hp = {'filter_size': 16}
        key, val = _fixed_step_config(
            args, lib, world, name, 'the headline workload with hyper-parameters {} '
            '(training.py:134-136 leaves them free): the block-diagonal tower of nets with up to 16 '
            'filters (rhs_mfma.h HalfTower; round 5: embedded in 32 filters, 25.7 %); fractions in '
            'TRUE-net FLOPs'.format(json.dumps(hp)), 4096,
            **dict(base, hparams=json.dumps(hp), steps=500))
      elif name == 'one_layer_b4096':
        key, val = _fixed_step_config(
            args, lib, world, name, 'num_layers = 1 (a hyper-parameter create_hparams admits; '
            'NOT what the reference\'s integration tests train -- integrate_test.py:48 never passes '
            'its model_kwargs): coefficients affine in the 5 neighbouring values, folded on the '
            'host; 111 FMA per grid point and evaluation, no matrix work', 4096,
            **dict(base, hparams=json.dumps({'num_layers': 1}), steps=1000))
      elif name in ('burgers_b256', 'burgers_b512'):
        small = 256 if name == 'burgers_b256' else 512
        key, val = _fixed_step_config(
            args, lib, world, name, 'the headline model on a SMALL ensemble ({} samples: a '
            '{} of the SIMDs would hold a 64-row wavefront): every sample on FOUR 16-row '
            'wavefronts, all layers on 16x16x4 MFMAs (rhs_mfma.h kQuad; round 5: two 32-row '
            'wavefronts, 31.6 % at 256)'.format(small, 'quarter' if small == 256 else 'half'),
            small, **dict(base, steps=1000))
      elif name == 'differentiator_b1':
        key, val = _differentiator_config(_variant(args, **base))
      elif name == 'adaptive_rk23':
        key, val = _adaptive_config(_variant(args, **base))
      elif name == 'adaptive_rk23_b256':
        key, val = _adaptive_config(_variant(args, **base), name, 256)
      elif name == 'adaptive_ks_n256_b8192':
        # the adaptive leg at the fixed-step leg's batch: 16 rounds of workgroups instead of
        # two, so the spread of the per-sample evaluation counts no longer shows as a tail
        key, val = _adaptive_config(_variant(args, **dict(base, equation='ks', num_points=256)),
                                    name, 8192, t_end=0.02, unique=256)
      elif name == 'ks_n256_b1024':
        # ... and the fixed-step kernel at the adaptive leg's batch (two workgroups per CU)
        key, val = _fixed_step_config(
            args, lib, world, name, 'KS N=256, conv-net stencils, batch 1024 (the batch of the '
            'adaptive_ks_n256_b1024 leg: one round of two workgroups per CU), midpoint', 1024,
            **dict(base, equation='ks', num_points=256, steps=400))
      elif name == 'weno_exact_n512_b2048':
        key, val = _weno_exact_config(args)
      elif name == 'adaptive_kdv_n64_b4096':
        key, val = _adaptive_config(_variant(args, **dict(base, equation='kdv')), name, 4096,
                                    t_end=0.2, unique=1024)
      else:
        key, val = _adaptive_config(_variant(args, **dict(base, equation='ks', num_points=256)),
                                    name, 1024, t_end=0.02, unique=256)
      out[key] = val
    except Exception as exc:   # one broken leg must not cost the headline line
      sys.stderr.write('bench.py: config {} failed: {!r}\n'.format(name, exc))
      out[name] = {'error': repr(exc)}
  return out



def relaunch_under_torchrun(args):
  """`python bench.py --gpus N` outside a torchrun environment: start the N
  ranks ourselves (one process per GPU, rendezvous on 127.0.0.1)."""
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  import tempfile
  log_dir = tempfile.mkdtemp(prefix='bench_ranks_')
  cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1',
         '--nproc-per-node', str(args.gpus), '--master-addr', '127.0.0.1',
         '--master-port', str(port), '--log-dir', log_dir, '--redirects', '2',
         os.path.abspath(__file__)] + sys.argv[1:]
  env = dict(os.environ)
  env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
  rc = subprocess.call(cmd, env=env)
  # every rank's stderr went to its own log file (stdout stays the one JSON line):
  # replayed here rank by rank, so that a dead rank never passes silently
  for path in sorted(glob.glob(os.path.join(log_dir, '**', 'stderr.log'), recursive=True)):
    try:
      with open(path) as f:
        tail = f.read()[-3000:]
    except OSError:
      continue
    if tail.strip():
      sys.stderr.write('---- {} ----\n{}\n'.format(os.path.relpath(path, log_dir), tail))
  if rc != 0:
    sys.stderr.write('bench.py: torch.distributed.run exited with {} for --gpus {} '
                     '(visible devices: see the rank logs above)\n'.format(rc, args.gpus))
  return rc


def main():
  args = parse_args()
  world = int(os.environ.get('WORLD_SIZE', '1'))
  if 'RANK' not in os.environ and args.gpus > 1:
    raise SystemExit(relaunch_under_torchrun(args))
  if world != args.gpus:
    raise SystemExit('--gpus {} but WORLD_SIZE {}'.format(args.gpus, world))
  import torch
  import torch.distributed as dist

  rank = int(os.environ.get('RANK', '0'))
  local_rank = int(os.environ.get('LOCAL_RANK', '0'))
  device = local_rank % torch.cuda.device_count() if args.backend == 'gloo' else local_rank
  torch.cuda.set_device(device)
  if world > 1:
    if args.backend == 'nccl':
      dist.init_process_group('nccl', device_id=torch.device('cuda', device))
    else:
      dist.init_process_group('gloo')

  import ddd1d_amd
  if args.library:
    lib = ddd1d_amd._lib.load_library(os.path.join(
        os.path.dirname(ddd1d_amd._lib.LIBRARY_PATH), 'libddd1d_{}.so'.format(args.library)))
  elif args.debug_option:
    # A/B switches live in the probe flavour of the library only (never a headline run)
    lib = ddd1d_amd._lib.load_probe_library()
  else:
    lib = ddd1d_amd._lib.load_library()   # raises if the HIP extension is missing
  for item in args.debug_option:
    name, _, value = item.partition('=')
    ddd1d_amd._lib.debug_set_option(name, int(value or '1'))
  stages = lib.ddd_scheme_stages(ddd1d_amd._lib.SCHEMES[args.scheme])
  batch = args.batch
  eq, model, forcing, y0_host = build_workload(args, rank, batch)
  dt = eq.time_step
  n = eq.grid.solution_num_points

  m = measure(args, model, y0_host, world, n, batch, sample_clocks=(rank == 0))
  secondary = None
  if world == 1 and args.secondary_batch > 0 and args.secondary_batch != batch:
    b2 = args.secondary_batch
    eq2, model2, _, y02 = build_workload(args, rank, b2)
    m2 = measure(args, model2, y02, world, n, b2)
    s2 = summarize(args, eq2, model2, m2, world, n, b2, stages)
    secondary = {
        'workload': 'BASELINE.json configs[1] batch' if b2 == 1024 else 'secondary batch',
        'batch_per_gpu': b2, 'value': s2['value'], 'unit': 'grid-point-steps/s',
        'ms_per_step': s2['ms_per_step'], 'reps': m2['reps'],
        'fp32_tflops': s2['roofline']['fp32_tflops'],
        'frac': s2['roofline']['frac'], 'frac_wall': s2['roofline']['frac_wall'],
        'kernel_ms_per_launch': s2['roofline']['kernel_ms_per_launch'],
        'kernel': model2.kernel_name, 'finite': m2['finite'],
    }

  configs = None
  if world == 1 and args.configs != 'none':
    configs = extra_configs(args, lib, world)

  if rank == 0:
    s = summarize(args, eq, model, m, world, n, batch, stages)
    dtype = 'f32' if args.state_dtype == 'float32' else 'f32 rhs / f64 state'
    result = {
        'metric': 'grid-point-steps/s',
        'value': s['value'],
        'unit': 'grid-point-steps/s',
        'n_gpus': world,
        'steps': args.steps,
        'warmup': args.warmup,
        'reps': m['reps'],
        'ms_per_step': s['ms_per_step'],
        'higher_is_better': True,
        'scaling': 'weak',
        'vs_baseline': None,
        'dtype': dtype,
        'data': 'synthetic',
        'config': {
            'workload': '{} N={} {} learned-stencil ensemble, batch {}/GPU{}, '
                        '{} steps, {} dt={:g}'.format(
                            args.equation, n,
                            'fixed-stencil' if args.baseline_stencils else 'conv-net',
                            batch,
                            ' (BASELINE configs[4]: {} samples weak-sharded over {} GPUs, '
                            'final states all-gathered over RCCL in the timed region)'
                            .format(batch * world, world) if world > 1 else
                            ' (north_star target; BASELINE configs[1] = batch 1024 under '
                            '"secondary")',
                            args.steps, args.scheme, dt),
            'equation': type(eq).__name__, 'num_points': n,
            'batch_per_gpu': batch, 'global_batch': batch * world,
            'scheme': args.scheme, 'stages': stages, 'dt': dt,
            'launch_mode': args.launch_mode, 'kernel': model.kernel_name,
            'state_dtype': args.state_dtype,
            'fma_per_point_eval': model.fma_per_point,
            'parallelism': 'ensemble-shard x{}'.format(world),
            'backend': args.backend if world > 1 else None,
            'visible_devices': torch.cuda.device_count(),
            'finite': m['finite'],
            'debug_options': args.debug_option, 'library': args.library, 'hparams': json.loads(args.hparams or '{}'),
            'preheat_ms': m['preheat_ms'], 'min_timed_ms': args.min_timed_ms,
            'timed_wall_ms': m['wall'] * 1e3,
        },
        'roofline': s['roofline'],
        'secondary': secondary,
        'configs': configs,
        'per_rank': m['per_rank'],
        'scaling_detail': scaling_detail(args, m, world, n, batch, s['value']),
        'clocks': m['clocks'],
    }
    if not m['finite']:
      result['invalid'] = 'non-finite state after the timed run'

    if world == 1 and args.cpu_seconds > 0:
      result['cpu_baseline'] = cpu_baseline(model, forcing, y0_host, args.scheme,
                                            dt, args.cpu_seconds)
    else:
      result['cpu_baseline'] = None
    print(json.dumps(result))
  if world > 1:
    dist.destroy_process_group()


def measured_traffic(equation, num_points, batch, launch_mode, fixed, state_dtype='float32',
                     hparams=None):
  """(HBM bytes per launch of the dominant kernel, source file) from the
  committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE collected in separate
  runs, gfx950 correction applied; profiles/r*_hbm_traffic.json), or
  (None, None) when this configuration was not profiled.  bench.py cannot
  collect counters itself; the table entry names the command it was measured
  with.  In persistent mode one launch reads y0 + forcing rows + weights and
  writes one snapshot whatever --steps is, so entries are keyed on the
  configuration, not on the step count; per-substep entries are per substep."""
  want = dict(equation=equation, num_points=num_points, batch_per_gpu=batch,
              launch_mode=launch_mode, fixed=bool(fixed))
  if state_dtype != 'float32':
    want['state_dtype'] = state_dtype
  if hparams:
    want['hparams'] = hparams   # (another net: another kernel, other weight traffic)
  for name in TRAFFIC_TABLES:
    path = os.path.join(ROOT, 'profiles', name)
    try:
      with open(path) as f:
        table = json.load(f)
    except (OSError, ValueError):
      continue
    for entry in table.get('entries', []):
      if entry.get('match') == want:
        return (entry['traffic_bytes_per_launch'], 'profiles/' + name,
                entry.get('command', table.get('command')))
  return None, None, None


if __name__ == '__main__':
  main()
