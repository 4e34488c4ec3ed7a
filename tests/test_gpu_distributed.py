"""HIP kernel + collective together (VERDICT r1 "what's weak" 3): two ranks each
integrate THEIR shard of a forced Burgers ensemble on the GPU and gather the
final states; the result must equal the single-process run of the whole
ensemble bit for bit (samples are independent; scripts/run_evaluation.py:212-221,
xarray_beam.py:127-154 is the reference's gather).

  * backend "nccl" (RCCL, device tensors): needs two GPUs -- RCCL refuses two
    ranks on one device -- and is skipped, loudly, on a one-GPU box;
  * backend "gloo": both ranks share cuda:0 for the kernel and gather the slabs
    through host memory; runs on any GPU box.
"""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT

pytestmark = pytest.mark.gpu

TOTAL, STEPS = 13, 40     # ragged split: 7 + 6 samples


def _ensemble(lo, hi):
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  from helpers import make_model, random_phase_ic
  from ddd1d_amd import model as model_lib
  model = make_model('burgers', True, num_points=64, resample_factor=8)
  forcing = model_lib.batched_forcing_parameters(range(lo, hi), nparams=20)
  y0 = random_phase_ic(model.equation, hi - lo, seed0=1000 + lo)
  model.set_forcing(forcing)
  return model, y0


def _worker(rank, world, port, backend, tmpdir):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                    WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                    HSA_ENABLE_IPC_MODE_LEGACY='0')
  from ddd1d_amd import distributed
  device = rank % torch.cuda.device_count()
  torch.cuda.set_device(device)
  if backend == 'nccl':
    dist.init_process_group('nccl', rank=rank, world_size=world,
                            device_id=torch.device('cuda', device))
  else:
    dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    lo, hi = distributed.shard_bounds(TOTAL, rank, world)
    model, y0 = _ensemble(lo, hi)
    final = model.integrate_fixed(y0, STEPS, dt=1e-3, save_every=STEPS)[0]   # HIP kernel
    assert model.kernel_name.startswith('mfma_f32')
    local = final if backend == 'nccl' else final.cpu()
    gathered = distributed.gather_states(local, total=TOTAL)                 # the collective
    if rank == 0:
      np.save(os.path.join(tmpdir, 'gathered.npy'), gathered.cpu().numpy())
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('backend', ['gloo', 'nccl'])
def test_two_ranks_hip_kernel_and_gather(tmp_path, backend):
  if backend == 'nccl' and torch.cuda.device_count() < 2:
    pytest.skip('RCCL needs one GPU per rank; this box has {}'.format(
        torch.cuda.device_count()))
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  mp.spawn(_worker, args=(2, port, backend, str(tmp_path)), nprocs=2, join=True)
  gathered = np.load(os.path.join(str(tmp_path), 'gathered.npy'))
  model, y0 = _ensemble(0, TOTAL)
  want = model.integrate_fixed(y0, STEPS, dt=1e-3, save_every=STEPS)[0].cpu().numpy()
  assert gathered.shape == (TOTAL, 64)
  np.testing.assert_array_equal(gathered, want)


def test_bench_two_ranks_on_this_box():
  """bench.py's N > 1 code path (self-launch under torch.distributed.run, agreed
  repetition count, gather inside the timed region, max over ranks) on whatever
  this box has: two ranks sharing cuda:0 with the gloo backend.  Not a
  performance number -- the ranks compete for one GPU."""
  env = {k: v for k, v in os.environ.items()
         if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2',
                        '--backend', 'gloo', '--steps', '20', '--warmup', '5', '--batch', '512',
                        '--preheat-ms', '20', '--min-timed-ms', '5', '--cpu-seconds', '0'],
                       env=env, capture_output=True, text=True, timeout=900)
  assert out.returncode == 0, out.stderr[-3000:]
  line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
  result = json.loads(line)
  assert result['n_gpus'] == 2 and result['config']['global_batch'] == 1024
  assert result['config']['backend'] == 'gloo' and result['config']['finite']
  assert result['steps'] == 20 and result['reps'] >= 1 and result['value'] > 0
  assert result['cpu_baseline'] is None and result['secondary'] is None
  # every rank's own times and the isolated gather cost ride in the line
  per_rank = result['per_rank']
  assert len(per_rank['kernel_ms']) == 2 and len(per_rank['wall_ms']) == 2
  assert len(per_rank['gather_ms_isolated']) == 2 and per_rank['gather_bytes_per_rank'] == 512 * 64 * 4
  assert result['configs'] is None


def test_bench_eight_ranks_on_this_box():
  """The command shape of the first 8-GPU run -- `bench.py --gpus 8`, self-launched
  under torch.distributed.run -- with eight ranks sharing cuda:0 over gloo: the
  8-rank rendezvous, the agreed repetition count, per-rank tables of length 8 and
  the gathered ensemble, which must equal the single-process run of the same 512
  global sample ids bit for bit."""
  import hashlib
  env = {k: v for k, v in os.environ.items()
         if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '8',
                        '--backend', 'gloo', '--batch', '64', '--steps', '5', '--warmup', '2',
                        '--preheat-ms', '10', '--min-timed-ms', '5', '--configs', 'none',
                        '--cpu-seconds', '0'], env=env, capture_output=True, text=True,
                       timeout=1200)
  assert out.returncode == 0, out.stderr[-3000:]
  line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
  result = json.loads(line)
  assert result['n_gpus'] == 8 and result['config']['global_batch'] == 512
  assert result['config']['visible_devices'] == torch.cuda.device_count()
  assert result['config']['finite'] and result['value'] > 0 and result['scaling'] == 'weak'
  per_rank = result['per_rank']
  for key in ('wall_ms', 'kernel_ms', 'gather_ms_isolated'):
    assert len(per_rank[key]) == 8 and all(v > 0 for v in per_rank[key]), key
  assert per_rank['gathered_shape'] == [512, 64]
  # the 1 -> N point measured in the same invocation: rank 0 alone on its shard
  detail = result['scaling_detail']
  assert detail['nranks'] == 8 and detail['backend'] == 'gloo' and detail['n1_batch'] == 64
  assert detail['n1_value'] > 0 and detail['n1_ms_per_step'] > 0
  assert abs(detail['efficiency'] - result['value'] / (8 * detail['n1_value'])) < 1e-9
  # the single-process run of global sample ids 0 .. 511 (rank r owns [64 r, 64 r + 64))
  import bench
  args = bench.parse_args(['--batch', '512', '--steps', '5'])
  eq, model, _, y0 = bench.build_workload(args, 0, 512)
  want = model.integrate_fixed(y0, 5, dt=eq.time_step, scheme='midpoint', save_every=5)[0]
  assert per_rank['gathered_sha1'] == hashlib.sha1(want.cpu().numpy().tobytes()).hexdigest()


def test_bench_reports_a_dead_rank(tmp_path):
  """A rank that dies must fail the whole command, with that rank's own stderr."""
  env = {k: v for k, v in os.environ.items()
         if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2',
                        '--backend', 'gloo', '--batch', '64', '--steps', '5',
                        '--configs', 'none', '--cpu-seconds', '0', '--equation', 'nonsense'],
                       env=env, capture_output=True, text=True, timeout=600)
  assert out.returncode != 0
  assert 'stderr.log' in out.stderr and 'torch.distributed.run exited with' in out.stderr


def test_bench_rccl_path_with_one_rank():
  """The RCCL calls bench.py makes with N > 1 -- init with device_id, broadcast of
  the repetition count, async all_gather_into_tensor of a [batch, x] slab into
  a [world * batch, x] buffer, wait, all_reduce(MAX), barrier -- on DEVICE
  tensors with backend "nccl" in a world of ONE rank, which one GPU can host.
  Catches API / shape mistakes before the first multi-GPU run."""
  code = r"""
import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='%d', RANK='0', WORLD_SIZE='1',
                  LOCAL_RANK='0')
torch.cuda.set_device(0)
dist.init_process_group('nccl', device_id=torch.device('cuda', 0))
r = torch.tensor([7], dtype=torch.int64, device='cuda'); dist.broadcast(r, 0)
final = torch.arange(512 * 64, dtype=torch.float32, device='cuda').reshape(1, 512, 64)
gathered = torch.empty((1 * 512, 64), dtype=torch.float32, device='cuda')
work = dist.all_gather_into_tensor(gathered, final[0], async_op=True)
work.wait()
t = torch.tensor([1.5, 2.5], dtype=torch.float64, device='cuda')
dist.all_reduce(t, op=dist.ReduceOp.MAX)
torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
assert int(r[0]) == 7 and torch.equal(gathered, final[0]) and float(t[1]) == 2.5
dist.destroy_process_group()
print('RCCL_ONE_RANK_OK')
"""
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
  out = subprocess.run([sys.executable, '-c', code % port], env=env, capture_output=True,
                       text=True, timeout=600)
  assert out.returncode == 0 and 'RCCL_ONE_RANK_OK' in out.stdout, out.stderr[-3000:]


def test_bench_self_launches_its_ranks():
  """`python bench.py --gpus 2` without a torchrun environment starts its own
  ranks (bench.py: relaunch_under_torchrun) and prints one JSON line."""
  if torch.cuda.device_count() < 2:
    pytest.skip('needs two GPUs; this box has {}'.format(torch.cuda.device_count()))
  env = {k: v for k, v in os.environ.items()
         if k not in ('RANK', 'WORLD_SIZE', 'LOCAL_RANK', 'MASTER_ADDR', 'MASTER_PORT')}
  out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2',
                        '--steps', '20', '--warmup', '5', '--batch', '2048',
                        '--cpu-seconds', '0'], env=env, capture_output=True, text=True,
                       timeout=900)
  assert out.returncode == 0, out.stderr[-2000:]
  line = [l for l in out.stdout.splitlines() if l.startswith('{')][-1]
  result = json.loads(line)
  assert result['n_gpus'] == 2 and result['config']['global_batch'] == 4096
  assert result['config']['finite'] and result['value'] > 0


def _eval_worker(rank, world, port, tmpdir):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank),
                    WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                    HSA_ENABLE_IPC_MODE_LEGACY='0')
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  from test_gpu_evaluation import _setup
  from helpers import random_phase_ic
  from ddd1d_amd import evaluation
  torch.cuda.set_device(0)
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    hp, model = _setup('burgers', 7)
    y0 = 0.3 * random_phase_ic(model.equation, 7)
    times = np.arange(0, 0.2 + 1e-9, 0.1)
    out = evaluation.run_integrate_batch(model, hp, y0, times)   # shard, adaptive RK23, gather
    if rank == 0:
      np.savez(os.path.join(tmpdir, 'eval.npz'), y=out['y'], num_evals=out['num_evals'],
               sample=out['sample'])
  finally:
    dist.destroy_process_group()


def test_two_ranks_evaluation_harness_adaptive(tmp_path):
  """run_integrate_batch under two ranks (the analogue of the reference's Beam
  fan-out + ConcatCombineFn('sample'), run_evaluation.py:212-221): each rank
  integrates its ragged slab (4 + 3 samples) with the on-device adaptive RK23 and
  its own seeds' forcing; trajectories AND per-sample evaluation counts are
  gathered and equal the single-process run."""
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  mp.spawn(_eval_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
  got = np.load(os.path.join(str(tmp_path), 'eval.npz'))
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  from test_gpu_evaluation import _setup
  from helpers import random_phase_ic
  from ddd1d_amd import evaluation
  hp, model = _setup('burgers', 7)
  y0 = 0.3 * random_phase_ic(model.equation, 7)
  want = evaluation.run_integrate_batch(model, hp, y0, np.arange(0, 0.2 + 1e-9, 0.1))
  assert got['y'].shape == (7, 3, 64) and got['y'].dtype == np.float64
  np.testing.assert_array_equal(got['y'], want['y'])
  np.testing.assert_array_equal(got['num_evals'], want['num_evals'])
  np.testing.assert_array_equal(got['sample'], np.arange(7))
