"""world_size-2 gloo test of the ensemble sharding + final gather (no GPU).

Each rank draws ITS shard of the per-sample forcing / initial conditions from
global sample ids, advances it with the CPU oracle standing in for the kernel
(the test is about the sharding and the collective, not the kernel), and the
slabs are gathered; the result must equal the single-process run of the whole
ensemble.
"""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from helpers import ROOT


def _worker(rank, world, port, total, tmpdir):
  os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port),
                    RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
  sys.path.insert(0, os.path.join(ROOT, 'tests'))
  from helpers import oracle, make_model, random_phase_ic
  from ddd1d_amd import distributed, model as model_lib
  dist.init_process_group('gloo', rank=rank, world_size=world)
  try:
    assert distributed.world_info() == (rank, rank, world)
    lo, hi = distributed.shard_bounds(total, rank, world)
    model = make_model('burgers', True, num_points=32, resample_factor=2)
    forcing = model_lib.batched_forcing_parameters(range(lo, hi), nparams=20)
    y0 = random_phase_ic(model.equation, hi - lo, seed0=1000 + lo)
    final = oracle.integrate_fixed(model.spec(), oracle.SCHEME_MIDPOINT, 0.0,
                                   1e-3, 3, 3, y0, forcing=forcing)[0]
    gathered = distributed.gather_states(torch.from_numpy(final), total=total)
    # without `total` the slab sizes are exchanged: ragged slabs still gather in order
    unsized = distributed.gather_states(torch.from_numpy(final))
    assert torch.equal(unsized, gathered)
    if rank == 0:
      np.save(os.path.join(tmpdir, 'gathered.npy'), gathered.numpy())
  finally:
    dist.destroy_process_group()


@pytest.mark.parametrize('total', [6, 7])   # equal and ragged slabs
def test_two_rank_shard_and_gather(tmp_path, total):
  import socket
  with socket.socket() as s:
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
  mp.spawn(_worker, args=(2, port, total, str(tmp_path)), nprocs=2, join=True)
  gathered = np.load(os.path.join(str(tmp_path), 'gathered.npy'))
  from helpers import oracle, make_model, random_phase_ic
  from ddd1d_amd import model as model_lib
  model = make_model('burgers', True, num_points=32, resample_factor=2)
  forcing = model_lib.batched_forcing_parameters(range(total), nparams=20)
  y0 = random_phase_ic(model.equation, total, seed0=1000)
  want = oracle.integrate_fixed(model.spec(), oracle.SCHEME_MIDPOINT, 0.0, 1e-3,
                                3, 3, y0, forcing=forcing)[0]
  assert gathered.shape == (total, 32)
  np.testing.assert_array_equal(gathered, want)
