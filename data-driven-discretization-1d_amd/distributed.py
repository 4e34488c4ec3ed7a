"""Ensemble sharding across the GPUs of one node (one process per GPU).

The batch of independent initial conditions is split into contiguous slabs,
one per rank; ranks never communicate while stepping.  The only collective is
the final gather of the per-rank result slabs -- the analogue of the
reference's ``beam.CombineGlobally(ConcatCombineFn('sample'))``
(scripts/run_evaluation.py:218, xarray_beam.py:127-154) -- done with
``torch.distributed`` (backend "nccl" = RCCL over xGMI on GPUs, "gloo" on CPU
tensors in the tests).
"""
import os
from typing import Optional, Tuple


def world_info() -> Tuple[int, int, int]:
  """(rank, local_rank, world_size) from the torchrun environment."""
  return (int(os.environ.get('RANK', '0')), int(os.environ.get('LOCAL_RANK', '0')),
          int(os.environ.get('WORLD_SIZE', '1')))


def shard_bounds(total: int, rank: int, world_size: int) -> Tuple[int, int]:
  """Contiguous, balanced slab [start, stop) of ``total`` samples for ``rank``.

  The first ``total % world_size`` ranks get one extra sample, so slabs differ
  by at most one and concatenating them in rank order restores sample order.
  """
  if not 0 <= rank < world_size:
    raise ValueError('rank {} outside world of {}'.format(rank, world_size))
  base, extra = divmod(total, world_size)
  start = rank * base + min(rank, extra)
  return start, start + base + (1 if rank < extra else 0)


def weak_shard_ids(per_rank: int, rank: int):
  """Weak scaling: every rank owns ``per_rank`` samples with global ids."""
  return range(rank * per_rank, (rank + 1) * per_rank)


def gather_states(local, total: Optional[int] = None, group=None):
  """All-gather per-rank slabs [b_r, ...] into [sum b_r, ...] in rank order.

  With ``total`` every rank derives the slab sizes from ``shard_bounds`` (the
  partition all callers use) and nothing is exchanged: equal slabs use ONE
  ``all_gather_into_tensor`` -- no pickled object collective, no host
  synchronisation --; ragged slabs (total not divisible by the world size) are
  padded to the largest slab and trimmed.  Without ``total`` the slab sizes are
  exchanged first (one all-gather of one int64 per rank, then a host read), so
  ragged slabs are gathered correctly instead of corrupting an undersized buffer.
  Works on any backend (RCCL for CUDA tensors, gloo for CPU tensors).
  """
  import torch
  import torch.distributed as dist
  if not dist.is_available() or not dist.is_initialized():
    return local
  world = dist.get_world_size(group)
  if world == 1:
    return local
  if total is None:
    mine = torch.tensor([int(local.shape[0])], dtype=torch.int64, device=local.device)
    everyone = torch.empty(world, dtype=torch.int64, device=local.device)
    dist.all_gather_into_tensor(everyone, mine, group=group)
    sizes = [int(v) for v in everyone.cpu()]
  else:
    bounds = [shard_bounds(total, r, world) for r in range(world)]
    sizes = [hi - lo for lo, hi in bounds]
    mine = sizes[dist.get_rank(group)]
    if int(local.shape[0]) != mine:
      raise ValueError('this rank holds {} samples, shard_bounds({}, ..) assigns it {}'
                       .format(int(local.shape[0]), total, mine))
  if len(set(sizes)) == 1:
    out = torch.empty((world * sizes[0],) + tuple(local.shape[1:]),
                      dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous(), group=group)
    return out
  largest = max(sizes)
  padded = torch.zeros((largest,) + tuple(local.shape[1:]), dtype=local.dtype,
                       device=local.device)
  padded[:local.shape[0]] = local
  pieces = [torch.empty_like(padded) for _ in range(world)]
  dist.all_gather(pieces, padded, group=group)
  return torch.cat([p[:s] for p, s in zip(pieces, sizes)], dim=0)
