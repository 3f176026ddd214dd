# coding=utf-8
"""Multi-GPU scheme of the hot path: one process per GPU, batch-sharded.

Trajectories are independent units in the forward, so N
ranks run N batch shards with NO data-path collective; torch.distributed
(backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests) is
used only for the timing barrier / max-over-ranks and for collecting
per-shard results on rank 0 when a caller wants the whole batch back.
"""

from __future__ import annotations

import numpy as np


def shard_range(n_items, rank, world):
  """Contiguous slice [lo, hi) of `n_items` owned by `rank`; sizes differ by at
  most one and every item belongs to exactly one rank."""
  base, rem = divmod(n_items, world)
  lo = rank * base + min(rank, rem)
  return lo, lo + base + (1 if rank < rem else 0)


def shard_feed(feed, rank, world, batch_size):
  """Slice a global-batch feed (as built by pred_models.build_feed_dict) into
  this rank's shard.  The scene table is replicated (25 KB per frame)."""
  lo, hi = shard_range(batch_size, rank, world)
  out = dict(feed)
  out["obs_scene"] = feed["obs_scene"][lo:hi]
  for key in ("grid_obs_labels", "grid_obs_regress", "grid_pred_labels",
              "grid_pred_regress"):
    if key in feed:
      out[key] = [None if a is None else a[lo:hi] for a in feed[key]]
  return out, (lo, hi)


def max_over_ranks(value, device=None):
  """MAX all-reduce of a python float (bench timing contract)."""
  import torch
  import torch.distributed as dist
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
    return float(value)
  t = torch.tensor([float(value)], dtype=torch.float64, device=device or "cpu")
  dist.all_reduce(t, op=dist.ReduceOp.MAX)
  return float(t.item())


def gather_to_rank0(array):
  """Concatenate per-rank numpy arrays along axis 0 on rank 0 (None elsewhere)."""
  import torch.distributed as dist
  if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
    return array
  parts = [None] * dist.get_world_size() if dist.get_rank() == 0 else None
  dist.gather_object(array, parts, dst=0)
  if dist.get_rank() != 0:
    return None
  return np.concatenate(parts, axis=0)
