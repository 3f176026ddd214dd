# coding=utf-8
"""Multi-GPU scheme of the hot path: one process per GPU, batch-sharded.

Trajectories are independent units in the forward, so N
ranks run N batch shards with NO data-path collective; torch.distributed
(backend "nccl" == RCCL over xGMI on the GPU box, "gloo" in the CPU tests) is
used only for the timing barrier / max-over-ranks and for collecting
per-shard results on rank 0 when a caller wants the whole batch back.

The training step has the path's one real exchange: the batch means in the
loss (reference code/pred_models.py:995, 1016-1022) couple the samples, so the
per-rank gradients of the local means are summed over the ranks and scaled by
1/world.  The engine keeps every parameter gradient in ONE flat device buffer
(85.4 MB fp32 for both scales).  On RCCL the reduction runs INSIDE the library
(`init_engine_comm` -> mv_allreduce_init: one bucket per ConvLSTM kernel on a side
stream, overlapped with the rest of the backward pass; torch.distributed only
broadcasts the 128-byte unique id); `allreduce_engine_grads` is the round-1 path --
one torch all_reduce on the zero-copy view of that buffer -- kept for gloo (the CPU /
one-GPU tests) and MV_ALLREDUCE=torch.
"""

from __future__ import annotations

import os

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


# ------------------------------------------------------------ training

def world_size():
  import torch.distributed as dist
  if dist.is_available() and dist.is_initialized():
    return dist.get_world_size()
  return 1


def rank():
  import torch.distributed as dist
  if dist.is_available() and dist.is_initialized():
    return dist.get_rank()
  return 0


class _DeviceArray(object):
  """A raw device pointer as a 1-D float32 `__cuda_array_interface__` object."""

  def __init__(self, ptr, n):
    self.__cuda_array_interface__ = {
        "shape": (int(n),), "typestr": "<f4", "data": (int(ptr), False),
        "version": 2, "strides": None}


def engine_grad_tensor(engine, device_index):
  """Zero-copy torch view of the engine's flat gradient buffer."""
  import torch
  ptr, n = engine.grad_buffer()
  return torch.as_tensor(_DeviceArray(ptr, n), device="cuda:%d" % device_index)


def allreduce_engine_grads(engine, device_index=None):
  """SUM all-reduce of the engine's flat gradient buffer over the ranks (RCCL).
  `mv_train_forward_backward` has synchronised the engine's stream before it
  returned; the collective runs on torch's stream and is waited for here, so
  `mv_train_apply` may follow immediately."""
  import torch
  import torch.distributed as dist
  if world_size() == 1:
    return
  if device_index is None:
    device_index = torch.cuda.current_device()
  t = engine_grad_tensor(engine, device_index)
  if dist.get_backend() == "nccl":
    dist.all_reduce(t, op=dist.ReduceOp.SUM)       # RCCL, in place on the engine's buffer
  else:                                            # gloo (single-GPU control-flow checks)
    h = t.cpu()
    dist.all_reduce(h, op=dist.ReduceOp.SUM)
    t.copy_(h)
  torch.cuda.synchronize(device_index)


def init_engine_comm(engine):
  """Give `engine` an RCCL communicator over the ranks of the initialised torch process
  group (used only as the bootstrap: rank 0's 128-byte unique id is broadcast through
  it).  Afterwards `Engine.train_step` all-reduces inside the library, bucketed and
  overlapped with the backward pass (include/multiverse_hip.h mv_allreduce_init).
  Returns False -- and changes nothing -- for world 1, a non-RCCL backend (gloo: several
  ranks may share one GPU there, which RCCL refuses) or MV_ALLREDUCE=torch."""
  import os
  import torch.distributed as dist
  from multiverse_amd import _lib
  mode = os.environ.get("MV_ALLREDUCE", "lib")
  if not (dist.is_available() and dist.is_initialized()):
    return False
  # "lib-force": also on a world of one rank (exercises this bootstrap on a 1-GPU box)
  if (world_size() == 1 and mode != "lib-force") or dist.get_backend() != "nccl" or \
      mode == "torch":
    return False
  box = [_lib.comm_unique_id() if dist.get_rank() == 0 else None]
  dist.broadcast_object_list(box, src=0)
  # ncclCommInitRank blocks until EVERY rank has called it: a rank that died, a wrong world
  # size or a fabric that does not come up would hang the job silently.  It runs under a
  # deadline (MV_COMM_INIT_TIMEOUT_S, default 180 s) and the process ends LOUDLY when it expires.
  deadline = float(os.environ.get("MV_COMM_INIT_TIMEOUT_S", "180"))
  call_with_deadline(
      lambda: engine.comm_init(dist.get_rank(), dist.get_world_size(), box[0]), deadline,
      "mv_allreduce_init (ncclCommInitRank) on rank %d of %d" % (dist.get_rank(),
                                                                 dist.get_world_size()))
  return True


# Called (with the description of what hung) just before a deadline ends the process: bench.py
# puts the emission of what it has measured so far here, so that a stalled in-library RCCL
# bootstrap in a SUB-workload cannot take the already measured headline line down with it.
deadline_hook = None


def call_with_deadline(fn, seconds, what):
  """Run a blocking native call (ctypes releases the GIL) under a deadline.  A call that
  does not come back cannot be cancelled, so on expiry this writes what hung, and why that
  usually happens, to stderr, runs `deadline_hook`, and ends the PROCESS with status 3 -- a
  launcher then tears the other ranks down instead of the job hanging until its wall-clock
  limit."""
  import sys
  import threading
  box = {}

  def run():
    try:
      box["r"] = fn()
    except BaseException as ex:   # pylint: disable=broad-except
      box["e"] = ex
  t = threading.Thread(target=run, daemon=True)
  t.start()
  t.join(seconds)
  if t.is_alive():
    sys.stderr.write(
        "[multiverse_amd] FATAL: %s did not return within %.0f s.  Every rank must reach it "
        "(same world size, one visible GPU per rank, HSA_ENABLE_IPC_MODE_LEGACY=0 for dmabuf "
        "IPC); NCCL_DEBUG=INFO shows where RCCL's bootstrap stands.  Ending this rank "
        "(MV_COMM_INIT_TIMEOUT_S changes the deadline).\n" % (what, seconds))
    sys.stderr.flush()
    if deadline_hook is not None:
      try:
        deadline_hook(what)
      except BaseException:   # pylint: disable=broad-except
        pass
    os._exit(3)
  if "e" in box:
    raise box["e"]
  return box.get("r")


def allreduce_mean_arrays(arrays):
  """In-place mean over ranks of a dict of numpy arrays (gloo / host path; used
  by the CPU tests of the data-parallel gradient rule)."""
  import torch
  import torch.distributed as dist
  w = world_size()
  if w == 1:
    return arrays
  for k in sorted(arrays):
    t = torch.from_numpy(np.ascontiguousarray(arrays[k]))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    arrays[k] = (t / w).numpy()
  return arrays


def mean_over_ranks(loss, parts):
  """Mean of the per-rank loss scalars (reporting only)."""
  import torch
  import torch.distributed as dist
  w = world_size()
  if w == 1:
    return loss, parts
  dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
  t = torch.tensor([loss] + list(parts), dtype=torch.float64, device=dev)
  dist.all_reduce(t, op=dist.ReduceOp.SUM)
  t = (t / w).tolist()
  return t[0], t[1:]
