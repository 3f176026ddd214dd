# coding=utf-8
"""CPU, world_size 2 over gloo: the batch-sharded multi-GPU scheme of
bench.py / multiverse_amd.parallel -- no data-path collective, shards
concatenate to the single-process result, timing is the MAX over ranks."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _worker(rank, world, port, outdir):
  sys.path.insert(0, ROOT)
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  torch.set_num_threads(2)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  from multiverse_amd import parallel, synth
  from oracle import multiverse_oracle as oracle
  N = 3                                    # uneven split: 2 + 1
  cfg = synth.default_config(batch_size=N, use_grids=(0, 1))
  params = synth.make_params(cfg, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=21)
  shard, (lo, hi) = parallel.shard_feed(feed, rank, world, N)
  scfg = synth.default_config(batch_size=hi - lo, use_grids=(0, 1))
  cls, reg, _ = oracle.forward(params, scfg, shard)      # stands in for the engine
  full_cls = parallel.gather_to_rank0(cls[1])
  full_reg = parallel.gather_to_rank0(reg[1])
  t = parallel.max_over_ranks(1.0 + rank)
  assert t == float(world)
  dist.barrier()
  if rank == 0:
    ref_cls, ref_reg, _ = oracle.forward(params, cfg, feed)
    np.savez(os.path.join(outdir, "res.npz"),
             dc=np.abs(full_cls - ref_cls[1]).max(), dr=np.abs(full_reg - ref_reg[1]).max(),
             n=full_cls.shape[0])
  dist.destroy_process_group()


def test_shard_range_partitions():
  from multiverse_amd import parallel
  for n in (1, 7, 64, 65):
    for world in (1, 2, 4, 8):
      spans = [parallel.shard_range(n, r, world) for r in range(world)]
      assert spans[0][0] == 0 and spans[-1][1] == n
      assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
      sizes = [b - a for a, b in spans]
      assert max(sizes) - min(sizes) <= 1


def test_two_rank_batch_sharding_matches_single_process(tmp_path):
  world = 2
  mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
  r = np.load(os.path.join(str(tmp_path), "res.npz"))
  assert int(r["n"]) == 3
  assert float(r["dc"]) < 1e-5 and float(r["dr"]) < 1e-5


def _grad_worker(rank, world, port, outdir):
  sys.path.insert(0, ROOT)
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  torch.set_num_threads(2)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  from multiverse_amd import parallel, synth
  from oracle import multiverse_oracle as oracle
  N = 4
  cfg = synth.default_config(batch_size=N, use_grids=(0, 1), is_train=True)
  params = synth.make_params(cfg, recurrent_gain=2.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=33)
  shard, (lo, hi) = parallel.shard_feed(feed, rank, world, N)
  scfg = synth.default_config(batch_size=hi - lo, use_grids=(0, 1), is_train=True)
  # the oracle stands in for mv_train_forward_backward on this rank's shard
  loss, wd, pgl, grads = oracle.loss_and_grads(params, scfg, shard)
  grads = parallel.allreduce_mean_arrays(grads)          # sum over ranks / world
  loss, pgl = parallel.mean_over_ranks(loss, pgl)
  if rank == 0:
    rloss, _, rpgl, rgrads = oracle.loss_and_grads(params, cfg, feed)
    worst = max(float(np.abs(grads[k] - rgrads[k]).max() /
                      max(np.abs(rgrads[k]).max(), 1e-30)) for k in rgrads)
    np.savez(os.path.join(outdir, "grad.npz"), worst=worst, dloss=abs(loss - rloss),
             dpgl=np.abs(np.asarray(pgl) - np.asarray(rpgl)).max())
  dist.barrier()
  dist.destroy_process_group()


def test_two_rank_gradient_allreduce_matches_single_process(tmp_path):
  """The data-parallel rule of the training step: per-rank gradients of the
  LOCAL batch-mean loss, all-reduce(sum)/world == tf.gradients of the
  global-batch loss (equal shards)."""
  world = 2
  mp.spawn(_grad_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world,
           join=True)
  r = np.load(os.path.join(str(tmp_path), "grad.npz"))
  assert float(r["worst"]) < 1e-4, float(r["worst"])
  assert float(r["dloss"]) < 1e-3 and float(r["dpgl"]) < 1e-3


# ---- the shared-memory RCCL stand-in of the two-ranks-on-one-GPU test (tests/fake_rccl)

_FAKE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fake_rccl", "libfakerccl.so")


@pytest.mark.skipif(not os.path.exists(_FAKE), reason="tests/fake_rccl not built (__graft_entry__.build())")
def test_fake_rccl_exports_what_comm_h_binds():
  """TEST INFRASTRUCTURE check: the stand-in carries the seven entry points
  multiverse_amd/csrc/comm.h resolves by dlsym, ids are unique and tagged, group calls nest,
  a world of one attaches without a peer, and foreign ids / dtypes are refused."""
  import ctypes as C
  lib = C.CDLL(_FAKE)
  for name in ("ncclGetUniqueId", "ncclCommInitRank", "ncclCommDestroy", "ncclAllReduce",
               "ncclGroupStart", "ncclGroupEnd", "ncclGetErrorString"):
    assert getattr(lib, name)
  lib.ncclGetErrorString.restype = C.c_char_p
  ids = []
  for _ in range(2):
    buf = (C.c_uint8 * 128)()
    assert lib.ncclGetUniqueId(buf) == 0
    assert bytes(buf)[24:30] == b"mvfake"
    ids.append(bytes(buf))
  assert ids[0] != ids[1]
  assert lib.ncclGroupEnd() != 0                       # not inside a group
  assert lib.ncclGroupStart() == 0 and lib.ncclGroupStart() == 0
  assert lib.ncclGroupEnd() == 0 and lib.ncclGroupEnd() == 0 and lib.ncclGroupEnd() != 0

  class Id(C.Structure):
    _fields_ = [("internal", C.c_uint8 * 128)]
  comm = C.c_void_p()
  uid = Id.from_buffer_copy(ids[0])
  assert lib.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0 and comm.value
  # float + sum only; a NULL buffer is an argument error, not a crash
  assert lib.ncclAllReduce(None, None, C.c_size_t(4), 7, 0, comm, None) != 0
  assert b"fake rccl" in lib.ncclGetErrorString(5)
  assert lib.ncclCommDestroy(comm) == 0
  foreign = Id()                                       # an id this library did not draw
  assert lib.ncclCommInitRank(C.byref(comm), 1, foreign, 0) != 0
  assert lib.ncclCommInitRank(C.byref(comm), 2, uid, 2) != 0      # rank out of range


_HOOKS = os.path.join(os.path.dirname(_FAKE), "libmultiverse_hip_testhooks.so")


@pytest.mark.skipif(not (os.path.exists(_FAKE) and os.path.exists(_HOOKS)),
                    reason="tests/fake_rccl not built (__graft_entry__.build())")
def test_mv_rccl_lib_is_a_hook_of_the_test_build_only():
  """MV_RCCL_LIB (comm.h) exists ONLY in the -DMV_TEST_HOOKS copy of the library (selected
  with MV_LIB_PATH): there it binds exactly the named RCCL -- here the stand-in, recognisable
  by the tag in the unique id it hands out -- and reports a missing file.  The SHIPPED library
  does not read the variable at all: its unique id never carries the stand-in's tag."""
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  code = ("import sys; sys.path.insert(0, %r)\n"
          "from multiverse_amd import _lib\n"
          "print(_lib.comm_unique_id()[24:30])\n" % root)
  hooks = dict(os.environ, MV_LIB_PATH=_HOOKS)
  r = subprocess.run([sys.executable, "-c", code], env=dict(hooks, MV_RCCL_LIB=_FAKE),
                     capture_output=True)
  assert r.returncode == 0 and b"mvfake" in r.stdout
  assert b"TEST-HOOKS BUILD: collectives bound to" in r.stderr      # announced, never silent
  r = subprocess.run([sys.executable, "-c", code],
                     env=dict(hooks, MV_RCCL_LIB="/nonexistent/librccl.so"), capture_output=True)
  assert r.returncode != 0 and b"MV_RCCL_LIB" in r.stderr
  # the shipped library: the variable is not a knob -- whatever it binds (the real RCCL, or
  # nothing on a box without one), it is not the stand-in
  env = dict(os.environ, MV_RCCL_LIB=_FAKE)
  env.pop("MV_LIB_PATH", None)
  r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True)
  assert b"mvfake" not in r.stdout and b"TEST-HOOKS" not in r.stderr


def test_comm_init_deadline_ends_the_process_loudly():
  """parallel.call_with_deadline (wrapped around mv_allreduce_init / ncclCommInitRank): a
  native call that never returns must end the rank with status 3 and say what hung -- the first
  real 8-GPU run must FAIL, not hang, if RCCL's bootstrap stalls; a call that returns in time
  hands its value (or its exception) through."""
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  code = ("import sys, time; sys.path.insert(0, %r)\n"
          "from multiverse_amd import parallel\n"
          "assert parallel.call_with_deadline(lambda: 7, 5.0, 'quick') == 7\n"
          "try:\n"
          "  parallel.call_with_deadline(lambda: 1 / 0, 5.0, 'raises')\n"
          "  raise SystemExit(9)\n"
          "except ZeroDivisionError:\n"
          "  pass\n"
          "parallel.deadline_hook = lambda what: print('HOOK: ' + what, flush=True)\n"
          "parallel.call_with_deadline(lambda: time.sleep(30), 0.3, 'stuck ncclCommInitRank')\n"
          "raise SystemExit(8)\n" % root)
  r = subprocess.run([sys.executable, "-c", code], capture_output=True, timeout=60)
  assert r.returncode == 3, (r.returncode, r.stderr)
  # the hook ran before the process ended (bench.py emits its measured headline there)
  assert b"HOOK: stuck ncclCommInitRank" in r.stdout
  assert b"FATAL: stuck ncclCommInitRank did not return within 0 s" in r.stderr
  assert b"MV_COMM_INIT_TIMEOUT_S" in r.stderr
