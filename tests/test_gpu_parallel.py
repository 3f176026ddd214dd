# coding=utf-8
"""GPU: the multi-rank path with the ENGINE on every rank.

Two processes share the one GPU of the test box (RCCL refuses duplicate devices, so the
process group is gloo -- the same `parallel.allreduce_engine_grads` call the RCCL job
makes, with the reduction staged through the host); each rank owns one batch shard:

  training   mv_train_forward_backward(shard) -> all-reduce(SUM) of the flat gradient
             buffer -> mv_train_apply(1/world)   ==   one process, mv_train_step on the
             global batch: parameters after two steps (incl. the LR staircase, which
             counts GLOBAL batches), Adadelta slots and global_step;
  inference  the shards' outputs concatenate BITWISE to the single-process forward.
"""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_GLOBAL, STEPS = 4, 2


def _free_port():
  s = socket.socket()
  s.bind(("127.0.0.1", 0))
  port = s.getsockname()[1]
  s.close()
  return port


def _cfg(batch, synth, is_train):
  cfg = synth.default_config(batch_size=batch, use_grids=(1, 1), is_train=is_train)
  cfg.train_num_examples = N_GLOBAL       # decay_steps = 2 global batches: LR moves at step 2
  cfg.num_epoch_per_decay = 2.0
  return cfg


def _worker(rank, world, port, outdir):
  sys.path.insert(0, ROOT)
  os.environ["MASTER_ADDR"] = "127.0.0.1"
  os.environ["MASTER_PORT"] = str(port)
  os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  import torch
  import torch.distributed as dist
  torch.cuda.set_device(0)
  dist.init_process_group("gloo", rank=rank, world_size=world)
  from multiverse_amd import _lib, parallel, synth
  gcfg = _cfg(N_GLOBAL, synth, True)
  params = synth.make_params(gcfg, seed=synth.SEED_BASE + 3, recurrent_gain=2.0,
                             bias_scale=0.1)
  lo, hi = parallel.shard_range(N_GLOBAL, rank, world)
  scfg = _cfg(hi - lo, synth, True)
  eng = _lib.Engine(scfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode("f16x3")
  eng.train_init(world=world)
  losses = []
  for step in range(STEPS):
    feed = synth.make_feed(gcfg, seed=synth.SEED_BASE + 200 + step)
    shard, _ = parallel.shard_feed(feed, rank, world, N_GLOBAL)
    loss, wd, pgl = eng.train_forward_backward(shard)
    parallel.allreduce_engine_grads(eng, 0)
    eng.train_apply(1.0 / world)
    loss, pgl = parallel.mean_over_ranks(loss, pgl)
    losses.append([loss, wd] + list(pgl))
  assert eng.global_step == STEPS
  out = {n: eng.get_param(n) for n, _ in eng.param_specs()}
  out.update({"slot0|" + n: eng.get_opt_slot(n, 0) for n, _ in eng.param_specs()})
  eng.close()
  # inference on the trained weights: this rank's shard of a fresh batch, on an inference
  # engine (a training engine keeps the dense x operand of the class chains for its backward
  # pass; the inference engine folds it into table terms -- same values, other rounding)
  feed = synth.make_feed(gcfg, seed=synth.SEED_BASE + 300)
  shard, _ = parallel.shard_feed(feed, rank, world, N_GLOBAL)
  ieng = _lib.Engine(_cfg(hi - lo, synth, False), device=0)
  ieng.set_params({n: out[n] for n in params})
  ieng.set_compute_mode("f16x3")
  cls, reg = ieng.forward_greedy(shard)
  ieng.close()
  full = [parallel.gather_to_rank0(a) for a in (cls[0], cls[1], reg[0], reg[1])]
  if rank == 0:
    out["losses"] = np.asarray(losses)
    for k, a in zip(("cls0", "cls1", "reg0", "reg1"), full):
      out[k] = a
    np.savez(os.path.join(outdir, "dp.npz"), **out)
  dist.barrier()
  dist.destroy_process_group()


def test_two_ranks_one_gpu_engine_equals_single_process(built_lib, tmp_path):
  from multiverse_amd import synth
  world = 2
  mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
  dp = np.load(os.path.join(str(tmp_path), "dp.npz"))
  gcfg = _cfg(N_GLOBAL, synth, True)
  params = synth.make_params(gcfg, seed=synth.SEED_BASE + 3, recurrent_gain=2.0,
                             bias_scale=0.1)
  eng = built_lib.Engine(gcfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode("f16x3")
  eng.train_init()
  for step in range(STEPS):
    feed = synth.make_feed(gcfg, seed=synth.SEED_BASE + 200 + step)
    loss, wd, pgl = eng.train_step(feed)
    ref = np.asarray([loss, wd] + list(pgl))
    print("step %d: single-process %s | 2 ranks (mean) %s" % (step, ref, dp["losses"][step]))
    assert np.allclose(dp["losses"][step], ref, rtol=2e-6, atol=1e-7)
  worst = 0.0
  for n, _ in eng.param_specs():
    ref = eng.get_param(n)
    upd = max(float(np.abs(ref - params[n]).max()), 1e-12)
    d = float(np.abs(dp[n] - ref).max())
    worst = max(worst, d / max(1.0, float(np.abs(ref).max())))
    assert d <= 1e-6 * max(1.0, float(np.abs(ref).max())) and d <= 2e-3 * upd, (n, d, upd)
    s_ref = eng.get_opt_slot(n, 0)
    assert np.abs(dp["slot0|" + n] - s_ref).max() <= 1e-4 * max(np.abs(s_ref).max(), 1e-30)
  print("parameters after %d data-parallel steps: max |2 ranks - 1 process| = %.2e"
        % (STEPS, worst))
  # inference: shards concatenate bitwise (same kernels, same per-row order)
  trained = {n: dp[n] for n, _ in eng.param_specs()}
  eng.close()
  icfg = _cfg(N_GLOBAL, synth, False)
  ieng = built_lib.Engine(icfg, device=0)
  ieng.set_params(trained)
  ieng.set_compute_mode("f16x3")
  cls, reg = ieng.forward_greedy(synth.make_feed(icfg, seed=synth.SEED_BASE + 300))
  ieng.close()
  assert (cls[0] == dp["cls0"]).all() and (cls[1] == dp["cls1"]).all()
  assert (reg[0] == dp["reg0"]).all() and (reg[1] == dp["reg1"]).all()


def test_in_library_rccl_allreduce_world1_is_the_identity(built_lib):
  """mv_allreduce_init + the bucketed side-stream all-reduce inside mv_train_step, on a
  world of one rank (RCCL refuses two ranks on one GPU): same bits as without a
  communicator, every element of the gradient buffer covered exactly once."""
  from multiverse_amd import synth
  cfg = _cfg(2, synth, True)
  params = synth.make_params(cfg, seed=synth.SEED_BASE + 3, recurrent_gain=2.0, bias_scale=0.1)
  outs = []
  for with_comm in (False, True):
    eng = built_lib.Engine(cfg, device=0)
    eng.set_params(params)
    eng.set_compute_mode("f16x3")
    eng.train_init()
    if with_comm:
      eng.comm_init(0, 1, built_lib.comm_unique_id())
    for step in range(2):
      eng.train_step(synth.make_feed(cfg, seed=synth.SEED_BASE + 200 + step))
    info = eng.comm_info()
    outs.append({n: eng.get_param(n) for n, _ in eng.param_specs()})
    if with_comm:
      _, nelem = eng.grad_buffer()
      print("in-library all-reduce:", info)
      assert info["world"] == 1 and info["rank"] == 0
      # 8 ConvLSTM buckets + the scene convs + two runs of small tensors per scale (the
      # embedding / hidden2grid kernels that follow each decoder cell in creation order)
      assert info["buckets"] == 8 + 1 + 4
      assert info["bytes"] == 4.0 * nelem
    else:
      assert info is None
    eng.close()
  for n in outs[0]:
    assert (outs[0][n] == outs[1][n]).all(), n


FAKE_RCCL = os.path.join(ROOT, "tests", "fake_rccl", "libfakerccl.so")
HOOKS_LIB = os.path.join(ROOT, "tests", "fake_rccl", "libmultiverse_hip_testhooks.so")


def _lib_worker(rank, world, outdir, extra_env=None):
  """One rank of the IN-LIBRARY data-parallel step: mv_allreduce_init + the bucketed
  side-stream all-reduce inside mv_train_forward_backward / mv_train_step, over the
  shared-memory RCCL stand-in (MV_RCCL_LIB), both ranks on GPU 0."""
  sys.path.insert(0, ROOT)
  os.environ["MV_RCCL_LIB"] = FAKE_RCCL
  os.environ["MV_LIB_PATH"] = HOOKS_LIB      # the -DMV_TEST_HOOKS copy: only it reads MV_RCCL_LIB
  os.environ.update(extra_env or {})
  os.environ["MV_FAKE_RCCL_LOG"] = os.path.join(outdir, "rccl_rank%d.log" % rank)
  os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
  import time
  from multiverse_amd import _lib, parallel, synth
  idfile = os.path.join(outdir, "unique_id.bin")
  if rank == 0:
    uid = _lib.comm_unique_id()
    with open(idfile + ".tmp", "wb") as f:
      f.write(bytes(uid))
    os.rename(idfile + ".tmp", idfile)
  else:
    t0 = time.time()
    while not os.path.exists(idfile):
      assert time.time() - t0 < 120, "rank 0 never published the unique id"
      time.sleep(0.05)
    uid = open(idfile, "rb").read()
  gcfg = _cfg(N_GLOBAL, synth, True)
  params = synth.make_params(gcfg, seed=synth.SEED_BASE + 3, recurrent_gain=2.0,
                             bias_scale=0.1)
  lo, hi = parallel.shard_range(N_GLOBAL, rank, world)
  eng = _lib.Engine(_cfg(hi - lo, synth, True), device=0)
  eng.set_params(params)
  eng.set_compute_mode("f16x3")
  eng.train_init(world=world)
  eng.comm_init(rank, world, bytes(uid))
  _, nelem = eng.grad_buffer()
  out, losses, infos = {}, [], []
  for step in range(STEPS):
    feed = synth.make_feed(gcfg, seed=synth.SEED_BASE + 200 + step)
    shard, _ = parallel.shard_feed(feed, rank, world, N_GLOBAL)
    if step == 0:     # the split form: the gradients in the buffer are the SUM over the ranks
      loss, wd, pgl = eng.train_forward_backward(shard)
      for n, _ in eng.param_specs():
        out["grad|" + n] = eng.get_grad(n)
      eng.train_apply(1.0 / world)
    else:             # the one-call form: reduce + 1/world + clip + optimizer inside
      loss, wd, pgl = eng.train_step(shard)
    losses.append([loss, wd] + list(pgl))
    infos.append(eng.comm_info())
  assert eng.global_step == STEPS
  for info in infos:
    assert info["world"] == world and info["rank"] == rank
    assert info["buckets"] == 8 + 1 + 4 and info["bytes"] == 4.0 * nelem, info
  out.update({n: eng.get_param(n) for n, _ in eng.param_specs()})
  out.update({"slot0|" + n: eng.get_opt_slot(n, 0) for n, _ in eng.param_specs()})
  out["losses"] = np.asarray(losses)
  out["nelem"] = np.asarray([nelem])
  eng.close()
  np.savez(os.path.join(outdir, "lib_rank%d.npz" % rank), **out)


def _two_ranks_vs_one_process(built_lib, tmp_path, extra_env):
  """Runs the two ranks (tests/fake_rccl under `extra_env`) and one process on the global batch;
  returns the deviations instead of asserting them, so that the negative controls can demand
  that they are LARGE: {"calls": per-rank collective log, "ranks_equal": both ranks bit-identical,
  "grad": worst |reduced gradient - global-batch gradient| / max|g|, "loss": worst relative loss
  difference, "param": worst |2 ranks - 1 process| / max(1, max|p|), "param_upd": the same as a
  fraction of the parameter's update, "slot": optimizer slot 0}."""
  from multiverse_amd import synth
  world = 2
  mp.spawn(_lib_worker, args=(world, str(tmp_path), extra_env), nprocs=world, join=True)
  r = [np.load(os.path.join(str(tmp_path), "lib_rank%d.npz" % k)) for k in range(world)]
  res = {"nelem": int(r[0]["nelem"][0]), "calls": []}
  for k in range(world):
    lines = open(os.path.join(str(tmp_path), "rccl_rank%d.log" % k)).read().split("\n")
    res["calls"].append([(int(l.split()[6]), int(l.split()[8])) for l in lines if l.strip()])
  gcfg = _cfg(N_GLOBAL, synth, True)
  params = synth.make_params(gcfg, seed=synth.SEED_BASE + 3, recurrent_gain=2.0,
                             bias_scale=0.1)
  res["ranks_equal"] = all((r[0][n] == r[1][n]).all() and
                           (r[0]["grad|" + n] == r[1]["grad|" + n]).all() for n in params)
  eng = built_lib.Engine(gcfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode("f16x3")
  eng.train_init()
  res["grad"] = res["loss"] = res["param"] = res["param_upd"] = res["slot"] = 0.0
  for step in range(STEPS):
    feed = synth.make_feed(gcfg, seed=synth.SEED_BASE + 200 + step)
    if step == 0:
      loss, wd, pgl = eng.train_forward_backward(feed)
      for n, _ in eng.param_specs():
        g = eng.get_grad(n)                  # gradient of the GLOBAL batch mean
        gs = r[0]["grad|" + n] / world       # sum over ranks of the shard means / world
        scale = max(float(np.abs(g).max()), 1e-30)
        res["grad"] = max(res["grad"], float(np.abs(gs - g).max()) / scale)
      eng.train_apply(1.0)
    else:
      loss, wd, pgl = eng.train_step(feed)
    ref = np.asarray([loss, wd] + list(pgl))
    mean = 0.5 * (r[0]["losses"][step] + r[1]["losses"][step])
    print("step %d: single-process %s | 2 ranks (mean) %s" % (step, ref, mean))
    res["loss"] = max(res["loss"], float(np.abs((mean - ref) / np.maximum(np.abs(ref), 1e-7)).max()))
  for n, _ in eng.param_specs():
    ref = eng.get_param(n)
    upd = max(float(np.abs(ref - params[n]).max()), 1e-12)
    d = float(np.abs(r[0][n] - ref).max())
    res["param"] = max(res["param"], d / max(1.0, float(np.abs(ref).max())))
    res["param_upd"] = max(res["param_upd"], d / upd)
    s_ref = eng.get_opt_slot(n, 0)
    res["slot"] = max(res["slot"], float(np.abs(r[0]["slot0|" + n] - s_ref).max()) /
                      max(float(np.abs(s_ref).max()), 1e-30))
  eng.close()
  print("2 ranks vs 1 process under %s: %s" % (extra_env, {k: v for k, v in res.items() if k != "calls"}))
  return res


@pytest.mark.parametrize("mode", ["sync", "async"])
def test_in_library_allreduce_two_ranks_one_gpu(built_lib, tmp_path, mode):
  """The library's own bucketed all-reduce (comm.h, engine_train.h comm_reduce_*) with TWO
  ranks: bucket order, event hand-offs between the main and the side stream, the 1 / world
  scale and clip-after-reduce -- against one process on the global batch.  RCCL refuses two
  ranks on one GPU, so the ranks load the shared-memory stand-in tests/fake_rccl through
  MV_RCCL_LIB (test infrastructure: same entry points, sum in rank order on the host); what
  is under test is everything on THIS side of ncclAllReduce.  mode "sync": the stand-in blocks
  the host inside the call; mode "async": it only enqueues work on the caller's stream and
  returns, as RCCL does (the exchange runs in a host function with an extra 20 ms delay) -- the
  mode in which a missing event wait is visible (next test)."""
  if not (os.path.exists(FAKE_RCCL) and os.path.exists(HOOKS_LIB)):
    pytest.skip("tests/fake_rccl/*.so not built (__graft_entry__.build())")
  env = ({"MV_FAKE_RCCL_ASYNC": "1", "MV_FAKE_RCCL_DELAY_MS": "20", "MV_FAKE_RCCL_POISON": "1"}
         if mode == "async" else {})
  res = _two_ranks_vs_one_process(built_lib, tmp_path, env)
  nelem, calls = res["nelem"], res["calls"]
  # what the stand-in saw: per step 13 collectives per rank, the same sizes in the same order
  # on both ranks, covering the gradient buffer exactly once; the last group inside
  # ncclGroupStart / End
  assert calls[0] == calls[1] and len(calls[0]) == STEPS * 13
  for st in range(STEPS):
    step_calls = calls[0][st * 13:(st + 1) * 13]
    assert sum(c for c, _ in step_calls) == nelem
    assert [g for _, g in step_calls[:8]] == [0] * 8            # ConvLSTM buckets, one by one
    assert all(g == 1 for _, g in step_calls[8:])               # the rest as one group
    assert min(c for c, _ in step_calls[:8]) > 2_000_000        # a kernel + its biases each
  # both ranks hold the same model afterwards, bit for bit ...
  assert res["ranks_equal"]
  # ... and it is the model one process makes of the global batch
  print("reduced gradients vs the global batch: worst %.2e of max|g|" % res["grad"])
  assert res["grad"] <= 2e-4 and res["loss"] <= 2e-6
  assert res["param"] <= 1e-6 and res["param_upd"] <= 2e-3 and res["slot"] <= 1e-4
  print("parameters after %d in-library data-parallel steps (%s stand-in): max |2 ranks - 1 "
        "process| = %.2e" % (STEPS, mode, res["param"]))


@pytest.mark.parametrize("fault,what", [("1", "ready"), ("2", "done")])
def test_two_rank_test_fails_when_an_event_wait_is_dropped(built_lib, tmp_path, fault, what):
  """Negative control of the test above: MV_COMM_FAULT makes the engine skip ONE of its two
  event waits around the collectives (1: the side stream no longer waits for the main stream's
  `ready` -- the collective reads gradients the wgrad kernels have not finished; 2: the main
  stream no longer waits for `done` -- clip + optimizer run on unreduced gradients).  Over the
  asynchronous stand-in either must be VISIBLE in the quantities the test above asserts;
  otherwise that test could not catch an ordering bug in comm_reduce_*."""
  if not (os.path.exists(FAKE_RCCL) and os.path.exists(HOOKS_LIB)):
    pytest.skip("tests/fake_rccl/*.so not built (__graft_entry__.build())")
  # (the stand-in poisons the receive buffer with NaNs until its copy-back lands: a main
  # stream that does not wait for `done` reads them whatever the timing)
  env = {"MV_FAKE_RCCL_ASYNC": "1", "MV_FAKE_RCCL_DELAY_MS": "20", "MV_FAKE_RCCL_POISON": "1",
         "MV_COMM_FAULT": fault}
  # `done`: deterministic (the poison).  `ready`: the collective must START before the wgrad
  # kernels it should have waited for have finished -- a race the side stream wins by a
  # millisecond in every run so far; it gets three attempts so that one lost race on a loaded
  # box cannot turn the negative control into a red suite
  broken = False
  for attempt in range(1 if what == "done" else 3):
    sub = tmp_path / ("attempt%d" % attempt)
    sub.mkdir()
    res = _two_ranks_vs_one_process(built_lib, sub, env)
    broken = (not res["ranks_equal"]) or not (res["grad"] <= 2e-4) or \
        not (res["param_upd"] <= 2e-3)
    print("dropped `%s` wait (attempt %d): ranks_equal %s grad %.2e param_upd %.2e -> %s" % (
        what, attempt, res["ranks_equal"], res["grad"], res["param_upd"],
        "DETECTED" if broken else "NOT detected"))
    if broken:
      break
  assert broken, "the two-rank test cannot see a missing `%s` wait" % what


def test_bench_spawns_its_own_ranks(built_lib):
  """`python bench.py --gpus 2` with NO launcher: bench.py starts the two ranks itself
  (torch.distributed.run on 127.0.0.1) and relays ONE JSON line whose `value` is the
  whole-job rate and whose training sub-object ran the data-parallel step on both ranks.
  MV_BENCH_BACKEND=gloo lets the ranks share the single GPU of the test box (RCCL refuses
  two ranks on one device; on a 2-GPU box the same command runs over RCCL and reports
  rccl_ranks = 2)."""
  import json
  import subprocess
  env = dict(os.environ, MV_BENCH_BACKEND="gloo")
  env.pop("WORLD_SIZE", None)
  env.pop("RANK", None)
  out = subprocess.check_output(
      [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3",
       "--warmup", "1", "--no-fp32-ref", "--only-sub", "train_n32"],
      env=env, timeout=900)
  lines = [l for l in out.decode().splitlines() if l.strip()]
  assert len(lines) == 1, lines
  d = json.loads(lines[0])
  assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1
  assert d["config"]["global_batch"] == 2 * d["config"]["batch_per_gpu"] == 128
  assert d["value"] > 0 and abs(d["value"] - 128 * 3 / (d["ms_per_step"] * 3e-3)) < 1e-2 * d["value"]
  t = d["train_n32"]
  assert "error" not in t, t
  assert t["global_batch"] == 64 and t["value"] > 0
  assert t["rccl_ranks"] == 0          # gloo here; 2 over RCCL on a 2-GPU box
  assert "host_path" not in d and "cpu_baseline" not in d     # rank 0 at N = 1 only
