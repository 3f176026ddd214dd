# coding=utf-8
"""TEST INFRASTRUCTURE: runs bench.py's real control flow -- argument handling, spawning its
own ranks, process group, per-rank feeds, barriers, max-over-ranks timing, the ONE JSON line
from rank 0 -- on a box WITHOUT a GPU, by standing a pure-Python engine in for
multiverse_amd._lib.Engine and no-ops in for the torch.cuda calls bench.py makes.  Nothing
here measures anything; tests/test_bench_multirank.py reads the line and the per-rank records
this harness leaves in $MV_HARNESS_DIR.  (bench.py re-launches sys.argv[0], i.e. THIS file,
for the N ranks of `--gpus N`.)"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


class FakeEngine(object):
  """The calls bench.measure() makes, with a rank-dependent step time."""

  def __init__(self, cfg, device=0):
    self.cfg, self.device = cfg, device
    self.rank = int(os.environ.get("RANK", "0"))
    self.steps = 0
    self.train = False

  # -- set-up ----------------------------------------------------------------------------
  def set_params(self, params): self.nparams = len(params)
  def upload(self, feed): self.feed_rows = int(feed["obs_scene"].shape[0])
  def upload_targets(self, feed): pass
  def set_graph_mode(self, on): pass
  def set_compute_mode(self, mode): self.mode = mode
  def train_init(self, world=1): self.train, self.world = True, world
  def comm_info(self): return None
  def synchronize(self): pass
  def set_profiling(self, on): pass
  def reset_kernel_stats(self): pass
  def close(self): pass

  # -- one step: rank r takes (r + 1) x 2 ms, so the SLOWEST rank sets ms_per_step --------
  def _work(self):
    time.sleep(0.002 * (self.rank + 1))
    self.steps += 1

  def run_resident(self, beam): self._work()
  def train_forward_backward(self, feed): self._work()
  def train_step(self, feed): self._work()
  def train_apply(self, scale): self.scale = scale

  def kernel_stats(self):
    one = {"launches": 20, "total_ms": 10.0, "flops": 1e12, "bytes": 1e9, "flops_dense": 1.1e12,
           "flops_mfma": 1.7e12}
    names = ["convlstm_step"] + (["convlstm_dgrad", "convlstm_wgrad"] if self.train else [])
    return {n: dict(one) for n in names}


def main():
  import torch
  from multiverse_amd import _lib, parallel, synth
  out_dir = os.environ["MV_HARNESS_DIR"]
  rank = int(os.environ.get("RANK", "-1"))
  torch.cuda.is_available = lambda: True
  torch.cuda.device_count = lambda: 8
  torch.cuda.set_device = lambda d: None
  torch.cuda.synchronize = lambda *a, **k: None
  _lib.Engine = FakeEngine
  record = {"rank": rank, "feed_seeds": [], "allreduce_calls": 0}
  real_feed = synth.make_feed

  def feed_spy(cfg, seed=None, **kw):
    record["feed_seeds"].append(seed)
    return real_feed(cfg, seed=seed, **kw)
  synth.make_feed = feed_spy
  # full-size random weights are 85 MB per rank and nothing here reads them
  synth.make_params = lambda cfg, **kw: {}

  def fake_allreduce(engine, device_index=None):
    import torch.distributed as dist
    t = torch.ones(4)
    dist.all_reduce(t)                      # the collective of the step, over gloo
    assert float(t[0]) == dist.get_world_size()
    record["allreduce_calls"] += 1
  parallel.allreduce_engine_grads = fake_allreduce
  if os.environ.get("MV_HARNESS_STUCK_COMM") == "1":
    # the in-library RCCL bootstrap of the training sub-workload never returns
    def stuck_comm(engine):
      parallel.call_with_deadline(lambda: time.sleep(120), 1.0,
                                  "mv_allreduce_init (ncclCommInitRank) [harness: stuck]")
      return True
    parallel.init_engine_comm = stuck_comm
  import bench
  try:
    bench.main()
  finally:
    if rank >= 0:
      with open(os.path.join(out_dir, "rank%d.json" % rank), "w") as f:
        json.dump(record, f)


if __name__ == "__main__":
  main()
