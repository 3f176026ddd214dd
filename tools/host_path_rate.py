#!/usr/bin/env python
"""PCIe-inclusive rates of the greedy forward at the benchmark size (N=64, both
scales): the same batch through (a) resident inputs (bench.py's `value`), (b)
mv_forward_greedy with host buffers (dense maps over PCIe), (c) the compact upload
(labels + (x, y) + uint8 masks, maps expanded in HBM), and (d)/(e) the same two
including the host-side feed construction from a Dataset batch
(`Model.get_feed_dict`).  Prints one JSON line.  Run on the GPU box."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multiverse_amd import _lib, pred_models, pred_utils, synth  # noqa: E402


def rate(fn, n, reps=5, warm=2):
  for _ in range(warm):
    fn()
  t0 = time.perf_counter()
  for _ in range(reps):
    fn()
  return n * reps / (time.perf_counter() - t0)


def main():
  N = 64
  cfg = synth.default_config(batch_size=N, use_grids=(1, 1))
  params = synth.make_params(cfg, seed=synth.SEED_BASE + 2)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 2)
  eng = _lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode("f16x3")
  out = {"batch": N, "unit": "trajectories/sec"}

  eng.upload(feed)

  def resident():
    eng.run_resident(False)
    eng.synchronize()
  out["resident_inputs"] = round(rate(resident, N), 1)
  out["host_buffers_dense"] = round(rate(lambda: eng.forward_greedy(feed), N), 1)
  out["host_buffers_compact"] = round(rate(lambda: eng.forward_greedy_compact(feed), N), 1)
  eng.close()

  data = synth.make_npz_data(cfg, N, seed=11, float32_traj=True)
  ds = pred_utils.dataset_from_npz_dict(data, "test", cfg)
  batch = next(ds.get_batches(N, full=True, shuffle=False))
  model = pred_models.get_model(cfg, 0)
  model.load_params(params)
  tester = pred_models.Tester(model, cfg)
  for compact in (False, True):
    cfg.compact_inputs = compact
    t_feed = time.perf_counter()
    for _ in range(3):
      model.get_feed_dict(batch[1], is_train=False)
    t_feed = (time.perf_counter() - t_feed) / 3
    key = "tester_step_compact" if compact else "tester_step_dense"
    out[key] = round(rate(lambda: tester.step(None, batch), N, reps=3, warm=1), 1)
    out[key + "_feed_ms"] = round(1e3 * t_feed, 2)
  model.close()
  h2d_dense = sum(a.nbytes for a in feed["grid_obs_regress"]) + feed["scene_feat"].nbytes
  out["h2d_MB_dense"] = round(h2d_dense / 1e6, 2)
  out["h2d_MB_compact"] = round((feed["obs_xy"].nbytes + feed["scene_feat_u8"].nbytes) / 1e6, 3)
  print(json.dumps(out))


if __name__ == "__main__":
  main()
