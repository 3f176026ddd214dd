# coding=utf-8
"""Diagnostic (GPU box): one row of the configs[3] launch (N = 128, beam 20) at recurrent gain 3
against the batch-1 oracle -- where do the logits rows differ, and what do the oracle's candidate
scores look like there?  usage: python tests/diag/beam_row_diag.py <row> [gain]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from multiverse_amd import _lib, synth
from oracle import multiverse_oracle as oracle

n = int(sys.argv[1]); gain = float(sys.argv[2]) if len(sys.argv) > 2 else 3.0
N, B = 128, 20
cfg = synth.default_config(batch_size=N, use_grids=(1, 0), beam_size=B)
params = synth.make_params(cfg, seed=synth.SEED_BASE + 2, recurrent_gain=gain, bias_scale=0.1)
feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 2)
out = {}
for mode in ("f16x3", "f32"):
  eng = _lib.Engine(cfg, device=0)
  eng.set_params(params); eng.set_compute_mode(mode)
  arrs, s = eng.forward_beam(feed)
  eng.close()
  out[mode] = {k: v[n] for k, v in arrs.items()}
cfg1 = synth.default_config(batch_size=1, use_grids=(1, 0), beam_size=B)
f1 = dict(feed)
f1["obs_scene"] = feed["obs_scene"][n:n + 1]
f1["grid_obs_labels"] = [a[n:n + 1] for a in feed["grid_obs_labels"]]
f1["grid_obs_regress"] = [a[n:n + 1] for a in feed["grid_obs_regress"]]
torch.set_num_threads(16)
for dt in (torch.float32, torch.float64):
  trace = {}
  _, oreg, ob = oracle.forward(params, cfg1, f1, trace=trace, dtype=dt)
  topv = np.stack(trace["beam_step_topvals"], axis=-1)[0]      # [B, T]
  print("==== oracle", dt)
  print("oracle final logprobs", np.round(ob[2][0], 5))
  for mode in ("f16x3", "f32"):
    a = out[mode]
    same_ids = [(a["ids"][b] == ob[1][0][b]).all() for b in range(B)]
    print(mode, "ids equal per beam:", "".join("1" if x else "0" for x in same_ids),
          " max|dlogprob| %.3g" % np.abs(a["logprobs"] - ob[2][0]).max())
    for b in range(B):
      for t in range(cfg.pred_len):
        err = np.abs(a["logits"][b, t] - ob[0][0][b, t]).max()
        if err > 1e-4:
          j = trace["beam_trace"][0][b, t]
          print("  %s b=%d t=%d err %.3g  oracle trace idx %d  gaps at t: %s  at t-1: %s" % (
              mode, b, t, err, j,
              np.round(np.abs(np.diff(topv[:, t]))[max(0, j - 2):j + 2] * 1e5, 2),
              np.round(np.abs(np.diff(topv[:, t - 1]))[max(0, j - 2):j + 2] * 1e5, 2) if t else None))
  if dt == torch.float32:
    print("selected-candidate scores t=0:", np.round(topv[:, 0], 5))
    print("selected-candidate scores t=1:", np.round(topv[:, 1], 5))
    print("ids t0", ob[1][0][:, 0], "\nids t1", ob[1][0][:, 1])
e, f = out["f16x3"], out["f32"]
print("engine f16x3 vs engine f32: ids equal", (e["ids"] == f["ids"]).all(),
      "max|dlogits| %.3g" % np.abs(e["logits"] - f["logits"]).max())
