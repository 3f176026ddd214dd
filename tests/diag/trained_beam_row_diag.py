# coding=utf-8
"""Diagnostic (GPU box): one row of the trained-weights beam-20 test (tests/test_gpu_trained_parity.py)
against the batch-1 oracle in fp32 AND fp64 -- which beams differ, at which step their id
sequences leave the oracle's, and what the oracle's candidate scores look like at the keep / drop
cut of that step.  usage: python tests/diag/trained_beam_row_diag.py <row> [<row> ...]"""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from multiverse_amd import _lib, synth
from oracle import multiverse_oracle as oracle

rows = [int(a) for a in sys.argv[1:]]
N_TRAIN, N_BEAM, B, STEPS = 16, 32, 20, 300
cfg = synth.default_config(batch_size=N_TRAIN, use_grids=(1, 1), is_train=True, optimizer="adam",
                           init_lr=2e-3)
params = synth.make_params(cfg, seed=synth.SEED_BASE + 61)
feeds = [synth.make_feed(cfg, seed=synth.SEED_BASE + 600 + i) for i in range(12)]
eng = _lib.Engine(cfg, device=0)
eng.set_params(params); eng.set_compute_mode("f16x3"); eng.train_init()
for it in range(STEPS):
  eng.train_step(feeds[it % len(feeds)])
trained = {n: eng.get_param(n) for n, _ in eng.param_specs()}
eng.close()
cfgN = synth.default_config(batch_size=N_BEAM, use_grids=(1, 0), beam_size=B)
want = synth.param_shapes(cfgN)
P = {n: np.ascontiguousarray(trained[n], dtype=np.float32) for n in want}
feed = synth.make_feed(cfgN, seed=synth.SEED_BASE + 901)
out = {}
for mode in ("f16x3", "f32"):
  e2 = _lib.Engine(cfgN, device=0)
  e2.set_params(P); e2.set_compute_mode(mode)
  arrs, s = e2.forward_beam(feed)
  e2.close()
  out[mode] = arrs
print("engine f16x3 vs engine f32: ids equal", bool((out["f16x3"]["ids"] == out["f32"]["ids"]).all()))
cfg1 = synth.default_config(batch_size=1, use_grids=(1, 0), beam_size=B)
torch.set_num_threads(16)
for n in rows:
  f1 = dict(feed)
  f1["obs_scene"] = feed["obs_scene"][n:n + 1]
  f1["grid_obs_labels"] = [a[n:n + 1] for a in feed["grid_obs_labels"]]
  f1["grid_obs_regress"] = [a[n:n + 1] for a in feed["grid_obs_regress"]]
  for dt in (torch.float32, torch.float64):
    trace = {}
    _, oreg, ob = oracle.forward(P, cfg1, f1, trace=trace, dtype=dt)
    oids = ob[1][0]
    cut = np.stack(trace["beam_step_cut_gap"], axis=-1)[0]
    topv = np.stack(trace["beam_step_topvals"], axis=-1)[0]
    print("==== row %d oracle %s: cut gaps per step %s" % (n, dt, np.array2string(cut, precision=6)))
    print("   min gap between selected neighbours per step", np.array2string(np.abs(np.diff(topv, axis=0)).min(axis=0), precision=6))
    for mode in ("f16x3", "f32"):
      a = out[mode]
      oset = {tuple(oids[b]) for b in range(B)}
      miss = [b for b in range(B) if tuple(a["ids"][n, b]) not in oset]
      print("  %s: beams whose id sequence is not among the oracle's 20: %s" % (mode, miss))
      for b in miss:
        seq = a["ids"][n, b]
        # longest common prefix with any oracle beam
        best = max(range(B), key=lambda bb: int(np.argmax(np.append(oids[bb] != seq, True))))
        lcp = int(np.argmax(np.append(oids[best] != seq, True)))
        print("    beam %d: logprob %.6f; leaves oracle beam %d at step %d (engine id %d, oracle id %d); "
              "oracle final logprobs around the cut: %s" % (
                  b, a["logprobs"][n, b], best, lcp, seq[lcp] if lcp < len(seq) else -1,
                  oids[best][lcp] if lcp < len(seq) else -1, np.round(ob[2][0][-3:], 6)))
        if lcp < len(seq):
          # the oracle's candidates at that step: per parent, log-prob of the two ids and the
          # parent's top candidates (the diversity penalty is log(gamma) x rank WITHIN a parent)
          lg = np.asarray(trace["beam_step_logits"][lcp], dtype=np.float64).reshape(B, -1)
          prev = np.asarray(trace["beam_step_prev_lp"][lcp], dtype=np.float64).reshape(B)
          m = lg.max(axis=1, keepdims=True)
          lp = prev[:, None] + (lg - m - np.log(np.exp(lg - m).sum(axis=1, keepdims=True)))
          for par in range(B):
            order = np.argsort(-lp[par], kind="stable")
            ra = int(np.where(order == seq[lcp])[0][0]); rb = int(np.where(order == oids[best][lcp])[0][0])
            if min(ra, rb) < 3:
              print("      parent %2d: rank / lp of engine id %d: %d / %.6f   of oracle id %d: %d / %.6f   "
                    "top-3 ids %s lp %s" % (par, seq[lcp], ra, lp[par, seq[lcp]], oids[best][lcp], rb,
                                            lp[par, oids[best][lcp]], order[:3],
                                            np.array2string(lp[par, order[:3]], precision=6)))
