#!/usr/bin/env python
# coding=utf-8
"""Regenerates the golden fixtures in this directory from the CPU oracle.

    python tests/golden/make_golden.py

The reference ships no golden vectors and TensorFlow 1.x
cannot run here, so these are outputs of `oracle/multiverse_oracle.py` on
seeded synthetic inputs -- frozen so that (a) an accidental change of the
oracle shows up as a diff against the fixture, and (b) the GPU tests can check
the HIP path against committed numbers on a box where only the repo exists.
`golden_shim_*.npz` (written by oracle/tf1_shim/make_shim_golden.py) come from
the reference's UNMODIFIED code/pred_models.py executed on the TF-1 API
emulation and need /root/reference.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from multiverse_amd import synth  # noqa: E402
from oracle import multiverse_oracle as oracle  # noqa: E402
sys.path.insert(0, HERE)
import cases  # noqa: E402


def kernel_cases():
  out = {}
  # one ConvLSTM step per input width of the four cells (Cx = 64, 2, 32)
  for Cx, (H, W) in cases.CELL_CASES:
    x, c, h, k, b = cases.cell_case(Cx, H, W)
    co, ho = oracle.convlstm_step_np(x, c, h, k, b)
    out.update({"cell%d_c_out" % Cx: co, "cell%d_h_out" % Cx: ho})
  h, sm = cases.gnn_case()
  out["gnn_out"] = oracle.gnn_np(h, sm)
  return out


def forward_case(name, cfg, seed, gain, bias):
  params = synth.make_params(cfg, seed=seed, recurrent_gain=gain, bias_scale=bias)
  feed = synth.make_feed(cfg, seed=seed)
  trace = {}
  cls, reg, beam = oracle.forward(params, cfg, feed, trace=trace)
  out = {"seed": np.array([seed]), "gain": np.array([gain]), "bias": np.array([bias])}
  for s in range(len(cfg.scene_grids)):
    if not cfg.use_grids[s]:
      continue
    out["cls_%d" % s] = cls[s]
    out["reg_%d" % s] = reg[s]
    if "greedy_ids_%d" % s in trace:
      out["ids_%d" % s] = np.stack(trace["greedy_ids_%d" % s], axis=1)
  if beam is not None:
    out["beam_logits"] = beam[0]
    out["beam_ids"] = beam[1]
    out["beam_logprobs"] = beam[2]
    out["beam_topvals"] = np.stack(trace["beam_step_topvals"], axis=-1)
    out["beam_trace"] = trace["beam_trace"]
  np.savez_compressed(os.path.join(HERE, name), **out)
  print("wrote", name, {k: v.shape for k, v in out.items()})


def main():
  np.savez_compressed(os.path.join(HERE, "golden_kernels.npz"), **kernel_cases())
  print("wrote golden_kernels.npz")
  # BASELINE config 1: single scale 18x32, N=4
  forward_case("golden_greedy_cfg1.npz",
               synth.default_config(batch_size=4, use_grids=(1, 0)),
               synth.SEED_BASE + 0, 3.0, 0.1)
  # both scales, reference initialisers, N=2
  forward_case("golden_greedy_both.npz",
               synth.default_config(batch_size=2, use_grids=(1, 1)),
               synth.SEED_BASE + 1, 1.0, 0.0)
  # beam search, scale 1, N=2, B=5
  forward_case("golden_beam_s1.npz",
               synth.default_config(batch_size=2, use_grids=(0, 1), beam_size=5),
               synth.SEED_BASE + 5, 3.0, 0.1)


if __name__ == "__main__":
  main()
