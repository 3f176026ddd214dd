# coding=utf-8
"""Diagnostic (GPU box): gradient cosines of a bf16-mode training step against the fp32 oracle
for --activation_func relu / lrelu / tanh models (the x k-steps of unbounded-activation models
run on an fp16 plane under a per-tensor exponent in bf16 mode).
usage: python tests/diag/bf16_relu_grad_cosine.py"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from multiverse_amd import _lib, synth
from oracle import multiverse_oracle as oracle

for act in ("tanh", "relu", "lrelu"):
  cfg = synth.default_config(batch_size=2, use_grids=(1, 1), is_train=True, activation_func=act)
  params = synth.make_params(cfg, seed=synth.SEED_BASE + 3, recurrent_gain=2.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 73)
  oloss, owd, opgl, og = oracle.loss_and_grads(params, cfg, feed)
  for mode in ("bf16", "f16x3"):
    eng = _lib.Engine(cfg, device=0)
    eng.set_params(params); eng.set_compute_mode(mode); eng.train_init()
    loss, wd, pgl = eng.train_forward_backward(feed)
    worst, wn = 2.0, ""
    for n, _ in eng.param_specs():
      if og[n] is None: continue
      a = eng.get_grad(n).reshape(-1).astype(np.float64); b = og[n].reshape(-1).astype(np.float64)
      if np.linalg.norm(b) == 0: continue
      c = float(a @ b / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))
      if c < worst: worst, wn = c, n
    eng.close()
    print("%-5s %-5s loss %.6f (oracle %.6f, rel %.2e)  worst gradient cosine %.5f  (%s)"
          % (act, mode, loss, oloss, abs(loss - oloss) / max(1, abs(oloss)), worst, wn))
