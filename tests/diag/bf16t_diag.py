# coding=utf-8
"""Diagnostic (GPU box): bf16 mode, row-triple tile (MV_BF16T=1) against the 32-cell body
(MV_BF16T=0), per decoder step.  Run each setting in its own process:
  MV_BF16T=1 python tests/diag/bf16t_diag.py out1.npz; MV_BF16T=0 ... out0.npz; then compare."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from multiverse_amd import _lib, synth
if len(sys.argv) == 3:
  a, b = np.load(sys.argv[1]), np.load(sys.argv[2])
  for k in a.files:
    d = np.abs(a[k].astype(np.float64) - b[k]).reshape(a[k].shape[0], a[k].shape[1], -1).max(-1)
    print(k, "range %.3g" % np.abs(b[k]).max(), "max|d| per step (row 0):", np.round(d[0], 5))
  sys.exit(0)
cfg = synth.default_config(batch_size=2, use_grids=(1, 1))
params = synth.make_params(cfg, recurrent_gain=1.0, bias_scale=0.0)
feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 22)
eng = _lib.Engine(cfg, device=0)
eng.set_params(params); eng.set_compute_mode("bf16")
cls, reg = eng.forward_greedy(feed)
eng.close()
np.savez(sys.argv[1], cls0=cls[0], cls1=cls[1], reg0=reg[0], reg1=reg[1])
