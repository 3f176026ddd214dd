# coding=utf-8
"""Diagnostic (GPU box): a long training run on synthetic data in one compute mode -- every loss
finite, the mean loss falling, no fp16-range flag from the device-side re-packs, weights finite
at the end.  usage: python tests/diag/train_soak.py <mode: f16x3|bf16|f32> [steps] [batch]"""
import sys, os, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from multiverse_amd import _lib, synth

mode = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
N = int(sys.argv[3]) if len(sys.argv) > 3 else 16
cfg = synth.default_config(batch_size=N, use_grids=(1, 1), is_train=True, optimizer="adam",
                           init_lr=1e-3)
params = synth.make_params(cfg, seed=synth.SEED_BASE + 61)
feeds = [synth.make_feed(cfg, seed=synth.SEED_BASE + 700 + i) for i in range(24)]
eng = _lib.Engine(cfg, device=0)
eng.set_params(params)
eng.set_compute_mode(mode)
eng.train_init()
t0 = time.time()
losses = []
for it in range(steps):
  loss, wd, _ = eng.train_step(feeds[it % len(feeds)])
  losses.append(loss)
  if not np.isfinite(loss):
    print("step %d: loss %r" % (it, loss)); sys.exit(1)
  if (it + 1) % (steps // 10) == 0:
    print("%s step %5d: mean loss of the last %d steps %.4f" % (mode, it + 1, steps // 10,
                                                                float(np.mean(losses[-(steps // 10):]))))
dt = time.time() - t0
trained = {n: eng.get_param(n) for n, _ in eng.param_specs()}
eng.close()
finite = all(np.isfinite(v).all() for v in trained.values())
moved = max(float(np.abs(trained[n] - params[n]).max()) for n in params)
print("%s: %d steps at batch %d in %.1f s (%.1f ms per step incl. the host loop); loss %.4f -> %.4f; "
      "all weights finite: %s; largest parameter move %.3g; max |w| %.3g"
      % (mode, steps, N, dt, 1e3 * dt / steps, losses[0], float(np.mean(losses[-24:])), finite, moved,
         max(float(np.abs(v).max()) for v in trained.values())))
assert finite and np.mean(losses[-24:]) < 0.5 * np.mean(losses[:24])
