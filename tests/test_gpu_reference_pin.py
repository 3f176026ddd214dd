# coding=utf-8
"""GPU: the HIP engine against the outputs of the reference's own unmodified
pred_models.py (frozen by oracle/tf1_shim/make_shim_golden.py; the reference
checkout is not needed here).  Bars from BASELINE.json north_star: argmax /
beam ids bit-exact, logits and regression within 1e-4."""
import numpy as np
import pytest

from oracle import multiverse_oracle as oracle

import shim_golden as sg
from beam_compare import compare_beams

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.mark.parametrize("name", ["golden_shim_greedy_cfg1.npz",
                                  "golden_shim_greedy_both.npz",
                                  "golden_shim_greedy_nognn.npz"])
def test_greedy_against_reference_run(built_lib, name):
  g, cfg, params, feed = sg.forward_case(name)
  eng = built_lib.Engine(cfg, device=0)
  ref = sg.var_table(g)
  ref.pop("global_step")
  assert dict(eng.param_specs()) == ref          # names / shapes the reference creates
  eng.set_params(params)
  cls, reg = eng.forward_greedy(feed)
  eng.close()
  N = cfg.batch_size
  for s in range(2):
    if not cfg.use_grids[s]:
      continue
    assert (cls[s].reshape(N, 12, -1).argmax(-1) ==
            g["cls_%d" % s].reshape(N, 12, -1).argmax(-1)).all()
    assert np.abs(cls[s] - g["cls_%d" % s]).max() < TOL
    assert np.abs(reg[s] - g["reg_%d" % s]).max() < TOL


@pytest.mark.parametrize("name,scale", [("golden_shim_beam_s1.npz", 1),
                                        ("golden_shim_beam20_s0.npz", 0),
                                        ("golden_shim_beam_plain_s1.npz", 1)])
def test_beam_against_reference_run(built_lib, name, scale):
  g, cfg, params, feed = sg.forward_case(name)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  arrs, s = eng.forward_beam(feed)
  eng.close()
  assert s == scale
  # the oracle reproduces the reference run (bitwise on the host that wrote the
  # fixture, to conv-summation-order noise on this box's CPU); its trace supplies
  # the per-step candidate scores the tie-aware comparison needs
  trace = {}
  _, oreg, obeam = oracle.forward(params, cfg, feed, trace=trace)
  assert (obeam[1] == g["beam_ids"]).all()
  assert np.abs(obeam[0] - g["beam_logits"]).max() <= 2e-5
  compare_beams(arrs, g["reg_%d" % scale], g["beam_logits"], g["beam_ids"],
                g["beam_logprobs"], np.stack(trace["beam_step_topvals"], axis=-1),
                trace["beam_trace"])


@pytest.mark.parametrize("name", ["golden_shim_train_both.npz",
                                  "golden_shim_train_s1_3steps.npz"])
def test_train_steps_against_reference_trainer(built_lib, name):
  """Trainer.step of the reference (tf.gradients, clip, Adadelta, LR staircase,
  global_step) vs mv_train_step."""
  g, cfg, params, feeds = sg.train_case(name)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.train_init()
  for step, feed in enumerate(feeds):
    loss, wd, pgl = eng.train_step(feed)
    ref = g["loss_%d" % step]
    print(name, "step", step, "loss", loss, "reference", ref[0])
    assert np.allclose([loss, wd] + pgl, ref, rtol=1e-4, atol=1e-5), (loss, ref)
    worst = 0.0
    for n, _ in eng.param_specs():
      e_s, e_a = sg.digest_err(eng.get_grad(n), g["grad_%d|%s" % (step, n)])
      worst = max(worst, e_s)
      assert e_s < 2e-3 and e_a < 2e-3, (n, e_s, e_a)
    print("  worst sampled gradient error (of max|g|): %.2e" % worst)
  for n, _ in eng.param_specs():
    e_s, e_a = sg.digest_err(eng.get_param(n), g["param|%s" % n])
    assert e_s < 1e-4 and e_a < 1e-5, (n, e_s, e_a)
  assert eng.global_step == int(g["global_step"][0])
  eng.close()
