# coding=utf-8
"""The oracle is pinned by the reference's own code: `golden_shim_*.npz` hold
outputs of /root/reference/code/pred_models.py executed UNMODIFIED on the
eager TF-1 shim (oracle/tf1_shim).  Here (CPU) the oracle restatement must
reproduce them; on the GPU box the HIP engine is held to the same files
(tests/test_gpu_reference_pin.py)."""
import numpy as np
import pytest

from multiverse_amd import synth
from oracle import multiverse_oracle as oracle
from oracle.tf1_shim import run_reference as rr

import shim_golden as sg


@pytest.mark.parametrize("name", sorted(sg.FORWARD_CASES))
def test_oracle_forward_equals_reference_run(name):
  g, cfg, params, feed = sg.forward_case(name)
  cls, reg, beam = oracle.forward(params, cfg, feed)
  for s in range(2):
    if not cfg.use_grids[s]:
      continue
    assert np.abs(cls[s] - g["cls_%d" % s]).max() <= 2e-5
    assert np.abs(reg[s] - g["reg_%d" % s]).max() <= 2e-5
    assert (cls[s].reshape(cfg.batch_size, 12, -1).argmax(-1) ==
            g["cls_%d" % s].reshape(cfg.batch_size, 12, -1).argmax(-1)).all()
  if beam is not None:
    assert (beam[1] == g["beam_ids"]).all()
    assert np.abs(beam[0] - g["beam_logits"]).max() <= 2e-5
    assert np.abs(beam[2] - g["beam_logprobs"]).max() <= 2e-5


def test_variable_names_are_the_ones_the_reference_creates():
  """tf.get_variable names / shapes requested by the reference code (under the
  shim's TF-1 scoping rules) == the table the engine and synth use."""
  for name, kw in sg.FORWARD_CASES.items():
    ref = sg.var_table(sg.load(name))
    assert ref.pop("global_step") == ()
    ours = synth.param_shapes(synth.default_config(**kw))
    if sum(kw["use_grids"]) == 1:   # scene_conv2 exists even when scale 1 is unused
      pass
    assert ours == ref, (set(ours) ^ set(ref))


@pytest.mark.parametrize("name", ["golden_shim_train_both.npz",
                                  "golden_shim_train_s1_3steps.npz"])
def test_oracle_training_equals_reference_trainer(name):
  g, cfg, params, feeds = sg.train_case(name)
  p, st = dict(params), oracle.adadelta_init(params)
  for step, feed in enumerate(feeds):
    loss, wd, pgl, p, st, grads = oracle.train_step(p, st, step, cfg, feed)
    ref = g["loss_%d" % step]
    assert np.allclose([loss, wd] + pgl, ref, rtol=2e-6, atol=1e-6), (loss, ref)
    for n, gr in grads.items():
      e_s, e_a = sg.digest_err(gr, g["grad_%d|%s" % (step, n)])
      assert e_s < 2e-5 and e_a < 1e-4, (n, e_s, e_a)
  for n in p:
    e_s, e_a = sg.digest_err(p[n], g["param|%s" % n])
    assert e_s < 1e-6 and e_a < 1e-6, (n, e_s, e_a)
  assert int(g["global_step"][0]) == len(feeds)


@pytest.mark.skipif(not rr.available(), reason="needs the /root/reference checkout")
def test_live_reference_run_reproduces_the_fixture():
  """Re-run the reference's code on the shim and compare bitwise with the
  committed fixture (guards against a stale golden)."""
  g, cfg, params, feed = sg.forward_case("golden_shim_beam_s1.npz")
  cls, reg, beam = rr.forward(cfg, params, feed)
  assert (np.asarray(cls[1]) == g["cls_1"]).all()
  assert (np.asarray(reg[1]) == g["reg_1"]).all()
  assert (np.asarray(beam[1]) == g["beam_ids"]).all()
  assert (np.asarray(beam[0]) == g["beam_logits"]).all()


@pytest.mark.parametrize("name", sorted(sg.VARIANT_CASES))
def test_oracle_variants_equal_reference_trainer(name):
  """Soft grid labels (with TensorFlow's registered softmax-xent gradient), fg-masked
  regression, teacher forcing, dense class feedback, input dropout and the three other
  optimizers: the oracle against the reference's own Trainer.step on the shim."""
  g, cfg, params, feeds = sg.variant_case(name)
  p, st = dict(params), oracle.optimizer_init(cfg, params)
  for step, feed in enumerate(feeds):
    loss, wd, pgl, p, st, grads = oracle.train_step(p, st, step, cfg, feed)
    ref = g["loss_%d" % step]
    # a second Adam step starts from parameters that already differ by ~lr where |g| is
    # at the fp32 noise floor (update ~ lr g / (|g| + 3e-7)): only its first step is tight
    loose = name == "adam" and step > 0
    assert np.allclose([loss, wd] + pgl, ref, rtol=1e-3 if loose else 2e-6, atol=1e-6), (loss, ref)
    for n, gr in grads.items():
      e_s, e_a = sg.digest_err(gr, g["grad_%d|%s" % (step, n)])
      assert e_s < (5e-2 if loose else 2e-5) and e_a < (5e-2 if loose else 1e-4), (n, e_s, e_a)
  if name != "adam":
    for n in p:
      e_s, e_a = sg.digest_err(p[n], g["param|%s" % n])
      assert e_s < 2e-6 and e_a < 2e-6, (n, e_s, e_a)
  if name == "adam":
    assert np.allclose(st[""], g["opt_scalars"], rtol=1e-6)
  assert int(g["global_step"][0]) == len(feeds)


def test_oracle_test_time_teacher_forcing_equals_reference():
  """--use_teacher_forcing without training: the class decoder eats its raw logits."""
  g = sg.load("golden_shim_variant_teacher_test.npz")
  cfg = synth.default_config(batch_size=2, use_grids=(0, 1), use_teacher_forcing=True)
  params = synth.make_params(cfg, seed=sg.VARIANT_SEED + 1, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=sg.VARIANT_SEED + 1)
  cls, reg, _ = oracle.forward(params, cfg, feed)
  assert np.abs(cls[1] - g["cls_1"]).max() <= 2e-5
  assert np.abs(reg[1] - g["reg_1"]).max() <= 2e-5


def test_oracle_single_decoder_equals_reference():
  """--use_single_decoder on both scales (shared decode_reg kernel, regression encoder
  without a gradient): greedy forward and two Trainer.steps of the reference on the shim."""
  g, (cfg, params, feed), (tcfg, tparams, feeds) = sg.single_decoder_case()
  names = sg.var_table(g)
  names.pop("global_step", None)
  assert sorted(names) == sorted(params)
  assert names["person_pred/decode_reg/out_dec_grid/W"] == (3, 3, 256, 2)
  assert not any("decoder_grid_reg" in n for n in names)
  cls, reg, _ = oracle.forward(params, cfg, feed)
  for s in range(2):
    assert np.abs(cls[s] - g["cls_%d" % s]).max() <= 2e-5
    assert np.abs(reg[s] - g["reg_%d" % s]).max() <= 2e-5
  p, st = dict(tparams), oracle.optimizer_init(tcfg, tparams)
  no_grad = sorted(str(n) for n in g["no_grad"])
  assert no_grad == sorted(n for n in tparams if "encoder_grid_reg" in n)
  for step, feed in enumerate(feeds):
    loss, wd, pgl, p, st, grads = oracle.train_step(p, st, step, tcfg, feed)
    assert np.allclose([loss, wd] + pgl, g["loss_%d" % step], rtol=2e-6, atol=1e-6)
    assert sorted(n for n in tparams if grads.get(n) is None) == no_grad
    for n, gr in grads.items():
      if gr is None:
        continue
      e_s, e_a = sg.digest_err(gr, g["grad_%d|%s" % (step, n)])
      assert e_s < 2e-5 and e_a < 1e-4, (n, e_s, e_a)
  for n in p:
    e_s, e_a = sg.digest_err(p[n], g["param|%s" % n])
    assert e_s < 2e-6 and e_a < 2e-6, (n, e_s, e_a)
  for n in no_grad:                       # never touched by apply_gradients
    assert (p[n] == tparams[n]).all()
