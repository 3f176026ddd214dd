# coding=utf-8
"""The oracle is pinned by the reference's own code: `golden_shim_*.npz` hold
outputs of /root/reference/code/pred_models.py executed UNMODIFIED on the
eager TF-1 shim (oracle/tf1_shim).  Here (CPU) the oracle restatement must
reproduce them; on the GPU box the HIP engine is held to the same files
(tests/test_gpu_reference_pin.py)."""
import copy

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


def test_oracle_single_decoder_with_beam_search_equals_reference():
  """--use_single_decoder + --use_beam_search (code/pred_models.py:274, 287-296): the cell
  outputs are traced back along every beam and the offsets decoded per beam,
  grid_pred_reg_decoded [N * beam, T, H, W, 2]."""
  from beam_compare import compare_beams
  g, cfg, params, feed = sg.single_decoder_beam_case()
  cls, reg, beam = oracle.forward(params, cfg, feed)
  N, B = cfg.batch_size, cfg.beam_size
  assert reg[1].shape == g["reg_1"].shape == (N * B, cfg.pred_len, 9, 16, 2)
  assert (beam[1] == g["beam_ids"]).all()
  assert np.abs(beam[0] - g["beam_logits"]).max() <= 2e-5
  assert np.abs(cls[1] - g["cls_1"]).max() <= 2e-5
  assert np.abs(reg[1] - g["reg_1"]).max() <= 2e-5


# ---- N4 (SimAug): the reference's own SimAug/code/pred_models.py, run on the shim with
# injected random draws and frozen in golden_simaug.npz (tests/simaug_cases.py)

def _simaug_gold():
  import simaug_cases as sc
  return sc, np.load(sc.GOLD)


def test_simaug_graph_forward_equals_reference():
  """SimAug's greedy decoder attends over the hidden state alone (its gnn_edge concatenates
  the scene features only under tile_to_beam, SimAug/code/pred_models.py:1219-1227)."""
  sc, g = _simaug_gold()
  cfg = sc.config(False)
  from multiverse_amd import simaug
  params, feed = sc.base_inputs(cfg)
  feed = simaug.per_step_scene_feed(cfg, feed)     # norm_input: masks mapped to [-1, 1]
  cls, reg, _ = oracle.forward(params, cfg, feed)
  assert np.abs(cls[1] - g["forward|cls_1"]).max() <= 2e-5
  assert np.abs(reg[1] - g["forward|reg_1"]).max() <= 2e-5
  base = copy.copy(cfg)
  base.simaug_graph = False                        # code/pred_models.py's attention differs
  bcls, _, _ = oracle.forward(params, base, feed)
  assert np.abs(bcls[1] - g["forward|cls_1"]).max() > 1e-3


@pytest.mark.parametrize("name", ["fgsm", "pgd3", "mix_clean", "mix_adv",
                                  "clean_start_norm_feat", "maybe_clean"])
def test_simaug_white_box_attack_equals_reference(name):
  from multiverse_amd import simaug
  from oracle import simaug_oracle
  sc, g = _simaug_gold()
  over, seed = sc.WHITE_BOX[name]
  cfg = sc.config(True, adv_train=True, **over)
  params, feed = sc.base_inputs(cfg)
  pf = simaug.per_step_scene_feed(cfg, feed)
  adv, target = simaug_oracle.white_box_attack(params, cfg, pf, simaug.Draws(seed), simaug,
                                               norm_feat=cfg.norm_feat)
  assert (target == g["wb|%s|target" % name]).all()
  gold = g["wb|%s|adv" % name]
  assert (np.abs(sc.samples(adv) - gold[3:]) < 1e-6).mean() > 0.9999
  assert abs(float(np.abs(adv).astype(np.float64).sum()) - gold[1]) <= 1e-6 * gold[1]
  loss, _, _, grads = oracle.loss_and_grads(params, sc.train_config(cfg),
                                            dict(pf, scene_feat=adv))
  assert abs(loss - float(g["wb|%s|loss" % name][0])) <= 2e-6 * abs(loss)
  if name == "fgsm":
    for n, gr in grads.items():
      e_s, e_a = sg.digest_err(gr, _restride(g["wb|fgsm|grad|%s" % n], gr, sc))
      assert e_s < 2e-5 and e_a < 1e-4, (n, e_s, e_a)


def _restride(gold, like, sc):
  """digest written with simaug_cases.STRIDE -> the (sum, sum|.|, max, samples) layout
  shim_golden.digest_err expects for an array `like` (which samples with sg.STRIDE)."""
  a = np.asarray(like, dtype=np.float32).reshape(-1)
  mine = a[::sc.STRIDE]
  assert mine.shape[0] == gold.shape[0] - 3
  # compare on the finer grid directly: return a digest whose samples sit at sg.STRIDE
  # positions that are multiples of both strides where possible, else fall back to stats only
  out = np.concatenate([gold[:3], a[::sg.STRIDE].astype(np.float64)])
  assert np.abs(mine - gold[3:]).max() <= 2e-5 * max(gold[2], 1e-30)
  return out


@pytest.mark.parametrize("name", ["exp1", "exp2", "exp4_maxw", "exp3_dw", "exp3_random_advloss"])
def test_simaug_multiview_equals_reference(name):
  from multiverse_amd import simaug
  from oracle import simaug_oracle
  sc, g = _simaug_gold()
  over, seed = sc.MULTIVIEW[name]
  cfg = sc.config(True, multiview_train=True, **over)
  params, feed = sc.base_inputs(cfg)
  f0, pf, extra_scene = sc.multiview_feed(cfg, feed)
  o = simaug_oracle.multiview_augmentation(params, cfg, pf, f0["grid_pred_labels_extra"][1],
                                           simaug.Draws(seed), simaug, extra_scene)
  gold = g["mv|%s|mixed" % name]
  assert abs(np.float32(o["weight"]) - np.float32(g["mv|%s|weight" % name][0])) < 1e-7
  assert (np.abs(sc.samples(o["mixed"]) - gold[3:]) < 1e-6).mean() > 0.9999
  tfeed = dict(pf, scene_feat=o["mixed"])
  if cfg.multiview_exp == 3:
    assert (o["select"] == g["mv|%s|select" % name]).all()
    assert np.allclose(o["focal"], g["mv|%s|focal" % name], rtol=1e-6)
    tfeed = sc.label_mixup_feed(pf, f0, o["mixed"], o["weight"], o["select"], o["focal"],
                                cfg.double_weighting)
  loss, _, _, grads = oracle.loss_and_grads(params, sc.train_config(cfg), tfeed)
  assert abs(loss - float(g["mv|%s|loss" % name][0])) <= 2e-6 * abs(loss)
  if name == "exp3_dw":
    for n, gr in grads.items():
      _restride(g["mv|exp3_dw|grad|%s" % n], gr, sc)


def test_simaug_beam_search_equals_reference():
  """SimAug's beam decoder tiles the scene features into its attention (tile_to_beam), its
  greedy regression decoder has no attention: with simaug_graph the oracle equals the frozen
  run of SimAug's own file -- ids bit-exact."""
  from multiverse_amd import simaug
  sc, g = _simaug_gold()
  cfg = sc.beam_config()
  params, feed = sc.base_inputs(cfg)
  cls, reg, beam = oracle.forward(params, cfg, simaug.per_step_scene_feed(cfg, feed))
  assert (np.asarray(beam[1]) == g["beam|ids"]).all()
  assert np.abs(cls[1] - g["beam|best"]).max() <= 2e-5
  assert np.abs(reg[1] - g["beam|reg"]).max() <= 2e-5
  assert np.abs(np.asarray(beam[0]) - g["beam|logits"]).max() <= 2e-5
  assert np.abs(np.asarray(beam[2]) - g["beam|logprobs"]).max() <= 1e-4
