# coding=utf-8
"""GPU: SimAug training extras through the C ABI -- the gradient of the
attack loss w.r.t. the scene features, targeted FGSM / PGD (+ mixup) and the multi-view
selection -- against the torch-autograd restatement oracle/simaug_oracle.py under identical
injected random draws (SimAug/code/pred_models.py:60-172, 346-543)."""
import copy

import numpy as np
import pytest

from multiverse_amd import simaug, synth
from oracle import simaug_oracle

pytestmark = pytest.mark.gpu
_MIX_ORACLE = {}     # fp64 oracle results shared by the f32 and f16x3 runs


def _setup(built_lib, N=2, **over):
  cfg = synth.default_config(batch_size=N, use_grids=(0, 1), is_train=True)
  for k, v in dict(norm_input=True, adv_train=True, adv_epsilon=0.1, adv_step_size=0.03,
                   adv_num_iter=3, adv_start_from_clean_prob=0.0, adv_use_fgsm=True,
                   use_mixup=False, mixup_alpha=1.0, mixup_mix_adv=False,
                   multiview_max_num=3, multiview_exp=1,
                   multiview_max_weight_for_first=False).items():
    setattr(cfg, k, v)
  for k, v in over.items():
    setattr(cfg, k, v)
  params = synth.make_params(cfg, seed=synth.SEED_BASE + 50, recurrent_gain=2.0, bias_scale=0.1)
  feed = simaug.per_step_scene_feed(cfg, synth.make_feed(cfg, seed=synth.SEED_BASE + 51))
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.train_init()
  return cfg, params, feed, eng


def test_input_gradient_of_the_attack_loss(built_lib):
  """d sum(CE(target)) / d scene_feat: the backward pass taken down to the input."""
  cfg, params, feed, eng = _setup(built_lib)
  s = 1
  target = simaug.random_targets(feed["grid_pred_labels"][s], 9 * 16, simaug.Draws(3))
  eng.train_init(simaug.attack_config(cfg))
  eng.upload(feed)
  eng.upload_targets(simaug._attack_feed(feed, s, target))
  eng.attack_begin()
  eng.train_forward_backward(None)
  g = eng.get_scene_grad()
  losses = eng.sample_losses(s)
  eng.close()
  tcfg = copy.copy(cfg)
  ol, og = simaug_oracle.class_loss_and_input_grad(params, tcfg, feed, feed["scene_feat"], target)
  # the engine differentiates the MEAN over N*T_p rows (grid_loss_weight 1): rescale
  og = og / float(cfg.batch_size * cfg.pred_len)
  err = np.abs(g - og).max() / np.abs(og).max()
  big = np.abs(og) > 1e-4 * np.abs(og).max()
  print("d loss / d scene_feat: rel err %.2e, sign agreement on |g| > 1e-4 max: %.6f; "
        "per-sample losses %s / %s" % (err, (np.sign(g[big]) == np.sign(og[big])).mean(),
                                       losses, ol))
  assert err < 2e-3
  assert (np.sign(g[big]) == np.sign(og[big])).mean() == 1.0
  assert np.allclose(losses, ol, rtol=1e-4)


@pytest.mark.parametrize("name,over", [
    ("fgsm", dict()),
    ("pgd3", dict(adv_use_fgsm=False)),
    ("fgsm_mix_clean", dict(use_mixup=True)),
    ("fgsm_mix_adv", dict(use_mixup=True, mixup_mix_adv=True)),
    ("fgsm_clean_start_norm_feat", dict(adv_start_from_clean_prob=1.0)),
])
def test_white_box_attack_matches_the_oracle(built_lib, name, over):
  cfg, params, feed, eng = _setup(built_lib, **over)
  adv, target = simaug.white_box_attack(eng, cfg, feed, simaug.Draws(11))
  # the engine trains on the adversarial features it now holds
  eng.upload_targets(feed)
  loss = eng.train_step(None)[0]
  eng.close()
  oadv, otarget = simaug_oracle.white_box_attack(params, cfg, feed, simaug.Draws(11), simaug)
  assert (target == otarget).all() and (target != feed["grid_pred_labels"][1]).all()
  clean = feed["scene_feat"]
  same = np.abs(adv - oadv) < 1e-6
  print("%s: adversarial features equal on %.5f of the elements; max |adv - clean| %.4f; "
        "training loss on them %.5f" % (name, same.mean(), np.abs(adv - clean).max(), loss))
  # sign(g) is discontinuous at g = 0: a handful of elements at the fp32 noise floor of the
  # gradient may take the other branch (by 2 * step); everything else is identical
  assert same.mean() > 0.999
  assert np.abs(adv - clean).max() <= cfg.adv_epsilon + 1e-6 and np.abs(adv).max() <= 1.0 + 1e-6
  assert np.isfinite(loss)


@pytest.mark.parametrize("exp", [1, 2, 4])
def test_multiview_augmentation(built_lib, exp):
  cfg, params, feed, eng = _setup(built_lib, multiview_exp=exp)
  eng.close()
  N, M = cfg.batch_size, cfg.multiview_max_num
  mcfg = copy.copy(cfg)
  mcfg.batch_size = N * M
  engm = built_lib.Engine(mcfg, device=0)
  engm.set_params(params)
  engm.train_init()
  rng = np.random.default_rng(5)
  extra = rng.integers(0, 9 * 16, size=(N, M, cfg.pred_len)).astype("int32")
  mixed, weight, adv_loss = simaug.multiview_augmentation(engm, cfg, feed, extra, simaug.Draws(21))
  engm.close()
  # oracle: the per-view FGSM step and losses, then the same selection / mix on the host
  d = simaug.Draws(21)
  clean = feed["scene_feat"]
  T = cfg.obs_len
  tiled = dict(feed)
  tiled["scene_feat"] = np.repeat(clean.reshape((N, T) + clean.shape[1:]), M, axis=0).reshape(
      (-1,) + clean.shape[1:])
  tiled["obs_scene"] = np.arange(N * M * T, dtype="int32").reshape(N * M, T)
  for key in ("grid_obs_labels", "grid_obs_regress"):
    tiled[key] = [None if a is None else np.repeat(np.asarray(a), M, axis=0) for a in feed[key]]
  start = simaug.start_adv(tiled["scene_feat"], cfg, d)
  tcfg = copy.copy(mcfg)
  ol, og = simaug_oracle.class_loss_and_input_grad(params, tcfg, tiled, start,
                                                   extra.reshape(N * M, -1))
  assert np.allclose(adv_loss.reshape(-1), ol, rtol=1e-4)
  oadv = simaug_oracle.fgsm_step(start, og, start, cfg.adv_epsilon, cfg.adv_epsilon)
  oadv = oadv.reshape((N, M, T) + clean.shape[1:])
  order = np.argsort(-ol.reshape(N, M), axis=1, kind="stable")
  rows = np.arange(N)
  if exp == 1:
    i1, i2 = order[:, 0], order[:, 1]
  elif exp == 4:
    i1, i2 = order[:, M - 1], order[:, M - 2]
  else:
    i1 = d.index(N, 0, M)
    i2 = np.mod(i1 + d.index(N, 1, M), M)
  w = d.beta(cfg.mixup_alpha)
  omixed = (oadv[rows, i1] * np.float32(w) + oadv[rows, i2] * np.float32(1 - w)).reshape(mixed.shape)
  assert abs(w - weight) < 1e-12 and (i1 != i2).all()
  same = np.abs(mixed - omixed) < 1e-6
  print("multiview exp %d: mixed features equal on %.5f of the elements, weight %.4f, "
        "attack losses %s" % (exp, same.mean(), weight, np.round(adv_loss, 4)))
  assert same.mean() > 0.999


@pytest.mark.parametrize("mode", ["f32", "f16x3"])
def test_label_mixup_training_step(built_lib, mode):
  """mv_set_label_mixup (multi-view experiment 3, SimAug/code/pred_models.py:616-636,
  1371-1398): two-hot class-encoder inputs and first decoder input, mixed-up targets under
  softmax_cross_entropy_with_logits_v2, per-sample focal weights -- losses and every gradient
  against the oracle's autograd in fp64; weight 1 is the plain training step."""
  import torch
  from oracle import multiverse_oracle as oracle
  cfg = synth.default_config(batch_size=3, use_grids=(0, 1), is_train=True)
  params = synth.make_params(cfg, seed=synth.SEED_BASE + 60, recurrent_gain=2.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 61)
  rng = np.random.default_rng(9)
  obs2 = [None, rng.integers(0, 144, size=(3, cfg.obs_len)).astype("int32")]
  pred2 = [None, rng.integers(0, 144, size=(3, cfg.pred_len)).astype("int32")]
  obs2[1][0] = feed["grid_obs_labels"][1][0]          # a sample whose two views coincide
  sw = np.array([0.4, 1.7, 1.0], dtype="float32")
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode(mode)
  eng.train_init()
  plain = eng.train_forward_backward(feed)
  g_plain = {n: eng.get_grad(n) for n, _ in eng.param_specs()}
  eng.set_label_mixup(obs2, pred2, 1.0)
  same = eng.train_forward_backward(None)
  # (the mixed-target loss is -sum q log p over all K cells, the sparse one lse - logit[label]:
  # equal up to fp32 rounding, not bit for bit)
  assert np.allclose([same[0]] + same[2], [plain[0]] + plain[2], rtol=2e-6)
  for n in g_plain:
    assert np.abs(eng.get_grad(n) - g_plain[n]).max() <= 2e-5 * max(np.abs(g_plain[n]).max(), 1e-30), n
  for w, weights in ((0.65, None), (0.3, sw)):
    eng.set_label_mixup(obs2, pred2, w, weights)
    loss, wd, pgl = eng.train_forward_backward(None)
    f = dict(feed, mix_weight=w, mix_obs_labels=obs2, mix_pred_labels=pred2,
             mix_sample_weight=weights)
    key = (w, weights is not None)
    if key not in _MIX_ORACLE:
      _MIX_ORACLE[key] = oracle.loss_and_grads(params, cfg, f, dtype=torch.float64)
    ol, owd, opgl, og = _MIX_ORACLE[key]
    worst = 0.0
    for n, _ in eng.param_specs():
      g = eng.get_grad(n)
      err = float(np.abs(g - og[n]).max() / max(np.abs(og[n]).max(), 1e-30))
      worst = max(worst, err)
      assert err < 2e-3, (n, err)
    print("label mixup %s w=%.2f%s: loss %.6f oracle %.6f, worst gradient error %.2e of max"
          % (mode, w, " + sample weights" if weights is not None else "", loss, ol, worst))
    assert np.allclose([loss, wd] + pgl, [ol, owd] + opgl, rtol=1e-4, atol=1e-5)
  eng.clear_label_mixup()
  again = eng.train_forward_backward(None)
  assert again[0] == plain[0] and again[2] == plain[2]     # cleared: the plain step, bitwise
  eng.close()


def test_multiview_experiment_3(built_lib):
  """Experiment 3: hardest view's adversarial features x clean features of the selected
  view, focal weights from the top attack loss; against the oracle under the same draws."""
  cfg, params, feed, eng = _setup(built_lib, multiview_exp=3, fl_gamma=2.0,
                                  multiview_random=False, multiview_use_adv_for_loss=False,
                                  double_weighting=True)
  eng.close()
  N, M, T = cfg.batch_size, cfg.multiview_max_num, cfg.obs_len
  mcfg = copy.copy(cfg)
  mcfg.batch_size = N * M
  engm = built_lib.Engine(mcfg, device=0)
  engm.set_params(params)
  engm.train_init()
  rng = np.random.default_rng(5)
  extra = rng.integers(0, 9 * 16, size=(N, M, cfg.pred_len)).astype("int32")
  clean = feed["scene_feat"]
  extra_scene = (rng.uniform(size=(N, M, T) + clean.shape[1:]) > 0.5).astype("float32")
  mixed, weight, select, focal, adv_loss = simaug.multiview_augmentation_exp3(
      engm, cfg, feed, extra, extra_scene, simaug.Draws(31))
  engm.close()
  d = simaug.Draws(31)
  tile = np.repeat(clean.reshape((N, T) + clean.shape[1:]), M, axis=0).reshape(
      (-1,) + clean.shape[1:])
  start = simaug.start_adv(tile, cfg, d)
  tiled = simaug._tile_feed(cfg, feed, M, start)
  ol, og = simaug_oracle.class_loss_and_input_grad(params, copy.copy(mcfg), tiled, start,
                                                   extra.reshape(N * M, -1))
  assert np.allclose(adv_loss.reshape(-1), ol, rtol=1e-4)
  oadv = simaug_oracle.fgsm_step(start, og, start, cfg.adv_epsilon, cfg.adv_epsilon).reshape(
      (N, M, T) + clean.shape[1:])
  order = np.argsort(-ol.reshape(N, M), axis=1, kind="stable")
  rows = np.arange(N)
  assert (select == order[:, 0]).all()
  top = ol.reshape(N, M)[rows, order[:, 0]]
  assert np.allclose(focal, (1.0 - np.exp(-top)) ** cfg.fl_gamma, rtol=1e-5)
  w = d.beta(cfg.mixup_alpha)
  omixed = (oadv[rows, order[:, 0]] * np.float32(w) +
            extra_scene[rows, order[:, 0]] * np.float32(1 - w)).reshape(mixed.shape)
  same = np.abs(mixed - omixed) < 1e-6
  print("multiview exp 3: mixed features equal on %.5f of the elements, weight %.4f, focal %s"
        % (same.mean(), weight, np.round(focal, 4)))
  assert abs(w - weight) < 1e-12 and same.mean() > 0.999


# ---- against the frozen runs of the reference's own SimAug/code/pred_models.py
# (tests/golden/golden_simaug.npz, tests/simaug_cases.py): same injected draws, SimAug's graph

def _engine(built_lib, cfg, params, batch=None):
  c = copy.copy(cfg)
  if batch is not None:
    c.batch_size = batch
  eng = built_lib.Engine(c, device=0)
  eng.set_params(params)
  eng.train_init(sc_train_config(c))
  return eng


def sc_train_config(cfg):
  import simaug_cases as sc
  return sc.train_config(cfg)


def test_simaug_graph_forward_against_reference_run(built_lib):
  import simaug_cases as sc
  g = np.load(sc.GOLD)
  cfg = sc.config(False)
  params, feed = sc.base_inputs(cfg)
  feed = simaug.per_step_scene_feed(cfg, feed)
  for mode in ("f32", "f16x3"):
    eng = built_lib.Engine(cfg, device=0)
    eng.set_params(params)
    eng.set_compute_mode(mode)
    cls, reg = eng.forward_greedy(feed)
    eng.close()
    assert np.abs(cls[1] - g["forward|cls_1"]).max() <= 1e-4
    assert np.abs(reg[1] - g["forward|reg_1"]).max() <= 1e-4
    assert (cls[1].reshape(2, 12, -1).argmax(-1) ==
            g["forward|cls_1"].reshape(2, 12, -1).argmax(-1)).all()


@pytest.mark.parametrize("name", ["fgsm", "pgd3", "mix_adv", "clean_start_norm_feat"])
def test_white_box_attack_against_reference_run(built_lib, name):
  import simaug_cases as sc
  g = np.load(sc.GOLD)
  over, seed = sc.WHITE_BOX[name]
  cfg = sc.config(True, adv_train=True, **over)
  params, feed = sc.base_inputs(cfg)
  pf = simaug.per_step_scene_feed(cfg, feed)
  eng = _engine(built_lib, cfg, params)
  adv, target = simaug.white_box_attack(eng, sc.train_config(cfg), pf, simaug.Draws(seed),
                                        norm_feat=cfg.norm_feat)
  eng.upload_targets(pf)
  loss = eng.train_step(None)[0]
  eng.close()
  gold = g["wb|%s|adv" % name]
  same = np.abs(sc.samples(adv) - gold[3:]) < 1e-6
  print("%s: adversarial features equal to the reference's on %.5f of the sampled elements, "
        "training loss %.6f (reference %.6f)" % (name, same.mean(), loss, g["wb|%s|loss" % name][0]))
  assert (target == g["wb|%s|target" % name]).all()
  assert same.mean() > 0.999
  assert abs(loss - float(g["wb|%s|loss" % name][0])) <= 1e-4 * abs(loss)


@pytest.mark.parametrize("name", ["exp1", "exp2", "exp3_dw", "exp3_random_advloss"])
def test_multiview_against_reference_run(built_lib, name):
  import simaug_cases as sc
  g = np.load(sc.GOLD)
  over, seed = sc.MULTIVIEW[name]
  cfg = sc.config(True, multiview_train=True, **over)
  params, feed = sc.base_inputs(cfg)
  f0, pf, extra_scene = sc.multiview_feed(cfg, feed)
  tcfg = sc.train_config(cfg)
  engm = _engine(built_lib, cfg, params, batch=sc.N * sc.M)
  if cfg.multiview_exp == 3:
    mixed, weight, select, focal, _ = simaug.multiview_augmentation_exp3(
        engm, tcfg, pf, f0["grid_pred_labels_extra"][1], extra_scene, simaug.Draws(seed))
  else:
    mixed, weight, _ = simaug.multiview_augmentation(
        engm, tcfg, pf, f0["grid_pred_labels_extra"][1], simaug.Draws(seed))
  engm.close()
  gold = g["mv|%s|mixed" % name]
  same = np.abs(sc.samples(mixed) - gold[3:]) < 1e-6
  assert abs(np.float32(weight) - np.float32(g["mv|%s|weight" % name][0])) < 1e-7
  assert same.mean() > 0.999
  eng = _engine(built_lib, cfg, params)
  eng.upload(dict(pf, scene_feat=mixed))
  eng.upload_targets(pf)
  if cfg.multiview_exp == 3:
    assert (select == g["mv|%s|select" % name]).all()
    assert np.allclose(focal, g["mv|%s|focal" % name], rtol=1e-5)
    t = sc.label_mixup_feed(pf, f0, mixed, weight, select, focal, cfg.double_weighting)
    eng.set_label_mixup(t["mix_obs_labels"], t["mix_pred_labels"], weight,
                        t["mix_sample_weight"])
  loss = eng.train_forward_backward(None)[0]
  print("%s: mixed features equal to the reference's on %.5f of the sampled elements, training "
        "loss %.6f (reference %.6f)" % (name, same.mean(), loss, g["mv|%s|loss" % name][0]))
  assert abs(loss - float(g["mv|%s|loss" % name][0])) <= 1e-4 * abs(loss)
  if name == "exp3_dw":
    for n, _ in eng.param_specs():
      gd = g["mv|exp3_dw|grad|%s" % n]
      gr = sc.samples(eng.get_grad(n))
      assert np.abs(gr - gd[3:]).max() <= 2e-3 * max(gd[2], 1e-30), n
  eng.close()


def test_simaug_beam_search_against_reference_run(built_lib):
  """SimAug's beam decode (attention WITH the scene features, unlike its greedy decoder)."""
  import simaug_cases as sc
  g = np.load(sc.GOLD)
  cfg = sc.beam_config()
  params, feed = sc.base_inputs(cfg)
  feed = simaug.per_step_scene_feed(cfg, feed)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode("f16x3")
  arrs, s = eng.forward_beam(feed)
  eng.close()
  assert s == 1
  assert (np.asarray(arrs["ids"]) == g["beam|ids"]).all()
  assert np.abs(np.asarray(arrs["logits"]) - g["beam|logits"]).max() <= 1e-4
  assert np.abs(np.asarray(arrs["logprobs"]) - g["beam|logprobs"]).max() <= 1e-3
  assert np.abs(np.asarray(arrs["grid_reg"]) - g["beam|reg"]).max() <= 1e-4
