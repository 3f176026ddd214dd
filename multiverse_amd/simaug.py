# coding=utf-8
"""SimAug training extras on the engine: the host-side mirror of

  white_box_attack          SimAug/code/pred_models.py:60-172   targeted FGSM / PGD on the
                            scene-semantics features, optional mixup with the clean or a
                            second adversarial view
  multiview_augmentation    SimAug/code/pred_models.py:346-543  one FGSM step per extra
                            camera view (targets = that view's own future grid labels),
                            selection of two views by attack loss (exp 1 / 4) or at random
                            (exp 2), Beta-weighted mixup; exp 3: the hardest view's
                            adversarial features mixed with the CLEAN features of a selected
                            extra view, the labels mixed up the same way
                            (mv_set_label_mixup) and the class loss focal-weighted

The network, its backward pass and the element-wise attack / mixup updates run in the HIP
engine (mv_attack_begin / mv_train_forward_backward / mv_attack_step / mv_scene_mix /
mv_get_sample_losses); what is left here is control flow and the random draws.  TensorFlow
draws those from unseeded ops, so `Draws` makes them explicit and injectable: parity is
"same draws -> same adversarial features" (tests/test_gpu_simaug.py against the torch-autograd
restatement in oracle/simaug_oracle.py).

Conventions: the attacked tensor is `obs_scene` AFTER the embedding lookup, one map per
(n, t): feeds carry scene_feat [N*T_o, SH, SW, SC] with obs_scene = arange (norm_input
features in [-1, 1] when config.norm_input).  Only one grid scale may be enabled
(`assert sum(config.use_grids) == 1`, :292, :305).
"""

from __future__ import annotations

import copy

import numpy as np

from multiverse_amd import _lib


class Draws(object):
  """The random draws of the two augmentations, in the reference's call order."""

  def __init__(self, seed=0):
    self.rng = np.random.default_rng(seed)

  def label_offset(self, shape, max_class):          # tf.random_uniform(minval=1, maxval=K, int32)
    return self.rng.integers(1, max_class, size=shape).astype("int32")

  def noise(self, shape, eps):                       # tf.random_uniform(-eps, eps)
    return self.rng.uniform(-eps, eps, size=shape).astype("float32")

  def scalar(self):                                  # tf.random_uniform(shape=[])
    return float(self.rng.uniform())

  def beta(self, alpha):                             # tf.distributions.Beta(a, a).sample()
    return float(self.rng.beta(alpha, alpha))

  def index(self, n, lo, hi):                        # tf.random.uniform([N], lo, hi, int32)
    return self.rng.integers(lo, hi, size=n).astype("int32")


def per_step_scene_feed(cfg, feed):
  """Feed whose scene table holds one map per (n, t) (the tensor the attacks perturb):
  scene_feat[n * T_o + t] = feed scene_feat[obs_scene[n, t]]; `config.norm_input` maps the
  0/1 masks to [-1, 1] (SimAug/code/pred_models.py:283-285)."""
  N, T = cfg.batch_size, cfg.obs_len
  out = dict(feed)
  sf = np.asarray(feed["scene_feat"], dtype="float32")[np.asarray(feed["obs_scene"]).reshape(-1)]
  if getattr(cfg, "norm_input", False):
    sf = sf * 2.0 - 1.0
  out["scene_feat"] = np.ascontiguousarray(sf)
  out["obs_scene"] = np.arange(N * T, dtype="int32").reshape(N, T)
  return out


def attack_config(cfg):
  """The loss of one_step_attack (:94-110) is the class cross entropy alone: no
  regression term, no weight decay, hard labels (sign() ignores its scale)."""
  a = copy.copy(cfg)
  a.grid_reg_loss_weight = 0.0
  a.wd = 0.0
  a.use_soft_grid_class = False
  a.mask_grid_regression = False
  return a


def random_targets(labels, max_class, draws):
  """create_random_target (:67-72): (label + U{1..K-1}) mod K, never the true class."""
  lab = np.asarray(labels, dtype="int32")
  return np.mod(lab + draws.label_offset(lab.shape, max_class), max_class).astype("int32")


def start_adv(feature, cfg, draws):
  """get_start_adv (:75-89 / :350-364)."""
  init = draws.noise(feature.shape, cfg.adv_epsilon)
  if cfg.adv_start_from_clean_prob >= 1.0:
    return feature
  if cfg.adv_start_from_clean_prob > 0:
    init = init * np.float32(draws.scalar() > cfg.adv_start_from_clean_prob)
  return (feature + init).astype("float32")


def _softmax_last(x):
  m = x.max(axis=-1, keepdims=True)
  e = np.exp(x - m)
  return (e / e.sum(axis=-1, keepdims=True)).astype("float32")


def _scale_of(cfg):
  if sum(bool(u) for u in cfg.use_grids) != 1:
    raise _lib.MvError("only one scale for adv / multiview train "
                       "(SimAug/code/pred_models.py:292, 305)")
  return [i for i, u in enumerate(cfg.use_grids) if u][0]


def _attack_feed(feed, s, target):
  f = dict(feed)
  f["grid_pred_labels"] = list(feed["grid_pred_labels"])
  f["grid_pred_labels"][s] = np.ascontiguousarray(target, dtype="int32")
  return f


def white_box_attack(engine, cfg, feed, draws, norm_feat=False):
  """-> (adversarial scene features [N*T_o, SH, SW, SC], target_label [N, T_p]).
  `engine`: a training engine (train_init done) of batch cfg.batch_size; `feed`: a
  per-step scene feed (per_step_scene_feed) with the TRUE future labels.  On return the
  engine's resident scene features are the adversarial ones and its training
  configuration is `cfg` again: `engine.upload_targets(feed); engine.train_step(None)`
  trains on them (the graph's tf.stop_gradient, :302, is implicit)."""
  s = _scale_of(cfg)
  h, w = cfg.scene_grids[s]
  target = random_targets(feed["grid_pred_labels"][s], h * w, draws)       # :132-134
  clean = np.asarray(feed["scene_feat"], dtype="float32")
  engine.train_init(attack_config(cfg))
  engine.upload(feed)
  engine.upload_targets(_attack_feed(feed, s, target))
  engine.attack_begin()                  # bounds come from the CLEAN features (:137-138)

  def attack(start, normalise=norm_feat):
    x0 = _softmax_last(start) if normalise else start                      # :143-144
    engine.set_scene_feat(x0)
    if cfg.adv_use_fgsm:                                                   # :146-147
      engine.train_forward_backward(None)
      engine.attack_step(cfg.adv_epsilon, cfg.adv_epsilon)                 # :118-119
    else:                                                                  # PGD (:121-122)
      for _ in range(int(cfg.adv_num_iter)):
        engine.train_forward_backward(None)
        engine.attack_step(cfg.adv_epsilon, cfg.adv_step_size)

  attack(start_adv(clean, cfg, draws))
  if getattr(cfg, "use_mixup", False):                                     # :159-171
    weight = draws.beta(cfg.mixup_alpha)
    if getattr(cfg, "mixup_mix_adv", False):
      assert cfg.adv_use_fgsm and cfg.adv_start_from_clean_prob < 1.0
      adv1 = engine.get_scene_feat()
      # the second view goes into one_step_attack WITHOUT the softmax (:162-163)
      attack(start_adv(clean, cfg, draws), normalise=False)
      engine.scene_mix(adv1, 1.0 - weight)      # adv2 * w + adv1 * (1 - w)
    else:
      engine.scene_mix(None, weight)            # clean * w + adv * (1 - w)
  adv = engine.get_scene_feat()
  engine.attack_end()
  engine.train_init(cfg)
  return adv, target


def multiview_augmentation(engine_m, cfg, feed, extra_pred_labels, draws):
  """multiview_augmentation (:346-543), experiments 1, 2 and 4.
  engine_m: a training engine of batch N * M (M = cfg.multiview_max_num) holding the same
  weights; feed: per-step scene feed of the N samples; extra_pred_labels [N, M, T_p]: the
  future grid labels seen from the M other camera views (grid_pred_labels_T_extra).
  -> (mixed features [N*T_o, SH, SW, SC], beta weight, per-view attack losses [N, M])."""
  s = _scale_of(cfg)
  N, M, T = cfg.batch_size, int(cfg.multiview_max_num), cfg.obs_len
  if cfg.multiview_exp == 3:
    raise _lib.MvError("multiview_exp 3 mixes labels too: use multiview_augmentation_exp3")
  if cfg.multiview_exp not in (1, 2, 4):
    raise _lib.MvError("Please set experiment number")                      # :523-525
  mcfg = copy.copy(cfg)
  mcfg.batch_size = N * M
  clean = np.asarray(feed["scene_feat"], dtype="float32")
  SH, SW, SC = clean.shape[1:]
  # tile every per-sample input M times (:418-448): row n * M + m
  tiled = {"pred_length": feed.get("pred_length", cfg.pred_len)}
  tiled["scene_feat"] = np.repeat(clean.reshape(N, T, SH, SW, SC), M, axis=0).reshape(-1, SH, SW, SC)
  tiled["obs_scene"] = np.arange(N * M * T, dtype="int32").reshape(N * M, T)
  for key in ("grid_obs_labels", "grid_obs_regress", "grid_pred_regress"):
    tiled[key] = [None if a is None else np.repeat(np.asarray(a), M, axis=0) for a in feed[key]]
  tiled["grid_pred_labels"] = [None] * len(cfg.scene_grids)
  tiled["grid_pred_labels"][s] = np.asarray(extra_pred_labels, dtype="int32").reshape(N * M, -1)
  engine_m.train_init(attack_config(mcfg))
  start = start_adv(tiled["scene_feat"], cfg, draws)       # one_step_attack starts from noise (:374)
  engine_m.upload(dict(tiled, scene_feat=start))
  engine_m.upload_targets(tiled)
  engine_m.attack_begin()              # NB: bounds here are start +- eps (:400-403), and the
  engine_m.train_forward_backward(None)                    # snapshot IS the noisy start
  adv_loss = engine_m.sample_losses(s).reshape(N, M)       # :404
  engine_m.attack_step(cfg.adv_epsilon, cfg.adv_epsilon)   # FGSM (:398)
  adv_out = engine_m.get_scene_feat().reshape(N, M, T, SH, SW, SC)
  engine_m.attack_end()
  engine_m.train_init(mcfg)
  # tf.nn.top_k(adv_loss, k=M, sorted): descending, ties -> lower index
  order = np.argsort(-adv_loss, axis=1, kind="stable")
  rows = np.arange(N)
  if cfg.multiview_exp == 1:                                # the two hardest views
    i1, i2 = order[:, 0], order[:, 1]
  elif cfg.multiview_exp == 4:                              # the two easiest
    i1, i2 = order[:, M - 1], order[:, M - 2]
  else:                                                     # exp 2: two distinct random views
    i1 = draws.index(N, 0, M)
    i2 = np.mod(i1 + draws.index(N, 1, M), M)
  feat1, feat2 = adv_out[rows, i1], adv_out[rows, i2]
  weight = draws.beta(cfg.mixup_alpha)                      # :528-529
  if getattr(cfg, "multiview_max_weight_for_first", False):
    weight = max(weight, 1.0 - weight)
  w32 = np.float32(weight)          # the Beta sample is a float32 tensor in the reference
  mixed = (feat1 * w32 + feat2 * (np.float32(1.0) - w32)).astype("float32")
  return mixed.reshape(N * T, SH, SW, SC), weight, adv_loss


def _tile_feed(cfg, feed, M, scene):
  """every per-sample input M times (:418-448), row n * M + m; `scene` [N*M*T_o, SH, SW, SC]"""
  N, T = cfg.batch_size, cfg.obs_len
  tiled = {"pred_length": feed.get("pred_length", cfg.pred_len), "scene_feat": scene}
  tiled["obs_scene"] = np.arange(N * M * T, dtype="int32").reshape(N * M, T)
  for key in ("grid_obs_labels", "grid_obs_regress", "grid_pred_regress"):
    tiled[key] = [None if a is None else np.repeat(np.asarray(a), M, axis=0) for a in feed[key]]
  return tiled


def multiview_augmentation_exp3(engine_m, cfg, feed, extra_pred_labels, extra_scene, draws):
  """Experiment 3 (:486-517): one FGSM step per extra view as in the other experiments; the
  view with the highest attack loss gives adv_feat1 and the focal weight
  (1 - exp(-loss))^fl_gamma; adv_feat2 = the CLEAN features `extra_scene`
  [N, M, T_o, SH, SW, SC] of the selected view (the hardest, or a random one with
  cfg.multiview_random); Beta-weighted mix.  With cfg.multiview_use_adv_for_loss the attack
  losses are re-evaluated at the adversarial features (a second attack pass whose features
  are discarded, :488-497).
  -> (mixed features [N*T_o, SH, SW, SC], beta weight, selected view index [N], focal
  weights [N], attack losses [N, M]).  The caller mixes the labels of the selected view with
  the same weight: engine.set_label_mixup(obs2, pred2, weight, focal if cfg.double_weighting)."""
  s = _scale_of(cfg)
  N, M, T = cfg.batch_size, int(cfg.multiview_max_num), cfg.obs_len
  mcfg = copy.copy(cfg)
  mcfg.batch_size = N * M
  clean = np.asarray(feed["scene_feat"], dtype="float32")
  SH, SW, SC = clean.shape[1:]
  tile_clean = np.repeat(clean.reshape(N, T, SH, SW, SC), M, axis=0).reshape(-1, SH, SW, SC)
  targets = [None] * len(cfg.scene_grids)
  targets[s] = np.asarray(extra_pred_labels, dtype="int32").reshape(N * M, -1)

  def one_step_attack(features):
    start = start_adv(features, cfg, draws)
    tiled = _tile_feed(cfg, feed, M, start)
    tiled["grid_pred_labels"] = targets
    engine_m.upload(tiled)
    engine_m.upload_targets(tiled)
    engine_m.attack_begin()                                  # bounds: start +- eps (:400-403)
    engine_m.train_forward_backward(None)
    loss = engine_m.sample_losses(s).reshape(N, M)
    engine_m.attack_step(cfg.adv_epsilon, cfg.adv_epsilon)
    out = engine_m.get_scene_feat()
    engine_m.attack_end()
    return out, loss

  engine_m.train_init(attack_config(mcfg))
  adv_flat, adv_loss = one_step_attack(tile_clean)
  if getattr(cfg, "multiview_use_adv_for_loss", False):      # :488-497
    _, adv_loss = one_step_attack(adv_flat)
  engine_m.train_init(mcfg)
  adv_out = adv_flat.reshape(N, M, T, SH, SW, SC)
  order = np.argsort(-adv_loss, axis=1, kind="stable")       # tf.nn.top_k(sorted)
  top = np.take_along_axis(adv_loss, order[:, :1], axis=1)[:, 0]
  focal = ((np.float32(1.0) - np.exp(-top.astype("float32"))) **
           np.float32(cfg.fl_gamma)).astype("float32")       # :502
  rows = np.arange(N)
  feat1 = adv_out[rows, order[:, 0]]
  select = order[:, 0].astype("int32")
  if getattr(cfg, "multiview_random", False):                # :511-513
    select = draws.index(N, 0, M)
  feat2 = np.asarray(extra_scene, dtype="float32")[rows, select]
  weight = draws.beta(cfg.mixup_alpha)
  if getattr(cfg, "multiview_max_weight_for_first", False):
    weight = max(weight, 1.0 - weight)
  w32 = np.float32(weight)          # the Beta sample is a float32 tensor in the reference
  mixed = (feat1 * w32 + feat2 * (np.float32(1.0) - w32)).astype("float32")
  return mixed.reshape(N * T, SH, SW, SC), weight, select, focal, adv_loss
