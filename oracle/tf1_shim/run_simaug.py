# coding=utf-8
"""Run the reference's UNMODIFIED SimAug/code/pred_models.py on the eager TF-1 shim  --
TEST INFRASTRUCTURE ONLY (needs /root/reference; the GPU box never runs this).

SimAug's Model builds its whole graph inside __init__ (placeholders, then white_box_attack /
multiview_augmentation, build_tower, build_loss), so the placeholder values are queued in
creation order (tf.preset_placeholders) and the unseeded random ops are replayed from an
injected source (tf.set_random_source: multiverse_amd.simaug.Draws).  make_simaug_golden.py
freezes these runs into tests/golden/golden_simaug_*.npz, which pin
multiverse_amd/simaug.py + the engine (tests/test_gpu_simaug.py) and
oracle/simaug_oracle.py (tests/test_reference_pin.py).
"""

from __future__ import annotations

import copy
import importlib
import importlib.util
import os
import sys

import numpy as np

SHIM_DIR = os.path.dirname(os.path.abspath(__file__))
SIMAUG_CODE = os.environ.get("MULTIVERSE_SIMAUG", "/root/reference/SimAug/code")

# switches SimAug's Model reads beyond the base config (SimAug/code/train.py); off by default
SIMAUG_DEFAULTS = dict(
    norm_input=False, norm_feat=False, adv_train=False, multiview_train=False,
    standard_aug=False, adv_epsilon=0.1, adv_step_size=0.03, adv_num_iter=3,
    adv_start_from_clean_prob=0.0, adv_use_fgsm=True, use_mixup=False, mixup_alpha=1.0,
    mixup_mix_adv=False, multiview_max_num=3, multiview_exp=1,
    multiview_max_weight_for_first=False, multiview_random=False,
    multiview_use_adv_for_loss=False, fl_gamma=2.0, double_weighting=False)


def available():
  return os.path.exists(os.path.join(SIMAUG_CODE, "pred_models.py"))


def import_simaug():
  if SHIM_DIR not in sys.path:
    sys.path.insert(0, SHIM_DIR)
  tf = importlib.import_module("tensorflow")
  assert "eager-shim" in tf.__version__, "a real tensorflow shadows the shim"
  if "simaug_pred_models" in sys.modules:
    return tf, sys.modules["simaug_pred_models"]
  spec = importlib.util.spec_from_file_location(
      "simaug_pred_models", os.path.join(SIMAUG_CODE, "pred_models.py"))
  mod = importlib.util.module_from_spec(spec)
  sys.modules["simaug_pred_models"] = mod
  spec.loader.exec_module(mod)
  return tf, mod


def placeholder_values(cfg, feed, is_train):
  """Values of Model.__init__'s placeholders in creation order
  (SimAug/code/pred_models.py:205-267)."""
  N, To = cfg.batch_size, cfg.obs_len
  Tp = int(feed.get("pred_length", cfg.pred_len))
  vals = [np.full([N], To, "int32"), np.full([N], Tp, "int32"), np.asarray(is_train),
          np.asarray(feed["obs_scene"], "int32"), np.asarray(feed["scene_feat"], "float32")]
  mv = bool(cfg.multiview_train)
  if mv:
    vals.append(np.asarray(feed["obs_scene_extra"], "int32"))
  for s, (h, w) in enumerate(cfg.scene_grids):
    used = cfg.use_grids[s]
    z = lambda *shape: np.zeros(shape, "float32")   # noqa: E731  unused scale: never read
    vals.append(np.asarray(feed["grid_obs_labels"][s], "int32") if used
                else np.zeros((N, To), "int32"))
    vals.append(np.asarray(feed["grid_obs_regress"][s], "float32") if used
                else z(N, To, h, w, 2))
    pl = feed.get("grid_pred_labels", [None] * 9)[s]
    vals.append(np.asarray(pl, "float32") if (used and pl is not None) else z(N, Tp))
    pr = feed.get("grid_pred_regress", [None] * 9)[s]
    vals.append(np.asarray(pr, "float32") if (used and pr is not None) else z(N, Tp, h, w, 2))
    if cfg.is_train and mv:
      M = cfg.multiview_max_num
      e = lambda key, shape, dt: (np.asarray(feed[key][s], dt) if used and feed.get(key)  # noqa: E731
                                  and feed[key][s] is not None else np.zeros(shape, dt))
      vals.append(e("grid_obs_labels_extra", (N, M, To), "int32"))
      vals.append(e("grid_pred_labels_extra", (N, M, Tp), "float32"))
      vals.append(e("grid_pred_regress_extra", (N, M, Tp, h, w, 2), "float32"))
      vals.append(e("grid_obs_regress_extra", (N, M, To, h, w, 2), "float32"))
  return vals


def build_model(cfg, params, feed, is_train, draws=None):
  """SimAug's Model(config, scope) executed eagerly on `feed`; returns (tf, module, model)."""
  tf, ref = import_simaug()
  cfg = copy.copy(cfg)
  for k, v in SIMAUG_DEFAULTS.items():
    if not hasattr(cfg, k):
      setattr(cfg, k, v)
  cfg.activation_func = tf.nn.tanh
  cfg.is_train = is_train
  p = dict(params)
  p["global_step"] = np.asarray(0, dtype="int32")
  tf.reset_default_graph(params=p, strict=True)
  tf.set_random_source(draws)
  tf.preset_placeholders(placeholder_values(cfg, feed, is_train))
  try:
    model = ref.get_model(cfg, 0)
  finally:
    tf.preset_placeholders(None)
  return tf, ref, model


def _np(tf, t):
  return np.asarray(tf._v(t).detach())


def _loss_and_grads(tf, model):
  """What SimAug's Trainer asks for (SimAug/code/pred_models.py:1996-2001):
  tf.gradients(model.loss, tf.trainable_variables())."""
  var = tf.trainable_variables()
  grads = tf.gradients(model.loss, var)
  return float(tf._v(model.loss).detach()), {
      v._name: (None if g is None else _np(tf, g)) for v, g in zip(var, grads)}


def forward(cfg, params, feed):
  """Tester.step outputs of SimAug's Model (test mode, greedy)."""
  tf, ref, model = build_model(cfg, params, feed, is_train=False)
  S = len(cfg.scene_grids)
  return ([_np(tf, model.grid_pred_decoded[s]) if cfg.use_grids[s] else [] for s in range(S)],
          [_np(tf, model.grid_pred_reg_decoded[s]) if cfg.use_grids[s] else [] for s in range(S)])


def run_white_box(cfg, params, feed, draws):
  """Model(config) with adv_train: the reference's own white_box_attack (its return values
  captured by wrapping the module-level function), build_tower on the adversarial features,
  build_loss.  -> dict(adv, target, loss, grads)."""
  tf, ref = import_simaug()
  cap = {}
  orig = ref.white_box_attack

  def wrapped(*a, **k):
    out = orig(*a, **k)
    cap["adv"], cap["target"] = _np(tf, out[0]), _np(tf, out[1])
    return out

  ref.white_box_attack = wrapped
  try:
    tf, ref, model = build_model(cfg, params, feed, is_train=True, draws=draws)
  finally:
    ref.white_box_attack = orig
  cap["loss"], cap["grads"] = _loss_and_grads(tf, model)
  return cap


def run_multiview(cfg, params, feed, draws):
  """Model(config) with multiview_train: the reference's own multiview_augmentation (return
  value and the attributes it leaves on the model captured), build_tower(mixup=True),
  build_loss.  -> dict(mixed, weight, [select, focal], loss, grads)."""
  tf, ref = import_simaug()
  cap = {}
  orig = ref.Model.multiview_augmentation

  def wrapped(self, obs_scene):
    out = orig(self, obs_scene)
    cap["mixed"] = _np(tf, out)
    cap["weight"] = float(tf._v(self.beta_weight))
    if hasattr(self, "focal_loss_weight"):
      cap["focal"] = _np(tf, self.focal_loss_weight)
      cap["select"] = _np(tf, self.selected_extra_indices)
    return out

  ref.Model.multiview_augmentation = wrapped
  try:
    tf, ref, model = build_model(cfg, params, feed, is_train=True, draws=draws)
  finally:
    ref.Model.multiview_augmentation = orig
  cap["loss"], cap["grads"] = _loss_and_grads(tf, model)
  return cap
