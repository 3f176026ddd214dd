# coding=utf-8
"""The SimAug (N4) cases frozen in tests/golden/golden_simaug.npz by
oracle/tf1_shim/make_simaug_golden.py -- runs of the reference's UNMODIFIED
SimAug/code/pred_models.py on the TF-1 shim with injected random draws -- and the inputs
that reproduce them (shared by the generator, the CPU oracle tests and the GPU tests)."""
import copy
import os

import numpy as np

from multiverse_amd import simaug, synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "golden_simaug.npz")
SEED = synth.SEED_BASE + 50
N, M = 2, 3
STRIDE = 97

SIMAUG_DEFAULTS = dict(
    norm_input=True, norm_feat=False, adv_train=False, multiview_train=False,
    standard_aug=False, adv_epsilon=0.1, adv_step_size=0.03, adv_num_iter=3,
    adv_start_from_clean_prob=0.0, adv_use_fgsm=True, use_mixup=False, mixup_alpha=1.0,
    mixup_mix_adv=False, multiview_max_num=M, multiview_exp=1,
    multiview_max_weight_for_first=False, multiview_random=False,
    multiview_use_adv_for_loss=False, fl_gamma=2.0, double_weighting=False)

WHITE_BOX = {          # name -> (config overrides, Draws seed)
    "fgsm": (dict(), 11),
    "pgd3": (dict(adv_use_fgsm=False), 11),
    "mix_clean": (dict(use_mixup=True), 11),
    "mix_adv": (dict(use_mixup=True, mixup_mix_adv=True), 11),
    "clean_start_norm_feat": (dict(adv_start_from_clean_prob=1.0, norm_feat=True), 11),
    "maybe_clean": (dict(adv_start_from_clean_prob=0.5), 11),
}
MULTIVIEW = {
    "exp1": (dict(multiview_exp=1), 21),
    "exp2": (dict(multiview_exp=2), 21),
    "exp4_maxw": (dict(multiview_exp=4, multiview_max_weight_for_first=True), 21),
    "exp3_dw": (dict(multiview_exp=3, double_weighting=True), 21),
    "exp3_random_advloss": (dict(multiview_exp=3, multiview_random=True,
                                 multiview_use_adv_for_loss=True), 21),
}


def config(is_train, **over):
  cfg = synth.default_config(batch_size=N, use_grids=(0, 1), simaug_graph=True,
                             is_train=is_train)
  for k, v in SIMAUG_DEFAULTS.items():
    setattr(cfg, k, v)
  for k, v in over.items():
    setattr(cfg, k, v)
  return cfg


def base_inputs(cfg):
  params = synth.make_params(cfg, seed=SEED, recurrent_gain=2.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=SEED + 1)
  return params, feed


def multiview_feed(cfg, feed):
  """The reference-side feed of a multi-view batch: the samples' own (raw 0/1) frames
  followed by M extra views per (n, t) in one scene table, and the extra views' labels.
  -> (reference feed, per-step engine feed with norm_input applied, extra_scene
  [N, M, T_o, SH, SW, SC] raw)."""
  rng = np.random.default_rng(5)
  To, Tp = cfg.obs_len, cfg.pred_len
  K = cfg.scene_grids[1][0] * cfg.scene_grids[1][1]
  own = np.asarray(feed["scene_feat"], "float32")[np.asarray(feed["obs_scene"]).reshape(-1)]
  extra = (rng.uniform(size=(N * M * To,) + own.shape[1:]) > 0.5).astype("float32")
  f0 = dict(feed)
  f0["scene_feat"] = np.concatenate([own, extra], axis=0)
  f0["obs_scene"] = np.arange(N * To, dtype="int32").reshape(N, To)
  f0["obs_scene_extra"] = (N * To + np.arange(N * M * To, dtype="int32")).reshape(N, M, To)
  f0["grid_obs_labels_extra"] = [None, rng.integers(0, K, size=(N, M, To)).astype("int32")]
  f0["grid_pred_labels_extra"] = [None, rng.integers(0, K, size=(N, M, Tp)).astype("int32")]
  pf = dict(f0)
  pf["scene_feat"] = own * np.float32(2.0) - np.float32(1.0)        # norm_input (:283-285)
  return f0, pf, extra.reshape((N, M, To) + own.shape[1:])


def digest(a):
  a = np.asarray(a, dtype=np.float32).reshape(-1)
  return np.concatenate([
      np.array([a.astype(np.float64).sum(), np.abs(a).astype(np.float64).sum(),
                np.abs(a).max()], dtype=np.float64), a[::STRIDE].astype(np.float64)])


def samples(a):
  return np.asarray(a, dtype=np.float32).reshape(-1)[::STRIDE]


def label_mixup_feed(pf, f0, mixed, weight, select, focal, double_weighting):
  """The training feed after experiment 3: mixed features + the mixed-up labels."""
  rows = np.arange(N)
  t = dict(pf, scene_feat=mixed)
  t["mix_weight"] = weight
  t["mix_obs_labels"] = [None, f0["grid_obs_labels_extra"][1][rows, select]]
  t["mix_pred_labels"] = [None, f0["grid_pred_labels_extra"][1][rows, select]]
  t["mix_sample_weight"] = focal if double_weighting else None
  return t


def train_config(cfg):
  """The config of the training step that FOLLOWS an augmentation: the features are already
  normalised / perturbed."""
  t = copy.copy(cfg)
  t.norm_input = False
  return t


def beam_config():
  """SimAug's beam-search decode (its gnn_edge DOES see the scene features there)."""
  cfg = config(False)
  cfg.use_beam_search, cfg.beam_size = True, 5
  cfg.diverse_beam, cfg.diverse_gamma, cfg.fix_num_timestep = True, 0.01, 1
  return cfg
