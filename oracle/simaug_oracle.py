# coding=utf-8
"""CPU ORACLE for the SimAug training extras  --  TEST INFRASTRUCTURE ONLY.

Restates `white_box_attack` (SimAug/code/pred_models.py:60-172) and
`multiview_augmentation` (:346-543, experiments 1 / 2 / 4) on the torch-CPU oracle of the
network (oracle/multiverse_oracle.py; SimAug's `build_tower` :544 is the same graph as
code/pred_models.py build_forward), with torch.autograd playing `tf.gradients(loss, input)`.

**Pinned to the reference** (round 2): SimAug/code/pred_models.py is executed UNMODIFIED on the
TF-1 shim (oracle/tf1_shim/run_simaug.py) with its unseeded random ops replayed from the same
`Draws` object, and frozen in tests/golden/golden_simaug.npz by make_simaug_golden.py: six
white_box_attack configurations, the four multi-view experiments (experiment 3 twice) and
SimAug's greedy forward.  tests/test_reference_pin.py holds the functions below to those runs
(adversarial / mixed features equal element for element, target labels, beta weight, selected
view, focal weights, the loss of the training step that follows to 2e-6, gradients);
tests/test_gpu_simaug.py holds the HIP engine + multiverse_amd/simaug.py to the same file.
That run also showed that SimAug's graph differs from code/pred_models.py in one place: its
greedy decoder's graph attention sees the hidden state alone (`simaug_graph`).
"""
from __future__ import annotations

import copy

import numpy as np
import torch

from oracle import multiverse_oracle as oracle


def _scale_of(cfg):
  assert sum(bool(u) for u in cfg.use_grids) == 1, "only one scale for adv train"   # :292
  return [i for i, u in enumerate(cfg.use_grids) if u][0]


def class_loss_and_input_grad(params, cfg, feed, scene, target, dtype=torch.float32):
  """one_step_attack's loss (:94-110): per-row sparse softmax cross entropy of the target
  scale's logits against `target`; returns (per-sample mean loss [N], d sum(loss) / d scene)."""
  s = _scale_of(cfg)
  H, W = cfg.scene_grids[s]
  P = oracle.Params(params, dtype)
  x = torch.tensor(np.asarray(scene), dtype=dtype, requires_grad=True)
  f = dict(feed)
  f["scene_feat"] = x
  cls_out, _, _ = oracle.forward_tensors(P, cfg, f, dtype)
  logits = cls_out[s].reshape(-1, H * W)
  lab = torch.from_numpy(np.asarray(target).astype("int64").reshape(-1))
  ce = torch.logsumexp(logits, dim=-1) - logits.gather(1, lab[:, None])[:, 0]
  g, = torch.autograd.grad(ce.sum(), x)
  N = cfg.batch_size
  return ce.detach().reshape(N, -1).mean(dim=1).numpy(), g.numpy()


def fgsm_step(x, g, clean, eps, step):
  """:118-123 -- x - step * sign(g), clipped to [clip(clean - eps, -1, 1), clip(clean + eps, -1, 1)]"""
  lo = np.clip(clean - eps, -1.0, 1.0)
  hi = np.clip(clean + eps, -1.0, 1.0)
  return np.minimum(np.maximum(x - np.sign(g) * np.float32(step), lo), hi).astype("float32")


def white_box_attack(params, cfg, feed, draws, mirror, norm_feat=False):
  """`mirror` = multiverse_amd.simaug (its pure-numpy helpers random_targets / start_adv
  consume `draws` in the reference's order); -> (adv features, target labels)."""
  s = _scale_of(cfg)
  h, w = cfg.scene_grids[s]
  target = mirror.random_targets(feed["grid_pred_labels"][s], h * w, draws)
  clean = np.asarray(feed["scene_feat"], dtype="float32")
  tcfg = copy.copy(cfg)
  tcfg.is_train = True

  def attack(start):
    x = mirror._softmax_last(start) if norm_feat else start
    if cfg.adv_use_fgsm:
      _, g = class_loss_and_input_grad(params, tcfg, feed, x, target)
      return fgsm_step(x, g, clean, cfg.adv_epsilon, cfg.adv_epsilon)
    for _ in range(int(cfg.adv_num_iter)):
      _, g = class_loss_and_input_grad(params, tcfg, feed, x, target)
      x = fgsm_step(x, g, clean, cfg.adv_epsilon, cfg.adv_step_size)
    return x

  adv = attack(mirror.start_adv(clean, cfg, draws))
  if getattr(cfg, "use_mixup", False):
    weight = np.float32(draws.beta(cfg.mixup_alpha))
    if getattr(cfg, "mixup_mix_adv", False):
      adv2 = attack(mirror.start_adv(clean, cfg, draws))
      adv = adv2 * weight + adv * (np.float32(1) - weight)
    else:
      adv = clean * weight + adv * (np.float32(1) - weight)
  return adv.astype("float32"), target


def multiview_augmentation(params, cfg, feed, extra_pred_labels, draws, mirror,
                           extra_scene=None):
  """`Model.multiview_augmentation` (SimAug/code/pred_models.py:346-543), all four
  experiments, on the oracle.  feed: per-step scene feed of the N samples (norm_input already
  applied); extra_pred_labels [N, M, T_p]; extra_scene [N, M, T_o, SH, SW, SC]: the RAW
  features of the extra views (experiment 3 only).
  -> dict(mixed, weight, adv_loss [N, M], and for experiment 3: select [N], focal [N])."""
  N, M, T = cfg.batch_size, int(cfg.multiview_max_num), cfg.obs_len
  mcfg = copy.copy(cfg)
  mcfg.batch_size = N * M
  mcfg.is_train = True
  clean = np.asarray(feed["scene_feat"], dtype="float32")
  shp = clean.shape[1:]
  tile_clean = np.repeat(clean.reshape((N, T) + shp), M, axis=0).reshape((-1,) + shp)
  targets = np.asarray(extra_pred_labels, dtype="int32").reshape(N * M, -1)

  def one_step_attack(features):
    start = mirror.start_adv(features, cfg, draws)          # bounds: start +- eps (:400-403)
    tiled = mirror._tile_feed(cfg, feed, M, start)
    loss, g = class_loss_and_input_grad(params, mcfg, tiled, start, targets)
    return fgsm_step(start, g, start, cfg.adv_epsilon, cfg.adv_epsilon), loss.reshape(N, M)

  adv_flat, adv_loss = one_step_attack(tile_clean)
  exp = int(cfg.multiview_exp)
  if exp == 3 and getattr(cfg, "multiview_use_adv_for_loss", False):
    _, adv_loss = one_step_attack(adv_flat)
  adv_out = adv_flat.reshape((N, M, T) + shp)
  order = np.argsort(-adv_loss, axis=1, kind="stable")      # tf.nn.top_k(sorted=True)
  rows = np.arange(N)
  out = {"adv_loss": adv_loss}
  if exp == 1:
    f1, f2 = adv_out[rows, order[:, 0]], adv_out[rows, order[:, 1]]
  elif exp == 4:
    f1, f2 = adv_out[rows, order[:, M - 1]], adv_out[rows, order[:, M - 2]]
  elif exp == 2:
    i1 = draws.index(N, 0, M)
    i2 = np.mod(i1 + draws.index(N, 1, M), M)
    f1, f2 = adv_out[rows, i1], adv_out[rows, i2]
  else:
    top = adv_loss[rows, order[:, 0]].astype("float32")
    out["focal"] = ((np.float32(1.0) - np.exp(-top)) ** np.float32(cfg.fl_gamma)).astype("float32")
    select = order[:, 0].astype("int32")
    if getattr(cfg, "multiview_random", False):
      select = draws.index(N, 0, M)
    out["select"] = select
    f1 = adv_out[rows, order[:, 0]]
    f2 = np.asarray(extra_scene, dtype="float32")[rows, select]
  weight = draws.beta(cfg.mixup_alpha)
  if getattr(cfg, "multiview_max_weight_for_first", False):
    weight = max(weight, 1.0 - weight)
  out["weight"] = weight
  w32 = np.float32(weight)          # the Beta sample is a float32 tensor in the reference
  out["mixed"] = (f1 * w32 + f2 * (np.float32(1.0) - w32)).astype("float32").reshape(
      (N * T,) + shp)
  return out
