# coding=utf-8
"""Deterministic synthetic weights and `data_{split}.npz`-contract inputs.

There is no network here (no ActEV / Forking-Paths data, no checkpoint), so the
parity tests, `bench.py` and `__graft_entry__.smoke()` all draw from this one
generator.  Everything follows the reference's own recipes:

* weights use the reference's initialisers -- `variance_scaling(2.0)` for the
  `W` tensors of `conv2d` (reference `code/pred_models.py:1358-1368`), glorot
  uniform / zero bias for `ConvLSTMCell` (tf.contrib default);
* grid classes / regression targets are derived from (x, y) exactly as
  `code/preprocess.py:442-475`, with the centres of `code/preprocess.py:97-106`;
* scene features are one-hot uint8 masks `[F, 36, 64, 11]`
  (`code/preprocess.py:831-864`).

The dict returned by `make_npz_data` has the keys `pred_utils.read_data`
(`code/pred_utils.py:208-300`) reads from `data_{split}.npz`.
"""

from __future__ import annotations

import argparse
import math

import numpy as np

SEED_BASE = 20200614


def default_config(batch_size=4, use_grids=(1, 0), beam_size=1,
                   is_train=False, **overrides):
  """Namespace with the fields `Model` reads (authoritative list: reference
  `code/multifuture_inference.py:419-452` + the train-only fields)."""
  cfg = argparse.Namespace(
      modelname="model",
      batch_size=batch_size,
      obs_len=8, pred_len=12,
      is_train=is_train,
      emb_size=32,
      enc_hidden_size=256, dec_hidden_size=256,
      activation_func="tanh",
      keep_prob=1.0,
      scene_h=36, scene_w=64, scene_class=11,
      scene_conv_kernel=3, scene_conv_dim=64,
      convlstm_kernel=3,
      scene_grid_strides=[2, 4],
      use_grids=[bool(g) for g in use_grids],
      scene_grids=[(18, 32), (9, 16)],
      use_scene_enc=True,
      use_gnn=True,
      use_single_decoder=False,
      simaug_graph=False,         # SimAug fork: greedy graph attention without scene features
      use_soft_grid_class=False,
      soft_grid=1,
      use_teacher_forcing=False,
      train_w_onehot=True,
      use_gt_grid=False,
      mask_grid_regression=False,
      use_beam_search=beam_size > 1,
      beam_size=beam_size,
      diverse_beam=beam_size > 1,
      diverse_gamma=0.01,
      fix_num_timestep=1,
      # training
      wd=0.001,
      grid_loss_weight=1.0,
      grid_reg_loss_weight=0.2,
      init_lr=0.3,
      emb_lr=1.0,
      optimizer="adadelta",
      learning_rate_decay=0.95,
      num_epoch_per_decay=2.0,
      use_cosine_lr=False,
      clip_gradient_norm=10.0,
      train_num_examples=1000,
      num_epochs=80,
      video_h=1080, video_w=1920,
  )
  for k, v in overrides.items():
    setattr(cfg, k, v)
  return cfg


# ----------------------------------------------------------------- weights

def param_shapes(cfg):
  """TF-1 variable names -> shapes.  The names are
  inferred from TF1 scoping rules of reference `code/pred_models.py:140-305`
  (`raw_rnn(scope="decoder_rnn")` puts the cell and `grid_emb` under
  `<decoder scope>/decoder_rnn/`; `hidden2grid` re-enters the top scope)."""
  C = cfg.enc_hidden_size
  assert cfg.dec_hidden_size == C
  k = cfg.convlstm_kernel
  sk = cfg.scene_conv_kernel
  D = cfg.scene_conv_dim
  E = cfg.emb_size
  shapes = {}
  cin = cfg.scene_class
  for i in range(len(cfg.scene_grid_strides)):
    shapes["person_pred/scene_conv%d/W" % (i + 1)] = (sk, sk, cin, D)
    shapes["person_pred/scene_conv%d/b" % (i + 1)] = (D,)
    cin = D
  for s, use in enumerate(cfg.use_grids):
    if not use:
      continue
    p = "person_pred/"
    shapes[p + "encoder_grid_class_%d/enc_grid_%d/kernel" % (s, s)] = (k, k, D + C, 4 * C)
    shapes[p + "encoder_grid_class_%d/enc_grid_%d/biases" % (s, s)] = (4 * C,)
    shapes[p + "encoder_grid_reg_%d/enc_grid_regress_%d/kernel" % (s, s)] = (k, k, 2 + C, 4 * C)
    shapes[p + "encoder_grid_reg_%d/enc_grid_regress_%d/biases" % (s, s)] = (4 * C,)
    shapes[p + "decoder_grid_class_%d/decoder_rnn/dec_grid_%d/kernel" % (s, s)] = (k, k, E + C, 4 * C)
    shapes[p + "decoder_grid_class_%d/decoder_rnn/dec_grid_%d/biases" % (s, s)] = (4 * C,)
    shapes[p + "decoder_grid_class_%d/decoder_rnn/grid_emb/W" % s] = (3, 3, 1, E)
    shapes[p + "decoder_grid_class_%d/decoder_rnn/grid_emb/b" % s] = (E,)
    if getattr(cfg, "use_single_decoder", False):
      # --use_single_decoder (code/pred_models.py:287-296): no regression decoder; the offsets
      # come from the class decoder's states through ONE kernel shared by the scales (the
      # scope "decode_reg" does not carry the scale index).  The regression ENCODER is still
      # built (:232-234), so its variables exist in the graph and in checkpoints.
      shapes[p + "hidden2grid_decoder_grid_class_%d/out_dec_grid/W" % s] = (3, 3, C, 1)
      shapes[p + "decode_reg/out_dec_grid/W"] = (3, 3, C, 2)
      continue
    # (insertion order is the order make_params draws in: the fixtures depend on it)
    shapes[p + "decoder_grid_reg_%d/decoder_rnn/dec_grid_reg_%d/kernel" % (s, s)] = (k, k, E + C, 4 * C)
    shapes[p + "decoder_grid_reg_%d/decoder_rnn/dec_grid_reg_%d/biases" % (s, s)] = (4 * C,)
    shapes[p + "decoder_grid_reg_%d/decoder_rnn/grid_emb/W" % s] = (3, 3, 2, E)
    shapes[p + "decoder_grid_reg_%d/decoder_rnn/grid_emb/b" % s] = (E,)
    shapes[p + "hidden2grid_decoder_grid_class_%d/out_dec_grid/W" % s] = (3, 3, C, 1)
    shapes[p + "hidden2grid_decoder_grid_reg_%d/out_dec_grid/W" % s] = (3, 3, C, 2)
  return shapes


def make_params(cfg, seed=SEED_BASE, recurrent_gain=1.0, bias_scale=0.0):
  """Random-init weights with the reference's initialisers.

  `recurrent_gain` scales the ConvLSTM kernels (SURVEY.md §8d suggests 1.5 to
  keep top-1/top-2 logit margins away from zero); `bias_scale` > 0 draws
  non-zero biases so that a bias bug cannot hide behind the zero initialiser.
  """
  rng = np.random.default_rng(seed)
  params = {}
  for name, shape in param_shapes(cfg).items():
    leaf = name.rsplit("/", 1)[1]
    if leaf == "kernel":  # glorot uniform (tf.contrib ConvLSTMCell default)
      fan_in = shape[0] * shape[1] * shape[2]
      fan_out = shape[0] * shape[1] * shape[3]
      lim = math.sqrt(6.0 / (fan_in + fan_out)) * recurrent_gain
      w = rng.uniform(-lim, lim, size=shape)
    elif leaf == "W":  # variance_scaling(2.0), fan_in, truncated normal at 2 sigma
      fan_in = shape[0] * shape[1] * shape[2]
      std = math.sqrt(2.0 / fan_in) / 0.87962566103423978
      w = rng.normal(0.0, 1.0, size=shape)
      bad = np.abs(w) > 2.0
      while bad.any():
        w[bad] = rng.normal(0.0, 1.0, size=int(bad.sum()))
        bad = np.abs(w) > 2.0
      w = w * std
    else:  # biases / b
      w = rng.normal(0.0, 1.0, size=shape) * bias_scale
    params[name] = np.ascontiguousarray(w, dtype=np.float32)
  return params


# ------------------------------------------------------------------ inputs

def grid_centers(cfg):
  """[H, W, 2] (x, y) centre of every cell, float64; reference
  `code/preprocess.py:97-106`."""
  centers = []
  for h, w in cfg.scene_grids:
    h_gap, w_gap = cfg.video_h * 1.0 / h, cfg.video_w * 1.0 / w
    cx = np.cumsum([w_gap for _ in range(w)]) - w_gap / 2.0
    cy = np.cumsum([h_gap for _ in range(h)]) - h_gap / 2.0
    cxx = np.tile(np.expand_dims(cx, axis=0), [h, 1])
    cyy = np.tile(np.expand_dims(cy, axis=1), [1, w])
    centers.append(np.stack((cxx, cyy), axis=-1))
  return centers


def grid_class_and_targets(cfg, traj):
  """traj [M, T, 2] float64 -> (classes [M, n_scale, T] int32,
  targets_all list_s [M, T, H, W, 2] float32).  Reference
  `code/preprocess.py:442-475`."""
  M, T, _ = traj.shape
  centers = grid_centers(cfg)
  classes = np.zeros((M, len(cfg.scene_grids), T), dtype="int32")
  targets = []
  for i, (center, (h, w)) in enumerate(zip(centers, cfg.scene_grids)):
    h_gap, w_gap = cfg.video_h * 1.0 / h, cfg.video_w * 1.0 / w
    xi = np.asarray(np.ceil(traj[:, :, 0] / w_gap), dtype="int")
    yi = np.asarray(np.ceil(traj[:, :, 1] / h_gap), dtype="int")
    xi[xi == 0] = 1
    yi[yi == 0] = 1
    xi -= 1
    yi -= 1
    classes[:, i, :] = yi * w + xi
    allt = traj[:, :, None, None, :] - center[None, None]  # [M,T,h,w,2] f64
    targets.append(allt.astype("float32"))
  return classes, targets


def make_trajectories(rng, M, T, cfg):
  """AR(1) random walks inside the frame."""
  start = np.stack([rng.uniform(200, cfg.video_w - 200, size=M),
                    rng.uniform(150, cfg.video_h - 150, size=M)], axis=-1)
  vel = rng.normal(0.0, 25.0, size=(M, 2))
  pts = np.zeros((M, T, 2), dtype="float64")
  pos = start.copy()
  for t in range(T):
    pts[:, t] = pos
    vel = 0.8 * vel + math.sqrt(1 - 0.64) * rng.normal(0.0, 25.0, size=(M, 2))
    pos = pos + vel
    pos[:, 0] = np.clip(pos[:, 0], 1.0, cfg.video_w - 1.0)
    pos[:, 1] = np.clip(pos[:, 1], 1.0, cfg.video_h - 1.0)
  return pts


def make_scene_feat(rng, F, cfg):
  """[F, SH, SW, SC] uint8 one-hot masks from random rectangles."""
  SH, SW, SC = cfg.scene_h, cfg.scene_w, cfg.scene_class
  out = np.zeros((F, SH, SW, SC), dtype="uint8")
  for f in range(F):
    lab = np.zeros((SH, SW), dtype="int64")
    for _ in range(int(rng.integers(4, 9))):
      y0 = int(rng.integers(0, SH - 2))
      x0 = int(rng.integers(0, SW - 2))
      y1 = int(rng.integers(y0 + 1, SH + 1))
      x1 = int(rng.integers(x0 + 1, SW + 1))
      lab[y0:y1, x0:x1] = int(rng.integers(1, SC))
    hh = np.repeat(np.arange(SH), SW).reshape(SH, SW)
    ww = np.tile(np.arange(SW), SH).reshape(SH, SW)
    out[f, hh, ww, lab] = 1
  return out


def make_npz_data(cfg, num_examples, seed=SEED_BASE, frames_per_group=4,
                  float32_traj=False):
  """The dict `np.savez(data_{split}.npz)` would hold
  (`code/preprocess.py:670-866`), restricted to the keys the hot path and
  `pred_utils.evaluate` read.  float32_traj: round the walk to float32 BEFORE the
  grid classes / regression maps are derived, as the reference does
  (`preprocess.py:308` reads the trajectory file into float32), so that
  `*_grid_target_all` == float32(float64(obs_traj) - grid_center) exactly; the
  default keeps the float64 walk the committed golden fixtures were made from."""
  rng = np.random.default_rng(seed)
  T = cfg.obs_len + cfg.pred_len
  traj = make_trajectories(rng, num_examples, T, cfg)
  if float32_traj:
    traj = traj.astype("float32").astype("float64")
  classes, targets = grid_class_and_targets(cfg, traj)
  F = int(math.ceil(num_examples / float(frames_per_group)))
  scene_feat = make_scene_feat(rng, F, cfg)
  scene_ids = (np.arange(num_examples) // frames_per_group).astype("int64")
  scene = np.tile(scene_ids[:, None, None], [1, T, 1])
  rel = np.zeros_like(traj)
  rel[:, 1:] = traj[:, 1:] - traj[:, :-1]
  data = {
      "obs_traj": traj[:, :cfg.obs_len].astype("float32"),
      "pred_traj": traj[:, cfg.obs_len:].astype("float32"),
      "obs_traj_rel": rel[:, :cfg.obs_len].astype("float32"),
      "pred_traj_rel": rel[:, cfg.obs_len:].astype("float32"),
      "obs_grid_class": classes[:, :, :cfg.obs_len],
      "pred_grid_class": classes[:, :, cfg.obs_len:],
      "obs_scene": scene[:, :cfg.obs_len],
      "pred_scene": scene[:, cfg.obs_len:],
      "scene_feat": scene_feat,
      "video_wh": (cfg.video_w, cfg.video_h),
      "scene_grid_strides": list(cfg.scene_grid_strides),
  }
  data["_traj64"] = traj       # the unrounded walk the maps below were derived from
  for i, c in enumerate(grid_centers(cfg)):
    data["grid_center_%d" % i] = c
    data["obs_grid_target_all_%d" % i] = targets[i][:, :cfg.obs_len]
    data["pred_grid_target_all_%d" % i] = targets[i][:, cfg.obs_len:]
  return data


def make_feed(cfg, seed=SEED_BASE, frames_per_group=4, pred_len=None):
  """One batch of engine inputs (the arrays `Model.get_feed_dict` produces,
  reference `code/pred_models.py:1042-1194`), built without the Dataset
  machinery -- used by bench.py and the kernel-level tests."""
  N = cfg.batch_size
  data = make_npz_data(cfg, N, seed=seed, frames_per_group=frames_per_group)
  feed = {
      "obs_scene": data["obs_scene"][:, :, 0].astype("int32"),
      "scene_feat": data["scene_feat"].astype("float32"),
      "pred_length": int(pred_len if pred_len is not None else cfg.pred_len),
      "grid_obs_labels": [],
      "grid_obs_regress": [],
      "grid_pred_labels": [],
      "grid_pred_regress": [],
  }
  for s in range(len(cfg.scene_grids)):
    feed["grid_obs_labels"].append(
        np.ascontiguousarray(data["obs_grid_class"][:, s, :], dtype="int32"))
    feed["grid_obs_regress"].append(
        np.ascontiguousarray(data["obs_grid_target_all_%d" % s], dtype="float32"))
    feed["grid_pred_labels"].append(
        np.ascontiguousarray(data["pred_grid_class"][:, s, :], dtype="int32"))
    feed["grid_pred_regress"].append(
        np.ascontiguousarray(data["pred_grid_target_all_%d" % s], dtype="float32"))
  # compact form of the same batch (Engine.upload_compact): coordinates + centres
  feed["obs_xy"] = np.ascontiguousarray(data["_traj64"][:, :cfg.obs_len])
  feed["pred_xy"] = np.ascontiguousarray(data["_traj64"][:, cfg.obs_len:])
  feed["grid_centers"] = [data["grid_center_%d" % s] for s in range(len(cfg.scene_grids))]
  feed["scene_feat_u8"] = data["scene_feat"]
  return feed
