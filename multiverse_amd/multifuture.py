# coding=utf-8
"""Multi-future (Forking Paths) host pipeline around the beam-search decode:
the callers either side of the hot path.  numpy only.

  inputs      code/multifuture_inference.py:78-272   traj txt + scene-seg npy -> arrays
  feed        code/multifuture_inference.py:304-385  PredictionModelInference.get_feed_dict
  decode      code/multifuture_inference.py:475-523  ids + offsets -> pixel trajectories
  minADE/FDE  code/multifuture_eval_trajs.py:16-88
  grid NLL    code/multifuture_eval_trajs_prob.py:19-119

File formats (forking_paths_dataset/code/get_prepared_data_multifuture.py:192-251):
  traj_2.5fps/<scene>_<moment>_<pid>_<cam>.txt   lines "frame\\tpid\\tx\\ty"
  scene_seg/<traj_id>/<traj_id>_F_%08d.npy        int class map [36, 64]
  multifuture/<traj_id>.p                         {future_id: {"x_agent_traj": [(frame, pid, x, y)]}}
  scene_id2name json                              {"oldid2new": {...}, "id2name": {...}}

The reference runs one sample at a time (batch 1, beam 20, run-time T_pred);
`run_inference` keeps that order of results and additionally batches samples
that share T_pred when the engine was created with batch_size > 1.
"""

from __future__ import annotations

import json
import os
import pickle

import numpy as np


def load_traj(traj_file):
  """code/multifuture_inference.py:78-85."""
  rows = []
  with open(traj_file, "r") as f:
    for line in f:
      line = line.strip()
      if line:
        rows.append(line.split("\t"))
  return np.array(rows, dtype="float32")


def add_grid(args):
  """code/multifuture_inference.py:88-113: grid sizes (Python round), cell
  centres and cell box sizes per scale."""
  args.scene_grid_strides = [int(o) for o in str(args.grid_strides).split(",")]
  assert args.scene_grid_strides
  args.num_scene_grid = len(args.scene_grid_strides)
  if isinstance(args.use_grids, str):
    args.use_grids = [bool(int(o)) for o in args.use_grids.split(",")]
  args.scene_grids = []
  for stride in args.scene_grid_strides:
    args.scene_grids.append((int(round(args.scene_h * 1.0 / stride)),
                             int(round(args.scene_w * 1.0 / stride))))
  args.scene_grid_centers = []
  args.grid_box_sizes = []
  for h, w in args.scene_grids:
    h_gap, w_gap = args.video_h * 1.0 / h, args.video_w * 1.0 / w
    args.grid_box_sizes.append((h_gap, w_gap))
    cx = np.cumsum([w_gap for _ in range(w)]) - w_gap / 2.0
    cy = np.cumsum([h_gap for _ in range(h)]) - h_gap / 2.0
    args.scene_grid_centers.append(np.stack(
        (np.tile(cx[None, :], [h, 1]), np.tile(cy[:, None], [1, w])), axis=-1))
  return args


def xy_to_grid_class(xy, h, w, video_h, video_w):
  """ceil(x / w_gap) (0 -> 1) - 1, likewise y; class = y * w + x
  (code/multifuture_inference.py:122-141, multifuture_eval_trajs_prob.py:47-66)."""
  xy = np.asarray(xy)
  xi = np.asarray(np.ceil(xy[:, 0] / (video_w * 1.0 / w)), dtype="int")
  yi = np.asarray(np.ceil(xy[:, 1] / (video_h * 1.0 / h)), dtype="int")
  xi[xi == 0] = 1
  yi[yi == 0] = 1
  return (yi - 1) * w + (xi - 1)


def get_grid_input(args, traj):
  """traj [obs_length, 2] -> (grid_class [n_scale, T] int32, targets list_s
  [T, h, w, 2] float32 = xy - centre).  code/multifuture_inference.py:115-156."""
  T = len(traj)
  grid_class = np.zeros([len(args.scene_grids), T], dtype="int32")
  targets = []
  for i, (center, (h, w)) in enumerate(zip(args.scene_grid_centers, args.scene_grids)):
    grid_class[i] = xy_to_grid_class(traj, h, w, args.video_h, args.video_w)
    targets.append((traj[:, None, None, :] - center[None]).astype("float32"))
  return grid_class, targets


def load_scene_id_map(path):
  """code/multifuture_inference.py:172-184 -> (oldid2new dict incl. 0 -> 0,
  total_scene_class)."""
  with open(path, "r") as f:
    m = json.load(f)
  old2new = {int(k): v for k, v in m["oldid2new"].items()}
  assert 0 not in old2new
  old2new[0] = 0
  id2name = dict(m["id2name"])
  id2name[0] = "BG"
  assert len(old2new) == len(id2name)
  return old2new, len(old2new)


def scene_to_masks(scene_feat, old2new, total_class):
  """int class map [H, W] -> uint8 one-hot [H, W, total_class]; ids missing from
  the map go to background 0 (code/multifuture_inference.py:242-258)."""
  sf = np.asarray(scene_feat)
  lut_size = int(max(int(sf.max()) if sf.size else 0, max(old2new))) + 1
  lut = np.zeros(lut_size, dtype=np.int64)
  for k, v in old2new.items():
    lut[k] = v
  neg = sf < 0
  new = lut[np.where(neg, 0, sf).astype(np.int64)]
  new[neg] = 0
  out = np.zeros(sf.shape + (total_class,), dtype="uint8")
  hh, ww = np.meshgrid(np.arange(sf.shape[0]), np.arange(sf.shape[1]), indexing="ij")
  out[hh, ww, new] = 1
  return out


def get_inputs(args, traj_files, gt_trajs):
  """code/multifuture_inference.py:158-272."""
  old2new, total_class = load_scene_id_map(args.scene_id2name)
  out = {"obs_traj": [], "obs_traj_rel": [], "obs_grid_class": [], "obs_grid_target": [],
         "obs_scene": [], "max_pred_lengths": []}
  scene_feats = []
  for traj_file in traj_files:
    traj_id = os.path.splitext(os.path.basename(traj_file))[0]
    _, _, x_agent_pid, _ = traj_id.split("_")
    x_agent_pid = int(x_agent_pid)
    data = load_traj(traj_file)
    frame_idxs = np.unique(data[:, 0]).tolist()
    obs = data[x_agent_pid == data[:, 1], 2:]
    assert len(obs) == args.obs_length, (traj_id, obs.shape)
    rel = np.zeros_like(obs)
    rel[1:] = obs[1:] - obs[:-1]
    grid_class, grid_target = get_grid_input(args, obs)
    featidx = np.zeros([args.obs_length, 1], dtype="int32")
    for i, frame_idx in enumerate(frame_idxs):
      featidx[i, 0] = len(scene_feats)
      scene_feats.append(np.load(os.path.join(
          args.scene_feat_path, traj_id, "%s_F_%08d.npy" % (traj_id, frame_idx))))
    out["obs_traj"].append(obs)
    out["obs_traj_rel"].append(rel)
    out["obs_scene"].append(featidx)
    out["obs_grid_class"].append(grid_class)
    out["obs_grid_target"].append(grid_target)
    out["max_pred_lengths"].append(max(
        len(gt_trajs[traj_id][fid]["x_agent_traj"]) for fid in gt_trajs[traj_id]))
  feats = np.zeros([len(scene_feats), args.scene_h, args.scene_w, total_class],
                   dtype="uint8")
  for k, sf in enumerate(scene_feats):
    feats[k] = scene_to_masks(sf, old2new, total_class)
  out["scene_feats"] = feats
  return out


def inference_feed(inputs, args, idxs, batch_size=None):
  """Engine feed for the samples `idxs` (all with the same T_pred), padded to
  `batch_size` by repeating the last one.  For one sample this is the feed of
  PredictionModelInference.get_feed_dict (code/multifuture_inference.py:304-385):
  the scene table is compacted to the frames the batch uses, in first-use order."""
  idxs = list(idxs)
  n_real = len(idxs)
  N = batch_size or n_real
  idxs = idxs + [idxs[-1]] * (N - n_real)
  T_in = args.obs_length
  T_pred = inputs["max_pred_lengths"][idxs[0]]
  assert all(inputs["max_pred_lengths"][i] == T_pred for i in idxs)
  feed = {"pred_length": int(T_pred), "grid_obs_labels": [], "grid_obs_regress": []}
  for j, (h, w) in enumerate(args.scene_grids):
    feed["grid_obs_labels"].append(np.stack(
        [inputs["obs_grid_class"][i][j] for i in idxs]).astype("int32"))
    if not args.use_grids[j]:
      feed["grid_obs_regress"].append(None)
      continue
    feed["grid_obs_regress"].append(np.stack(
        [inputs["obs_grid_target"][i][j] for i in idxs]).astype("float32"))
  old2new = {}
  obs_scene = np.zeros((N, T_in), dtype="int32")
  for r, i in enumerate(idxs):
    for t in range(T_in):
      old = int(inputs["obs_scene"][i][t][0])
      if old not in old2new:
        old2new[old] = len(old2new)
      obs_scene[r, t] = old2new[old]
  scene_feat = np.zeros((len(old2new), args.scene_h, args.scene_w, args.scene_class),
                        dtype="float32")
  for old, new in old2new.items():
    scene_feat[new] = inputs["scene_feats"][old]
  feed["obs_scene"] = obs_scene
  feed["scene_feat"] = scene_feat
  return feed, n_real


def decode_trajectories(args, class_output, reg_output, beam_outputs, pred_len,
                        use_grid_idx):
  """One sample's outputs -> `num_out` trajectories of `pred_len` (x, y) points
  (code/multifuture_inference.py:475-517).  class_output [T,H,W,1] (greedy),
  reg_output [T,H,W,2], beam_outputs (logits [B,T,K], ids [B,T], logprobs [B])."""
  reg = np.asarray(reg_output).reshape([pred_len, -1, 2])
  centers = args.scene_grid_centers[use_grid_idx].reshape([-1, 2])

  def point(t, cls):
    return centers[cls] if args.center_only else centers[cls] + reg[t, cls]

  if args.greedy:
    sel = np.argmax(np.asarray(class_output).reshape([pred_len, -1]), axis=1)
    one = [point(t, sel[t]) for t in range(pred_len)]
    return [one for _ in range(args.num_out)]
  ids = beam_outputs[1]
  return [[point(t, ids[j, t]) for t in range(pred_len)] for j in range(args.num_out)]


def model_config(args, batch_size=1, max_pred_len=None):
  """The Namespace the reference builds for the model
  (code/multifuture_inference.py:419-452)."""
  import argparse
  return argparse.Namespace(
      modelname="model", batch_size=batch_size,
      beam_size=args.num_out, use_beam_search=not args.greedy,
      diverse_beam=args.diverse_beam, diverse_gamma=args.diverse_gamma,
      fix_num_timestep=args.fix_num_timestep,
      use_teacher_forcing=False, is_train=False,
      scene_h=args.scene_h, scene_w=args.scene_w, scene_class=args.scene_class,
      use_soft_grid_class=args.use_soft_grid_class,
      use_single_decoder=args.use_single_decoder,
      obs_len=args.obs_length, pred_len=12,
      max_pred_len=max_pred_len or 12,
      emb_size=args.emb_size, enc_hidden_size=args.enc_hidden_size,
      dec_hidden_size=args.dec_hidden_size, activation_func="tanh",
      scene_conv_kernel=args.scene_conv_kernel, use_scene_enc=args.use_scene_enc,
      scene_conv_dim=args.scene_conv_dim, convlstm_kernel=args.convlstm_kernel,
      use_gnn=args.use_gnn, keep_prob=1.0,
      scene_grid_strides=args.scene_grid_strides, scene_grids=args.scene_grids,
      use_grids=args.use_grids)


def run_inference(args, model, inputs, traj_ids):
  """The per-sample loop of code/multifuture_inference.py:458-523 ->
  (output_data {traj_id: [num_out][T][2]}, beam_prob {traj_id: (logits
  [1,B,T,K], logprobs [1,B])}).  `model.run_forward(feed)` is one sess.run."""
  use_grid_idx = list(args.use_grids).index(True)
  N = model.config.batch_size
  by_len = {}
  for i in range(len(traj_ids)):
    by_len.setdefault(inputs["max_pred_lengths"][i], []).append(i)
  output_data, beam_prob = {}, {}
  for T_pred in sorted(by_len):
    group = by_len[T_pred]
    for lo in range(0, len(group), N):
      idxs = group[lo:lo + N]
      feed, n_real = inference_feed(inputs, args, idxs, batch_size=N)
      cls, reg, beam = model.run_forward(feed)
      # --use_single_decoder with beam search: the offsets come per beam, [N*B, T, H, W, 2]
      # (code/pred_models.py:287-296); the reference's script reshapes the B rows of its one
      # sample as [1, T, -1, 2] (code/multifuture_inference.py:478) -- kept as it is
      rows_per = reg[use_grid_idx].shape[0] // N
      for r in range(n_real):
        i = idxs[r]
        b = None if beam is None else (beam[0][r], beam[1][r], beam[2][r])
        reg_r = (reg[use_grid_idx][r] if rows_per == 1
                 else reg[use_grid_idx][r * rows_per:(r + 1) * rows_per])
        output_data[traj_ids[i]] = decode_trajectories(
            args, cls[use_grid_idx][r], reg_r, b, T_pred, use_grid_idx)
        if b is not None and getattr(args, "save_prob_file", None) is not None:
          beam_prob[traj_ids[i]] = (b[0][None], b[2][None])
  ordered = {t: output_data[t] for t in traj_ids}
  return ordered, ({t: beam_prob[t] for t in traj_ids if t in beam_prob})


# ------------------------------------------------------------------ metrics

def _get_min(errors):
  sums = [sum(e) for e in errors]
  idx = sums.index(min(sums))
  return errors[idx], idx


def eval_min_ade_fde(gt_by_traj, prediction):
  """minADE_K / minFDE_K (code/multifuture_eval_trajs.py:16-88): for every GT
  future, the prediction with the smallest summed (resp. final) error; errors
  pooled over all GT timesteps; cam4 = top-down, the rest = 45-degree.
  Returns {"ade": {...}, "fde": {...}} over ("45-degree", "top-down", "all")."""
  keys = ("45-degree", "top-down", "all")
  ade = {k: [] for k in keys}
  fde = {k: [] for k in keys}
  for traj_id in prediction:
    camera = traj_id.split("_")[-1]
    gt = gt_by_traj[traj_id]
    for future_id in gt:
      gt_traj = np.array([one[2:] for one in gt[future_id]["x_agent_traj"]])
      pred_len = len(gt_traj)
      a_err, f_err = [], []
      for pred_out in prediction[traj_id]:
        assert len(pred_out) >= pred_len
        d = np.sqrt(np.sum((gt_traj - np.asarray(pred_out)[:pred_len]) ** 2, axis=1))
        a_err.append(d.tolist())
        f_err.append([d[-1]])
      min_a, _ = _get_min(a_err)
      min_f, _ = _get_min(f_err)
      view = "top-down" if camera == "cam4" else "45-degree"
      ade[view] += min_a
      fde[view] += min_f
      ade["all"] += min_a
      fde["all"] += min_f
  mean = lambda v: float(np.mean(v)) if len(v) else float("nan")
  return {"ade": {k: mean(ade[k]) for k in keys}, "fde": {k: mean(fde[k]) for k in keys}}


def _softmax(x, axis=None):
  x = x - x.max(axis=axis, keepdims=True)
  y = np.exp(x)
  return y / y.sum(axis=axis, keepdims=True)


def eval_grid_nll(gt_by_traj, predictions, scene_h=18, scene_w=32, video_h=1080,
                  video_w=1920, time_list=(0, 1, 2, 3, 4)):
  """Grid negative log-likelihood at T=1..5 (code/multifuture_eval_trajs_prob.py:
  69-119): beams' per-step softmax maps mixed with softmax(beam logprobs)."""
  nlls = {"T=%d" % (t + 1): [] for t in time_list}
  for traj_id in predictions:
    gt = gt_by_traj[traj_id]
    beams, logprobs = predictions[traj_id]
    probs = _softmax(np.squeeze(logprobs))
    beams = _softmax(np.squeeze(beams), axis=-1)          # [B, T, K]
    assert beams.shape[-1] == scene_h * scene_w
    for t in time_list:
      xys = [gt[fid]["x_agent_traj"][t][2:] for fid in gt
             if len(gt[fid]["x_agent_traj"]) > t]
      if not xys:
        continue
      grid = (beams[:, t, :].astype("float32") * probs[:, None].astype("float32")).sum(0)
      idx = xy_to_grid_class(np.asarray(xys), scene_h, scene_w, video_h, video_w)
      nll = float(np.mean([-np.log(grid[k] + np.finfo(float).eps) for k in idx]))
      nlls["T=%d" % (t + 1)].append(nll)
  return {k: (float(np.mean(v)) if v else float("nan")) for k, v in nlls.items()}, \
      {k: len(v) for k, v in nlls.items()}


def load_gt(multifuture_path, traj_ids):
  out = {}
  for traj_id in traj_ids:
    with open(os.path.join(multifuture_path, "%s.p" % traj_id), "rb") as f:
      out[traj_id] = pickle.load(f)
  return out
