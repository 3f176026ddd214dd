# coding=utf-8
"""Host-side data / evaluation utilities with the reference's semantics
(`code/pred_utils.py`), TensorFlow-free.  Plain numpy; nothing here touches the
GPU -- these are the callers either side of the hot path:

  process_args        code/pred_utils.py:70-146   (grid dims from strides)
  read_data           code/pred_utils.py:208-300  (data_{split}.npz -> Dataset)
  Dataset.get_batches code/pred_utils.py:609-706  (padding, per-batch scene table)
  evaluate            code/pred_utils.py:354-586  (grid acc, ADE / FDE)
  initialize          code/pred_utils.py:149-205  (weights -> model; here from
                      an .npz keyed by TF variable name, see save_params)
"""

from __future__ import annotations

import collections
import itertools
import math
import os
import pickle
import random

import numpy as np


def process_args(args):
  """Derive the fields the model reads from the CLI flags.  Grid sizes use
  Python round() like the reference (code/pred_utils.py:127-132); the engine
  separately checks they match the stride-2 conv chain."""
  act = getattr(args, "activation_func", "tanh")
  if not callable(act) and act not in ("tanh", "relu", "lrelu"):
    # code/pred_utils.py:92-94: "unrecognied activation function, using relu..."
    print("unrecognied activation function, using relu...")
    args.activation_func = "relu"
  args.seq_len = args.obs_len + args.pred_len
  if getattr(args, "outbasepath", None) is not None:
    args.outpath = os.path.join(args.outbasepath, args.modelname,
                                str(args.runId).zfill(2))
    args.save_dir = os.path.join(args.outpath, "save")
    args.save_dir_model = os.path.join(args.save_dir, "save")
    args.save_dir_best = os.path.join(args.outpath, "best")
    args.save_dir_best_model = os.path.join(args.save_dir_best, "save-best")
    for d in (args.outpath, args.save_dir, args.save_dir_best):
      os.makedirs(d, exist_ok=True)
  if isinstance(args.scene_grid_strides, str):
    args.scene_grid_strides = [int(o) for o in args.scene_grid_strides.split(",")]
  if isinstance(args.use_grids, str):
    args.use_grids = [bool(int(o)) for o in args.use_grids.split(",")]
  assert len(args.scene_grid_strides) == len(args.use_grids)
  assert sum(args.use_grids) <= 2, \
      "Currently only supports at most two scale training at a time"
  args.scene_grids = []
  for stride in args.scene_grid_strides:
    this_h = int(round(args.scene_h * 1.0 / stride))
    this_w = int(round(args.scene_w * 1.0 / stride))
    args.scene_grids.append((this_h, this_w))
  if getattr(args, "load_best", False) or getattr(args, "load_from", None) is not None:
    args.load = True
  if not getattr(args, "is_train", False):
    args.load = True
    args.num_epochs = 1
    args.keep_prob = 1.0
  return args


# ------------------------------------------------------------------ weights

def save_params(path, params):
  """Light checkpoint = .npz keyed by TF-1 variable name, HWIO layout."""
  np.savez(path, **{k.replace("/", "__"): v for k, v in params.items()})


def load_params(path):
  with np.load(path) as z:
    return {k.replace("__", "/"): z[k] for k in z.files}


def _find_weights(path):
  """A weights source from a path: an .npz, a TF checkpoint prefix / .index, or
  a directory holding either (tf.train.get_checkpoint_state semantics for TF
  checkpoints, code/pred_utils.py:186-204; newest .npz otherwise)."""
  if os.path.isdir(path):
    if os.path.exists(os.path.join(path, "checkpoint")):
      return "tf", path
    cands = sorted(f for f in os.listdir(path) if f.endswith(".npz"))
    if cands:
      return "npz", os.path.join(path, cands[-1])
    raise IOError("Model not exists: no checkpoint under %s" % path)
  if path.endswith(".npz") and os.path.exists(path):
    return "npz", path
  if os.path.exists(path + ".index") or (path.endswith(".index") and os.path.exists(path)):
    return "tf", path
  raise IOError("Model not exists: %s" % path)


def load_weights(path, scope="person_pred"):
  """{TF variable name: array} under `scope`, optimizer slots and global_step
  dropped (code/pred_utils.py:166-174, multifuture_inference.py:282-289)."""
  from multiverse_amd import tf_checkpoint
  kind, src = _find_weights(path)
  if kind == "tf":
    return tf_checkpoint.load_checkpoint(src, scope=scope)
  return {k: v for k, v in load_params(src).items()
          if k.split("/")[0] == scope and
          k.split("/")[-1] not in tf_checkpoint.OPTIMIZER_SLOT_NAMES}


def initialize(load, load_best, args, model):
  """code/pred_utils.py:149-205 role: put weights into the model.  `model`
  plays the session's part.  Reads a TensorFlow checkpoint (the reference's own
  `tf.train.Saver` files, via multiverse_amd.tf_checkpoint) or an .npz; only the
  `person_pred` scope is consumed, like the inference script
  (code/multifuture_inference.py:287-289)."""
  if not load:
    raise ValueError("random initialisation lives in multiverse_amd.synth.make_params")
  path = getattr(args, "load_from", None)
  if path is None:
    path = args.save_dir_best if load_best else args.save_dir
  model.load_params(load_weights(path))


OPT_SLOT_NAMES = {        # TF-1.15 slot variable suffixes by --optimizer
    "adadelta": ("Adadelta", "Adadelta_1"), "momentum": ("Momentum",),
    "adam": ("Adam", "Adam_1"), "rmsprop": ("RMSProp", "RMSProp_1")}


class Saver(object):
  """tf.train.Saver(max_to_keep=5) role (code/train.py:170-171, 222, 244):
  `save(model, path, global_step)` writes a TensorFlow-format checkpoint with
  the variables, the optimizer slots under TF's slot names (and Adam's beta powers)
  and global_step, so a run can resume and the reference's own tools can read the
  weights."""

  def __init__(self, max_to_keep=5):
    self.max_to_keep = max_to_keep
    self._last_checkpoints = []    # only what THIS saver wrote is ever deleted

  def save(self, model, path, global_step=None):
    from multiverse_amd import tf_checkpoint
    eng = model.engine
    variables = model.get_params()
    try:
      step = eng.global_step
      slots = OPT_SLOT_NAMES[getattr(model.config, "optimizer", "adadelta")]
      for n in list(variables):
        for i, suffix in enumerate(slots):
          variables[n + "/" + suffix] = eng.get_opt_slot(n, i)
      if slots[0] == "Adam":
        b1p, b2p = eng.opt_scalars()
        variables["beta1_power"] = np.asarray(b1p, dtype="float32")
        variables["beta2_power"] = np.asarray(b2p, dtype="float32")
    except _lib_error():
      step = 0                       # inference-only engine: weights only
    variables["global_step"] = np.asarray(step, dtype="int32")
    return tf_checkpoint.save_checkpoint(path, variables, global_step=global_step,
                                         max_to_keep=self.max_to_keep,
                                         written=self._last_checkpoints)

  def restore(self, model, path, with_optimizer=True):
    """Resume: weights + (when present) optimizer slots and global_step."""
    from multiverse_amd import tf_checkpoint
    allv = tf_checkpoint.load_checkpoint(path, skip_optimizer_slots=False)
    model.load_params({k: v for k, v in allv.items()
                       if k.startswith("person_pred/") and
                       k.split("/")[-1] not in tf_checkpoint.OPTIMIZER_SLOT_NAMES})
    if with_optimizer and "global_step" in allv:
      eng = model.engine
      slots = OPT_SLOT_NAMES[getattr(model.config, "optimizer", "adadelta")]
      for n, _ in eng.param_specs():
        for i, suffix in enumerate(slots):
          if n + "/" + suffix in allv:
            eng.set_opt_slot(n, i, allv[n + "/" + suffix])
      if slots[0] == "Adam" and "beta1_power" in allv:
        eng.set_opt_scalars(float(allv["beta1_power"]), float(allv["beta2_power"]))
      eng.global_step = int(allv["global_step"])


def _lib_error():
  from multiverse_amd import _lib
  return _lib.MvError


# ------------------------------------------------------------------ data

def read_data(args, data_type):
  """code/pred_utils.py:208-300: npz -> Dataset(per-example data, shared)."""
  data_path = os.path.join(args.prepropath, "data_%s.npz" % data_type)
  data = dict(np.load(data_path, allow_pickle=True))
  return dataset_from_npz_dict(data, data_type, args)


def dataset_from_npz_dict(data, data_type, args):
  shares = ["scene_feat", "video_wh", "scene_grid_strides", "vid2name",
            "person_boxkey2id", "person_boxid2key"]
  excludes = ["seq_start_end", "obs_kp_rel", "obs_kp", "cur_activity", "obs_box",
              "future_activity", "pred_kp", "obs_other_box", "person_boxid2key"]
  if "video_wh" in data:
    args.box_img_w, args.box_img_h = [int(v) for v in np.asarray(data["video_wh"])]
  else:
    args.box_img_w, args.box_img_h = 1920, 1080
  for i in range(len(args.scene_grid_strides)):
    shares.append("grid_center_%d" % i)
  shared = {}
  for key in data:
    if key in shares:
      val = np.asarray(data[key]) if not isinstance(data[key], np.ndarray) else data[key]
      shared[key] = val.item() if not val.shape else val
  num_examples = len(data["obs_traj"])
  newdata = {}
  for key in data:
    if key in excludes or key in shares:
      continue
    if len(data[key]) != num_examples:
      print("warning, ignoring %s.." % key)
      continue
    newdata[key] = data[key]
  assert np.asarray(shared["scene_grid_strides"])[0] == args.scene_grid_strides[0]
  if "person_boxid2key" in shared:
    boxid2key = shared["person_boxid2key"]
    newdata["traj_key"] = [boxid2key[newdata["obs_boxid"][i][0]]
                           for i in range(num_examples)]
  return Dataset(newdata, data_type, shared=shared, config=args)


class Dataset(object):
  """Batching with the reference's rules (code/pred_utils.py:586-706)."""

  def __init__(self, data, data_type, config=None, shared=None):
    self.data = data
    self.data_type = data_type
    self.valid_idxs = range(self.get_data_size())
    self.num_examples = len(self.valid_idxs)
    self.shared = shared
    self.config = config

  def get_data_size(self):
    return len(self.data["obs_traj"])

  def get_by_idxs(self, idxs):
    out = collections.defaultdict(list)
    for key, val in self.data.items():
      out[key].extend(val[idx] for idx in idxs)
    return out

  def get_batches(self, batch_size, num_steps=0, shuffle=True, cap=False,
                  full=False):
    """Yields (batch_idxs, Dataset).  Short batches are padded with their last
    item to `batch_size` and carry `original_batch_size`; every batch gets a
    compacted scene table (`batch_scene_feat` float32 [U,SH,SW,SC],
    `batch_obs_scene` int32 [N,T_o,1]).  Shuffling draws ONE permutation that
    every epoch reuses, like the reference (:638-644)."""
    per_epoch = int(math.ceil(self.num_examples / float(batch_size)))
    if full:
      num_steps = per_epoch
    if cap and num_steps > per_epoch:
      num_steps = per_epoch
    num_epochs = int(math.ceil(num_steps / float(per_epoch))) if per_epoch else 0
    order = (random.sample(list(self.valid_idxs), len(self.valid_idxs))
             if shuffle else list(self.valid_idxs))

    def groups():
      for i in range(0, len(order), batch_size):
        yield tuple(order[i:i + batch_size])

    stream = itertools.chain.from_iterable(groups() for _ in range(num_epochs))
    config = self.config
    for _ in range(num_steps):
      batch_idxs = next(stream)
      original = len(batch_idxs)
      if original < batch_size:
        batch_idxs = tuple(list(batch_idxs) + [batch_idxs[-1]] * (batch_size - original))
      batch_data = self.get_by_idxs(batch_idxs)
      batch_data["original_batch_size"] = original
      old2new = {}
      new_obs_scene = np.zeros((config.batch_size, config.obs_len, 1), dtype="int32")
      for i, seq in enumerate(batch_data["obs_scene"]):
        for j in range(len(seq)):
          oldid = int(np.asarray(seq[j]).reshape(-1)[0])
          if oldid not in old2new:
            old2new[oldid] = len(old2new)
          new_obs_scene[i, j, 0] = old2new[oldid]
      scene_feat = np.zeros((len(old2new), config.scene_h, config.scene_w,
                             config.scene_class), dtype="float32")
      for oldid, newid in old2new.items():
        scene_feat[newid] = self.shared["scene_feat"][oldid]
      batch_data["batch_obs_scene"] = new_obs_scene
      batch_data["batch_scene_feat"] = scene_feat
      yield batch_idxs, Dataset(batch_data, self.data_type, shared=self.shared)


# ------------------------------------------------------------------ metrics

def evaluate(dataset, config, sess, tester):
  """code/pred_utils.py:354-586: grid classification accuracy (overall and per
  step), trajectory ADE / FDE from centre + regressed offset, and the
  centre-only variants; same keys, same arithmetic, vectorised per sample."""
  pred_len = config.pred_len
  n_scale = len(config.scene_grids)
  l2 = [[] for _ in range(n_scale)]
  l2_center = [[] for _ in range(n_scale)]
  hits = [[] for _ in range(n_scale)]
  per_scene = bool(getattr(config, "per_scene_eval", False))
  if per_scene:       # code/pred_utils.py:374-378 (the ActEV cameras)
    assert sum(config.use_grids) == 1, "per scene eval is for one grid only"
    scenes = ["0000", "0002", "0400", "0401", "0500"]
    l2_scenes = [[] for _ in scenes]
  out_data = None
  if getattr(config, "save_output", None) is not None:
    out_data = {"obs_list": [], "pred_gt_list": [], "seq_ids": []}
    for i in range(n_scale):
      out_data["grid%s_class" % i] = []
      out_data["grid%s_gt_class" % i] = []
      out_data["grid%s_pred_traj" % i] = []
      out_data["grid_center_%d" % i] = dataset.shared["grid_center_%d" % i]
    if getattr(config, "use_beam_search", False):
      out_data["beam_grid_ids"] = []
      out_data["beam_logprobs"] = []
  # tester.steps == tester.step per batch, with the next batch's feed and the previous
  # batch's fetch overlapped with the current batch's kernels when the tester offers it
  batches = dataset.get_batches(config.batch_size, full=True, shuffle=False)
  stepper = (tester.steps(sess, batches) if hasattr(tester, "steps")
             else ((b, tester.step(sess, b)) for b in batches))
  for evalbatch, (grid_pred_class, grid_pred_reg, beam_outputs) in stepper:
    _, batch = evalbatch
    N = batch.data["original_batch_size"]
    if getattr(config, "use_beam_search", False):
      assert sum(config.use_grids) == 1
      _, beam_grid_ids, beam_logprobs = beam_outputs
    for j, (H, W) in enumerate(config.scene_grids):
      if not config.use_grids[j]:
        continue
      grid_class = np.asarray(grid_pred_class[j])[:N].reshape([N, pred_len, H * W])
      selected = np.argmax(grid_class, axis=2)
      gt_class = np.array([np.asarray(batch.data["pred_grid_class"][i])[j, :]
                           for i in range(N)])
      if getattr(config, "use_gt_grid", False):
        selected = gt_class
      grid_reg = np.asarray(grid_pred_reg[j])[:N].reshape([N, pred_len, H * W, 2])
      centers = np.asarray(batch.shared["grid_center_%s" % j]).reshape([-1, 2])
      for i in range(N):
        hits[j].append(gt_class[i] == selected[i])
        this_center = centers[selected[i]]                       # [T, 2]
        offs = grid_reg[i, np.arange(pred_len), selected[i], :]   # [T, 2]
        traj = this_center + offs
        gt_traj = np.asarray(batch.data["pred_traj"][i])
        l2[j].append(np.sqrt(np.sum((gt_traj - traj) ** 2, axis=1)))
        l2_center[j].append(np.sqrt(np.sum((gt_traj - this_center) ** 2, axis=1)))
        if per_scene:   # :514-517; a camera outside the list raises, as scenes.index does
          l2_scenes[scenes.index(get_scene(batch.data["traj_key"][i]))].append(l2[j][-1])
        if out_data is not None:
          if j == 0 and "traj_key" in batch.data:
            out_data["seq_ids"].append(batch.data["traj_key"][i])
          if j == 0:
            out_data["obs_list"].append(batch.data["obs_traj"][i])
            out_data["pred_gt_list"].append(batch.data["pred_traj"][i])
          out_data["grid%s_pred_traj" % j].append(traj)
          out_data["grid%s_gt_class" % j].append(gt_class[i])
          out_data["grid%s_class" % j].append(grid_class[i])
          if getattr(config, "use_beam_search", False):
            out_data["beam_grid_ids"].append(beam_grid_ids[i])
            out_data["beam_logprobs"].append(beam_logprobs[i])
  p = {}
  for j in range(n_scale):
    if not config.use_grids[j]:
      continue
    h = np.array(hits[j])                      # [M, T] bool
    p["grid%d_acc" % j] = np.mean(h)
    for t in range(pred_len):
      p["grid%d_acc_@T=%d" % (j, t)] = np.mean(h[:, t])
    d = np.array(l2[j])
    p["grid%d_traj_ade" % j] = np.mean(d)
    p["grid%d_traj_fde" % j] = np.mean(d[:, -1])
    dc = np.array(l2_center[j])
    p["grid%d_traj_centerOnly_ade" % j] = np.mean(dc)
    p["grid%d_traj_centerOnly_fde" % j] = np.mean(dc[:, -1])
  if per_scene:         # :569-578
    for scene, diffs in zip(scenes, l2_scenes):
      p["%s_ade" % scene] = np.mean([t for l in diffs for t in l]) if diffs else 0.0
      p["%s_fde" % scene] = np.mean([l[-1] for l in diffs]) if diffs else 0.0
  if out_data is not None:
    with open(config.save_output, "wb") as f:
      pickle.dump(out_data, f)
  return p


def get_scene(videoname_):
  """The scene camera of an ActEV video name, e.g. `VIRAT_S_040003_02_...` -> "0400"
  (code/pred_utils.py:303-307)."""
  return videoname_.split("_S_")[-1].split("_")[0][:4]


def relative_to_abs(rel_traj, start_pos):
  """code/pred_utils.py:735-749."""
  return np.cumsum(rel_traj, axis=0) + np.array([start_pos])


class FIFO_ME(object):
  """Moving average over the last N values (code/pred_utils.py:310-331)."""

  def __init__(self, N):
    self.N = N
    self.lst = []

  def put(self, val):
    if val is None:
      return None
    self.lst.append(val)
    if len(self.lst) > self.N:
      self.lst.pop(0)
    return 1

  def me(self):
    return float(np.mean(self.lst)) if self.lst else None
