# coding=utf-8
"""Run the reference's UNMODIFIED `code/pred_models.py` on the eager TF-1 shim
(oracle/tf1_shim/tensorflow)  --  TEST INFRASTRUCTURE ONLY.

Needs /root/reference (this container only; the GPU box never runs this).
`tests/golden/make_shim_golden.py` uses it to freeze outputs of the reference's
own code into tests/golden/, which then pin oracle/multiverse_oracle.py.
"""

from __future__ import annotations

import importlib
import os
import sys

import numpy as np

SHIM_DIR = os.path.dirname(os.path.abspath(__file__))
REFERENCE_CODE = os.environ.get("MULTIVERSE_REFERENCE", "/root/reference/code")


def available():
  return os.path.exists(os.path.join(REFERENCE_CODE, "pred_models.py"))


def import_reference():
  """(tf shim module, the reference's pred_models module)."""
  if SHIM_DIR not in sys.path:
    sys.path.insert(0, SHIM_DIR)
  tf = importlib.import_module("tensorflow")
  assert "eager-shim" in tf.__version__, "a real tensorflow shadows the shim"
  if "pred_models" in sys.modules and \
      getattr(sys.modules["pred_models"], "__file__", "").startswith(REFERENCE_CODE):
    return tf, sys.modules["pred_models"]
  sys.path.insert(0, REFERENCE_CODE)
  try:
    sys.modules.pop("pred_models", None)
    ref = importlib.import_module("pred_models")
  finally:
    sys.path.remove(REFERENCE_CODE)
  assert os.path.abspath(ref.__file__).startswith(os.path.abspath(REFERENCE_CODE))
  return tf, ref


class Batch(object):
  """What `Dataset.get_batches` yields as batch[1] (code/pred_utils.py:672-706):
  an object whose .data dict has the per-example lists and the per-batch
  compacted scene table."""

  def __init__(self, data):
    self.data = data


def batch_from_feed(cfg, feed):
  """Engine-style feed (multiverse_amd.synth.make_feed) -> the reference's batch
  dict, so that the reference's own get_feed_dict rebuilds the placeholders."""
  N = cfg.batch_size
  nscale = len(cfg.scene_grids)
  T_o, T_p = cfg.obs_len, cfg.pred_len
  data = {"obs_grid_class": [], "pred_grid_class": [],
          "batch_scene_feat": np.asarray(feed["scene_feat"], dtype="float32"),
          "batch_obs_scene": [[[int(feed["obs_scene"][i, t])] for t in range(T_o)]
                              for i in range(N)]}
  for i in range(N):
    data["obs_grid_class"].append(np.stack(
        [np.asarray(feed["grid_obs_labels"][s][i]) for s in range(nscale)]))
    data["pred_grid_class"].append(np.stack(
        [np.asarray(feed["grid_pred_labels"][s][i]) if feed["grid_pred_labels"][s]
         is not None else np.zeros(T_p, "int32") for s in range(nscale)]))
  for s, (h, w) in enumerate(cfg.scene_grids):
    if not cfg.use_grids[s]:
      continue
    data["obs_grid_target_all_%d" % s] = [feed["grid_obs_regress"][s][i] for i in range(N)]
    data["pred_grid_target_all_%d" % s] = [feed["grid_pred_regress"][s][i] for i in range(N)]
  return Batch(data)


def _activation(tf, cfg):
  # process_args maps the flag to a TF function (code/pred_utils.py:112-121)
  if callable(cfg.activation_func):
    return cfg.activation_func
  return {"tanh": tf.nn.tanh, "relu": tf.nn.relu, "lrelu": tf.nn.leaky_relu}[cfg.activation_func]


def build_model(cfg, params, feed, is_train=False, opt_slots=None, global_step=0,
                dropout_seed=0):
  """Instantiate the reference's Model eagerly on `feed`:
  1. Model.__init__ with build_forward / build_loss stubbed -> placeholders only;
  2. the reference's own get_feed_dict(batch, is_train) -> values, bound;
  3. the reference's own (unmodified) build_forward / build_loss, executed."""
  import copy
  tf, ref = import_reference()
  cfg = copy.copy(cfg)
  cfg.activation_func = _activation(tf, cfg)
  cfg.is_train = is_train
  p = dict(params)
  p["global_step"] = np.asarray(global_step, dtype="int32")
  tf.reset_default_graph(params=p, strict=True, opt_slots=opt_slots,
                         dropout_seed=dropout_seed)
  real_fwd, real_loss = ref.Model.build_forward, ref.Model.build_loss
  ref.Model.build_forward = lambda self: None
  ref.Model.build_loss = lambda self: None
  try:
    model = ref.get_model(cfg, 0)
  finally:
    ref.Model.build_forward, ref.Model.build_loss = real_fwd, real_loss
  batch = batch_from_feed(cfg, feed)
  fd = model.get_feed_dict(batch, is_train=is_train)
  T_pred = int(feed.get("pred_length", cfg.pred_len))
  if T_pred != cfg.pred_len:   # multifuture_inference.py:311: run-time T_pred
    fd[model.pred_length] = np.full([cfg.batch_size], T_pred, dtype="int32")
  tf.bind_feed(fd)
  model.build_forward()
  if is_train:
    model.build_loss()
  return tf, ref, model, cfg, batch


def variable_names():
  tf, _ = import_reference()
  return [(v._name, tuple(v.value.shape)) for v in tf.global_variables()]


def forward(cfg, params, feed):
  """Tester.step of the reference on the shim -> (cls list, reg list, beam)."""
  tf, ref, model, rcfg, batch = build_model(cfg, params, feed, is_train=False)
  tester = ref.Tester(model, rcfg, sess=None)
  return tester.step(tf.Session(), (None, batch))


def train_step(cfg, params, feed, opt_slots, global_step):
  """Trainer.step of the reference on the shim.
  -> (loss, wd_loss, pred_grid_loss, grads {name: array}, new params, slots)
  feed["dropout_seed"] seeds the shared dropout mask generator (keep_prob < 1)."""
  tf, ref, model, rcfg, batch = build_model(cfg, params, feed, is_train=True,
                                            opt_slots=opt_slots,
                                            global_step=global_step,
                                            dropout_seed=int(feed.get("dropout_seed", 0)))
  trainer = ref.Trainer(model, rcfg)
  var = tf.trainable_variables()
  grads = {v._name: (None if g is None else g.numpy().copy())
           for v, g in zip(var, tf.gradients(model.loss, var))}
  loss, _, wd_loss, pgl = trainer.step(tf.Session(), (None, batch))
  new_params = {v._name: v.numpy().copy() for v in tf.global_variables()}
  gs = int(new_params.pop("global_step"))
  slots = tf.shim_state().opt_slots
  return (float(loss), float(wd_loss), [float(x) for x in pgl], grads, new_params,
          slots, gs)
