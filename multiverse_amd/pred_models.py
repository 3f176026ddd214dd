# coding=utf-8
"""Drop-in for the reference's `pred_models` call boundary, backed by
libmultiverse_hip.so (no TensorFlow, no CPU fallback).

What the reference's callers use and where it lives there:

  get_model(config, gpuid)            code/pred_models.py:19-30
  Model(config, scope)                code/pred_models.py:32-121
    .get_feed_dict(batch, is_train)   code/pred_models.py:1042-1194
    .grid_pred_decoded / .grid_pred_reg_decoded / .beam_outputs (fetch names)
  Tester(model, config, sess).step    code/pred_models.py:1745-1790
  Trainer(model, config).step         code/pred_models.py:1636-1742

`sess` arguments are accepted and ignored: the engine handle plays the role of
the TF session.  A `Model` owns one engine (one HIP stream, weights resident in
HBM); `Tester.step` is one `mv_forward_greedy` / `mv_forward_beam` call, i.e.
exactly one `sess.run` of the reference.
"""

from __future__ import annotations

import numpy as np

from multiverse_amd import _lib


def get_model(config, gpuid=0):
  """code/pred_models.py:19-30 -- build the model and pin it to one GPU."""
  return Model(config, "%s" % getattr(config, "modelname", "model"), gpuid=gpuid)


def compact_inputs_enabled(cfg):
  """Device-side batch assembly: config.compact_inputs or
  MV_COMPACT_INPUTS=1."""
  import os
  return bool(getattr(cfg, "compact_inputs", False)) or \
      os.environ.get("MV_COMPACT_INPUTS", "0") not in ("", "0")


_compact_warned = set()


def compact_inputs_consistent(cfg, batch):
  """Is the compact hand-over bit-identical to the dense feed for THIS batch?

  The engine rebuilds `*_grid_target_all_<s>` as float32(float64(xy) - centre).  That
  holds when the npz was written from `obs_traj` in pixel coordinates
  (code/preprocess.py:463-475), NOT when preprocess ran with `--traj_pixel_lst`
  (:428-475: the maps come from the alternate pixel trajectory while `obs_traj` stays
  in world coordinates), and the uint8 hand-over of the scene table needs 0/1 masks
  (:831).  Checked per batch on its first and last row wherever the dense maps are
  present; a mismatch falls back to the dense feed (once-per-reason warning).
  -> (ok, reason)"""
  data = batch.data
  sf = np.asarray(data["batch_scene_feat"])
  if sf.dtype != np.uint8 and not ((sf == 0) | (sf == 1)).all():
    return False, "scene_feat is not a 0/1 mask"
  n_have = len(data["obs_grid_class"])
  if not n_have:
    return True, ""
  with_targets = getattr(cfg, "is_train", False) or getattr(cfg, "use_gt_grid", False)
  for j in range(len(cfg.scene_grids)):
    if not cfg.use_grids[j]:
      continue
    centre = np.asarray(batch.shared["grid_center_%d" % j], dtype=np.float64)
    for key, traj in (("obs_grid_target_all_%d" % j, "obs_traj"),
                      ("pred_grid_target_all_%d" % j, "pred_traj")):
      if key not in data or traj not in data:
        continue
      if traj == "pred_traj" and not with_targets:
        continue
      for i in sorted({0, n_have - 1}):
        xy = np.asarray(data[traj][i], dtype=np.float64)                 # [T, 2]
        want = (xy[:, None, None, :] - centre[None]).astype(np.float32)  # [T, H, W, 2]
        have = np.asarray(data[key][i], dtype=np.float32)
        if have.shape != want.shape or not (have == want).all():
          return False, ("%s is not float32(%s - grid_center_%d) (an npz written with "
                         "--traj_pixel_lst?)" % (key, traj, j))
  return True, ""


def build_compact_feed_dict(cfg, batch, is_train=False):
  """The same batch as `build_feed_dict`, handed over as labels + one (x, y) per
  step + the uint8 scene masks; the engine derives the dense regression maps in HBM
  (`Engine.upload_compact`), bit-identical to `data["*_grid_target_all_<s>"]`
  because those are float32(obs_traj - grid_center) by construction
  (code/preprocess.py:463-475).  No O(N T H W) host loops, 16 B per step over PCIe."""
  N, T_in, T_pred = cfg.batch_size, cfg.obs_len, cfg.pred_len
  data = batch.data
  n_have = len(data["obs_grid_class"])
  with_targets = is_train or getattr(cfg, "use_gt_grid", False)
  feed = {"is_train": is_train, "pred_length": T_pred, "num_rows": n_have,
          "grid_obs_labels": [], "grid_pred_labels": [],
          "grid_centers": [batch.shared["grid_center_%d" % j]
                           for j in range(len(cfg.scene_grids))]}
  for j in range(len(cfg.scene_grids)):
    labels = np.zeros([N, T_in], dtype="int32")
    plab = np.zeros([N, T_pred], dtype="int32")
    if n_have:
      labels[:n_have] = np.stack(
          [np.asarray(data["obs_grid_class"][i])[j, :] for i in range(n_have)])
      if with_targets:
        plab[:n_have] = np.stack(
            [np.asarray(data["pred_grid_class"][i])[j, :] for i in range(n_have)])
    feed["grid_obs_labels"].append(labels)
    feed["grid_pred_labels"].append(plab if with_targets and cfg.use_grids[j] else None)
  xy = np.zeros([N, T_in, 2], dtype="float64")
  if n_have:
    xy[:n_have] = np.asarray(data["obs_traj"], dtype="float64")[:n_have]
  feed["obs_xy"] = xy
  if with_targets:
    pxy = np.zeros([N, T_pred, 2], dtype="float64")
    if n_have:
      pxy[:n_have] = np.asarray(data["pred_traj"], dtype="float64")[:n_have]
    feed["pred_xy"] = pxy
  obs_scene = np.zeros((N, T_in), dtype="int32")
  bos = data["batch_obs_scene"]
  for i in range(len(bos)):
    row = np.asarray(bos[i]).reshape(-1)[:T_in]
    obs_scene[i, :len(row)] = row
  feed["obs_scene"] = obs_scene
  feed["scene_feat"] = np.asarray(data["batch_scene_feat"]).astype("uint8", copy=False)
  feed["compact"] = True
  return feed


def build_feed_dict(cfg, batch, is_train=False):
  """numpy batch -> engine inputs; same contents as the reference feed_dict
  (code/pred_models.py:1042-1194): rows beyond len(data) stay zero, GT future
  only when training / use_gt_grid.  Pure host code (no engine needed)."""
  N, T_in, T_pred = cfg.batch_size, cfg.obs_len, cfg.pred_len
  data = batch.data
  n_have = len(data["obs_grid_class"])
  feed = {"is_train": is_train, "pred_length": T_pred,
          "grid_obs_labels": [], "grid_obs_regress": [],
          "grid_pred_labels": [], "grid_pred_regress": []}
  for j, (h, w) in enumerate(cfg.scene_grids):
    labels = np.zeros([N, T_in], dtype="int32")
    if n_have:
      labels[:n_have] = np.stack(
          [np.asarray(data["obs_grid_class"][i])[j, :] for i in range(n_have)])
    feed["grid_obs_labels"].append(labels)
    if not cfg.use_grids[j]:
      feed["grid_obs_regress"].append(None)
      feed["grid_pred_labels"].append(None)
      feed["grid_pred_regress"].append(None)
      continue
    reg = np.zeros([N, T_in, h, w, 2], dtype="float32")
    for i in range(n_have):
      reg[i] = data["obs_grid_target_all_%d" % j][i]
    feed["grid_obs_regress"].append(reg)
    if is_train or getattr(cfg, "use_gt_grid", False):
      plab = np.zeros([N, T_pred], dtype="int32")
      preg = np.zeros([N, T_pred, h, w, 2], dtype="float32")
      for i in range(n_have):
        plab[i] = np.asarray(data["pred_grid_class"][i])[j, :]
        preg[i] = data["pred_grid_target_all_%d" % j][i]
      feed["grid_pred_labels"].append(plab)
      feed["grid_pred_regress"].append(preg)
    else:
      feed["grid_pred_labels"].append(None)
      feed["grid_pred_regress"].append(None)
  obs_scene = np.zeros((N, T_in), dtype="int32")
  bos = data["batch_obs_scene"]
  for i in range(len(bos)):
    row = np.asarray(bos[i]).reshape(-1)[:T_in]
    obs_scene[i, :len(row)] = row
  feed["obs_scene"] = obs_scene
  feed["scene_feat"] = np.asarray(data["batch_scene_feat"], dtype="float32")
  return feed


def f16x3_out_of_range(params, limit=60000.0 / 256.0):
  """The bound engine_setup.h ensure_packed16 enforces, on the host copy: max |w| of every
  3 x 3 gate kernel and of its Winograd-transformed rows ((g0 +- g1 + g2) / 2, (g0 + 2 g1 +
  4 g2) / 6 and its mirror, column by column) must stay below 60 000 / 256.  Returns None, or
  (name, max |w|, reach) of the first kernel that does not."""
  for name in sorted(params):
    w = np.asarray(params[name])
    if not name.endswith("/kernel") or w.ndim != 4 or w.shape[0] != 3 or w.shape[1] != 3:
      continue
    g0, g1, g2 = (w[i].astype(np.float32) for i in range(3))
    mx = float(np.abs(w).max())
    reach = max(mx,
                float(np.abs(g0 + g1 + g2).max()) * 0.5, float(np.abs(g0 - g1 + g2).max()) * 0.5,
                float(np.abs(g0 + 2.0 * g1 + 4.0 * g2).max()) / 6.0,
                float(np.abs(4.0 * g0 + 2.0 * g1 + g2).max()) / 6.0)
    if not reach < limit:
      return name, mx, reach
  return None


class Model(object):
  """One engine instance with the reference Model's host-side protocol."""

  def __init__(self, config, scope, gpuid=0):
    self.scope = scope
    self.config = config
    self.N = config.batch_size
    self.beam_size = getattr(config, "beam_size", 1)
    self._check_config(config)
    self.engine = _lib.Engine(config, device=gpuid)
    # gate-convolution arithmetic of the inference forward: "f16x3" (default; fp32
    # operands as two pre-scaled fp16 planes, three fp16 MFMAs per product, fp32
    # accumulate -- fp32-class error, same parity bars) or "f32" (fp32 MFMA);
    # config.compute_mode or the MV_COMPUTE environment variable override it
    import os
    self.compute_mode = getattr(config, "compute_mode", None) or \
        os.environ.get("MV_COMPUTE", "f16x3")
    # (relu / lrelu models run in f16x3 too: their unbounded x operands carry a per-tensor
    # exponent, csrc/convlstm_f16x3.h ConvLstm16Args::x_exp)
    if getattr(config, "convlstm_kernel", 3) != 3 and self.compute_mode != "f32":
      # --convlstm_kernel other than 3 (code/train.py:70): the matrix-pipe gate kernels are
      # 3 x 3 stencils; such models run the generic fp32 loops (csrc/convlstm_generic.h)
      import logging
      logging.getLogger("multiverse_amd").warning(
          "compute mode %s overridden to f32: convlstm_kernel %d runs the generic fp32 path",
          self.compute_mode, config.convlstm_kernel)
      self.compute_mode = "f32"
    if any(use and h * w < 32 for (h, w), use in zip(config.scene_grids, config.use_grids)):
      # the fp16-pipe kernels' epilogue lets a 32-cell wave tile span at most two images
      # (engine_forward.h run_conv_group_f16x3 refuses smaller grids); such toy grids run on the
      # fp32 matrix pipe, whose kernel wraps over any number of images
      if self.compute_mode != "f32":
        import logging
        logging.getLogger("multiverse_amd").warning(
            "compute mode %s overridden to f32: a used grid has fewer than 32 cells",
            self.compute_mode)
      self.compute_mode = "f32"
    self.engine.set_compute_mode(self.compute_mode)
    self.global_step = 0
    # names of the fetches, kept for callers that introspect them
    self.grid_pred_decoded = ["grid_pred_decoded_%d" % i
                              for i in range(len(config.scene_grids))]
    self.grid_pred_reg_decoded = ["grid_pred_reg_decoded_%d" % i
                                  for i in range(len(config.scene_grids))]
    self.beam_outputs = (["beam_logits", "beam_ids", "beam_logprobs"]
                         if getattr(config, "use_beam_search", False) else None)
    self.loss = None

  @staticmethod
  def _check_config(config):
    """Same unsupported-combination asserts as the reference graph builder
    (code/pred_models.py:261-262) plus the engine's own scope limits."""
    _lib.activation_code(getattr(config, "activation_func", "tanh"))   # tanh / relu / lrelu
    if not getattr(config, "use_scene_enc", True):
      raise _lib.MvError("only the published --use_scene_enc wiring is built")
    if getattr(config, "use_beam_search", False):
      assert not getattr(config, "is_train", False)
      assert sum(config.use_grids) == 1, "only one scale test at a time"

  # -- weights (tf.train.Saver role) --------------------------------------
  def param_specs(self):
    return self.engine.param_specs()

  def load_params(self, params):
    self.engine.set_params(params)
    if self.compute_mode == "f16x3":
      bad = f16x3_out_of_range(params)
      if bad is not None:
        # the f16x3 planes hold 256 w (and, in the Winograd packs, 256 x the transformed
        # kernel rows) in fp16: engine_setup.h ensure_packed16 refuses a model that leaves that
        # range.  Such a model decodes on the fp32 matrix pipe instead (same results
        # contract, 1/5 of the rate) -- the toy-grid pattern above.
        import logging
        logging.getLogger("multiverse_amd").warning(
            "compute mode f16x3 overridden to f32: |%s| reaches %.4g (%.4g in the Winograd "
            "kernel planes), outside the scaled fp16 range", bad[0], bad[1], bad[2])
        self.compute_mode = "f32"
        self.engine.set_compute_mode("f32")

  def get_params(self):
    return {n: self.engine.get_param(n) for n, _ in self.engine.param_specs()}

  def close(self):
    self.engine.close()

  # -- feed ----------------------------------------------------------------
  def get_feed_dict(self, batch, is_train=False):
    if compact_inputs_enabled(self.config) and "obs_traj" in batch.data and \
        all(("grid_center_%d" % j) in batch.shared
            for j in range(len(self.config.scene_grids))):
      ok, why = compact_inputs_consistent(self.config, batch)
      if ok:
        return build_compact_feed_dict(self.config, batch, is_train=is_train)
      if why not in _compact_warned:
        _compact_warned.add(why)
        import sys
        sys.stderr.write("compact_inputs: falling back to the dense feed: %s\n" % why)
    return build_feed_dict(self.config, batch, is_train=is_train)

  # -- one sess.run ----------------------------------------------------------
  def run_forward(self, feed):
    """-> (grid_pred_class list_s, grid_pred_reg list_s, beam_outputs)."""
    cfg = self.config
    compact = bool(feed.get("compact", False))
    if getattr(cfg, "use_beam_search", False):
      arrs, s = (self.engine.forward_beam_compact(feed) if compact
                 else self.engine.forward_beam(feed))
      cls = [[] for _ in cfg.scene_grids]
      reg = [[] for _ in cfg.scene_grids]
      cls[s] = arrs["best_beam"]
      reg[s] = arrs["grid_reg"]
      return cls, reg, [arrs["logits"], arrs["ids"], arrs["logprobs"]]
    cls, reg = (self.engine.forward_greedy_compact(feed) if compact
                else self.engine.forward_greedy(feed))
    return cls, reg, None


class Tester(object):
  """code/pred_models.py:1745-1790."""

  def __init__(self, model, config, sess=None):
    self.config = config
    self.model = model
    self.sess = sess
    self.grid_pred_decoded = model.grid_pred_decoded
    self.grid_pred_reg_decoded = model.grid_pred_reg_decoded
    self.beam_outputs = model.beam_outputs

  def step(self, sess, batch):
    """One inferencing step: (grid_pred_class, grid_pred_reg, beam_outputs)
    with grid_pred_class[s] [N,T_p,H,W,1], grid_pred_reg[s] [N,T_p,H,W,2],
    `[]` for unused scales, beam_outputs None or [logits, ids, logprobs]."""
    _, batch_data = batch
    feed = self.model.get_feed_dict(batch_data, is_train=False)
    return self.model.run_forward(feed)

  def steps(self, sess, batches, depth=2):
    """Yield (batch, step(sess, batch)) for every batch of an iterable -- what the loop of
    `evaluate` (code/pred_utils.py:415) computes, with the feed of batch k+1 and the fetch
    of batch k-1 overlapped with the kernels of batch k (mv_submit_greedy /
    mv_collect_greedy; greedy decode with the dense feed, else plain `step`).  Results are
    bitwise those of `step`."""
    cfg = self.config
    if getattr(cfg, "use_beam_search", False) or getattr(cfg, "no_pipeline", False):
      for batch in batches:
        yield batch, self.step(sess, batch)
      return
    pending = []
    eng = self.model.engine
    try:
      for batch in batches:
        feed = self.model.get_feed_dict(batch[1], is_train=False)
        if feed.get("compact", False):          # compact feeds take the blocking path
          while pending:
            b0 = pending.pop(0)
            cls, reg = eng.collect_greedy()
            yield b0, (cls, reg, None)
          yield batch, self.model.run_forward(feed)
          continue
        if len(pending) >= depth:
          b0 = pending.pop(0)
          cls, reg = eng.collect_greedy()
          yield b0, (cls, reg, None)
        eng.submit_greedy(feed, depth)
        pending.append(batch)
      while pending:
        b0 = pending.pop(0)
        cls, reg = eng.collect_greedy()
        yield b0, (cls, reg, None)
    finally:
      # a consumer that stops early (break, an exception in its loop body) must not leave
      # submissions in the engine's pipeline: the next caller would collect THEIR outputs
      # Best effort: when the loop ended on an ENGINE error the same call may fail again --
      # that secondary error must not mask the original one, and the drain stops there.
      while pending:
        pending.pop(0)
        try:
          eng.collect_greedy()
        except Exception as drain_err:  # pylint: disable=broad-except
          import logging
          logging.getLogger("multiverse_amd").warning(
              "Tester.steps: pipeline drain stopped on %r (%d submissions abandoned)",
              drain_err, len(pending))
          pending.clear()
          break


class Trainer(object):
  """code/pred_models.py:1636-1742: learning-rate schedule, Adadelta,
  tf.gradients, element-wise clip, apply_gradients + global_step.  All of it
  runs inside the engine (`mv_train_step`); `step` is one `sess.run([loss,
  train_op, wd_loss, pred_grid_loss])`.

  When torch.distributed is initialised with more than one rank, every rank
  holds one batch shard and the step becomes forward+backward ->
  all-reduce(sum) of the flat gradient buffer (RCCL over xGMI) ->
  `mv_train_apply(1/world)`: the clip and the optimizer see the global-batch
  mean gradient, as a single-GPU step over the global batch would."""

  def __init__(self, model, config):
    self.config = config
    self.model = model
    if not getattr(config, "is_train", False):
      raise _lib.MvError("Trainer needs a config with is_train=True")
    from multiverse_amd import parallel
    model.engine.train_init(config, world=parallel.world_size())
    # data parallel over RCCL: the all-reduce runs inside the library
    self.lib_allreduce = parallel.init_engine_comm(model.engine)

  @property
  def global_step(self):
    return self.model.engine.global_step

  def step(self, sess, batch):
    """One training step -> (loss, train_op(None), wd_loss, pred_grid_loss)."""
    from multiverse_amd import parallel
    _, batch_data = batch
    feed = self.model.get_feed_dict(batch_data, is_train=True)
    eng = self.model.engine
    world = parallel.world_size()
    if self.config.keep_prob < 1.0:
      # DropoutWrapper masks: TF draws them unseeded; here one seed per step from
      # config.dropout_seed (default 0) and the step count -- every rank of a
      # data-parallel job draws different masks for its own rows
      base = int(getattr(self.config, "dropout_seed", 0))
      rank = parallel.rank()
      eng.set_dropout_seed((base * 1000003 + eng.global_step * 8191 + rank * 131071)
                           & 0xFFFFFFFF)
    if feed.get("compact", False):      # batch assembled in HBM, then the resident calls
      eng.upload_compact(feed)
      feed = None
    if world == 1:
      loss, wd_loss, pred_grid_loss = eng.train_step(feed)
    elif self.lib_allreduce:     # buckets reduced on a side stream during the backward pass
      loss, wd_loss, pred_grid_loss = eng.train_step(feed)
      loss, pred_grid_loss = parallel.mean_over_ranks(loss, pred_grid_loss)
    else:
      loss, wd_loss, pred_grid_loss = eng.train_forward_backward(feed)
      parallel.allreduce_engine_grads(eng)
      eng.train_apply(1.0 / world)
      loss, pred_grid_loss = parallel.mean_over_ranks(loss, pred_grid_loss)
    self.model.global_step = eng.global_step
    return loss, None, wd_loss, pred_grid_loss
