# coding=utf-8
"""ctypes binding of libmultiverse_hip.so (C ABI: include/multiverse_hip.h).

There is NO CPU fallback: if the HIP library is missing or a call fails this
module raises.  The library is built in-tree by `__graft_entry__.build()`
(`hipcc --offload-arch=gfx950`), so it travels with the source snapshot.
"""

from __future__ import annotations

import ctypes as C
import os

import numpy as np

MV_MAX_SCALES = 2
MV_ABI_VERSION = 5

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmultiverse_hip.so")

# every symbol include/multiverse_hip.h declares
EXPORTED_SYMBOLS = [
    "mv_create", "mv_destroy", "mv_last_error", "mv_abi_version",
    "mv_set_param", "mv_get_param", "mv_num_params", "mv_param_info",
    "mv_forward_greedy", "mv_forward_beam",
    "mv_upload_inputs", "mv_run_greedy_resident", "mv_run_beam_resident",
    "mv_synchronize", "mv_download_outputs", "mv_download_beam_outputs",
    "mv_set_graph_mode", "mv_set_compute_mode", "mv_set_profiling", "mv_reset_kernel_stats", "mv_num_kernel_stats",
    "mv_kernel_stat", "mv_kernel_stat_dense_flops", "mv_kernel_stat_mfma_flops",
    "mv_time_greedy_resident",
    "mv_time_beam_resident",
    "mv_op_convlstm_step", "mv_op_convlstm_step16", "mv_op_gnn", "mv_op_hidden2grid", "mv_op_beam_step",
    "mv_train_init", "mv_train_step", "mv_train_forward_backward",
    "mv_upload_targets", "mv_grad_buffer", "mv_train_apply", "mv_get_grad", "mv_get_global_step",
    "mv_set_global_step", "mv_get_opt_slot", "mv_set_opt_slot",
    "mv_set_dropout_seed", "mv_get_opt_scalars", "mv_set_opt_scalars",
    "mv_comm_unique_id", "mv_allreduce_init", "mv_allreduce_info",
    "mv_attack_begin", "mv_attack_end", "mv_set_scene_feat", "mv_get_scene_feat",
    "mv_get_scene_grad", "mv_attack_step", "mv_scene_mix", "mv_get_sample_losses",
    "mv_set_label_mixup", "mv_clear_label_mixup",
    "mv_op_convlstm_bwd", "mv_op_gnn_bwd",
    "mv_set_grid_centers", "mv_upload_inputs_compact", "mv_upload_targets_compact",
    "mv_pipeline_create", "mv_submit_greedy", "mv_collect_greedy",
]


class MvError(RuntimeError):
  pass


class mv_config(C.Structure):
  _fields_ = [
      ("abi_version", C.c_int32),
      ("batch_size", C.c_int32),
      ("obs_len", C.c_int32),
      ("max_pred_len", C.c_int32),
      ("scene_h", C.c_int32), ("scene_w", C.c_int32), ("scene_class", C.c_int32),
      ("scene_conv_dim", C.c_int32), ("scene_conv_kernel", C.c_int32),
      ("emb_size", C.c_int32),
      ("hidden_size", C.c_int32),
      ("convlstm_kernel", C.c_int32),
      ("num_scales", C.c_int32),
      ("grid_h", C.c_int32 * MV_MAX_SCALES),
      ("grid_w", C.c_int32 * MV_MAX_SCALES),
      ("use_grid", C.c_int32 * MV_MAX_SCALES),
      ("use_gnn", C.c_int32),
      ("beam_size", C.c_int32),
      ("diverse_beam", C.c_int32),
      ("diverse_gamma", C.c_float),
      ("fix_num_timestep", C.c_int32),
      ("class_feedback_dense", C.c_int32),
      ("use_single_decoder", C.c_int32),
      ("simaug_graph", C.c_int32),
      ("activation", C.c_int32),
  ]


_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int32)


class mv_inputs(C.Structure):
  _fields_ = [
      ("obs_scene", _ip),
      ("scene_feat", _fp),
      ("num_scene_frames", C.c_int32),
      ("pred_len", C.c_int32),
      ("grid_obs_labels", _ip * MV_MAX_SCALES),
      ("grid_obs_regress", _fp * MV_MAX_SCALES),
  ]


_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)


class mv_inputs_compact(C.Structure):
  _fields_ = [
      ("obs_scene", _ip),
      ("scene_feat_u8", _u8p),
      ("num_scene_frames", C.c_int32),
      ("pred_len", C.c_int32),
      ("grid_obs_labels", _ip * MV_MAX_SCALES),
      ("obs_xy", _dp),
      ("num_rows", C.c_int32),
  ]


class mv_targets_compact(C.Structure):
  _fields_ = [
      ("grid_pred_labels", _ip * MV_MAX_SCALES),
      ("pred_xy", _dp),
      ("num_rows", C.c_int32),
  ]


class mv_outputs(C.Structure):
  _fields_ = [
      ("grid_pred_class", _fp * MV_MAX_SCALES),
      ("grid_pred_reg", _fp * MV_MAX_SCALES),
  ]


class mv_beam_outputs(C.Structure):
  _fields_ = [
      ("best_beam", _fp), ("grid_reg", _fp), ("logits", _fp), ("ids", _ip),
      ("logprobs", _fp),
  ]


class mv_train_config(C.Structure):
  _fields_ = [
      ("optimizer", C.c_int32),
      ("init_lr", C.c_float), ("emb_lr", C.c_float),
      ("use_cosine_lr", C.c_int32),
      ("has_decay", C.c_int32), ("learning_rate_decay", C.c_float),
      ("decay_steps", C.c_int32), ("max_steps", C.c_int32),
      ("do_clip", C.c_int32), ("clip_gradient_norm", C.c_float),
      ("wd", C.c_float),
      ("grid_loss_weight", C.c_float), ("grid_reg_loss_weight", C.c_float),
      ("class_feedback", C.c_int32), ("reg_teacher_forcing", C.c_int32),
      ("use_soft_grid_class", C.c_int32), ("soft_kernel_size", C.c_int32),
      ("soft_kernel", C.c_float * 25),
      ("mask_grid_regression", C.c_int32), ("keep_prob", C.c_float),
  ]


class mv_targets(C.Structure):
  _fields_ = [
      ("grid_pred_labels", _ip * MV_MAX_SCALES),
      ("grid_pred_regress", _fp * MV_MAX_SCALES),
  ]


class mv_losses(C.Structure):
  _fields_ = [
      ("loss", C.c_float), ("wd_loss", C.c_float),
      ("pred_grid_loss", C.c_float * (2 * MV_MAX_SCALES)),
      ("num_pred_grid_loss", C.c_int32),
  ]


_lib = None


def load():
  """dlopen the in-tree library; raises MvError if it has not been built."""
  global _lib
  if _lib is not None:
    return _lib
  path = os.environ.get("MV_LIB_PATH", LIB_PATH)   # A/B builds of the same ABI (tuning)
  if not os.path.exists(path):
    raise MvError(
        "%s not found: the HIP extension has not been built "
        "(run `python -c 'import __graft_entry__ as g; g.build()'`). "
        "There is no CPU fallback." % LIB_PATH)
  lib = C.CDLL(path)
  h = C.c_void_p
  lib.mv_last_error.restype = C.c_char_p
  lib.mv_last_error.argtypes = [h]
  lib.mv_abi_version.argtypes = []
  lib.mv_create.argtypes = [C.POINTER(mv_config), C.c_int, C.POINTER(h)]
  lib.mv_destroy.argtypes = [h]
  lib.mv_set_param.argtypes = [h, C.c_char_p, _fp, C.POINTER(C.c_int64), C.c_int32]
  lib.mv_get_param.argtypes = [h, C.c_char_p, _fp, C.c_int64]
  lib.mv_num_params.argtypes = [h]
  lib.mv_param_info.argtypes = [h, C.c_int32, C.c_char_p, C.c_int32,
                                C.POINTER(C.c_int64)]
  lib.mv_forward_greedy.argtypes = [h, C.POINTER(mv_inputs), C.POINTER(mv_outputs)]
  lib.mv_forward_beam.argtypes = [h, C.POINTER(mv_inputs),
                                  C.POINTER(mv_beam_outputs)]
  lib.mv_upload_inputs.argtypes = [h, C.POINTER(mv_inputs)]
  lib.mv_pipeline_create.argtypes = [h, C.c_int32]
  lib.mv_submit_greedy.argtypes = [h, C.POINTER(mv_inputs)]
  lib.mv_collect_greedy.argtypes = [h, C.POINTER(mv_outputs), C.POINTER(C.c_int32)]
  lib.mv_set_grid_centers.argtypes = [h, C.c_int32, _dp]
  lib.mv_upload_inputs_compact.argtypes = [h, C.POINTER(mv_inputs_compact)]
  lib.mv_upload_targets_compact.argtypes = [h, C.POINTER(mv_targets_compact)]
  lib.mv_run_greedy_resident.argtypes = [h]
  lib.mv_run_beam_resident.argtypes = [h]
  lib.mv_synchronize.argtypes = [h]
  lib.mv_download_outputs.argtypes = [h, C.POINTER(mv_outputs)]
  lib.mv_download_beam_outputs.argtypes = [h, C.POINTER(mv_beam_outputs)]
  lib.mv_set_profiling.argtypes = [h, C.c_int32]
  lib.mv_set_graph_mode.argtypes = [h, C.c_int32]
  lib.mv_set_compute_mode.argtypes = [h, C.c_int32]
  lib.mv_reset_kernel_stats.argtypes = [h]
  lib.mv_num_kernel_stats.argtypes = [h]
  lib.mv_kernel_stat.argtypes = [h, C.c_int32, C.c_char_p, C.c_int32,
                                 C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                 C.POINTER(C.c_double), C.POINTER(C.c_double)]
  lib.mv_kernel_stat_dense_flops.argtypes = [h, C.c_int32, C.POINTER(C.c_double)]
  lib.mv_kernel_stat_mfma_flops.argtypes = [h, C.c_int32, C.POINTER(C.c_double)]
  lib.mv_time_greedy_resident.argtypes = [h, C.c_int32, _fp]
  lib.mv_time_beam_resident.argtypes = [h, C.c_int32, _fp]
  lib.mv_op_convlstm_step.argtypes = [C.c_int, _fp, _fp, _fp, _fp, _fp] + \
      [C.c_int32] * 5 + [_fp, _fp]
  lib.mv_op_convlstm_step16.argtypes = [C.c_int, C.c_int32, _fp, _fp, _fp, _fp, _fp] + \
      [C.c_int32] * 5 + [_fp, _fp, _fp]
  lib.mv_op_gnn.argtypes = [C.c_int, _fp, _fp] + [C.c_int32] * 5 + [_fp]
  lib.mv_op_hidden2grid.argtypes = [C.c_int, _fp, _fp] + [C.c_int32] * 5 + [_fp]
  lib.mv_op_beam_step.argtypes = [C.c_int, _fp, _fp, C.c_int32, C.c_int32,
                                  C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                  C.c_int32, _fp, _ip, _ip]
  lib.mv_train_init.argtypes = [h, C.POINTER(mv_train_config)]
  lib.mv_train_step.argtypes = [h, C.POINTER(mv_inputs), C.POINTER(mv_targets),
                                C.POINTER(mv_losses)]
  lib.mv_train_forward_backward.argtypes = lib.mv_train_step.argtypes
  lib.mv_upload_targets.argtypes = [h, C.POINTER(mv_targets)]
  lib.mv_grad_buffer.argtypes = [h, C.POINTER(_fp), C.POINTER(C.c_int64)]
  lib.mv_train_apply.argtypes = [h, C.c_float]
  lib.mv_get_grad.argtypes = [h, C.c_char_p, _fp, C.c_int64]
  lib.mv_get_global_step.argtypes = [h, C.POINTER(C.c_int64)]
  lib.mv_set_global_step.argtypes = [h, C.c_int64]
  lib.mv_comm_unique_id.argtypes = [_u8p]
  lib.mv_allreduce_init.argtypes = [h, C.c_int32, C.c_int32, _u8p]
  lib.mv_allreduce_info.argtypes = [h, _ip, _ip, _ip, _dp]
  lib.mv_attack_begin.argtypes = [h]
  lib.mv_attack_end.argtypes = [h]
  lib.mv_set_scene_feat.argtypes = [h, _fp]
  lib.mv_get_scene_feat.argtypes = [h, _fp]
  lib.mv_get_scene_grad.argtypes = [h, _fp]
  lib.mv_attack_step.argtypes = [h, C.c_float, C.c_float]
  lib.mv_scene_mix.argtypes = [h, _fp, C.c_float]
  lib.mv_get_sample_losses.argtypes = [h, C.c_int32, _fp]
  lib.mv_set_label_mixup.argtypes = [h, C.POINTER(_ip), C.POINTER(_ip), C.c_float, _fp]
  lib.mv_clear_label_mixup.argtypes = [h]
  lib.mv_set_dropout_seed.argtypes = [h, C.c_uint32]
  lib.mv_get_opt_scalars.argtypes = [h, _fp, _fp]
  lib.mv_set_opt_scalars.argtypes = [h, C.c_float, C.c_float]
  lib.mv_get_opt_slot.argtypes = [h, C.c_char_p, C.c_int32, _fp, C.c_int64]
  lib.mv_set_opt_slot.argtypes = [h, C.c_char_p, C.c_int32, _fp, C.c_int64]
  lib.mv_op_convlstm_bwd.argtypes = [C.c_int] + [_fp] * 7 + [C.c_int32] * 5 + [_fp] * 5
  lib.mv_op_gnn_bwd.argtypes = [C.c_int, _fp, _fp, _fp] + [C.c_int32] * 5 + [_fp, _fp]
  if lib.mv_abi_version() != MV_ABI_VERSION:
    raise MvError("ABI version mismatch: library %d, binding %d"
                  % (lib.mv_abi_version(), MV_ABI_VERSION))
  _lib = lib
  return lib


def f32(a):
  return np.ascontiguousarray(a, dtype=np.float32)


def i32(a):
  return np.ascontiguousarray(a, dtype=np.int32)


def fptr(a):
  return a.ctypes.data_as(_fp) if a is not None else _fp()


def iptr(a):
  return a.ctypes.data_as(_ip) if a is not None else _ip()


def check(rc, handle=None):
  if rc != 0:
    msg = load().mv_last_error(handle)
    raise MvError(msg.decode("utf-8", "replace") if msg else "error %d" % rc)


ACTIVATIONS = {"tanh": 0, "relu": 1, "lrelu": 2, "leaky_relu": 2}


def activation_code(act):
  """--activation_func (code/pred_utils.py:86-94): a name, or the TF function process_args
  replaced it with (its __name__: tanh / relu / leaky_relu)."""
  name = act if isinstance(act, str) else getattr(act, "__name__", None)
  if name not in ACTIVATIONS:
    raise MvError("activation_func %r: tanh, relu or lrelu" % (act,))
  return ACTIVATIONS[name]


def make_config(cfg):
  """argparse.Namespace (after process_args) -> mv_config."""
  c = mv_config()
  c.abi_version = MV_ABI_VERSION
  c.batch_size = int(cfg.batch_size)
  c.obs_len = int(cfg.obs_len)
  c.max_pred_len = int(getattr(cfg, "max_pred_len", None) or cfg.pred_len)
  c.scene_h, c.scene_w, c.scene_class = int(cfg.scene_h), int(cfg.scene_w), \
      int(cfg.scene_class)
  c.scene_conv_dim = int(cfg.scene_conv_dim)
  c.scene_conv_kernel = int(cfg.scene_conv_kernel)
  c.emb_size = int(cfg.emb_size)
  if int(cfg.enc_hidden_size) != int(cfg.dec_hidden_size):
    raise MvError("enc_hidden_size != dec_hidden_size is not supported")
  c.hidden_size = int(cfg.enc_hidden_size)
  c.convlstm_kernel = int(cfg.convlstm_kernel)
  if len(cfg.scene_grids) > MV_MAX_SCALES:
    raise MvError("at most %d scales" % MV_MAX_SCALES)
  c.num_scales = len(cfg.scene_grids)
  for s, (h, w) in enumerate(cfg.scene_grids):
    c.grid_h[s], c.grid_w[s] = int(h), int(w)
    c.use_grid[s] = 1 if cfg.use_grids[s] else 0
  c.use_gnn = 1 if cfg.use_gnn else 0
  c.beam_size = int(cfg.beam_size) if getattr(cfg, "use_beam_search", False) else 1
  c.diverse_beam = 1 if getattr(cfg, "diverse_beam", False) else 0
  c.diverse_gamma = float(getattr(cfg, "diverse_gamma", 1.0))
  c.fix_num_timestep = int(getattr(cfg, "fix_num_timestep", 0))
  # --use_teacher_forcing at test time: the class decoder is fed its raw logits
  # (code/pred_models.py:388-406, the not-training arm of the teacher_forcing branch)
  c.class_feedback_dense = 1 if (getattr(cfg, "use_teacher_forcing", False) and
                                 not getattr(cfg, "is_train", False)) else 0
  c.use_single_decoder = 1 if getattr(cfg, "use_single_decoder", False) else 0
  c.simaug_graph = 1 if getattr(cfg, "simaug_graph", False) else 0
  c.activation = activation_code(getattr(cfg, "activation_func", "tanh"))
  return c


def soft_grid_kernel(soft_grid):
  """The `--soft_grid` stamp of Model.get_feed_dict (code/pred_models.py:1084-1120)."""
  table = {1: (0.1, 1.0), 2: (0.01, 1.0), 3: (0.05, 1.0), 4: (0.0125, 0.9),
           5: (0.05, 0.6), 6: (0.1, 0.2)}
  if soft_grid == 7:
    k = np.full((5, 5), 0.0625)
    k[1:4, 1:4] = 0.0125
    k[2, 2] = 0.8
    return k
  if soft_grid not in table:
    raise MvError("soft_grid %r not in 1..7" % (soft_grid,))
  ring, centre = table[soft_grid]
  k = np.full((3, 3), ring)
  k[1, 1] = centre
  return k


def make_train_config(cfg, world=1):
  """Trainer.__init__ / Model.build_loss fields of the config
  (reference code/pred_models.py:1636-1717, 961-1040) -> mv_train_config.
  `world`: data-parallel ranks.  cfg.batch_size is the PER-RANK batch; the schedules
  count optimizer steps over the global batch (batch_size * world), as a single-GPU
  run of the reference with that batch size would."""
  kinds = {"adadelta": 0, "momentum": 1, "adam": 2, "rmsprop": 3}
  if cfg.optimizer not in kinds:
    raise MvError("Optimizer not implemented: %r" % (cfg.optimizer,))   # pred_models.py:1681
  if not (0.0 < float(cfg.keep_prob) <= 1.0):
    raise MvError("keep_prob %r not in (0, 1]" % (cfg.keep_prob,))
  t = mv_train_config()
  t.optimizer = kinds[cfg.optimizer]
  tf_mode = bool(getattr(cfg, "use_teacher_forcing", False))
  # decoder_loop_fn (code/pred_models.py:388-436): teacher forcing wins over
  # input_onehot = not is_train or train_w_onehot (:285)
  t.class_feedback = 2 if tf_mode else (0 if cfg.train_w_onehot else 1)
  t.reg_teacher_forcing = 1 if tf_mode else 0
  t.use_soft_grid_class = 1 if getattr(cfg, "use_soft_grid_class", False) else 0
  k = soft_grid_kernel(int(getattr(cfg, "soft_grid", 1))) if t.use_soft_grid_class \
      else np.zeros((3, 3))
  t.soft_kernel_size = int(k.shape[0])
  for i, v in enumerate(np.asarray(k, dtype=np.float64).reshape(-1)):
    t.soft_kernel[i] = float(v)
  t.mask_grid_regression = 1 if getattr(cfg, "mask_grid_regression", False) else 0
  t.keep_prob = float(cfg.keep_prob)
  t.init_lr = float(cfg.init_lr)
  t.emb_lr = float(cfg.emb_lr)
  t.use_cosine_lr = 1 if getattr(cfg, "use_cosine_lr", False) else 0
  t.has_decay = 0 if cfg.learning_rate_decay is None else 1
  t.learning_rate_decay = float(cfg.learning_rate_decay or 1.0)
  gbatch = cfg.batch_size * max(1, int(world))
  t.decay_steps = int(cfg.train_num_examples / gbatch * cfg.num_epoch_per_decay)
  t.max_steps = int(cfg.train_num_examples / gbatch * cfg.num_epochs)
  t.do_clip = 0 if cfg.clip_gradient_norm is None else 1
  t.clip_gradient_norm = float(cfg.clip_gradient_norm or 0.0)
  t.wd = float(cfg.wd or 0.0)
  t.grid_loss_weight = float(cfg.grid_loss_weight)
  t.grid_reg_loss_weight = float(cfg.grid_reg_loss_weight)
  return t


MV_COMM_ID_BYTES = 128


def comm_unique_id():
  """128-byte RCCL unique id (rank 0 draws it, every rank passes it to comm_init)."""
  buf = (C.c_uint8 * MV_COMM_ID_BYTES)()
  check(load().mv_comm_unique_id(buf))
  return bytes(buf)


class Engine(object):
  """Owns one mv_handle."""

  def __init__(self, cfg, device=0):
    self.lib = load()
    self.cfg = cfg
    self.c_cfg = make_config(cfg)
    self.handle = C.c_void_p()
    rc = self.lib.mv_create(C.byref(self.c_cfg), int(device), C.byref(self.handle))
    if rc != 0:
      msg = self.lib.mv_last_error(None)
      self.handle = None
      raise MvError(msg.decode("utf-8", "replace"))
    self._keep = []

  def close(self):
    if getattr(self, "handle", None):
      self.lib.mv_destroy(self.handle)
      self.handle = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # pylint: disable=broad-except
      pass

  # ---- weights
  def param_specs(self):
    out = []
    name = C.create_string_buffer(512)
    shape = (C.c_int64 * 4)()
    for i in range(self.lib.mv_num_params(self.handle)):
      rank = self.lib.mv_param_info(self.handle, i, name, 512, shape)
      out.append((name.value.decode(), tuple(int(shape[d]) for d in range(rank))))
    return out

  def set_param(self, name, value):
    a = f32(value)
    shape = (C.c_int64 * max(a.ndim, 1))(*a.shape)
    check(self.lib.mv_set_param(self.handle, name.encode(), fptr(a), shape, a.ndim),
          self.handle)

  def set_params(self, params):
    for name, _ in self.param_specs():
      if name not in params:
        raise MvError("missing parameter %s" % name)
      self.set_param(name, params[name])

  def get_param(self, name):
    shape = dict(self.param_specs())[name]
    out = np.empty(shape, dtype=np.float32)
    check(self.lib.mv_get_param(self.handle, name.encode(), fptr(out), out.size),
          self.handle)
    return out

  # ---- inputs / outputs
  def _inputs(self, feed):
    cfg = self.cfg
    N, T = cfg.batch_size, cfg.obs_len
    inp = mv_inputs()
    keep = []
    obs_scene = i32(feed["obs_scene"]).reshape(N, T)
    scene_feat = f32(feed["scene_feat"])
    keep += [obs_scene, scene_feat]
    inp.obs_scene = iptr(obs_scene)
    inp.scene_feat = fptr(scene_feat)
    inp.num_scene_frames = int(scene_feat.shape[0])
    self._num_frames = int(scene_feat.shape[0])
    inp.pred_len = int(feed.get("pred_length", cfg.pred_len))
    for s, (h, w) in enumerate(cfg.scene_grids):
      if not cfg.use_grids[s]:
        continue
      lab = i32(feed["grid_obs_labels"][s]).reshape(N, T)
      reg = f32(feed["grid_obs_regress"][s]).reshape(N, T, h, w, 2)
      keep += [lab, reg]
      inp.grid_obs_labels[s] = iptr(lab)
      inp.grid_obs_regress[s] = fptr(reg)
    self._keep = keep
    return inp

  def _alloc_outputs(self, Tp):
    cfg = self.cfg
    N = cfg.batch_size
    out = mv_outputs()
    cls, reg = [], []
    for s, (h, w) in enumerate(cfg.scene_grids):
      if not cfg.use_grids[s]:
        cls.append([])
        reg.append([])
        continue
      a = np.empty((N, Tp, h, w, 1), dtype=np.float32)
      b = np.empty((N, Tp, h, w, 2), dtype=np.float32)
      out.grid_pred_class[s] = fptr(a)
      out.grid_pred_reg[s] = fptr(b)
      cls.append(a)
      reg.append(b)
    return out, cls, reg

  def _alloc_beam(self, Tp):
    cfg = self.cfg
    N, B = cfg.batch_size, cfg.beam_size
    s = [i for i, u in enumerate(cfg.use_grids) if u][0]
    h, w = cfg.scene_grids[s]
    out = mv_beam_outputs()
    arrs = {
        "best_beam": np.empty((N, Tp, h, w, 1), dtype=np.float32),
        # --use_single_decoder: per beam, [N*B, T, h, w, 2] (code/pred_models.py:287-296)
        "grid_reg": np.empty((N * B if getattr(cfg, "use_single_decoder", False) else N,
                              Tp, h, w, 2), dtype=np.float32),
        "logits": np.empty((N, B, Tp, h * w), dtype=np.float32),
        "ids": np.empty((N, B, Tp), dtype=np.int32),
        "logprobs": np.empty((N, B), dtype=np.float32),
    }
    out.best_beam = fptr(arrs["best_beam"])
    out.grid_reg = fptr(arrs["grid_reg"])
    out.logits = fptr(arrs["logits"])
    out.ids = iptr(arrs["ids"])
    out.logprobs = fptr(arrs["logprobs"])
    return out, arrs, s

  def forward_greedy(self, feed):
    inp = self._inputs(feed)
    out, cls, reg = self._alloc_outputs(inp.pred_len)
    check(self.lib.mv_forward_greedy(self.handle, C.byref(inp), C.byref(out)),
          self.handle)
    return cls, reg

  def forward_beam(self, feed):
    inp = self._inputs(feed)
    out, arrs, s = self._alloc_beam(inp.pred_len)
    check(self.lib.mv_forward_beam(self.handle, C.byref(inp), C.byref(out)),
          self.handle)
    return arrs, s

  # ---- resident-input path (bench)
  def upload(self, feed):
    inp = self._inputs(feed)
    self._pred_len = inp.pred_len
    check(self.lib.mv_upload_inputs(self.handle, C.byref(inp)), self.handle)

  # ---- compact inputs (device-side batch assembly)
  def set_grid_centers(self, centers):
    """centers: list over scales of float64 [H, W, 2] (data["grid_center_<s>"])."""
    sent = []
    for s, (h, w) in enumerate(self.cfg.scene_grids):
      if not self.cfg.use_grids[s]:
        sent.append(None)
        continue
      c = np.ascontiguousarray(np.asarray(centers[s], dtype=np.float64).reshape(h, w, 2))
      check(self.lib.mv_set_grid_centers(self.handle, s, c.ctypes.data_as(_dp)),
            self.handle)
      sent.append(c.copy())
    self._centers_sent = sent

  def _centers_current(self, centers):
    """Are the centres resident in the engine the ones of this feed?  (A second
    dataset -- val after train -- may carry different grid centres.)"""
    sent = getattr(self, "_centers_sent", None)
    if sent is None:
      return False
    for s, (h, w) in enumerate(self.cfg.scene_grids):
      if not self.cfg.use_grids[s]:
        continue
      c = np.asarray(centers[s], dtype=np.float64).reshape(h, w, 2)
      if sent[s] is None or not np.array_equal(sent[s], c):
        return False
    return True

  def upload_compact(self, feed):
    """feed: obs_scene, scene_feat (0/1, any dtype), grid_obs_labels, obs_xy
    [N, T_o, 2], optional num_rows, pred_length; with grid_pred_labels + pred_xy
    also the training targets (after train_init)."""
    cfg = self.cfg
    N, T = cfg.batch_size, cfg.obs_len
    if not self._centers_current(feed["grid_centers"]):
      self.set_grid_centers(feed["grid_centers"])
    scene_any = np.asarray(feed["scene_feat"])
    if scene_any.dtype != np.uint8 and not ((scene_any == 0) | (scene_any == 1)).all():
      raise MvError("upload_compact: scene_feat must be 0/1 masks (it is handed over as "
                    "uint8); use the dense upload for real-valued scene features")
    inp = mv_inputs_compact()
    obs_scene = i32(feed["obs_scene"]).reshape(N, T)
    scene = np.ascontiguousarray(np.asarray(feed["scene_feat"]).astype(np.uint8, copy=False))
    xy = np.ascontiguousarray(np.asarray(feed["obs_xy"], dtype=np.float64).reshape(N, T, 2))
    keep = [obs_scene, scene, xy]
    inp.obs_scene = iptr(obs_scene)
    inp.scene_feat_u8 = scene.ctypes.data_as(_u8p)
    inp.num_scene_frames = int(scene.shape[0])
    inp.pred_len = int(feed.get("pred_length", cfg.pred_len))
    inp.obs_xy = xy.ctypes.data_as(_dp)
    inp.num_rows = int(feed.get("num_rows", N))
    for s in range(len(cfg.scene_grids)):
      if not cfg.use_grids[s]:
        continue
      lab = i32(feed["grid_obs_labels"][s]).reshape(N, T)
      keep.append(lab)
      inp.grid_obs_labels[s] = iptr(lab)
    self._pred_len = inp.pred_len
    check(self.lib.mv_upload_inputs_compact(self.handle, C.byref(inp)), self.handle)
    if feed.get("pred_xy") is not None and getattr(self, "_tc", None) is not None:
      Tp = inp.pred_len
      tg = mv_targets_compact()
      pxy = np.ascontiguousarray(
          np.asarray(feed["pred_xy"], dtype=np.float64).reshape(N, -1, 2)[:, :Tp])
      keep.append(pxy)
      tg.pred_xy = pxy.ctypes.data_as(_dp)
      tg.num_rows = inp.num_rows
      for s in range(len(cfg.scene_grids)):
        if not cfg.use_grids[s]:
          continue
        lab = i32(feed["grid_pred_labels"][s]).reshape(N, Tp)
        keep.append(lab)
        tg.grid_pred_labels[s] = iptr(lab)
      check(self.lib.mv_upload_targets_compact(self.handle, C.byref(tg)), self.handle)
    del keep      # both calls copy synchronously

  # ---- pipelined greedy forward: feed of batch k+1 / fetch of batch k-1 under batch k
  def submit_greedy(self, feed, depth=2):
    """Queue one batch (H2D -> forward -> D2H on the engine's streams) and return; the
    caller's arrays are free again on return.  At most `depth` batches outstanding."""
    if not getattr(self, "_pipe_depth", 0):
      check(self.lib.mv_pipeline_create(self.handle, int(depth)), self.handle)
      self._pipe_depth = int(depth)
    inp = self._inputs(feed)
    check(self.lib.mv_submit_greedy(self.handle, C.byref(inp)), self.handle)
    self._pipe_lens = getattr(self, "_pipe_lens", []) + [inp.pred_len]

  def collect_greedy(self):
    """-> (cls, reg) of the OLDEST submitted batch (blocks until it is on the host)."""
    Tp = self._pipe_lens.pop(0)
    out, cls, reg = self._alloc_outputs(Tp)
    check(self.lib.mv_collect_greedy(self.handle, C.byref(out), None), self.handle)
    return cls, reg

  def forward_greedy_pipelined(self, feeds, depth=2):
    """[(cls, reg)] of a sequence of batches, bitwise those of forward_greedy on each; batch
    k+1 is submitted before batch k is collected."""
    feeds = list(feeds)
    outs = []
    for k, feed in enumerate(feeds):
      if k >= depth:
        outs.append(self.collect_greedy())
      self.submit_greedy(feed, depth)
    while len(outs) < len(feeds):
      outs.append(self.collect_greedy())
    return outs

  def forward_greedy_compact(self, feed):
    self.upload_compact(feed)
    self.run_resident(False)
    return self.download()

  def forward_beam_compact(self, feed):
    self.upload_compact(feed)
    self.run_resident(True)
    return self.download_beam()

  def train_step_compact(self, feed):
    self.upload_compact(feed)
    return self.train_step(None)

  def run_resident(self, beam=False):
    fn = self.lib.mv_run_beam_resident if beam else self.lib.mv_run_greedy_resident
    check(fn(self.handle), self.handle)

  def synchronize(self):
    check(self.lib.mv_synchronize(self.handle), self.handle)

  def time_resident(self, iters, beam=False):
    ms = C.c_float()
    fn = self.lib.mv_time_beam_resident if beam else self.lib.mv_time_greedy_resident
    check(fn(self.handle, int(iters), C.byref(ms)), self.handle)
    return float(ms.value)

  def download(self):
    out, cls, reg = self._alloc_outputs(self._pred_len)
    check(self.lib.mv_download_outputs(self.handle, C.byref(out)), self.handle)
    return cls, reg

  def download_beam(self):
    out, arrs, s = self._alloc_beam(self._pred_len)
    check(self.lib.mv_download_beam_outputs(self.handle, C.byref(out)), self.handle)
    return arrs, s

  # ---- training (Trainer.step)
  def train_init(self, cfg=None, world=None):
    """world None = the world of the previous train_init of this engine (1 for the first):
    re-initialising with another configuration (the SimAug attacks switch to their loss
    and back) must not reset the data-parallel LR / decay schedule to a single rank."""
    if world is None:
      world = getattr(self, "_train_world", 1)
    self._train_world = int(world)
    self._tc = make_train_config(cfg or self.cfg, world=world)
    check(self.lib.mv_train_init(self.handle, C.byref(self._tc)), self.handle)

  def _targets(self, feed):
    cfg = self.cfg
    N, Tp = cfg.batch_size, int(feed.get("pred_length", cfg.pred_len))
    tg = mv_targets()
    keep = []
    for s, (h, w) in enumerate(cfg.scene_grids):
      if not cfg.use_grids[s]:
        continue
      lab = i32(feed["grid_pred_labels"][s]).reshape(N, Tp)
      reg = f32(feed["grid_pred_regress"][s]).reshape(N, Tp, h, w, 2)
      keep += [lab, reg]
      tg.grid_pred_labels[s] = iptr(lab)
      tg.grid_pred_regress[s] = fptr(reg)
    self._keep_t = keep
    return tg

  @staticmethod
  def _losses(L):
    return (float(L.loss), float(L.wd_loss),
            [float(L.pred_grid_loss[i]) for i in range(L.num_pred_grid_loss)])

  def train_step(self, feed=None):
    """-> (loss, wd_loss, pred_grid_loss list); feed None = resident batch."""
    L = mv_losses()
    if feed is None:
      rc = self.lib.mv_train_step(self.handle, None, None, C.byref(L))
    else:
      inp, tg = self._inputs(feed), self._targets(feed)
      rc = self.lib.mv_train_step(self.handle, C.byref(inp), C.byref(tg), C.byref(L))
    check(rc, self.handle)
    return self._losses(L)

  def train_forward_backward(self, feed=None):
    L = mv_losses()
    if feed is None:
      rc = self.lib.mv_train_forward_backward(self.handle, None, None, C.byref(L))
    else:
      inp, tg = self._inputs(feed), self._targets(feed)
      rc = self.lib.mv_train_forward_backward(self.handle, C.byref(inp), C.byref(tg),
                                              C.byref(L))
    check(rc, self.handle)
    return self._losses(L)

  def upload_targets(self, feed):
    tg = self._targets(feed)
    check(self.lib.mv_upload_targets(self.handle, C.byref(tg)), self.handle)

  def train_apply(self, grad_scale=1.0):
    check(self.lib.mv_train_apply(self.handle, float(grad_scale)), self.handle)

  def grad_buffer(self):
    """(device pointer, element count) of the flat gradient buffer."""
    p, n = _fp(), C.c_int64()
    check(self.lib.mv_grad_buffer(self.handle, C.byref(p), C.byref(n)), self.handle)
    return C.cast(p, C.c_void_p).value, int(n.value)

  def get_grad(self, name):
    shape = dict(self.param_specs())[name]
    out = np.empty(shape, dtype=np.float32)
    check(self.lib.mv_get_grad(self.handle, name.encode(), fptr(out), out.size),
          self.handle)
    return out

  def get_opt_slot(self, name, slot):
    shape = dict(self.param_specs())[name]
    out = np.empty(shape, dtype=np.float32)
    check(self.lib.mv_get_opt_slot(self.handle, name.encode(), int(slot), fptr(out),
                                   out.size), self.handle)
    return out

  def set_opt_slot(self, name, slot, value):
    a = f32(value)
    check(self.lib.mv_set_opt_slot(self.handle, name.encode(), int(slot), fptr(a),
                                   a.size), self.handle)

  # ---- in-library gradient all-reduce (RCCL)
  def comm_init(self, rank, world, unique_id):
    if len(unique_id) != MV_COMM_ID_BYTES:
      raise MvError("unique id must be %d bytes" % MV_COMM_ID_BYTES)
    buf = (C.c_uint8 * MV_COMM_ID_BYTES).from_buffer_copy(unique_id)
    check(self.lib.mv_allreduce_init(self.handle, int(rank), int(world), buf), self.handle)
    self.comm_world = int(world)

  def comm_info(self):
    r, w, b, by = C.c_int32(), C.c_int32(), C.c_int32(), C.c_double()
    if self.lib.mv_allreduce_info(self.handle, C.byref(r), C.byref(w), C.byref(b),
                                  C.byref(by)) != 0:
      return None
    return {"rank": r.value, "world": w.value, "buckets": b.value, "bytes": by.value}

  # ---- SimAug extras: attacks on the resident scene features
  def _scene_shape(self):
    cfg = self.cfg
    return (self._num_frames, cfg.scene_h, cfg.scene_w, cfg.scene_class)

  def attack_begin(self):
    check(self.lib.mv_attack_begin(self.handle), self.handle)

  def attack_end(self):
    check(self.lib.mv_attack_end(self.handle), self.handle)

  def set_scene_feat(self, a):
    a = f32(a).reshape(self._scene_shape())
    check(self.lib.mv_set_scene_feat(self.handle, fptr(a)), self.handle)

  def get_scene_feat(self):
    out = np.empty(self._scene_shape(), dtype=np.float32)
    check(self.lib.mv_get_scene_feat(self.handle, fptr(out)), self.handle)
    return out

  def get_scene_grad(self):
    out = np.empty(self._scene_shape(), dtype=np.float32)
    check(self.lib.mv_get_scene_grad(self.handle, fptr(out)), self.handle)
    return out

  def attack_step(self, epsilon, step):
    check(self.lib.mv_attack_step(self.handle, float(epsilon), float(step)), self.handle)

  def scene_mix(self, other, weight):
    o = None if other is None else f32(other).reshape(self._scene_shape())
    check(self.lib.mv_scene_mix(self.handle, fptr(o), float(weight)), self.handle)

  def sample_losses(self, scale):
    out = np.empty((self.cfg.batch_size,), dtype=np.float32)
    check(self.lib.mv_get_sample_losses(self.handle, int(scale), fptr(out)), self.handle)
    return out

  def set_label_mixup(self, obs_labels2, pred_labels2, weight, sample_weight=None):
    """SimAug multi-view experiment 3: train on w * one_hot(label) + (1 - w) * one_hot(label2)
    (lists over scales of [N, T_o] / [N, T_p] int arrays, None for unused scales) until
    clear_label_mixup(); sample_weight [N]: per-sample weights of the class loss."""
    S = len(self.cfg.scene_grids)
    keep, po, pp = [], (_ip * MV_MAX_SCALES)(), (_ip * MV_MAX_SCALES)()
    for s in range(S):
      if not self.cfg.use_grids[s]:
        continue
      o = np.ascontiguousarray(obs_labels2[s], dtype=np.int32).reshape(self.cfg.batch_size, -1)
      p = np.ascontiguousarray(pred_labels2[s], dtype=np.int32).reshape(self.cfg.batch_size, -1)
      if o.shape[1] != self.cfg.obs_len or p.shape[1] != getattr(self, "_pred_len", p.shape[1]):
        raise MvError("set_label_mixup: labels of scale %d have shapes %s / %s" %
                      (s, o.shape, p.shape))
      keep += [o, p]
      po[s], pp[s] = o.ctypes.data_as(_ip), p.ctypes.data_as(_ip)
    sw = None
    if sample_weight is not None:
      sw = np.ascontiguousarray(sample_weight, dtype=np.float32).reshape(self.cfg.batch_size)
    check(self.lib.mv_set_label_mixup(self.handle, po, pp, float(weight),
                                      fptr(sw) if sw is not None else None), self.handle)

  def clear_label_mixup(self):
    check(self.lib.mv_clear_label_mixup(self.handle), self.handle)

  def set_dropout_seed(self, seed):
    check(self.lib.mv_set_dropout_seed(self.handle, int(seed) & 0xFFFFFFFF), self.handle)

  def opt_scalars(self):
    a, b = C.c_float(), C.c_float()
    check(self.lib.mv_get_opt_scalars(self.handle, C.byref(a), C.byref(b)), self.handle)
    return float(a.value), float(b.value)

  def set_opt_scalars(self, b1p, b2p):
    check(self.lib.mv_set_opt_scalars(self.handle, float(b1p), float(b2p)), self.handle)

  @property
  def global_step(self):
    v = C.c_int64()
    if self.lib.mv_get_global_step(self.handle, C.byref(v)) != 0:
      raise MvError("mv_train_init has not been called")
    return int(v.value)

  @global_step.setter
  def global_step(self, step):
    if self.lib.mv_set_global_step(self.handle, int(step)) != 0:
      raise MvError("mv_train_init has not been called")

  def set_compute_mode(self, mode):
    """0 / "f32": fp32 MFMA; 1 / "f16x3": split-fp16 matrix pipe at fp32 accuracy;
    2 / "bf16": bf16 operands, fp32 accumulate and state (BASELINE configs[4])."""
    mode = {"f32": 0, "f16x3": 1, "bf16": 2}.get(mode, mode)
    check(self.lib.mv_set_compute_mode(self.handle, int(mode)), self.handle)

  def set_graph_mode(self, on):
    check(self.lib.mv_set_graph_mode(self.handle, 1 if on else 0), self.handle)

  # ---- measurement
  def set_profiling(self, on):
    check(self.lib.mv_set_profiling(self.handle, 1 if on else 0), self.handle)

  def reset_kernel_stats(self):
    check(self.lib.mv_reset_kernel_stats(self.handle), self.handle)

  def kernel_stats(self):
    out = {}
    name = C.create_string_buffer(256)
    n, ms, fl, by, dn = C.c_int64(), C.c_double(), C.c_double(), C.c_double(), C.c_double()
    mf = C.c_double()
    for i in range(self.lib.mv_num_kernel_stats(self.handle)):
      self.lib.mv_kernel_stat(self.handle, i, name, 256, C.byref(n), C.byref(ms),
                              C.byref(fl), C.byref(by))
      self.lib.mv_kernel_stat_dense_flops(self.handle, i, C.byref(dn))
      self.lib.mv_kernel_stat_mfma_flops(self.handle, i, C.byref(mf))
      # flops: algorithmic FLOPs the launches executed; flops_dense: the same steps as
      # the reference computes them (a zero-state encoder step still multiplies h = 0);
      # flops_mfma: FLOPs issued to the matrix pipe (f16x3: 3 MFMAs per product in the
      # direct form, 2 in the Winograd form)
      out[name.value.decode()] = {"launches": int(n.value), "total_ms": ms.value,
                                  "flops": fl.value, "bytes": by.value,
                                  "flops_dense": dn.value, "flops_mfma": mf.value}
    return out


# ------------------------------------------------ single-kernel entry points

def op_convlstm_step(x, c, h, kernel, biases, device=0):
  lib = load()
  x = f32(x)
  M, H, W, Cx = x.shape
  Cc = int(kernel.shape[3]) // 4
  kernel, biases = f32(kernel), f32(biases)
  c_out = np.empty((M, H, W, Cc), dtype=np.float32)
  h_out = np.empty((M, H, W, Cc), dtype=np.float32)
  if c is None:
    cp, hp = _fp(), _fp()
  else:
    c, h = f32(c), f32(h)
    cp, hp = fptr(c), fptr(h)
  check(lib.mv_op_convlstm_step(device, fptr(x), cp, hp, fptr(kernel), fptr(biases),
                                M, H, W, Cx, Cc, fptr(c_out), fptr(h_out)))
  return c_out, h_out


def op_convlstm_step16(x, c, h, kernel, biases, variant, device=0):
  """The step on the fp16 matrix pipe (f16x3 planes): variant 1 = direct form, 2 = Winograd
  F(2,3) over image rows.  Returns c', h' and the h' operand planes decoded to fp32."""
  lib = load()
  x = f32(x)
  M, H, W, Cx = x.shape
  Cc = int(kernel.shape[3]) // 4
  kernel, biases = f32(kernel), f32(biases)
  c_out = np.empty((M, H, W, Cc), dtype=np.float32)
  h_out = np.empty((M, H, W, Cc), dtype=np.float32)
  h16 = np.empty((M, H, W, Cc), dtype=np.float32)
  if c is None:
    cp, hp = _fp(), _fp()
  else:
    c, h = f32(c), f32(h)
    cp, hp = fptr(c), fptr(h)
  check(lib.mv_op_convlstm_step16(device, int(variant), fptr(x), cp, hp, fptr(kernel),
                                  fptr(biases), M, H, W, Cx, Cc, fptr(c_out), fptr(h_out),
                                  fptr(h16)))
  return c_out, h_out, h16


def op_gnn(h, scene_mean, device=0):
  lib = load()
  h, scene_mean = f32(h), f32(scene_mean)
  M, H, W, Cc = h.shape
  out = np.empty_like(h)
  check(lib.mv_op_gnn(device, fptr(h), fptr(scene_mean), M, H, W, Cc,
                      scene_mean.shape[-1], fptr(out)))
  return out


def op_hidden2grid(h, w, device=0):
  lib = load()
  h, w = f32(h), f32(w)
  M, H, W, Cc = h.shape
  P = w.shape[-1]
  out = np.empty((M, H, W, P), dtype=np.float32)
  check(lib.mv_op_hidden2grid(device, fptr(h), fptr(w), M, H, W, Cc, P, fptr(out)))
  return out


def op_beam_step(logits, prev_lp, time, diverse, gamma, fix_num_timestep, device=0):
  lib = load()
  logits, prev_lp = f32(logits), f32(prev_lp)
  N, B, K = logits.shape
  new_lp = np.empty((N, B), dtype=np.float32)
  ids = np.empty((N, B), dtype=np.int32)
  parents = np.empty((N, B), dtype=np.int32)
  check(lib.mv_op_beam_step(device, fptr(logits), fptr(prev_lp), N, B, K, int(time),
                            1 if diverse else 0, float(gamma), int(fix_num_timestep),
                            fptr(new_lp), iptr(ids), iptr(parents)))
  return new_lp, ids, parents


def op_convlstm_bwd(x, c, h, kernel, biases, dh_new, dc_new, device=0):
  """-> (dx, dh, dc, dkernel, dbiases)"""
  lib = load()
  x = f32(x)
  M, H, W, Cx = x.shape
  Cc = int(kernel.shape[3]) // 4
  kernel, biases = f32(kernel), f32(biases)
  dh_new, dc_new = f32(dh_new), f32(dc_new)
  if c is None:
    cp, hp = _fp(), _fp()
  else:
    c, h = f32(c), f32(h)
    cp, hp = fptr(c), fptr(h)
  dx = np.empty((M, H, W, Cx), dtype=np.float32)
  dh = np.empty((M, H, W, Cc), dtype=np.float32)
  dc = np.empty((M, H, W, Cc), dtype=np.float32)
  dk = np.empty(kernel.shape, dtype=np.float32)
  db = np.empty(biases.shape, dtype=np.float32)
  check(lib.mv_op_convlstm_bwd(device, fptr(x), cp, hp, fptr(kernel), fptr(biases),
                               fptr(dh_new), fptr(dc_new), M, H, W, Cx, Cc, fptr(dx),
                               fptr(dh), fptr(dc), fptr(dk), fptr(db)))
  return dx, dh, dc, dk, db


def op_gnn_bwd(h, scene_mean, g, device=0):
  lib = load()
  h, scene_mean, g = f32(h), f32(scene_mean), f32(g)
  M, H, W, Cc = h.shape
  dh = np.empty_like(h)
  ds = np.empty_like(scene_mean)
  check(lib.mv_op_gnn_bwd(device, fptr(h), fptr(scene_mean), fptr(g), M, H, W, Cc,
                          scene_mean.shape[-1], fptr(dh), fptr(ds)))
  return dh, ds
