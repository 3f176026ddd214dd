# coding=utf-8
"""Helpers for the fixtures written by oracle/tf1_shim/make_shim_golden.py
(outputs of the reference's unmodified pred_models.py on the TF-1 shim)."""
import os

import numpy as np

from multiverse_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
STRIDE = 997

FORWARD_CASES = {
    "golden_shim_greedy_cfg1.npz": dict(batch_size=4, use_grids=(1, 0)),
    "golden_shim_greedy_both.npz": dict(batch_size=2, use_grids=(1, 1)),
    "golden_shim_beam_s1.npz": dict(batch_size=2, use_grids=(0, 1), beam_size=5),
    "golden_shim_beam20_s0.npz": dict(batch_size=1, use_grids=(1, 0), beam_size=20),
    # --use_gnn off (the graph attention is optional in the reference, code/train.py)
    "golden_shim_greedy_nognn.npz": dict(batch_size=2, use_grids=(1, 1), use_gnn=False),
    # plain beam search: no diversity penalty, log-probs accumulated from the first step
    "golden_shim_beam_plain_s1.npz": dict(batch_size=2, use_grids=(0, 1), beam_size=4,
                                          diverse_beam=False, fix_num_timestep=0),
}


def load(name):
  return np.load(os.path.join(GOLD, name))


def forward_case(name):
  g = load(name)
  cfg = synth.default_config(**FORWARD_CASES[name])
  seed = int(g["seed"][0])
  params = synth.make_params(cfg, seed=seed, recurrent_gain=float(g["gain"][0]),
                             bias_scale=float(g["bias"][0]))
  feed = synth.make_feed(cfg, seed=seed)
  return g, cfg, params, feed


def train_case(name):
  g = load(name)
  seed, steps = int(g["seed"][0]), int(g["steps"][0])
  if name == "golden_shim_train_both.npz":
    cfg = synth.default_config(batch_size=2, use_grids=(1, 1), is_train=True)
  else:
    cfg = synth.default_config(batch_size=2, use_grids=(0, 1), is_train=True)
    cfg.train_num_examples = 2
  params = synth.make_params(cfg, seed=seed, recurrent_gain=2.0, bias_scale=0.1)
  feeds = [synth.make_feed(cfg, seed=seed + 100 + s) for s in range(steps)]
  return g, cfg, params, feeds


def digest(a):
  a = np.asarray(a, dtype=np.float32).reshape(-1)
  return np.concatenate([
      np.array([a.astype(np.float64).sum(), np.abs(a).astype(np.float64).sum(),
                np.abs(a).max()], dtype=np.float64),
      a[::STRIDE].astype(np.float64)])


def digest_err(a, gold):
  """max |sample - gold sample| / max|gold|, and the relative error of sum|.|"""
  d = digest(a)
  scale = max(gold[2], 1e-30)
  return (float(np.abs(d[3:] - gold[3:]).max() / scale),
          float(abs(d[1] - gold[1]) / max(gold[1], 1e-30)))


def var_table(g):
  out = {}
  for s in g["var_names"]:
    n, shp = str(s).split("|")
    out[n] = tuple(int(x) for x in shp.split(",")) if shp else ()
  return out


# ---- switches beyond the published run: mirror of oracle/tf1_shim/make_shim_golden.py
VARIANT_CASES = {
    "soft1": (dict(use_soft_grid_class=True, soft_grid=1), 1),
    "soft7_mask": (dict(use_soft_grid_class=True, soft_grid=7, mask_grid_regression=True), 1),
    "mask": (dict(mask_grid_regression=True), 1),
    "teacher": (dict(use_teacher_forcing=True), 1),
    "teacher_soft4": (dict(use_teacher_forcing=True, use_soft_grid_class=True, soft_grid=4), 1),
    "no_onehot": (dict(train_w_onehot=False), 1),
    "dropout07": (dict(keep_prob=0.7), 1),
    "momentum": (dict(optimizer="momentum", init_lr=0.01), 2),
    "rmsprop": (dict(optimizer="rmsprop", init_lr=0.001), 2),
    "adam": (dict(optimizer="adam", init_lr=0.001), 2),
    "cosine": (dict(use_cosine_lr=True, num_epochs=4, optimizer="momentum", init_lr=0.01), 3),
    "sck1": (dict(scene_conv_kernel=1), 1),
    "relu": (dict(activation_func="relu"), 1),
    "lrelu": (dict(activation_func="lrelu"), 1),
    "emb128": (dict(emb_size=128), 1),
    "hidden128": (dict(enc_hidden_size=128, dec_hidden_size=128), 1),
    "hidden512": (dict(enc_hidden_size=512, dec_hidden_size=512), 1),
    "ck1": (dict(convlstm_kernel=1), 1),
    "ck5": (dict(convlstm_kernel=5), 1),
    "scd128": (dict(scene_conv_dim=128), 1),
}
VARIANT_SEED = synth.SEED_BASE + 40


def variant_case(name):
  """(fixture, cfg, params, feeds) of one reference Trainer run with a non-published
  switch; feeds carry the dropout seed and (soft labels) the oracle's label maps."""
  from oracle import multiverse_oracle as oracle
  over, steps = VARIANT_CASES[name]
  g = load("golden_shim_variant_%s.npz" % name)
  cfg = synth.default_config(batch_size=2, use_grids=(0, 1), is_train=True, **over)
  cfg.train_num_examples = 2
  params = synth.make_params(cfg, seed=VARIANT_SEED, recurrent_gain=2.0, bias_scale=0.1)
  feeds = []
  for step in range(steps):
    feed = synth.make_feed(cfg, seed=VARIANT_SEED + 100 + step)
    feed["dropout_seed"] = 4242 + step
    if cfg.use_soft_grid_class:
      feed["grid_pred_soft"] = [
          oracle.soft_grid_labels(feed["grid_pred_labels"][s], h, w, cfg.soft_grid)
          for s, (h, w) in enumerate(cfg.scene_grids)]
    feeds.append(feed)
  return g, cfg, params, feeds


def single_decoder_case():
  """golden_shim_single_decoder.npz: (fixture, (cfg, params, feed) of the greedy run,
  (cfg, params, feeds) of the two training steps)."""
  g = load("golden_shim_single_decoder.npz")
  cfg = synth.default_config(batch_size=2, use_grids=(1, 1), use_single_decoder=True)
  params = synth.make_params(cfg, seed=VARIANT_SEED + 2, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=VARIANT_SEED + 2)
  tcfg = synth.default_config(batch_size=2, use_grids=(1, 1), use_single_decoder=True,
                              is_train=True)
  tcfg.train_num_examples = 2
  tparams = synth.make_params(tcfg, seed=VARIANT_SEED + 3, recurrent_gain=2.0, bias_scale=0.1)
  feeds = [synth.make_feed(tcfg, seed=VARIANT_SEED + 103 + s) for s in range(int(g["steps"][0]))]
  return g, (cfg, params, feed), (tcfg, tparams, feeds)


def single_decoder_beam_case():
  """golden_shim_single_decoder_beam.npz: --use_single_decoder with beam search (scale 1,
  N = 2, beam 4): (fixture, cfg, params, feed)."""
  g = load("golden_shim_single_decoder_beam.npz")
  cfg = synth.default_config(batch_size=2, use_grids=(0, 1), use_single_decoder=True,
                             beam_size=4)
  params = synth.make_params(cfg, seed=VARIANT_SEED + 4, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=VARIANT_SEED + 4)
  return g, cfg, params, feed
