#!/usr/bin/env python
# coding=utf-8
"""Freeze outputs of the reference's UNMODIFIED code/pred_models.py (executed
on the eager TF-1 shim next to this file) into tests/golden/golden_shim_*.npz.

    python oracle/tf1_shim/make_shim_golden.py          # needs /root/reference

These fixtures are what pins oracle/multiverse_oracle.py and, on the GPU box
(where /root/reference does not exist), the HIP engine itself:
  - greedy forward (Tester.step): BASELINE config 1 (single scale, N=4), both scales;
  - beam search: scale 1 N=2 B=5, and the reference's own inference
    configuration batch 1 / beam 20 / gamma 0.01 / fix_num_timestep 1 on scale 0;
  - one Trainer.step on both scales and two consecutive steps on scale 1:
    loss, wd_loss, pred_grid_loss, and for every variable the gradient and the
    updated value as (sum, sum|.|, max|.|, every STRIDE-th element);
  - the variable names / shapes the reference code asks tf.get_variable for.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")

from multiverse_amd import synth  # noqa: E402
from oracle.tf1_shim import run_reference as rr  # noqa: E402

STRIDE = 997


def digest(a):
  a = np.asarray(a, dtype=np.float32).reshape(-1)
  return np.concatenate([
      np.array([a.astype(np.float64).sum(), np.abs(a).astype(np.float64).sum(),
                np.abs(a).max()], dtype=np.float64),
      a[::STRIDE].astype(np.float64)])


def forward_case(name, cfg, seed, gain, bias):
  params = synth.make_params(cfg, seed=seed, recurrent_gain=gain, bias_scale=bias)
  feed = synth.make_feed(cfg, seed=seed)
  cls, reg, beam = rr.forward(cfg, params, feed)
  out = {"seed": np.array([seed]), "gain": np.array([gain]), "bias": np.array([bias]),
         "var_names": np.array(["%s|%s" % (n, ",".join(map(str, s)))
                                for n, s in rr.variable_names()])}
  for s in range(len(cfg.scene_grids)):
    if not cfg.use_grids[s]:
      assert cls[s] == [] and reg[s] == []
      continue
    out["cls_%d" % s] = np.asarray(cls[s])
    out["reg_%d" % s] = np.asarray(reg[s])
  if beam is not None:
    out["beam_logits"], out["beam_ids"], out["beam_logprobs"] = [np.asarray(b) for b in beam]
  np.savez_compressed(os.path.join(GOLD, name), **out)
  print("wrote", name, {k: v.shape for k, v in out.items()})


def train_case(name, cfg, seed, steps):
  params = synth.make_params(cfg, seed=seed, recurrent_gain=2.0, bias_scale=0.1)
  out = {"seed": np.array([seed]), "steps": np.array([steps])}
  slots, gs = {}, 0
  for step in range(steps):
    feed = synth.make_feed(cfg, seed=seed + 100 + step)
    loss, wd, pgl, grads, params, slots, gs = rr.train_step(cfg, params, feed, slots, gs)
    out["loss_%d" % step] = np.array([loss, wd] + pgl, dtype=np.float64)
    for n in sorted(grads):
      out["grad_%d|%s" % (step, n)] = digest(grads[n])
    print(name, "step", step, "loss", loss, "global_step", gs)
  for n in sorted(params):
    out["param|%s" % n] = digest(params[n])
  out["global_step"] = np.array([gs])
  np.savez_compressed(os.path.join(GOLD, name), **out)
  print("wrote", name)


# Switches of the reference beyond the published run: every case
# is ONE config override set on scale 1 (9x16), batch 2; the frozen numbers are the
# reference's own Trainer.step on the shim.  tests/shim_golden.py VARIANT_CASES mirrors
# this table.
VARIANT_CASES = [
    # name, config overrides, steps
    ("soft1", dict(use_soft_grid_class=True, soft_grid=1), 1),
    ("soft7_mask", dict(use_soft_grid_class=True, soft_grid=7, mask_grid_regression=True), 1),
    ("mask", dict(mask_grid_regression=True), 1),
    ("teacher", dict(use_teacher_forcing=True), 1),
    ("teacher_soft4", dict(use_teacher_forcing=True, use_soft_grid_class=True, soft_grid=4), 1),
    ("no_onehot", dict(train_w_onehot=False), 1),
    ("dropout07", dict(keep_prob=0.7), 1),
    ("momentum", dict(optimizer="momentum", init_lr=0.01), 2),
    ("rmsprop", dict(optimizer="rmsprop", init_lr=0.001), 2),
    ("adam", dict(optimizer="adam", init_lr=0.001), 2),
    # --use_cosine_lr (code/pred_models.py:1646-1654): max_steps = 4, lr 1, 0.854, 0.5 x init
    ("cosine", dict(use_cosine_lr=True, num_epochs=4, optimizer="momentum", init_lr=0.01), 3),
    # --scene_conv_kernel 1 (code/train.py:65): the scene stack as two strided 1x1 projections
    ("sck1", dict(scene_conv_kernel=1), 1),
    # --activation_func relu / lrelu (code/train.py:58-59, code/pred_utils.py:86-94): the scene
    # convolutions and grid_emb; tf.nn.leaky_relu's default alpha 0.2
    ("relu", dict(activation_func="relu"), 1),
    ("lrelu", dict(activation_func="lrelu"), 1),
    # shapes other than the published run's (code/train.py:53-57): --emb_size 128 is the flag's
    # own default, --enc/dec_hidden_size 128 / 512 the neighbouring widths
    ("emb128", dict(emb_size=128), 1),
    ("hidden128", dict(enc_hidden_size=128, dec_hidden_size=128), 1),
    ("hidden512", dict(enc_hidden_size=512, dec_hidden_size=512), 1),
    # --convlstm_kernel 1 / 5 (code/train.py:70) and --scene_conv_dim 128 (code/train.py:69):
    # the engine's generic-tap fp32 path (csrc/convlstm_generic.h), the two-channels-per-lane
    # graph attention
    ("ck1", dict(convlstm_kernel=1), 1),
    ("ck5", dict(convlstm_kernel=5), 1),
    ("scd128", dict(scene_conv_dim=128), 1),
]
VARIANT_SEED = synth.SEED_BASE + 40


def variant_train_case(name, over, steps):
  cfg = synth.default_config(batch_size=2, use_grids=(0, 1), is_train=True, **over)
  cfg.train_num_examples = 2
  params = synth.make_params(cfg, seed=VARIANT_SEED, recurrent_gain=2.0, bias_scale=0.1)
  out = {"steps": np.array([steps])}
  slots, gs = {}, 0
  for step in range(steps):
    feed = synth.make_feed(cfg, seed=VARIANT_SEED + 100 + step)
    feed["dropout_seed"] = 4242 + step
    loss, wd, pgl, grads, params, slots, gs = rr.train_step(cfg, params, feed, slots, gs)
    out["loss_%d" % step] = np.array([loss, wd] + pgl, dtype=np.float64)
    for n in sorted(grads):
      out["grad_%d|%s" % (step, n)] = digest(grads[n])
    print(name, "step", step, "loss", loss)
  for n in sorted(params):
    out["param|%s" % n] = digest(params[n])
  for n in sorted(slots):
    if n == "":
      out["opt_scalars"] = np.asarray(slots[n], dtype=np.float64)
    else:
      for i, sl in enumerate(slots[n]):
        out["slot%d|%s" % (i, n)] = digest(np.asarray(sl))
  out["global_step"] = np.array([gs])
  np.savez_compressed(os.path.join(GOLD, "golden_shim_variant_%s.npz" % name), **out)
  print("wrote golden_shim_variant_%s.npz" % name)


def teacher_test_forward_case():
  """--use_teacher_forcing at TEST time: the class decoder is fed its raw logits."""
  cfg = synth.default_config(batch_size=2, use_grids=(0, 1), use_teacher_forcing=True)
  params = synth.make_params(cfg, seed=VARIANT_SEED + 1, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=VARIANT_SEED + 1)
  cls, reg, _ = rr.forward(cfg, params, feed)
  np.savez_compressed(os.path.join(GOLD, "golden_shim_variant_teacher_test.npz"),
                      cls_1=np.asarray(cls[1]), reg_1=np.asarray(reg[1]))
  print("wrote golden_shim_variant_teacher_test.npz")


def single_decoder_case():
  """--use_single_decoder (code/pred_models.py:274,287-296) on BOTH scales: the offset kernel
  person_pred/decode_reg/out_dec_grid/W is shared by the scales, the regression encoder's
  variables exist without a gradient.  A greedy Tester.step and two Trainer.steps."""
  out = {}
  cfg = synth.default_config(batch_size=2, use_grids=(1, 1), use_single_decoder=True)
  params = synth.make_params(cfg, seed=VARIANT_SEED + 2, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=VARIANT_SEED + 2)
  cls, reg, _ = rr.forward(cfg, params, feed)
  out["var_names"] = np.array(["%s|%s" % (n, ",".join(map(str, s)))
                               for n, s in rr.variable_names()])
  for s in range(2):
    out["cls_%d" % s], out["reg_%d" % s] = np.asarray(cls[s]), np.asarray(reg[s])
  cfg = synth.default_config(batch_size=2, use_grids=(1, 1), use_single_decoder=True,
                             is_train=True)
  cfg.train_num_examples = 2
  params = synth.make_params(cfg, seed=VARIANT_SEED + 3, recurrent_gain=2.0, bias_scale=0.1)
  slots, gs, steps = {}, 0, 2
  for step in range(steps):
    feed = synth.make_feed(cfg, seed=VARIANT_SEED + 103 + step)
    loss, wd, pgl, grads, params, slots, gs = rr.train_step(cfg, params, feed, slots, gs)
    out["loss_%d" % step] = np.array([loss, wd] + pgl, dtype=np.float64)
    for n in sorted(grads):
      if grads[n] is not None:
        out["grad_%d|%s" % (step, n)] = digest(grads[n])
    out["no_grad"] = np.array(sorted(n for n in grads if grads[n] is None))
    print("single_decoder step", step, "loss", loss)
  for n in sorted(params):
    out["param|%s" % n] = digest(params[n])
  out["steps"] = np.array([steps])
  out["global_step"] = np.array([gs])
  np.savez_compressed(os.path.join(GOLD, "golden_shim_single_decoder.npz"), **out)
  print("wrote golden_shim_single_decoder.npz")


def single_decoder_beam_case():
  """--use_single_decoder WITH --use_beam_search (code/pred_models.py:274, 287-296): the
  decoder states are traced back along every beam and the offsets decoded from them per
  beam, grid_pred_reg_decoded [N * beam, T, H, W, 2]."""
  cfg = synth.default_config(batch_size=2, use_grids=(0, 1), use_single_decoder=True,
                             beam_size=4)
  params = synth.make_params(cfg, seed=VARIANT_SEED + 4, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=VARIANT_SEED + 4)
  cls, reg, beam = rr.forward(cfg, params, feed)
  np.savez_compressed(os.path.join(GOLD, "golden_shim_single_decoder_beam.npz"),
                      cls_1=np.asarray(cls[1]), reg_1=np.asarray(reg[1]),
                      beam_logits=np.asarray(beam[0]), beam_ids=np.asarray(beam[1]),
                      beam_logprobs=np.asarray(beam[2]))
  print("wrote golden_shim_single_decoder_beam.npz", np.asarray(reg[1]).shape)


def main():
  assert rr.available(), "needs the reference checkout (/root/reference)"
  if len(sys.argv) > 1 and sys.argv[1] == "single_beam":
    single_decoder_beam_case()
    return
  if len(sys.argv) > 1 and sys.argv[1] == "variants":
    only = sys.argv[2:] or None
    for name, over, steps in VARIANT_CASES:
      if only is None or name in only:
        variant_train_case(name, over, steps)
    teacher_test_forward_case()
    return
  if len(sys.argv) > 1 and sys.argv[1] == "nognn":
    forward_case("golden_shim_greedy_nognn.npz",
                 synth.default_config(batch_size=2, use_grids=(1, 1), use_gnn=False),
                 synth.SEED_BASE + 10, 3.0, 0.1)
    return
  if len(sys.argv) > 1 and sys.argv[1] == "beam_plain":
    forward_case("golden_shim_beam_plain_s1.npz",
                 synth.default_config(batch_size=2, use_grids=(0, 1), beam_size=4,
                                      diverse_beam=False, fix_num_timestep=0),
                 synth.SEED_BASE + 9, 3.0, 0.1)
    return
  if len(sys.argv) > 1 and sys.argv[1] == "single":
    single_decoder_case()
    return
  forward_case("golden_shim_greedy_cfg1.npz",
               synth.default_config(batch_size=4, use_grids=(1, 0)),
               synth.SEED_BASE + 0, 3.0, 0.1)
  forward_case("golden_shim_greedy_both.npz",
               synth.default_config(batch_size=2, use_grids=(1, 1)),
               synth.SEED_BASE + 1, 1.0, 0.0)
  forward_case("golden_shim_beam_s1.npz",
               synth.default_config(batch_size=2, use_grids=(0, 1), beam_size=5),
               synth.SEED_BASE + 5, 3.0, 0.1)
  forward_case("golden_shim_beam20_s0.npz",
               synth.default_config(batch_size=1, use_grids=(1, 0), beam_size=20),
               synth.SEED_BASE + 6, 3.0, 0.1)
  train_case("golden_shim_train_both.npz",
             synth.default_config(batch_size=2, use_grids=(1, 1), is_train=True),
             synth.SEED_BASE + 7, 1)
  cfg = synth.default_config(batch_size=2, use_grids=(0, 1), is_train=True)
  cfg.train_num_examples = 2     # decay_steps = 2: the LR staircase moves inside the run
  train_case("golden_shim_train_s1_3steps.npz", cfg, synth.SEED_BASE + 8, 3)


if __name__ == "__main__":
  main()
