# coding=utf-8
"""Command lines of the reference, re-hosted on the MI355X engine (same flags,
same printed metrics): code/train.py, code/test.py, code/multifuture_inference.py,
code/multifuture_eval_trajs.py, code/multifuture_eval_trajs_prob.py.

The reference scripts own a tf.Session / tf.train.Saver (code/train.py:170-180,
code/test.py:143-150); here the engine handle plays the session and
`pred_utils.Saver` writes / reads the same TensorFlow checkpoint format.  Flags
are declared from one table so that train / test keep their (different)
reference defaults in one place.  argparse prefix matching stays on: the
documented commands rely on it (`--use_scene` for `--use_scene_enc`,
`--obs_len` for `--obs_length`; TRAINING.md:32-39, TESTING.md:84-93).
"""

from __future__ import annotations

import argparse
import math
import os
import pickle
import sys
from glob import glob

import numpy as np

B = "store_true"
# (flag, type or "store_true", train default, test default, scripts)
_FLAGS = [
    ("--runId", int, 0, 0, "tT"), ("--gpuid", int, 0, 0, "tT"),
    ("--load", B, None, None, "tT"), ("--load_best", B, None, None, "tT"),
    ("--load_from", str, None, None, "tT"), ("--save_output", str, None, None, "tT"),
    ("--obs_len", int, 8, 8, "tT"), ("--pred_len", int, 12, 12, "tT"),
    ("--per_scene_eval", B, None, None, "tT"),
    ("--show_grid_acc_at_T", B, None, None, "T"), ("--show_center_only", B, None, None, "T"),
    ("--emb_size", int, 128, 128, "tT"),
    ("--enc_hidden_size", int, 256, 256, "tT"), ("--dec_hidden_size", int, 256, 256, "tT"),
    ("--activation_func", str, "tanh", "tanh", "tT"),
    ("--scene_conv_kernel", int, 3, 3, "tT"), ("--scene_h", int, 36, 36, "tT"),
    ("--scene_w", int, 64, 64, "tT"), ("--scene_class", int, 11, 11, "tT"),
    ("--scene_conv_dim", int, 64, 64, "tT"), ("--convlstm_kernel", int, 3, 3, "tT"),
    ("--pool_scale_idx", int, 0, 0, "T"),
    ("--scene_grid_strides", str, "2,4,8", "4,8,16", "tT"),
    ("--use_grids", str, "1,1,1", "1,1,1", "tT"), ("--val_grid_num", int, 1, 1, "tT"),
    ("--use_beam_search", B, None, None, "tT"), ("--diverse_beam", B, None, None, "tT"),
    ("--beam_size", int, 5, 5, "tT"),
    ("--diverse_gamma", float, 1.0, 1.0, "T"), ("--fix_num_timestep", int, 0, 0, "T"),
    ("--use_gn", B, None, None, "tT"),
    ("--use_teacher_forcing", B, None, None, "tT"), ("--train_w_onehot", B, None, None, "t"),
    ("--use_soft_grid_class", B, None, None, "tT"), ("--soft_grid", int, 1, 1, "tT"),
    ("--mask_grid_regression", B, None, None, "tT"), ("--use_gnn", B, None, None, "tT"),
    ("--use_scene_enc", B, None, None, "tT"), ("--use_single_decoder", B, None, None, "tT"),
    ("--use_gt_grid", B, None, None, "tT"), ("--check_model", B, None, None, "t"),
    ("--loss_moving_avg_step", int, 100, 100, "tT"),
    ("--grid_loss_weight", float, 1.0, 1.0, "tT"),
    ("--grid_reg_loss_weight", float, 0.1, 1.0, "tT"),
    ("--save_period", int, 300, 300, "tT"), ("--batch_size", int, 64, 64, "tT"),
    ("--num_epochs", int, 100, 100, "tT"), ("--keep_prob", float, 1.0, 1.0, "tT"),
    ("--wd", float, 0.0001, 0.0001, "tT"), ("--clip_gradient_norm", float, 10, 10, "tT"),
    ("--optimizer", str, "adadelta", "adadelta", "tT"), ("--use_cosine_lr", B, None, None, "tT"),
    ("--learning_rate_decay", float, 0.95, 0.95, "tT"),
    ("--num_epoch_per_decay", float, 2.0, 2.0, "tT"),
    ("--init_lr", float, 0.2, 0.2, "tT"), ("--emb_lr", float, 1.0, 1.0, "tT"),
    # not a reference flag: device-side batch assembly (pred_models.compact_inputs_enabled)
    ("--compact_inputs", B, None, None, "tT"),
]


def model_parser(kind):
  """kind: "t" = train.py (code/train.py:25-138), "T" = test.py (code/test.py:22-134)."""
  p = argparse.ArgumentParser()
  p.add_argument("prepropath", type=str)
  p.add_argument("outbasepath", type=str,
                 help="full path will be outbasepath/modelname/runId")
  p.add_argument("modelname", type=str)
  for flag, typ, d_train, d_test, where in _FLAGS:
    if kind not in where:
      continue
    if typ == B:
      p.add_argument(flag, action="store_true")
    else:
      p.add_argument(flag, type=typ, default=d_train if kind == "t" else d_test)
  return p


def _finish_args(args, is_train):
  from multiverse_amd import pred_utils
  args.is_train = is_train
  args.is_test = not is_train
  for missing, val in (("diverse_gamma", 1.0), ("fix_num_timestep", 0),
                       ("train_w_onehot", False)):
    if not hasattr(args, missing):
      setattr(args, missing, val)
  return pred_utils.process_args(args)


def train_main(argv=None):
  """code/train.py:142-281."""
  from multiverse_amd import pred_models, pred_utils
  args = _finish_args(model_parser("t").parse_args(argv), True)
  train_data = pred_utils.read_data(args, "train")
  val_data = pred_utils.read_data(args, "val")
  args.train_num_examples = train_data.num_examples
  model = pred_models.get_model(args, gpuid=args.gpuid)
  if args.check_model:
    print("--------------- Model Weights -----------------")
    for name, shape in model.param_specs():
      print("%s:0 %s\n" % (name, tuple(shape)))
    return
  trainer = pred_models.Trainer(model, args)
  tester = pred_models.Tester(model, args)
  saver, bestsaver = pred_utils.Saver(max_to_keep=5), pred_utils.Saver(max_to_keep=5)
  if args.load or args.load_best:
    pred_utils.initialize(load=True, load_best=args.load_best, args=args, model=model)
  else:
    # tf.global_variables_initializer(): the reference's own initialisers
    from multiverse_amd import synth
    model.load_params(synth.make_params(args, seed=synth.SEED_BASE + args.runId))
  steps_per_epoch = int(math.ceil(train_data.num_examples / float(args.batch_size)))
  num_steps = steps_per_epoch * args.num_epochs
  print(" batch_size:%s, epoch:%s, %s step every epoch, total step:%s,"
        " eval/save every %s steps" % (args.batch_size, args.num_epochs, steps_per_epoch,
                                       num_steps, args.save_period))
  metric = "grid%d_traj_ade" % args.val_grid_num
  best = {metric: 999999, "step": -1}
  finalperf, is_start = None, True
  loss, wd_loss = [pred_utils.FIFO_ME(args.loss_moving_avg_step) for _ in range(2)]
  pred_grid_loss = [pred_utils.FIFO_ME(args.loss_moving_avg_step)
                    for _ in range(sum(args.use_grids))] * 2
  global_step = 0
  for batch in train_data.get_batches(args.batch_size, num_steps=num_steps):
    global_step = trainer.global_step + 1
    if (global_step % args.save_period == 0) or ((args.load_best or args.load) and is_start):
      print("\tsaving model %s..." % global_step)
      saver.save(model, args.save_dir_model, global_step=global_step)
      evalperf = pred_utils.evaluate(val_data, args, None, tester)
      print(("\tmoving average of %s steps: loss:%s, wd_loss:%s, pred_grid_loss:%s,"
             " eval on validation:%s, (best %s:%s at step %s) ") % (
                 args.loss_moving_avg_step, loss, wd_loss, pred_grid_loss,
                 ["%s: %.4f" % (k, evalperf[k]) for k in sorted(evalperf.keys())],
                 metric, best[metric], best["step"]))
      if evalperf[metric] < best[metric]:
        best[metric], best["step"] = evalperf[metric], global_step
        print("\t saving best model...")
        bestsaver.save(model, args.save_dir_best_model, global_step=global_step)
      finalperf, is_start = evalperf, False
    this_loss, _, this_wd_loss, this_pgl = trainer.step(None, batch)
    if math.isnan(this_loss):        # code/train.py:256-259
      print("nan loss.")
      print(this_pgl)
      sys.exit()
    loss.put(this_loss)
    wd_loss.put(this_wd_loss)
    for i in range(len(pred_grid_loss)):
      pred_grid_loss[i].put(this_pgl[i])
  if global_step % args.save_period != 0:
    saver.save(model, args.save_dir_model, global_step=global_step)
  if finalperf is not None:
    print("best eval on val %s: %s at %s step, final step %s %s is %s" % (
        metric, best[metric], best["step"], global_step, metric, finalperf[metric]))
  model.close()


def test_main(argv=None):
  """code/test.py:137-191."""
  from multiverse_amd import pred_models, pred_utils
  args = _finish_args(model_parser("T").parse_args(argv), False)
  test_data = pred_utils.read_data(args, "test")
  print("total test samples:%s" % test_data.num_examples)
  model = pred_models.get_model(args, gpuid=args.gpuid)
  pred_utils.initialize(load=True, load_best=args.load_best, args=args, model=model)
  tester = pred_models.Tester(model, args, None)
  perf = pred_utils.evaluate(test_data, args, None, tester)
  print("performance:")
  key_metrics = []
  for i in range(len(args.scene_grids)):
    if not args.use_grids[i]:
      continue
    key_metrics += ["grid%d_acc" % i, "grid%d_traj_ade" % i, "grid%d_traj_fde" % i]
    if args.show_center_only:
      key_metrics += ["grid%d_centerOnly_traj_ade" % i, "grid%d_centerOnly_traj_fde" % i]
    if args.show_grid_acc_at_T:
      key_metrics += ["grid%d_acc_@T=%d" % (i, t) for t in (0, 4, 9, 11)]
  if args.per_scene_eval:
    scenes = ["0000", "0002", "0400", "0401", "0500"]
    key_metrics += ["%s_ade" % s for s in scenes] + ["%s_fde" % s for s in scenes]
  numbers = []
  for k in sorted(perf.keys()):
    print("%s, %s" % (k, perf[k]))
    if k in key_metrics:
      numbers.append(("%s" % perf[k], k))
  print(" ".join(k for _, k in numbers))
  print(" ".join(v for v, _ in numbers))
  model.close()
  return perf


_MF_FLAGS = [
    ("--num_out", int, 20), ("--save_prob_file", str, None), ("--greedy", B, None),
    ("--center_only", B, None), ("--cap_reg", B, None), ("--gpuid", int, 0),
    ("--obs_length", int, 8), ("--emb_size", int, 128), ("--enc_hidden_size", int, 256),
    ("--dec_hidden_size", int, 256), ("--grid_strides", str, "2,4"),
    ("--use_grids", str, "1,0"), ("--use_gn", B, None), ("--use_gnn", B, None),
    ("--use_scene_enc", B, None), ("--use_single_decoder", B, None),
    ("--use_soft_grid_class", B, None), ("--diverse_beam", B, None),
    ("--diverse_gamma", float, 1.0), ("--fix_num_timestep", int, 0),
    ("--scene_feat_path", str, None), ("--scene_id2name", str, None),
    ("--scene_h", int, 36), ("--scene_w", int, 64), ("--scene_class", int, 11),
    ("--convlstm_kernel", int, 3), ("--scene_conv_dim", int, 64),
    ("--scene_conv_kernel", int, 3), ("--video_h", int, 1080), ("--video_w", int, 1920),
    # not in the reference: samples with equal T_pred are decoded together
    ("--batch_size", int, 1),
]


def multifuture_inference_parser():
  """The argument table of code/multifuture_inference.py:22-76 (+ --batch_size)."""
  p = argparse.ArgumentParser()
  for pos in ("traj_path", "multifuture_path", "model_path"):
    p.add_argument(pos)
  p.add_argument("output_file", help="a pickle, traj_id -> all output")
  for flag, typ, default in _MF_FLAGS:
    if typ == B:
      p.add_argument(flag, action="store_true")
    else:
      p.add_argument(flag, type=typ, default=default)
  return p


def multifuture_inference_main(argv=None):
  """code/multifuture_inference.py:387-530."""
  from multiverse_amd import multifuture as mf, pred_models, pred_utils
  args = multifuture_inference_parser().parse_args(argv)
  mf.add_grid(args)
  assert sum(args.use_grids) == 1
  traj_files = glob(os.path.join(args.traj_path, "*.txt"))
  traj_ids = [os.path.splitext(os.path.basename(one))[0] for one in traj_files]
  gt_trajs = mf.load_gt(args.multifuture_path, traj_ids)
  inputs = mf.get_inputs(args, traj_files, gt_trajs)
  cfg = mf.model_config(args, batch_size=args.batch_size,
                        max_pred_len=max(inputs["max_pred_lengths"] + [12]))
  model = pred_models.Model(cfg, cfg.modelname, gpuid=args.gpuid)
  model.load_params(pred_utils.load_weights(args.model_path, scope="person_pred"))
  output_data, beam_prob = mf.run_inference(args, model, inputs, traj_ids)
  model.close()
  with open(args.output_file, "wb") as f:
    pickle.dump(output_data, f)
  if args.save_prob_file is not None:
    with open(args.save_prob_file, "wb") as f:
      pickle.dump(beam_prob, f)


def multifuture_eval_trajs_main(argv=None):
  """code/multifuture_eval_trajs.py."""
  from multiverse_amd import multifuture as mf
  p = argparse.ArgumentParser()
  p.add_argument("gt_path")
  p.add_argument("prediction_file")
  args = p.parse_args(argv)
  with open(args.prediction_file, "rb") as f:
    prediction = pickle.load(f)
  res = mf.eval_min_ade_fde(mf.load_gt(args.gt_path, list(prediction)), prediction)
  keys = ["45-degree", "top-down", "all"]
  print("ADE/FDE:")
  print(" ".join(keys + keys))
  print(" ".join(["%s" % res["ade"][k] for k in keys] + ["%s" % res["fde"][k] for k in keys]))
  return res


def multifuture_eval_trajs_prob_main(argv=None):
  """code/multifuture_eval_trajs_prob.py."""
  from multiverse_amd import multifuture as mf
  p = argparse.ArgumentParser()
  p.add_argument("gt_path")
  p.add_argument("prediction_file")
  for flag, default in (("--scene_h", 18), ("--scene_w", 32), ("--video_h", 1080),
                        ("--video_w", 1920)):
    p.add_argument(flag, type=int, default=default)
  args = p.parse_args(argv)
  with open(args.prediction_file, "rb") as f:
    predictions = pickle.load(f)
  nll, counts = mf.eval_grid_nll(mf.load_gt(args.gt_path, list(predictions)), predictions,
                                 args.scene_h, args.scene_w, args.video_h, args.video_w)
  keys = sorted(nll)
  print([counts[k] for k in keys])
  print("NLL:")
  print(" ".join(keys))
  print(" ".join("%s" % nll[k] for k in keys))
  return nll


_ = np
