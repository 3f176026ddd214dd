# coding=utf-8
"""GPU: the re-hosted command lines end to end on synthetic files --
train.py (Trainer + Saver writing TensorFlow-format checkpoints + periodic
evaluate), test.py (restore + evaluate), multifuture_inference.py (raw traj /
scene-seg files -> beam decode -> trajectories pickle) and the two eval scripts."""
import os
import pickle
from glob import glob

import numpy as np
import pytest

from multiverse_amd import cli, multifuture as mf, pred_utils, synth, tf_checkpoint
from oracle import multiverse_oracle as oracle

import mf_fixture

pytestmark = pytest.mark.gpu

MODEL_FLAGS = ["--emb_size", "32", "--scene_grid_strides", "2,4", "--use_grids", "0,1",
               "--use_scene_enc", "--use_gnn", "--val_grid_num", "1", "--batch_size", "4",
               "--wd", "0.001", "--init_lr", "0.3", "--grid_reg_loss_weight", "0.2"]


def _write_npz(prepro, cfg):
  os.makedirs(prepro, exist_ok=True)
  for split, n, seed in (("train", 10, 1), ("val", 6, 2), ("test", 7, 3)):
    np.savez(os.path.join(prepro, "data_%s.npz" % split),
             **synth.make_npz_data(cfg, n, seed=seed))


def test_train_test_cli_roundtrip(built_lib, tmp_path, capsys):
  cfg = synth.default_config(batch_size=4, use_grids=(0, 1))
  prepro, out = str(tmp_path / "prepro"), str(tmp_path / "out")
  _write_npz(prepro, cfg)
  cli.train_main([prepro, out, "mv", "--train_w_onehot", "--num_epochs", "2",
                  "--save_period", "3"] + MODEL_FLAGS)
  log = capsys.readouterr().out
  assert "saving model 3" in log and "saving best model" in log
  save_dir = os.path.join(out, "mv", "00", "save")
  best_dir = os.path.join(out, "mv", "00", "best")
  assert os.path.exists(os.path.join(save_dir, "checkpoint"))
  assert os.path.exists(os.path.join(best_dir, "checkpoint"))
  # 10 examples / batch 4 -> 3 steps per epoch, 2 epochs -> final save at step 6
  names = dict((n, s) for n, s, _ in tf_checkpoint.list_variables(save_dir))
  assert tf_checkpoint.resolve_checkpoint(save_dir).endswith("save-6")
  k = "person_pred/decoder_grid_class_1/decoder_rnn/dec_grid_1/kernel"
  assert names[k] == (3, 3, 288, 1024) and names[k + "/Adadelta"] == names[k]
  allv = tf_checkpoint.load_checkpoint(save_dir, skip_optimizer_slots=False)
  assert int(allv["global_step"].item()) == 5          # steps applied before the last save
  assert np.abs(allv[k + "/Adadelta"]).max() > 0
  perf = cli.test_main([prepro, out, "mv", "--load_best"] + MODEL_FLAGS)
  log = capsys.readouterr().out
  assert "grid1_traj_ade" in log and "total test samples:7" in log
  assert 0 <= perf["grid1_acc"] <= 1 and perf["grid1_traj_ade"] > 0
  # the restored weights are the saved ones, and evaluation is the engine's
  w = pred_utils.load_weights(best_dir)
  assert sorted(w) == sorted(synth.param_shapes(cfg))


def test_multifuture_cli(built_lib, tmp_path, capsys):
  ds = mf_fixture.make_dataset(str(tmp_path / "fp"), n_traj=4)
  cfg = synth.default_config(batch_size=1, use_grids=(0, 1), beam_size=5)
  params = synth.make_params(cfg, seed=synth.SEED_BASE + 31, recurrent_gain=3.0,
                             bias_scale=0.1)
  model_dir = str(tmp_path / "model")
  tf_checkpoint.save_checkpoint(os.path.join(model_dir, "save-best"), params,
                                global_step=100)
  out_file, prob_file = str(tmp_path / "out.p"), str(tmp_path / "prob.p")
  argv = [ds["traj_path"], ds["multifuture_path"], model_dir, out_file,
          "--save_prob_file", prob_file, "--num_out", "5", "--emb_size", "32",
          "--use_grids", "0,1", "--use_gnn", "--use_scene_enc", "--diverse_beam",
          "--diverse_gamma", "0.01", "--fix_num_timestep", "1",
          "--scene_feat_path", ds["scene_feat_path"], "--scene_id2name", ds["scene_id2name"],
          "--obs_len", "8"]                     # prefix of --obs_length, as TESTING.md:88
  cli.multifuture_inference_main(argv)
  out = pickle.load(open(out_file, "rb"))
  prob = pickle.load(open(prob_file, "rb"))
  ids = sorted(out)
  assert len(ids) == 4
  # batched decode of samples sharing T_pred == the reference's one-at-a-time loop
  out2_file = str(tmp_path / "out2.p")
  cli.multifuture_inference_main(argv[:3] + [out2_file] + argv[4:] + ["--batch_size", "3"])
  out2 = pickle.load(open(out2_file, "rb"))
  for t in ids:
    assert np.abs(np.asarray(out[t]) - np.asarray(out2[t])).max() < 1e-3
  # against the oracle for one sample: beam ids bit-exact -> identical cells
  args = mf.add_grid(cli.argparse.Namespace(
      grid_strides="2,4", use_grids="0,1", scene_h=36, scene_w=64, video_h=1080,
      video_w=1920, obs_length=8, scene_id2name=ds["scene_id2name"],
      scene_feat_path=ds["scene_feat_path"], scene_class=11))
  files = sorted(glob(os.path.join(ds["traj_path"], "*.txt")))
  fids = [os.path.splitext(os.path.basename(f))[0] for f in files]
  gt = mf.load_gt(ds["multifuture_path"], fids)
  inputs = mf.get_inputs(args, files, gt)
  i = 2
  feed, _ = mf.inference_feed(inputs, args, [i])
  ocfg = synth.default_config(batch_size=1, use_grids=(0, 1), beam_size=5)
  _, oreg, obeam = oracle.forward(params, ocfg, feed)
  centers = args.scene_grid_centers[1].reshape(-1, 2)
  T = inputs["max_pred_lengths"][i]
  want = np.array([[centers[obeam[1][0, b, t]] +
                    oreg[1][0].reshape(T, -1, 2)[t, obeam[1][0, b, t]]
                    for t in range(T)] for b in range(5)])
  assert np.abs(np.asarray(out[fids[i]]) - want).max() < 1e-2       # pixels
  assert prob[fids[i]][0].shape == (1, 5, T, 144)
  capsys.readouterr()
  res = cli.multifuture_eval_trajs_main([ds["multifuture_path"], out_file])
  nll = cli.multifuture_eval_trajs_prob_main([ds["multifuture_path"], prob_file,
                                              "--scene_h", "9", "--scene_w", "16"])
  log = capsys.readouterr().out
  assert "ADE/FDE:" in log and "NLL:" in log
  assert res["ade"]["all"] > 0 and nll["T=1"] > 0
