"""`multifuture.get_inputs` / `add_grid` / `get_grid_input` (trajectory txt + scene-seg
npy + id map -> observed trajectories, grid classes, regression targets, one-hot scene
masks, T_pred per sample) against the functions of the reference's own
`code/multifuture_inference.py`, imported unmodified (TensorFlow = the eager shim, not
called on this path) and run on the same synthetic on-disk dataset.  Needs
/root/reference; skipped elsewhere."""
import argparse
import copy
import importlib
import os
import sys
from glob import glob

import numpy as np
import pytest

from multiverse_amd import multifuture as mf

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(os.path.dirname(HERE), "oracle", "tf1_shim"))
import mf_fixture  # noqa: E402
import run_reference as rr  # noqa: E402

pytestmark = pytest.mark.skipif(not rr.available(), reason="needs the /root/reference checkout")


def _reference_module():
  rr.import_reference()
  sys.path.insert(0, rr.REFERENCE_CODE)
  argv = sys.argv
  try:
    sys.argv = ["multifuture_inference.py"]
    for m in ("multifuture_inference", "pred_utils"):
      sys.modules.pop(m, None)
    mod = importlib.import_module("multifuture_inference")
  finally:
    sys.argv = argv
    sys.path.remove(rr.REFERENCE_CODE)
  assert os.path.abspath(mod.__file__).startswith(os.path.abspath(rr.REFERENCE_CODE))
  return mod


def _args(ds):
  return argparse.Namespace(
      traj_path=ds["traj_path"], multifuture_path=ds["multifuture_path"],
      scene_feat_path=ds["scene_feat_path"], scene_id2name=ds["scene_id2name"],
      num_out=3, obs_length=8, grid_strides="2,4", use_grids="0,1",
      scene_h=36, scene_w=64, scene_class=11, video_h=1080, video_w=1920)


def test_get_inputs_matches_the_reference(tmp_path):
  ref = _reference_module()
  ds = mf_fixture.make_dataset(str(tmp_path), n_traj=5)
  files = sorted(glob(os.path.join(ds["traj_path"], "*.txt")))
  ids = [os.path.splitext(os.path.basename(f))[0] for f in files]
  gt = mf.load_gt(ds["multifuture_path"], ids)

  a_ref = copy.deepcopy(_args(ds))
  ref.add_grid(a_ref)                                # the reference mutates args in place
  a_me = mf.add_grid(copy.deepcopy(_args(ds)))
  assert a_ref.scene_grids == a_me.scene_grids
  assert a_ref.scene_grid_strides == a_me.scene_grid_strides
  for c0, c1 in zip(a_ref.scene_grid_centers, a_me.scene_grid_centers):
    assert (np.asarray(c0) == np.asarray(c1)).all()

  r = ref.get_inputs(a_ref, files, gt)
  m = mf.get_inputs(a_me, files, gt)
  assert sorted(r) == sorted(m)
  assert r["max_pred_lengths"] == m["max_pred_lengths"]
  assert (np.asarray(r["scene_feats"]) == np.asarray(m["scene_feats"])).all()
  assert r["scene_feats"].dtype == m["scene_feats"].dtype
  for key in ("obs_traj", "obs_traj_rel", "obs_grid_class", "obs_scene"):
    for x, y in zip(r[key], m[key]):
      assert (np.asarray(x) == np.asarray(y)).all(), key
  for x, y in zip(r["obs_grid_target"], m["obs_grid_target"]):
    for xs, ys in zip(x, y):                         # per scale
      assert (np.asarray(xs) == np.asarray(ys)).all()


class _Placeholders(object):
  """Stand-in for the reference model object: get_feed_dict only uses its
  placeholder attributes as dictionary keys."""

  def __init__(self, n_scale):
    for k in ("obs_length", "pred_length", "is_train", "obs_scene", "obs_scene_mask",
              "scene_feat"):
      setattr(self, k, k)
    for k in ("grid_obs_labels", "grid_obs_regress", "grid_pred_regress",
              "grid_pred_labels_T"):
      setattr(self, k, ["%s_%d" % (k, j) for j in range(n_scale)])


def test_inference_feed_matches_the_reference(tmp_path):
  ref = _reference_module()
  ds = mf_fixture.make_dataset(str(tmp_path), n_traj=4)
  files = sorted(glob(os.path.join(ds["traj_path"], "*.txt")))
  ids = [os.path.splitext(os.path.basename(f))[0] for f in files]
  gt = mf.load_gt(ds["multifuture_path"], ids)
  args = mf.add_grid(copy.deepcopy(_args(ds)))
  args.use_soft_grid_class = False
  inputs = mf.get_inputs(args, files, gt)
  ph = _Placeholders(len(args.scene_grids))
  for idx in range(len(files)):
    want = ref.PredictionModelInference.get_feed_dict(ph, inputs, args, idx)
    feed, n_real = mf.inference_feed(inputs, args, [idx])
    assert n_real == 1
    assert feed["pred_length"] == int(want["pred_length"][0])
    assert (feed["obs_scene"] == want["obs_scene"]).all()
    assert feed["scene_feat"].dtype == want["scene_feat"].dtype == np.float32
    assert (feed["scene_feat"] == want["scene_feat"]).all()
    for j in range(len(args.scene_grids)):
      assert (feed["grid_obs_labels"][j] == want["grid_obs_labels_%d" % j]).all()
      if args.use_grids[j]:
        # the reference feeds float64 into a float32 placeholder: same values after the cast
        assert (feed["grid_obs_regress"][j] ==
                want["grid_obs_regress_%d" % j].astype("float32")).all()
      else:
        assert feed["grid_obs_regress"][j] is None
