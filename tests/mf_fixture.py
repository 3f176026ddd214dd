# coding=utf-8
"""Synthetic Forking-Paths style files (traj txt, scene-seg npy, multifuture
pickles, scene_id2name json) in the formats multifuture_inference.py reads."""
import json
import os
import pickle

import numpy as np


def make_dataset(root, n_traj=5, obs_length=8, seed=0):
  rng = np.random.default_rng(seed)
  traj_dir = os.path.join(root, "traj_2.5fps")
  seg_dir = os.path.join(root, "scene_seg")
  mf_dir = os.path.join(root, "multifuture")
  for d in (traj_dir, seg_dir, mf_dir):
    os.makedirs(d, exist_ok=True)
  # raw segmentation ids (ADE20k-like) -> 10 kept classes + background
  old_ids = [3, 6, 9, 11, 12, 21, 40, 52, 76, 91]
  id_map = {"oldid2new": {str(o): i + 1 for i, o in enumerate(old_ids)},
            "id2name": {str(i + 1): "class%d" % o for i, o in enumerate(old_ids)}}
  id_file = os.path.join(root, "scene36_64_id2name_top10.json")
  with open(id_file, "w") as f:
    json.dump(id_map, f)
  traj_ids = []
  for k in range(n_traj):
    cam = "cam4" if k % 2 == 0 else "cam%d" % (k % 3 + 1)
    pid = 10 + k
    traj_id = "%04d_%d_%d_%s" % (k, k % 3, pid, cam)
    traj_ids.append(traj_id)
    frames = [100 + 12 * t for t in range(obs_length)]
    pos = rng.uniform([300, 200], [1600, 900])
    vel = rng.normal(0, 20, size=2)
    lines = []
    pts = []
    for fr in frames:
      pts.append(pos.copy())
      lines.append("%d\t%d\t%.3f\t%.3f" % (fr, pid, pos[0], pos[1]))
      other = pos + rng.normal(0, 200, size=2)       # another agent in the frame
      lines.append("%d\t%d\t%.3f\t%.3f" % (fr, pid + 500, other[0], other[1]))
      pos = np.clip(pos + vel, [1, 1], [1919, 1079])
    with open(os.path.join(traj_dir, traj_id + ".txt"), "w") as f:
      f.write("\n".join(lines) + "\n")
    os.makedirs(os.path.join(seg_dir, traj_id), exist_ok=True)
    for fr in frames:
      seg = rng.choice(old_ids + [0, 150, 7], size=(36, 64))   # 150, 7: ids outside the map
      np.save(os.path.join(seg_dir, traj_id, "%s_F_%08d.npy" % (traj_id, fr)), seg)
    futures = {}
    for a in range(2 + k % 2):
      T = 12 + 2 * ((k + a) % 3)
      p = pts[-1].copy()
      v = vel + rng.normal(0, 10, size=2)
      tr = []
      for t in range(T):
        p = np.clip(p + v, [1, 1], [1919, 1079])
        tr.append((frames[-1] + 12 * (t + 1), pid, float(p[0]), float(p[1])))
      futures["annotator%d" % a] = {"x_agent_traj": tr}
    with open(os.path.join(mf_dir, traj_id + ".p"), "wb") as f:
      pickle.dump(futures, f)
  return {"traj_path": traj_dir, "scene_feat_path": seg_dir, "multifuture_path": mf_dir,
          "scene_id2name": id_file, "traj_ids": traj_ids}
