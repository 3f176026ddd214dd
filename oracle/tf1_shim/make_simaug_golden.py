#!/usr/bin/env python
# coding=utf-8
"""Freeze runs of the reference's UNMODIFIED SimAug/code/pred_models.py (on the eager TF-1
shim, random ops replayed from multiverse_amd.simaug.Draws) into tests/golden/golden_simaug.npz.

    python oracle/tf1_shim/make_simaug_golden.py          # needs /root/reference

Cases (tests/simaug_cases.py): the greedy Tester forward of SimAug's graph (whose greedy
graph attention ignores the scene features, SimAug/code/pred_models.py:1219-1227) and its
diverse beam-5 decode (whose attention does see them); six
white_box_attack configurations; the four multi-view experiments (experiment 3 twice).  Per
case: the augmented features (digest), target labels / beta weight / selected view / focal
weights, the training loss of the step that follows, and for two cases every gradient.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import simaug_cases as sc  # noqa: E402
from multiverse_amd import simaug  # noqa: E402
from oracle.tf1_shim import run_simaug as rs  # noqa: E402


def main():
  out = {}
  cfg = sc.config(False)
  params, feed = sc.base_inputs(cfg)
  cls, reg = rs.forward(cfg, params, feed)
  out["forward|cls_1"], out["forward|reg_1"] = cls[1], reg[1]
  bcfg = sc.beam_config()
  tf, ref, model = rs.build_model(bcfg, params, feed, is_train=False)
  out["beam|best"] = rs._np(tf, model.grid_pred_decoded[1])
  out["beam|reg"] = rs._np(tf, model.grid_pred_reg_decoded[1])
  out["beam|logits"], out["beam|ids"], out["beam|logprobs"] = [
      rs._np(tf, t) for t in model.beam_outputs]
  for name, (over, seed) in sc.WHITE_BOX.items():
    cfg = sc.config(True, adv_train=True, **over)
    r = rs.run_white_box(cfg, params, feed, simaug.Draws(seed))
    out["wb|%s|adv" % name] = sc.digest(r["adv"])
    out["wb|%s|target" % name] = r["target"].astype("int32")
    out["wb|%s|loss" % name] = np.array([r["loss"]], dtype=np.float64)
    if name == "fgsm":
      for n, g in r["grads"].items():
        if g is not None:
          out["wb|fgsm|grad|%s" % n] = sc.digest(g)
    print("white box", name, "loss", r["loss"])
  for name, (over, seed) in sc.MULTIVIEW.items():
    cfg = sc.config(True, multiview_train=True, **over)
    f0, _, _ = sc.multiview_feed(cfg, feed)
    r = rs.run_multiview(cfg, params, f0, simaug.Draws(seed))
    out["mv|%s|mixed" % name] = sc.digest(r["mixed"])
    out["mv|%s|weight" % name] = np.array([r["weight"]], dtype=np.float64)
    out["mv|%s|loss" % name] = np.array([r["loss"]], dtype=np.float64)
    if "select" in r:
      out["mv|%s|select" % name] = r["select"].astype("int32")
      out["mv|%s|focal" % name] = r["focal"].astype("float32")
    if name == "exp3_dw":
      for n, g in r["grads"].items():
        if g is not None:
          out["mv|exp3_dw|grad|%s" % n] = sc.digest(g)
    print("multiview", name, "loss", r["loss"], "weight", r["weight"])
  np.savez_compressed(sc.GOLD, **out)
  print("wrote", sc.GOLD, os.path.getsize(sc.GOLD), "bytes")


if __name__ == "__main__":
  main()
