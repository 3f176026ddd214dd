# coding=utf-8
"""CPU: the pure-numpy half of multiverse_amd/simaug.py (no engine): draws in the reference's
call order, per-step scene feeds, tiling, attack configuration -- against the frozen runs of
SimAug's own model file where a reference value exists (tests/golden/golden_simaug.npz)."""
import copy

import numpy as np
import pytest

from multiverse_amd import _lib, simaug, synth

import simaug_cases as sc


def test_random_targets_equal_the_reference_draw():
  """create_random_target (SimAug/code/pred_models.py:67-72) replayed from Draws(11): the
  labels the reference's own run attacked towards; never the true class."""
  g = np.load(sc.GOLD)
  cfg = sc.config(True, adv_train=True)
  _, feed = sc.base_inputs(cfg)
  t = simaug.random_targets(feed["grid_pred_labels"][1], 9 * 16, simaug.Draws(11))
  assert (t == g["wb|fgsm|target"]).all()
  assert (t != feed["grid_pred_labels"][1]).all() and t.min() >= 0 and t.max() < 144


def test_start_adv_draw_order_and_clean_start():
  cfg = sc.config(True)
  x = np.zeros((4, 3, 3, 2), "float32")
  d = simaug.Draws(5)
  a = simaug.start_adv(x, cfg, d)
  assert np.abs(a).max() <= cfg.adv_epsilon and np.abs(a).max() > 0
  cfg.adv_start_from_clean_prob = 1.0          # the noise is still drawn (graph order), unused
  d1, d2 = simaug.Draws(5), simaug.Draws(5)
  assert simaug.start_adv(x, cfg, d1) is x
  d2.noise(x.shape, cfg.adv_epsilon)
  assert d1.scalar() == d2.scalar()            # both consumed exactly one noise draw
  cfg.adv_start_from_clean_prob = 0.5          # one scalar decides noise vs clean
  outs = {float(np.abs(simaug.start_adv(x, cfg, simaug.Draws(s))).max() > 0) for s in range(12)}
  assert outs == {0.0, 1.0}


def test_per_step_scene_feed_and_tiling():
  cfg = sc.config(True)
  _, feed = sc.base_inputs(cfg)
  pf = simaug.per_step_scene_feed(cfg, feed)
  N, T = cfg.batch_size, cfg.obs_len
  assert pf["scene_feat"].shape[0] == N * T and (pf["obs_scene"] == np.arange(N * T).reshape(N, T)).all()
  raw = np.asarray(feed["scene_feat"])[np.asarray(feed["obs_scene"]).reshape(-1)]
  assert np.array_equal(pf["scene_feat"], raw * 2.0 - 1.0)          # norm_input
  assert set(np.unique(pf["scene_feat"])) <= {-1.0, 1.0}
  M = 3
  tiled = simaug._tile_feed(cfg, pf, M, np.repeat(
      pf["scene_feat"].reshape((N, T) + pf["scene_feat"].shape[1:]), M, axis=0).reshape(
          (-1,) + pf["scene_feat"].shape[1:]))
  assert tiled["obs_scene"].shape == (N * M, T)
  lab = np.asarray(pf["grid_obs_labels"][1])
  assert (tiled["grid_obs_labels"][1] == np.repeat(lab, M, axis=0)).all()      # row n * M + m


def test_attack_config_and_scale_checks():
  cfg = sc.config(True)
  a = simaug.attack_config(cfg)
  assert a.grid_reg_loss_weight == 0.0 and a.wd == 0.0 and not a.use_soft_grid_class
  assert cfg.wd != 0.0                                     # a copy, not the caller's config
  two = copy.copy(cfg)
  two.use_grids = [True, True]
  with pytest.raises(_lib.MvError, match="only one scale"):
    simaug._scale_of(two)
  cfg3 = copy.copy(cfg)
  cfg3.multiview_exp = 3
  with pytest.raises(_lib.MvError, match="exp3"):
    simaug.multiview_augmentation(None, cfg3, {"scene_feat": np.zeros((16, 2, 2, 1))}, None,
                                  simaug.Draws(0))


def test_draws_are_reproducible_and_typed():
  a, b = simaug.Draws(3), simaug.Draws(3)
  assert (a.label_offset((2, 5), 7) == b.label_offset((2, 5), 7)).all()
  o = a.label_offset((100,), 7)
  assert o.dtype == np.int32 and o.min() >= 1 and o.max() <= 6
  n = a.noise((50,), 0.1)
  assert n.dtype == np.float32 and np.abs(n).max() <= 0.1
  i = a.index(100, 1, 3)
  assert i.dtype == np.int32 and set(np.unique(i)) <= {1, 2}
  assert 0.0 < a.beta(1.0) < 1.0 and 0.0 <= a.scalar() < 1.0
