# coding=utf-8
"""GPU: edge cases of the forward in both compute modes against the oracle --
batch 1, T_pred = 1 and T_pred = 25 (the longest multi-future horizon,
code/multifuture_eval_trajs_prob.py:75), every (n, t) on its own scene frame,
trajectories pinned to the grid corners (border handling of every 3x3 op),
beam variants (no diversity penalty, fix_num_timestep 0, beam 2)."""
import numpy as np
import pytest

from multiverse_amd import synth
from oracle import multiverse_oracle as oracle

from beam_compare import compare_beams

pytestmark = pytest.mark.gpu
TOL = 1e-4
MODES = ["f32", "f16x3"]


def _run_greedy(built_lib, cfg, params, feed, mode):
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode(mode)
  cls, reg = eng.forward_greedy(feed)
  eng.close()
  return cls, reg


def _check(cfg, cls, reg, ocls, oreg, Tp):
  N = cfg.batch_size
  for s in range(2):
    if not cfg.use_grids[s]:
      continue
    assert cls[s].shape == ocls[s].shape
    assert (cls[s].reshape(N, Tp, -1).argmax(-1) == ocls[s].reshape(N, Tp, -1).argmax(-1)).all()
    assert np.abs(cls[s] - ocls[s]).max() < TOL
    assert np.abs(reg[s] - oreg[s]).max() < TOL


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("Tp", [1, 25])
def test_batch1_and_extreme_pred_len(built_lib, mode, Tp):
  cfg = synth.default_config(batch_size=1, use_grids=(0, 1))
  cfg.max_pred_len = 25
  params = synth.make_params(cfg, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 60, pred_len=Tp)
  cls, reg = _run_greedy(built_lib, cfg, params, feed, mode)
  ocls, oreg, _ = oracle.forward(params, cfg, feed)
  _check(cfg, cls, reg, ocls, oreg, Tp)


@pytest.mark.parametrize("mode", MODES)
def test_corner_cells_and_all_unique_frames(built_lib, mode):
  """Observed cells in the four grid corners (every 3x3 stencil clipped), and
  one scene frame per (n, t) (U = N * T_o, no frame sharing)."""
  cfg = synth.default_config(batch_size=4, use_grids=(1, 1))
  params = synth.make_params(cfg, recurrent_gain=2.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 61, frames_per_group=1)
  N, T = 4, cfg.obs_len
  rng = np.random.default_rng(5)
  U = N * T
  feed["scene_feat"] = synth.make_scene_feat(rng, U, cfg).astype("float32")
  feed["obs_scene"] = np.arange(U, dtype="int32").reshape(N, T)
  for s, (h, w) in enumerate(cfg.scene_grids):
    corners = [0, w - 1, (h - 1) * w, h * w - 1]
    lab = np.array([[corners[n]] * T for n in range(N)], dtype="int32")
    lab[:, ::2] = np.array(corners)[::-1][:, None]      # hop between corners
    feed["grid_obs_labels"][s] = lab
  cls, reg = _run_greedy(built_lib, cfg, params, feed, mode)
  ocls, oreg, _ = oracle.forward(params, cfg, feed)
  _check(cfg, cls, reg, ocls, oreg, cfg.pred_len)


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("B,diverse,fix", [(2, False, 0), (4, True, 0), (3, False, 2)])
def test_beam_variants(built_lib, mode, B, diverse, fix):
  cfg = synth.default_config(batch_size=3, use_grids=(0, 1), beam_size=B)
  cfg.diverse_beam = diverse
  cfg.fix_num_timestep = fix
  params = synth.make_params(cfg, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 62 + B)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode(mode)
  arrs, s = eng.forward_beam(feed)
  eng.close()
  trace = {}
  _, oreg, obeam = oracle.forward(params, cfg, feed, trace=trace)
  compare_beams(arrs, oreg[1], obeam[0], obeam[1], obeam[2],
                np.stack(trace["beam_step_topvals"], axis=-1), trace["beam_trace"])
