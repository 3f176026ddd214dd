# coding=utf-8
"""GPU: the f16x3 compute mode (gate convolution on the fp16 matrix pipe, every
product as three fp16 MFMAs over two pre-scaled planes per operand) is held to
the SAME parity bars as the fp32-MFMA path: argmax / beam ids bit-exact, logits
and regression maps within 1e-4 of the fp32 oracle and of the frozen runs of
the reference's own code; its error against an fp64 oracle is printed next to
the fp32 path's."""
import numpy as np
import pytest
import torch

from multiverse_amd import synth
from oracle import multiverse_oracle as oracle

import shim_golden as sg
from beam_compare import compare_beams

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _engine(built_lib, cfg, params, mode):
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode(mode)
  return eng


@pytest.mark.parametrize("use_grids,N,gain,bias", [((1, 1), 3, 1.0, 0.0),
                                                    ((1, 0), 4, 3.0, 0.1),
                                                    ((0, 1), 2, 3.0, 0.1)])
def test_greedy_f16x3_matches_oracle(built_lib, use_grids, N, gain, bias):
  cfg = synth.default_config(batch_size=N, use_grids=use_grids)
  params = synth.make_params(cfg, recurrent_gain=gain, bias_scale=bias)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 41)
  outs = {}
  for mode in ("f32", "f16x3"):
    eng = _engine(built_lib, cfg, params, mode)
    outs[mode] = eng.forward_greedy(feed)
    eng.close()
  o32 = oracle.forward(params, cfg, feed)
  o64 = oracle.forward(params, cfg, feed, dtype=torch.float64)
  for s in range(2):
    if not cfg.use_grids[s]:
      continue
    for k, name in ((0, "cls"), (1, "reg")):
      e16 = np.abs(outs["f16x3"][k][s] - o64[k][s]).max()
      e32 = np.abs(outs["f32"][k][s] - o64[k][s]).max()
      print("scale %d %s: |f16x3 - fp64| %.2e  |fp32 MFMA - fp64| %.2e" % (s, name, e16, e32))
      assert np.abs(outs["f16x3"][k][s] - o32[k][s]).max() < TOL
    ids = outs["f16x3"][0][s].reshape(N, 12, -1).argmax(-1)
    assert (ids == o32[0][s].reshape(N, 12, -1).argmax(-1)).all()


@pytest.mark.parametrize("name", ["golden_shim_greedy_cfg1.npz", "golden_shim_greedy_both.npz"])
def test_greedy_f16x3_against_reference_run(built_lib, name):
  g, cfg, params, feed = sg.forward_case(name)
  eng = _engine(built_lib, cfg, params, "f16x3")
  cls, reg = eng.forward_greedy(feed)
  eng.set_graph_mode(True)                      # hipGraph replay of the same mode
  cls2, reg2 = eng.forward_greedy(feed)
  eng.close()
  N = cfg.batch_size
  for s in range(2):
    if not cfg.use_grids[s]:
      continue
    assert (cls[s] == cls2[s]).all() and (reg[s] == reg2[s]).all()
    assert (cls[s].reshape(N, 12, -1).argmax(-1) ==
            g["cls_%d" % s].reshape(N, 12, -1).argmax(-1)).all()
    assert np.abs(cls[s] - g["cls_%d" % s]).max() < TOL
    assert np.abs(reg[s] - g["reg_%d" % s]).max() < TOL


@pytest.mark.parametrize("name,scale", [("golden_shim_beam_s1.npz", 1),
                                        ("golden_shim_beam20_s0.npz", 0)])
def test_beam_f16x3_against_reference_run(built_lib, name, scale):
  g, cfg, params, feed = sg.forward_case(name)
  eng = _engine(built_lib, cfg, params, "f16x3")
  arrs, s = eng.forward_beam(feed)
  eng.close()
  trace = {}
  oracle.forward(params, cfg, feed, trace=trace)
  compare_beams(arrs, g["reg_%d" % scale], g["beam_logits"], g["beam_ids"],
                g["beam_logprobs"], np.stack(trace["beam_step_topvals"], axis=-1),
                trace["beam_trace"])


def test_f16x3_runtime_pred_len_no_gnn_and_mode_switch(built_lib):
  cfg = synth.default_config(batch_size=2, use_grids=(0, 1), use_gnn=False)
  cfg.max_pred_len = 16
  params = synth.make_params(cfg, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 2, pred_len=15)
  eng = _engine(built_lib, cfg, params, "f16x3")
  a, ar = eng.forward_greedy(feed)
  eng.set_compute_mode("f32")
  b, br = eng.forward_greedy(feed)
  eng.set_compute_mode("f16x3")
  c, cr = eng.forward_greedy(feed)
  eng.close()
  assert a[1].shape[1] == 15
  assert (a[1] == c[1]).all() and (ar[1] == cr[1]).all()       # deterministic
  assert np.abs(a[1] - b[1]).max() < TOL and np.abs(ar[1] - br[1]).max() < TOL


def test_f16x3_refuses_weights_whose_transformed_planes_leave_the_fp16_range(built_lib):
  """The f16x3 range guard covers the TRANSFORMED kernel planes of the Winograd packs: three
  same-sign taps of |w| = 180 in one stencil column are inside the scaled fp16 range one by one
  (256 * 180 = 46 080) but (g0 + g1 + g2) / 2 = 270 is not (69 120 > 65 504) -- the forward must
  refuse loudly, not run on infinities; the same weights spread over different columns run, and
  the fp32 matrix pipe takes either."""
  cfg = synth.default_config(batch_size=2, use_grids=(1, 0))
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 5)
  name = "person_pred/decoder_grid_class_0/decoder_rnn/dec_grid_0/kernel"
  spread = synth.make_params(cfg)
  spread[name] = spread[name].copy()
  for ky in range(3):
    spread[name][ky, ky, 40 + ky, 7] = 180.0            # three different columns: |U| <= 90
  stacked = synth.make_params(cfg)
  stacked[name] = stacked[name].copy()
  stacked[name][:, 1, 40, 7] = 180.0                    # one column: (g0 + g1 + g2) / 2 = 270
  eng = _engine(built_lib, cfg, spread, "f16x3")
  eng.forward_greedy(feed)
  eng.close()
  eng = _engine(built_lib, cfg, stacked, "f32")
  eng.forward_greedy(feed)
  eng.close()
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(stacked)
  with pytest.raises(Exception) as err:
    eng.set_compute_mode("f16x3")
    eng.forward_greedy(feed)
  assert "transformed kernel planes" in str(err.value)
  eng.close()


def test_model_falls_back_to_f32_when_the_f16x3_range_guard_would_trip(built_lib, caplog):
  """The host mirror (pred_models.Model.load_params) applies the engine's range bound to the
  checkpoint it is handed and, instead of letting the DEFAULT compute mode raise at the first
  forward, switches that model to the fp32 matrix pipe with a warning -- the pattern it already
  follows for toy grids.  The engine itself stays loud (test above)."""
  import logging
  from multiverse_amd import pred_models
  cfg = synth.default_config(batch_size=2, use_grids=(1, 0))
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 5)
  name = "person_pred/decoder_grid_class_0/decoder_rnn/dec_grid_0/kernel"
  stacked = synth.make_params(cfg)
  stacked[name] = stacked[name].copy()
  stacked[name][:, 1, 40, 7] = 180.0                    # (g0 + g1 + g2) / 2 = 270 in one column
  assert pred_models.f16x3_out_of_range(synth.make_params(cfg)) is None
  bad = pred_models.f16x3_out_of_range(stacked)
  assert bad is not None and bad[0] == name and abs(bad[2] - 270.0) < 1.0
  model = pred_models.get_model(cfg, 0)
  assert model.compute_mode == "f16x3"
  with caplog.at_level(logging.WARNING, logger="multiverse_amd"):
    model.load_params(stacked)
  assert model.compute_mode == "f32"
  assert any("overridden to f32" in r.getMessage() for r in caplog.records)
  cls, reg, _ = model.run_forward(feed)
  ref = _engine(built_lib, cfg, stacked, "f32")
  rc, rr = ref.forward_greedy(feed)
  ref.close()
  model.close()
  assert (np.asarray(cls[0]) == rc[0]).all() and (np.asarray(reg[0]) == rr[0]).all()


def test_training_repacks_that_leave_the_fp16_range_fail_loudly(built_lib):
  """The f16x3 packs of a TRAINING engine are rebuilt from the device weights after every
  optimizer step (direct and Winograd forms, forward and dgrad), where no host copy exists for
  ensure_packed16's range check: the pack kernels flag a scaled value at or beyond the guard
  and mv_train_step / mv_train_apply fail at their closing synchronisation -- instead of the
  next step running on fp16 infinities.  Forced here by one momentum step at a learning rate
  of 1e7 (clipped gradients of up to 10 move every weight far outside 60 000 / 256)."""
  cfg = synth.default_config(batch_size=2, use_grids=(0, 1), is_train=True,
                             optimizer="momentum", init_lr=1e7)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 6)
  eng = _engine(built_lib, cfg, synth.make_params(cfg), "f16x3")
  eng.train_init()
  with pytest.raises(Exception) as err:
    eng.train_step(feed)
  assert "left the scaled fp16 range" in str(err.value)
  eng.close()
  # a sane step on a fresh engine afterwards: the flag was cleared
  cfg2 = synth.default_config(batch_size=2, use_grids=(0, 1), is_train=True)
  eng = _engine(built_lib, cfg2, synth.make_params(cfg2), "f16x3")
  eng.train_init()
  loss, _, _ = eng.train_step(feed)
  assert np.isfinite(loss)
  eng.close()
