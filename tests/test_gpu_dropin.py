# coding=utf-8
"""GPU: the drop-in boundary (get_model / Tester.step / evaluate) and the
committed golden fixtures -- these do not need the oracle at run time, only
the frozen numbers in tests/golden/."""
import os

import numpy as np
import pytest

from multiverse_amd import pred_models, pred_utils, synth

from beam_compare import compare_beams

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = 1e-4


def _engine_for_golden(built_lib, name, cfg):
  g = np.load(os.path.join(GOLD, name))
  params = synth.make_params(cfg, seed=int(g["seed"][0]),
                             recurrent_gain=float(g["gain"][0]),
                             bias_scale=float(g["bias"][0]))
  feed = synth.make_feed(cfg, seed=int(g["seed"][0]))
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  return g, eng, feed


def test_golden_greedy_cfg1(built_lib):
  cfg = synth.default_config(batch_size=4, use_grids=(1, 0))
  g, eng, feed = _engine_for_golden(built_lib, "golden_greedy_cfg1.npz", cfg)
  cls, reg = eng.forward_greedy(feed)
  eng.close()
  ids = cls[0].reshape(4, 12, -1).argmax(-1)
  assert (ids == g["ids_0"]).all()                      # bit-exact argmax
  assert np.abs(cls[0] - g["cls_0"]).max() < TOL
  assert np.abs(reg[0] - g["reg_0"]).max() < TOL


def test_golden_greedy_both_scales(built_lib):
  cfg = synth.default_config(batch_size=2, use_grids=(1, 1))
  g, eng, feed = _engine_for_golden(built_lib, "golden_greedy_both.npz", cfg)
  cls, reg = eng.forward_greedy(feed)
  eng.close()
  for s in (0, 1):
    assert (cls[s].reshape(2, 12, -1).argmax(-1) == g["ids_%d" % s]).all()
    assert np.abs(cls[s] - g["cls_%d" % s]).max() < TOL
    assert np.abs(reg[s] - g["reg_%d" % s]).max() < TOL


def test_golden_beam(built_lib):
  cfg = synth.default_config(batch_size=2, use_grids=(0, 1), beam_size=5)
  g, eng, feed = _engine_for_golden(built_lib, "golden_beam_s1.npz", cfg)
  arrs, s = eng.forward_beam(feed)
  eng.close()
  assert s == 1
  compare_beams(arrs, g["reg_1"], g["beam_logits"], g["beam_ids"], g["beam_logprobs"],
                g["beam_topvals"], g["beam_trace"])


def test_tester_step_and_evaluate_through_the_boundary(built_lib):
  """test.py's flow: read_data -> get_model -> Tester -> evaluate."""
  cfg = synth.default_config(batch_size=4, use_grids=(1, 1))
  data = synth.make_npz_data(cfg, 6, seed=9)
  ds = pred_utils.dataset_from_npz_dict(data, "test", cfg)
  model = pred_models.get_model(cfg, 0)
  model.load_params(synth.make_params(cfg, recurrent_gain=3.0, bias_scale=0.1))
  tester = pred_models.Tester(model, cfg, sess=None)
  batch = next(ds.get_batches(4, full=True, shuffle=False))
  cls, reg, beam = tester.step(None, batch)
  assert beam is None
  assert cls[0].shape == (4, 12, 18, 32, 1) and reg[1].shape == (4, 12, 9, 16, 2)
  p = pred_utils.evaluate(ds, cfg, None, tester)
  for j in (0, 1):
    assert 0.0 <= p["grid%d_acc" % j] <= 1.0
    assert np.isfinite(p["grid%d_traj_ade" % j]) and p["grid%d_traj_fde" % j] > 0
  # weights round-trip through the engine (Saver role)
  back = model.get_params()
  ref = synth.make_params(cfg, recurrent_gain=3.0, bias_scale=0.1)
  assert all((back[k] == ref[k]).all() for k in ref)
  model.close()


def test_abandoned_steps_generator_leaves_no_stale_batches(built_lib):
  """Tester.steps pipelines feed / kernels / fetch over the engine's submit / collect slots.
  A consumer that stops early (break, an exception in its loop body) must not leave
  submissions behind: the next pass over the data would collect THOSE outputs and pair
  them with the wrong batch (or die with `pipeline full`)."""
  cfg = synth.default_config(batch_size=2, use_grids=(1, 1))
  data = synth.make_npz_data(cfg, 8, seed=11)
  ds = pred_utils.dataset_from_npz_dict(data, "test", cfg)
  model = pred_models.get_model(cfg, 0)
  model.load_params(synth.make_params(cfg, recurrent_gain=3.0, bias_scale=0.1))
  tester = pred_models.Tester(model, cfg, sess=None)
  batches = list(ds.get_batches(2, full=True, shuffle=False))
  assert len(batches) == 4
  want = [tester.step(None, bt) for bt in batches]
  gen = tester.steps(None, batches[::-1], depth=2)       # reversed order, abandoned after one
  first = next(gen)
  assert first[0] is batches[-1]
  gen.close()                                            # GeneratorExit -> drains the pipeline
  with pytest.raises(RuntimeError, match="consumer"):
    for _ in tester.steps(None, batches, depth=2):
      raise RuntimeError("consumer failed")
  got = list(tester.steps(None, batches, depth=2))
  assert len(got) == 4
  for (bt, (cls, reg, beam)), w, b0 in zip(got, want, batches):
    assert bt is b0 and beam is None
    for s in (0, 1):
      assert (cls[s] == w[0][s]).all() and (reg[s] == w[1][s]).all()
  model.close()


def test_toy_grids_select_the_fp32_pipe(built_lib):
  """Grids of fewer than 32 cells cannot run the fp16-pipe kernels (a 32-cell wave tile would
  span more than two images); Model picks the fp32 MFMA path for them instead of failing at
  the first forward, the C ABI keeps refusing an explicit f16x3 request."""
  cfg = synth.default_config(batch_size=2, use_grids=(0, 1), scene_h=12, scene_w=20,
                             scene_grids=[(6, 10), (3, 5)])     # scale 1: 15 cells
  model = pred_models.get_model(cfg, 0)
  assert model.compute_mode == "f32"
  params = synth.make_params(cfg, recurrent_gain=3.0, bias_scale=0.1)
  model.load_params(params)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 3)
  cls, reg = model.engine.forward_greedy(feed)
  from oracle import multiverse_oracle as oracle
  ocls, oreg, _ = oracle.forward(params, cfg, feed)
  assert np.abs(cls[1] - ocls[1]).max() < 1e-4 and np.abs(reg[1] - oreg[1]).max() < 1e-4
  model.engine.set_compute_mode("f16x3")
  with pytest.raises(built_lib.MvError, match="at least 32 cells"):
    model.engine.forward_greedy(feed)
  model.close()


def test_full_size_properties_batch64(built_lib):
  """BASELINE configs[1] size (N=64, both scales): the oracle would take
  minutes, so check size-independent properties: (a) batch independence -- a
  sample's outputs equal its outputs inside a batch of 4; (b) determinism."""
  cfg = synth.default_config(batch_size=64, use_grids=(1, 1))
  params = synth.make_params(cfg, seed=5)
  feed = synth.make_feed(cfg, seed=5)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  cls, reg = eng.forward_greedy(feed)
  cls2, reg2 = eng.forward_greedy(feed)
  eng.close()
  for s in (0, 1):
    assert np.isfinite(cls[s]).all() and np.isfinite(reg[s]).all()
    assert (cls[s] == cls2[s]).all() and (reg[s] == reg2[s]).all()
  sub = dict(feed)
  lo = 20
  sub["obs_scene"] = feed["obs_scene"][lo:lo + 4]
  sub["grid_obs_labels"] = [a[lo:lo + 4] for a in feed["grid_obs_labels"]]
  sub["grid_obs_regress"] = [a[lo:lo + 4] for a in feed["grid_obs_regress"]]
  cfg4 = synth.default_config(batch_size=4, use_grids=(1, 1))
  eng4 = built_lib.Engine(cfg4, device=0)
  eng4.set_params(params)
  c4, r4 = eng4.forward_greedy(sub)
  eng4.close()
  for s in (0, 1):
    assert (c4[s] == cls[s][lo:lo + 4]).all()      # same kernels, same order: bitwise
    assert (r4[s] == reg[s][lo:lo + 4]).all()


def test_full_size_properties_batch256(built_lib):
  """north_star's "batch 256" (the `greedy_b256` line of bench.py), both scales, f16x3 and f32:
  (a) determinism; (b) batch independence at THAT size -- rows 64 .. 127 of the batch equal, bit
  for bit, the same 64 trajectories run as a batch of 64 (whose every row tests/
  test_gpu_at_size.py holds to the oracle): the 256-row launch tiles, maps to XCDs and groups
  its problems differently, the arithmetic per row must not notice."""
  cfg = synth.default_config(batch_size=256, use_grids=(1, 1))
  params = synth.make_params(cfg, seed=synth.SEED_BASE + 2, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 8)
  lo = 64
  sub = dict(feed)
  sub["obs_scene"] = feed["obs_scene"][lo:lo + 64]
  sub["grid_obs_labels"] = [a[lo:lo + 64] for a in feed["grid_obs_labels"]]
  sub["grid_obs_regress"] = [a[lo:lo + 64] for a in feed["grid_obs_regress"]]
  cfg64 = synth.default_config(batch_size=64, use_grids=(1, 1))
  for mode in ("f16x3", "f32"):
    eng = built_lib.Engine(cfg, device=0)
    eng.set_params(params)
    eng.set_compute_mode(mode)
    cls, reg = eng.forward_greedy(feed)
    cls2, reg2 = eng.forward_greedy(feed)
    eng.close()
    eng64 = built_lib.Engine(cfg64, device=0)
    eng64.set_params(params)
    eng64.set_compute_mode(mode)
    c64, r64 = eng64.forward_greedy(sub)
    eng64.close()
    for s in (0, 1):
      assert np.isfinite(cls[s]).all() and np.isfinite(reg[s]).all()
      assert (cls[s] == cls2[s]).all() and (reg[s] == reg2[s]).all()
      assert (c64[s] == cls[s][lo:lo + 64]).all(), (mode, s)
      assert (r64[s] == reg[s][lo:lo + 64]).all(), (mode, s)


def test_compact_inputs_are_bit_identical(built_lib):
  """SURVEY 8f N3 (device-side batch assembly): labels + one (x, y) per step +
  uint8 masks, expanded in HBM, against the dense upload of the same batch --
  greedy outputs, beam outputs and one training step, bit for bit; then the same
  through Tester.step with config.compact_inputs on a reference-style npz."""
  cfg = synth.default_config(batch_size=3, use_grids=(1, 1))
  params = synth.make_params(cfg, seed=synth.SEED_BASE + 4)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 31)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  cls, reg = eng.forward_greedy(feed)
  ccls, creg = eng.forward_greedy_compact(dict(feed, num_rows=3))
  for s in range(2):
    assert (cls[s] == ccls[s]).all() and (reg[s] == creg[s]).all()
  # padded rows: maps zero, as rows beyond len(data) in the reference feed
  dense2 = dict(feed)
  dense2["grid_obs_regress"] = [a.copy() for a in feed["grid_obs_regress"]]
  for a in dense2["grid_obs_regress"]:
    a[2:] = 0.0
  cls2, reg2 = eng.forward_greedy(dense2)
  ccls2, creg2 = eng.forward_greedy_compact(dict(feed, num_rows=2))
  for s in range(2):
    assert (cls2[s] == ccls2[s]).all() and (reg2[s] == creg2[s]).all()
  eng.close()

  bcfg = synth.default_config(batch_size=2, use_grids=(1, 0), use_beam_search=True,
                              beam_size=5, diverse_beam=True)
  beng = built_lib.Engine(bcfg, device=0)
  beng.set_params(synth.make_params(bcfg, seed=synth.SEED_BASE + 4))
  bfeed = synth.make_feed(bcfg, seed=synth.SEED_BASE + 32)
  a, _ = beng.forward_beam(bfeed)
  b, _ = beng.forward_beam_compact(bfeed)
  for k in a:
    assert (a[k] == b[k]).all(), k
  beng.close()

  tcfg = synth.default_config(batch_size=2, use_grids=(1, 1), is_train=True)
  tparams = synth.make_params(tcfg, seed=synth.SEED_BASE + 4)
  tfeed = synth.make_feed(tcfg, seed=synth.SEED_BASE + 33)
  out = []
  for compact in (False, True):
    e = built_lib.Engine(tcfg, device=0)
    e.set_params(tparams)
    e.train_init()
    out.append((e.train_step_compact(tfeed) if compact else e.train_step(tfeed),
                e.get_param("person_pred/scene_conv1/W")))
    e.close()
  assert out[0][0] == out[1][0]
  assert (out[0][1] == out[1][1]).all()

  # through the host mirror: Tester.step on a reference-style dataset
  cfg4 = synth.default_config(batch_size=4, use_grids=(1, 1))
  data = synth.make_npz_data(cfg4, 6, seed=9, float32_traj=True)
  ds = pred_utils.dataset_from_npz_dict(data, "test", cfg4)
  model = pred_models.get_model(cfg4, 0)
  model.load_params(synth.make_params(cfg4, seed=synth.SEED_BASE + 4))
  tester = pred_models.Tester(model, cfg4)
  res = []
  for compact in (False, True):
    cfg4.compact_inputs = compact
    res.append([tester.step(None, bt) for bt in ds.get_batches(4, full=True, shuffle=False)])
  for (c0, r0, _), (c1, r1, _) in zip(*res):
    for s in range(2):
      assert (c0[s] == c1[s]).all() and (r0[s] == r1[s]).all()
  model.close()

  # and Trainer.step: two steps over the same batches, dense vs compact feeds
  tr = []
  for compact in (False, True):
    cfg5 = synth.default_config(batch_size=4, use_grids=(1, 1), is_train=True)
    cfg5.compact_inputs = compact
    cfg5.train_num_examples = 6
    ds5 = pred_utils.dataset_from_npz_dict(data, "train", cfg5)
    m5 = pred_models.get_model(cfg5, 0)
    m5.load_params(synth.make_params(cfg5, seed=synth.SEED_BASE + 4))
    trainer = pred_models.Trainer(m5, cfg5)
    losses = [trainer.step(None, bt)
              for bt in ds5.get_batches(4, num_steps=2, full=True, shuffle=False)]
    tr.append((losses, m5.get_params()["person_pred/scene_conv2/W"]))
    m5.close()
  assert [l[0] for l in tr[0][0]] == [l[0] for l in tr[1][0]]
  assert (tr[0][1] == tr[1][1]).all()
