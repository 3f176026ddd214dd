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


@pytest.mark.parametrize("mode", MODES)
def test_scene_conv_kernel_1_is_a_dense_projection_on_mfma(built_lib, mode):
  """--scene_conv_kernel 1 (code/train.py:65): the scene stack as two strided 1x1
  projections, run as fp32 MFMA GEMMs (scene_proj1x1_mfma_kernel); forward parity and one
  training step's gradients of the projection weights."""
  import torch
  cfg = synth.default_config(batch_size=3, use_grids=(1, 1), scene_conv_kernel=1)
  params = synth.make_params(cfg, recurrent_gain=2.0, bias_scale=0.1)
  assert params["person_pred/scene_conv1/W"].shape == (1, 1, 11, 64)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 66)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode(mode)
  eng.set_profiling(True)
  cls, reg = eng.forward_greedy(feed)
  stats = eng.kernel_stats()
  eng.set_profiling(False)
  eng.close()
  assert stats["scene_proj1x1_mfma"]["launches"] == 2 and "scene_conv_s2_tanh" not in stats
  ocls, oreg, _ = oracle.forward(params, cfg, feed)
  _check(cfg, cls, reg, ocls, oreg, cfg.pred_len)
  tcfg = synth.default_config(batch_size=2, use_grids=(0, 1), scene_conv_kernel=1, is_train=True)
  tparams = synth.make_params(tcfg, recurrent_gain=2.0, bias_scale=0.1)
  tfeed = synth.make_feed(tcfg, seed=synth.SEED_BASE + 67)
  eng = built_lib.Engine(tcfg, device=0)
  eng.set_params(tparams)
  eng.set_compute_mode(mode)
  eng.train_init()
  loss, _, _ = eng.train_forward_backward(tfeed)
  ol, _, _, og = oracle.loss_and_grads(tparams, tcfg, tfeed, dtype=torch.float64)
  assert abs(loss - ol) < 1e-4 * abs(ol)
  for n in ("person_pred/scene_conv1/W", "person_pred/scene_conv2/W", "person_pred/scene_conv2/b"):
    g = eng.get_grad(n)
    assert np.abs(g - og[n]).max() <= 2e-3 * max(np.abs(og[n]).max(), 1e-30), n
  eng.close()


def test_second_engine_with_smaller_beam_does_not_shrink_the_lds_limit(built_lib):
  """hipFuncAttributeMaxDynamicSharedMemorySize of the beam step is process
  wide; a later, smaller engine / mv_op_beam_step must not lower it under a live one."""
  big = synth.default_config(batch_size=1, use_grids=(1, 0), beam_size=20)
  params = synth.make_params(big, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(big, seed=synth.SEED_BASE + 68)
  e1 = built_lib.Engine(big, device=0)
  e1.set_params(params)
  a, _ = e1.forward_beam(feed)
  small = synth.default_config(batch_size=1, use_grids=(0, 1), beam_size=2)
  e2 = built_lib.Engine(small, device=0)          # 2 * 144 candidates: a much smaller LDS ask
  rng = np.random.default_rng(0)
  built_lib.op_beam_step(rng.normal(size=(1, 2, 16)).astype("float32"),
                         np.zeros((1, 2), "float32"), 2, False, 1.0, 0)
  b, _ = e1.forward_beam(feed)                    # 94 KB of LDS again
  e1.close()
  e2.close()
  for k in a:
    assert (a[k] == b[k]).all(), k


_TILED_FIRST_STEP = r"""
import sys
import numpy as np
sys.path.insert(0, %(root)r)
from multiverse_amd import _lib, synth
cfg = synth.default_config(batch_size=3, use_grids=(1, 0), beam_size=5)
cfg.diverse_beam = True
for gnn in (True, False):
  cfg.use_gnn = gnn
  params = synth.make_params(cfg, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 71)
  eng = _lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode("f16x3")
  arrs, s = eng.forward_beam(feed)
  eng.close()
  np.savez(%(out)r %% int(gnn), **{k: np.asarray(v) for k, v in arrs.items()})
"""


def test_shared_first_beam_step_is_bitwise_the_tiled_one(built_lib, tmp_path):
  """The first beam-decoder step runs once per sample (the B tiled rows are identical,
  code/pred_models.py:497-502): every output equals, bit for bit, the run that tiles the
  state first (MV_BEAM_SHARED_FIRST=0, read once per process -> a subprocess)."""
  import os
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  out = str(tmp_path / "tiled_%d.npz")
  env = dict(os.environ, MV_BEAM_SHARED_FIRST="0")
  subprocess.check_call([sys.executable, "-c", _TILED_FIRST_STEP % dict(root=root, out=out)],
                        env=env)
  cfg = synth.default_config(batch_size=3, use_grids=(1, 0), beam_size=5)
  cfg.diverse_beam = True
  for gnn in (True, False):
    cfg.use_gnn = gnn
    params = synth.make_params(cfg, recurrent_gain=3.0, bias_scale=0.1)
    feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 71)
    eng = built_lib.Engine(cfg, device=0)
    eng.set_params(params)
    eng.set_compute_mode("f16x3")
    eng.set_profiling(True)
    arrs, s = eng.forward_beam(feed)
    stats = eng.kernel_stats()
    eng.close()
    assert "beam_tile_state" not in stats
    tiled = np.load(out % int(gnn))
    for k, v in arrs.items():
      assert (np.asarray(v) == tiled[k]).all(), (gnn, k)


_DENSE_X = r"""
import sys
import numpy as np
sys.path.insert(0, %(root)r)
from multiverse_amd import _lib, synth
out = {}
for mode in ("f16x3", "bf16"):
  cfg = synth.default_config(batch_size=3, use_grids=(1, 1))
  params = synth.make_params(cfg, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 72)
  eng = _lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode(mode)
  cls, reg = eng.forward_greedy(feed)
  eng.close()
  for s in range(2):
    out["%%s_cls%%d" %% (mode, s)] = cls[s]
    out["%%s_reg%%d" %% (mode, s)] = reg[s]
np.savez(%(out)r, **out)
"""


def test_sparse_x_table_terms_equal_the_dense_x_operand(built_lib, tmp_path):
  """The class chains' x operands as epilogue table terms (sparse_x.h) against the same
  engine multiplying the dense operand (MV_SPARSE_X=0, read once per process): fp32-class
  agreement in f16x3, the same argmax everywhere; the x k-steps are gone from the FLOPs the
  launches report."""
  import os
  import subprocess
  import sys
  root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
  out = str(tmp_path / "dense_x.npz")
  subprocess.check_call([sys.executable, "-c", _DENSE_X % dict(root=root, out=out)],
                        env=dict(os.environ, MV_SPARSE_X="0"))
  dense = np.load(out)
  for mode, tol in (("f16x3", 2e-5), ("bf16", 3e-2)):
    cfg = synth.default_config(batch_size=3, use_grids=(1, 1))
    params = synth.make_params(cfg, recurrent_gain=3.0, bias_scale=0.1)
    feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 72)
    eng = built_lib.Engine(cfg, device=0)
    eng.set_params(params)
    eng.set_compute_mode(mode)
    eng.set_profiling(True)
    cls, reg = eng.forward_greedy(feed)
    stats = eng.kernel_stats()
    eng.close()
    assert "sx_encoder_corr" in stats and "enc_class_input" not in stats
    assert stats["convlstm_step"]["flops"] < 0.93 * stats["convlstm_step"]["flops_dense"]
    worst, flipped = 0.0, 0
    for s in range(2):
      ref = dense["%s_cls%d" % (mode, s)]
      scale = float(np.abs(ref).max())
      gi = cls[s].reshape(3, cfg.pred_len, -1).argmax(-1)
      ri = ref.reshape(3, cfg.pred_len, -1).argmax(-1)
      if mode == "f16x3":
        assert (gi == ri).all()
      # bf16 carries ~1e-2 relative noise per logit: where two runs of it pick different cells
      # the decoder feedback sends the trajectories apart, so a row is compared up to and
      # including its first differing step (the argmax audit of the mode itself is
      # tests/test_gpu_bf16.py)
      for n in range(3):
        bad = np.nonzero(gi[n] != ri[n])[0]
        upto = (bad[0] + 1) if bad.size else cfg.pred_len
        flipped += int(bad.size > 0)
        d = float(np.abs(cls[s][n, :upto] - ref[n, :upto]).max())
        worst = max(worst, d / scale)
        assert d <= tol * scale, (mode, s, n, d, scale)
      # the regression chain is not fed by the class decoder's choices
      assert np.abs(reg[s] - dense["%s_reg%d" % (mode, s)]).max() <= tol * max(
          1.0, float(np.abs(dense["%s_reg%d" % (mode, s)]).max()))
    # (how many of the six bf16 rows meet a near-tie depends on the last bits of the fp32 sums
    # around the matrix products: 1 with the second graph-attention kernel, 2 with the third;
    # each is held to the tolerance up to and including its first differing step)
    assert flipped <= (0 if mode == "f16x3" else 3), (mode, flipped)
    print("%s: sparse-x vs dense-x class logits, max rel diff %.2e, rows branched %d"
          % (mode, worst, flipped))


@pytest.mark.parametrize("mode", ["f32", "f16x3"])
def test_literal_baseline_grids_36x18_and_18x9(built_lib, mode):
  """BASELINE.json words the grids as 36x18 / 18x9 (648 / 162 cells over 72x36 scene maps);
  the reference's own are 18x32 / 9x16.  W = 18 and 9 do not divide 32:
  the gate kernels run their generic (every-tap-loaded) template instance, image rows end
  inside the 32-cell operand tiles, the attention falls to 1-D groups that straddle rows.
  Same bars as everywhere: argmax exact, 1e-4."""
  cfg = synth.default_config(batch_size=2, use_grids=(1, 1), scene_h=72, scene_w=36,
                             scene_grids=[(36, 18), (18, 9)])
  params = synth.make_params(cfg, recurrent_gain=2.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 5)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode(mode)
  cls, reg = eng.forward_greedy(feed)
  eng.close()
  ocls, oreg, _ = oracle.forward(params, cfg, feed)
  for s in range(2):
    assert cls[s].shape == ocls[s].shape
    assert (cls[s].reshape(2, 12, -1).argmax(-1) == ocls[s].reshape(2, 12, -1).argmax(-1)).all()
    assert np.abs(cls[s] - ocls[s]).max() < 1e-4 and np.abs(reg[s] - oreg[s]).max() < 1e-4


@pytest.mark.parametrize("E", [128, 64])
def test_emb_size_other_than_32(built_lib, E):
  """--emb_size (code/train.py:53, flag default 128; the published runs use 32): the
  decoders' x operand, the decode tail, the sparse-x tables and the training step must not be
  tied to one embedding width.  Greedy both scales (f16x3 and f32) and diverse beam search
  against the oracle; every gradient of one training step against the fp64 oracle."""
  import torch
  from oracle import multiverse_oracle as oracle
  cfg = synth.default_config(batch_size=2, use_grids=(1, 1), emb_size=E)
  params = synth.make_params(cfg, seed=synth.SEED_BASE + E, recurrent_gain=3.0, bias_scale=0.1)
  assert params["person_pred/decoder_grid_class_0/decoder_rnn/dec_grid_0/kernel"].shape == \
      (3, 3, E + 256, 1024)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 11)
  ocls, oreg, _ = oracle.forward(params, cfg, feed)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  for mode in ("f16x3", "f32"):
    eng.set_compute_mode(mode)
    cls, reg = eng.forward_greedy(feed)
    for s in range(2):
      dc, dr = np.abs(cls[s] - ocls[s]).max(), np.abs(reg[s] - oreg[s]).max()
      same = (cls[s].reshape(2, 12, -1).argmax(-1) == ocls[s].reshape(2, 12, -1).argmax(-1)).all()
      print("emb_size %d %-5s scale %d: max|dcls| %.3g max|dreg| %.3g" % (E, mode, s, dc, dr))
      assert dc < 1e-4 and dr < 1e-4 and same
  eng.close()
  # diverse beam search, scale 1
  from beam_compare import compare_beams
  bcfg = synth.default_config(batch_size=2, use_grids=(0, 1), beam_size=5, emb_size=E)
  bparams = synth.make_params(bcfg, seed=synth.SEED_BASE + E, recurrent_gain=3.0, bias_scale=0.1)
  bfeed = synth.make_feed(bcfg, seed=synth.SEED_BASE + 12)
  beng = built_lib.Engine(bcfg, device=0)
  beng.set_params(bparams)
  beng.set_compute_mode("f16x3")
  arrs, s = beng.forward_beam(bfeed)
  beng.close()
  trace = {}
  _, obreg, (ologits, oids, olp) = oracle.forward(bparams, bcfg, bfeed, trace=trace)
  assert s == 1
  compare_beams(arrs, obreg[1], ologits, oids, olp,
                np.stack(trace["beam_step_topvals"], axis=-1), trace["beam_trace"])
  # one training step: loss parts and every gradient tensor
  tcfg = synth.default_config(batch_size=2, use_grids=(1, 1), is_train=True, emb_size=E)
  tparams = synth.make_params(tcfg, seed=synth.SEED_BASE + E, recurrent_gain=2.0, bias_scale=0.1)
  tfeed = synth.make_feed(tcfg, seed=synth.SEED_BASE + 13)
  oloss, owd, opgl, _ = oracle.loss_and_grads(tparams, tcfg, tfeed)
  og64 = oracle.loss_and_grads(tparams, tcfg, tfeed, dtype=torch.float64)[3]
  for mode in ("f16x3", "f32"):
    teng = built_lib.Engine(tcfg, device=0)
    teng.set_params(tparams)
    teng.set_compute_mode(mode)
    teng.train_init()
    loss, wd, pgl = teng.train_forward_backward(tfeed)
    assert abs(loss - oloss) < 1e-4 * max(1.0, abs(oloss)) and np.allclose(pgl, opgl, rtol=1e-4, atol=1e-5)
    worst = 0.0
    for name, _ in teng.param_specs():
      g = teng.get_grad(name).astype("float64")
      worst = max(worst, float(np.abs(g - og64[name]).max() / max(np.abs(og64[name]).max(), 1e-30)))
    print("emb_size %d %-5s training step: worst gradient error %.2e of max|g|" % (E, mode, worst))
    teng.close()
    assert worst < 2e-3


def test_create_rejects_unsupported_shapes(built_lib):
  """Shape limits are reported by mv_create (Engine()), not by the first forward."""
  for kw, msg in ((dict(emb_size=48), "emb_size"), (dict(emb_size=16), "emb_size"),
                  (dict(enc_hidden_size=192, dec_hidden_size=192), "hidden_size"),
                  (dict(enc_hidden_size=1024, dec_hidden_size=1024), "hidden_size")):
    cfg = synth.default_config(batch_size=1, use_grids=(0, 1), **kw)
    with pytest.raises(built_lib.MvError, match=msg):
      built_lib.Engine(cfg, device=0)


@pytest.mark.parametrize("act,gain", [("relu", 1.0), ("lrelu", 1.0), ("relu", 8.0)])
def test_unbounded_activations_run_on_the_fp16_pipe(built_lib, act, gain):
  """--activation_func relu / lrelu (code/pred_utils.py:112-121): the embeddings that feed the
  gate convolutions are unbounded, so in f16x3 mode their operand planes carry a per-tensor
  exponent taken from max |x| (split_planes_dyn_kernel; the accumulators are rescaled once the
  x k-steps are done).  Greedy forward, both scales, against the fp64 oracle in f16x3 AND f32.
  gain 8 scales the regression embedding so that 256 x leaves the fp16 range (a fixed-scale
  plane would overflow to inf); such inputs drive the recurrence hard, so there the bar is
  "as close to fp64 as the fp32 matrix pipe is" (x 3), not an absolute one."""
  import torch
  from oracle import multiverse_oracle as oracle
  cfg = synth.default_config(batch_size=2, use_grids=(1, 1), activation_func=act)
  params = synth.make_params(cfg, seed=synth.SEED_BASE + 21, recurrent_gain=2.0, bias_scale=0.1)
  for s in (0, 1):
    k = "person_pred/decoder_grid_reg_%d/decoder_rnn/grid_emb/W" % s
    params[k] = (params[k] * gain).astype("float32")
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 22)
  ocls, oreg, _ = oracle.forward(params, cfg, feed, dtype=torch.float64)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  err = {}
  for mode in ("f16x3", "f32"):
    eng.set_compute_mode(mode)
    cls, reg = eng.forward_greedy(feed)
    for s in range(2):
      assert np.isfinite(cls[s]).all() and np.isfinite(reg[s]).all()
      dc = float(np.abs(cls[s] - np.asarray(ocls[s])).max())
      dr = float(np.abs(reg[s] - np.asarray(oreg[s])).max())
      err[mode, s] = (dc, dr)
      print("%s gain %g %-5s scale %d: max|dcls| %.3g max|dreg| %.3g (ranges %.3g, %.3g)"
            % (act, gain, mode, s, dc, dr, np.abs(ocls[s]).max(), np.abs(oreg[s]).max()))
  eng.close()
  for s in range(2):
    for k in (0, 1):
      bar = 1e-4 if gain == 1.0 else max(1e-4, 3.0 * err["f32", s][k])
      assert err["f16x3", s][k] < bar, (s, k, err)
      if gain == 1.0:
        assert err["f32", s][k] < 1e-4


@pytest.mark.parametrize("Ch,grids", [(128, (1, 1)), (512, (0, 1))])
def test_hidden_size_other_than_256(built_lib, Ch, grids):
  """--enc_hidden_size / --dec_hidden_size (code/train.py:54-57; the published runs use 256):
  128 and 512 through every kernel of the path -- gate convolutions (fp32 pipe, f16x3 direct
  and Winograd forms), graph attention, hidden2grid, beam search, and one training step
  (dgrad, wgrad, graph-attention backward) -- against the oracle.  Scale 1 only at 512 (the
  CPU oracle's cost grows with the square of the width)."""
  import torch
  from oracle import multiverse_oracle as oracle
  from beam_compare import compare_beams
  kw = dict(enc_hidden_size=Ch, dec_hidden_size=Ch)
  cfg = synth.default_config(batch_size=2, use_grids=grids, **kw)
  params = synth.make_params(cfg, seed=synth.SEED_BASE + Ch, recurrent_gain=3.0, bias_scale=0.1)
  s_on = [s for s in range(2) if grids[s]]
  kname = "person_pred/encoder_grid_class_%d/enc_grid_%d/kernel" % (s_on[0], s_on[0])
  assert params[kname].shape == (3, 3, 64 + Ch, 4 * Ch)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 31)
  ocls, oreg, _ = oracle.forward(params, cfg, feed)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  for mode in ("f16x3", "f32"):
    eng.set_compute_mode(mode)
    cls, reg = eng.forward_greedy(feed)
    for s in s_on:
      dc, dr = np.abs(cls[s] - ocls[s]).max(), np.abs(reg[s] - oreg[s]).max()
      same = (cls[s].reshape(2, 12, -1).argmax(-1) == ocls[s].reshape(2, 12, -1).argmax(-1)).all()
      print("hidden %d %-5s scale %d: max|dcls| %.3g max|dreg| %.3g" % (Ch, mode, s, dc, dr))
      assert dc < 1e-4 and dr < 1e-4 and same
  eng.close()
  # diverse beam search, scale 1
  bcfg = synth.default_config(batch_size=2, use_grids=(0, 1), beam_size=5, **kw)
  bparams = synth.make_params(bcfg, seed=synth.SEED_BASE + Ch, recurrent_gain=3.0, bias_scale=0.1)
  bfeed = synth.make_feed(bcfg, seed=synth.SEED_BASE + 32)
  beng = built_lib.Engine(bcfg, device=0)
  beng.set_params(bparams)
  beng.set_compute_mode("f16x3")
  arrs, s = beng.forward_beam(bfeed)
  beng.close()
  trace = {}
  _, obreg, (ologits, oids, olp) = oracle.forward(bparams, bcfg, bfeed, trace=trace)
  compare_beams(arrs, obreg[1], ologits, oids, olp,
                np.stack(trace["beam_step_topvals"], axis=-1), trace["beam_trace"])
  # one training step: loss parts and every gradient tensor, scale 1
  tcfg = synth.default_config(batch_size=2, use_grids=(0, 1), is_train=True, **kw)
  tparams = synth.make_params(tcfg, seed=synth.SEED_BASE + Ch, recurrent_gain=2.0, bias_scale=0.1)
  tfeed = synth.make_feed(tcfg, seed=synth.SEED_BASE + 33)
  oloss, owd, opgl, _ = oracle.loss_and_grads(tparams, tcfg, tfeed)
  og64 = oracle.loss_and_grads(tparams, tcfg, tfeed, dtype=torch.float64)[3]
  for mode in ("f16x3", "f32"):
    teng = built_lib.Engine(tcfg, device=0)
    teng.set_params(tparams)
    teng.set_compute_mode(mode)
    teng.train_init()
    loss, wd, pgl = teng.train_forward_backward(tfeed)
    assert abs(loss - oloss) < 1e-4 * max(1.0, abs(oloss)) and np.allclose(pgl, opgl, rtol=1e-4, atol=1e-5)
    worst = 0.0
    for name, _ in teng.param_specs():
      g = teng.get_grad(name).astype("float64")
      worst = max(worst, float(np.abs(g - og64[name]).max() / max(np.abs(og64[name]).max(), 1e-30)))
    print("hidden %d %-5s training step: worst gradient error %.2e of max|g|" % (Ch, mode, worst))
    # two optimizer steps run (device-side repacks of every weight form)
    teng.train_apply(1.0)
    teng.train_step(tfeed)
    teng.close()
    assert worst < 2e-3
