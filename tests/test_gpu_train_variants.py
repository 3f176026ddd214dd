# coding=utf-8
"""GPU: the switches of the training / decoding path beyond the published run, through the
C ABI, against (a) the frozen runs of the reference's own Trainer on the TF-1 shim
(tests/golden/golden_shim_variant_*.npz) and (b) the CPU oracle:

  --use_soft_grid_class (+ --soft_grid 1 / 4 / 7)   code/pred_models.py:974-990, 1077-1124
  --mask_grid_regression                            :999-1018
  --use_teacher_forcing (training AND test time)    :283-285, 388-406
  training without --train_w_onehot                 :285, 427-435 (differentiable feedback)
  --keep_prob < 1 (DropoutWrapper input dropout)    :130-132, 194-202, 241-249
  --optimizer momentum / adam / rmsprop             :1667-1681
  --use_cosine_lr                                   :1646-1654
  --scene_conv_kernel 1                             code/train.py:65
  --emb_size 128, --enc/dec_hidden_size 128 / 512   code/train.py:53-57
  --activation_func relu / lrelu                    code/train.py:58-59, code/pred_utils.py:86-94
                                                    (f16x3: per-tensor exponent of the x planes)

Same bars as the published path: losses 1e-4 relative, every gradient tensor within 2e-3
of its max (measured ~1e-6), parameters after the optimizer step(s).
"""
import numpy as np
import pytest
import torch

from multiverse_amd import synth
from oracle import multiverse_oracle as oracle

import shim_golden as sg

pytestmark = pytest.mark.gpu
LOSS_VARIANTS = ["soft1", "soft7_mask", "mask", "teacher", "teacher_soft4", "no_onehot",
                 "dropout07", "sck1",       # sck1: --scene_conv_kernel 1 (1x1 projections on MFMA)
                 "relu", "lrelu",           # --activation_func
                 "emb128", "hidden128", "hidden512"]   # --emb_size / --enc,dec_hidden_size
OPTIMIZERS = ["momentum", "rmsprop", "adam", "cosine"]     # cosine: --use_cosine_lr + momentum
SLOTS = {"momentum": 1, "rmsprop": 2, "adam": 2, "cosine": 1}


_ORACLE64 = {}


def _oracle64(key, params, cfg, feed):
  """fp64 oracle gradients, computed once per case (the f32 and f16x3 runs share them:
  they are 10-20 s of CPU autograd each)."""
  if key not in _ORACLE64:
    _ORACLE64[key] = oracle.loss_and_grads(params, cfg, feed, dtype=torch.float64)
  return _ORACLE64[key]


def _rel(a, b):
  a = np.asarray(a, dtype=np.float64)
  b = np.asarray(b, dtype=np.float64)
  return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.mark.parametrize("mode", ["f32", "f16x3"])
@pytest.mark.parametrize("name", LOSS_VARIANTS)
def test_loss_and_decoder_switches(built_lib, name, mode):
  g, cfg, params, feeds = sg.variant_case(name)
  feed = feeds[0]
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  # relu / lrelu in f16x3: the unbounded x operands of the gate convolutions carry a per-tensor
  # exponent (ConvLstm16Args::x_exp) -- same bars as every other switch
  eng.set_compute_mode(mode)
  eng.train_init()
  eng.set_dropout_seed(feed["dropout_seed"])
  loss, wd, pgl = eng.train_forward_backward(feed)
  ref = g["loss_0"]
  print("%s/%s: loss %.6f reference run %.6f parts %s / %s" % (name, mode, loss, ref[0], pgl,
                                                              ref[2:]))
  assert np.allclose([loss, wd] + pgl, ref, rtol=1e-4, atol=1e-5), (loss, ref)
  _, _, _, og64 = _oracle64(name, params, cfg, feed)
  worst = 0.0
  for n, _ in eng.param_specs():
    gr = eng.get_grad(n)
    e_s, e_a = sg.digest_err(gr, g["grad_0|%s" % n])      # the reference's own gradients
    e64 = _rel(gr, og64[n])                                # and the fp64 oracle, every element
    worst = max(worst, e_s, e64)
    assert e_s < 2e-3 and e_a < 2e-3 and e64 < 2e-3, (n, e_s, e_a, e64)
  print("  worst gradient error (of max|g|): %.2e" % worst)
  eng.train_apply(1.0)
  for n, _ in eng.param_specs():
    e_s, e_a = sg.digest_err(eng.get_param(n), g["param|%s" % n])
    assert e_s < 1e-4 and e_a < 1e-5, (n, e_s, e_a)
  eng.close()


@pytest.mark.parametrize("name", OPTIMIZERS)
def test_optimizers(built_lib, name):
  g, cfg, params, feeds = sg.variant_case(name)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.train_init()
  p, st = dict(params), oracle.optimizer_init(cfg, params)
  for step, feed in enumerate(feeds):
    loss, wd, pgl = eng.train_step(feed)
    ol, _, _, p, st, _ = oracle.train_step(p, st, step, cfg, feed)
    ref = g["loss_%d" % step]
    print("%s step %d: loss %.6f oracle %.6f reference run %.6f" % (name, step, loss, ol, ref[0]))
    tol = 2e-3 if (name == "adam" and step > 0) else 1e-4
    assert abs(loss - ref[0]) < tol * max(1.0, abs(ref[0]))
  lr = cfg.init_lr
  for n, _ in eng.param_specs():
    got = eng.get_param(n)
    if name == "adam":
      # Adam's step is ~lr * g / (|g| + 3e-7): wherever |g| sits at the fp32 noise floor
      # the sign is noise, so parameters agree to a fraction of lr, the moments tightly
      assert np.abs(got - p[n]).max() <= 2.0 * lr * len(feeds), n
      assert np.median(np.abs(got - p[n])) <= 1e-3 * lr, n
    else:
      e_s, e_a = sg.digest_err(got, g["param|%s" % n])
      upd = max(float(np.abs(p[n] - params[n]).max()), 1e-12)
      assert np.abs(got - p[n]).max() <= 5e-3 * upd, (n, np.abs(got - p[n]).max(), upd)
      assert e_s < 1e-4 and e_a < 1e-5, (n, e_s, e_a)
    for i in range(SLOTS[name]):
      sl = eng.get_opt_slot(n, i)
      tol = 2e-2 if name == "adam" else 5e-3
      assert _rel(sl, st[n][i]) < tol, (n, i, _rel(sl, st[n][i]))
  if name == "adam":
    assert np.allclose(eng.opt_scalars(), g["opt_scalars"], rtol=1e-6)
    assert np.allclose(eng.opt_scalars(), (0.9 ** 3, 0.999 ** 3), rtol=1e-6)
  assert eng.global_step == len(feeds)
  eng.close()


def test_unknown_optimizer_fails_like_the_reference(built_lib):
  cfg = synth.default_config(batch_size=2, use_grids=(0, 1), is_train=True)
  cfg.optimizer = "sgd"
  eng = built_lib.Engine(cfg, device=0)
  with pytest.raises(built_lib.MvError, match="Optimizer not implemented"):
    eng.train_init()
  eng.close()


@pytest.mark.parametrize("mode", ["f32", "f16x3"])
def test_test_time_teacher_forcing_feeds_raw_logits(built_lib, mode):
  g = sg.load("golden_shim_variant_teacher_test.npz")
  cfg = synth.default_config(batch_size=2, use_grids=(0, 1), use_teacher_forcing=True)
  params = synth.make_params(cfg, seed=sg.VARIANT_SEED + 1, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=sg.VARIANT_SEED + 1)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode(mode)
  cls, reg = eng.forward_greedy(feed)
  eng.close()
  print("test-time teacher forcing %s: max|dcls| %.2e max|dreg| %.2e"
        % (mode, np.abs(cls[1] - g["cls_1"]).max(), np.abs(reg[1] - g["reg_1"]).max()))
  assert np.abs(cls[1] - g["cls_1"]).max() < 1e-4
  assert np.abs(reg[1] - g["reg_1"]).max() < 1e-4
  # and it is NOT the one-hot feedback
  cfg0 = synth.default_config(batch_size=2, use_grids=(0, 1))
  eng0 = built_lib.Engine(cfg0, device=0)
  eng0.set_params(params)
  cls0, _ = eng0.forward_greedy(feed)
  eng0.close()
  assert np.abs(cls0[1][:, 1:] - cls[1][:, 1:]).max() > 1e-3


def test_dropout_mask_statistics_and_seed(built_lib):
  """The engine's dropout: different seeds give different losses, the same seed the same
  bits; the shared generator keeps ~keep_prob of the elements."""
  g, cfg, params, feeds = sg.variant_case("dropout07")
  feed = feeds[0]
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.train_init()
  out = []
  for seed in (1, 1, 2):
    eng.set_dropout_seed(seed)
    out.append(eng.train_forward_backward(feed)[0])
  eng.close()
  assert out[0] == out[1] and out[0] != out[2]
  m = oracle.dropout_keep_mask((1 << 20,), 0.7, 99, 3)
  assert abs(m.mean() - 0.7) < 2e-3


@pytest.mark.parametrize("mode", ["f32", "f16x3"])
def test_single_decoder(built_lib, mode):
  """--use_single_decoder (code/pred_models.py:274,287-296), both scales: offsets decoded
  from the class decoder's states by ONE kernel shared by the scales; the regression
  encoder's variables exist but are neither run nor trained.  Greedy forward and two
  training steps against the frozen reference runs, gradients also against the fp64 oracle."""
  g, (cfg, params, feed), (tcfg, tparams, feeds) = sg.single_decoder_case()
  eng = built_lib.Engine(cfg, device=0)
  assert sorted(n for n, _ in eng.param_specs()) == sorted(params)
  eng.set_params(params)
  eng.set_compute_mode(mode)
  eng.set_profiling(True)
  cls, reg = eng.forward_greedy(feed)
  stats = eng.kernel_stats()
  eng.close()
  # two chains per scale instead of four
  assert stats["convlstm_step"]["flops_dense"] < 0.55 * 2 * 2 * 20 * 2.0 * 9 * 288 * 1024 * (576 + 144)
  for s in range(2):
    K = cfg.scene_grids[s][0] * cfg.scene_grids[s][1]
    assert (cls[s].reshape(2, -1, K).argmax(-1) == g["cls_%d" % s].reshape(2, -1, K).argmax(-1)).all()
    assert np.abs(cls[s] - g["cls_%d" % s]).max() <= 1e-4
    assert np.abs(reg[s] - g["reg_%d" % s]).max() <= 1e-4
  eng = built_lib.Engine(tcfg, device=0)
  eng.set_params(tparams)
  eng.set_compute_mode(mode)
  eng.train_init()
  no_grad = sorted(str(n) for n in g["no_grad"])
  for step, fd in enumerate(feeds):
    p_before = {n: eng.get_param(n) for n in tparams}
    loss, wd, pgl = eng.train_forward_backward(fd)
    ref = g["loss_%d" % step]
    print("single decoder/%s step %d: loss %.6f reference run %.6f" % (mode, step, loss, ref[0]))
    assert np.allclose([loss, wd] + pgl, ref, rtol=1e-4, atol=1e-5), (loss, ref)
    if step == 0:
      _, _, _, og64 = _oracle64("single", p_before, tcfg, fd)
      for n, _ in eng.param_specs():
        gr = eng.get_grad(n)
        if n in no_grad:
          assert not gr.any() and og64.get(n) is None
          continue
        e_s, e_a = sg.digest_err(gr, g["grad_0|%s" % n])
        e64 = _rel(gr, og64[n])
        assert e_s < 2e-3 and e_a < 2e-3 and e64 < 2e-3, (n, e_s, e_a, e64)
    eng.train_apply(1.0)
  assert eng.global_step == len(feeds)
  for n, _ in eng.param_specs():
    e_s, e_a = sg.digest_err(eng.get_param(n), g["param|%s" % n])
    assert e_s < 1e-5 and e_a < 1e-5, (n, e_s, e_a)
  for n in no_grad:
    assert (eng.get_param(n) == tparams[n]).all()
  eng.close()


@pytest.mark.parametrize("mode", ["f32", "f16x3"])
def test_single_decoder_with_beam_search(built_lib, mode):
  """--use_single_decoder + --use_beam_search (code/pred_models.py:274, 287-296): the offsets
  are hidden2grid of the class decoder's states traced back along every beam,
  grid_pred_reg_decoded [N * beam, T, H, W, 2].  Against the frozen run of the reference's own
  graph (golden_shim_single_decoder_beam.npz) and, with hipGraph replay, against itself."""
  g, cfg, params, feed = sg.single_decoder_beam_case()
  N, B, Tp = cfg.batch_size, cfg.beam_size, cfg.pred_len
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode(mode)
  arrs, s = eng.forward_beam(feed)
  assert s == 1 and arrs["grid_reg"].shape == (N * B, Tp, 9, 16, 2)
  assert (arrs["ids"] == g["beam_ids"]).all()          # no tied selections in this case
  assert np.abs(arrs["logits"] - g["beam_logits"]).max() < 1e-4
  assert np.abs(arrs["logprobs"] - g["beam_logprobs"]).max() < 1e-3
  assert np.abs(arrs["best_beam"] - g["cls_1"]).max() < 1e-4
  assert np.abs(arrs["grid_reg"] - g["reg_1"]).max() < 1e-4
  eng.set_graph_mode(True)
  again, _ = eng.forward_beam(feed)
  again, _ = eng.forward_beam(feed)
  eng.close()
  for k in arrs:
    assert (again[k] == arrs[k]).all(), k
