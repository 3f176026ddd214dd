# coding=utf-8
"""GPU: the reference flags the matrix-pipe kernels do not cover still RUN.

  --convlstm_kernel 1 / 5  (code/train.py:70): tf.contrib.rnn.ConvLSTMCell with a k x k kernel,
      SAME padding.  Every fast gate kernel is a 3 x 3 stencil; other sizes take the plain fp32
      loops of csrc/convlstm_generic.h (forward, dgrad, wgrad) in compute mode 0 -- slow by
      design, held to the same bars: the frozen run of the reference's own Trainer.step on the
      TF-1 shim (golden_shim_variant_ck1 / ck5.npz), the fp64 oracle on every gradient element,
      the greedy forward against the oracle; modes 1 / 2 are refused loudly.
  --scene_conv_dim 128     (code/train.py:69): the graph attention with two scene channels per
      lane (one-wave-per-cell form), the class encoder's 128-channel x operand; every compute
      mode (golden_shim_variant_scd128.npz)."""
import numpy as np
import pytest
import torch

from multiverse_amd import synth
from oracle import multiverse_oracle as oracle

import shim_golden as sg

pytestmark = pytest.mark.gpu


def _rel(a, b):
  a = np.asarray(a, dtype=np.float64)
  b = np.asarray(b, dtype=np.float64)
  return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def _train_pin(built_lib, name, mode):
  g, cfg, params, feeds = sg.variant_case(name)
  feed = feeds[0]
  eng = built_lib.Engine(cfg, device=0)
  assert sorted(n for n, _ in eng.param_specs()) == sorted(params)
  for n, shape in eng.param_specs():
    assert tuple(shape) == params[n].shape, (n, shape, params[n].shape)
  eng.set_params(params)
  eng.set_compute_mode(mode)
  eng.train_init()
  eng.set_dropout_seed(feed["dropout_seed"])
  loss, wd, pgl = eng.train_forward_backward(feed)
  ref = g["loss_0"]
  print("%s/%s: loss %.6f reference run %.6f" % (name, mode, loss, ref[0]))
  assert np.allclose([loss, wd] + pgl, ref, rtol=1e-4, atol=1e-5), (loss, ref)
  _, _, _, og64 = oracle.loss_and_grads(params, cfg, feed, dtype=torch.float64)
  worst = 0.0
  for n, _ in eng.param_specs():
    gr = eng.get_grad(n)
    e_s, e_a = sg.digest_err(gr, g["grad_0|%s" % n])
    e64 = _rel(gr, og64[n])
    worst = max(worst, e_s, e64)
    assert e_s < 2e-3 and e_a < 2e-3 and e64 < 2e-3, (n, e_s, e_a, e64)
  print("  worst gradient error (of max|g|): %.2e" % worst)
  eng.train_apply(1.0)
  for n, _ in eng.param_specs():
    e_s, e_a = sg.digest_err(eng.get_param(n), g["param|%s" % n])
    assert e_s < 1e-4 and e_a < 1e-5, (n, e_s, e_a)
  eng.close()


@pytest.mark.parametrize("name", ["ck1", "ck5"])
def test_convlstm_kernel_sizes_train_step_vs_reference_run(built_lib, name):
  _train_pin(built_lib, name, "f32")


@pytest.mark.parametrize("k", [2, 4])
def test_even_convlstm_kernel_train_step_vs_oracle(built_lib, k):
  """Even kernel sizes: TF's SAME padding puts the extra row / column at the bottom / right
  (pad_before = (k - 1) // 2); forward, dgrad and wgrad of csrc/convlstm_generic.h each carry
  that convention in their own index arithmetic.  Loss and EVERY gradient element against the
  fp64 oracle (whose conv2d SAME the TF-1 shim's reference runs pin for odd and even sizes)."""
  cfg = synth.default_config(batch_size=2, use_grids=(1, 1), is_train=True, convlstm_kernel=k)
  params = synth.make_params(cfg, seed=synth.SEED_BASE + 40 + k, recurrent_gain=2.0)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 41 + k)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode("f32")
  eng.train_init()
  loss, wd, pgl = eng.train_forward_backward(feed)
  grads = {n: eng.get_grad(n) for n, _ in eng.param_specs()}
  eng.close()
  oloss, owd, opgl, og64 = oracle.loss_and_grads(params, cfg, feed, dtype=torch.float64)
  print("convlstm_kernel %d: loss %.6f oracle %.6f" % (k, loss, oloss))
  assert abs(loss - oloss) < 1e-4 * max(1.0, abs(oloss))
  assert np.allclose(pgl, opgl, rtol=1e-4, atol=1e-5)
  worst = 0.0
  for n in sorted(grads):
    e = _rel(grads[n], og64[n])
    worst = max(worst, e)
    assert e < 2e-3, (n, e)
  print("  worst gradient error (of max|g|): %.2e" % worst)


@pytest.mark.parametrize("k", [1, 2, 4, 5])
def test_convlstm_kernel_sizes_greedy_forward_vs_oracle(built_lib, k):
  """Both scales, greedy decode: argmax ids bit-exact, logits / offsets within 1e-4 (k = 2: an
  even kernel -- SAME pads bottom / right)."""
  cfg = synth.default_config(batch_size=2, use_grids=(1, 1), convlstm_kernel=k)
  params = synth.make_params(cfg, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 60 + k)
  ocls, oreg, _ = oracle.forward(params, cfg, feed)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode("f32")
  cls, reg = eng.forward_greedy(feed)
  for s in range(2):
    K = cfg.scene_grids[s][0] * cfg.scene_grids[s][1]
    oc, gc = ocls[s].reshape(2, -1, K), cls[s].reshape(2, -1, K)
    # A 1 x 1 cell mixes nothing spatially: every cell away from the trajectory carries the
    # same state, hidden2grid gives those cells EXACTLY tied logits in the oracle, and the
    # argmax fed back to the next step hangs on rounding.  Steps are compared up to the first
    # one whose oracle top-1 / top-2 margin is below 1e-4.
    top2 = np.sort(oc, axis=-1)[..., -2:]
    margin = top2[..., 1] - top2[..., 0]
    steps = oc.shape[1]
    for n in range(2):
      tied = np.nonzero(margin[n] < 1e-4)[0]
      upto = int(tied[0]) + 1 if len(tied) else steps
      dc = np.abs(gc[n, :upto] - oc[n, :upto]).max()
      print("convlstm_kernel %d scale %d row %d: %d of %d steps compared (oracle margin), "
            "max|dcls| %.3g" % (k, s, n, upto, steps, dc))
      assert dc < 1e-4
      ok_steps = upto - 1 if len(tied) else steps
      assert (gc[n, :ok_steps].argmax(-1) == oc[n, :ok_steps].argmax(-1)).all()
    dr = np.abs(reg[s] - oreg[s]).max()
    print("convlstm_kernel %d scale %d: max|dreg| %.3g" % (k, s, dr))
    assert dr < 1e-4
  # the matrix-pipe modes are 3 x 3 only: refused, not silently wrong
  for mode in ("f16x3", "bf16"):
    with pytest.raises(Exception) as err:
      eng.set_compute_mode(mode)
    assert "convlstm_kernel" in str(err.value)
  eng.close()


def test_convlstm_kernel_5_beam_search_vs_oracle(built_lib):
  """Diverse beam search (B = 4) through the generic gate kernel: ids / logits / log-probs
  against the oracle with the usual tie handling (tests/beam_compare.py)."""
  from beam_compare import compare_beams
  cfg = synth.default_config(batch_size=2, use_grids=(0, 1), beam_size=4, convlstm_kernel=5)
  params = synth.make_params(cfg, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 66)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode("f32")
  arrs, s = eng.forward_beam(feed)
  eng.close()
  trace = {}
  ocls, oreg, obeam = oracle.forward(params, cfg, feed, trace=trace)
  ologits, oids, olp = obeam
  assert s == 1
  compare_beams(arrs, oreg[1], ologits, oids, olp,
                np.stack(trace["beam_step_topvals"], axis=-1), trace["beam_trace"])


@pytest.mark.parametrize("mode", ["f32", "f16x3"])
def test_scene_conv_dim_128_train_step_vs_reference_run(built_lib, mode):
  _train_pin(built_lib, "scd128", mode)


@pytest.mark.parametrize("mode", ["f32", "f16x3"])
def test_scene_conv_dim_128_greedy_forward_vs_oracle(built_lib, mode):
  cfg = synth.default_config(batch_size=2, use_grids=(1, 1), scene_conv_dim=128)
  params = synth.make_params(cfg, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=synth.SEED_BASE + 71)
  ocls, oreg, _ = oracle.forward(params, cfg, feed)
  eng = built_lib.Engine(cfg, device=0)
  eng.set_params(params)
  eng.set_compute_mode(mode)
  cls, reg = eng.forward_greedy(feed)
  eng.close()
  for s in range(2):
    K = cfg.scene_grids[s][0] * cfg.scene_grids[s][1]
    dc, dr = np.abs(cls[s] - ocls[s]).max(), np.abs(reg[s] - oreg[s]).max()
    print("scene_conv_dim 128 / %s scale %d: max|dcls| %.3g max|dreg| %.3g" % (mode, s, dc, dr))
    assert dc < 1e-4 and dr < 1e-4
    assert (cls[s].reshape(2, -1, K).argmax(-1) == ocls[s].reshape(2, -1, K).argmax(-1)).all()
