# coding=utf-8
"""CPU: the TF-1 primitives the reference's graph is built from, as the shim
(oracle/tf1_shim/tensorflow) emulates them, against THIRD-PARTY implementations of the same
published operations -- torch.optim / torch.nn.functional / torch.optim.lr_scheduler /
scipy.signal.  The goldens of tests/golden/golden_shim_*.npz come from the reference's own
model file executed on this shim; a misread primitive would be wrong in the shim and in the
oracle alike (both were written here), so each one the training and decoding goldens depend
on is held to a library that was not.  What is TF-specific and has no library counterpart is
restated from TensorFlow 1.15's documentation in the test itself and marked so."""
import math

import numpy as np
import pytest
import scipy.signal
import torch
import torch.nn.functional as F

from oracle.tf1_shim import tensorflow as tf

F64 = torch.float64


@pytest.fixture(autouse=True)
def _float64_shim():
  tf.set_float_dtype(torch.float64)       # the comparison is about formulas, not rounding
  yield
  tf.set_float_dtype(torch.float32)
  tf.reset_default_graph()


def _var(name, value):
  tf.reset_default_graph(params={name: value})
  return tf.get_variable(name, shape=list(value.shape), dtype=tf.float64)


def _run_steps(make_tf_opt, make_torch_opt, steps=7, prime=None, seed=0):
  rng = np.random.default_rng(seed)
  w0 = rng.normal(size=(5, 3))
  grads = [rng.normal(size=(5, 3)) * (0.1 + k) for k in range(steps)]
  var = _var("w", w0)
  opt = make_tf_opt()
  sess = tf.Session()
  p = torch.nn.Parameter(torch.from_numpy(w0.copy()))
  topt = make_torch_opt([p])
  if prime:
    prime(topt, p)
  for g in grads:
    sess.run(opt.apply_gradients([(tf.constant(g, dtype=tf.float64), var)]))
    p.grad = torch.from_numpy(g.copy())
    topt.step()
  return var.value.detach().numpy(), p.detach().numpy()


def test_adadelta_is_torch_adadelta():
  # tf.train.AdadeltaOptimizer(lr, 0.95, 1e-8) -- the reference's optimizer (code/pred_models.py:1671)
  a, b = _run_steps(lambda: tf.train.AdadeltaOptimizer(0.3, rho=0.95, epsilon=1e-8),
                    lambda ps: torch.optim.Adadelta(ps, lr=0.3, rho=0.95, eps=1e-8))
  assert np.abs(a - b).max() < 1e-12


def test_momentum_is_torch_sgd_momentum():
  a, b = _run_steps(lambda: tf.train.MomentumOptimizer(0.05, momentum=0.9),
                    lambda ps: torch.optim.SGD(ps, lr=0.05, momentum=0.9))
  assert np.abs(a - b).max() < 1e-12


def test_adam_is_torch_adam_up_to_the_epsilon_placement():
  # TF: var -= lr sqrt(1-b2^t)/(1-b1^t) m / (sqrt(v) + eps); torch puts eps beside
  # sqrt(v / (1-b2^t)): the two differ by eps (1 - sqrt(1-b2^t)) / sqrt(v), ~1e-8 relative here
  a, b = _run_steps(lambda: tf.train.AdamOptimizer(0.01, beta1=0.9, beta2=0.999, epsilon=1e-8),
                    lambda ps: torch.optim.Adam(ps, lr=0.01, betas=(0.9, 0.999), eps=1e-8))
  assert np.abs(a - b).max() < 2e-6
  # the bias-correction powers are kept in float32 by TF (non-slot variables): visible at 1e-8


def test_rmsprop_is_torch_rmsprop_started_from_ones():
  # TF-1.15 RMSPropOptimizer creates its "rms" slot with ones_initializer (rmsprop.py
  # _create_slots) and puts eps INSIDE the square root; torch starts from zeros, eps outside
  def prime(topt, p):
    p.grad = torch.zeros_like(p)
    topt.step()                                   # creates the state, moves nothing
    topt.state[p]["square_avg"].fill_(1.0)
  a, b = _run_steps(lambda: tf.train.RMSPropOptimizer(0.01, decay=0.9, momentum=0.5, epsilon=1e-10),
                    lambda ps: torch.optim.RMSprop(ps, lr=0.01, alpha=0.9, momentum=0.5, eps=1e-10),
                    prime=prime)
  assert np.abs(a - b).max() < 1e-8


def test_learning_rate_schedules_are_torch_schedulers():
  lr0 = 0.3
  p = torch.nn.Parameter(torch.zeros(1))
  opt = torch.optim.SGD([p], lr=lr0)
  step = torch.optim.lr_scheduler.StepLR(opt, step_size=4, gamma=0.95)
  for gs in range(14):
    tf.reset_default_graph()
    g = tf.get_variable("global_step", shape=[], dtype=tf.int32,
                        initializer=tf.constant_initializer(gs), trainable=False)
    got = tf.train.exponential_decay(lr0, g, 4, 0.95, staircase=True)   # code/pred_models.py:1650-1657
    assert abs(float(got) - opt.param_groups[0]["lr"]) < 1e-12, gs
    opt.step(); step.step()
  opt = torch.optim.SGD([p], lr=lr0)
  cos = torch.optim.lr_scheduler.CosineAnnealingLR(opt, T_max=10, eta_min=0.0)
  for gs in range(11):
    tf.reset_default_graph()
    g = tf.get_variable("global_step", shape=[], dtype=tf.int32,
                        initializer=tf.constant_initializer(gs), trainable=False)
    got = tf.train.cosine_decay(lr0, g, 10)                            # code/pred_models.py:1659-1664
    assert abs(float(got) - opt.param_groups[0]["lr"]) < 1e-12, gs
    opt.step(); cos.step()


def test_losses_are_torch_functional_losses():
  rng = np.random.default_rng(3)
  logits = torch.from_numpy(rng.normal(size=(6, 11)) * 3)
  labels = torch.from_numpy(rng.integers(0, 11, size=6))
  got = tf.nn.sparse_softmax_cross_entropy_with_logits(labels=tf.constant(labels.numpy()),
                                                       logits=tf.constant(logits.numpy(), dtype=tf.float64))
  assert np.abs(got.numpy() - F.cross_entropy(logits, labels, reduction="none").numpy()).max() < 1e-12
  soft = torch.from_numpy(rng.random(size=(6, 11)))          # un-normalised soft labels
  got = tf.nn.softmax_cross_entropy_with_logits(labels=tf.constant(soft.numpy(), dtype=tf.float64),
                                                logits=tf.constant(logits.numpy(), dtype=tf.float64))
  assert np.abs(got.numpy() + (soft * F.log_softmax(logits, -1)).sum(-1).numpy()).max() < 1e-12
  pred = torch.from_numpy(rng.normal(size=(40, 2)) * 2)
  lab = torch.from_numpy(rng.normal(size=(40, 2)))
  got = tf.losses.huber_loss(labels=tf.constant(lab.numpy(), dtype=tf.float64),
                             predictions=tf.constant(pred.numpy(), dtype=tf.float64),
                             reduction=tf.losses.Reduction.MEAN)     # code/pred_models.py:1016-1022
  assert abs(float(got.numpy()) - float(F.huber_loss(pred, lab, delta=1.0))) < 1e-12
  x = torch.from_numpy(rng.normal(size=(7, 5, 320)))
  got = tf.nn.l2_normalize(tf.constant(x.numpy(), dtype=tf.float64), -1)   # gnn_edge, :845
  assert np.abs(got.numpy() - F.normalize(x, dim=-1).numpy()).max() < 1e-12
  got = tf.nn.softmax(tf.constant(logits.numpy(), dtype=tf.float64))
  assert np.abs(got.numpy() - F.softmax(logits, -1).numpy()).max() < 1e-12


# TensorFlow 1.15 documentation (tf.nn.convolution, "SAME"): out = ceil(in / stride); the total
# padding max((out - 1) * stride + k - in, 0) goes half (rounded down) in front, the rest behind
def _tf_same_pad_doc(n, k, s):
  out = -(-n // s)
  total = max((out - 1) * s + k - n, 0)
  return total // 2, total - total // 2, out


@pytest.mark.parametrize("H,W,k,s", [(36, 64, 3, 2), (18, 32, 3, 2), (9, 16, 3, 1), (5, 7, 3, 2),
                                     (6, 6, 1, 1), (4, 5, 3, 1)])
def test_conv2d_same_is_scipy_correlation_on_the_documented_padding(H, W, k, s):
  rng = np.random.default_rng(H * 10 + k)
  x = rng.normal(size=(2, H, W, 3))
  w = rng.normal(size=(k, k, 3, 4))
  got = tf.nn.conv2d(tf.constant(x, dtype=tf.float64), tf.constant(w, dtype=tf.float64),
                     [1, s, s, 1], "SAME").numpy()
  pt, pb, Ho = _tf_same_pad_doc(H, k, s)
  pl, pr, Wo = _tf_same_pad_doc(W, k, s)
  assert got.shape == (2, Ho, Wo, 4)
  xp = np.pad(x, ((0, 0), (pt, pb), (pl, pr), (0, 0)))
  for n in range(2):
    for co in range(4):
      acc = sum(scipy.signal.correlate2d(xp[n, :, :, ci], w[:, :, ci, co], mode="valid")
                for ci in range(3))
      assert np.abs(got[n, :, :, co] - acc[::s, ::s][:Ho, :Wo]).max() < 1e-10


def test_top_k_order_and_ties():
  # TF's TopK: descending values, equal values by ascending index (the beam search relies on
  # it, code/pred_models.py:560-575)
  a = np.array([[1.0, 3.0, 3.0, -2.0, 3.0, 1.0], [0.0, 0.0, 0.0, 0.0, 0.0, 0.0]])
  vals, idx = tf.nn.top_k(tf.constant(a, dtype=tf.float64), k=4)
  want = [sorted(range(6), key=lambda i, r=r: (-a[r, i], i))[:4] for r in range(2)]
  assert idx.numpy().tolist() == want
  assert np.array_equal(vals.numpy(), np.take_along_axis(a, np.array(want), 1))
  tv, ti = torch.topk(torch.from_numpy(a[0]), 4)                   # same values as torch's
  assert np.array_equal(vals.numpy()[0], tv.numpy())


def test_convlstm_cell_on_one_pixel_is_torch_lstm_cell():
  """tf.contrib.rnn.ConvLSTMCell with a 1x1 kernel on a 1x1 image is an LSTM cell.  Against
  torch.nn.LSTMCell: TF orders the gate blocks (i, j, f, o) and adds forget_bias = 1 to f
  (contrib/rnn/python/ops/rnn_cell.py, restated from the source: no library shares that
  layout); torch orders them (i, f, g, o).  With the blocks permuted and the forget bias
  moved into the bias vector the two cells must agree -- the update c' = sig(f) c + sig(i)
  tanh(j), h' = sig(o) tanh(c') is the part a library can pin."""
  rng = np.random.default_rng(9)
  N, Cx, C = 5, 7, 6
  kernel = rng.normal(size=(1, 1, Cx + C, 4 * C)) * 0.5
  biases = rng.normal(size=(4 * C,)) * 0.3
  x = rng.normal(size=(N, 1, 1, Cx))
  c0 = rng.normal(size=(N, 1, 1, C))
  h0 = np.tanh(rng.normal(size=(N, 1, 1, C)))
  tf.reset_default_graph(params={"cell/kernel": kernel, "cell/biases": biases})
  cell = tf.contrib.rnn.ConvLSTMCell(conv_ndims=2, input_shape=[1, 1, Cx], output_channels=C,
                                     kernel_shape=[1, 1], name="cell")
  out, (c1, h1) = cell(tf.constant(x, dtype=tf.float64),
                       tf.nn.rnn_cell.LSTMStateTuple(tf.constant(c0, dtype=tf.float64),
                                                     tf.constant(h0, dtype=tf.float64)))
  ref = torch.nn.LSTMCell(Cx, C).double()
  W = kernel[0, 0]                                       # [Cx + C, 4C], TF blocks i j f o
  perm = np.concatenate([np.arange(0, C), np.arange(2 * C, 3 * C), np.arange(C, 2 * C),
                         np.arange(3 * C, 4 * C)])       # -> torch blocks i f g o
  b = biases.copy()
  b[2 * C:3 * C] += 1.0                                  # forget_bias
  with torch.no_grad():
    ref.weight_ih.copy_(torch.from_numpy(W[:Cx, perm].T.copy()))
    ref.weight_hh.copy_(torch.from_numpy(W[Cx:, perm].T.copy()))
    ref.bias_ih.copy_(torch.from_numpy(b[perm]))
    ref.bias_hh.zero_()
    rh, rc = ref(torch.from_numpy(x.reshape(N, Cx)),
                 (torch.from_numpy(h0.reshape(N, C)), torch.from_numpy(c0.reshape(N, C))))
  assert np.abs(c1.numpy().reshape(N, C) - rc.numpy()).max() < 1e-12
  assert np.abs(h1.numpy().reshape(N, C) - rh.numpy()).max() < 1e-12
  assert np.abs(out.numpy() - h1.numpy()).max() == 0.0
