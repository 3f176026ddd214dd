# coding=utf-8
"""CPU: the oracle against its independent naive twin, against the committed
golden fixtures, and against the properties the reference's wiring implies."""
import os
import sys

import numpy as np
import torch

from multiverse_amd import synth
from oracle import multiverse_oracle as oracle
from oracle import naive_twin

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
sys.path.insert(0, GOLD)
import cases  # noqa: E402


def _g(name):
  return np.load(os.path.join(GOLD, name))


def test_same_padding_matches_tf_rule():
  # out = ceil(in/s); extra pad goes bottom/right
  assert oracle.same_pads(36, 3, 2) == (0, 1)
  assert oracle.same_pads(18, 3, 2) == (0, 1)
  assert oracle.same_pads(18, 3, 1) == (1, 1)
  assert oracle.same_pads(5, 3, 2) == (1, 1)


def test_conv_twin_agreement_stride1_and_2():
  rng = np.random.default_rng(0)
  x = rng.normal(size=(2, 6, 9, 3)).astype("f4")
  w = rng.normal(size=(3, 3, 3, 5)).astype("f4")
  for stride in (1, 2):
    a = oracle.conv2d_same(torch.from_numpy(x), torch.from_numpy(w), stride).numpy()
    b = naive_twin.conv2d_same_naive(x, w, stride)
    assert a.shape == b.shape
    assert np.abs(a - b).max() < 1e-5


def test_convlstm_cell_twin_agreement_fp64():
  rng = np.random.default_rng(1)
  x = rng.normal(size=(2, 5, 6, 3))
  h = rng.normal(size=(2, 5, 6, 8))
  c = rng.normal(size=(2, 5, 6, 8))
  k = rng.normal(size=(3, 3, 11, 32)) * 0.2
  b = rng.normal(size=32)
  c1, h1 = oracle.convlstm_step_np(x, c, h, k, b, dtype=torch.float64)
  c2, h2 = naive_twin.convlstm_cell_naive(x, c, h, k, b)
  assert np.abs(c1 - c2).max() < 1e-12 and np.abs(h1 - h2).max() < 1e-12


def test_convlstm_gate_order_and_forget_bias():
  """i, j, f, o split order and the +1.0 forget bias (tf.contrib ConvLSTMCell)."""
  C = 4
  x = np.zeros((1, 3, 3, 1), "f4")
  h = np.zeros((1, 3, 3, C), "f4")
  c = np.ones((1, 3, 3, C), "f4")
  k = np.zeros((3, 3, 1 + C, 4 * C), "f4")
  b = np.zeros(4 * C, "f4")
  b[0 * C:1 * C] = -50.0   # input gate closed
  b[2 * C:3 * C] = 0.0     # forget gate: sigmoid(0 + 1.0)
  b[3 * C:4 * C] = 50.0    # output gate open
  cn, hn = oracle.convlstm_step_np(x, c, h, k, b)
  sig1 = 1.0 / (1.0 + np.exp(-1.0))
  assert np.allclose(cn, sig1, atol=1e-6)
  assert np.allclose(hn, np.tanh(sig1), atol=1e-6)


def test_gnn_dense_equals_stencil():
  h, sm = cases.gnn_case()
  a = oracle.gnn_np(h, sm)
  b = naive_twin.gnn_stencil_naive(h, sm)
  assert np.abs(a - b).max() < 1e-6


def test_neighbor_mask_counts():
  m = oracle.neighbor_mask(4, 5, torch.float32).numpy()
  counts = m.sum(1).reshape(4, 5)
  assert counts[0, 0] == 4 and counts[0, 2] == 6 and counts[1, 2] == 9
  assert (m == m.T).all()


def test_grid_emb_onehot_closed_form():
  """tanh(conv3x3(one_hot)+b) == table lookup by offset from the hot cell --
  the identity the HIP grid_emb_onehot kernel relies on."""
  rng = np.random.default_rng(3)
  H, W, E = 5, 7, 32
  Wemb = rng.normal(size=(3, 3, 1, E)).astype("f4")
  bemb = rng.normal(size=E).astype("f4")
  for (py, px) in ((0, 0), (2, 3), (4, 6), (0, 6)):
    oh = np.zeros((1, H, W, 1), "f4")
    oh[0, py, px, 0] = 1
    ref = oracle.conv_layer(torch.from_numpy(oh), torch.from_numpy(Wemb),
                            torch.from_numpy(bemb), act=torch.tanh).numpy()[0]
    cf = naive_twin.grid_emb_onehot_closed_form(py, px, H, W, Wemb, bemb)
    assert np.abs(ref - cf).max() < 1e-6


def test_rank_and_topk_tie_order():
  x = torch.tensor([[0.5, 2.0, 2.0, -1.0, 0.5]])
  assert oracle.rank_desc_stable(x).tolist() == [[2, 0, 1, 4, 3]]
  vals, idx = oracle.topk_stable(x, 3)
  assert idx.tolist() == [[1, 2, 0]]


def test_beam_step_twin_agreement():
  rng = np.random.default_rng(5)
  B, K = 4, 30
  logits = rng.normal(size=(1, B, K)).astype("f4")
  logits[0, 1] = logits[0, 0]
  prev = rng.normal(size=(1, B)).astype("f4")
  prev[0, 1] = prev[0, 0]
  for time in (1, 2, 3):
    lp = oracle.log_softmax_tf(torch.from_numpy(logits))
    lp = torch.from_numpy(prev).unsqueeze(-1) + lp
    lp = oracle.add_div_penalty(lp, 0.01)
    flat = lp.reshape(1, B * K) if time > 1 else lp[:, 0]
    vals, idx = oracle.topk_stable(flat, B)
    nl, ids, par = naive_twin.beam_step_naive(logits[0], prev[0], time, 0.01, 1)
    assert (idx[0] % K == ids).all() and (idx[0] // K == par).all()


def test_golden_kernels_reproduce():
  g = _g("golden_kernels.npz")
  for Cx, (H, W) in cases.CELL_CASES:
    x, c, h, k, b = cases.cell_case(Cx, H, W)
    co, ho = oracle.convlstm_step_np(x, c, h, k, b)
    assert np.abs(co - g["cell%d_c_out" % Cx]).max() < 1e-6
    assert np.abs(ho - g["cell%d_h_out" % Cx]).max() < 1e-6
  h, sm = cases.gnn_case()
  assert np.abs(oracle.gnn_np(h, sm) - g["gnn_out"]).max() < 1e-6


def test_golden_greedy_cfg1_reproduces():
  """BASELINE config 1 (single scale 18x32, N=4, CPU): oracle == fixture."""
  g = _g("golden_greedy_cfg1.npz")
  cfg = synth.default_config(batch_size=4, use_grids=(1, 0))
  params = synth.make_params(cfg, seed=int(g["seed"][0]),
                             recurrent_gain=float(g["gain"][0]),
                             bias_scale=float(g["bias"][0]))
  feed = synth.make_feed(cfg, seed=int(g["seed"][0]))
  trace = {}
  cls, reg, beam = oracle.forward(params, cfg, feed, trace=trace)
  assert beam is None and cls[1] == [] and reg[1] == []
  assert np.abs(cls[0] - g["cls_0"]).max() < 2e-5
  assert np.abs(reg[0] - g["reg_0"]).max() < 2e-5
  assert (np.stack(trace["greedy_ids_0"], 1) == g["ids_0"]).all()


def test_golden_beam_reproduces_and_is_consistent():
  g = _g("golden_beam_s1.npz")
  cfg = synth.default_config(batch_size=2, use_grids=(0, 1), beam_size=5)
  params = synth.make_params(cfg, seed=int(g["seed"][0]),
                             recurrent_gain=float(g["gain"][0]),
                             bias_scale=float(g["bias"][0]))
  feed = synth.make_feed(cfg, seed=int(g["seed"][0]))
  cls, reg, beam = oracle.forward(params, cfg, feed)
  assert (beam[1] == g["beam_ids"]).all()
  assert np.abs(beam[0] - g["beam_logits"]).max() < 2e-5
  assert np.abs(beam[2] - g["beam_logprobs"]).max() < 1e-4
  # best beam == beam 0, non-increasing final scores (top_k sorted)
  assert np.abs(cls[1].reshape(2, 12, -1) - beam[0][:, 0]).max() == 0
  assert (np.diff(beam[2], axis=1) <= 1e-6).all()


def test_greedy_is_batch_independent():
  """Trajectories are independent units: a sample's outputs do
  not depend on what else is in the batch -> batch sharding needs no exchange."""
  cfg4 = synth.default_config(batch_size=4, use_grids=(0, 1))
  params = synth.make_params(cfg4, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg4, seed=11)
  cls4, reg4, _ = oracle.forward(params, cfg4, feed)
  cfg2 = synth.default_config(batch_size=2, use_grids=(0, 1))
  sub = dict(feed)
  sub["obs_scene"] = feed["obs_scene"][2:4]
  sub["grid_obs_labels"] = [a[2:4] for a in feed["grid_obs_labels"]]
  sub["grid_obs_regress"] = [a[2:4] for a in feed["grid_obs_regress"]]
  cls2, reg2, _ = oracle.forward(params, cfg2, sub)
  assert np.abs(cls4[1][2:4] - cls2[1]).max() < 1e-5
  assert np.abs(reg4[1][2:4] - reg2[1]).max() < 1e-5


def test_gnn_against_scikit_learn_and_scipy():
  """The graph attention (code/pred_models.py:808-909) written with third-party pieces:
  cosine similarity from scikit-learn, the neighbourhood mask as the reference builds it --
  a one-hot image convolved with 3x3 ones (:890-902) -- through scipy.signal.convolve2d,
  softmax from scipy.special; against the oracle's torch restatement."""
  import scipy.signal
  import scipy.special
  from sklearn.metrics.pairwise import cosine_similarity
  rng = np.random.default_rng(21)
  M, H, W, C, D = 2, 5, 7, 256, 64
  h = np.tanh(rng.normal(size=(M, H, W, C)))
  sm = np.tanh(rng.normal(size=(M, H, W, D)))
  K = H * W
  mask = np.zeros((K, K))
  for k in range(K):
    one = np.zeros((H, W)); one.flat[k] = 1.0
    mask[k] = scipy.signal.convolve2d(one, np.ones((3, 3)), mode="same").reshape(K)
  want = np.empty_like(h)
  for m in range(M):
    feat = np.concatenate([h[m].reshape(K, C), sm[m].reshape(K, D)], -1)
    e = cosine_similarity(feat)
    e = np.where(mask > 0, e, -1e30)                      # exp_mask, :1399-1401
    a = scipy.special.softmax(e, axis=-1)
    want[m] = (h[m].reshape(K, C) + a @ h[m].reshape(K, C)).reshape(H, W, C)
  got = oracle.gnn_np(h, sm, dtype=torch.float64)
  assert np.abs(got - want).max() < 1e-12


def test_beam_comparison_rules_on_the_oracle_against_itself():
  """tests/beam_compare.py (the checker of every beam-search parity test) on the oracle's own
  outputs: identical outputs pass with nothing tolerated; a swapped-in hypothesis is refused
  unless the oracle's trace shows a keep / drop cut or a within-parent rank pair tied to float32
  resolution at some step of that row ("beam_step_cut_gap", "beam_step_rank_gap": the diversity
  penalty is log(gamma) x rank within a parent, code/pred_models.py:1197-1223)."""
  import pytest
  from beam_compare import compare_beams
  cfg = synth.default_config(batch_size=2, use_grids=(0, 1), beam_size=5)
  cfg.diverse_beam = True
  params = synth.make_params(cfg, seed=5, recurrent_gain=3.0, bias_scale=0.1)
  feed = synth.make_feed(cfg, seed=5)
  trace = {}
  cls, reg, beam = oracle.forward(params, cfg, feed, trace=trace)
  T = cfg.pred_len
  cut = np.stack(trace["beam_step_cut_gap"], axis=-1)
  rank = np.stack(trace["beam_step_rank_gap"], axis=-1)
  assert cut.shape == (2, T) and rank.shape == (2, T)
  assert (cut >= 0).all() and (rank >= 0).all() and np.isfinite(rank).all()
  topv = np.stack(trace["beam_step_topvals"], axis=-1)
  arrs = {"ids": beam[1].copy(), "logits": beam[0].copy(), "logprobs": beam[2].copy(),
          "best_beam": beam[0][:, 0].copy(), "grid_reg": reg[1].copy()}
  args = (reg[1], beam[0], beam[1], beam[2], topv, trace["beam_trace"])
  assert compare_beams(arrs, *args, cut_gap=np.minimum(cut, rank)) == 0
  assert compare_beams.unmatched == 0
  # a hypothesis the oracle does not hold, in the last beam of row 1
  bad = {k: v.copy() for k, v in arrs.items()}
  bad["ids"][1, -1, T - 1] = (bad["ids"][1, -1, T - 1] + 1) % beam[0].shape[-1]
  with pytest.raises(AssertionError):
    compare_beams(bad, *args)
  with pytest.raises(AssertionError):       # gaps of this model are far from a float32 tie
    compare_beams(bad, *args, cut_gap=np.minimum(cut, rank))
  tied = np.minimum(cut, rank).copy()
  tied[1, 3] = 1e-6                         # ... unless the trace says the row has one
  compare_beams(bad, *args, cut_gap=tied)
  assert compare_beams.unmatched == 1
