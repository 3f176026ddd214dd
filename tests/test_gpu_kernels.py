# coding=utf-8
"""GPU parity, kernel by kernel, through the C ABI (include/multiverse_hip.h)
against the CPU oracle on the same seeded inputs.

Tolerances: the ConvLSTM gate GEMM runs on v_mfma_f32_32x32x2_f32, an exact
fp32 fmaf chain in a different summation order than the oracle's conv ->
abs 2e-5 on O(1) gate sums; integer outputs (argmax, beam ids/parents) are
bit-exact.
"""

import numpy as np
import pytest

from oracle import multiverse_oracle as oracle
from oracle import naive_twin

pytestmark = pytest.mark.gpu


def _rng(seed):
  return np.random.default_rng(seed)


@pytest.mark.parametrize("M,H,W,Cx,zero", [
    (2, 18, 32, 64, False),   # class encoder
    (2, 18, 32, 2, False),    # regression encoder (packed small-x chunk)
    (3, 9, 16, 32, False),    # decoders, scale 1, M*K not a multiple of 128
    (1, 5, 7, 32, False),     # ragged: one partial tile
    (2, 9, 16, 64, True),     # first encoder step: zero state
    (2, 9, 16, 2, True),
])
def test_convlstm_step(built_lib, M, H, W, Cx, zero):
  rng = _rng(M * 1000 + H * 10 + Cx)
  C = 256
  x = rng.normal(size=(M, H, W, Cx)).astype("f4")
  lim = np.sqrt(6.0 / (9 * (Cx + C) + 9 * 4 * C)) * 3.0
  kernel = rng.uniform(-lim, lim, size=(3, 3, Cx + C, 4 * C)).astype("f4")
  biases = (0.1 * rng.normal(size=4 * C)).astype("f4")
  if zero:
    c = h = None
    co, ho = oracle.convlstm_step_np(x, np.zeros((M, H, W, C), "f4"),
                                     np.zeros((M, H, W, C), "f4"), kernel, biases)
  else:
    c = rng.normal(size=(M, H, W, C)).astype("f4")
    h = np.tanh(rng.normal(size=(M, H, W, C))).astype("f4")
    co, ho = oracle.convlstm_step_np(x, c, h, kernel, biases)
  cg, hg = built_lib.op_convlstm_step(x, c, h, kernel, biases)
  assert np.abs(cg - co).max() < 2e-5
  assert np.abs(hg - ho).max() < 2e-5


def test_convlstm_step_transpose_detecting(built_lib):
  """One hot input cell / one hot weight tap: catches tap-orientation and
  row<->column swaps that symmetric data would hide."""
  M, H, W, Cx, C = 1, 6, 8, 32, 256
  x = np.zeros((M, H, W, Cx), "f4")
  x[0, 2, 5, 3] = 1.0
  kernel = np.zeros((3, 3, Cx + C, 4 * C), "f4")
  kernel[0, 2, 3, 1 * C + 17] = 2.0      # tap (ky=0,kx=2), gate j, channel 17
  kernel[:, :, :, 0 * C:1 * C] = 0.0
  biases = np.zeros(4 * C, "f4")
  biases[0 * C:1 * C] = 5.0              # input gate ~ open
  c = np.zeros((M, H, W, C), "f4")
  h = np.zeros((M, H, W, C), "f4")
  co, ho = oracle.convlstm_step_np(x, c, h, kernel, biases)
  cg, hg = built_lib.op_convlstm_step(x, c, h, kernel, biases)
  # out(y,x) sees in(y+ky-1, x+kx-1): hot input (2,5) with tap (0,2) -> out (3,4)
  assert abs(co[0, 3, 4, 17]) > 0.5
  assert np.abs(cg - co).max() < 1e-6
  assert np.abs(hg - ho).max() < 1e-6


# 36x18 / 18x9: BASELINE.json's literal grids (a quad of cells straddles image rows there);
# 3x31, 2x7: widths that divide nothing; M chosen so that the last 64-cell group is partial
@pytest.mark.parametrize("M,H,W", [(2, 18, 32), (3, 9, 16), (1, 4, 5), (2, 36, 18), (3, 18, 9),
                                   (1, 3, 31), (5, 2, 7)])
def test_gnn(built_lib, M, H, W):
  rng = _rng(H)
  h = np.tanh(rng.normal(size=(M, H, W, 256))).astype("f4")
  sm = np.tanh(rng.normal(size=(M, H, W, 64))).astype("f4")
  ref = oracle.gnn_np(h, sm)           # dense K x K form of the reference
  out = built_lib.op_gnn(h, sm)
  assert np.abs(out - ref).max() < 2e-6
  if H * W <= 20:
    twin = naive_twin.gnn_stencil_naive(h, sm)
    assert np.abs(out - twin).max() < 2e-6


def test_gnn_kernel_versions_and_position_independence(built_lib, monkeypatch):
  """The three kernels of the graph attention (MV_GNN=v1 one wave per cell, v2 LDS-tiled,
  default: register-blocked + LDS-DMA) against the oracle on the same input; and the
  default one gives a row the same bits wherever the row sits in its 64-cell groups
  (the engine's layout A/B tests rely on that)."""
  rng = _rng(77)
  M, H, W = 9, 9, 16                    # K = 144 = 2.25 groups: every row is aligned differently
  h = np.tanh(rng.normal(size=(M, H, W, 256))).astype("f4")
  sm = np.tanh(rng.normal(size=(M, H, W, 64))).astype("f4")
  ref = oracle.gnn_np(h, sm)
  outs = {}
  for ver in ("v1", "v2", ""):
    if ver:
      monkeypatch.setenv("MV_GNN", ver)
    else:
      monkeypatch.delenv("MV_GNN", raising=False)
    outs[ver] = built_lib.op_gnn(h, sm)
    assert np.abs(outs[ver] - ref).max() < 2e-6, ver
  assert np.abs(outs[""] - outs["v2"]).max() < 1e-6
  for lo in (1, 2, 3):
    part = built_lib.op_gnn(h[lo:], sm[lo:])
    assert (part == outs[""][lo:]).all(), lo


@pytest.mark.parametrize("P", [1, 2])
def test_hidden2grid(built_lib, P):
  rng = _rng(P)
  M, H, W, C = 3, 9, 16, 256
  h = np.tanh(rng.normal(size=(M, H, W, C))).astype("f4")
  w = (rng.normal(size=(3, 3, C, P)) * 0.05).astype("f4")
  import torch
  ref = oracle.conv2d_same(torch.from_numpy(h), torch.from_numpy(w)).numpy()
  out = built_lib.op_hidden2grid(h, w)
  assert np.abs(out - ref).max() < 2e-5


@pytest.mark.parametrize("time", [1, 2, 5])
@pytest.mark.parametrize("diverse", [True, False])
def test_beam_step(built_lib, time, diverse):
  rng = _rng(time)
  N, B, K = 3, 20, 576
  logits = rng.normal(size=(N, B, K)).astype("f4")
  # force exact ties inside a row and across beams
  logits[0, 0, 10] = logits[0, 0, 400] = logits[0, 0].max() + 1.0
  logits[1, 3] = logits[1, 2]
  prev = (rng.normal(size=(N, B)) * (time > 1)).astype("f4")
  prev[1, 3] = prev[1, 2]
  new_lp, ids, parents = built_lib.op_beam_step(logits, prev, time, diverse, 0.01, 1)
  for n in range(N):
    rl, ri, rp = naive_twin.beam_step_naive(logits[n], prev[n], time, 0.01, 1,
                                            diverse=diverse)
    assert (ids[n] == ri).all(), (n, ids[n], ri)
    assert (parents[n] == rp).all()
    assert np.abs(new_lp[n] - rl).max() < 1e-4
