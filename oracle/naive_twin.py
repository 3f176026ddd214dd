# coding=utf-8
"""Independent naive twin of the oracle's primitives -- TEST INFRASTRUCTURE ONLY.

Plain numpy float64 tap loops, written without looking at
`multiverse_oracle.py`'s torch formulation, so that a mistake in one
restatement (padding side, tap orientation, gate order, tie order, mask) shows
up as a disagreement between the two.  Small shapes only.

Reference anchors are the same as the oracle's: code/pred_models.py:1333-1373
(conv), :808-909 (graph attention), :1197-1223 (diversity penalty), :557-591
(beam expansion) and tf.contrib.rnn.ConvLSTMCell (TF 1.15).
"""

from __future__ import annotations

import math

import numpy as np


def conv2d_same_naive(x, w, stride=1):
  """x [N,H,W,Ci], w [kh,kw,Ci,Co]; TF SAME (extra pad bottom/right)."""
  x = np.asarray(x, dtype=np.float64)
  w = np.asarray(w, dtype=np.float64)
  N, H, W, Ci = x.shape
  kh, kw, _, Co = w.shape
  Ho, Wo = -(-H // stride), -(-W // stride)
  ph = max((Ho - 1) * stride + kh - H, 0)
  pw = max((Wo - 1) * stride + kw - W, 0)
  pt, pl = ph // 2, pw // 2
  out = np.zeros((N, Ho, Wo, Co), dtype=np.float64)
  for oy in range(Ho):
    for ox in range(Wo):
      for ky in range(kh):
        iy = oy * stride + ky - pt
        if iy < 0 or iy >= H:
          continue
        for kx in range(kw):
          ix = ox * stride + kx - pl
          if ix < 0 or ix >= W:
            continue
          out[:, oy, ox, :] += x[:, iy, ix, :] @ w[ky, kx]
  return out


def sigmoid(z):
  return 1.0 / (1.0 + np.exp(-z))


def convlstm_cell_naive(x, c, h, kernel, biases, forget_bias=1.0):
  g = conv2d_same_naive(np.concatenate([x, h], axis=-1), kernel) + biases
  C = g.shape[-1] // 4
  i, j, f, o = g[..., :C], g[..., C:2 * C], g[..., 2 * C:3 * C], g[..., 3 * C:]
  nc = sigmoid(f + forget_bias) * c + sigmoid(i) * np.tanh(j)
  nh = np.tanh(nc) * sigmoid(o)
  return nc, nh


def gnn_stencil_naive(h, scene_mean):
  """h + sum_j softmax_j(cos(f_i, f_j)) h_j over the <=9 in-bounds
  neighbours j of cell i, f = l2norm([h ; scene_mean]).  The masked entries of
  the reference's dense form are exp(-1e30 - max) == 0 exactly, so the stencil
  is the same function."""
  h = np.asarray(h, dtype=np.float64)
  sm = np.asarray(scene_mean, dtype=np.float64)
  M, H, W, C = h.shape
  f = np.concatenate([h, sm], axis=-1)
  f = f / np.sqrt(np.maximum((f * f).sum(-1, keepdims=True), 1e-12))
  out = np.array(h)
  for m in range(M):
    for y in range(H):
      for x in range(W):
        nb = [(yy, xx) for yy in (y - 1, y, y + 1) for xx in (x - 1, x, x + 1)
              if 0 <= yy < H and 0 <= xx < W]
        e = np.array([f[m, y, x] @ f[m, yy, xx] for yy, xx in nb])
        a = np.exp(e - e.max())
        a = a / a.sum()
        for wgt, (yy, xx) in zip(a, nb):
          out[m, y, x] += wgt * h[m, yy, xx]
  return out


def grid_emb_onehot_closed_form(py, px, H, W, Wemb, bemb):
  """tanh(conv3x3_SAME(one_hot(py,px)) + b) without the conv: the cell at
  offset (dy,dx) from the hot cell sees tap (1-dy, 1-dx)."""
  E = Wemb.shape[-1]
  out = np.tile(np.tanh(np.asarray(bemb, dtype=np.float64)), (H, W, 1))
  for dy in (-1, 0, 1):
    for dx in (-1, 0, 1):
      y, x = py + dy, px + dx
      if 0 <= y < H and 0 <= x < W:
        out[y, x] = np.tanh(np.asarray(Wemb[1 - dy, 1 - dx, 0], dtype=np.float64)
                            + bemb)
  return out


def beam_step_naive(logits, prev_lp, time, gamma, fix_num_timestep,
                    diverse=True):
  """logits [B,K] (one sample), prev_lp [B] -> (new_lp [B], ids [B],
  parents [B]) with python sorts keyed (-value, index)."""
  B, K = logits.shape
  lp = np.zeros((B, K), dtype=np.float32)
  for b in range(B):
    row = logits[b].astype(np.float32)
    m = row.max()
    sh = row - m
    lse = np.log(np.exp(sh).sum(dtype=np.float32)).astype(np.float32)
    lp[b] = (sh - lse).astype(np.float32)
    lp[b] = np.float32(prev_lp[b]) + lp[b]
    if diverse:
      order = sorted(range(K), key=lambda v: (-lp[b][v], v))
      rank = np.zeros(K, dtype=np.float32)
      for r, v in enumerate(order):
        rank[v] = r
      lp[b] = lp[b] + np.log(np.float32(gamma)) * rank
  cand = lp.reshape(-1) if time > 1 else lp[0]
  order = sorted(range(cand.shape[0]), key=lambda v: (-cand[v], v))[:B]
  new_lp = np.array([cand[v] for v in order], dtype=np.float32)
  if not time > fix_num_timestep:
    new_lp = np.zeros(B, dtype=np.float32)
  ids = np.array([v % K for v in order], dtype=np.int32)
  parents = np.array([v // K for v in order], dtype=np.int32)
  return new_lp, ids, parents
