#!/usr/bin/env python
# coding=utf-8
"""CPU study (numpy): what would F(4,3) over image rows cost in accuracy with f16x3 operands?

The shipped gate kernel pairs rows with F(2,3): 4 products per 2 output rows x 3 taps (2/3 of the
direct MFMAs; csrc/convlstm_wino.h, numpy twin in tests/test_wino_model.py).  F(4,3) needs 6 products
per 4 output rows (1/2).  Its transforms carry the constants 1/4, 1/6, 1/24 (kernel side), 4, 5 (input
side) and 8 (output side), so rounding errors of the operands and of the fp32 accumulators are
amplified.  This script measures by how much, under the BEST case for the in-kernel input transform
(V = B^T d formed exactly from the plane pairs, then re-split into two fp16 planes -- what an
error-free TwoSum chain approaches), on operands shaped like the model's: |h| <= 1, Glorot gate
kernels, K = 9 * 320 products per pre-activation.

    python tools/winograd_f43_numerics.py        # prints the table quoted in DESIGN.md section 8
"""
import numpy as np

F16 = np.float16
AT = np.array([[1, 1, 1, 1, 1, 0], [0, 1, -1, 2, -2, 0], [0, 1, 1, 4, 4, 0], [0, 1, -1, 8, -8, 1]], float)
G = np.array([[1 / 4, 0, 0], [-1 / 6, -1 / 6, -1 / 6], [-1 / 6, 1 / 6, -1 / 6],
              [1 / 24, 1 / 12, 1 / 6], [1 / 24, -1 / 12, 1 / 6], [0, 0, 1]])
BT = np.array([[4, 0, -5, 0, 1, 0], [0, -4, -4, 1, 1, 0], [0, 4, -4, -1, 1, 0],
               [0, -2, -1, 2, 1, 0], [0, 2, -1, -2, 1, 0], [0, 4, 0, -5, 0, 1]], float)
# F(3,3), points 0, 1, -1, 2, inf: 5 products per 3 output rows x 3 taps (5/9 of the direct MFMAs);
# 18 = 6 x 3 and 9 = 3 x 3 rows: no partial tiles on either scale of the published configuration
AT3 = np.array([[1, 1, 1, 1, 0], [0, 1, -1, 2, 0], [0, 1, 1, 4, 1]], float)
G3 = np.array([[1 / 2, 0, 0], [-1 / 2, -1 / 2, -1 / 2], [-1 / 6, 1 / 6, -1 / 6], [1 / 6, 1 / 3, 2 / 3],
               [0, 0, 1]])
BT3 = np.array([[2, -1, -2, 1, 0], [0, -2, -1, 1, 0], [0, 2, -3, 1, 0], [0, -1, 0, 1, 0],
                [0, 2, -1, -2, 1]], float)
AT2 = np.array([[1, 1, 1, 0], [0, 1, -1, -1]], float)
G2 = np.array([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]])
BT2 = np.array([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], float)


def planes(v, scale):
  """fp64 -> (hi, lo) fp16 planes of scale * v, returned as fp32 arrays"""
  s = v * scale
  hi = s.astype(F16)
  lo = (s - hi.astype(np.float64)).astype(F16)
  return hi.astype(np.float32), lo.astype(np.float32)


def conv_direct64(d, w):
  H, W, Ci = d.shape
  pad = np.zeros((H + 2, W + 2, Ci))
  pad[1:-1, 1:-1] = d
  out = np.zeros((H, W, w.shape[3]))
  for ky in range(3):
    for kx in range(3):
      out += pad[ky:ky + H, kx:kx + W] @ w[ky, kx]
  return out


def conv_direct_f16x3(d, w):
  """the direct f16x3 form: planes of 256 d and 256 w, three products, fp32 accumulation"""
  H, W, Ci = d.shape
  dh, dl = planes(d, 256.0)
  out = np.zeros((H, W, w.shape[3]), np.float32)
  pad = lambda a: np.pad(a, ((1, 1), (1, 1), (0, 0)))
  ph, pl = pad(dh), pad(dl)
  for ky in range(3):
    for kx in range(3):
      wh, wl = planes(w[ky, kx], 256.0)
      a, b = ph[ky:ky + H, kx:kx + W], pl[ky:ky + H, kx:kx + W]
      out += (b @ wh).astype(np.float32)
      out += (a @ wl).astype(np.float32)
      out += (a @ wh).astype(np.float32)
  return out.astype(np.float64) * 2.0 ** -16


def conv_wino_rows(d, w, m):
  """F(m,3) over rows, m = 2 or 4; dx taps direct.  Operand planes as the kernel would hold them:
  V = B^T d exactly from the (hi + lo) values of d's planes, re-split under the same 2^8 scale;
  U = G g in fp64, planes under 2^8; M_c in fp32 from three products; y = A^T M in fp32."""
  at, g, bt = {2: (AT2, G2, BT2), 3: (AT3, G3, BT3), 4: (AT, G, BT)}[m]
  nc = m + 2
  H, W, Ci = d.shape
  N = w.shape[3]
  dh, dl = planes(d, 256.0)
  dq = (dh.astype(np.float64) + dl.astype(np.float64)) / 256.0        # what the planes hold
  rows = np.zeros((H + m + 2, W + 2, Ci))
  rows[1:H + 1, 1:W + 1] = dq
  U = np.einsum("ck,kxio->cxio", g, w)                                 # [nc][3 dx][Ci][N]
  out = np.zeros((H + m, W, N), np.float32)
  for t in range((H + m - 1) // m):
    tile = rows[t * m:t * m + nc]                                      # input rows m t - 1 ... m t + m
    V = np.einsum("cr,rxi->cxi", bt, tile)                             # [nc][W + 2][Ci]
    M = np.zeros((nc, W, N), np.float32)
    for c in range(nc):
      vh, vl = planes(V[c], 256.0)
      for dx in range(3):
        uh, ul = planes(U[c, dx], 256.0)
        a, b = vh[dx:dx + W], vl[dx:dx + W]
        M[c] += (b @ uh).astype(np.float32)
        M[c] += (a @ ul).astype(np.float32)
        M[c] += (a @ uh).astype(np.float32)
    y = np.zeros((m, W, N), np.float32)
    for o in range(m):
      acc = np.zeros((W, N), np.float32)
      for c in range(nc):
        if at[o, c] != 0:
          acc = (acc + np.float32(at[o, c]) * M[c]).astype(np.float32)
      y[o] = acc
    out[t * m:t * m + m] = y
  return out[:H].astype(np.float64) * 2.0 ** -16


def main():
  rng = np.random.default_rng(20200614)
  H, W, Ci, N = 18, 8, 320, 128
  lim = np.sqrt(6.0 / (9 * Ci + 9 * 1024))
  print("| operands | max abs error of the pre-activation vs fp64: direct f16x3 | F(2,3) rows | F(3,3) rows | F(4,3) rows | max abs pre-activation |")
  print("|---|---|---|---|---|---|")
  for name, gain, dscale in (("reference initialisers, |h| ~ tanh", 1.0, 1.0),
                             ("recurrent gain 3", 3.0, 1.0),
                             ("rows alternating |h| ~ 1 and ~ 1e-3", 1.0, None)):
    w = rng.uniform(-lim, lim, size=(3, 3, Ci, N)) * gain
    d = np.tanh(rng.normal(size=(H, W, Ci)) * 1.5)
    if dscale is None:
      d[1::2] *= 1e-3
    ref = conv_direct64(d, w)
    e0 = np.abs(conv_direct_f16x3(d, w) - ref).max()
    e2 = np.abs(conv_wino_rows(d, w, 2) - ref).max()
    e3 = np.abs(conv_wino_rows(d, w, 3) - ref).max()
    e4 = np.abs(conv_wino_rows(d, w, 4) - ref).max()
    print("| %s | %.2e | %.2e | %.2e | %.2e | %.2f |" % (name, e0, e2, e3, e4, np.abs(ref).max()))


if __name__ == "__main__":
  main()
