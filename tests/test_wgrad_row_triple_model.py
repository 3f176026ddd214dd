"""CPU twin of the row-triple (Winograd F(3,3)) form of the gate-kernel weight gradient
(csrc/convlstm_wgrad_f16x3.h: `wino3_transpose_g_kernel`, `wino3_transpose_a3_kernel`, the 15-tap
mode of the GEMM kernels, `wgrad_wino3_reduce_kernel`), restated in numpy: the triple-cell ->
raw-cell arithmetic of both transposes, rows outside the image, the column-shifted copies, the
components in fp32, the split into two scaled fp16 planes in the GEMMs' cell-blocked layout, the
15 taps (component x dx) as plain f16x3 products over a third of the cells, and the three-row
output combination -- held against the definition
    dW[tap][ci][n] = sum_m in[m + d_tap][ci] G[m][n]
(reference code/pred_models.py:1694-1717: tf.gradients of the ConvLSTM kernel).  No GPU: the GPU
parity tests (tests/test_gpu_train.py, test_gpu_at_size.py, test_gpu_edge.py) run the kernels."""
import numpy as np
import pytest


def plane_index(m, r, R):
  """wg16_plane_index: cells blocked by 32, R rows (channels) per block"""
  return ((m >> 5) * R + r) * 32 + (m & 31)


def split_planes(v32, e):
  s = (v32.astype(np.float32) * np.float32(2.0 ** e)).astype(np.float32)
  hi = s.astype(np.float16)
  lo = (s - hi.astype(np.float32)).astype(np.float16)
  assert np.isfinite(hi.astype(np.float32)).all()
  return hi, lo


def transpose_g(g, Mtot3, Cc, Mrow3, W, e):
  """wino3_transpose_g_kernel: out[comp][plane] flat halves + the bias partial (-2 sum U1)"""
  out = np.zeros((5, 2, Mrow3 * Cc), dtype=np.float16)
  m3 = np.arange(Mtot3)
  q, x = m3 // W, m3 % W
  raw0 = 3 * q * W + x
  g0, g1, g2 = (g[raw0 + i * W].astype(np.float32) for i in range(3))
  half, sixth = np.float32(0.5), np.float32(1.0 / 6.0)
  U = [half * g0, -half * ((g0 + g1) + g2), ((g1 - g0) - g2) * sixth,
       ((g0 + np.float32(2) * g1) + np.float32(4) * g2) * sixth, -g2]
  for c in range(5):
    hi, lo = split_planes(U[c], e)
    for ch in range(Cc):
      idx = plane_index(m3, ch, Cc)
      out[c, 0, idx] = hi[:, ch]
      out[c, 1, idx] = lo[:, ch]
  return out, -2.0 * U[1].astype(np.float64).sum(axis=0)


def transpose_a3(a, Mtot3, Cc, Mrow3, H, W, e):
  """wino3_transpose_a3_kernel: out[dx][comp][plane] flat halves"""
  out = np.zeros((3, 5, 2, Mrow3 * Cc), dtype=np.float16)
  H3 = H // 3
  m3 = np.arange(Mtot3)
  q, x = m3 // W, m3 % W
  t3 = (q % H3) * 3
  d = []
  for i in range(5):
    y = t3 - 1 + i
    ok = (y >= 0) & (y < H)
    cell = np.where(ok, (3 * q - 1 + i) * W + x, 0)
    d.append(np.where(ok[:, None], a[cell], 0).astype(np.float32))
  two = np.float32(2)
  v3, d32 = d[3] - d[1], d[3] - d[2]
  V = [two * (d[0] - d[2]) + v3, d32 - two * d[1], two * (d[1] - d[2]) + d32, v3,
       two * v3 + (d[2] - d[4])]
  for c in range(5):
    for dd in range(3):
      dx = dd - 1
      ok = (x + dx >= 0) & (x + dx < W)
      src = np.where(ok, m3 + dx, 0)       # the neighbour triple-cell of the same image row
      sv = np.where(ok[:, None], V[c][src], np.float32(0))
      hi, lo = split_planes(sv, e)
      for ch in range(Cc):
        idx = plane_index(m3, ch, Cc)
        out[dd, c, 0, idx] = hi[:, ch]
        out[dd, c, 1, idx] = lo[:, ch]
  return out


def gather(flat, Mtot3, Cc):
  """planes back as [cell][channel] float64"""
  m3 = np.arange(Mtot3)
  return np.stack([flat[plane_index(m3, ch, Cc)] for ch in range(Cc)], axis=1).astype(np.float64)


def definition(a, g, R, H, W):
  Mtot = R * H * W
  m = np.arange(Mtot)
  y, x = (m // W) % H, m % W
  ad, gd = a.astype(np.float64), g.astype(np.float64)
  want = np.zeros((9, a.shape[1], g.shape[1]))
  for tap in range(9):
    dy, dx = tap // 3 - 1, tap % 3 - 1
    ok = (y + dy >= 0) & (y + dy < H) & (x + dx >= 0) & (x + dx < W)
    want[tap] = ad[(m + dy * W + dx)[ok]].T @ gd[m[ok]]
  return want


@pytest.mark.parametrize("H,W,R,Ca,bounded", [(18, 32, 2, 8, True), (9, 16, 3, 2, False),
                                              (3, 16, 5, 5, False), (6, 16, 1, 4, True)])
def test_row_triple_form_gives_the_weight_gradient(H, W, R, Ca, bounded):
  N4 = 12
  rng = np.random.default_rng(3 + H * W + R + Ca)
  Mtot = R * H * W
  Mtot3 = Mtot // 3
  Mrow3 = (Mtot3 + 63) // 64 * 64
  a = (rng.uniform(-1, 1, (Mtot, Ca)) if bounded else rng.standard_normal((Mtot, Ca)) * 40).astype(np.float32)
  g = (rng.standard_normal((Mtot, N4)) * np.exp(rng.standard_normal((Mtot, N4))) * 1e-3).astype(np.float32)
  # exponents as the engine sets them: h 2^8; an unbounded x operand 10 - ilogb(max) (|V| <= 6 max);
  # G 13 - ilogb(max) (|U| <= 1.5 max)
  a_exp = 8 if bounded else 10 - int(np.floor(np.log2(np.abs(a).max())))
  g_exp = 13 - int(np.floor(np.log2(np.abs(g).max())))
  gt, bias = transpose_g(g, Mtot3, N4, Mrow3, W, g_exp)
  at = transpose_a3(a, Mtot3, Ca, Mrow3, H, W, a_exp)
  assert np.allclose(bias, g.astype(np.float64).sum(axis=0), rtol=1e-5, atol=1e-7)
  # cells past the tensor stay zero (the GEMMs run to the end of a split's last k-step)
  assert not gt[:, :, plane_index(np.arange(Mtot3, Mrow3), 0, N4)].any()
  # 15 taps: tap = comp * 3 + (dx + 1); three MFMAs per product
  M = np.zeros((15, Ca, N4))
  for comp in range(5):
    g_hi, g_lo = gather(gt[comp, 0], Mtot3, N4), gather(gt[comp, 1], Mtot3, N4)
    for dd in range(3):
      a_hi, a_lo = gather(at[dd, comp, 0], Mtot3, Ca), gather(at[dd, comp, 1], Mtot3, Ca)
      M[comp * 3 + dd] = (a_lo.T @ g_hi + a_hi.T @ g_lo + a_hi.T @ g_hi) * 2.0 ** -(a_exp + g_exp)
  # wgrad_wino3_reduce_kernel
  got = np.zeros((9, Ca, N4))
  for dd in range(3):
    m = [M[c * 3 + dd] for c in range(5)]
    got[0 * 3 + dd] = m[0] + m[1] + m[2] + m[3]
    got[1 * 3 + dd] = m[1] - m[2] + 2 * m[3]
    got[2 * 3 + dd] = m[1] + m[2] + 4 * m[3] + m[4]
  want = definition(a, g, R, H, W)
  err = np.abs(got - want).max() / np.abs(want).max()
  assert err < 2e-6, err


def test_row_triple_tiles_of_the_x_rows_hold_one_component_each():
  """x rows in 15-tap mode: a tile's rows are (dx, channel) pairs of ONE component, nrbc tiles per
  component; every (tap, channel) is covered exactly once and dead rows are past 3 Cx."""
  for tile in (128,):
    for Cx in (2, 32, 64, 128):
      nrbc = (3 * Cx + tile - 1) // tile
      seen = set()
      for rb in range(5 * nrbc):
        comp, rbi = rb // nrbc, rb % nrbc
        for row in range(tile):
          Rr = rbi * tile + row
          if Rr < 3 * Cx:
            tq, ci = Rr // Cx, Rr % Cx
            tap = comp * 3 + tq
            assert (tap, ci) not in seen
            seen.add((tap, ci))
      assert seen == {(t, c) for t in range(15) for c in range(Cx)}
