# coding=utf-8
"""CPU: a numpy twin of what csrc/convlstm_wino.h computes and of the index maps it relies on.

No GPU here: these tests hold (a) the ALGEBRA of the Winograd F(2,3) row-pair form with f16x3
operands -- the fp64 kernel transform split into two fp16 planes of 256 U, the in-kernel input
transform as an error-free TwoSum on fp16 plane pairs, three fp16 products per component
accumulated in fp32, the output transform -- against a direct fp64 3x3 SAME convolution, and
(b) the LAYOUT bookkeeping the kernel's correctness rests on: pair-cell tiling, the DPP column
shift condition, the XCD-pair column-block map, the LDS state-tile transposition, the plane
addresses of the permlane32-swap stores, the weight pack order.  The GPU tests
(tests/test_gpu_wino.py) hold the kernel itself."""
import numpy as np
import pytest

F16 = np.float16


def split_planes(v, scale=256.0):
  """fp32 -> (hi, lo) fp16 planes of scale * v (convlstm_f16x3.h split_planes_item)."""
  s = (np.asarray(v, dtype=np.float32) * np.float32(scale)).astype(np.float32)
  hi = s.astype(F16)
  lo = (s - hi.astype(np.float32)).astype(F16)
  return hi, lo


def two_sum_planes(a_hi, a_lo, b_hi, b_lo, sub):
  """wn_combine: (a_hi + a_lo) +- (b_hi + b_lo) as a plane pair, every operation rounded to
  fp16 like v_pk_add_f16 / v_pk_fma_f16(x, -1, y) (a - b is ONE rounding either way)."""
  bh = -b_hi if sub else b_hi
  bl = -b_lo if sub else b_lo
  s = (a_hi + bh).astype(F16)
  bb = (s - a_hi).astype(F16)
  t = (s - bb).astype(F16)
  e1 = (a_hi - t).astype(F16)
  e2 = (bh - bb).astype(F16)
  lo = ((e1 + e2).astype(F16) + (a_lo + bl).astype(F16)).astype(F16)
  return s, lo


def test_two_sum_of_plane_pairs_is_exact_to_the_residual_class():
  rng = np.random.default_rng(0)
  a = np.tanh(rng.normal(size=20000) * 2).astype(np.float32)
  b = np.tanh(rng.normal(size=20000) * 2).astype(np.float32)
  b[::3] = a[::3] * np.float32(1.0 + 2.0 ** -9)          # near-cancellation
  a[1::7] *= np.float32(1e-3)                             # low plane subnormal
  ah, al = split_planes(a)
  bh, bl = split_planes(b)
  for sub in (False, True):
    hi, lo = two_sum_planes(ah, al, bh, bl, sub)
    exact = (ah.astype(np.float64) + al.astype(np.float64)) + \
        (-1.0 if sub else 1.0) * (bh.astype(np.float64) + bl.astype(np.float64))
    got = hi.astype(np.float64) + lo.astype(np.float64)
    # TwoSum of the high planes is error-free: what is lost is one fp16 rounding of the LOW part
    scale = 256.0 * np.maximum(np.abs(a), np.abs(b)).astype(np.float64)
    err = np.abs(got - exact)
    assert (err <= 2.0 ** -20 * np.maximum(scale, 2.0 ** -4)).all(), err.max()
    # ... and the high plane alone is the correctly rounded sum of the high planes
    assert (hi == (ah.astype(np.float32) + (-1 if sub else 1) * bh.astype(np.float32)).astype(F16)).all()


def _conv3x3_same(d, w):
  """d [H, W, Ci], w [3, 3, Ci, N] (fp64) -> [H, W, N], out(y, x) = sum d(y+ky-1, x+kx-1) w[ky, kx]."""
  H, W, Ci = d.shape
  pad = np.zeros((H + 2, W + 2, Ci))
  pad[1:-1, 1:-1] = d
  out = np.zeros((H, W, w.shape[3]))
  for ky in range(3):
    for kx in range(3):
      out += pad[ky:ky + H, kx:kx + W] @ w[ky, kx]
  return out


def _wino_rows_f16x3(d, w):
  """The kernel's arithmetic for one image: rows paired, dx direct; operands as fp16 plane
  pairs, products a0 w0 + a0 w1 + a1 w0 with fp32 accumulation (pairwise here), output
  transform in fp32, result / 2^16."""
  H, W, Ci = d.shape
  N = w.shape[3]
  Hp = (H + 1) // 2
  # kernel transform in fp64, planes of 256 U (pack_wino_kernel)
  g0, g1, g2 = w[0], w[1], w[2]                       # [3 dx][Ci][N] each
  U = [g0, 0.5 * (g0 + g1 + g2), 0.5 * (g0 - g1 + g2), g2]
  Uh, Ul = [], []
  for u in U:
    sv = u * 256.0
    h = sv.astype(F16)
    Uh.append(h)
    Ul.append((sv - h.astype(np.float64)).astype(F16))
  dh, dl = split_planes(d)
  zero = np.zeros((W, Ci), F16)

  def row(r):
    return (dh[r], dl[r]) if 0 <= r < H else (zero, zero)

  out = np.zeros((H, W, N), np.float32)
  for t in range(Hp):
    rm1, r0, r1, r2 = row(2 * t - 1), row(2 * t), row(2 * t + 1), row(2 * t + 2)
    V = [two_sum_planes(rm1[0], rm1[1], r1[0], r1[1], True),      # d(-1) - d(+1)
         two_sum_planes(r0[0], r0[1], r1[0], r1[1], False),       # d(0) + d(+1)
         two_sum_planes(r1[0], r1[1], r0[0], r0[1], True),        # d(+1) - d(0)
         two_sum_planes(r0[0], r0[1], r2[0], r2[1], True)]        # d(0) - d(+2)
    M = []
    for c in range(4):
      acc = np.zeros((W, N), np.float32)
      for dx in range(3):
        vh = np.zeros((W, Ci), np.float32)
        vl = np.zeros((W, Ci), np.float32)
        lo_x, hi_x = max(0, 1 - dx), min(W, W + 1 - dx)           # x + dx - 1 inside the row
        vh[lo_x:hi_x] = V[c][0][lo_x + dx - 1:hi_x + dx - 1]
        vl[lo_x:hi_x] = V[c][1][lo_x + dx - 1:hi_x + dx - 1]
        uh, ul = Uh[c][dx].astype(np.float32), Ul[c][dx].astype(np.float32)
        acc += (vl @ uh).astype(np.float32)
        acc += (vh @ ul).astype(np.float32)
        acc += (vh @ uh).astype(np.float32)
      M.append(acc)
    y0 = (M[0] + M[1]) + M[2]
    y1 = (M[1] - M[2]) - M[3]
    out[2 * t] = y0 * np.float32(2.0 ** -16)
    if 2 * t + 1 < H:
      out[2 * t + 1] = y1 * np.float32(2.0 ** -16)
  return out


@pytest.mark.parametrize("H,W", [(18, 32), (9, 16), (6, 8), (2, 32)])
def test_winograd_row_pairs_with_f16x3_operands_equal_the_direct_convolution(H, W):
  rng = np.random.default_rng(H * 100 + W)
  Ci, N = 48, 24
  d = np.tanh(rng.normal(size=(H, W, Ci))).astype(np.float32)
  w = (rng.normal(size=(3, 3, Ci, N)) * 0.06).astype(np.float32)
  ref = _conv3x3_same(d.astype(np.float64), w.astype(np.float64))
  got = _wino_rows_f16x3(d, w.astype(np.float64))
  err = np.abs(got - ref).max()
  print("H %d W %d: max |winograd f16x3 - fp64 direct| = %.3g (max |y| %.3g)" % (H, W, err,
                                                                             np.abs(ref).max()))
  assert err < 4e-6 * max(1.0, np.abs(ref).max())


def test_one_hot_taps_land_on_the_right_output_rows():
  """Every (ky, kx) tap, even and odd source rows: the row pairing and the sign pattern of the
  output transform (what a transposed or mis-signed component would break)."""
  H, W, Ci, N = 6, 8, 16, 4
  for ky in range(3):
    for kx in range(3):
      for (sy, sx) in ((2, 5), (3, 2), (0, 0), (5, 7)):
        d = np.zeros((H, W, Ci), np.float32)
        d[sy, sx, 3] = 0.75
        w = np.zeros((3, 3, Ci, N))
        w[ky, kx, 3, 1] = 0.5
        ref = _conv3x3_same(d.astype(np.float64), w)
        got = _wino_rows_f16x3(d, w)
        assert np.abs(got - ref).max() < 1e-7, (ky, kx, sy, sx)


# ------------------------------------------------------------------ index maps

def _pair_cell(q, H, W):
  Hp = (H + 1) // 2
  Kp = Hp * W
  r, pc = divmod(q, Kp)
  t, x = divmod(pc, W)
  return r, 2 * t, x


@pytest.mark.parametrize("rows,H,W", [(3, 18, 32), (5, 9, 16), (2, 6, 8), (1, 9, 16)])
def test_pair_cell_tiling_covers_every_cell_once_and_keeps_the_dpp_shift_legal(rows, H, W):
  Hp = (H + 1) // 2
  Q = rows * Hp * W
  seen = np.zeros((rows, H, W), np.int32)
  for q in range(Q):
    r, y0, x = _pair_cell(q, H, W)
    for e in (0, 1):
      if y0 + e < H:
        seen[r, y0 + e, x] += 1
    # lanes of a wave tile are 32 consecutive q: W | 32 => lane % 32 == 0 is x == 0, so the
    # lane a wave_shr:1 / wave_shl:1 brings in across a tile or half-wave edge is always masked
    lane = q % 32
    assert x == lane % W
    if lane == 0:
      assert x == 0
    if lane == 31:
      assert x == W - 1
  assert (seen == 1).all()


def test_column_block_map_puts_the_halves_of_a_state_line_on_one_xcd():
  ncb = 16                                            # C = 256
  for mtiles in (1, 3, 72):
    got = {}
    for block in range(ncb * mtiles):
      grp, w16 = divmod(block, 16)
      cb16 = (grp % (ncb // 16)) * 16 + 2 * (w16 & 7) + (w16 >> 3)
      mt = grp // (ncb // 16)
      assert (cb16, mt) not in got
      got[(cb16, mt)] = block % 8                     # XCD of the workgroup
    assert len(got) == ncb * mtiles
    for (cb16, mt), xcd in got.items():
      assert xcd == cb16 // 2                         # blocks 2x, 2x+1 = one 128-byte line


def test_lds_state_tile_transposition_is_a_bijection_onto_cell_rows():
  """Epilogue: lane (col, half) writes 16 B at [e*32 + col][rb*8 + half*4]; transposed, lane l
  moves row (l >> 2) + 16 k, piece l & 3.  Both cover the 64 x 16-float tile exactly once, and
  the transposed row belongs to the pair-cell of lane (row & 31), e = row >> 5."""
  tile = np.zeros((64, 16), np.int32)
  for lane in range(64):
    col, half = lane & 31, lane >> 5
    for e in (0, 1):
      for rb in (0, 1):
        idx = col * 16 + half * 4 + e * 512 + rb * 8
        assert idx % 4 == 0
        tile.reshape(-1)[idx:idx + 4] += 1
  assert (tile == 1).all()
  tile[:] = 0
  for lane in range(64):
    for k in range(4):
      idx = (lane >> 2) * 16 + (lane & 3) * 4 + k * 256
      tile.reshape(-1)[idx:idx + 4] += 1
      row = idx // 16
      assert row == (lane >> 2) + 16 * k
      assert row & 31 == (lane >> 2) + 16 * (k & 1) and row >> 5 == k >> 1   # bpermute source, e
  assert (tile == 1).all()


def _plane_index(m, c, C):
  return ((m >> 5) * (C >> 4) + (c >> 4)) * 512 + (((c >> 3) & 1) * 256 + (m & 31) * 8 + (c & 7))


def test_plane_store_addresses_of_the_half_wave_swap():
  """After v_permlane32_swap the lower half-wave holds channels 0-7 of row block rb for the
  e = 0 cells, the upper half-wave for the e = 1 cells; the 16-byte store lands at
  plane_index(cell, cb16*16 + rb*8, C)."""
  C, H, W = 256, 18, 32
  HW = H * W
  for cb16 in (0, 5, 15):
    for (r, y0) in ((0, 0), (2, 16)):
      for rb in (0, 1):
        for half in (0, 1):
          addrs = []
          for col in range(32):
            mc = r * HW + (y0 + half) * W + col
            o0 = ((mc >> 5) * (C >> 4) + cb16) * 512 + (mc & 31) * 8
            addrs.append(o0 + rb * 256)
            assert o0 + rb * 256 == _plane_index(mc, cb16 * 16 + rb * 8, C)
          # 32 lanes = one contiguous 512-byte run (8 halves = 16 bytes per lane)
          assert addrs == list(range(addrs[0], addrs[0] + 256, 8))


def test_weight_pack_order_matches_the_stage_reads():
  """pack_wino_kernel's element order against the kernel's LDS reads: vector
  ((((ci*3 + dx)*2 + plane)*2 + rb)*64 + lane of stage s = 2*chunk + (comp >> 1) holds, at
  element e, U_comp[dx][chunk*16 + 8*(lane >> 5) + e][gate (lane & 31) >> 3][channel
  cb16*16 + rb*8 + (lane & 7)]."""
  Cx16, C = 32, 64
  nxc, nst = Cx16 // 16, 2 * (Cx16 // 16 + C // 16)
  seen = set()
  total = (C // 16) * nst * 2 * 3 * 2 * 64 * 8
  for idx in range(0, total, 97):                     # a stride coprime to the layout
    e = idx & 7
    l = (idx >> 3) & 63
    rb = (idx >> 9) & 1
    t = idx >> 10
    dx = t % 3; t //= 3
    ci = t & 1; t >>= 1
    s, cb16 = t % nst, t // nst
    chunk, comp = s >> 1, (s & 1) * 2 + ci
    assert 0 <= comp < 4 and cb16 < C // 16
    is_x = chunk < nxc
    cin = (0 if is_x else Cx16) + (chunk if is_x else chunk - nxc) * 16 + 8 * (l >> 5) + e
    n = ((l & 31) >> 3) * C + cb16 * 16 + rb * 8 + (l & 7)
    base = ((((cb16 * nst + s) * 2 + ci) * 3 + dx) * (2 * 2 * 64 * 8))
    for plane in (0, 1):
      out = base + ((plane * 2 + rb) * 64 + l) * 8 + e
      vec_in_stage = (out - (cb16 * nst + s) * (2 * 3 * 2 * 2 * 64 * 8)) // 8
      assert vec_in_stage == (((ci * 3 + dx) * 2 + plane) * 2 + rb) * 64 + l
      assert (out, ) not in seen
      seen.add((out, ))
    assert 0 <= cin < Cx16 + C and 0 <= n < 4 * C


def test_f43_row_form_is_numerically_affordable_with_f16x3_operands():
  """The study behind DESIGN.md section 8 item 5a (tools/winograd_f43_numerics.py): F(4,3) over
  image rows -- 6 products per 4 output rows x 3 taps, half of the direct MFMAs -- with operands
  held as two fp16 planes costs about four times the F(2,3) form's error and stays far inside the
  1e-4 bar; its F(2,3) branch reproduces the direct convolution like the kernel's twin above."""
  import importlib.util
  import os
  spec = importlib.util.spec_from_file_location(
      "winograd_f43_numerics",
      os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools",
                   "winograd_f43_numerics.py"))
  m = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(m)
  # exact algebra first: with fp64 planes-free arithmetic A^T [(G g) . (B^T d)] is the correlation
  rng = np.random.default_rng(5)
  g = rng.normal(size=3)
  d = rng.normal(size=6)
  y = m.AT @ ((m.G @ g) * (m.BT @ d))
  assert np.allclose(y, [d[i:i + 3] @ g for i in range(4)], atol=1e-12)
  y2 = m.AT2 @ ((m.G2 @ g) * (m.BT2 @ d[:4]))
  assert np.allclose(y2, [d[i:i + 3] @ g for i in range(2)], atol=1e-12)
  # f16x3 operands, model-like magnitudes; H = 10 exercises a partial last 4-row tile
  H, W, Ci, N = 10, 8, 320, 64
  lim = np.sqrt(6.0 / (9 * Ci + 9 * 1024))
  w = rng.uniform(-lim, lim, size=(3, 3, Ci, N))
  x = np.tanh(rng.normal(size=(H, W, Ci)) * 1.5)
  ref = m.conv_direct64(x, w)
  e0 = np.abs(m.conv_direct_f16x3(x, w) - ref).max()
  e2 = np.abs(m.conv_wino_rows(x, w, 2) - ref).max()
  e4 = np.abs(m.conv_wino_rows(x, w, 4) - ref).max()
  print("direct f16x3 %.2e, F(2,3) rows %.2e, F(4,3) rows %.2e (max |pre-activation| %.2f)"
        % (e0, e2, e4, np.abs(ref).max()))
  assert e0 < 3e-6 and e2 < 5e-6 and e4 < 2e-5


# ------------------------------------------------------------------ F(3,3) over rows (round 5)
# csrc/convlstm_wino3.h: five products per THREE output rows and tap column (5/9 of the direct
# MFMAs; 18 = 6 x 3 and 9 = 3 x 3 rows: no partial tiles on the published grids).  Points
# 0, 1, -1, 2, inf; with d0..d4 = input rows 3t-1 .. 3t+3 and g0..g2 the kernel rows of one
# stencil column:
#   V0 = 2 (d0 - d2) + V3        U0 = g0 / 2                  y(3t)   = M0 + M1 + M2 + M3
#   V1 = (d3 - d2) - 2 d1        U1 = -(g0 + g1 + g2) / 2     y(3t+1) = M1 - M2 + 2 M3
#   V2 = 2 (d1 - d2) + (d3 - d2) U2 = (-g0 + g1 - g2) / 6     y(3t+2) = M1 + M2 + 4 M3 + M4
#   V3 = d3 - d1                 U3 = (g0 + 2 g1 + 4 g2) / 6
#   V4 = 2 V3 + (d2 - d4)        U4 = -g2
# Every V is a chain of wn_lin<KA, KB> steps (KA a + KB b on plane pairs, TwoSum of the high
# planes with the factors 2 folded into the fused multiply-adds).


def lin_planes(a_hi, a_lo, b_hi, b_lo, ka, kb):
  """wn_lin<KA, KB>: ka (a_hi + a_lo) + kb (b_hi + b_lo) as a plane pair; ka in {1, 2}, kb in
  {1, -1, -2}.  Every line is ONE v_pk_fma_f16 / v_pk_add_f16: the products by 1, 2 are exact,
  so each operation rounds once, to fp16."""
  ka, kb = F16(ka), F16(kb)
  s = (ka * a_hi + kb * b_hi).astype(F16)
  bb = (s - ka * a_hi).astype(F16)
  t = (s - bb).astype(F16)
  ne1 = (t - ka * a_hi).astype(F16)          # -(ka a_hi - t)
  ne2 = (bb - kb * b_hi).astype(F16)         # -(kb b_hi - bb)
  l1 = (ka * a_lo + kb * b_lo).astype(F16)
  ne = (ne1 + ne2).astype(F16)
  lo = (l1 - ne).astype(F16)
  return s, lo


def test_lin_planes_is_exact_to_the_residual_class():
  rng = np.random.default_rng(1)
  a = np.tanh(rng.normal(size=20000) * 2).astype(np.float32)
  b = np.tanh(rng.normal(size=20000) * 2).astype(np.float32)
  b[::3] = a[::3] * np.float32(1.0 + 2.0 ** -9)
  a[1::7] *= np.float32(1e-3)
  ah, al = split_planes(a)
  bh, bl = split_planes(b)
  for ka, kb in ((1, -1), (2, 1), (1, -2), (1, 1)):
    hi, lo = lin_planes(ah, al, bh, bl, ka, kb)
    exact = ka * (ah.astype(np.float64) + al.astype(np.float64)) + \
        kb * (bh.astype(np.float64) + bl.astype(np.float64))
    got = hi.astype(np.float64) + lo.astype(np.float64)
    scale = 256.0 * (abs(ka) * np.abs(a) + abs(kb) * np.abs(b)).astype(np.float64)
    err = np.abs(got - exact)
    assert (err <= 2.0 ** -19 * np.maximum(scale, 2.0 ** -4)).all(), (ka, kb, err.max())
    assert (hi == (ka * ah.astype(np.float32) + kb * bh.astype(np.float32)).astype(F16)).all()


def _wino3_transform(rows):
  """rows: five (hi, lo) plane pairs d0..d4 -> the five components, in the kernel's order of
  operations (convlstm_wino3.h wn3_transform)."""
  d0, d1, d2, d3, d4 = rows
  v3 = lin_planes(d3[0], d3[1], d1[0], d1[1], 1, -1)
  t = lin_planes(d0[0], d0[1], d2[0], d2[1], 1, -1)
  v0 = lin_planes(t[0], t[1], v3[0], v3[1], 2, 1)
  t3 = lin_planes(d3[0], d3[1], d2[0], d2[1], 1, -1)
  v1 = lin_planes(t3[0], t3[1], d1[0], d1[1], 1, -2)
  t4 = lin_planes(d1[0], d1[1], d2[0], d2[1], 1, -1)
  v2 = lin_planes(t4[0], t4[1], t3[0], t3[1], 2, 1)
  t5 = lin_planes(d2[0], d2[1], d4[0], d4[1], 1, -1)
  v4 = lin_planes(v3[0], v3[1], t5[0], t5[1], 2, 1)
  return [v0, v1, v2, v3, v4]


def _wino3_kernel_transform(w):
  g0, g1, g2 = w[0], w[1], w[2]
  return [0.5 * g0, -0.5 * (g0 + g1 + g2), (-g0 + g1 - g2) / 6.0, (g0 + 2.0 * g1 + 4.0 * g2) / 6.0,
          -g2]


def _wino3_rows_f16x3(d, w):
  """The F(3,3) kernel's arithmetic for one image (same conventions as _wino_rows_f16x3)."""
  H, W, Ci = d.shape
  N = w.shape[3]
  Ht = (H + 2) // 3
  Uh, Ul = [], []
  for u in _wino3_kernel_transform(w):
    sv = u * 256.0
    h = sv.astype(F16)
    Uh.append(h)
    Ul.append((sv - h.astype(np.float64)).astype(F16))
  dh, dl = split_planes(d)
  zero = np.zeros((W, Ci), F16)

  def row(r):
    return (dh[r], dl[r]) if 0 <= r < H else (zero, zero)

  out = np.zeros((H, W, N), np.float32)
  for t in range(Ht):
    V = _wino3_transform([row(3 * t - 1 + i) for i in range(5)])
    M = []
    for c in range(5):
      acc = np.zeros((W, N), np.float32)
      for dx in range(3):
        vh = np.zeros((W, Ci), np.float32)
        vl = np.zeros((W, Ci), np.float32)
        lo_x, hi_x = max(0, 1 - dx), min(W, W + 1 - dx)
        vh[lo_x:hi_x] = V[c][0][lo_x + dx - 1:hi_x + dx - 1]
        vl[lo_x:hi_x] = V[c][1][lo_x + dx - 1:hi_x + dx - 1]
        uh, ul = Uh[c][dx].astype(np.float32), Ul[c][dx].astype(np.float32)
        acc += (vl @ uh).astype(np.float32)
        acc += (vh @ ul).astype(np.float32)
        acc += (vh @ uh).astype(np.float32)
      M.append(acc)
    two, four = np.float32(2.0), np.float32(4.0)
    ys = [((M[0] + M[1]) + (M[2] + M[3])),
          ((M[1] - M[2]) + two * M[3]),
          ((M[1] + M[2]) + (four * M[3] + M[4]))]
    for e in range(3):
      if 3 * t + e < H:
        out[3 * t + e] = ys[e] * np.float32(2.0 ** -16)
  return out


@pytest.mark.parametrize("H,W", [(18, 32), (9, 16), (6, 8), (7, 8), (3, 32), (4, 16)])
def test_winograd_row_triples_with_f16x3_operands_equal_the_direct_convolution(H, W):
  rng = np.random.default_rng(H * 100 + W + 7)
  Ci, N = 48, 24
  d = np.tanh(rng.normal(size=(H, W, Ci))).astype(np.float32)
  w = (rng.normal(size=(3, 3, Ci, N)) * 0.06).astype(np.float32)
  ref = _conv3x3_same(d.astype(np.float64), w.astype(np.float64))
  got3 = _wino3_rows_f16x3(d, w.astype(np.float64))
  got2 = _wino_rows_f16x3(d, w.astype(np.float64))
  e3, e2 = np.abs(got3 - ref).max(), np.abs(got2 - ref).max()
  print("H %d W %d: max |F(3,3) f16x3 - fp64 direct| = %.3g (F(2,3): %.3g; max |y| %.3g)" % (
      H, W, e3, e2, np.abs(ref).max()))
  assert e3 < 1.2e-5 * max(1.0, np.abs(ref).max())


def test_f33_one_hot_taps_land_on_the_right_output_rows():
  """Every (ky, kx) tap and every row class of the 3-row tile (source rows 3t, 3t+1, 3t+2, the
  image's first and last rows): what a transposed or mis-signed component would break."""
  H, W, Ci, N = 9, 8, 16, 4
  for ky in range(3):
    for kx in range(3):
      for (sy, sx) in ((3, 5), (4, 2), (5, 1), (0, 0), (8, 7), (2, 3)):
        d = np.zeros((H, W, Ci), np.float32)
        d[sy, sx, 3] = 0.75
        w = np.zeros((3, 3, Ci, N))
        w[ky, kx, 3, 1] = 0.5
        ref = _conv3x3_same(d.astype(np.float64), w)
        got = _wino3_rows_f16x3(d, w)
        assert np.abs(got - ref).max() < 1e-6, (ky, kx, sy, sx, np.abs(got - ref).max())


# ---- layout / schedule bookkeeping of csrc/convlstm_wino3.h (numpy restatements of the index
# arithmetic the kernel's correctness rests on; the GPU tests hold the kernel itself)

def test_f33_stage_sequence_walks_every_component_of_every_chunk_once():
  """The (chunk, component) pairs are ONE sequence g = 0 .. 5 n - 1; an LDS stage holds g = 2 s,
  2 s + 1; the kernel unrolls five stages with the COMPILE-TIME accumulator pairs (0,1) (2,3)
  (4,0) (1,2) (3,4) per two chunks and ends an odd chunk count on (0,1) (2,3) (4)."""
  pairs = [(0, 1), (2, 3), (4, 0), (1, 2), (3, 4)]
  for nck in range(1, 20):
    seq = []
    for _ in range(nck // 2):
      seq += pairs
    if nck & 1:
      seq += [(0, 1), (2, 3), (4, -1)]
    flat = [c for p in seq for c in p if c >= 0]
    assert flat == [g % 5 for g in range(5 * nck)]                 # accumulator of component g
    assert len(seq) == (5 * nck + 1) // 2                          # stages
    # slot ci of component g inside its stage = g & 1; stage bytes 24 KB = 2 x 768 vectors
    for g, c in enumerate(flat):
      assert (g >> 1, g & 1) == divmod(g, 2)


def test_f33_pack_layout_is_the_stage_image_the_kernel_reads():
  """pack_wino3_kernel writes [cb][chunk][comp][dx][plane][rb][lane][8]; the kernel's stage st of
  a column block starts at vector (ck_lo * 5 + 2 st) * 768 and reads fragment (slot ci, dx, plane,
  rb) at ((ci * 3 + dx) * 2 + plane) * 2 + rb) * 64 + lane."""
  nrb, nch = 2, 3
  comp_vec = 3 * 2 * nrb * 64

  def pack_vec(cb, chunk, comp, dx, plane, rb, lane):
    return ((((((cb * nch + chunk) * 5 + comp) * 3 + dx) * 2 + plane) * nrb + rb) * 64) + lane

  assert comp_vec == 768
  for ck_lo in (0, 1):
    for g in range(5 * (nch - ck_lo)):
      st, ci = g >> 1, g & 1
      chunk, comp = ck_lo + g // 5, g % 5
      for dx in range(3):
        for plane in range(2):
          for rb in range(nrb):
            kernel_vec = (1 * nch + ck_lo) * 5 * comp_vec + st * 2 * comp_vec + \
                (((ci * 3 + dx) * 2 + plane) * nrb + rb) * 64 + 17
            assert kernel_vec == pack_vec(1, chunk, comp, dx, plane, rb, 17)


@pytest.mark.parametrize("H,W,rows", [(18, 32, 3), (9, 16, 5), (36, 18, 2), (7, 33, 2)])
def test_f33_operand_addresses_of_pre_pass_and_gate_kernel_agree(H, W, rows):
  """wino3_transform_kernel writes triple-cell q, channel group cg, component c, plane p, k half
  hf at vector ((q >> 5) * KG + cg) * 640 + (c * 2 + p) * 64 + hf * 32 + (q & 31); the gate
  kernel reads, for lane l = hf * 32 + column, tile base (q_wave >> 5) * KG * 10240 bytes +
  cg * 10240 + c * 2048 + p * 1024 + l * 16 (plain tiling) or, per lane, (q >> 5) * KG * 10240 +
  ((q & 31) + 32 hf) * 16 (halo tiling: q = 30 t - 1 + column)."""
  KG = 3
  Kt = ((H + 2) // 3) * W
  Q = rows * Kt
  halo = 32 % W != 0
  own = 30 if halo else 32

  def written(q, cg, c, p, hf):
    return (((q >> 5) * KG + cg) * 640 + (c * 2 + p) * 64 + hf * 32 + (q & 31)) * 16

  seen = set()
  ntile = (Q + own - 1) // own
  for t in range(ntile):
    q_own = t * own
    q_wave = q_own - 1 if halo else q_own
    for lane in range(64):
      col, hf = lane & 31, lane >> 5
      q = q_wave + col
      if not (0 <= q < Q):
        continue
      for cg in range(KG):
        for c in range(5):
          for p in range(2):
            so = cg * 10240 + c * 2048 + p * 1024
            if halo:
              addr = (q >> 5) * KG * 10240 + ((q & 31) + 32 * hf) * 16 + so
            else:
              addr = (q_wave >> 5) * KG * 10240 + lane * 16 + so
            assert addr == written(q, cg, c, p, hf)
      if (not halo) or 1 <= col <= 30:
        seen.add(q)
  assert seen == set(range(Q))                      # every triple-cell is owned by exactly one lane


def test_f33_xcd_map_modes_cover_every_block_once():
  """Block -> (column block, row tile) of convlstm_step_wino3_kernel: mode 1 (two adjacent column
  blocks per XCD), modes 2 / 3 (four / eight column blocks per XCD, every second / fourth row
  tile; the launch rounds the row tiles up to whole groups)."""
  ncb = 16
  for mode, per in ((1, 2), (2, 4), (3, 8)):
    for mtiles in (1, 2, 5, 8):
      nt = 1 if mode == 1 else per // 2
      mt_padded = (mtiles + nt - 1) // nt * nt
      nblocks = mt_padded * ncb
      seen = {}
      for block in range(nblocks):
        if mode == 1:
          grp, w16 = divmod(block, 16)
          cb = (grp % (ncb // 16)) * 16 + 2 * (w16 & 7) + (w16 >> 3)
          mt = grp // (ncb // 16)
        else:
          grp, w = divmod(block, 16 * nt)
          xcd, j = w & 7, w >> 3
          cb = per * (xcd % (16 // per)) + j
          mt = grp * nt + xcd // (16 // per)
        assert 0 <= cb < ncb and 0 <= mt < mt_padded
        assert (cb, mt) not in seen
        seen[(cb, mt)] = block % 8                   # the XCD (round-robin dispatch)
      assert len(seen) == nblocks
      # the XCDs that touch a row tile (= fetch its operands): 8, 4, 2
      for mt in range(mt_padded):
        xcds = {x for (cb, m), x in seen.items() if m == mt}
        assert len(xcds) == 16 // per, (mode, mt, xcds)


def test_fused_lstm_update_of_the_f33_inference_epilogue():
  """csrc/convlstm_wino3.h (inference, no gates to save): c' and h' over common denominators --
  five exponentials and TWO reciprocals per element instead of five and five.  fp32 twin against
  the fp64 definition, incl. saturated gates (the +-28 clamp keeps D_i D_j D_f finite)."""
  rng = np.random.default_rng(0)
  n = 200000
  pre = (rng.normal(size=(4, n)) * 4).astype(np.float32)
  pre[0, :50] = -80.0; pre[2, 50:100] = 90.0; pre[3, 100:150] = -100.0; pre[1, 150:200] = 50.0
  pre[:, 200:250] = -70.0; pre[:, 250:300] = 70.0
  c = (rng.normal(size=n) * 3).astype(np.float32)
  f32 = np.float32
  L = f32(1.4426950408889634)
  xi = np.clip(pre[0], -28, 28).astype(f32)
  xf = np.clip(pre[2] + f32(1.0), -28, 28).astype(f32)
  xo = np.clip(pre[3], -28, 28).astype(f32)
  Di = (f32(1) + np.exp2(-L * xi).astype(f32)).astype(f32)
  Df = (f32(1) + np.exp2(-L * xf).astype(f32)).astype(f32)
  Do = (f32(1) + np.exp2(-L * xo).astype(f32)).astype(f32)
  Ej = np.exp2(f32(-2) * L * np.abs(pre[1])).astype(f32)
  Dj = (f32(1) + Ej).astype(f32)
  tn = np.copysign((f32(1) - Ej).astype(f32), pre[1]).astype(f32)
  DiDj = (Di * Dj).astype(f32)
  R = (f32(1) / (DiDj * Df).astype(f32)).astype(f32)
  cn = ((c * DiDj + (tn * Df).astype(f32)).astype(f32) * R).astype(f32)
  Ec = np.exp2(f32(-2) * L * np.abs(cn)).astype(f32)
  hn = (np.copysign((f32(1) - Ec).astype(f32), cn) *
        (f32(1) / ((f32(1) + Ec) * Do).astype(f32)).astype(f32)).astype(f32)
  p64, c64 = pre.astype(np.float64), c.astype(np.float64)
  sig = lambda x: 1.0 / (1.0 + np.exp(-x))
  rc = sig(p64[2] + 1.0) * c64 + sig(p64[0]) * np.tanh(p64[1])
  rh = np.tanh(rc) * sig(p64[3])
  assert np.isfinite(cn).all() and np.isfinite(hn).all()
  assert np.abs(cn - rc).max() < 4e-6 and np.abs(hn - rh).max() < 1e-6
