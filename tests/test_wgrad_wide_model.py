"""CPU twin of the wide f16x3 wgrad tile (csrc/convlstm_wgrad_f16x3.h,
`convlstm_wgrad_f16x3_wide_kernel`): the kernel's index arithmetic restated thread by thread in
numpy -- block decode, the per-thread copy slots and their global addresses in the
cell-contiguous plane layout, the swizzled LDS image, the fragment reads of each lane, the
accumulator -> (channel, column) map of the 32x32x16 MFMA, the split ranges -- and held against
the plain definition  dW[tap][ci][n] = sum_m h[m + d_tap][ci] G[m][n]  (reference
code/pred_models.py:1694-1717: tf.gradients of the ConvLSTM kernel).  No GPU: this is the
bookkeeping the GPU parity tests (tests/test_gpu_train.py, test_gpu_at_size.py) then run for
real."""
import numpy as np
import pytest

A_ROWS, G_ROWS = 128, 256


def swz(row):
  return ((row >> 1) & 3) ^ ((row >> 3) & 1)


# lane groups of ds_read_b128 / ds_write_b128 (MI355X guide, LDS table)
READ_GROUPS = [
    list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
]
READ_GROUPS += [[l + 32 for l in g] for g in READ_GROUPS]


def test_wgrad_wide_lds_swizzle_is_conflict_free():
  """Fragment read: lane (li, hf) reads 16 B at row base + li, chunk kk*2 + hf.  Every 16-lane
  group must touch 16 different 16-byte slots of the 256-byte bank row; a store group of 8
  consecutive threads (two rows x four chunks) 8 different slots of a 128-byte bank row."""
  for base in (0, 32, 64, 96, 224):
    for kk in range(2):
      for grp in READ_GROUPS:
        slots = set()
        for lane in grp:
          li, hf = lane & 31, lane >> 5
          row = base + li
          byte = (row * 4 + ((kk * 2 + hf) ^ swz(li))) * 16
          slots.add((byte % 256) // 16)
        assert len(slots) == 16
  for t0 in range(0, 256, 8):
    slots = set()
    for tid in range(t0, t0 + 8):
      trow, vec = tid >> 2, tid & 3
      byte = (trow * 4 + (vec ^ swz(trow))) * 16
      slots.add((byte % 128) // 16)
    assert len(slots) == 8
  # rows r and r + 64 k share the swizzle (the copy slots of a thread differ by 64 rows)
  assert all(swz(r) == swz(r + 64) == swz(r + 128) for r in range(64))


def decode(b, C, map_mode):
  """blockIdx -> (split, tap, channel block, column block), the kernel's three XCD maps."""
  ncib, nnb = C // A_ROWS, 4 * C // G_ROWS
  xcd, j = b & 7, b >> 3
  if map_mode == 1:
    tps = 9 * ncib * nnb
    split = xcd + 8 * (j // tps)
    j %= tps
    tap, j = j % 9, j // 9
    return split, tap, j % ncib, j // ncib
  if map_mode == 2:
    tps = 9 * nnb
    split = (xcd >> 1) + 4 * (j // tps)
    j %= tps
    return split, j % 9, xcd & 1, j // 9
  gpx = (ncib * nnb) >> 3
  group = xcd + 8 * (j % gpx)
  j //= gpx
  return j // 9, j % 9, group % ncib, group // ncib


@pytest.mark.parametrize("C,nsplit,map_mode", [(256, 7, 0), (256, 21, 0), (512, 7, 0), (256, 3, 0),
                                               (256, 24, 1), (512, 8, 1), (256, 20, 2),
                                               (256, 4, 2)])
def test_wgrad_wide_block_decode_covers_every_tile_of_every_split_once(C, nsplit, map_mode):
  ncib, nnb = C // A_ROWS, 4 * C // G_ROWS
  nblocks = nsplit * 9 * ncib * nnb
  seen = set()
  per_xcd = [0] * 8
  streams = [set() for _ in range(8)]      # (operand, block, split) an XCD's L2 sees
  for b in range(nblocks):
    split, tap, cib, nb = decode(b, C, map_mode)
    assert 0 <= split < nsplit and 0 <= tap < 9 and cib < ncib and nb < nnb
    seen.add((split, tap, cib, nb))
    per_xcd[b & 7] += 1
    streams[b & 7].add(("G", nb, split))
    streams[b & 7].add(("A", cib, split))
  assert len(seen) == nblocks
  assert len(set(per_xcd)) == 1          # every XCD the same number of workgroups
  # operand bytes through the eight L2s, in units of (one column block | channel block) x split:
  # G blocks weigh 256 columns, A blocks 3 x 128 rows
  traffic = sum(256 if o == "G" else 384 for st in streams for (o, _, _) in st)
  unique = nsplit * (nnb * 256 + ncib * 384)
  ratio = traffic / unique
  if map_mode == 1:
    assert ratio == 1.0
  elif map_mode == 2:
    assert ratio < 1.6
  elif C == 256:
    assert 2.8 < ratio < 2.9             # every XCD streams a quarter of G and half of A: 20 / 7


def _planes(v):
  """fp32 -> (hi, lo) fp16 planes as float64 (the kernels' split: hi = f16(v), lo = f16(v - hi))."""
  hi = v.astype(np.float16)
  lo = (v - hi.astype(np.float32)).astype(np.float16)
  return hi.astype(np.float64), lo.astype(np.float64)


def _transpose_split(src, Mtot, Cc, Mrow, W, dx, e):
  """transpose_split_kernel: out[plane][m >> 5][c][m & 31] = src[m + dx][c] * 2^e, zero where
  x(m) + dx leaves the image row."""
  out = np.zeros((2, Mrow // 32, Cc, 32))
  m = np.arange(Mtot)
  x = m % W
  ok = (x + dx >= 0) & (x + dx < W) & (m + dx >= 0) & (m + dx < Mtot)
  v = np.zeros((Mtot, Cc), np.float32)
  v[ok] = src[(m + dx)[ok]] * np.float32(2.0 ** e)
  hi, lo = _planes(v)
  out[0, m >> 5, :, m & 31] = hi
  out[1, m >> 5, :, m & 31] = lo
  return out.reshape(2, -1)            # [plane][(m >> 5) * Cc * 32 + c * 32 + (m & 31)]


def _wide_kernel_model(at, gt, Mrow, ksteps_total, H, W, C, ksteps_per_split, nsplit, blocks,
                       map_mode=0, xrows=False, Ca=None):
  """The kernel, block by block.  h rows: {(split, tap, ci0, n0): [128][256] partial}; x rows
  (tile rows enumerate (tap, channel) pairs): {(split, rb, n0): [128][256]}."""
  N4 = 4 * C
  Ca = C if Ca is None else Ca
  wk = W // 16
  nnb = N4 // G_ROWS
  nrb = (9 * Ca + A_ROWS - 1) // A_ROWS
  tid = np.arange(256)
  vec, trow = tid & 3, tid >> 2
  wslot = trow * 4 + (vec ^ swz(trow))            # in 16-byte slots; + 64 rows: + 256 slots
  out = {}
  for b in blocks:
    if xrows:
      j = b
      nb, j = j % nnb, j // nnb
      rb, split = j % nrb, j // nrb
      n0 = nb * G_ROWS
    else:
      split, tap, cib, nb = decode(b, C, map_mode)
      ci0, n0 = cib * A_ROWS, nb * G_ROWS
    # copy slot k of a thread: tile row trow + 64 k -> (operand copy, channel row, row shift)
    slot_dx, slot_ci, slot_dy = [], [], []
    for k in range(2):
      if xrows:
        R = rb * A_ROWS + k * 64 + trow
        live = R < 9 * Ca
        tp = np.where(live, R // Ca, 4)
        slot_ci.append(np.where(live, R - tp * Ca, 0))
        slot_dy.append(np.where(live, tp // 3 - 1, 1 << 20))
        slot_dx.append(tp % 3)
      else:
        slot_ci.append(ci0 + k * 64 + trow)
        slot_dy.append(np.full(256, tap // 3 - 1))
        slot_dx.append(np.full(256, tap % 3))
    ks0 = split * ksteps_per_split
    ks1 = min(ks0 + ksteps_per_split, ksteps_total)
    nstages = (ks1 - ks0 + 1) // 2 if ks1 > ks0 else 0
    acc = np.zeros((A_ROWS, G_ROWS))
    kin = ks0 % (H * wk)
    ly, lxk = kin // wk, kin % wk
    for st in range(nstages):
      ksb = ks0 + 2 * st
      ys = []
      for _ in range(2):
        ys.append(ly)
        lxk += 1
        if lxk == wk:
          lxk = 0
          ly = (ly + 1) % H
      inr = [ksb < ks1, ksb + 1 < ks1]
      if xrows:
        nv = inr
      else:
        dy = tap // 3 - 1
        nv = [inr[0] and 0 <= ys[0] + dy < H, inr[1] and 0 <= ys[1] + dy < H]
      m0 = ksb * 16
      go = m0 * N4
      ysel = np.where(vec & 2, ys[1], ys[0])
      insel = np.where(vec & 2, inr[1], inr[0])
      ldsA = np.zeros((2, A_ROWS * 4, 8))
      ldsG = np.zeros((2, G_ROWS * 4, 8))
      for k in range(2):
        if xrows:
          ok = insel & (ysel + slot_dy[k] >= 0) & (ysel + slot_dy[k] < H)
        else:
          ok = np.where(vec & 2, nv[1], nv[0])
        cell = m0 + vec * 8 + np.where(ok, slot_dy[k] * W, 0)
        ao = (cell >> 5) * (Ca * 32) + (cell & 31)
        src = slot_ci[k] * 32 + ao
        for pl in range(2):
          vals = np.stack([at[d][pl][src[:, None] + np.arange(8)] for d in range(3)])
          v = vals[slot_dx[k], np.arange(256)]
          if xrows:
            v = np.where(ok[:, None], v, 0.0)
          ldsA[pl, k * 256 + wslot] = v
      for pl in range(2):
        for k in range(4):
          src = (n0 + trow) * 32 + vec * 8 + go + k * 2048
          ldsG[pl, k * 256 + wslot] = gt[pl][src[:, None] + np.arange(8)]
      # fragments: lane (li, hf) of k-step kk reads slot (row * 4 + ((kk*2+hf) ^ swz(li)))
      for kk in range(2):
        if not nv[kk]:
          continue
        fa = np.zeros((2, A_ROWS, 16))
        fg = np.zeros((2, G_ROWS, 16))
        for hf in range(2):
          for li in range(32):
            c = (kk * 2 + hf) ^ swz(li)
            for base in range(0, A_ROWS, 32):
              fa[:, base + li, hf * 8:hf * 8 + 8] = ldsA[:, (base + li) * 4 + c]
            for base in range(0, G_ROWS, 32):
              fg[:, base + li, hf * 8:hf * 8 + 8] = ldsG[:, (base + li) * 4 + c]
        # the three MFMAs of a product: a_lo g_hi + a_hi g_lo + a_hi g_hi
        acc += fa[1] @ fg[0].T + fa[0] @ fg[1].T + fa[0] @ fg[0].T
    out[(split, rb, n0) if xrows else (split, tap, ci0, n0)] = acc
  return out


@pytest.mark.parametrize("H,W,R,nsplit", [(3, 16, 3, 1), (2, 32, 4, 3), (5, 16, 2, 2)])
def test_wgrad_wide_index_arithmetic_gives_the_weight_gradient(H, W, R, nsplit):
  C = 256
  N4 = 4 * C
  rng = np.random.default_rng(17 + H * W + R)
  Mtot = R * H * W
  Mrow = (Mtot + 63) // 64 * 64
  h = rng.uniform(-1, 1, (Mtot, C)).astype(np.float32)
  g = (rng.standard_normal((Mtot, N4)) * 1e-3).astype(np.float32)
  a_exp = 8
  g_exp = 13 - int(np.floor(np.log2(np.abs(g).max())))
  at = [_transpose_split(h, Mtot, C, Mrow, W, d - 1, a_exp) for d in range(3)]
  gt = _transpose_split(g, Mtot, N4, Mrow, W, 0, g_exp)
  ksteps_total = Mtot // 16
  per = (ksteps_total + nsplit - 1) // nsplit
  per = (per + 1) & ~1
  nblocks = nsplit * 9 * (C // A_ROWS) * (N4 // G_ROWS)
  # every tap on one (channel block, column block) per test, all splits: blocks of XCD `xcd`
  # every tap and split of ONE (channel block, column block)
  pick = ((H + R) % 2, (W // 16 + R) % 4)
  blocks = [b for b in range(nblocks) if decode(b, C, 0)[2:] == pick]
  assert len(blocks) == 9 * nsplit
  parts = _wide_kernel_model(at, gt, Mrow, ksteps_total, H, W, C, per, nsplit, blocks)
  scale = 2.0 ** -(a_exp + g_exp)
  hd, gd = h.astype(np.float64), g.astype(np.float64)
  m = np.arange(Mtot)
  y, x = (m // W) % H, m % W
  taps_seen = set()
  sums = {}
  for (split, tap, ci0, n0), acc in parts.items():
    sums.setdefault((tap, ci0, n0), np.zeros((A_ROWS, G_ROWS)))
    sums[(tap, ci0, n0)] += acc * scale
    taps_seen.add(tap)
  assert taps_seen == set(range(9))
  for (tap, ci0, n0), got in sums.items():
    dy, dx = tap // 3 - 1, tap % 3 - 1
    ok = (y + dy >= 0) & (y + dy < H) & (x + dx >= 0) & (x + dx < W)
    src = (m + dy * W + dx)[ok]
    want = hd[src][:, ci0:ci0 + A_ROWS].T @ gd[m[ok]][:, n0:n0 + G_ROWS]
    err = np.abs(got - want).max() / max(np.abs(want).max(), 1e-30)
    assert err < 2e-6, (tap, ci0, n0, err)


@pytest.mark.parametrize("H,W,R,nsplit,Cx", [(3, 16, 3, 2, 64), (2, 32, 3, 1, 2), (4, 16, 2, 3, 32)])
def test_wgrad_wide_x_rows_tile_rows_enumerate_tap_channel_pairs(H, W, R, nsplit, Cx):
  """The x rows on the wide tile (XROWS): rows R = tap * Cx + ci, every copy slot with its own
  operand copy and row shift, invalid (row, k-step) pairs staged as zeros."""
  C = 256
  N4 = 4 * C
  rng = np.random.default_rng(5 + H * W + R + Cx)
  Mtot = R * H * W
  Mrow = (Mtot + 63) // 64 * 64
  x = (rng.standard_normal((Mtot, Cx)) * 3).astype(np.float32)
  g = (rng.standard_normal((Mtot, N4)) * 1e-3).astype(np.float32)
  a_exp = 13 - int(np.floor(np.log2(np.abs(x).max())))
  g_exp = 13 - int(np.floor(np.log2(np.abs(g).max())))
  at = [_transpose_split(x, Mtot, Cx, Mrow, W, d - 1, a_exp) for d in range(3)]
  gt = _transpose_split(g, Mtot, N4, Mrow, W, 0, g_exp)
  ksteps_total = Mtot // 16
  per = (ksteps_total + nsplit - 1) // nsplit
  per = (per + 1) & ~1
  nrb, nnb = (9 * Cx + A_ROWS - 1) // A_ROWS, N4 // G_ROWS
  nb_pick = (H + R) % nnb
  blocks = [b for b in range(nsplit * nrb * nnb) if b % nnb == nb_pick]
  parts = _wide_kernel_model(at, gt, Mrow, ksteps_total, H, W, C, per, nsplit, blocks,
                             xrows=True, Ca=Cx)
  scale = 2.0 ** -(a_exp + g_exp)
  got = np.zeros((nrb * A_ROWS, G_ROWS))
  for (split, rb, n0), acc in parts.items():
    assert n0 == nb_pick * G_ROWS
    got[rb * A_ROWS:(rb + 1) * A_ROWS] += acc * scale
  assert np.all(got[9 * Cx:] == 0)              # dead rows of the last tile
  xd, gd = x.astype(np.float64), g.astype(np.float64)
  m = np.arange(Mtot)
  y, xx = (m // W) % H, m % W
  for tap in range(9):
    dy, dx = tap // 3 - 1, tap % 3 - 1
    ok = (y + dy >= 0) & (y + dy < H) & (xx + dx >= 0) & (xx + dx < W)
    src = (m + dy * W + dx)[ok]
    want = xd[src].T @ gd[m[ok]][:, nb_pick * G_ROWS:(nb_pick + 1) * G_ROWS]
    err = np.abs(got[tap * Cx:(tap + 1) * Cx] - want).max() / max(np.abs(want).max(), 1e-30)
    assert err < 2e-6, (tap, err)
