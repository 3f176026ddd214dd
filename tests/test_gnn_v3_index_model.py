# coding=utf-8
"""CPU: the index scheme of the third graph-attention kernel (gnn_attend_v3_kernel,
multiverse_amd/csrc/kernels_misc.h) restated thread by thread in numpy and held against the
oracle's dense graph attention (gnn_np: code/pred_models.py:808-909 as the reference computes
it).  This is the model the kernel was written against before its first GPU run; it pins the
parts no GPU tolerance would localise: which cell a tile slot holds (clamped at both ends of the
tensor), the lane-linear DMA fill and the part swizzle of the node pass, the three row windows of
a quad and the one read that would fall in front of the tile, the halo cells' squared norms, the
masks.  LDS starts as NaN: a read of a byte nobody staged that reached a result would show.
The arithmetic ORDER is not modelled (sums run in fp64); the GPU tests hold the kernel itself to
the oracle (tests/test_gpu_kernels.py::test_gnn*)."""
import numpy as np
import pytest

from oracle import multiverse_oracle as oracle

CELLS, STAGE, TILE = 64, 128, 128 * 64

def run(h, sm, src_row=None):
  M, H, W, C = h.shape
  D = sm.shape[-1] if sm is not None else 0
  K = H * W
  Mtot = M * K
  hf = h.reshape(-1, C).astype(np.float32)
  smf = sm.reshape(-1, D).astype(np.float32) if D else None
  out = np.full((Mtot, C), np.nan, np.float32)
  ngroups = (Mtot + CELLS - 1) // CELLS
  for g in range(ngroups):
    m0 = g * CELLS
    own0 = W if CELLS % W == 0 else W + 1
    s0 = m0 - own0
    smem = np.full(2 * TILE + 1024, np.nan, np.float32)
    hsrc = np.zeros(STAGE, np.int64); ssrc = np.zeros(STAGE, np.int64)
    for t in range(STAGE):
      m = min(max(s0 + t, 0), Mtot - 1)
      r, c = divmod(m, K)
      hsrc[t] = (src_row[r] if src_row is not None else r) * K + c
      ssrc[t] = r * K + c
    def dma(q, tile, swz, scene=False):
      for tid in range(256):
        part, ls0 = tid & 15, tid >> 4
        for k in range(8):
          ls = ls0 + 16 * k
          if scene:
            src = smf[ssrc[ls], part * 4:part * 4 + 4]
          else:
            p = part
            if swz:
              p = part ^ ((ls0 >> 1) + 8 * (k & 1))
              assert ((ls >> 1) & 15) == (ls0 >> 1) + 8 * (k & 1)
            src = hf[hsrc[ls], q * 64 + p * 4: q * 64 + p * 4 + 4]
          slot = tid + 256 * k
          assert slot == ls * 16 + part
          smem[tile + slot * 4: tile + slot * 4 + 4] = src
    nchunk = 5 if D else 4
    pd = np.zeros((16, 16, 4, 9), np.float64)   # cp, sl, k, t   (fp64 accumulate: emulation of indices only)
    hq = np.zeros((16, 16, 4), np.float64)
    for c in range(nchunk):
      tile = (c & 1) * TILE
      if c < 4: dma(c, tile, False)
      else: dma(0, tile, False, scene=True)
      for tid in range(256):
        cp, sl = tid >> 4, tid & 15
        lq = own0 + cp * 4
        o0 = (lq - 1) * 64 + sl * 4
        om = o0 - W * 64
        om0 = sl * 4 if om < 0 else om
        def rd(off):
          a = tile + off
          assert 0 <= a and a + 4 <= smem.size, (a, g, tid)
          return smem[a:a + 4]
        r0 = [rd(o0 + j * 64) for j in range(6)]
        rm = [rd(om0 if j == 0 else om + j * 64) for j in range(6)]
        rp = [rd(o0 + W * 64 + j * 64) for j in range(6)]
        for k in range(4):
          own = r0[k + 1]
          for t, vec in ((3, r0[k]), (4, own), (5, r0[k + 2]), (0, rm[k]), (1, rm[k + 1]), (2, rm[k + 2]),
                         (6, rp[k]), (7, rp[k + 1]), (8, rp[k + 2])):
            pd[cp, sl, k, t] += float(np.dot(own.astype(np.float64), vec.astype(np.float64)))
        for i in range(4):
          hc = cp + 16 * i
          l = hc if hc < own0 else hc + CELLS
          v = rd(l * 64 + sl * 4).astype(np.float64)
          hq[cp, sl, i] += float(np.dot(v, v))
    ea = np.full((CELLS, 9), np.nan); ssq = np.full(STAGE, np.nan)
    for cp in range(16):
      for k in range(4):
        ea[cp * 4 + k] = pd[cp, :, k, :].sum(0)
        ssq[own0 + cp * 4 + k] = pd[cp, :, k, 4].sum()
      for i in range(4):
        hc = cp + 16 * i
        l = hc if hc < own0 else hc + CELLS
        ssq[l] = hq[cp, :, i].sum()
    alpha = np.zeros((CELLS, 9))
    for tid in range(CELLS):
      m = m0 + tid
      if m >= Mtot: continue
      cc = m % K; y, x = divmod(cc, W)
      li = own0 + tid
      e = np.full(9, -np.inf)
      for t in range(9):
        yy, xx = y + t // 3 - 1, x + t % 3 - 1
        if 0 <= yy < H and 0 <= xx < W:
          lj = li + (t // 3 - 1) * W + (t % 3 - 1)
          assert 0 <= lj < STAGE
          val = ea[tid, t] / np.sqrt(max(ssq[li], 1e-12)) / np.sqrt(max(ssq[lj], 1e-12))
          assert np.isfinite(val), (g, tid, t)
          e[t] = val
      w = np.exp(e - e.max()); w[~np.isfinite(e)] = 0
      alpha[tid] = w / w.sum()
    for q in range(4):
      c = nchunk + q
      tile = (c & 1) * TILE
      dma(q, tile, True)
      for tid in range(256):
        pi, c8 = tid & 31, tid >> 5
        lp = own0 + pi * 2
        for k in range(2):
          m2 = m0 + pi * 2 + k
          if m2 >= Mtot: continue
          node = np.zeros(8)
          for half in range(2):
            p = c8 * 2 + half
            for dy in (-1, 0, 1):
              for dx in (-1, 0, 1):
                l = lp - 1 + (k + 1 + dx) + dy * W
                l = min(max(l, 0), STAGE - 1)
                a = tile + l * 64 + ((p ^ ((l >> 1) & 15)) << 2)
                vec = smem[a:a + 4]
                assert np.isfinite(vec).all()
                wgt = alpha[pi * 2 + k, (dy + 1) * 3 + dx + 1] + (1.0 if (dy == 0 and dx == 0) else 0.0)
                node[half * 4:half * 4 + 4] += wgt * vec
          out[m2, q * 64 + c8 * 8: q * 64 + c8 * 8 + 8] = node
  return out.reshape(M, H, W, C)


# 4x5: one partial group; 2x7 x 5 rows and 3x31: widths that divide nothing, quads straddling
# image rows and images; 9x16: W divides 64 (the corner neighbours are not staged)
@pytest.mark.parametrize("M,H,W", [(1, 4, 5), (5, 2, 7), (1, 3, 31), (2, 9, 16)])
def test_v3_index_scheme_equals_the_dense_graph_attention(M, H, W):
  rng = np.random.default_rng(H * 100 + W)
  h = np.tanh(rng.normal(size=(M, H, W, 256))).astype("f4")
  sm = np.tanh(rng.normal(size=(M, H, W, 64))).astype("f4")
  ref = oracle.gnn_np(h, sm)
  got = run(h, sm)
  assert np.isfinite(got).all()
  assert np.abs(got - ref).max() < 1e-5


def test_v3_index_scheme_with_beam_parent_rows():
  """src_row: output row r reads state row src_row[r] (beam parents), scene rows stay r."""
  rng = np.random.default_rng(5)
  M, H, W = 3, 2, 7
  h = np.tanh(rng.normal(size=(M, H, W, 256))).astype("f4")
  sm = np.tanh(rng.normal(size=(M, H, W, 64))).astype("f4")
  src = np.array([2, 0, 2])
  ref = oracle.gnn_np(h[src], sm)
  got = run(h, sm, src_row=src)
  assert np.abs(got - ref).max() < 1e-5
