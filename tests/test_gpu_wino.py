# coding=utf-8
"""GPU: the Winograd forms of the f16x3 ConvLSTM step -- F(3,3) over row triples
(csrc/convlstm_wino3.h, variant 3: the default gate kernel of the f16x3 compute mode whenever
the grid widths divide 32) and F(2,3) over row pairs (csrc/convlstm_wino.h, variant 2: its
fall-back, and the dgrad form) -- against the CPU oracle, the direct f16x3 form and their own
emitted operand planes, kernel by kernel through the C ABI (mv_op_convlstm_step16).  The end-to-end parity tests (forward, beam,
training, at-size) run through the same kernel inside the engine.

Tolerance: 2e-5 absolute on O(1) gate sums -- the bar of the fp32-MFMA kernel test
(tests/test_gpu_kernels.py); measured errors are printed."""

import numpy as np
import pytest

from oracle import multiverse_oracle as oracle

pytestmark = pytest.mark.gpu


def _case(M, H, W, Cx, zero, seed):
  rng = np.random.default_rng(seed)
  C = 256
  x = rng.normal(size=(M, H, W, Cx)).astype("f4")
  if Cx > 3:
    x = np.tanh(x)                # f16x3 operand range: embeddings / scene features are in [-1, 1]
  else:
    x = (x * 300.0).astype("f4")  # pixel offsets of the regression encoder
  lim = np.sqrt(6.0 / (9 * (Cx + C) + 9 * 4 * C)) * 3.0
  kernel = rng.uniform(-lim, lim, size=(3, 3, Cx + C, 4 * C)).astype("f4")
  if Cx <= 3:
    kernel[:, :, :Cx, :] *= 1.0 / 300.0
  biases = (0.1 * rng.normal(size=4 * C)).astype("f4")
  if zero:
    c = h = None
    c0 = h0 = np.zeros((M, H, W, C), "f4")
  else:
    c = c0 = rng.normal(size=(M, H, W, C)).astype("f4")
    h = h0 = np.tanh(rng.normal(size=(M, H, W, C))).astype("f4")
  co, ho = oracle.convlstm_step_np(x, c0, h0, kernel, biases)
  return x, c, h, kernel, biases, co, ho


def _where(err, name):
  """Localise a mismatch: max error by image, row parity, row, column, channel mod 16 and
  channel block -- a wrong lane / register / tile map shows up as a pattern here."""
  M, H, W, C = err.shape
  lines = ["%s: max %.3g at %s" % (name, err.max(), np.unravel_index(err.argmax(), err.shape))]
  lines.append("  by image      " + " ".join("%.1e" % err[m].max() for m in range(M)))
  lines.append("  by row        " + " ".join("%.1e" % err[:, y].max() for y in range(H)))
  lines.append("  by column     " + " ".join("%.1e" % err[:, :, xx].max() for xx in range(W)))
  lines.append("  by ch mod 16  " + " ".join("%.1e" % err[..., k::16].max() for k in range(16)))
  lines.append("  by ch block16 " + " ".join("%.1e" % err[..., b * 16:(b + 1) * 16].max()
                                            for b in range(C // 16)))
  return "\n".join(lines)


SHAPES = [
    (2, 18, 32, 64, False),   # class encoder, dense x (training form)
    (2, 18, 32, 2, False),    # regression encoder: fp32 x chunk
    (3, 9, 16, 32, False),    # decoders, scale 1: odd H (last row pair half empty), 80 pairs/image
    (2, 9, 16, 64, True),     # first encoder step: zero state, x only
    (2, 9, 16, 2, True),      # zero state + fp32 x chunk only
    (5, 9, 16, 16, False),    # a wave tile straddles three images' worth of pairs
    (1, 6, 8, 32, False),     # W = 8: four row pairs per wave tile, partial workgroup
    (3, 18, 32, 32, False),   # M tiles not a multiple of the workgroup
    (2, 7, 16, 32, False),    # H = 7: the last row triple holds ONE row (F(3,3) remainder 1)
    (2, 8, 32, 64, False),    # H = 8: the last row triple holds two rows (remainder 2)
    (1, 3, 32, 2, False),     # a single row triple per image, fp32 x chunk
]
FORMS = [(2, "F(2,3)"), (3, "F(3,3)")]


@pytest.mark.parametrize("variant,form", FORMS)
@pytest.mark.parametrize("M,H,W,Cx,zero", SHAPES)
def test_wino_step_vs_oracle(built_lib, M, H, W, Cx, zero, variant, form):
  x, c, h, kernel, biases, co, ho = _case(M, H, W, Cx, zero, M * 1000 + H * 10 + Cx)
  cg, hg, h16 = built_lib.op_convlstm_step16(x, c, h, kernel, biases, variant=variant)
  ec, eh = np.abs(cg - co), np.abs(hg - ho)
  print("winograd %s M=%d %dx%d Cx=%d zero=%s: max|dc| %.3g max|dh| %.3g, planes vs h' %.3g"
        % (form, M, H, W, Cx, zero, ec.max(), eh.max(), np.abs(h16 - hg).max()))
  assert ec.max() < 2e-5, _where(ec, "c'")
  assert eh.max() < 2e-5, _where(eh, "h'")
  # the operand planes the epilogue emitted for the next step ARE h' (two fp16 planes of
  # 256 h': 22 bits)
  ep = np.abs(h16 - hg)
  assert ep.max() < 1e-6, _where(ep, "h' planes")


@pytest.mark.parametrize("M,H,W,Cx,zero", SHAPES[:4])
def test_wino_agrees_with_direct_form(built_lib, M, H, W, Cx, zero):
  """Same operands, same planes: the two forms differ by fp32 summation order and the
  ~2^-21 of the in-kernel input transform."""
  x, c, h, kernel, biases, co, ho = _case(M, H, W, Cx, zero, 77 + M + Cx)
  c1, h1, p1 = built_lib.op_convlstm_step16(x, c, h, kernel, biases, variant=1)
  c2, h2, p2 = built_lib.op_convlstm_step16(x, c, h, kernel, biases, variant=2)
  c3, h3, p3 = built_lib.op_convlstm_step16(x, c, h, kernel, biases, variant=3)
  d1, d2, d3 = np.abs(c1 - co).max(), np.abs(c2 - co).max(), np.abs(c3 - co).max()
  print("max|dc| vs the oracle: direct f16x3 %.3g, F(2,3) %.3g, F(3,3) %.3g; F(2,3) - direct %.3g, "
        "F(3,3) - direct %.3g" % (d1, d2, d3, np.abs(c1 - c2).max(), np.abs(c1 - c3).max()))
  assert d1 < 2e-5 and d2 < 2e-5 and d3 < 2e-5
  assert np.abs(c1 - c2).max() < 1e-5 and np.abs(h1 - h2).max() < 1e-5
  assert np.abs(p1 - p2).max() < 1e-5
  assert np.abs(c1 - c3).max() < 1.5e-5 and np.abs(h1 - h3).max() < 1.5e-5
  assert np.abs(p1 - p3).max() < 1.5e-5


@pytest.mark.parametrize("variant,form", FORMS)
def test_wino_transpose_detecting(built_lib, variant, form):
  """One hot input cell / one hot weight tap per case: tap orientation, the row class inside
  the tile (even / odd rows of a pair; rows 3t, 3t+1, 3t+2 of a triple; the image's first and
  last rows) and row <-> column swaps that symmetric data would hide.  Every (ky, kx) tap."""
  M, H, W, Cx, C = 1, 6, 8, 32, 256
  for ky in range(3):
    for kx in range(3):
      for (sy, sx) in ((2, 5), (3, 2), (4, 6), (0, 0), (5, 7)):
        x = np.zeros((M, H, W, Cx), "f4")
        x[0, sy, sx, 3] = 1.0
        kernel = np.zeros((3, 3, Cx + C, 4 * C), "f4")
        kernel[ky, kx, 3, 1 * C + 17] = 2.0      # gate j, channel 17
        biases = np.zeros(4 * C, "f4")
        biases[0 * C:1 * C] = 5.0                # input gate ~ open
        c = np.zeros((M, H, W, C), "f4")
        h = np.zeros((M, H, W, C), "f4")
        co, ho = oracle.convlstm_step_np(x, c, h, kernel, biases)
        cg, hg, _ = built_lib.op_convlstm_step16(x, c, h, kernel, biases, variant=variant)
        oy, ox = sy - (ky - 1), sx - (kx - 1)      # out(y,x) sees in(y+ky-1, x+kx-1)
        if 0 <= oy < H and 0 <= ox < W:
          assert abs(co[0, oy, ox, 17]) > 0.5
        err = np.abs(cg - co)
        assert err.max() < 1e-6, "tap (%d,%d) source (%d,%d)\n%s" % (ky, kx, sy, sx,
                                                                    _where(err, "c'"))
        assert np.abs(hg - ho).max() < 1e-6


def test_wino_dynamic_range(built_lib):
  """Operands at the edges of the scaled fp16 range: |h| up to 1 next to |h| ~ 1e-3 (whose low
  plane is subnormal), weights of mixed magnitude -- the TwoSum of the input transform and the
  fp64 weight transform must hold the f16x3 error class."""
  M, H, W, Cx, C = 2, 8, 16, 32, 256
  rng = np.random.default_rng(5)
  x = np.tanh(rng.normal(size=(M, H, W, Cx)) * 3.0).astype("f4")
  h = np.tanh(rng.normal(size=(M, H, W, C)) * 3.0).astype("f4")
  h[:, ::2] *= 1e-3                                  # alternate rows tiny: cancellation in V
  cst = rng.normal(size=(M, H, W, C)).astype("f4")
  kernel = (rng.normal(size=(3, 3, Cx + C, 4 * C)) * 0.02).astype("f4")
  kernel[..., ::7] *= 8.0
  biases = np.zeros(4 * C, "f4")
  co, ho = oracle.convlstm_step_np(x, cst, h, kernel, biases)
  c1, h1, _ = built_lib.op_convlstm_step16(x, cst, h, kernel, biases, variant=1)
  c2, h2, _ = built_lib.op_convlstm_step16(x, cst, h, kernel, biases, variant=2)
  c3, h3, _ = built_lib.op_convlstm_step16(x, cst, h, kernel, biases, variant=3)
  print("dynamic range: direct %.3g, F(2,3) %.3g, F(3,3) %.3g" % (
      np.abs(c1 - co).max(), np.abs(c2 - co).max(), np.abs(c3 - co).max()))
  assert np.abs(c2 - co).max() < 2e-5 and np.abs(h2 - ho).max() < 2e-5
  assert np.abs(c3 - co).max() < 3e-5 and np.abs(h3 - ho).max() < 3e-5


HALO_SHAPES = [
    (2, 36, 18, 32, False),   # BASELINE.json's literal 36 x 18 grid
    (3, 18, 9, 64, False),    # ... and 18 x 9: three image rows and a bit per wave tile
    (2, 9, 24, 32, False),    # W = 24
    (1, 12, 12, 2, False),    # fp32 x chunk on a halo tiling
    (2, 6, 18, 16, True),     # zero state
    (2, 7, 33, 32, False),    # W > 32, H not a multiple of 3
]


@pytest.mark.parametrize("M,H,W,Cx,zero", HALO_SHAPES)
def test_wino3_widths_that_do_not_divide_32(built_lib, M, H, W, Cx, zero):
  """F(3,3) on grids whose width does not divide 32 (csrc/convlstm_wino3.h HALO tiling: a wave's
  tile starts one triple-cell early, owns its inner 30 lanes): against the oracle, the direct
  f16x3 form and its own operand planes."""
  x, c, h, kernel, biases, co, ho = _case(M, H, W, Cx, zero, M * 1000 + H * 10 + W + Cx)
  cg, hg, h16 = built_lib.op_convlstm_step16(x, c, h, kernel, biases, variant=3)
  c1, h1, _ = built_lib.op_convlstm_step16(x, c, h, kernel, biases, variant=1)
  ec, eh = np.abs(cg - co), np.abs(hg - ho)
  print("winograd F(3,3) halo M=%d %dx%d Cx=%d zero=%s: max|dc| %.3g max|dh| %.3g (direct %.3g), "
        "planes vs h' %.3g" % (M, H, W, Cx, zero, ec.max(), eh.max(), np.abs(c1 - co).max(),
                               np.abs(h16 - hg).max()))
  assert ec.max() < 2e-5, _where(ec, "c'")
  assert eh.max() < 2e-5, _where(eh, "h'")
  assert np.abs(h16 - hg).max() < 1e-6, _where(np.abs(h16 - hg), "h' planes")


def test_wino3_halo_transpose_detecting(built_lib):
  """One hot taps on a W = 18 grid: sources on both sides of a wave tile's seams (lanes 0 / 1 and
  30 / 31 of the halo tiling) and of the image rows' ends."""
  M, H, W, Cx, C = 1, 6, 18, 32, 256
  for ky in range(3):
    for kx in range(3):
      for (sy, sx) in ((2, 11), (2, 12), (3, 5), (0, 0), (5, 17), (1, 17), (2, 0)):
        x = np.zeros((M, H, W, Cx), "f4")
        x[0, sy, sx, 3] = 1.0
        kernel = np.zeros((3, 3, Cx + C, 4 * C), "f4")
        kernel[ky, kx, 3, 1 * C + 17] = 2.0
        biases = np.zeros(4 * C, "f4")
        biases[0 * C:1 * C] = 5.0
        c = np.zeros((M, H, W, C), "f4")
        h = np.zeros((M, H, W, C), "f4")
        co, ho = oracle.convlstm_step_np(x, c, h, kernel, biases)
        cg, hg, _ = built_lib.op_convlstm_step16(x, c, h, kernel, biases, variant=3)
        err = np.abs(cg - co)
        assert err.max() < 1e-6, "tap (%d,%d) source (%d,%d)\n%s" % (ky, kx, sy, sx,
                                                                    _where(err, "c'"))
        assert np.abs(hg - ho).max() < 1e-6


def _bf16_round(a):
  """round-to-nearest-even to bf16, as the planes hold the operands in compute mode 2"""
  u = np.ascontiguousarray(a, dtype=np.float32).view(np.uint32).astype(np.uint64)
  u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
  return u.astype(np.uint32).view(np.float32).reshape(np.shape(a))


@pytest.mark.parametrize("M,H,W,Cx,zero", SHAPES + HALO_SHAPES)
def test_bf16_row_triple_tile_vs_oracle_on_bf16_operands(built_lib, M, H, W, Cx, zero):
  """Compute mode 2's gate kernel on the row-triple tile (csrc/convlstm_wino3.h BF16D, op variant
  4): ONE bf16 plane per operand, fp32 accumulate.  Held to the oracle evaluated ON THE ROUNDED
  OPERANDS (h and the 16-channel x groups and the kernel rows they meet rounded to bf16; the
  2-channel fp32 chunk and its kernel rows exact), so that what is left is summation order:
  the same 2e-5 as the f16x3 forms.  Every tiling case of the F(3,3) kernel incl. the halo."""
  x, c, h, kernel, biases, _, _ = _case(M, H, W, Cx, zero, M * 1000 + H * 10 + Cx + 5)
  kr = kernel.copy()
  if Cx > 3:
    kr = _bf16_round(kernel)
    xr = _bf16_round(x)
  else:
    kr[:, :, Cx:, :] = _bf16_round(kernel[:, :, Cx:, :])
    xr = x
  C = kernel.shape[3] // 4
  c0 = np.zeros((M, H, W, C), "f4") if zero else c
  h0 = np.zeros((M, H, W, C), "f4") if zero else _bf16_round(h)
  co, ho = oracle.convlstm_step_np(xr, c0, h0, kr, biases)
  cg, hg, h16 = built_lib.op_convlstm_step16(x, c, h, kernel, biases, variant=4)
  ec, eh = np.abs(cg - co), np.abs(hg - ho)
  print("bf16 row-triple tile M=%d %dx%d Cx=%d zero=%s: max|dc| %.3g max|dh| %.3g, plane vs h' %.3g"
        % (M, H, W, Cx, zero, ec.max(), eh.max(), np.abs(h16 - _bf16_round(hg)).max()))
  assert ec.max() < 2e-5, _where(ec, "c'")
  assert eh.max() < 2e-5, _where(eh, "h'")
  # the emitted plane IS h' rounded to bf16
  assert np.abs(h16 - _bf16_round(hg)).max() == 0, _where(np.abs(h16 - _bf16_round(hg)), "plane")
