"""CPU emulation of the f16x3 arithmetic: an fp32 operand v travels as two
fp16 planes of the pre-scaled value, v0 = half(s v), v1 = half(s v - v0); a product is
a0 w0 + a0 w1 + a1 w0 with fp16 x fp16 products exact in fp32 and fp32 accumulation; the
result is rescaled by a power of two.  Checks the claims the GPU kernels rely on: the
split loses ~2^-22, the dropped a1 w1 term is negligible, the dot product of a gate
convolution (K = 9 * 320) is as accurate as fp32 arithmetic, and a power-of-two operand
scale chosen from max |.| (dgrad / wgrad operands) keeps every plane finite -- the fixed
2^8 does NOT for pixel offsets, which is why the x operand of the wgrad carries its own
exponent (multiverse_amd/csrc/convlstm_wgrad_f16x3.h)."""
import numpy as np


def split(v, scale):
  s = (v.astype(np.float32) * np.float32(scale)).astype(np.float32)
  v0 = s.astype(np.float16)
  v1 = (s - v0.astype(np.float32)).astype(np.float16)
  return v0, v1


def dot_f16x3(a, w, sa=256.0, sw=256.0):
  a0, a1 = split(a, sa)
  w0, w1 = split(w, sw)
  f = lambda x: x.astype(np.float32)
  acc = np.float32(0)
  # three MFMAs per k-step of 16: fp32 accumulate in k order, products exact in fp32
  for k in range(0, a.size, 16):
    sl = slice(k, k + 16)
    for x, y in ((a1, w0), (a0, w1), (a0, w0)):
      acc = np.float32(acc + np.sum(f(x[sl]) * f(y[sl]), dtype=np.float32))
  return np.float32(acc) / np.float32(sa * sw)


def test_split_keeps_22_bits():
  rng = np.random.default_rng(0)
  v = rng.uniform(-1, 1, 100000).astype(np.float32)
  v0, v1 = split(v, 256.0)
  back = (v0.astype(np.float64) + v1.astype(np.float64)) / 256.0
  rel = np.abs(back - v) / np.maximum(np.abs(v), 1e-3)
  assert rel.max() < 2.0 ** -20
  assert np.isfinite(v0.astype(np.float32)).all()


def test_gate_convolution_dot_is_fp32_class():
  rng = np.random.default_rng(1)
  K = 9 * 320
  worst_x3, worst_f32 = 0.0, 0.0
  for _ in range(200):
    a = np.tanh(rng.normal(0, 1, K)).astype(np.float32)          # |h| <= 1
    w = rng.uniform(-0.05, 0.05, K).astype(np.float32)           # glorot-sized kernel
    ref = float(np.dot(a.astype(np.float64), w.astype(np.float64)))
    f32 = np.float32(0)
    for k in range(K):
      f32 = np.float32(f32 + a[k] * w[k])
    norm = float(np.sum(np.abs(a.astype(np.float64) * w)))
    worst_x3 = max(worst_x3, abs(float(dot_f16x3(a, w)) - ref) / norm)
    worst_f32 = max(worst_f32, abs(float(f32) - ref) / norm)
  print("f16x3 %.2e  fp32 %.2e (relative to sum |a w|)" % (worst_x3, worst_f32))
  assert worst_x3 < 2e-7
  assert worst_x3 < 4 * worst_f32 + 1e-8


def test_fixed_scale_overflows_on_pixel_offsets_and_the_dynamic_one_does_not():
  x = np.array([1234.5, -960.25, 17.0, 0.03], dtype=np.float32)   # grid_obs_regress values
  with np.errstate(over="ignore"):
    v0, _ = split(x, 256.0)
  assert not np.isfinite(v0.astype(np.float32)).all()          # 1234.5 * 256 > 65504
  e = 13 - int(np.floor(np.log2(np.abs(x).max())))             # chain_exp_kernel
  v0, v1 = split(x, 2.0 ** e)
  assert np.isfinite(v0.astype(np.float32)).all()
  assert 2.0 ** 13 <= np.abs(v0.astype(np.float32)).max() < 2.0 ** 14
  back = (v0.astype(np.float64) + v1.astype(np.float64)) / 2.0 ** e
  assert np.abs(back - x).max() <= np.abs(x).max() * 2.0 ** -21
