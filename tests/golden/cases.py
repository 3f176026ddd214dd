# coding=utf-8
"""Seeded inputs of the kernel-level golden cases (shared by make_golden.py and
the tests, so only the OUTPUTS need to be stored in golden_kernels.npz)."""
import numpy as np

CELL_CASES = ((64, (5, 7)), (2, (4, 6)), (32, (9, 16)))


def cell_case(Cx, H, W, C=256):
  rng = np.random.default_rng(1000 + Cx)
  x = rng.normal(size=(1, H, W, Cx)).astype("f4")
  c = rng.normal(size=(1, H, W, C)).astype("f4")
  h = np.tanh(rng.normal(size=(1, H, W, C))).astype("f4")
  lim = np.sqrt(6.0 / (9 * (Cx + C) + 9 * 4 * C)) * 3.0
  k = rng.uniform(-lim, lim, size=(3, 3, Cx + C, 4 * C)).astype("f4")
  b = (0.1 * rng.normal(size=4 * C)).astype("f4")
  return x, c, h, k, b


def gnn_case():
  rng = np.random.default_rng(2000)
  h = np.tanh(rng.normal(size=(2, 4, 5, 256))).astype("f4")
  sm = np.tanh(rng.normal(size=(2, 4, 5, 64))).astype("f4")
  return h, sm
