// wgrad of the ConvLSTM gate convolution on v_mfma_f32_32x32x2_f32:
//
//   dW[tap][ci][n] = sum_m  xh[m + d_tap][ci] * G[m][n]
//
// over ALL time steps of one cell at once (the training forward keeps x, the h
// operand and the gate gradients G of every step, time-major, so the T steps
// are simply T*N more images).  tf.gradients of tf.nn.conv2d w.r.t. its filter
// inside tf.contrib.rnn.ConvLSTMCell (reference code/pred_models.py:189-249,
// Trainer :1694-1698).
//
// GEMM view: rows i = input channel (for one tap), columns j = gate column,
// reduction k = cell.  One wave owns a 64 (ci) x 64 (n) tile of ONE tap:
// 4 accumulators, A = two dwords of the (tap-shifted, zero-padded) activation
// cell, B = two dwords of that cell's G row; an MFMA consumes two cells
// (k = lane>>5).  The cell range is split across workgroups (deterministic
// split-K: partial tiles are written per split and summed by a second pass,
// no atomics, so gradients are bitwise reproducible).  The four waves of a
// workgroup share (split, tap, ci block) and take four adjacent n blocks, so
// the A operand is fetched once per workgroup through L1.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "convlstm_mfma.h"

namespace mv {

constexpr int kWgTile = 64;    // ci x n tile of a wave
constexpr int kWgUnroll = 4;   // k-steps (2 cells each) per register set

struct WgradArgs {
  const float* x;     // [R, HW, Cx] contiguous (may be NULL when Cx == 0)
  const float* h;     // [R, HW, C]
  const float* g;     // [R, HW, 4C]
  float* partial;     // [nsplit][9][Cx + C][4C]
  int32_t R, H, W, Cx, C;
  int32_t cells_per_split;   // even
  int32_t nsplit;
  int32_t n_xblocks;         // ceil(Cx / 64)
  int32_t n_ciblocks;        // n_xblocks + C / 64
};

struct WFrag {
  float a0[kWgUnroll], a1[kWgUnroll], b0[kWgUnroll], b1[kWgUnroll];
};

__global__ __launch_bounds__(256, 2)
void convlstm_wgrad_kernel(const WgradArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int li = lane & 31, half = lane >> 5;
  const int H = a.H, W = a.W, HW = H * W, C = a.C, Cx = a.Cx, N4 = 4 * C;
  // blockIdx -> (split, tap, ci block, n quad); split slowest so that the
  // workgroups resident at one time walk the same cells (L2 / MALL reuse of G)
  int b = blockIdx.x;
  const int nquads = N4 / (4 * kWgTile);   // 4C = 1024 -> 16 n blocks = 4 quads
  const int nq = b % nquads; b /= nquads;
  const int cib = b % a.n_ciblocks; b /= a.n_ciblocks;
  const int tap = b % 9;
  const int split = b / 9;
  const int n0 = (nq * 4 + wave) * kWgTile;
  if (n0 >= N4) return;

  const bool is_x = cib < a.n_xblocks;
  const float* src = is_x ? a.x : a.h;
  const int cs = is_x ? Cx : C;
  const int ci0 = (is_x ? cib : cib - a.n_xblocks) * kWgTile;
  const int cvalid = min(cs - ci0, kWgTile);     // channels of this block
  const bool a0_lane = li < cvalid;
  const bool a1_on = cvalid > 32;                // wave-uniform
  const bool a1_lane = (li + 32) < cvalid;
  const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
  const int doff = dy * W + dx;

  const long long m_total = (long long)a.R * HW;
  const long long m_begin = (long long)split * a.cells_per_split;
  long long m_end = m_begin + a.cells_per_split;
  if (m_end > m_total) m_end = m_total;
  // lane state: cell m (this half's cell of the next k-step to LOAD), its (y, x)
  long long m = m_begin + half;
  int yy, xx;
  {
    const long long cell = m % HW;
    yy = (int)(cell / W);
    xx = (int)(cell - (long long)yy * W);
  }
  // running element offsets of this lane's A / B dwords (advance 2 cells per k-step)
  size_t aoff = (size_t)(m + doff) * cs + ci0 + li;   // only dereferenced when ok
  size_t goff = (size_t)m * N4 + n0 + li;

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc00[i] = 0.f; acc01[i] = 0.f; acc10[i] = 0.f; acc11[i] = 0.f; }

  auto load_set = [&](WFrag& f) {
#pragma unroll
    for (int u = 0; u < kWgUnroll; ++u) {
      const bool live = m < m_end;
      const int ty = yy + dy, tx = xx + dx;
      const bool ok = live & (ty >= 0) & (ty < H) & (tx >= 0) & (tx < W);
      // clamped addresses: a dead / padded lane reads element 0 and is zeroed
      const float va0 = src[(ok & a0_lane) ? aoff : 0];
      float va1 = 0.f;
      if (a1_on) va1 = src[(ok & a1_lane) ? aoff + 32 : 0];
      const float vb0 = a.g[live ? goff : 0];
      const float vb1 = a.g[live ? goff + 32 : 0];
      f.a0[u] = (ok & a0_lane) ? va0 : 0.f;
      f.a1[u] = (ok & a1_lane) ? va1 : 0.f;
      f.b0[u] = live ? vb0 : 0.f;
      f.b1[u] = live ? vb1 : 0.f;
      // advance two cells
      m += 2;
      aoff += (size_t)2 * cs;
      goff += (size_t)2 * N4;
      xx += 2;
      if (xx >= W) { xx -= W; yy += 1; }
      if (xx >= W) { xx -= W; yy += 1; }   // W == 1
      if (yy >= H) yy -= H;
      if (yy >= H) yy -= H;
    }
  };
  auto mma_set = [&](const WFrag& f) {
#pragma unroll
    for (int u = 0; u < kWgUnroll; ++u) {
      acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[u], f.b0[u], acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[u], f.b1[u], acc01, 0, 0, 0);
      if (a1_on) {
        acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[u], f.b0[u], acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[u], f.b1[u], acc11, 0, 0, 0);
      }
    }
  };

  const long long span = m_end > m_begin ? m_end - m_begin : 0;
  const int per_set = 2 * kWgUnroll;                       // cells per register set
  const int nsets = (int)((span + per_set - 1) / per_set);
  WFrag f0, f1;
  if (nsets > 0) load_set(f0);
  for (int s = 0; s < nsets; s += 2) {
    load_set(f1);           // past-the-end sets load nothing live (all masked)
    mma_set(f0);
    load_set(f0);
    if (s + 1 < nsets) mma_set(f1);
  }

  // ---- store the partial tile: D col j = lane&31 (n), row i = (reg&3)+8*(reg>>2)+4*(lane>>5) (ci)
  const int Cin = Cx + C;
  float* pt = a.partial + ((size_t)split * 9 + tap) * (size_t)Cin * N4;
  const int cbase = (is_x ? 0 : Cx) + ci0;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int i = (reg & 3) + 8 * (reg >> 2) + 4 * half;
    if (i < cvalid) {
      float* row = pt + (size_t)(cbase + i) * N4 + n0 + li;
      row[0] = acc00[reg];
      row[32] = acc01[reg];
    }
    if (a1_on && i + 32 < cvalid) {
      float* row = pt + (size_t)(cbase + i + 32) * N4 + n0 + li;
      row[0] = acc10[reg];
      row[32] = acc11[reg];
    }
  }
}

static inline void wgrad_plan(WgradArgs& a, int target_blocks) {
  a.n_xblocks = (a.Cx + kWgTile - 1) / kWgTile;
  a.n_ciblocks = a.n_xblocks + a.C / kWgTile;
  const long long m_total = (long long)a.R * a.H * a.W;
  const int per_split_blocks = 9 * a.n_ciblocks * ((4 * a.C) / (4 * kWgTile));
  int nsplit = target_blocks / per_split_blocks;
  if (nsplit < 1) nsplit = 1;
  // at least 64 cells per split, at most 64 splits
  long long maxsplit = m_total / 64;
  if (maxsplit < 1) maxsplit = 1;
  if (nsplit > maxsplit) nsplit = (int)maxsplit;
  if (nsplit > 64) nsplit = 64;
  long long cps = (m_total + nsplit - 1) / nsplit;
  cps = (cps + 1) & ~1LL;
  a.cells_per_split = (int32_t)cps;
  a.nsplit = (int32_t)((m_total + cps - 1) / cps);
}

static inline unsigned wgrad_blocks(const WgradArgs& a) {
  return (unsigned)a.nsplit * 9u * (unsigned)a.n_ciblocks * (unsigned)((4 * a.C) / (4 * kWgTile));
}

static inline size_t wgrad_partial_elems(const WgradArgs& a) {
  return (size_t)a.nsplit * 9 * (size_t)(a.Cx + a.C) * 4 * a.C;
}

}  // namespace mv
