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
#include <stdlib.h>

#include "convlstm_mfma.h"

namespace mv {

constexpr int kWgTile = 64;    // ci x n tile of a wave
#ifndef MV_WG_UNROLL
#define MV_WG_UNROLL 2   // measured on MI355X: 2 -> 86.9 TF, 4 -> 84.2, 8 -> 71.7
#endif
constexpr int kWgUnroll = MV_WG_UNROLL;   // k-steps (2 cells each) per register set

struct WgradArgs {
  const float* x;     // [R, HW, Cx] contiguous (may be NULL when Cx == 0)
  const float* h;     // [R, HW, C]
  const float* g;     // [R, HW, 4C]
  float* partial;     // [nsplit][9][Cx + C][4C]
  int32_t R, H, W, Cx, C;
  int32_t cells_per_split;   // generic path: cells (even); fast path: IMAGES per split
  int32_t nsplit;
  int32_t n_xblocks;         // ceil(Cx / 64)
  int32_t n_ciblocks;        // n_xblocks + C / 64
  int32_t only_x;            // fast path: compute the x rows only (the h rows come from
                             // the fp16-pipe kernel, convlstm_wgrad_f16x3.h)
};

struct WFrag {
  float a0[kWgUnroll], a1[kWgUnroll], b0[kWgUnroll], b1[kWgUnroll];
};

// Addressing is SCALAR: every index that does not depend on the lane (cell pair,
// its (y, x), tap validity, base pointers) lives in SGPRs and is advanced by the
// scalar unit; a lane adds only a constant offset (cell parity * row stride +
// channel).  The first version computed (y, x), masks and 64-bit addresses per
// lane: ~50 VALU instructions per 4 MFMAs kept the matrix pipe 45 % busy
// (profiles/r1_train_pmc_*.json).
// Operand buffers need kWgradPad floats of slack before and after (a masked
// lane may read one cell outside the tensor; engine-owned buffers have it).
constexpr int kWgradPad = 2048;

__global__ __launch_bounds__(256, 2)
void convlstm_wgrad_kernel(const WgradArgs a) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, half = lane >> 5;
  const int H = a.H, W = a.W, HW = H * W, C = a.C, Cx = a.Cx, N4 = 4 * C;
  // blockIdx -> (split, tap, ci block, n quad).  Workgroups round-robin over the
  // 8 XCDs by linear id, so id % 8 picks the XCD: every split (= cell range) is
  // pinned to ONE XCD and all its 9 x ci x n tiles run there back to back, so
  // the split's A and G rows are shared through that XCD's L2.
  const int nquads = N4 / (4 * kWgTile);   // 4C = 1024 -> 16 n blocks = 4 quads
  const int tiles_per_split = 9 * a.n_ciblocks * nquads;
  const int xcd = blockIdx.x & 7;
  int b = blockIdx.x >> 3;
  const int split = xcd + 8 * (b / tiles_per_split);
  b = b % tiles_per_split;
  const int nq = b % nquads; b /= nquads;
  const int cib = b % a.n_ciblocks;
  const int tap = b / a.n_ciblocks;
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

  // ---- scalar state of the NEXT cell pair to load: cells mu, mu + 1
  long long mu = m_begin;
  int y0, x0;
  {
    const long long cell = mu % HW;
    y0 = (int)(cell / W);
    x0 = (int)(cell - (long long)y0 * W);
  }
  const float* abase = src + (mu + doff) * (long long)cs + ci0;   // cell mu + tap shift
  const float* gbase = a.g + mu * (long long)N4 + n0;
  const float* asafe = src + ci0;
  const float* gsafe = a.g + n0;
  // ---- per-lane constant offsets
  const int a_lane = half * cs + li;
  const int g_lane = half * N4 + li;

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc00[i] = 0.f; acc01[i] = 0.f; acc10[i] = 0.f; acc11[i] = 0.f; }

  auto load_set = [&](WFrag& f) {
#pragma unroll
    for (int u = 0; u < kWgUnroll; ++u) {
      // second cell of the pair (may wrap to the next row / image)
      int x1 = x0 + 1, y1 = y0;
      if (x1 >= W) { x1 = 0; y1 = y0 + 1; if (y1 >= H) y1 = 0; }
      const bool live0 = mu < m_end, live1 = mu + 1 < m_end;
      const bool ok0 = live0 & (y0 + dy >= 0) & (y0 + dy < H) & (x0 + dx >= 0) & (x0 + dx < W);
      const bool ok1 = live1 & (y1 + dy >= 0) & (y1 + dy < H) & (x1 + dx >= 0) & (x1 + dx < W);
      // scalar clamps: a pair with no valid tap / no live cell reads the tensor start
      const float* ap = (ok0 | ok1) ? abase : asafe;
      const float* gp = live0 ? gbase : gsafe;
      const bool ok = half ? ok1 : ok0;
      const bool live = half ? live1 : live0;
      const float va0 = ap[a_lane];
      float va1 = 0.f;
      if (a1_on) va1 = ap[a_lane + 32];
      const float vb0 = gp[g_lane];
      const float vb1 = gp[g_lane + 32];
      f.a0[u] = (ok & a0_lane) ? va0 : 0.f;
      f.a1[u] = (ok & a1_lane) ? va1 : 0.f;
      f.b0[u] = live ? vb0 : 0.f;
      f.b1[u] = live ? vb1 : 0.f;
      // advance two cells (scalar)
      mu += 2;
      abase += 2 * cs;
      gbase += 2 * N4;
      x0 += 2;
      if (x0 >= W) { x0 -= W; y0 += 1; if (y0 >= H) y0 = 0; }
      if (x0 >= W) { x0 -= W; y0 += 1; if (y0 >= H) y0 = 0; }   // W == 1
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

// ---------------------------------------------------------------- fast path
// W % (2*kWgUnroll) == 0 (18x32 and 9x16 grids): the split is a range of whole IMAGES
// and the loop walks (image, row, chunk of 2*kWgUnroll cells).  A whole register set
// then lies inside one image row, so
//   * rows whose tap row y + dy is outside the image are SKIPPED (no loads, no
//     MFMAs: 1/H of the work of the six dy != 0 taps),
//   * the only padded cells left are x = 0 for dx = -1 (first cell of chunk 0)
//     and x = W-1 for dx = +1 (last cell of the last chunk): two scalar flags
//     per set instead of per-cell (y, x) bookkeeping,
//   * G needs no masking at all.
// Per set the scalar unit does ~20 instructions (the per-cell version needed
// ~240, and the scalar ALU is shared by the four SIMDs of a CU: it was the
// bottleneck at 45 % MFMA busy).
template <bool A1, bool PARTIAL>
__device__ __forceinline__ void wgrad_fast_body(const WgradArgs& a, int split, int tap,
                                                int cib, int n0) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, half = lane >> 5;
  const int H = a.H, W = a.W, HW = H * W, C = a.C, Cx = a.Cx, N4 = 4 * C;
  const bool is_x = cib < a.n_xblocks;
  const float* src = is_x ? a.x : a.h;
  const int cs = is_x ? Cx : C;
  const int ci0 = (is_x ? cib : cib - a.n_xblocks) * kWgTile;
  const int cvalid = min(cs - ci0, kWgTile);
  const bool a0_lane = li < cvalid, a1_lane = (li + 32) < cvalid;
  const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
  const int doff = dy * W + dx;
  const int cpr = W / (2 * kWgUnroll);                     // chunks per row
  const int ylo = dy < 0 ? 1 : 0, yhi = dy > 0 ? H - 1 : H;   // rows with a valid tap row
  const int rows = yhi - ylo;
  const int img0 = split * a.cells_per_split;              // images per split (fast path)
  int img1 = img0 + a.cells_per_split;
  if (img1 > a.R) img1 = a.R;
  const int nimg = img1 > img0 ? img1 - img0 : 0;
  const int nsets = nimg * rows * cpr;

  // WIDE (full 64-channel h blocks): 8-byte loads, lane li holds channels /
  // columns 2*li, 2*li+1, i.e. accumulator (a, b) row i <-> ci0 + 2i + a and
  // column j <-> n0 + 2j + b; otherwise 4-byte loads with (i + 32a, j + 32b).
  constexpr bool WIDE = A1 && !PARTIAL;
  const unsigned a_lane = (unsigned)(half * cs + (WIDE ? 2 * li : li));
  const unsigned g_lane = (unsigned)(half * N4 + (WIDE ? 2 * li : li));
  const float* abase0 = src + (long long)doff * cs + ci0;
  const float* gbase0 = a.g + n0;
  const bool lo_half = half == 0;

  f32x16 acc00, acc01, acc10, acc11;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc00[i] = 0.f; acc01[i] = 0.f; acc10[i] = 0.f; acc11[i] = 0.f; }

  // scalar iterator over sets: (image, row, chunk)
  int it_img = img0, it_y = ylo, it_ch = 0;
  auto load_set = [&](WFrag& f) {
    const long long m = ((long long)it_img * H + it_y) * W + it_ch * (2 * kWgUnroll);   // first cell
    const float* ap = abase0 + m * cs;
    const float* gp = gbase0 + m * N4;
    const bool kill_first = (dx < 0) & (it_ch == 0);          // cell x = 0, tap x = -1
    const bool kill_last = (dx > 0) & (it_ch == cpr - 1);     // cell x = W-1, tap x = W
#pragma unroll
    for (int u = 0; u < kWgUnroll; ++u) {
      float va0, va1 = 0.f;
      if (WIDE) {
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        const f32x2 av = *reinterpret_cast<const f32x2*>(ap + a_lane + (unsigned)(2 * u) * cs);
        const f32x2 gv = *reinterpret_cast<const f32x2*>(gp + g_lane + (unsigned)(2 * u) * N4);
        va0 = av[0]; va1 = av[1];
        f.b0[u] = gv[0]; f.b1[u] = gv[1];
      } else {
        va0 = ap[a_lane + (unsigned)(2 * u) * cs];
        if (A1) va1 = ap[a_lane + (unsigned)(2 * u) * cs + 32];
        f.b0[u] = gp[g_lane + (unsigned)(2 * u) * N4];
        f.b1[u] = gp[g_lane + (unsigned)(2 * u) * N4 + 32];
      }
      if (u == 0) {
        const bool k = kill_first & lo_half;
        va0 = k ? 0.f : va0; va1 = k ? 0.f : va1;
      }
      if (u == kWgUnroll - 1) {
        const bool k = kill_last & !lo_half;
        va0 = k ? 0.f : va0; va1 = k ? 0.f : va1;
      }
      if (PARTIAL) { va0 = a0_lane ? va0 : 0.f; va1 = a1_lane ? va1 : 0.f; }
      f.a0[u] = va0; f.a1[u] = va1;
    }
    // advance, saturating at the last set (the pipeline's extra loads re-read it)
    int ch = it_ch + 1, y = it_y, im = it_img;
    if (ch == cpr) { ch = 0; y += 1; }
    if (y == yhi) { y = ylo; im += 1; }
    if (im < img1) { it_ch = ch; it_y = y; it_img = im; }
  };
  auto mma_set = [&](const WFrag& f) {
#pragma unroll
    for (int u = 0; u < kWgUnroll; ++u) {
      acc00 = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[u], f.b0[u], acc00, 0, 0, 0);
      acc01 = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a0[u], f.b1[u], acc01, 0, 0, 0);
      if (A1) {
        acc10 = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[u], f.b0[u], acc10, 0, 0, 0);
        acc11 = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a1[u], f.b1[u], acc11, 0, 0, 0);
      }
    }
  };
  WFrag f0, f1;
  if (nsets > 0) load_set(f0);
  for (int s = 0; s < nsets; s += 2) {
    load_set(f1);
    mma_set(f0);
    load_set(f0);
    if (s + 1 < nsets) mma_set(f1);
  }

  const int Cin = Cx + C;
  float* pt = a.partial + ((size_t)split * 9 + tap) * (size_t)Cin * N4;
  const int cbase = (is_x ? 0 : Cx) + ci0;
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int i = (reg & 3) + 8 * (reg >> 2) + 4 * half;
    if (WIDE) {
      typedef float f32x2 __attribute__((ext_vector_type(2)));
      float* r0 = pt + (size_t)(cbase + 2 * i) * N4 + n0 + 2 * li;
      *reinterpret_cast<f32x2*>(r0) = f32x2{acc00[reg], acc01[reg]};
      *reinterpret_cast<f32x2*>(r0 + N4) = f32x2{acc10[reg], acc11[reg]};
    } else {
      if (i < cvalid) {
        float* row = pt + (size_t)(cbase + i) * N4 + n0 + li;
        row[0] = acc00[reg];
        row[32] = acc01[reg];
      }
      if (A1 && i + 32 < cvalid) {
        float* row = pt + (size_t)(cbase + i + 32) * N4 + n0 + li;
        row[0] = acc10[reg];
        row[32] = acc11[reg];
      }
    }
  }
}

__global__ __launch_bounds__(256, 2)
void convlstm_wgrad_fast_kernel(const WgradArgs a) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int N4 = 4 * a.C;
  const int nquads = N4 / (4 * kWgTile);
  // Block order inside one XCD (xcd = id % 8, measured 1:1 on MI355X with
  // tools/xcc_probe.py): all FULL (h) tiles of its splits, split by split, then
  // all partial (x) tiles.  The workgroups resident on an XCD at one time then
  // run at the same speed over the same cell range, so they stay in phase and
  // share the range's A / G rows through the 4 MiB L2; a partial tile does half
  // the MFMAs per cell, runs ahead and would pull its rows through L2 a second
  // time (measured: 9.4 GB fetched per launch for 0.7 GB of operands when the
  // two kinds were interleaved).
  const int xcd = blockIdx.x & 7;
  int j = blockIdx.x >> 3;
  const int splits_per_xcd = a.nsplit >> 3;
  const int n_hblocks = a.only_x ? 0 : a.n_ciblocks - a.n_xblocks;
  const int n_h_tiles = 9 * n_hblocks * nquads;
  const int n_x_tiles = 9 * a.n_xblocks * nquads;
  int sl, cib, tap, nq;
  if (j < splits_per_xcd * n_h_tiles) {
    sl = j / n_h_tiles;
    int t = j - sl * n_h_tiles;
    nq = t % nquads; t /= nquads;
    cib = a.n_xblocks + t % n_hblocks;
    tap = t / n_hblocks;
  } else {
    j -= splits_per_xcd * n_h_tiles;
    sl = j / n_x_tiles;
    int t = j - sl * n_x_tiles;
    nq = t % nquads; t /= nquads;
    cib = t % a.n_xblocks;
    tap = t / a.n_xblocks;
  }
  const int split = xcd + 8 * sl;
  const int n0 = (nq * 4 + wave) * kWgTile;
  if (n0 >= N4) return;
  const bool is_x = cib < a.n_xblocks;
  const int cs = is_x ? a.Cx : a.C;
  const int ci0 = (is_x ? cib : cib - a.n_xblocks) * kWgTile;
  const int cvalid = min(cs - ci0, kWgTile);
  if (cvalid == kWgTile) wgrad_fast_body<true, false>(a, split, tap, cib, n0);
  else if (cvalid > 32) wgrad_fast_body<true, true>(a, split, tap, cib, n0);
  else wgrad_fast_body<false, true>(a, split, tap, cib, n0);
}

static inline bool wgrad_fast_ok(const WgradArgs& a) { return (a.W % (2 * kWgUnroll)) == 0; }

static inline void wgrad_plan(WgradArgs& a, int target_blocks) {
  a.n_xblocks = (a.Cx + kWgTile - 1) / kWgTile;
  a.n_ciblocks = a.n_xblocks + a.C / kWgTile;
  if (wgrad_fast_ok(a)) {      // splits are ranges of whole images
    const int per_split = 9 * a.n_ciblocks * ((4 * a.C) / (4 * kWgTile));
    int nsplit = target_blocks / per_split;
    if (const char* ev = getenv("MV_WGRAD_SPLITS")) nsplit = atoi(ev);   // tuning knob
    if (nsplit > 256) nsplit = 256;
    nsplit = (nsplit + 7) & ~7;
    if (nsplit < 8) nsplit = 8;
    a.cells_per_split = (a.R + nsplit - 1) / nsplit;     // IMAGES per split
    if (a.cells_per_split < 1) a.cells_per_split = 1;
    a.nsplit = nsplit;
    return;
  }
  const long long m_total = (long long)a.R * a.H * a.W;
  const int per_split_blocks = 9 * a.n_ciblocks * ((4 * a.C) / (4 * kWgTile));
  int nsplit = target_blocks / per_split_blocks;
  if (nsplit > 64) nsplit = 64;
  nsplit = (nsplit + 7) & ~7;           // one split set per XCD
  if (nsplit < 8) nsplit = 8;
  long long cps = (m_total + nsplit - 1) / nsplit;
  cps = (cps + 1) & ~1LL;
  if (cps < 2) cps = 2;
  a.cells_per_split = (int32_t)cps;
  a.nsplit = nsplit;                    // trailing splits may be empty (zero tiles)
}

static inline unsigned wgrad_blocks(const WgradArgs& a) {
  const unsigned cib = a.only_x ? (unsigned)a.n_xblocks : (unsigned)a.n_ciblocks;
  return (unsigned)a.nsplit * 9u * cib * (unsigned)((4 * a.C) / (4 * kWgTile));
}

static inline size_t wgrad_partial_elems(const WgradArgs& a) {
  return (size_t)a.nsplit * 9 * (size_t)(a.Cx + a.C) * 4 * a.C;
}

}  // namespace mv

namespace mv {
static inline void launch_convlstm_wgrad(const WgradArgs& a, hipStream_t stream) {
  if (wgrad_fast_ok(a))
    hipLaunchKernelGGL(convlstm_wgrad_fast_kernel, dim3(wgrad_blocks(a)), dim3(256), 0,
                       stream, a);
  else
    hipLaunchKernelGGL(convlstm_wgrad_kernel, dim3(wgrad_blocks(a)), dim3(256), 0, stream, a);
}
}  // namespace mv
