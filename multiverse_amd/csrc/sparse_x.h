// Sparse x operand of the class chains (f16x3 / bf16 inference; ConvLstmArgs::sx_*).
//
//   class encoder  x = scene_conv * one_hot(cell)  (code/pred_models.py:174-175, 210): zero
//                  except at the occupied cell -> the x part of the gate convolution is
//                  W_x[t]^T s at the <= 9 cells around it, s = the 64 scene channels there;
//   class decoder  x = grid_emb(one_hot(id)) = tanh(b) everywhere except the 3x3 around the
//                  hot cell (:912-923 in closed form) -> a constant per border class plus a
//                  correction within two cells of the hot cell, both functions of the
//                  WEIGHTS only.
// The gate kernel skips the x k-steps (20 % of an encoder launch, 11 % of a decoder one) and
// adds these table terms in its epilogue.  Tables are built on the device from the current
// weights (after mv_set_param / every optimizer step), accumulated in fp64.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mv {

__global__ void cell_yx_kernel(uint32_t* __restrict__ out, int H, int W) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < H * W) out[i] = ((uint32_t)(i / W) << 16) | (uint32_t)(i % W);
}

// is offset d (-1, 0, +1) inside the image for a coordinate of border class c (0 first,
// 1 interior, 2 last)?
__device__ __forceinline__ bool sx_inside(int c, int d) {
  return !((c == 0 && d < 0) || (c == 2 && d > 0));
}

// bias_tab [9][4C]; corr [9][25][4C].  w: gate kernel HWIO [3,3,Cx+C,4C] (x channels
// first), emb_w [3,3,1,E], emb_b [E], E == Cx.  One thread per (table row, column).
__global__ void sx_decoder_tables_kernel(const float* __restrict__ w,
                                         const float* __restrict__ bias,
                                         const float* __restrict__ emb_w,
                                         const float* __restrict__ emb_b, int Cx, int C,
                                         float* __restrict__ bias_tab,
                                         float* __restrict__ corr, int act = 0) {
  const int N4 = 4 * C, Cin = Cx + C;
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  const int rows = 9 + 9 * 25;
  if (idx >= rows * N4) return;
  const int row = idx / N4, col = idx - row * N4;
  if (row < 9) {          // bias + conv of the constant part over the taps inside the image
    const int cy = row / 3, cx = row % 3;
    double acc = 0.0;
    for (int t = 0; t < 9; ++t) {
      if (!sx_inside(cy, t / 3 - 1) || !sx_inside(cx, t % 3 - 1)) continue;
      for (int ch = 0; ch < Cx; ++ch)
        acc += (double)w[((size_t)t * Cin + ch) * N4 + col] * (double)act_apply(act, emb_b[ch]);
    }
    bias_tab[idx] = (float)((double)bias[col] + acc);
    return;
  }
  const int q = row - 9;
  const int hcls = q / 25, o = q - hcls * 25;
  const int oy = o / 5 - 2, ox = o % 5 - 2;      // cell - hot cell
  const int hcy = hcls / 3, hcx = hcls % 3;
  double acc = 0.0;
  for (int e = 0; e < 9; ++e) {                   // j = hot + e: a cell whose x is not tanh(b)
    const int ey = e / 3 - 1, ex = e % 3 - 1;
    if (!sx_inside(hcy, ey) || !sx_inside(hcx, ex)) continue;
    const int dy = ey - oy, dx = ex - ox;         // tap offset from the cell to j
    if (dy < -1 || dy > 1 || dx < -1 || dx > 1) continue;
    const int t = (dy + 1) * 3 + (dx + 1);
    const float* ew = emb_w + ((1 - ey) * 3 + (1 - ex)) * Cx;
    for (int ch = 0; ch < Cx; ++ch) {
      const double xd = (double)act_apply(act, ew[ch] + emb_b[ch]) - (double)act_apply(act, emb_b[ch]);
      acc += (double)w[((size_t)t * Cin + ch) * N4 + col] * xd;
    }
  }
  corr[(size_t)q * N4 + col] = (float)acc;
}

// Class-encoder step t: corr [N][9][4C], corr[n][o][col] = sum_ch W_x[tap with offset
// -o][ch][col] * scene_conv[frame(n, t)][hot][ch]  (o = cell - hot cell).  A workgroup =
// 256 columns of one offset for kSxRows rows: every weight is read once per kSxRows rows
// (one thread per output re-read the 2.4 MB of W_x per row: 37 us per launch).
constexpr int kSxRows = 8;
__global__ __launch_bounds__(256)
void sx_encoder_corr_kernel(const float* __restrict__ w, const float* __restrict__ conv,
                            const int32_t* __restrict__ obs_scene,
                            const int32_t* __restrict__ labels, int N, int T, int t, int K,
                            int D, int C, float* __restrict__ corr) {
  __shared__ float s[kSxRows][64];          // D <= 64 (host-checked)
  const int N4 = 4 * C, Cin = D + C;
  const int col = blockIdx.x * 256 + threadIdx.x;
  // blockIdx.z = step * row groups + row group: all T_o steps of the encoder in one launch
  // (t < 0), or the single step t
  const int ngrp = (N + kSxRows - 1) / kSxRows;
  const int o = blockIdx.y, n0 = (int)(blockIdx.z % ngrp) * kSxRows;
  if (t < 0) {
    t = (int)(blockIdx.z / ngrp);
    corr += (size_t)t * N * 9 * 4 * C;
  }
  for (int i = threadIdx.x; i < kSxRows * D; i += 256) {
    const int k = i / D, ch = i - k * D;
    const int n = n0 + k < N ? n0 + k : N - 1;
    s[k][ch] = conv[((size_t)obs_scene[n * T + t] * K + labels[n * T + t]) * D + ch];
  }
  __syncthreads();
  if (col >= N4) return;
  const int oy = o / 3 - 1, ox = o % 3 - 1;
  const int tap = (1 - oy) * 3 + (1 - ox);
  const float* wp = w + (size_t)tap * Cin * N4 + col;
  float acc[kSxRows];
#pragma unroll
  for (int k = 0; k < kSxRows; ++k) acc[k] = 0.f;
  for (int ch = 0; ch < D; ++ch) {
    const float wv = wp[(size_t)ch * N4];
#pragma unroll
    for (int k = 0; k < kSxRows; ++k) acc[k] = fmaf(wv, s[k][ch], acc[k]);
  }
#pragma unroll
  for (int k = 0; k < kSxRows; ++k)
    if (n0 + k < N) corr[((size_t)(n0 + k) * 9 + o) * N4 + col] = acc[k];
}

}  // namespace mv
