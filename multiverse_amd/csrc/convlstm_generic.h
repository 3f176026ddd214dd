// ConvLSTM step, its dgrad and its wgrad for ANY square kernel size (--convlstm_kernel,
// code/train.py:70; tf.contrib.rnn.ConvLSTMCell kernel_shape = [k, k], SAME padding: tap t
// reads the cell at offset t - (k - 1) / 2, the extra pad of an even k goes bottom / right).
//
// Every fast gate kernel of this library -- fp32 MFMA, f16x3 direct, Winograd F(2,3) / F(3,3),
// their dgrad and wgrad -- is a 3 x 3 stencil by construction (tap enumeration in the packs,
// DPP column shifts, row pairs / triples).  The published configuration and every documented
// command line use 3; these kernels exist so that the OTHER values of the flag run at all:
// plain fp32 FMA loops straight from the HWIO kernel (no pack), register-tiled enough to be
// usable on small jobs, nowhere near the matrix pipe.  Same function, same epilogue, same
// gradient definitions as the 3 x 3 path (tests/test_gpu_generic_taps.py holds them to the
// oracle and to frozen runs of the reference).
#pragma once
#include "convlstm_mfma.h"

namespace mv {

constexpr int kGenCells = 8;        // cells of one image row per workgroup (forward / dgrad)

struct ConvGenericArgs {
  ConvLstmArgs f;          // geometry, x / h / c, bias, outputs, gates_out, row indirection
  const float* w;          // HWIO kernel [k, k, Cx + C, 4C]
  int32_t ksize;
};

// Forward.  Block = (8 consecutive cells of one image row, 256 channels); thread = channel:
// 8 x 4 gate accumulators; per (tap, input channel) 4 coalesced weight loads and 8 broadcast
// operand loads feed 32 FMAs.
__global__ __launch_bounds__(256)
void convlstm_step_generic_kernel(const ConvGenericArgs p) {
  const ConvLstmArgs& a = p.f;
  const int H = a.H, W = a.W, HW = H * W, C = a.C, Cx = a.Cx, k = p.ksize, pad = (k - 1) / 2;
  const int xt = (W + kGenCells - 1) / kGenCells;
  const int ch = blockIdx.y * 256 + threadIdx.x;
  int b = blockIdx.x;
  const int xb = b % xt; b /= xt;
  const int y = b % H;
  const int r = b / H;
  const int x0 = xb * kGenCells;
  const int srh = a.src_row_h ? a.src_row_h[r] : r;
  const int src = a.src_row_c ? a.src_row_c[r] : r;
  const bool live = ch < C;
  const int chc = live ? ch : 0;
  const int Cin = Cx + C, N4 = 4 * C;
  float acc[kGenCells][4];
#pragma unroll
  for (int j = 0; j < kGenCells; ++j)
#pragma unroll
    for (int g = 0; g < 4; ++g) acc[j][g] = 0.f;
  for (int ky = 0; ky < k; ++ky) {
    const int yy = y + ky - pad;
    if (yy < 0 || yy >= H) continue;
    for (int kx = 0; kx < k; ++kx) {
      const float* wt = p.w + (size_t)(ky * k + kx) * Cin * N4 + chc;
      for (int ci = 0; ci < Cin; ++ci) {
        const bool is_x = ci < Cx;
        if (!is_x && a.zero_state) break;
        float wv[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) wv[g] = wt[(size_t)ci * N4 + g * C];
#pragma unroll
        for (int j = 0; j < kGenCells; ++j) {
          const int xx = x0 + j + kx - pad;
          float v = 0.f;
          if (xx >= 0 && xx < W && x0 + j < W)
            v = is_x ? a.x[(size_t)r * a.x_row_stride + (size_t)(yy * W + xx) * Cx + ci]
                     : a.h[((size_t)srh * HW + yy * W + xx) * C + (ci - Cx)];
#pragma unroll
          for (int g = 0; g < 4; ++g) acc[j][g] = fmaf(v, wv[g], acc[j][g]);
        }
      }
    }
  }
  if (!live) return;
  const float bi = a.bias[ch], bj = a.bias[C + ch], bf = a.bias[2 * C + ch], bo = a.bias[3 * C + ch];
#pragma unroll
  for (int j = 0; j < kGenCells; ++j) {
    const int x = x0 + j;
    if (x >= W) break;
    const size_t m = (size_t)r * HW + y * W + x;
    const float cprev = a.zero_state ? 0.f : a.c[((size_t)src * HW + y * W + x) * C + ch];
    const float si = sigm_(acc[j][0] + bi), tj = tanh_(acc[j][1] + bj),
                sf = sigm_(acc[j][2] + bf + a.forget_bias), so = sigm_(acc[j][3] + bo);
    float cn = sf * cprev;
    cn = cn + si * tj;
    const float hn = tanh_(cn) * so;
    a.c_out[m * C + ch] = cn;
    a.h_out[m * C + ch] = hn;
    if (a.gates_out) {
      float* gp = a.gates_out + m * 4 * C + ch;
      gp[0] = si; gp[C] = tj; gp[2 * C] = sf; gp[3 * C] = so;
    }
  }
}

static inline void launch_convlstm_generic_step(const ConvGenericArgs& p, hipStream_t stream) {
  const ConvLstmArgs& a = p.f;
  const unsigned xt = (unsigned)((a.W + kGenCells - 1) / kGenCells);
  hipLaunchKernelGGL(convlstm_step_generic_kernel, dim3(xt * a.H * a.rows, (a.C + 255) / 256),
                     dim3(256), 0, stream, p);
}

// dgrad: d in[m][ci] = sum_{tap, n} G[m - d_tap][n] W[tap][ci][n]  (in(m) feeds out(m - d_tap)).
// ci < Cx -> out1 [M][Cx] (d x), else out0 [M][C] (d h).  Block = (8 cells of a row, 256 input
// channels); thread = input channel, 8 accumulators, its own weight row walked along n.
struct ConvGenericDgradArgs {
  const float* g;          // [M][4C] gate gradients
  const float* w;          // HWIO kernel
  float* dh;               // [M][C] or null
  float* dx;               // [M][Cx] or null
  int32_t rows, H, W, Cx, C, ksize;
};
__global__ __launch_bounds__(256)
void convlstm_dgrad_generic_kernel(const ConvGenericDgradArgs p) {
  const int H = p.H, W = p.W, HW = H * W, C = p.C, Cx = p.Cx, k = p.ksize, pad = (k - 1) / 2;
  const int Cin = Cx + C, N4 = 4 * C;
  const int xt = (W + kGenCells - 1) / kGenCells;
  const int ci = blockIdx.y * 256 + threadIdx.x;
  int b = blockIdx.x;
  const int xb = b % xt; b /= xt;
  const int y = b % H;
  const int r = b / H;
  const int x0 = xb * kGenCells;
  const bool live = ci < Cin;
  const int cic = live ? ci : 0;
  float acc[kGenCells];
#pragma unroll
  for (int j = 0; j < kGenCells; ++j) acc[j] = 0.f;
  for (int ky = 0; ky < k; ++ky) {
    const int yy = y - (ky - pad);               // the output row this tap of in(y) feeds
    if (yy < 0 || yy >= H) continue;
    for (int kx = 0; kx < k; ++kx) {
      const float* wr = p.w + ((size_t)(ky * k + kx) * Cin + cic) * N4;
      for (int n = 0; n < N4; ++n) {
        const float wv = wr[n];
#pragma unroll
        for (int j = 0; j < kGenCells; ++j) {
          const int xx = x0 + j - (kx - pad);
          float gv = 0.f;
          if (xx >= 0 && xx < W && x0 + j < W) gv = p.g[((size_t)r * HW + yy * W + xx) * N4 + n];
          acc[j] = fmaf(gv, wv, acc[j]);
        }
      }
    }
  }
  if (!live) return;
#pragma unroll
  for (int j = 0; j < kGenCells; ++j) {
    const int x = x0 + j;
    if (x >= W) break;
    const size_t m = (size_t)r * HW + y * W + x;
    if (ci < Cx) { if (p.dx) p.dx[m * Cx + ci] = acc[j]; }
    else if (p.dh) p.dh[m * C + (ci - Cx)] = acc[j];
  }
}
static inline void launch_convlstm_generic_dgrad(const ConvGenericDgradArgs& p,
                                                 hipStream_t stream) {
  const unsigned xt = (unsigned)((p.W + kGenCells - 1) / kGenCells);
  hipLaunchKernelGGL(convlstm_dgrad_generic_kernel,
                     dim3(xt * p.H * p.rows, (p.Cx + p.C + 255) / 256), dim3(256), 0, stream, p);
}

// wgrad: dW[tap][ci][n] = sum_m in[m + d_tap][ci] G[m][n], in = [x | h] of the chain's R images.
// Block = (tap, 8 input channels, 256 columns); thread = column n: 8 accumulators, the sum over
// ALL cells in a fixed order (deterministic, no split).  Writes (not adds) dW.
constexpr int kGenWgCi = 8;
struct ConvGenericWgradArgs {
  const float* x;          // [R][HW][Cx] or null
  const float* h;          // [R][HW][C]
  const float* g;          // [R][HW][4C]
  float* dw;               // [k][k][Cx + C][4C]
  int32_t R, H, W, Cx, C, ksize;
};
__global__ __launch_bounds__(256)
void convlstm_wgrad_generic_kernel(const ConvGenericWgradArgs p) {
  const int H = p.H, W = p.W, HW = H * W, C = p.C, Cx = p.Cx, k = p.ksize, pad = (k - 1) / 2;
  const int Cin = Cx + C, N4 = 4 * C;
  const int n = blockIdx.x * 256 + threadIdx.x;
  const int ci0 = blockIdx.y * kGenWgCi;
  const int tap = blockIdx.z;
  const int dy = tap / k - pad, dx = tap % k - pad;
  const bool live = n < N4;
  const int nc = live ? n : 0;
  float acc[kGenWgCi];
#pragma unroll
  for (int j = 0; j < kGenWgCi; ++j) acc[j] = 0.f;
  for (int r = 0; r < p.R; ++r)
    for (int y = 0; y < H; ++y) {
      const int yy = y + dy;
      if (yy < 0 || yy >= H) continue;
      for (int x = 0; x < W; ++x) {
        const int xx = x + dx;
        if (xx < 0 || xx >= W) continue;
        const float gv = p.g[((size_t)r * HW + y * W + x) * N4 + nc];
        const size_t mi = (size_t)r * HW + yy * W + xx;
#pragma unroll
        for (int j = 0; j < kGenWgCi; ++j) {
          const int ci = ci0 + j;
          float v = 0.f;
          if (ci < Cx) v = p.x[mi * Cx + ci];
          else if (ci < Cin) v = p.h[mi * C + (ci - Cx)];
          acc[j] = fmaf(v, gv, acc[j]);
        }
      }
    }
  if (!live) return;
#pragma unroll
  for (int j = 0; j < kGenWgCi; ++j) {
    const int ci = ci0 + j;
    if (ci < Cin) p.dw[((size_t)tap * Cin + ci) * N4 + n] = acc[j];
  }
}
static inline void launch_convlstm_generic_wgrad(const ConvGenericWgradArgs& p,
                                                 hipStream_t stream) {
  hipLaunchKernelGGL(convlstm_wgrad_generic_kernel,
                     dim3((4 * p.C + 255) / 256, (p.Cx + p.C + kGenWgCi - 1) / kGenWgCi,
                          p.ksize * p.ksize),
                     dim3(256), 0, stream, p);
}

}  // namespace mv
