// ConvLSTM step as ONE kernel: 3x3 SAME gate convolution over concat[x, h] as
// an fp32 implicit GEMM on v_mfma_f32_32x32x2_f32, with the LSTM pointwise
// update fused into the accumulator epilogue.
//
// Replaces tf.contrib.rnn.ConvLSTMCell.call as constructed at reference
// code/pred_models.py:189-193,196-200,236-240,243-247:
//   g = conv2d_SAME(concat([x,h]), kernel[3,3,Cx+C,4C]) + biases
//   (i,j,f,o) = split(g,4);  c' = sigm(f+1)*c + sigm(i)*tanh(j);  h' = tanh(c')*sigm(o)
//
// GEMM view: M = rows*H*W cells, N = 4C gate columns, K = 9*(Cx+C).
// Workgroup tile: 128 cells x 128 columns, the 128 columns being the FOUR gates
// of ONE block of 32 channels (column order fixed at weight-pack time), so a
// lane's accumulators hold i,j,f,o of the same (cell, channel) and the LSTM
// update runs in registers: c is read once, c'/h' are written once, the
// [M,4C] pre-activation tensor never exists in memory.
// 4 waves, each 32 cells x 128 columns = 4 MFMA 32x32 tiles (64 accumulators).
// K is walked in chunks of 32: (channel group of 32) x (tap), channel-major so
// the nine shifted re-reads of one activation slab hit L1/L2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace mv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBM = 128;      // cells per workgroup
constexpr int kBN = 128;      // gate columns per workgroup (4 gates x 32 ch)
constexpr int kBK = 32;       // K per chunk
constexpr int kLdsStride = 36;  // floats per LDS row: 32 + 4 pad -> conflict-free b128
constexpr int kChBlock = 32;  // channels per workgroup

struct ConvLstmArgs {
  const float* x;       // [rows, H, W, Cx] dense input (x_mode 0/1)
  const float* h;       // [src rows, H, W, C]
  const float* c;       // [src rows, H, W, C]
  const int32_t* src_row_h;  // optional [rows]: row indirection for h (beam parents)
  const int32_t* src_row_c;  // optional [rows]: row indirection for c (beam parents)
  const float* wpack;   // packed weights, see pack_convlstm_weights()
  const float* bias;    // [4C] TF order (i|j|f|o)
  float* h_out;         // [rows, H, W, C]
  float* c_out;         // [rows, H, W, C]
  int32_t rows, H, W, Cx, C;
  int32_t n_xchunks;    // K chunks taken from x
  int32_t n_hchunks;    // K chunks taken from h (0 when the state is known zero)
  int32_t x_small;      // 1: 9*Cx <= 32, all taps of x packed in ONE chunk
  int32_t zero_state;   // 1: h == c == 0 (first encoder step): skip h, c reads
  int32_t n_mtiles;
  int32_t w_chunks;     // chunks per channel block in wpack (x + all h chunks)
  float forget_bias;
};

// Number of K chunks for an x operand of Cx channels.
static inline int convlstm_xchunks(int Cx) {
  if (Cx == 0) return 0;
  if (9 * Cx <= kBK) return 1;
  return 9 * (Cx / kBK);
}
static inline bool convlstm_cx_supported(int Cx) {
  return Cx == 0 || 9 * Cx <= kBK || (Cx % kBK) == 0;
}

// Host-side weight pack: TF HWIO kernel [3,3,Cx+C,4C] ->
//   wpack[cb][chunk][col(128)][k(32)]   (col = gate*32 + j <-> n = gate*C + cb*32 + j)
// chunk order: x chunks (channel-group-major, tap-minor; or the single packed
// small chunk k = tap*Cx + ch), then h chunks (channel-group-major, tap-minor).
static inline size_t convlstm_wpack_elems(int Cx, int C) {
  size_t nch = (size_t)convlstm_xchunks(Cx) + 9 * (size_t)(C / kBK);
  return (size_t)(C / kChBlock) * nch * kBN * kBK;
}
static inline void pack_convlstm_weights(const float* w, int Cx, int C,
                                         float* out) {
  const int Cin = Cx + C, N4 = 4 * C;
  const int nx = convlstm_xchunks(Cx), nh = 9 * (C / kBK), nch = nx + nh;
  const bool small = (Cx > 0 && 9 * Cx <= kBK);
  for (int cb = 0; cb < C / kChBlock; ++cb)
    for (int q = 0; q < nch; ++q)
      for (int col = 0; col < kBN; ++col) {
        const int gate = col / 32, j = col % 32;
        const int n = gate * C + cb * kChBlock + j;
        float* dst = out + (((size_t)cb * nch + q) * kBN + col) * kBK;
        for (int k = 0; k < kBK; ++k) {
          int tap = -1, ci = -1;
          if (q < nx) {
            if (small) {
              if (k < 9 * Cx) { tap = k / Cx; ci = k % Cx; }
            } else {
              tap = q % 9; ci = (q / 9) * kBK + k;
            }
          } else {
            const int qq = q - nx;
            tap = qq % 9; ci = Cx + (qq / 9) * kBK + k;
          }
          dst[k] = (tap < 0) ? 0.f : w[((size_t)tap * Cin + ci) * N4 + n];
        }
      }
}

__device__ __forceinline__ float sigmoidf_(float v) {
  return 1.0f / (1.0f + __expf(-v));
}

// Accurate-enough fp32 tanh / sigmoid: use the ocml implementations.
__device__ __forceinline__ float tanh_(float v) { return tanhf(v); }
__device__ __forceinline__ float sigm_(float v) { return 1.0f / (1.0f + expf(-v)); }

__global__ __launch_bounds__(256, 2)
void convlstm_step_kernel(const ConvLstmArgs a) {
  __shared__ __attribute__((aligned(16))) float lds[2 * kBM * kLdsStride];
  float* As = lds;
  float* Bs = lds + kBM * kLdsStride;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  // block -> (channel block, m tile).  Blocks are observed to round-robin over
  // the 8 XCDs by linear id, so cb = id % 8 keeps each XCD's L2 on ONE 1/8
  // slice of the packed weights (speed only, never correctness).
  const int ncb = a.C / kChBlock;
  const int cb = blockIdx.x % ncb;
  const int mt = blockIdx.x / ncb;
  const int H = a.H, W = a.W, HW = H * W, C = a.C, Cx = a.Cx;
  const int M_total = a.rows * HW;
  const int nchunks = a.n_xchunks + a.n_hchunks;

  // ---- staging roles: 8 threads cover the 32 channels (128 B) of one cell
  const int q4 = tid & 7;      // float4 slot within the chunk's 32 channels
  const int rr = tid >> 3;     // 0..31; rows rr + 32*p
  int ypos[4], xpos[4];
  ptrdiff_t xbase[4], hbase[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int m = mt * kBM + rr + 32 * p;
    if (m < M_total) {
      const int r = m / HW, cell = m - r * HW;
      const int y = cell / W;
      ypos[p] = y; xpos[p] = cell - y * W;
      const int sr = a.src_row_h ? a.src_row_h[r] : r;
      xbase[p] = (ptrdiff_t)m * Cx;
      hbase[p] = ((ptrdiff_t)sr * HW + cell) * C;
    } else {
      ypos[p] = -100000; xpos[p] = -100000; xbase[p] = 0; hbase[p] = 0;
    }
  }
  const float* wblk = a.wpack + (size_t)cb * a.w_chunks * kBN * kBK;

  f32x4 pa[4], pb[4];
  auto load_chunk = [&](int q) {
    // B: 128 cols x 32 k, contiguous 16 KB in the packed weights
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(wblk + (size_t)q * kBN * kBK);
#pragma unroll
    for (int i = 0; i < 4; ++i) pb[i] = wsrc[tid + 256 * i];
    // A: 128 cells x 32 k gathered at the chunk's tap
    if (q < a.n_xchunks) {
      if (a.x_small) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int k = q4 * 4 + j;
            if (k < 9 * Cx) {
              const int tap = k / Cx, ch = k - tap * Cx;
              const int dy = tap / 3 - 1, dx = tap % 3 - 1;
              const int yy = ypos[p] + dy, xx = xpos[p] + dx;
              if (yy >= 0 && yy < H && xx >= 0 && xx < W)
                v[j] = a.x[xbase[p] + (ptrdiff_t)(dy * W + dx) * Cx + ch];
            }
          }
          pa[p] = v;
        }
      } else {
        const int cg = q / 9, tap = q - cg * 9;
        const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
          const int yy = ypos[p] + dy, xx = xpos[p] + dx;
          f32x4 v = {0.f, 0.f, 0.f, 0.f};
          if (yy >= 0 && yy < H && xx >= 0 && xx < W)
            v = *reinterpret_cast<const f32x4*>(
                a.x + xbase[p] + (ptrdiff_t)(dy * W + dx) * Cx + cg * kBK + q4 * 4);
          pa[p] = v;
        }
      }
    } else {
      const int qq = q - a.n_xchunks;
      const int cg = qq / 9, tap = qq - cg * 9;
      const int dy = tap / 3 - 1, dx = tap % 3 - 1;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int yy = ypos[p] + dy, xx = xpos[p] + dx;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (yy >= 0 && yy < H && xx >= 0 && xx < W)
          v = *reinterpret_cast<const f32x4*>(
              a.h + hbase[p] + (ptrdiff_t)(dy * W + dx) * C + cg * kBK + q4 * 4);
        pa[p] = v;
      }
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;

  const float* Arow = As + (wave * 32 + (lane & 31)) * kLdsStride + (lane >> 5) * 4;
  const float* Bcol = Bs + (lane & 31) * kLdsStride + (lane >> 5) * 4;

  if (nchunks > 0) load_chunk(0);
  for (int q = 0; q < nchunks; ++q) {
    __syncthreads();  // all waves finished reading the previous chunk
#pragma unroll
    for (int p = 0; p < 4; ++p)
      *reinterpret_cast<f32x4*>(As + (rr + 32 * p) * kLdsStride + q4 * 4) = pa[p];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int f = tid + 256 * i;
      *reinterpret_cast<f32x4*>(Bs + (f >> 3) * kLdsStride + (f & 7) * 4) = pb[i];
    }
    __syncthreads();
    if (q + 1 < nchunks) load_chunk(q + 1);  // in flight under the MFMAs below
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const f32x4 av = *reinterpret_cast<const f32x4*>(Arow + kk * 8);
      f32x4 bv[4];
#pragma unroll
      for (int g = 0; g < 4; ++g)
        bv[g] = *reinterpret_cast<const f32x4*>(Bcol + g * 32 * kLdsStride + kk * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[j], bv[g][j], acc[g], 0, 0, 0);
    }
  }

  // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int ch = cb * kChBlock + (lane & 31);
  const float bi = a.bias[ch], bj = a.bias[C + ch], bf = a.bias[2 * C + ch],
              bo = a.bias[3 * C + ch];
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    const int m = mt * kBM + wave * 32 + row;
    if (m < M_total) {
      float cprev = 0.f;
      if (!a.zero_state) {
        const int r = m / HW, cell = m - r * HW;
        const int sr = a.src_row_c ? a.src_row_c[r] : r;
        cprev = a.c[((size_t)sr * HW + cell) * C + ch];
      }
      const float gi = acc[0][reg] + bi, gj = acc[1][reg] + bj,
                  gf = acc[2][reg] + bf, go = acc[3][reg] + bo;
      float cn = sigm_(gf + a.forget_bias) * cprev;
      cn = cn + sigm_(gi) * tanh_(gj);
      const float hn = tanh_(cn) * sigm_(go);
      a.c_out[(size_t)m * C + ch] = cn;
      a.h_out[(size_t)m * C + ch] = hn;
    }
  }
}

}  // namespace mv
