// The decoder tail of one step -- hidden2grid + argmax + the next step's input
// embedding (reference code/pred_models.py:925-959 hidden2grid, :411-415 argmax_onehot,
// :912-923 grid_emb) -- as TWO launches for all chains of a step
// instead of five per chain, each reading h' from HBM exactly once:
//
//   h2g_q_kernel      q[cell][tap*P + p] = sum_c h'[cell][c] * W[tap][c][p]
//                     a plain GEMM [cells x 256] x [256 x 9P] on v_mfma_f32_32x32x2_f32
//                     (exact fp32 fmaf chain).  The 3x3 convolution is linear, so the
//                     nine taps are applied to the cell's OWN vector and gathered
//                     afterwards: out[cell] = sum_tap q[cell + d_tap][tap].  The first
//                     version convolved in place: one wave per cell read its nine
//                     neighbours' 1 KB vectors (9x the L1/L2 traffic; 1.06 TB/s of
//                     algorithmic bytes, profiles/r2_*).
//   decode_tail_kernel  one workgroup per (chain, row): the 9-tap gather of q (36 B per
//                     cell), the output row (class logits / (dx, dy) maps), and, for
//                     the greedy decoders, argmax (lowest index on ties, tf.argmax) and
//                     the next input embedding with its operand planes as 16-byte
//                     stores.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "convlstm_mfma.h"
#include "kernels_misc.h"
#include "plane_layout.h"

namespace mv {

constexpr int kTailMax = 4;          // chains per grouped launch (class / regression x scales)

// W [3,3,C,P] HWIO -> wq[kk (C/8)][lane 64][4]: element j of lane l is
// Wq[c = kk*8 + (l>>5)*4 + j][col = l&31], Wq[c][tap*P + p] = W[tap][c][p], 0 for col >= 9P
// (the A side of the MFMA uses the same (lane, j) -> channel map, convlstm_mfma.h).
__global__ void pack_h2g_kernel(const float* __restrict__ w, float* __restrict__ wq, int C,
                                int P) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= C * 32) return;                       // (C/8) * 64 * 4
  const int j = idx & 3, l = (idx >> 2) & 63, kk = idx >> 8;
  const int c = kk * 8 + (l >> 5) * 4 + j, col = l & 31;
  float v = 0.f;
  if (col < 9 * P) {
    const int tap = col / P, p = col - tap * P;
    v = w[((size_t)tap * C + c) * P + p];
  }
  wq[idx] = v;
}

struct H2gQProblem {
  const float* h;        // [cells][C]
  const float* wq;       // pack_h2g_kernel
  float* q;              // [cells][9P]
  int32_t cells, P;
};
struct H2gQGroup {
  H2gQProblem p[kTailMax];
  int32_t block_end[kTailMax];
  int32_t n;
};

// One wave = 32 cells.  Round 1-3 form: two 16-byte loads of h' in flight per wave, each
// touching 32 cell rows (1 KB apart) -- ~2 KB per wave, ~22 KB per CU in flight, and the four
// loads that share a 128-byte line spread over two loop trips: 2.6 TB/s at batch 64 (latency
// x bytes in flight), 3.5 TB/s at 2 560 beam rows.  Now the wave issues the loads of a whole
// 256-channel chunk (32 x 16 bytes per lane = the cell's full 1 KB row: 128 registers) before
// the MFMA chain starts, the four loads of a line back to back, and the packed weights
// (C x 128 bytes, shared by the workgroup's four waves) wait in LDS instead of competing
// for the vector memory queue.  Same products in the same order: bit-identical results.
template <int NK>        // k-steps (8 channels each) per chunk: 32, or 16 when C == 128
__device__ __forceinline__ void h2g_q_body(const H2gQProblem& a, int block, int C, float* wl) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  {
    const f32x4* wg = reinterpret_cast<const f32x4*>(a.wq);
    f32x4* wd = reinterpret_cast<f32x4*>(wl);
    for (int i = threadIdx.x; i < C * 8; i += 256) wd[i] = wg[i];
  }
  __syncthreads();
  const int m_wave = block * 128 + wave * 32;
  if (m_wave >= a.cells) return;
  const int m = m_wave + (lane & 31);
  const float* src = a.h + (size_t)(m < a.cells ? m : 0) * C + (lane >> 5) * 4;
  const f32x4* wsrc = reinterpret_cast<const f32x4*>(wl) + lane;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int c0 = 0; c0 < C; c0 += NK * 8) {
    f32x4 av[NK];
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) av[kk] = *reinterpret_cast<const f32x4*>(src + c0 + kk * 8);
    // the machine scheduler would sink every load to its use (2 in flight, 40 registers)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int kk = 0; kk < NK; ++kk) {
      const f32x4 bv = wsrc[((c0 >> 3) + kk) * 64];
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[kk][j], bv[j], acc, 0, 0, 0);
    }
  }
  const int col = lane & 31, QS = 9 * a.P;
  if (col < QS) {
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
      const int mm = m_wave + row;
      if (mm < a.cells) a.q[(size_t)mm * QS + col] = acc[reg];
    }
  }
}

template <int NK>
__global__ __launch_bounds__(256)
void h2g_q_kernel(const H2gQGroup g, int C) {
  extern __shared__ __attribute__((aligned(16))) float h2g_wlds[];
  int block = blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int i = 0; i < kTailMax - 1; ++i)
    if (i + 1 < g.n && (int)blockIdx.x >= g.block_end[i]) pi = i + 1;
  if (pi > 0) block -= g.block_end[pi - 1];
  switch (pi) {
    case 0: h2g_q_body<NK>(g.p[0], block, C, h2g_wlds); break;
    case 1: h2g_q_body<NK>(g.p[1], block, C, h2g_wlds); break;
    case 2: h2g_q_body<NK>(g.p[2], block, C, h2g_wlds); break;
    default: h2g_q_body<NK>(g.p[3], block, C, h2g_wlds); break;
  }
}

// C % 128 == 0 (mv_create: hidden_size 128 / 256 / 512)
static inline void launch_h2g_q(const H2gQProblem* probs, int n, int C, hipStream_t stream) {
  H2gQGroup g{};
  g.n = n;
  unsigned total = 0;
  for (int i = 0; i < n; ++i) {
    g.p[i] = probs[i];
    total += (unsigned)((probs[i].cells + 127) / 128);
    g.block_end[i] = (int32_t)total;
  }
  for (int i = n; i < kTailMax; ++i) g.block_end[i] = (int32_t)total;
  const size_t lds = (size_t)C * 128;            // <= 64 KB
  if (C % 256 == 0)
    hipLaunchKernelGGL(h2g_q_kernel<32>, dim3(total), dim3(256), lds, stream, g, C);
  else
    hipLaunchKernelGGL(h2g_q_kernel<16>, dim3(total), dim3(256), lds, stream, g, C);
}

// ---- grid_emb with 16-byte plane stores.  item = (cell, group of 8 output channels);
// the fp32 row [E] of a cell is E / 8 x 32 B, the planes one 16-byte vector per
// item and plane (plane_index: tile (cell>>5, c8>>1), k half c8&1).
__device__ __forceinline__ void emb_store8(float* __restrict__ x, _Float16* p16, size_t pst,
                                           size_t m, int c8, int E, const float (&v)[8]) {
  f32x4 lo = {v[0], v[1], v[2], v[3]}, hi = {v[4], v[5], v[6], v[7]};
  f32x4* dst = reinterpret_cast<f32x4*>(x + m * E + c8 * 8);
  dst[0] = lo; dst[1] = hi;
  if (p16) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    h8 a, b;
    const size_t o = plane_index((long long)m, c8 * 8, E);
    if (pst == 0) {        // bf16 mode: one unscaled plane
#pragma unroll
      for (int j = 0; j < 8; ++j) a[j] = bf16_half_bits(v[j]);
      *reinterpret_cast<h8*>(p16 + o) = a;
      return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float s = v[j] * 256.0f;
      const _Float16 h0 = (_Float16)s;
      a[j] = h0;
      b[j] = (_Float16)(s - (float)h0);
    }
    *reinterpret_cast<h8*>(p16 + o) = a;
    *reinterpret_cast<h8*>(p16 + pst + o) = b;
  }
}

// grid_emb(one_hot(id)) in closed form (kernels_misc.h grid_emb_onehot_kernel), rows x K
// cells, E % 8 == 0; row m reads ids[(m / ids_div) * ids_stride].
__global__ __launch_bounds__(256)
void grid_emb_onehot8_kernel(const int32_t* __restrict__ ids, int ids_stride, int ids_div,
                             const float* __restrict__ w, const float* __restrict__ b,
                             float* __restrict__ out, int M, int H, int W, int E,
                             _Float16* p16, size_t p16_stride, int act = 0) {
  const int ng = E >> 3;
  const size_t item = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int K = H * W;
  if (item >= (size_t)M * K * ng) return;
  const int c8 = (int)(item % ng);
  const size_t mc = item / ng;                     // flat cell
  const int m = (int)(mc / K), cell = (int)(mc - (size_t)m * K);
  const int yy = cell / W, xx = cell - yy * W;
  const int id = ids[(size_t)(m / ids_div) * ids_stride];
  const int py = id / W, px = id - py * W;
  const int dy = yy - py, dx = xx - px;
  const bool hot = dy >= -1 && dy <= 1 && dx >= -1 && dx <= 1;
  const float* wp = w + (hot ? (1 - dy) * 3 + (1 - dx) : 4) * E + c8 * 8;   // always in range
  float v[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float acc = hot ? wp[j] : 0.f;
    v[j] = act_apply(act, acc + b[c8 * 8 + j]);
  }
  emb_store8(out, p16, p16_stride, mc, c8, E, v);
}

struct TailProblem {
  const float* q;          // [rows*K][9P]
  float* out;              // row r, cell, p -> out[r * out_row_stride + cell*P + p]
  int64_t out_row_stride;
  int32_t* ids_out;        // class, optional: argmax per row
  const float* emb_w;      // optional: next-input embedding [3,3,P,E], emb_b [E]
  const float* emb_b;
  float* x_out;            // [rows*K][E]
  _Float16* x16;           // optional operand planes of x_out
  int64_t x16_stride;
  int32_t rows, H, W, P, E;
  int32_t onehot;          // 1: class chain (argmax -> one-hot -> closed-form embedding)
  int32_t act;             // activation of grid_emb (mv_config.activation)
};
struct TailGroup {
  TailProblem p[kTailMax];
  int32_t block_end[kTailMax];
  int32_t n;
};

// dynamic LDS: vals [K*2] floats | emb W [9*2*E] | reduction scratch
__device__ __forceinline__ void decode_tail_body(const TailProblem& a, int row, float* sm) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int H = a.H, W = a.W, K = H * W, P = a.P, E = a.E, QS = 9 * P;
  float* vals = sm;                         // [K * P]
  float* wl = sm + 2 * 1024;                // [9 * P * E] (K * 2 <= 2048 host-checked)
  float* red = wl + 9 * 2 * E;              // [16] + [16] ints
  int* redi = reinterpret_cast<int*>(red + 16);
  const float* qrow = a.q + (size_t)row * K * QS;
  if (a.emb_w && !a.onehot)
    for (int i = tid; i < 9 * P * E; i += blockDim.x) wl[i] = a.emb_w[i];
  // ---- 9-tap gather: out[cell][p] = sum_t q[cell + d_t][t*P + p], taps in order
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int idx = tid; idx < K * P; idx += blockDim.x) {
    const int cell = idx / P, p = idx - cell * P;
    const int y = cell / W, x = cell - y * W;
    float acc = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W)
        acc += qrow[(size_t)(yy * W + xx) * QS + t * P + p];
    }
    vals[idx] = acc;
    a.out[(size_t)row * a.out_row_stride + idx] = acc;
    if (acc > best || (acc == best && idx < bi)) { best = acc; bi = idx; }
  }
  if (!a.emb_w && !a.ids_out) return;
  int hot = 0;
  if (a.onehot) {          // argmax over the K logits, lowest index wins ties
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { red[wave] = best; redi[wave] = bi; }
    __syncthreads();
    best = red[0]; bi = redi[0];
    for (int wv = 1; wv < (int)(blockDim.x >> 6); ++wv) {
      const float ov = red[wv];
      const int oi = redi[wv];
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    hot = bi;
    if (a.ids_out && tid == 0) a.ids_out[row] = hot;
  } else {
    __syncthreads();       // vals + wl complete
  }
  if (!a.emb_w) return;
  // ---- next input: x = tanh(conv3x3(one_hot(hot) | vals) + b), 8 channels per item
  const int ng = E >> 3;
  const int py = hot / W, px = hot - py * W;
  for (int item = tid; item < K * ng; item += blockDim.x) {
    const int cell = item / ng, c8 = item - cell * ng;
    const int y = cell / W, x = cell - y * W;
    float v[8];
    if (a.onehot) {
      const int dy = y - py, dx = x - px;
      const bool in = dy >= -1 && dy <= 1 && dx >= -1 && dx <= 1;
      const float* wp = a.emb_w + (in ? (1 - dy) * 3 + (1 - dx) : 4) * E + c8 * 8;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = act_apply(a.act, (in ? wp[j] : 0.f) + a.emb_b[c8 * 8 + j]);
    } else {
      float acc[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = 0.f;
      // same accumulation order as grid_emb_dense_kernel: taps, then input channels
      for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
        const float* ip = vals + (yy * W + xx) * P;
        for (int p = 0; p < P; ++p) {
          const float iv = ip[p];
          const float* wp = wl + (t * P + p) * E + c8 * 8;
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] = fmaf(iv, wp[j], acc[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = act_apply(a.act, acc[j] + a.emb_b[c8 * 8 + j]);
    }
    emb_store8(a.x_out, a.x16, (size_t)a.x16_stride, (size_t)row * K + cell, c8, E, v);
  }
}

// dynamic LDS of a decode-tail workgroup: the gathered values, the embedding kernel of the
// regression chain [9][2][E] and the argmax scratch; E = the largest emb_size of the launch
static inline size_t tail_lds_bytes(int E) {
  return ((size_t)2 * 1024 + (size_t)9 * 2 * E + 32) * sizeof(float);
}
constexpr int kTailThreads = 1024;    // one workgroup per (chain, row): rows are few, wide WGs

__global__ __launch_bounds__(kTailThreads)
void decode_tail_kernel(const TailGroup g) {
  extern __shared__ __attribute__((aligned(16))) float tail_sm[];
  int block = blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int i = 0; i < kTailMax - 1; ++i)
    if (i + 1 < g.n && (int)blockIdx.x >= g.block_end[i]) pi = i + 1;
  if (pi > 0) block -= g.block_end[pi - 1];
  switch (pi) {
    case 0: decode_tail_body(g.p[0], block, tail_sm); break;
    case 1: decode_tail_body(g.p[1], block, tail_sm); break;
    case 2: decode_tail_body(g.p[2], block, tail_sm); break;
    default: decode_tail_body(g.p[3], block, tail_sm); break;
  }
}

static inline void launch_decode_tail(const TailProblem* probs, int n, hipStream_t stream) {
  TailGroup g{};
  g.n = n;
  unsigned total = 0;
  for (int i = 0; i < n; ++i) {
    g.p[i] = probs[i];
    total += (unsigned)probs[i].rows;
    g.block_end[i] = (int32_t)total;
  }
  for (int i = n; i < kTailMax; ++i) g.block_end[i] = (int32_t)total;
  int emax = 32;
  for (int i = 0; i < n; ++i) emax = probs[i].E > emax ? probs[i].E : emax;
  hipLaunchKernelGGL(decode_tail_kernel, dim3(total), dim3(kTailThreads), tail_lds_bytes(emax),
                     stream, g);
}

}  // namespace mv
