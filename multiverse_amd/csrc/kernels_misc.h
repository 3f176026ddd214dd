// The HBM-bound members of the Multiverse hot path: scene encoder, decoder
// input embeddings, graph attention, hidden2grid + argmax, beam expansion.
// Every kernel cites the reference lines it replaces.
#pragma once
#include <hip/hip_runtime.h>
#include "plane_layout.h"
#include <stdint.h>
#include <math.h>

namespace mv {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

// f16x3 compute mode (convlstm_f16x3.h): a producer of a ConvLSTM operand can
// emit the two pre-scaled fp16 planes of its output next to the fp32 value, so
// that no separate split pass is needed.  p16 == nullptr: off.
// stride == 0 selects the bf16 mode: ONE plane holding bf16(v) (compute mode 2).
__device__ __forceinline__ _Float16 bf16_half_bits(float v) {
  uint32_t u = __builtin_bit_cast(uint32_t, v);
  if ((u & 0x7f800000u) != 0x7f800000u) u += 0x7fffu + ((u >> 16) & 1u);   // RNE
  return __builtin_bit_cast(_Float16, (uint16_t)(u >> 16));
}
__device__ __forceinline__ void emit_planes(_Float16* p16, size_t stride, size_t idx, int C,
                                            float v) {
  if (!p16) return;
  const size_t m = idx / (size_t)C;
  const size_t o = plane_index((long long)m, (int)(idx - m * C), C);
  if (stride == 0) { p16[o] = bf16_half_bits(v); return; }
  const float s = v * 256.0f;
  const _Float16 h0 = (_Float16)s;
  p16[o] = h0;
  p16[stride + o] = (_Float16)(s - (float)h0);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}

// ---------------------------------------------------------------- activation
// --activation_func of the reference (code/train.py:58-59 -> code/pred_utils.py:86-94):
// tanh (published), relu, lrelu = tf.nn.leaky_relu with its default alpha 0.2.  It is the
// activation of the scene convolutions (code/pred_models.py:155-165) and of grid_emb
// (:444, :664); the ConvLSTM cells and hidden2grid do not use it.  mv_config.activation.
constexpr int kActTanh = 0, kActRelu = 1, kActLrelu = 2;
__device__ __forceinline__ float act_apply(int act, float v) {
  if (act == kActTanh) return tanhf(v);
  if (act == kActRelu) return v > 0.f ? v : 0.f;
  return v > 0.f ? v : 0.2f * v;
}
// d act / d pre-activation from the OUTPUT y (lrelu: y > 0 <=> pre-activation > 0)
__device__ __forceinline__ float act_grad(int act, float y) {
  if (act == kActTanh) return 1.f - y * y;
  if (act == kActRelu) return y > 0.f ? 1.f : 0.f;
  return y > 0.f ? 1.f : 0.2f;
}

// ---------------------------------------------------------------- scene conv
// conv k x k, stride 2, SAME, + b, tanh (reference code/pred_models.py:155-160
// via conv2d :1333-1373) on the U unique frames of the batch; one thread per
// output element.  in [U, Hi, Wi, Ci], w [k,k,Ci,Co] HWIO, out [U, Ho, Wo, Co].
__global__ void scene_conv_s2_tanh_kernel(const float* __restrict__ in,
                                          const float* __restrict__ w,
                                          const float* __restrict__ b,
                                          float* __restrict__ out, int U, int Hi,
                                          int Wi, int Ci, int Ho, int Wo, int Co,
                                          int k, int pad_t, int pad_l, int act = 0) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)U * Ho * Wo * Co;
  if (idx >= total) return;
  const int co = idx % Co;
  size_t r = idx / Co;
  const int ox = r % Wo; r /= Wo;
  const int oy = r % Ho;
  const int u = r / Ho;
  float acc = 0.f;
  for (int ky = 0; ky < k; ++ky) {
    const int iy = oy * 2 + ky - pad_t;
    if (iy < 0 || iy >= Hi) continue;
    for (int kx = 0; kx < k; ++kx) {
      const int ix = ox * 2 + kx - pad_l;
      if (ix < 0 || ix >= Wi) continue;
      const float* ip = in + (((size_t)u * Hi + iy) * Wi + ix) * Ci;
      const float* wp = w + ((size_t)(ky * k + kx) * Ci) * Co + co;
      for (int ci = 0; ci < Ci; ++ci) acc = fmaf(ip[ci], wp[(size_t)ci * Co], acc);
    }
  }
  out[idx] = act_apply(act, acc + b[co]);
}

// --scene_conv_kernel 1 (code/train.py:65, conv2d code/pred_models.py:155-165): the scene
// stack degenerates to a strided 1x1 projection, a true dense GEMM
//   out[(u, oy, ox)][co] = tanh(sum_ci in[u, 2 oy, 2 ox, ci] W[ci][co] + b[co])
// (SAME padding of a 1x1 stride-2 conv is zero: the even input positions), M = U Ho Wo
// rows, K = Ci (11 / 64), N = Co <= 64 -- the one place BASELINE.json's north_star puts on
// the matrix cores.  v_mfma_f32_32x32x2_f32: exact fp32, the same ci-ordered fmaf chain as
// the generic kernel above.  A wave owns 32 rows x Co columns (two accumulators).
typedef float f32x16_t __attribute__((ext_vector_type(16)));
__global__ __launch_bounds__(256)
void scene_proj1x1_mfma_kernel(const float* __restrict__ in, const float* __restrict__ w,
                               const float* __restrict__ b, float* __restrict__ out, int U,
                               int Hi, int Wi, int Ci, int Ho, int Wo, int Co, int act = 0) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int M = U * Ho * Wo;
  const int m_wave = (blockIdx.x * 4 + wave) * 32;
  if (m_wave >= M) return;
  const int m = m_wave + (lane & 31);
  const int kh = lane >> 5;                       // which of the k-step's two channels
  size_t src = 0;
  if (m < M) {
    const int ox = m % Wo, r = m / Wo, oy = r % Ho, u = r / Ho;
    src = (((size_t)u * Hi + 2 * oy) * Wi + 2 * ox) * Ci;
  }
  f32x16_t acc0, acc1;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
  const int col = lane & 31;
  for (int kk = 0; kk < (Ci + 1) / 2; ++kk) {
    const int ci = 2 * kk + kh;
    const bool okc = ci < Ci;
    const float a = (okc && m < M) ? in[src + ci] : 0.f;
    const float b0 = (okc && col < Co) ? w[(size_t)ci * Co + col] : 0.f;
    const float b1 = (okc && 32 + col < Co) ? w[(size_t)ci * Co + 32 + col] : 0.f;
    acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b0, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b1, acc1, 0, 0, 0);
  }
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    const int mm = m_wave + row;
    if (mm >= M) continue;
    if (col < Co) out[(size_t)mm * Co + col] = act_apply(act, acc0[reg] + b[col]);
    if (32 + col < Co) out[(size_t)mm * Co + 32 + col] = act_apply(act, acc1[reg] + b[32 + col]);
  }
}

// mean over the T_o observed frames of the per-sample scene feature
// (tf.reduce_mean(scene_features, axis=1), code/pred_models.py:826-828).
// conv [U, K, D] (K = H*W), obs_scene [N, T], out [N, K, D].
__global__ void scene_mean_kernel(const float* __restrict__ conv,
                                  const int32_t* __restrict__ obs_scene,
                                  float* __restrict__ out, int N, int T, int K,
                                  int D) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per = (size_t)K * D;
  if (idx >= (size_t)N * per) return;
  const int n = idx / per;
  const size_t off = idx - (size_t)n * per;
  float s = 0.f;
  for (int t = 0; t < T; ++t) s += conv[(size_t)obs_scene[n * T + t] * per + off];
  out[idx] = s / (float)T;
}

// Class-encoder input: scene feature masked to the occupied cell
// (tf.multiply(scene_convs[i], one_hot), code/pred_models.py:174-175,210).
// out [N, K, D] for time step t.
__global__ void enc_class_input_kernel(const float* __restrict__ conv,
                                       const int32_t* __restrict__ obs_scene,
                                       const int32_t* __restrict__ labels,
                                       float* __restrict__ out, int N, int T,
                                       int t, int K, int D, _Float16* p16 = nullptr,
                                       size_t p16_stride = 0,
                                       const int32_t* __restrict__ labels2 = nullptr,
                                       float mixw = 1.f) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per = (size_t)K * D;
  if (idx >= (size_t)N * per) return;
  const int n = idx / per;
  const size_t off = idx - (size_t)n * per;
  const int cell = off / D;
  float v = 0.f;
  if (labels2) {
    // label mixup (SimAug/code/pred_models.py:616-636): the one-hot map is
    // w * one_hot(l1) + one_hot(l2) * (1 - w), then the same product
    const float m = mixw * (cell == labels[n * T + t] ? 1.f : 0.f) +
                    (cell == labels2[n * T + t] ? 1.f : 0.f) * (1.f - mixw);
    if (m != 0.f) v = conv[(size_t)obs_scene[n * T + t] * per + off] * m;
  } else if (cell == labels[n * T + t])
    v = conv[(size_t)obs_scene[n * T + t] * per + off];
  out[idx] = v;
  emit_planes(p16, p16_stride, idx, D, v);
}

// ---------------------------------------------------------------- grid_emb
// tanh(conv3x3_SAME(x) + b), P in {1,2} input channels, E outputs
// (grid_emb, code/pred_models.py:912-919).  Dense form, one thread per output.
// x [M, H, W, P] with row stride x_row_stride (elements) so a time slice of a
// [N, T, H, W, P] tensor can be embedded without a copy.
__global__ void grid_emb_dense_kernel(const float* __restrict__ x,
                                      size_t x_row_stride,
                                      const float* __restrict__ w,
                                      const float* __restrict__ b,
                                      float* __restrict__ out, int M, int H,
                                      int W, int P, int E, _Float16* p16 = nullptr,
                                      size_t p16_stride = 0, int act = 0) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)M * H * W * E;
  if (idx >= total) return;
  const int e = idx % E;
  size_t r = idx / E;
  const int xx = r % W; r /= W;
  const int yy = r % H;
  const int m = r / H;
  float acc = 0.f;
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = yy + ky - 1;
    if (iy < 0 || iy >= H) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = xx + kx - 1;
      if (ix < 0 || ix >= W) continue;
      const float* ip = x + (size_t)m * x_row_stride + ((size_t)iy * W + ix) * P;
      for (int p = 0; p < P; ++p)
        acc = fmaf(ip[p], w[((ky * 3 + kx) * P + p) * E + e], acc);
    }
  }
  const float v = act_apply(act, acc + b[e]);
  out[idx] = v;
  emit_planes(p16, p16_stride, idx, E, v);
}

// One-hot input in closed form: the cell at offset (dy,dx) from the hot cell
// sees exactly tap (1-dy, 1-dx); every other cell sees tanh(b).
// (grid_emb of tf.one_hot(argmax), code/pred_models.py:411-425, 442-446,
// 602-606.)  Row m reads ids[(m / ids_div) * ids_stride], so the obs labels
// [N,T] column T-1, tiled over beams (ids_div = B), or per-beam ids are used in
// place.  out [M, H, W, E].
__global__ void grid_emb_onehot_kernel(const int32_t* __restrict__ ids,
                                       int ids_stride, int ids_div,
                                       const float* __restrict__ w,
                                       const float* __restrict__ b,
                                       float* __restrict__ out, int M, int H,
                                       int W, int E, _Float16* p16 = nullptr,
                                       size_t p16_stride = 0, int act = 0) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)M * H * W * E;
  if (idx >= total) return;
  const int e = idx % E;
  size_t r = idx / E;
  const int xx = r % W; r /= W;
  const int yy = r % H;
  const int m = r / H;
  const int id = ids[(size_t)(m / ids_div) * ids_stride];
  const int py = id / W, px = id - py * W;
  const int dy = yy - py, dx = xx - px;
  float acc = 0.f;
  if (dy >= -1 && dy <= 1 && dx >= -1 && dx <= 1)
    acc = w[((1 - dy) * 3 + (1 - dx)) * E + e];  // P == 1
  const float v = act_apply(act, acc + b[e]);
  out[idx] = v;
  emit_planes(p16, p16_stride, idx, E, v);
}

// ---------------------------------------------------------------- graph attention
// h_out = h + sum_j softmax_j(<f_i, f_j>) h_j over the <= 9 in-bounds 3x3
// neighbours, f = l2_normalize([h ; scene_mean]).  The reference builds the
// dense K x K edge matrix, adds -1e30 off the neighbourhood and softmaxes
// (gnn_edge / gnn_mask_edge / gnn_node, code/pred_models.py:808-909,
// exp_mask :1399-1401); exp(-1e30 - max) is exactly 0 in fp32, so the 9-point
// stencil is the same function.  One wave per cell: lane l holds channels
// 256 g + 4l..4l+3 of h (NG groups: C <= 256 NG) and channels l, l + 64 of the scene mean
// (D <= 128).
// scene_mean rows are indexed by m / sm_div (beam tiling, :831-834).
// src_row: optional state-row indirection (beam parents): input row for
// output row r is src_row[r].
// Channel groups of the one-wave-per-cell kernels: lane l owns channels 256 g + 4 l .. + 3 of
// group g < NG (C <= 256 NG: hidden sizes 128 / 256 / 512; lanes past C idle with zeros).
template <int NG>
struct CVec {
  f32x4_t v[NG];
};
template <int NG>
__device__ __forceinline__ CVec<NG> cvec_load(const float* __restrict__ row, int lane, int C) {
  CVec<NG> r;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int c0 = g * 256 + lane * 4;
    r.v[g] = c0 < C ? *reinterpret_cast<const f32x4_t*>(row + c0) : f32x4_t{0.f, 0.f, 0.f, 0.f};
  }
  return r;
}
// the scene part of a cell's feature in the one-wave-per-cell kernels: lane l owns scene
// channels l and l + 64 (--scene_conv_dim up to 128; the published 64 leaves .b zero)
struct SVec { float a, b; };
__device__ __forceinline__ SVec svec_load(const float* __restrict__ row, int lane, int D) {
  SVec r;
  r.a = lane < D ? row[lane] : 0.f;
  r.b = lane + 64 < D ? row[lane + 64] : 0.f;
  return r;
}
__device__ __forceinline__ float svec_dot(const SVec& x, const SVec& y) {
  return x.a * y.a + x.b * y.b;
}
template <int NG>
__device__ __forceinline__ float cvec_dot(const CVec<NG>& a, const CVec<NG>& b) {
  float s = 0.f;
#pragma unroll
  for (int g = 0; g < NG; ++g)
    s += a.v[g][0] * b.v[g][0] + a.v[g][1] * b.v[g][1] + a.v[g][2] * b.v[g][2] +
         a.v[g][3] * b.v[g][3];
  return s;
}

template <int NG>
__global__ __launch_bounds__(256)
void gnn_attend_kernel(const float* __restrict__ h,
                       const float* __restrict__ scene_mean,
                       const int32_t* __restrict__ src_row,
                       float* __restrict__ out, int M, int H, int W, int C,
                       int D, int sm_div, _Float16* p16 = nullptr, size_t p16_stride = 0) {
  const int lane = threadIdx.x & 63;
  const size_t cell_id = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int K = H * W;
  if (cell_id >= (size_t)M * K) return;
  const int m = cell_id / K;
  const int cell = cell_id - (size_t)m * K;
  const int y = cell / W, x = cell - y * W;
  const int ms = src_row ? src_row[m] : m;
  const float* hrow = h + (size_t)ms * K * C;
  const float* srow = scene_mean + (size_t)(m / sm_div) * K * D;

  const CVec<NG> hi = cvec_load<NG>(hrow + (size_t)cell * C, lane, C);
  const SVec si = svec_load(srow + (size_t)cell * D, lane, D);
  float ssi = cvec_dot<NG>(hi, hi) + svec_dot(si, si);
  ssi = wave_sum(ssi);
  const float invi = rsqrtf(fmaxf(ssi, 1e-12f));

  CVec<NG> hj[9];
  float e[9];
  bool ok[9];
  float emax = -INFINITY;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
    ok[t] = (yy >= 0 && yy < H && xx >= 0 && xx < W);
    e[t] = 0.f;
#pragma unroll
    for (int g = 0; g < NG; ++g) hj[t].v[g] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    if (ok[t]) {  // wave-uniform
      const int cj = yy * W + xx;
      hj[t] = cvec_load<NG>(hrow + (size_t)cj * C, lane, C);
      const SVec sj = svec_load(srow + (size_t)cj * D, lane, D);
      float ssj = cvec_dot<NG>(hj[t], hj[t]) + svec_dot(sj, sj);
      float dot = cvec_dot<NG>(hi, hj[t]) + svec_dot(si, sj);
      ssj = wave_sum(ssj);
      dot = wave_sum(dot);
      e[t] = dot * invi * rsqrtf(fmaxf(ssj, 1e-12f));
      emax = fmaxf(emax, e[t]);
    }
  }
  float den = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    e[t] = ok[t] ? expf(e[t] - emax) : 0.f;
    den += e[t];
  }
  const float inv = 1.0f / den;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int c0 = g * 256 + lane * 4;
    if (c0 >= C) continue;
    f32x4_t node = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float a = e[t] * inv;
      node[0] = fmaf(a, hj[t].v[g][0], node[0]);
      node[1] = fmaf(a, hj[t].v[g][1], node[1]);
      node[2] = fmaf(a, hj[t].v[g][2], node[2]);
      node[3] = fmaf(a, hj[t].v[g][3], node[3]);
    }
    f32x4_t o = {hi.v[g][0] + node[0], hi.v[g][1] + node[1], hi.v[g][2] + node[2],
                 hi.v[g][3] + node[3]};
    *reinterpret_cast<f32x4_t*>(out + ((size_t)m * K + cell) * C + c0) = o;
    if (p16) {
      typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
      f16x4_t a, b;
      const size_t idx = plane_index((long long)m * K + cell, c0, C);   // 4 of one 8-group
      if (p16_stride == 0) {
#pragma unroll
        for (int j = 0; j < 4; ++j) a[j] = bf16_half_bits(o[j]);
        *reinterpret_cast<f16x4_t*>(p16 + idx) = a;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float sc = o[j] * 256.0f;
          const _Float16 h0 = (_Float16)sc;
          a[j] = h0;
          b[j] = (_Float16)(sc - (float)h0);
        }
        *reinterpret_cast<f16x4_t*>(p16 + idx) = a;
        *reinterpret_cast<f16x4_t*>(p16 + p16_stride + idx) = b;
      }
    }
  }
}

// ---- graph attention, second version: LDS-tiled.
// The first version (above) gives one wave to a cell: it gathers the nine neighbours'
// 1.25 KB feature vectors from global memory (9x the L1/L2 traffic of the tensor) and
// spends most of its instructions in 19 wave-wide reductions per cell; measured 1.2 TB/s
// of algorithmic bytes (profiles/r2_*).  Here a workgroup owns 32 consecutive cells of
// the flat [rows*K] index (= one operand-plane tile row) and stages the contiguous range
// of cells that holds every in-image neighbour (32 + 2W + 2 <= 98 cells) through LDS in
// chunks of 64 channels:
//   pass 1  (4 chunks of h + 1 of scene_mean)  partial dot products: thread = (cell,
//           8-channel slice) accumulates its 9 dots over the chunks; |u|^2 of every staged
//           cell the same way; 8-lane shuffles at the end -> e_ij, softmax weights in LDS;
//   pass 2  (4 chunks of h again, an L2 hit)   out = h_i + sum_j a_ij h_j: thread = (cell,
//           8 channels), results leave as 16-byte operand-plane vectors (1 KB contiguous
//           per wave) and, when a consumer needs them, fp32 rows.
// Consecutive 32-cell groups are mapped to the same XCD (block % 8 -> contiguous range of
// groups) so that a group's halo cells are its neighbours' own cells in that L2.
constexpr int kGnnPitch = 68;        // floats per staged cell and chunk (64 + 4: bank spread)
// cells per workgroup: MV_GNN_CELLS = 32 (256 threads, 98 staged cells: every staged byte is
// read 3.1 x per pass) or 64 (512 threads, 130 staged cells: 2.0 x).  Same arithmetic per cell
// in the same order: bitwise the same result.  Measured (beam 20, 2 560 rows, same box): 64
// cells 16.7 ms per forward, 32 cells 15.8 -- the kernel is not bound by its L2 re-reads.
#ifndef MV_GNN_CELLS
#define MV_GNN_CELLS 32
#endif
constexpr int kGnnCells = MV_GNN_CELLS;
constexpr int kGnnThreads = kGnnCells * 8;
constexpr int kGnnStage = kGnnCells + 2 * 32 + 2;   // + a halo of W + 1 <= 33 cells each side

// One launch serves up to two problems (the two grid scales of a greedy decode step):
// blocks [0, nblocks0) belong to p[0], the rest to p[1]; both counts are multiples of 8,
// so the block -> XCD mapping below holds for each.
struct GnnProblem {
  const float* h; const float* scene_mean; const int32_t* src_row; float* out;
  _Float16* p16; size_t p16_stride;
  int M, H, W, sm_div, ngroups;
  const int32_t* row_ref;   // optional [M]: rows with 0 are not computed (no beam continues them)
};
struct GnnGroup { GnnProblem p[2]; unsigned nblocks0; };

__global__ __launch_bounds__(kGnnThreads)
void gnn_attend_v2_kernel(const GnnGroup grp, int C, int D) {
  __shared__ __attribute__((aligned(16))) float buf[kGnnStage * kGnnPitch];
  __shared__ float ssq[kGnnStage];
  __shared__ float edot[kGnnCells * 9];
  __shared__ float alpha[kGnnCells * 9];
  __shared__ int hsrc[kGnnStage], ssrc[kGnnStage];
  const bool second = blockIdx.x >= grp.nblocks0;
  const GnnProblem& pr = grp.p[second ? 1 : 0];
  const unsigned blk = second ? blockIdx.x - grp.nblocks0 : blockIdx.x;
  const float* __restrict__ h = pr.h;
  const float* __restrict__ scene_mean = pr.scene_mean;
  const int32_t* __restrict__ src_row = pr.src_row;
  float* __restrict__ out = pr.out;
  _Float16* p16 = pr.p16;
  const size_t p16_stride = pr.p16_stride;
  const int M = pr.M, H = pr.H, W = pr.W, sm_div = pr.sm_div, ngroups = pr.ngroups;
  const int per = (ngroups + 7) >> 3;
  const int g = (blk & 7) * per + (blk >> 3);
  if (g >= ngroups) return;
  const int tid = threadIdx.x;
  const int K = H * W;
  const long long Mtot = (long long)M * K;
  const long long m0 = (long long)g * kGnnCells;
  if (pr.row_ref) {     // beam decode: skip the rows no surviving beam descends from
    const long long mlast = m0 + kGnnCells - 1 < Mtot ? m0 + kGnnCells - 1 : Mtot - 1;
    bool any = false;
    for (long long r = m0 / K; r <= mlast / K; ++r) any |= pr.row_ref[r] != 0;
    if (!any) return;
  }
  const long long s0 = m0 - W - 1 > 0 ? m0 - W - 1 : 0;
  const long long s1 = m0 + kGnnCells + W + 1 < Mtot ? m0 + kGnnCells + W + 1 : Mtot;
  const int NS = (int)(s1 - s0);
  if (tid < NS) {
    const long long m = s0 + tid;
    const int r = (int)(m / K), c = (int)(m - (long long)r * K);
    hsrc[tid] = (src_row ? src_row[r] : r) * K + c;
    ssrc[tid] = (r / sm_div) * K + c;
    ssq[tid] = 0.f;
  }
  // pass-1 role: own cell i1 = tid >> 3, channel slice sl = tid & 7
  const int i1 = tid >> 3, sl = tid & 7;
  // pass-2 role: own cell i2 = tid % cells, channel group c8 = tid / cells
  const int i2 = tid % kGnnCells, c8 = tid / kGnnCells;
  auto cell_info = [&](int i, int& li, int& mask) {
    const long long m = m0 + i;
    li = (int)(m - s0);
    mask = 0;
    if (m < Mtot) {
      const int c = (int)(m % K);
      const int y = c / W, x = c - y * W;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) mask |= 1 << t;
      }
    }
  };
  int li1, mask1, li2, mask2;
  cell_info(i1, li1, mask1);
  cell_info(i2, li2, mask2);

  // A chunk travels global -> registers -> LDS; the loads of chunk q+1 are issued
  // before chunk q is computed on, so their HBM / L2 latency runs under the compute
  // phase.  (First version: load -> wait -> store inside a run-time loop: seven
  // serialised round trips per chunk, 48 us per workgroup.)
  constexpr int kIt = (kGnnStage * 16 + kGnnThreads - 1) / kGnnThreads;     // 7 (32 cells) / 5 (64)
  f32x4_t val[kIt];
  auto stage_load = [&](int q) {     // q < 4: channels q*64.. of h; q == 4: scene_mean
#pragma unroll
    for (int k = 0; k < kIt; ++k) {
      const int v = tid + k * kGnnThreads;
      const int ls = v >> 4, part = v & 15;
      val[k] = f32x4_t{0.f, 0.f, 0.f, 0.f};
      if (v < NS * 16) {
        if (q < 4) {
          val[k] = *reinterpret_cast<const f32x4_t*>(h + (size_t)hsrc[ls] * C + q * 64 + part * 4);
        } else if (D == 64) {
          val[k] = *reinterpret_cast<const f32x4_t*>(scene_mean + (size_t)ssrc[ls] * 64 + part * 4);
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j)
            val[k][j] = (part * 4 + j < D) ? scene_mean[(size_t)ssrc[ls] * D + part * 4 + j] : 0.f;
        }
      }
    }
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int k = 0; k < kIt; ++k) {
      const int v = tid + k * kGnnThreads;
      if (v < NS * 16)
        *reinterpret_cast<f32x4_t*>(&buf[(v >> 4) * kGnnPitch + (v & 15) * 4]) = val[k];
    }
  };
  auto dot8 = [&](const float* a, const float* b) {
    const f32x4_t a0 = *reinterpret_cast<const f32x4_t*>(a), a1 = *reinterpret_cast<const f32x4_t*>(a + 4);
    const f32x4_t b0 = *reinterpret_cast<const f32x4_t*>(b), b1 = *reinterpret_cast<const f32x4_t*>(b + 4);
    float d = a0[0] * b0[0];
    d = fmaf(a0[1], b0[1], d); d = fmaf(a0[2], b0[2], d); d = fmaf(a0[3], b0[3], d);
    d = fmaf(a1[0], b1[0], d); d = fmaf(a1[1], b1[1], d); d = fmaf(a1[2], b1[2], d);
    d = fmaf(a1[3], b1[3], d);
    return d;
  };

  float pd[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) pd[t] = 0.f;
  __syncthreads();                    // hsrc / ssrc / ssq visible
  const int nchunk = D > 0 ? 5 : 4;
  stage_load(0);
  for (int q = 0; q < nchunk; ++q) {
    stage_store();
    __syncthreads();
    stage_load(q + 1 < nchunk ? q + 1 : 0);   // next chunk; after the last: pass 2's first
    if (mask1) {
      const float* fi = &buf[li1 * kGnnPitch + sl * 8];
#pragma unroll
      for (int t = 0; t < 9; ++t)
        if ((mask1 >> t) & 1)
          pd[t] += dot8(fi, &buf[(li1 + (t / 3 - 1) * W + (t % 3 - 1)) * kGnnPitch + sl * 8]);
    }
    for (int base = 0; base < NS * 8; base += kGnnThreads) {     // |u|^2 of every staged cell
      const int idx = base + tid;
      const int ls = idx >> 3;
      float pp = 0.f;
      if (idx < NS * 8) {
        const float* f = &buf[ls * kGnnPitch + sl * 8];
        pp = dot8(f, f);
      }
      pp += __shfl_xor(pp, 1, 64);
      pp += __shfl_xor(pp, 2, 64);
      pp += __shfl_xor(pp, 4, 64);
      if (idx < NS * 8 && sl == 0) ssq[ls] += pp;        // one writer per staged cell
    }
    __syncthreads();
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    float v = pd[t];
    v += __shfl_xor(v, 1, 64);
    v += __shfl_xor(v, 2, 64);
    v += __shfl_xor(v, 4, 64);
    if (sl == 0) edot[i1 * 9 + t] = v;
  }
  __syncthreads();
  if (tid < kGnnCells) {              // softmax over the in-image neighbours of cell tid
    int li, mask;
    cell_info(tid, li, mask);
    float e[9];
    float emax = -INFINITY;
    const float invi = mask ? rsqrtf(fmaxf(ssq[li], 1e-12f)) : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      e[t] = 0.f;
      if ((mask >> t) & 1) {
        const int lj = li + (t / 3 - 1) * W + (t % 3 - 1);
        e[t] = edot[tid * 9 + t] * invi * rsqrtf(fmaxf(ssq[lj], 1e-12f));
        emax = fmaxf(emax, e[t]);
      }
    }
    float den = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      e[t] = ((mask >> t) & 1) ? expf(e[t] - emax) : 0.f;
      den += e[t];
    }
    const float inv = 1.0f / den;
#pragma unroll
    for (int t = 0; t < 9; ++t) alpha[tid * 9 + t] = e[t] * inv;
  }
  __syncthreads();
  const long long m2 = m0 + i2;
  for (int q = 0; q < 4; ++q) {
    stage_store();
    __syncthreads();
    if (q < 3) stage_load(q + 1);
    if (mask2) {
      const float* hi = &buf[li2 * kGnnPitch + c8 * 8];
      float node[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) node[j] = 0.f;
#pragma unroll
      for (int t = 0; t < 9; ++t)
        if ((mask2 >> t) & 1) {
          const float a = alpha[i2 * 9 + t];
          const float* hj = &buf[(li2 + (t / 3 - 1) * W + (t % 3 - 1)) * kGnnPitch + c8 * 8];
          const f32x4_t h0 = *reinterpret_cast<const f32x4_t*>(hj);
          const f32x4_t h1 = *reinterpret_cast<const f32x4_t*>(hj + 4);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            node[j] = fmaf(a, h0[j], node[j]);
            node[4 + j] = fmaf(a, h1[j], node[4 + j]);
          }
        }
      const f32x4_t s0v = *reinterpret_cast<const f32x4_t*>(hi);
      const f32x4_t s1v = *reinterpret_cast<const f32x4_t*>(hi + 4);
      float o[8];
#pragma unroll
      for (int j = 0; j < 4; ++j) { o[j] = s0v[j] + node[j]; o[4 + j] = s1v[j] + node[4 + j]; }
      const int ch = q * 64 + c8 * 8;
      if (out) {
        f32x4_t* dst = reinterpret_cast<f32x4_t*>(out + (size_t)m2 * C + ch);
        dst[0] = f32x4_t{o[0], o[1], o[2], o[3]};
        dst[1] = f32x4_t{o[4], o[5], o[6], o[7]};
      }
      if (p16) {
        typedef _Float16 h8 __attribute__((ext_vector_type(8)));
        h8 pa, pb;
        const size_t idx = plane_index(m2, ch, C);
        if (p16_stride == 0) {
#pragma unroll
          for (int j = 0; j < 8; ++j) pa[j] = bf16_half_bits(o[j]);
          *reinterpret_cast<h8*>(p16 + idx) = pa;
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float sc = o[j] * 256.0f;
            const _Float16 h0h = (_Float16)sc;
            pa[j] = h0h;
            pb[j] = (_Float16)(sc - (float)h0h);
          }
          *reinterpret_cast<h8*>(p16 + idx) = pa;
          *reinterpret_cast<h8*>(p16 + p16_stride + idx) = pb;
        }
      }
    }
    __syncthreads();
  }
}
static inline unsigned gnn_v2_blocks(size_t cells, int* ngroups) {
  const size_t g = (cells + kGnnCells - 1) / kGnnCells;
  *ngroups = (int)g;
  return (unsigned)(((g + 7) / 8) * 8);
}

// ---- graph attention, third version: the same two-pass scheme, register-blocked, with the
// staging done by the LDS-DMA.
// The second version is bound by its LDS reads, not by memory: every thread owns ONE cell
// and reads ten staged vectors (itself + 9 neighbours) for 72 FMAs -- 26.9 KB of LDS reads
// per cell over the two passes plus 7 KB of staging writes -- and a chunk travels global ->
// registers -> LDS, 28 registers per thread that live across the whole compute phase.
// Here a workgroup of 256 threads owns 64 consecutive cells of the flat index (<= 128
// staged) and a thread owns several ADJACENT cells, so a row of staged vectors is shared:
//   pass 1  thread = (4 cells, 4 channels of the 64-channel chunk): 18 16-byte LDS reads
//           feed the 36 dot products of the quad (v2: 40 reads of twice the channels); the
//           16 channel slices of a cell are 16 consecutive lanes, folded once, after the
//           last chunk, with four DPP adds per value.  |u|^2 of an owned cell is its self
//           dot product; the <= 64 halo cells get theirs from four more reads per thread,
//           computed exactly the way their owner computes it (same chain, same fold);
//   pass 2  thread = (2 cells, 8 channels): 24 reads for 144 FMAs, 16-byte plane stores.
// 13 KB of LDS reads per cell and no staging writes by the waves: chunk q + 1 is copied
// global -> LDS by the DMA (16 bytes per lane, lane-linear destination) into the second
// of two tiles while the waves compute on chunk q; one barrier per chunk.
// LDS tile: 128 cells x 256 bytes, no pad.  Pass 1: part p of staged cell l at l*256 + p*16:
// its readers (16 consecutive lanes = the 16 parts of one cell) are conflict-free and every
// read is one of three row bases + an immediate; a window may reach one cell past its tile
// -- the other tile, or the small arrays behind them -- where it reads garbage that only
// ever feeds dot products of out-of-image neighbours, which nothing uses.  Pass 2: the lane that fills slot p of cell l fetches part
// p ^ ((l >> 1) & 15) instead, which makes the pass-2 pattern (16 consecutive lanes = 16
// consecutive cell pairs, same part) hit 16 different bank groups; reads are clamped into
// the tile, every slot of which holds a real cell (indices are clamped into the tensor): an
// out-of-image neighbour is a finite value times a weight of exactly 0.
// Arithmetic per cell depends only on the cell's data, not on where the cell sits in its
// workgroup (the engine's A/B tests compare layouts bit for bit).
constexpr int kGnn3Cells = 64;
constexpr int kGnn3Threads = 256;
constexpr int kGnn3Stage = kGnn3Cells + 2 * 32;       // halo: W (+ 1 unless W divides 64) per side
constexpr int kGnn3It = kGnn3Stage * 16 / kGnn3Threads;   // 8 DMA pieces per thread and chunk
constexpr int kGnn3Tile = kGnn3Stage * 64;            // floats per chunk tile
constexpr int kGnn3Small = 1024;                      // floats behind the tiles: e_ij / weights, |u|^2, row maps
static_assert(kGnn3Stage * 16 % kGnn3Threads == 0, "the DMA pieces cover the tile exactly");

template <int CTRL>
__device__ __forceinline__ float dpp_lane_f(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
// Sum over the 16 lanes of a DPP row; every lane of the row ends with the total.  All 16
// lanes must be active.
__device__ __forceinline__ float row16_sum(float v) {
  v += dpp_lane_f<0xB1>(v);     // quad_perm [1,0,3,2]
  v += dpp_lane_f<0x4E>(v);     // quad_perm [2,3,0,1]
  v += dpp_lane_f<0x141>(v);    // row_half_mirror: lane i <- lane 7 - i of its half row
  v += dpp_lane_f<0x140>(v);    // row_mirror:      lane i <- lane 15 - i
  return v;
}
__device__ __forceinline__ float dot4_acc(const f32x4_t a, const f32x4_t b, float acc) {
  acc = fmaf(a[0], b[0], acc); acc = fmaf(a[1], b[1], acc);
  acc = fmaf(a[2], b[2], acc); acc = fmaf(a[3], b[3], acc);
  return acc;
}

__global__ __launch_bounds__(kGnn3Threads, 2)
void gnn_attend_v3_kernel(const GnnGroup grp, int C, int D) {
  typedef __attribute__((address_space(3))) void lds_void;
  // The two tiles come first: every LDS-DMA destination then lies below 64 KB.
  __shared__ __attribute__((aligned(16))) float smem[2 * kGnn3Tile + kGnn3Small];
  float* const buf0 = smem;                                 // tiles: buf0, buf0 + kGnn3Tile
  float* const ea = smem + 2 * kGnn3Tile;                   // [64][9] e_ij, then the softmax weights
  float* const ssq = ea + kGnn3Cells * 9;                   // [128]
  int* const hsrc = reinterpret_cast<int*>(ssq + kGnn3Stage);      // [128]
  int* const ssrc = hsrc + kGnn3Stage;                             // [128]
  static_assert(kGnn3Cells * 9 + 3 * kGnn3Stage <= kGnn3Small, "small arrays");
  const bool second = blockIdx.x >= grp.nblocks0;
  const GnnProblem& pr = grp.p[second ? 1 : 0];
  const unsigned blk = second ? blockIdx.x - grp.nblocks0 : blockIdx.x;
  const float* __restrict__ h = pr.h;
  const float* __restrict__ scene_mean = pr.scene_mean;
  const int32_t* __restrict__ src_row = pr.src_row;
  float* __restrict__ out = pr.out;
  _Float16* p16 = pr.p16;
  const size_t p16_stride = pr.p16_stride;
  const int M = pr.M, H = pr.H, W = pr.W, sm_div = pr.sm_div, ngroups = pr.ngroups;
  const int per = (ngroups + 7) >> 3;
  const int g = (blk & 7) * per + (blk >> 3);
  if (g >= ngroups) return;
  const int tid = threadIdx.x;
  const int K = H * W;
  const long long Mtot = (long long)M * K;
  const long long m0 = (long long)g * kGnn3Cells;
  if (pr.row_ref) {     // beam decode: skip the rows no surviving beam descends from
    const long long mlast = m0 + kGnn3Cells - 1 < Mtot ? m0 + kGnn3Cells - 1 : Mtot - 1;
    bool any = false;
    for (long long r = m0 / K; r <= mlast / K; ++r) any |= pr.row_ref[r] != 0;
    if (!any) return;
  }
  // When W divides 64 the group starts at x == 0 and ends at x == W - 1: the corner
  // neighbours m0 - W - 1 and m0 + 64 + W are out of the image and need no staging.
  // Tile slot l holds cell s0 + l, clamped into the tensor (the first and the last group:
  // cells that do not exist are nobody's in-image neighbour); the owned cells are slots
  // own0 .. own0 + 63.
  const int own0 = kGnn3Cells % W == 0 ? W : W + 1;
  const long long s0 = m0 - own0;
  if (tid < kGnn3Stage) {
    long long m = s0 + tid;
    m = m < 0 ? 0 : (m > Mtot - 1 ? Mtot - 1 : m);
    const int r = (int)(m / K), c = (int)(m - (long long)r * K);
    hsrc[tid] = (src_row ? src_row[r] : r) * K + c;
    ssrc[tid] = (r / sm_div) * K + c;
  }
  __syncthreads();

  // DMA role: slot (tid & 15) of staged cells (tid >> 4) + 16 k.  A wave's 64 lanes fill 64
  // consecutive 16-byte slots of the tile.
  // Buffer addressing: one uniform descriptor per tensor + a 32-bit lane offset (rows of h
  // and of the scene means lie below 4 GB: host-checked).
  const int part = tid & 15, ls0 = tid >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const __amdgpu_buffer_rsrc_t h_rs = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(const_cast<float*>(h)), 0, 0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t sm_rs = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(const_cast<float*>(scene_mean)), 0, 0xffffffffu, 0x00020000);
  uint32_t goff[kGnn3It];                     // byte offset of the cell's row in h
#pragma unroll
  for (int k = 0; k < kGnn3It; ++k) {
    goff[k] = (uint32_t)hsrc[ls0 + 16 * k] * (uint32_t)(C * 4);
  }
  // (ls >> 1) & 15 of cell ls0 + 16 k is (ls0 >> 1) + 8 (k & 1)
  const uint32_t p_plain = part * 16;
  const uint32_t p_swz0 = (part ^ (ls0 >> 1)) * 16, p_swz1 = (part ^ ((ls0 >> 1) + 8)) * 16;
  auto dma_h = [&](int q, float* tile, bool swizzled) {     // channels q*64 .. q*64+63 of h
#pragma unroll
    for (int k = 0; k < kGnn3It; ++k)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          h_rs, (lds_void*)(tile + (wave * 64 + k * kGnn3Threads) * 4), 16,
          goff[k] + (swizzled ? ((k & 1) ? p_swz1 : p_swz0) : p_plain), q * 256, 0, 0);
  };
  auto dma_scene = [&](float* tile) {         // the 64 scene-mean channels (D is 0 or 64: host-checked)
#pragma unroll
    for (int k = 0; k < kGnn3It; ++k)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          sm_rs, (lds_void*)(tile + (wave * 64 + k * kGnn3Threads) * 4), 16,
          (uint32_t)ssrc[ls0 + 16 * k] * 256u + p_plain, 0, 0, 0);
  };

  // ---- pass 1: thread = (quad of cells 4 cp .. 4 cp + 3, channel slice sl); plain layout
  const int cp = tid >> 4, sl = tid & 15;
  const int lq = own0 + cp * 4;
  float pd[4][9], hq[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    hq[k] = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) pd[k][t] = 0.f;
  }
  const int o0 = (lq - 1) * 64 + sl * 4;      // row dy = 0, window cell j: + j * 64 (floats)
  // The window of row dy = -1 starts at slot own0 - 1 - W + 4 cp >= -1: slot -1 (quad 0 when
  // W divides 64) is the out-of-image corner and is read from slot 0 instead.  Windows end
  // at slot 128 at most: one cell past a tile is the other tile or the small arrays.
  const int om = o0 - W * 64, om0 = om < 0 ? sl * 4 : om;
  // halo cells of this thread: the i-th is staged cell hc, hc = cp + 16 i below the owned
  // range, + 64 above it
  int ho[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int hc = cp + 16 * i;
    ho[i] = (hc < own0 ? hc : hc + kGnn3Cells) * 64 + sl * 4;
  }
  auto dots1 = [&](const float* tile) {
    f32x4_t own[4], r[6];
#pragma unroll
    for (int j = 0; j < 6; ++j) r[j] = *reinterpret_cast<const f32x4_t*>(tile + o0 + j * 64);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      own[k] = r[k + 1];
      pd[k][3] = dot4_acc(own[k], r[k], pd[k][3]);
      pd[k][4] = dot4_acc(own[k], own[k], pd[k][4]);
      pd[k][5] = dot4_acc(own[k], r[k + 2], pd[k][5]);
    }
#pragma unroll
    for (int j = 0; j < 6; ++j)
      r[j] = *reinterpret_cast<const f32x4_t*>(tile + (j == 0 ? om0 : om + j * 64));
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pd[k][0] = dot4_acc(own[k], r[k], pd[k][0]);
      pd[k][1] = dot4_acc(own[k], r[k + 1], pd[k][1]);
      pd[k][2] = dot4_acc(own[k], r[k + 2], pd[k][2]);
    }
#pragma unroll
    for (int j = 0; j < 6; ++j) r[j] = *reinterpret_cast<const f32x4_t*>(tile + o0 + W * 64 + j * 64);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      pd[k][6] = dot4_acc(own[k], r[k], pd[k][6]);
      pd[k][7] = dot4_acc(own[k], r[k + 1], pd[k][7]);
      pd[k][8] = dot4_acc(own[k], r[k + 2], pd[k][8]);
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4_t v = *reinterpret_cast<const f32x4_t*>(tile + ho[i]);
      hq[i] = dot4_acc(v, v, hq[i]);
    }
  };
  // Chunk c of the kernel's nchunk + 4 lands in tile c & 1.  The barrier at the top of a
  // step carries the vmcnt(0) of this wave's pending DMA pieces and closes the reads of
  // the tile the next DMA overwrites.
  const int nchunk = D > 0 ? 5 : 4;
  dma_h(0, buf0, false);
#pragma unroll 1
  for (int c = 0; c < nchunk; ++c) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of chunk c
    __syncthreads();
    float* const next = buf0 + ((c + 1) & 1) * kGnn3Tile;
    if (c + 1 < 4) dma_h(c + 1, next, false);
    else if (c + 1 < nchunk) dma_scene(next);
    else dma_h(0, next, true);                // pass 2's first chunk
    dots1(buf0 + (c & 1) * kGnn3Tile);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const float v = row16_sum(pd[k][t]);
      if (sl == 0) {
        ea[(cp * 4 + k) * 9 + t] = v;
        if (t == 4) ssq[lq + k] = v;
      }
    }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float v = row16_sum(hq[i]);
    const int hc = cp + 16 * i;
    const int l = hc < own0 ? hc : hc + kGnn3Cells;
    if (sl == 0) ssq[l] = v;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // pass 2's first chunk
  __syncthreads();

  auto cell_mask = [&](long long m) {
    int mask = 0;
    if (m < Mtot) {
      const int c = (int)(m % K);
      const int y = c / W, x = c - y * W;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) mask |= 1 << t;
      }
    }
    return mask;
  };
  if (tid < kGnn3Cells) {              // softmax over the in-image neighbours of cell tid
    const int mask = cell_mask(m0 + tid);
    const int li = own0 + tid;
    float e[9];
    float emax = -INFINITY;
    const float invi = mask ? rsqrtf(fmaxf(ssq[li], 1e-12f)) : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      e[t] = 0.f;
      if ((mask >> t) & 1) {
        const int lj = li + (t / 3 - 1) * W + (t % 3 - 1);
        e[t] = ea[tid * 9 + t] * invi * rsqrtf(fmaxf(ssq[lj], 1e-12f));
        emax = fmaxf(emax, e[t]);
      }
    }
    float den = 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      e[t] = ((mask >> t) & 1) ? expf(e[t] - emax) : 0.f;
      den += e[t];
    }
    const float inv = mask ? 1.0f / den : 0.f;
#pragma unroll
    for (int t = 0; t < 9; ++t) ea[tid * 9 + t] = e[t] * inv;
  }
  __syncthreads();                     // weights visible; pass 2's first chunk has landed

  // ---- pass 2: thread = (pair of cells 2 pi, 2 pi + 1, channels c8*8 .. +7 of the chunk);
  // swizzled layout, reads clamped into the tile
  const int pi = tid & 31, c8 = tid >> 5;
  const int lp = own0 + pi * 2;
  const long long m2 = m0 + pi * 2;
  // h_i + sum_t a_t h_t = sum_t (a_t + [t == 4]) h_t: the residual rides on the self weight
  float al[2][9];
#pragma unroll
  for (int k = 0; k < 2; ++k)
#pragma unroll
    for (int t = 0; t < 9; ++t) al[k][t] = ea[(pi * 2 + k) * 9 + t] + (t == 4 ? 1.0f : 0.f);
  // The stores of chunk q are issued at the top of step q + 1, in front of that step's DMA:
  // the vmcnt(0) in front of a barrier then only waits for requests that are a whole compute
  // phase old (vmcnt counts loads and stores alike).
  auto store_pair = [&](const float (&o2)[2][8], int ch) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (m2 + k < Mtot) {
        const float* o = o2[k];
        if (out) {
          f32x4_t* dst = reinterpret_cast<f32x4_t*>(out + (size_t)(m2 + k) * C + ch);
          dst[0] = f32x4_t{o[0], o[1], o[2], o[3]};
          dst[1] = f32x4_t{o[4], o[5], o[6], o[7]};
        }
        if (p16) {
          typedef _Float16 h8 __attribute__((ext_vector_type(8)));
          h8 pa, pb;
          const size_t idx = plane_index(m2 + k, ch, C);
          if (p16_stride == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) pa[j] = bf16_half_bits(o[j]);
            *reinterpret_cast<h8*>(p16 + idx) = pa;
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
              const float sc = o[j] * 256.0f;
              const _Float16 h0h = (_Float16)sc;
              pa[j] = h0h;
              pb[j] = (_Float16)(sc - (float)h0h);
            }
            *reinterpret_cast<h8*>(p16 + idx) = pa;
            *reinterpret_cast<h8*>(p16 + p16_stride + idx) = pb;
          }
        }
      }
    }
  };
  float node[2][8];
#pragma unroll 1
  for (int q = 0; q < 4; ++q) {
    const int c = nchunk + q;
    if (q > 0) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wave's pieces of chunk c
      __syncthreads();
      store_pair(node, (q - 1) * 64 + c8 * 8);
    }
    if (q < 3) dma_h(q + 1, buf0 + ((c + 1) & 1) * kGnn3Tile, true);
    const float* const tile = buf0 + (c & 1) * kGnn3Tile;
    auto staged = [&](int l, int p) -> f32x4_t {
      l = l < 0 ? 0 : (l > kGnn3Stage - 1 ? kGnn3Stage - 1 : l);
      return *reinterpret_cast<const f32x4_t*>(&tile[l * 64 + ((p ^ ((l >> 1) & 15)) << 2)]);
    };
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) node[k][j] = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy) {
        f32x4_t r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) r[j] = staged(lp - 1 + j + dy * W, c8 * 2 + half);
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
          for (int dx = -1; dx <= 1; ++dx) {
            const float a = al[k][(dy + 1) * 3 + dx + 1];
#pragma unroll
            for (int j = 0; j < 4; ++j)
              node[k][half * 4 + j] = fmaf(a, r[k + 1 + dx][j], node[k][half * 4 + j]);
          }
      }
    }
  }
  store_pair(node, 3 * 64 + c8 * 8);
}
static inline unsigned gnn_v3_blocks(size_t cells, int* ngroups) {
  const size_t g = (cells + kGnn3Cells - 1) / kGnn3Cells;
  *ngroups = (int)g;
  return (unsigned)(((g + 7) / 8) * 8);
}

// ---------------------------------------------------------------- hidden2grid
// conv3x3 SAME, no bias, identity, C -> P (P in {1,2}); hidden2grid,
// code/pred_models.py:925-959.  One wave per cell, lane l holds channels
// 4l..4l+3.  out has row stride out_row_stride elements so step t of a
// [N, T, H, W, P] output tensor is written in place.
template <int P>
__global__ __launch_bounds__(256)
void hidden2grid_kernel(const float* __restrict__ h, const float* __restrict__ w,
                        float* __restrict__ out, size_t out_row_stride, int M,
                        int H, int W, int C) {
  const int lane = threadIdx.x & 63;
  const size_t cell_id = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int K = H * W;
  if (cell_id >= (size_t)M * K) return;
  const int m = cell_id / K;
  const int cell = cell_id - (size_t)m * K;
  const int y = cell / W, x = cell - y * W;
  const float* hrow = h + (size_t)m * K * C;
  float acc[P];
#pragma unroll
  for (int p = 0; p < P; ++p) acc[p] = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {
      for (int c0 = lane * 4; c0 < C; c0 += 256) {     // channel groups of 256 (any C % 4 == 0)
        const f32x4_t hv = *reinterpret_cast<const f32x4_t*>(
            hrow + (size_t)(yy * W + xx) * C + c0);
        const float* wp = w + ((size_t)t * C + c0) * P;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int p = 0; p < P; ++p) acc[p] = fmaf(hv[j], wp[j * P + p], acc[p]);
      }
    }
  }
#pragma unroll
  for (int p = 0; p < P; ++p) acc[p] = wave_sum(acc[p]);
  if (lane == 0) {
#pragma unroll
    for (int p = 0; p < P; ++p)
      out[(size_t)m * out_row_stride + (size_t)cell * P + p] = acc[p];
  }
}

// argmax over K logits per row, lowest index wins ties (tf.argmax,
// code/pred_models.py:411-415).  One wave per row.
__global__ __launch_bounds__(64)
void argmax_rows_kernel(const float* __restrict__ logits, size_t row_stride,
                        int32_t* __restrict__ ids, int M, int K) {
  const int m = blockIdx.x;
  const int lane = threadIdx.x;
  const float* p = logits + (size_t)m * row_stride;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int k = lane; k < K; k += 64) {
    const float v = p[k];
    if (v > best || (v == best && k < bi)) { best = v; bi = k; }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(best, off, 64);
    const int oi = __shfl_xor(bi, off, 64);
    if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
  }
  if (lane == 0) ids[m] = bi;
}

// ---------------------------------------------------------------- beam step
// One workgroup per sample.  (code/pred_models.py:557-591, add_div_penalty
// :1197-1223.)  logits [N,B,K]:
//   lp = log_softmax(logits) + prev_lp[b]
//   if diverse: lp += log(gamma) * rank_desc(lp within (n,b)), ties -> lower idx
//   candidates = time > 1 ? lp[N, B*K] : lp[:, 0]
//   (new_lp, idx) = top_k(candidates, B) descending, ties -> lower index
//   new_lp = 0 unless time > fix_num_timestep; ids = idx % K; parents = idx / K
// The rank of element v is the number of elements that precede it in the
// stable descending order: #{u : lp[u] > lp[v]} + #{u < v : lp[u] == lp[v]}.
// Dynamic LDS: lp [B*K] + 512 words of reduction scratch + pen [B*K].
__global__ __launch_bounds__(512)
void beam_step_kernel(const float* __restrict__ logits,
                      const float* __restrict__ prev_lp, int B, int K, int time,
                      int diverse, float log_gamma, int fix_num_timestep,
                      float* __restrict__ new_lp, int32_t* __restrict__ ids,
                      int32_t* __restrict__ parents,
                      int32_t* __restrict__ state_src_row, int state_rows_per_sample,
                      int32_t* __restrict__ row_ref = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* lp = sm;                 // [B*K]
  float* red = sm + B * K;        // [256] reduction scratch
  int* redi = reinterpret_cast<int*>(red + 256);  // [256]
  const int n = blockIdx.x;
  const int tid = threadIdx.x;
  const int nthr = blockDim.x;
  const int lane = tid & 63, wave = tid >> 6, nwave = nthr >> 6;
  const float* lg = logits + (size_t)n * B * K;

  // log_softmax per beam: one wave per beam row (strided over beams)
  for (int b = wave; b < B; b += nwave) {
    const float* row = lg + (size_t)b * K;
    float mx = -INFINITY;
    for (int k = lane; k < K; k += 64) mx = fmaxf(mx, row[k]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    float s = 0.f;
    for (int k = lane; k < K; k += 64) s += expf(row[k] - mx);
    s = wave_sum(s);
    const float lse = logf(s);
    const float pl = prev_lp[(size_t)n * B + b];
    for (int k = lane; k < K; k += 64) lp[b * K + k] = pl + ((row[k] - mx) - lse);
  }
  __syncthreads();
  if (diverse) {
    // rank by counting over the un-penalised values; penalties go to a second
    // LDS plane and are applied once every rank has been taken.
    float* pen = sm + B * K + 512;
    const int total = B * K;
    for (int v = tid; v < total; v += nthr) {
      const int b = v / K, kv = v - b * K;
      const float val = lp[v];
      const float* row = lp + b * K;
      int rank = 0;
      for (int u = 0; u < K; ++u) {
        const float o = row[u];
        rank += (o > val) || (o == val && u < kv);
      }
      pen[v] = log_gamma * (float)rank;
    }
    __syncthreads();
    for (int v = tid; v < total; v += nthr) lp[v] = lp[v] + pen[v];
    __syncthreads();
  }
  // top-B selection by repeated arg-max with (value desc, index asc) order.
  const int ncand = (time > 1) ? B * K : K;
  for (int sel = 0; sel < B; ++sel) {
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int v = tid; v < ncand; v += nthr) {
      const float val = lp[v];
      if (val > best || (val == best && v < bi)) { best = val; bi = v; }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(best, off, 64);
      const int oi = __shfl_xor(bi, off, 64);
      if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0) { red[wave] = best; redi[wave] = bi; }
    __syncthreads();
    if (tid == 0) {
      for (int wv = 1; wv < nwave; ++wv) {
        const float ov = red[wv];
        const int oi = redi[wv];
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
      }
      const int par = bi / K;
      new_lp[(size_t)n * B + sel] = (time > fix_num_timestep) ? best : 0.f;
      ids[(size_t)n * B + sel] = bi - par * K;
      parents[(size_t)n * B + sel] = par;
      if (state_src_row) state_src_row[(size_t)n * B + sel] = n * state_rows_per_sample + par;
      if (row_ref) row_ref[n * state_rows_per_sample + par] = 1;   // a state row some beam continues
      lp[bi] = -INFINITY;  // remove from the candidate set
    }
    __syncthreads();
  }
}

// The same step in two launches, so that the O(K^2) rank count of the diversity penalty
// spreads over the whole chip instead of one CU per sample (447 us -> ~60 us per step at
// N=128, B=20, K=576):
//   beam_rank_kernel    one WAVE per (n, b) row, the row held in registers (lane l owns
//                       k = l + 64 j); element u is broadcast with v_readlane and compared
//                       against the 64 * J owned values; writes the candidates lp [R, K].
//                       Same arithmetic, in the same order, as beam_step_kernel.
//   beam_select_kernel  one workgroup per sample: candidates in LDS, every thread caches
//                       the best of the ones it owns, B rounds of (wave reduce -> 16-entry
//                       LDS table -> every thread re-derives the winner); only the winner's
//                       owner rescans.  One barrier per round.
constexpr int kBeamRankJ = 16;   // K <= 1024

template <int J>
__global__ __launch_bounds__(256)
void beam_rank_kernel(const float* __restrict__ logits, const float* __restrict__ prev_lp,
                      int R, int B, int K, int time, int diverse, float log_gamma,
                      float* __restrict__ cand) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  if (time <= 1 && (r % B) != 0) return;   // only beam 0 is a candidate at the first step
  const float* row = logits + (size_t)r * K;
  float v[J];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < J; ++j) {
    const int k = lane + 64 * j;
    v[j] = k < K ? row[k] : -INFINITY;
    mx = fmaxf(mx, v[j]);
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < J; ++j)
    if (lane + 64 * j < K) s += expf(v[j] - mx);
  s = wave_sum(s);
  const float lse = logf(s);
  const float pl = prev_lp[r];
#pragma unroll
  for (int j = 0; j < J; ++j) v[j] = pl + ((v[j] - mx) - lse);
  if (diverse) {
    int rank[J];
#pragma unroll
    for (int j = 0; j < J; ++j) rank[j] = 0;
#pragma unroll
    for (int jj = 0; jj < J; ++jj) {
      const int lim = min(64, K - 64 * jj);      // uniform
      for (int l = 0; l < lim; ++l) {
        const float o = __builtin_bit_cast(float, __builtin_amdgcn_readlane(
                                                      __builtin_bit_cast(int, v[jj]), l));
#pragma unroll
        for (int j = 0; j < J; ++j) {
          // u = l + 64 jj precedes kv = lane + 64 j  <=>  jj < j, or jj == j and l < lane
          const bool before = jj < j || (jj == j && l < lane);
          rank[j] += (o > v[j] || (before && o == v[j])) ? 1 : 0;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < J; ++j) v[j] = v[j] + log_gamma * (float)rank[j];
  }
  float* out = cand + (size_t)r * K;
#pragma unroll
  for (int j = 0; j < J; ++j)
    if (lane + 64 * j < K) out[lane + 64 * j] = v[j];
}

__device__ __forceinline__ bool beam_better(float v, int i, float bv, int bi) {
  return v > bv || (v == bv && i < bi);
}

// Dynamic LDS: lp [B*K] + 2 x 16 (value, index) reduction entries.
__global__ __launch_bounds__(1024)
void beam_select_kernel(const float* __restrict__ cand, int B, int K, int time,
                        int fix_num_timestep, float* __restrict__ new_lp,
                        int32_t* __restrict__ ids, int32_t* __restrict__ parents,
                        int32_t* __restrict__ state_src_row, int state_rows_per_sample,
                        int32_t* __restrict__ row_ref = nullptr) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ncand = (time > 1) ? B * K : K;
  float* lp = sm;
  float* redv = sm + B * K;                              // [2][16]
  int* redi = reinterpret_cast<int*>(redv + 32);         // [2][16]
  const float* src = cand + (size_t)n * B * K;
  float best = -INFINITY;
  int bi = 0x7fffffff;
  for (int v = tid; v < ncand; v += 1024) {
    const float x = src[v];
    lp[v] = x;
    if (beam_better(x, v, best, bi)) { best = x; bi = v; }
  }
  for (int sel = 0; sel < B; ++sel) {
    float wv = best;
    int wi = bi;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float ov = __shfl_xor(wv, off, 64);
      const int oi = __shfl_xor(wi, off, 64);
      if (beam_better(ov, oi, wv, wi)) { wv = ov; wi = oi; }
    }
    const int pp = (sel & 1) * 16;
    if (lane == 0) { redv[pp + wave] = wv; redi[pp + wave] = wi; }
    __syncthreads();
    wv = redv[pp];
    wi = redi[pp];
#pragma unroll
    for (int w = 1; w < 16; ++w) {
      const float ov = redv[pp + w];
      const int oi = redi[pp + w];
      if (beam_better(ov, oi, wv, wi)) { wv = ov; wi = oi; }
    }
    if (tid == 0) {
      const int par = wi / K;
      new_lp[(size_t)n * B + sel] = (time > fix_num_timestep) ? wv : 0.f;
      ids[(size_t)n * B + sel] = wi - par * K;
      parents[(size_t)n * B + sel] = par;
      if (state_src_row) state_src_row[(size_t)n * B + sel] = n * state_rows_per_sample + par;
      if (row_ref) row_ref[n * state_rows_per_sample + par] = 1;   // a state row some beam continues
    }
    if ((wi & 1023) == tid) {          // the owner removes the winner and rescans its own
      lp[wi] = -INFINITY;
      best = -INFINITY;
      bi = 0x7fffffff;
      for (int v = tid; v < ncand; v += 1024) {
        const float x = lp[v];
        if (beam_better(x, v, best, bi)) { best = x; bi = v; }
      }
    }
  }
}

// ------------------------------------------------------------ batch assembly
// Dense regression maps from one (x, y) per row-step: out[r, cell, :] =
// (float)(xy[r, :] - centre[cell, :]) in double, the rounding of
// code/preprocess.py:463-475; rows of samples n >= num_rows are zero.
__global__ void regress_from_xy_kernel(const double* __restrict__ xy,
                                       const double* __restrict__ centers,
                                       float* __restrict__ out, int rows, int T, int K,
                                       int num_rows) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // (row-step, cell)
  if (idx >= (size_t)rows * K) return;
  const int r = (int)(idx / K), cell = (int)(idx - (size_t)r * K);
  float2 v = {0.f, 0.f};
  if (r / T < num_rows) {
    v.x = (float)(xy[2 * (size_t)r] - centers[2 * cell]);
    v.y = (float)(xy[2 * (size_t)r + 1] - centers[2 * cell + 1]);
  }
  reinterpret_cast<float2*>(out)[idx] = v;
}

__global__ void u8_to_f32_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                 size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = (float)in[i];
}

}  // namespace mv
