// Backward / loss / optimizer kernels of the training step
// (reference: Model.build_loss code/pred_models.py:961-1040, Trainer
// :1636-1742 = tf.gradients + clip_by_value + AdadeltaOptimizer).
// Everything except the two gate-conv GEMMs (convlstm_mfma.h dgrad,
// convlstm_wgrad.h) is HBM-bound elementwise / stencil / reduction work.
// All reductions are two-stage with a fixed order: gradients are bitwise
// reproducible run to run.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>

#include "kernels_misc.h"
#include "convlstm_f16x3.h"

namespace mv {

// ------------------------------------------------------------ LSTM pointwise
// Backward of  c' = sf*c + si*tj ; h' = tanh(c')*so  given the saved
// activations gates = [si | tj | sf | so] ([M,4,C]), c (previous), c' and the
// incoming dh', dc'.  Writes the pre-activation gate gradients G [M,4C] in the
// reference's column order (i|j|f|o) IN PLACE over `gates`, and dc (w.r.t. the
// previous cell state) in place over dc_io.
__global__ void lstm_gate_bwd_kernel(float* __restrict__ gates,
                                     const float* __restrict__ c_prev,
                                     const float* __restrict__ c_new,
                                     const float* __restrict__ dh,
                                     float* __restrict__ dc_io, size_t total, int C,
                                     int32_t* __restrict__ gmax_bits = nullptr) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;       // total is a multiple of the block size (C = 256)
  const size_t m = idx / C;
  const int ch = (int)(idx - m * C);
  float* gp = gates + m * 4 * (size_t)C + ch;
  const float si = gp[0], tj = gp[C], sf = gp[2 * C], so = gp[3 * C];
  const float tc = tanhf(c_new[idx]);
  const float dhv = dh[idx];
  const float dcv = dc_io[idx] + dhv * so * (1.f - tc * tc);
  const float cp = c_prev ? c_prev[idx] : 0.f;
  const float gi = dcv * tj * (si * (1.f - si));
  const float gj = dcv * si * (1.f - tj * tj);
  const float gf = dcv * cp * (sf * (1.f - sf));
  const float go = dhv * tc * (so * (1.f - so));
  gp[0] = gi; gp[C] = gj; gp[2 * C] = gf; gp[3 * C] = go;
  dc_io[idx] = dcv * sf;
  if (gmax_bits) {   // max |G| of the tensor (f16x3 dgrad scale); max is order-independent
    float m = fmaxf(fmaxf(fabsf(gi), fabsf(gj)), fmaxf(fabsf(gf), fabsf(go)));
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    // one atomic per WORKGROUP, spread over 64 addresses per tensor (block id mod
    // 64): one atomic per wave on one hot address serialised 73 k atomics per
    // launch (measured: 2.3 -> 41 ms per step; 64 addresses alone: 17 ms)
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
      m = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
      if (m > 0.f && m < INFINITY)
        atomicMax(gmax_bits + (blockIdx.x & 63), __float_as_int(m));
    }
  }
}

// The same, four channels per thread (16-byte accesses; C % 4 == 0): the scalar form
// above moves 13 floats per element with 4-byte loads and sits at 2.5 TB/s.
__global__ __launch_bounds__(256)
void lstm_gate_bwd4_kernel(float* __restrict__ gates, const float* __restrict__ c_prev,
                           const float* __restrict__ c_new, const float* __restrict__ dh,
                           float* __restrict__ dc_io, size_t total4, int C,
                           int32_t* __restrict__ gmax_bits = nullptr) {
  const size_t i4 = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  float mx = 0.f;
  if (i4 < total4) {
    const int c4 = C >> 2;
    const size_t m = i4 / c4;
    const int ch = (int)(i4 - m * c4) * 4;
    float* gp = gates + m * 4 * (size_t)C + ch;
    const size_t idx = m * (size_t)C + ch;
    const f32x4_t si = *reinterpret_cast<const f32x4_t*>(gp),
                  tj = *reinterpret_cast<const f32x4_t*>(gp + C),
                  sf = *reinterpret_cast<const f32x4_t*>(gp + 2 * C),
                  so = *reinterpret_cast<const f32x4_t*>(gp + 3 * C);
    const f32x4_t cn = *reinterpret_cast<const f32x4_t*>(c_new + idx);
    const f32x4_t dhv = *reinterpret_cast<const f32x4_t*>(dh + idx);
    const f32x4_t dci = *reinterpret_cast<const f32x4_t*>(dc_io + idx);
    f32x4_t cp = {0.f, 0.f, 0.f, 0.f};
    if (c_prev) cp = *reinterpret_cast<const f32x4_t*>(c_prev + idx);
    f32x4_t gi, gj, gf, go, dco;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float tc = tanhf(cn[j]);
      const float dcv = dci[j] + dhv[j] * so[j] * (1.f - tc * tc);
      gi[j] = dcv * tj[j] * (si[j] * (1.f - si[j]));
      gj[j] = dcv * si[j] * (1.f - tj[j] * tj[j]);
      gf[j] = dcv * cp[j] * (sf[j] * (1.f - sf[j]));
      go[j] = dhv[j] * tc * (so[j] * (1.f - so[j]));
      dco[j] = dcv * sf[j];
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(gi[j]), fabsf(gj[j])), fmaxf(fabsf(gf[j]), fabsf(go[j]))));
    }
    *reinterpret_cast<f32x4_t*>(gp) = gi;
    *reinterpret_cast<f32x4_t*>(gp + C) = gj;
    *reinterpret_cast<f32x4_t*>(gp + 2 * C) = gf;
    *reinterpret_cast<f32x4_t*>(gp + 3 * C) = go;
    *reinterpret_cast<f32x4_t*>(dc_io + idx) = dco;
  }
  if (gmax_bits) {   // as above: one atomic per workgroup over 64 addresses
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    __shared__ float wm[4];
    if ((threadIdx.x & 63) == 0) wm[threadIdx.x >> 6] = mx;
    __syncthreads();
    if (threadIdx.x == 0) {
      mx = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
      if (mx > 0.f && mx < INFINITY)
        atomicMax(gmax_bits + (blockIdx.x & 63), __float_as_int(mx));
    }
  }
}

// The same for compute mode 2 with the bf16 backward: G leaves as fp32 (the wgrad's operand) AND
// as ONE bf16 plane in the dgrad's tiled operand layout (plane_layout.h) -- the bits
// split_plane_bf16_kernel would produce from the fp32 G in a pass of its own (2.7 ms per training
// step at 64 / GPU).  The plane wants 32 consecutive cells of an 8-channel group contiguous (512
// bytes), the state tensors 4C consecutive floats of a cell: a workgroup takes 32 cells x 32
// channels (8 threads x 16 bytes = 128-byte runs per cell and tensor), computes as above, and
// passes the bf16 values through LDS so that they leave as whole 512-byte runs.  C % 32 == 0.
__global__ __launch_bounds__(256)
void lstm_gate_bwd4_plane_kernel(float* __restrict__ gates, const float* __restrict__ c_prev,
                                 const float* __restrict__ c_new, const float* __restrict__ dh,
                                 float* __restrict__ dc_io, long long cells, int C,
                                 _Float16* __restrict__ plane,
                                 int32_t* __restrict__ gmax_bits) {
  typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
  float mx = 0.f;
  __shared__ __attribute__((aligned(16))) _Float16 tile[4][4][32][8];   // [gate][group of 8][cell][8]
  const int ncb = C >> 5;
  const long long blk = blockIdx.x;
  const long long m0 = (blk / ncb) * 32;
  const int cbase = (int)(blk % ncb) * 32;
  const int tid = threadIdx.x;
  const int cl = tid >> 3, c4 = (tid & 7) * 4;
  const long long m = m0 + cl;
  if (m < cells) {
    const int ch = cbase + c4;
    float* gp = gates + (size_t)m * 4 * (size_t)C + ch;
    const size_t idx = (size_t)m * (size_t)C + ch;
    const f32x4_t si = *reinterpret_cast<const f32x4_t*>(gp),
                  tj = *reinterpret_cast<const f32x4_t*>(gp + C),
                  sf = *reinterpret_cast<const f32x4_t*>(gp + 2 * C),
                  so = *reinterpret_cast<const f32x4_t*>(gp + 3 * C);
    const f32x4_t cn = *reinterpret_cast<const f32x4_t*>(c_new + idx);
    const f32x4_t dhv = *reinterpret_cast<const f32x4_t*>(dh + idx);
    const f32x4_t dci = *reinterpret_cast<const f32x4_t*>(dc_io + idx);
    f32x4_t cp = {0.f, 0.f, 0.f, 0.f};
    if (c_prev) cp = *reinterpret_cast<const f32x4_t*>(c_prev + idx);
    f32x4_t gi, gj, gf, go, dco;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float tc = tanhf(cn[j]);
      const float dcv = dci[j] + dhv[j] * so[j] * (1.f - tc * tc);
      gi[j] = dcv * tj[j] * (si[j] * (1.f - si[j]));
      gj[j] = dcv * si[j] * (1.f - tj[j] * tj[j]);
      gf[j] = dcv * cp[j] * (sf[j] * (1.f - sf[j]));
      go[j] = dhv[j] * tc * (so[j] * (1.f - so[j]));
      dco[j] = dcv * sf[j];
      mx = fmaxf(mx, fmaxf(fmaxf(fabsf(gi[j]), fabsf(gj[j])), fmaxf(fabsf(gf[j]), fabsf(go[j]))));
    }
    *reinterpret_cast<f32x4_t*>(gp) = gi;
    *reinterpret_cast<f32x4_t*>(gp + C) = gj;
    *reinterpret_cast<f32x4_t*>(gp + 2 * C) = gf;
    *reinterpret_cast<f32x4_t*>(gp + 3 * C) = go;
    *reinterpret_cast<f32x4_t*>(dc_io + idx) = dco;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4_t v = g == 0 ? gi : (g == 1 ? gj : (g == 2 ? gf : go));
      f16x4_t b;
#pragma unroll
      for (int j = 0; j < 4; ++j) b[j] = bf16_as_half(v[j]);
      *reinterpret_cast<f16x4_t*>(&tile[g][c4 >> 3][cl][c4 & 7]) = b;
    }
  }
  __shared__ float wm[4];
  if (gmax_bits) {   // max |G| of the step (the wgrad's chain exponent): one atomic per workgroup
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
    if ((tid & 63) == 0) wm[tid >> 6] = mx;
  }
  __syncthreads();
  if (gmax_bits && tid == 0) {
    mx = fmaxf(fmaxf(wm[0], wm[1]), fmaxf(wm[2], wm[3]));
    if (mx > 0.f && mx < INFINITY)
      atomicMax(gmax_bits + (blockIdx.x & 63), __float_as_int(mx));
  }
  // 16 runs (gate, group of 8) of 32 cells x 16 bytes; thread -> (run = tid >> 4, two cells)
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int item = it * 256 + tid;                 // 512 items = 16 runs x 32 cells
    const int run = item >> 5, cell = item & 31;
    const int g = run >> 2, grp = run & 3;
    if (m0 + cell < cells) {
      const f16x8 v = *reinterpret_cast<const f16x8*>(&tile[g][grp][cell][0]);
      *reinterpret_cast<f16x8*>(plane + plane_index(m0 + cell, g * C + cbase + grp * 8, 4 * C)) = v;
    }
  }
}

// ------------------------------------------------------------ graph attention
// Backward of h_out = h + sum_j a_ij h_j, a = softmax_j(f_i . f_j),
// f = l2_normalize([h ; s]) (gnn_edge/gnn_mask_edge/gnn_node,
// code/pred_models.py:808-909), 9-neighbour form.
// Pass A (one wave per cell i): recompute n_i = rsqrt(max(|u_i|^2,1e-12)),
// e_ij, a_ij; da_ij = g_i . h_j; de_ij = a_ij (da_ij - sum_k a_ik da_ik).
// Stores a, de [M*K, 9] (0 for out-of-grid taps) and n [M*K].
template <int NG>
__global__ __launch_bounds__(256)
void gnn_bwd_a_kernel(const float* __restrict__ h, const float* __restrict__ smean,
                      const float* __restrict__ g, float* __restrict__ a_out,
                      float* __restrict__ de_out, float* __restrict__ n_out, int M,
                      int H, int W, int C, int D) {
  const int lane = threadIdx.x & 63;
  const size_t cell_id = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int K = H * W;
  if (cell_id >= (size_t)M * K) return;
  const int m = cell_id / K;
  const int cell = cell_id - (size_t)m * K;
  const int y = cell / W, x = cell - y * W;
  const float* hrow = h + (size_t)m * K * C;
  const float* srow = smean + (size_t)m * K * D;
  const CVec<NG> hi = cvec_load<NG>(hrow + (size_t)cell * C, lane, C);
  const CVec<NG> gi = cvec_load<NG>(g + cell_id * C, lane, C);
  const SVec si = svec_load(srow + (size_t)cell * D, lane, D);
  float ssi = cvec_dot<NG>(hi, hi) + svec_dot(si, si);
  ssi = wave_sum(ssi);
  const float invi = rsqrtf(fmaxf(ssi, 1e-12f));
  float e[9], da[9];
  bool ok[9];
  float emax = -INFINITY;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
    ok[t] = (yy >= 0 && yy < H && xx >= 0 && xx < W);
    e[t] = 0.f; da[t] = 0.f;
    if (ok[t]) {
      const int cj = yy * W + xx;
      const CVec<NG> hj = cvec_load<NG>(hrow + (size_t)cj * C, lane, C);
      const SVec sj = svec_load(srow + (size_t)cj * D, lane, D);
      float ssj = cvec_dot<NG>(hj, hj) + svec_dot(sj, sj);
      float dot = cvec_dot<NG>(hi, hj) + svec_dot(si, sj);
      float dg = cvec_dot<NG>(gi, hj);
      ssj = wave_sum(ssj);
      dot = wave_sum(dot);
      da[t] = wave_sum(dg);
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
  float dsum = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    e[t] = e[t] * inv;            // a_t
    dsum += e[t] * da[t];
  }
  float my_a = 0.f, my_de = 0.f;
#pragma unroll
  for (int t = 0; t < 9; ++t)
    if (lane == t) { my_a = e[t]; my_de = e[t] * (da[t] - dsum); }
  if (lane < 9) {
    a_out[cell_id * 9 + lane] = my_a;
    de_out[cell_id * 9 + lane] = my_de;
  }
  if (lane == 0) n_out[cell_id] = invi;
}

// Pass B (one wave per cell j), gathers instead of scattering:
//   dh_j = g_j + sum_t a[j+d_t][8-t] g_{j+d_t} + du_j[:C]
//   df_j = sum_t (de[j][t] + de[j+d_t][8-t]) f_{j+d_t},  f_k = n_k u_k
//   du_j = n_j (df_j - f_j (f_j . df_j))      (n_j clamped: du_j = n_j df_j)
//   ds_j = du_j[C:]
template <int NG>
__global__ __launch_bounds__(256)
void gnn_bwd_b_kernel(const float* __restrict__ h, const float* __restrict__ smean,
                      const float* __restrict__ g, const float* __restrict__ a_in,
                      const float* __restrict__ de_in, const float* __restrict__ n_in,
                      float* __restrict__ dh, float* __restrict__ ds, int M, int H,
                      int W, int C, int D, int ds_accumulate) {
  const int lane = threadIdx.x & 63;
  const size_t cell_id = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int K = H * W;
  if (cell_id >= (size_t)M * K) return;
  const int m = cell_id / K;
  const int cell = cell_id - (size_t)m * K;
  const int y = cell / W, x = cell - y * W;
  const size_t rbase = (size_t)m * K;
  const float* hrow = h + rbase * C;
  const float* srow = smean + rbase * D;
  const float* grow = g + rbase * C;
  const CVec<NG> hj = cvec_load<NG>(hrow + (size_t)cell * C, lane, C);
  const SVec sj = svec_load(srow + (size_t)cell * D, lane, D);
  const float nj = n_in[cell_id];
  CVec<NG> acc = cvec_load<NG>(grow + (size_t)cell * C, lane, C);   // residual + weighted neighbours
  CVec<NG> dfh;                        // df_j, h part
#pragma unroll
  for (int gq = 0; gq < NG; ++gq) dfh.v[gq] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  SVec dfs = {0.f, 0.f};               // df_j, scene part (channels lane, lane + 64)
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
    if (yy >= 0 && yy < H && xx >= 0 && xx < W) {   // wave-uniform
      const int ck = yy * W + xx;
      const size_t kid = rbase + ck;
      const float a_kj = a_in[kid * 9 + (8 - t)];
      const float w = de_in[cell_id * 9 + t] + de_in[kid * 9 + (8 - t)];
      const float nk = n_in[kid];
      const CVec<NG> gk = cvec_load<NG>(grow + (size_t)ck * C, lane, C);
      const CVec<NG> hk = cvec_load<NG>(hrow + (size_t)ck * C, lane, C);
      const SVec sk = svec_load(srow + (size_t)ck * D, lane, D);
      const float wn = w * nk;
#pragma unroll
      for (int gq = 0; gq < NG; ++gq)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          acc.v[gq][q] = fmaf(a_kj, gk.v[gq][q], acc.v[gq][q]);
          dfh.v[gq][q] = fmaf(wn, hk.v[gq][q], dfh.v[gq][q]);
        }
      dfs.a = fmaf(wn, sk.a, dfs.a);
      dfs.b = fmaf(wn, sk.b, dfs.b);
    }
  }
  // projection f_j . df_j
  float proj = (cvec_dot<NG>(hj, dfh) + svec_dot(sj, dfs)) * nj;
  proj = wave_sum(proj);
  const bool clamped = nj >= 1.0e6f;   // |u|^2 <= 1e-12: l2_normalize is u * 1e6
#pragma unroll
  for (int gq = 0; gq < NG; ++gq) {
    const int c0 = gq * 256 + lane * 4;
    if (c0 >= C) continue;
    f32x4_t o;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float du = clamped ? nj * dfh.v[gq][q]
                               : nj * (dfh.v[gq][q] - (hj.v[gq][q] * nj) * proj);
      o[q] = acc.v[gq][q] + du;
    }
    *reinterpret_cast<f32x4_t*>(dh + cell_id * C + c0) = o;
  }
  if (lane < D) {
    const float du = clamped ? nj * dfs.a : nj * (dfs.a - (sj.a * nj) * proj);
    float* o2 = ds + cell_id * D + lane;
    *o2 = ds_accumulate ? *o2 + du : du;
  }
  if (lane + 64 < D) {
    const float du = clamped ? nj * dfs.b : nj * (dfs.b - (sj.b * nj) * proj);
    float* o2 = ds + cell_id * D + lane + 64;
    *o2 = ds_accumulate ? *o2 + du : du;
  }
}

// ------------------------------------------------------------ small 3x3 convs
// dgrad of a 3x3 SAME conv with few output channels (hidden2grid: Ci=C, Co=P;
// reg-decoder grid_emb: Ci=2, Co=E):
//   din[m][ci] (+)= sum_tap sum_co dout[m - d_tap][co] * W[tap][ci][co]
// One thread per (cell, ci).  Row strides in elements (time slices in place).
__global__ void conv3x3_small_dgrad_kernel(const float* __restrict__ dout,
                                           size_t dout_row_stride,
                                           const float* __restrict__ w,
                                           float* __restrict__ din,
                                           size_t din_row_stride, int M, int H, int W,
                                           int Ci, int Co, int accumulate) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)M * H * W * Ci;
  if (idx >= total) return;
  const int ci = idx % Ci;
  size_t r = idx / Ci;
  const int xx = r % W; r /= W;
  const int yy = r % H;
  const int m = r / H;
  float acc = 0.f;
  const bool vec4 = (Co % 4 == 0) && (dout_row_stride % 4 == 0) &&
                    ((reinterpret_cast<uintptr_t>(dout) | reinterpret_cast<uintptr_t>(w)) % 16 == 0);
  for (int t = 0; t < 9; ++t) {
    const int sy = yy - (t / 3 - 1), sx = xx - (t % 3 - 1);
    if (sy < 0 || sy >= H || sx < 0 || sx >= W) continue;
    const float* dp = dout + (size_t)m * dout_row_stride + ((size_t)sy * W + sx) * Co;
    const float* wp = w + ((size_t)t * Ci + ci) * Co;
    if (vec4) {          // 16-byte loads (Co = emb_size = 32: eight per tap instead of 32)
      for (int co = 0; co < Co; co += 4) {
        const f32x4_t d4 = *reinterpret_cast<const f32x4_t*>(dp + co);
        const f32x4_t w4 = *reinterpret_cast<const f32x4_t*>(wp + co);
        acc = fmaf(d4[0], w4[0], acc); acc = fmaf(d4[1], w4[1], acc);
        acc = fmaf(d4[2], w4[2], acc); acc = fmaf(d4[3], w4[3], acc);
      }
    } else {
    for (int co = 0; co < Co; ++co) acc = fmaf(dp[co], wp[co], acc);
    }
  }
  float* o = din + (size_t)m * din_row_stride + ((size_t)yy * W + xx) * Ci + ci;
  *o = accumulate ? *o + acc : acc;
}

// The same for Ci % 4 == 0 and Co <= 2 (hidden2grid): one thread per (cell, 4 input
// channels), 16-byte accesses of din / W, 32-bit index arithmetic.
template <int CO>
__global__ void conv3x3_small_dgrad4_kernel(const float* __restrict__ dout,
                                            size_t dout_row_stride,
                                            const float* __restrict__ w,
                                            float* __restrict__ din, size_t din_row_stride,
                                            int M, int H, int W, int Ci, int accumulate) {
  const int q = Ci >> 2;
  const unsigned idx = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned total = (unsigned)M * H * W * q;
  if (idx >= total) return;
  const int c4 = (int)(idx % (unsigned)q) * 4;
  unsigned r = idx / (unsigned)q;
  const int xx = r % (unsigned)W; r /= (unsigned)W;
  const int yy = r % (unsigned)H;
  const int m = r / (unsigned)H;
  f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
  const float* dbase = dout + (size_t)m * dout_row_stride;
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int sy = yy - (t / 3 - 1), sx = xx - (t % 3 - 1);
    if (sy < 0 || sy >= H || sx < 0 || sx >= W) continue;
    const float* dp = dbase + (size_t)(sy * W + sx) * CO;
    const float* wp = w + ((size_t)t * Ci + c4) * CO;       // [ci][co], 4 * CO floats
    if (CO == 1) {
      const f32x4_t wv = *reinterpret_cast<const f32x4_t*>(wp);
      const float d = dp[0];
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = fmaf(d, wv[i], acc[i]);
    } else {
      const f32x4_t w0 = *reinterpret_cast<const f32x4_t*>(wp);
      const f32x4_t w1 = *reinterpret_cast<const f32x4_t*>(wp + 4);
      const float d0 = dp[0], d1 = dp[1];
      // same order as the scalar kernel: co = 0 then co = 1 per channel
      acc[0] = fmaf(d1, w0[1], fmaf(d0, w0[0], acc[0]));
      acc[1] = fmaf(d1, w0[3], fmaf(d0, w0[2], acc[1]));
      acc[2] = fmaf(d1, w1[1], fmaf(d0, w1[0], acc[2]));
      acc[3] = fmaf(d1, w1[3], fmaf(d0, w1[2], acc[3]));
    }
  }
  f32x4_t* o = reinterpret_cast<f32x4_t*>(din + (size_t)m * din_row_stride +
                                          (size_t)(yy * W + xx) * Ci + c4);
  if (accumulate) {
    const f32x4_t old = *o;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = old[i] + acc[i];
  }
  *o = acc;
}

// wgrad of a 3x3 SAME conv with Ci*Co <= 512 (hidden2grid, grid_emb):
//   partial[blk][tap][ci][co] = sum_m in[m + d_tap][ci] * dout[m][co]
// in [R, HW, Ci], dout [R, HW, Co] contiguous; one thread per (ci, co), one
// workgroup per slab of cells; reduce with colsum_kernel afterwards.
// The wide operand is read ONCE per cell and the narrow one nine times (broadcast
// loads): WIDE_IN = false walks the output cells m of the slab (grid_emb: Co = 32
// wide, Ci <= 2), WIDE_IN = true walks the INPUT cells m' = m + d_tap (hidden2grid:
// Ci = 256 wide, Co <= 2), i.e. acc[tap] += in[m'] * dout[m' - d_tap].
// A workgroup = G cell groups x (Ci*Co) threads: group g takes the cells m0 + g,
// m0 + g + G, ... of the slab (narrow convs would otherwise run one wave per
// workgroup with a 200-cell serial loop); the groups' sums are folded through LDS
// in group order, so the result does not depend on timing.
template <bool WIDE_IN>
__global__ __launch_bounds__(512)
void conv3x3_small_wgrad_kernel(const float* __restrict__ in,
                                const float* __restrict__ dout,
                                float* __restrict__ partial, int R, int H,
                                int W, int Ci, int Co, int cells_per_block, int G) {
  extern __shared__ float wred[];          // [G][9][Ci*Co] when G > 1
  const int P = Ci * Co;
  const int tid = threadIdx.x;
  const int g = tid / P, pq = tid - g * P;
  const bool live = g < G;
  const int ci = pq / Co, co = pq - ci * Co;
  const long long total = (long long)R * H * W;
  const long long m0 = (long long)blockIdx.x * cells_per_block;
  long long m1 = m0 + cells_per_block;
  if (m1 > total) m1 = total;
  const int HW = H * W;
  float acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = 0.f;
  if (live) {
    int cell = (int)((m0 + g) % HW);
    for (long long m = m0 + g; m < m1; m += G) {
      const int y = cell / W, x = cell - y * W;
      if (WIDE_IN) {
        const float v = in[(size_t)m * Ci + ci];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int dy = t / 3 - 1, dx = t % 3 - 1;
          const int yy = y - dy, xx = x - dx;          // the output cell this input feeds
          if (yy >= 0 && yy < H && xx >= 0 && xx < W)
            acc[t] = fmaf(v, dout[(size_t)(m - dy * W - dx) * Co + co], acc[t]);
        }
      } else {
        const float d = dout[(size_t)m * Co + co];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int yy = y + t / 3 - 1, xx = x + t % 3 - 1;
          if (yy >= 0 && yy < H && xx >= 0 && xx < W)
            acc[t] = fmaf(in[(size_t)(m + (t / 3 - 1) * W + (t % 3 - 1)) * Ci + ci], d, acc[t]);
        }
      }
      cell += G;
      while (cell >= HW) cell -= HW;
    }
  }
  float* p = partial + (size_t)blockIdx.x * 9 * P;
  if (G == 1) {
    if (live) {
#pragma unroll
      for (int t = 0; t < 9; ++t) p[(size_t)t * P + pq] = acc[t];
    }
    return;
  }
  if (live) {
#pragma unroll
    for (int t = 0; t < 9; ++t) wred[(g * 9 + t) * P + pq] = acc[t];
  }
  __syncthreads();
  for (int i = tid; i < 9 * P; i += blockDim.x) {
    float sum = 0.f;
    for (int gg = 0; gg < G; ++gg) sum += wred[gg * 9 * P + i];
    p[i] = sum;
  }
}

// The WIDE_IN = true case (hidden2grid: Ci = 256 wide, Co <= 2) again, without the
// per-thread (y, x) division and the nine broadcast global loads per cell: a workgroup
// stages the d out values of its slab plus a halo of W + 1 cells, and a 9-bit tap mask per
// cell, in LDS once; thread = input channel ci, CO accumulators per tap.  Cells in slab
// order, one fma per (cell, tap, co): the sums are those of the kernel above, bit for bit.
// partial layout unchanged: [block][tap][ci * CO + co].  Dynamic LDS:
// (cells_per_block + 2 W + 2) * CO floats + cells_per_block ints.
template <int CO>
__global__ __launch_bounds__(256)
void h2g_wgrad_kernel(const float* __restrict__ in, const float* __restrict__ dout,
                      float* __restrict__ partial, int R, int H, int W, int Ci,
                      int cells_per_block) {
  extern __shared__ float sm_w[];
  const int halo = W + 1;
  const int nst = cells_per_block + 2 * halo;
  float* sd = sm_w;                                             // [nst][CO]
  int* smask = reinterpret_cast<int*>(sm_w + (size_t)nst * CO); // [cells_per_block]
  const int tid = threadIdx.x;
  const long long total = (long long)R * H * W;
  const long long m0 = (long long)blockIdx.x * cells_per_block;
  long long m1 = m0 + cells_per_block;
  if (m1 > total) m1 = total;
  const int HW = H * W;
  for (int i = tid; i < nst * CO; i += blockDim.x) {
    const long long m = m0 - halo + i / CO;
    sd[i] = (m >= 0 && m < total) ? dout[(size_t)m * CO + (i % CO)] : 0.f;
  }
  for (int i = tid; i < (int)(m1 - m0); i += blockDim.x) {
    const int cell = (int)((m0 + i) % HW);
    const int y = cell / W, x = cell - y * W;
    int mask = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int yy = y - (t / 3 - 1), xx = x - (t % 3 - 1);    // the output cell this input feeds
      if (yy >= 0 && yy < H && xx >= 0 && xx < W) mask |= 1 << t;
    }
    smask[i] = mask;
  }
  __syncthreads();
  for (int ci = tid; ci < Ci; ci += blockDim.x) {     // one pass per 256 input channels
    float acc[9][CO];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int co = 0; co < CO; ++co) acc[t][co] = 0.f;
    const float* ip = in + (size_t)m0 * Ci + ci;
    const int n = (int)(m1 - m0);
    for (int i = 0; i < n; ++i) {
      const float v = ip[(size_t)i * Ci];
      const int mask = smask[i];
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        if ((mask >> t) & 1) {
          const float* dp = sd + (size_t)(i + halo - (t / 3 - 1) * W - (t % 3 - 1)) * CO;
#pragma unroll
          for (int co = 0; co < CO; ++co) acc[t][co] = fmaf(v, dp[co], acc[t][co]);
        }
      }
    }
    float* p = partial + (size_t)blockIdx.x * 9 * Ci * CO;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
      for (int co = 0; co < CO; ++co) p[(size_t)t * Ci * CO + ci * CO + co] = acc[t][co];
  }
}

// dpre = dy * act'(pre) from the output y (tanh: 1 - y^2; relu / lrelu: the slope of y's
// sign); in place over dy allowed.
__global__ void tanh_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                float* __restrict__ dpre, size_t total, int act = 0) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  dpre[idx] = dy[idx] * act_grad(act, y[idx]);
}

// the same when y was stored scaled by 1 / keep_prob (dropped input of a cell):
// dpre = dy * (1 - (y * keep)^2)
__global__ void tanh_bwd_scaled_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                       float* __restrict__ dpre, size_t total, float keep,
                                       int act = 0) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  dpre[idx] = dy[idx] * act_grad(act, y[idx] * keep);
}

// out[s][col] = sum over the rows of slab s of in[row][col]; grid (nslab,
// ceil(ncols/256)).  Used twice for a deterministic column sum (bias
// gradients) and once with one slab to fold split-K partials.
__global__ void colsum_kernel(const float* __restrict__ in, float* __restrict__ out,
                              size_t rows, size_t ncols, size_t rows_per_slab) {
  const size_t col = (size_t)blockIdx.y * blockDim.x + threadIdx.x;
  if (col >= ncols) return;
  const size_t r0 = (size_t)blockIdx.x * rows_per_slab;
  size_t r1 = r0 + rows_per_slab;
  if (r1 > rows) r1 = rows;
  float acc = 0.f;
  for (size_t r = r0; r < r1; ++r) acc += in[r * ncols + col];
  out[(size_t)blockIdx.x * ncols + col] = acc;
}

// The same for narrow matrices (ncols divides 256, ncols <= 128): with one thread per
// column a 32-column sum ran 32 threads per workgroup on 128-byte rows (0.1 TB/s).  Here
// the 256 threads are 256/ncols row lanes x ncols columns (1 KB contiguous per pass); row
// lane l takes rows r0 + l, r0 + l + L, ...; the lanes are folded through LDS in lane
// order, so the result is deterministic.
__global__ __launch_bounds__(256)
void colsum_narrow_kernel(const float* __restrict__ in, float* __restrict__ out, size_t rows,
                          int ncols, size_t rows_per_slab) {
  __shared__ float red[256];
  const int L = 256 / ncols;
  const int l = threadIdx.x / ncols, col = threadIdx.x - l * ncols;
  const size_t r0 = (size_t)blockIdx.x * rows_per_slab;
  size_t r1 = r0 + rows_per_slab;
  if (r1 > rows) r1 = rows;
  float acc = 0.f;
  for (size_t r = r0 + l; r < r1; r += L) acc += in[r * ncols + col];
  red[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x < ncols) {
    float sum = 0.f;
    for (int k = 0; k < L; ++k) sum += red[k * ncols + threadIdx.x];
    out[(size_t)blockIdx.x * ncols + threadIdx.x] = sum;
  }
}

// out[0] = scale * sum(in[0..n)) (optionally of squares); ONE workgroup, fixed
// order: thread-strided partial sums, then an LDS tree.
__global__ __launch_bounds__(256)
void reduce_sum_kernel(const float* __restrict__ in, size_t n, float* __restrict__ out,
                       float scale, int squares) {
  __shared__ float red[256];
  float acc = 0.f;
  for (size_t i = threadIdx.x; i < n; i += 256) {
    const float v = in[i];
    acc += squares ? v * v : v;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0] * scale;
}

// partial[blockIdx.x] = sum of in[blockIdx.x * 4096 ..+4096): first stage of a long sum.  The
// one-workgroup kernel above walks n / 256 dependent strided loads per thread: 440 us for the
// 442 k Huber terms of the 18x32 scale; with this in front the pair takes < 10 us.
constexpr int kSumChunk = 4096;
__global__ __launch_bounds__(256)
void chunk_sum_kernel(const float* __restrict__ in, size_t n, float* __restrict__ partial) {
  __shared__ float red[256];
  const size_t base = (size_t)blockIdx.x * kSumChunk;
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < kSumChunk / 256; ++j) {
    const size_t i = base + (size_t)j * 256 + threadIdx.x;
    if (i < n) acc += in[i];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// ------------------------------------------------------------ losses
// sparse softmax cross entropy (code/pred_models.py:991-995) on the
// time-major logits [T, N, K]; labels [N, T] as the host hands them.
// One wave per (t, n) row: loss_row = lse - logit[label];
// dlogits = (softmax - onehot) * scale,  scale = grid_loss_weight / (N*T).
__global__ __launch_bounds__(64)
void ce_loss_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                    float* __restrict__ loss_row, float* __restrict__ dlogits, int T,
                    int N, int K, float scale) {
  const int r = blockIdx.x;          // t * N + n
  const int t = r / N, n = r - t * N;
  const int lane = threadIdx.x;
  const float* p = logits + (size_t)r * K;
  float mx = -INFINITY;
  for (int k = lane; k < K; k += 64) mx = fmaxf(mx, p[k]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += expf(p[k] - mx);
  s = wave_sum(s);
  const float lse = logf(s) + mx;
  const int lab = labels[(size_t)n * T + t];
  if (lane == 0) loss_row[r] = lse - p[lab];
  const float inv = 1.0f / s;
  float* d = dlogits + (size_t)r * K;
  for (int k = lane; k < K; k += 64) {
    const float sm = expf(p[k] - mx) * inv;
    d[k] = (sm - (k == lab ? 1.f : 0.f)) * scale;
  }
}

// Huber (delta = 1), tf.losses.huber_loss(reduction=MEAN)
// (code/pred_models.py:1020-1022): pred time-major [T, N, KP], target [N, T, KP].
// loss_elem = 0.5 q^2 + (|e| - q), q = min(|e|, 1);  dpred = clamp(e,-1,1) * scale.
__global__ void huber_loss_kernel(const float* __restrict__ pred,
                                  const float* __restrict__ target,
                                  float* __restrict__ loss_elem,
                                  float* __restrict__ dpred, int T, int N, int KP,
                                  float scale) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)T * N * KP;
  if (idx >= total) return;
  const int kp = idx % KP;
  const size_t r = idx / KP;
  const int n = r % N;
  const int t = r / N;
  const float e = pred[idx] - target[((size_t)n * T + t) * KP + kp];
  const float ab = fabsf(e);
  const float q = fminf(ab, 1.f);
  loss_elem[idx] = 0.5f * q * q + (ab - q);
  dpred[idx] = fminf(fmaxf(e, -1.f), 1.f) * scale;
}

// ------------------------------------------------------------ label maps
// Class ground-truth maps, time-major [T][N][K], from the labels [N][T]:
//   ks == 0  tf.one_hot(label)                       (code/pred_models.py:259-262)
//   ks > 0   the one-hot map stamped with the --soft_grid kernel (ks x ks, row-major;
//            scipy.ndimage.convolve(mode='constant') of a symmetric kernel, :1077-1124)
// Used as soft labels (:988-990), as teacher-forcing inputs (:398) and as the
// foreground mask of --mask_grid_regression (:1004-1010).
struct SoftKernel { float k[25]; };
__global__ void gt_class_maps_kernel(const int32_t* __restrict__ labels,
                                     float* __restrict__ out, int T, int N, int H, int W,
                                     int ks, const SoftKernel sk) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int K = H * W;
  if (idx >= (size_t)T * N * K) return;
  const int cell = (int)(idx % K);
  const size_t r = idx / K;
  const int n = (int)(r % N), t = (int)(r / N);
  const int lab = labels[(size_t)n * T + t];
  const int y = cell / W, x = cell - y * W, py = lab / W, px = lab - py * W;
  float v;
  if (ks == 0) {
    v = (cell == lab) ? 1.f : 0.f;
  } else {
    const int rad = ks >> 1, dy = y - py, dx = x - px;
    v = (dy >= -rad && dy <= rad && dx >= -rad && dx <= rad)
            ? sk.k[(dy + rad) * ks + (dx + rad)] : 0.f;
  }
  out[idx] = v;
}

// number of elements > 0 (the fg cells of --mask_grid_regression); *count zeroed before;
// integer atomics: exact and order-independent
__global__ void count_positive_kernel(const float* __restrict__ in, size_t n,
                                      int32_t* __restrict__ count) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  int c = 0;
  for (; i < n; i += (size_t)gridDim.x * blockDim.x) c += in[i] > 0.f;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
  if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, c);
}

// softmax_cross_entropy_with_logits on soft labels (code/pred_models.py:988-990):
// loss_row = -sum_k lab_k log_softmax(logits)_k; dlogits = (softmax - lab) * scale -- TF's
// REGISTERED gradient (xent_op backprop), exact only for labels that sum to one; the
// reference trains on it with un-normalised labels.  logits, labels time-major [T*N, K].
__global__ __launch_bounds__(64)
void ce_soft_loss_kernel(const float* __restrict__ logits, const float* __restrict__ soft,
                         float* __restrict__ loss_row, float* __restrict__ dlogits, int K,
                         float scale) {
  const int r = blockIdx.x, lane = threadIdx.x;
  const float* p = logits + (size_t)r * K;
  const float* q = soft + (size_t)r * K;
  float mx = -INFINITY;
  for (int k = lane; k < K; k += 64) mx = fmaxf(mx, p[k]);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off, 64));
  float s = 0.f;
  for (int k = lane; k < K; k += 64) s += expf(p[k] - mx);
  s = wave_sum(s);
  const float ls = logf(s);
  float acc = 0.f;
  for (int k = lane; k < K; k += 64) acc += q[k] * ((p[k] - mx) - ls);
  acc = wave_sum(acc);
  if (lane == 0) loss_row[r] = -acc;
  const float inv = 1.0f / s;
  float* d = dlogits + (size_t)r * K;
  for (int k = lane; k < K; k += 64) d[k] = (expf(p[k] - mx) * inv - q[k]) * scale;
}

// Huber over the foreground cells only (--mask_grid_regression, :999-1014): the mean runs
// over count[0] * 2 elements; pred / fg time-major [T, N, K(, 2)], target [N, T, K, 2].
__global__ void huber_masked_loss_kernel(const float* __restrict__ pred,
                                         const float* __restrict__ target,
                                         const float* __restrict__ fg,
                                         const int32_t* __restrict__ count,
                                         float* __restrict__ loss_elem,
                                         float* __restrict__ dpred, int T, int N, int K,
                                         float weight) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)T * N * K * 2;
  if (idx >= total) return;
  const int kp = (int)(idx % ((size_t)K * 2));
  const size_t r = idx / ((size_t)K * 2);
  const int n = (int)(r % N), t = (int)(r / N);
  const int cnt = count[0];
  const float scale = cnt > 0 ? weight / (2.0f * (float)cnt) : 0.f;
  const bool on = fg[idx >> 1] > 0.f;
  const float e = pred[idx] - target[((size_t)n * T + t) * K * 2 + kp];
  const float ab = fabsf(e);
  const float q = fminf(ab, 1.f);
  loss_elem[idx] = on ? 0.5f * q * q + (ab - q) : 0.f;
  dpred[idx] = on ? fminf(fmaxf(e, -1.f), 1.f) * scale : 0.f;
}
// loss = weight * sum(loss_elem) / (2 count)
__global__ void masked_mean_kernel(const float* __restrict__ sum, const int32_t* __restrict__ count,
                                   float* __restrict__ out, float weight) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int cnt = count[0];
    out[0] = cnt > 0 ? sum[0] / (2.0f * (float)cnt) * weight : 0.f;
  }
}

// ------------------------------------------------------------ dropout
// x *= keep_mask / keep_prob in place (tf.nn.rnn_cell.DropoutWrapper input dropout,
// code/pred_models.py:194-202, 241-249; also its backward on d x).  Element i of draw
// `stream` is kept iff the top 24 bits of hash32(i, seed, stream) < keep_prob * 2^24 --
// the generator oracle/multiverse_oracle.py dropout_keep_mask restates.
__device__ __forceinline__ bool dropout_keep(uint32_t i, uint32_t seed, uint32_t stream,
                                             uint32_t thr) {
  uint32_t x = i * 0x9E3779B1u + seed * 0x85EBCA77u + stream * 0xC2B2AE3Du;
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return (x >> 8) < thr;
}
__global__ void dropout_kernel(float* __restrict__ x, size_t n, uint32_t seed, uint32_t stream,
                               uint32_t thr, float inv_keep) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  x[i] = dropout_keep((uint32_t)i, seed, stream, thr) ? x[i] * inv_keep : 0.f;
}

// ------------------------------------------------------------ scene stack
// d scene_conv_s [U, K, D] gathered from (a) the graph attention's d scene_mean
// [N, K, D] (reduce_mean over T_o: each observed frame gets 1/T) and (b) the
// class encoder's dx_t [T, N, K, D] at the occupied cell
// (scene_conv * one_hot, code/pred_models.py:174-175, 210).
__global__ void scene_grad_gather_kernel(const float* __restrict__ dmean,
                                         const float* __restrict__ dxenc,
                                         const int32_t* __restrict__ obs_scene,
                                         const int32_t* __restrict__ labels,
                                         float* __restrict__ dsc, int U, int N, int T,
                                         int K, int D,
                                         const int32_t* __restrict__ labels2 = nullptr,
                                         float mixw = 1.f) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t per = (size_t)K * D;
  if (idx >= (size_t)U * per) return;
  const int u = idx / per;
  const size_t off = idx - (size_t)u * per;
  const int cell = off / D;
  float acc = 0.f;
  for (int n = 0; n < N; ++n)
    for (int t = 0; t < T; ++t) {
      if (obs_scene[n * T + t] != u) continue;
      if (dmean) acc += dmean[(size_t)n * per + off] / (float)T;
      if (labels2) {       // label mixup: the class-encoder input was scene_conv * m
        const float m = mixw * (labels[n * T + t] == cell ? 1.f : 0.f) +
                        (labels2[n * T + t] == cell ? 1.f : 0.f) * (1.f - mixw);
        if (m != 0.f) acc += dxenc[((size_t)t * N + n) * per + off] * m;
      } else if (labels[n * T + t] == cell)
        acc += dxenc[((size_t)t * N + n) * per + off];
    }
  dsc[idx] = acc;
}

// ------------------------------------------------------------ label mixup (SimAug, exp 3)
// out [T][N][K] (time-major) = w * one_hot(l1[n * sn + t * st]) + one_hot(l2[..]) * (1 - w)
__global__ void twohot_map_kernel(const int32_t* __restrict__ l1, const int32_t* __restrict__ l2,
                                  int sn, int st, float w, float* __restrict__ out, int T,
                                  int N, int K) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)T * N * K) return;
  const int k = (int)(idx % K);
  const size_t r = idx / K;
  const int n = (int)(r % N), t = (int)(r / N);
  const size_t li = (size_t)n * sn + (size_t)t * st;
  out[idx] = w * (l1[li] == k ? 1.f : 0.f) + (l2[li] == k ? 1.f : 0.f) * (1.f - w);
}

// per-sample weights on the class loss (double_weighting, :1391-1398): rows time-major
// r = t * N + n; the loss row and its gradient row are scaled by sw[n]
__global__ void scale_rows_kernel(float* __restrict__ loss_row, float* __restrict__ drows,
                                  const float* __restrict__ sw, int rows, int N, int K) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)rows * K) return;
  const size_t r = idx / K;
  const float f = sw[r % N];
  drows[idx] = drows[idx] * f;
  if (idx - r * K == 0) loss_row[r] = loss_row[r] * f;
}

// dgrad of conv k x k, stride 2, SAME: din[u][iy][ix][ci] (+)= sum dpre[u][oy][ox][co] W[ky][kx][ci][co]
// over (ky,kx) with 2*oy + ky - pad_t == iy.
__global__ void conv_s2_dgrad_kernel(const float* __restrict__ dpre,
                                     const float* __restrict__ w,
                                     float* __restrict__ din, int U, int Hi, int Wi,
                                     int Ci, int Ho, int Wo, int Co, int k, int pad_t,
                                     int pad_l, int accumulate) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)U * Hi * Wi * Ci;
  if (idx >= total) return;
  const int ci = idx % Ci;
  size_t r = idx / Ci;
  const int ix = r % Wi; r /= Wi;
  const int iy = r % Hi;
  const int u = r / Hi;
  float acc = 0.f;
  for (int ky = 0; ky < k; ++ky) {
    const int ty = iy + pad_t - ky;
    if (ty < 0 || (ty & 1)) continue;
    const int oy = ty >> 1;
    if (oy >= Ho) continue;
    for (int kx = 0; kx < k; ++kx) {
      const int tx = ix + pad_l - kx;
      if (tx < 0 || (tx & 1)) continue;
      const int ox = tx >> 1;
      if (ox >= Wo) continue;
      const float* dp = dpre + (((size_t)u * Ho + oy) * Wo + ox) * Co;
      const float* wp = w + ((size_t)(ky * k + kx) * Ci + ci) * Co;
      for (int co = 0; co < Co; ++co) acc = fmaf(dp[co], wp[co], acc);
    }
  }
  din[idx] = accumulate ? din[idx] + acc : acc;
}

// wgrad of conv k x k, stride 2, SAME; one thread per (ky,kx,ci,co), direct sum over
// the output rows [blockIdx.y * rows_per_slab, ...) of the flattened (u, oy) index;
// dw is [gridDim.y][k*k*Ci*Co] partials, folded by colsum_kernel.
__global__ void conv_s2_wgrad_kernel(const float* __restrict__ in,
                                     const float* __restrict__ dpre,
                                     float* __restrict__ dw, int U, int Hi, int Wi,
                                     int Ci, int Ho, int Wo, int Co, int k, int pad_t,
                                     int pad_l, int rows_per_slab) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)k * k * Ci * Co;
  if (idx >= total) return;
  const int co = idx % Co;
  size_t r = idx / Co;
  const int ci = r % Ci; r /= Ci;
  const int kx = r % k;
  const int ky = r / k;
  float acc = 0.f;
  const int r0 = blockIdx.y * rows_per_slab;
  const int r1 = min(r0 + rows_per_slab, U * Ho);
  for (int row = r0; row < r1; ++row) {
    const int u = row / Ho, oy = row - u * Ho;
    const int iy = oy * 2 + ky - pad_t;
    if (iy < 0 || iy >= Hi) continue;
    for (int ox = 0; ox < Wo; ++ox) {
      const int ix = ox * 2 + kx - pad_l;
      if (ix < 0 || ix >= Wi) continue;
      acc = fmaf(in[(((size_t)u * Hi + iy) * Wi + ix) * Ci + ci],
                 dpre[(((size_t)u * Ho + oy) * Wo + ox) * Co + co], acc);
    }
  }
  dw[(size_t)blockIdx.y * total + idx] = acc;
}

// ------------------------------------------------------------ SimAug (N4)
// One targeted FGSM / PGD step on the scene features (SimAug/code/pred_models.py:94-126
// one_step_attack): x <- clip(x - step * sign(g), lo, hi) with lo = clip(clean - eps, -1, 1),
// hi = clip(clean + eps, -1, 1) (:137-138); tf.sign(0) = 0.
__global__ void adv_step_kernel(float* __restrict__ x, const float* __restrict__ g,
                                const float* __restrict__ clean, float eps, float step,
                                size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gv = g[i];
  const float sg = gv > 0.f ? 1.f : (gv < 0.f ? -1.f : 0.f);
  const float lo = fminf(fmaxf(clean[i] - eps, -1.f), 1.f);
  const float hi = fminf(fmaxf(clean[i] + eps, -1.f), 1.f);
  const float v = x[i] - sg * step;
  x[i] = fminf(fmaxf(v, lo), hi);          // tf.clip_by_value: min(max(v, lo), hi)
}
// mixup (:151-166, :529-536): x <- other * weight + x * (1 - weight)
__global__ void mix_kernel(float* __restrict__ x, const float* __restrict__ other, float weight,
                           size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  x[i] = other[i] * weight + x[i] * (1.f - weight);
}
// per-sample mean over the T steps of the per-row cross entropy (:403-404), rows time-major
__global__ void loss_rows_mean_kernel(const float* __restrict__ loss_row, float* __restrict__ out,
                                      int T, int N) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x;
  if (n >= N) return;
  float s = 0.f;
  for (int t = 0; t < T; ++t) s += loss_row[(size_t)t * N + n];
  out[n] = s / (float)T;
}

// ------------------------------------------------------------ optimizer
// grad += wd * w  (the d/dw of wd * l2_loss(w), wd_cost code/pred_models.py:1253-1275)
__global__ void add_scaled_kernel(float* __restrict__ g, const float* __restrict__ w,
                                  float s, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) g[i] = g[i] + s * w[i];
}

// The same with the weight-decay cost's sum of squares taken from the pass (the parameter is
// read once per step instead of twice: a one-workgroup reduce_sum over a 20 MB kernel took
// 100-430 us).  A workgroup owns 4 096 consecutive elements: thread-strided partial sums in a
// fixed order, LDS tree, partial[blockIdx.x]; reduce_sum_kernel folds the partials.
constexpr int kWdChunk = 4096;
__global__ __launch_bounds__(256)
void add_scaled_sumsq_kernel(float* __restrict__ g, const float* __restrict__ w, float s,
                             size_t n, float* __restrict__ partial) {
  __shared__ float red[256];
  const size_t base = (size_t)blockIdx.x * kWdChunk;
  float acc = 0.f;
#pragma unroll
  for (int j = 0; j < kWdChunk / 256; ++j) {
    const size_t i = base + (size_t)j * 256 + threadIdx.x;
    if (i < n) {
      const float wv = w[i];
      g[i] = g[i] + s * wv;
      acc += wv * wv;
    }
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if ((int)threadIdx.x < off) red[threadIdx.x] += red[threadIdx.x + off];
    __syncthreads();
  }
  if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// clip_by_value (code/pred_models.py:1700-1705) + TF ApplyAdadelta
// (AdadeltaOptimizer(lr, rho=0.95, epsilon=1e-8), :1671-1672):
//   accum = rho accum + (1-rho) g^2
//   update = sqrt(accum_update + eps) * rsqrt(accum + eps) * g
//   var -= lr * update;  accum_update = rho accum_update + (1-rho) update^2
__global__ void adadelta_kernel(float* __restrict__ var, float* __restrict__ accum,
                                float* __restrict__ accum_update,
                                const float* __restrict__ grad, float gscale, float clip,
                                int do_clip, float lr, float rho, float eps, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float g = grad[i] * gscale;
  if (do_clip) g = fminf(fmaxf(g, -clip), clip);
  const float a = accum[i] * rho + g * g * (1.f - rho);
  const float upd = sqrtf(accum_update[i] + eps) * (1.0f / sqrtf(a + eps)) * g;
  var[i] = var[i] - upd * lr;
  accum_update[i] = accum_update[i] * rho + upd * upd * (1.f - rho);
  accum[i] = a;
}

// clip_by_value + TF ApplyMomentum (MomentumOptimizer(lr, 0.9), no Nesterov, :1668):
//   accum = accum * momentum + g;  var -= lr * accum
__global__ void momentum_kernel(float* __restrict__ var, float* __restrict__ accum,
                                const float* __restrict__ grad, float gscale, float clip,
                                int do_clip, float lr, float momentum, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float g = grad[i] * gscale;
  if (do_clip) g = fminf(fmaxf(g, -clip), clip);
  const float a = accum[i] * momentum + g;
  var[i] = var[i] - a * lr;
  accum[i] = a;
}

// clip_by_value + TF ApplyAdam (AdamOptimizer(lr): beta 0.9 / 0.999, eps 1e-8, :1674):
//   m += (g - m)(1 - beta1);  v += (g^2 - v)(1 - beta2);  var -= m * alpha / (sqrt(v) + eps)
// alpha = lr sqrt(1 - beta2_power) / (1 - beta1_power), computed on the host in float32
__global__ void adam_kernel(float* __restrict__ var, float* __restrict__ m,
                            float* __restrict__ v, const float* __restrict__ grad,
                            float gscale, float clip, int do_clip, float alpha, float omb1,
                            float omb2, float eps, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float g = grad[i] * gscale;
  if (do_clip) g = fminf(fmaxf(g, -clip), clip);
  const float mm = m[i] + (g - m[i]) * omb1;
  const float vv = v[i] + (g * g - v[i]) * omb2;
  var[i] = var[i] - (mm * alpha) / (sqrtf(vv) + eps);
  m[i] = mm; v[i] = vv;
}

// clip_by_value + TF ApplyRMSProp (RMSPropOptimizer(lr): decay 0.9, momentum 0,
// eps 1e-10, :1677):  ms += (g^2 - ms)(1 - decay);  mom = mom * momentum +
// (g * lr) * rsqrt(ms + eps);  var -= mom     (ms starts at ONE)
__global__ void rmsprop_kernel(float* __restrict__ var, float* __restrict__ ms,
                               float* __restrict__ mom, const float* __restrict__ grad,
                               float gscale, float clip, int do_clip, float lr, float omd,
                               float momentum, float eps, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float g = grad[i] * gscale;
  if (do_clip) g = fminf(fmaxf(g, -clip), clip);
  const float s2 = ms[i] + (g * g - ms[i]) * omd;
  const float mo = mom[i] * momentum + (g * lr) * (1.0f / sqrtf(s2 + eps));
  var[i] = var[i] - mo;
  ms[i] = s2; mom[i] = mo;
}

__global__ void fill_kernel(float* __restrict__ x, float v, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = v;
}

// ------------------------------------------------------------ layout helpers
// dense one-hot map [R, K] from ids (row r reads ids[r * stride])
__global__ void onehot_map_kernel(const int32_t* __restrict__ ids, int stride,
                                  float* __restrict__ out, int R, int K) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)R * K) return;
  const int r = idx / K, k = idx - (size_t)r * K;
  out[idx] = (ids[(size_t)r * stride] == k) ? 1.f : 0.f;
}

// [N, T, E] -> [T, N, E]
__global__ void transpose_nt_kernel(const float* __restrict__ in, float* __restrict__ out,
                                    int N, int T, size_t E) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (size_t)N * T * E) return;
  const size_t e = idx % E;
  const size_t r = idx / E;
  const int n = r % N;
  const int t = r / N;
  out[idx] = in[((size_t)n * T + t) * E + e];
}

// Device-side weight packs (same maps as pack_convlstm_weights /
// pack_convlstm_dgrad_weights in convlstm_mfma.h), run after every optimizer
// step.  One thread per packed element.
__global__ void pack_fwd_kernel(const float* __restrict__ w, float* __restrict__ out,
                                int Cx, int C, int nx, int nch, int small, size_t total) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int j = idx & 3;
  const int l = (idx >> 2) & 63;
  const int g = (idx >> 8) & 3;
  const int kk = (idx >> 10) & 3;
  const size_t tile = idx >> 12;
  const int q = tile % nch;
  const int cb = tile / nch;
  const int k = kk * 8 + (l >> 5) * 4 + j;
  const int n = g * C + cb * 32 + (l & 31);
  const int Cin = Cx + C, N4 = 4 * C;
  int tap = -1, ci = -1;
  if (q < nx) {
    if (small) {
      if (k < 9 * Cx) { tap = k / Cx; ci = k % Cx; }
    } else {
      tap = q % 9; ci = (q / 9) * 32 + k;
    }
  } else {
    const int qq = q - nx;
    tap = qq % 9; ci = Cx + (qq / 9) * 32 + k;
  }
  out[idx] = (tap < 0) ? 0.f : w[((size_t)tap * Cin + ci) * N4 + n];
}

__global__ void pack_dgrad_kernel(const float* __restrict__ w, float* __restrict__ out,
                                  int Cx, int C, int nch, size_t total) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int j = idx & 3;
  const int l = (idx >> 2) & 63;
  const int g = (idx >> 8) & 3;
  const int kk = (idx >> 10) & 3;
  const size_t tile = idx >> 12;
  const int q = tile % nch;
  const int cb = tile / nch;
  const int grp = q / 9, tap = q % 9;
  const int k = kk * 8 + (l >> 5) * 4 + j;
  const int n = grp * 32 + k;
  const int col = cb * 128 + g * 32 + (l & 31);
  const int Cin = Cx + C, N4 = 4 * C;
  int ci = -1;
  if (col < C) ci = Cx + col;
  else if (col - C < Cx) ci = col - C;
  out[idx] = (ci < 0) ? 0.f : w[((size_t)(8 - tap) * Cin + ci) * N4 + n];
}

}  // namespace mv
