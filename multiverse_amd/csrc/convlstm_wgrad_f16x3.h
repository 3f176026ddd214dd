// wgrad of the ConvLSTM gate convolution on the fp16 matrix pipe (f16x3 split,
// see convlstm_f16x3.h), h rows of the kernel gradient:
//
//   dW[tap][Cx + ci][n] = sum_m  h[m + d_tap][ci] * G[m][n]
//
// v_mfma_f32_32x32x16_f16 wants 8 consecutive REDUCTION elements per lane, and the
// reduction runs over cells, so both operands are first re-laid cell-contiguous
// ("transposed") as two scaled fp16 planes:
//   GT        [plane][Mrow/32][4C][32]     scale 2^e (chain-wide, from max |G|)
//   AT[dx]    [plane][Mrow/32][C][32]      dx=-1,0,1: AT[dx][ci][m] = h[m + dx][ci], zero where
//                                          x(m) + dx leaves the image row; scale 2^8
//                                          (|h| <= 1); an x operand (pixel offsets are
//                                          not bounded) gets its own exponent from max|x|
// (cells blocked by 32 = one pipeline stage, so the tile rows a workgroup stages are
// ONE contiguous 8 KB run per plane instead of 64-byte pieces at a row stride of
// hundreds of KB -- DRAM-page and TLB friendly.)
// With W % 16 == 0 a k-step of 16 cells lies inside one image row, so the column
// shift of a tap is baked into the operand copy (three copies), the row shift
// dy*W is a multiple of 16 cells = an ALIGNED offset, and a k-step whose tap row
// is outside the image is simply skipped: no per-element masks in the GEMM.
//
// GEMM tile: a workgroup owns 128 (ci) x 128 (n) of one tap, four waves 2 x 2,
// each 64 x 64 (2 x 2 accumulators); a stage = 32 cells = two k-steps; the A and
// G tiles of a stage (2 x 20 KB) are copied global -> LDS (16-B vectors along
// cells) into a double buffer, fragments come back with ds_read_b128; 8 reads
// and 12 MFMAs per k-step and wave, the same ratio as the forward kernel.
// Split-K over cell ranges pinned to XCDs, per-split partial tiles summed by the
// common reduction pass (bitwise reproducible).  The x rows (Cx = 64 / 32 / 2
// channels) use the same tile with rows enumerating (tap, channel) pairs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>

#include "convlstm_f16x3.h"

namespace mv {

// element (cell m, row r) of a plane with R rows: cells blocked by 32
__device__ __forceinline__ size_t wg16_plane_index(long long m, int r, int R) {
  return ((size_t)(m >> 5) * R + r) * 32 + (size_t)(m & 31);
}

// in fp32 [Mtot][Cc] -> out [2][Mrow/32][Cc][32] halves; out[c][m] = in[m+dx][c] * 2^e,
// zero when the shifted cell leaves its image row or the tensor.  Block = 64 cells
// x 64 channels through LDS; grid (Mrow/64, Cc/64).
__global__ __launch_bounds__(256)
void transpose_split_kernel(const float* __restrict__ in, _Float16* __restrict__ out,
                            long long Mtot, int Cc, long long Mrow, int W, int dx,
                            const int32_t* __restrict__ exp_ptr, int exp_const,
                            float* __restrict__ colsum_out, int nplanes = 2) {
  __shared__ float tile[64][65];
  const int e = exp_ptr ? exp_ptr[0] : exp_const;
  const float sc2e = __int_as_float((127 + e) << 23);     // 2^e, |e| <= 100 (one multiply = ldexpf)
  const long long m0 = (long long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
  // load: 16 threads x float4 cover the 64 channels of a cell, 16 cells per pass
  const int lc = (tid & 15) * 4, lj = tid >> 4;
#pragma unroll
  for (int pass = 0; pass < 4; ++pass) {
    const int j = pass * 16 + lj;
    const long long m = m0 + j;
    const int x = (int)(m % W);
    const long long src = m + dx;
    const bool ok = (m < Mtot) & (x + dx >= 0) & (x + dx < W) & (src >= 0) & (src < Mtot);
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok) v = *reinterpret_cast<const f32x4*>(in + (size_t)src * Cc + c0 + lc);
    tile[j][lc + 0] = v[0]; tile[j][lc + 1] = v[1];
    tile[j][lc + 2] = v[2]; tile[j][lc + 3] = v[3];
  }
  __syncthreads();
  if (colsum_out && tid < 64) {          // per-block column sums (bias gradient partials)
    float sum = 0.f;
#pragma unroll 8
    for (int jj = 0; jj < 64; ++jj) sum += tile[jj][tid];
    colsum_out[(size_t)blockIdx.x * Cc + c0 + tid] = sum;
  }
  // store: item = (channel, group of 8 cells): 64 x 8 = 512 items, 2 per thread
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const int item = it * 256 + tid;
    const int ch = item >> 3, grp = item & 7;
    f16x8 p0, p1;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float s = tile[grp * 8 + q][ch] * sc2e;
      const _Float16 h0 = (_Float16)s;
      p0[q] = h0;
      p1[q] = (_Float16)(s - (float)h0);
    }
    const size_t o = wg16_plane_index(m0 + grp * 8, c0 + ch, Cc);
    *reinterpret_cast<f16x8*>(out + o) = p0;
    if (nplanes == 2) *reinterpret_cast<f16x8*>(out + (size_t)Cc * Mrow + o) = p1;
  }
}

// The three column-shifted copies (dx = -1, 0, +1) of one operand from ONE read: the block
// stages its 64 cells plus one neighbour each side.  Same arithmetic per element as
// transpose_split_kernel (bit-identical planes); grid (Mrow/64, Cc/64).
__global__ __launch_bounds__(256)
void transpose_split3_kernel(const float* __restrict__ in, _Float16* __restrict__ out0,
                             _Float16* __restrict__ out1, _Float16* __restrict__ out2,
                             long long Mtot, int Cc, long long Mrow, int W,
                             const int32_t* __restrict__ exp_ptr, int exp_const, int nplanes) {
  __shared__ float tile[66][65];                   // tile[j] = cell m0 - 1 + j
  const int e = exp_ptr ? exp_ptr[0] : exp_const;
  const float sc2e = __int_as_float((127 + e) << 23);
  const long long m0 = (long long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 64;
  const int tid = threadIdx.x;
  const int lc = (tid & 15) * 4, lj = tid >> 4;
#pragma unroll
  for (int pass = 0; pass < 5; ++pass) {
    const int j = pass * 16 + lj;
    if (j < 66) {
      const long long src = m0 - 1 + j;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (src >= 0 && src < Mtot) v = *reinterpret_cast<const f32x4*>(in + (size_t)src * Cc + c0 + lc);
      tile[j][lc + 0] = v[0]; tile[j][lc + 1] = v[1];
      tile[j][lc + 2] = v[2]; tile[j][lc + 3] = v[3];
    }
  }
  __syncthreads();
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    _Float16* const out = d == 0 ? out0 : (d == 1 ? out1 : out2);
    const int dx = d - 1;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int item = it * 256 + tid;
      const int ch = item >> 3, grp = item & 7;
      f16x8 p0, p1;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const long long m = m0 + grp * 8 + q;
        const int x = (int)(m % W);
        const bool ok = (m < Mtot) & (x + dx >= 0) & (x + dx < W);
        const float sv = ok ? tile[grp * 8 + q + 1 + dx][ch] * sc2e : 0.f;
        const _Float16 h0 = (_Float16)sv;
        p0[q] = h0;
        p1[q] = (_Float16)(sv - (float)h0);
      }
      const size_t o = wg16_plane_index(m0 + grp * 8, c0 + ch, Cc);
      *reinterpret_cast<f16x8*>(out + o) = p0;
      if (nplanes == 2) *reinterpret_cast<f16x8*>(out + (size_t)Cc * Mrow + o) = p1;
    }
  }
}

// the same for a narrow tensor (Cc < 64, any Cc): one block = 64 cells x all channels
__global__ __launch_bounds__(256)
void transpose_split_narrow_kernel(const float* __restrict__ in, _Float16* __restrict__ out,
                                   long long Mtot, int Cc, long long Mrow, int W, int dx,
                                   const int32_t* __restrict__ exp_ptr, int nplanes = 2) {
  __shared__ float tile[64][65];
  const int exp_const = exp_ptr[0];
  const float sc2e = __int_as_float((127 + exp_const) << 23);
  const long long m0 = (long long)blockIdx.x * 64;
  const int tid = threadIdx.x;
  for (int idx = tid; idx < 64 * Cc; idx += 256) {
    const int jj = idx / Cc, c = idx - jj * Cc;
    const long long m = m0 + jj;
    const int x = (int)(m % W);
    const long long src = m + dx;
    const bool ok = (m < Mtot) & (x + dx >= 0) & (x + dx < W) & (src >= 0) & (src < Mtot);
    tile[jj][c] = ok ? in[(size_t)src * Cc + c] : 0.f;
  }
  __syncthreads();
  for (int item = tid; item < Cc * 8; item += 256) {
    const int ch = item >> 3, grp = item & 7;
    f16x8 p0, p1;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float s = tile[grp * 8 + q][ch] * sc2e;
      const _Float16 h0 = (_Float16)s;
      p0[q] = h0;
      p1[q] = (_Float16)(s - (float)h0);
    }
    const size_t o = wg16_plane_index(m0 + grp * 8, ch, Cc);
    *reinterpret_cast<f16x8*>(out + o) = p0;
    if (nplanes == 2) *reinterpret_cast<f16x8*>(out + (size_t)Cc * Mrow + o) = p1;
  }
}

// ---------------------------------------------------------------- the row-triple form (round 6)
// dW in Winograd F(3,3) form over image-row triples -- the forward's algebra (convlstm_wino3.h)
// with the roles of kernel rows and output rows exchanged: per stencil column dx and triple t,
//     dg_k = sum_i G(3t + i) d_{i+k}            (d0..d4 = input rows 3t-1 .. 3t+3, k = 0..2)
// is the correlation the forward computes with the gate-gradient rows as its "kernel", so
//     U0 = G0 / 2   U1 = -(G0 + G1 + G2) / 2   U2 = (-G0 + G1 - G2) / 6   U3 = (G0 + 2 G1 + 4 G2) / 6
//     U4 = -G2      V0..V4 = the forward's input components of d
//     M_c = sum over triple-cells of V_c U_c     (the GEMMs: 15 "taps" = 5 components x 3 dx)
//     dg_0 = M0 + M1 + M2 + M3    dg_1 = M1 - M2 + 2 M3    dg_2 = M1 + M2 + 4 M3 + M4
// and the sums over cells commute with the output combination: the GEMMs run on a third of the
// cells with 15 taps instead of 9 -- 5/9 of the MFMAs -- and the three-row combination is applied
// once, to the summed partials (wgrad_wino3_reduce_kernel).  The operands are transformed in fp32
// BEFORE the split into planes (no extra rounding in the planes); both transposes below write
// the GEMM kernels' cell-blocked layout per component.  |V| <= 6 |d|, |U| <= 1.5 |G|: the
// exponents of the direct form keep every plane finite.
//
// G operand: in fp32 [Mtot][Cc] -> out [5][2][Mrow3/32][Cc][32]; triple-cell m3 = (q, x), q = image
// * H/3 + triple, reads the raw cells (3q + i) W + x.  Bias partials: sum of the three rows =
// -2 U1.  Block = 64 triple-cells x 32 channels; grid (Mrow3/64, Cc/32).
__global__ __launch_bounds__(256)
void wino3_transpose_g_kernel(const float* __restrict__ in, _Float16* __restrict__ out,
                              long long Mtot3, int Cc, long long Mrow3, int W,
                              const int32_t* __restrict__ exp_ptr, float* __restrict__ colsum_out,
                              long long comp_stride, int nplanes) {
  __shared__ float tile[5][64][33];
  const int e = exp_ptr[0];
  const float sc2e = __int_as_float((127 + e) << 23);
  const long long m0 = (long long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 32;
  const int tid = threadIdx.x;
  const int lc = (tid & 7) * 4, lj = tid >> 3;
#pragma unroll
  for (int pass = 0; pass < 2; ++pass) {
    const int j = pass * 32 + lj;
    const long long m3 = m0 + j;
    f32x4 g0 = {0.f, 0.f, 0.f, 0.f}, g1 = g0, g2 = g0;
    if (m3 < Mtot3) {
      const long long q = m3 / W;
      const int x = (int)(m3 - q * W);
      const float* src = in + (size_t)(3 * q * W + x) * Cc + c0 + lc;
      g0 = *reinterpret_cast<const f32x4*>(src);
      g1 = *reinterpret_cast<const f32x4*>(src + (size_t)W * Cc);
      g2 = *reinterpret_cast<const f32x4*>(src + (size_t)2 * W * Cc);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      tile[0][j][lc + k] = 0.5f * g0[k];
      tile[1][j][lc + k] = -0.5f * ((g0[k] + g1[k]) + g2[k]);
      tile[2][j][lc + k] = ((g1[k] - g0[k]) - g2[k]) * (1.0f / 6.0f);
      tile[3][j][lc + k] = ((g0[k] + 2.0f * g1[k]) + 4.0f * g2[k]) * (1.0f / 6.0f);
      tile[4][j][lc + k] = -g2[k];
    }
  }
  __syncthreads();
  if (colsum_out && tid < 32) {          // per-block column sums of G (bias gradient partials)
    float sum = 0.f;
#pragma unroll 8
    for (int jj = 0; jj < 64; ++jj) sum += tile[1][jj][tid];
    colsum_out[(size_t)blockIdx.x * Cc + c0 + tid] = -2.0f * sum;
  }
  const int ch = tid >> 3, grp = tid & 7;
  const size_t o = wg16_plane_index(m0 + grp * 8, c0 + ch, Cc);
#pragma unroll
  for (int comp = 0; comp < 5; ++comp) {
    f16x8 p0, p1;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const float sv = tile[comp][grp * 8 + q][ch] * sc2e;
      const _Float16 h0 = (_Float16)sv;
      p0[q] = h0;
      p1[q] = (_Float16)(sv - (float)h0);
    }
    _Float16* const oc = out + (size_t)comp * (size_t)comp_stride + o;
    *reinterpret_cast<f16x8*>(oc) = p0;
    if (nplanes == 2) *reinterpret_cast<f16x8*>(oc + (size_t)Cc * Mrow3) = p1;
  }
}

// A operand (h, or the x rows' input): in fp32 [Mtot][Cc] -> the three column-shifted copies
// out{0,1,2} [5][2][Mrow3/32][Cc][32] of the components V0..V4; rows outside the image are zero
// (H = image rows, a multiple of 3), a shifted cell outside its image row is zero.  Any Cc (the
// pixel-offset inputs have 2 channels): block = 64 triple-cells (+ one neighbour each side) x 32
// channels; grid (Mrow3/64, ceil(Cc/32)).
__global__ __launch_bounds__(256)
void wino3_transpose_a3_kernel(const float* __restrict__ in, _Float16* __restrict__ out0,
                               _Float16* __restrict__ out1, _Float16* __restrict__ out2,
                               long long Mtot3, int Cc, long long Mrow3, int H, int W,
                               const int32_t* __restrict__ exp_ptr, int exp_const,
                               long long comp_stride, int nplanes) {
  __shared__ float tile[5][66][33];                // tile[c][j] = triple-cell m0 - 1 + j
  const int e = exp_ptr ? exp_ptr[0] : exp_const;
  const float sc2e = __int_as_float((127 + e) << 23);
  const long long m0 = (long long)blockIdx.x * 64;
  const int c0 = blockIdx.y * 32;
  const int tid = threadIdx.x;
  const int lc = (tid & 7) * 4, lj = tid >> 3;
  const int H3 = H / 3;
  const bool vec4 = (Cc & 3) == 0;
#pragma unroll
  for (int pass = 0; pass < 3; ++pass) {
    const int j = pass * 32 + lj;
    if (j < 66) {
      const long long m3 = m0 - 1 + j;
      f32x4 d[5];
#pragma unroll
      for (int i = 0; i < 5; ++i) d[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (m3 >= 0 && m3 < Mtot3 && c0 + lc < Cc) {
        const long long q = m3 / W;
        const int x = (int)(m3 - q * W);
        const int t3 = (int)(q % H3) * 3;          // first image row of the triple
#pragma unroll
        for (int i = 0; i < 5; ++i) {
          const int y = t3 - 1 + i;
          if (y >= 0 && y < H) {
            const float* src = in + (size_t)((3 * q - 1 + i) * W + x) * Cc + c0 + lc;
            if (vec4) d[i] = *reinterpret_cast<const f32x4*>(src);
            else
              for (int k = 0; k < 4; ++k)
                if (c0 + lc + k < Cc) d[i][k] = src[k];
          }
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float v3 = d[3][k] - d[1][k];
        const float d32 = d[3][k] - d[2][k];
        tile[0][j][lc + k] = 2.0f * (d[0][k] - d[2][k]) + v3;
        tile[1][j][lc + k] = d32 - 2.0f * d[1][k];
        tile[2][j][lc + k] = 2.0f * (d[1][k] - d[2][k]) + d32;
        tile[3][j][lc + k] = v3;
        tile[4][j][lc + k] = 2.0f * v3 + (d[2][k] - d[4][k]);
      }
    }
  }
  __syncthreads();
  const int ch = tid >> 3, grp = tid & 7;
  if (c0 + ch >= Cc) return;
  const size_t o = wg16_plane_index(m0 + grp * 8, c0 + ch, Cc);
#pragma unroll
  for (int comp = 0; comp < 5; ++comp) {
#pragma unroll
    for (int dd = 0; dd < 3; ++dd) {
      _Float16* const out = (dd == 0 ? out0 : (dd == 1 ? out1 : out2)) + (size_t)comp * (size_t)comp_stride;
      const int dx = dd - 1;
      f16x8 p0, p1;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const long long m = m0 + grp * 8 + q;
        const int x = (int)(m % W);
        const bool ok = (m < Mtot3) & (x + dx >= 0) & (x + dx < W);
        const float sv = ok ? tile[comp][grp * 8 + q + 1 + dx][ch] * sc2e : 0.f;
        const _Float16 h0 = (_Float16)sv;
        p0[q] = h0;
        p1[q] = (_Float16)(sv - (float)h0);
      }
      *reinterpret_cast<f16x8*>(out + o) = p0;
      if (nplanes == 2) *reinterpret_cast<f16x8*>(out + (size_t)Cc * Mrow3 + o) = p1;
    }
  }
}

// Sum of the splits' partials [nsplit][15][per] in split order (bitwise reproducible) and the
// output combination of the row-triple form -> dW [9][per] (tap = (dy + 1) * 3 + (dx + 1)).
// per = (Cx + C) x N4; the x rows (input channel < Cx) were written by nsplit_x <= nsplit splits
// (their GEMM balances on its own split count, see wgrad16_x_splits).
__global__ __launch_bounds__(256)
void wgrad_wino3_reduce_kernel(const float* __restrict__ partial, int nsplit, int nsplit_x,
                               size_t x_elems, size_t per, float* __restrict__ out) {
  const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= 3 * per) return;
  const size_t dxi = idx / per, el = idx - dxi * per;
  const int ns = el < x_elems ? nsplit_x : nsplit;
  float m[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
  for (int sp = 0; sp < ns; ++sp)
#pragma unroll
    for (int c = 0; c < 5; ++c) m[c] += partial[((size_t)sp * 15 + c * 3 + dxi) * per + el];
  out[(0 * 3 + dxi) * per + el] = ((m[0] + m[1]) + m[2]) + m[3];
  out[(1 * 3 + dxi) * per + el] = (m[1] - m[2]) + 2.0f * m[3];
  out[(2 * 3 + dxi) * per + el] = ((m[1] + m[2]) + 4.0f * m[3]) + m[4];
}

// max |in| as float bits (non-negative floats order like ints); *out zeroed before
__global__ __launch_bounds__(256)
void absmax_bits_kernel(const float* __restrict__ in, size_t n, int32_t* __restrict__ out) {
  __shared__ int32_t red[4];
  int32_t mb = 0;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
    mb = max(mb, __float_as_int(fabsf(in[i])));
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) mb = max(mb, __shfl_xor(mb, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mb;
  __syncthreads();
  if (threadIdx.x == 0) atomicMax(out, max(max(red[0], red[1]), max(red[2], red[3])));
}

// exponent of the chain-wide G scale: e = top - ilogb(max over the chain's steps); top = 13
// leaves the planes below 2^14 (10 for an operand the row-triple form multiplies by up to 6)
__global__ void chain_exp_kernel(const int32_t* __restrict__ gmax, int nsteps, int step_stride,
                                 int32_t* __restrict__ exp_out, int top = 13) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  int mb = 0;
  for (int s = 0; s < nsteps; ++s)
    for (int i = 0; i < 64; ++i) mb = max(mb, gmax[(size_t)s * step_stride + i]);
  const float mx = __int_as_float(mb);
  int e = 0;
  if (mx > 0.f) {
    e = top - ilogbf(mx);
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
  }
  exp_out[0] = e;
}

struct Wgrad16Args {
  const _Float16* at[3];     // A operand planes [2][Ca][Mrow], dx = -1, 0, +1
  const _Float16* gt;
  float* partial;            // [nsplit][9][Cx + C][4C]
  const int32_t* g_exp;      // scale exponents of G and of the A operand
  const int32_t* a_exp;
  long long Mrow;            // cells per operand row (multiple of 64)
  int32_t ksteps_total;      // Mtot / 16
  int32_t H, W, Cx, C;
  int32_t Ca;                // channels of the A operand: C (h rows) or Cx (x rows)
  int32_t ksteps_per_split;  // even
  int32_t nsplit;            // multiple of 8 (split -> XCD), or the wide kernel's own count
  int32_t map_mode;          // wide kernel: tiles -> XCDs (0 by group, 1 by split, 2 mixed)
  // Winograd F(3,3) form over image-row triples (see "the row-triple form" below): ntaps = 15,
  // tap = component * 3 + (dx + 1), no row shift; cells are triple-cells (H = triples per image),
  // the operands hold five components each, *_comp_stride halves apart.  0 / 9 = the direct form.
  int32_t ntaps;
  int64_t a_comp_stride, g_comp_stride;
};
__host__ __device__ __forceinline__ int wg16_ntaps(const Wgrad16Args& a) { return a.ntaps == 15 ? 15 : 9; }

constexpr int kWg16Pitch = 40;                         // halves per LDS row (32 cells + pad)
constexpr int kWg16Tile = 2 * 128 * kWg16Pitch;        // halves per operand tile (2 planes)
// NP = MFMAs per product: 3 = the f16x3 split (both planes of both operands); 1 = the leading
// fp16 plane of each operand only -- 11 significand bits, fp32 accumulate: the backward of
// the reduced-precision compute mode (BASELINE.json configs[4]; the lower planes are neither
// loaded nor staged, a tile is half the LDS).
template <int NP> constexpr int wg16_tile() { return (NP == 1 ? 1 : 2) * 128 * kWg16Pitch; }

// XROWS = false: h rows.  A workgroup's 128 tile rows are 128 channels of ONE tap,
//   so "tap row outside the image" is uniform per k-step and its MFMAs are skipped.
// XROWS = true: x rows (Ca = Cx channels, any width).  Tile rows enumerate
//   (tap, channel) pairs, R = tap * Cx + ci < 9 Cx, so each row has its own operand
//   copy / row shift and an invalid (row, k-step) is staged as zeros instead.
template <bool XROWS, int NP = 3>
__global__ __launch_bounds__(256, 2)
void convlstm_wgrad_f16x3_kernel(const Wgrad16Args a) {
  extern __shared__ __attribute__((aligned(16))) _Float16 lds_raw[];   // [buffer][A | G][tile]
  constexpr int kTile = wg16_tile<NP>();
  constexpr int NQ = NP == 1 ? 2 : 4;                  // copy slots per thread and operand
  constexpr int NPL = NP == 1 ? 1 : 2;                 // planes staged
  _Float16 (*lds)[2][kTile] = reinterpret_cast<_Float16 (*)[2][kTile]>(lds_raw);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1;
  const int H = a.H, W = a.W, C = a.C, N4 = 4 * C, Ca = a.Ca;
  const int wk = W / 16;                               // k-steps per image row
  const int nnb = N4 / 128;
  const int NT = wg16_ntaps(a);
  const bool wn = NT == 15;
  // x rows of the row-triple form: a tile's rows are (dx, channel) pairs of ONE component (the G
  // tile is that component's), nrbc tiles per component
  const int nrbc = (3 * Ca + 127) / 128;
  const int nrb = XROWS ? (wn ? 5 * nrbc : (9 * Ca + 127) / 128) : NT * (C / 128);   // row blocks
  const int tiles_per_split = nrb * nnb;
  const int xcd = blockIdx.x & 7;                      // split -> XCD (see convlstm_wgrad.h)
  int j = blockIdx.x >> 3;
  int split = xcd + 8 * (j / tiles_per_split);
  j = j % tiles_per_split;
  if (a.nsplit & 7) {                                  // a split count of the wide kernel: plain order
    split = (int)blockIdx.x / tiles_per_split;
    j = (int)blockIdx.x % tiles_per_split;
  }
  const int nb = j % nnb;
  const int rb = j / nnb;
  const int n0 = nb * 128;
  // h rows: rb = tap * (C/128) + cib
  const int tap_u = XROWS ? 0 : rb / (C / 128);
  const int ci0_u = XROWS ? 0 : (rb - tap_u * (C / 128)) * 128;
  const int dy_u = wn ? 0 : tap_u / 3 - 1;
  const int comp_u = wn ? (XROWS ? rb / nrbc : tap_u / 3) : 0;     // the workgroup's component
  const int rbi = (XROWS && wn) ? rb - comp_u * nrbc : rb;
  const size_t aco = (size_t)comp_u * (size_t)a.a_comp_stride, gco = (size_t)comp_u * (size_t)a.g_comp_stride;
  const long long Mrow = a.Mrow;

  const int ks0 = split * a.ksteps_per_split;
  int ks1 = ks0 + a.ksteps_per_split;
  if (ks1 > a.ksteps_total) ks1 = a.ksteps_total;
  const int nstages = ks1 > ks0 ? (ks1 - ks0 + 1) / 2 : 0;

  f32x16 acc[2][2];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[x][y][i] = 0.f;

  // per-thread copy slots: q -> plane q>>1, tile row (q&1)*64 + tid>>2, 8-cell vector tid&3
  const int vec = tid & 3;
  const _Float16* ap[NQ];
  const _Float16* gp[NQ];
  int dyq[NQ];                                         // XROWS: row shift of slot q (huge = dead row)
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int plane = q >> 1, row = (q & 1) * 64 + (tid >> 2);
    gp[q] = a.gt + gco + (size_t)plane * N4 * Mrow + (size_t)(n0 + row) * 32 + vec * 8;
    if (XROWS) {
      const int R = rbi * 128 + row;
      const bool live = R < (wn ? 3 : 9) * Ca;
      const int tap = live ? R / Ca : (wn ? 1 : 4);
      const int ci = live ? R - tap * Ca : 0;
      dyq[q] = live ? (wn ? 0 : tap / 3 - 1) : (1 << 20);
      ap[q] = a.at[tap - (tap / 3) * 3] + aco + (size_t)plane * Ca * Mrow + (size_t)ci * 32;
    } else {
      dyq[q] = dy_u;
      ap[q] = a.at[tap_u - (tap_u / 3) * 3] + aco + (size_t)plane * Ca * Mrow + (size_t)(ci0_u + row) * 32;
    }
  }

  // image row of the two k-steps of the stage being LOADED (uniform, incremental)
  int ly, lxk;
  {
    const int kin = ks0 % (H * wk);
    ly = kin / wk; lxk = kin - ly * wk;
  }
  auto advance = [&](int& y, int& xk) { if (++xk == wk) { xk = 0; if (++y == H) y = 0; } };

  // Two register sets of staged operands: the loads of stage st + 2 are issued while stage st
  // multiplies and stage st + 1 waits in the other set -- two stages (~50 MFMAs per wave) of
  // lead.  With ONE set (rounds 1-3: loads issued at the top of a stage, written to LDS at its
  // bottom) the lead was 24 MFMAs = 770-1500 cycles against an HBM round trip of 2-4 k cycles
  // under this kernel's streaming load (both operands are read once, nothing is reused across
  // the reduction): MFMA busy 0.425 at 1.7 waves / SIMD, LDS and L2 far from saturated.  The
  // workgroup count per CU is set by LDS (2 x 80 KB), so the 32 extra VGPRs are free.
  f16x8 sa[NQ], sg[NQ], sb[NQ], sgb[NQ];
  bool cv0 = false, cv1 = false;                       // h rows: k-step validity, stage in LDS
  auto stage_load = [&](int st, f16x8 (&ra)[NQ], f16x8 (&rg)[NQ], bool& nv0, bool& nv1) {
    const int ksb = ks0 + 2 * st;
    const int y0 = ly; advance(ly, lxk);
    const int y1 = ly; advance(ly, lxk);
    const bool in0 = ksb < ks1, in1 = ksb + 1 < ks1;
    const int m0 = ksb * 16;                           // stage = one 32-cell block
    const size_t gofs = (size_t)m0 * N4;
    auto aofs = [&](int cell) -> size_t {              // cell is a multiple of 8
      return (size_t)(cell >> 5) * ((size_t)Ca * 32) + (size_t)(cell & 31);
    };
    const int ysel = (vec & 2) ? y1 : y0;
    const bool insel = (vec & 2) ? in1 : in0;
    if (!XROWS) {
      nv0 = in0 & ((unsigned)(y0 + dy_u) < (unsigned)H);
      nv1 = in1 & ((unsigned)(y1 + dy_u) < (unsigned)H);
      const int sh = ((vec & 2) ? nv1 : nv0) ? dy_u * W : 0;   // skipped k-step: unshifted
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        ra[q] = *reinterpret_cast<const f16x8*>(ap[q] + aofs(m0 + vec * 8 + sh));
        rg[q] = *reinterpret_cast<const f16x8*>(gp[q] + gofs);
      }
    } else {
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        const bool ok = insel & ((unsigned)(ysel + dyq[q]) < (unsigned)H);
        f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (ok) v = *reinterpret_cast<const f16x8*>(ap[q] + aofs(m0 + vec * 8 + dyq[q] * W));
        ra[q] = v;
        rg[q] = *reinterpret_cast<const f16x8*>(gp[q] + gofs);
      }
      nv0 = in0; nv1 = in1;
    }
  };
  auto stage_store = [&](int buf, const f16x8 (&ra)[NQ], const f16x8 (&rg)[NQ]) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int plane = q >> 1, row = (q & 1) * 64 + (tid >> 2);
      const int o = (plane * 128 + row) * kWg16Pitch + vec * 8;
      *reinterpret_cast<f16x8*>(&lds[buf][0][o]) = ra[q];
      *reinterpret_cast<f16x8*>(&lds[buf][1][o]) = rg[q];
    }
  };
  const int li = lane & 31, k8 = (lane >> 5) * 8;
  auto stage_mfma = [&](int buf) {
    const _Float16* A = &lds[buf][0][0];
    const _Float16* G = &lds[buf][1][0];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (kk ? cv1 : cv0) {                   // uniform
        f16x8 fa[2][NPL], fg[2][NPL];         // [sub-block][plane]
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) {
            fa[s2][pl] = *reinterpret_cast<const f16x8*>(
                A + (pl * 128 + wi * 64 + s2 * 32 + li) * kWg16Pitch + kk * 16 + k8);
            fg[s2][pl] = *reinterpret_cast<const f16x8*>(
                G + (pl * 128 + wj * 64 + s2 * 32 + li) * kWg16Pitch + kk * 16 + k8);
          }
        if constexpr (NP == 3) {
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[x][NPL - 1], fg[y][0], acc[x][y], 0, 0, 0);
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[x][0], fg[y][NPL - 1], acc[x][y], 0, 0, 0);
        }
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int y = 0; y < 2; ++y)
            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[x][0], fg[y][0], acc[x][y], 0, 0, 0);
      }
    }
  };

  if (nstages > 0) {
    bool av0 = false, av1 = false, bv0 = false, bv1 = false;   // validity of the staged sets
    stage_load(0, sa, sg, av0, av1);
    stage_store(0, sa, sg);
    cv0 = av0; cv1 = av1;
    if (nstages > 1) stage_load(1, sa, sg, av0, av1);          // set A: stage 1, in flight
    __syncthreads();
    // iteration st: the set holding stage st + 1 goes to LDS after the MFMAs, the other set
    // takes the loads of stage st + 2
    for (int st = 0; st < nstages; st += 2) {
      // even step: stage st + 1 is in set A, stage st + 2 goes to set B
      if (st + 2 < nstages) stage_load(st + 2, sb, sgb, bv0, bv1);
      stage_mfma(st & 1);
      if (st + 1 < nstages) { stage_store((st + 1) & 1, sa, sg); cv0 = av0; cv1 = av1; }
      __syncthreads();
      if (st + 1 >= nstages) break;
      // odd step: stage st + 2 is in set B, stage st + 3 goes to set A
      if (st + 3 < nstages) stage_load(st + 3, sa, sg, av0, av1);
      stage_mfma((st + 1) & 1);
      if (st + 2 < nstages) { stage_store((st + 2) & 1, sb, sgb); cv0 = bv0; cv1 = bv1; }
      __syncthreads();
    }
  }

  // D: col j = lane&31 (n), row i = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const float scale = ldexpf(1.0f, -(a.a_exp[0] + a.g_exp[0]));
  const int Cin = a.Cx + C;
  float* ps = a.partial + (size_t)split * NT * Cin * N4;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 2; ++y)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        const int r = wi * 64 + x * 32 + i;
        const int n = n0 + wj * 64 + y * 32 + (lane & 31);
        if (XROWS) {
          const int R = rbi * 128 + r;
          if (R < (wn ? 3 : 9) * Ca) {
            const int tq = R / Ca, ci = R - tq * Ca;
            const int tap = comp_u * 3 + tq;           // direct form: comp_u == 0
            ps[((size_t)tap * Cin + ci) * N4 + n] = acc[x][y][reg] * scale;
          }
        } else {
          ps[((size_t)tap_u * Cin + a.Cx + ci0_u + r) * N4 + n] = acc[x][y][reg] * scale;
        }
      }
}

// ---------------------------------------------------------------- the wide tile (round 5)
// What limits the kernel above (MFMA busy 0.445 under no power cap) is not the matrix pipe but
// the three paths that feed it, all near their rate at once.  Per workgroup and stage (32 cells,
// 96 MFMAs) it moves 32 KB global -> VGPR (vector L1: 64 B / clk / CU), stores them with 8
// ds_write_b128 per lane (the wide stores run at ~79 B / clk / CU, MI355X guide "LDS": 13 cycles
// per wave instruction) and reads 64 fragments back: per CU (two workgroups) 1 024 + 830 + 512
// cycles beside 1 536 cycles of MFMAs.
// Here a workgroup owns 128 input channels x 256 gate columns of one tap and a wave 64 x 128
// (2 x 4 accumulators): 48 KB per 192 MFMAs -- 0.25 KB of staging per MFMA instead of 0.33, 12
// fragment reads per 24 MFMAs instead of 8 per 12 -- in ONE 48 KB LDS stage: two workgroups
// per CU, whose waves fill each other's store / barrier phases (a stage is: loads of the next
// stage -> MFMAs -> barrier -> stores -> barrier).
// LDS rows are the 64-byte cell runs of the planes, unpadded; the 16-byte chunk c of row r sits
// at position c ^ s(r), s(r) = ((r >> 1) & 3) ^ ((r >> 3) & 1): the 16-lane groups of
// ds_read_b128 ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}: rows = lanes, one chunk) then touch
// 16 different 16-byte slots of the 256-byte bank row, and the 8-lane groups of ds_write_b128
// two whole rows (tests/test_wgrad_wide_model.py: the swizzle, the block decode and the whole
// index arithmetic restated in numpy against the definition of the gradient).
// Tiles -> XCDs by (channel block, column block), not by split: any split count balances
// (9 taps x nsplit workgroups per XCD and group on 64 slots: 7 or 14 splits), every workgroup
// of an XCD streams the same G columns and the same A rows through its L2.
// h rows, C % 256 == 0 (NP = 3: both planes; NP = 1: the leading planes, half the stage);
// everything else stays on the kernel above.
constexpr int kWgwA = 128, kWgwG = 256;
constexpr int kWgwLdsHalves = 2 * (kWgwA + kWgwG) * 32;          // 24 576 halves = 48 KB
__host__ __device__ __forceinline__ constexpr int wgw_swz(int row) {
  return ((row >> 1) & 3) ^ ((row >> 3) & 1);
}

// NP = 3: both planes (f16x3); 1: the leading plane only (compute mode 2).
// XROWS: the x rows -- tile rows enumerate (tap, channel) pairs R = tap * Cx + ci < 9 Cx, so each
// copy slot has its own operand copy / row shift and an invalid (row, k-step) is staged as zeros
// (the kernel above, same rule); tiles in plain order (5 x 4 per split at Cx = 64).
template <int NP, bool XROWS = false>
__global__ __launch_bounds__(256, 2)
void convlstm_wgrad_f16x3_wide_kernel(const Wgrad16Args a) {
  constexpr int NPL = NP == 1 ? 1 : 2;
  __shared__ __attribute__((aligned(16))) _Float16 lds[kWgwLdsHalves / 2 * NPL];
  _Float16* const ldsA = lds;                          // [plane][128 rows][4 chunks][8]
  _Float16* const ldsG = lds + NPL * kWgwA * 32;       // [plane][256 rows][4 chunks][8]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wi = wave >> 1, wj = wave & 1;
  const int H = a.H, W = a.W, C = a.C, N4 = 4 * C, Ca = a.Ca;
  const int wk = W / 16;
  const int ncib = C / kWgwA, nnb = N4 / kWgwG;
  const int xcd = blockIdx.x & 7;
  const int NT = wg16_ntaps(a);
  const bool wn = NT == 15;
  const int nrbc = (3 * Ca + kWgwA - 1) / kWgwA;     // x rows, row-triple form: tiles per component
  int j = blockIdx.x >> 3;
  int tap = wn ? 1 : 4, split, cib = 0, nb, rb = 0;
  if (XROWS) {
    const int nrb = wn ? 5 * nrbc : (9 * Ca + kWgwA - 1) / kWgwA;
    j = blockIdx.x;
    nb = j % nnb; j /= nnb;
    rb = j % nrb;
    split = j / nrb;
    if (wn) { tap = (rb / nrbc) * 3 + 1; rb -= (rb / nrbc) * nrbc; }   // tap / 3 = the component
  } else if (a.map_mode == 1) {             // split -> XCD: every operand byte through ONE L2
    const int tps = NT * ncib * nnb;
    split = xcd + 8 * (j / tps);
    j %= tps;
    tap = j % NT; j /= NT;
    cib = j % ncib; nb = j / ncib;
  } else if (a.map_mode == 2) {      // (channel block, split mod 4) -> XCD; ncib == 2
    const int tps = NT * nnb;
    cib = xcd & 1;
    split = (xcd >> 1) + 4 * (j / tps);
    j %= tps;
    tap = j % NT; nb = j / NT;
  } else {                           // (channel block, column block) -> XCD
    const int gpx = (ncib * nnb) >> 3;
    const int group = xcd + 8 * (j % gpx);
    j /= gpx;
    tap = j % NT;
    split = j / NT;
    cib = group % ncib; nb = group / ncib;
  }
  const int ci0 = cib * kWgwA;
  const int n0 = nb * kWgwG;
  const int dy = wn ? 0 : tap / 3 - 1;
  const int comp = wn ? tap / 3 : 0;
  const size_t aco = (size_t)comp * (size_t)a.a_comp_stride, gco = (size_t)comp * (size_t)a.g_comp_stride;
  const long long Mrow = a.Mrow;

  const int ks0 = split * a.ksteps_per_split;
  int ks1 = ks0 + a.ksteps_per_split;
  if (ks1 > a.ksteps_total) ks1 = a.ksteps_total;
  const int nstages = ks1 > ks0 ? (ks1 - ks0 + 1) / 2 : 0;

  f32x16 acc[2][4];
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[x][y][i] = 0.f;

  // copy slots of a thread: row (tid >> 2) + 64 k of an operand plane, 8-cell chunk tid & 3
  const int vec = tid & 3, trow = tid >> 2;
  const _Float16* abase[2];                            // slot k: tile row trow + 64 k
  int dyq[2];                                          // XROWS: its row shift (huge = dead row)
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    if (XROWS) {
      const int R = rb * kWgwA + k * 64 + trow;
      const bool live = R < (wn ? 3 : 9) * Ca;
      const int tp = live ? R / Ca : (wn ? 1 : 4);
      const int ci = live ? R - tp * Ca : 0;
      dyq[k] = live ? (wn ? 0 : tp / 3 - 1) : (1 << 20);
      abase[k] = a.at[tp - (tp / 3) * 3] + aco + (size_t)ci * 32;
    } else {
      dyq[k] = dy;
      abase[k] = a.at[tap - (tap / 3) * 3] + aco + (size_t)(ci0 + k * 64 + trow) * 32;
    }
  }
  const _Float16* const gbase = a.gt + gco + (size_t)(n0 + trow) * 32 + vec * 8;
  const size_t apl = (size_t)Ca * Mrow, gpl = (size_t)N4 * Mrow;    // plane strides
  const int wslot = (trow * 4 + (vec ^ wgw_swz(trow))) * 8;         // halves; + 64 rows: + 2048
  static_assert(wgw_swz(64) == 0 && wgw_swz(5 + 64) == wgw_swz(5), "row + 64 keeps its swizzle");

  int ly, lxk;
  {
    const int kin = ks0 % (H * wk);
    ly = kin / wk; lxk = kin - ly * wk;
  }
  auto advance = [&](int& y, int& xk) { if (++xk == wk) { xk = 0; if (++y == H) y = 0; } };

  f16x8 sa[2 * NPL], sg[4 * NPL];                      // [plane * 2 + k], [plane * 4 + k]
  bool cv0 = false, cv1 = false, nv0 = false, nv1 = false;
  auto stage_load = [&](int st) {
    const int ksb = ks0 + 2 * st;
    const int y0 = ly; advance(ly, lxk);
    const int y1 = ly; advance(ly, lxk);
    const bool in0 = ksb < ks1, in1 = ksb + 1 < ks1;
    nv0 = in0 & (XROWS || (unsigned)(y0 + dy) < (unsigned)H);
    nv1 = in1 & (XROWS || (unsigned)(y1 + dy) < (unsigned)H);
    const int m0 = ksb * 16;                           // one 32-cell block
    const size_t go = (size_t)m0 * N4;
    const int ysel = (vec & 2) ? y1 : y0;
    const bool insel = (vec & 2) ? in1 : in0;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      // h rows: a skipped k-step loads unshifted (never multiplied); x rows: zeros
      const bool ok = XROWS ? (insel & ((unsigned)(ysel + dyq[k]) < (unsigned)H))
                            : ((vec & 2) ? nv1 : nv0);
      const int cell = m0 + vec * 8 + (ok ? dyq[k] * W : 0);
      const size_t ao = (size_t)(cell >> 5) * ((size_t)Ca * 32) + (size_t)(cell & 31);
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        f16x8 v = {0, 0, 0, 0, 0, 0, 0, 0};
        if (!XROWS || ok) v = *reinterpret_cast<const f16x8*>(abase[k] + pl * apl + ao);
        sa[pl * 2 + k] = v;
      }
    }
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
      for (int k = 0; k < 4; ++k)
        sg[pl * 4 + k] = *reinterpret_cast<const f16x8*>(gbase + pl * gpl + go + k * 2048);
  };
  auto stage_store = [&]() {
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
#pragma unroll
      for (int k = 0; k < 2; ++k)
        *reinterpret_cast<f16x8*>(ldsA + pl * (kWgwA * 32) + k * 2048 + wslot) = sa[pl * 2 + k];
#pragma unroll
      for (int k = 0; k < 4; ++k)
        *reinterpret_cast<f16x8*>(ldsG + pl * (kWgwG * 32) + k * 2048 + wslot) = sg[pl * 4 + k];
    }
  };
  // fragment of lane (li, half) for k-step kk: chunk kk * 2 + half of row base + li
  const int li = lane & 31, hf = lane >> 5;
  const int rsw = wgw_swz(li);                         // row bases are multiples of 32
  auto stage_mfma = [&]() {
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
      if (kk ? cv1 : cv0) {                            // uniform
        const int co = (li * 4 + ((kk * 2 + hf) ^ rsw)) * 8;
        f16x8 fa[2][NPL];
#pragma unroll
        for (int x = 0; x < 2; ++x)
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl)
            fa[x][pl] = *reinterpret_cast<const f16x8*>(
                ldsA + pl * (kWgwA * 32) + (wi * 64 + x * 32) * 32 + co);
#pragma unroll
        for (int y = 0; y < 4; ++y) {
          f16x8 fg[NPL];
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl)
            fg[pl] = *reinterpret_cast<const f16x8*>(
                ldsG + pl * (kWgwG * 32) + (wj * 128 + y * 32) * 32 + co);
#pragma unroll
          for (int x = 0; x < 2; ++x) {
            if constexpr (NP == 3) {
              acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[x][NPL - 1], fg[0], acc[x][y], 0, 0, 0);
              acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[x][0], fg[NPL - 1], acc[x][y], 0, 0, 0);
            }
            acc[x][y] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[x][0], fg[0], acc[x][y], 0, 0, 0);
          }
        }
      }
    }
  };

  if (nstages > 0) {
    stage_load(0);
    stage_store();
    cv0 = nv0; cv1 = nv1;
    __syncthreads();
    for (int st = 0; st < nstages; ++st) {
      const bool more = st + 1 < nstages;
      if (more) stage_load(st + 1);
      __builtin_amdgcn_sched_barrier(0);               // the loads stay ahead of the MFMAs
      stage_mfma();
      __syncthreads();                                 // every wave is done reading the stage
      if (more) { stage_store(); cv0 = nv0; cv1 = nv1; }
      __syncthreads();
    }
  }

  // D: col j = lane&31 (n), row i = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const float scale = ldexpf(1.0f, -(a.a_exp[0] + a.g_exp[0]));
  const int Cin = a.Cx + C;
  float* ps = a.partial + (size_t)split * NT * Cin * N4;
#pragma unroll
  for (int x = 0; x < 2; ++x)
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
      for (int reg = 0; reg < 16; ++reg) {
        const int i = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
        const int r = wi * 64 + x * 32 + i;
        const int n = n0 + wj * 128 + y * 32 + (lane & 31);
        if (XROWS) {
          const int R = rb * kWgwA + r;
          if (R < (wn ? 3 : 9) * Ca) {
            const int tq = R / Ca, ci = R - tq * Ca;
            const int tp = comp * 3 + tq;              // direct form: comp == 0
            ps[((size_t)tp * Cin + ci) * N4 + n] = acc[x][y][reg] * scale;
          }
        } else {
          ps[((size_t)tap * Cin + a.Cx + ci0 + r) * N4 + n] = acc[x][y][reg] * scale;
        }
      }
}

static inline bool wgrad16_wide_ok(int W, int C) {
  static const bool on = !(getenv("MV_WGRAD_WIDE") && atoi(getenv("MV_WGRAD_WIDE")) == 0);
  return on && (W % 16) == 0 && (C % 256) == 0;
}
// Split count of the wide kernel: 9 taps x nsplit workgroups per XCD and (channel, column) group
// on 64 slots -- 7 splits = one full round, 14 = two, 21 = three -- never more than the planned count
// (the partial buffer), and at least ~64 k-steps per split.
static inline int wgrad16_wide_map_mode(int C) {
  static const int m = getenv("MV_WGRAD_WIDE_MAP") ? atoi(getenv("MV_WGRAD_WIDE_MAP")) : 0;
  if (m == 2 && C != 2 * kWgwA) return 0;
  return m >= 0 && m <= 2 ? m : 0;
}
static inline int wgrad16_wide_splits(long long Mtot, int planned, int map_mode) {
  int v = 0;
  if (const char* ev = getenv("MV_WGRAD_WIDE_SPLITS")) v = atoi(ev);
  const int mult = map_mode == 1 ? 8 : (map_mode == 2 ? 4 : 1);
  if (v >= 1 && v <= planned && v % mult == 0) return v;
  if (map_mode == 1) return planned;                 // a multiple of 8 by construction
  if (map_mode == 2) return planned >= 20 ? 20 : (planned / 4) * 4;
  // measured (training step configs[2], profiles/r5wg_*): the wide kernel takes the same time
  // at 7, 14 and 21 splits; the x rows kernel, which follows the count, is fastest at 21
  const long long ks = Mtot / 16;
  if (planned >= 21 && ks >= 21 * 64) return 21;
  if (planned >= 14 && ks >= 14 * 64) return 14;
  return planned >= 7 ? 7 : planned;
}
static inline unsigned wgrad16_wide_blocks(const Wgrad16Args& a, bool xrows = false) {
  const bool wn = a.ntaps == 15;
  if (xrows)
    return (unsigned)a.nsplit *
           (unsigned)(wn ? 5 * ((3 * a.Ca + kWgwA - 1) / kWgwA) : (9 * a.Ca + kWgwA - 1) / kWgwA) *
           (unsigned)(4 * a.C / kWgwG);
  return (unsigned)a.nsplit * (unsigned)wg16_ntaps(a) * (unsigned)(a.C / kWgwA) *
         (unsigned)(4 * a.C / kWgwG);
}
// The x rows on the wide tile measured the same as on the 128 x 128 tile (2.59 vs 2.56 ms per
// training step, profiles/r5wg_wgrad_wide_tile_ab.md: 20 fat workgroups per split fill the chip
// worse than 40 small ones, which eats the tile's gain): off unless MV_WGRAD_WIDE_X=1.
static inline bool wgrad16_wide_x_enabled() {
  static const bool on = getenv("MV_WGRAD_WIDE_X") && atoi(getenv("MV_WGRAD_WIDE_X")) == 1;
  return on;
}

constexpr size_t kWg16LdsBytes = (size_t)2 * 2 * kWg16Tile * sizeof(_Float16);   // 80 KB
constexpr size_t kWg16LdsBytes1 = (size_t)2 * 2 * wg16_tile<1>() * sizeof(_Float16);   // NP = 1: 40 KB

static inline bool wgrad16_ok(int W, int C) { return (W % 16) == 0 && (C % 128) == 0; }
// the row-triple form: whole triples only; MV_WGRAD_WINO=0 keeps the direct form
// (one_plane: the reduced-precision mode's single fp16 plane per operand, MV_WGRAD_WINO_BF16=0)
static inline bool wgrad16_wino3_ok(int H, bool one_plane) {
  static const bool on = !(getenv("MV_WGRAD_WINO") && atoi(getenv("MV_WGRAD_WINO")) == 0);
  static const bool on1 = !(getenv("MV_WGRAD_WINO_BF16") && atoi(getenv("MV_WGRAD_WINO_BF16")) == 0);
  return on && (on1 || !one_plane) && H >= 3 && H % 3 == 0;
}

static inline void wgrad16_plan(Wgrad16Args& a, long long Mtot, int nsplit) {
  a.ksteps_total = (int32_t)(Mtot / 16);
  int per = (a.ksteps_total + nsplit - 1) / nsplit;
  per = (per + 1) & ~1;
  a.ksteps_per_split = per;
  a.nsplit = nsplit;
}

// Split count of the x rows in the row-triple form.  Their 128 x 128 tiles number 40 or 80 per
// split (5 components x 1 or 2 row tiles x 8 column blocks) on 512 workgroup slots: 12 splits =
// 480 / 960 workgroups fill one / two rounds to 94 % (the h rows' 21: 1.64 / 3.28 rounds).
// Measured: 1.94 against 2.17 - 2.28 ms per training step (profiles/r6t, MV_WGRAD_WIDE_SPLITS
// sweep).  MV_WGRAD_X_SPLITS overrides; never more than the h rows' count (the partial buffer).
static inline int wgrad16_x_splits(long long Mgemm, int nsplit_h) {
  int v = 12;
  if (const char* ev = getenv("MV_WGRAD_X_SPLITS")) v = atoi(ev);
  if (v < 1 || v > nsplit_h || Mgemm / 16 < (long long)v * 32) return nsplit_h;
  return v;
}

static inline unsigned wgrad16_blocks(const Wgrad16Args& a, bool xrows) {
  const bool wn = a.ntaps == 15;
  const unsigned nrb = xrows ? (unsigned)(wn ? 5 * ((3 * a.Ca + 127) / 128) : (9 * a.Ca + 127) / 128)
                             : (unsigned)wg16_ntaps(a) * (unsigned)(a.C / 128);
  return (unsigned)a.nsplit * nrb * (unsigned)(4 * a.C / 128);
}

}  // namespace mv
