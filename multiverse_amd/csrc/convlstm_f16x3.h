// ConvLSTM step with the gate convolution on the fp16 matrix pipe at fp32
// accuracy ("f16x3"): every fp32 operand v is carried as two fp16 planes of the
// pre-scaled value 256 v,
//     v0 = half(256 v),  v1 = half(256 v - v0)      (|256 v - v0 - v1| <= 2^-22 |256 v|)
// and every product a*w is formed as  a0 w0 + a0 w1 + a1 w0  by three
// v_mfma_f32_32x32x16_f16 accumulating in fp32 (fp16 x fp16 products are exact in
// fp32); the dropped a1 w1 term and the plane residuals are ~2^-22 relative, i.e.
// fp32-roundoff class.  The accumulators carry 2^16 x the pre-activation and are
// rescaled (exactly) in the epilogue.  Measured against fp64 on the full forward
// the error equals the fp32-MFMA path's own; the parity
// bars (argmax bit-exact, 1e-4 on logits / offsets) are asserted for this path
// by the same GPU tests.
//
// Why: the fp32 MFMA (32x32x2) peaks at 157 TFLOP/s, the fp16 MFMA (32x32x16)
// at 2.5 PFLOP/s; three fp16 MFMAs per product put the ceiling of this
// MFMA-bound sweep at 2.5 PF / 3 = 833 "fp32-equivalent" TFLOP/s.
//
// Same tile as the fp32 kernel (convlstm_mfma.h): a wave owns 32 cells x 128
// columns = the four gates of one block of 32 channels, so the LSTM update still
// runs in the accumulator registers.  One k-step is 16 input channels of one tap:
// A = 2 planes x 16 B of the lane's cell (register-direct from the tiled operand
// planes, plane_layout.h), B = 2 planes x 4 gates x 16 B in fragment order, shared by
// the eight waves of a workgroup through an LDS double buffer filled by LDS-DMA; 12 MFMAs.
// The same body serves the forward step (LSTM epilogue, optional gate activations for
// training, optional h' operand planes, optional sparse-x table terms), dgrad (store
// epilogue, split-K) and, with ONE bf16 plane per operand, the bf16 mode.
// The regression encoder's 2-channel pixel-offset input (|x| up to 1920, outside
// the scaled fp16 range) keeps its single fp32 chunk on v_mfma_f32_32x32x2_f32,
// with weights pre-scaled by 2^16 so that it lands in the same accumulators.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "convlstm_mfma.h"
#include "plane_layout.h"

namespace mv {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// float -> bf16 bits, round to nearest even (operands of the bf16 mode)
__host__ __device__ __forceinline__ uint16_t bf16_bits(float v) {
  uint32_t u = __builtin_bit_cast(uint32_t, v);
  if ((u & 0x7f800000u) == 0x7f800000u) return (uint16_t)(u >> 16);   // inf / nan: truncate
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}
__host__ __device__ __forceinline__ _Float16 bf16_as_half(float v) {
  return __builtin_bit_cast(_Float16, bf16_bits(v));
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

// a pointer the compiler must treat as wave-uniform (SGPR pair): buffer resources built
// from it then need no waterfall loop
template <typename T>
__device__ __forceinline__ T* uniform_ptr(T* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<T*>(((uint64_t)hi << 32) | lo);
}

constexpr int kPlanePad = 16;                       // zero halves in front of every operand plane
constexpr float kF16Scale = 256.0f;                 // per operand
constexpr float kF16Unscale = 1.0f / 65536.0f;      // per product

struct ConvLstm16Args {
  ConvLstmArgs f;              // geometry, fp32 x (x_small only), c, bias, outputs, src rows
  const _Float16* x16;         // [2 planes][rows*H*W*Cx]   (unused when f.x_small)
  const _Float16* h16;         // [2 planes][src rows*H*W*C]
  const _Float16* wp16;        // [cb][kstep][plane][gate][lane][8]
  const float* wx32;           // x_small: fp32 fragment-order chunk [cb][4][4][64][4], x 2^16
  const int32_t* g_exp;        // dgrad: exponent e of the G planes' scale 2^e
  const int32_t* x_exp;        // forward, unbounded activations (relu / lrelu): the x planes hold
                               // 2^e x with a per-tensor e (split_planes_dyn_kernel) instead of
                               // 256 x; the accumulators are rescaled by 2^(8 - e) once the x
                               // k-steps are done (they come first).  null: e = 8
  _Float16* h16_out;           // optional: planes of h' for the next step's h operand
  int64_t h16_out_stride;
  int64_t x_plane_stride;      // elements between the two planes
  int64_t h_plane_stride;
  int32_t n_xk;                // f16 k-steps taken from x (9 * Cx/16; 0 when x_small)
  int32_t n_hk;                // f16 k-steps taken from h (9 * C/16; 0 for the zero state)
  int32_t w_ksteps;            // k-steps per channel block in wp16 (x + all h)
  // dgrad split-K: the channel groups of G are cut into n_kslice ranges, each range
  // of each column block a workgroup of its own; slice s stores its partial sums to
  // part0/part1 + s * rows*H*W*cols, summed afterwards by sum_slices_kernel
  int32_t n_kslice;            // 0/1 = off
  float* part0;
  float* part1;
};

struct ConvLstm16Group {
  ConvLstm16Args p[kMaxGroup];
  int32_t block_end[kMaxGroup];
  int32_t n;
  int32_t map_mode;    // forward step: 0 = column block per XCD, 1 = row tile per XCD
};

static inline int f16x3_xksteps(int Cx) { return (Cx % 16 == 0) ? 9 * (Cx / 16) : 0; }
static inline bool f16x3_cx_supported(int Cx) {
  return Cx == 0 || (Cx % 16) == 0 || 9 * Cx <= kBK;
}
static inline size_t f16x3_wpack_elems(int Cx, int C) {   // in halves
  return (size_t)(C / kChBlock) * (size_t)(f16x3_xksteps(Cx) + 9 * (C / 16)) * 2 * 4 * 64 * 8;
}

// Host-side pack: TF HWIO kernel [3,3,Cx+C,4C] -> wp16.  k-steps: x channel
// groups of 16 (group-major, tap-minor), then h groups; element e of lane l is
// k = 8*(l>>5) + e (the same (half, element) -> k map on the A side, so the
// hardware's pairing of A and B elements is respected whatever its k labels).
static inline void pack_f16x3_weights(const float* w, int Cx, int C, _Float16* out) {
  const int Cin = Cx + C, N4 = 4 * C;
  const int nxk = f16x3_xksteps(Cx), nk = nxk + 9 * (C / 16);
  for (int cb = 0; cb < C / kChBlock; ++cb)
    for (int s = 0; s < nk; ++s) {
      const bool is_x = s < nxk;
      const int q = is_x ? s : s - nxk;
      const int cg = q / 9, tap = q % 9;
      for (int g = 0; g < 4; ++g)
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 8; ++e) {
            const int k = 8 * (l >> 5) + e;
            const int ci = (is_x ? 0 : Cx) + cg * 16 + k;
            const int n = g * C + cb * kChBlock + (l & 31);
            const float v = w[((size_t)tap * Cin + ci) * N4 + n] * kF16Scale;
            const _Float16 v0 = (_Float16)v;
            const _Float16 v1 = (_Float16)(v - (float)v0);
            const size_t base = ((size_t)cb * nk + s) * 2 * 4 * 64 * 8;
            out[base + ((size_t)(0 * 4 + g) * 64 + l) * 8 + e] = v0;
            out[base + ((size_t)(1 * 4 + g) * 64 + l) * 8 + e] = v1;
          }
    }
}

// Device-side twin of pack_f16x3_weights (run after every optimizer step): one
// thread per (cb, k-step, gate, lane, e); writes both planes.
// The packs below are rebuilt from the DEVICE weights after every optimizer step, where no host
// copy exists to range-check (engine_setup.h ensure_packed16 does that for a loaded checkpoint).  A
// scaled value at or beyond the guard (60 000; fp16 overflows to inf at 65 520) sets this flag;
// the training step tests it at its closing synchronisation and fails loudly instead of
// training on infinities (engine_train.h train_apply).  One flag per process.
__device__ int g_pack_overflow;
__device__ __forceinline__ void note_pack_range(double scaled) {
  if (!(fabs(scaled) < 60000.0)) atomicOr(&g_pack_overflow, 1);
}

__global__ void pack_f16x3_kernel(const float* __restrict__ w, _Float16* __restrict__ out,
                                  int Cx_total, int Cx16, int C, size_t total) {
  // Cx_total: x channels in the HWIO kernel; Cx16: x channels packed as f16 k-steps
  // (0 when the x part stays on the fp32 chunk)
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int e = idx & 7;
  const int l = (idx >> 3) & 63;
  const int g = (idx >> 9) & 3;
  const size_t t = idx >> 11;                    // cb * nk + s
  const int nxk = 9 * (Cx16 / 16), nk = nxk + 9 * (C / 16);
  const int s = t % nk, cb = t / nk;
  const bool is_x = s < nxk;
  const int q = is_x ? s : s - nxk;
  const int cg = q / 9, tap = q - cg * 9;
  const int k = 8 * (l >> 5) + e;
  const int ci = (is_x ? 0 : Cx_total) + cg * 16 + k;
  const int n = g * C + cb * kChBlock + (l & 31);
  const int Cin = Cx_total + C, N4 = 4 * C;
  const float v = w[((size_t)tap * Cin + ci) * N4 + n] * kF16Scale;
  note_pack_range(v);
  const _Float16 v0 = (_Float16)v;
  const size_t base = t * (2 * 4 * 64 * 8);
  out[base + ((size_t)(0 * 4 + g) * 64 + l) * 8 + e] = v0;
  out[base + ((size_t)(1 * 4 + g) * 64 + l) * 8 + e] = (_Float16)(v - (float)v0);
}

// x_small: the fp32 x chunk of the fp32 fragment pack, scaled by 2^16
__global__ void scale_xchunk_kernel(const float* __restrict__ wpack, float* __restrict__ wx32,
                                    int nch, size_t total, float scale) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const size_t cb = idx / (kBN * kBK), i = idx - cb * (kBN * kBK);
  wx32[idx] = wpack[(cb * nch + 0) * (size_t)(kBN * kBK) + i] * scale;
}

// fp32 [M][C] -> two scaled fp16 planes in the operand layout (plane_index,
// kernels_misc.h): tiles of 32 cells x 16 channels in MFMA A-fragment order.  One
// thread per (cell, 8 channels), cells fastest inside a tile: 16-byte stores,
// contiguous over the 32 cells of a tile half.
__device__ __forceinline__ void split_planes_item(const float* __restrict__ in,
                                                  _Float16* __restrict__ p0,
                                                  _Float16* __restrict__ p1, int M, int C,
                                                  size_t i) {
  const int c8n = C >> 3;
  const size_t per_tile = (size_t)32 * c8n;
  const size_t mb = i / per_tile;
  const int r = (int)(i - mb * per_tile);
  const int c8 = r >> 5, cell = r & 31;
  const long long m = (long long)mb * 32 + cell;
  if (m >= M) return;
  const f32x4* src = reinterpret_cast<const f32x4*>(in + (size_t)m * C + c8 * 8);
  const f32x4 v0 = src[0], v1 = src[1];
  f16x8 a, b;
  const size_t o = plane_index(m, c8 * 8, C);
  if (!p1) {          // bf16 mode: one unscaled plane
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = bf16_as_half(j < 4 ? v0[j] : v1[j - 4]);
    *reinterpret_cast<f16x8*>(p0 + o) = a;
    return;
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float sv = (j < 4 ? v0[j] : v1[j - 4]) * kF16Scale;
    const _Float16 h0 = (_Float16)sv;
    a[j] = h0;
    b[j] = (_Float16)(sv - (float)h0);
  }
  *reinterpret_cast<f16x8*>(p0 + o) = a;
  *reinterpret_cast<f16x8*>(p1 + o) = b;
}
__global__ void split_planes_kernel(const float* __restrict__ in, _Float16* __restrict__ p0,
                                    _Float16* __restrict__ p1, int M, int C) {
  split_planes_item(in, p0, p1, M, C, (size_t)blockIdx.x * blockDim.x + threadIdx.x);
}
static inline unsigned split_planes_blocks(size_t M, int C) {
  return (unsigned)((((M + 31) / 32) * 32 * (size_t)(C >> 3) + 255) / 256);
}
// The x and h operands of up to four grouped ConvLSTM problems in ONE launch (a training
// forward step issued up to eight of these 4-6 us launches in front of every gate kernel).
constexpr int kSplitGroup = 8;
struct SplitGroup {
  const float* in[kSplitGroup];
  _Float16* p0[kSplitGroup];
  _Float16* p1[kSplitGroup];
  int M[kSplitGroup], C[kSplitGroup];
  unsigned blk_end[kSplitGroup];      // running block count: item j owns [blk_end[j-1], blk_end[j])
  int n;
};
__global__ void split_planes_group_kernel(const SplitGroup g) {
  int j = 0;
  while (j + 1 < g.n && blockIdx.x >= g.blk_end[j]) ++j;
  const unsigned b0 = j ? g.blk_end[j - 1] : 0u;
  split_planes_item(g.in[j], g.p0[j], g.p1[j], g.M[j], g.C[j],
                    (size_t)(blockIdx.x - b0) * blockDim.x + threadIdx.x);
}

// ---------------------------------------------------------------- v2: B through LDS
// v1 streams 10 KB of operands per k-step (384 MFMA cycles) per wave from L1:
// 104 B/clk/CU at full matrix rate against an L1 of 64 B/clk/CU (measured v1:
// 266 TF-equivalent = 32 % of the fp16 pipe).  The four waves of a workgroup
// need the SAME weight fragments, so v2 stages them through LDS: each stage of
// kKpb k-steps (kKpb x 8 KB) is copied global -> LDS once per workgroup
// (coalesced 16-B loads, one quarter per wave) into a double buffer and read back
// as fragments with conflict-free ds_read_b128; one barrier per stage.  L1
// traffic per workgroup and k-step drops from 40 KB to 16 KB.
#ifndef MV_F16_KPB
#define MV_F16_KPB 3
#endif
// Waves per workgroup.  Eight waves (256 cells) share one LDS weight stage: with the
// LDS-DMA copy the kernel needs 120 VGPRs, so 2 workgroups x 8 waves = 4 waves per SIMD
// fit (4 waves per workgroup: 3 workgroups of 48 KB LDS = 3 waves per SIMD), and every
// staged weight byte feeds twice the MFMAs: 1.185 -> 1.137 ms per grouped launch
// (greedy), 12.82 -> 12.08 ms (beam 20).
#ifndef MV_CONV_WAVES
#define MV_CONV_WAVES 8
#endif
#ifndef MV_BF16_UNITS
#define MV_BF16_UNITS 2         // bf16 kernel: row units (of 3 k-steps) per LDS stage
#endif
#ifndef MV_BF16_ADIST
#define MV_BF16_ADIST 1          // bf16 kernel: A operands requested this many k-steps ahead (2: measured -1.5 %)
#endif
// Cache policy of the three streams (gfx940+ aux bits of the buffer builtins: 1 = sc0,
// 2 = nt, 16 = sc1): the weight stage DMA, the epilogue's read-once c state and its
// write-once c' / h' / gate stores.  Tuned per DESIGN.md section 5.
#ifndef MV_DMA_AUX
#define MV_DMA_AUX 0
#endif
#ifndef MV_EPI_LD_AUX
#define MV_EPI_LD_AUX 0
#endif
#ifndef MV_EPI_ST_AUX
#define MV_EPI_ST_AUX 0
#endif
constexpr int kWaves16 = MV_CONV_WAVES;           // waves per workgroup of the f16x3 kernels
constexpr int kThreads16 = kWaves16 * 64;
constexpr int kBlockRows16 = kWaves16 * kWaveRows; // cells per workgroup
constexpr int kKpb = MV_F16_KPB;                 // k-steps per LDS stage (2, 3 or 6 divide every k-step count)
constexpr int kStageVec = kKpb * 2 * 4 * 64;     // f16x8 elements per stage (kKpb x 8 KB)
// LDS of a step / dgrad workgroup: two stage buffers, and at least the 4 KB per wave the
// epilogue's h' plane transposes take (16-wave workgroups: 64 KB)
constexpr int kLdsVec16 = 2 * kStageVec > kWaves16 * 256 ? 2 * kStageVec : kWaves16 * 256;

// EPI = kEpiLstm (forward, NG = 4) or kEpiStore (dgrad: the operand "h" is the gate
// gradient G with 4C channels, the columns are input channels; NG active 32-column
// sub-blocks in this column block; result scaled back by 2^-(8 + *g_exp)).
// NPL = operand planes: 2 = f16x3 (fp16 planes of 256 v, three MFMAs per product),
// 1 = bf16 (ONE bf16 plane of v itself, one v_mfma_f32_32x32x16_bf16 per product, fp32
// accumulate: the reduced-precision mode of BASELINE.json configs[4]; plane buffers and
// packs hold bf16 bit patterns in the same 16-bit containers, same tile layout).
template <int EPI, int NG, int NPL = 2, bool SHIFT = false, bool XF16 = false>
__device__ __forceinline__ void convlstm16_lds_body(const ConvLstm16Args& p, int cb, int mt,
                                                    int kslice, int n_kslice,
                                                    f16x8* lds /* [2][kKpb * NPL * 4 * 64] */) {
  const ConvLstmArgs& a = p.f;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int H = a.H, W = a.W, HW = H * W, C = a.C, Cx = a.Cx;
  const int M_total = a.rows * HW;
  const int m_wave = mt * kBlockRows16 + wave * kWaveRows;
  const bool wave_live = m_wave < M_total;     // dead waves still copy and hit barriers

  int ypos, xpos, xoff, xcell, hcell;    // xcell / hcell: flat cell index in the operand planes
  {
    const int m = m_wave + (lane & 31);
    if (m < M_total) {
      const int r = m / HW, cell = m - r * HW;
      const int y = cell / W;
      ypos = y; xpos = cell - y * W;
      const int sr = a.src_row_h ? a.src_row_h[r] : r;
      xoff = r * a.x_row_stride + cell * Cx;
      xcell = m;
      hcell = sr * HW + cell;
    } else {
      ypos = -100000; xpos = -100000; xoff = 0; xcell = 0; hcell = 0;
    }
  }
  const int khalf = (lane >> 5) * 256;   // second k half of a 32 x 16 operand tile

  f32x16 acc[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;

  if constexpr (EPI == kEpiLstm) if (a.x_small && wave_live) {
    const int khalf4 = (lane >> 5) * 4;
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(p.wx32 + (size_t)cb * kBN * kBK) + lane;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      f32x4 b[4];
#pragma unroll
      for (int g = 0; g < 4; ++g) b[g] = wsrc[(kk * 4 + g) * 64];
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = kk * 8 + khalf4 + j;
        const int tap = k / Cx, ch = k - tap * Cx;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int yy = ypos + dy, xx = xpos + dx;
        const bool ok = (k < 9 * Cx) & (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W);
        int off = xoff + (dy * W + dx) * Cx + ch;
        off = ok ? off : 0;
        const float tv = a.x[off];
        v[j] = ok ? tv : 0.f;
      }
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(v[j], b[g][j], acc[g], 0, 0, 0);
    }
  }

  // ---- f16 k-steps.  k-step s = (channel group cg, tap); a stage = the three
  // taps of one image ROW of the stencil (tap = 3 j + kk, j = dy + 1 uniform per
  // stage, kk = dx + 1 compile-time), so the per-k-step address is one add and
  // the border handling one select: an out-of-image tap reads the 16 zero halves
  // that precede every operand plane (kPlanePad) instead of being masked with
  // eight v_and.  (The first version spent 8.4 non-MFMA instructions per MFMA --
  // 3.7 VALU + 3.5 SALU -- against ~6 issue slots per 32-cycle MFMA: issue-bound at
  // 49 % MFMA busy, profiles/r1_f16x3_pmc_v2.json.)
  static_assert(kKpb == 3, "a row unit is one stencil row");
  // An LDS stage holds R row units (R x 3 k-steps).  f16x3: R = 1 (24 KB).  bf16: one MFMA per
  // product makes a row unit only 384 matrix-pipe cycles long, so the per-stage barrier and
  // DMA drain weigh three times as much; R = 2 puts the same 24 KB behind each barrier.
  constexpr int R = (NPL == 1) ? MV_BF16_UNITS : 1;
  constexpr int kBufVec = R * kKpb * NPL * 4 * 64;   // 16-B vectors per LDS stage buffer
  const int nxk = p.n_xk;
  const int nsteps = nxk + p.n_hk;
  const int nstages = nsteps / 3;
  const int nxst = nxk / 3;
  const f16x8* wblk = reinterpret_cast<const f16x8*>(p.wp16) +
                      (size_t)cb * p.w_ksteps * (NPL * 4 * 64);
  constexpr int kSV = R * 3 * NPL * NG * 64;   // 16-B vectors per stage (NG sub-blocks)
  constexpr int kCopy = (kSV + kThreads16 - 1) / kThreads16;     // per thread
  // stage vector v = ((kk*NPL + plane)*NG + g)*64 + lane  ->  its place in the pack,
  // which keeps four sub-block slots per (k-step, plane)
  // stage vector v = ((kk*NPL + plane)*NG + g)*64 + lane  ->  its place in the stage's
  // slice of the pack, which keeps four sub-block slots per (k-step, plane); a row unit
  // (3 k-steps) is 3 * NPL * 256 vectors
  auto pack_rel = [&](int v) -> uint32_t {
    const int ln = v & 63;
    int t = v >> 6;
    const int g = t % NG; t /= NG;
    const int plane = t % NPL, kk = t / NPL;
    return (uint32_t)(((kk * NPL + plane) * 256 + g * 64 + ln) * 16);
  };
  constexpr uint32_t kUnitBytes = 3 * NPL * 256 * 16;
  // Buffer addressing (V# in SGPRs + one 32-bit VGPR offset + one SGPR offset) for the
  // weight stream and the operand planes: no 64-bit address arithmetic per lane.
  const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(const_cast<f16x8*>(wblk)), 0, 0x7fffffff, 0x00020000);

  const bool okx0 = (xpos - 1 >= 0) & (xpos - 1 < W), okx1 = (xpos >= 0) & (xpos < W),
             okx2 = (xpos + 1 >= 0) & (xpos + 1 < W);
  // row validity as a bit mask (a 3-way select of captured bools went through
  // scratch memory and a flat byte load)
  const int okymask = (int)((ypos - 1 >= 0) & (ypos - 1 < H)) |
                      ((int)((ypos >= 0) & (ypos < H)) << 1) |
                      ((int)((ypos + 1 >= 0) & (ypos + 1 < H)) << 2);
  // per-stage address state as plain scalars (a struct of pointers here ended up
  // in scratch memory and turned the loads into flat_load).  The operand planes
  // are tiled (plane_index): cell m', channel group cg, k half -> one 16-byte
  // vector at ((m' >> 5) * KG + cg) * 512 + khalf + (m' & 31) * 8, so a wave's
  // load of 32 consecutive cells is two contiguous 512-byte runs instead of 32
  // separate cache lines.
  const int KGx = Cx >> 4, KGh = C >> 4;
  // bf16 mode with unbounded activations (x_exp set, NPL == 1): the x stages come in THREE
  // passes over the same (channel group, stencil row) sequence -- x hi plane x w hi, x lo x w hi,
  // x hi x w lo (the f16x3 split of the x part alone, on the fp16 MFMA; the pack holds the x
  // rows three times accordingly, pack_bf16_*) -- while the h stages stay one bf16 plane
  // (XF16 is a TEMPLATE parameter: compiled into the ordinary bf16 kernel the extra stage state
  // pushed it from 117 registers into 1 900 spills -- 16x slower; found by the bench, not by the
  // tests)
  static_assert(!XF16 || (EPI == kEpiLstm && NPL == 1), "x passes: bf16 forward only");
  constexpr int xpasses = XF16 ? 3 : 1;
  const int nxr = nxst / xpasses;                        // x stages of one pass
  auto xq = [&](int st) {                                // st < nxst -> stage inside its pass
    int q = st;
    if (q >= nxr) q -= nxr;
    if (q >= nxr) q -= nxr;
    return q;
  };
  auto stage_isx = [&](int st) { return st < nxst; };
  auto stage_xlo = [&](int st) { return XF16 && st >= nxr && st < 2 * nxr; };
  auto stage_rowoff = [&](int st) {       // (stencil row - 1) * W + first cell of the lane
    const bool is_x = st < nxst;
    const int q = is_x ? xq(st) : st - nxst;
    const int j = q - (q / 3) * 3;
    return (is_x ? xcell : hcell) + (j - 1) * W;
  };
  auto stage_cg = [&](int st) {
    const int q = (st < nxst) ? xq(st) : st - nxst;
    return q / 3;
  };
  auto stage_rowok = [&](int st) {
    const int q = (st < nxst) ? xq(st) : st - nxst;
    const int j = q - (q / 3) * 3;
    return ((okymask >> j) & 1) != 0;
  };
  const _Float16* const x16 = p.x16;
  const _Float16* const h16 = p.h16;
  const int64_t xps = p.x_plane_stride, hps = p.h_plane_stride;

  // The bases point at the zero pad in FRONT of a plane, so the offset of an out-of-image
  // tap is simply 0.  Host-checked: a plane pair is shorter than 2 GB.
  const __amdgpu_buffer_rsrc_t xrs0 = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(const_cast<_Float16*>(x16 ? x16 - kPlanePad : h16 - kPlanePad)), 0, 0x7fffffff,
      0x00020000);
  const __amdgpu_buffer_rsrc_t xrs1 = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(const_cast<_Float16*>(x16 ? x16 + xps - kPlanePad : h16 - kPlanePad)), 0,
      0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t hrs0 = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(const_cast<_Float16*>(h16 - kPlanePad)), 0, 0x7fffffff, 0x00020000);
  const __amdgpu_buffer_rsrc_t hrs1 = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(const_cast<_Float16*>(h16 + hps - kPlanePad)), 0, 0x7fffffff, 0x00020000);
#define MV_LOAD_A(ISX, XLO, ROWOFF, CG, ROWOK, KK, A0, A1)                             \
  do {                                                                                  \
    const bool ok_ = (ROWOK) & ((KK) == 0 ? okx0 : ((KK) == 1 ? okx1 : okx2));          \
    const int mm_ = (ROWOFF) + ((KK) - 1);                                              \
    const int off_ =                                                                    \
        ok_ ? (((mm_ >> 5) * ((ISX) ? KGx : KGh) + (CG)) * 512 + khalf +                \
               (mm_ & 31) * 8 + kPlanePad) * 2                                          \
            : 0;                                                                        \
    A0 = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(               \
                                       (ISX) ? ((XLO) ? xrs1 : xrs0) : hrs0, off_, 0, 0)); \
    if (NPL == 2)                                                                       \
      A1 = __builtin_bit_cast(f16x8, __builtin_amdgcn_raw_buffer_load_b128(             \
                                         (ISX) ? xrs1 : hrs1, off_, 0, 0));             \
  } while (0)

  // stage range of this workgroup (split-K: n_kslice equal ranges)
  int st_lo = (nstages / n_kslice) * kslice;
  const int st_hi = n_kslice > 1 ? st_lo + nstages / n_kslice : nstages;
  if constexpr (EPI == kEpiLstm)
    if (a.sx_corr) st_lo = nxst;           // sparse x: its k-steps are table terms (epilogue)
  // The stage copy is an LDS-DMA (global_load_lds, 16 B per lane: the pack IS the LDS
  // image, lane-linear, so a wave's piece is one 1 KB run on both sides).  Round 1 staged
  // global -> VGPR -> ds_write: 24 more VGPRs (154 vs 126) and a write pass; 1.226 ->
  // 1.207 ms per grouped launch.  Tried on top of it and dropped: a second
  // register set of B fragments read one k-step ahead (180 VGPRs, 2 waves per SIMD:
  // 1.317 ms; capped at 168 registers it spills: 1.380 ms) and A fragments requested
  // two k-steps ahead (140 VGPRs: 1.192 vs 1.184 ms on the same box) -- neither the
  // LDS nor the L2 latency of a single wave is what limits the matrix pipe here.
  if (st_hi > st_lo) {
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    uint32_t dma_voff[kCopy];              // this thread's pieces inside a stage slice
#pragma unroll
    for (int i = 0; i < kCopy; ++i) dma_voff[i] = pack_rel((i * kWaves16 + wave_u) * 64 + lane);
    auto stage_dma = [&](int st, f16x8* dstbuf) {
#pragma unroll
      for (int i = 0; i < kCopy; ++i) {
        const int v0 = (i * kWaves16 + wave_u) * 64;              // wave-uniform piece of 64 vectors
        if (kSV % kThreads16 == 0 || v0 < kSV)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(
              wrs, (__attribute__((address_space(3))) void*)(dstbuf + v0), 16, dma_voff[i],
              (uint32_t)st * kUnitBytes, 0, MV_DMA_AUX);
      }
    };
    // LDS stages of R row units each; (st_hi - st_lo) % R == 0 (host-checked for bf16)
    const int nsg = (st_hi - st_lo) / R;
    {
    stage_dma(st_lo, lds);
    bool c_isx = stage_isx(st_lo), c_xlo = stage_xlo(st_lo);
    int c_rowoff = stage_rowoff(st_lo), c_cg = stage_cg(st_lo);
    bool c_rowok = stage_rowok(st_lo);
    // A operands.  The three k-steps of a stencil row read the SAME 32 x 16 tile of the lane's
    // plane shifted by one cell (dx = -1, 0, +1).  SHIFT (every W of the launch divides 32, so
    // a tile never ends inside an image row: lane 0 / 31 of a k half is x == 0 / W - 1, a
    // masked tap anyway): only the centre fragment is loaded, one stage ahead; the dx = -/+1
    // fragments are the centre moved one lane up / down the wave (v_mov_b32_dpp wave_shr:1 /
    // wave_shl:1) and zeroed where the neighbour is outside the image row -- one third of
    // the operand requests, issued a whole stage (36 MFMAs) before their use.  Other widths
    // load every tap, one k-step ahead (cc = the fragments of the NEXT k-step there).
    auto lane_shift = [&](const f16x8& v, bool up, bool ok) -> f16x8 {
      const u32x4 w = __builtin_bit_cast(u32x4, v);
      u32x4 r;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const uint32_t t = up ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[j], 0x138, 0xf, 0xf, false)
                              : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[j], 0x130, 0xf, 0xf, false);
        r[j] = ok ? t : 0u;
      }
      return __builtin_bit_cast(f16x8, r);
    };
    f16x8 cc0, cc1;                        // SHIFT: centre fragments of the current stage
    MV_LOAD_A(c_isx, c_xlo, c_rowoff, c_cg, c_rowok, SHIFT ? 1 : 0, cc0, cc1);
    if constexpr (NPL == 1) cc1 = cc0;
    __syncthreads();                       // carries the vmcnt(0) of the pending LDS-DMA
    for (int sg = 0; sg < nsg; ++sg) {
      const bool more = sg + 1 < nsg;
      const f16x8* buf = lds + (sg & 1) * kBufVec;
#pragma unroll
      for (int u = 0; u < R; ++u) {
      const int st = st_lo + sg * R + u;
      const int stn = st + 1 < st_hi ? st + 1 : st;
      const bool n_isx = stage_isx(stn), n_xlo = stage_xlo(stn);
      const int n_rowoff = stage_rowoff(stn), n_cg = stage_cg(stn);
      const bool n_rowok = stage_rowok(stn);
      f16x8 cn0, cn1;                      // SHIFT: centre fragments of the next stage
      if constexpr (SHIFT) {
        MV_LOAD_A(n_isx, n_xlo, n_rowoff, n_cg, n_rowok, 1, cn0, cn1);   // (re-read at the very end)
        if constexpr (NPL == 1) cn1 = cn0;
      }
#pragma unroll
      for (int kk = 0; kk < 3; ++kk) {
        const int kq = u * 3 + kk;         // k-step inside the LDS stage
        f16x8 fa0, fa1;
        if constexpr (SHIFT) {
          if (kk == 1) { fa0 = cc0; fa1 = cc1; }
          else {
            const bool okx = kk == 0 ? okx0 : okx2;
            fa0 = lane_shift(cc0, kk == 0, okx);
            if constexpr (NPL == 2) fa1 = lane_shift(cc1, kk == 0, okx);
          }
        } else {
          fa0 = cc0; fa1 = cc1;            // loaded one k-step ago
          if (kk < 2) MV_LOAD_A(c_isx, c_xlo, c_rowoff, c_cg, c_rowok, kk + 1, cc0, cc1);
          else MV_LOAD_A(n_isx, n_xlo, n_rowoff, n_cg, n_rowok, 0, cc0, cc1);
        }
        // the DMA of the next stage goes out behind the first k-step's operands (vmcnt
        // retires in order); its target buffer was last read before the previous barrier
        if (kq == 1 && more)
          stage_dma(st_lo + (sg + 1) * R, lds + ((sg + 1) & 1) * kBufVec);
        if constexpr (NPL == 2) {
          f16x8 b0[NG], b1[NG];
#pragma unroll
          for (int g = 0; g < NG; ++g) {
            b0[g] = buf[((kq * 2 + 0) * NG + g) * 64 + lane];
            b1[g] = buf[((kq * 2 + 1) * NG + g) * 64 + lane];
          }
#pragma unroll
          for (int g = 0; g < NG; ++g)
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa1, b0[g], acc[g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < NG; ++g)
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0, b1[g], acc[g], 0, 0, 0);
#pragma unroll
          for (int g = 0; g < NG; ++g)
            acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0, b0[g], acc[g], 0, 0, 0);
        } else {
          f16x8 b0[NG];
#pragma unroll
          for (int g = 0; g < NG; ++g) b0[g] = buf[(kq * NG + g) * 64 + lane];
          if (XF16 && c_isx) {                     // fp16 x plane x fp16 x rows (uniform)
#pragma unroll
            for (int g = 0; g < NG; ++g)
              acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa0, b0[g], acc[g], 0, 0, 0);
          } else {
#pragma unroll
            for (int g = 0; g < NG; ++g)
              acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                  __builtin_bit_cast(bf16x8, fa0), __builtin_bit_cast(bf16x8, b0[g]), acc[g], 0,
                  0, 0);
          }
        }
      }
      if constexpr (SHIFT) { cc0 = cn0; cc1 = cn1; }
      c_isx = n_isx; c_xlo = n_xlo; c_rowoff = n_rowoff; c_cg = n_cg; c_rowok = n_rowok;
      if constexpr (EPI == kEpiLstm && (NPL == 2 || XF16)) {
        if (p.x_exp && st == nxst - 1) {
          // x planes at 2^e: bring the sums to 2^16 (f16x3: the h products follow at 2^8 * 2^8)
          // or, bf16 mode, back to 1 (fp16 x plane at 2^e x fp16 x rows at 2^8; the h products
          // follow unscaled)
          const float f = __int_as_float((127 + (NPL == 2 ? 8 : -8) - p.x_exp[0]) << 23);
#pragma unroll
          for (int g = 0; g < NG; ++g)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[g][i] *= f;
        }
      }
      }
      __syncthreads();
    }
    }
  }
#undef MV_LOAD_A
  if (!wave_live) return;
  // The epilogue's per-lane addresses are functions of `lane`; computed from this laundered
  // copy they cannot be hoisted above the main loop, where they cost it registers (the
  // stage loop runs at the 128-VGPR edge of two 8-wave workgroups per CU).
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));

  if constexpr (EPI == kEpiStore) {
    // NPL == 1: bf16 operands are unscaled (bf16 has fp32's exponent range)
    const float scale = NPL == 2 ? ldexpf(1.0f, -(8 + (p.g_exp ? p.g_exp[0] : 0))) : 1.0f;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int col = cb * kBN + g * 32 + (lane_e & 31);
      float* dst = nullptr;
      int stride = 0, cc = col;
      if (col < a.out0_cols) {
        stride = a.out0_cols;
        dst = n_kslice > 1 ? (a.out0 ? p.part0 + (size_t)kslice * M_total * stride : nullptr)
                           : a.out0;
      } else if (col - a.out0_cols < a.out1_cols) {
        stride = a.out1_cols; cc = col - a.out0_cols;
        dst = n_kslice > 1 ? (a.out1 ? p.part1 + (size_t)kslice * M_total * stride : nullptr)
                           : a.out1;
      }
      if (dst) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane_e >> 5);
          const int m = m_wave + row;
          if (m < M_total) dst[(size_t)m * stride + cc] = acc[g][reg] * scale;
        }
      }
    }
  } else {
  const int ch = cb * kChBlock + (lane_e & 31);
    const int half = lane_e >> 5;
    // sparse x: the registers hold the INTERIOR class of the bias table
    const float* const b0 = a.sx_bias ? a.sx_bias + (size_t)4 * 4 * C : a.bias;
    const float bi = b0[ch], bj = b0[C + ch], bf = b0[2 * C + ch], bo = b0[3 * C + ch];
    _Float16* const tl = reinterpret_cast<_Float16*>(lds) + wave * (1024 * NPL);
    // The wave's 32 cells start in image r0 at cell0 and cross at most one image boundary
    // (HW >= 32, host-checked for this kernel): rows >= wrap_at belong to image r0 + 1.
    const int r0 = __builtin_amdgcn_readfirstlane(m_wave / HW);
    const int cell0 = __builtin_amdgcn_readfirstlane(m_wave - r0 * HW);
    const int wrap_at = HW - cell0;
    // sparse x: hot cells of the two images of the tile
    int hot_y[2] = {0, 0}, hot_x[2] = {0, 0};
    if (a.sx_corr) {
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        int rr = r0 + k;
        if (rr >= a.rows) rr = a.rows - 1;
        const int hr = a.sx_hot_div > 1 ? rr / a.sx_hot_div : rr;
        const uint32_t hyx = a.sx_cellyx[a.sx_hot[(size_t)hr * a.sx_hot_stride]];
        hot_y[k] = (int)(hyx >> 16); hot_x[k] = (int)(hyx & 0xffffu);
      }
    }
    // ---- pass 1: every state load of the epilogue goes out BEFORE any of its stores.
    // Written as one loop (load c, update, store c', h' per register) the 16 iterations ran
    // one after the other -- a load cannot be hoisted over the previous iteration's stores
    // -- so each waited out its own HBM round trip: the epilogue was 0.29 of the 1.03 ms
    // launch (profiles/r3_ablation_gate_kernel_s1.md).  Buffer addressing throughout: rows
    // past M_total fall outside num_records (loads return 0, stores are dropped), so there
    // is no per-element bounds branch.  The whole byte offset sits in the VGPR operand: the
    // hardware's range check does not look at the scalar offset.
    // Source rows of the two images are looked up once, scalar.
    int sr0 = r0, sr1 = r0 + 1;
    if (a.src_row_c && !a.zero_state) {
      sr0 = a.src_row_c[r0 < a.rows ? r0 : a.rows - 1];
      sr1 = a.src_row_c[r0 + 1 < a.rows ? r0 + 1 : a.rows - 1];
    }
    sr0 = __builtin_amdgcn_readfirstlane(sr0);
    sr1 = __builtin_amdgcn_readfirstlane(sr1);
    const uint32_t rowb = (uint32_t)C * 4u;                         // bytes per cell
    const __amdgpu_buffer_rsrc_t c_rs = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(const_cast<float*>(a.zero_state ? a.c_out : a.c)), 0,
        (uint32_t)M_total * (uint32_t)C * 4u, 0x00020000);
    const uint32_t c_off0 = ((uint32_t)(sr0 * HW + cell0 + 4 * half) * (uint32_t)C + ch) * 4u;
    const uint32_t c_wrap = (uint32_t)((sr1 - sr0 - 1) * HW) * rowb;   // added once wrapped
    float cprev[16];
  #pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int rc = (reg & 3) + 8 * (reg >> 2);
      const bool wrapped = rc + 4 * half >= wrap_at;
      cprev[reg] = 0.f;
      if (!a.zero_state)
        cprev[reg] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(
            c_rs, (int)(c_off0 + (wrapped ? c_wrap : 0u) + (uint32_t)rc * rowb), 0,
            MV_EPI_LD_AUX));
    }
    const uint32_t out_bytes = (uint32_t)M_total * rowb;
    const __amdgpu_buffer_rsrc_t co_rs = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(a.c_out), 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ho_rs = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(a.h_out), 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t go_rs = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(a.gates_out ? a.gates_out : a.h_out), 0, a.gates_out ? 4u * out_bytes : 0u,
        0x00020000);
    const uint32_t o_off0 = ((uint32_t)(m_wave + 4 * half) * (uint32_t)C + ch) * 4u;
    const uint32_t wmagic = 0xFFFFFFFFu / (uint32_t)W + 1u;         // cell / W = umulhi(cell, wmagic)
  #pragma unroll
    for (int reg = 0; reg < 16; ++reg) {
      const int rc = (reg & 3) + 8 * (reg >> 2);
      const int row = rc + 4 * half;
      const int m = m_wave + row;
      float hn_keep = 0.f;                 // cells past the end: zero planes
      {
        constexpr float kUn = NPL == 2 ? kF16Unscale : 1.0f;   // bf16 operands are unscaled
        float gi = acc[0][reg] * kUn, gj = acc[1][reg] * kUn, gf = acc[2][reg] * kUn,
              go = acc[3][reg] * kUn;
        if (a.sx_corr) {
          const bool wrapped = row >= wrap_at;
          const int r = wrapped ? r0 + 1 : r0;
          const int cell = cell0 + row - (wrapped ? HW : 0);
          const int y = (int)__umulhi((uint32_t)cell, wmagic), x = cell - y * W;
          float ci = bi, cj = bj, cf = bf, co = bo;
          if (a.sx_bias) {
            const int cls = 3 * (y == 0 ? 0 : (y == H - 1 ? 2 : 1)) +
                            (x == 0 ? 0 : (x == W - 1 ? 2 : 1));
            if (cls != 4 && m < M_total) {
              const float* bt = a.sx_bias + (size_t)cls * 4 * C + ch;
              ci = bt[0]; cj = bt[C]; cf = bt[2 * C]; co = bt[3 * C];
            }
          }
          const int hy = wrapped ? hot_y[1] : hot_y[0], hx = wrapped ? hot_x[1] : hot_x[0];
          const int dy = y - hy, dx = x - hx, rad = a.sx_rad;
          if (dy >= -rad && dy <= rad && dx >= -rad && dx <= rad && m < M_total) {
            const int side = 2 * rad + 1;
            const int idx = a.sx_by_class
                                ? 3 * (hy == 0 ? 0 : (hy == H - 1 ? 2 : 1)) +
                                      (hx == 0 ? 0 : (hx == W - 1 ? 2 : 1))
                                : r;
            const float* ct = a.sx_corr +
                              ((size_t)idx * side * side + (dy + rad) * side + (dx + rad)) * 4 * C + ch;
            ci += ct[0]; cj += ct[C]; cf += ct[2 * C]; co += ct[3 * C];
          }
          gi += ci; gj += cj; gf += cf; go += co;
        } else {
          gi += bi; gj += bj; gf += bf; go += bo;
        }
        const float si = sigm_(gi), tj = tanh_(gj), sf = sigm_(gf + a.forget_bias), so = sigm_(go);
        float cn = sf * cprev[reg];
        cn = cn + si * tj;
        const float hn = tanh_(cn) * so;
        const int o_off = (int)(o_off0 + (uint32_t)rc * rowb);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, cn), co_rs, o_off, 0, MV_EPI_ST_AUX);
        if (!a.skip_h32)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, hn), ho_rs, o_off, 0, MV_EPI_ST_AUX);
        hn_keep = m < M_total ? hn : 0.f;
        if (a.gates_out) {
          const uint32_t g0 = ((uint32_t)(m_wave + row) * 4u * (uint32_t)C + ch) * 4u;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, si), go_rs,
                                                (int)g0, 0, MV_EPI_ST_AUX);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, tj), go_rs,
                                                (int)(g0 + rowb), 0, MV_EPI_ST_AUX);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, sf), go_rs,
                                                (int)(g0 + 2 * rowb), 0, MV_EPI_ST_AUX);
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, so), go_rs,
                                                (int)(g0 + 3 * rowb), 0, MV_EPI_ST_AUX);
        }
      }
      if (p.h16_out) {
        // operand planes of h' for the next step: the wave's 32 cells x 32 channels are
        // two plane tiles (32 cells x 16 channels, [k half][cell][8 ch]); a lane holds
        // ONE channel of 16 cells, so the tile is assembled in LDS (the weight stage
        // buffers are free after the last barrier; 4 KB per wave, wave-private) and
        // goes out as 16-byte vectors, 1 KB contiguous per tile and plane.  (Stored
        // straight from the accumulator layout these were 2-byte stores at a 16-byte
        // stride: slower than a separate split pass)
        const int chl = lane_e & 31;
        const int tofs = (chl >> 4) * 512 + ((chl >> 3) & 1) * 256 + (chl & 7) + row * 8;
        if constexpr (NPL == 2) {
          const float sc = hn_keep * kF16Scale;
          const _Float16 h0 = (_Float16)sc;
          tl[tofs] = h0;
          tl[1024 + tofs] = (_Float16)(sc - (float)h0);
        } else {
          tl[tofs] = bf16_as_half(hn_keep);
        }
      }
    }
    if (p.h16_out) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // LDS ops of a wave retire in order
      const size_t tile0 = ((size_t)(m_wave >> 5) * (size_t)(C >> 4) + (size_t)cb * 2) * 512;
#pragma unroll
      for (int q = 0; q < 2 * NPL; ++q) {  // q = plane * 2 + tile
        const f16x8 v = *reinterpret_cast<const f16x8*>(tl + q * 512 + lane_e * 8);
        *reinterpret_cast<f16x8*>(p.h16_out + (size_t)(q >> 1) * p.h16_out_stride + tile0 +
                                  (size_t)(q & 1) * 512 + lane_e * 8) = v;
      }
    }
  }
}

// SHIFT form of the step kernels: every problem's W divides 32 (MV_CONV_SHIFT=0: never)
static inline bool conv_group_shift(const ConvLstm16Args* probs, int n) {
  static const bool off = getenv("MV_CONV_SHIFT") && atoi(getenv("MV_CONV_SHIFT")) == 0;
  if (off) return false;
  for (int i = 0; i < n; ++i)
    if (probs[i].f.W <= 0 || 32 % probs[i].f.W != 0) return false;
  return true;
}

// block -> (column block cb, row tile mt) of the forward step.  A workgroup reads the
// operand planes of its 256 cells and ONE column block's weights.
//   mode 0  cb = block % 8: a column block per XCD -- its L2 keeps that block's 1.3 MB of
//           weights, and every XCD streams ALL the activations (8x the activation bytes
//           over the fabric);
//   mode 1  mt % 8 = block % 8: a row tile lives on one XCD, whose eight column-block
//           workgroups (consecutive on that XCD) share the tile's planes through its L2;
//           the weights (10.6 MB per gate kernel) stream to every XCD instead.  The grid
//           is padded to a multiple of 8 row tiles (dead workgroups exit at once).
__device__ __forceinline__ bool step_block_map(const ConvLstmArgs& a, int block, int mode,
                                               int& cb, int& mt) {
  const int ncb = a.n_colblocks;
  if (mode == 0) {
    cb = block % ncb;
    mt = block / ncb;
    return true;
  }
  const int q = block >> 3;
  cb = q % ncb;
  mt = (q / ncb) * 8 + (block & 7);
  const int mtiles = (a.rows * a.H * a.W + kBlockRows16 - 1) / kBlockRows16;
  return mt < mtiles;
}

#ifndef MV_CONV_MINWAVES
#define MV_CONV_MINWAVES (MV_CONV_WAVES == 8 ? 4 : 2)
#endif
template <bool SHIFT>
__global__ __launch_bounds__(kThreads16, MV_CONV_MINWAVES)
void convlstm_step_f16x3_lds_kernel(const ConvLstm16Group g) {
  __shared__ f16x8 lds[kLdsVec16];
  int block = blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int i = 0; i < kMaxGroup - 1; ++i)
    if (i + 1 < g.n && (int)blockIdx.x >= g.block_end[i]) pi = i + 1;
  if (pi > 0) block -= g.block_end[pi - 1];
  int cb, mt;
  switch (pi) {
    case 0: if (step_block_map(g.p[0].f, block, g.map_mode, cb, mt)) convlstm16_lds_body<kEpiLstm, 4, 2, SHIFT>(g.p[0], cb, mt, 0, 1, lds); break;
    case 1: if (step_block_map(g.p[1].f, block, g.map_mode, cb, mt)) convlstm16_lds_body<kEpiLstm, 4, 2, SHIFT>(g.p[1], cb, mt, 0, 1, lds); break;
    case 2: if (step_block_map(g.p[2].f, block, g.map_mode, cb, mt)) convlstm16_lds_body<kEpiLstm, 4, 2, SHIFT>(g.p[2], cb, mt, 0, 1, lds); break;
    default: if (step_block_map(g.p[3].f, block, g.map_mode, cb, mt)) convlstm16_lds_body<kEpiLstm, 4, 2, SHIFT>(g.p[3], cb, mt, 0, 1, lds); break;
  }
}

// The same step with ONE bf16 plane per operand (compute mode 2, BASELINE configs[4]).
// XF16: unbounded-activation models, three fp16 passes over the x k-steps (body: xpasses).
template <bool SHIFT, bool XF16 = false>
__global__ __launch_bounds__(kThreads16, XF16 ? 2 : MV_CONV_MINWAVES)
void convlstm_step_bf16_kernel(const ConvLstm16Group g) {
  __shared__ f16x8 lds[MV_BF16_UNITS * kStageVec];   // 2 buffers x (units x 3 k-steps) x 4 KB
  int block = blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int i = 0; i < kMaxGroup - 1; ++i)
    if (i + 1 < g.n && (int)blockIdx.x >= g.block_end[i]) pi = i + 1;
  if (pi > 0) block -= g.block_end[pi - 1];
  int cb, mt;
  switch (pi) {
    case 0: if (step_block_map(g.p[0].f, block, g.map_mode, cb, mt)) convlstm16_lds_body<kEpiLstm, 4, 1, SHIFT, XF16>(g.p[0], cb, mt, 0, 1, lds); break;
    case 1: if (step_block_map(g.p[1].f, block, g.map_mode, cb, mt)) convlstm16_lds_body<kEpiLstm, 4, 1, SHIFT, XF16>(g.p[1], cb, mt, 0, 1, lds); break;
    case 2: if (step_block_map(g.p[2].f, block, g.map_mode, cb, mt)) convlstm16_lds_body<kEpiLstm, 4, 1, SHIFT, XF16>(g.p[2], cb, mt, 0, 1, lds); break;
    default: if (step_block_map(g.p[3].f, block, g.map_mode, cb, mt)) convlstm16_lds_body<kEpiLstm, 4, 1, SHIFT, XF16>(g.p[3], cb, mt, 0, 1, lds); break;
  }
}

// bf16 pack of the gate kernel: [cb][k-step][gate][lane][8], the f16x3 pack's order
// with one unscaled bf16 plane.  Host and device twins (the device one runs after
// every optimizer step).
// xf16 (unbounded-activation models, ConvLstm16Args::x_exp): the x k-steps appear THREE times
// -- fp16 of 256 w, the same again, fp16 of the residual 256 w - hi -- for the kernel's three x
// passes (x hi x w hi, x lo x w hi, x hi x w lo); the h rows stay one bf16 plane
static inline size_t bf16_wpack_elems(int Cx, int C, bool xf16 = false) {
  return (size_t)(C / kChBlock) * (size_t)((xf16 ? 3 : 1) * f16x3_xksteps(Cx) + 9 * (C / 16)) *
         4 * 64 * 8;
}
static inline void pack_bf16_weights(const float* w, int Cx, int C, _Float16* out,
                                     bool xf16 = false) {
  const int Cin = Cx + C, N4 = 4 * C;
  const int nx1 = f16x3_xksteps(Cx), nxk = (xf16 ? 3 : 1) * nx1, nk = nxk + 9 * (C / 16);
  for (int cb = 0; cb < C / kChBlock; ++cb)
    for (int s = 0; s < nk; ++s) {
      const bool is_x = s < nxk;
      const int pass = is_x ? s / (nx1 > 0 ? nx1 : 1) : 0;
      const int q = is_x ? s - pass * nx1 : s - nxk;
      const int cg = q / 9, tap = q % 9;
      for (int g = 0; g < 4; ++g)
        for (int l = 0; l < 64; ++l)
          for (int e = 0; e < 8; ++e) {
            const int k = 8 * (l >> 5) + e;
            const int ci = (is_x ? 0 : Cx) + cg * 16 + k;
            const int n = g * C + cb * kChBlock + (l & 31);
            const float wv = w[((size_t)tap * Cin + ci) * N4 + n];
            const _Float16 w0 = (_Float16)(wv * kF16Scale);
            out[(((size_t)cb * nk + s) * 4 + g) * 512 + (size_t)l * 8 + e] =
                !(xf16 && is_x) ? bf16_as_half(wv)
                                : (pass < 2 ? w0 : (_Float16)(wv * kF16Scale - (float)w0));
          }
    }
}
__global__ void pack_bf16_kernel(const float* __restrict__ w, _Float16* __restrict__ out,
                                 int Cx_total, int Cx16, int C, size_t total, int xf16) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int e = idx & 7;
  const int l = (idx >> 3) & 63;
  const int g = (idx >> 9) & 3;
  const size_t t = idx >> 11;                    // cb * nk + s
  const int nx1 = 9 * (Cx16 / 16), nxk = (xf16 ? 3 : 1) * nx1, nk = nxk + 9 * (C / 16);
  const int s = t % nk, cb = t / nk;
  const bool is_x = s < nxk;
  const int pass = is_x ? s / (nx1 > 0 ? nx1 : 1) : 0;
  const int q = is_x ? s - pass * nx1 : s - nxk;
  const int cg = q / 9, tap = q - cg * 9;
  const int k = 8 * (l >> 5) + e;
  const int ci = (is_x ? 0 : Cx_total) + cg * 16 + k;
  const int n = g * C + cb * kChBlock + (l & 31);
  const int Cin = Cx_total + C, N4 = 4 * C;
  const float wv = w[((size_t)tap * Cin + ci) * N4 + n];
  if (xf16 && is_x) {
    note_pack_range(wv * kF16Scale);
    const _Float16 w0 = (_Float16)(wv * kF16Scale);
    out[idx] = pass < 2 ? w0 : (_Float16)(wv * kF16Scale - (float)w0);
  } else {
    out[idx] = bf16_as_half(wv);
  }
}

static inline void launch_convlstm_bf16_steps(const ConvLstm16Args* probs, int n,
                                              hipStream_t stream);

// dgrad on the fp16 matrix pipe: d[h | x] = conv3x3(G, W^T flipped) with G as two
// fp16 planes under a per-tensor power-of-two scale (split_planes_dyn_kernel) and
// the transposed, tap-flipped kernel as planes (pack_f16x3_dgrad_kernel).
template <bool SHIFT, int NPL = 2>
__device__ __forceinline__ void convlstm16_dgrad_dispatch(const ConvLstm16Args& p, int block,
                                                          f16x8* lds) {
  // block -> (column block, k slice, row tile).  Full-width column blocks first,
  // combo = block % (n_main * nks) = (k slice, column block): with 2 x 4 combos a
  // combo is an XCD (blockIdx % 8), whose L2 then keeps that combo's 1.2 MB of
  // kernel planes while the G planes stream through; the narrow last block (d x
  // columns) follows as its own region.
  const int ncb = p.f.n_colblocks;
  const int nks = p.n_kslice > 1 ? p.n_kslice : 1;
  const bool narrow = p.f.ng_last != 4;
  const int n_main = narrow ? ncb - 1 : ncb;
  const int M_total = p.f.rows * p.f.H * p.f.W;
  const int mtiles = (M_total + kBlockRows16 - 1) / kBlockRows16;
  const int main_blocks = mtiles * n_main * nks;
  int cb, ks, mt;
  if (block < main_blocks) {
    const int combo = block % (n_main * nks);
    mt = block / (n_main * nks);
    cb = combo % n_main; ks = combo / n_main;
  } else {
    const int b2 = block - main_blocks;
    cb = ncb - 1; ks = b2 % nks; mt = b2 / nks;
  }
  if (cb == ncb - 1 && p.f.ng_last == 1)
    convlstm16_lds_body<kEpiStore, 1, NPL, SHIFT>(p, cb, mt, ks, nks, lds);
  else if (cb == ncb - 1 && p.f.ng_last == 2)
    convlstm16_lds_body<kEpiStore, 2, NPL, SHIFT>(p, cb, mt, ks, nks, lds);
  else
    convlstm16_lds_body<kEpiStore, 4, NPL, SHIFT>(p, cb, mt, ks, nks, lds);
}

template <bool SHIFT>
__global__ __launch_bounds__(kThreads16, MV_CONV_MINWAVES)
void convlstm_dgrad_f16x3_kernel(const ConvLstm16Group g) {
  __shared__ f16x8 lds[2 * kStageVec];
  int block = blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int i = 0; i < kMaxGroup - 1; ++i)
    if (i + 1 < g.n && (int)blockIdx.x >= g.block_end[i]) pi = i + 1;
  if (pi > 0) block -= g.block_end[pi - 1];
  switch (pi) {
    case 0: convlstm16_dgrad_dispatch<SHIFT>(g.p[0], block, lds); break;
    case 1: convlstm16_dgrad_dispatch<SHIFT>(g.p[1], block, lds); break;
    case 2: convlstm16_dgrad_dispatch<SHIFT>(g.p[2], block, lds); break;
    default: convlstm16_dgrad_dispatch<SHIFT>(g.p[3], block, lds); break;
  }
}

// The same dgrad with ONE bf16 plane per operand (compute mode 2): G and the transposed,
// tap-flipped kernel as bf16 bit patterns (split_plane_bf16_kernel, pack_bf16_dgrad_kernel),
// one v_mfma_f32_32x32x16_bf16 per product, fp32 accumulate, nothing scaled.
template <bool SHIFT>
__global__ __launch_bounds__(kThreads16, MV_CONV_MINWAVES)
void convlstm_dgrad_bf16_kernel(const ConvLstm16Group g) {
  __shared__ f16x8 lds[MV_BF16_UNITS * kStageVec];
  int block = blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int i = 0; i < kMaxGroup - 1; ++i)
    if (i + 1 < g.n && (int)blockIdx.x >= g.block_end[i]) pi = i + 1;
  if (pi > 0) block -= g.block_end[pi - 1];
  switch (pi) {
    case 0: convlstm16_dgrad_dispatch<SHIFT, 1>(g.p[0], block, lds); break;
    case 1: convlstm16_dgrad_dispatch<SHIFT, 1>(g.p[1], block, lds); break;
    case 2: convlstm16_dgrad_dispatch<SHIFT, 1>(g.p[2], block, lds); break;
    default: convlstm16_dgrad_dispatch<SHIFT, 1>(g.p[3], block, lds); break;
  }
}

static inline unsigned convlstm16_blocks(const ConvLstmArgs& a) {
  const size_t M = (size_t)a.rows * a.H * a.W;
  return (unsigned)((M + kBlockRows16 - 1) / kBlockRows16) * (unsigned)a.n_colblocks;
}
// MV_CONV_MAP=1: row tile per XCD (step_block_map); grid padded to 8 row tiles
static inline int conv_step_map_mode() {
  static const int m = getenv("MV_CONV_MAP") ? atoi(getenv("MV_CONV_MAP")) : 0;
  return m;
}
static inline unsigned convlstm16_step_blocks(const ConvLstmArgs& a, int mode) {
  if (mode == 0) return convlstm16_blocks(a);
  const size_t M = (size_t)a.rows * a.H * a.W;
  const size_t mt = (M + kBlockRows16 - 1) / kBlockRows16;
  return (unsigned)(((mt + 7) / 8) * 8) * (unsigned)a.n_colblocks;
}

static inline void launch_convlstm16_dgrads(const ConvLstm16Args* probs, int n,
                                            hipStream_t stream, bool bf16 = false) {
  ConvLstm16Group g{};
  g.n = n;
  unsigned total = 0;
  for (int i = 0; i < n; ++i) {
    g.p[i] = probs[i];
    total += convlstm16_blocks(probs[i].f) * (unsigned)(probs[i].n_kslice > 1 ? probs[i].n_kslice : 1);
    g.block_end[i] = (int32_t)total;
  }
  for (int i = n; i < kMaxGroup; ++i) g.block_end[i] = (int32_t)total;
  const bool shift = conv_group_shift(probs, n);
  if (bf16 && shift)
    hipLaunchKernelGGL(convlstm_dgrad_bf16_kernel<true>, dim3(total), dim3(kThreads16), 0, stream, g);
  else if (bf16)
    hipLaunchKernelGGL(convlstm_dgrad_bf16_kernel<false>, dim3(total), dim3(kThreads16), 0, stream, g);
  else if (shift)
    hipLaunchKernelGGL(convlstm_dgrad_f16x3_kernel<true>, dim3(total), dim3(kThreads16), 0, stream, g);
  else
    hipLaunchKernelGGL(convlstm_dgrad_f16x3_kernel<false>, dim3(total), dim3(kThreads16), 0, stream, g);
}

// out[seg][i] = sum over the k slices (in slice order) of part[seg][s][i]
struct SumSlicesArgs {
  const float* part[2 * kMaxGroup];
  float* out[2 * kMaxGroup];
  unsigned long long n[2 * kMaxGroup];     // elements per slice (multiple of 4)
  unsigned block_end[2 * kMaxGroup];
  int nseg, nslice;
  int nslice_seg[2 * kMaxGroup];           // per segment; 0 = nslice
};
__global__ __launch_bounds__(256)
void sum_slices_kernel(const SumSlicesArgs a) {
  int seg = 0;
#pragma unroll
  for (int i = 0; i < 2 * kMaxGroup - 1; ++i)
    if (i + 1 < a.nseg && blockIdx.x >= a.block_end[i]) seg = i + 1;
  const unsigned b = blockIdx.x - (seg ? a.block_end[seg - 1] : 0u);
  const size_t i4 = (size_t)b * 256 + threadIdx.x;
  const size_t n4 = a.n[seg] / 4;
  if (i4 >= n4) return;
  const f32x4* src = reinterpret_cast<const f32x4*>(a.part[seg]);
  f32x4 v = src[i4];
  const int ns = a.nslice_seg[seg] > 0 ? a.nslice_seg[seg] : a.nslice;
  for (int s2 = 1; s2 < ns; ++s2) {
    const f32x4 w = src[(size_t)s2 * n4 + i4];
    v[0] += w[0]; v[1] += w[1]; v[2] += w[2]; v[3] += w[3];
  }
  reinterpret_cast<f32x4*>(a.out[seg])[i4] = v;
}

// Pack of the transposed, tap-flipped kernel as fp16 planes (cf.
// pack_convlstm_dgrad_weights): k-step s = (group of 16 gate columns, tap'),
// k = 8*(l>>5) + e -> gate column n = grp*16 + k; output column
// col = cb*128 + g*32 + (l&31): col < C -> input channel Cx + col, else col - C.
__global__ void pack_f16x3_dgrad_kernel(const float* __restrict__ w, _Float16* __restrict__ out,
                                        int Cx, int C, size_t total) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int e = idx & 7;
  const int l = (idx >> 3) & 63;
  const int g = (idx >> 9) & 3;
  const size_t t = idx >> 11;                    // cb * nk + s
  const int nk = 9 * (4 * C / 16);
  const int s = t % nk, cb = t / nk;
  const int grp = s / 9, tap = s - grp * 9;
  const int k = 8 * (l >> 5) + e;
  const int n = grp * 16 + k;
  const int col = cb * kBN + g * 32 + (l & 31);
  const int Cin = Cx + C, N4 = 4 * C;
  int ci = -1;
  if (col < C) ci = Cx + col;
  else if (col - C < Cx) ci = col - C;
  const float v = (ci < 0) ? 0.f : w[((size_t)(8 - tap) * Cin + ci) * N4 + n] * kF16Scale;
  note_pack_range(v);
  const _Float16 v0 = (_Float16)v;
  const size_t base = t * (2 * 4 * 64 * 8);
  out[base + ((size_t)(0 * 4 + g) * 64 + l) * 8 + e] = v0;
  out[base + ((size_t)(1 * 4 + g) * 64 + l) * 8 + e] = (_Float16)(v - (float)v0);
}
static inline size_t f16x3_dgrad_wpack_elems(int Cx, int C) {   // in halves
  return (size_t)convlstm_dgrad_colblocks(Cx, C) * 9 * (size_t)(4 * C / 16) * 2 * 4 * 64 * 8;
}
// The same pack as ONE unscaled bf16 plane: [cb][k-step][sub-block][lane][8]
__global__ void pack_bf16_dgrad_kernel(const float* __restrict__ w, _Float16* __restrict__ out,
                                       int Cx, int C, size_t total) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int e = idx & 7;
  const int l = (idx >> 3) & 63;
  const int g = (idx >> 9) & 3;
  const size_t t = idx >> 11;                    // cb * nk + s
  const int nk = 9 * (4 * C / 16);
  const int s = t % nk, cb = t / nk;
  const int grp = s / 9, tap = s - grp * 9;
  const int k = 8 * (l >> 5) + e;
  const int n = grp * 16 + k;
  const int col = cb * kBN + g * 32 + (l & 31);
  const int Cin = Cx + C, N4 = 4 * C;
  int ci = -1;
  if (col < C) ci = Cx + col;
  else if (col - C < Cx) ci = col - C;
  out[idx] = bf16_as_half((ci < 0) ? 0.f : w[((size_t)(8 - tap) * Cin + ci) * N4 + n]);
}
static inline size_t bf16_dgrad_wpack_elems(int Cx, int C) {    // in halves
  return f16x3_dgrad_wpack_elems(Cx, C) / 2;
}
// fp32 [M][C] -> one bf16 plane in the tiled operand layout (plane_index)
__global__ void split_plane_bf16_kernel(const float* __restrict__ in, _Float16* __restrict__ p0,
                                        int M, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int c8n = C >> 3;
  const size_t per_tile = (size_t)32 * c8n;
  const size_t tb = i / per_tile;
  const int r = (int)(i - tb * per_tile);
  const int c8 = r >> 5, cell = r & 31;
  const long long m = (long long)tb * 32 + cell;
  if (m >= M) return;
  const f32x4* src = reinterpret_cast<const f32x4*>(in + (size_t)m * C + c8 * 8);
  const f32x4 v0 = src[0], v1 = src[1];
  f16x8 a;
#pragma unroll
  for (int j = 0; j < 8; ++j) a[j] = bf16_as_half(j < 4 ? v0[j] : v1[j - 4]);
  *reinterpret_cast<f16x8*>(p0 + plane_index(m, c8 * 8, C)) = a;
}

// fp32 -> two fp16 planes under a per-tensor power-of-two scale 2^e chosen from
// the tensor's max |.| (tracked as int bits by lstm_gate_bwd_kernel): the largest
// element lands in [2^13, 2^14).  e is stored for the consumer's epilogue.
__global__ void split_planes_dyn_kernel(const float* __restrict__ in, _Float16* __restrict__ p0,
                                        _Float16* __restrict__ p1, int M, int C,
                                        const int32_t* __restrict__ max_bits,
                                        int32_t* __restrict__ exp_out) {
  int mb = 0;
  for (int i = 0; i < 64; ++i) mb = max(mb, max_bits[i]);    // uniform: scalar loads
  const float mx = __int_as_float(mb);
  int e = 0;
  if (mx > 0.f) {
    e = 13 - ilogbf(mx);
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
  }
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) exp_out[0] = e;
  const int c8n = C >> 3;
  const size_t per_tile = (size_t)32 * c8n;
  const size_t tb = i / per_tile;
  const int r = (int)(i - tb * per_tile);
  const int c8 = r >> 5, cell = r & 31;
  const long long m = (long long)tb * 32 + cell;
  if (m >= M) return;
  const f32x4* src = reinterpret_cast<const f32x4*>(in + (size_t)m * C + c8 * 8);
  const f32x4 v0 = src[0], v1 = src[1];
  f16x8 a, b;
  // x * 2^e as one multiply (|e| <= 100: 2^e is a normal float, the product is exactly
  // ldexpf's); ocml ldexpf was ~12 instructions per element of an HBM-bound kernel
  const float sc2e = __int_as_float((127 + e) << 23);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float sv = (j < 4 ? v0[j] : v1[j - 4]) * sc2e;
    const _Float16 h0 = (_Float16)sv;
    a[j] = h0;
    b[j] = (_Float16)(sv - (float)h0);
  }
  const size_t o = plane_index(m, c8 * 8, C);
  *reinterpret_cast<f16x8*>(p0 + o) = a;
  *reinterpret_cast<f16x8*>(p1 + o) = b;
}

static inline void launch_convlstm_bf16_steps(const ConvLstm16Args* probs, int n,
                                              hipStream_t stream) {
  ConvLstm16Group g{};
  g.n = n;
  g.map_mode = conv_step_map_mode();
  unsigned total = 0;
  for (int i = 0; i < n; ++i) {
    g.p[i] = probs[i];
    total += convlstm16_step_blocks(probs[i].f, g.map_mode);
    g.block_end[i] = (int32_t)total;
  }
  for (int i = n; i < kMaxGroup; ++i) g.block_end[i] = (int32_t)total;
  // unbounded-activation models (a problem carries an x exponent): the three-pass x kernel
  bool xf16 = false;
  for (int i = 0; i < n; ++i) xf16 = xf16 || probs[i].x_exp != nullptr;
  const bool shift = conv_group_shift(probs, n);
  if (xf16) {
    if (shift)
      hipLaunchKernelGGL((convlstm_step_bf16_kernel<true, true>), dim3(total), dim3(kThreads16), 0, stream, g);
    else
      hipLaunchKernelGGL((convlstm_step_bf16_kernel<false, true>), dim3(total), dim3(kThreads16), 0, stream, g);
  } else if (shift) {
    hipLaunchKernelGGL(convlstm_step_bf16_kernel<true>, dim3(total), dim3(kThreads16), 0, stream, g);
  } else {
    hipLaunchKernelGGL(convlstm_step_bf16_kernel<false>, dim3(total), dim3(kThreads16), 0, stream, g);
  }
}

static inline void launch_convlstm16_steps(const ConvLstm16Args* probs, int n,
                                           hipStream_t stream) {
  ConvLstm16Group g{};
  g.n = n;
  g.map_mode = conv_step_map_mode();
  unsigned total = 0;
  for (int i = 0; i < n; ++i) {
    g.p[i] = probs[i];
    total += convlstm16_step_blocks(probs[i].f, g.map_mode);
    g.block_end[i] = (int32_t)total;
  }
  for (int i = n; i < kMaxGroup; ++i) g.block_end[i] = (int32_t)total;
  if (conv_group_shift(probs, n))
    hipLaunchKernelGGL(convlstm_step_f16x3_lds_kernel<true>, dim3(total), dim3(kThreads16), 0, stream, g);
  else
    hipLaunchKernelGGL(convlstm_step_f16x3_lds_kernel<false>, dim3(total), dim3(kThreads16), 0, stream, g);
}

}  // namespace mv
