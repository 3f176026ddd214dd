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
// A wave owns 32 cells x 128 columns, the 128 columns being the FOUR gates of
// ONE block of 32 channels (column order fixed at weight-pack time), so a
// lane's accumulators hold i,j,f,o of the same (cell, channel) and the LSTM
// update runs in registers: c is read once, c'/h' are written once, the
// [M,4C] pre-activation tensor never exists in memory.
// K is walked in chunks of 32 = (channel group of 32) x (tap), channel-major so
// the nine shifted re-reads of one activation slab hit L1/L2; each chunk is
// four k-steps of 8 (16 MFMAs per wave per k-step).
//
// Operands are REGISTER-DIRECT: no LDS, no workgroup barrier.  fp32 MFMA runs
// at 1/16 of the bf16 rate (64 cycles per 32x32x2), so one k-step of a wave
// (1024 pipe cycles) needs only 5 KB of operands; at that intensity the LDS
// round trip and its barriers cost more matrix-pipe idle time than they save
// in traffic (measured on MI355X, DESIGN.md section 5: LDS-staged 128x128 tile
// 111 TF with MFMA busy 73 % and 11 % of wave cycles parked at s_barrier;
// register-direct 122 TF, MFMA busy 80 %, 2.5 % parked).  The weights are
// pre-packed in MFMA *fragment order* so every B load is one coalesced 1 KB
// wave read that the waves of a CU (same channel block) share through L1; the
// A fragment is 16 B of the lane's own cell.  The next k-step's five loads are
// in flight while the current one multiplies.
//
// Up to kMaxGroup independent steps (the class / regression chains of the two
// grid scales advance in lockstep) are issued as ONE launch: the small 9x16
// problems alone fill only 2.25 workgroups per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace mv {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kBN = 128;       // gate columns per wave tile (4 gates x 32 ch)
constexpr int kBK = 32;        // K per chunk
constexpr int kChBlock = 32;   // channels per workgroup
constexpr int kWaveRows = 32;  // cells per wave
constexpr int kBlockRows = 128;  // cells per workgroup (4 independent waves)
constexpr int kMaxGroup = 4;

struct ConvLstmArgs {
  const float* x;       // [rows, H, W, Cx] dense input
  const float* h;       // [src rows, H, W, C]
  const float* c;       // [src rows, H, W, C]
  const int32_t* src_row_h;  // optional [rows]: row indirection for h (beam parents)
  const int32_t* src_row_c;  // optional [rows]: row indirection for c (beam parents)
  const float* wpack;   // fragment-order weights, see pack_convlstm_weights()
  const float* bias;    // [4C] TF order (i|j|f|o)
  float* h_out;         // [rows, H, W, C]
  float* c_out;         // [rows, H, W, C]
  int32_t rows, H, W, Cx, C;
  int32_t x_row_stride; // elements between consecutive rows of x (0: H*W*Cx), so a
                        // time slice of an [N, T, H, W, Cx] tensor is read in place
  int32_t n_xchunks;    // K chunks taken from x
  int32_t n_hchunks;    // K chunks taken from h (0 when the state is known zero)
  int32_t w_chunks;     // chunks per channel block in wpack (x + all h chunks)
  int32_t x_small;      // 1: 9*Cx <= 32, all taps of x packed in ONE chunk
  int32_t zero_state;   // 1: h == c == 0 (first encoder step): skip h, c reads
  float forget_bias;
};

struct ConvLstmGroup {
  ConvLstmArgs p[kMaxGroup];
  int32_t block_end[kMaxGroup];   // exclusive prefix of workgroups per problem
  int32_t n;
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
static inline size_t convlstm_wpack_elems(int Cx, int C) {
  const size_t nch = (size_t)convlstm_xchunks(Cx) + 9 * (size_t)(C / kBK);
  return (size_t)(C / kChBlock) * nch * kBN * kBK;
}

// Host-side weight pack: TF HWIO kernel [3,3,Cx+C,4C] -> MFMA fragment order
//   wpack[cb][chunk][kk(4)][gate(4)][lane(64)][4]
// element (lane, j) = W[k = kk*8 + (lane>>5)*4 + j][n = gate*C + cb*32 + (lane&31)]
// where chunk-local k maps to (tap, input channel):
//   x chunks first -- channel-group-major, tap-minor (k = channel within the
//   group), or, for Cx <= 3, ONE packed chunk with k = tap*Cx + ch -- then the
//   h chunks, channel-group-major, tap-minor.
static inline void pack_convlstm_weights(const float* w, int Cx, int C, float* out) {
  const int Cin = Cx + C, N4 = 4 * C;
  const int nx = convlstm_xchunks(Cx), nh = 9 * (C / kBK), nch = nx + nh;
  const bool small = (Cx > 0 && 9 * Cx <= kBK);
  for (int cb = 0; cb < C / kChBlock; ++cb)
    for (int q = 0; q < nch; ++q) {
      float* tile = out + ((size_t)cb * nch + q) * kBN * kBK;
      for (int kk = 0; kk < 4; ++kk)
        for (int g = 0; g < 4; ++g)
          for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
              const int k = kk * 8 + (l >> 5) * 4 + j;
              const int n = g * C + cb * kChBlock + (l & 31);
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
              tile[((kk * 4 + g) * 64 + l) * 4 + j] =
                  (tap < 0) ? 0.f : w[((size_t)tap * Cin + ci) * N4 + n];
            }
    }
}

// fp32 tanh / sigmoid from the ocml math library (1-2 ulp).
__device__ __forceinline__ float tanh_(float v) { return tanhf(v); }
__device__ __forceinline__ float sigm_(float v) { return 1.0f / (1.0f + expf(-v)); }

struct ConvFrag {
  f32x4 a;        // A: 4 consecutive k of this lane's cell
  f32x4 b[4];     // B: one fragment per gate
  uint32_t ok;    // all-ones when the tap is inside the image, else 0
};

__device__ __forceinline__ void convlstm_step_body(const ConvLstmArgs& a, int block) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  // block -> (channel block, m tile).  Workgroups are observed to round-robin
  // over the 8 XCDs by linear id, so cb = id % 8 keeps each XCD's L2 on ONE 1/8
  // slice of the packed weights (speed only, never correctness).
  const int ncb = a.C / kChBlock;
  const int cb = block % ncb;
  const int mt = block / ncb;
  const int H = a.H, W = a.W, HW = H * W, C = a.C, Cx = a.Cx;
  const int M_total = a.rows * HW;
  const int nx = a.n_xchunks;
  const int nchunks = nx + a.n_hchunks;
  const int m_wave = mt * kBlockRows + wave * kWaveRows;
  if (m_wave >= M_total) return;   // whole wave past the end (no barriers here)

  // each lane owns cell (lane & 31) of its wave's 32 cells; xoff / hoff are the
  // element offsets of that cell in x and h (all offsets < 2^31, host-checked)
  int ypos, xpos, xoff, hoff;
  {
    const int m = m_wave + (lane & 31);
    if (m < M_total) {
      const int r = m / HW, cell = m - r * HW;
      const int y = cell / W;
      ypos = y; xpos = cell - y * W;
      const int sr = a.src_row_h ? a.src_row_h[r] : r;
      xoff = r * a.x_row_stride + cell * Cx;
      hoff = (sr * HW + cell) * C;
    } else {
      ypos = -100000; xpos = -100000; xoff = 0; hoff = 0;
    }
  }
  const int khalf = (lane >> 5) * 4;
  const f32x4* wblk = reinterpret_cast<const f32x4*>(
      a.wpack + (size_t)cb * a.w_chunks * kBN * kBK) + lane;
  const bool xsmall = a.x_small != 0;

  // k-step s = chunk q * 4 + kk.  Branch-free gather: an out-of-image tap reads
  // element 0 of the tensor (always mapped) and is zeroed by a bit mask when it
  // is CONSUMED, so the prefetch never waits for its own data and the loop body
  // has no divergent control flow around its loads.
  auto load_step = [&](int s, ConvFrag& f) {
    const int q = s >> 2, kk = s & 3;
    const f32x4* wsrc = wblk + (size_t)s * (4 * 64);
#pragma unroll
    for (int g = 0; g < 4; ++g) f.b[g] = wsrc[g * 64];
    if (xsmall && q == 0) {
      // all 9 taps x Cx (<= 3) channels of x packed into one chunk: k = tap*Cx + ch
      f32x4 v;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = kk * 8 + khalf + j;
        const int tap = k / Cx, ch = k - tap * Cx;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int yy = ypos + dy, xx = xpos + dx;
        const bool ok = (k < 9 * Cx) & (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W);
        int off = xoff + (dy * W + dx) * Cx + ch;
        off = ok ? off : 0;
        const float tv = a.x[off];
        v[j] = ok ? tv : 0.f;
      }
      f.a = v;
      f.ok = 0xffffffffu;
    } else {
      const bool is_x = q < nx;
      const int qq = is_x ? q : q - nx;
      const int cg = qq / 9, tap = qq - cg * 9;
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      const float* base = is_x ? a.x : a.h;
      const int cs = is_x ? Cx : C;
      const int yy = ypos + dy, xx = xpos + dx;
      const bool ok = (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W);
      int off = (is_x ? xoff : hoff) + (dy * W + dx) * cs + cg * kBK + kk * 8 + khalf;
      off = ok ? off : 0;
      f.a = *reinterpret_cast<const f32x4*>(base + off);
      f.ok = ok ? 0xffffffffu : 0u;
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;

  auto mma_step = [&](const ConvFrag& f) {
    f32x4 am;
#pragma unroll
    for (int j = 0; j < 4; ++j) am[j] = __uint_as_float(__float_as_uint(f.a[j]) & f.ok);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g)
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(am[j], f.b[g][j], acc[g], 0, 0, 0);
  };

  // Two-deep register ring, unrolled by two so the buffer index is static.
  // The (clamped) prefetch is unconditional: a straight-line body lets the
  // compiler use counted vmcnt waits; the final re-read is harmless.
  const int nsteps = nchunks * 4;
  ConvFrag f0, f1;
  if (nsteps > 0) load_step(0, f0);
  for (int s = 0; s < nsteps; s += 2) {     // nsteps is a multiple of 4
    load_step(s + 1, f1);
    mma_step(f0);
    load_step(min(s + 2, nsteps - 1), f0);
    mma_step(f1);
  }

  // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  const int ch = cb * kChBlock + (lane & 31);
  const float bi = a.bias[ch], bj = a.bias[C + ch], bf = a.bias[2 * C + ch],
              bo = a.bias[3 * C + ch];
#pragma unroll
  for (int reg = 0; reg < 16; ++reg) {
    const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
    const int m = m_wave + row;
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

__global__ __launch_bounds__(256, 2)
void convlstm_step_kernel(const ConvLstmGroup g) {
  // problem lookup: wave-uniform compares against the workgroup prefix sums
  int block = blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int i = 0; i < kMaxGroup - 1; ++i)
    if (i + 1 < g.n && (int)blockIdx.x >= g.block_end[i]) pi = i + 1;
  if (pi > 0) block -= g.block_end[pi - 1];
  // select by value so every field of the chosen problem lives in SGPRs
  switch (pi) {
    case 0: convlstm_step_body(g.p[0], block); break;
    case 1: convlstm_step_body(g.p[1], block); break;
    case 2: convlstm_step_body(g.p[2], block); break;
    default: convlstm_step_body(g.p[3], block); break;
  }
}

static inline unsigned convlstm_blocks(const ConvLstmArgs& a) {
  const size_t M = (size_t)a.rows * a.H * a.W;
  return (unsigned)((M + kBlockRows - 1) / kBlockRows) * (unsigned)(a.C / kChBlock);
}

// Fill the derived fields of one problem.
static inline void convlstm_finish_args(ConvLstmArgs& a, bool zero_state) {
  a.n_xchunks = convlstm_xchunks(a.Cx);
  a.n_hchunks = zero_state ? 0 : 9 * (a.C / kBK);
  a.w_chunks = a.n_xchunks + 9 * (a.C / kBK);
  a.x_small = (a.Cx > 0 && 9 * a.Cx <= kBK) ? 1 : 0;
  a.zero_state = zero_state ? 1 : 0;
  if (a.x_row_stride == 0) a.x_row_stride = a.H * a.W * a.Cx;
  a.forget_bias = 1.0f;   // tf.contrib.rnn.ConvLSTMCell default
}

// Launch n (<= kMaxGroup) independent ConvLSTM steps as one grid.
static inline void launch_convlstm_steps(const ConvLstmArgs* probs, int n,
                                         hipStream_t stream) {
  ConvLstmGroup g{};
  g.n = n;
  unsigned total = 0;
  for (int i = 0; i < n; ++i) {
    g.p[i] = probs[i];
    total += convlstm_blocks(probs[i]);
    g.block_end[i] = (int32_t)total;
  }
  for (int i = n; i < kMaxGroup; ++i) g.block_end[i] = (int32_t)total;
  hipLaunchKernelGGL(convlstm_step_kernel, dim3(total), dim3(256), 0, stream, g);
}

}  // namespace mv
