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
  float* gates_out;     // optional [rows, H, W, 4, C]: sigm(i), tanh(j), sigm(f+fb),
                        // sigm(o) saved for the backward pass (training forward)
  // --- plain-store epilogue (dgrad of the gate conv, see convlstm_dgrad_kernel):
  // column block cb covers output columns [cb*128, cb*128+128); columns
  // < out0_cols go to out0 [M, out0_cols], the next out1_cols to out1.
  float* out0;
  float* out1;
  int32_t out0_cols, out1_cols;
  int32_t n_colblocks;  // column blocks of 128 (LSTM epilogue: C/32)
  int32_t ng_last;      // active 32-column sub-blocks in the last column block
  int32_t rows, H, W, Cx, C;
  int32_t x_row_stride; // elements between consecutive rows of x (0: H*W*Cx), so a
                        // time slice of an [N, T, H, W, Cx] tensor is read in place
  int32_t n_xchunks;    // K chunks taken from x
  int32_t n_hchunks;    // K chunks taken from h (0 when the state is known zero)
  int32_t w_chunks;     // chunks per channel block in wpack (x + all h chunks)
  int32_t x_small;      // 1: 9*Cx <= 32, all taps of x packed in ONE chunk
  int32_t zero_state;   // 1: h == c == 0 (first encoder step): skip h, c reads
  float forget_bias;
  int32_t want_h16;     // host-side hint (f16x3 mode): the next consumer of h' is a gate
                        // convolution, emit its operand planes from the epilogue
  int32_t skip_h32;     // f16x3 / bf16 inference: NOTHING reads the fp32 h' of this step (its
                        // only consumer is the next gate convolution, through the planes):
                        // the epilogue does not store it (encoder steps)
  // --- sparse x (f16x3 / bf16 inference, class chains).  The x operand of the class
  // encoder is zero except at ONE cell per row, the class decoder's is a constant
  // vector except within one cell of the hot cell: their k-steps (20 % / 11 % of the
  // launch) multiply zeros or recompute constants.  With sx_corr set the kernel skips
  // the x k-steps and the epilogue adds
  //   sx_bias[border class of the cell][4C]        (bias + conv of the constant part)
  //   sx_corr[index][offset of the cell from the hot cell][4C]   within sx_rad cells,
  // index = row (encoder: per-row table of the step) or border class of the hot cell
  // (decoder: weights-only table).  Border class = 3 * (y == 0 ? 0 : y == H-1 ? 2 : 1) +
  // (x == 0 ? 0 : x == W-1 ? 2 : 1).
  const float* sx_bias;        // [9][4C] or null (plain bias)
  const float* sx_corr;        // null: dense x
  const int32_t* sx_hot;       // hot cell of row r: sx_hot[(r / sx_hot_div) * sx_hot_stride]
  const uint32_t* sx_cellyx;   // [H*W]: y << 16 | x
  int32_t sx_hot_stride, sx_hot_div, sx_rad, sx_by_class;
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

// fp32 sigmoid / tanh of the LSTM update on the hardware transcendentals: v_exp_f32 (2^x)
// and v_rcp_f32 are 1 ulp each, so sigm_ is good to ~3 ulp and tanh_ to ~2e-7 ABSOLUTE
// (the (1 - e) difference cancels for |v| << 1; gates and states are O(1) quantities held
// to a 1e-4 absolute bar).  The ocml tanhf / expf / IEEE division they replace were ~400
// VALU instructions per (cell, channel): with every workgroup of a round reaching its
// epilogue at the same time that was 0.25 of the 0.98 ms gate launch
// (profiles/r3_ablation_gate_kernel_s2.md).  Saturation: 2^(+big) = inf -> rcp = 0.
__device__ __forceinline__ float sigm_(float v) {
  const float e = __builtin_amdgcn_exp2f(-1.4426950408889634f * v);       // e^-v
  return __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float tanh_(float v) {
  const float e = __builtin_amdgcn_exp2f(-2.8853900817779268f * __builtin_fabsf(v));  // e^-2|v|
  const float t = (1.0f - e) * __builtin_amdgcn_rcpf(1.0f + e);
  return __builtin_copysignf(t, v);
}

template <int NG>
struct ConvFrag {
  f32x4 a;        // A: 4 consecutive k of this lane's cell
  f32x4 b[NG];    // B: one fragment per gate / 32-column sub-block
  uint32_t ok;    // all-ones when the tap is inside the image, else 0
};

constexpr int kEpiLstm = 0;    // fused LSTM update (forward)
constexpr int kEpiStore = 1;   // plain store of the NG x 32 columns (dgrad)

template <int EPI, int NG>
__device__ __forceinline__ void convlstm_step_body(const ConvLstmArgs& a, int block) {
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  // block -> (channel block, m tile).  Workgroups are observed to round-robin
  // over the 8 XCDs by linear id, so cb = id % 8 keeps each XCD's L2 on ONE 1/8
  // slice of the packed weights (speed only, never correctness).
  const int ncb = a.n_colblocks;
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
  auto load_step = [&](int s, ConvFrag<NG>& f) {
    const int q = s >> 2, kk = s & 3;
    const f32x4* wsrc = wblk + (size_t)s * (4 * 64);
#pragma unroll
    for (int g = 0; g < NG; ++g) f.b[g] = wsrc[g * 64];
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

  f32x16 acc[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[g][i] = 0.f;

  auto mma_step = [&](const ConvFrag<NG>& f) {
    f32x4 am;
#pragma unroll
    for (int j = 0; j < 4; ++j) am[j] = __uint_as_float(__float_as_uint(f.a[j]) & f.ok);
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int g = 0; g < NG; ++g)
        acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(am[j], f.b[g][j], acc[g], 0, 0, 0);
  };

  // Two-deep register ring, unrolled by two so the buffer index is static.
  // The (clamped) prefetch is unconditional: a straight-line body lets the
  // compiler use counted vmcnt waits; the final re-read is harmless.
  const int nsteps = nchunks * 4;
  ConvFrag<NG> f0, f1;
  if (nsteps > 0) load_step(0, f0);
  for (int s = 0; s < nsteps; s += 2) {     // nsteps is a multiple of 4
    load_step(s + 1, f1);
    mma_step(f0);
    load_step(min(s + 2, nsteps - 1), f0);
    mma_step(f1);
  }

  // ---- epilogue: C/D layout col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)
  if constexpr (EPI == kEpiLstm) {
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
        const float si = sigm_(gi), tj = tanh_(gj), sf = sigm_(gf + a.forget_bias),
                    so = sigm_(go);
        float cn = sf * cprev;
        cn = cn + si * tj;
        const float hn = tanh_(cn) * so;
        a.c_out[(size_t)m * C + ch] = cn;
        a.h_out[(size_t)m * C + ch] = hn;
        if (a.gates_out) {
          float* gp = a.gates_out + (size_t)m * 4 * C + ch;
          gp[0] = si; gp[C] = tj; gp[2 * C] = sf; gp[3 * C] = so;
        }
      }
    }
  } else {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int col = cb * kBN + g * 32 + (lane & 31);
      float* dst = nullptr;
      int stride = 0, cc = col;
      if (col < a.out0_cols) { dst = a.out0; stride = a.out0_cols; }
      else if (col - a.out0_cols < a.out1_cols) {
        dst = a.out1; stride = a.out1_cols; cc = col - a.out0_cols;
      }
      if (dst) {
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
          const int row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5);
          const int m = m_wave + row;
          if (m < M_total) dst[(size_t)m * stride + cc] = acc[g][reg];
        }
      }
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
    case 0: convlstm_step_body<kEpiLstm, 4>(g.p[0], block); break;
    case 1: convlstm_step_body<kEpiLstm, 4>(g.p[1], block); break;
    case 2: convlstm_step_body<kEpiLstm, 4>(g.p[2], block); break;
    default: convlstm_step_body<kEpiLstm, 4>(g.p[3], block); break;
  }
}

// dgrad of the gate convolution: d[h | x] = conv3x3_SAME(G, W^T flipped), the
// same implicit GEMM with the saved gate gradients G [M, 4C] as the "h"
// operand (4C input channels) and the Cx + C input channels as output
// columns (pack_convlstm_dgrad_weights).  The last column block usually has
// fewer than four active 32-column sub-blocks (Cx = 32 -> 1, Cx = 64 -> 2).
__device__ __forceinline__ void convlstm_dgrad_dispatch(const ConvLstmArgs& a, int block) {
  const int cb = block % a.n_colblocks;
  if (cb == a.n_colblocks - 1 && a.ng_last == 1)
    convlstm_step_body<kEpiStore, 1>(a, block);
  else if (cb == a.n_colblocks - 1 && a.ng_last == 2)
    convlstm_step_body<kEpiStore, 2>(a, block);
  else
    convlstm_step_body<kEpiStore, 4>(a, block);
}

__global__ __launch_bounds__(256, 2)
void convlstm_dgrad_kernel(const ConvLstmGroup g) {
  int block = blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int i = 0; i < kMaxGroup - 1; ++i)
    if (i + 1 < g.n && (int)blockIdx.x >= g.block_end[i]) pi = i + 1;
  if (pi > 0) block -= g.block_end[pi - 1];
  switch (pi) {
    case 0: convlstm_dgrad_dispatch(g.p[0], block); break;
    case 1: convlstm_dgrad_dispatch(g.p[1], block); break;
    case 2: convlstm_dgrad_dispatch(g.p[2], block); break;
    default: convlstm_dgrad_dispatch(g.p[3], block); break;
  }
}

static inline unsigned convlstm_blocks(const ConvLstmArgs& a) {
  const size_t M = (size_t)a.rows * a.H * a.W;
  return (unsigned)((M + kBlockRows - 1) / kBlockRows) * (unsigned)a.n_colblocks;
}

// Fill the derived fields of one problem.
static inline void convlstm_finish_args(ConvLstmArgs& a, bool zero_state) {
  a.n_xchunks = convlstm_xchunks(a.Cx);
  a.n_hchunks = zero_state ? 0 : 9 * (a.C / kBK);
  a.w_chunks = a.n_xchunks + 9 * (a.C / kBK);
  a.x_small = (a.Cx > 0 && 9 * a.Cx <= kBK) ? 1 : 0;
  a.zero_state = zero_state ? 1 : 0;
  if (a.x_row_stride == 0) a.x_row_stride = a.H * a.W * a.Cx;
  a.n_colblocks = a.C / kChBlock;
  a.ng_last = 4;
  a.forget_bias = 1.0f;   // tf.contrib.rnn.ConvLSTMCell default
}

// ---- dgrad problem: G [rows,H,W,4C] -> dh [rows,H,W,C], dx [rows,H,W,Cx]
static inline int convlstm_dgrad_colblocks(int Cx, int C) {
  return (C + Cx + kBN - 1) / kBN;
}
static inline size_t convlstm_dgrad_wpack_elems(int Cx, int C) {
  return (size_t)convlstm_dgrad_colblocks(Cx, C) * 9 * (size_t)(4 * C / kBK) * kBN * kBK;
}
// Host-side pack of the transposed, tap-flipped kernel for the dgrad launch:
//   wd[cb][chunk][kk][g][lane][4], chunk = (gate-column group of 32, tap'),
//   element (lane, j): k = kk*8 + (lane>>5)*4 + j -> gate column n = grp*32 + k,
//   output column col = cb*128 + g*32 + (lane&31):
//     col <  C       -> input channel Cx + col  (an h channel)
//     col <  C + Cx  -> input channel col - C   (an x channel)
//   value = W[8 - tap'][ci][n]   (d in[m] = sum_tap G[m - d_tap] W[tap], so the
//   launch's tap' reads offset d_tap' = -d_tap).
static inline void pack_convlstm_dgrad_weights(const float* w, int Cx, int C, float* out) {
  const int Cin = Cx + C, N4 = 4 * C;
  const int ncb = convlstm_dgrad_colblocks(Cx, C), nch = 9 * (N4 / kBK);
  for (int cb = 0; cb < ncb; ++cb)
    for (int q = 0; q < nch; ++q) {
      float* tile = out + ((size_t)cb * nch + q) * kBN * kBK;
      const int grp = q / 9, tap = q % 9;
      for (int kk = 0; kk < 4; ++kk)
        for (int g = 0; g < 4; ++g)
          for (int l = 0; l < 64; ++l)
            for (int j = 0; j < 4; ++j) {
              const int k = kk * 8 + (l >> 5) * 4 + j;
              const int n = grp * kBK + k;
              const int col = cb * kBN + g * 32 + (l & 31);
              int ci = -1;
              if (col < C) ci = Cx + col;
              else if (col - C < Cx) ci = col - C;
              tile[((kk * 4 + g) * 64 + l) * 4 + j] =
                  (ci < 0) ? 0.f : w[((size_t)(8 - tap) * Cin + ci) * N4 + n];
            }
    }
}
// Fill a dgrad problem: g = G [rows,H,W,4C]; dh [rows,H,W,C]; dx [rows,H,W,Cx]
// (dx may be NULL: its columns are still computed in the shared last block
// when Cx > 0 and need_dx, else the block is dropped).
static inline void convlstm_dgrad_args(ConvLstmArgs& a, const float* g, const float* wdpack,
                                       float* dh, float* dx, int rows, int H, int W,
                                       int Cx, int C, bool need_dh, bool need_dx) {
  a = ConvLstmArgs{};
  a.x = nullptr; a.h = g; a.c = nullptr; a.wpack = wdpack; a.bias = nullptr;
  a.rows = rows; a.H = H; a.W = W; a.Cx = 0; a.C = 4 * C;
  a.n_xchunks = 0; a.n_hchunks = 9 * (4 * C / kBK); a.w_chunks = a.n_hchunks;
  a.x_small = 0; a.zero_state = 0; a.x_row_stride = 0; a.forget_bias = 0.f;
  a.out0 = need_dh ? dh : nullptr; a.out0_cols = C;
  a.out1 = need_dx ? dx : nullptr; a.out1_cols = Cx;
  const int ncb = convlstm_dgrad_colblocks(Cx, C);
  a.n_colblocks = need_dx ? ncb : C / kBN;
  const int rem = (C + Cx) - (ncb - 1) * kBN;   // columns in the last block
  a.ng_last = need_dx ? (rem <= 32 ? 1 : rem <= 64 ? 2 : 4) : 4;
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

static inline void launch_convlstm_dgrads(const ConvLstmArgs* probs, int n,
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
  hipLaunchKernelGGL(convlstm_dgrad_kernel, dim3(total), dim3(256), 0, stream, g);
}

}  // namespace mv
