// ConvLSTM step, f16x3 arithmetic, with the 3x3 gate convolution in Winograd F(2,3) form
// along the image's ROW axis: two thirds of the matrix-pipe work of convlstm_f16x3.h for
// the same pre-activations.
//
// Why: the direct f16x3 gate kernel runs at the package
// power cap (MFMA busy 0.78 at 1.64-1.75 GHz), so scheduling work no longer pays; what is
// left is energy per result, i.e. MFMAs per product.  For a pair of output rows (2t, 2t+1)
// at column x and the three kernel rows g0, g1, g2 of one stencil column dx,
//     y(2t)   = M0 + M1 + M2        M0 = (d(2t-1) - d(2t+1)) g0
//     y(2t+1) = M1 - M2 - M3        M1 = (d(2t)   + d(2t+1)) (g0 + g1 + g2) / 2
//                                   M2 = (d(2t+1) - d(2t)  ) (g0 - g1 + g2) / 2
//                                   M3 = (d(2t)   - d(2t+2)) g2
// four products instead of six, each still summed over (dx, input channel): four
// independent GEMMs  M_c[pair-cell][column] = sum_{dx, ci} V_c[pair-cell + dx][ci] U_c[dx][ci][column].
// Rows, not columns, are paired so that the dx taps stay what they are in the direct
// kernel: ONE operand fragment per (component, 16 channels) moved one lane up / down the
// wave by DPP (a wave's 32 pair-cells are consecutive x of whole image rows; every W of
// the launch divides 32).
//
// Operands.  U_c is formed from the fp32 kernel in fp64 and stored, like every f16x3
// operand, as two fp16 planes of 256 U (pack_wino_kernel).  V_c is formed IN the kernel
// from the ordinary operand planes the producers already write (plane_layout.h; no
// producer changes, beam row indirection as before): with d = hi + lo per row,
//     V = (a_hi +- b_hi) + (a_lo +- b_lo)
// is evaluated in packed fp16 with an error-free TwoSum of the high planes (hi' = fl(a_hi
// +- b_hi), lo' = err + (a_lo +- b_lo)): |V - hi' - lo'| <= ~2^-21 max(|a|, |b|), the
// class of the plane residuals themselves.  32 packed fp16 instructions per component and 16 channels
// against 18 MFMAs (576 matrix-pipe cycles).
//
// Tile.  The matrix roles are swapped against the direct kernel: the weights are the A
// operand (rows = 4 gates x 8 channels), the activations the B operand (columns = 32
// pair-cells), so a lane's accumulator registers hold i, j, f, o of FOUR consecutive
// channels of ONE pair-cell: the LSTM update stays in registers, state loads / stores
// are 16-byte vectors, and the h' operand planes leave the registers as 8-byte stores that
// a wave lays down as contiguous 512-byte runs (no LDS transpose).  A wave owns 32
// pair-cells (64 cells) x 16 channels x 4 gates x 4 Winograd components = 128
// accumulator registers; a workgroup = 4 (or 8) waves = 128 (256) pair-cells of one 16-channel
// column block, sharing that block's weight stage (24 KB = 2 components x 3 dx x 2 planes x 2
// row blocks) through a double LDS buffer filled by LDS-DMA; 36 MFMAs per wave and stage.
// Two 4-wave workgroups (or one 8-wave workgroup) per CU: 2 waves per SIMD, <= 256 registers.
//
// The regression encoder's 2-channel pixel-offset input keeps its fp32 chunk
// (v_mfma_f32_32x32x2_f32, weights x 2^16 read straight from the HWIO kernel): direct
// form, row 2t into M0 and row 2t+1, negated, into M3.
#pragma once
#include "convlstm_f16x3.h"

namespace mv {

// Waves per workgroup (template parameter WAVES): 8 waves = 256 pair-cells share one weight
// stage, ONE workgroup per CU; 4 waves (the default) = 128 pair-cells, TWO workgroups per CU at
// twice the stage traffic from L2.
constexpr int kWnCh = 16;                            // output channels per workgroup
constexpr int kWnStageVec = 2 * 3 * 2 * 2 * 64;      // 16-byte vectors per LDS stage (24 KB)
constexpr uint32_t kWnStageBytes = kWnStageVec * 16;

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

struct ConvLstmWinoArgs {
  ConvLstm16Args b;        // geometry, operand planes, outputs (wp16 / wx32 unused)
  const _Float16* wpw;     // [cb16][stage][comp in stage 2][dx 3][plane 2][row block 2][lane 64][8]
  const float* w_hwio;     // the fp32 kernel [3,3,Cx+C,4C] (x_small chunk)
  const _Float16* v3x;     // F(3,3), pre-transformed operands (convlstm_wino3.h
  const _Float16* v3h;     // wino3_transform_kernel): components of the x / h planes, or null
  int32_t n_xc;            // 16-channel x chunks present in the pack (0 when x_small)
  int32_t nks_main, nks_x; // dgrad: split-K slices of the d h column blocks / of the d x blocks
};

struct ConvLstmWinoGroup {
  ConvLstmWinoArgs p[kMaxGroup];
  int32_t block_end[kMaxGroup];
  int32_t n;
  int32_t map_mode;          // 1: adjacent 16-channel blocks (one 128-byte state line) per XCD
};

static inline size_t wino_wpack_elems(int Cx16, int C) {   // in halves
  return (size_t)(C / kWnCh) * 2 * (size_t)(Cx16 / 16 + C / 16) * kWnStageVec * 8;
}

// The pack, from the CURRENT device weights: one thread per (cb16, stage, comp, dx, row
// block, lane, element), both planes.  Element e of lane l: A-operand row l & 31 = gate
// (row >> 3), channel cb16*16 + rb*8 + (row & 7); k = 8 (l >> 5) + e = input channel of the
// chunk.  Stage s = 2 * chunk + (comp >> 1); chunks = x groups of 16 first, then h groups.
__global__ void pack_wino_kernel(const float* __restrict__ w, _Float16* __restrict__ out,
                                 int Cx_total, int Cx16, int C, size_t total) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int e = idx & 7;
  const int l = (idx >> 3) & 63;
  const int rb = (idx >> 9) & 1;
  size_t t = idx >> 10;                           // ((cb16 * nst + s) * 2 + ci) * 3 + dx
  const int dx = (int)(t % 3); t /= 3;
  const int ci = (int)(t & 1); t >>= 1;
  const int nxc = Cx16 / 16, nst = 2 * (nxc + C / 16);
  const int s = (int)(t % nst), cb16 = (int)(t / nst);
  const int chunk = s >> 1, comp = (s & 1) * 2 + ci;
  const bool is_x = chunk < nxc;
  const int cg = is_x ? chunk : chunk - nxc;
  const int k = 8 * (l >> 5) + e;
  const int cin = (is_x ? 0 : Cx_total) + cg * 16 + k;
  const int row = l & 31;
  const int n = (row >> 3) * C + cb16 * kWnCh + rb * 8 + (row & 7);
  const int Cin = Cx_total + C, N4 = 4 * C;
  const double g0 = w[((size_t)(0 * 3 + dx) * Cin + cin) * N4 + n];
  const double g1 = w[((size_t)(1 * 3 + dx) * Cin + cin) * N4 + n];
  const double g2 = w[((size_t)(2 * 3 + dx) * Cin + cin) * N4 + n];
  const double u = comp == 0 ? g0 : (comp == 1 ? 0.5 * (g0 + g1 + g2)
                                               : (comp == 2 ? 0.5 * (g0 - g1 + g2) : g2));
  const double sv = u * 256.0;
  note_pack_range(sv);
  const _Float16 v0 = (_Float16)sv;
  const _Float16 v1 = (_Float16)(sv - (double)v0);
  const size_t base = ((((size_t)cb16 * nst + s) * 2 + ci) * 3 + dx) * (2 * 2 * 64 * 8);
  out[base + ((size_t)(0 * 2 + rb) * 64 + l) * 8 + e] = v0;
  out[base + ((size_t)(1 * 2 + rb) * 64 + l) * 8 + e] = v1;
}

// a - b on packed halves as ONE v_pk_fma_f16 (b * -1 + a, exactly rounded like the
// subtraction).  Written as a - b, hipcc scalarises a <8 x half> subtraction into v_sub_f16 /
// SDWA / v_pack triples (there is no v_pk_sub_f16 and the fsub lowering does not use the neg
// modifiers); m1 = (-1, -1) comes from a laundered SGPR so that the fma is not folded back.
__device__ __forceinline__ f16x8 wn_sub(const f16x8 a, const f16x8 b, const f16x8 m1) {
  return __builtin_elementwise_fma(b, m1, a);
}
// (a_hi + a_lo) +- (b_hi + b_lo) as a plane pair: error-free TwoSum of the high planes
// (s + err == a_hi +- b_hi exactly), everything else into the low plane.  8 packed
// instructions per register.
template <bool SUB>
__device__ __forceinline__ void wn_combine(const f16x8 a_hi, const f16x8 a_lo, const f16x8 b_hi,
                                           const f16x8 b_lo, const f16x8 m1, f16x8& hi,
                                           f16x8& lo) {
  const f16x8 s = SUB ? wn_sub(a_hi, b_hi, m1) : a_hi + b_hi;
  const f16x8 bb = wn_sub(s, a_hi, m1);              // the part of +-b_hi that arrived in s
  const f16x8 t = wn_sub(s, bb, m1);                 // the part of a_hi that arrived
  const f16x8 e1 = wn_sub(a_hi, t, m1);
  const f16x8 e2 = SUB ? wn_sub(-bb, b_hi, m1) : wn_sub(b_hi, bb, m1);   // (+-b_hi) - bb
  hi = s;
  lo = (e1 + e2) + (SUB ? wn_sub(a_lo, b_lo, m1) : a_lo + b_lo);
}

// the fragment moved one lane up (dx = -1: lane l takes lane l - 1) or down the wave
__device__ __forceinline__ f16x8 wn_lane_shift(const f16x8& v, bool up, bool ok) {
  const u32x4 w = __builtin_bit_cast(u32x4, v);
  u32x4 r;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const uint32_t t =
        up ? (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[j], 0x138, 0xf, 0xf, true)
           : (uint32_t)__builtin_amdgcn_update_dpp(0, (int)w[j], 0x130, 0xf, 0xf, true);
    r[j] = ok ? t : 0u;
  }
  return __builtin_bit_cast(f16x8, r);
}

// EPI = kEpiLstm: the forward step (LSTM epilogue).  EPI = kEpiStore: dgrad of the gate
// convolution -- the operand is the gate gradient G [M][4C] as planes under its own exponent
// (g_exp), the A rows are 64 OUTPUT COLUMNS (d h channels, then d x channels) instead of 4 gates
// x 16 channels, the chunk range is one of n_kslice split-K slices, and the epilogue stores
// y * 2^-(8 + e) to out0 / out1 (or their per-slice partial tiles).
template <int WAVES, int EPI = kEpiLstm>
__device__ __forceinline__ void convlstm_wino_body(const ConvLstmWinoArgs& p, int cb16, int mt,
                                                   f16x8* lds /* [2][kWnStageVec] */,
                                                   int kslice = 0, int n_kslice = 1) {
  constexpr int kWnPairs = WAVES * 32;             // pair-cells per workgroup
  const ConvLstm16Args& q = p.b;
  const ConvLstmArgs& a = q.f;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int H = a.H, W = a.W, HW = H * W, C = a.C, Cx = a.Cx;
  const int Hp = (H + 1) >> 1, Kp = Hp * W;
  const int Q_total = a.rows * Kp;
  const int q_wave = mt * kWnPairs + wave * 32;
  const bool wave_live = q_wave < Q_total;       // dead waves still copy and hit barriers
  const int col = lane & 31, half = lane >> 5;

  int r = 0, y0 = 0, xpos = 0;
  bool valid;
  {
    const int qq = q_wave + col;
    valid = qq < Q_total;
    if (valid) {
      r = qq / Kp;
      const int pc = qq - r * Kp;
      const int t = pc / W;
      y0 = 2 * t;
      xpos = pc - t * W;
    }
  }
  const int srh = (valid && a.src_row_h) ? a.src_row_h[r] : r;
  const bool okx0 = valid & (xpos > 0), okx2 = valid & (xpos + 1 < W);

  // operand rows y0 - 1 .. y0 + 2 of the lane's column: byte offsets of the lane's 16-byte
  // vector in channel group 0 of the tiled planes (from the zero pad in front of a plane,
  // so that offset 0 reads zeros); an out-of-image row reads offset 0
  bool rok[4];
  uint32_t roffx[4], roffh[4];
  {
    const int KGx = Cx >> 4, KGh = C >> 4;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int rho = y0 - 1 + i;
      rok[i] = valid & (rho >= 0) & (rho < H);
      const int cx = r * HW + rho * W + xpos, chh = srh * HW + rho * W + xpos;
      roffx[i] = (uint32_t)((((cx >> 5) * KGx) * 512 + half * 256 + (cx & 31) * 8 + kPlanePad) * 2);
      roffh[i] = (uint32_t)((((chh >> 5) * KGh) * 512 + half * 256 + (chh & 31) * 8 + kPlanePad) * 2);
    }
  }

  f32x16 acc[4][2];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[c][rb][i] = 0.f;

  // ---- the wave's tile of the cell state c (64 cells x 16 channels, the epilogue's operand)
  // starts its way into LDS NOW, by LDS-DMA: no registers, and the HBM round trip that the
  // epilogue used to open with is hidden behind the whole main loop.  Tile layout and the
  // lane -> (cell row, 16-byte piece) map: see the epilogue.  c is not written by this launch
  // (c' goes to the other buffer of the ping-pong pair).
  float* const ctile = reinterpret_cast<float*>(lds + 2 * kWnStageVec) + wave * 1024;
  if (EPI == kEpiLstm && wave_live && !a.zero_state) {
    const uint32_t rowb0 = (uint32_t)C * 4u;
    const int src_c0 = (valid && a.src_row_c) ? a.src_row_c[r] : r;
    uint32_t coff0[2];
#pragma unroll
    for (int e = 0; e < 2; ++e)
      coff0[e] = (valid & (y0 + e < H))
                     ? (uint32_t)(src_c0 * HW + (y0 + e) * W + xpos) * rowb0 : 0xffffffffu;
    const __amdgpu_buffer_rsrc_t c_rs0 = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(const_cast<float*>(a.c)), 0, (uint32_t)(a.rows * HW) * rowb0, 0x00020000);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int srcl = ((lane >> 2) + 16 * (k & 1)) * 4;
      const uint32_t ro = (uint32_t)__builtin_amdgcn_ds_bpermute(srcl, (int)coff0[k >> 1]);
      // rows that own no cell read offset 0 (the value is never used)
      const uint32_t off = ro != 0xffffffffu
                               ? ro + (uint32_t)(cb16 * kWnCh) * 4u + (uint32_t)(lane & 3) * 16u : 0u;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(
          c_rs0, (__attribute__((address_space(3))) void*)(ctile + k * 256), 16, off, 0, 0,
          MV_EPI_LD_AUX);
    }
  }

  // ---- the 2-channel fp32 x chunk (regression encoder), direct form
  if (EPI == kEpiLstm && a.x_small && wave_live) {
    const int Cin = Cx + C, N4 = 4 * C;
    const int nk = 9 * Cx;
    const int n0 = (col >> 3) * C + cb16 * kWnCh + (col & 7);
    for (int k2 = 0; 2 * k2 < nk; ++k2) {
      const int k = 2 * k2 + half;
      const bool kok = k < nk;
      const int tap = kok ? k / Cx : 0, chn = kok ? k - tap * Cx : 0;
      const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
      float wv[2];
#pragma unroll
      for (int rb = 0; rb < 2; ++rb) {
        const float tw = p.w_hwio[((size_t)tap * Cin + chn) * N4 + n0 + rb * 8];
        wv[rb] = kok ? tw * 65536.0f : 0.f;
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        const int yy = y0 + e + dy, xx = xpos + dx;
        const bool ok = kok & valid & (yy >= 0) & (yy < H) & (xx >= 0) & (xx < W);
        const int off = ok ? r * a.x_row_stride + (yy * W + xx) * Cx + chn : 0;
        const float tv = a.x[off];
        const float v = ok ? (e ? -tv : tv) : 0.f;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
          acc[e ? 3 : 0][rb] =
              __builtin_amdgcn_mfma_f32_32x32x2f32(wv[rb], v, acc[e ? 3 : 0][rb], 0, 0, 0);
      }
    }
  }

  // ---- f16 chunks of 16 input channels: x chunks first, then h chunks
  const int nxc = p.n_xc;
  int ck_lo = a.sx_corr ? nxc : 0;                          // sparse x: table terms instead
  int ck_hi = a.zero_state ? nxc : nxc + (C >> 4);
  if (EPI == kEpiStore && n_kslice > 1) {                   // dgrad split-K: equal chunk ranges
    const int per = (C >> 4) / n_kslice;
    ck_lo = kslice * per;
    ck_hi = ck_lo + per;
  }
  if (ck_hi > ck_lo) {
    const f16x8* wblk = reinterpret_cast<const f16x8*>(p.wpw) +
                        (size_t)cb16 * 2 * (nxc + (C >> 4)) * kWnStageVec;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(const_cast<f16x8*>(wblk)), 0, 0x7fffffff, 0x00020000);
    const _Float16* const x16 = q.x16 ? q.x16 : q.h16;
    const int64_t xps = q.x16 ? q.x_plane_stride : 0;
    const _Float16* const h16 = q.h16 ? q.h16 : q.x16;
    const int64_t hps = q.h16 ? q.h_plane_stride : 0;
    const __amdgpu_buffer_rsrc_t xrs0 = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(const_cast<_Float16*>(x16 - kPlanePad)), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrs1 = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(const_cast<_Float16*>(x16 + xps - kPlanePad)), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t hrs0 = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(const_cast<_Float16*>(h16 - kPlanePad)), 0, 0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t hrs1 = __builtin_amdgcn_make_buffer_rsrc(
        uniform_ptr(const_cast<_Float16*>(h16 + hps - kPlanePad)), 0, 0x7fffffff, 0x00020000);

    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // stage copy: the pack IS the LDS image; 24 pieces of 64 vectors, 24 / WAVES per wave
    auto stage_dma = [&](int s, f16x8* dstbuf) {
#pragma unroll
      for (int i = 0; i < 24 / WAVES; ++i) {
        const int v0 = (i * WAVES + wave_u) * 64;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(
            wrs, (__attribute__((address_space(3))) void*)(dstbuf + v0), 16,
            (uint32_t)(v0 + lane) * 16u, (uint32_t)s * kWnStageBytes, 0, MV_DMA_AUX);
      }
    };
    f16x8 raw[4][2];
    auto load_raw = [&](int ck) {
      const bool is_x = ck < nxc;
      const uint32_t cgo = (uint32_t)(is_x ? ck : ck - nxc) * 1024u;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const uint32_t o = (is_x ? roffx[i] : roffh[i]) + cgo;
        const int off = rok[i] ? (int)o : 0;
        raw[i][0] = __builtin_bit_cast(
            f16x8, __builtin_amdgcn_raw_buffer_load_b128(is_x ? xrs0 : hrs0, off, 0, 0));
        raw[i][1] = __builtin_bit_cast(
            f16x8, __builtin_amdgcn_raw_buffer_load_b128(is_x ? xrs1 : hrs1, off, 0, 0));
      }
    };
    // one component: 3 dx x 2 row blocks x 3 MFMAs from stage buffer `buf`, slot ci
#define MV_WN_COMP(COMP, CI, VHI, VLO, BUF)                                                   \
  do {                                                                                        \
    _Pragma("unroll") for (int dx = 0; dx < 3; ++dx) {                                        \
      const bool sh = dx != 1;                                                                \
      const f16x8 b0 = !sh ? (VHI) : wn_lane_shift((VHI), dx == 0, dx == 0 ? okx0 : okx2);    \
      const f16x8 b1 = !sh ? (VLO) : wn_lane_shift((VLO), dx == 0, dx == 0 ? okx0 : okx2);    \
      f16x8 w0[2], w1[2];                                                                     \
      _Pragma("unroll") for (int rb = 0; rb < 2; ++rb) {                                      \
        w0[rb] = (BUF)[((((CI) * 3 + dx) * 2 + 0) * 2 + rb) * 64 + lane];                     \
        w1[rb] = (BUF)[((((CI) * 3 + dx) * 2 + 1) * 2 + rb) * 64 + lane];                     \
      }                                                                                       \
      _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                        \
        acc[COMP][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1[rb], b0, acc[COMP][rb], 0, 0, 0); \
      _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                        \
        acc[COMP][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[rb], b1, acc[COMP][rb], 0, 0, 0); \
      _Pragma("unroll") for (int rb = 0; rb < 2; ++rb)                                        \
        acc[COMP][rb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0[rb], b0, acc[COMP][rb], 0, 0, 0); \
    }                                                                                         \
  } while (0)

    uint32_t mone = 0xBC00BC00u;              // (-1, -1) as packed halves, opaque to hipcc
    asm volatile("" : "+s"(mone));
    const f16x8 m1 = __builtin_bit_cast(f16x8, u32x4{mone, mone, mone, mone});
    f16x8* const bufA = lds;
    f16x8* const bufB = lds + kWnStageVec;
    load_raw(ck_lo);
    stage_dma(2 * ck_lo, bufA);
    __syncthreads();                         // carries the vmcnt(0) of the pending LDS-DMA
    {
    for (int ck = ck_lo; ck < ck_hi; ++ck) {
      const bool more = ck + 1 < ck_hi;
      f16x8 vh0, vl0, vh1, vl1;
      // stage A: components 0, 1
      stage_dma(2 * ck + 1, bufB);           // its buffer was last read before the barrier
      wn_combine<true>(raw[0][0], raw[0][1], raw[2][0], raw[2][1], m1, vh0, vl0);   // d(-1) - d(+1)
      wn_combine<false>(raw[1][0], raw[1][1], raw[2][0], raw[2][1], m1, vh1, vl1);  // d(0) + d(+1)
      MV_WN_COMP(0, 0, vh0, vl0, bufA);
      MV_WN_COMP(1, 1, vh1, vl1, bufA);
      __syncthreads();
      // stage B: components 2, 3
      wn_combine<true>(raw[2][0], raw[2][1], raw[1][0], raw[1][1], m1, vh0, vl0);   // d(+1) - d(0)
      wn_combine<true>(raw[1][0], raw[1][1], raw[3][0], raw[3][1], m1, vh1, vl1);   // d(0) - d(+2)
      if (more) {
        stage_dma(2 * ck + 2, bufA);
        load_raw(ck + 1);                    // a whole stage (36 MFMAs) ahead of its use
      }
      MV_WN_COMP(2, 0, vh0, vl0, bufB);
      MV_WN_COMP(3, 1, vh1, vl1, bufB);
      if (q.x_exp && ck == nxc - 1) {        // x planes at 2^e (relu / lrelu): sums to 2^16
        const float f = __int_as_float((127 + 8 - q.x_exp[0]) << 23);
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
          for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[c][rb][i] *= f;
      }
      __syncthreads();
    }
    }
#undef MV_WN_COMP
  }
  if (!wave_live) return;

  if constexpr (EPI == kEpiStore) {
    // registers of acc[c][rb]: output column cb*64 + rb*32 + 8*(reg >> 2) + 4*half + (reg & 3)
    // of the lane's pair-cell, rows y0 (e = 0) and y0 + 1 (e = 1)
    const float scale = ldexpf(1.0f, -(8 + (q.g_exp ? q.g_exp[0] : 0)));
    const int M_total = a.rows * HW;
    const int half_s = lane >> 5;
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const bool okc = valid & (y0 + e < H);
      const size_t m = (size_t)r * HW + (size_t)(y0 + e) * W + xpos;
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int jg = 0; jg < 4; ++jg) {
          const int col = cb16 * 64 + rb * 32 + 8 * jg + 4 * half_s;
          f32x4 v;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int reg = jg * 4 + j;
            const float m0 = acc[0][rb][reg], m1 = acc[1][rb][reg], m2 = acc[2][rb][reg],
                        m3 = acc[3][rb][reg];
            v[j] = (e == 0 ? (m0 + m1) + m2 : (m1 - m2) - m3) * scale;
          }
          float* dst = nullptr;
          int stride = 0, cc = col;
          if (col < a.out0_cols) {
            stride = a.out0_cols;
            dst = n_kslice > 1 ? (a.out0 ? q.part0 + (size_t)kslice * M_total * stride : nullptr)
                               : a.out0;
          } else if (col - a.out0_cols < a.out1_cols) {
            stride = a.out1_cols; cc = col - a.out0_cols;
            dst = n_kslice > 1 ? (a.out1 ? q.part1 + (size_t)kslice * M_total * stride : nullptr)
                               : a.out1;
          }
          if (dst && okc) {
            float* o = dst + m * stride + cc;
            if (cc + 4 <= stride && (stride & 3) == 0) {
              *reinterpret_cast<f32x4*>(o) = v;
            } else {
#pragma unroll
              for (int j = 0; j < 4; ++j)
                if (cc + j < stride) o[j] = v[j];
            }
          }
        }
    }
    return;
  }
  // ---------------------------------------------------------------- epilogue
  // registers of acc[c][rb]: gate = reg >> 2, channel = cb16*16 + rb*8 + 4*half + (reg & 3);
  // the lane's pair-cell gives rows y0 (e = 0) and y0 + 1 (e = 1).
  int lane_e = lane;
  asm volatile("" : "+v"(lane_e));          // keep the address arithmetic below the loop
  const int half_e = lane_e >> 5;
  const int ch0 = cb16 * kWnCh + 4 * half_e;                 // + rb * 8
  const uint32_t rowb = (uint32_t)C * 4u;                    // bytes per cell
  const uint32_t out_bytes = (uint32_t)(a.rows * HW) * rowb;
  const __amdgpu_buffer_rsrc_t co_rs = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.c_out), 0, out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t ho_rs = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.h_out), 0, out_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t go_rs = __builtin_amdgcn_make_buffer_rsrc(
      uniform_ptr(a.gates_out ? a.gates_out : a.h_out), 0, a.gates_out ? 4u * out_bytes : 0u,
      0x00020000);
  bool okc[2];
  int cell[2];                                // cell index inside its image
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    okc[e] = valid & (y0 + e < H);
    cell[e] = (y0 + e) * W + xpos;
  }
  // ---- state I/O through LDS.  A lane owns 4 + 4 channels of the two cells of its pair-cell:
  // stored from the accumulator layout, every lane of a wave instruction would touch its own
  // cache line (16 bytes at a stride of C floats).  Instead the wave's tile of a state tensor
  // -- 64 cells x 16 channels fp32, rows = cells (e * 32 + column) -- passes through a
  // wave-private 4 KB LDS tile (the weight stages are dead after the last barrier) and is
  // moved to / from memory with lane l on cell row (l >> 2) + 16 k, 16-byte piece l & 3:
  // four lanes cover the 64 contiguous bytes the workgroup's 16 channels have in a cell.
  // Row r of the tile belongs to lane (r & 31)'s pair-cell: its offsets come by ds_bpermute.
  float* const tl0 = ctile;                   // c in, then c' out
  float* const tl1 = reinterpret_cast<float*>(lds) + wave * 1024;     // h' out (a dead weight stage)
  constexpr uint32_t kNone = 0xffffffffu;
  const uint32_t colb = (uint32_t)(cb16 * kWnCh) * 4u;       // the workgroup's first channel
  uint32_t ooff[2];                           // byte offsets of the lane's output cells
#pragma unroll
  for (int e = 0; e < 2; ++e)
    ooff[e] = okc[e] ? (uint32_t)(r * HW + cell[e]) * rowb : kNone;
  const int piece = lane_e & 3;
  uint32_t rooff[4];                          // output offsets of the four transposed rows of the lane
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int srcl = ((lane_e >> 2) + 16 * (k & 1)) * 4;
    rooff[k] = (uint32_t)__builtin_amdgcn_ds_bpermute(srcl, (int)ooff[k >> 1]);
  }
  const int wr_idx = (lane_e & 31) * 16 + half_e * 4;        // + e * 512 + rb * 8 (floats)
  const int rd_idx = (lane_e >> 2) * 16 + piece * 4;         // + k * 256
  // pass 1: every state load goes out before any store
  f32x4 cprev[2][2];
#pragma unroll
  for (int e = 0; e < 2; ++e)
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) cprev[e][rb] = f32x4{0.f, 0.f, 0.f, 0.f};
  if (!a.zero_state) {
    // the tile was requested before the main loop; its barriers carried the vmcnt(0) -- the
    // explicit wait covers a launch without f16 chunks
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int e = 0; e < 2; ++e)
#pragma unroll
      for (int rb = 0; rb < 2; ++rb)
        cprev[e][rb] = *reinterpret_cast<const f32x4*>(tl0 + wr_idx + e * 512 + rb * 8);
  }
  // sparse x: hot cell of the lane's image
  int hot_y = 0, hot_x = 0;
  if (a.sx_corr) {
    const int hr = a.sx_hot_div > 1 ? r / a.sx_hot_div : r;
    const uint32_t hyx = a.sx_cellyx[a.sx_hot[(size_t)hr * a.sx_hot_stride]];
    hot_y = (int)(hyx >> 16); hot_x = (int)(hyx & 0xffffu);
  }
  const float un = kF16Unscale;
  const bool planes = q.h16_out != nullptr;
  u32x2 ph[2][2], pl[2][2];                   // h' as plane halves (hi, lo), [e][rb]
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int y = y0 + e;
    const uint32_t mcell = (uint32_t)(r * HW + cell[e]);     // output cell, flat
#pragma unroll
    for (int rb = 0; rb < 2; ++rb) {
      const int ch = ch0 + rb * 8;
      // per-gate additive terms (4 channels each)
      f32x4 add[4];
      {
        const float* bsrc = a.bias;
        if (a.sx_corr && a.sx_bias) {
          const int cls = 3 * (y == 0 ? 0 : (y == H - 1 ? 2 : 1)) +
                          (xpos == 0 ? 0 : (xpos == W - 1 ? 2 : 1));
          bsrc = a.sx_bias + (size_t)(okc[e] ? cls : 4) * 4 * C;
        }
#pragma unroll
        for (int g = 0; g < 4; ++g)
          add[g] = *reinterpret_cast<const f32x4*>(bsrc + g * C + ch);
        if (a.sx_corr) {
          // UNCONDITIONAL loads (a branch around them would serialise one L2 round trip per
          // row and row block behind the bias row's): a cell outside the radius reads the
          // clamped table row and drops it
          const int dy = y - hot_y, dxh = xpos - hot_x, rad = a.sx_rad;
          const bool inr = okc[e] && dy >= -rad && dy <= rad && dxh >= -rad && dxh <= rad;
          const int dyc = dy < -rad ? -rad : (dy > rad ? rad : dy);
          const int dxc = dxh < -rad ? -rad : (dxh > rad ? rad : dxh);
          const int side = 2 * rad + 1;
          const int idx = a.sx_by_class
                              ? 3 * (hot_y == 0 ? 0 : (hot_y == H - 1 ? 2 : 1)) +
                                    (hot_x == 0 ? 0 : (hot_x == W - 1 ? 2 : 1))
                              : (valid ? r : 0);
          const float* ct = a.sx_corr +
              ((size_t)idx * side * side + (dyc + rad) * side + (dxc + rad)) * 4 * C + ch;
          const float keep = inr ? 1.0f : 0.0f;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 cv = *reinterpret_cast<const f32x4*>(ct + g * C);
            add[g][0] = __builtin_fmaf(cv[0], keep, add[g][0]);
            add[g][1] = __builtin_fmaf(cv[1], keep, add[g][1]);
            add[g][2] = __builtin_fmaf(cv[2], keep, add[g][2]);
            add[g][3] = __builtin_fmaf(cv[3], keep, add[g][3]);
          }
        }
      }
      f32x4 cn4, hn4, si4, tj4, sf4, so4;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float pre[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int reg = g * 4 + j;
          const float m0 = acc[0][rb][reg], m1 = acc[1][rb][reg], m2 = acc[2][rb][reg],
                      m3 = acc[3][rb][reg];
          const float yv = e == 0 ? (m0 + m1) + m2 : (m1 - m2) - m3;
          pre[g] = __builtin_fmaf(yv, un, add[g][j]);   // un = 2^-16: the product is exact
        }
        const float si = sigm_(pre[0]), tj = tanh_(pre[1]), sf = sigm_(pre[2] + a.forget_bias),
                    so = sigm_(pre[3]);
        float cn = sf * cprev[e][rb][j];
        cn = cn + si * tj;
        const float hn = tanh_(cn) * so;
        cn4[j] = cn; hn4[j] = hn; si4[j] = si; tj4[j] = tj; sf4[j] = sf; so4[j] = so;
      }
      // c' and h' into the wave's two LDS tiles (the c tile was read into cprev above)
      *reinterpret_cast<f32x4*>(tl0 + wr_idx + e * 512 + rb * 8) = cn4;
      if (!a.skip_h32) *reinterpret_cast<f32x4*>(tl1 + wr_idx + e * 512 + rb * 8) = hn4;
      if (a.gates_out && okc[e]) {
        // training forward: the four gate activations [m][4][C] (stored from the lane)
        const uint32_t g0 = (mcell * 4u * (uint32_t)C + (uint32_t)ch) * 4u;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, si4), go_rs, (int)g0,
                                               0, MV_EPI_ST_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, tj4), go_rs,
                                               (int)(g0 + rowb), 0, MV_EPI_ST_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, sf4), go_rs,
                                               (int)(g0 + 2 * rowb), 0, MV_EPI_ST_AUX);
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, so4), go_rs,
                                               (int)(g0 + 3 * rowb), 0, MV_EPI_ST_AUX);
      }
      if (planes) {
        f16x4 p0, p1;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float sc = hn4[j] * kF16Scale;
          const _Float16 h0 = (_Float16)sc;
          p0[j] = h0;
          p1[j] = (_Float16)(sc - (float)h0);
        }
        ph[e][rb] = __builtin_bit_cast(u32x2, p0);
        pl[e][rb] = __builtin_bit_cast(u32x2, p1);
      }
    }
  }
  // ---- c' / h' out: four lanes per cell, 64 contiguous bytes each
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32x4 cv = *reinterpret_cast<const u32x4*>(tl0 + rd_idx + k * 256);
      u32x4 hv = cv;
      if (!a.skip_h32) hv = *reinterpret_cast<const u32x4*>(tl1 + rd_idx + k * 256);
      if (rooff[k] != kNone) {
        const int off = (int)(rooff[k] + colb + (uint32_t)piece * 16u);
        __builtin_amdgcn_raw_buffer_store_b128(cv, co_rs, off, 0, MV_EPI_ST_AUX);
        if (!a.skip_h32) __builtin_amdgcn_raw_buffer_store_b128(hv, ho_rs, off, 0, MV_EPI_ST_AUX);
      }
    }
  }
  // ---- operand planes of h' for the next gate convolution: tile (m >> 5, cb16), k half rb,
  // 8 channels = one 16-byte vector per cell.  A lane holds 4 of them for each of its two
  // cells; v_permlane32_swap hands the lower half-wave the complete vectors of the e = 0 cells
  // and the upper half-wave those of the e = 1 cells: every store is 16 bytes per lane, the 32
  // lanes of a half-wave one contiguous 512-byte run per plane.
  if (planes) {
    const uint32_t mc = (uint32_t)(r * HW + (half_e ? cell[1] : cell[0]));
    const bool okp = half_e ? okc[1] : okc[0];
    const size_t o0 = ((size_t)(mc >> 5) * (size_t)(C >> 4) + (size_t)cb16) * 512 +
                      (size_t)((int)(mc & 31u) * 8);
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
      for (int pn = 0; pn < 2; ++pn) {
        const u32x2 A = pn ? pl[0][rb] : ph[0][rb], B = pn ? pl[1][rb] : ph[1][rb];
        const u32x2 s0 = __builtin_amdgcn_permlane32_swap(A[0], B[0], false, false);
        const u32x2 s1 = __builtin_amdgcn_permlane32_swap(A[1], B[1], false, false);
        // lower lanes: (own A | partner's A) = e 0, channels 0-3 | 4-7; upper lanes:
        // (partner's B | own B) = e 1, channels 0-3 | 4-7
        const u32x4 v = {s0[0], s1[0], s0[1], s1[1]};
        if (okp)
          *reinterpret_cast<u32x4*>(q.h16_out + (size_t)pn * q.h16_out_stride + o0 + rb * 256) = v;
      }
  }
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2)
void convlstm_step_wino_kernel(const ConvLstmWinoGroup g) {
  // two weight stage buffers (the epilogue reuses them as one 4 KB h' tile per wave) | one
  // 4 KB cell-state tile per wave (c in by LDS-DMA from the kernel's start, c' out)
  extern __shared__ __attribute__((aligned(16))) f16x8 lds[];
  int block = blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int i = 0; i < kMaxGroup - 1; ++i)
    if (i + 1 < g.n && (int)blockIdx.x >= g.block_end[i]) pi = i + 1;
  if (pi > 0) block -= g.block_end[pi - 1];
  // block -> (16-channel column block, row tile).  Mode 0: cb16 = block % 16 (XCD = block % 8
  // holds blocks x and x + 8); mode 1: XCD x holds the ADJACENT blocks 2x, 2x + 1 -- the two
  // halves of every 128-byte line of the state tensors, fetched and written through one L2
  auto cbmap = [&](int ncb, int& cb16, int& mt) {
    if (g.map_mode == 1 && (ncb & 15) == 0) {
      const int grp = block / 16, w16 = block - grp * 16;      // 16 consecutive blocks
      cb16 = (grp % (ncb / 16)) * 16 + 2 * (w16 & 7) + (w16 >> 3);
      mt = grp / (ncb / 16);
    } else {
      cb16 = block % ncb; mt = block / ncb;
    }
  };
  int cb16, mt;
  switch (pi) {
    case 0: cbmap(g.p[0].b.f.C / kWnCh, cb16, mt); convlstm_wino_body<WAVES>(g.p[0], cb16, mt, lds); break;
    case 1: cbmap(g.p[1].b.f.C / kWnCh, cb16, mt); convlstm_wino_body<WAVES>(g.p[1], cb16, mt, lds); break;
    case 2: cbmap(g.p[2].b.f.C / kWnCh, cb16, mt); convlstm_wino_body<WAVES>(g.p[2], cb16, mt, lds); break;
    default: cbmap(g.p[3].b.f.C / kWnCh, cb16, mt); convlstm_wino_body<WAVES>(g.p[3], cb16, mt, lds); break;
  }
}

// Waves per workgroup: 4 (two workgroups per CU, twice the L2 -> LDS weight traffic of one
// 8-wave workgroup) measured +2 ... +4 % on the greedy workload in five same-box sessions of
// round 4, equal on beam-20 and training; the 8-wave build is gone.
static inline int wino_waves() { return 4; }

static inline size_t wino_lds_bytes(int waves) {
  return (size_t)2 * kWnStageBytes + (size_t)waves * 4096;
}
static inline unsigned convlstm_wino_blocks(const ConvLstmArgs& a, int waves) {
  const size_t Q = (size_t)a.rows * ((a.H + 1) / 2) * a.W;
  const size_t pairs = (size_t)waves * 32;
  return (unsigned)((Q + pairs - 1) / pairs) * (unsigned)(a.C / kWnCh);
}

// The Winograd form serves a group when every problem's W divides 32 (the DPP column shift)
// and C is a multiple of 16.  MV_WINO=0 keeps the direct kernel (A/B runs).
static inline bool wino_enabled() {
  static const bool off = getenv("MV_WINO") && atoi(getenv("MV_WINO")) == 0;
  return !off;
}
static inline bool wino_dgrad_enabled() {
  static const bool off = getenv("MV_WINO_DGRAD") && atoi(getenv("MV_WINO_DGRAD")) == 0;
  return !off;
}
static inline bool wino_geometry_ok(const ConvLstmArgs& a) {
  return a.W > 0 && 32 % a.W == 0 && a.C % kWnCh == 0 && (a.Cx % 16 == 0 || a.x_small) &&
         a.H >= 2;
}

// dynamic LDS above the 64 KB default: the attribute is set once per process (the engine calls
// this when it packs the weights, i.e. never inside a graph capture)
static inline void wino_init_attributes() {
  static const bool done = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(convlstm_step_wino_kernel<4>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)wino_lds_bytes(4));
    return true;
  }();
  (void)done;
}

static inline void launch_convlstm_wino_steps(const ConvLstmWinoArgs* probs, int n,
                                              hipStream_t stream) {
  ConvLstmWinoGroup g{};
  g.n = n;
  const int waves = wino_waves();
  // MV_WINO_MAP: block -> column block map (1, default: the two halves of a 128-byte state
  // line on one XCD)
  static const int map_mode = getenv("MV_WINO_MAP") ? atoi(getenv("MV_WINO_MAP")) : 1;
  g.map_mode = map_mode;
  unsigned total = 0;
  for (int i = 0; i < n; ++i) {
    g.p[i] = probs[i];
    total += convlstm_wino_blocks(probs[i].b.f, waves);
    g.block_end[i] = (int32_t)total;
  }
  for (int i = n; i < kMaxGroup; ++i) g.block_end[i] = (int32_t)total;
  wino_init_attributes();
  hipLaunchKernelGGL(convlstm_step_wino_kernel<4>, dim3(total), dim3(256), wino_lds_bytes(4),
                     stream, g);
}

// ------------------------------------------------------------------ dgrad in Winograd form
// d[h | x][m][col] = conv3x3(G, W')[m][col], W'[ky'][kx'][n][col] = W[2 - ky'][2 - kx'][ci(col)][n]
// (the transposed, tap-flipped kernel; col < C -> ci = Cx + col, else ci = col - C).  Pack of
// its F(2,3) row transform: [cb64][stage][comp in stage][dx][plane][row block][lane][8]; A row
// l & 31 of row block rb = output column cb64*64 + rb*32 + (l & 31); k = 8 (l >> 5) + e = gate
// column of the chunk.  Columns past C + Cx are zero.
__global__ void pack_wino_dgrad_kernel(const float* __restrict__ w, _Float16* __restrict__ out,
                                       int Cx, int C, size_t total) {
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= total) return;
  const int e = idx & 7;
  const int l = (idx >> 3) & 63;
  const int rb = (idx >> 9) & 1;
  size_t t = idx >> 10;                           // ((cb64 * nst + s) * 2 + ci) * 3 + dx
  const int dx = (int)(t % 3); t /= 3;
  const int ci2 = (int)(t & 1); t >>= 1;
  const int N4 = 4 * C, nst = 2 * (N4 / 16);
  const int s = (int)(t % nst), cb = (int)(t / nst);
  const int chunk = s >> 1, comp = (s & 1) * 2 + ci2;
  const int n = chunk * 16 + 8 * (l >> 5) + e;    // gate column = reduction index
  const int col = cb * 64 + rb * 32 + (l & 31);
  const int Cin = Cx + C;
  int ci = -1;
  if (col < C) ci = Cx + col;
  else if (col - C < Cx) ci = col - C;
  double u = 0.0;
  if (ci >= 0) {
    const double g0 = w[((size_t)((2 - 0) * 3 + (2 - dx)) * Cin + ci) * N4 + n];
    const double g1 = w[((size_t)((2 - 1) * 3 + (2 - dx)) * Cin + ci) * N4 + n];
    const double g2 = w[((size_t)((2 - 2) * 3 + (2 - dx)) * Cin + ci) * N4 + n];
    u = comp == 0 ? g0 : (comp == 1 ? 0.5 * (g0 + g1 + g2)
                                    : (comp == 2 ? 0.5 * (g0 - g1 + g2) : g2));
  }
  const double sv = u * 256.0;
  note_pack_range(sv);
  const _Float16 v0 = (_Float16)sv;
  const _Float16 v1 = (_Float16)(sv - (double)v0);
  const size_t base = ((((size_t)cb * nst + s) * 2 + ci2) * 3 + dx) * (2 * 2 * 64 * 8);
  out[base + ((size_t)(0 * 2 + rb) * 64 + l) * 8 + e] = v0;
  out[base + ((size_t)(1 * 2 + rb) * 64 + l) * 8 + e] = v1;
}
static inline int wino_dgrad_colblocks(int Cx, int C) { return (C + Cx + 63) / 64; }
static inline size_t wino_dgrad_wpack_elems(int Cx, int C) {   // in halves
  return (size_t)wino_dgrad_colblocks(Cx, C) * 2 * (size_t)(4 * C / 16) * kWnStageVec * 8;
}

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2)
void convlstm_dgrad_wino_kernel(const ConvLstmWinoGroup g) {
  extern __shared__ __attribute__((aligned(16))) f16x8 lds[];
  int block = blockIdx.x;
  int pi = 0;
#pragma unroll
  for (int i = 0; i < kMaxGroup - 1; ++i)
    if (i + 1 < g.n && (int)blockIdx.x >= g.block_end[i]) pi = i + 1;
  if (pi > 0) block -= g.block_end[pi - 1];
  // block -> (column block of 64, k slice, row tile).  The d h column blocks first, their
  // (block, slice) combos numbering EIGHT (C = 256: 4 blocks x 2 slices), combo = block % 8 =
  // XCD: an L2 then keeps exactly one combo's weight planes (1.5 MB) while G streams through;
  // the d x blocks follow as a region of their own with eight slices (384 KB per combo).  (One
  // region of 5 blocks x 4 slices put five combos = 3.8 MB on every XCD's 4 MB L2.)
  auto run = [&](const ConvLstmWinoArgs& p) {
    const ConvLstmArgs& a = p.b.f;
    const int ncb = a.n_colblocks;                   // host: wino_dgrad_colblocks (or C / 64)
    const int ncm = a.out0_cols / 64;                // d h column blocks
    const int nkm = p.nks_main > 1 ? p.nks_main : 1, nkx = p.nks_x > 1 ? p.nks_x : 1;
    const int Q = a.rows * ((a.H + 1) >> 1) * a.W;
    const int mtiles = (Q + WAVES * 32 - 1) / (WAVES * 32);
    const int main_blocks = mtiles * ncm * nkm;
    if (block < main_blocks) {
      const int combo = block % (ncm * nkm);
      convlstm_wino_body<WAVES, kEpiStore>(p, combo % ncm, block / (ncm * nkm), lds, combo / ncm,
                                           nkm);
    } else {
      const int b2 = block - main_blocks, nxb = ncb - ncm;
      const int combo = b2 % (nxb * nkx);
      convlstm_wino_body<WAVES, kEpiStore>(p, ncm + combo % nxb, b2 / (nxb * nkx), lds,
                                           combo / nxb, nkx);
    }
  };
  switch (pi) {
    case 0: run(g.p[0]); break;
    case 1: run(g.p[1]); break;
    case 2: run(g.p[2]); break;
    default: run(g.p[3]); break;
  }
}

static inline void launch_convlstm_wino_dgrads(const ConvLstmWinoArgs* probs, int n,
                                               hipStream_t stream) {
  ConvLstmWinoGroup g{};
  g.n = n;
  const int waves = wino_waves();
  unsigned total = 0;
  for (int i = 0; i < n; ++i) {
    g.p[i] = probs[i];
    const ConvLstmArgs& a = probs[i].b.f;
    const size_t Q = (size_t)a.rows * ((a.H + 1) / 2) * a.W;
    const size_t pairs = (size_t)waves * 32;
    const unsigned mtiles = (unsigned)((Q + pairs - 1) / pairs);
    const unsigned ncm = (unsigned)(a.out0_cols / 64), nxb = (unsigned)a.n_colblocks - ncm;
    total += mtiles * (ncm * (unsigned)(probs[i].nks_main > 1 ? probs[i].nks_main : 1) +
                       nxb * (unsigned)(probs[i].nks_x > 1 ? probs[i].nks_x : 1));
    g.block_end[i] = (int32_t)total;
  }
  for (int i = n; i < kMaxGroup; ++i) g.block_end[i] = (int32_t)total;
  static const bool attr = [] {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(convlstm_dgrad_wino_kernel<4>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)wino_lds_bytes(4));
    return true;
  }();
  (void)attr;
  hipLaunchKernelGGL(convlstm_dgrad_wino_kernel<4>, dim3(total), dim3(256), wino_lds_bytes(4),
                     stream, g);
}

}  // namespace mv
