// Layout of the fp16 operand planes of the f16x3 gate convolution (shared by the
// producers in kernels_misc.h and the consumers in convlstm_f16x3.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>

namespace mv {

// Operand-plane layout of the f16x3 gate convolution: element (cell m, channel c) of
// an [M][C] tensor (C % 16 == 0) lives in tile (m >> 5, c >> 4) of 32 cells x 16
// channels, stored [k half = (c >> 3) & 1][cell & 31][8 channels] -- MFMA
// A-fragment order: lane l of a wave reads the 16 bytes of cell (l & 31), k half
// (l >> 5), and 32 consecutive cells are one contiguous 512-byte run.
__host__ __device__ __forceinline__ size_t plane_index(long long m, int c, int C) {
  return ((size_t)(m >> 5) * (size_t)(C >> 4) + (size_t)(c >> 4)) * 512 +
         (size_t)(((c >> 3) & 1) * 256 + (int)(m & 31) * 8 + (c & 7));
}
constexpr size_t kPlaneSlack = 32 * 1024;   // halves: the last, partial 32-cell tile row


}  // namespace mv
