// Device helpers shared by the fp32 streaming BatchNorm kernels (bn_f32.hip) and the encode stage that works
// without its pre-BatchNorm tensor (encode_f32.hip): the dropout keep word of an 8-row x 4-column patch.
#pragma once
#include "common.h"
#include "philox.h"

namespace blh {

// keep word of rows base + 8w .. base + 8w + 7, columns col .. col + 3
__device__ __forceinline__ uint32_t f2_keep_word(const DropoutSrc& d, int64_t base, int w, int col, int W,
                                                 int64_t batch) {
  if (d.keep) {                      // explicit masks (parity tests): [B][W] bytes
    uint32_t word = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t r = base + 8 * w + j;
      if (r < batch) {
        const uchar4 k = *reinterpret_cast<const uchar4*>(d.keep + r * (int64_t)W + col);
        word |= ((k.x ? 1u : 0u) | (k.y ? 2u : 0u) | (k.z ? 4u : 0u) | (k.w ? 8u : 0u)) << (4 * j);
      }
    }
    return word;
  }
  const Philox128 p = dropout_patch(d.seed, dropout_step(d), d.layer, base + d.row_offset, col);
  return w == 0 ? p.w[0] : (w == 1 ? p.w[1] : (w == 2 ? p.w[2] : p.w[3]));
}

}  // namespace blh
