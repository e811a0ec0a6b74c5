// Counter-based dropout mask for the Linear->BN->ReLU->Dropout(0.5) stages
// (/root/reference/model/bilinear.py:12).  Philox4x32-10 (Salmon et al., SC'11)
// keyed by the 64-bit seed; the counter is (row/32, column/4, layer, step), so one
// call yields the 128 keep-bits of a 32-row x 4-column patch: a thread that walks 4
// columns down 32 rows pays one Philox call per 128 elements, and backward
// regenerates exactly the bits forward used (no mask tensor in HBM).  The global
// row index is used, so the mask does not depend on how the batch is sharded
// across GPUs.
#pragma once
#include <stdint.h>

namespace blh {

struct Philox128 {
  uint32_t w[4];
};

__device__ __forceinline__ Philox128 philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2,
                                                   uint32_t c3, uint32_t k0, uint32_t k1) {
  constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint32_t hi0 = __umulhi(M0, c0), lo0 = M0 * c0;
    const uint32_t hi1 = __umulhi(M1, c2), lo1 = M1 * c2;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += W0; k1 += W1;
  }
  return Philox128{{c0, c1, c2, c3}};
}

// keep-bits of the 32x4 patch that contains (global_row, col)
__device__ __forceinline__ Philox128 dropout_patch(uint64_t seed, uint64_t step, int layer,
                                                   int64_t global_row, int col) {
  const uint32_t c0 = (uint32_t)(global_row >> 5);
  const uint32_t c1 = (uint32_t)(col >> 2);
  const uint32_t c2 = (uint32_t)layer | ((uint32_t)((global_row >> 37) & 0xFFFF) << 16);
  const uint32_t c3 = (uint32_t)step;
  const uint32_t k0 = (uint32_t)seed ^ (uint32_t)(step >> 32);
  const uint32_t k1 = (uint32_t)(seed >> 32);
  return philox4x32_10(c0, c1, c2, c3, k0, k1);
}

// keep-bit of element (global_row, col) inside its patch: bit (row%32)*4 + col%4
// (select chain, not p.w[runtime]: a runtime-indexed register array is lowered to a very
//  slow indirect access — measured 14 us on a 6 us kernel)
__device__ __forceinline__ uint32_t patch_nibble(const Philox128& p, int row_in_patch) {
  const int wi = row_in_patch >> 3;
  const uint32_t word = wi == 0 ? p.w[0] : (wi == 1 ? p.w[1] : (wi == 2 ? p.w[2] : p.w[3]));
  return (word >> ((row_in_patch & 7) * 4)) & 0xFu;
}

}  // namespace blh
