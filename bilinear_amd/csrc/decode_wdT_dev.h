// The decode weight's bf16 image for the second phase of the bf16-storage one-pass decode (skinny.hip:
// decode_fwd_mse_h_kernel<RT, true>): for column group g (128 columns), MFMA j, k-step ks one 1 KiB block of
// 64 x 16 bytes, lane (q, n) -> Wd[32 ks + 8 q .. + 7][128 g + 8 n + j], output features >= OF zero.  One thread per
// 16 bytes; runs as extra blocks of the forward's cast launch (gemm_bf16s.hip: cast2_f32_bf16_kernel).
#pragma once
#include "common.h"

namespace blh {

static __device__ __forceinline__ void wdT_image_block(const float* __restrict__ Wd, uint16_t* __restrict__ WdT, int W,
                                                       int OF, int block) {
  const int idx = block * 256 + (int)threadIdx.x;          // ((g * 8 + j) * 2 + ks) * 64 + lane
  if (idx >= W * 8) return;
  const int lane = idx & 63, ks = (idx >> 6) & 1, j = (idx >> 7) & 7, g = idx >> 10;
  const int n = lane & 15, q = lane >> 4;
  const int col = 128 * g + 8 * n + j, o0 = 32 * ks + 8 * q;
  uint32_t w[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const float lo = (o0 + 2 * t < OF) ? Wd[(int64_t)(o0 + 2 * t) * W + col] : 0.f;
    const float hi = (o0 + 2 * t + 1 < OF) ? Wd[(int64_t)(o0 + 2 * t + 1) * W + col] : 0.f;
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
    const bf2 pr = {(__bf16)lo, (__bf16)hi};
    w[t] = *reinterpret_cast<const uint32_t*>(&pr);
  }
  *reinterpret_cast<uint4*>(WdT + (int64_t)idx * 8) = make_uint4(w[0], w[1], w[2], w[3]);
}

}  // namespace blh
