// Pure host arithmetic shared by the library and by the host-side sanitizer driver
// (tools/host_sanitize.cpp): no HIP headers, no device code.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>

namespace blh {

static inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t round_up(int64_t a, int64_t b) { return ceil_div(a, b) * b; }

static constexpr int EW_COLS_PER_BLOCK = 256;    // 64 lanes x float4, 4 waves share the rows
static constexpr int EW_THREADS_HOST = 256;
static constexpr int WAMAX_PARTS = 64;
static constexpr int SUMSQ_MAX_PARTS = 1024;
// sum-of-squares partials written by the kernels that PRODUCE the gradient ranges of the fp32 backward
// (slab sums, bias reduction, gamma / beta finalize): the fused step then needs no pass over the arena
static constexpr int SUMSQ_FOLD_PARTS = 4096;
// bf16 storage: the batched slab sum leaves (items + 1) x blocks partials (elementwise.hip: launch_sum_slabs_batched_sq)
static constexpr int SUMSQ_FOLD_PARTS_H = 4096;   // (clip_adam requests up to 4096 partials in one round trip)

// rows handled by one block of the streaming BatchNorm kernels: whole 32-row Philox patches,
// at most 128 row chunks (= partials of the column sums)
static inline int ew_row_chunk(int64_t batch) {
  return (int)(32 * std::max<int64_t>(1, ceil_div(batch, 32 * 128)));
}
static inline int ew_num_row_chunks(int64_t batch) { return (int)ceil_div(batch, ew_row_chunk(batch)); }
// bf16-storage kernels (512 columns per block): smaller row chunks, up to `max_chunks` of them,
// so that the grid still holds several blocks per CU at W = 1024
static inline int ew_h_max_chunks() {
  static const int v = [] {                     // BLH_EW_H_CHUNKS: developer tuning knob
    const char* e = std::getenv("BLH_EW_H_CHUNKS");
    const int n = e ? std::atoi(e) : 0;
    return (n >= 32 && n <= 4096) ? n : 256;      // measured at B = 16384, W = 1024: 128 -> 2.42, 256 -> 2.28, 512 -> 2.38 ms/step
  }();
  return v;
}
static inline int ew_row_chunk_h(int64_t batch) {
  return (int)(32 * std::max<int64_t>(1, ceil_div(batch, 32 * (int64_t)ew_h_max_chunks())));
}
static inline int ew_num_row_chunks_h(int64_t batch) {
  return (int)ceil_div(batch, ew_row_chunk_h(batch));
}
// one max-|value| partial per wave of the fp32 streaming kernels (gemm_dtype 3)
static inline int ew_num_amax_parts(int64_t batch, int W) {
  return (int)(ceil_div(W, EW_COLS_PER_BLOCK) * ew_num_row_chunks(batch) * (EW_THREADS_HOST / 64));
}

}  // namespace blh
