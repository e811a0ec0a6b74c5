// Host-side dispatch of the fp32 MFMA GEMM (kernel template: gemm_f32_ring.h).
#include <atomic>
#include <cstdlib>

#include "gemm_f32_ring.h"
#include "gemm_split_kernel.h"
#include "gemm_f16x2_kernel.h"

namespace blh {

// ------------------------------------------------------------------ host ----
// The dynamic-LDS attribute of a kernel is per device: one bit per device id and kernel, set
// with an atomic OR after the attribute call succeeded (two host threads racing here both call
// hipFuncSetAttribute, which is idempotent).
static int ensure_lds_attr(std::atomic<uint64_t>& done, const void* kern, size_t lds) {
  int dev = 0;
  BLH_HIP_TRY(hipGetDevice(&dev));
  const uint64_t bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    BLH_HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    done.fetch_or(bit, std::memory_order_release);
  }
  return BLH_OK;
}

// Ring geometry of the shipped fp32 kernel (gemm_f32_ring.h), chosen by tools/gemm_bench on
// MI355X (profiles/r02_gemm_variants.md).  The 128x128 tile exists in two forms:
//   * BK 64, 2 stages (128 KB LDS, one workgroup per CU): fewest barriers, the fastest form for a
//     launch that is alone on the chip with at most one tile per CU — the forward Linear at
//     B <= 4096 (67.5 us);
//   * BK 32, 2 stages (64 KB LDS, TWO workgroups per CU): the fill / drain of one tile (prologue
//     latency, epilogue, the 16.8 MB output burst) hides under the main loop of its co-resident
//     neighbour — of the same launch when the grid has more tiles than CUs (B = 16384: 252 vs
//     262 us), of the OTHER stream's launch in the two-stream backward (dgrad || wgrad pair 126.6
//     vs 132.1 us).
static constexpr int RING_BKT_SKINNY = 32, RING_STAGES_SKINNY = 3;

// The backward pair (data gradient on the main stream, weight gradient on the side stream).  When each
// of the two launches is at most one workgroup per CU (B x W <= 4 Mi at 128x128 tiles), both take the
// 128 KB form: they cannot share a CU, so the workgroup dispatcher itself serialises them — the
// weight-gradient workgroups are placed CU by CU as the data-gradient workgroups retire (no cross-queue
// latency, no 50/50 split of a CU's matrix pipes), each GEMM runs at its solo rate and the HBM-bound
// BatchNorm-backward chain of the next stage runs beside the weight gradient.  Step 1.035 against
// 1.057 ms at configs[1] (profiles/r03_backward_schedule.md).  BLH_F32_BWD_EXCL=0: the 64 KB form
// everywhere (A/B knob).
bool gemm_f32_backward_exclusive(int64_t wgs, int K, int k_per_launch_slice) {
  static const bool off = [] { const char* e = getenv("BLH_F32_BWD_EXCL"); return e && e[0] == '0'; }();
  return !off && wgs <= 256 && K > 32 && (k_per_launch_slice % 64) == 0;
}

template <int BM, int BN, int WM, int WN, int LA, int LB, int EPI, int BKT, int STAGES>
static int launch_ring(hipStream_t s, const GemmParams& p, int splits) {
  constexpr int NT = 64 * WM * WN;
  constexpr size_t lds = gemm_ring_lds_bytes<BM, BN, BKT, STAGES>();
  static std::atomic<uint64_t> attr_done{0};
  if (splits > 1 && (p.k_per_split % BKT) != 0) return BLH_ERR_SHAPE;
  auto kern = gemm_f32_ring_kernel<BM, BN, WM, WN, LA, LB, EPI, BKT, STAGES>;
  BLH_TRY(ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), lds));
  const int tiles = (int)(ceil_div(p.M, BM) * ceil_div(p.N, BN));
  dim3 grid(tiles, 1, splits);
  launch_kernel(kern, grid, dim3(NT), lds, s, p);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

template <int BM, int BN, int WM, int WN, int LA, int LB, int EPI, int PIPE>
static int launch_cfg(hipStream_t s, const GemmParams& p, int splits) {
  if constexpr (BM == 128 && BN == 128) {
    const int64_t wgs = ceil_div(p.M, BM) * ceil_div(p.N, BN) * splits;
    const bool forward = (LA == ROWK && LB == ROWK);
    // (K <= 32, the encode Linear: a 64-deep K tile would be half zero page)
    if (forward && wgs <= 256 && p.K > 32) return launch_ring<BM, BN, WM, WN, LA, LB, EPI, 64, 2>(s, p, splits);
    // backward GEMMs of a launch that covers the chip once: also the one-workgroup-per-CU form, see
    // gemm_f32_backward_exclusive()
    if (!forward && gemm_f32_backward_exclusive(wgs, p.K, splits > 1 ? p.k_per_split : p.K))
      return launch_ring<BM, BN, WM, WN, LA, LB, EPI, 64, 2>(s, p, splits);
    return launch_ring<BM, BN, WM, WN, LA, LB, EPI, 32, 2>(s, p, splits);
  } else {
    return launch_ring<BM, BN, WM, WN, LA, LB, EPI, RING_BKT_SKINNY, RING_STAGES_SKINNY>(s, p, splits);
  }
}

// The library ships the LDS-DMA ring kernel (gemm_f32_ring.h); the round-1 kernels of
// gemm_f32_kernel.h (register-staged PIPE 1, per-lane-pointer ring PIPE 3) are built by
// tools/gemm_bench only, as the A/B baseline.
#define BLH_CASE(BM_, BN_, WM_, WN_, LA_, LB_, EPI_) \
  if (la == LA_ && lb == LB_ && epi == EPI_)         \
    return launch_cfg<BM_, BN_, WM_, WN_, LA_, LB_, EPI_, 3>(s, p, splits);

// Only the (tile, layout, epilogue) combinations the network and the unit tests use are
// instantiated (each is a separate gfx950 kernel).
static int launch_128x128(hipStream_t s, int la, int lb, int epi, const GemmParams& p, int splits) {
  BLH_CASE(128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS)   // hidden / encode forward (train)
  BLH_CASE(128, 128, 4, 2, ROWK, ROWK, EPI_BIAS)         // forward (eval, small batch)
  BLH_CASE(128, 128, 4, 2, ROWK, ROWK, EPI_BN_RELU)      // forward (eval): Linear + BN + ReLU (+ skip)
  BLH_CASE(128, 128, 4, 2, ROWK, ROWK, EPI_STORE)
  BLH_CASE(128, 128, 4, 2, ROWK, KROW, EPI_STORE)        // dgrad
  BLH_CASE(128, 128, 4, 2, ROWK, KROW, EPI_ADD)          // dgrad + residual gradient
  BLH_CASE(128, 128, 4, 2, KROW, KROW, EPI_STORE)        // wgrad (split over the batch)
  BLH_CASE(128, 128, 4, 2, KROW, KROW, EPI_STORE_SQ)     // batched small-batch wgrad + its norm partials
  BLH_CASE(128, 128, 4, 2, KROW, ROWK, EPI_STORE)
  return BLH_ERR_INVALID_ARGUMENT;
}
static int launch_128x64(hipStream_t s, int la, int lb, int epi, const GemmParams& p, int splits) {
  BLH_CASE(128, 64, 2, 2, ROWK, ROWK, EPI_BIAS)          // decode forward
  BLH_CASE(128, 64, 2, 2, ROWK, ROWK, EPI_STORE)
  BLH_CASE(128, 64, 2, 2, ROWK, KROW, EPI_STORE)
  BLH_CASE(128, 64, 2, 2, KROW, KROW, EPI_STORE)
  return BLH_ERR_INVALID_ARGUMENT;
}
static int launch_64x128(hipStream_t s, int la, int lb, int epi, const GemmParams& p, int splits) {
  BLH_CASE(64, 128, 2, 2, ROWK, ROWK, EPI_BIAS_STATS)    // encode forward (K = 32): 2 workgroups per CU
  BLH_CASE(64, 128, 2, 2, ROWK, ROWK, EPI_BN_RELU)       // eval forward of a half-chip batch (mid_tile64)
  BLH_CASE(64, 128, 2, 2, KROW, KROW, EPI_STORE)         // decode wgrad (M = 48)
  BLH_CASE(64, 128, 2, 2, ROWK, ROWK, EPI_STORE)
  BLH_CASE(64, 128, 2, 2, ROWK, ROWK, EPI_BIAS)
  BLH_CASE(64, 128, 2, 2, ROWK, KROW, EPI_STORE)         // data gradient of a half-chip batch (step_f32.hip: mid64)
  BLH_CASE(64, 128, 2, 2, ROWK, KROW, EPI_ADD)
  return BLH_ERR_INVALID_ARGUMENT;
}
static int launch_128x32(hipStream_t s, int la, int lb, int epi, const GemmParams& p, int splits) {
  BLH_CASE(128, 32, 4, 1, KROW, KROW, EPI_STORE)         // encode wgrad (N = 32)
  BLH_CASE(128, 32, 4, 1, ROWK, ROWK, EPI_STORE)
  BLH_CASE(128, 32, 4, 1, ROWK, ROWK, EPI_BIAS)
  BLH_CASE(128, 32, 4, 1, ROWK, KROW, EPI_STORE)
  return BLH_ERR_INVALID_ARGUMENT;
}
#undef BLH_CASE

// ---- bf16x3 split instantiations (gemm_dtype = 2: fp32 accuracy on the bf16 matrix cores) ---
template <int LA, int LB, int EPI>
static int launch_cfg_split(hipStream_t s, const GemmParams& p, int splits) {
  constexpr size_t lds = gemm_split_lds_bytes<128, 128>();
  static std::atomic<uint64_t> attr_done{0};
  auto kern = gemm_split_kernel<128, 128, 2, 2, LA, LB, EPI>;
  BLH_TRY(ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), lds));
  const int tiles = (int)(ceil_div(p.M, 128) * ceil_div(p.N, 128));
  launch_kernel(kern, dim3(tiles, 1, splits), dim3(256), lds, s, p);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

#define BLH_CASE3(LA_, LB_, EPI_) \
  if (la == LA_ && lb == LB_ && epi == EPI_) return launch_cfg_split<LA_, LB_, EPI_>(s, p, splits);

// Only the 128x128 tile exists in split form (the 1024-wide Linears, 98 % of the FLOPs); the
// skinny encode / decode contractions of this mode run on the exact-fp32 MFMA kernels.
static int launch_split_128x128(hipStream_t s, int la, int lb, int epi, const GemmParams& p,
                                int splits) {
  BLH_CASE3(ROWK, ROWK, EPI_BIAS_STATS)
  BLH_CASE3(ROWK, ROWK, EPI_BIAS)
  BLH_CASE3(ROWK, ROWK, EPI_BN_RELU)
  BLH_CASE3(ROWK, ROWK, EPI_STORE)
  BLH_CASE3(ROWK, KROW, EPI_STORE)
  BLH_CASE3(ROWK, KROW, EPI_ADD)
  BLH_CASE3(KROW, KROW, EPI_STORE)
  return BLH_ERR_INVALID_ARGUMENT;
}
#undef BLH_CASE3

// ---- fp16x2 split instantiations (gemm_dtype = 3) ------------------------------------------
template <int LA, int LB, int EPI>
static int launch_cfg_f16x2(hipStream_t s, const GemmParams& p, int splits) {
  constexpr size_t lds = gemm_f16x2_lds_bytes<128, 128>();
  static std::atomic<uint64_t> attr_done{0};
  auto kern = gemm_f16x2_kernel<LA, LB, EPI>;
  BLH_TRY(ensure_lds_attr(attr_done, reinterpret_cast<const void*>(kern), lds));
  const int tiles = (int)(ceil_div(p.M, 128) * ceil_div(p.N, 128));
  launch_kernel(kern, dim3(tiles, 1, splits), dim3(256), lds, s, p);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

#define BLH_CASE4(LA_, LB_, EPI_) \
  if (la == LA_ && lb == LB_ && epi == EPI_) return launch_cfg_f16x2<LA_, LB_, EPI_>(s, p, splits);

static int launch_f16x2_128x128(hipStream_t s, int la, int lb, int epi, const GemmParams& p,
                                int splits) {
  BLH_CASE4(ROWK, ROWK, EPI_BIAS_STATS)
  BLH_CASE4(ROWK, ROWK, EPI_BIAS)
  BLH_CASE4(ROWK, ROWK, EPI_STORE)
  BLH_CASE4(ROWK, KROW, EPI_STORE)
  BLH_CASE4(ROWK, KROW, EPI_ADD)
  BLH_CASE4(KROW, KROW, EPI_STORE)
  return BLH_ERR_INVALID_ARGUMENT;
}
#undef BLH_CASE4

int gemm_stat_tile_rows(GemmTile tile) { return tile == TILE_64x128 ? 64 : 128; }

int gemm_grid_blocks(GemmTile tile, int M, int N) {
  switch (tile) {
    case TILE_128x128: return (int)(ceil_div(M, 128) * ceil_div(N, 128));
    case TILE_128x64: return (int)(ceil_div(M, 128) * ceil_div(N, 64));
    case TILE_64x128: return (int)(ceil_div(M, 64) * ceil_div(N, 128));
    case TILE_128x32: return (int)(ceil_div(M, 128) * ceil_div(N, 32));
  }
  return 0;
}

int launch_gemm(hipStream_t s, GemmTile tile, int la, int lb, int epi, const GemmParams& p,
                int splits, int dtype) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || splits < 1) return BLH_ERR_INVALID_ARGUMENT;
  if ((la == ROWK || lb == ROWK) && p.K % 4 != 0) return BLH_ERR_SHAPE;
  if (la == KROW && p.M % 4 != 0) return BLH_ERR_SHAPE;
  if (lb == KROW && p.N % 4 != 0) return BLH_ERR_SHAPE;
  if (splits > 1 && (p.k_per_split % 32 != 0)) return BLH_ERR_SHAPE;   // whole 32-deep K tiles per slab
  if (dtype == 1) return BLH_ERR_INVALID_ARGUMENT;   // (round 1's mixed mode: removed, superseded by gemm_dtype 4)
  // the split kernels take reductions in whole K tiles of 32 (every slab); other shapes (the
  // K = 48 decode dgrad) run on the exact kernel
  const bool whole_tiles = (p.K % 32 == 0) && (p.k_per_split % 32 == 0) &&
                           (int64_t)std::max(p.M, p.K) * p.lda < (1ll << 28) &&
                           (int64_t)std::max(p.N, p.K) * p.ldb < (1ll << 28);   // operands < 1 GiB (32-bit offsets)
  if (dtype == 3 && whole_tiles && tile == TILE_128x128 && p.a_amax && p.b_amax && p.a_namax > 0 && p.b_namax > 0) {
    const int rc = launch_f16x2_128x128(s, la, lb, epi, p, splits);
    if (rc != BLH_ERR_INVALID_ARGUMENT) return rc;
  }
  // (fp16x2 without operand maxima — stand-alone stages, the encode / decode contractions — runs
  //  on the range-safe bf16 split)
  if (dtype >= 2 && whole_tiles && tile == TILE_128x128) {   // combinations not built in split form: exact fp32
    const int rc = launch_split_128x128(s, la, lb, epi, p, splits);
    if (rc != BLH_ERR_INVALID_ARGUMENT) return rc;
  }
  switch (tile) {
    case TILE_128x128: return launch_128x128(s, la, lb, epi, p, splits);
    case TILE_128x64: return launch_128x64(s, la, lb, epi, p, splits);
    case TILE_64x128: return launch_64x128(s, la, lb, epi, p, splits);
    case TILE_128x32: return launch_128x32(s, la, lb, epi, p, splits);
  }
  return BLH_ERR_INVALID_ARGUMENT;
}

}  // namespace blh
