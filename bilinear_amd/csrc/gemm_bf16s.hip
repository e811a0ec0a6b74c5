// Host-side dispatch of the bf16-storage GEMM (kernel: gemm_bf16s_kernel.h) and the cast kernels
// between the fp32 master tensors and their bf16 images.
#include <atomic>

#include <cstdlib>

#include "gemm_bf16s_kernel.h"
#include "decode_wdT_dev.h"
#include "gemm_bf16s_256.h"
#include "gemm_bf16s_128x256.h"

namespace blh {

static int ensure_lds_attr_h(std::atomic<uint64_t>& done, const void* kern, size_t lds) {
  int dev = 0;
  BLH_HIP_TRY(hipGetDevice(&dev));
  const uint64_t bit = 1ull << (dev & 63);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    BLH_HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    done.fetch_or(bit, std::memory_order_release);
  }
  return BLH_OK;
}

// K tile / ring depth, measured with tools/bf16s_bench (profiles/r02_bf16s_gemm.md):
//   forward / dgrad (a ROWK operand): 64 elements x 2 stages = 64 KB of LDS, two workgroups
//   (8 waves) per CU: 664 / 695 TFLOP/s at B = 16384, W = 1024 (96 / 128 KB variants: 470-495);
//   wgrad (both operands through the transposing read): 128 x 2 stages, one workgroup per CU:
//   697 TFLOP/s (64 x 2: 628).
template <int LA, int LB, int EPI, bool OUT_BF16>
static int launch_h(hipStream_t s, const GemmParamsH& p, int splits) {
  constexpr int H_BKE = (LA == KROW && LB == KROW) ? 128 : 64, H_STAGES = 2;
  if (splits > 1 && (p.k_per_split % H_BKE) != 0) return BLH_ERR_SHAPE;
  constexpr size_t lds = gemm_bf16s_lds_bytes<H_BKE, H_STAGES>();
  static std::atomic<uint64_t> attr_done{0};
  auto kern = gemm_bf16s_kernel<LA, LB, EPI, OUT_BF16, H_BKE, H_STAGES>;
  BLH_TRY(ensure_lds_attr_h(attr_done, reinterpret_cast<const void*>(kern), lds));
  const int tiles = (int)(ceil_div(p.M, 128) * ceil_div(p.N, 128));
  launch_kernel(kern, dim3(tiles, 1, splits), dim3(256), lds, s, p);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

template <int LA, int LB, int EPI, bool OUT_BF16>
static int launch_h256(hipStream_t s, const GemmParamsH& p, int splits) {
  static std::atomic<uint64_t> attr_done{0};
  auto kern = gemm_bf16s_256_kernel<LA, LB, EPI, OUT_BF16>;
  BLH_TRY(ensure_lds_attr_h(attr_done, reinterpret_cast<const void*>(kern), H256_LDS_BYTES));
  const int tiles = (int)(ceil_div(p.M, 256) * (p.N / 256));
  launch_kernel(kern, dim3(tiles, 1, splits), dim3(512), H256_LDS_BYTES, s, p);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

template <int LA, int LB, int EPI, bool OUT_BF16>
static int launch_h128x256(hipStream_t s, const GemmParamsH& p, int splits) {
  static std::atomic<uint64_t> attr_done{0};
  auto kern = gemm_bf16s_128x256_kernel<LA, LB, EPI, OUT_BF16>;
  BLH_TRY(ensure_lds_attr_h(attr_done, reinterpret_cast<const void*>(kern), H128_LDS_BYTES));
  const int tiles = (int)(ceil_div(p.M, 128) * (p.N / 256));
  launch_kernel(kern, dim3(tiles, 1, splits), dim3(512), H128_LDS_BYTES, s, p);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

// Which kernel serves a contraction.  H_TILE_256 (gemm_bf16s_256.h: 256 x 256 tiles, one workgroup
// per CU, 8-phase schedule) when the launch has enough such tiles to fill the chip and its shape meets
// that kernel's restrictions; H_TILE_128x256 (gemm_bf16s_128x256.h: the same schedule on 128-row
// tiles and a three-deep ring) when only the half-height tiles fill it (M = 8192 at W = 1024:
// BASELINE configs[3] per GPU); else H_TILE_128 (gemm_bf16s_kernel.h).  BLH_BF16S_TILE = 128 | 256 |
// 384 forces one (the big tiles only where the shape allows them) for A/B measurements.
static std::atomic<int> g_forced_tile{-1};
static int forced_tile() {
  int v = g_forced_tile.load(std::memory_order_relaxed);
  if (v < 0) {
    const char* e = getenv("BLH_BF16S_TILE");
    v = e ? atoi(e) : 0;
    g_forced_tile.store(v, std::memory_order_relaxed);
  }
  return v;
}
void gemm_bf16s_force_tile(int tile) { g_forced_tile.store(tile < 0 ? -1 : tile, std::memory_order_relaxed); }

int gemm_bf16s_pick_tile(int la, int lb, bool out_bf16, const GemmParamsH& p, int splits) {
  const int vec = out_bf16 ? 8 : 4;
  const int slabs = p.batch_splits > 0 ? p.batch_splits : splits;     // per GEMM of a batched launch
  const int64_t extent = (slabs > 1) ? p.k_per_split : p.K;
  // what both big-tile kernels need: whole 256-column tiles, 16-byte C rows / addend rows / slabs
  const bool common_ok = (p.N % 256 == 0) && extent >= 128 &&
                         (slabs == 1 || (int64_t)slabs * p.k_per_split == p.K) &&
                         (p.batch_splits == 0 || splits % p.batch_splits == 0) &&
                         (p.ldc % vec == 0) && ((reinterpret_cast<uintptr_t>(p.C) & 15) == 0) &&
                         (!p.addend || ((p.ldadd % 8 == 0) && ((reinterpret_cast<uintptr_t>(p.addend) & 15) == 0))) &&
                         (!out_bf16 || slabs == 1 || (p.c_split_stride % 8 == 0)) &&
                         (out_bf16 || slabs == 1 || (p.c_split_stride % 4 == 0)) &&
                         (p.batch_splits == 0 || p.c_batch_stride % 8 == 0) &&
                         ((int64_t)256 * std::max(p.lda, p.ldb) * 2 < (1ll << 31));
  const bool ok256 = common_ok && (extent % 128 == 0) && (la == ROWK || p.M % 256 == 0);
  const bool ok128x256 = common_ok && (extent % 64 == 0) && p.batch_splits == 0 && (la == ROWK || p.M % 128 == 0);
  const int force = forced_tile();
  if (force == H_TILE_128) return H_TILE_128;
  if (force == H_TILE_256 && ok256) return H_TILE_256;
  if (force == H_TILE_128x256 && ok128x256) return H_TILE_128x256;
  if (ok256) {
    const int64_t wgs = ceil_div(p.M, 256) * (p.N / 256) * splits;
    // Weight gradients (both operands KROW, fp32 slabs): only with at most 4 slabs (api_layout.h: wgrad_plan_h)
    static const bool min_tiles_forced = getenv("BLH_WGRAD256_MIN_TILES") != nullptr;   // (developer knob, read once)
    const bool wgrad_many_slabs = la == KROW && slabs > 4 && p.batch_splits == 0 && !min_tiles_forced;
    if (!wgrad_many_slabs && wgs >= 224) return H_TILE_256;
    // forward / data gradient with 129 .. 223 tiles of 256 x 256 (batch 8448 .. 14080 at W = 1024): one round of
    // 256 x 256 tiles on part of the chip (34 us at K = 1024) still beats TWO rounds of 128 x 256 tiles (2 x 22 us):
    // batch 12288 1.349 -> 1.277 ms per step, 10240 1.279 -> 1.195 (tools_dev/batch_sweep_bf16s.py)
    if (la == ROWK && splits == 1 && wgs > 128) return H_TILE_256;
  }
  if (ok128x256 && la == ROWK) {
    // (forward / data gradient; the weight gradient keeps its measured plans)
    const int64_t wgs = ceil_div(p.M, 128) * (p.N / 256) * splits;
    // (from half a chip of them on: batch 4096 / 6144 at W = 1024 0.768 / 0.861 ms per step against 0.792 / 0.885 on
    //  128 x 128 tiles)
    if (wgs >= 128) return H_TILE_128x256;
  }
  // fewer workgroups: the 128 x 128 grid fills the 256 CUs better
  return H_TILE_128;
}

int gemm_bf16s_tile_rows(int tile) { return tile == H_TILE_256 ? 256 : 128; }
int gemm_bf16s_tile_cols(int tile) { return tile == H_TILE_128 ? 128 : 256; }

#define BLH_CASEH(LA_, LB_, EPI_, OUT_)                                          \
  if (la == LA_ && lb == LB_ && epi == EPI_ && out_bf16 == OUT_)                 \
    return tile == H_TILE_256      ? launch_h256<LA_, LB_, EPI_, OUT_>(s, p, splits)    \
           : tile == H_TILE_128x256 ? launch_h128x256<LA_, LB_, EPI_, OUT_>(s, p, splits) \
                                    : launch_h<LA_, LB_, EPI_, OUT_>(s, p, splits);

int launch_gemm_bf16s(hipStream_t s, int la, int lb, int epi, bool out_bf16, const GemmParamsH& p,
                      int splits) {
  if (p.M <= 0 || p.N <= 0 || p.K <= 0 || splits < 1) return BLH_ERR_INVALID_ARGUMENT;
  // 16-byte DMA chunks: every contiguous extent a multiple of 8 elements, rows 16-byte aligned
  if (p.K % 8 != 0 || p.lda % 8 != 0 || p.ldb % 8 != 0) return BLH_ERR_SHAPE;
  if (la == KROW && p.M % 8 != 0) return BLH_ERR_SHAPE;
  if (lb == KROW && p.N % 8 != 0) return BLH_ERR_SHAPE;
  if (out_bf16 && (p.N % 2 != 0 || p.ldc % 2 != 0)) return BLH_ERR_SHAPE;
  if (((uintptr_t)p.A | (uintptr_t)p.B) & 15) return BLH_ERR_INVALID_ARGUMENT;
  // 32-bit per-lane byte offsets inside one tile row panel
  if ((int64_t)128 * std::max(p.lda, p.ldb) * 2 >= (1ll << 31)) return BLH_ERR_SHAPE;
  const int tile = gemm_bf16s_pick_tile(la, lb, out_bf16, p, splits);
  if (p.batch_splits > 0 && tile != H_TILE_256) return BLH_ERR_SHAPE;   // (batched launches exist on the 256 x 256 kernel only)
  BLH_CASEH(ROWK, ROWK, EPI_BIAS_STATS, true)    // forward (train): Z bf16 + BatchNorm partials
  BLH_CASEH(ROWK, ROWK, EPI_BIAS, true)          // forward (eval)
  BLH_CASEH(ROWK, ROWK, EPI_BIAS, false)         // decode forward: fp32 prediction
  BLH_CASEH(ROWK, ROWK, EPI_STORE, true)
  BLH_CASEH(ROWK, ROWK, EPI_STORE, false)
  BLH_CASEH(ROWK, KROW, EPI_STORE, true)         // dgrad
  BLH_CASEH(ROWK, KROW, EPI_ADD, true)           // dgrad + block-skip gradient
  BLH_CASEH(ROWK, KROW, EPI_STORE, false)
  BLH_CASEH(KROW, KROW, EPI_STORE, false)        // wgrad: fp32 slabs
  if (epi == EPI_BN_BWD || epi == EPI_BN_BWD_ADD) {   // dgrad + the BatchNorm-backward reductions of the stage below
    if (tile == H_TILE_128 || la != ROWK || lb != KROW || !out_bf16 || splits != 1 || !p.bn_z || !p.bn_keep ||
        !p.bn_scale || !p.bn_shift || !p.stat_part || (p.ldz % 8) != 0 || (epi == EPI_BN_BWD_ADD && !p.addend))
      return BLH_ERR_SHAPE;
    if (tile == H_TILE_256)
      return epi == EPI_BN_BWD ? launch_h256<ROWK, KROW, EPI_BN_BWD, true>(s, p, 1)
                               : launch_h256<ROWK, KROW, EPI_BN_BWD_ADD, true>(s, p, 1);
    return epi == EPI_BN_BWD ? launch_h128x256<ROWK, KROW, EPI_BN_BWD, true>(s, p, 1)
                             : launch_h128x256<ROWK, KROW, EPI_BN_BWD_ADD, true>(s, p, 1);
  }
  return BLH_ERR_INVALID_ARGUMENT;
}
#undef BLH_CASEH

// ---- casts ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ src,
                                                            bf16_bits* __restrict__ dst, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float4 v = *reinterpret_cast<const float4*>(src + i * 4);
    uint2 o;
    o.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
    o.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
    *reinterpret_cast<uint2*>(dst + i * 4) = o;
  }
}

__global__ __launch_bounds__(256) void cast_bf16_f32_kernel(const bf16_bits* __restrict__ src,
                                                            float* __restrict__ dst, int64_t n) {
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const uint2 v = *reinterpret_cast<const uint2*>(src + i * 4);
    float4 o;
    o.x = __uint_as_float(v.x << 16); o.y = __uint_as_float(v.x & 0xffff0000u);
    o.z = __uint_as_float(v.y << 16); o.w = __uint_as_float(v.y & 0xffff0000u);
    *reinterpret_cast<float4*>(dst + i * 4) = o;
  }
}

int launch_cast_f32_bf16(hipStream_t s, const float* src, uint16_t* dst, int64_t n) {
  if (n % 4 != 0) return BLH_ERR_SHAPE;
  const int64_t blocks = std::min<int64_t>(ceil_div(n / 4, 256), 4096);
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, n);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

// two tensors in one launch (the parameter arena and the network input at the head of the bf16-storage forward:
// a launch of its own costs the 1 MB input cast 4.9 us): blocks [0, b0) take the first, the rest the second
// (blocks behind the first b01 = b0 + b1: the decode weight's phase-2 image, decode_wdT_dev.h — wd_src != nullptr)
__global__ __launch_bounds__(256) void cast2_f32_bf16_kernel(const float* __restrict__ src0, bf16_bits* __restrict__ dst0,
                                                             int64_t n0, int b0, const float* __restrict__ src1,
                                                             bf16_bits* __restrict__ dst1, int64_t n1, int b01,
                                                             const float* __restrict__ wd_src, bf16_bits* __restrict__ wdT,
                                                             int wd_W, int wd_OF) {
  if ((int)blockIdx.x >= b01) {
    wdT_image_block(wd_src, wdT, wd_W, wd_OF, (int)blockIdx.x - b01);
    return;
  }
  const bool first = (int)blockIdx.x < b0;
  const float* __restrict__ src = first ? src0 : src1;
  bf16_bits* __restrict__ dst = first ? dst0 : dst1;
  const int64_t n4 = (first ? n0 : n1) >> 2;
  const int64_t blk = first ? blockIdx.x : blockIdx.x - b0, nblk = first ? b0 : (int64_t)b01 - b0;
  for (int64_t i = blk * (int64_t)blockDim.x + threadIdx.x; i < n4; i += nblk * blockDim.x) {
    const float4 v = *reinterpret_cast<const float4*>(src + i * 4);
    uint2 o;
    o.x = (uint32_t)f32_to_bf16(v.x) | ((uint32_t)f32_to_bf16(v.y) << 16);
    o.y = (uint32_t)f32_to_bf16(v.z) | ((uint32_t)f32_to_bf16(v.w) << 16);
    *reinterpret_cast<uint2*>(dst + i * 4) = o;
  }
}

int launch_cast2_f32_bf16(hipStream_t s, const float* src0, uint16_t* dst0, int64_t n0, const float* src1,
                          uint16_t* dst1, int64_t n1, const float* wd_src, uint16_t* wdT, int wd_W, int wd_OF) {
  if (n0 % 4 != 0 || n1 % 4 != 0 || n0 <= 0 || n1 <= 0) return BLH_ERR_SHAPE;
  if (wd_src && (wd_W % 128 != 0 || wd_OF > 64 || !wdT)) return BLH_ERR_SHAPE;
  const int64_t b0 = std::min<int64_t>(ceil_div(n0 / 4, 256), 4096), b1 = std::min<int64_t>(ceil_div(n1 / 4, 256), 1024);
  const int64_t b2 = wd_src ? ceil_div((int64_t)wd_W * 8, 256) : 0;
  hipLaunchKernelGGL(cast2_f32_bf16_kernel, dim3((unsigned)(b0 + b1 + b2)), dim3(256), 0, s, src0, dst0, n0, (int)b0, src1,
                     dst1, n1, (int)(b0 + b1), wd_src, wdT, wd_W, wd_OF);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

// thread = 4 consecutive columns n of one row m of dW; products of two bf16 values are exact in fp32
__global__ __launch_bounds__(256) void wgrad_tail_h_kernel(const bf16_bits* __restrict__ dz, int64_t ld_dz,
                                                           const bf16_bits* __restrict__ act, int64_t ld_act, int rows,
                                                           int M, int N, float* __restrict__ out, int accumulate) {
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const int n4 = N >> 2;
  if (idx >= (int64_t)M * n4) return;
  const int m = (int)(idx / n4), n = (int)(idx % n4) * 4;
  float4 acc = accumulate ? *reinterpret_cast<const float4*>(out + (int64_t)m * N + n) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (int r = 0; r < rows; ++r) {
    const float d = bf16_to_f32(dz[r * ld_dz + m]);
    const uint2 a = *reinterpret_cast<const uint2*>(act + r * ld_act + n);
    acc.x = fmaf(d, __uint_as_float(a.x << 16), acc.x); acc.y = fmaf(d, __uint_as_float(a.x & 0xffff0000u), acc.y);
    acc.z = fmaf(d, __uint_as_float(a.y << 16), acc.z); acc.w = fmaf(d, __uint_as_float(a.y & 0xffff0000u), acc.w);
  }
  *reinterpret_cast<float4*>(out + (int64_t)m * N + n) = acc;
}

int launch_wgrad_tail_h(hipStream_t s, const uint16_t* dz, int64_t ld_dz, const uint16_t* act, int64_t ld_act, int rows,
                        int M, int N, float* out, bool accumulate) {
  if (rows <= 0 || rows > 7 || M <= 0 || N <= 0 || N % 4 != 0 || ld_act % 4 != 0) return BLH_ERR_SHAPE;
  const int64_t threads = (int64_t)M * (N / 4);
  hipLaunchKernelGGL(wgrad_tail_h_kernel, dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, s, dz, ld_dz, act, ld_act,
                     rows, M, N, out, accumulate ? 1 : 0);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

int launch_cast_bf16_f32(hipStream_t s, const uint16_t* src, float* dst, int64_t n) {
  if (n % 4 != 0) return BLH_ERR_SHAPE;
  const int64_t blocks = std::min<int64_t>(ceil_div(n / 4, 256), 4096);
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, dst, n);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

}  // namespace blh
