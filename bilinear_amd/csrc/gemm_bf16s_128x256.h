// bf16-storage GEMM for gfx950, 128 x 256 tile on a three-deep LDS ring — the form of the eight-phase
// kernel (gemm_bf16s_256.h) for launches that a 256 x 256 grid cannot spread over the 256 CUs:
// M = 8192 at W = 1024 (BASELINE configs[3]: 8192 poses per GPU, every strong-scaling split of
// configs[2]) is 128 tiles of 256 x 256 but 256 of 128 x 256.  Same contractions, operand layouts,
// half-tile LDS images, fragment reads and epilogue as that kernel (its header has the maps).
//
//   * 512 threads = 8 waves as 2 (M) x 4 (N); a wave owns 64 x 64 outputs = two quadrants of 64 x 32
//     (q_n = 0, 1: tile columns q_n*128 + w_c*32 .. +31), acc[q_n][4][2] of f32x4 = 64 VGPRs, on
//     v_mfma_f32_16x16x32_bf16.  Per 64-deep K tile a wave reads its A fragments once (8 ds_read_b128)
//     and each B quadrant once (4 + 4): 16 reads per 32 MFMAs (the 256 x 128 orientation needs 20).
//   * K tile 64; one LDS buffer = {A, B0, B1} half-tiles of 16 KiB (128 rows x 64 k each); THREE
//     buffers = 144 KiB.  A K tile is only two phases of 16 MFMAs, about half a microsecond: with two
//     buffers a half-tile could be requested at most two phases before its first read, less than an
//     LDS-DMA round trip through L2; the third buffer lets the DMA stream run a whole K tile further ahead.
//   * One phase = {fragment reads | three LDS-DMA instructions (1.5 half-tiles) | raw s_barrier |
//     16 MFMAs under s_setprio 1 | raw s_barrier}.  K tile T (buffer T % 3): phase 1 reads B0 and A,
//     issues B0 and the first half of A of tile T + 2; phase 2 reads B1, issues the rest of A and B1.
//     The buffer being refilled was last read one K tile earlier: every half-tile is re-filled at least
//     two phases after its last read (the rule of the eight-phase template; no lgkmcnt trick needed).
//   * `s_waitcnt vmcnt(6)` once per K tile, in phase 2 in front of the first barrier: it retires the six
//     DMAs of tile T + 1 (requested during tile T - 1) and leaves tile T + 2's in flight; tile T + 1 is
//     read from the next phase on.  The waves 4-7 run one barrier behind the waves 0-3.
//
// Restrictions (the host falls back to the 128 x 128 kernel otherwise): N % 256 == 0, every reduction
// slab a multiple of 64 and at least 128 deep, 16-byte rows; a KROW A operand needs M % 128 == 0.
// M may be ragged for ROWK A (rows beyond M are clamped on load and not stored).
#pragma once
#include "gemm_bf16s_256.h"

namespace blh {

static constexpr int H128_BUF_BYTES = 3 * H256_HALF_BYTES;      // A, B0, B1
static constexpr size_t H128_LDS_BYTES = 3 * H128_BUF_BYTES;    // 144 KiB

// DMA plan of one operand with NH half-tiles of 128 rows (A: 1, B: 2): per-lane byte offsets from
// the tile's first row, a wave-uniform 64-bit base advanced by one K tile with advance()
template <int LAYOUT, int NH>
struct PlanH128 {
  uint32_t voff[NH][2];    // [half][chunk]
  const bf16_bits* sbase;
  int64_t tile_step;
  uint32_t wave_off;

  __device__ inline void init(const bf16_bits* __restrict__ base, int64_t ld, int row0, int rows_limit,
                              int k_first, int tid) {
    wave_off = __builtin_amdgcn_readfirstlane((uint32_t)(tid & ~63) * 16u);
    const int last = rows_limit - 1 - row0;       // >= 0: the tile exists
    if (LAYOUT == ROWK) {
      sbase = base + (int64_t)row0 * ld + k_first;
      tile_step = 64;
    } else {
      sbase = base + (int64_t)k_first * ld + row0;
      tile_step = (int64_t)64 * ld;
    }
#pragma unroll
    for (int h = 0; h < NH; ++h)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int q = tid + p * 512;              // 16-B chunk of the half-tile image
        if (LAYOUT == ROWK) {
          const int r = q >> 3, c = q & 7;        // image row, slot
          const int kk = (c ^ rowk_swz_h<64>(r)) << 3;
          voff[h][p] = (uint32_t)(((int64_t)min(h * 128 + r, last) * ld + kk) * 2);
        } else {
          // image chunk q (16 B = 8 m of one k row): the KROW image map of gemm_bf16s_256.h
          const int k = 32 * (q >> 9) + 16 * ((q >> 6) & 1) + 8 * ((q >> 3) & 1) + 4 * ((q >> 8) & 1) + ((q >> 1) & 3);
          const int m8 = h * 128 + 64 * ((q >> 7) & 1) + 16 * ((q >> 4) & 3) + 8 * (q & 1);
          voff[h][p] = (uint32_t)(((int64_t)k * ld + min(m8, last - 7)) * 2);   // rows_limit % 8 == 0
        }
      }
  }
  // one DMA instruction: chunk P (8 KiB of the workgroup) of half-tile HALF into the image at lds_half
  template <int HALF, int P, bool FIRST>
  __device__ inline void issue(uint32_t lds_half) {
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_half + wave_off + (uint32_t)P * 512u * 16u);
    lds_dma16_sbase<FIRST>(voff[HALF][P], reinterpret_cast<const float*>(sbase), dst);
  }
  __device__ inline void advance() { sbase += tile_step; }
};

// ABL (tools/bf16s_bench only): 1 no DMA in the loop, 2 no fragment reads, 3 no MFMAs (results are then wrong);
// 4: correct results + 100 MHz real-time stamps of wave 0 {entry, loop start, loop end, stores issued, stores
// drained} per workgroup into the buffer p.addend points to
// SYNC (experiment, tools/bf16s_bench): 0 = two barriers per phase, waves 4-7 one barrier behind (shipped);
// 1 = ONE barrier per phase (in front of the MFMAs), no stagger — legal on the three-deep ring: a half-tile is
// re-filled a whole K tile after its last read, and every wave has consumed those reads (its MFMAs waited for
// them) before it arrives at the barrier the re-filling wave has passed
template <int LA, int LB, int EPI, bool OUT_BF16, int ABL = 0, int SYNC = 0>
__global__ __launch_bounds__(512, 2) void gemm_bf16s_128x256_kernel(GemmParamsH p) {
  constexpr int BM = 128, BN = 256;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const bf16_bits* lds = reinterpret_cast<const bf16_bits*>(smem);

  uint64_t stamp[5] = {0, 0, 0, 0, 0};
  if (ABL == 4) stamp[0] = __builtin_amdgcn_s_memrealtime();
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = p.N / BN;
  int tile = 0, slab = blockIdx.z;
  if (!(gridDim.z > 1 && xcd_remap_split(blockIdx.x, blockIdx.z, gridDim.x, gridDim.z, (p.M + BM - 1) / BM,
                                         tiles_n, &tile, &slab)))
    tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kz0 = slab * p.k_per_split;
  const int k_end = min(p.K, kz0 + p.k_per_split);
  const int nkt = (k_end - kz0) >> 6;            // K tiles (host: extent % 64 == 0, >= 128)
  const int64_t coff = (int64_t)slab * p.c_split_stride;
  void* C = OUT_BF16 ? (void*)(reinterpret_cast<bf16_bits*>(p.C) + coff)
                     : (void*)(reinterpret_cast<float*>(p.C) + coff);

  f32x4 acc[1][2][4][2];
#pragma unroll
  for (int b = 0; b < 2; ++b)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[0][b][i][j][r] = 0.f;

  PlanH128<LA, 1> planA;
  PlanH128<LB, 2> planB;
  planA.init(p.A, p.lda, m0, p.M, kz0, tid);
  planB.init(p.B, p.ldb, n0, p.N, kz0, tid);
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) float*)smem);
  constexpr uint32_t OA = 0, OB0 = H256_HALF_BYTES, OB1 = 2 * H256_HALF_BYTES;

  bf16x8_t fa[4][2], fb[2][2][2];

#define BLH_SB() __builtin_amdgcn_sched_barrier(0)
#define BLH_LOAD_A(BOFF) if (ABL != 2) read_frags_256<LA, 4>(fa, lds + ((BOFF) + OA) / 2, wr * 64, lane)
#define BLH_LOAD_B(QN, BOFF) if (ABL != 2) read_frags_256<LB, 2>(fb[QN], lds + ((BOFF) + ((QN) ? OB1 : OB0)) / 2, wc * 32, lane)
#define BLH_MFMA(QN)                                                                              \
  __builtin_amdgcn_s_setprio(1);                                                                  \
  if (ABL != 3) {                                                                                 \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                   \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                   \
    acc[0][QN][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][ks], fb[QN][j][ks],          \
                                                               acc[0][QN][i][j], 0, 0, 0);        \
  } else { _Pragma("unroll") for (int i = 0; i < 4; ++i) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) \
    { asm volatile("" :: "v"(fa[i][ks])); } _Pragma("unroll") for (int j = 0; j < 2; ++j)         \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) { asm volatile("" :: "v"(fb[QN][j][ks])); } } \
  __builtin_amdgcn_s_setprio(0);
#define BLH_BAR() do { BLH_SB(); __builtin_amdgcn_s_barrier(); BLH_SB(); } while (0)
#define BLH_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
  // the six DMA instructions of one K tile into the buffer at byte offset ROFF, as two groups of three
#define BLH_STAGE_1(ROFF)                                                                         \
  planB.template issue<0, 0, true>(lds0 + (ROFF) + OB0);                                          \
  planB.template issue<0, 1, false>(lds0 + (ROFF) + OB0);                                         \
  planA.template issue<0, 0, true>(lds0 + (ROFF) + OA);
#define BLH_STAGE_2(ROFF)                                                                         \
  planA.template issue<0, 1, false>(lds0 + (ROFF) + OA);                                          \
  planB.template issue<1, 0, false>(lds0 + (ROFF) + OB1);                                         \
  planB.template issue<1, 1, false>(lds0 + (ROFF) + OB1);                                         \
  planA.advance(); planB.advance();

  // prologue: K tiles 0 and 1 requested (buffers 0 and 1), tile 0 landed
  BLH_STAGE_1(0u) BLH_STAGE_2(0u)
  BLH_STAGE_1((uint32_t)H128_BUF_BYTES) BLH_STAGE_2((uint32_t)H128_BUF_BYTES)
  BLH_WAIT_VM(6);
  BLH_BAR();
  if (SYNC == 0 && wr == 1) BLH_BAR();           // waves 4-7 run one barrier behind waves 0-3
  if (ABL == 4) stamp[1] = __builtin_amdgcn_s_memrealtime();

  // K tile in the buffer at byte offset BOFF; STG: request tile + 2 into the buffer at ROFF
#define BLH_KTILE(BOFF, STG, ROFF, WAIT)                                                          \
  /* phase 1: quadrant 0 */                                                                       \
  BLH_LOAD_B(0, BOFF); BLH_SB(); BLH_LOAD_A(BOFF); BLH_SB();                                      \
  if (STG) { BLH_STAGE_1(ROFF) }                                                                  \
  BLH_BAR(); BLH_MFMA(0) if (SYNC == 0) BLH_BAR();                                                \
  /* phase 2: quadrant 1 */                                                                       \
  BLH_LOAD_B(1, BOFF); BLH_SB();                                                                  \
  if (STG) { BLH_STAGE_2(ROFF) }                                                                  \
  WAIT;                                                                                           \
  BLH_BAR(); BLH_MFMA(1) if (SYNC == 0) BLH_BAR();

  uint32_t boff = 0, roff = 2u * H128_BUF_BYTES;  // buffer of the current tile / of tile + 2
  if (ABL == 2) {   // (fragments that no read defines)
#pragma unroll
    for (int i = 0; i < 4; ++i) for (int ks = 0; ks < 2; ++ks) fa[i][ks] = bf16x8_t{};
#pragma unroll
    for (int q = 0; q < 2; ++q) for (int j = 0; j < 2; ++j) for (int ks = 0; ks < 2; ++ks) fb[q][j][ks] = bf16x8_t{};
  }
  for (int t = 0; t < nkt - 2; ++t) {
    BLH_KTILE(boff, ABL != 1, roff, BLH_WAIT_VM(6))
    roff = boff;
    boff = (boff == 2u * H128_BUF_BYTES) ? 0u : boff + (uint32_t)H128_BUF_BYTES;
  }
  BLH_KTILE(boff, false, 0u, BLH_WAIT_VM(0))
  boff = (boff == 2u * H128_BUF_BYTES) ? 0u : boff + (uint32_t)H128_BUF_BYTES;
  BLH_KTILE(boff, false, 0u, (void)0)
  if (SYNC == 0 && wr == 0) BLH_BAR();           // the barrier waves 4-7 took at the start

#undef BLH_KTILE
#undef BLH_STAGE_1
#undef BLH_STAGE_2
#undef BLH_LOAD_A
#undef BLH_LOAD_B
#undef BLH_MFMA
#undef BLH_BAR
#undef BLH_WAIT_VM
#undef BLH_SB
  __syncthreads();
  if (ABL == 4) {
    stamp[2] = __builtin_amdgcn_s_memrealtime();
    GemmParamsH q = p;
    q.addend = nullptr;
    gemm_epilogue_256<EPI, OUT_BF16, 1>(acc, q, C, smem, m0, n0, tile_m);
    stamp[3] = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    stamp[4] = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
      uint64_t* dbg = reinterpret_cast<uint64_t*>(const_cast<bf16_bits*>(p.addend)) + (size_t)blockIdx.x * 8;
#pragma unroll
      for (int i = 0; i < 5; ++i) dbg[i] = stamp[i];
      dbg[5] = (uint64_t)tile;
    }
    return;
  }
  if constexpr (EPI == EPI_BN_BWD) gemm_epilogue_256_bnbwd_packed<1>(acc, p, C, smem, m0, n0, tile_m);
  else if constexpr (EPI == EPI_BN_BWD_ADD) gemm_epilogue_256_bnbwd<1, true>(acc, p, C, smem, m0, n0, tile_m);
  else gemm_epilogue_256<EPI, OUT_BF16, 1>(acc, p, C, smem, m0, n0, tile_m);
}

}  // namespace blh
