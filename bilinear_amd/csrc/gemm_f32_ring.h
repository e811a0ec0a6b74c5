// fp32 MFMA GEMM for gfx950, LDS-DMA ring pipeline — the shipped form of the dominant kernel.
//
//   C[M,N] = A (M x K) * B (K x N) on v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD,
//   157.3 TFLOP/s peak).  Contractions, operand layouts (ROWK / KROW) and epilogues are those of
//   tools/gemm_f32_kernel.h (the register-staged reference kernel kept for tools/gemm_bench); this
//   file differs in how the operands reach the matrix cores:
//
//   * HBM/L2 -> LDS by LDS-DMA in its SCALAR-BASE form (`global_load_lds_dwordx4 voff, s[base]`):
//     the per-lane 32-bit byte offsets are loop invariants, the tile advance is one 64-bit
//     scalar add per operand per K tile, so a DMA costs 3 instructions in the loop (M0 write,
//     nop, load) instead of a 64-bit vector pointer add + select per DMA.
//     Rows beyond the operand are CLAMPED to the last valid row instead of zero-filled: such
//     rows only feed output rows / columns that are never stored (each output element depends
//     on its own A row and B column only).  Only a ragged LAST K tile needs true zeros (both
//     operands' k beyond K multiply each other): that one tile takes the per-lane-pointer path
//     with the 16-byte zero page.
//   * STAGES-deep ring of unpadded BKT-wide tiles, tile kt+STAGES-1 in flight while tile kt is
//     multiplied, ONE raw s_barrier per K tile behind a counted s_waitcnt vmcnt.
//   * ROWK fragment reads are ds_read_b128; the DMA writes LDS linearly, so the bank-conflict
//     fix is an XOR swizzle on the per-lane SOURCE chunk and again on the read.  The swizzle key
//     must make the 16 lanes of a ds_read_b128 lane group ({0-3,12-15,20-27}, {4-11,16-19,28-31}
//     of each half-wave) hit 16 different 16-B slots of the 256-B bank row: with 128-B rows
//     (BKT 32) two rows share a bank row, so the key is (row >> 1) & 7 — the round-1 key
//     row & 7 left every group 2-way conflicted (SQ_LDS_BANK_CONFLICT = 50 % of
//     SQ_LDS_IDX_ACTIVE, profiles/r01_pmc_gemm.md); with 256-B rows (BKT 64) it is row & 15.
#pragma once
#include "common.h"
#include "gemm_epilogue.h"
#include "gemm_dma.h"          // g_zero16, lds_dma16_asm, xcd_remap

namespace blh {

template <int BKT>
__device__ __forceinline__ int ring_swz(int row) {
  return BKT == 32 ? ((row >> 1) & 7) : (row & 15);
}

// LDS-DMA, scalar base + per-lane 32-bit byte offset.  `first`: the base SGPR pair may have been
// written by the SALU just before this statement (tile advance): pad the SALU-write -> VMEM-read
// hazard inside the string (hipcc pads nothing inside asm).
template <bool FIRST>
__device__ __forceinline__ void lds_dma16_sbase(uint32_t voff, const float* sbase,
                                                uint32_t lds_byte_addr_uniform) {
  if (FIRST)
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(voff), "s"(sbase), "s"(lds_byte_addr_uniform)
        : "memory", "m0");
  else
    asm volatile(
        "s_mov_b32 m0, %2\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %0, %1"
        :
        : "v"(voff), "s"(sbase), "s"(lds_byte_addr_uniform)
        : "memory", "m0");
}

template <int LAYOUT, int R, int NT, int BKT>
struct RingPlan {
  static constexpr int CHUNKS = (R * BKT / 4) / NT;   // 16-B chunks per thread per tile
  static_assert((R * BKT / 4) % NT == 0, "tile not divisible among threads");
  uint32_t voff[CHUNKS];   // byte offset of this lane's chunk from the tile base
  int koff[CHUNKS];        // first k of the chunk inside the tile (ragged last tile only)
  const float* sbase;      // wave-uniform: operand at (row0, first k of the current tile)
  int64_t tile_step;       // floats per K tile
  uint32_t wave_off;       // (tid & ~63) * 16
  bool ragged_k;

  __device__ inline void init(const float* __restrict__ base, int64_t ld, int row0, int rows_limit,
                              int k_first, int k_end, int tid) {
    ragged_k = ((k_end - k_first) % BKT) != 0;
    wave_off = __builtin_amdgcn_readfirstlane((uint32_t)(tid & ~63) * 16u);
    const int last = rows_limit - 1 - row0;   // >= 0: the tile exists
    if (LAYOUT == ROWK) {
      sbase = base + (int64_t)row0 * ld + k_first;
      tile_step = BKT;
    } else {
      sbase = base + (int64_t)k_first * ld + row0;
      tile_step = (int64_t)BKT * ld;
    }
#pragma unroll
    for (int p = 0; p < CHUNKS; ++p) {
      const int q = tid + p * NT;
      if (LAYOUT == ROWK) {
        constexpr int CPR = BKT / 4;            // chunks per row
        const int r = q / CPR, c = q % CPR;     // LDS row, LDS slot
        const int kk = (c ^ ring_swz<BKT>(r)) << 2;
        koff[p] = kk;
        voff[p] = (uint32_t)(((int64_t)min(r, last) * ld + kk) * 4);
      } else {
        constexpr int CPR = R / 4;              // chunks per k row
        const int kk = q / CPR, r4 = (q % CPR) << 2;
        koff[p] = kk;
        voff[p] = (uint32_t)(((int64_t)kk * ld + min(r4, last - 3)) * 4);   // rows_limit % 4 == 0
      }
    }
  }

  // DMA the tile whose first k is k0 into the LDS tile at byte address lds_tile, then advance
  __device__ inline void issue(uint32_t lds_tile, int k0, int k_end) {
    if (ragged_k && k0 + BKT > k_end) {
#pragma unroll
      for (int p = 0; p < CHUNKS; ++p) {
        const float* g = (k0 + koff[p] < k_end)
                             ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(sbase) + voff[p])
                             : reinterpret_cast<const float*>(&g_zero16);
        lds_dma16_asm(g, lds_tile + wave_off + (uint32_t)(p * NT * 16));
      }
    } else {
      lds_dma16_sbase<true>(voff[0], sbase, lds_tile + wave_off);
#pragma unroll
      for (int p = 1; p < CHUNKS; ++p)
        lds_dma16_sbase<false>(voff[p], sbase, lds_tile + wave_off + (uint32_t)(p * NT * 16));
    }
    sbase += tile_step;
  }
};

// LDS -> MFMA fragments of k-group s (8 consecutive k) of a BKT-wide tile
template <int LAYOUT, int R, int T, int BKT>
__device__ inline void ring_read_frags(float (&frag)[T][4], const float* lds, int row_base, int s,
                                       int lane) {
  const int h = lane >> 5, lr = lane & 31;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (LAYOUT == ROWK) {
      const int row = row_base + t * 32 + lr;
      const float4 v = *reinterpret_cast<const float4*>(
          lds + row * BKT + (((2 * s + h) ^ ring_swz<BKT>(row)) << 2));
      frag[t][0] = v.x; frag[t][1] = v.y; frag[t][2] = v.z; frag[t][3] = v.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        frag[t][j] = lds[(8 * s + 4 * h + j) * R + row_base + t * 32 + lr];
    }
  }
}

template <int BM, int BN, int BKT, int STAGES>
constexpr size_t gemm_ring_lds_bytes() {
  return (size_t)STAGES * (BM + BN) * BKT * sizeof(float);
}

template <int N>
__device__ __forceinline__ void ring_wait_vm_lgkm() {
  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(N) : "memory");
}

// STAMP: diagnostic builds only (tools/gemm_bench): 100 MHz real-time / shader-clock stamps at
// entry, main-loop start, main-loop end and kernel end, written to p.loss_part (8 u64 per WG).
// STAMP 2..5 additionally ablate one part of the steady-state loop (results are then wrong; only
// the timing is read): 2 = no barrier, 3 = no DMA issue, 4 = no fragment reads, 5 = no wait + barrier,
// 6 = none of them (MFMAs only), 7 = no MFMAs (DMA, fragment reads and barriers only).
// Measured (M = 4096, N = K = 1024, bk64 x 2): full loop 146.3 k cycles against the MFMA-ideal
// 131.1 k; without the barrier 141.6 k, without the DMA 139.4 k, without the fragment reads
// 143.1 k: no single part explains the last 6 % (round 1 measured 33.4 cycles per 32-cycle bf16 MFMA in a loop of
// MFMAs and fragment reads only: back-to-back issue itself sits ~4 % above the instruction's pass count).
// STAMP 6 (MFMAs only): 134.6 k cycles, 62.8 us per launch = the ceiling of this launch shape (0.87 of peak).
// PRE (experiment, tools/k3_bench; never 1 in the library): SURVEY K3 "normalise on load" — the A
// operand is the PRE-BatchNorm tensor Z of the producing stage and BatchNorm-apply + ReLU (+ the
// dropout factor 2) run on the A fragments between the LDS read and the MFMAs, with the per-feature
// (= per k) scale / shift in an LDS table behind the ring (p.bn_gamma = 2 scale, p.bn_beta = 2 shift).
// Measured at M = 4096, N = K = 1024 (profiles/r03_k3_normalise_on_load.md).
template <int BM, int BN, int WM, int WN, int LA, int LB, int EPI, int BKT = 32, int STAGES = 3, int STAMP = 0,
          int PRE = 0>
__global__ __launch_bounds__(64 * WM * WN) void gemm_f32_ring_kernel(GemmParams p) {
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile must be at least 32x32");
  static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
  constexpr int RING = (BM + BN) * BKT;          // floats per stage
  constexpr int NG = BKT / 8;                    // k-groups per tile
  using PA = RingPlan<LA, BM, NT, BKT>;
  using PB = RingPlan<LB, BN, NT, BKT>;
  constexpr int G = PA::CHUNKS + PB::CHUNKS;     // DMAs per thread per tile
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned long long stamp_entry = 0, stamp_c0 = 0, stamp_r0 = 0;
  if (STAMP) stamp_entry = __builtin_amdgcn_s_memrealtime();

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (p.N + BN - 1) / BN;
  int tile = 0, slab = blockIdx.z;
  if (!(gridDim.z > 1 && xcd_remap_split(blockIdx.x, blockIdx.z, gridDim.x, gridDim.z, (p.M + BM - 1) / BM,
                                         tiles_n, &tile, &slab)))
    tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kz0 = slab * p.k_per_split;
  const int k_end = min(p.K, kz0 + p.k_per_split);
  float* __restrict__ C = p.C + (int64_t)slab * p.c_split_stride;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nkt = (k_end - kz0 + BKT - 1) / BKT;
  PA planA;
  PB planB;
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) float*)smem);
  planA.init(p.A, p.lda, m0, p.M, kz0, k_end, tid);
  planB.init(p.B, p.ldb, n0, p.N, kz0, k_end, tid);
  float fa[2][TM][4], fb[2][TN][4];
  if (STAMP) { stamp_c0 = __builtin_amdgcn_s_memtime(); stamp_r0 = __builtin_amdgcn_s_memrealtime(); }
  // PRE: scale / shift table [2][K] behind the ring, per-lane prefetch registers for one k-group
  float* pre_tab = smem + STAGES * RING;
  float4 psc[2], psh[2];
  if (PRE) {
    for (int k = tid; k < p.K; k += NT) { pre_tab[k] = p.bn_gamma[k]; pre_tab[p.K + k] = p.bn_beta[k]; }
    __syncthreads();
  }
  auto pre_load = [&](int buf, int kbase) {   // this lane's 4 k of the k-group starting at kbase
    if (PRE) {
      const int k = min(kbase + 4 * (lane >> 5), p.K - 4);
      psc[buf] = *reinterpret_cast<const float4*>(pre_tab + k);
      psh[buf] = *reinterpret_cast<const float4*>(pre_tab + p.K + k);
    }
  };
  auto pre_apply = [&](int buf) {
    if (PRE) {
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        fa[buf][i][0] = fmaxf(fmaf(fa[buf][i][0], psc[buf].x, psh[buf].x), 0.f);
        fa[buf][i][1] = fmaxf(fmaf(fa[buf][i][1], psc[buf].y, psh[buf].y), 0.f);
        fa[buf][i][2] = fmaxf(fmaf(fa[buf][i][2], psc[buf].z, psh[buf].z), 0.f);
        fa[buf][i][3] = fmaxf(fmaf(fa[buf][i][3], psc[buf].w, psh[buf].w), 0.f);
      }
    }
  };

  // prologue: tiles 0 .. STAGES-2 in flight; wait for tile 0 only
  if (nkt > 0) {
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
      if (t < nkt) {
        planA.issue(lds0 + t * (RING * 4), kz0 + t * BKT, k_end);
        planB.issue(lds0 + t * (RING * 4) + BM * BKT * 4, kz0 + t * BKT, k_end);
      }
    const int ahead = min(nkt, STAGES - 1) - 1;   // tiles issued beyond tile 0
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    ring_read_frags<LA, BM, TM, BKT>(fa[0], smem, wm * (TM * 32), 0, lane);
    ring_read_frags<LB, BN, TN, BKT>(fb[0], smem + BM * BKT, wn * (TN * 32), 0, lane);
    pre_load(0, kz0);
  }
  // One k-group: issue the fragment reads of the NEXT group, then the MFMAs of this one.
  // The order is pinned with sched_barrier: left alone, hipcc sinks the ds_reads below the MFMAs
  // and waits for them right in front of the next group's first MFMA (seen in the .s of the
  // round-1 kernel): the register double buffer is defeated and every k-group boundary exposes
  // the LDS latency, with both waves of a SIMD in lockstep.
#define BLH_RING_MFMAS(CUR)                                                                     \
  if (STAMP != 7)                                                                               \
  _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                 \
  _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                \
  _Pragma("unroll") for (int jn = 0; jn < TN; ++jn)                                             \
    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[CUR][i][j], fb[CUR][jn][j], acc[i][jn], 0, 0, 0);

  // steady state: a next tile exists (straight-line body: the compiler's lgkmcnt bookkeeping
  // stays exact, one branch-free iteration per K tile apart from the DMA issue and the wait count)
  int st_cur = 0;                                  // kt % STAGES
  for (int kt = 0; kt < nkt - 1; ++kt) {
    const int st_nxt = (st_cur == STAGES - 1) ? 0 : st_cur + 1;
    const int st_new = (st_cur == 0) ? STAGES - 1 : st_cur - 1;   // (kt + STAGES - 1) % STAGES
    const float* sA = smem + st_cur * RING;
    const float* sB = sA + BM * BKT;
    const float* nA = smem + st_nxt * RING;
    // Group 0's fragments were read right after the previous barrier, a whole MFMA group ago:
    // this wait is free, and — being the builtin, which hipcc's waitcnt bookkeeping understands —
    // it tells the compiler that nothing is outstanding at the loop head (its merge of the loop
    // entry and the back edge is conservative and otherwise puts lgkmcnt(0) BEHIND the reads of
    // group 1, exposing one LDS latency per K tile).  0xC07F = lgkmcnt(0), vmcnt/expcnt untouched.
    __builtin_amdgcn_s_waitcnt(0xC07F);
    // the stage being refilled is the one read in iteration kt-1: every wave's reads of it were
    // retired (lgkmcnt(0)) before that iteration's barrier
    if (kt + STAGES - 1 < nkt && STAMP != 3 && STAMP != 6) {
      const int k0 = kz0 + (kt + STAGES - 1) * BKT;
      planA.issue(lds0 + st_new * (RING * 4), k0, k_end);
      planB.issue(lds0 + st_new * (RING * 4) + BM * BKT * 4, k0, k_end);
    }
#pragma unroll
    for (int s = 0; s < NG; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      if (STAMP == 4 || STAMP == 6) {
        // (ablation: fragments are not re-read)
      } else if (s < NG - 1) {
        ring_read_frags<LA, BM, TM, BKT>(fa[nxt], sA, wm * (TM * 32), s + 1, lane);
        ring_read_frags<LB, BN, TN, BKT>(fb[nxt], sB, wn * (TN * 32), s + 1, lane);
        pre_load(nxt, kz0 + kt * BKT + 8 * (s + 1));
      } else {
        ring_read_frags<LA, BM, TM, BKT>(fa[nxt], nA, wm * (TM * 32), 0, lane);
        ring_read_frags<LB, BN, TN, BKT>(fb[nxt], nA + BM * BKT, wn * (TN * 32), 0, lane);
        pre_load(nxt, kz0 + (kt + 1) * BKT);
      }
      __builtin_amdgcn_sched_barrier(0);
      pre_apply(cur);
      BLH_RING_MFMAS(cur)
      __builtin_amdgcn_sched_barrier(0);
      if (s == NG - 2) {
        // tile kt+1 must have landed (it is read from the next k-group on); tiles kt+2 ..
        // kt+STAGES-1, as far as they exist, stay in flight across the barrier
        const int ahead = min(nkt - 1, kt + STAGES - 1) - (kt + 1);
        if (STAMP == 5 || STAMP == 6) {
          // (ablation: neither wait nor barrier)
        } else if (STAGES >= 4 && ahead >= 2) ring_wait_vm_lgkm<2 * G>();
        else if (STAGES >= 3 && ahead >= 1) ring_wait_vm_lgkm<G>();
        else ring_wait_vm_lgkm<0>();
        if (STAMP != 2 && STAMP != 5 && STAMP != 6) __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
    st_cur = st_nxt;
  }
  if (nkt > 0) {   // last tile: nothing left to prefetch
    const float* sA = smem + st_cur * RING;
    const float* sB = sA + BM * BKT;
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
    for (int s = 0; s < NG; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      if (s < NG - 1) {
        ring_read_frags<LA, BM, TM, BKT>(fa[nxt], sA, wm * (TM * 32), s + 1, lane);
        ring_read_frags<LB, BN, TN, BKT>(fb[nxt], sB, wn * (TN * 32), s + 1, lane);
        pre_load(nxt, kz0 + (nkt - 1) * BKT + 8 * (s + 1));
      }
      __builtin_amdgcn_sched_barrier(0);
      pre_apply(cur);
      BLH_RING_MFMAS(cur)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef BLH_RING_MFMAS
  __syncthreads();
  if (STAMP) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.loss_part) + 8 * (blockIdx.x + gridDim.x * blockIdx.z);   // (any order)
      o[0] = c1 - stamp_c0; o[1] = r1 - stamp_r0; o[2] = stamp_entry; o[3] = stamp_r0; o[4] = r1;
    }
  }
  gemm_epilogue<BM, BN, WM, WN, EPI>(acc, p, C, smem, m0, n0, tile_m, true);
  if (STAMP) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long r2 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0)
      (reinterpret_cast<unsigned long long*>(p.loss_part) + 8 * (blockIdx.x + gridDim.x * blockIdx.z))[5] = r2;
  }
}

}  // namespace blh
