// fp32 MFMA GEMM for gfx950 — device templates (included by gemm_f32.hip and tools/) (MI355X), the dominant kernel of the lifter MLP.
//
//   C[M,N] = A (M x K) * B (K x N), fp32 in / fp32 accumulate on
//   v_mfma_f32_32x32x2_f32 (exact fp32: a k-ordered fmaf chain, no TF32 exists
//   on gfx950).  Peak 157.3 TFLOP/s (64 FLOP/clk/SIMD).
//
// One kernel template serves the three contractions of a Linear layer
// (reference call-sites: nn.Linear in /root/reference/model/bilinear.py:9,29 and
// its autograd, train_bilinear.py:79):
//   forward  Z  = A  W^T        A=[B,K] ROWK , W=[N,K]  ROWK
//   dgrad    dA = dZ W          dZ=[B,N'] ROWK, W=[N',K'] KROW (reduction index is W's row)
//   wgrad    dW = dZ^T A        dZ=[B,N'] KROW, A=[B,K'] KROW (reduction over the batch)
// ROWK = the reduction index is the contiguous one in memory; KROW = the
// output index is contiguous.
//
// Data movement per workgroup (512 threads = 8 waves of 64, tile 128x128, K tile 32), PIPE 3:
//   HBM/L2 --global_load_lds_dwordx4 (LDS-DMA: 16 B/lane, coalesced along the contiguous axis,
//   no VGPR staging, no ds_write)--> 3-stage ring of unpadded LDS tiles, tile kt+2 in flight
//   while tile kt is multiplied; counted s_waitcnt vmcnt + one raw s_barrier per K tile
//   --ds_read_b128 (ROWK; XOR swizzle on the DMA source address and on the read) or
//   ds_read_b32 (KROW; 32 consecutive lanes = 32 consecutive banks)--> MFMA, with the
//   fragments of k-group s+1 read while the MFMAs of group s issue.
//   PIPE 1 is the register-staged double buffer (global_load_dwordx4 -> VGPR -> ds_write_b128
//   into 36-float padded rows), kept as the A/B baseline of tools/gemm_bench.hip together
//   with the STAMP (in-kernel clock stamps) and ABLATE (timing-only) diagnostic switches.
// Each wave owns a (BM/WM) x (BN/WN) sub-tile as TM x TN accumulators of 32x32.
//
// Fragment/k ordering: a lane of half h = lane>>5 reads 4 consecutive k
// (8s+4h .. 8s+4h+3) of its row; MFMA j of group s therefore contracts
// k = 8s+j (half 0) and 8s+4+j (half 1).  Both operands use the same map, so
// the permutation of k inside a group is immaterial.
#pragma once
#include "../common.h"
#include "../gemm_epilogue.h"
#include "../gemm_dma.h"

namespace blh {

static constexpr int BK = 32;
static constexpr int ROWK_PITCH = BK + 4;   // floats; 36*4 B rows -> ds_read_b128 conflict-free

template <int LAYOUT, int R>
struct TileGeom {
  static constexpr int LDS_FLOATS = (LAYOUT == ROWK) ? R * ROWK_PITCH : BK * R;
};

// ---- global -> registers ----------------------------------------------------
template <int LAYOUT, int R, int NT>
struct TileIO {
  static constexpr int CHUNKS = (R * BK / 4) / NT;   // float4 chunks per thread
  static_assert((R * BK / 4) % NT == 0, "tile not divisible among threads");

  // rows_limit: number of valid rows of this operand (M or N); k_end: end of the
  // reduction range of this workgroup.
  __device__ static inline void load(float4 (&reg)[CHUNKS], const float* __restrict__ base,
                                     int64_t ld, int row0, int rows_limit, int k0, int k_end,
                                     int tid) {
#pragma unroll
    for (int p = 0; p < CHUNKS; ++p) {
      const int q = tid + p * NT;
      int row, k;
      if (LAYOUT == ROWK) {
        row = row0 + (q >> 3);
        k = k0 + ((q & 7) << 2);
      } else {
        constexpr int CPR = R / 4;
        row = row0 + ((q % CPR) << 2);
        k = k0 + (q / CPR);
      }
      const bool ok = (row < rows_limit) && (k < k_end);
      const int64_t off = (LAYOUT == ROWK) ? ((int64_t)row * ld + k) : ((int64_t)k * ld + row);
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ok) v = *reinterpret_cast<const float4*>(base + off);
      reg[p] = v;
    }
  }

  __device__ static inline void store(const float4 (&reg)[CHUNKS], float* lds, int tid) {
#pragma unroll
    for (int p = 0; p < CHUNKS; ++p) {
      const int q = tid + p * NT;
      int idx;
      if (LAYOUT == ROWK) {
        idx = (q >> 3) * ROWK_PITCH + ((q & 7) << 2);
      } else {
        constexpr int CPR = R / 4;
        idx = (q / CPR) * R + ((q % CPR) << 2);
      }
      *reinterpret_cast<float4*>(lds + idx) = reg[p];
    }
  }
};

// ---- LDS -> MFMA fragments for k-group s (8 consecutive k) -----------------
// frag[t][j]: operand value of 32-row sub-tile t for MFMA j of the group.
template <int LAYOUT, int R, int T>
__device__ inline void read_frags(float (&frag)[T][4], const float* lds, int row_base, int s,
                                  int lane) {
  const int h = lane >> 5, lr = lane & 31;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (LAYOUT == ROWK) {
      const float4 v = *reinterpret_cast<const float4*>(
          lds + (row_base + t * 32 + lr) * ROWK_PITCH + 8 * s + 4 * h);
      frag[t][0] = v.x; frag[t][1] = v.y; frag[t][2] = v.z; frag[t][3] = v.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        frag[t][j] = lds[(8 * s + 4 * h + j) * R + row_base + t * 32 + lr];
    }
  }
}

// Loop-invariant part of a thread's DMA work hoisted out of the K loop: source pointers at
// the first tile (advanced by a constant stride per tile), row validity, and the wave-uniform
// LDS byte offsets.  What is left per DMA is a pointer add, a k-range compare + select of the
// zero page, two SALU adds and the 5-instruction M0 sequence.
template <int LAYOUT, int R, int NT>
struct DmaPlan {
  static constexpr int CHUNKS = (R * BK / 4) / NT;
  static_assert((R * BK / 4) % NT == 0, "tile not divisible among threads");
  const float* src[CHUNKS];   // chunk source at the current tile (zero page for invalid rows)
  int koff[CHUNKS];
  int64_t step[CHUNKS];       // floats per K tile (0 for invalid rows: they stay on the zero page)
  uint32_t wave_off;          // (tid & ~63) * 16, wave-uniform (SGPR)
  bool ragged_k;              // the reduction range is not a multiple of BK (kernel-uniform)

  __device__ inline void init(const float* __restrict__ base, int64_t ld, int row0, int rows_limit,
                              int k_first, int k_end, int tid) {
    const int64_t tile_step = (LAYOUT == ROWK) ? (int64_t)BK : (int64_t)BK * ld;
    ragged_k = ((k_end - k_first) % BK) != 0;
    wave_off = __builtin_amdgcn_readfirstlane((uint32_t)(tid & ~63) * 16u);
#pragma unroll
    for (int p = 0; p < CHUNKS; ++p) {
      const int q = tid + p * NT;
      int row, kk;
      if (LAYOUT == ROWK) {
        const int r = q >> 3;
        row = row0 + r;
        kk = (((q & 7) ^ (r & 7))) << 2;
      } else {
        constexpr int CPR = R / 4;
        row = row0 + ((q % CPR) << 2);
        kk = q / CPR;
      }
      const bool rowok = row < rows_limit;
      koff[p] = kk;
      const int64_t off = (LAYOUT == ROWK) ? ((int64_t)row * ld + k_first + kk)
                                           : ((int64_t)(k_first + kk) * ld + row);
      src[p] = rowok ? (base + off) : reinterpret_cast<const float*>(&g_zero16);
      step[p] = rowok ? tile_step : 0;
    }
  }

  // DMA tile whose first k is k0 into the LDS tile at byte address lds_tile (wave-uniform)
  __device__ inline void issue(uint32_t lds_tile, int k0, int k_end) {
#pragma unroll
    for (int p = 0; p < CHUNKS; ++p) {
      const float* g = src[p];
      if (ragged_k && !(k0 + koff[p] < k_end)) g = reinterpret_cast<const float*>(&g_zero16);
      lds_dma16_asm(g, lds_tile + wave_off + (uint32_t)(p * NT * 16));
      src[p] += step[p];
    }
  }
};

template <int LAYOUT, int R, int T>
__device__ inline void read_frags_dma(float (&frag)[T][4], const float* lds, int row_base, int s,
                                      int lane) {
  const int h = lane >> 5, lr = lane & 31;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    if (LAYOUT == ROWK) {
      const int row = row_base + t * 32 + lr;
      const float4 v = *reinterpret_cast<const float4*>(
          lds + row * BK + (((2 * s + h) ^ (row & 7)) << 2));
      frag[t][0] = v.x; frag[t][1] = v.y; frag[t][2] = v.z; frag[t][3] = v.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        frag[t][j] = lds[(8 * s + 4 * h + j) * R + row_base + t * 32 + lr];
    }
  }
}

template <int BM, int BN, int LA, int LB, int PIPE>
constexpr size_t gemm_lds_bytes() {
  return PIPE == 3 ? 3 * (size_t)(BM + BN) * BK * sizeof(float)
                   : 2 * (size_t)(TileGeom<LA, BM>::LDS_FLOATS + TileGeom<LB, BN>::LDS_FLOATS) * sizeof(float);
}

template <int BM, int BN, int WM, int WN, int LA, int LB, int EPI, int PIPE = 1, int STAMP = 0, int ABLATE = 0>
__global__ __launch_bounds__(64 * WM * WN) void gemm_f32_kernel(GemmParams p) {
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static_assert(TM >= 1 && TN >= 1, "wave tile must be at least 32x32");
  constexpr int A_FLOATS = TileGeom<LA, BM>::LDS_FLOATS;
  constexpr int B_FLOATS = TileGeom<LB, BN>::LDS_FLOATS;
  constexpr int STAGE = A_FLOATS + B_FLOATS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  unsigned long long stamp_entry = 0;
  if (STAMP) stamp_entry = __builtin_amdgcn_s_memrealtime();

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int nwg = gridDim.x;
  const int tile = xcd_remap(blockIdx.x, nwg);
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kz0 = blockIdx.z * p.k_per_split;
  const int k_end = min(p.K, kz0 + p.k_per_split);
  float* __restrict__ C = p.C + (int64_t)blockIdx.z * p.c_split_stride;

  using IOA = TileIO<LA, BM, NT>;
  using IOB = TileIO<LB, BN, NT>;
  float4 ra[IOA::CHUNKS], rb[IOB::CHUNKS];

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nkt = (k_end - kz0 + BK - 1) / BK;
  // STAMP: diagnostic builds only (tools/gemm_bench): shader clock vs 100 MHz real-time clock
  unsigned long long stamp_c0 = 0, stamp_r0 = 0;
  if (STAMP) { stamp_c0 = __builtin_amdgcn_s_memtime(); stamp_r0 = __builtin_amdgcn_s_memrealtime(); }
  if (PIPE != 3) {
    if (nkt > 0) {
      IOA::load(ra, p.A, p.lda, m0, p.M, kz0, k_end, tid);
      IOB::load(rb, p.B, p.ldb, n0, p.N, kz0, k_end, tid);
      IOA::store(ra, smem, tid);
      IOB::store(rb, smem + A_FLOATS, tid);
    }
    __syncthreads();
  }

  if (PIPE == 3) {
    // LDS-DMA pipeline: a 3-stage ring of unpadded tiles filled by global_load_lds (no VGPR
    // staging, no ds_write); tile kt+2 is in flight while tile kt is multiplied.  One raw
    // s_barrier per K tile, behind a COUNTED vmcnt that retires only tile kt+1's DMAs (a
    // __syncthreads() would drain the ring) and lgkmcnt(0) (this wave's reads of the stage
    // that the next iteration's DMA overwrites are complete).
    DmaPlan<LA, BM, NT> planA;
    DmaPlan<LB, BN, NT> planB;
    const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) float*)smem);
    planA.init(p.A, p.lda, m0, p.M, kz0, k_end, tid);
    planB.init(p.B, p.ldb, n0, p.N, kz0, k_end, tid);
    constexpr int RING = (BM + BN) * BK;          // floats per stage
    constexpr int G = DmaPlan<LA, BM, NT>::CHUNKS + DmaPlan<LB, BN, NT>::CHUNKS;   // DMAs per thread per tile
    float fa[2][TM][4], fb[2][TN][4];
    if (nkt > 0) {
      planA.issue(lds0, kz0, k_end);
      planB.issue(lds0 + BM * BK * 4, kz0, k_end);
      if (nkt > 1) {
        planA.issue(lds0 + RING * 4, kz0 + BK, k_end);
        planB.issue(lds0 + (RING + BM * BK) * 4, kz0 + BK, k_end);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      read_frags_dma<LA, BM, TM>(fa[0], smem, wm * (TM * 32), 0, lane);
      read_frags_dma<LB, BN, TN>(fb[0], smem + BM * BK, wn * (TN * 32), 0, lane);
    }
    int st_cur = 0;                                // kt % 3
    for (int kt = 0; kt < nkt; ++kt) {
      const int st_nxt = (st_cur == 2) ? 0 : st_cur + 1;
      const int st_nn = (st_nxt == 2) ? 0 : st_nxt + 1;
      const float* sA = smem + st_cur * RING;
      const float* sB = sA + BM * BK;
      const float* nA = smem + st_nxt * RING;
      const bool more = (kt + 1 < nkt), more2 = (kt + 2 < nkt);
      if (more2 && !(ABLATE & 8)) {   // ABLATE: timing-only diagnostic builds (wrong results)
        // (measured and not kept: spreading these DMAs between the MFMA groups, 72.1 vs 72.3 us;
        //  a hand-interleaved group with one asm statement per MFMA and the next group's LDS
        //  reads in between, 74.4 / 74.7 / 74.6 vs 72.2 / 74.1 / 71.3 us)
        const int k0 = kz0 + (kt + 2) * BK;
        planA.issue(lds0 + st_nn * (RING * 4), k0, k_end);
        planB.issue(lds0 + st_nn * (RING * 4) + BM * BK * 4, k0, k_end);
      }
#pragma unroll
      for (int s = 0; s < BK / 8; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        if (s < BK / 8 - 1) {
          read_frags_dma<LA, BM, TM>(fa[nxt], sA, wm * (TM * 32), s + 1, lane);
          read_frags_dma<LB, BN, TN>(fb[nxt], sB, wn * (TN * 32), s + 1, lane);
        } else if (more) {
          read_frags_dma<LA, BM, TM>(fa[nxt], nA, wm * (TM * 32), 0, lane);
          read_frags_dma<LB, BN, TN>(fb[nxt], nA + BM * BK, wn * (TN * 32), 0, lane);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
              acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i][j], fb[cur][jn][j],
                                                                acc[i][jn], 0, 0, 0);
        if (s == BK / 8 - 2 && more && !(ABLATE & 16)) {
          if (more2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G) : "memory");
          else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          asm volatile("" ::: "memory");
        }
      }
      st_cur = st_nxt;
    }
    __syncthreads();
  } else {
    // Software-pipelined main loop.  The fragments of k-group s+1 are read from LDS while the
    // MFMAs of group s execute (two register sets, static indices), so a wave never waits on
    // LDS between MFMA groups; the next K tile goes global -> VGPR at the top of the tile,
    // VGPR -> other LDS stage after group 1, and the single barrier sits after group 2, so
    // that group 3 already prefetches group 0 of the next stage.
    float fa[2][TM][4], fb[2][TN][4];
    if (nkt > 0) {
      read_frags<LA, BM, TM>(fa[0], smem, wm * (TM * 32), 0, lane);
      read_frags<LB, BN, TN>(fb[0], smem + A_FLOATS, wn * (TN * 32), 0, lane);
    }
    for (int kt = 0; kt < nkt; ++kt) {
      const float* sA = smem + (kt & 1) * STAGE;
      const float* sB = sA + A_FLOATS;
      float* nA = smem + ((kt + 1) & 1) * STAGE;
      const bool more = (kt + 1 < nkt);
      if (more && !(ABLATE & 1)) {   // ABLATE: timing-only diagnostic builds (wrong results)
        const int k0 = kz0 + (kt + 1) * BK;
        IOA::load(ra, p.A, p.lda, m0, p.M, k0, k_end, tid);
        IOB::load(rb, p.B, p.ldb, n0, p.N, k0, k_end, tid);
      }
#pragma unroll
      for (int s = 0; s < BK / 8; ++s) {
        const int cur = s & 1, nxt = cur ^ 1;
        if (s < BK / 8 - 1) {
          read_frags<LA, BM, TM>(fa[nxt], sA, wm * (TM * 32), s + 1, lane);
          read_frags<LB, BN, TN>(fb[nxt], sB, wn * (TN * 32), s + 1, lane);
        } else if (more) {
          read_frags<LA, BM, TM>(fa[nxt], nA, wm * (TM * 32), 0, lane);
          read_frags<LB, BN, TN>(fb[nxt], nA + A_FLOATS, wn * (TN * 32), 0, lane);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
              acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][i][j], fb[cur][jn][j],
                                                                acc[i][jn], 0, 0, 0);
        if (s == 1 && more && !(ABLATE & 2)) {
          IOA::store(ra, nA, tid);
          IOB::store(rb, nA + A_FLOATS, tid);
        }
        if (s == BK / 8 - 2 && !(ABLATE & 4)) __syncthreads();
      }
    }
    __syncthreads();
  }

  if (STAMP) {
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0) {
      unsigned long long* o = reinterpret_cast<unsigned long long*>(p.loss_part) + 8 * (blockIdx.x + gridDim.x * blockIdx.z);
      o[0] = c1 - stamp_c0; o[1] = r1 - stamp_r0; o[2] = stamp_entry; o[3] = stamp_r0; o[4] = r1;
    }
  }
  // ------------------------------------------------------------- epilogue --
  gemm_epilogue<BM, BN, WM, WN, EPI>(acc, p, C, smem, m0, n0, tile_m, true);
  if (STAMP) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long r2 = __builtin_amdgcn_s_memrealtime();
    if (tid == 0)
      (reinterpret_cast<unsigned long long*>(p.loss_part) + 8 * (blockIdx.x + gridDim.x * blockIdx.z))[5] = r2;
  }
}

}  // namespace blh
