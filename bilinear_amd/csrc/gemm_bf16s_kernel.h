// bf16-storage GEMM for gfx950 ("bf16s", gemm_dtype = 4: BASELINE configs 3-5).
//
//   C[M,N] = A (M x K) * B (K x N): both operands are bf16 IN MEMORY (activations, gradients and
//   the bf16 shadow of the fp32 master weights), multiplied on v_mfma_f32_32x32x16_bf16 with fp32
//   accumulation (dense peak ~2.5 PFLOP/s), C written as bf16 (activations / gradients) or fp32
//   (weight-gradient slabs).  Same three contractions as gemm_f32_ring.h:
//     forward  Z  = A  W^T     A [B,K] ROWK,  W  [N,K]  ROWK    (+bias, BatchNorm partials)
//     dgrad    dA = dZ W       dZ [B,N'] ROWK, W [N',K'] KROW   (+ block-skip gradient)
//     wgrad    dW = dZ^T A     dZ [B,N'] KROW, A [B,K'] KROW    (fp32 slabs, split over B)
//
// Data path: no conversion and no register staging anywhere.
//   * HBM/L2 -> LDS by LDS-DMA (global_load_lds_dwordx4, scalar base + per-lane 32-bit offset,
//     gemm_f32_ring.h), STAGES-deep ring of BKE-element K tiles, one raw s_barrier per K tile
//     behind a counted vmcnt.
//   * ROWK operand: LDS image [row][k] (k contiguous, BKE * 2 bytes per row); a lane's MFMA
//     operand (8 consecutive k of its row) is ONE ds_read_b128; 16-B chunks are XOR-swizzled on
//     the DMA source and on the read with the key that makes every ds_read_b128 lane group hit
//     16 different slots (gemm_f32_ring.h: ring_swz).
//   * KROW operand (reduction index is the memory ROW): the DMA lands the tile as it lies in
//     memory, [k][m] with 256-B k rows, and the MFMA operand (8 consecutive k of one m) comes from
//     two ds_read_b64_tr_b16 — the hardware transposing read: per 16-lane group a block of
//     4 k-rows x 16 m-columns is delivered column-major, lane 4q+p supplying the address of row q,
//     columns 4p..4p+3, lane i receiving column i of the 4 rows.  16-B chunks of a k row are
//     XOR-swizzled with ((k & 3) << 2) | ((k >> 2) & 3), the key that keeps both this read and a
//     row read conflict-free on 256-B rows.
// Workgroup = 256 threads = 4 waves (2 x 2), tile 128 x 128, each wave 64 x 64 = 2 x 2 MFMA tiles:
// 4 fragment reads per 4 MFMAs per k-step (one ds_read_b128 per MFMA keeps the LDS below its
// 256 B/clk; three per two MFMAs, as an 8-wave tiling would need, saturates it).
#pragma once
#include "common.h"
#include "gemm_epilogue.h"
#include "gemm_f32_ring.h"

namespace blh {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef short s16x4_t __attribute__((ext_vector_type(4)));
typedef uint16_t bf16_bits;
typedef uint16_t bf16_bits_t;

#ifndef BLH_H_BKE_DEFAULT
#define BLH_H_BKE_DEFAULT 64
#endif
#ifndef BLH_H_STAGES_DEFAULT
#define BLH_H_STAGES_DEFAULT 2
#endif

struct GemmParamsH {
  const bf16_bits* A;
  const bf16_bits* B;
  void* C;                   // bf16 or fp32 (template OUT_BF16)
  int64_t lda, ldb, ldc;     // elements
  int M, N, K;
  int k_per_split;
  int64_t c_split_stride;    // elements of C between slabs
  const float* bias;         // [N] fp32
  const bf16_bits* addend;   // [M][ldadd] bf16 (EPI_ADD)
  int64_t ldadd;
  float* stat_part;          // [tiles_m][2][N]
  // batched launch (gemm_bf16s_256_kernel only): grid z = batch item * batch_splits + slab; item b reads
  // A + b * a_batch_stride, B + b * b_batch_stride and writes C + b * c_batch_stride (elements).
  // batch_splits == 0: an ordinary launch (grid z = slab).
  int batch_splits;
  int64_t a_batch_stride, b_batch_stride, c_batch_stride;
  // EPI_BN_BWD (big-tile kernels): the stage whose OUTPUT gradient this GEMM produces — its pre-BatchNorm
  // tensor Z [M][ldz] (bf16), dropout keep bits ([ceil(M/4)][N/8] words, bn_bf16.hip), scale / shift [N];
  // stat_part receives [tiles_m][2][N]: sum over the tile's rows of dY' z and of dY'
  const bf16_bits* bn_z;
  int64_t ldz;
  const uint32_t* bn_keep;
  const float* bn_scale;
  const float* bn_shift;
};

__device__ __forceinline__ float bf16_to_f32(bf16_bits v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ bf16_bits f32_to_bf16(float x) {
  const __bf16 b = (__bf16)x;      // v_cvt_pk_bf16_f32: round to nearest even, NaN stays NaN
  return *reinterpret_cast<const bf16_bits*>(&b);
}

template <int BKE>
__device__ __forceinline__ int rowk_swz_h(int row) {     // 16-B chunk key of a ROWK image
  return (BKE * 2 == 128) ? ((row >> 1) & 7) : (row & 15);
}
__device__ __forceinline__ int krow_swz_h(int k) { return ((k & 3) << 2) | ((k >> 2) & 3); }

// LDS-DMA plan of one operand: like RingPlan (gemm_f32_ring.h) for 2-byte elements
template <int LAYOUT, int R, int NT, int BKE>
struct RingPlanH {
  static constexpr int CHUNKS = (R * BKE / 8) / NT;   // 16-B chunks per thread per tile
  static_assert((R * BKE / 8) % NT == 0, "tile not divisible among threads");
  static_assert(LAYOUT == ROWK || R == 128, "KROW image has 256-byte k rows");
  uint32_t voff[CHUNKS];
  int koff[CHUNKS];
  const bf16_bits* sbase;
  int64_t tile_step;       // elements per K tile
  uint32_t wave_off;
  bool ragged_k;

  __device__ inline void init(const bf16_bits* __restrict__ base, int64_t ld, int row0,
                              int rows_limit, int k_first, int k_end, int tid) {
    ragged_k = ((k_end - k_first) % BKE) != 0;
    wave_off = __builtin_amdgcn_readfirstlane((uint32_t)(tid & ~63) * 16u);
    const int last = rows_limit - 1 - row0;
    if (LAYOUT == ROWK) {
      sbase = base + (int64_t)row0 * ld + k_first;
      tile_step = BKE;
    } else {
      sbase = base + (int64_t)k_first * ld + row0;
      tile_step = (int64_t)BKE * ld;
    }
#pragma unroll
    for (int p = 0; p < CHUNKS; ++p) {
      const int q = tid + p * NT;
      if (LAYOUT == ROWK) {
        constexpr int CPR = BKE / 8;
        const int r = q / CPR, c = q % CPR;
        const int kk = (c ^ rowk_swz_h<BKE>(r)) << 3;
        koff[p] = kk;
        voff[p] = (uint32_t)(((int64_t)min(r, last) * ld + kk) * 2);
      } else {
        constexpr int CPR = R / 8;              // 16 chunks per k row
        const int k = q / CPR, slot = q % CPR;
        const int m8 = (slot ^ krow_swz_h(k)) << 3;
        koff[p] = k;
        voff[p] = (uint32_t)(((int64_t)k * ld + min(m8, last - 7)) * 2);   // rows_limit % 8 == 0
      }
    }
  }

  __device__ inline void issue(uint32_t lds_tile, int k0, int k_end) {
    if (ragged_k && k0 + BKE > k_end) {
#pragma unroll
      for (int p = 0; p < CHUNKS; ++p) {
        const float* g = (k0 + koff[p] < k_end)
                             ? reinterpret_cast<const float*>(reinterpret_cast<const char*>(sbase) + voff[p])
                             : reinterpret_cast<const float*>(&g_zero16);
        lds_dma16_asm(g, lds_tile + wave_off + (uint32_t)(p * NT * 16));
      }
    } else {
      lds_dma16_sbase<true>(voff[0], reinterpret_cast<const float*>(sbase), lds_tile + wave_off);
#pragma unroll
      for (int p = 1; p < CHUNKS; ++p)
        lds_dma16_sbase<false>(voff[p], reinterpret_cast<const float*>(sbase),
                               lds_tile + wave_off + (uint32_t)(p * NT * 16));
    }
    sbase += tile_step;
  }
};

// MFMA operands of k-step kk (16 consecutive k) for T 32-row sub-tiles starting at row_base
template <int LAYOUT, int R, int T, int BKE>
__device__ inline void read_frags_h(bf16x8_t (&frag)[T], const bf16_bits* lds, int row_base, int kk,
                                    int lane) {
  if (LAYOUT == ROWK) {
    const int h = lane >> 5, lr = lane & 31;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int row = row_base + t * 32 + lr;
      frag[t] = *reinterpret_cast<const bf16x8_t*>(
          lds + row * BKE + (((2 * kk + h) ^ rowk_swz_h<BKE>(row)) << 3));
    }
  } else {
    const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const int mcol = row_base + t * 32 + 16 * (g & 1) + 4 * pp;
      s16x4_t v[2];
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int krow = 16 * kk + 8 * (g >> 1) + 4 * r + q;
        const bf16_bits* addr = lds + krow * R + (((mcol >> 3) ^ krow_swz_h(krow)) << 3) + (mcol & 7);
        v[r] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
            (__attribute__((address_space(3))) s16x4_t*)(addr));
      }
      union { s16x4_t s[2]; bf16x8_t b; } u;
      u.s[0] = v[0]; u.s[1] = v[1];
      frag[t] = u.b;
    }
  }
}

// host-side entry points (gemm_bf16s.hip)
int launch_gemm_bf16s(hipStream_t s, int la, int lb, int epi, bool out_bf16, const GemmParamsH& p,
                      int splits);
// the tile (kernel) launch_gemm_bf16s uses for this contraction (gemm_bf16s.hip)
enum HTile : int { H_TILE_128 = 128, H_TILE_256 = 256, H_TILE_128x256 = 384 };
int gemm_bf16s_pick_tile(int la, int lb, bool out_bf16, const GemmParamsH& p, int splits);
int gemm_bf16s_tile_rows(int tile);   // rows of one tile = rows per BatchNorm-partials record
int gemm_bf16s_tile_cols(int tile);
void gemm_bf16s_force_tile(int tile);  // 0: automatic; -1: re-read BLH_BF16S_TILE
int launch_cast_f32_bf16(hipStream_t s, const float* src, uint16_t* dst, int64_t n);
// out[m][n] (+)= sum_{r < rows} dz[r][m] * act[r][n]  (rows <= 7: the rows a ragged batch leaves behind the 8-row
// groups of the weight-gradient GEMM); accumulate: add to what `out` holds, else overwrite
int launch_wgrad_tail_h(hipStream_t s, const uint16_t* dz, int64_t ld_dz, const uint16_t* act, int64_t ld_act, int rows,
                        int M, int N, float* out, bool accumulate);
int launch_cast2_f32_bf16(hipStream_t s, const float* src0, uint16_t* dst0, int64_t n0, const float* src1,
                          uint16_t* dst1, int64_t n1, const float* wd_src = nullptr, uint16_t* wdT = nullptr,
                          int wd_W = 0, int wd_OF = 0);   // two tensors, one launch
int launch_cast_bf16_f32(hipStream_t s, const uint16_t* src, float* dst, int64_t n);

template <int BKE, int STAGES>
constexpr size_t gemm_bf16s_lds_bytes() { return (size_t)STAGES * (128 + 128) * BKE * 2; }

// ---- epilogue: fp32 accumulators -> bf16 / fp32 C -------------------------------------------
// OUT_BF16: consecutive-column pairs are exchanged between neighbouring lanes so that a lane
// stores 4 bytes (even lanes row r, odd lanes row r + 1).
template <int EPI, bool OUT_BF16>
__device__ inline void gemm_epilogue_h(f32x16 (&acc)[2][2], const GemmParamsH& p, void* Cv, float* smem,
                                       int m0, int n0, int tile_m) {
  constexpr int BN = 128, WM = 2, TM = 2, TN = 2;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int h = lane >> 5, lc = lane & 31;
  const int row_w = m0 + wm * 64 + 4 * h;
  const int col_w = n0 + wn * 64 + lc;

  if (EPI == EPI_BIAS || EPI == EPI_BIAS_STATS) {
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      const int col = col_w + jn * 32;
      const float bv = (col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][jn][r] += bv;
    }
  }
  static_assert(EPI != EPI_ADD || OUT_BF16, "the skip-gradient epilogue writes bf16");
  // ---- C stores.  Measured (tools/bf16s_bench, M = 16384, W = 1024): writing the tile straight
  // from the MFMA register layout — a lane owns 4 consecutive ROWS of one column, so even with
  // neighbouring lanes exchanging values a store instruction covers four 64-byte row segments —
  // cost 20 us of a 59 us forward GEMM (the same kernel without its C stores: 38 us).  The tile
  // therefore goes through LDS (free after the main loop): fp32 values, 64 rows at a time (the
  // two wave rows in turn), row pitch 136 floats (conflict-free for the 32-lane row writes of both
  // lane halves); then every thread reads 8 (bf16 out) or 4 (fp32 out) consecutive columns of one
  // row and issues ONE 16-byte store: a wave writes four full 256-byte (512-byte) rows per
  // instruction.  The skip-gradient addend (EPI_ADD) is loaded with the same 16-byte pattern and
  // added in fp32 before the single rounding to bf16; it may alias C (in place): a thread reads
  // exactly the 16 bytes it then writes.
  constexpr int SP = 136;                         // staging row pitch in floats
  constexpr int VEC = OUT_BF16 ? 8 : 4;           // columns per 16-byte store
  const bool vec_ok = (p.N % VEC == 0) && (p.ldc % VEC == 0) && ((reinterpret_cast<uintptr_t>(Cv) & 15) == 0) &&
                      (EPI != EPI_ADD || ((p.ldadd % 8 == 0) && ((reinterpret_cast<uintptr_t>(p.addend) & 15) == 0)));
  if (vec_ok) {
    float* stg = smem;                            // [64][SP] floats = 34 KB
    // (EPI_ADD: the thread's 8 addend pieces are requested here, once, in front of the staging passes)
    constexpr bool PREADD = (EPI == EPI_ADD) && OUT_BF16;
    uint4 adv[PREADD ? 2 : 1][PREADD ? 4 : 1];
    if constexpr (PREADD) {
#pragma unroll
      for (int half = 0; half < 2; ++half)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
          const int q = it * 256 + tid;
          const int row = m0 + half * 64 + q / (BN / VEC), col = n0 + (q % (BN / VEC)) * VEC;
          adv[half][it] = (row < p.M && col < p.N)
                              ? *reinterpret_cast<const uint4*>(p.addend + (int64_t)row * p.ldadd + col)
                              : uint4{0u, 0u, 0u, 0u};
        }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if (wm == half) {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn)
#pragma unroll
            for (int r = 0; r < 16; ++r)
              stg[(i * 32 + 8 * (r >> 2) + 4 * h + (r & 3)) * SP + wn * 64 + jn * 32 + lc] = acc[i][jn][r];
      }
      __syncthreads();
      constexpr int CPR = BN / VEC;               // 16-byte chunks per tile row
#pragma unroll
      for (int q0 = 0; q0 < 64 * CPR; q0 += 256) {
        const int q = q0 + tid;
        const int lrow = q / CPR, cv = (q % CPR) * VEC;
        const int row = m0 + half * 64 + lrow, col = n0 + cv;
        if (row < p.M && col < p.N) {
          const float4 v0 = *reinterpret_cast<const float4*>(stg + lrow * SP + cv);
          if (OUT_BF16) {
            const float4 v1 = *reinterpret_cast<const float4*>(stg + lrow * SP + cv + 4);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            if constexpr (EPI == EPI_ADD) {
              const uint4 ad = adv[half][q0 / 256];
              v[0] += __uint_as_float(ad.x << 16); v[1] += __uint_as_float(ad.x & 0xffff0000u);
              v[2] += __uint_as_float(ad.y << 16); v[3] += __uint_as_float(ad.y & 0xffff0000u);
              v[4] += __uint_as_float(ad.z << 16); v[5] += __uint_as_float(ad.z & 0xffff0000u);
              v[6] += __uint_as_float(ad.w << 16); v[7] += __uint_as_float(ad.w & 0xffff0000u);
            }
            uint4 o;
            o.x = (uint32_t)f32_to_bf16(v[0]) | ((uint32_t)f32_to_bf16(v[1]) << 16);
            o.y = (uint32_t)f32_to_bf16(v[2]) | ((uint32_t)f32_to_bf16(v[3]) << 16);
            o.z = (uint32_t)f32_to_bf16(v[4]) | ((uint32_t)f32_to_bf16(v[5]) << 16);
            o.w = (uint32_t)f32_to_bf16(v[6]) | ((uint32_t)f32_to_bf16(v[7]) << 16);
            *reinterpret_cast<uint4*>(reinterpret_cast<bf16_bits*>(Cv) + (int64_t)row * p.ldc + col) = o;
          } else {
            *reinterpret_cast<float4*>(reinterpret_cast<float*>(Cv) + (int64_t)row * p.ldc + col) = v0;
          }
        }
      }
      __syncthreads();
    }
  } else if (OUT_BF16) {
    // (shapes without 16-byte rows: element pairs, neighbouring lanes exchanging values)
    bf16_bits* __restrict__ C = reinterpret_cast<bf16_bits*>(Cv);
    const bool odd = lane & 1;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jn = 0; jn < TN; ++jn)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          const float give = odd ? acc[i][jn][r] : acc[i][jn][r + 1];
          const float got = __shfl_xor(give, 1);
          float lo = odd ? got : acc[i][jn][r];
          float hi = odd ? acc[i][jn][r + 1] : got;
          const int row = row_w + i * 32 + ((r + (odd ? 1 : 0)) & 3) + 8 * (r >> 2);
          const int col = col_w + jn * 32 - (odd ? 1 : 0);     // even column of the pair
          if (row < p.M && col < p.N) {                        // N is even: the pair is in range
            if (EPI == EPI_ADD) {
              const uint32_t ad = *reinterpret_cast<const uint32_t*>(p.addend + (int64_t)row * p.ldadd + col);
              lo += __uint_as_float(ad << 16);
              hi += __uint_as_float(ad & 0xffff0000u);
            }
            const uint32_t packed = (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
            *reinterpret_cast<uint32_t*>(C + (int64_t)row * p.ldc + col) = packed;
          }
        }
  } else {
    float* __restrict__ C = reinterpret_cast<float*>(Cv);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row_w + i * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
          const int col = col_w + jn * 32;
          if (row < p.M && col < p.N) C[(int64_t)row * p.ldc + col] = acc[i][jn][r];
        }
      }
  }

  if (EPI == EPI_BIAS_STATS) {
    // per-tile column (mean, M2) of the fp32 values (before rounding to bf16), as gemm_epilogue.h
    float* red = smem;   // [WM][BN]
    const int cnt = min(128, p.M - m0);
    float mean[TN];
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row_w + i * 32 + (r & 3) + 8 * (r >> 2);
          if (row < p.M) s += acc[i][jn][r];
        }
      s += __shfl_xor(s, 32);
      if (h == 0) red[wm * BN + wn * 64 + jn * 32 + lc] = s;
    }
    lds_barrier();
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) t += red[w * BN + wn * 64 + jn * 32 + lc];
      mean[jn] = t / (float)cnt;
    }
    lds_barrier();
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row_w + i * 32 + (r & 3) + 8 * (r >> 2);
          const float dlt = acc[i][jn][r] - mean[jn];
          if (row < p.M) s += dlt * dlt;
        }
      s += __shfl_xor(s, 32);
      if (h == 0) red[wm * BN + wn * 64 + jn * 32 + lc] = s;
    }
    lds_barrier();
    if (wm == 0 && h == 0) {
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) {
        const int col = col_w + jn * 32;
        float m2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) m2 += red[w * BN + wn * 64 + jn * 32 + lc];
        if (col < p.N) {
          p.stat_part[((int64_t)tile_m * 2 + 0) * p.N + col] = mean[jn];
          p.stat_part[((int64_t)tile_m * 2 + 1) * p.N + col] = m2;
        }
      }
    }
  }
}

template <int LA, int LB, int EPI, bool OUT_BF16, int BKE = 64, int STAGES = 2>
__global__ __launch_bounds__(256) void gemm_bf16s_kernel(GemmParamsH p) {
  constexpr int BM = 128, BN = 128, NT = 256, TM = 2, TN = 2;
  static_assert(STAGES >= 2 && STAGES <= 4, "ring depth");
  constexpr int RING = (BM + BN) * BKE;          // elements per stage
  constexpr int NG = BKE / 16;                   // k-steps per tile
  static_assert(NG >= 2, "K tile of at least two k-steps");
  using PA = RingPlanH<LA, BM, NT, BKE>;
  using PB = RingPlanH<LB, BN, NT, BKE>;
  constexpr int G = PA::CHUNKS + PB::CHUNKS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  bf16_bits* lds = reinterpret_cast<bf16_bits*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_n = (p.N + BN - 1) / BN;
  int tile = 0, slab = blockIdx.z;   // (split grids: slab-major XCD mapping, gemm_dma.h: xcd_remap_split)
  if (!(gridDim.z > 1 && xcd_remap_split(blockIdx.x, blockIdx.z, gridDim.x, gridDim.z, (p.M + BM - 1) / BM,
                                         tiles_n, &tile, &slab)))
    tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kz0 = slab * p.k_per_split;
  const int k_end = min(p.K, kz0 + p.k_per_split);
  void* C = OUT_BF16 ? (void*)(reinterpret_cast<bf16_bits*>(p.C) + (int64_t)slab * p.c_split_stride)
                     : (void*)(reinterpret_cast<float*>(p.C) + (int64_t)slab * p.c_split_stride);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nkt = (k_end - kz0 + BKE - 1) / BKE;
  PA planA;
  PB planB;
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) float*)smem);
  planA.init(p.A, p.lda, m0, p.M, kz0, k_end, tid);
  planB.init(p.B, p.ldb, n0, p.N, kz0, k_end, tid);
  bf16x8_t fa[2][TM], fb[2][TN];

  if (nkt > 0) {
#pragma unroll
    for (int t = 0; t < STAGES - 1; ++t)
      if (t < nkt) {
        planA.issue(lds0 + t * (RING * 2), kz0 + t * BKE, k_end);
        planB.issue(lds0 + t * (RING * 2) + BM * BKE * 2, kz0 + t * BKE, k_end);
      }
    const int ahead = min(nkt, STAGES - 1) - 1;
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    read_frags_h<LA, BM, TM, BKE>(fa[0], lds, wm * 64, 0, lane);
    read_frags_h<LB, BN, TN, BKE>(fb[0], lds + BM * BKE, wn * 64, 0, lane);
  }

#define BLH_H_MFMAS(CUR)                                                                        \
  _Pragma("unroll") for (int i = 0; i < TM; ++i)                                                \
  _Pragma("unroll") for (int jn = 0; jn < TN; ++jn)                                             \
    acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[CUR][i], fb[CUR][jn], acc[i][jn], 0, 0, 0);

  int st_cur = 0;
  for (int kt = 0; kt < nkt - 1; ++kt) {
    const int st_nxt = (st_cur == STAGES - 1) ? 0 : st_cur + 1;
    const int st_new = (st_cur == 0) ? STAGES - 1 : st_cur - 1;
    const bf16_bits* sA = lds + st_cur * RING;
    const bf16_bits* sB = sA + BM * BKE;
    const bf16_bits* nA = lds + st_nxt * RING;
    __builtin_amdgcn_s_waitcnt(0xC07F);       // lgkmcnt(0): free here, keeps hipcc's counts exact
    if (kt + STAGES - 1 < nkt) {
      const int k0 = kz0 + (kt + STAGES - 1) * BKE;
      planA.issue(lds0 + st_new * (RING * 2), k0, k_end);
      planB.issue(lds0 + st_new * (RING * 2) + BM * BKE * 2, k0, k_end);
    }
#pragma unroll
    for (int s = 0; s < NG; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      if (s < NG - 1) {
        read_frags_h<LA, BM, TM, BKE>(fa[nxt], sA, wm * 64, s + 1, lane);
        read_frags_h<LB, BN, TN, BKE>(fb[nxt], sB, wn * 64, s + 1, lane);
      } else {
        read_frags_h<LA, BM, TM, BKE>(fa[nxt], nA, wm * 64, 0, lane);
        read_frags_h<LB, BN, TN, BKE>(fb[nxt], nA + BM * BKE, wn * 64, 0, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
      BLH_H_MFMAS(cur)
      __builtin_amdgcn_sched_barrier(0);
      if (s == NG - 2) {
        const int ahead = min(nkt - 1, kt + STAGES - 1) - (kt + 1);
        if (STAGES >= 4 && ahead >= 2) ring_wait_vm_lgkm<2 * G>();
        else if (STAGES >= 3 && ahead >= 1) ring_wait_vm_lgkm<G>();
        else ring_wait_vm_lgkm<0>();
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
      }
    }
    st_cur = st_nxt;
  }
  if (nkt > 0) {
    const bf16_bits* sA = lds + st_cur * RING;
    const bf16_bits* sB = sA + BM * BKE;
    __builtin_amdgcn_s_waitcnt(0xC07F);
#pragma unroll
    for (int s = 0; s < NG; ++s) {
      const int cur = s & 1, nxt = cur ^ 1;
      if (s < NG - 1) {
        read_frags_h<LA, BM, TM, BKE>(fa[nxt], sA, wm * 64, s + 1, lane);
        read_frags_h<LB, BN, TN, BKE>(fb[nxt], sB, wn * 64, s + 1, lane);
      }
      __builtin_amdgcn_sched_barrier(0);
      BLH_H_MFMAS(cur)
      __builtin_amdgcn_sched_barrier(0);
    }
  }
#undef BLH_H_MFMAS
  __syncthreads();
  gemm_epilogue_h<EPI, OUT_BF16>(acc, p, C, smem, m0, n0, tile_m);
}

}  // namespace blh
