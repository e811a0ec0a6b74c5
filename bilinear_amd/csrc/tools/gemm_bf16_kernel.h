// bf16-MFMA GEMM for gfx950 with fp32 storage ("mixed" mode, gemm_dtype = 1):
//   C[M,N] (fp32) = A * B, operands read as fp32 from HBM/L2, rounded to bf16 (RNE,
//   v_cvt_pk_bf16_f32) on their way into LDS, multiplied on v_mfma_f32_32x32x16_bf16 with fp32
//   accumulation (dense peak ~2.5 PFLOP/s, 16x the fp32 MFMA rate).
// Same contractions, operand layouts (ROWK / KROW) and epilogues as gemm_f32_kernel.h; with
// fp32 operands in memory the kernel is bound by the L2 -> LDS traffic, not by the MFMA.
//
// LDS image: both operands as [row][k] bf16 with k contiguous (K tile = 128, row pitch 136 bf16
// = 272 B so that the 16-B fragment reads of 16 consecutive rows hit 16 different slots).
//   ROWK operand: a lane loads 4 consecutive k (float4), writes 4 bf16 (ds_write_b64).
//   KROW operand: a lane loads a 8(k) x 4(rows) patch (8 x float4, 512 B contiguous per k row
//   across 32 lanes), transposes it in registers and writes 4 x ds_write_b128 (8 k of one row).
// Fragment: lane (r = lane&31, h = lane>>5) of k-step kk reads the 8 bf16 at
// [row r][16 kk + 8 h .. +7] — exactly the A / B operand of the 32x32x16 MFMA.
#pragma once
#include "../common.h"
#include "../gemm_epilogue.h"
#include "../gemm_dma.h"          // g_zero16, lds_dma16_asm, xcd_remap

namespace blh {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4_t __attribute__((ext_vector_type(4)));

#ifndef BLH_BKH
#define BLH_BKH 128
#endif
static constexpr int BKH = BLH_BKH;       // K tile in elements (tools may override)
static constexpr int PITCHH = BKH + 8;    // bf16 per LDS row

template <int R>
constexpr int bf16_tile_elems() { return R * PITCHH; }

template <int BM, int BN>
constexpr size_t gemm_bf16_lds_bytes() {
  return 2 * (size_t)(bf16_tile_elems<BM>() + bf16_tile_elems<BN>()) * sizeof(__bf16);
}

// ---- global fp32 -> registers -> bf16 LDS -----------------------------------------------
template <int LAYOUT, int R, int NT>
struct TileCvt {
  // ROWK: R*16 float4 chunks; KROW: (R/4)*8 patches of 8 float4
  static constexpr int ITEMS = (LAYOUT == ROWK) ? (R * BKH / 4) : (R / 4) * (BKH / 8);
  static constexpr int PER = (ITEMS + NT - 1) / NT;
  static constexpr int REGS = (LAYOUT == ROWK) ? PER : PER * 8;

  __device__ static inline void load(float4 (&reg)[REGS], const float* __restrict__ base,
                                     int64_t ld, int row0, int rows_limit, int k0, int k_end,
                                     int tid) {
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int q = tid + p * NT;
      if (LAYOUT == ROWK) {
        constexpr int CPR = BKH / 4;
        const int row = row0 + (q / CPR), k = k0 + ((q % CPR) << 2);
        const bool ok = (ITEMS % NT == 0 || q < ITEMS) && row < rows_limit && k < k_end;
        reg[p] = ok ? *reinterpret_cast<const float4*>(base + (int64_t)row * ld + k)
                    : make_float4(0.f, 0.f, 0.f, 0.f);
      } else {
        constexpr int MG = R / 4;
        const int row = row0 + ((q % MG) << 2), kb = k0 + ((q / MG) << 3);
        const bool okr = (ITEMS % NT == 0 || q < ITEMS) && row < rows_limit;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const bool ok = okr && (kb + j < k_end);
          reg[p * 8 + j] = ok ? *reinterpret_cast<const float4*>(base + (int64_t)(kb + j) * ld + row)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
  }

  __device__ static inline void store(const float4 (&reg)[REGS], __bf16* lds, int tid) {
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int q = tid + p * NT;
      if (ITEMS % NT != 0 && q >= ITEMS) continue;
      if (LAYOUT == ROWK) {
        bf16x4_t v;
        v[0] = (__bf16)reg[p].x; v[1] = (__bf16)reg[p].y; v[2] = (__bf16)reg[p].z; v[3] = (__bf16)reg[p].w;
        constexpr int CPR = BKH / 4;
        *reinterpret_cast<bf16x4_t*>(lds + (q / CPR) * PITCHH + ((q % CPR) << 2)) = v;
      } else {
        constexpr int MG = R / 4;
        const int r = (q % MG) << 2, kb = (q / MG) << 3;
        bf16x8_t v0, v1, v2, v3;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          v0[j] = (__bf16)reg[p * 8 + j].x; v1[j] = (__bf16)reg[p * 8 + j].y;
          v2[j] = (__bf16)reg[p * 8 + j].z; v3[j] = (__bf16)reg[p * 8 + j].w;
        }
        *reinterpret_cast<bf16x8_t*>(lds + (r + 0) * PITCHH + kb) = v0;
        *reinterpret_cast<bf16x8_t*>(lds + (r + 1) * PITCHH + kb) = v1;
        *reinterpret_cast<bf16x8_t*>(lds + (r + 2) * PITCHH + kb) = v2;
        *reinterpret_cast<bf16x8_t*>(lds + (r + 3) * PITCHH + kb) = v3;
      }
    }
  }
};

template <int BM, int BN, int WM, int WN, int LA, int LB, int EPI>
__global__ __launch_bounds__(64 * WM * WN) void gemm_bf16_kernel(GemmParams p) {
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  constexpr int A_EL = bf16_tile_elems<BM>(), B_EL = bf16_tile_elems<BN>(), STAGE = A_EL + B_EL;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* lds = reinterpret_cast<__bf16*>(smem);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kz0 = blockIdx.z * p.k_per_split;
  const int k_end = min(p.K, kz0 + p.k_per_split);
  float* __restrict__ C = p.C + (int64_t)blockIdx.z * p.c_split_stride;

  using IOA = TileCvt<LA, BM, NT>;
  using IOB = TileCvt<LB, BN, NT>;
  float4 ra[IOA::REGS], rb[IOB::REGS];

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nkt = (k_end - kz0 + BKH - 1) / BKH;
  if (nkt > 0) {
    IOA::load(ra, p.A, p.lda, m0, p.M, kz0, k_end, tid);
    IOB::load(rb, p.B, p.ldb, n0, p.N, kz0, k_end, tid);
    IOA::store(ra, lds, tid);
    IOB::store(rb, lds + A_EL, tid);
  }
  __syncthreads();

  const int h = lane >> 5, lr = lane & 31;
  for (int kt = 0; kt < nkt; ++kt) {
    const __bf16* sA = lds + (kt & 1) * STAGE;
    const __bf16* sB = sA + A_EL;
    const bool more = (kt + 1 < nkt);
    if (more) {
      const int k0 = kz0 + (kt + 1) * BKH;
      IOA::load(ra, p.A, p.lda, m0, p.M, k0, k_end, tid);
      IOB::load(rb, p.B, p.ldb, n0, p.N, k0, k_end, tid);
    }
#pragma unroll
    for (int kk = 0; kk < BKH / 16; ++kk) {
      bf16x8_t fa[TM], fb[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i)
        fa[i] = *reinterpret_cast<const bf16x8_t*>(
            sA + (wm * (TM * 32) + i * 32 + lr) * PITCHH + 16 * kk + 8 * h);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        fb[j] = *reinterpret_cast<const bf16x8_t*>(
            sB + (wn * (TN * 32) + j * 32 + lr) * PITCHH + 16 * kk + 8 * h);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      __bf16* nA = lds + ((kt + 1) & 1) * STAGE;
      IOA::store(ra, nA, tid);
      IOB::store(rb, nA + A_EL, tid);
    }
    __syncthreads();
  }

  gemm_epilogue<BM, BN, WM, WN, EPI>(acc, p, C, smem, m0, n0, tile_m, true);
}

}  // namespace blh
