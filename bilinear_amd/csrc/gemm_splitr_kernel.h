// bf16x3 split GEMM, "split on read" form (gemm_dtype = 2): fp32 accuracy on the bf16 matrix
// cores of gfx950.  See gemm_split_kernel.h for the arithmetic (x = h + m + l exactly, six bf16
// MFMAs per product, fp32 accumulate); this file differs in WHERE the split happens:
//
//   HBM/L2 --global_load_lds_dwordx4 (LDS-DMA, no VGPR staging, no ds_write)--> 3-stage ring of
//   fp32 tiles in LDS (the data path of gemm_f32_kernel.h PIPE 3: same DmaPlan, same XOR swizzle)
//   --ds_read_b128 / ds_read_b32--> fp32 fragments in registers --split (11 VALU per value pair,
//   in the shadow of the MFMAs)--> three bf16x8 operands per 32-row sub-tile --> 6 MFMAs.
//
// Measured motivation (tools/split_bench, B=4096, W=1024): splitting on the way INTO LDS costs
// 6 B of ds_write per value, and the VGPR->LDS store path (~80 B/clk/CU, in-order issue) stalls
// the MFMA stream for 12 cycles per MFMA slot; here LDS is written by the DMA engine only.  The
// price is that a value is split by every wave that reads it (2x for the 2 x 2 wave grid).
//
// Workgroup = 256 threads = 4 waves (2 x 2), tile 128 x 128, K tile 32 = two k-steps of 16.
// Phase n (one k-step): MFMAs on planes P[n&1]; split floats F[(n+1)&1] -> P[(n+1)&1]; read the
// floats of k-step n+2 into F[n&1].  While tile kt is multiplied, the LDS reads target tile kt+1
// and tiles kt+2, kt+3 are in flight; one counted vmcnt + s_barrier per K tile.
#pragma once
#include "common.h"
#include "gemm_bf16_kernel.h"   // bf16x8_t
#include "gemm_epilogue.h"
#include "gemm_f32_kernel.h"    // DmaPlan, lds_dma16_asm, xcd_remap, BK

namespace blh {

template <int BM, int BN>
constexpr size_t gemm_splitr_lds_bytes() {
  return 3 * (size_t)(BM + BN) * BK * sizeof(float);
}

typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));

// fp32 fragment source of one 32-row sub-tile for one k-step of 16: 8 floats per lane.
//   slot j < 4: k = 16 kk + 4 h + j ; slot j >= 4: k = 16 kk + 8 + 4 h + (j - 4)   (h = lane >> 5)
// Both operands use this map, so the order of k inside the k-step is immaterial.
// A "unit" is one LDS instruction: ROWK 2 units (ds_read_b128), KROW 8 units (ds_read_b32).
template <int LAYOUT>
struct FragSrc {
  static constexpr int UNITS = (LAYOUT == ROWK) ? 2 : 8;
  template <int R>
  __device__ static inline void read(float (&f)[8], const float* lds, int row, int kk, int h, int unit) {
    if (LAYOUT == ROWK) {
      const int s = 2 * kk + unit;
      const float4 v = *reinterpret_cast<const float4*>(lds + row * BK + (((2 * s + h) ^ (row & 7)) << 2));
      f[4 * unit + 0] = v.x; f[4 * unit + 1] = v.y; f[4 * unit + 2] = v.z; f[4 * unit + 3] = v.w;
    } else {
      const int s = 2 * kk + (unit >> 2), j = unit & 3;
      f[unit] = lds[(8 * s + 4 * h + j) * R + row];
    }
  }
};

template <int BM, int BN, int LA, int LB, int EPI>
__global__ __launch_bounds__(256) void gemm_splitr_kernel(GemmParams p) {
  constexpr int NT = 256, WN = 2, TM = 2, TN = 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kz0 = blockIdx.z * p.k_per_split;
  const int k_end = min(p.K, kz0 + p.k_per_split);
  float* __restrict__ C = p.C + (int64_t)blockIdx.z * p.c_split_stride;
  const int nkt = (k_end - kz0 + BK - 1) / BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  DmaPlan<LA, BM, NT> planA;
  DmaPlan<LB, BN, NT> planB;
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) float*)smem);
  planA.init(p.A, p.lda, m0, p.M, kz0, k_end, tid);
  planB.init(p.B, p.ldb, n0, p.N, kz0, k_end, tid);
  planA.ragged_k = true;   // requests past the last tile must read the zero page (branch-free loop)
  planB.ragged_k = true;
  constexpr int RING = (BM + BN) * BK;   // floats per stage
  constexpr int G = DmaPlan<LA, BM, NT>::CHUNKS + DmaPlan<LB, BN, NT>::CHUNKS;
  auto issue = [&](int stage, int kt) {
    const int k0 = kz0 + kt * BK;
    planA.issue(lds0 + stage * (RING * 4), k0, k_end);
    planB.issue(lds0 + stage * (RING * 4) + BM * BK * 4, k0, k_end);
  };

  const int h = lane >> 5, lr = lane & 31;
  const int rowa = wm * 64 + lr, rowb = wn * 64 + lr;

  struct Floats { float a[TM][8], b[TN][8]; };
  struct Planes { u32x4_t a[3][TM], b[3][TN]; };
  Floats F[2];
  Planes P[2];

  constexpr int UA = FragSrc<LA>::UNITS, UB = FragSrc<LB>::UNITS, UNITS = TM * UA + TN * UB;
  // LDS read unit `u` of a k-step (0 .. UNITS-1) into fl
  auto read_unit = [&](Floats& fl, const float* st, int kk, int u) {
    if (u < TM * UA)
      FragSrc<LA>::template read<BM>(fl.a[u / UA], st, rowa + 32 * (u / UA), kk, h, u % UA);
    else {
      const int v = u - TM * UA;
      FragSrc<LB>::template read<BN>(fl.b[v / UB], st + BM * BK, rowb + 32 * (v / UB), kk, h, v % UB);
    }
  };
  auto read_all = [&](Floats& fl, const float* st, int kk) {
#pragma unroll
    for (int u = 0; u < UNITS; ++u) read_unit(fl, st, kk, u);
  };

  // pair-split e (0..15) of a k-step: operand/sub-tile t = e / 4 (0,1: A; 2,3: B), pair q = e % 4
  // (floats 2q, 2q+1) -> element q of the three plane vectors.  Three stages (5, 5, 1 VALU).
  float sx[2], sy[2];   // residuals of the (up to two) pair-splits in flight
  uint32_t t0, t1;
  // ---- prologue ---------------------------------------------------------------------------
  issue(0, 0);
  issue(1, 1);
  issue(2, 2);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * G) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  read_all(F[0], smem, 0);
  read_all(F[1], smem, 1);

  auto split_stage = [&](Floats& fl, Planes& pl, int e, int stg, int lane_slot) {
    const int t = e >> 2, q = e & 3;
    float& x = sx[lane_slot];
    float& y = sy[lane_slot];
    if (stg == 0) {
      x = (t < 2) ? fl.a[t][2 * q] : fl.b[t - 2][2 * q];
      y = (t < 2) ? fl.a[t][2 * q + 1] : fl.b[t - 2][2 * q + 1];
    }
    uint32_t out;
    if (stg < 2) {
      asm volatile(
          "v_cvt_pk_bf16_f32 %0, %1, %2\n\t"
          "v_lshlrev_b32 %3, 16, %0\n\t"
          "v_and_b32 %4, 0xffff0000, %0\n\t"
          "v_sub_f32 %1, %1, %3\n\t"
          "v_sub_f32 %2, %2, %4"
          : "=&v"(out), "+v"(x), "+v"(y), "=&v"(t0), "=&v"(t1));
    } else {
      asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(out) : "v"(x), "v"(y));
    }
    if (t < 2) pl.a[stg][t][q] = out; else pl.b[stg][t - 2][q] = out;
  };

  // whole k-step split without MFMAs (prologue only)
#pragma unroll
  for (int e = 0; e < 16; ++e)
#pragma unroll
    for (int stg = 0; stg < 3; ++stg) split_stage(F[0], P[0], e, stg, 0);

  asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G) : "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  // One phase: 24 slots = {MFMA; two split stages of the next k-step}; the LDS reads of the
  // k-step after that are spread between the slots.  Split stage list: entry g = 0..47 is
  // (pair e = g / 3, stage g % 3); slot s takes entries 2s and 2s+1 — consecutive entries of one
  // pair chain through sx/sy[slot parity], so at most two pairs are in flight.
  auto phase = [&](const Planes& pc, Floats& fsplit, Planes& pn, Floats& fread, const float* rst, int kkr) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
#pragma unroll
    for (int s = 0; s < 24; ++s) {
      const int t = s >> 2, i = (s >> 1) & 1, j = s & 1;
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"
                   : "+a"(acc[i][j]) : "v"(pc.a[PA[t]][i]), "v"(pc.b[PB[t]][j]) : "memory");
#pragma unroll
      for (int g = 2 * s; g < 2 * s + 2; ++g) split_stage(fsplit, pn, g / 3, g % 3, (g / 3) & 1);
#pragma unroll
      for (int u = (UNITS * s) / 24; u < (UNITS * (s + 1)) / 24; ++u) read_unit(fread, rst, kkr, u);
    }
  };

  int st_rd = 1;   // ring stage of tile kt + 1 (the one the phases of tile kt read)
  for (int kt = 0; kt < nkt; ++kt) {
    const int st_free = (st_rd == 0) ? 2 : st_rd - 1;   // stage of tile kt: all its reads are done
    issue(st_free, kt + 3);
    const float* rst = smem + st_rd * RING;
    phase(P[0], F[1], P[1], F[0], rst, 0);
    phase(P[1], F[0], P[0], F[1], rst, 1);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(G) : "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    st_rd = (st_rd == 2) ? 0 : st_rd + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  gemm_epilogue<BM, BN, 2, 2, EPI>(acc, p, C, smem, m0, n0, tile_m, true);
}

}  // namespace blh
