// bf16x3 split GEMM, 8-wave form (gemm_dtype = 2).  Arithmetic and LDS image: gemm_split_kernel.h
// (x = h + m + l exactly, six bf16 MFMAs per product, fp32 accumulate; planes [row][k] bf16 with an
// 80-byte row pitch).  This form runs TWO waves per SIMD: tools/issue_probe shows that beside one
// v_mfma_f32_32x32x16_bf16 a single wave hides 4 VALU and pays ~5 cycles for each further one,
// while two waves on the SIMD hide ~6 and pay ~3.5 — and the split needs 5-6 VALU per MFMA.
//
// Workgroup = 512 threads = 8 waves (4 x 2), tile 128 x 128, K tile 32 = two k-steps; a wave owns
// 32 x 64 (1 x 2 accumulators): a k-step is 9 fragment reads and 12 MFMAs, and a thread splits
// 8 values of one operand per k-step = 4 pair-splits x 3 stages = one stage per MFMA slot.
#pragma once
#include "common.h"
#include "gemm_bf16_kernel.h"    // bf16x8_t
#include "gemm_epilogue.h"
#include "gemm_f32_kernel.h"     // xcd_remap, g_zero16
#include "gemm_split_kernel.h"   // SBK, SPITCH, IntC

namespace blh {

typedef float f32x2_t __attribute__((ext_vector_type(2)));

// Per-thread staging of one operand's share of a K tile (8 values) for 512 threads.
//   ROWK: two float4 chunks (4 consecutive k of one row each): chunk q = tid + 512 p -> row q >> 3,
//         k = 4 (q & 7).
//   KROW: one patch of 4 k x 2 rows loaded as four float2 (k = 4 (tid & 7) + j, rows 2 (tid >> 3),
//         +1) and transposed in registers: each row's 4 k go out as one 8-byte LDS write per plane.
template <int LAYOUT, int R>
struct Stage8 {
  static_assert(R == 128, "512 threads x 8 values = 128 rows x 32 k");
  static constexpr int NT = 512;
  static constexpr int PLANE = R * SPITCH;
  using reg_t = typename std::conditional<LAYOUT == ROWK, f32x4_t, f32x2_t>::type;
  static constexpr int REGS = (LAYOUT == ROWK) ? 2 : 4;

  const float* src[(LAYOUT == ROWK) ? 2 : 1];
  int64_t step[(LAYOUT == ROWK) ? 2 : 1];
  int64_t kstride;
  int koff;

  __device__ inline void init(const float* __restrict__ base, int64_t ld, int row0, int rows_limit,
                              int k_first, int tid) {
    const float* zp = reinterpret_cast<const float*>(&g_zero16);
    koff = (tid & 7) << 2;
    if (LAYOUT == ROWK) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int row = row0 + ((tid + p * NT) >> 3);
        const bool ok = row < rows_limit;
        src[p] = ok ? base + (int64_t)row * ld + k_first + koff : zp;
        step[p] = ok ? (int64_t)SBK : 0;
      }
      kstride = 0;
    } else {
      const int row = row0 + ((tid >> 3) << 1);
      const bool ok = row < rows_limit;   // rows come in multiples of 4 (launch_gemm checks)
      src[0] = ok ? base + (int64_t)(k_first + koff) * ld + row : zp;
      step[0] = ok ? (int64_t)SBK * ld : 0;
      kstride = ok ? ld : 0;
    }
  }

  // request the next K tile (first k = k0); values at or beyond k_end read the zero page
  __device__ inline void load(reg_t (&reg)[REGS], int k0, int k_end) {
    const float* zp = reinterpret_cast<const float*>(&g_zero16);
    if (LAYOUT == ROWK) {
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const float* g = (k0 + koff < k_end) ? src[p] : zp;
        reg[p] = *reinterpret_cast<const reg_t*>(g);
        src[p] += step[p];
      }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float* g = (k0 + koff + j < k_end) ? src[0] + j * kstride : zp;
        reg[j] = *reinterpret_cast<const reg_t*>(g);
      }
      src[0] += step[0];
    }
  }

  // pair-split u = 0..3: the two values, and (after u = 1, 3) the LDS row of put u >> 1
  __device__ static inline void pair(const reg_t (&reg)[REGS], int u, float& x, float& y) {
    if (LAYOUT == ROWK) {
      x = reg[u >> 1][2 * (u & 1)];
      y = reg[u >> 1][2 * (u & 1) + 1];
    } else {
      x = reg[2 * (u & 1)][u >> 1];
      y = reg[2 * (u & 1) + 1][u >> 1];
    }
  }
  __device__ static inline __bf16* row_ptr(__bf16* lds, int put, int tid) {
    if (LAYOUT == ROWK) {
      const int q = tid + put * NT;
      return lds + (q >> 3) * SPITCH + ((q & 7) << 2);
    }
    return lds + (((tid >> 3) << 1) + put) * SPITCH + ((tid & 7) << 2);
  }
  // whole share at once (prologue)
  __device__ static inline void store(const reg_t (&reg)[REGS], __bf16* lds, int tid) {
#pragma unroll
    for (int put = 0; put < 2; ++put) {
      float a, b, c, d;
      pair(reg, 2 * put, a, b);
      pair(reg, 2 * put + 1, c, d);
      uint32_t h0, m0, l0, h1, m1, l1;
      split3(a, b, h0, m0, l0);
      split3(c, d, h1, m1, l1);
      __bf16* at = row_ptr(lds, put, tid);
      *reinterpret_cast<uint2*>(at) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(at + PLANE) = make_uint2(m0, m1);
      *reinterpret_cast<uint2*>(at + 2 * PLANE) = make_uint2(l0, l1);
    }
  }
};

template <int BM, int BN, int LA, int LB, int EPI>
__global__ __launch_bounds__(512) void gemm_split8_kernel(GemmParams p) {
  constexpr int WN = 2, TN = 2;   // 4 x 2 waves, each 32 x 64
  using IOA = Stage8<LA, BM>;
  using IOB = Stage8<LB, BN>;
  constexpr int A_EL = 3 * IOA::PLANE, B_EL = 3 * IOB::PLANE, STAGE = A_EL + B_EL;
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
  const int nkt = (k_end - kz0 + SBK - 1) / SBK;

  typename IOA::reg_t ra[2][IOA::REGS];
  typename IOB::reg_t rb[2][IOB::REGS];

  f32x16 acc[1][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;

  IOA ioa;
  IOB iob;
  ioa.init(p.A, p.lda, m0, p.M, kz0, tid);
  iob.init(p.B, p.ldb, n0, p.N, kz0, tid);

  // prologue: tile 0 complete in stage 0, the A half of tile 1 in stage 1, B of tile 1 and all of
  // tile 2 requested
  ioa.load(ra[0], kz0, k_end);
  iob.load(rb[0], kz0, k_end);
  ioa.load(ra[1], kz0 + SBK, k_end);
  iob.load(rb[1], kz0 + SBK, k_end);
  IOA::store(ra[0], lds, tid);
  IOB::store(rb[0], lds + A_EL, tid);
  ioa.load(ra[0], kz0 + 2 * SBK, k_end);
  iob.load(rb[0], kz0 + 2 * SBK, k_end);
  IOA::store(ra[1], lds + STAGE, tid);
  __syncthreads();

  const int h = lane >> 5, lr = lane & 31;
  const int a_off = (wm * 32 + lr) * SPITCH + 8 * h;
  const int b_off = (wn * 64 + lr) * SPITCH + 8 * h;

  struct Frags { bf16x8_t a[3], b[3][TN]; };
  // fragment read r = 0..8 of k-step kk
  auto read_frag = [&](Frags& f, const __bf16* sA, int kk, int r) {
    if (r < 3)
      f.a[r] = *reinterpret_cast<const bf16x8_t*>(sA + r * IOA::PLANE + a_off + 16 * kk);
    else
      f.b[(r - 3) % 3][(r - 3) / 3] = *reinterpret_cast<const bf16x8_t*>(
          sA + A_EL + ((r - 3) % 3) * IOB::PLANE + b_off + ((r - 3) / 3) * 32 * SPITCH + 16 * kk);
  };

  // One phase = the 12 MFMAs of a k-step (6 partial products x 2 accumulators, small terms first),
  // the 9 fragment reads of the NEXT k-step and the split of 8 values of one operand, written as
  // 12 slots of ONE asm statement each {MFMA; one stage of a pair-split}: the VALU issue in the
  // shadow of THAT MFMA (issue is in order; left to itself the compiler bunches the MFMAs).
  //   stage 0: h = cvt_pk(x, y); x -= hi(h); y -= lo(h)   (5 VALU)
  //   stage 1: m = cvt_pk(x, y); x -= ...;   y -= ...     (5 VALU)
  //   stage 2: l = cvt_pk(x, y)                           (1 VALU), then the put's 3 ds_write_b64
  auto phase = [&](const Frags& fc, Frags& fn, const __bf16* rA, int kkn, auto io, const auto& regs,
                   __bf16* dst) {
    using IO = decltype(io);
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    uint32_t hh[2], mm[2], ll[2];
    float x = 0.f, y = 0.f;
    uint32_t t0, t1;
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      const int t = s >> 1, j = s & 1;        // MFMA s: product term t, accumulator j
      const int u = s / 3, stage = s % 3;     // pair-split u, stage
      if (stage == 0) {
        IO::pair(regs, u, x, y);
        asm volatile(
            "v_mfma_f32_32x32x16_bf16 %0, %6, %7, %0\n\t"
            "v_cvt_pk_bf16_f32 %1, %2, %3\n\t"
            "v_lshlrev_b32 %4, 16, %1\n\t"
            "v_and_b32 %5, 0xffff0000, %1\n\t"
            "v_sub_f32 %2, %2, %4\n\t"
            "v_sub_f32 %3, %3, %5"
            : "+a"(acc[0][j]), "=&v"(hh[u & 1]), "+v"(x), "+v"(y), "=&v"(t0), "=&v"(t1)
            : "v"(fc.a[PA[t]]), "v"(fc.b[PB[t]][j]) : "memory");
      } else if (stage == 1) {
        asm volatile(
            "v_mfma_f32_32x32x16_bf16 %0, %6, %7, %0\n\t"
            "v_cvt_pk_bf16_f32 %1, %2, %3\n\t"
            "v_lshlrev_b32 %4, 16, %1\n\t"
            "v_and_b32 %5, 0xffff0000, %1\n\t"
            "v_sub_f32 %2, %2, %4\n\t"
            "v_sub_f32 %3, %3, %5"
            : "+a"(acc[0][j]), "=&v"(mm[u & 1]), "+v"(x), "+v"(y), "=&v"(t0), "=&v"(t1)
            : "v"(fc.a[PA[t]]), "v"(fc.b[PB[t]][j]) : "memory");
      } else {
        asm volatile(
            "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\t"
            "v_cvt_pk_bf16_f32 %1, %2, %3"
            : "+a"(acc[0][j]), "=&v"(ll[u & 1]) : "v"(x), "v"(y), "v"(fc.a[PA[t]]), "v"(fc.b[PB[t]][j]) : "memory");
        if (u & 1) {
          __bf16* at = IO::row_ptr(dst, u >> 1, tid);
          *reinterpret_cast<uint2*>(at) = make_uint2(hh[0], hh[1]);
          *reinterpret_cast<uint2*>(at + IO::PLANE) = make_uint2(mm[0], mm[1]);
          *reinterpret_cast<uint2*>(at + 2 * IO::PLANE) = make_uint2(ll[0], ll[1]);
        }
      }
      if (s < 9) read_frag(fn, rA, kkn, s);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  auto fence = [&]() {
#pragma unroll
    for (int j = 0; j < TN; ++j) asm volatile("" : "+a"(acc[0][j]));
    __builtin_amdgcn_sched_barrier(0);
  };

  Frags f0, f1;
#pragma unroll
  for (int r = 0; r < 9; ++r) read_frag(f0, lds, 0, r);

  // Iteration kt (parity P = kt & 1) = two symmetric phases around ONE barrier; see
  // gemm_split_kernel.h for the hand-off (stage P = tile kt, stage P^1 = A half of tile kt+1,
  // rb[P^1] = B of tile kt+1, ra[P] / rb[P] = tile kt+2 in flight):
  //   phase 0: request A(kt+3) | MFMAs k-step 0 | read k-step-1 fragments | split B(kt+1) -> stage P^1
  //   barrier
  //   phase 1: request B(kt+3) | MFMAs k-step 1 | read k-step-0 fragments of tile kt+1 | split A(kt+2) -> stage P
  // Branch-free: requests past the last tile return zeros; an odd tile count is rounded up.
  auto iter = [&](auto pc, int kt) {
    constexpr int P = decltype(pc)::value;
    __bf16* sA = lds + P * STAGE;
    __bf16* nA = lds + (P ^ 1) * STAGE;
    const int k3 = kz0 + (kt + 3) * SBK;
    ioa.load(ra[P ^ 1], k3, k_end);
#pragma unroll
    for (int r = 0; r < IOB::REGS; ++r) asm volatile("" : "+v"(rb[P ^ 1][r]));   // B(kt+1) arrives: counted vmcnt
    fence();
    phase(f0, f1, sA, 1, iob, rb[P ^ 1], nA + A_EL);
    __syncthreads();
    fence();
    iob.load(rb[P ^ 1], k3, k_end);
#pragma unroll
    for (int r = 0; r < IOA::REGS; ++r) asm volatile("" : "+v"(ra[P][r]));
    fence();
    phase(f1, f0, nA, 0, ioa, ra[P], sA);
  };

  for (int kt = 0; kt < nkt; kt += 2) {
    iter(IntC<0>{}, kt);
    iter(IntC<1>{}, kt + 1);
  }
  __syncthreads();

  gemm_epilogue<BM, BN, 4, 2, EPI>(acc, p, C, smem, m0, n0, tile_m, true);
}

}  // namespace blh
