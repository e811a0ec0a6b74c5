// PROTOTYPE (tools only, not part of the library): fp32 GEMM from a TWO-piece fp16 split,
// x = hi + lo with hi = RNE_f16(x), lo = RNE_f16(x - hi)  (|x - hi - lo| <= 2^-24 |x| inside
// fp16's exponent range), three v_mfma_f32_32x32x16_f16 per product (lo*hi + hi*lo + hi*hi)
// instead of the six bf16 MFMAs of gemm_split_kernel.h, and 4 B of LDS per value instead of 6.
// No range management: operands must sit in fp16's comfortable range (|x| roughly 2^-6 .. 2^12),
// which a shipped version would have to arrange with per-tensor power-of-two scales.
// Same tile / wave / phase structure as gemm_split_kernel.h (whose TileSplit loader it reuses).
#pragma once
#include "../gemm_split_kernel.h"

namespace blh {

template <int BM, int BN>
constexpr size_t gemm_f16x2_lds_bytes() {
  return 2 * 2 * (size_t)(BM + BN) * SPITCH * sizeof(__bf16);
}

__device__ __forceinline__ void split2_f16(float x, float y, uint32_t& h, uint32_t& l) {
  typedef _Float16 h2 __attribute__((ext_vector_type(2)));
  h2 hv = {(_Float16)x, (_Float16)y};
  h = __builtin_bit_cast(uint32_t, hv);
  h2 lv = {(_Float16)(x - (float)hv[0]), (_Float16)(y - (float)hv[1])};
  l = __builtin_bit_cast(uint32_t, lv);
}

template <int LA, int LB, int EPI>
__global__ __launch_bounds__(256) void gemm_f16x2_kernel(GemmParams p) {
  constexpr int BM = 128, BN = 128, NT = 256, WN = 2, TM = 2, TN = 2;
  using IOA = TileSplit<LA, BM, NT>;
  using IOB = TileSplit<LB, BN, NT>;
  constexpr int A_EL = 2 * IOA::PLANE, B_EL = 2 * IOB::PLANE, STAGE = A_EL + B_EL;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* lds = reinterpret_cast<__bf16*>(smem);   // 16-bit elements (fp16 payload)

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

  f32x4_t ra[2][IOA::REGS], rb[2][IOB::REGS];
  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  IOA ioa;
  IOB iob;
  ioa.init(p.A, p.lda, m0, p.M, kz0, k_end, tid);
  iob.init(p.B, p.ldb, n0, p.N, kz0, k_end, tid);

  auto store_all = [&](auto io, const f32x4_t (&regs)[4], __bf16* dst) {
    using IO = decltype(io);
#pragma unroll
    for (int put = 0; put < 4; ++put) {
      float a, b, c, d;
      IO::pair(regs, 2 * put, a, b);
      IO::pair(regs, 2 * put + 1, c, d);
      uint32_t h0, l0, h1, l1;
      split2_f16(a, b, h0, l0);
      split2_f16(c, d, h1, l1);
      __bf16* at = IO::row_ptr(dst, put, tid);
      *reinterpret_cast<uint2*>(at) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(at + IO::PLANE) = make_uint2(l0, l1);
    }
  };

  ioa.load(ra[0], kz0, k_end);
  iob.load(rb[0], kz0, k_end);
  ioa.load(ra[1], kz0 + SBK, k_end);
  iob.load(rb[1], kz0 + SBK, k_end);
  store_all(ioa, ra[0], lds);
  store_all(iob, rb[0], lds + A_EL);
  ioa.load(ra[0], kz0 + 2 * SBK, k_end);
  iob.load(rb[0], kz0 + 2 * SBK, k_end);
  store_all(ioa, ra[1], lds + STAGE);
  __syncthreads();

  const int h = lane >> 5, lr = lane & 31;
  const int a_off = (wm * (TM * 32) + lr) * SPITCH + 8 * h;
  const int b_off = (wn * (TN * 32) + lr) * SPITCH + 8 * h;

  struct Frags { bf16x8_t a[2][TM], b[2][TN]; };
  auto read_frag = [&](Frags& f, const __bf16* sA, int kk, int r) {   // r = 0..7
    const __bf16* sB = sA + A_EL;
    if (r < 4)
      f.a[r % 2][r / 2] = *reinterpret_cast<const bf16x8_t*>(sA + (r % 2) * IOA::PLANE + a_off + (r / 2) * 32 * SPITCH + 16 * kk);
    else
      f.b[(r - 4) % 2][(r - 4) / 2] = *reinterpret_cast<const bf16x8_t*>(sB + ((r - 4) % 2) * IOB::PLANE + b_off + ((r - 4) / 2) * 32 * SPITCH + 16 * kk);
  };

  auto fence = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" : "+a"(acc[i][j]));
    __builtin_amdgcn_sched_barrier(0);
  };

  // phase: 12 MFMAs (3 partial products x 4 accumulators, small terms first), 8 fragment reads
  // of the next k-step, the split of 8 value pairs: slot s < 8 carries stage 0 of pair s (5 VALU)
  // and stage 1 of pair s - 1 (1 VALU)
  auto phase = [&](const Frags& fc, Frags& fn, const __bf16* rA, int kkn, auto io,
                   const f32x4_t (&regs)[4], __bf16* dst) {
    using IO = decltype(io);
    constexpr int PA[3] = {1, 0, 0}, PB[3] = {0, 1, 0};
    uint32_t hh[2], ll[2];
    float xs[2] = {0.f, 0.f}, ys[2] = {0.f, 0.f};
    uint32_t t0, t1;
#pragma unroll
    for (int s = 0; s < 12; ++s) {
      const int t = s >> 2, i = (s >> 1) & 1, j = s & 1;
      if (s >= 1 && s <= 8) {   // MFMA + stage 1 (the lo piece) of pair u = s - 1
        const int u = s - 1;
        asm volatile(
            "v_mfma_f32_32x32x16_f16 %0, %4, %5, %0\n\t"
            "v_cvt_pk_f16_f32 %1, %2, %3"
            : "+a"(acc[i][j]), "=&v"(ll[u & 1]) : "v"(xs[u & 1]), "v"(ys[u & 1]), "v"(fc.a[PA[t]][i]), "v"(fc.b[PB[t]][j]) : "memory");
        if (u & 1) {   // both pairs of put u >> 1 are split
          __bf16* at = IO::row_ptr(dst, u >> 1, tid);
          *reinterpret_cast<uint2*>(at) = make_uint2(hh[0], hh[1]);
          *reinterpret_cast<uint2*>(at + IO::PLANE) = make_uint2(ll[0], ll[1]);
        }
      } else {
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0"
                     : "+a"(acc[i][j]) : "v"(fc.a[PA[t]][i]), "v"(fc.b[PB[t]][j]) : "memory");
      }
      if (s < 8) {   // stage 0 of pair s: hi piece and the exact residuals
        IO::pair(regs, s, xs[s & 1], ys[s & 1]);
        asm volatile(
            "v_cvt_pk_f16_f32 %0, %1, %2\n\t"
            "v_cvt_f32_f16 %3, %0\n\t"
            "v_cvt_f32_f16_sdwa %4, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
            "v_sub_f32 %1, %1, %3\n\t"
            "v_sub_f32 %2, %2, %4"
            : "=&v"(hh[s & 1]), "+v"(xs[s & 1]), "+v"(ys[s & 1]), "=&v"(t0), "=&v"(t1) : : "memory");
        read_frag(fn, rA, kkn, s);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  Frags f0, f1;
#pragma unroll
  for (int r = 0; r < 8; ++r) read_frag(f0, lds, 0, r);

  auto iter = [&](auto pc, int kt) {
    constexpr int P = decltype(pc)::value;
    __bf16* sA = lds + P * STAGE;
    __bf16* nA = lds + (P ^ 1) * STAGE;
    const int k3 = kz0 + (kt + 3) * SBK;
    ioa.load(ra[P ^ 1], k3, k_end);
#pragma unroll
    for (int r = 0; r < IOB::REGS; ++r) asm volatile("" : "+v"(rb[P ^ 1][r]));
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

  gemm_epilogue<BM, BN, 2, 2, EPI>(acc, p, C, smem, m0, n0, tile_m, true);
}

}  // namespace blh
