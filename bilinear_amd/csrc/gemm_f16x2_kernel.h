// fp32 GEMM from a TWO-piece fp16 split (gemm_dtype = 3, "fp16x2"): fp32 accuracy on the f16
// matrix cores of gfx950 at half the MFMA work of the bf16 split.
//
//   Every operand value is scaled by a per-tensor power of two and written as  x s = hi + lo,
//   hi = RNE_f16(x s), lo = RNE_f16(x s - hi)  (the residual is exact in fp32; hi + lo reproduce
//   x s to 2^-24 relative while lo stays a normal fp16 number, and to 2^-25 of the tensor's scaled
//   maximum below that), and a product is accumulated in fp32 from THREE v_mfma_f32_32x32x16_f16:
//   lo*hi + hi*lo + hi*hi  (the dropped lo*lo is below 2^-24 |a b|).  Against the bf16 split
//   (gemm_split_kernel.h): 3 MFMAs instead of 6, 7 VALU per value pair instead of 11, 4 B of LDS
//   per value instead of 6.
//
//   Range: fp16 has 5 exponent bits, so each operand tensor gets a scale 2^e that puts its largest
//   magnitude in [2^12, 2^13).  The maxima come as partials from the kernels that produce the
//   tensors (bn_apply / bn_bwd_apply: one per wave; wamax_kernel for the weights) and every
//   workgroup reduces them in its prologue — same input, same scale, deterministic.  The scale is
//   applied for free inside the split (v_fma_mix: f16(x*s), then x*s - hi by v_fma_f32) and
//   removed exactly in the epilogue (acc * 2^-(ea+eb)).  A tensor of zeros gets scale 1.
//
// Tile / wave / phase structure, loader (TileSplit) and LDS image (two planes per operand):
// gemm_split_kernel.h.
#pragma once
#include "gemm_split_kernel.h"

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
  int tile = 0, slab = blockIdx.z;   // (split grids: slab-major XCD mapping, gemm_dma.h: xcd_remap_split)
  if (!(gridDim.z > 1 && xcd_remap_split(blockIdx.x, blockIdx.z, gridDim.x, gridDim.z, (p.M + BM - 1) / BM,
                                         tiles_n, &tile, &slab)))
    tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kz0 = slab * p.k_per_split;
  const int k_end = min(p.K, kz0 + p.k_per_split);
  float* __restrict__ C = p.C + (int64_t)slab * p.c_split_stride;
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
  ioa.init(p.A, p.lda, m0, p.M, p.K, tid);
  iob.init(p.B, p.ldb, n0, p.N, p.K, tid);

  auto store_all = [&](auto io, const f32x4_t (&regs)[4], __bf16* dst, float sc) {
    using IO = decltype(io);
#pragma unroll
    for (int put = 0; put < 4; ++put) {
      float a, b, c, d;
      IO::pair(regs, 2 * put, a, b);
      IO::pair(regs, 2 * put + 1, c, d);
      uint32_t h0, l0, h1, l1;
      split2_f16(a * sc, b * sc, h0, l0);
      split2_f16(c * sc, d * sc, h1, l1);
      __bf16* at = IO::row_ptr(dst, put, tid);
      *reinterpret_cast<uint2*>(at) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(at + IO::PLANE) = make_uint2(l0, l1);
    }
  };

  ioa.load(ra[0], kz0, k_end);
  iob.load(rb[0], kz0, k_end);
  ioa.load(ra[1], kz0 + SBK, k_end);
  iob.load(rb[1], kz0 + SBK, k_end);

  // (the first two K tiles are already requested: the reduction below runs in their shadow)
  // ---- per-tensor scales from the amax partials (identical in every workgroup) -------------
  float red_a = 0.f, red_b = 0.f;
  for (int i = tid; i < p.a_namax; i += NT) red_a = fmaxf(red_a, p.a_amax[i]);
  for (int i = tid; i < p.b_namax; i += NT) red_b = fmaxf(red_b, p.b_amax[i]);
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    red_a = fmaxf(red_a, __shfl_xor(red_a, o));
    red_b = fmaxf(red_b, __shfl_xor(red_b, o));
  }
  if (lane == 0) { smem[wave] = red_a; smem[4 + wave] = red_b; }
  __syncthreads();
  red_a = fmaxf(fmaxf(smem[0], smem[1]), fmaxf(smem[2], smem[3]));
  red_b = fmaxf(fmaxf(smem[4], smem[5]), fmaxf(smem[6], smem[7]));
  __syncthreads();
  // e = 12 - floor(log2(amax)); amax == 0 (or denormal) -> e = 0.  The maxima are taken over
  // finite values only (elementwise.hip: amax4), so ex <= 254 and e >= -115; the upper clamp keeps
  // the un-scale factor 2^-e a normal float (e <= 126: tensors below 2^-114 keep full accuracy
  // relative to 2^-126 * 2^13, i.e. to the smallest normal float — nothing is lost there either).
  auto scale_exp = [](float amax) {
    const int ex = (int)((__float_as_uint(amax) >> 23) & 0xffu);
    return (ex == 0 || ex == 255) ? 0 : max(-115, min(126, 12 - (ex - 127)));
  };
  const int ea = __builtin_amdgcn_readfirstlane(scale_exp(red_a));
  const int eb = __builtin_amdgcn_readfirstlane(scale_exp(red_b));
  const float sa = __uint_as_float((uint32_t)(ea + 127) << 23);
  const float sb = __uint_as_float((uint32_t)(eb + 127) << 23);
  // (ea + eb can exceed the exponent range of one float: unscale in two exact steps)
  // un-scale by 2^-(ea+eb) in two exact steps of half the exponent each: |ea + eb| can exceed the
  // exponent range of one float, and with halves the intermediate acc * 2^u1 lies between acc and
  // the final value on the log scale — whenever both of those are finite floats, so is it
  // (un-scaling by 2^-ea then 2^-eb overflowed for a 1e37-sized operand times a 1e-6-sized one)
  const int un_total = -(ea + eb), un_half = un_total / 2;
  const float un_a = __uint_as_float((uint32_t)(127 + un_half) << 23);
  const float un_b = __uint_as_float((uint32_t)(127 + (un_total - un_half)) << 23);

  store_all(ioa, ra[0], lds, sa);
  store_all(iob, rb[0], lds + A_EL, sb);
  ioa.load(ra[0], kz0 + 2 * SBK, k_end);
  iob.load(rb[0], kz0 + 2 * SBK, k_end);
  store_all(ioa, ra[1], lds + STAGE, sa);
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
                   const f32x4_t (&regs)[4], __bf16* dst, float sc) {
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
        asm volatile(   // hi = f16(x s) (RNE), residual x s - hi (exact)
            "v_fma_mixlo_f16 %0, %1, %5, 0\n\t"
            "v_fma_mixhi_f16 %0, %2, %5, 0\n\t"
            "v_cvt_f32_f16 %3, %0\n\t"
            "v_cvt_f32_f16_sdwa %4, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1\n\t"
            "v_fma_f32 %1, %1, %5, -%3\n\t"
            "v_fma_f32 %2, %2, %5, -%4"
            : "=&v"(hh[s & 1]), "+v"(xs[s & 1]), "+v"(ys[s & 1]), "=&v"(t0), "=&v"(t1) : "s"(sc) : "memory");
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
    if (!(BLH_SPLIT_ABLATE & 1)) ioa.load(ra[P ^ 1], k3, k_end);
#pragma unroll
    for (int r = 0; r < IOB::REGS; ++r) asm volatile("" : "+v"(rb[P ^ 1][r]));
    fence();
    phase(f0, f1, sA, 1, iob, rb[P ^ 1], nA + A_EL, sb);
    if (!(BLH_SPLIT_ABLATE & 4)) __syncthreads();
    fence();
    if (!(BLH_SPLIT_ABLATE & 1)) iob.load(rb[P ^ 1], k3, k_end);
#pragma unroll
    for (int r = 0; r < IOA::REGS; ++r) asm volatile("" : "+v"(ra[P][r]));
    fence();
    phase(f1, f0, nA, 0, ioa, ra[P], sA, sa);
  };
  for (int kt = 0; kt < nkt; kt += 2) {
    iter(IntC<0>{}, kt);
    iter(IntC<1>{}, kt + 1);
  }
  __syncthreads();

#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = (acc[i][j][r] * un_a) * un_b;
  gemm_epilogue<BM, BN, 2, 2, EPI>(acc, p, C, smem, m0, n0, tile_m, true);
}

}  // namespace blh
