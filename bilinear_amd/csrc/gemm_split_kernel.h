// fp32 GEMM on the bf16 matrix cores of gfx950 by operand splitting (gemm_dtype = 2, "bf16x3").
//
//   C[M,N] (fp32) = A * B with fp32 operands in memory.  On their way into LDS every operand
//   value x is split into three bf16 pieces  x = h + m + l  (h = RNE_bf16(x), m = RNE_bf16(x - h),
//   l = RNE_bf16(x - h - m); both residuals are exact in fp32, so the three pieces carry all 24
//   significand bits of x), and the product is accumulated in fp32 from the six partial products
//        a_l b_h + a_h b_l + a_m b_m + a_m b_h + a_h b_m + a_h b_h
//   on v_mfma_f32_32x32x16_bf16 (each bf16 x bf16 product is exact in fp32).  The three dropped
//   terms (a_m b_l, a_l b_m, a_l b_l) are below 2^-25 |a b|, i.e. below the rounding of one fp32
//   multiply, so the result carries fp32 accuracy (tests/test_gpu_parity.py compares its error
//   against the fp64 product with that of the exact-fp32 MFMA kernel).
//   Why: gfx950 multiplies fp32 on the matrix cores at 157 TFLOP/s (v_mfma_f32_32x32x2_f32) but
//   bf16 at 2.5 PFLOP/s; six bf16 MFMAs per 16 k cost 6/16 of the fp32-MFMA cycles.
//
// Same contractions, operand layouts (ROWK / KROW) and epilogues as gemm_f32_ring.h.
// Workgroup = 256 threads = 4 waves (2 x 2), tile 128 x 128, K tile 32; each wave owns 64 x 64
// (2 x 2 accumulators of 32 x 32), so a k-step of 16 is 12 fragment reads for 24 MFMAs.
//
// Data path: global fp32 --global_load_dwordx4--> registers (requested 1.5 K tiles ahead) --split (11 VALU per 2 values)--> LDS image
// [stage][operand][plane h|m|l][row][k] bf16, k contiguous, row pitch 40 bf16 = 80 B (the 16-B
// fragment reads of a ds_read_b128 lane group then cover all 64 banks) --ds_read_b128--> MFMA.
// KROW operands (reduction index is the slow one in memory) are transposed in registers: a lane
// loads a 4(k) x 4(row) patch as four float4 and writes four 8-B rows per plane.
#pragma once
#include "common.h"
#include "common.h"
#include "gemm_epilogue.h"
#include "gemm_dma.h"
#include "gemm_epilogue.h"
#include "gemm_dma.h"          // g_zero16, lds_dma16_asm, xcd_remap

namespace blh {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

#ifndef BLH_SPLIT_ABLATE
#define BLH_SPLIT_ABLATE 0   // tools only: 1 no in-loop loads, 2 no split/store, 4 no barrier (wrong results)
#endif
static constexpr int SBK = 32;          // K tile (elements)
static constexpr int SPITCH = SBK + 8;  // bf16 per LDS row

template <int BM, int BN>
constexpr size_t gemm_split_lds_bytes() {
  return 2 * 3 * (size_t)(BM + BN) * SPITCH * sizeof(__bf16);
}

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// (x, y) -> packed bf16 pairs of the three pieces
__device__ __forceinline__ void split3(float x, float y, uint32_t& h, uint32_t& m, uint32_t& l) {
  bf16x2_t hv = {(__bf16)x, (__bf16)y};
  h = __builtin_bit_cast(uint32_t, hv);
  const float rx = x - __uint_as_float(h << 16), ry = y - __uint_as_float(h & 0xffff0000u);
  bf16x2_t mv = {(__bf16)rx, (__bf16)ry};
  m = __builtin_bit_cast(uint32_t, mv);
  const float sx = rx - __uint_as_float(m << 16), sy = ry - __uint_as_float(m & 0xffff0000u);
  bf16x2_t lv = {(__bf16)sx, (__bf16)sy};
  l = __builtin_bit_cast(uint32_t, lv);
}

// ---- global fp32 -> registers -> three bf16 planes in LDS ---------------------------------
template <int LAYOUT, int R, int NT>
struct TileSplit {
  // ROWK: float4 chunks of 4 consecutive k; KROW: patches of 4 k x 4 rows (4 float4)
  static constexpr int ITEMS = (LAYOUT == ROWK) ? (R * SBK / 4) : (R / 4) * (SBK / 4);
  static_assert(ITEMS % NT == 0, "tile not divisible among threads");
  static constexpr int PER = ITEMS / NT;
  static constexpr int REGS = (LAYOUT == ROWK) ? PER : PER * 4;
  static constexpr int PLANE = R * SPITCH;   // bf16 elements per plane

  // ROWK: chunk q -> tile row.  Rows are visited in the order 0,4,1,5,2,6,3,7 inside each block of 8,
  // so that the two rows a 16-lane ds_write_b64 group covers are 4 apart: with the 80-byte pitch
  // their 16-dword spans then fall on disjoint halves of the 32 write banks (adjacent rows overlap
  // on 12 banks).  The global side is unaffected (8 lanes still read one 128-byte row segment).
  __device__ static inline int rowk_row(int q) {
    const int g = q >> 3;
    return (g & ~7) + ((g >> 1) & 3) + ((g & 1) << 2);
  }

  // Loads go through a buffer descriptor over the whole operand: the address of a request is
  // descriptor base + a per-thread 32-bit byte offset (set up once) + a wave-uniform byte offset
  // of the K tile (one SALU add per request; no 64-bit pointer arithmetic in the loop), and the
  // hardware range check returns zeros for everything outside the operand — rows past its end
  // (their per-thread offset is set beyond the descriptor's size) and whole tiles past the last
  // one (the tile offset is) — so the K loop needs neither a branch nor a select.  The reduction
  // range of a launch must be a multiple of the K tile (the dispatcher sends other shapes to the
  // exact kernel).
  __amdgpu_buffer_rsrc_t rsrc;
  uint32_t voff[REGS];     // bytes: ROWK one per chunk, KROW one per k of the 4 x 4 patch
  uint32_t tile_bytes;     // bytes from one K tile to the next
  uint32_t nbytes;         // size of the operand = the offset that is out of range for sure

  __device__ inline void init(const float* __restrict__ base, int64_t ld, int row0, int rows_limit,
                              int k_total, int tid) {
    // operand extent: ROWK [rows_limit][ld], KROW [k_total][ld]
    const uint32_t bytes = (uint32_t)(((LAYOUT == ROWK) ? (int64_t)rows_limit : (int64_t)k_total) * ld * 4);
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, bytes, 0x00020000);
    nbytes = bytes;
    tile_bytes = (LAYOUT == ROWK) ? (uint32_t)(SBK * 4) : (uint32_t)(SBK * ld * 4);
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int q = tid + p * NT;
      const int row = row0 + ((LAYOUT == ROWK) ? rowk_row(q) : ((q >> 3) << 2));
      const int kk = (q & 7) << 2;
      const bool ok = row < rows_limit;
      if (LAYOUT == ROWK) {
        voff[p] = ok ? (uint32_t)(((int64_t)row * ld + kk) * 4) : nbytes;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          voff[p * 4 + j] = ok ? (uint32_t)(((int64_t)(kk + j) * ld + row) * 4) : nbytes;
      }
    }
  }

  // request the K tile whose first k is k0 (any order); k0 >= k_end returns zeros
  __device__ inline void load(f32x4_t (&reg)[REGS], int k0, int k_end) {
    typedef unsigned int u32x4_t __attribute__((ext_vector_type(4)));
    // (written without a conditional: a branch here would split the K loop into blocks and make
    //  the compiler shuttle the accumulators between AGPRs and VGPRs at every block boundary)
    const uint32_t in_range = 0u - (uint32_t)(k0 < k_end);
    const uint32_t soff = (((uint32_t)(k0 / SBK) * tile_bytes) & in_range) | (nbytes & ~in_range);
#pragma unroll
    for (int r = 0; r < REGS; ++r) {
      const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)voff[r], (int)soff, 0);
      reg[r] = __builtin_bit_cast(f32x4_t, v);
    }
  }

  __device__ static inline void put(__bf16* at, float a, float b, float c, float d) {
    uint32_t h0, m0, l0, h1, m1, l1;
    split3(a, b, h0, m0, l0);
    split3(c, d, h1, m1, l1);
    *reinterpret_cast<uint2*>(at) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(at + PLANE) = make_uint2(m0, m1);
    *reinterpret_cast<uint2*>(at + 2 * PLANE) = make_uint2(l0, l1);
  }

  // Incremental form used inside the K loop: the half tile of a thread is 4 "puts" (4 values of
  // one LDS row each) = 8 pair-splits u = 0..7; pair(u) selects the two values, row_ptr(u >> 1)
  // the LDS row they go to.
  static_assert(REGS == 4, "K-loop interleave assumes 16 values per thread and operand");
  __device__ static inline void pair(const f32x4_t (&reg)[REGS], int u, float& x, float& y) {
    if (LAYOUT == ROWK) {
      const f32x4_t c = reg[u >> 1];
      x = (u & 1) ? c.z : c.x;
      y = (u & 1) ? c.w : c.y;
    } else {
      const int j = u >> 1;
      x = (u & 1) ? reg[2][j] : reg[0][j];
      y = (u & 1) ? reg[3][j] : reg[1][j];
    }
  }
  __device__ static inline __bf16* row_ptr(__bf16* lds, int put, int tid) {
    if (LAYOUT == ROWK) {
      const int q = tid + put * NT;
      return lds + rowk_row(q) * SPITCH + ((q & 7) << 2);
    }
    return lds + (((tid >> 3) << 2) + put) * SPITCH + ((tid & 7) << 2);
  }

  __device__ static inline void store(const f32x4_t (&reg)[REGS], __bf16* lds, int tid) {
#pragma unroll
    for (int p = 0; p < PER; ++p) {
      const int q = tid + p * NT;
      if (LAYOUT == ROWK) {
        put(lds + rowk_row(q) * SPITCH + ((q & 7) << 2), reg[p].x, reg[p].y, reg[p].z, reg[p].w);
      } else {
        __bf16* at = lds + ((q >> 3) << 2) * SPITCH + ((q & 7) << 2);
        const f32x4_t &k0 = reg[p * 4], &k1 = reg[p * 4 + 1], &k2 = reg[p * 4 + 2], &k3 = reg[p * 4 + 3];
        put(at, k0.x, k1.x, k2.x, k3.x);
        put(at + SPITCH, k0.y, k1.y, k2.y, k3.y);
        put(at + 2 * SPITCH, k0.z, k1.z, k2.z, k3.z);
        put(at + 3 * SPITCH, k0.w, k1.w, k2.w, k3.w);
      }
    }
  }
};

template <int V>
struct IntC { static constexpr int value = V; };

template <int BM, int BN, int WM, int WN, int LA, int LB, int EPI>
__global__ __launch_bounds__(64 * WM * WN) void gemm_split_kernel(GemmParams p) {
  constexpr int NT = 64 * WM * WN;
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  using IOA = TileSplit<LA, BM, NT>;
  using IOB = TileSplit<LB, BN, NT>;
  constexpr int A_EL = 3 * IOA::PLANE, B_EL = 3 * IOB::PLANE, STAGE = A_EL + B_EL;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  __bf16* lds = reinterpret_cast<__bf16*>(smem);

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

  f32x4_t ra[2][IOA::REGS], rb[2][IOB::REGS];   // two K tiles of look-ahead, static indices

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  const int nkt = (k_end - kz0 + SBK - 1) / SBK;
  IOA ioa;
  IOB iob;
  ioa.init(p.A, p.lda, m0, p.M, p.K, tid);
  iob.init(p.B, p.ldb, n0, p.N, p.K, tid);
  // prologue: tile 0 complete in stage 0, the A half of tile 1 in stage 1, B of tile 1 and all
  // of tile 2 requested
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
  const int a_off = (wm * (TM * 32) + lr) * SPITCH + 8 * h;
  const int b_off = (wn * (TN * 32) + lr) * SPITCH + 8 * h;

  // fragments of one k-step of 16: 3 planes x (TM + TN) ds_read_b128
  struct Frags { bf16x8_t a[3][TM], b[3][TN]; };
  auto read_frags = [&](Frags& f, const __bf16* sA, const __bf16* sB, int kk) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
        f.a[pl][i] = *reinterpret_cast<const bf16x8_t*>(sA + pl * IOA::PLANE + a_off + i * 32 * SPITCH + 16 * kk);
#pragma unroll
      for (int j = 0; j < TN; ++j)
        f.b[pl][j] = *reinterpret_cast<const bf16x8_t*>(sB + pl * IOB::PLANE + b_off + j * 32 * SPITCH + 16 * kk);
    }
  };
  // Phase fence.  __builtin_amdgcn_sched_barrier only fences the machine scheduler; instruction
  // selection is free to emit side-effect-free nodes (the MFMAs) on either side of it.  Passing
  // the accumulators through an empty volatile asm ties every MFMA to its place.
  auto fence = [&]() {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) asm volatile("" : "+a"(acc[i][j]));
    __builtin_amdgcn_sched_barrier(0);
  };

  // One phase = the 24 MFMAs of a k-step (6 partial products x 2 x 2 accumulators, small terms
  // first), the 12 fragment reads of the NEXT k-step and the split of one operand's half tile
  // (8 value pairs per thread).  A 32x32x16 MFMA occupies the matrix core for 32 cycles but holds
  // the wave's issue port for only 8 of them: 5 VALU and one LDS instruction fit in its shadow —
  // but only if they follow THAT MFMA (issue is in order: two MFMAs back to back stall the wave
  // for the first one's 32 cycles and the VALU queue up behind).  So the phase is written as 24
  // slots, each ONE asm statement {MFMA; one third of a pair-split}, with the LDS traffic between
  // the statements; left to itself the compiler bunches the MFMAs.
  //   pair-split stages:  0: h = cvt_pk(x, y); x -= hi(h), y -= lo(h)        (5 VALU)
  //                       1: m = cvt_pk(x, y); x -= ..., y -= ...             (5 VALU)
  //                       2: l = cvt_pk(x, y)                                 (1 VALU)
  static_assert(TM == 2 && TN == 2, "phase layout assumes 2 x 2 accumulators per wave");
  auto phase = [&](const Frags& fc, Frags& fn, const __bf16* rA, const __bf16* rB, int kkn,
                   auto io, const f32x4_t (&regs)[4], __bf16* dst) {
    using IO = decltype(io);
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    uint32_t hh[2], mm[2], ll[2];
    float x = 0.f, y = 0.f;
    uint32_t t0, t1;
#pragma unroll
    for (int s = 0; s < 24; ++s) {
      const int t = s >> 2, i = (s >> 1) & 1, j = s & 1;     // MFMA s: product term t, accumulator (i, j)
      const int u = s / 3, stage = s % 3;                    // pair-split u, stage
      if (BLH_SPLIT_ABLATE & 2) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"
                     : "+a"(acc[i][j]) : "v"(fc.a[PA[t]][i]), "v"(fc.b[PB[t]][j]) : "memory");
      } else if (BLH_SPLIT_ABLATE & 16) {   // timing experiment: 5 independent VALU per slot
        if (stage == 0) IO::pair(regs, u, x, y);
        asm volatile(
            "v_mfma_f32_32x32x16_bf16 %0, %6, %7, %0\n\t"
            "v_add_f32 %1, %2, %2\n\t"
            "v_add_f32 %4, %3, %3\n\t"
            "v_add_f32 %5, %2, %3\n\t"
            "v_mul_f32 %1, %2, %3\n\t"
            "v_mul_f32 %4, %2, %2"
            : "+a"(acc[i][j]), "=&v"(hh[u & 1]), "+v"(x), "+v"(y), "=&v"(t0), "=&v"(t1)
            : "v"(fc.a[PA[t]][i]), "v"(fc.b[PB[t]][j]) : "memory");
        mm[u & 1] = t0; ll[u & 1] = t1;
        if (stage == 2 && (u & 1) && !(BLH_SPLIT_ABLATE & 8)) {
          __bf16* at = IO::row_ptr(dst, u >> 1, tid);
          *reinterpret_cast<uint2*>(at) = make_uint2(hh[0], hh[1]);
          *reinterpret_cast<uint2*>(at + IO::PLANE) = make_uint2(mm[0], mm[1]);
          *reinterpret_cast<uint2*>(at + 2 * IO::PLANE) = make_uint2(ll[0], ll[1]);
        }
      } else if (stage == 0) {
        IO::pair(regs, u, x, y);
        asm volatile(
            "v_mfma_f32_32x32x16_bf16 %0, %6, %7, %0\n\t"
            "v_cvt_pk_bf16_f32 %1, %2, %3\n\t"
            "v_lshlrev_b32 %4, 16, %1\n\t"
            "v_and_b32 %5, 0xffff0000, %1\n\t"
            "v_sub_f32 %2, %2, %4\n\t"
            "v_sub_f32 %3, %3, %5"
            : "+a"(acc[i][j]), "=&v"(hh[u & 1]), "+v"(x), "+v"(y), "=&v"(t0), "=&v"(t1)
            : "v"(fc.a[PA[t]][i]), "v"(fc.b[PB[t]][j]) : "memory");
      } else if (stage == 1) {
        asm volatile(
            "v_mfma_f32_32x32x16_bf16 %0, %6, %7, %0\n\t"
            "v_cvt_pk_bf16_f32 %1, %2, %3\n\t"
            "v_lshlrev_b32 %4, 16, %1\n\t"
            "v_and_b32 %5, 0xffff0000, %1\n\t"
            "v_sub_f32 %2, %2, %4\n\t"
            "v_sub_f32 %3, %3, %5"
            : "+a"(acc[i][j]), "=&v"(mm[u & 1]), "+v"(x), "+v"(y), "=&v"(t0), "=&v"(t1)
            : "v"(fc.a[PA[t]][i]), "v"(fc.b[PB[t]][j]) : "memory");
      } else {
        asm volatile(
            "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\t"
            "v_cvt_pk_bf16_f32 %1, %2, %3"
            : "+a"(acc[i][j]), "=&v"(ll[u & 1]) : "v"(x), "v"(y), "v"(fc.a[PA[t]][i]), "v"(fc.b[PB[t]][j]) : "memory");
        if ((u & 1) && !(BLH_SPLIT_ABLATE & 8)) {
          __bf16* at = IO::row_ptr(dst, u >> 1, tid);
          *reinterpret_cast<uint2*>(at) = make_uint2(hh[0], hh[1]);
          *reinterpret_cast<uint2*>(at + IO::PLANE) = make_uint2(mm[0], mm[1]);
          *reinterpret_cast<uint2*>(at + 2 * IO::PLANE) = make_uint2(ll[0], ll[1]);
        }
      }
      if (!(s & 1)) {   // fragment read r = s / 2 of the next k-step
        const int r = s >> 1;
        if (r < 6)
          fn.a[r % 3][r / 3] = *reinterpret_cast<const bf16x8_t*>(rA + (r % 3) * IOA::PLANE + a_off + (r / 3) * 32 * SPITCH + 16 * kkn);
        else
          fn.b[(r - 6) % 3][(r - 6) / 3] = *reinterpret_cast<const bf16x8_t*>(rB + ((r - 6) % 3) * IOB::PLANE + b_off + ((r - 6) / 3) * 32 * SPITCH + 16 * kkn);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  Frags f0, f1;
  read_frags(f0, lds, lds + A_EL, 0);

  // Iteration kt (parity P = kt & 1) = two symmetric phases around ONE barrier.  The split of a
  // tile is spread over two phases (its A half in phase 1 of iteration t-2, its B half in phase 0
  // of iteration t-1) so that each phase carries 24 MFMAs and half of the VALU work, and every
  // request has 1.5 iterations to arrive.  On entry: LDS stage P = tile kt (complete), f0 = its
  // k-step-0 fragments, stage P^1 = the A half of tile kt+1, rb[P^1] = B of tile kt+1,
  // ra[P] / rb[P] = tile kt+2 (in flight).
  //   phase 0: request A(kt+3) -> ra[P^1] | MFMAs k-step 0 | read k-step-1 fragments (stage P)
  //            | split B(kt+1) -> stage P^1
  //   barrier: stage P^1 complete; nobody reads stage P any more
  //   phase 1: request B(kt+3) -> rb[P^1] | MFMAs k-step 1 | read k-step-0 fragments of tile kt+1
  //            | split A(kt+2) -> stage P
  // The body has no branch: requests past the last tile return zeros and an odd tile count is
  // rounded up with an all-zero tile, so the loop is one basic block (accumulators stay in place).
  auto iter = [&](auto pc, int kt) {
    constexpr int P = decltype(pc)::value;
    __bf16* sA = lds + P * STAGE;
    __bf16* nA = lds + (P ^ 1) * STAGE;
    const int k3 = kz0 + (kt + 3) * SBK;
    if (!(BLH_SPLIT_ABLATE & 1)) ioa.load(ra[P ^ 1], k3, k_end);
    // B(kt+1) enters here (a counted vmcnt: the younger requests stay in flight)
#pragma unroll
    for (int r = 0; r < IOB::REGS; ++r) asm volatile("" : "+v"(rb[P ^ 1][r]));
    fence();
    phase(f0, f1, sA, sA + A_EL, 1, iob, rb[P ^ 1], nA + A_EL);
    if (!(BLH_SPLIT_ABLATE & 4)) __syncthreads();
    fence();
    if (!(BLH_SPLIT_ABLATE & 1)) iob.load(rb[P ^ 1], k3, k_end);
#pragma unroll
    for (int r = 0; r < IOA::REGS; ++r) asm volatile("" : "+v"(ra[P][r]));
    fence();
    phase(f1, f0, nA, nA + A_EL, 0, ioa, ra[P], sA);
  };

#ifdef BLH_SPLIT_STAMP   // tools only: shader clock vs 100 MHz real-time clock around the K loop
  const unsigned long long sc0 = __builtin_amdgcn_s_memtime(), sr0 = __builtin_amdgcn_s_memrealtime();
#endif
  for (int kt = 0; kt < nkt; kt += 2) {
    iter(IntC<0>{}, kt);
    iter(IntC<1>{}, kt + 1);
  }
  __syncthreads();
#ifdef BLH_SPLIT_STAMP
  if (tid == 0 && p.loss_part) {
    unsigned long long* o = reinterpret_cast<unsigned long long*>(p.loss_part) + 2 * (blockIdx.x + gridDim.x * blockIdx.z);
    o[0] = __builtin_amdgcn_s_memtime() - sc0;
    o[1] = __builtin_amdgcn_s_memrealtime() - sr0;
  }
#endif

  gemm_epilogue<BM, BN, WM, WN, EPI>(acc, p, C, smem, m0, n0, tile_m, true);
}

}  // namespace blh
