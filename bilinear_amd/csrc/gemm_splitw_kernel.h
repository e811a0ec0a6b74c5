// bf16x3 split GEMM with a PRE-SPLIT B operand (gemm_dtype = 2, forward and dgrad of the W x W
// Linears).  Arithmetic: gemm_split_kernel.h (x = h + m + l exactly, six bf16 MFMAs per product,
// fp32 accumulate).  The weights change once per step but are multiplied by 32 row tiles per
// contraction, so their three bf16 planes are written once per step by wplanes_kernel
// (elementwise.hip) — as [plane][n][k] for forward and as the transpose [plane][k][n] for dgrad —
// and this kernel takes B straight from those planes: in both contractions B is "rows = output
// column, reduction index contiguous".  Only A (activations / dZ, fp32, row-major [M][K]) is split
// here, which halves the VALU + ds_write work that bounds gemm_split_kernel.
//
// Every byte enters LDS by LDS-DMA (global_load_lds_dwordx4 from inline asm, counted vmcnt, as in
// gemm_f32_kernel.h PIPE 3); there is no VMEM load into registers in the K loop:
//   B planes  -> 3-stage ring of [plane][128 rows][64 B] images, 16-B chunks XOR-swizzled on the
//                source side (chunk c of row r sits at c ^ ((r >> 2) & 3): conflict-free
//                ds_read_b128 fragments from unpadded rows);
//   A (fp32)  -> a per-thread staging area: lane l of DMA i lands at (64 i + l) * 16 B and the
//                SAME thread reads it back (ds_read_b128) one and a half K tiles later, splits it
//                and writes the three planes of its 4 values into the A image (same swizzled
//                layout, 2 stages) — a register-staged load whose "registers" are LDS, so the
//                compiler never sees an asynchronous register write.
// Workgroup = 256 threads = 4 waves (2 x 2), tile 128 x 128, K tile 32 = two k-steps of 16
// (12 fragment reads + 24 MFMAs per wave each).  An iteration is two phases around ONE barrier:
//   top:     DMA Lo(kt+3) [rows 0-63 of A]; wait; read Hi(kt+1) staging
//   phase 0: MFMAs k-step 0 | read k-step-1 fragments | split Hi(kt+1) -> A image (kt+1)&1
//   barrier
//   mid:     DMA Bplanes(kt+3), Hi(kt+3); wait; read Lo(kt+2) staging
//   phase 1: MFMAs k-step 1 | read k-step-0 fragments of tile kt+1 | split Lo(kt+2) -> A image kt&1
// Requirements (checked by the dispatcher, which otherwise uses gemm_split_kernel): K % 32 == 0,
// one reduction slab (no split-K), B planes 16-byte aligned with K % 8 == 0.
#pragma once
#include "common.h"
#include "gemm_bf16_kernel.h"    // bf16x8_t
#include "gemm_epilogue.h"
#include "gemm_f32_kernel.h"     // lds_dma16_asm, xcd_remap, g_zero16
#include "gemm_split_kernel.h"   // f32x4_t

namespace blh {

#ifndef BLH_SW_ABLATE
#define BLH_SW_ABLATE 0   // tools only: 1 no B DMA, 2 no A DMA, 4 no barrier, 8 no split/ds_write (wrong results)
#endif
static constexpr int SW_PLANE_BYTES = 128 * 64;                  // one plane of one operand tile
static constexpr int SW_IMG_BYTES = 3 * SW_PLANE_BYTES;          // 24 KB
static constexpr int SW_A_OFF = 0;                               // 2 A images
static constexpr int SW_B_OFF = 2 * SW_IMG_BYTES;                // 3 B images
static constexpr int SW_STG_OFF = 5 * SW_IMG_BYTES;              // staging: Lo[2], Hi[2] of 8 KB
static constexpr int SW_LDS_BYTES = SW_STG_OFF + 4 * 8192;       // 155,648 B

template <int EPI>
__global__ __launch_bounds__(256) void gemm_splitw_kernel(GemmParams p) {
  constexpr int BM = 128, BN = 128, TM = 2, TN = 2, WN = 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  char* lds = reinterpret_cast<char*>(smem);
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) float*)smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (p.N + BN - 1) / BN;
  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int nkt = p.K / 32;
  float* __restrict__ C = p.C;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // ---- DMA plans (loop-invariant part of every source address) ----------------------------
  const char* zp = reinterpret_cast<const char*>(&g_zero16);
  // A: chunk q = tid + 256 c (c = 0..3): row q >> 3, k = 4 (q & 7); Lo = c 0,1 ; Hi = c 2,3
  const char* a_src[4];
  int a_step[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int q = tid + 256 * c, row = m0 + (q >> 3);
    const bool ok = row < p.M;
    a_src[c] = ok ? reinterpret_cast<const char*>(p.A + (int64_t)row * p.lda + ((q & 7) << 2)) : zp;
    a_step[c] = ok ? 128 : 0;   // bytes per K tile
  }
  // B planes: wave-instruction i = 6 wave + d covers plane i / 8, rows 16 (i % 8) .. +15; a lane
  // brings the 16-B chunk that belongs at its LDS position under the swizzle
  const char* b_src[6];
  int b_step[6];
  const char* bplanes = reinterpret_cast<const char*>(p.B);
#pragma unroll
  for (int d = 0; d < 6; ++d) {
    const int i = 6 * wave + d, plane = i >> 3, row = 16 * (i & 7) + (lane >> 2);
    const int c = (lane & 3) ^ ((row >> 2) & 3);
    const bool ok = n0 + row < p.N;
    b_src[d] = ok ? bplanes + 2 * ((int64_t)plane * p.b_plane_stride + (int64_t)(n0 + row) * p.ldb + 8 * c) : zp;
    b_step[d] = ok ? 64 : 0;
  }
  const uint32_t wave_off = __builtin_amdgcn_readfirstlane((uint32_t)(tid & ~63) * 16u);

  // requests past the last tile re-read the last tile (valid memory, never multiplied)
  auto dma_a = [&](int c0, uint32_t slot_base, int t) {   // chunks c0, c0 + 1 of tile t
    const int adv = (t + 1 < nkt) ? 1 : 0;
#pragma unroll
    for (int c = c0; c < c0 + 2; ++c) {
      lds_dma16_asm(reinterpret_cast<const float*>(a_src[c]), slot_base + (uint32_t)((c - c0) * 4096) + wave_off);
      a_src[c] += adv * a_step[c];
    }
  };
  auto dma_b = [&](uint32_t img_base, int t) {
    const int adv = (t + 1 < nkt) ? 1 : 0;
#pragma unroll
    for (int d = 0; d < 6; ++d) {
      lds_dma16_asm(reinterpret_cast<const float*>(b_src[d]), img_base + (uint32_t)((6 * wave + d) * 1024));
      b_src[d] += adv * b_step[d];
    }
  };
  auto stg_lo = [&](int t) { return lds0 + SW_STG_OFF + (uint32_t)((t & 1) * 8192); };
  auto stg_hi = [&](int t) { return lds0 + SW_STG_OFF + 16384 + (uint32_t)((t & 1) * 8192); };
  auto bimg = [&](int t) { return (uint32_t)(SW_B_OFF + (t % 3) * SW_IMG_BYTES); };
  auto aimg = [&](int t) { return (uint32_t)(SW_A_OFF + (t & 1) * SW_IMG_BYTES); };

  // this thread's two staged chunks of a half tile (8 floats)
  auto read_stage = [&](f32x4_t (&r)[2], uint32_t slot_off) {
    r[0] = *reinterpret_cast<const f32x4_t*>(lds + slot_off + tid * 16);
    r[1] = *reinterpret_cast<const f32x4_t*>(lds + slot_off + 4096 + tid * 16);
  };
  // LDS byte offset (inside an image) of the 8-byte piece (row, k = 4 c4 .. 4 c4 + 3)
  auto piece_off = [&](int row, int c4) {
    return row * 64 + ((((c4 >> 1) ^ ((row >> 2) & 3)) << 4) | ((c4 & 1) << 3));
  };
  // full split of a staged half (prologue)
  auto split_store = [&](const f32x4_t (&r)[2], uint32_t img, int half) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const int q = tid + 256 * (2 * half + c);
      uint32_t h0, m0_, l0, h1, m1, l1;
      split3(r[c].x, r[c].y, h0, m0_, l0);
      split3(r[c].z, r[c].w, h1, m1, l1);
      char* at = lds + img + piece_off(q >> 3, q & 7);
      *reinterpret_cast<uint2*>(at) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(at + SW_PLANE_BYTES) = make_uint2(m0_, m1);
      *reinterpret_cast<uint2*>(at + 2 * SW_PLANE_BYTES) = make_uint2(l0, l1);
    }
  };

  // ---- fragments ---------------------------------------------------------------------------
  const int h = lane >> 5, lr = lane & 31;
  struct Frags { bf16x8_t a[3][TM], b[3][TN]; };
  // fragment r = 0..11 of k-step kk: r < 6: A (plane r % 3, sub-tile r / 3), else B
  auto read_frag = [&](Frags& f, uint32_t a_img, uint32_t b_img, int kk, int r) {
    const bool isa = r < 6;
    const int rr = isa ? r : r - 6, pl = rr % 3, t = rr / 3;
    const int row = (isa ? wm : wn) * 64 + t * 32 + lr;
    const uint32_t off = (isa ? a_img : b_img) + pl * SW_PLANE_BYTES + row * 64 + ((((2 * kk + h) ^ ((row >> 2) & 3))) << 4);
    const bf16x8_t v = *reinterpret_cast<const bf16x8_t*>(lds + off);
    if (isa) f.a[pl][t] = v; else f.b[pl][t] = v;
  };

  // One phase: 24 slots {MFMA [+ one stage of a pair-split]}, 12 fragment reads of the next
  // k-step, the split of this thread's staged half tile (8 values = 4 pairs x 3 stages).  The first
  // slots carry no split stage: the staged values were requested from LDS just before the phase.
  auto phase = [&](const Frags& fc, Frags& fn, uint32_t ra_img, uint32_t rb_img, int kkn,
                   const f32x4_t (&st)[2], uint32_t dst_img, int half) {
    constexpr int PA[6] = {2, 0, 1, 1, 0, 0}, PB[6] = {0, 2, 1, 0, 1, 0};
    // slot of split stage g = 0..11 (pair g / 3, stage g % 3)
    constexpr int SLOT[12] = {5, 6, 7, 9, 10, 11, 14, 15, 16, 19, 20, 21};
    uint32_t hh[2], mm[2], ll[2];
    float x = 0.f, y = 0.f;
    uint32_t t0, t1;
    int g = 0;
#pragma unroll
    for (int s = 0; s < 24; ++s) {
      const int t = s >> 2, i = (s >> 1) & 1, j = s & 1;
      if (g < 12 && SLOT[g] == s && !(BLH_SW_ABLATE & 8)) {
        const int u = g / 3, stage = g % 3;
        if (stage == 0) {
          x = st[u >> 1][2 * (u & 1)];
          y = st[u >> 1][2 * (u & 1) + 1];
        }
        if (stage < 2) {
          uint32_t& outp = (stage == 0) ? hh[u & 1] : mm[u & 1];
          asm volatile(
              "v_mfma_f32_32x32x16_bf16 %0, %6, %7, %0\n\t"
              "v_cvt_pk_bf16_f32 %1, %2, %3\n\t"
              "v_lshlrev_b32 %4, 16, %1\n\t"
              "v_and_b32 %5, 0xffff0000, %1\n\t"
              "v_sub_f32 %2, %2, %4\n\t"
              "v_sub_f32 %3, %3, %5"
              : "+a"(acc[i][j]), "=&v"(outp), "+v"(x), "+v"(y), "=&v"(t0), "=&v"(t1)
              : "v"(fc.a[PA[t]][i]), "v"(fc.b[PB[t]][j]) : "memory");
        } else {
          asm volatile(
              "v_mfma_f32_32x32x16_bf16 %0, %4, %5, %0\n\t"
              "v_cvt_pk_bf16_f32 %1, %2, %3"
              : "+a"(acc[i][j]), "=&v"(ll[u & 1]) : "v"(x), "v"(y), "v"(fc.a[PA[t]][i]), "v"(fc.b[PB[t]][j]) : "memory");
          if (u & 1) {
            const int q = tid + 256 * (2 * half + (u >> 1));
            char* at = lds + dst_img + piece_off(q >> 3, q & 7);
            *reinterpret_cast<uint2*>(at) = make_uint2(hh[0], hh[1]);
            *reinterpret_cast<uint2*>(at + SW_PLANE_BYTES) = make_uint2(mm[0], mm[1]);
            *reinterpret_cast<uint2*>(at + 2 * SW_PLANE_BYTES) = make_uint2(ll[0], ll[1]);
          }
        }
        ++g;
      } else {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0"
                     : "+a"(acc[i][j]) : "v"(fc.a[PA[t]][i]), "v"(fc.b[PB[t]][j]) : "memory");
      }
      if (!(s & 1)) read_frag(fn, ra_img, rb_img, kkn, s >> 1);
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue ---------------------------------------------------------------------------
  // (issue order = the steady-state order  Lo(t) | Bplanes(t), Hi(t)  so that the counted waits of
  //  the first iterations count what they expect)
  f32x4_t rlo[2], rhi[2];
  dma_a(0, stg_lo(0), 0);
  dma_b(lds0 + bimg(0), 0);
  dma_a(2, stg_hi(0), 0);
  dma_a(0, stg_lo(1), 1);
  dma_b(lds0 + bimg(1), 1);
  dma_a(2, stg_hi(1), 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  read_stage(rlo, SW_STG_OFF + 0);
  read_stage(rhi, SW_STG_OFF + 16384);
  split_store(rlo, aimg(0), 0);
  split_store(rhi, aimg(0), 1);
  read_stage(rlo, SW_STG_OFF + 8192);
  split_store(rlo, aimg(1), 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  dma_a(0, stg_lo(2), 2);
  dma_b(lds0 + bimg(2), 2);
  dma_a(2, stg_hi(2), 2);
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  Frags f0, f1;
#pragma unroll
  for (int r = 0; r < 12; ++r) read_frag(f0, aimg(0), bimg(0), 0, r);

  for (int kt = 0; kt < nkt; ++kt) {
    // ---- top: Lo(kt+3) -> staging; Hi(kt+1) has landed (and with it Bplanes(kt+1))
    if (!(BLH_SW_ABLATE & 2)) dma_a(0, stg_lo(kt + 3), kt + 3);
    if (!(BLH_SW_ABLATE & 3)) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    read_stage(rhi, SW_STG_OFF + 16384 + ((kt + 1) & 1) * 8192);
    phase(f0, f1, aimg(kt), bimg(kt), 1, rhi, aimg(kt + 1), 1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(BLH_SW_ABLATE & 4)) __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    // ---- mid: Bplanes(kt+3) over tile kt's image, Hi(kt+3) over Hi(kt+1)'s slot; Lo(kt+2) has landed
    if (!(BLH_SW_ABLATE & 1)) dma_b(lds0 + bimg(kt + 3), kt + 3);
    if (!(BLH_SW_ABLATE & 2)) dma_a(2, stg_hi(kt + 3), kt + 3);
    if (!(BLH_SW_ABLATE & 3)) asm volatile("s_waitcnt vmcnt(18)" ::: "memory");
    read_stage(rlo, SW_STG_OFF + (kt & 1) * 8192);
    phase(f1, f0, aimg(kt + 1), bimg(kt + 1), 0, rlo, aimg(kt), 0);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();

  gemm_epilogue<BM, BN, 2, 2, EPI>(acc, p, C, smem, m0, n0, tile_m, true);
}

}  // namespace blh
