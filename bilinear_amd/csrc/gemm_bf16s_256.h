// bf16-storage GEMM for gfx950, 256 x 256 tile, 8-phase schedule — the large-batch form of
// gemm_bf16s_kernel.h (same contractions, operand layouts and epilogues; BASELINE configs 3-5).
//
// Why a second kernel.  The 128 x 128 kernel (4 waves, two to four workgroups per CU) is bound by
// its LDS-DMA issue: 128 x 128 x 64 per workgroup-step needs 64 B/clk/CU from the texture path for
// the MFMAs to run at their rate (profiles/r02_bf16s_gemm.md: 0.32-0.41 of the bf16 peak).  A
// 256 x 256 tile halves the bytes per MFMA, but only pays when the DMA, the fragment reads and
// the MFMAs of ONE workgroup overlap each other (one workgroup per CU: 128 KiB of LDS), which is
// what the 8-phase structure of cdna_hip_programming.md ("The 256^2 8-phase template") does:
//
//   * 512 threads = 8 waves as 2 (M) x 4 (N); a wave owns 128 x 64 outputs = four quadrants of
//     64 x 32 (q_m, q_n), on v_mfma_f32_16x16x32_bf16: acc[q_m][q_n][4][2] of f32x4 = 128 VGPRs.
//   * K tile 64.  LDS = 2 buffers (even / odd K tile) x 4 half-tiles of 16 KiB: A0, A1 (tile rows
//     0-127 / 128-255), B0, B1 (tile columns likewise).  Quadrant q_m of EVERY wave lies in
//     half-tile A(q_m) (wave w_r owns rows q_m*128 + w_r*64 .. +63), q_n in B(q_n) (columns
//     q_n*128 + w_c*32 .. +31): a half-tile is dead as soon as one phase has read it.
//   * One phase = {fragment reads of the quadrant that changes | ONE half-tile DMA (2 x
//     global_load_lds_dwordx4 per thread) | raw s_barrier | 16 MFMAs (one quadrant x K 64) under
//     s_setprio 1 | raw s_barrier}; four phases per K tile in the order (0,0) (0,1) (1,1) (1,0), so
//     a phase reads B(q_n) + A(q_m) (12 x ds_read_b128), B only (4), A only (8) or nothing.
//   * The waves 4-7 run one barrier behind the waves 0-3: while one half of the workgroup is in
//     its MFMA block the other issues its reads and DMAs — each SIMD hosts one wave of either half.
//   * The DMA stream B0 A0 B1 A1 | B0 ... runs 3 half-tiles ahead: `s_waitcnt vmcnt(6)` in phases
//     4 and 8 only (never 0 in the loop), the buffer it retires is read from the next phase on.
//     A half-tile is re-filled two phases after its last read; B0 one phase after, its four reads
//     being issued first and retired by `s_waitcnt lgkmcnt(#A reads)` before the phase's barrier.
//
// Operand images (one 16 KiB half-tile = 128 rows x 64 k):
//   ROWK (k contiguous in memory): [128 rows][64 k], 128-byte rows, 16-B chunks XOR-swizzled with
//     (row >> 1) & 7 on the DMA source and on the read: conflict-free ds_read_b128 for the
//     16x16x32 operand map (lane l = row l & 15, k chunk l >> 4); a wave's DMA instruction moves
//     8 rows x 128 B = eight whole cache lines.
//   KROW (the reduction index is the memory row): read with ds_read_b64_tr_b16 (lane 4q+p of a
//     16-lane group g supplies k row q, m columns 4p..4p+3 of a 4 x 16 block, lane i receives
//     column i).  The image is laid out so that the byte a lane reads is
//         const(k-step, r, m sub-tile) + [(g >> 1) * 1024 + (g & 1) * 128 + (l & 15) * 8]:
//     element (k = 32 ks + 16 gh + 8 g0 + 4 r + q, m = 64 th + 16 tl + 4 p + e) lives at byte
//         ks*8192 + r*4096 + th*2048 + gh*1024 + tl*256 + g0*128 + q*32 + p*8 + e*2.
//     One address register per operand serves every read (the 256-byte-row image of the 128 x 128
//     kernel needs one per read: its XOR swizzle is not additive), the 32 lanes of a half-wave
//     read 256 consecutive bytes (conflict-free), and a 1 KiB piece (one DMA instruction) is
//     8 k rows x 64 m = eight whole 128-byte cache lines.
//
// Restrictions (the host falls back to the 128 x 128 kernel otherwise): N % 256 == 0, every
// reduction slab a multiple of 128 (two K tiles per loop iteration), 16-byte rows.  M may be
// ragged (rows beyond M are clamped on load and not stored).
#pragma once
#include "gemm_bf16s_kernel.h"

namespace blh {

typedef float f32x4 __attribute__((ext_vector_type(4)));

static constexpr int H256_HALF_BYTES = 128 * 64 * 2;          // 16 KiB
static constexpr size_t H256_LDS_BYTES = 8 * H256_HALF_BYTES; // 2 buffers x (A0 A1 B0 B1) = 128 KiB

// DMA plan of one operand's two half-tiles: per-lane byte offsets from the tile's first row
// (wave-uniform 64-bit base advanced by one K tile after the second half was issued)
template <int LAYOUT>
struct PlanH256 {
  uint32_t voff[2][2];     // [half][chunk]
  const bf16_bits* sbase;
  int64_t tile_step;
  uint32_t wave_off;

  __device__ inline void init(const bf16_bits* __restrict__ base, int64_t ld, int row0, int rows_limit,
                              int k_first, int tid) {
    wave_off = __builtin_amdgcn_readfirstlane((uint32_t)(tid & ~63) * 16u);
    const int last = rows_limit - 1 - row0;       // >= 0: the tile exists
    if (LAYOUT == ROWK) {
      sbase = base + (int64_t)row0 * ld + k_first;
      tile_step = 64;
    } else {
      sbase = base + (int64_t)k_first * ld + row0;
      tile_step = (int64_t)64 * ld;
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int p = 0; p < 2; ++p) {
        const int q = tid + p * 512;              // 16-B chunk of the half-tile image
        if (LAYOUT == ROWK) {
          const int r = q >> 3, c = q & 7;        // image row, slot
          const int kk = (c ^ rowk_swz_h<64>(r)) << 3;
          voff[h][p] = (uint32_t)(((int64_t)min(h * 128 + r, last) * ld + kk) * 2);
        } else {
          // image chunk q (16 B = 8 m of one k row): see the KROW image map at the top
          const int k = 32 * (q >> 9) + 16 * ((q >> 6) & 1) + 8 * ((q >> 3) & 1) + 4 * ((q >> 8) & 1) + ((q >> 1) & 3);
          const int m8 = h * 128 + 64 * ((q >> 7) & 1) + 16 * ((q >> 4) & 3) + 8 * (q & 1);
          voff[h][p] = (uint32_t)(((int64_t)k * ld + min(m8, last - 7)) * 2);   // rows_limit % 8 == 0
        }
      }
  }
  template <int HALF>
  __device__ inline void issue(uint32_t lds_half) {
    // (readfirstlane: the sum is wave-uniform by construction; it keeps hipcc from folding it
    //  into a vector address computation it shares with the fragment reads)
    const uint32_t dst = __builtin_amdgcn_readfirstlane(lds_half + wave_off);
    lds_dma16_sbase<true>(voff[HALF][0], reinterpret_cast<const float*>(sbase), dst);
    lds_dma16_sbase<false>(voff[HALF][1], reinterpret_cast<const float*>(sbase), dst + 512u * 16u);
    if (HALF == 1) sbase += tile_step;
  }
};

// fragments of a 16 x 16 x 32 MFMA operand: T sub-tiles of 16 rows starting at row_base, both
// k-steps (32 k each) of the 64-deep tile
template <int LAYOUT, int T>
__device__ __forceinline__ void read_frags_256(bf16x8_t (&frag)[T][2], const bf16_bits* half, int row_base,
                                               int lane) {
  const int g = lane >> 4, i16 = lane & 15;
  if (LAYOUT == ROWK) {
    const int key = i16 >> 1;                     // (row >> 1) & 7: row_base is a multiple of 16
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
        frag[t][ks] = *reinterpret_cast<const bf16x8_t*>(
            half + (row_base + t * 16 + i16) * 64 + (((g + 4 * ks) ^ key) << 3));
  } else {
    const char* base = reinterpret_cast<const char*>(half) + (g >> 1) * 1024 + (g & 1) * 128 + i16 * 8;
    const int t16b = row_base >> 4;               // first 16-column sub-tile (wave-uniform)
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int t16 = t16b + t;
        s16x4_t v[2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const char* addr = base + ks * 8192 + r * 4096 + (t16 >> 2) * 2048 + (t16 & 3) * 256;
          v[r] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(addr));
        }
        union { s16x4_t s[2]; bf16x8_t b; } u;
        u.s[0] = v[0]; u.s[1] = v[1];
        frag[t][ks] = u.b;
      }
  }
}

// per-tile column (mean, M2) over the tile's 128 NQM rows, from the fp32 accumulators (before rounding):
// p.stat_part[tile_m][2][N].  Uses the first 4 KiB of smem; the caller separates it from other LDS use.
template <int NQM>
__device__ inline void tile_stats_256(f32x4 (&acc)[NQM][2][4][2], const GemmParamsH& p, float* smem, int m0, int n0,
                                      int tile_m) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int g = lane >> 4, c16 = lane & 15;
  // per-tile column (mean, M2) over the tile's 128 NQM rows, from the fp32 values (before rounding).
  // Each wave row (half of the rows) forms its own (n, mean, M2) in registers + two shuffles; one
  // LDS exchange merges the two with Chan's formula.
  float* red = smem;                  // [2 wave rows][256 columns][mean, M2]
  int nrow = 0;                       // valid rows of this wave row (wave-uniform)
#pragma unroll
  for (int qm = 0; qm < NQM; ++qm) nrow += max(0, min(64, p.M - (m0 + qm * 128 + wr * 64)));
  const float inv_n = nrow > 0 ? 1.0f / (float)nrow : 0.f;
#pragma unroll
  for (int qn = 0; qn < 2; ++qn)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float s = 0.f;
#pragma unroll
      for (int qm = 0; qm < NQM; ++qm)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = m0 + qm * 128 + wr * 64 + i * 16 + 4 * g + r;
            if (row < p.M) s += acc[qm][qn][i][j][r];
          }
      s += __shfl_xor(s, 16);
      s += __shfl_xor(s, 32);
      const float mean = s * inv_n;
      float q = 0.f;
#pragma unroll
      for (int qm = 0; qm < NQM; ++qm)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int row = m0 + qm * 128 + wr * 64 + i * 16 + 4 * g + r;
            const float dlt = acc[qm][qn][i][j][r] - mean;
            if (row < p.M) q += dlt * dlt;
          }
      q += __shfl_xor(q, 16);
      q += __shfl_xor(q, 32);
      if (g == 0) {
        const int lc = qn * 128 + wc * 32 + j * 16 + c16;
        red[(wr * 256 + lc) * 2 + 0] = mean;
        red[(wr * 256 + lc) * 2 + 1] = q;
      }
    }
  lds_barrier();
  if (wr == 0 && g == 0) {
    const int n0r = nrow;             // this wave row's count; the other one's:
    int n1r = 0;
#pragma unroll
    for (int qm = 0; qm < NQM; ++qm) n1r += max(0, min(64, p.M - (m0 + qm * 128 + 64)));
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int lc = qn * 128 + wc * 32 + j * 16 + c16;
        const float ma = red[lc * 2], qa = red[lc * 2 + 1];
        const float mb = red[(256 + lc) * 2], qb = red[(256 + lc) * 2 + 1];
        const float na = (float)n0r, nb = (float)n1r, nt = na + nb;
        const float d = mb - ma;
        const float mean = ma + d * (nb / nt);
        const float m2 = qa + qb + d * d * (na * nb / nt);
        p.stat_part[((int64_t)tile_m * 2 + 0) * p.N + n0 + lc] = mean;
        p.stat_part[((int64_t)tile_m * 2 + 1) * p.N + n0 + lc] = m2;
      }
  }
}

// ---- epilogue ------------------------------------------------------------------------------------
// acc[qm][qn][i][j][reg]: row m0 + qm*128 + wr*64 + i*16 + 4*(lane>>4) + reg,
//                         column n0 + qn*128 + wc*32 + j*16 + (lane & 15)
// NQM: quadrant rows of the tile (2: the 256 x 256 kernel, 1: the 128 x 256 kernel of gemm_bf16s_128x256.h)
template <int EPI, bool OUT_BF16, int NQM = 2>
__device__ inline void gemm_epilogue_256(f32x4 (&acc)[NQM][2][4][2], const GemmParamsH& p, void* Cv, float* smem,
                                         int m0, int n0, int tile_m) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int g = lane >> 4, c16 = lane & 15;

  if (EPI == EPI_BIAS || EPI == EPI_BIAS_STATS) {
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float bv = p.bias[n0 + qn * 128 + wc * 32 + j * 16 + c16];
#pragma unroll
        for (int qm = 0; qm < NQM; ++qm)
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[qm][qn][i][j][r] += bv;
      }
  }
  static_assert(EPI != EPI_ADD || OUT_BF16, "the skip-gradient epilogue writes bf16");
  // C stores through LDS (free after the main loop), 64 rows at a time = the rows of one
  // (quadrant row, wave row) pair: fp32 values, row pitch 260 floats (the two 16-lane groups of a
  // ds_write_b32 half-wave sit 4 rows = 16 banks apart), then every thread moves 16-byte pieces
  // of whole rows: a wave writes two full 512-byte (bf16) or 1024-byte (fp32) rows per instruction.
  // (Measured and not kept: 128 rows per pass at pitch 256 — half the barriers, 2-way ds_write_b32
  //  conflicts: 32.6 / 30.4 us against 31.6 / 28.8 us forward / dgrad at M = 16384, W = 1024.)
  constexpr int SP = 260;
  constexpr int VEC = OUT_BF16 ? 8 : 4;
  constexpr int CPR = 256 / VEC;
  float* stg = smem;
  // skip-gradient addend: all 16 pieces of this thread (4 passes x 4) are requested here, in front of
  // the staging passes, so that their latency is paid once per tile and not once per pass
  constexpr int NPS = 2 * NQM;        // staging passes of 64 rows
  uint4 adv[(EPI == EPI_ADD && OUT_BF16) ? NPS : 1][(EPI == EPI_ADD && OUT_BF16) ? 4 : 1];
  if constexpr (EPI == EPI_ADD && OUT_BF16) {
#pragma unroll
    for (int ps = 0; ps < NPS; ++ps)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int q = it * 512 + tid;
        const int row = m0 + (ps >> 1) * 128 + (ps & 1) * 64 + q / CPR, col = n0 + (q % CPR) * VEC;
        adv[ps][it] = row < p.M ? *reinterpret_cast<const uint4*>(p.addend + (int64_t)row * p.ldadd + col)
                                : uint4{0u, 0u, 0u, 0u};
      }
  }
#pragma unroll
  for (int ps = 0; ps < NPS; ++ps) {
    const int qm = ps >> 1, wrp = ps & 1;
    if (wr == wrp) {
#pragma unroll
      for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              stg[(i * 16 + 4 * g + r) * SP + qn * 128 + wc * 32 + j * 16 + c16] = acc[qm][qn][i][j][r];
    }
    __syncthreads();
    const int row_base = m0 + qm * 128 + wrp * 64;
#pragma unroll
    for (int q0 = 0; q0 < 64 * CPR; q0 += 512) {
      const int q = q0 + tid;
      const int lrow = q / CPR, cv = (q % CPR) * VEC;
      const int row = row_base + lrow, col = n0 + cv;
      if (row < p.M) {
        const float4 v0 = *reinterpret_cast<const float4*>(stg + lrow * SP + cv);
        if (OUT_BF16) {
          const float4 v1 = *reinterpret_cast<const float4*>(stg + lrow * SP + cv + 4);
          float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
          if constexpr (EPI == EPI_ADD) {
            const uint4 ad = adv[ps][q0 / 512];
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
    if (ps < NPS - 1 || EPI == EPI_BIAS_STATS) __syncthreads();
  }

  if (EPI == EPI_BIAS_STATS) tile_stats_256<NQM>(acc, p, smem, m0, n0, tile_m);
}

// ---- epilogue of a data gradient that feeds a BatchNorm backward -------------------------
// The GEMM dA = dZ W (+ skip gradient) produces the gradient with respect to the OUTPUT of the stage below;
// that stage's BatchNorm backward starts with two column reductions over (dA, Z) — bn_bwd_reduce_h2, a
// streaming kernel that re-reads dA and Z from memory — and only then forms dZ.  Here the tile is still in
// the accumulators: the store phase reads the matching Z rows and the keep-bit word of each 4 x 8 patch,
// forms dY' = 2 keep [z scale + shift > 0] bf16(dA) (exactly the value the streaming kernel would form from
// the stored bf16 dA: doubling commutes with the rounding), stores dY' in dA's place (bn_bwd_apply_h2 then
// needs neither the bits nor the gate) and adds dY' z and dY' into per-thread column sums, which one LDS
// exchange per tile turns into the [tile row][2][N] partials bn_bwd_finalize_h2 consumes.
// Store-phase mapping: 64 rows per pass = 16 groups of 4 rows x 32 chunks of 8 columns = one (group, chunk)
// per thread: a thread owns exactly one keep-bit word per pass, a wave stores two whole 512-byte row
// segments per instruction.
template <int NQM, bool HAS_ADD>
__device__ inline void gemm_epilogue_256_bnbwd(f32x4 (&acc)[NQM][2][4][2], const GemmParamsH& p, void* Cv,
                                               float* smem, int m0, int n0, int tile_m) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int g = lane >> 4, c16 = lane & 15;
  constexpr int SP = 260;
  constexpr int NPS = 2 * NQM;
  float* stg = smem;
  const int rg = tid >> 5, ch = tid & 31;          // 4-row group of the pass, 8-column chunk
  const int col = n0 + ch * 8;
  const int W8 = p.N >> 3;
  float s1[8], s2[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) s1[c] = s2[c] = 0.f;
  bf16_bits* C = reinterpret_cast<bf16_bits*>(Cv);
  // (register budget: the 256 x 256 kernel holds 128 accumulator registers through this phase, so the
  //  operands of a pass — four Z rows, four addend rows, the keep word — are requested at the head of the
  //  pass, in front of the staging writes and the barrier that hide most of their latency, not a pass ahead;
  //  scale / shift are re-read per pass: 64 bytes from L1)
#pragma unroll
  for (int ps = 0; ps < NPS; ++ps) {
    const int qm = ps >> 1, wrp = ps & 1;
    uint4 zq[4], aq[HAS_ADD ? 4 : 1];
    uint32_t kw;
    {
      const int row0 = m0 + qm * 128 + wrp * 64 + 4 * rg;
      kw = row0 < p.M ? p.bn_keep[(int64_t)(row0 >> 2) * W8 + (col >> 3)] : 0u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t r = min(row0 + j, p.M - 1);
        zq[j] = *reinterpret_cast<const uint4*>(p.bn_z + r * p.ldz + col);
        if constexpr (HAS_ADD) aq[j] = *reinterpret_cast<const uint4*>(p.addend + r * p.ldadd + col);
      }
    }
    if (wr == wrp) {
#pragma unroll
      for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
              stg[(i * 16 + 4 * g + r) * SP + qn * 128 + wc * 32 + j * 16 + c16] = acc[qm][qn][i][j][r];
    }
    __syncthreads();
    float sc[8], sh[8];
    {
      const float4 a0 = *reinterpret_cast<const float4*>(p.bn_scale + col), a1 = *reinterpret_cast<const float4*>(p.bn_scale + col + 4);
      const float4 b0 = *reinterpret_cast<const float4*>(p.bn_shift + col), b1 = *reinterpret_cast<const float4*>(p.bn_shift + col + 4);
      sc[0] = a0.x; sc[1] = a0.y; sc[2] = a0.z; sc[3] = a0.w; sc[4] = a1.x; sc[5] = a1.y; sc[6] = a1.z; sc[7] = a1.w;
      sh[0] = b0.x; sh[1] = b0.y; sh[2] = b0.z; sh[3] = b0.w; sh[4] = b1.x; sh[5] = b1.y; sh[6] = b1.z; sh[7] = b1.w;
    }
    const int row0 = m0 + qm * 128 + wrp * 64 + 4 * rg;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = row0 + j;
      const float4 v0 = *reinterpret_cast<const float4*>(stg + (4 * rg + j) * SP + ch * 8);
      const float4 v1 = *reinterpret_cast<const float4*>(stg + (4 * rg + j) * SP + ch * 8 + 4);
      float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
      if constexpr (HAS_ADD) {
        const uint4 ad = aq[j];
        v[0] += __uint_as_float(ad.x << 16); v[1] += __uint_as_float(ad.x & 0xffff0000u);
        v[2] += __uint_as_float(ad.y << 16); v[3] += __uint_as_float(ad.y & 0xffff0000u);
        v[4] += __uint_as_float(ad.z << 16); v[5] += __uint_as_float(ad.z & 0xffff0000u);
        v[6] += __uint_as_float(ad.w << 16); v[7] += __uint_as_float(ad.w & 0xffff0000u);
      }
      const uint4 zz = zq[j];
      const float z[8] = {__uint_as_float(zz.x << 16), __uint_as_float(zz.x & 0xffff0000u),
                          __uint_as_float(zz.y << 16), __uint_as_float(zz.y & 0xffff0000u),
                          __uint_as_float(zz.z << 16), __uint_as_float(zz.z & 0xffff0000u),
                          __uint_as_float(zz.w << 16), __uint_as_float(zz.w & 0xffff0000u)};
      const uint32_t bits = row < p.M ? (kw >> (8 * j)) : 0u;
      float dy[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const float gr = bf16_to_f32(f32_to_bf16(v[c]));          // the gradient as bf16 storage would hold it
        dy[c] = (((bits >> c) & 1u) && (fmaf(z[c], sc[c], sh[c]) > 0.f)) ? gr * 2.f : 0.f;
        s2[c] += dy[c];
        s1[c] = fmaf(dy[c], z[c], s1[c]);
      }
      if (row < p.M) {
        uint4 o;
        o.x = (uint32_t)f32_to_bf16(dy[0]) | ((uint32_t)f32_to_bf16(dy[1]) << 16);
        o.y = (uint32_t)f32_to_bf16(dy[2]) | ((uint32_t)f32_to_bf16(dy[3]) << 16);
        o.z = (uint32_t)f32_to_bf16(dy[4]) | ((uint32_t)f32_to_bf16(dy[5]) << 16);
        o.w = (uint32_t)f32_to_bf16(dy[6]) | ((uint32_t)f32_to_bf16(dy[7]) << 16);
        *reinterpret_cast<uint4*>(C + (int64_t)row * p.ldc + col) = o;
      }
    }
    __syncthreads();
  }
  // column sums of the tile: 16 row groups x 256 columns, one tensor at a time through LDS
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    float* red = smem;                       // [16][256]
#pragma unroll
    for (int c = 0; c < 8; c += 4)
      *reinterpret_cast<float4*>(red + rg * 256 + ch * 8 + c) =
          which == 0 ? make_float4(s1[c], s1[c + 1], s1[c + 2], s1[c + 3]) : make_float4(s2[c], s2[c + 1], s2[c + 2], s2[c + 3]);
    __syncthreads();
    if (tid < 256) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += red[r * 256 + tid];
      p.stat_part[((int64_t)tile_m * 2 + which) * p.N + n0 + tid] = t;
    }
    __syncthreads();
  }
}

// The same without an addend, staged as PACKED bf16: the gated gradient only needs bf16(acc), so the 128 (64)
// accumulator registers are first rounded and packed in pairs of rows (v_cvt_pk_bf16_f32: registers r, r + 1 of an
// MFMA tile are rows 4 g + r, 4 g + r + 1 of one column), which (1) halves the LDS traffic of the staging,
// (2) lets a whole 128-row half of the tile be staged at once (66 KB of 32-bit row-pair words: one barrier pair
// per quadrant row instead of two), and (3) frees the accumulator registers, so that the Z rows and keep words
// of the NEXT half are requested while this one is processed (the fp32-staged form above, with 128 accumulator
// registers alive, has to request its operands at the head of every pass and waits for memory four times per tile:
// 256 x 256 dgrad 30 -> 50 us at M = 16384, profiles/r04_k9.md).
template <int NQM>
__device__ inline void gemm_epilogue_256_bnbwd_packed(f32x4 (&acc)[NQM][2][4][2], const GemmParamsH& p, void* Cv,
                                                      float* smem, int m0, int n0, int tile_m) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int g = lane >> 4, c16 = lane & 15;
  constexpr int SPW = 260;                           // words per row pair
  uint32_t* stg = reinterpret_cast<uint32_t*>(smem);
  const int rg0 = tid >> 5, ch = tid & 31;           // 4-row group (of 16; + 16 for the second item), 8-column chunk
  const int col = n0 + ch * 8;
  const int W8 = p.N >> 3;
  bf16_bits* C = reinterpret_cast<bf16_bits*>(Cv);
  float s1[8], s2[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) s1[c] = s2[c] = 0.f;
  // operands of one 128-row half: two (4-row group, chunk) items per thread
  uint4 zq[2][2][4];       // [buffer][item][row]
  uint32_t kw[2][2];
  auto request = [&](int qm, int buf) {
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row0 = m0 + qm * 128 + 4 * (it * 16 + rg0);
      kw[buf][it] = row0 < p.M ? p.bn_keep[(int64_t)(row0 >> 2) * W8 + (col >> 3)] : 0u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t r = min(row0 + j, p.M - 1);
        zq[buf][it][j] = *reinterpret_cast<const uint4*>(p.bn_z + r * p.ldz + col);
      }
    }
  };
  request(0, 0);
  // round + pack: pk[qm][qn][i][j][h] = rows (4 g + 2 h, 4 g + 2 h + 1) of the MFMA tile, one column
  uint32_t pk[NQM][2][4][2][2];
  {
    typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int qm = 0; qm < NQM; ++qm)
#pragma unroll
      for (int qn = 0; qn < 2; ++qn)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const bf2 b = {(__bf16)acc[qm][qn][i][j][2 * h], (__bf16)acc[qm][qn][i][j][2 * h + 1]};
              pk[qm][qn][i][j][h] = *reinterpret_cast<const uint32_t*>(&b);
            }
  }
  float sc[8], sh[8];
  {
    const float4 a0 = *reinterpret_cast<const float4*>(p.bn_scale + col), a1 = *reinterpret_cast<const float4*>(p.bn_scale + col + 4);
    const float4 b0 = *reinterpret_cast<const float4*>(p.bn_shift + col), b1 = *reinterpret_cast<const float4*>(p.bn_shift + col + 4);
    sc[0] = a0.x; sc[1] = a0.y; sc[2] = a0.z; sc[3] = a0.w; sc[4] = a1.x; sc[5] = a1.y; sc[6] = a1.z; sc[7] = a1.w;
    sh[0] = b0.x; sh[1] = b0.y; sh[2] = b0.z; sh[3] = b0.w; sh[4] = b1.x; sh[5] = b1.y; sh[6] = b1.z; sh[7] = b1.w;
  }
#pragma unroll
  for (int qm = 0; qm < NQM; ++qm) {
    const int buf = qm & 1;
    // stage the 128 rows of this quadrant row: row pair (wr * 32 + i * 8 + 2 g + h), 256 columns
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            stg[(wr * 32 + i * 8 + 2 * g + h) * SPW + qn * 128 + wc * 32 + j * 16 + c16] = pk[qm][qn][i][j][h];
    if (qm + 1 < NQM) request(qm + 1, buf ^ 1);
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int rgi = it * 16 + rg0;                 // 4-row group of the half: rows 4 rgi .. 4 rgi + 3
      const int row0 = m0 + qm * 128 + 4 * rgi;
#pragma unroll
      for (int hp = 0; hp < 2; ++hp) {               // row pair 2 rgi + hp = rows 4 rgi + 2 hp, + 1
        const uint4 w0 = *reinterpret_cast<const uint4*>(stg + (2 * rgi + hp) * SPW + ch * 8);
        const uint4 w1 = *reinterpret_cast<const uint4*>(stg + (2 * rgi + hp) * SPW + ch * 8 + 4);
        const uint32_t w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int e = 0; e < 2; ++e) {                // even / odd row of the pair
          const int j = 2 * hp + e;
          const int row = row0 + j;
          const uint4 zz = zq[buf][it][j];
          const float z[8] = {__uint_as_float(zz.x << 16), __uint_as_float(zz.x & 0xffff0000u),
                              __uint_as_float(zz.y << 16), __uint_as_float(zz.y & 0xffff0000u),
                              __uint_as_float(zz.z << 16), __uint_as_float(zz.z & 0xffff0000u),
                              __uint_as_float(zz.w << 16), __uint_as_float(zz.w & 0xffff0000u)};
          const uint32_t bits = row < p.M ? (kw[buf][it] >> (8 * j)) : 0u;
          float dy[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const float gr = e == 0 ? __uint_as_float(w[c] << 16) : __uint_as_float(w[c] & 0xffff0000u);
            dy[c] = (((bits >> c) & 1u) && (fmaf(z[c], sc[c], sh[c]) > 0.f)) ? gr * 2.f : 0.f;
            s2[c] += dy[c];
            s1[c] = fmaf(dy[c], z[c], s1[c]);
          }
          if (row < p.M) {
            uint4 o;
            o.x = (uint32_t)f32_to_bf16(dy[0]) | ((uint32_t)f32_to_bf16(dy[1]) << 16);
            o.y = (uint32_t)f32_to_bf16(dy[2]) | ((uint32_t)f32_to_bf16(dy[3]) << 16);
            o.z = (uint32_t)f32_to_bf16(dy[4]) | ((uint32_t)f32_to_bf16(dy[5]) << 16);
            o.w = (uint32_t)f32_to_bf16(dy[6]) | ((uint32_t)f32_to_bf16(dy[7]) << 16);
            *reinterpret_cast<uint4*>(C + (int64_t)row * p.ldc + col) = o;
          }
        }
      }
    }
    __syncthreads();
  }
  // column sums of the tile: 16 thread rows x 256 columns, one tensor at a time through LDS
#pragma unroll
  for (int which = 0; which < 2; ++which) {
    float* red = smem;                       // [16][256]
#pragma unroll
    for (int c = 0; c < 8; c += 4)
      *reinterpret_cast<float4*>(red + rg0 * 256 + ch * 8 + c) =
          which == 0 ? make_float4(s1[c], s1[c + 1], s1[c + 2], s1[c + 3]) : make_float4(s2[c], s2[c + 1], s2[c + 2], s2[c + 3]);
    __syncthreads();
    if (tid < 256) {
      float t = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) t += red[r * 256 + tid];
      p.stat_part[((int64_t)tile_m * 2 + which) * p.N + n0 + tid] = t;
    }
    __syncthreads();
  }
}

}  // namespace blh
namespace blh {

// ---- kernel ----------------------------------------------------------------------------------------
template <int LA, int LB, int EPI, bool OUT_BF16>
__global__ __launch_bounds__(512, 2) void gemm_bf16s_256_kernel(GemmParamsH p) {
  constexpr int BM = 256, BN = 256;
  // fragment reads per phase (ROWK: one ds_read_b128 per fragment, KROW: two transposing reads)
  constexpr int NA_READS = (LA == ROWK) ? 8 : 16;
  constexpr int LGKM_AFTER_B = NA_READS > 15 ? 15 : NA_READS;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const bf16_bits* lds = reinterpret_cast<const bf16_bits*>(smem);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int tiles_n = p.N / BN;
  int tile = 0, slab = blockIdx.z;
  if (p.batch_splits > 0) {
    // batched launch: the (tile, z) space as one list, an XCD takes a contiguous piece of it — all tiles
    // of one (item, slab) pair sit on one XCD, whose L2 then serves the operand panels they share
    const int logical = xcd_remap(blockIdx.z * gridDim.x + blockIdx.x, gridDim.x * gridDim.z);
    tile = logical % (int)gridDim.x;
    slab = logical / (int)gridDim.x;
  } else if (!(gridDim.z > 1 && xcd_remap_split(blockIdx.x, blockIdx.z, gridDim.x, gridDim.z, (p.M + BM - 1) / BM,
                                                tiles_n, &tile, &slab)))
    tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tile_m = tile / tiles_n, tile_n = tile % tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  // batched launch: the z index carries (batch item, slab); the items are independent GEMMs of one shape
  int item = 0;
  if (p.batch_splits > 0) {
    item = slab / p.batch_splits;
    slab -= item * p.batch_splits;
  }
  const bf16_bits* Aop = p.A + (int64_t)item * p.a_batch_stride;
  const bf16_bits* Bop = p.B + (int64_t)item * p.b_batch_stride;
  const int kz0 = slab * p.k_per_split;
  const int k_end = min(p.K, kz0 + p.k_per_split);
  const int nit = (k_end - kz0) >> 7;            // iterations of two K tiles (host: extent % 128 == 0, >= 128)
  const int64_t coff = (int64_t)item * p.c_batch_stride + (int64_t)slab * p.c_split_stride;
  void* C = OUT_BF16 ? (void*)(reinterpret_cast<bf16_bits*>(p.C) + coff)
                     : (void*)(reinterpret_cast<float*>(p.C) + coff);

  f32x4 acc[2][2][4][2];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[a][b][i][j][r] = 0.f;

  PlanH256<LA> planA;
  PlanH256<LB> planB;
  planA.init(Aop, p.lda, m0, p.M, kz0, tid);
  planB.init(Bop, p.ldb, n0, p.N, kz0, tid);
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) float*)smem);
  // LDS map: A halves of both buffers in the first 64 KiB, B halves in the second (every fragment
  // read of an operand is then base register + a 16-bit immediate): [EV.A0 EV.A1 OD.A0 OD.A1 | B likewise]
  constexpr uint32_t OA0 = 0, OA1 = H256_HALF_BYTES, OB0 = 4 * H256_HALF_BYTES, OB1 = 5 * H256_HALF_BYTES;
  constexpr uint32_t EV = 0, OD = 2 * H256_HALF_BYTES;

  bf16x8_t fa[4][2], fb[2][2][2];

#define BLH_SB() __builtin_amdgcn_sched_barrier(0)
#define BLH_LOAD_A(QM, BUF) \
  read_frags_256<LA, 4>(fa, lds + ((BUF) + ((QM) ? OA1 : OA0)) / 2, wr * 64, lane)
#define BLH_LOAD_B(QN, BUF) \
  read_frags_256<LB, 2>(fb[QN], lds + ((BUF) + ((QN) ? OB1 : OB0)) / 2, wc * 32, lane)
#define BLH_MFMA(QM, QN)                                                                          \
  __builtin_amdgcn_s_setprio(1);                                                                  \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                   \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                   \
    acc[QM][QN][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][ks], fb[QN][j][ks],         \
                                                                acc[QM][QN][i][j], 0, 0, 0);      \
  __builtin_amdgcn_s_setprio(0);
#define BLH_BAR() do { BLH_SB(); __builtin_amdgcn_s_barrier(); BLH_SB(); } while (0)
#define BLH_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define BLH_WAIT_LGKM(N) asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory")

  // prologue: K tile 0 whole (even buffer) and B0 A0 B1 of tile 1 (odd buffer) in flight
  planB.template issue<0>(lds0 + EV + OB0);
  planA.template issue<0>(lds0 + EV + OA0);
  planB.template issue<1>(lds0 + EV + OB1);
  planA.template issue<1>(lds0 + EV + OA1);
  planB.template issue<0>(lds0 + OD + OB0);
  planA.template issue<0>(lds0 + OD + OA0);
  planB.template issue<1>(lds0 + OD + OB1);
  BLH_WAIT_VM(6);
  BLH_BAR();
  if (wr == 1) BLH_BAR();                        // waves 4-7 run one barrier behind waves 0-3

  // one K tile = four phases; STG: the DMAs of this iteration exist (all but the last iteration)
#define BLH_KTILE(BUF, STG, ST1, ST2, ST3, ST4, WAIT4)                                            \
  /* phase 1: quadrant (0,0) */                                                                   \
  BLH_LOAD_B(0, BUF); BLH_SB(); BLH_LOAD_A(0, BUF); BLH_SB();                                     \
  ST1;                                                                                            \
  BLH_WAIT_LGKM(LGKM_AFTER_B);                                                                    \
  BLH_BAR(); BLH_MFMA(0, 0) BLH_BAR();                                                            \
  /* phase 2: quadrant (0,1) */                                                                   \
  BLH_LOAD_B(1, BUF); BLH_SB();                                                                   \
  if (STG) { ST2; }                                                                               \
  BLH_BAR(); BLH_MFMA(0, 1) BLH_BAR();                                                            \
  /* phase 3: quadrant (1,1) */                                                                   \
  BLH_LOAD_A(1, BUF); BLH_SB();                                                                   \
  if (STG) { ST3; }                                                                               \
  BLH_BAR(); BLH_MFMA(1, 1) BLH_BAR();                                                            \
  /* phase 4: quadrant (1,0) */                                                                   \
  if (STG) { ST4; }                                                                               \
  WAIT4;                                                                                          \
  BLH_BAR(); BLH_MFMA(1, 0) BLH_BAR();

  for (int it = 0; it < nit - 1; ++it) {
    BLH_KTILE(EV, true, planA.template issue<1>(lds0 + OD + OA1), planB.template issue<0>(lds0 + EV + OB0),
              planA.template issue<0>(lds0 + EV + OA0), planB.template issue<1>(lds0 + EV + OB1), BLH_WAIT_VM(6))
    BLH_KTILE(OD, true, planA.template issue<1>(lds0 + EV + OA1), planB.template issue<0>(lds0 + OD + OB0),
              planA.template issue<0>(lds0 + OD + OA0), planB.template issue<1>(lds0 + OD + OB1), BLH_WAIT_VM(6))
  }
  // last iteration: only the A1 half of the last K tile is still to be issued
  BLH_KTILE(EV, false, planA.template issue<1>(lds0 + OD + OA1), (void)0, (void)0, (void)0, BLH_WAIT_VM(0))
  BLH_KTILE(OD, false, (void)0, (void)0, (void)0, (void)0, (void)0)
  if (wr == 0) BLH_BAR();                        // the barrier waves 4-7 took at the start

#undef BLH_KTILE
#undef BLH_LOAD_A
#undef BLH_LOAD_B
#undef BLH_MFMA
#undef BLH_BAR
#undef BLH_WAIT_VM
#undef BLH_WAIT_LGKM
#undef BLH_SB
  __syncthreads();
  if constexpr (EPI == EPI_BN_BWD) gemm_epilogue_256_bnbwd_packed<2>(acc, p, C, smem, m0, n0, tile_m);
  else if constexpr (EPI == EPI_BN_BWD_ADD) gemm_epilogue_256_bnbwd<2, true>(acc, p, C, smem, m0, n0, tile_m);
  else gemm_epilogue_256<EPI, OUT_BF16>(acc, p, C, smem, m0, n0, tile_m);
}

}  // namespace blh
