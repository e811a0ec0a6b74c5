// bf16-storage GEMM for gfx950, PERSISTENT 128 x 256 tiles: one workgroup per CU walks several output tiles
// back to back on the three-deep LDS ring of gemm_bf16s_128x256.h and lets the C tile of tile t leave
// through a 32-row LDS window WHILE the K loop of tile t + 1 runs.
//
// Why (profiles/r04_128x256_stamps_8192.txt): a launch of one tile per CU spends a third of its
// time on things that do not depend on K — filling the ring (1.5 us), and the output burst of all 256 CUs at
// once (4-8 us at the fabric's write rate) during which no matrix instruction runs.  M = 16384 at W = 1024
// (BASELINE configs[2]) is 512 tiles of 128 x 256 = two per CU: here the ring never drains between them
// (the DMA stream simply continues with the next tile's operand panels, two K tiles ahead of the MFMAs), and
// the finished tile is rounded to bf16, kept in 32 registers per lane (row pairs packed with
// v_cvt_pk_bf16_f32) and written out SPREAD OVER THE WHOLE K LOOP of its successor (the first build drained it
// in four K tiles: 64 KiB per CU in 2.6 us is 6.4 TB/s chip-wide, more than the fabric writes, and every
// store that was late held up the counted DMA wait behind it: no gain).  The tile leaves in eight windows of
// 16 rows (8 KiB of LDS beside the 144 KiB ring), window w during K tiles [w nkt/8, (w+1) nkt/8):
//     phase 1 of the window's first K tile: the four waves that own rows 16 w .. 16 w + 15 put them into the
//              window, 8 ds_write_b32 each;
//     phase 2 of each of its K tiles: 16 / (nkt/8) rows go out, one 16-byte global store per thread of the
//              first waves, issued BEHIND the K tile's counted `s_waitcnt vmcnt`, i.e. between the DMA
//              groups of two K tiles in the wave's in-order memory queue; the wait of the NEXT K tile counts
//              one more (7) in the storing waves, so a store has two K tiles to complete.
// Only the last tile of a workgroup pays the ordinary epilogue (gemm_epilogue_256).  At a tile boundary the
// two wave halves close their one-barrier stagger, add the bias (which an LDS-DMA of one wave fetched three
// K tiles earlier: a compiler-visible global load here would make hipcc wait for vmcnt(0), i.e. drain the
// ring), take the BatchNorm partials of the tile from the fp32 accumulators (tile_stats_256), pack, clear
// the accumulators and re-open the stagger: about 1 us without matrix work per boundary.
//
// Restrictions (the host falls back to the non-persistent kernels): A is ROWK, C is bf16, epilogues
// EPI_STORE / EPI_BIAS / EPI_BIAS_STATS, no split, M % 128 == 0, N % 256 == 0, K % 64 == 0 and at least
// K = 512, 1024 or 2048 (8, 16 or 32 K tiles: a window per nkt/8 K tiles), grid a multiple of 8 or one tile
// per workgroup.  Results are bit-identical to gemm_bf16s_128x256_kernel (same MFMA order per element, same
// statistics code per 128-row tile).
#pragma once
#include "../gemm_bf16s_128x256.h"

namespace blh {

static constexpr int HP128_SPARE_BYTES = 16384;            // window 8 KiB | bias slot 1 KiB | statistics exchange 4 KiB
static constexpr size_t HP128_LDS_BYTES = H128_LDS_BYTES + HP128_SPARE_BYTES;   // 160 KiB
static inline bool gemm_bf16s_p128x256_k_ok(int K) { return K == 512 || K == 1024 || K == 2048; }

// DBG (tools/bf16s_bench only; results are then wrong): 1 the output windows are staged but not stored, 2 no window
// traffic at all, 3 stores without the windows' LDS traffic
template <int LB, int EPI, int DBG = 0>
__global__ __launch_bounds__(512, 2) void gemm_bf16s_p128x256_kernel(GemmParamsH p, int tiles_total) {
  constexpr int BM = 128, BN = 256;
  constexpr bool HAS_BIAS = (EPI == EPI_BIAS || EPI == EPI_BIAS_STATS);
  static_assert(EPI == EPI_STORE || EPI == EPI_BIAS || EPI == EPI_BIAS_STATS, "epilogue not built in the persistent form");
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const bf16_bits* lds = reinterpret_cast<const bf16_bits*>(smem);
  float* spare = smem + H128_LDS_BYTES / 4;
  uint32_t* win = reinterpret_cast<uint32_t*>(spare);          // [8 row pairs][256 columns] words
  float* bias_slot = spare + 2048;                             // 256 floats
  float* stat_xchg = spare + 2048 + 256;                       // 4 KiB (tile_stats_256)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 2, wc = wave & 3;
  const int g = lane >> 4, c16 = lane & 15;
  const int tiles_n = p.N / BN;
  const int G = gridDim.x, b = blockIdx.x;
  const int ntw = (tiles_total - b + G - 1) / G;         // tiles of this workgroup (>= 1: host launches G <= tiles)
  const int nkt = p.K >> 6;
  const int total = ntw * nkt;
  const int per = nkt >> 3;                              // K tiles per 16-row output window (1, 2, 4)
  const int rpp = 16 / per;                              // rows stored per K tile
  const bool st_wave = wave * 64 < rpp * 32;             // this wave stores (one 16-byte piece per thread)
  bf16_bits* C = reinterpret_cast<bf16_bits*>(p.C);

  f32x4 acc[1][2][4][2];
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[0][q][i][j][r] = 0.f;
  uint32_t pk[2][4][2][2];                               // the previous tile, bf16 row pairs: [qn][i][j][h]
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j) pk[q][i][j][0] = pk[q][i][j][1] = 0u;

  // tile of round r (the XCD of a workgroup is b & 7 in every round: G % 8 == 0 or one round)
  int ctile = xcd_remap(b, tiles_total);
  int ctm = __builtin_amdgcn_readfirstlane(ctile / tiles_n);
  int cm0 = ctm * BM, cn0 = (ctile - ctm * tiles_n) * BN;     // tile being computed
  int dm0 = 0, dn0 = 0;                                       // tile being written out

  PlanH128<ROWK, 1> planA;
  PlanH128<LB, 2> planB;
  planA.init(p.A, p.lda, cm0, p.M, 0, tid);
  planB.init(p.B, p.ldb, cn0, p.N, 0, tid);
  int ik = 0, ir = 0;                                         // K tile / round of the next DMA group
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) float*)smem);
  constexpr uint32_t OA = 0, OB0 = H256_HALF_BYTES, OB1 = 2 * H256_HALF_BYTES;
  const uint32_t bias_voff = (uint32_t)lane * 16u;
  const uint32_t bias_lds = lds0 + (uint32_t)H128_LDS_BYTES + 8192u;

  bf16x8_t fa[4][2], fb[2][2][2];

#define BLH_SB() __builtin_amdgcn_sched_barrier(0)
#define BLH_LOAD_A(BOFF) read_frags_256<ROWK, 4>(fa, lds + ((BOFF) + OA) / 2, wr * 64, lane)
#define BLH_LOAD_B(QN, BOFF) read_frags_256<LB, 2>(fb[QN], lds + ((BOFF) + ((QN) ? OB1 : OB0)) / 2, wc * 32, lane)
#define BLH_MFMA(QN)                                                                              \
  __builtin_amdgcn_s_setprio(1);                                                                  \
  _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                \
  _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                   \
  _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                   \
    acc[0][QN][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i][ks], fb[QN][j][ks],          \
                                                               acc[0][QN][i][j], 0, 0, 0);        \
  __builtin_amdgcn_s_setprio(0);
#define BLH_BAR() do { BLH_SB(); __builtin_amdgcn_s_barrier(); BLH_SB(); } while (0)
#define BLH_WAIT_VM(N) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory")
#define BLH_STAGE_1(ROFF)                                                                         \
  planB.template issue<0, 0, true>(lds0 + (ROFF) + OB0);                                          \
  planB.template issue<0, 1, false>(lds0 + (ROFF) + OB0);                                         \
  planA.template issue<0, 0, true>(lds0 + (ROFF) + OA);
  // (after the last K tile of a round the operand bases move to the next tile of this workgroup)
#define BLH_STAGE_2(ROFF)                                                                         \
  planA.template issue<0, 1, false>(lds0 + (ROFF) + OA);                                          \
  planB.template issue<1, 0, false>(lds0 + (ROFF) + OB1);                                         \
  planB.template issue<1, 1, false>(lds0 + (ROFF) + OB1);                                         \
  planA.advance(); planB.advance();                                                               \
  if (++ik == nkt) {                                                                              \
    ik = 0; ++ir;                                                                                 \
    if (ir < ntw) {                                                                               \
      const int t_ = xcd_remap(b + ir * G, tiles_total);                                          \
      const int tm_ = __builtin_amdgcn_readfirstlane(t_ / tiles_n), tn_ = t_ - tm_ * tiles_n;     \
      planA.sbase = p.A + (int64_t)tm_ * BM * p.lda;                                              \
      planB.sbase = (LB == ROWK) ? p.B + (int64_t)tn_ * BN * p.ldb : p.B + tn_ * BN;              \
    }                                                                                             \
  }
  // one K tile (two phases).  HOOK1 runs in front of phase 1's first barrier, HOOK2 behind the counted wait of
  // phase 2 (global stores / DMAs issued there fall between the DMA groups of two K tiles: counts unchanged)
#define BLH_KTILE(BOFF, STG, ROFF, WAIT, HOOK1, HOOK2)                                            \
  BLH_LOAD_B(0, BOFF); BLH_SB(); BLH_LOAD_A(BOFF); BLH_SB();                                      \
  if (STG) { BLH_STAGE_1(ROFF) }                                                                  \
  HOOK1;                                                                                          \
  BLH_BAR(); BLH_MFMA(0) BLH_BAR();                                                               \
  BLH_LOAD_B(1, BOFF); BLH_SB();                                                                  \
  if (STG) { BLH_STAGE_2(ROFF) }                                                                  \
  WAIT;                                                                                           \
  HOOK2;                                                                                          \
  BLH_BAR(); BLH_MFMA(1) BLH_BAR();
#define BLH_ROTATE() do { roff = boff; boff = (boff == 2u * H128_BUF_BYTES) ? 0u : boff + (uint32_t)H128_BUF_BYTES; } while (0)

  // window w (rows 16 w .. 16 w + 15 of the finished tile): written by the wave row that owns them — always from
  // pk[.][0][.][.]: a wave that has written a window shifts its remaining three 16-row slices down, so the window
  // index is a run-time value and the K loop has ONE body (a body per window costs registers: 256 + spills) ...
#define BLH_WIN_WRITE(W)                                                                          \
  if (DBG < 2 && wr == ((W) >> 2)) {                                                                         \
    _Pragma("unroll") for (int q = 0; q < 2; ++q)                                                 \
    _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                 \
    _Pragma("unroll") for (int h = 0; h < 2; ++h) {                                               \
      win[(2 * g + h) * 256 + q * 128 + wc * 32 + j * 16 + c16] = pk[q][0][j][h];                 \
      pk[q][0][j][h] = pk[q][1][j][h]; pk[q][1][j][h] = pk[q][2][j][h]; pk[q][2][j][h] = pk[q][3][j][h]; \
    }                                                                                             \
  }
  // ... and stored rpp rows per K tile: thread = (row tid >> 5 of part S, 8-column chunk tid & 31)
#define BLH_WIN_STORE(W, S)                                                                       \
  if (DBG != 2 && st_wave) {                                                                      \
    const int rl = (S) * rpp + (tid >> 5), ch = tid & 31;                                         \
    uint4 w0 = uint4{pk[0][0][0][0], pk[0][0][0][1], pk[0][0][1][0], pk[0][0][1][1]}, w1 = w0;    \
    if (DBG != 3) {                                                                               \
      w0 = *reinterpret_cast<const uint4*>(win + (rl >> 1) * 256 + ch * 8);                       \
      w1 = *reinterpret_cast<const uint4*>(win + (rl >> 1) * 256 + ch * 8 + 4);                   \
    }                                                                                             \
    uint4 o;                                                                                      \
    if (rl & 1) {                                                                                 \
      o.x = (w0.x >> 16) | (w0.y & 0xffff0000u); o.y = (w0.z >> 16) | (w0.w & 0xffff0000u);       \
      o.z = (w1.x >> 16) | (w1.y & 0xffff0000u); o.w = (w1.z >> 16) | (w1.w & 0xffff0000u);       \
    } else {                                                                                      \
      o.x = (w0.x & 0xffffu) | (w0.y << 16); o.y = (w0.z & 0xffffu) | (w0.w << 16);               \
      o.z = (w1.x & 0xffffu) | (w1.y << 16); o.w = (w1.z & 0xffffu) | (w1.w << 16);               \
    }                                                                                             \
    if (DBG != 1) *reinterpret_cast<uint4*>(C + (int64_t)(dm0 + 16 * (W) + rl) * p.ldc + dn0 + ch * 8) = o; \
    else asm volatile("" :: "v"(o.x), "v"(o.y), "v"(o.z), "v"(o.w));                              \
  }

  // prologue: K tiles 0 and 1 of the first tile requested (buffers 0 and 1), tile 0 landed
  BLH_STAGE_1(0u) BLH_STAGE_2(0u)
  BLH_STAGE_1((uint32_t)H128_BUF_BYTES) BLH_STAGE_2((uint32_t)H128_BUF_BYTES)
  BLH_WAIT_VM(6);
  BLH_BAR();
  if (wr == 1) BLH_BAR();                        // waves 4-7 run one barrier behind waves 0-3

  uint32_t boff = 0, roff = 2u * H128_BUF_BYTES;  // buffer of the current K tile / of K tile + 2
  int gt = 0;                                     // K tiles done, over all tiles of this workgroup
  const int per_shift = per == 4 ? 2 : per == 2 ? 1 : 0;
  for (int r = 0;; ++r) {
    if (r == 0) {
      // the first tile: nothing to write out
      for (int kt = 0; kt < nkt; ++kt, ++gt) {
        const bool more = gt + 2 < total;           // K tile gt + 2 exists: request it, leave it in flight
        BLH_KTILE(boff, more, roff,
                  if (more) BLH_WAIT_VM(6); else BLH_WAIT_VM(0),
                  (void)0,
                  if (HAS_BIAS && kt == nkt - 3 && wave == 0)
                    lds_dma16_sbase<true>(bias_voff, p.bias + cn0, bias_lds))
        BLH_ROTATE();
      }
    } else {
      // a later tile: its K tiles carry the previous tile's output, window kt / per, part kt % per.  Counted wait: a
      // storing wave leaves its store of the PREVIOUS K tile in flight as well (7); the first K tile behind a
      // boundary waits for everything older (6); the last two K tiles of the workgroup request nothing.
      for (int kt = 0; kt < nkt; ++kt, ++gt) {
        const bool more = gt + 2 < total;
        const int w_ = kt >> per_shift, s_ = kt & (per - 1);
        BLH_KTILE(boff, more, roff,
                  if (!more) BLH_WAIT_VM(0); else if (DBG == 0 && st_wave && kt > 0) BLH_WAIT_VM(7); else BLH_WAIT_VM(6),
                  if (s_ == 0) { BLH_WIN_WRITE(w_) },
                  BLH_WIN_STORE(w_, s_)
                  if (HAS_BIAS && kt == nkt - 3 && wave == 0)
                    lds_dma16_sbase<true>(bias_voff, p.bias + cn0, bias_lds))
        BLH_ROTATE();
      }
    }
    if (r == ntw - 1) break;

    // ---- tile boundary: bias, BatchNorm partials, pack; the ring keeps filling ---------------------------
    if (wr == 0) BLH_BAR();                      // close the stagger (waves 0-3 wait for 4-7's last MFMA block)
    if constexpr (HAS_BIAS) {
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float bv = bias_slot[q * 128 + wc * 32 + j * 16 + c16];
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) acc[0][q][i][j][rr] += bv;
        }
    }
    if constexpr (EPI == EPI_BIAS_STATS) tile_stats_256<1>(acc, p, stat_xchg, cm0, cn0, ctm);
    {
      typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
              const bf2 v = {(__bf16)acc[0][q][i][j][2 * h], (__bf16)acc[0][q][i][j][2 * h + 1]};
              pk[q][i][j][h] = *reinterpret_cast<const uint32_t*>(&v);
              acc[0][q][i][j][2 * h] = 0.f; acc[0][q][i][j][2 * h + 1] = 0.f;
            }
    }
    dm0 = cm0; dn0 = cn0;
    ctile = xcd_remap(b + (r + 1) * G, tiles_total);
    ctm = __builtin_amdgcn_readfirstlane(ctile / tiles_n);
    cm0 = ctm * BM; cn0 = (ctile - ctm * tiles_n) * BN;
    if (wr == 1) BLH_BAR();                      // re-open the stagger
  }
  if (wr == 0) BLH_BAR();                        // the barrier waves 4-7 took at the start

#undef BLH_KTILE
#undef BLH_STAGE_1
#undef BLH_STAGE_2
#undef BLH_LOAD_A
#undef BLH_LOAD_B
#undef BLH_MFMA
#undef BLH_BAR
#undef BLH_WAIT_VM
#undef BLH_SB
#undef BLH_ROTATE
#undef BLH_WIN_WRITE
#undef BLH_WIN_STORE
  __syncthreads();
  gemm_epilogue_256<EPI, true, 1>(acc, p, C, smem, cm0, cn0, ctm);
}

}  // namespace blh
