// Forward stage of the bf16-storage path as ONE launch: Linear + BatchNorm (batch statistics) + ReLU + Dropout
// (+ block skip), i.e. /root/reference/model/bilinear.py:9-12,34-38 — the epilogue of the big-tile GEMM kernels
// behind a grid-wide barrier ("apply in the epilogue").
//
// Unfused, a stage is three launches: the GEMM writes Z and per-row-tile statistics, bn_fwd_finalize merges them
// (a 5 us kernel that is nothing but latency), bn_apply_h2 reads Z (+ skip) back and writes the activation A and
// the dropout keep bits.  Here every workgroup keeps its tile in registers: it publishes the tile statistics,
// waits until all workgroups of the launch have (they are all resident: one workgroup per CU, the host only
// takes this path when the grid has at most as many workgroups as the device has CUs), merges the statistics of
// its own 256 columns exactly as bn_fwd_finalize does (same order of operations in fp64: the saved statistics,
// scale / shift and running statistics are bit-identical), and then writes Z, A and the keep bits from the tile it
// still holds.  What it saves per stage: the finalize launch, two launch boundaries and apply's read of Z.
//
// The tile is rounded to bf16 and packed in row pairs as soon as its statistics exist (Z IS stored as bf16 and
// the unfused apply normalises the stored values: normalising bf16(z) here keeps A bit-identical), which frees the
// accumulator registers for the skip rows and halves the LDS traffic of the staging (gemm_bf16s_256.h,
// gemm_epilogue_256_bnbwd_packed, has the same staging).
#pragma once
#include "philox.h"

namespace blh {

// One grid-wide barrier per launch on two words in device memory (arrivals, generation), zeroed once when the
// context is created.  Sense reversal: the last arriver resets the count and bumps the generation, which the
// others poll (relaxed, with s_sleep); a third word counts timeouts (a spin is bounded: a launch whose
// workgroups are not all resident must not hang the device — its results are then wrong and the host can see
// why).  Caller: every wave has drained its stores (s_waitcnt vmcnt(0)) and the workgroup has passed a barrier.
__device__ inline void grid_barrier_once(uint32_t* bar, uint32_t nwg) {
  if (threadIdx.x == 0) {
    const uint32_t gen = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t old = __hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (old == nwg - 1) {
      __hip_atomic_store(&bar[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_fetch_add(&bar[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      int spins = 0;
      while (__hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gen) {
        __builtin_amdgcn_s_sleep(2);
        if (++spins > (1 << 22)) {      // ~0.3 s: give up instead of hanging
          __hip_atomic_fetch_add(&bar[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __syncthreads();
}

template <int NQM>
__device__ inline void gemm_epilogue_256_bnfwd(f32x4 (&acc)[NQM][2][4][2], const GemmParamsH& p, float* smem,
                                               int m0, int n0, int tile_m, int tile_n) {
  const BnFwdParams& f = p.fwd;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 2, wc = wave & 3;
  const int g = lane >> 4, c16 = lane & 15;
  // ---- 1. bias, tile statistics (from the fp32 values), then round + pack -------------------------------
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
  tile_stats_256<NQM>(acc, p, smem, m0, n0, tile_m);
  uint32_t pk[NQM][2][4][2][2];          // rows (4 g + 2 h, + 1) of an MFMA tile, one column
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
  // (the cumulative-average factor reads the batch counter: before the barrier, tile (0, 0) bumps it after)
  const int64_t nbt_before = f.nbt ? f.nbt[0] : 0;
  // ---- 2. publish the partials, wait for every workgroup of the launch ------------------------------------
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  grid_barrier_once(f.bar, gridDim.x);
  if (tile_m == 0 && tile_n == 0 && tid == 0 && f.nbt) f.nbt[0] = nbt_before + 1;
  // ---- 3. batch statistics of this tile's 256 columns: the arithmetic of bn_fwd_finalize_kernel ----------
  const int tiles = (p.M + f.tile_rows - 1) / f.tile_rows;
  double* red = reinterpret_cast<double*>(smem);                 // [8][64]
  float* scs = smem + 2 * 8 * 64;                                // [256] scale, then [256] shift
  float* shs = scs + 256;
  {
    const int cl = tid & 63, sl = tid >> 6;
    constexpr int UF = 16;                                       // tiles <= 128 (host)
#pragma unroll 1
    for (int rnd = 0; rnd < 4; ++rnd) {
      const int lc = rnd * 64 + cl, colg = n0 + lc;
      float mu[UF], m2t[UF];
#pragma unroll
      for (int u = 0; u < UF; ++u) {
        const int64_t tt = min(sl + 8 * u, tiles - 1);
        mu[u] = p.stat_part[(tt * 2 + 0) * p.N + colg];
        m2t[u] = p.stat_part[(tt * 2 + 1) * p.N + colg];
      }
      double acc1 = 0.0;
#pragma unroll
      for (int u = 0; u < UF; ++u) {
        const int t = sl + 8 * u;
        const double n = (double)min((int64_t)f.tile_rows, (int64_t)p.M - (int64_t)t * f.tile_rows);
        if (t < tiles) acc1 += n * (double)mu[u];
      }
      red[sl * 64 + cl] = acc1;
      __syncthreads();
      double mean1 = 0.0;
#pragma unroll
      for (int s8 = 0; s8 < 8; ++s8) mean1 += red[s8 * 64 + cl];
      mean1 /= (double)p.M;
      __syncthreads();
      double acc2 = 0.0;
#pragma unroll
      for (int u = 0; u < UF; ++u) {
        const int t = sl + 8 * u;
        const double n = (double)min((int64_t)f.tile_rows, (int64_t)p.M - (int64_t)t * f.tile_rows);
        const double d = (double)mu[u] - mean1;
        if (t < tiles) acc2 += (double)m2t[u] + n * d * d;
      }
      red[sl * 64 + cl] = acc2;
      __syncthreads();
      if (sl == 0) {
        double m2 = 0.0;
#pragma unroll
        for (int s8 = 0; s8 < 8; ++s8) m2 += red[s8 * 64 + cl];
        const double var = m2 / (double)p.M;
        const float invstd = (float)(1.0 / sqrt(var + (double)1e-5f));     // (BN_EPS of elementwise.hip, a float)
        const float mu1 = (float)mean1;
        const float sc = f.gamma[colg] * invstd;
        const float sh = f.beta[colg] - mu1 * sc;
        scs[lc] = sc;
        shs[lc] = sh;
        if (tile_m == 0) {        // one workgroup per column tile writes the saved / running statistics
          f.saved[0 * (int64_t)p.N + colg] = mu1;
          f.saved[1 * (int64_t)p.N + colg] = invstd;
          f.saved[2 * (int64_t)p.N + colg] = sc;
          f.saved[3 * (int64_t)p.N + colg] = sh;
          const double fac = (f.momentum >= 0.f) ? (double)f.momentum : 1.0 / (double)(nbt_before + 1);
          const double unbiased = m2 / (double)(p.M > 1 ? p.M - 1 : 1);
          f.running_mean[colg] = (float)((1.0 - fac) * (double)f.running_mean[colg] + fac * mean1);
          f.running_var[colg] = (float)((1.0 - fac) * (double)f.running_var[colg] + fac * unbiased);
        }
      }
      __syncthreads();
    }
  }
  // ---- 4. Z, A, keep bits from the packed tile ----------------------------------------------------------
  constexpr int SPW = 260;
  uint32_t* stg = reinterpret_cast<uint32_t*>(smem);
  const int rg0 = tid >> 5, ch = tid & 31;          // 8-row group of the 128-row half, 8-column chunk
  const int lcol = ch * 8, col = n0 + lcol;
  const int W8 = p.N >> 3;
  float sc[8], sh[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { sc[c] = scs[lcol + c]; sh[c] = shs[lcol + c]; }
  __syncthreads();                                  // (scs / shs lie inside the staging area)
  bf16_bits* Z = reinterpret_cast<bf16_bits*>(p.C);
#pragma unroll
  for (int qm = 0; qm < NQM; ++qm) {
    const int R0 = m0 + qm * 128 + 8 * rg0;         // first of this thread's 8 rows (two items of 4)
    // skip rows of both items, requested in front of the staging
    uint4 kq[2][4];
    if (f.skip) {
#pragma unroll
      for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t r = min(R0 + 4 * it + j, p.M - 1);
          kq[it][j] = *reinterpret_cast<const uint4*>(f.skip + r * f.ldskip + col);
        }
    }
    // keep words of the two 4-row groups (bn_bf16.hip, keep_words: the same bits)
    uint32_t kw[2] = {0u, 0u};
    if (f.drop.keep) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int64_t r = R0 + 4 * it + j;
          if (r < p.M) {
            const uint2 k = *reinterpret_cast<const uint2*>(f.drop.keep + r * (int64_t)p.N + col);
            uint32_t b = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              b |= ((k.x >> (8 * c)) & 0xFFu) ? (1u << c) : 0u;
              b |= ((k.y >> (8 * c)) & 0xFFu) ? (16u << c) : 0u;
            }
            word |= b << (8 * j);
          }
        }
        kw[it] = word;
      }
    } else {
      const int64_t base = (int64_t)(R0 & ~31);
      const int w = (R0 & 31) >> 3;
      const Philox128 p0 = dropout_patch(f.drop.seed, dropout_step(f.drop), f.drop.layer, base + f.drop.row_offset, col);
      const Philox128 p1 = dropout_patch(f.drop.seed, dropout_step(f.drop), f.drop.layer, base + f.drop.row_offset, col + 4);
      const uint32_t a = w == 0 ? p0.w[0] : (w == 1 ? p0.w[1] : (w == 2 ? p0.w[2] : p0.w[3]));
      const uint32_t b = w == 0 ? p1.w[0] : (w == 1 ? p1.w[1] : (w == 2 ? p1.w[2] : p1.w[3]));
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        uint32_t word = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int n = 4 * it + j;
          word |= (((a >> (4 * n)) & 0xFu) | (((b >> (4 * n)) & 0xFu) << 4)) << (8 * j);
        }
        kw[it] = word;
      }
    }
    // stage the 128 rows of this quadrant row as row-pair words
#pragma unroll
    for (int qn = 0; qn < 2; ++qn)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            stg[(wr * 32 + i * 8 + 2 * g + h) * SPW + qn * 128 + wc * 32 + j * 16 + c16] = pk[qm][qn][i][j][h];
    __syncthreads();
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int rgi = 2 * rg0 + it;                  // 4-row group of the half
      const int row0 = m0 + qm * 128 + 4 * rgi;
      if (row0 < p.M && f.keepbits) f.keepbits[(int64_t)(row0 >> 2) * W8 + (col >> 3)] = kw[it];
#pragma unroll
      for (int hp = 0; hp < 2; ++hp) {
        const uint4 w0 = *reinterpret_cast<const uint4*>(stg + (2 * rgi + hp) * SPW + lcol);
        const uint4 w1 = *reinterpret_cast<const uint4*>(stg + (2 * rgi + hp) * SPW + lcol + 4);
        const uint32_t w[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
        for (int e = 0; e < 2; ++e) {
          const int j = 2 * hp + e;
          const int row = row0 + j;
          if (row >= p.M) continue;
          uint4 zo;                                  // this row of Z: 8 bf16
          if (e == 0) {
            zo.x = (w[0] & 0xffffu) | (w[1] << 16); zo.y = (w[2] & 0xffffu) | (w[3] << 16);
            zo.z = (w[4] & 0xffffu) | (w[5] << 16); zo.w = (w[6] & 0xffffu) | (w[7] << 16);
          } else {
            zo.x = (w[0] >> 16) | (w[1] & 0xffff0000u); zo.y = (w[2] >> 16) | (w[3] & 0xffff0000u);
            zo.z = (w[4] >> 16) | (w[5] & 0xffff0000u); zo.w = (w[6] >> 16) | (w[7] & 0xffff0000u);
          }
          *reinterpret_cast<uint4*>(Z + (int64_t)row * p.ldc + col) = zo;
          const uint32_t bits = kw[it] >> (8 * j);
          float a[8];
#pragma unroll
          for (int c = 0; c < 8; ++c) {
            const float z = e == 0 ? __uint_as_float(w[c] << 16) : __uint_as_float(w[c] & 0xffff0000u);
            const float y = fmaxf(fmaf(z, sc[c], sh[c]), 0.f);
            a[c] = ((bits >> c) & 1u) ? y * 2.f : 0.f;
          }
          if (f.skip) {
            const uint4 kk = kq[it][j];
            a[0] += __uint_as_float(kk.x << 16); a[1] += __uint_as_float(kk.x & 0xffff0000u);
            a[2] += __uint_as_float(kk.y << 16); a[3] += __uint_as_float(kk.y & 0xffff0000u);
            a[4] += __uint_as_float(kk.z << 16); a[5] += __uint_as_float(kk.z & 0xffff0000u);
            a[6] += __uint_as_float(kk.w << 16); a[7] += __uint_as_float(kk.w & 0xffff0000u);
          }
          uint4 o;
          o.x = (uint32_t)f32_to_bf16(a[0]) | ((uint32_t)f32_to_bf16(a[1]) << 16);
          o.y = (uint32_t)f32_to_bf16(a[2]) | ((uint32_t)f32_to_bf16(a[3]) << 16);
          o.z = (uint32_t)f32_to_bf16(a[4]) | ((uint32_t)f32_to_bf16(a[5]) << 16);
          o.w = (uint32_t)f32_to_bf16(a[6]) | ((uint32_t)f32_to_bf16(a[7]) << 16);
          *reinterpret_cast<uint4*>(f.A + (int64_t)row * f.lda_out + col) = o;
        }
      }
    }
    __syncthreads();
  }
}

}  // namespace blh
