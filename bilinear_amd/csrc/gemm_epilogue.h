// Shared epilogue of the MFMA GEMM kernels (fp32 accumulators of 32x32 MFMA tiles; the C/D
// register layout is the same for the f32 and bf16 MFMA shapes on gfx950):
//   EPI_STORE / EPI_BIAS / EPI_ADD, EPI_BIAS_STATS (per-128-row-tile BatchNorm partials),
//   EPI_BN_RELU (eval-mode heavy_linear in one kernel: bias, BatchNorm with running statistics,
//   ReLU, optional block skip).
//   See gemm_f32_ring.h for the contractions it serves.
#pragma once
#include "common.h"

namespace blh {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Workgroup barrier that orders LDS traffic only (a __syncthreads() would also wait for the
// output stores that are still draining).
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

template <int BM, int BN, int WM, int WN, int EPI>
__device__ inline void gemm_epilogue(f32x16 (&acc)[BM / WM / 32][BN / WN / 32], const GemmParams& p,
                                     float* __restrict__ C, float* smem, int m0, int n0, int tile_m,
                                     bool is_cons) {
  constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int h = lane >> 5, lc = lane & 31;
  const int row_w = m0 + wm * (TM * 32) + 4 * h;   // + tm*32 + (r&3) + 8*(r>>2)
  const int col_w = n0 + wn * (TN * 32) + lc;      // + tn*32

  if (EPI == EPI_BN_RELU) {
    // same operations, in the same order, as bn_apply_kernel<false> on Z = acc + bias:
    // sc = gamma / sqrt(var + eps); sh = beta - mean * sc; a = max(fma(z, sc, sh), 0) (+ skip)
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      const int col = col_w + jn * 32;
      const bool okc = col < p.N;
      const float bv = okc ? p.bias[col] : 0.f;
      const float sc = okc ? p.bn_gamma[col] * (1.0f / sqrtf(p.bn_var[col] + 1e-5f)) : 0.f;
      const float sh = okc ? p.bn_beta[col] - p.bn_mean[col] * sc : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][jn][r] = fmaxf(fmaf(acc[i][jn][r] + bv, sc, sh), 0.f);
    }
  }
  if (EPI == EPI_BIAS || EPI == EPI_BIAS_STATS) {
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      const int col = col_w + jn * 32;
      const float bv = (col < p.N) ? p.bias[col] : 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][jn][r] += bv;
    }
  }

  // the output stores go first: they drain while the statistics below are reduced
  if (EPI == EPI_ADD || EPI == EPI_BN_RELU) {
    // all addend loads are issued before the first add: one memory latency for the tile instead
    // of one per element (the add + store chain otherwise serialises on every load)
    float add[TM][TN][16];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row_w + i * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
          const int col = col_w + jn * 32;
          add[i][jn][r] = (p.addend && row < p.M && col < p.N) ? p.addend[(int64_t)row * p.ldadd + col] : 0.f;
        }
      }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int jn = 0; jn < TN; ++jn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][jn][r] += add[i][jn][r];
  }
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = row_w + i * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) {
        const int col = col_w + jn * 32;
        if (row < p.M && col < p.N && is_cons) C[(int64_t)row * p.ldc + col] = acc[i][jn][r];
      }
    }

  if (EPI == EPI_STORE_SQ) {
    // one sum of squares of the tile per workgroup (fixed order: lanes, then waves)
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = row_w + i * 32 + (r & 3) + 8 * (r >> 2);
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) {
          const int col = col_w + jn * 32;
          if (row < p.M && col < p.N && is_cons) s += (double)acc[i][jn][r] * (double)acc[i][jn][r];
        }
      }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    double* red = reinterpret_cast<double*>(smem);   // (stage buffers are dead: the barrier closed the main loop)
    if (lane == 0) red[wave] = s;
    lds_barrier();
    if (tid == 0) {
      double t = 0.0;
      for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += red[w];
      p.sq_part[(int64_t)blockIdx.z * gridDim.x + blockIdx.x] = t;
    }
  }
  if (EPI == EPI_BIAS_STATS) {
    // Per-tile column statistics in the shifted (Welford/Chan) form: tile mean and
    // M2 = sum (z - tile_mean)^2, merged across tiles by bn_fwd_finalize.  Avoids the
    // cancellation of sum(z^2) - sum(z)^2/n at large batch (SURVEY.md hazard H1).
    float* red = smem;   // [WM][BN]; stage buffers are dead (barrier closed the main loop)
    const int cnt = min(BM, p.M - m0);
    float mean[TN];
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row_w + i * 32 + (r & 3) + 8 * (r >> 2);
          if (row < p.M) s += acc[i][jn][r];
        }
      s += __shfl_xor(s, 32);
      if (h == 0 && is_cons) red[wm * BN + wn * (TN * 32) + jn * 32 + lc] = s;
    }
    lds_barrier();
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      float t = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) t += red[w * BN + wn * (TN * 32) + jn * 32 + lc];
      mean[jn] = t / (float)cnt;
    }
    lds_barrier();
#pragma unroll
    for (int jn = 0; jn < TN; ++jn) {
      float s = 0.f;
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = row_w + i * 32 + (r & 3) + 8 * (r >> 2);
          const float dlt = acc[i][jn][r] - mean[jn];
          if (row < p.M) s += dlt * dlt;
        }
      s += __shfl_xor(s, 32);
      if (h == 0 && is_cons) red[wm * BN + wn * (TN * 32) + jn * 32 + lc] = s;
    }
    lds_barrier();
    if (wm == 0 && h == 0 && is_cons) {
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) {
        const int col = col_w + jn * 32;
        float m2 = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) m2 += red[w * BN + wn * (TN * 32) + jn * 32 + lc];
        if (col < p.N) {
          p.stat_part[((int64_t)tile_m * 2 + 0) * p.N + col] = mean[jn];
          p.stat_part[((int64_t)tile_m * 2 + 1) * p.N + col] = m2;
        }
      }
    }
  }
}

}  // namespace blh
