// BatchNorm forward finalize, shared by bn_fwd_finalize_kernel (elementwise.hip) and by the GEMM
// epilogue (gemm_epilogue.h: the LAST workgroup of a column tile to finish merges the tile
// partials of its columns, which removes one ~5 us launch from the dependent chain of every stage).
//
// Merges per-row-tile (mean, M2) partials (Chan et al.) into the batch mean / biased variance;
// emits scale = gamma*invstd, shift = beta - mean*scale, the saved mean / invstd for backward and
// the running statistics (unbiased variance, momentum or cumulative average:
// torch.nn.BatchNorm1d semantics, /root/reference/model/bilinear.py:10,43-55).
#pragma once
#include "common.h"

namespace blh {

// NTHREADS threads handle COLS columns starting at col0: thread = (column, slice of the row tiles);
// red = [NTHREADS / COLS][COLS] doubles of LDS.  Every thread of the block must call it.
// COHERENT: the partials were written by other workgroups of the SAME kernel (agent-scope relaxed
// atomic stores, i.e. written through the non-coherent per-XCD L2): read them with agent-scope
// atomic loads, which bypass stale cache lines; no cache invalidate is needed.
template <bool COHERENT>
__device__ __forceinline__ float ld_part(const float* p) {
  if (COHERENT) return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return *p;
}

template <int NTHREADS, int COLS, bool COHERENT = false>
__device__ __forceinline__ void bn_finalize_cols(const float* __restrict__ part, int tiles, int tile_rows,
                                                 int64_t batch, int W, int col0, const BnFin& f,
                                                 double* red) {
  constexpr int NS = NTHREADS / COLS;   // slices
  constexpr int U = 8;                  // tile partials in flight per thread
  const int cl = threadIdx.x % COLS, sl = threadIdx.x / COLS;
  const int col = col0 + cl;
  const bool ok = col < W;
  const int cc = ok ? col : 0;
  double acc = 0.0;
  for (int t0 = sl; t0 < tiles; t0 += NS * U) {
    float mu[U];
#pragma unroll
    for (int u = 0; u < U; ++u) mu[u] = ld_part<COHERENT>(part + ((int64_t)min(t0 + NS * u, tiles - 1) * 2 + 0) * W + cc);
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + NS * u;
      const double n = (double)min((int64_t)tile_rows, batch - (int64_t)t * tile_rows);
      if (t < tiles) acc += n * (double)mu[u];
    }
  }
  red[sl * COLS + cl] = acc;
  __syncthreads();
  double mean = 0.0;
#pragma unroll
  for (int s = 0; s < NS; ++s) mean += red[s * COLS + cl];
  mean /= (double)batch;
  __syncthreads();
  acc = 0.0;
  for (int t0 = sl; t0 < tiles; t0 += NS * U) {
    float mu[U], m2t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t tt = min(t0 + NS * u, tiles - 1);
      mu[u] = ld_part<COHERENT>(part + (tt * 2 + 0) * W + cc);
      m2t[u] = ld_part<COHERENT>(part + (tt * 2 + 1) * W + cc);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + NS * u;
      const double n = (double)min((int64_t)tile_rows, batch - (int64_t)t * tile_rows);
      const double d = (double)mu[u] - mean;
      if (t < tiles) acc += (double)m2t[u] + n * d * d;
    }
  }
  red[sl * COLS + cl] = acc;
  __syncthreads();
  if (sl == 0 && ok) {
    double m2 = 0.0;
#pragma unroll
    for (int s = 0; s < NS; ++s) m2 += red[s * COLS + cl];
    const double var = m2 / (double)batch;
    const float invstd = (float)(1.0 / sqrt(var + (double)BN_EPS));
    const float mu = (float)mean;
    const float sc = f.gamma[col] * invstd;
    f.saved[col] = mu;
    f.saved[W + col] = invstd;
    f.saved[2 * W + col] = sc;
    f.saved[3 * W + col] = f.beta[col] - mu * sc;
    const double mom = (f.momentum >= 0.f) ? (double)f.momentum : 1.0 / (double)(f.nbt[0] + 1);
    const double unbiased = m2 / (double)(batch > 1 ? batch - 1 : 1);
    f.running_mean[col] = (float)((1.0 - mom) * (double)f.running_mean[col] + mom * mean);
    f.running_var[col] = (float)((1.0 - mom) * (double)f.running_var[col] + mom * unbiased);
  }
}

}  // namespace blh
