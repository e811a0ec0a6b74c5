// HBM-bound kernels of the lifter step: BatchNorm statistics / apply / backward
// fused with ReLU + Dropout + residual add, reductions, MSE, fused clip + Adam.
// (/root/reference/model/bilinear.py:10-12,38; train_bilinear.py:78-83.)
//
// Layout: every [B,W] tensor is row-major; a thread owns 4 consecutive columns
// (one float4 = 16 B per lane, a wave reads 1 KiB contiguous) and walks down the
// rows of its row-chunk, so column reductions need no cross-thread traffic and a
// Philox call (128 keep-bits = 32 rows x 4 columns) is shared by a whole patch.
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "bn_stats_dev.h"
#include "philox.h"

namespace blh {

// The BatchNorm-backward kernels run beside the weight-gradient GEMM of the previous stage (two-stream
// backward).  At equal priority the GEMM's waves win most issue slots and bn_bwd_apply takes 60 us
// instead of 10; raised to the highest wave priority it takes 37 us and the GEMM loses nothing
// measurable (it is bound by the matrix pipe, not by issue): step 1.101 -> 1.078 ms
// (profiles/r02_step_timeline.md).  Harmless when the kernel runs alone.
#define BLH_EW_PRIO() __builtin_amdgcn_s_setprio(3)

static constexpr int EW_THREADS = 256;

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// 4 consecutive elements of a [B,W] tensor stored as bf16 (gemm_dtype 4): 8 bytes per lane
typedef uint16_t bf16_bits;
__device__ __forceinline__ float4 ld4(const bf16_bits* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u),
                     __uint_as_float(v.y << 16), __uint_as_float(v.y & 0xffff0000u));
}
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  bf2 b;
  b[0] = (__bf16)lo; b[1] = (__bf16)hi;          // v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
  return *reinterpret_cast<const uint32_t*>(&b);
}
__device__ __forceinline__ void st4(bf16_bits* p, float4 v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16x2(v.x, v.y), pack_bf16x2(v.z, v.w));
}
// what a consumer of the stored tensor will read back (identity for fp32 storage)
__device__ __forceinline__ float4 as_stored(const float*, float4 v) { return v; }
__device__ __forceinline__ float4 as_stored(const bf16_bits*, float4 v) {
  const uint32_t a = pack_bf16x2(v.x, v.y), b = pack_bf16x2(v.z, v.w);
  return make_float4(__uint_as_float(a << 16), __uint_as_float(a & 0xffff0000u),
                     __uint_as_float(b << 16), __uint_as_float(b & 0xffff0000u));
}

// max |v| over the FINITE values of a float4, folded into m.  Inf / NaN elements do not take
// part: the fp16-split GEMM derives its power-of-two scale from this maximum, and one overflowed
// element must not push every finite value of the tensor out of fp16's range (the non-finite
// element itself stays non-finite through the split and poisons only its own row / column, as
// in the exact kernel).
__device__ __forceinline__ float finite_abs(float x) {
  const float a = fabsf(x);
  return a <= 3.402823466e+38f ? a : 0.f;   // false for Inf and NaN
}
__device__ __forceinline__ float amax4(float m, float4 v) {
  return fmaxf(fmaxf(m, fmaxf(finite_abs(v.x), finite_abs(v.y))),
               fmaxf(finite_abs(v.z), finite_abs(v.w)));
}
// one partial per wave (all 64 lanes must be active): part[wave_global]
__device__ __forceinline__ void wave_amax_store(float m, float* __restrict__ part) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0)
    part[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (EW_THREADS / 64) + (threadIdx.x >> 6)] = m;
}

struct DropState {
  Philox128 patch;
};

// keep nibble (bit i = keep column col+i) for row r
__device__ __forceinline__ uint32_t keep_nibble(const DropoutSrc& d, DropState& st, int64_t r,
                                                int col, int W, bool first) {
  if (d.keep) {
    const uchar4 k = *reinterpret_cast<const uchar4*>(d.keep + r * (int64_t)W + col);
    return (k.x ? 1u : 0u) | (k.y ? 2u : 0u) | (k.z ? 4u : 0u) | (k.w ? 8u : 0u);
  }
  const int64_t grow = r + d.row_offset;
  if (first || (grow & 31) == 0) st.patch = dropout_patch(d.seed, dropout_step(d), d.layer, grow, col);
  return patch_nibble(st.patch, (int)(grow & 31));
}

// Sum of in[s*ld + col] over s = first, first+stride, ... < S in fp64.  The loads of a batch of
// U rows are issued together (clamped row index, select AFTER the load): these tiny reductions
// are latency-bound, and a loop that waits for each load in turn pays one L2/Infinity-Cache round
// trip per row (the round-1 colreduce: 16 dependent round trips = 11 us for 1 MB of partials).
template <int U>
__device__ __forceinline__ double strided_colsum(const float* __restrict__ in, int first, int stride,
                                                 int S, int64_t ld, int col) {
  double acc = 0.0;
  for (int s0 = first; s0 < S; s0 += U * stride) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = in[(int64_t)min(s0 + u * stride, S - 1) * ld + col];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += (s0 + u * stride < S) ? (double)v[u] : 0.0;
  }
  return acc;
}

// ---------------------------------------------------------------------------
// forward BN finalize: merge per-tile (mean, M2) partials (Chan et al.) into the
// batch mean / biased variance; emit scale = gamma*invstd, shift = beta - mean*scale;
// update running stats (unbiased variance, PyTorch BatchNorm1d semantics).
// block = 32 columns x 8 tile-slices.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void bn_fwd_finalize_kernel(
    const float* __restrict__ part, int tiles, int tile_rows, int64_t batch, int W,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean,
    float* running_var, const int64_t* nbt, float momentum, float* saved_mean,
    float* saved_invstd, float* scale, float* shift) {
  __shared__ double red[8][32];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + cl;
  const bool ok = col < W;
  // both passes read every tile partial of this column with the loads of 8 tiles in flight
  constexpr int U = 8;
  const int cc = ok ? col : 0;
  BnColumnIn pre{};
  if (sl == 0) pre = bn_finalize_prefetch(cc, gamma, beta, running_mean, running_var);
  auto fast = [&](auto uc) {
    // every partial this thread needs — (mean, M2) of up to UF tiles — is requested at once and
    // kept in registers for the second step: one memory round trip instead of two (this kernel is
    // nothing but latency).  UF = 8 covers 64 tiles (batch <= 8192 at 128-row tiles), 16 covers 128.
    constexpr int UF = decltype(uc)::value;
    float mu[UF], m2t[UF];
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const int64_t tt = min(sl + 8 * u, tiles - 1);
      mu[u] = part[(tt * 2 + 0) * W + cc];
      m2t[u] = part[(tt * 2 + 1) * W + cc];
    }
    double acc1 = 0.0;
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const int t = sl + 8 * u;
      const double n = (double)min((int64_t)tile_rows, batch - (int64_t)t * tile_rows);
      if (t < tiles) acc1 += n * (double)mu[u];
    }
    red[sl][cl] = acc1;
    __syncthreads();
    double mean1 = 0.0;
#pragma unroll
    for (int s = 0; s < 8; ++s) mean1 += red[s][cl];
    mean1 /= (double)batch;
    __syncthreads();
    double acc2 = 0.0;
#pragma unroll
    for (int u = 0; u < UF; ++u) {
      const int t = sl + 8 * u;
      const double n = (double)min((int64_t)tile_rows, batch - (int64_t)t * tile_rows);
      const double d = (double)mu[u] - mean1;
      if (t < tiles) acc2 += (double)m2t[u] + n * d * d;
    }
    red[sl][cl] = acc2;
    __syncthreads();
    if (sl == 0 && ok) {
      double m2 = 0.0;
#pragma unroll
      for (int s = 0; s < 8; ++s) m2 += red[s][cl];
      bn_finalize_store(mean1, m2, batch, col, pre, running_mean, running_var, nbt, momentum,
                        saved_mean, saved_invstd, scale, shift);
    }
  };
  if (tiles <= 64) { fast(std::integral_constant<int, 8>{}); return; }
  if (tiles <= 128) { fast(std::integral_constant<int, 16>{});
    return;
  }
  double acc = 0.0;
  for (int t0 = sl; t0 < tiles; t0 += 8 * U) {
    float mu[U];
#pragma unroll
    for (int u = 0; u < U; ++u) mu[u] = part[((int64_t)min(t0 + 8 * u, tiles - 1) * 2 + 0) * W + cc];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + 8 * u;
      const double n = (double)min((int64_t)tile_rows, batch - (int64_t)t * tile_rows);
      if (t < tiles) acc += n * (double)mu[u];
    }
  }
  red[sl][cl] = acc;
  __syncthreads();
  double mean = 0.0;
#pragma unroll
  for (int s = 0; s < 8; ++s) mean += red[s][cl];
  mean /= (double)batch;
  __syncthreads();
  acc = 0.0;
  for (int t0 = sl; t0 < tiles; t0 += 8 * U) {
    float mu[U], m2t[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t tt = min(t0 + 8 * u, tiles - 1);
      mu[u] = part[(tt * 2 + 0) * W + cc];
      m2t[u] = part[(tt * 2 + 1) * W + cc];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int t = t0 + 8 * u;
      const double n = (double)min((int64_t)tile_rows, batch - (int64_t)t * tile_rows);
      const double d = (double)mu[u] - mean;
      if (t < tiles) acc += (double)m2t[u] + n * d * d;
    }
  }
  red[sl][cl] = acc;
  __syncthreads();
  if (sl == 0 && ok) {
    double m2 = 0.0;
#pragma unroll
    for (int s = 0; s < 8; ++s) m2 += red[s][cl];
    bn_finalize_store(mean, m2, batch, col, gamma, beta, running_mean, running_var, nbt, momentum,
                      saved_mean, saved_invstd, scale, shift);
  }
}

int launch_bn_fwd_finalize(hipStream_t s, const float* stat_part, int tiles, int tile_rows,
                           int64_t batch, int W, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, int64_t* nbt,
                           float momentum, float* saved_mean, float* saved_invstd, float* scale,
                           float* shift) {
  hipLaunchKernelGGL(bn_fwd_finalize_kernel, dim3((unsigned)ceil_div(W, 32)), dim3(256), 0, s,
                     stat_part, tiles, tile_rows, batch, W, gamma, beta, running_mean,
                     running_var, nbt, momentum, saved_mean, saved_invstd, scale, shift);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

// (The streaming BatchNorm-apply / BatchNorm-backward kernels live in bn_f32.hip and bn_bf16.hip.)


// ---------------------------------------------------------------------------
// out[c] = sum_s in[s][c], c < ncols.  block = 32 columns x 8 slices, fp64 sums.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colreduce_kernel(const float* __restrict__ in, int S,
                                                        int64_t ld, int ncols,
                                                        float* __restrict__ out) {
  BLH_EW_PRIO();
  __shared__ double red[8][32];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + cl;
  const double acc = strided_colsum<8>(in, sl, 8, S, ld, min(col, ncols - 1));
  red[sl][cl] = acc;
  __syncthreads();
  if (sl == 0 && col < ncols) {
    double t = 0.0;
#pragma unroll
    for (int s = 0; s < 8; ++s) t += red[s][cl];
    out[col] = (float)t;
  }
}

int launch_colreduce(hipStream_t s, const float* in, int S, int64_t ld, int ncols, float* out) {
  hipLaunchKernelGGL(colreduce_kernel, dim3((unsigned)ceil_div(ncols, 32)), dim3(256), 0, s, in,
                     S, ld, ncols, out);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

// all stages' Linear-bias gradients in one launch: out[stage][c] = sum_s part[stage][s][c];
// an optional extra entry (the decode bias) has its own partial array, row count and width
struct BiasOffsets {
  int64_t off[32];
  const float* extra_part; int extra_S; int extra_cols; int64_t extra_off;
};
__global__ __launch_bounds__(256) void bias_colreduce_kernel(const float* __restrict__ part,
                                                             int64_t stage_stride, int S, int W,
                                                             int num_stages,
                                                             float* __restrict__ grads,
                                                             BiasOffsets offs, double* __restrict__ sq) {
  __shared__ double red[8][32];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + cl;
  const bool extra = (int)blockIdx.y == num_stages;
  const float* in = extra ? offs.extra_part : part + (int64_t)blockIdx.y * stage_stride;
  const int rows = extra ? offs.extra_S : S, cols = extra ? offs.extra_cols : W;
  const int64_t out = extra ? offs.extra_off : offs.off[blockIdx.y];
  const double acc = strided_colsum<8>(in, sl, 8, rows, cols, min(col, cols - 1));
  red[sl][cl] = acc;
  __syncthreads();
  float written = 0.f;
  if (sl == 0 && col < cols) {
    double t = 0.0;
#pragma unroll
    for (int s = 0; s < 8; ++s) t += red[s][cl];
    written = (float)t;
    grads[out + col] = written;
  }
  if (sq) {   // sum of squares of the 32 gradients this block wrote (wave 0 holds them)
    if (sl < 2) {
      double q = (double)written * (double)written;
#pragma unroll
      for (int o = 16; o >= 1; o >>= 1) q += __shfl_xor(q, o);
      if (threadIdx.x == 0) sq[blockIdx.y * gridDim.x + blockIdx.x] = q;
    }
  }
}

int bias_colreduce_blocks(int W, int num_stages, bool extra) {
  return (int)ceil_div(W, 32) * (num_stages + (extra ? 1 : 0));
}

int launch_bias_colreduce(hipStream_t s, const float* part, int64_t stage_stride, int S, int W,
                          int num_stages, const int64_t* out_offsets, float* grads,
                          const float* extra_part, int extra_S, int extra_cols,
                          int64_t extra_off, double* sq) {
  if (num_stages > 32) return BLH_ERR_SHAPE;
  BiasOffsets o{};
  for (int i = 0; i < num_stages; ++i) o.off[i] = out_offsets[i];
  o.extra_part = extra_part; o.extra_S = extra_S; o.extra_cols = extra_cols; o.extra_off = extra_off;
  hipLaunchKernelGGL(bias_colreduce_kernel,
                     dim3((unsigned)ceil_div(W, 32), num_stages + (extra_part ? 1 : 0)), dim3(256),
                     0, s, part, stage_stride, S, W, num_stages, grads, o, sq);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}


// ---------------------------------------------------------------------------
// out = sum of `splits` slabs (split-K partial products of the wgrad GEMM)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ slabs,
                                                        int64_t count, int splits,
                                                        const float* addend,
                                                        float* out) {
  const int64_t n4 = count >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 a = ld4(slabs + i * 4);
    int s = 1;
    for (; s + 3 <= splits; s += 3) {   // three further slabs per round trip (4 splits = one)
      const float4 b0 = ld4(slabs + (int64_t)(s + 0) * count + i * 4);
      const float4 b1 = ld4(slabs + (int64_t)(s + 1) * count + i * 4);
      const float4 b2 = ld4(slabs + (int64_t)(s + 2) * count + i * 4);
      a.x += b0.x; a.y += b0.y; a.z += b0.z; a.w += b0.w;
      a.x += b1.x; a.y += b1.y; a.z += b1.z; a.w += b1.w;
      a.x += b2.x; a.y += b2.y; a.z += b2.z; a.w += b2.w;
    }
    for (; s < splits; ++s) {
      const float4 b = ld4(slabs + (int64_t)s * count + i * 4);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    if (addend) {     // may alias out (element-wise read-then-write by the same thread)
      const float4 b = ld4(addend + i * 4);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    st4(out + i * 4, a);
  }
}

int launch_sum_slabs_add(hipStream_t s, const float* slabs, int64_t count, int splits,
                         const float* addend, float* out) {
  if (count % 4 != 0) return BLH_ERR_SHAPE;
  const int64_t blocks = std::min<int64_t>(ceil_div(count / 4, 256), 2048);
  hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)blocks), dim3(256), 0, s, slabs, count,
                     splits, addend, out);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

int launch_sum_slabs(hipStream_t s, const float* slabs, int64_t count, int splits, float* out) {
  return launch_sum_slabs_add(s, slabs, count, splits, nullptr, out);
}

// out = sum of the slabs, and one sum-of-squares partial of `out` per block (the fused fp32 step: the
// gradient norm of clip_grad_norm_ is gathered by the kernels that write the gradients).  1024 threads per
// block so that a few hundred partials cover a 1024 x 1024 weight gradient at full memory-level parallelism.
__global__ __launch_bounds__(1024) void sum_slabs_sq_kernel(const float* __restrict__ slabs, int64_t count,
                                                            int splits, float* __restrict__ out,
                                                            double* __restrict__ sq) {
  __shared__ double sh[16];
  const int64_t n4 = count >> 2;
  double acc = 0.0;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 a = ld4(slabs + i * 4);
    int s = 1;
    for (; s + 3 <= splits; s += 3) {
      const float4 b0 = ld4(slabs + (int64_t)(s + 0) * count + i * 4);
      const float4 b1 = ld4(slabs + (int64_t)(s + 1) * count + i * 4);
      const float4 b2 = ld4(slabs + (int64_t)(s + 2) * count + i * 4);
      a.x += b0.x; a.y += b0.y; a.z += b0.z; a.w += b0.w;
      a.x += b1.x; a.y += b1.y; a.z += b1.z; a.w += b1.w;
      a.x += b2.x; a.y += b2.y; a.z += b2.z; a.w += b2.w;
    }
    for (; s < splits; ++s) {
      const float4 b = ld4(slabs + (int64_t)s * count + i * 4);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    st4(out + i * 4, a);
    acc += (double)((a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += sh[w];
    sq[blockIdx.x] = t;
  }
}

int sum_slabs_sq_blocks(int64_t count, int max_blocks) {
  return (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(count / 4, 1024), max_blocks));
}

int launch_sum_slabs_sq(hipStream_t s, const float* slabs, int64_t count, int splits, float* out, double* sq,
                        int max_blocks) {
  if (count % 4 != 0 || !sq) return BLH_ERR_SHAPE;
  hipLaunchKernelGGL(sum_slabs_sq_kernel, dim3((unsigned)sum_slabs_sq_blocks(count, max_blocks)), dim3(1024), 0, s,
                     slabs, count, splits, out, sq);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

// The same for `items` independent outputs in one launch (the batched weight-gradient GEMM):
// item b sums slabs + b * slab_item_stride into out + b * out_item_stride.
__global__ __launch_bounds__(256) void sum_slabs_batched_kernel(const float* __restrict__ slabs, int64_t count,
                                                                int splits, int64_t slab_item_stride,
                                                                float* __restrict__ out, int64_t out_item_stride) {
  const float* sl = slabs + (int64_t)blockIdx.y * slab_item_stride;
  float* o = out + (int64_t)blockIdx.y * out_item_stride;
  const int64_t n4 = count >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 a = ld4(sl + i * 4);
    for (int s = 1; s < splits; ++s) {
      const float4 b = ld4(sl + (int64_t)s * count + i * 4);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    st4(o + i * 4, a);
  }
}

int launch_sum_slabs_batched(hipStream_t s, const float* slabs, int64_t count, int splits, int items,
                             int64_t slab_item_stride, float* out, int64_t out_item_stride) {
  if (count % 4 != 0 || slab_item_stride % 4 != 0 || out_item_stride % 4 != 0 || items < 1) return BLH_ERR_SHAPE;
  const int64_t blocks = std::min<int64_t>(ceil_div(count / 4, 256), std::max<int64_t>(1, 4096 / items));
  hipLaunchKernelGGL(sum_slabs_batched_kernel, dim3((unsigned)blocks, (unsigned)items), dim3(256), 0, s, slabs,
                     count, splits, slab_item_stride, out, out_item_stride);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

// ... and the gradient norm's partials on the way (bf16-storage fused step, r06): block (x, y) sums its share of item
// y's slabs and leaves the sum of squares of what it wrote, plus that of its share of every OTHER range of the gradient
// arena (stage 0, the biases / gamma / beta of the stages, the decode layer: all final when this kernel runs — the
// caller orders the bias reduction in front of it), so that no pass over the arena is left between the last gradient
// and clip + Adam.  One partial per block, sq[y * gridDim.x + x]: dense, every slot written by exactly one block.
// (The other ranges as a grid slice of their own — (items + 1) x blocks — made the launch 2304 blocks: one full round
//  of 2048 resident blocks and a straggling eighth, 21.2 us against 15.7 for the plain kernel.)
__device__ __forceinline__ int64_t sq_range_offset(const SqRanges& rg, int64_t idx) {
  int64_t off = 0;
  bool found = false;
#pragma unroll
  for (int r = 0; r < 36; ++r) {
    if (r < rg.n) {
      const int64_t n4 = rg.cnt[r] >> 2;
      if (!found && idx < n4) { found = true; off = rg.off[r] + idx * 4; }
      else if (!found) idx -= n4;
    }
  }
  return off;
}

__global__ __launch_bounds__(256) void sum_slabs_batched_sq_kernel(const float* __restrict__ slabs, int64_t count,
                                                                   int splits, int64_t slab_item_stride,
                                                                   float* __restrict__ out, int64_t out_item_stride,
                                                                   const float* __restrict__ arena, SqRanges rg,
                                                                   double* __restrict__ sq) {
  __shared__ double sh[4];
  double acc = 0.0;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  // the other ranges as ONE index space (their float4 counts laid end to end): a thread finds the range of its element
  // with the (uniform, unrolled) table walk and requests it HERE, in front of the slab loop; it is squared at the end.
  // (A loop over the ranges with a load inside was ten dependent round trips for the first blocks — 20.5 us for the
  //  kernel against 16.5 without this part; one load behind the slab loop 18.6.)
  const int64_t xfirst = ((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x;
  const int64_t xall = stride * gridDim.y;
  float4 ex = make_float4(0.f, 0.f, 0.f, 0.f);
  if (xfirst < rg.total4) ex = ld4(arena + sq_range_offset(rg, xfirst));
  {
    const float* sl = slabs + (int64_t)blockIdx.y * slab_item_stride;
    float* o = out + (int64_t)blockIdx.y * out_item_stride;
    const int64_t n4 = count >> 2;
    // (two elements per trip, all their loads first: as many requests in flight per thread as the plain kernel's
    //  larger grid has)
    int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    for (; i + stride < n4; i += 2 * stride) {
      float4 a = ld4(sl + i * 4), c = ld4(sl + (i + stride) * 4);
      for (int s = 1; s < splits; ++s) {
        const float4 b = ld4(sl + (int64_t)s * count + i * 4), e = ld4(sl + (int64_t)s * count + (i + stride) * 4);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
        c.x += e.x; c.y += e.y; c.z += e.z; c.w += e.w;
      }
      st4(o + i * 4, a);
      st4(o + (i + stride) * 4, c);
      acc += (double)((a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w));
      acc += (double)((c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w));
    }
    for (; i < n4; i += stride) {
      float4 a = ld4(sl + i * 4);
      for (int s = 1; s < splits; ++s) {
        const float4 b = ld4(sl + (int64_t)s * count + i * 4);
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
      }
      st4(o + i * 4, a);
      acc += (double)((a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w));
    }
  }
  acc += (double)((ex.x * ex.x + ex.y * ex.y) + (ex.z * ex.z + ex.w * ex.w));
  for (int64_t idx0 = xfirst + xall; idx0 < rg.total4; idx0 += xall) {     // (never taken at the shapes of the step)
    const float4 a = ld4(arena + sq_range_offset(rg, idx0));
    acc += (double)((a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w));
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) acc += __shfl_xor(acc, o);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) sq[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

int launch_sum_slabs_batched_sq(hipStream_t s, const float* slabs, int64_t count, int splits, int items,
                                int64_t slab_item_stride, float* out, int64_t out_item_stride, const float* arena,
                                const SqRanges& ranges, double* sq, int max_parts, int* nparts) {
  if (count % 4 != 0 || slab_item_stride % 4 != 0 || out_item_stride % 4 != 0 || items < 1 || !sq || !nparts)
    return BLH_ERR_SHAPE;
  SqRanges rg = ranges;
  rg.total4 = 0;
  for (int r = 0; r < rg.n; ++r) {
    if (rg.off[r] % 4 != 0 || rg.cnt[r] % 4 != 0 || rg.cnt[r] < 0) return BLH_ERR_SHAPE;
    rg.total4 += rg.cnt[r] >> 2;
  }
  // at most 2048 blocks in all (one round of resident blocks: 8 per CU), a power of two per item (even trips)
  int64_t blocks = 1;
  while (blocks * 2 <= std::min<int64_t>(ceil_div(count / 4, 256), std::max<int64_t>(1, std::min(max_parts, 2048) / items)))
    blocks *= 2;
  if (blocks * items > max_parts) return BLH_ERR_SHAPE;
  hipLaunchKernelGGL(sum_slabs_batched_sq_kernel, dim3((unsigned)blocks, (unsigned)items), dim3(256), 0, s, slabs,
                     count, splits, slab_item_stride, out, out_item_stride, arena, rg, sq);
  BLH_HIP_TRY(hipGetLastError());
  *nparts = (int)(blocks * items);
  return BLH_OK;
}

// ---------------------------------------------------------------------------
// small-batch forward: Z = sum of split-K slabs + bias (streaming, fully parallel), then the
// column (mean, M2) over all M rows as ONE statistics tile (Welford per thread, Chan merge
// across the 8 row lanes).  block = 32 columns x 8 row lanes.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sum_slabs_bias_kernel(const float* __restrict__ slabs,
                                                             int64_t count, int splits, int N,
                                                             const float* __restrict__ bias,
                                                             float* __restrict__ out) {
  const int64_t n4 = count >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 a = ld4(bias + (int)((i * 4) % N));
    int s = 0;
    for (; s + 4 <= splits; s += 4) {
      const float4 b0 = ld4(slabs + (int64_t)(s + 0) * count + i * 4);
      const float4 b1 = ld4(slabs + (int64_t)(s + 1) * count + i * 4);
      const float4 b2 = ld4(slabs + (int64_t)(s + 2) * count + i * 4);
      const float4 b3 = ld4(slabs + (int64_t)(s + 3) * count + i * 4);
      a.x += (b0.x + b1.x) + (b2.x + b3.x); a.y += (b0.y + b1.y) + (b2.y + b3.y);
      a.z += (b0.z + b1.z) + (b2.z + b3.z); a.w += (b0.w + b1.w) + (b2.w + b3.w);
    }
    for (; s < splits; ++s) {
      const float4 b = ld4(slabs + (int64_t)s * count + i * 4);
      a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    st4(out + i * 4, a);
  }
}

// Column statistics of Z in 64-row chunks (the tile form bn_fwd_finalize merges): block = 32 columns x 8 row
// lanes of 8 rows; grid (N / 32, chunks).  (Round 1's form was ONE block per 32 columns walking all rows with a
// division per element: 44 us at M = 1024, a quarter of that step.)
constexpr int FWD_FINISH_ROWS = 64;
__global__ __launch_bounds__(256) void col_stats_chunk_kernel(const float* __restrict__ Z, int64_t M, int N,
                                                              float* __restrict__ stat_part) {
  __shared__ float red[8][32];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + cl;
  const int64_t r0 = (int64_t)blockIdx.y * FWD_FINISH_ROWS;
  const int cnt = (int)min<int64_t>(FWD_FINISH_ROWS, M - r0);
  const bool okc = col < N;
  float x[8];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int r = rl * 8 + j;
    x[j] = (okc && r < cnt) ? Z[(r0 + r) * N + col] : 0.f;
    s += x[j];
  }
  red[rl][cl] = s;
  __syncthreads();
  float t = 0.f;
#pragma unroll
  for (int w = 0; w < 8; ++w) t += red[w][cl];
  const float mean = t / (float)cnt;
  __syncthreads();
  float d2 = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const float d = x[j] - mean;
    if (rl * 8 + j < cnt) d2 = fmaf(d, d, d2);
  }
  red[rl][cl] = d2;
  __syncthreads();
  if (rl == 0 && okc) {
    float m2 = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) m2 += red[w][cl];
    stat_part[((int64_t)blockIdx.y * 2 + 0) * N + col] = mean;
    stat_part[((int64_t)blockIdx.y * 2 + 1) * N + col] = m2;
  }
}

int fwd_finish_stat_rows() { return FWD_FINISH_ROWS; }

int launch_fwd_finish(hipStream_t s, const float* slabs, int splits, int64_t M, int N,
                      const float* bias, float* Z, float* stat_part) {
  const int64_t count = M * N;
  if (count % 4 != 0 || N % 4 != 0) return BLH_ERR_SHAPE;
  const int64_t blocks = std::min<int64_t>(ceil_div(count / 4, 256), 2048);
  hipLaunchKernelGGL(sum_slabs_bias_kernel, dim3((unsigned)blocks), dim3(256), 0, s, slabs, count,
                     splits, N, bias, Z);
  BLH_HIP_TRY(hipGetLastError());
  if (stat_part) {
    hipLaunchKernelGGL(col_stats_chunk_kernel, dim3((unsigned)ceil_div(N, 32), (unsigned)ceil_div(M, FWD_FINISH_ROWS)),
                       dim3(256), 0, s, Z, M, N, stat_part);
    BLH_HIP_TRY(hipGetLastError());
  }
  return BLH_OK;
}

// ---------------------------------------------------------------------------
// column sums of a narrow matrix X[rows][ld] (decode bias gradient: 48 columns)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_partial_kernel(const float* __restrict__ X,
                                                             int64_t rows, int cols, int64_t ld,
                                                             int rows_per_block,
                                                             float* __restrict__ part) {
  extern __shared__ float4 sred[];   // [slots][cg]
  const int cg = cols >> 2;
  const int slots = 256 / cg;
  const int c = threadIdx.x % cg, slot = threadIdx.x / cg;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(rows, r0 + rows_per_block);
  float4 a = make_float4(0, 0, 0, 0);
  if (slot < slots)
    for (int64_t r = r0 + slot; r < r1; r += slots) {
      const float4 v = ld4(X + r * ld + c * 4);
      a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
    }
  if (slot < slots) sred[slot * cg + c] = a;
  __syncthreads();
  if (threadIdx.x < cg) {
    float4 t = make_float4(0, 0, 0, 0);
    for (int sidx = 0; sidx < slots; ++sidx) {
      const float4 v = sred[sidx * cg + threadIdx.x];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    st4(part + (int64_t)blockIdx.x * cols + threadIdx.x * 4, t);
  }
}

int launch_colsum(hipStream_t s, const float* X, int64_t rows, int cols, int64_t ld, float* part,
                  float* out) {
  if (cols % 4 != 0 || cols > 1024 || cols <= 0) return BLH_ERR_SHAPE;
  const int rpb = 256;
  const int blocks = (int)ceil_div(rows, rpb);
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(blocks), dim3(256), 256 * sizeof(float4), s, X,
                     rows, cols, ld, rpb, part);
  BLH_HIP_TRY(hipGetLastError());
  return launch_colreduce(s, part, blocks, cols, cols, out);
}

// ---------------------------------------------------------------------------
// MSE: dpred = scale*(pred-target); per-block partial sums of squared error
// ---------------------------------------------------------------------------
__device__ __forceinline__ float block_sum_f32(float v, float* sh) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) v += __shfl_xor(v, o);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  float t = 0.f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
  return t;
}

__global__ __launch_bounds__(256) void mse_kernel(const float* __restrict__ pred,
                                                  const float* __restrict__ target, int64_t n,
                                                  float scale, float* __restrict__ dpred,
                                                  float* __restrict__ part) {
  __shared__ float sh[4];
  float acc = 0.f;
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const float4 p = ld4(pred + i * 4), t = ld4(target + i * 4);
    float4 d;
    d.x = p.x - t.x; d.y = p.y - t.y; d.z = p.z - t.z; d.w = p.w - t.w;
    acc += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
    d.x *= scale; d.y *= scale; d.z *= scale; d.w *= scale;
    st4(dpred + i * 4, d);
  }
  const float t = block_sum_f32(acc, sh);
  if (threadIdx.x == 0) part[blockIdx.x] = t;
}

int launch_mse(hipStream_t s, const float* pred, const float* target, int64_t n, float scale,
               float* dpred, float* part, int* nparts) {
  if (n % 4 != 0) return BLH_ERR_SHAPE;
  const int blocks = (int)std::min<int64_t>(ceil_div(n / 4, 256), 1024);
  hipLaunchKernelGGL(mse_kernel, dim3(blocks), dim3(256), 0, s, pred, target, n, scale, dpred,
                     part);
  BLH_HIP_TRY(hipGetLastError());
  *nparts = blocks;
  return BLH_OK;
}

__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ part, int n,
                                                            double denom, float* loss_out) {
  __shared__ double sh[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < n; i += 256) a += (double)part[i];
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_out[0] = (float)(sh[0] / denom);
}

int launch_loss_finalize(hipStream_t s, const float* part, int n, double denom, float* loss_out) {
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, s, part, n, denom, loss_out);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

// ---------------------------------------------------------------------------
// optimiser: global L2 norm partials, then fused clip + Adam over the flat arena
// ---------------------------------------------------------------------------
template <typename TG>
__global__ __launch_bounds__(256) void sumsq_kernel(const TG* __restrict__ g, int64_t count, float gscale,
                                                    double* __restrict__ part) {
  __shared__ double sh[256];
  double acc = 0.0;
  const int64_t n4 = count >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = ld4(g + i * 4);
    v.x *= gscale; v.y *= gscale; v.z *= gscale; v.w *= gscale;
    acc += (double)(v.x * v.x + v.y * v.y) + (double)(v.z * v.z + v.w * v.w);
  }
  sh[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = sh[0];
}

int launch_sumsq(hipStream_t s, const float* g, int64_t count, double* part, int* nparts) {
  if (count % 4 != 0) return BLH_ERR_SHAPE;
  const int blocks = (int)std::min<int64_t>(ceil_div(count / 4, 256 * 4), SUMSQ_MAX_PARTS);
  hipLaunchKernelGGL(sumsq_kernel<float>, dim3(blocks), dim3(256), 0, s, g, count, 1.0f, part);
  BLH_HIP_TRY(hipGetLastError());
  *nparts = blocks;
  return BLH_OK;
}

int launch_sumsq_bf16(hipStream_t s, const uint16_t* g, int64_t count, float gscale, double* part, int* nparts) {
  if (count % 4 != 0) return BLH_ERR_SHAPE;
  const int blocks = (int)std::min<int64_t>(ceil_div(count / 4, 256 * 4), SUMSQ_MAX_PARTS);
  hipLaunchKernelGGL(sumsq_kernel<bf16_bits>, dim3(blocks), dim3(256), 0, s, g, count, gscale, part);
  BLH_HIP_TRY(hipGetLastError());
  *nparts = blocks;
  return BLH_OK;
}

// block 0 of the optimiser kernel also finishes the MSE loss (sum of the partials / denom),
// which saves a 1-block launch per step
__device__ __forceinline__ void finish_loss(const LossFinish& lf, double* sh) {
  if (lf.part == nullptr || blockIdx.x != 0) return;
  double a = 0.0;
  for (int i = threadIdx.x; i < lf.n; i += 256) a += (double)lf.part[i];
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) lf.out[0] = (float)(sh[0] / lf.denom);
  __syncthreads();
}

// clip + Adam over the flat arena.  gin: the gradient as it arrives (fp32 arena, or the bf16 buckets
// of the compressed data-parallel exchange, times gscale); gout: the fp32 arena, which receives the
// clipped gradient (the reference's .grad after clip_grad_norm_); shadow (optional): the bf16 image
// of the updated parameters for the next bf16-storage forward.

template <typename TG>
__device__ __forceinline__ void clip_adam_body(float* __restrict__ p, const TG* gin, float* gout,
                                               float* __restrict__ m, float* __restrict__ v, int64_t count,
                                               const AdamConsts c, float gscale,
                                               const double* __restrict__ sumsq_part, int nparts,
                                               float* stats_out, const LossFinish& lf,
                                               const ShadowDst shadow, double* sh) {
  // The first elements of this thread and ALL of its norm partials are requested before anything waits: the norm
  // reduction (up to sixteen partials per thread, then a tree with eight barriers) used to be a chain of dependent
  // round trips in front of a kernel that is otherwise a pure stream.  Same sums in the same order.
  const int64_t n4 = count >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t i0 = i < n4 ? i : 0;                   // (clamped: an unconditional load, selected below)
  float4 gv = ld4(gin + i0 * 4), mv = ld4(m + i0 * 4), vv = ld4(v + i0 * 4), pv = ld4(p + i0 * 4);
  constexpr int NP = 16;                               // 256 threads x 16 = 4096 partials at most (SUMSQ_*_PARTS)
  double part_v[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) part_v[k] = sumsq_part[min((int)threadIdx.x + 256 * k, max(nparts - 1, 0))];
  finish_loss(lf, sh);
  double a = 0.0;
#pragma unroll
  for (int k = 0; k < NP; ++k)
    if ((int)threadIdx.x + 256 * k < nparts) a += part_v[k];
  for (int k = (int)threadIdx.x + 256 * NP; k < nparts; k += 256) a += sumsq_part[k];      // (never taken today)
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  const float total_norm = (float)sqrt(sh[0]);
  float coef = 1.0f;
  if (c.max_norm > 0.f) coef = fminf(c.max_norm / (total_norm + 1e-6f), 1.0f);
  if (stats_out && blockIdx.x == 0 && threadIdx.x == 0) {
    stats_out[0] = total_norm;
    stats_out[1] = coef;
  }
  const float gmul = coef * gscale;      // (gscale == 1 on the fp32 path: bit-identical to coef)
  while (i < n4) {
    float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x; float* pp = &pv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = gp[k] * gmul;
      gp[k] = gg;
      mp[k] = mp[k] + (gg - mp[k]) * c.one_minus_b1;
      vp[k] = vp[k] * c.b2 + (c.one_minus_b2 * gg) * gg;
      const float denom = sqrtf(vp[k]) / c.bc2_sqrt + c.eps;
      pp[k] = pp[k] - c.step_size * (mp[k] / denom);
    }
    st4(gout + i * 4, gv); st4(m + i * 4, mv); st4(v + i * 4, vv); st4(p + i * 4, pv);
    if (shadow.plain) {
      st4(shadow.plain + i * 4, pv);
      // the decode weight's K-major image (one-pass decode, decode_wdT_dev.h): element (o, col) of Wd goes to
      // 16-byte block ((col / 128 * 8 + col % 8) * 2 + o / 32) * 64 + 16 (o / 8 % 4) + col / 8 % 16, half word o % 8
      const int64_t e = i * 4 - shadow.dec_w;
      if (shadow.wdT && e >= 0 && e < (int64_t)shadow.OF * shadow.W) {
        const int o = (int)(e / shadow.W), col0 = (int)(e % shadow.W);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int col = col0 + k;
          const int idx = (((col >> 7) * 8 + (col & 7)) * 2 + (o >> 5)) * 64 + 16 * ((o >> 3) & 3) + ((col >> 3) & 15);
          shadow.wdT[(int64_t)idx * 8 + (o & 7)] = (bf16_bits)(pack_bf16x2(pp[k], 0.f) & 0xFFFFu);
        }
      }
    }
    i += stride;
    if (i < n4) { gv = ld4(gin + i * 4); mv = ld4(m + i * 4); vv = ld4(v + i * 4); pv = ld4(p + i * 4); }
  }
}

template <typename TG>
__global__ __launch_bounds__(256) void clip_adam_kernel(
    float* __restrict__ p, const TG* gin, float* gout, float* __restrict__ m, float* __restrict__ v,
    int64_t count, AdamConsts c, float gscale, const double* __restrict__ sumsq_part, int nparts,
    float* stats_out, LossFinish lf, ShadowDst shadow) {
  __shared__ double sh[256];
  clip_adam_body<TG>(p, gin, gout, m, v, count, c, gscale, sumsq_part, nparts, stats_out, lf, shadow, sh);
}

AdamConsts adam_consts(const blh_adam_hyper& h) {
  // torch.optim.Adam forms these scalars in double (Python floats) and rounds each once
  const double bc1 = 1.0 - pow(h.beta1, (double)h.step);
  const double bc2 = 1.0 - pow(h.beta2, (double)h.step);
  return AdamConsts{(float)(1.0 - h.beta1), (float)h.beta2, (float)(1.0 - h.beta2), (float)(h.lr / bc1),
                    (float)sqrt(bc2), (float)h.eps, (float)h.max_norm};
}

int launch_clip_adam(hipStream_t s, float* p, float* g, float* m, float* v, int64_t count,
                     const blh_adam_hyper& h, const double* sumsq_part, int nparts,
                     float* stats_out, LossFinish lf, ShadowDst shadow) {
  if (count % 4 != 0) return BLH_ERR_SHAPE;
  if (h.step < 1) return BLH_ERR_INVALID_ARGUMENT;
  const int blocks = (int)std::min<int64_t>(ceil_div(count / 4, 256), 2048);
  hipLaunchKernelGGL(clip_adam_kernel<float>, dim3(blocks), dim3(256), 0, s, p, (const float*)g, g, m, v, count,
                     adam_consts(h), 1.0f, sumsq_part, nparts, stats_out, lf, shadow);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

int launch_clip_adam_bf16(hipStream_t s, float* p, const uint16_t* g_bf16, float gscale, float* gout, float* m,
                          float* v, int64_t count, const blh_adam_hyper& h, const double* sumsq_part,
                          int nparts, float* stats_out, ShadowDst shadow) {
  if (count % 4 != 0) return BLH_ERR_SHAPE;
  if (h.step < 1) return BLH_ERR_INVALID_ARGUMENT;
  const int blocks = (int)std::min<int64_t>(ceil_div(count / 4, 256), 2048);
  hipLaunchKernelGGL(clip_adam_kernel<bf16_bits>, dim3(blocks), dim3(256), 0, s, p, g_bf16, gout, m, v, count,
                     adam_consts(h), gscale, sumsq_part, nparts, stats_out,
                     LossFinish{nullptr, 0, 1.0, nullptr}, shadow);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

}  // namespace blh

// ---------------------------------------------------------------------------
// stand-alone nn.utils.clip_grad_norm_ (drop-in path; the fast path fuses it into Adam)
// ---------------------------------------------------------------------------
namespace blh {

__global__ __launch_bounds__(256) void clip_scale_kernel(float* __restrict__ g, int64_t count,
                                                         float max_norm,
                                                         const double* __restrict__ sumsq_part,
                                                         int nparts, float* stats_out) {
  __shared__ double sh[256];
  double a = 0.0;
  for (int i = threadIdx.x; i < nparts; i += 256) a += sumsq_part[i];
  sh[threadIdx.x] = a;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  const float total_norm = (float)sqrt(sh[0]);
  const float coef = fminf(max_norm / (total_norm + 1e-6f), 1.0f);
  if (stats_out && blockIdx.x == 0 && threadIdx.x == 0) {
    stats_out[0] = total_norm;
    stats_out[1] = coef;
  }
  const int64_t n4 = count >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = ld4(g + i * 4);
    v.x *= coef; v.y *= coef; v.z *= coef; v.w *= coef;
    st4(g + i * 4, v);
  }
}

int launch_clip_scale(hipStream_t s, float* g, int64_t count, float max_norm,
                      const double* sumsq_part, int nparts, float* stats_out) {
  if (count % 4 != 0) return BLH_ERR_SHAPE;
  const int blocks = (int)std::min<int64_t>(ceil_div(count / 4, 256), 2048);
  hipLaunchKernelGGL(clip_scale_kernel, dim3(blocks), dim3(256), 0, s, g, count, max_norm,
                     sumsq_part, nparts, stats_out);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

}  // namespace blh

// ---------------------------------------------------------------------------
// materialise the Philox keep-mask (tests: feed the oracle the mask the kernels used)
// ---------------------------------------------------------------------------
namespace blh {

__global__ __launch_bounds__(256) void dropout_mask_kernel(uint8_t* __restrict__ out,
                                                           int64_t batch, int W, DropoutSrc drop) {
  const int64_t n4 = batch * (int64_t)(W / 4);
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    const int64_t r = i / (W / 4);
    const int col = (int)(i % (W / 4)) * 4;
    DropState st;
    const uint32_t nib = keep_nibble(drop, st, r, col, W, true);
    uchar4 k;
    k.x = nib & 1u; k.y = (nib >> 1) & 1u; k.z = (nib >> 2) & 1u; k.w = (nib >> 3) & 1u;
    *reinterpret_cast<uchar4*>(out + r * (int64_t)W + col) = k;
  }
}

int launch_dropout_mask(hipStream_t s, uint8_t* out, int64_t batch, int W, const DropoutSrc& drop) {
  const int64_t blocks = std::min<int64_t>(ceil_div(batch * (W / 4), 256), 4096);
  hipLaunchKernelGGL(dropout_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, s, out, batch, W,
                     drop);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

}  // namespace blh

// ---------------------------------------------------------------------------
// decode epilogue: pred = sum of split-K slabs + bias; optionally the fused MSE
// (dpred = scale*(pred-target), per-block partial sums of squared error)
// ---------------------------------------------------------------------------
namespace blh {

// grid = multiple of 3 blocks of 256 threads, so that a thread's 4 columns never change
// (256*4*3 = 3072 = 64*48): per-thread sums of dpred give the decode-bias gradient partials
__global__ __launch_bounds__(256) void decode_finish_kernel(
    const float* __restrict__ slabs, int splits, int64_t n, int out_features,
    const float* __restrict__ bias, float* __restrict__ pred, const float* __restrict__ target,
    float scale, float* __restrict__ dpred, float* __restrict__ loss_part,
    float* __restrict__ dbias_part) {
  __shared__ float sh[4];
  __shared__ float4 shc[256];
  float acc = 0.f;
  float4 dsum = make_float4(0.f, 0.f, 0.f, 0.f);
  const int64_t n4 = n >> 2;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n4;
       i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = ld4(slabs + i * 4);
    for (int s = 1; s < splits; ++s) {
      const float4 u = ld4(slabs + (int64_t)s * n + i * 4);
      v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
    }
    const float4 b = ld4(bias + (int)((i * 4) % out_features));
    v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    st4(pred + i * 4, v);
    if (target) {
      const float4 t = ld4(target + i * 4);
      float4 d;
      d.x = v.x - t.x; d.y = v.y - t.y; d.z = v.z - t.z; d.w = v.w - t.w;
      acc += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
      d.x *= scale; d.y *= scale; d.z *= scale; d.w *= scale;
      dsum.x += d.x; dsum.y += d.y; dsum.z += d.z; dsum.w += d.w;
      st4(dpred + i * 4, d);
    }
  }
  if (target) {
    const float t = block_sum_f32(acc, sh);
    if (threadIdx.x == 0) loss_part[blockIdx.x] = t;
    if (dbias_part) {
      // thread t of block b owns column group (b*256 + t) % (out_features/4); fixed-order sums
      shc[threadIdx.x] = dsum;
      __syncthreads();
      const int cg = out_features >> 2;
      if ((int)threadIdx.x < cg) {
        const int first = (int)((threadIdx.x + cg - (blockIdx.x * 256) % cg) % cg);
        float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int k = first; k < 256; k += cg) {
          const float4 u = shc[k];
          a.x += u.x; a.y += u.y; a.z += u.z; a.w += u.w;
        }
        st4(dbias_part + (int64_t)blockIdx.x * out_features + threadIdx.x * 4, a);
      }
    }
  }
}

int launch_decode_finish(hipStream_t s, const float* slabs, int splits, int64_t batch,
                         int out_features, const float* bias, float* pred, const float* target,
                         float scale, float* dpred, float* loss_part, int* nparts,
                         float* dbias_part) {
  const int64_t n = batch * out_features;
  if (out_features % 4 != 0) return BLH_ERR_SHAPE;
  int blocks = (int)std::min<int64_t>(ceil_div(n / 4, 256), 1023);
  if (dbias_part) {
    // the fixed thread -> column mapping needs (gridDim*256) % (out_features/4) == 0
    if ((256 * 3) % (out_features / 4) != 0) return BLH_ERR_SHAPE;
    blocks = (int)round_up(blocks, 3);
  }
  hipLaunchKernelGGL(decode_finish_kernel, dim3(blocks), dim3(256), 0, s, slabs, splits, n,
                     out_features, bias, pred, target, scale, dpred, loss_part, dbias_part);
  BLH_HIP_TRY(hipGetLastError());
  if (nparts) *nparts = blocks;
  return BLH_OK;
}

}  // namespace blh

// ---------------------------------------------------------------------------
// device-resident step state (hipGraph replay)
// ---------------------------------------------------------------------------
namespace blh {

__global__ void step_state_advance_kernel(blh_step_state* st) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const int t = st->step + 1;
    st->step = t;
    st->rng_step += 1;
    const double bc1 = 1.0 - pow(st->beta1, (double)t);
    const double bc2 = 1.0 - pow(st->beta2, (double)t);
    st->step_size = (float)(st->lr / bc1);
    st->bc2_sqrt = (float)sqrt(bc2);
  }
}

int launch_step_state_advance(hipStream_t s, blh_step_state* st) {
  hipLaunchKernelGGL(step_state_advance_kernel, dim3(1), dim3(64), 0, s, st);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

__global__ __launch_bounds__(256) void clip_adam_dev_kernel(
    float* __restrict__ p, float* g, float* __restrict__ m, float* __restrict__ v,
    int64_t count, const blh_step_state* __restrict__ st, const double* __restrict__ sumsq_part,
    int nparts, float* stats_out, LossFinish lf, ShadowDst shadow) {
  __shared__ double sh[256];
  const AdamConsts c{(float)(1.0 - st->beta1), (float)st->beta2, (float)(1.0 - st->beta2), st->step_size,
                     st->bc2_sqrt, (float)st->eps, (float)st->max_norm};
  clip_adam_body<float>(p, g, g, m, v, count, c, 1.0f, sumsq_part, nparts, stats_out, lf, shadow, sh);
}

int launch_clip_adam_dev(hipStream_t s, float* p, float* g, float* m, float* v, int64_t count,
                         const blh_step_state* st, const double* sumsq_part, int nparts,
                         float* stats_out, LossFinish lf, ShadowDst shadow) {
  if (count % 4 != 0) return BLH_ERR_SHAPE;
  const int blocks = (int)std::min<int64_t>(ceil_div(count / 4, 256), 2048);
  hipLaunchKernelGGL(clip_adam_dev_kernel, dim3(blocks), dim3(256), 0, s, p, g, m, v, count, st,
                     sumsq_part, nparts, stats_out, lf, shadow);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

}  // namespace blh

// ---------------------------------------------------------------------------
// validation metric (/root/reference/valid_bilinear.py:53-70): de-normalise prediction and
// ground truth with the train-set mean / stddev, per-sample sum over the 16 joints of the
// Euclidean distance, then per-action sums (replaces the reference's per-sample Python loop)
// ---------------------------------------------------------------------------
namespace blh {

__global__ __launch_bounds__(256) void mpjpe_kernel(const float* __restrict__ pred,
                                                    const float* __restrict__ target,
                                                    const float* __restrict__ mean,
                                                    const float* __restrict__ stddev,
                                                    int64_t batch, int joints,
                                                    float* __restrict__ dist) {
  const int64_t b = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (b >= batch) return;
  const int F = joints * 3;
  float acc = 0.f;
  for (int j = 0; j < joints; ++j) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int f = j * 3 + c;
      const float p = stddev[f] * pred[b * F + f] + mean[f];
      const float g = stddev[f] * target[b * F + f] + mean[f];
      const float d = p - g;
      s += d * d;
    }
    acc += sqrtf(s);
  }
  dist[b] = acc;
}

// one block per segment: sum (fp64) and count of dist[b] with ids[b] == segment
__global__ __launch_bounds__(256) void segment_sum_kernel(const float* __restrict__ dist,
                                                          const int32_t* __restrict__ ids,
                                                          int64_t batch, double* __restrict__ sum,
                                                          int64_t* __restrict__ count) {
  __shared__ double sh[256];
  __shared__ int sc[256];
  const int seg = blockIdx.x;
  double a = 0.0;
  int c = 0;
  for (int64_t b = threadIdx.x; b < batch; b += 256)
    if (ids[b] == seg) { a += (double)dist[b]; ++c; }
  sh[threadIdx.x] = a; sc[threadIdx.x] = c;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) { sh[threadIdx.x] += sh[threadIdx.x + o]; sc[threadIdx.x] += sc[threadIdx.x + o]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { sum[seg] += sh[0]; count[seg] += sc[0]; }
}

int launch_mpjpe(hipStream_t s, const float* pred, const float* target, const float* mean,
                 const float* stddev, int64_t batch, int joints, float* dist) {
  hipLaunchKernelGGL(mpjpe_kernel, dim3((unsigned)ceil_div(batch, 256)), dim3(256), 0, s, pred,
                     target, mean, stddev, batch, joints, dist);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

int launch_segment_sum(hipStream_t s, const float* dist, const int32_t* ids, int64_t batch,
                       int segments, double* sum, int64_t* count) {
  hipLaunchKernelGGL(segment_sum_kernel, dim3(segments), dim3(256), 0, s, dist, ids, batch, sum,
                     count);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

}  // namespace blh

// ---------------------------------------------------------------------------
// SyncBN (data parallel, statistics over the GLOBAL batch): the local tile partials are
// reduced to fp64 column sums [sum z | sum z^2] that the host all-reduces across ranks; the
// finalize then works from the global sums.  (fp64 keeps sum z^2 - (sum z)^2/n benign.)
// ---------------------------------------------------------------------------
namespace blh {

__global__ __launch_bounds__(256) void bn_fwd_local_sums_kernel(const float* __restrict__ part,
                                                                int tiles, int tile_rows,
                                                                int64_t batch, int W,
                                                                double* __restrict__ sums) {
  __shared__ double r1[8][32], r2[8][32];
  const int cl = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int col = blockIdx.x * 32 + cl;
  double a1 = 0.0, a2 = 0.0;
  if (col < W)
    for (int t = sl; t < tiles; t += 8) {
      const double n = (double)min((int64_t)tile_rows, batch - (int64_t)t * tile_rows);
      const double mu = (double)part[((int64_t)t * 2 + 0) * W + col];
      a1 += n * mu;
      a2 += (double)part[((int64_t)t * 2 + 1) * W + col] + n * mu * mu;
    }
  r1[sl][cl] = a1; r2[sl][cl] = a2;
  __syncthreads();
  if (sl == 0 && col < W) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int s = 0; s < 8; ++s) { t1 += r1[s][cl]; t2 += r2[s][cl]; }
    sums[col] = t1;
    sums[W + col] = t2;
  }
}

__global__ __launch_bounds__(256) void bn_fwd_finalize_sums_kernel(
    const double* __restrict__ sums, int64_t n_global, int W, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* running_mean, float* running_var, const int64_t* nbt,
    float momentum, float* saved_mean, float* saved_invstd, float* scale, float* shift) {
  const int col = blockIdx.x * blockDim.x + threadIdx.x;
  if (col >= W) return;
  const double n = (double)n_global;
  const double mean = sums[col] / n;
  double m2 = sums[W + col] - n * mean * mean;
  if (m2 < 0.0) m2 = 0.0;
  const double var = m2 / n;
  const float invstd = (float)(1.0 / sqrt(var + (double)BN_EPS));
  const float mu = (float)mean;
  const float sc = gamma[col] * invstd;
  saved_mean[col] = mu;
  saved_invstd[col] = invstd;
  scale[col] = sc;
  shift[col] = beta[col] - mu * sc;
  const double f = (momentum >= 0.f) ? (double)momentum : 1.0 / (double)(nbt[0] + 1);
  const double unbiased = m2 / (double)(n_global > 1 ? n_global - 1 : 1);
  running_mean[col] = (float)((1.0 - f) * (double)running_mean[col] + f * mean);
  running_var[col] = (float)((1.0 - f) * (double)running_var[col] + f * unbiased);
}

int launch_bn_fwd_local_sums(hipStream_t s, const float* stat_part, int tiles, int tile_rows,
                             int64_t batch, int W, double* sums) {
  hipLaunchKernelGGL(bn_fwd_local_sums_kernel, dim3((unsigned)ceil_div(W, 32)), dim3(256), 0, s,
                     stat_part, tiles, tile_rows, batch, W, sums);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

int launch_bn_fwd_finalize_sums(hipStream_t s, const double* sums, int64_t n_global, int W,
                                const float* gamma, const float* beta, float* running_mean,
                                float* running_var, int64_t* nbt, float momentum,
                                float* saved_mean, float* saved_invstd, float* scale,
                                float* shift) {
  hipLaunchKernelGGL(bn_fwd_finalize_sums_kernel, dim3((unsigned)ceil_div(W, 256)), dim3(256), 0, s,
                     sums, n_global, W, gamma, beta, running_mean, running_var, nbt, momentum,
                     saved_mean, saved_invstd, scale, shift);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

}  // namespace blh

// ---------------------------------------------------------------------------
// gemm_dtype = 3 (fp16 two-piece split): max |w| of each hidden Linear weight, as 64 partials
// per layer that the GEMM reduces in its prologue.  W [layers][count] with layer stride w_stride.
// ---------------------------------------------------------------------------
namespace blh {

__global__ __launch_bounds__(256) void wamax_kernel(const float* __restrict__ W, int64_t w_stride,
                                                    int64_t count, float* __restrict__ part) {
  __shared__ float red[4];
  const float* __restrict__ Wl = W + (int64_t)blockIdx.y * w_stride;
  float m = 0.f;
  for (int64_t i = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4; i < count; i += (int64_t)gridDim.x * 1024)
    m = amax4(m, *reinterpret_cast<const float4*>(Wl + i));
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)
    part[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

int launch_wamax(hipStream_t s, const float* W, int64_t w_stride, int layers, int64_t count,
                 float* part) {
  if (layers <= 0) return BLH_OK;
  if (count % 4 != 0) return BLH_ERR_SHAPE;
  hipLaunchKernelGGL(wamax_kernel, dim3(WAMAX_PARTS, (unsigned)layers), dim3(256), 0, s, W, w_stride,
                     count, part);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

}  // namespace blh
