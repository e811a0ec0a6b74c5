// Streaming BatchNorm kernels of the fp32-storage path (gemm_dtype 0..3), second generation: the
// same restructuring as bn_bf16.hip for float tensors.
//
//   forward   A  = 2 keep relu(Z scale + shift) (+ skip)          bn_apply_f2      (writes the keep bits)
//   backward  S1 = sum_b dY z, S2 = sum_b dY per row chunk        bn_bwd_reduce_f2 (dY = 2 dA keep [y > 0])
//             dgamma = invstd (S1 - mean S2), dbeta = S2          bn_bwd_finalize_h2 (bn_bf16.hip, fp64)
//             dZ = scale dY + a z + b, column sums of dZ          bn_bwd_apply_f2
//               a = -scale c2 invstd, b = scale (c2 invstd mean - c1), c1 = dbeta / B, c2 = dgamma / B
//               (/root/reference/model/bilinear.py:10, BatchNorm1d backward with zhat expanded)
//
// A lane owns 4 consecutive columns (one 16-byte access per row and tensor), a block = 4 waves x
// 256 columns; inside a 32-row Philox patch wave w owns rows 8w .. 8w+7 — exactly word w of the
// patch, which is also the keep word the backward kernels read: keep[row / 8][col / 4], nibble j =
// row 8 (row/8) + j, bit c = column 4 (col/4) + c.  One word per 8 rows x 4 columns replaces the
// Philox regeneration in both backward kernels; per column they carry scale, shift (the ReLU gate)
// and a, b (two FMAs per element) instead of six vectors, ~80 registers instead of 180: beside a
// GEMM that owns the matrix pipes several of their waves fit on a SIMD and each needs fewer of
// the vector-issue slots the GEMM leaves (profiles/r02_step_timeline.md: the first generation ran
// 3x slower beside a GEMM than alone).
#include "common.h"
#include "philox.h"
#include "bn_f32_dev.h"

namespace blh {

static constexpr int F2_THREADS = 256;
static constexpr int F2_COLS = 256;
static constexpr float F2_BN_EPS = 1e-5f;
#define BLH_F2_PRIO() __builtin_amdgcn_s_setprio(3)

__device__ __forceinline__ float4 f2_ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float f2_finite_abs(float x) {
  const float a = fabsf(x);
  return a <= 3.402823466e+38f ? a : 0.f;      // false for Inf and NaN
}
// one max-|value| partial per wave (gemm_dtype 3: the fp16-split GEMM picks its scale from them)
__device__ __forceinline__ void f2_wave_amax_store(float m, float* __restrict__ part) {
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  if ((threadIdx.x & 63) == 0)
    part[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (F2_THREADS / 64) + (threadIdx.x >> 6)] = m;
}
// sum the per-lane float4 partials of the block's 4 waves: out[col0 .. col0 + 255]
__device__ __forceinline__ void f2_block_colsum(float4 v, float* red, float* out, int col0, int W) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  reinterpret_cast<float4*>(red)[w * 64 + lane] = v;
  __syncthreads();
  const int t = threadIdx.x;
  if (col0 + t < W) out[col0 + t] = (red[t] + red[256 + t]) + (red[512 + t] + red[768 + t]);
}

template <bool TRAIN>
__global__ __launch_bounds__(F2_THREADS) void bn_apply_f2_kernel(
    const float* __restrict__ Z, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ running_mean, const float* __restrict__ running_var,
    const float* __restrict__ skip, float* __restrict__ A, uint32_t* __restrict__ keepbits, int64_t batch,
    int W, int row_chunk, DropoutSrc drop, int64_t* nbt, float* __restrict__ amax_part) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = blockIdx.x * F2_COLS + lane * 4;
  if (TRAIN && nbt && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) nbt[0] += 1;
  float am = 0.f;
  const bool okc = col < W;          // (no early return: the amax reduction needs whole waves)
  const int cc = okc ? col : 0;
  float4 sc, sh;
  if (TRAIN) {
    sc = f2_ld(scale + cc);
    sh = f2_ld(shift + cc);
  } else {
    const float4 g = f2_ld(gamma + cc), b = f2_ld(beta + cc), rm = f2_ld(running_mean + cc),
                 rv = f2_ld(running_var + cc);
    sc.x = g.x * (1.0f / sqrtf(rv.x + F2_BN_EPS)); sc.y = g.y * (1.0f / sqrtf(rv.y + F2_BN_EPS));
    sc.z = g.z * (1.0f / sqrtf(rv.z + F2_BN_EPS)); sc.w = g.w * (1.0f / sqrtf(rv.w + F2_BN_EPS));
    sh.x = b.x - rm.x * sc.x; sh.y = b.y - rm.y * sc.y;
    sh.z = b.z - rm.z * sc.z; sh.w = b.w - rm.w * sc.w;
  }
  const int W4 = W >> 2;
  const int64_t r0 = (int64_t)blockIdx.y * row_chunk;
  const int64_t r1 = min(batch, r0 + row_chunk);
  for (int64_t base = r0; okc && base < r1; base += 32) {
    const int64_t rg = base + 8 * w;
    if (rg >= batch) break;
    float4 z[8], k[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t r = min(rg + j, batch - 1);
      z[j] = f2_ld(Z + r * W + col);
      if (skip) k[j] = f2_ld(skip + r * W + col);
    }
    uint32_t kw = 0xffffffffu;
    if (TRAIN) {
      kw = f2_keep_word(drop, base, w, col, W, batch);
      if (keepbits) keepbits[(rg >> 3) * W4 + (col >> 2)] = kw;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t nib = kw >> (4 * j);
      float4 a;
      a.x = fmaxf(fmaf(z[j].x, sc.x, sh.x), 0.f); a.y = fmaxf(fmaf(z[j].y, sc.y, sh.y), 0.f);
      a.z = fmaxf(fmaf(z[j].z, sc.z, sh.z), 0.f); a.w = fmaxf(fmaf(z[j].w, sc.w, sh.w), 0.f);
      if (TRAIN) {
        a.x = (nib & 1u) ? a.x * 2.f : 0.f; a.y = (nib & 2u) ? a.y * 2.f : 0.f;
        a.z = (nib & 4u) ? a.z * 2.f : 0.f; a.w = (nib & 8u) ? a.w * 2.f : 0.f;
      }
      if (skip) { a.x += k[j].x; a.y += k[j].y; a.z += k[j].z; a.w += k[j].w; }
      if (rg + j < batch) {
        *reinterpret_cast<float4*>(A + (rg + j) * W + col) = a;
        am = fmaxf(fmaxf(am, fmaxf(f2_finite_abs(a.x), f2_finite_abs(a.y))),
                   fmaxf(f2_finite_abs(a.z), f2_finite_abs(a.w)));
      }
    }
  }
  if (amax_part) f2_wave_amax_store(am, amax_part);
}

__global__ __launch_bounds__(F2_THREADS) void bn_bwd_reduce_f2_kernel(
    const float* __restrict__ dA, const float* __restrict__ Z, const float* __restrict__ scale,
    const float* __restrict__ shift, const uint32_t* __restrict__ keepbits, float* __restrict__ part,
    int64_t batch, int W, int row_chunk) {
  BLH_F2_PRIO();
  __shared__ __attribute__((aligned(16))) float red[4 * 256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col0 = blockIdx.x * F2_COLS;
  const int col = col0 + lane * 4;
  const bool ok = col < W;
  const int cc = ok ? col : 0;
  const float4 sc = f2_ld(scale + cc), sh = f2_ld(shift + cc);
  float4 s1 = make_float4(0, 0, 0, 0), s2 = make_float4(0, 0, 0, 0);
  const int W4 = W >> 2;
  const int64_t r0 = (int64_t)blockIdx.y * row_chunk;
  const int64_t r1 = min(batch, r0 + row_chunk);
  if (ok)
    for (int64_t rg = r0 + 8 * w; rg < r1; rg += 32) {     // wave w: 8-row groups w, w+4, ...
      const uint32_t kw = keepbits[(rg >> 3) * W4 + (col >> 2)];
      float4 z[8], g[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t r = min(rg + j, batch - 1);
        z[j] = f2_ld(Z + r * W + col);
        g[j] = f2_ld(dA + r * W + col);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t nib = (rg + j < batch) ? (kw >> (4 * j)) : 0u;
        const float dx = ((nib & 1u) && (fmaf(z[j].x, sc.x, sh.x) > 0.f)) ? g[j].x * 2.f : 0.f;
        const float dy = ((nib & 2u) && (fmaf(z[j].y, sc.y, sh.y) > 0.f)) ? g[j].y * 2.f : 0.f;
        const float dz = ((nib & 4u) && (fmaf(z[j].z, sc.z, sh.z) > 0.f)) ? g[j].z * 2.f : 0.f;
        const float dw = ((nib & 8u) && (fmaf(z[j].w, sc.w, sh.w) > 0.f)) ? g[j].w * 2.f : 0.f;
        s2.x += dx; s2.y += dy; s2.z += dz; s2.w += dw;
        s1.x = fmaf(dx, z[j].x, s1.x); s1.y = fmaf(dy, z[j].y, s1.y);
        s1.z = fmaf(dz, z[j].z, s1.z); s1.w = fmaf(dw, z[j].w, s1.w);
      }
    }
  f2_block_colsum(s1, red, part + ((int64_t)blockIdx.y * 2 + 0) * W, col0, W);
  f2_block_colsum(s2, red, part + ((int64_t)blockIdx.y * 2 + 1) * W, col0, W);
}

__global__ __launch_bounds__(F2_THREADS) void bn_bwd_apply_f2_kernel(
    const float* __restrict__ dA, const float* __restrict__ Z, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ dgamma, const float* __restrict__ dbeta,
    const uint32_t* __restrict__ keepbits, float* __restrict__ dZ, float* __restrict__ colsum_part,
    int64_t batch, int W, int row_chunk, int64_t norm_batch, float* __restrict__ amax_part) {
  BLH_F2_PRIO();
  __shared__ __attribute__((aligned(16))) float red[4 * 256];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col0 = blockIdx.x * F2_COLS;
  const int col = col0 + lane * 4;
  const bool ok = col < W;
  const int cc = ok ? col : 0;
  const float4 sc = f2_ld(scale + cc), sh = f2_ld(shift + cc);
  float4 ca, cb;
  {
    const float4 mu = f2_ld(mean + cc), is = f2_ld(invstd + cc), dg = f2_ld(dgamma + cc), db = f2_ld(dbeta + cc);
    const float inv_b = 1.0f / (float)norm_batch;
    const float tx = sc.x * (dg.x * inv_b) * is.x, ty = sc.y * (dg.y * inv_b) * is.y,
                tz = sc.z * (dg.z * inv_b) * is.z, tw = sc.w * (dg.w * inv_b) * is.w;
    ca = make_float4(-tx, -ty, -tz, -tw);
    cb.x = fmaf(tx, mu.x, -sc.x * (db.x * inv_b)); cb.y = fmaf(ty, mu.y, -sc.y * (db.y * inv_b));
    cb.z = fmaf(tz, mu.z, -sc.z * (db.z * inv_b)); cb.w = fmaf(tw, mu.w, -sc.w * (db.w * inv_b));
  }
  float4 cs = make_float4(0, 0, 0, 0);
  float am = 0.f;
  const int W4 = W >> 2;
  const int64_t r0 = (int64_t)blockIdx.y * row_chunk;
  const int64_t r1 = min(batch, r0 + row_chunk);
  if (ok)
    for (int64_t rg = r0 + 8 * w; rg < r1; rg += 32) {
      const uint32_t kw = keepbits[(rg >> 3) * W4 + (col >> 2)];
      float4 z[8], g[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int64_t r = min(rg + j, batch - 1);
        z[j] = f2_ld(Z + r * W + col);
        g[j] = f2_ld(dA + r * W + col);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t nib = kw >> (4 * j);
        const float dx = ((nib & 1u) && (fmaf(z[j].x, sc.x, sh.x) > 0.f)) ? g[j].x * 2.f : 0.f;
        const float dy = ((nib & 2u) && (fmaf(z[j].y, sc.y, sh.y) > 0.f)) ? g[j].y * 2.f : 0.f;
        const float dz = ((nib & 4u) && (fmaf(z[j].z, sc.z, sh.z) > 0.f)) ? g[j].z * 2.f : 0.f;
        const float dw = ((nib & 8u) && (fmaf(z[j].w, sc.w, sh.w) > 0.f)) ? g[j].w * 2.f : 0.f;
        float4 o;
        o.x = fmaf(sc.x, dx, fmaf(ca.x, z[j].x, cb.x)); o.y = fmaf(sc.y, dy, fmaf(ca.y, z[j].y, cb.y));
        o.z = fmaf(sc.z, dz, fmaf(ca.z, z[j].z, cb.z)); o.w = fmaf(sc.w, dw, fmaf(ca.w, z[j].w, cb.w));
        if (rg + j < batch) {
          cs.x += o.x; cs.y += o.y; cs.z += o.z; cs.w += o.w;
          *reinterpret_cast<float4*>(dZ + (rg + j) * W + col) = o;
          am = fmaxf(fmaxf(am, fmaxf(f2_finite_abs(o.x), f2_finite_abs(o.y))),
                     fmaxf(f2_finite_abs(o.z), f2_finite_abs(o.w)));
        }
      }
    }
  f2_block_colsum(cs, red, colsum_part + (int64_t)blockIdx.y * W, col0, W);
  if (amax_part) f2_wave_amax_store(am, amax_part);
}

// ---- host --------------------------------------------------------------------------------------
static dim3 f2_grid(int64_t batch, int W) {
  return dim3((unsigned)ceil_div(W, F2_COLS), (unsigned)ew_num_row_chunks(batch));
}

int64_t bn_keepbits_words_f32(int64_t batch, int W) { return ceil_div(batch, 8) * (int64_t)(W / 4); }

int launch_bn_apply_f2(hipStream_t s, bool train, const float* Z, const float* scale, const float* shift,
                       const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, const float* skip, float* A, uint32_t* keepbits,
                       int64_t batch, int W, const DropoutSrc& drop, int64_t* nbt, float* amax_part) {
  if (W % 4 != 0) return BLH_ERR_SHAPE;
  const int rc = ew_row_chunk(batch);
  if (train)
    hipLaunchKernelGGL(bn_apply_f2_kernel<true>, f2_grid(batch, W), dim3(F2_THREADS), 0, s, Z, scale, shift,
                       gamma, beta, running_mean, running_var, skip, A, keepbits, batch, W, rc, drop, nbt,
                       amax_part);
  else
    hipLaunchKernelGGL(bn_apply_f2_kernel<false>, f2_grid(batch, W), dim3(F2_THREADS), 0, s, Z, scale, shift,
                       gamma, beta, running_mean, running_var, skip, A, keepbits, batch, W, rc, drop, nbt,
                       amax_part);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

int launch_bn_bwd_reduce_f2(hipStream_t s, const float* dA, const float* Z, const float* scale,
                            const float* shift, const uint32_t* keepbits, float* part, int64_t batch, int W) {
  if (W % 4 != 0) return BLH_ERR_SHAPE;
  hipLaunchKernelGGL(bn_bwd_reduce_f2_kernel, f2_grid(batch, W), dim3(F2_THREADS), 0, s, dA, Z, scale, shift,
                     keepbits, part, batch, W, ew_row_chunk(batch));
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

int launch_bn_bwd_apply_f2(hipStream_t s, const float* dA, const float* Z, const float* scale,
                           const float* shift, const float* mean, const float* invstd, const float* dgamma,
                           const float* dbeta, const uint32_t* keepbits, float* dZ, float* colsum_part,
                           int64_t batch, int W, int64_t norm_batch, float* amax_part) {
  if (W % 4 != 0) return BLH_ERR_SHAPE;
  launch_kernel(bn_bwd_apply_f2_kernel, f2_grid(batch, W), dim3(F2_THREADS), 0, s, dA, Z, scale, shift, mean,
                invstd, dgamma, dbeta, keepbits, dZ, colsum_part, batch, W, ew_row_chunk(batch), norm_batch,
                amax_part);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

}  // namespace blh
