// Streaming BatchNorm kernels of the bf16-storage path (gemm_dtype 4), second generation.
//
//   forward   A  = 2 keep relu(Z scale + shift) (+ skip)          bn_apply_h2      (writes the keep bits)
//   backward  S1 = sum_b dY z, S2 = sum_b dY per row chunk        bn_bwd_reduce_h2 (dY = 2 dA keep [y > 0])
//             dgamma = invstd (S1 - mean S2), dbeta = S2          bn_bwd_finalize_h2 (fp64)
//             dZ = scale dY + a z + b, column sums of dZ          bn_bwd_apply_h2
//               with a = -scale c2 invstd, b = scale (c2 invstd mean - c1), c1 = dbeta / B, c2 = dgamma / B
//               (= gamma invstd (dY - dbeta/B - zhat dgamma/B) of /root/reference/model/bilinear.py:10's
//                BatchNorm1d backward, with zhat = (z - mean) invstd expanded: two FMAs per element)
//
// Against the first generation (elementwise.hip: bn_*_h_kernel, 180-256 VGPRs, two waves per SIMD,
// 3.4-4.5 TB/s):
//   * the dropout keep bits are written ONCE by the forward kernel, one bit per element
//     (keep[row / 4][col / 8] = 32 bits: byte j = row 4 (row/4) + j, bit c = column 8 (col/8) + c; 2 MB
//     per stage at B = 16384, W = 1024), and the two backward kernels read one 32-bit word per
//     4 rows x 8 columns instead of regenerating two Philox patches per 32 x 8 elements;
//   * the backward kernels carry per column only what their two FMAs need (scale, shift for the
//     ReLU gate; a, b) — 16-32 registers of constants instead of 48;
//   * tensors stay packed (two bf16 per register) until the row that uses them: 4 rows x 2 tensors
//     in flight per lane cost 32 registers, the kernels fit 3-4 waves per SIMD.
// Row mapping: a block = 4 waves x 512 columns (a lane owns 8 consecutive columns = one 16-byte
// access per row and tensor); inside a 32-row Philox patch wave w owns rows 8w .. 8w+7 = word w of
// the patch = two 4-row groups.
#include "common.h"
#include "philox.h"

namespace blh {

typedef uint16_t bf16_bits;
static constexpr int H2_THREADS = 256;
static constexpr int H2_COLS = 512;
static constexpr float H2_BN_EPS = 1e-5f;

#ifndef BLH_H2_WAVES
#define BLH_H2_WAVES 3
#endif
// (beside a GEMM that keeps every matrix pipe busy a streaming kernel only issues in the gaps:
//  raise the wave priority, elementwise.hip BLH_EW_PRIO)
#define BLH_H2_PRIO() __builtin_amdgcn_s_setprio(3)

__device__ __forceinline__ float bflo(uint32_t v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bfhi(uint32_t v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 b = {(__bf16)lo, (__bf16)hi};          // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return *reinterpret_cast<const uint32_t*>(&b);
}
__device__ __forceinline__ void unpack8(const uint4& q, float (&f)[8]) {
  f[0] = bflo(q.x); f[1] = bfhi(q.x); f[2] = bflo(q.y); f[3] = bfhi(q.y);
  f[4] = bflo(q.z); f[5] = bfhi(q.z); f[6] = bflo(q.w); f[7] = bfhi(q.w);
}
__device__ __forceinline__ uint4 pack8(const float (&f)[8]) {
  return make_uint4(pack2(f[0], f[1]), pack2(f[2], f[3]), pack2(f[4], f[5]), pack2(f[6], f[7]));
}
__device__ __forceinline__ void ldc8(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}

// sum the per-lane 8-column partials of the block's 4 waves: out[col0 .. col0 + 511]
__device__ __forceinline__ void block_colsum8(const float (&v)[8], float* red, float* out, int col0, int W) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  __syncthreads();
  reinterpret_cast<float4*>(red)[w * 128 + lane * 2 + 0] = make_float4(v[0], v[1], v[2], v[3]);
  reinterpret_cast<float4*>(red)[w * 128 + lane * 2 + 1] = make_float4(v[4], v[5], v[6], v[7]);
  __syncthreads();
  for (int t = threadIdx.x; t < 512; t += H2_THREADS)
    if (col0 + t < W) out[col0 + t] = (red[t] + red[512 + t]) + (red[1024 + t] + red[1536 + t]);
}

// keep bits of rows base + 8w .. base + 8w + 7, columns col .. col + 7: two words, byte j of word h
// = row base + 8w + 4h + j (low nibble columns 0-3, high nibble columns 4-7)
__device__ __forceinline__ void keep_words(const DropoutSrc& d, int64_t base, int w, int col, int W,
                                           int64_t batch, uint32_t (&kw)[2]) {
  if (d.keep) {                      // explicit masks (parity tests): [B][W] bytes
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      uint32_t word = 0;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t r = base + 8 * w + 4 * h + j;
        if (r < batch) {
          const uint2 k = *reinterpret_cast<const uint2*>(d.keep + r * (int64_t)W + col);
          uint32_t b = 0;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            b |= ((k.x >> (8 * c)) & 0xFFu) ? (1u << c) : 0u;
            b |= ((k.y >> (8 * c)) & 0xFFu) ? (16u << c) : 0u;
          }
          word |= b << (8 * j);
        }
      }
      kw[h] = word;
    }
    return;
  }
  // Philox: one call = the 32 x 4 patch; rows 8w .. 8w+7 are word w (row r of the patch: word r >> 3,
  // nibble r & 7)
  const Philox128 p0 = dropout_patch(d.seed, dropout_step(d), d.layer, base + d.row_offset, col);
  const Philox128 p1 = dropout_patch(d.seed, dropout_step(d), d.layer, base + d.row_offset, col + 4);
  const uint32_t a = w == 0 ? p0.w[0] : (w == 1 ? p0.w[1] : (w == 2 ? p0.w[2] : p0.w[3]));
  const uint32_t b = w == 0 ? p1.w[0] : (w == 1 ? p1.w[1] : (w == 2 ? p1.w[2] : p1.w[3]));
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    uint32_t word = 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = 4 * h + j;
      word |= (((a >> (4 * n)) & 0xFu) | (((b >> (4 * n)) & 0xFu) << 4)) << (8 * j);
    }
    kw[h] = word;
  }
}

// ---------------------------------------------------------------------------------------------
template <bool TRAIN>
__global__ __launch_bounds__(H2_THREADS, BLH_H2_WAVES) void bn_apply_h2_kernel(
    const bf16_bits* __restrict__ Z, const float* __restrict__ scale, const float* __restrict__ shift,
    const float* __restrict__ gamma, const float* __restrict__ beta,
    const float* __restrict__ running_mean, const float* __restrict__ running_var,
    const bf16_bits* __restrict__ skip, bf16_bits* __restrict__ A, uint32_t* __restrict__ keepbits,
    int64_t batch, int W, int row_chunk, DropoutSrc drop, int64_t* nbt) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = blockIdx.x * H2_COLS + lane * 8;
  if (TRAIN && nbt && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) nbt[0] += 1;
  if (col >= W) return;
  float sc[8], sh[8];
  if (TRAIN) {
    ldc8(scale + col, sc);
    ldc8(shift + col, sh);
  } else {
    float g[8], b[8], rm[8], rv[8];
    ldc8(gamma + col, g); ldc8(beta + col, b); ldc8(running_mean + col, rm); ldc8(running_var + col, rv);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      sc[c] = g[c] * (1.0f / sqrtf(rv[c] + H2_BN_EPS));
      sh[c] = b[c] - rm[c] * sc[c];
    }
  }
  const int W8 = W >> 3;
  const int64_t r0 = (int64_t)blockIdx.y * row_chunk;
  const int64_t r1 = min(batch, r0 + row_chunk);
  for (int64_t base = r0; base < r1; base += 32) {
    uint32_t kw[2] = {0xffffffffu, 0xffffffffu};
    if (TRAIN) keep_words(drop, base, w, col, W, batch, kw);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int64_t rg = base + 8 * w + 4 * h;          // first row of the 4-row group
      if (rg >= batch) break;
      uint4 zq[4], kq[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t r = min(rg + j, batch - 1);
        zq[j] = *reinterpret_cast<const uint4*>(Z + r * W + col);
        if (skip) kq[j] = *reinterpret_cast<const uint4*>(skip + r * W + col);
      }
      if (TRAIN && keepbits) keepbits[(rg >> 2) * W8 + (col >> 3)] = kw[h];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float z[8], k[8], a[8];
        unpack8(zq[j], z);
        if (skip) unpack8(kq[j], k);
        const uint32_t bits = kw[h] >> (8 * j);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          float y = fmaxf(fmaf(z[c], sc[c], sh[c]), 0.f);
          if (TRAIN) y = ((bits >> c) & 1u) ? y * 2.f : 0.f;
          a[c] = skip ? y + k[c] : y;
        }
        if (rg + j < batch) *reinterpret_cast<uint4*>(A + (rg + j) * W + col) = pack8(a);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(H2_THREADS, BLH_H2_WAVES) void bn_bwd_reduce_h2_kernel(
    const bf16_bits* __restrict__ dA, const bf16_bits* __restrict__ Z, const float* __restrict__ scale,
    const float* __restrict__ shift, const uint32_t* __restrict__ keepbits, float* __restrict__ part,
    int64_t batch, int W, int row_chunk) {
  BLH_H2_PRIO();
  __shared__ __attribute__((aligned(16))) float red[4 * 512];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col0 = blockIdx.x * H2_COLS;
  const int col = col0 + lane * 8;
  const bool ok = col < W;
  const int cc = ok ? col : 0;
  float sc[8], sh[8], s1[8], s2[8];
  ldc8(scale + cc, sc);
  ldc8(shift + cc, sh);
#pragma unroll
  for (int c = 0; c < 8; ++c) s1[c] = s2[c] = 0.f;
  const int W8 = W >> 3;
  const int64_t r0 = (int64_t)blockIdx.y * row_chunk;
  const int64_t r1 = min(batch, r0 + row_chunk);
  if (ok)
    for (int64_t rg = r0 + 4 * w; rg < r1; rg += 16) {     // wave w: 4-row groups w, w+4, ...
      uint4 zq[4], gq[4];
      const uint32_t kw = keepbits[(rg >> 2) * W8 + (col >> 3)];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t r = min(rg + j, batch - 1);
        zq[j] = *reinterpret_cast<const uint4*>(Z + r * W + col);
        gq[j] = *reinterpret_cast<const uint4*>(dA + r * W + col);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float z[8], g[8];
        unpack8(zq[j], z);
        unpack8(gq[j], g);
        const uint32_t bits = (rg + j < batch) ? (kw >> (8 * j)) : 0u;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float dy = (((bits >> c) & 1u) && (fmaf(z[c], sc[c], sh[c]) > 0.f)) ? g[c] * 2.f : 0.f;
          s2[c] += dy;
          s1[c] = fmaf(dy, z[c], s1[c]);
        }
      }
    }
  block_colsum8(s1, red, part + ((int64_t)blockIdx.y * 2 + 0) * W, col0, W);
  block_colsum8(s2, red, part + ((int64_t)blockIdx.y * 2 + 1) * W, col0, W);
}

// dgamma[c] = invstd (sum_s S1 - mean sum_s S2), dbeta[c] = sum_s S2; fp64 (the subtraction cancels
// when |mean| >> 1 / invstd).  block = 16 columns x 16 slices; a thread requests up to 8 rows of
// both sums at once (these tiny reductions are latency-bound: one round trip per batch of loads).
__global__ __launch_bounds__(256) void bn_bwd_finalize_h2_kernel(const float* __restrict__ part, int S, int W,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd,
                                                                 float* __restrict__ dgamma,
                                                                 float* __restrict__ dbeta,
                                                                 double* __restrict__ sq) {
  __shared__ double r1[16][16], r2[16][16];
  const int cl = threadIdx.x & 15, sl = threadIdx.x >> 4;
  const int col = blockIdx.x * 16 + cl;
  double a1 = 0.0, a2 = 0.0;
  // (what the storing thread needs at the end: requested with the partials, not in a round trip of its own)
  float p_mean = 0.f, p_invstd = 0.f;
  if (sl == 0 && col < W) { p_mean = mean[col]; p_invstd = invstd[col]; }
  if (col < W)
    for (int s0 = sl; s0 < S; s0 += 16 * 8) {
      float v1[8], v2[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int s = min(s0 + 16 * u, S - 1);
        v1[u] = part[((int64_t)s * 2 + 0) * W + col];
        v2[u] = part[((int64_t)s * 2 + 1) * W + col];
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (s0 + 16 * u < S) { a1 += (double)v1[u]; a2 += (double)v2[u]; }
    }
  r1[sl][cl] = a1; r2[sl][cl] = a2;
  __syncthreads();
  if (sl == 0 && col < W) {
    double t1 = 0.0, t2 = 0.0;
#pragma unroll
    for (int s = 0; s < 16; ++s) { t1 += r1[s][cl]; t2 += r2[s][cl]; }
    const float dg = (float)((double)p_invstd * (t1 - (double)p_mean * t2)), db = (float)t2;
    dgamma[col] = dg;
    dbeta[col] = db;
    if (sq) {   // sum of squares of the 32 gradients this block wrote (lanes 0-15 of wave 0 hold them)
      double q = (double)dg * (double)dg + (double)db * (double)db;
#pragma unroll
      for (int o = 8; o >= 1; o >>= 1) q += __shfl_xor(q, o);
      if (threadIdx.x == 0) sq[blockIdx.x] = q;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// PREGATED: dA already is dY' = 2 keep [y > 0] dA (written by the data-gradient GEMM's EPI_BN_BWD epilogue,
// gemm_bf16s_256.h): neither the keep bits nor the gate are needed
template <bool PREGATED>
__global__ __launch_bounds__(H2_THREADS, BLH_H2_WAVES) void bn_bwd_apply_h2_kernel(
    const bf16_bits* __restrict__ dA, const bf16_bits* __restrict__ Z, const float* __restrict__ scale,
    const float* __restrict__ shift, const float* __restrict__ mean, const float* __restrict__ invstd,
    const float* __restrict__ dgamma, const float* __restrict__ dbeta,
    const uint32_t* __restrict__ keepbits, bf16_bits* __restrict__ dZ, float* __restrict__ colsum_part,
    int64_t batch, int W, int row_chunk, int64_t norm_batch) {
  BLH_H2_PRIO();
  __shared__ __attribute__((aligned(16))) float red[4 * 512];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col0 = blockIdx.x * H2_COLS;
  const int col = col0 + lane * 8;
  const bool ok = col < W;
  const int cc = ok ? col : 0;
  float sc[8], sh[8], ca[8], cb[8], cs[8];
  ldc8(scale + cc, sc);
  ldc8(shift + cc, sh);
  {
    float mu[8], is[8], dg[8], db[8];
    ldc8(mean + cc, mu); ldc8(invstd + cc, is); ldc8(dgamma + cc, dg); ldc8(dbeta + cc, db);
    const float inv_b = 1.0f / (float)norm_batch;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float c1 = db[c] * inv_b, c2 = dg[c] * inv_b;
      const float t = sc[c] * c2 * is[c];
      ca[c] = -t;
      cb[c] = fmaf(t, mu[c], -sc[c] * c1);
      cs[c] = 0.f;
    }
  }
  const int W8 = W >> 3;
  const int64_t r0 = (int64_t)blockIdx.y * row_chunk;
  const int64_t r1 = min(batch, r0 + row_chunk);
  if (ok)
    for (int64_t rg = r0 + 4 * w; rg < r1; rg += 16) {
      uint4 zq[4], gq[4];
      const uint32_t kw = PREGATED ? 0u : keepbits[(rg >> 2) * W8 + (col >> 3)];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int64_t r = min(rg + j, batch - 1);
        zq[j] = *reinterpret_cast<const uint4*>(Z + r * W + col);
        gq[j] = *reinterpret_cast<const uint4*>(dA + r * W + col);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float z[8], g[8], o[8];
        unpack8(zq[j], z);
        unpack8(gq[j], g);
        const uint32_t bits = kw >> (8 * j);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
          const float dy = PREGATED ? g[c]
                                    : ((((bits >> c) & 1u) && (fmaf(z[c], sc[c], sh[c]) > 0.f)) ? g[c] * 2.f : 0.f);
          o[c] = fmaf(sc[c], dy, fmaf(ca[c], z[c], cb[c]));
        }
        if (rg + j < batch) {
          const uint4 q = pack8(o);
          *reinterpret_cast<uint4*>(dZ + (rg + j) * W + col) = q;
          float st[8];
          unpack8(q, st);                         // the bias gradient sums what the wgrad GEMM reads
#pragma unroll
          for (int c = 0; c < 8; ++c) cs[c] += st[c];
        }
      }
    }
  block_colsum8(cs, red, colsum_part + (int64_t)blockIdx.y * W, col0, W);
}

// ---- host --------------------------------------------------------------------------------------
static dim3 h2_grid(int64_t batch, int W) {
  return dim3((unsigned)ceil_div(W, H2_COLS), (unsigned)ew_num_row_chunks_h(batch));
}

int64_t bn_keepbits_words(int64_t batch, int W) { return ceil_div(batch, 4) * (int64_t)(W / 8); }

int launch_bn_apply_h2(hipStream_t s, bool train, const uint16_t* Z, const float* scale, const float* shift,
                       const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, const uint16_t* skip, uint16_t* A, uint32_t* keepbits,
                       int64_t batch, int W, const DropoutSrc& drop, int64_t* nbt) {
  if (W % 8 != 0) return BLH_ERR_SHAPE;
  const int rc = ew_row_chunk_h(batch);
  if (train)
    hipLaunchKernelGGL(bn_apply_h2_kernel<true>, h2_grid(batch, W), dim3(H2_THREADS), 0, s, Z, scale, shift,
                       gamma, beta, running_mean, running_var, skip, A, keepbits, batch, W, rc, drop, nbt);
  else
    hipLaunchKernelGGL(bn_apply_h2_kernel<false>, h2_grid(batch, W), dim3(H2_THREADS), 0, s, Z, scale, shift,
                       gamma, beta, running_mean, running_var, skip, A, keepbits, batch, W, rc, drop, nbt);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

int launch_bn_bwd_reduce_h2(hipStream_t s, const uint16_t* dA, const uint16_t* Z, const float* scale,
                            const float* shift, const uint32_t* keepbits, float* part, int64_t batch, int W) {
  if (W % 8 != 0) return BLH_ERR_SHAPE;
  hipLaunchKernelGGL(bn_bwd_reduce_h2_kernel, h2_grid(batch, W), dim3(H2_THREADS), 0, s, dA, Z, scale, shift,
                     keepbits, part, batch, W, ew_row_chunk_h(batch));
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

int bn_bwd_finalize_blocks(int W) { return (int)ceil_div(W, 16); }

int launch_bn_bwd_finalize_h2(hipStream_t s, const float* part, int chunks, int W, const float* mean,
                              const float* invstd, float* dgamma, float* dbeta, double* sq) {
  if (sq && W % 16 != 0) return BLH_ERR_SHAPE;     // (a ragged last block would reduce over idle lanes)
  hipLaunchKernelGGL(bn_bwd_finalize_h2_kernel, dim3((unsigned)ceil_div(W, 16)), dim3(256), 0, s, part, chunks,
                     W, mean, invstd, dgamma, dbeta, sq);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

int launch_bn_bwd_apply_h2(hipStream_t s, const uint16_t* dA, const uint16_t* Z, const float* scale,
                           const float* shift, const float* mean, const float* invstd, const float* dgamma,
                           const float* dbeta, const uint32_t* keepbits, uint16_t* dZ, float* colsum_part,
                           int64_t batch, int W, int64_t norm_batch, bool pregated) {
  if (W % 8 != 0) return BLH_ERR_SHAPE;
  if (pregated)
    launch_kernel(bn_bwd_apply_h2_kernel<true>, h2_grid(batch, W), dim3(H2_THREADS), 0, s, dA, Z, scale, shift, mean,
                  invstd, dgamma, dbeta, keepbits, dZ, colsum_part, batch, W, ew_row_chunk_h(batch), norm_batch);
  else
    launch_kernel(bn_bwd_apply_h2_kernel<false>, h2_grid(batch, W), dim3(H2_THREADS), 0, s, dA, Z, scale, shift, mean,
                  invstd, dgamma, dbeta, keepbits, dZ, colsum_part, batch, W, ew_row_chunk_h(batch), norm_batch);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

}  // namespace blh
