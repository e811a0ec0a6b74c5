// The encode stage of the lifter without its pre-BatchNorm tensor (fp32 storage, exact fp32 arithmetic).
// (/root/reference/model/bilinear.py:22 `heavy_linear(16 * 2, 1024)`, :34)
//
// z = x W0^T + b0 has only 32 inputs, so everything BatchNorm needs from Z0 [B][W] follows from quantities of x:
//   forward   the batch mean and variance of column j are  w_j . xbar + b_j  and  w_j^T Cov(x) w_j  (fp64 from the
//             33 x 32 sums of x: enc_xstats, on the matrix cores, + enc_bn_finalize), so ONE kernel maps x -> A0 = 2 keep relu(z scale + shift)
//             and the keep bits (enc_fwd): no 16.8 MB Z0 write, no re-read by a bn_apply pass.
//   backward  the encode Linear has no data gradient; what is wanted are dgamma, dbeta, db0 and
//             dW0 = dZ0^T X with dZ0 = scale dY' + a z + b' (bn_f32.hip).  The forward leaves keep AND [y > 0] as the
//             stage's "keep" bits, so dY' = 2 dA0 keep' needs no z; P = dY'^T X is a contraction over the batch
//             (enc_bwd: partials per row block, with S2 = sum dY'), and the rest is analytic again (enc_bwd_finish):
//                 S1 = sum dY' z = W0 . P + b0 S2            (row-wise dot of W0 and P)
//                 z^T X = W0 (X^T X) + b0 xs^T,             sum z = W0 xs + B b0
//                 dW0 = diag(scale) P + diag(a) (z^T X) + b' xs^T,     db0 = scale S2 + a sum(z) + b' B.
//             The stage reads dA0 once and nothing else of size [B][W] — the streaming path read dA0 and Z0 twice
//             each, wrote dZ0 and read it again for the weight-gradient GEMM — and issues 32 MFMAs per 16 x 64
//             tile (it runs beside the side stream's weight-gradient GEMM, which owns the matrix pipes: the first
//             form re-computed z with 32 more and took 59 us there).
//
// MFMA layouts (v_mfma_f32_16x16x4_f32; lane l: n = l & 15, q = l >> 4).  A wave owns 64 columns c0 .. c0 + 63:
//   z tile [16 rows][64 cols]: A operand = x (lane: row n, float4 of k = 16 h + 4 q .. + 3: component j feeds MFMA j,
//     a permutation of the contraction index used on both operands), B operand = W0 (lane: float4 of row
//     c0 + 4 n + jj, same k): MFMA group jj produces columns c0 + 4 n + jj, so acc[jj][reg] of a lane are rows
//     4 q + reg x FOUR CONSECUTIVE columns c0 + 4 n .. + 3 — one 16-byte access per row for A0 / dA0 / keep bits.
//   dY'^T X: the z tile's layout (column on the lane, rows in q and the registers) — in which the backward kernel
//     loads dA0 and forms dY' — is exactly a B operand whose
//     contraction index is the row (cdna_hip_programming.md 3, "An accumulator tile as the next MFMA's operand"):
//     step reg contracts rows 4 q + reg; A operand = x^T (lane: feature 16 h + n, row 4 q + reg).  The result holds
//     features 16 h + 4 q + reg' of column c0 + 4 n + jj: 16-byte stores into the [col][32] partial.
#include <type_traits>

#include "common.h"
#include "philox.h"
#include "bn_f32_dev.h"
#include "bn_stats_dev.h"

namespace blh {

typedef float encf4 __attribute__((ext_vector_type(4)));
typedef uint16_t enc_bf16;

// H = bf16 storage (gemm_dtype 4): x (the cast input xh), W0 (the parameter shadow), A0 and dA0 are bf16 in memory;
// the arithmetic is the fp32 path's on the exactly-converted values (products of two bf16 numbers are exact in
// fp32: the contraction is what the bf16 MFMA path computes, up to the order of the fp32 sums), z is rounded to
// bf16 before BatchNorm normalises it — the stored Z of the GEMM path is what bn_apply_h2 reads.
template <bool H> struct EncT { typedef float T; };
template <> struct EncT<true> { typedef enc_bf16 T; };
__device__ __forceinline__ float enc_ld1(const float* p) { return *p; }
__device__ __forceinline__ float enc_ld1(const enc_bf16* p) { return __uint_as_float((uint32_t)*p << 16); }
__device__ __forceinline__ float4 enc_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float4 enc_ld4(const enc_bf16* p) {
  const uint2 v = *reinterpret_cast<const uint2*>(p);
  return make_float4(__uint_as_float(v.x << 16), __uint_as_float(v.x & 0xffff0000u), __uint_as_float(v.y << 16),
                     __uint_as_float(v.y & 0xffff0000u));
}
__device__ __forceinline__ float enc_round_bf16(float v) {
  const __bf16 b = (__bf16)v;
  return __uint_as_float((uint32_t)(*reinterpret_cast<const uint16_t*>(&b)) << 16);
}
__device__ __forceinline__ uint32_t enc_pack2(float lo, float hi) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 b = {(__bf16)lo, (__bf16)hi};          // v_cvt_pk_bf16_f32: RNE
  return *reinterpret_cast<const uint32_t*>(&b);
}
__device__ __forceinline__ void enc_st4(float* p, const float4& a) { *reinterpret_cast<float4*>(p) = a; }
__device__ __forceinline__ void enc_st4(enc_bf16* p, const float4& a) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  const bf2 lo = {(__bf16)a.x, (__bf16)a.y}, hi = {(__bf16)a.z, (__bf16)a.w};
  uint2 o;
  o.x = *reinterpret_cast<const uint32_t*>(&lo); o.y = *reinterpret_cast<const uint32_t*>(&hi);
  *reinterpret_cast<uint2*>(p) = o;
}

static constexpr int ENC_IF = 32;                        // input features (16 joints x 2)
static constexpr int ENC_XN = ENC_IF + ENC_IF * ENC_IF;  // colsum(x) | X^T X
static constexpr int ENC_XBLOCKS = 32;                   // at most this many row blocks of enc_xstats: every block of
                                                         // enc_bn_finalize adds them all, 16 per round trip (64 at
                                                         // B = 16384 were four rounds: 12.8 us against 6 at B = 4096)
static constexpr int ENC_XWAVES = 4;                     // waves per block of enc_xstats
static constexpr int ENC_XROWS = 64 * ENC_XWAVES;        // rows per block of enc_xstats: a multiple of this (two rounds of
                                                         // 32 rows per wave, requested together)

// ---- x statistics -------------------------------------------------------------------------------------------
// Moments of y = x - c with c = row 0 of the batch (the shifted-data form: the variance below is a difference of
// second moments, and moments of the raw x lose log2(1 + mean^2 / var) bits to that difference — inputs that are not
// standardised would pay for it; any row of the batch is within a few standard deviations of the mean):
// xpart[b][0 .. 31] = sum_r y[r][f], xpart[b][32 + a * 32 + c] = sum_r y[r][a] y[r][c] over the block's rows (fp32
// sums over at most a few hundred rows per wave; the partials are added in fp64).  Y^T Y on the matrix
// cores: per 4 rows both operands are the same registers — A = x^T (lane: feature n + 16 h, row 4 s + q), B = x
// (lane: row 4 s + q, feature n + 16 h') — 4 MFMAs per 4 rows.  (Sixteen waves per block and at most 16 blocks — so
// that the finalize kernel had one round of partials — measured 7-12 us against 5: r06, profiles/r06_encode.md.)
// CAST (bf16 storage with the weight image kept by Adam: no cast launch in front of the forward): x is the caller's fp32
// input; every element is rounded to bf16 here, the statistics are those of the rounded values and the bf16 image
// xh — what every later kernel of the stage reads — is written on the way (each element is touched exactly once).
template <typename TX, bool CAST = false>
__global__ __launch_bounds__(64 * ENC_XWAVES) void enc_xstats_kernel(const TX* __restrict__ x, int64_t batch,
                                                                     int rows_per_block, float* __restrict__ xpart,
                                                                     enc_bf16* __restrict__ xh = nullptr) {
  __shared__ float red[ENC_XWAVES][ENC_XN];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int64_t r0 = (int64_t)blockIdx.x * rows_per_block;
  const int64_t r1 = min(batch, r0 + rows_per_block);
  encf4 acc[2][2];
  acc[0][0] = acc[0][1] = acc[1][0] = acc[1][1] = encf4{0.f, 0.f, 0.f, 0.f};
  float cs0 = 0.f, cs1 = 0.f;
  float c0 = enc_ld1(x + n), c1 = enc_ld1(x + 16 + n);                // the shift: row 0
  if (CAST) { c0 = enc_round_bf16(c0); c1 = enc_round_bf16(c1); }
  // rounds of 8 steps of 4 rows, TWO rounds per trip with all their loads first (the launch plans at most two rounds
  // per wave up to 16384 rows: one memory round trip for the whole kernel)
  for (int64_t base = r0 + 64 * wave; base < r1; base += ENC_XROWS) {
    float xv[2][8][2];
#pragma unroll
    for (int rd = 0; rd < 2; ++rd)
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        // (unconditional loads — the row is clamped — and a select afterwards: a load under a lane mask compiles to
        //  branch + load + full wait, sixteen dependent round trips: 8.7 instead of 5 us)
        const TX* xr = x + min(base + 32 * rd + 4 * st + q, batch - 1) * ENC_IF + n;
        xv[rd][st][0] = enc_ld1(xr);
        xv[rd][st][1] = enc_ld1(xr + 16);
      }
    // (nothing below may move above this line: left to itself the scheduler sinks each load to its use and waits for
    //  it there — 32 dependent round trips, 12 us)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rd = 0; rd < 2; ++rd)
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int64_t row = base + 32 * rd + 4 * st + q;
        const bool ok = row < r1;
        if (CAST) {
          const uint32_t pk = enc_pack2(xv[rd][st][0], xv[rd][st][1]);
          xv[rd][st][0] = __uint_as_float(pk << 16); xv[rd][st][1] = __uint_as_float(pk & 0xffff0000u);
          if (ok) { xh[row * ENC_IF + n] = (enc_bf16)(pk & 0xFFFFu); xh[row * ENC_IF + 16 + n] = (enc_bf16)(pk >> 16); }
        }
        const float y0 = ok ? xv[rd][st][0] - c0 : 0.f, y1 = ok ? xv[rd][st][1] - c1 : 0.f;
        const float yy[2] = {y0, y1};
        cs0 += y0; cs1 += y1;
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int h2 = 0; h2 < 2; ++h2)
            acc[h][h2] = __builtin_amdgcn_mfma_f32_16x16x4f32(yy[h], yy[h2], acc[h][h2], 0, 0, 0);
      }
  }
  cs0 += __shfl_xor(cs0, 16); cs0 += __shfl_xor(cs0, 32);
  cs1 += __shfl_xor(cs1, 16); cs1 += __shfl_xor(cs1, 32);
  if (q == 0) { red[wave][n] = cs0; red[wave][16 + n] = cs1; }
  // acc[h][h2][reg] = XtX[a = 16 h + 4 q + reg][c = 16 h2 + n]
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) red[wave][ENC_IF + (16 * h + 4 * q + reg) * ENC_IF + 16 * h2 + n] = acc[h][h2][reg];
  __syncthreads();
  float* out = xpart + (int64_t)blockIdx.x * ENC_XN;
  for (int e = threadIdx.x; e < ENC_XN; e += 256) out[e] = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
}

// Sums the partials and finishes BatchNorm's forward statistics of 16 columns per block (thread = column t >> 4,
// features g = t & 15 and g + 16); leaves for the backward, per column, T[col][f] = (z^T X)[col][f] =
// sum_k W0[col][k] XtX[k][f] + b0[col] xs[f] and zs[col] = sum of z over the batch, and xs[32] (block 0).
// With d = sum y, S = Y^T Y (y = x - c, enc_xstats):  xs = d + B c,  XtX = S + c d^T + d c^T + B c c^T,
// sum (z - mean)^2 = w^T S w - (w . d)^2 / B  (neither c nor the bias appears in it).
template <typename TW>
__global__ __launch_bounds__(256) void enc_bn_finalize_kernel(
    const TW* __restrict__ x, const float* __restrict__ xpart, int nparts, float* __restrict__ xs_out, float* __restrict__ ttab,
    float* __restrict__ zs_out, const TW* __restrict__ W0, const float* __restrict__ b0, int64_t batch, int W,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean, float* running_var,
    const int64_t* nbt, float momentum, float* saved_mean, float* saved_invstd, float* scale, float* shift) {
  __shared__ double xs[ENC_XN];      // d | S
  __shared__ double cc[ENC_IF];      // c
  const int t = threadIdx.x;
  if (t < ENC_IF) cc[t] = (double)enc_ld1(x + t);
  // this thread's column; its bias and its own two weights do not depend on the partial sums below — requested first
  const int col = blockIdx.x * 16 + (t >> 4), g = t & 15;
  const bool ok = col < W;
  const TW* w = W0 + (int64_t)(ok ? col : 0) * ENC_IF;
  const float b_f = b0[ok ? col : 0];
  const float wf0_f = enc_ld1(w + g), wf1_f = enc_ld1(w + g + 16);     // this thread's features (unconditional loads)
  // (what the storing thread of a column reads at the end: requested now, one round trip less at the tail)
  BnColumnIn pre{};
  {
    const int pcol = blockIdx.x * 16 + (t >> 4);
    if ((t & 15) == 0 && pcol < W) pre = bn_finalize_prefetch(pcol, gamma, beta, running_mean, running_var);
  }
  {
    // (the loads of 16 partials of all of the thread's entries are requested together: nparts / 16 round trips)
    constexpr int NE = (ENC_XN + 255) / 256;
    double tot[NE];
#pragma unroll
    for (int i = 0; i < NE; ++i) tot[i] = 0.0;
    for (int p0 = 0; p0 < nparts; p0 += 16) {
      float v[NE][16];
#pragma unroll
      for (int i = 0; i < NE; ++i)
#pragma unroll
        for (int p = 0; p < 16; ++p)
          v[i][p] = xpart[(int64_t)min(p0 + p, nparts - 1) * ENC_XN + min(t + 256 * i, ENC_XN - 1)];
#pragma unroll
      for (int i = 0; i < NE; ++i)
#pragma unroll
        for (int p = 0; p < 16; ++p) tot[i] += p0 + p < nparts ? (double)v[i][p] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < NE; ++i) {
      const int e = t + 256 * i;
      if (e < ENC_XN) xs[e] = tot[i];
    }
  }
  __syncthreads();
  const double B = (double)batch;
  if (blockIdx.x == 0 && t < ENC_IF) xs_out[t] = (float)(xs[t] + B * cc[t]);
  // (the whole weight row: 16 lanes of a column read the same 128 bytes; requested here, behind the partial sums —
  //  in front of them it only delayed their arrival: 7.1 against 6.0-6.5 us)
  float wr[ENC_IF];
#pragma unroll
  for (int k = 0; k < ENC_IF; k += 4) {
    const float4 v = enc_ld4(w + k);
    wr[k] = v.x; wr[k + 1] = v.y; wr[k + 2] = v.z; wr[k + 3] = v.w;
  }
  const double b = ok ? (double)b_f : 0.0;
  // u[f] = sum_k w[k] S[k][f] for f = g, g + 16;   w^T S w = sum_f w[f] u[f];   w . d;   w . c
  double u0 = 0.0, u1 = 0.0;
#pragma unroll
  for (int k = 0; k < ENC_IF; ++k) {
    u0 = fma((double)wr[k], xs[ENC_IF + k * ENC_IF + g], u0);
    u1 = fma((double)wr[k], xs[ENC_IF + k * ENC_IF + g + 16], u1);
  }
  const double wf0 = ok ? (double)wf0_f : 0.0, wf1 = ok ? (double)wf1_f : 0.0;
  double quad = wf0 * u0 + wf1 * u1, dotd = wf0 * xs[g] + wf1 * xs[g + 16], dotc = wf0 * cc[g] + wf1 * cc[g + 16];
#pragma unroll
  for (int o = 8; o >= 1; o >>= 1) {
    quad += __shfl_xor(quad, o); dotd += __shfl_xor(dotd, o); dotc += __shfl_xor(dotc, o);
  }
  if (ok) {
    // T[f] = (w XtX)[f] + b xs[f] = u[f] + (w.c) d[f] + (w.d) c[f] + B (w.c) c[f] + b (d[f] + B c[f])
    const double kc = dotd + B * dotc + B * b;               // = sum of z: multiplies c[f]
    ttab[(int64_t)col * ENC_IF + g] = (float)(u0 + (dotc + b) * xs[g] + kc * cc[g]);
    ttab[(int64_t)col * ENC_IF + g + 16] = (float)(u1 + (dotc + b) * xs[g + 16] + kc * cc[g + 16]);
    if (g == 0) {
      zs_out[col] = (float)kc;
      const double m2 = quad - dotd * dotd / B;              // sum (z - mean)^2
      bn_finalize_store(dotd / B + dotc + b, m2 > 0.0 ? m2 : 0.0, batch, col, pre, running_mean, running_var,
                        nbt, momentum, saved_mean, saved_invstd, scale, shift);
    }
  }
}

// ---- operands shared by the forward and the backward kernel ---------------------------------------------------
// A lane owns CPL consecutive columns: 4 with fp32 storage (one float4 per row), 8 with bf16 storage (16 bytes per
// row as well — with 4 columns a bf16 lane moved 8 bytes per access and the forward took 26 us for 33.5 MB).
template <int CPL>
struct EncCols {
  float4 wb[CPL][2];                 // W0 rows col + jj, k = 16 h + 4 q .. + 3
  float bias[CPL], sc[CPL], sh[CPL]; // columns col .. col + CPL - 1
};
template <int CPL, typename TW>
__device__ __forceinline__ void enc_load_cols(EncCols<CPL>& c, const TW* __restrict__ W0, const float* __restrict__ b0,
                                              const float* __restrict__ scale, const float* __restrict__ shift,
                                              int col, int q) {
#pragma unroll
  for (int jj = 0; jj < CPL; ++jj)
#pragma unroll
    for (int h = 0; h < 2; ++h)
      c.wb[jj][h] = enc_ld4(W0 + (int64_t)(col + jj) * ENC_IF + 16 * h + 4 * q);
#pragma unroll
  for (int jj = 0; jj < CPL; jj += 4) {
    const float4 bv = *reinterpret_cast<const float4*>(b0 + col + jj), sv = *reinterpret_cast<const float4*>(scale + col + jj),
                 hv = *reinterpret_cast<const float4*>(shift + col + jj);
    c.bias[jj] = bv.x; c.bias[jj + 1] = bv.y; c.bias[jj + 2] = bv.z; c.bias[jj + 3] = bv.w;
    c.sc[jj] = sv.x; c.sc[jj + 1] = sv.y; c.sc[jj + 2] = sv.z; c.sc[jj + 3] = sv.w;
    c.sh[jj] = hv.x; c.sh[jj + 1] = hv.y; c.sh[jj + 2] = hv.z; c.sh[jj + 3] = hv.w;
  }
}
// z of the 16-row tile at `base`: zt[jj][reg] = row base + 4 q + reg, column col + jj (without the bias)
template <int CPL, typename TX>
__device__ __forceinline__ void enc_z_tile(encf4 (&zt)[CPL], const EncCols<CPL>& c, const TX* __restrict__ x,
                                           int64_t base, int64_t batch, int n, int q) {
  const TX* xr = x + min(base + n, batch - 1) * ENC_IF + 4 * q;
  const float4 xa0 = enc_ld4(xr), xa1 = enc_ld4(xr + 16);
#pragma unroll
  for (int jj = 0; jj < CPL; ++jj) {
    encf4 a = encf4{0.f, 0.f, 0.f, 0.f};
    a = __builtin_amdgcn_mfma_f32_16x16x4f32(xa0.x, c.wb[jj][0].x, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x4f32(xa0.y, c.wb[jj][0].y, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x4f32(xa0.z, c.wb[jj][0].z, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x4f32(xa0.w, c.wb[jj][0].w, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x4f32(xa1.x, c.wb[jj][1].x, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x4f32(xa1.y, c.wb[jj][1].y, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x4f32(xa1.z, c.wb[jj][1].z, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x4f32(xa1.w, c.wb[jj][1].w, a, 0, 0, 0);
    zt[jj] = a;
  }
}
__device__ __forceinline__ void enc_ld_f8(const float* p, float (&f)[8]) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
// the same from x operands already in registers
template <int CPL>
__device__ __forceinline__ void enc_z_tile_x(encf4 (&zt)[CPL], const EncCols<CPL>& c, const float4& xa0, const float4& xa1) {
#pragma unroll
  for (int jj = 0; jj < CPL; ++jj) {
    encf4 a = encf4{0.f, 0.f, 0.f, 0.f};
    a = __builtin_amdgcn_mfma_f32_16x16x4f32(xa0.x, c.wb[jj][0].x, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x4f32(xa0.y, c.wb[jj][0].y, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x4f32(xa0.z, c.wb[jj][0].z, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x4f32(xa0.w, c.wb[jj][0].w, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x4f32(xa1.x, c.wb[jj][1].x, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x4f32(xa1.y, c.wb[jj][1].y, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x4f32(xa1.z, c.wb[jj][1].z, a, 0, 0, 0);
    a = __builtin_amdgcn_mfma_f32_16x16x4f32(xa1.w, c.wb[jj][1].w, a, 0, 0, 0);
    zt[jj] = a;
  }
}
// CPL values of one row, 16 bytes: a float4 (fp32 storage) or 8 bf16 (bf16 storage)
__device__ __forceinline__ void enc_st_row(float* p, const float (&a)[4]) {
  *reinterpret_cast<float4*>(p) = make_float4(a[0], a[1], a[2], a[3]);
}
__device__ __forceinline__ void enc_st_row(enc_bf16* p, const float (&a)[8]) {
  typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
  uint4 o;
  const bf2 v0 = {(__bf16)a[0], (__bf16)a[1]}, v1 = {(__bf16)a[2], (__bf16)a[3]}, v2 = {(__bf16)a[4], (__bf16)a[5]},
            v3 = {(__bf16)a[6], (__bf16)a[7]};
  o.x = *reinterpret_cast<const uint32_t*>(&v0); o.y = *reinterpret_cast<const uint32_t*>(&v1);
  o.z = *reinterpret_cast<const uint32_t*>(&v2); o.w = *reinterpret_cast<const uint32_t*>(&v3);
  *reinterpret_cast<uint4*>(p) = o;
}
__device__ __forceinline__ void enc_ld_row(const float* p, float (&a)[4]) {
  const float4 v = *reinterpret_cast<const float4*>(p);
  a[0] = v.x; a[1] = v.y; a[2] = v.z; a[3] = v.w;
}
__device__ __forceinline__ void enc_ld_row(const enc_bf16* p, float (&a)[8]) {
  const uint4 v = *reinterpret_cast<const uint4*>(p);
  a[0] = __uint_as_float(v.x << 16); a[1] = __uint_as_float(v.x & 0xffff0000u);
  a[2] = __uint_as_float(v.y << 16); a[3] = __uint_as_float(v.y & 0xffff0000u);
  a[4] = __uint_as_float(v.z << 16); a[5] = __uint_as_float(v.z & 0xffff0000u);
  a[6] = __uint_as_float(v.w << 16); a[7] = __uint_as_float(v.w & 0xffff0000u);
}

// ---- forward: x -> A0 (+ keep bits) ----------------------------------------------------------------------------
// block = 4 waves x 16 CPL columns, ONE 32-row Philox patch of rows (two 16-row tiles).  Blocks are numbered so that
// the column blocks of one row block sit on the same XCD (the hardware deals consecutive workgroup ids round-robin
// over the 8 XCDs): x — re-read by every column block — then comes from HBM once, not once per XCD L2 (PMC, r05:
// 1.20 x the algorithmic bytes in the forward, 1.48 x in the backward, most of it these re-reads).
//   id -> xcd = id & 7, slot = id >> 3, column block cb = slot % ncb, row block rb = 8 (slot / ncb) + xcd
__device__ __forceinline__ bool enc_block(int ncb, int64_t nrb, int& cb, int64_t& rb) {
  const int id = blockIdx.x, xcd = id & 7, slot = id >> 3;
  cb = slot % ncb;
  rb = (int64_t)(slot / ncb) * 8 + xcd;
  return rb < nrb;
}
static unsigned enc_grid(int ncb, int64_t nrb) { return (unsigned)(ncb * round_up(nrb, 8)); }

// Keep-bit words: fp32 storage [B/8][W/4] (nibble j = row 8 g + j, bit c = column 4 (col/4) + c: bn_f32.hip) — a lane's
// 4 rows x 4 columns are half a word, the other half is lane l ^ 16.
__global__ __launch_bounds__(256) void enc_fwd_kernel(const float* __restrict__ x, const float* __restrict__ W0,
                                                      const float* __restrict__ b0, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, float* __restrict__ A,
                                                      uint32_t* __restrict__ keepbits, int64_t batch, int W, int ncb,
                                                      DropoutSrc drop, int64_t* nbt) {
  constexpr int CPL = 4;
  int cb; int64_t rb;
  if (!enc_block(ncb, (batch + 31) >> 5, cb, rb)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int col = (cb * 4 + wave) * (16 * CPL) + CPL * n;
  if (nbt && blockIdx.x == 0 && threadIdx.x == 0) nbt[0] += 1;
  EncCols<CPL> c;
  enc_load_cols<CPL>(c, W0, b0, scale, shift, col, q);
#pragma unroll
  for (int jj = 0; jj < CPL; ++jj) { c.sc[jj] *= 2.f; c.sh[jj] *= 2.f; }      // a = 2 y: exact, folded into the FMA
  const int64_t r0 = rb * 32;
  // both tiles' x rows are requested before the first MFMA
  float4 xa[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const float* xr = x + min(r0 + 16 * t + n, batch - 1) * ENC_IF + 4 * q;
    xa[t][0] = enc_ld4(xr); xa[t][1] = enc_ld4(xr + 16);
  }
  // FULL: every row of the block exists (all blocks but the last of a ragged batch): no per-row predicates
  auto tiles = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int64_t base = r0 + 16 * t;
      if (!FULL && base >= batch) break;
      encf4 zt[CPL];
      enc_z_tile_x<CPL>(zt, c, xa[t][0], xa[t][1]);
      // Philox / mask bits of the 8-row group this lane's rows belong to (nibble 4 (q & 1) + reg = row base + 4 q + reg),
      // one word per 4 columns
      const int64_t rg = base + 8 * (q >> 1);
      const uint32_t kw = f2_keep_word(drop, r0, 2 * t + (q >> 1), col, W, batch) >> (16 * (q & 1));
      uint32_t word = 0;            // keep AND [y > 0]: nibble reg of this lane's half word
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int64_t row = base + 4 * q + reg;
        float a[CPL];
        uint32_t bits = 0;
#pragma unroll
        for (int jj = 0; jj < CPL; ++jj) {
          const float z = zt[jj][reg] + c.bias[jj];
          const float y2 = fmaxf(fmaf(z, c.sc[jj], c.sh[jj]), 0.f);      // (scale and shift carry dropout's factor 2)
          const uint32_t av = __float_as_uint(y2) & (uint32_t)__builtin_amdgcn_sbfe(kw, 4 * reg + jj, 1);
          a[jj] = __uint_as_float(av);
          bits |= min(av, 1u) << jj;                                     // keep AND [y > 0] = [a != 0]
        }
        word |= bits << (CPL * reg);
        if (FULL || row < batch) enc_st_row(A + row * W + col, a);
      }
      const uint32_t full = (word << (16 * (q & 1))) | (__shfl_xor(word, 16) << (16 * ((q & 1) ^ 1)));
      if ((q & 1) == 0 && (FULL || rg < batch)) keepbits[(rg >> 3) * (W >> 2) + (col >> 2)] = full;
    }
  };
  if (r0 + 32 <= batch) tiles(std::true_type{});
  else tiles(std::false_type{});
}

// bf16 storage (gemm_dtype 4).  A lane owns 8 consecutive columns (16 bytes per row); z comes from ONE
// v_mfma_f32_16x16x32_bf16 per 16 rows x 16 columns (K = 32 is the whole contraction: A operand = 8 bf16 of an x row,
// B operand = 8 bf16 of a W0 row — the products are the exact products of the stored values, summed in fp32, which is
// what the bf16 GEMM of the materialised path computes; the first form ran eight exact-fp32 MFMAs per tile on the
// converted values: 16 x the matrix-pipe cycles, 30 us at B = 16384 with one workgroup per CU).  MFMA group jj produces
// columns col + jj of the lane (B operand = W0 row c0 + 8 n + jj), so acc[jj][reg] are rows 4 q + reg x the lane's 8
// columns.  z is rounded to bf16 before BatchNorm normalises it (the stored Z of the GEMM path is what bn_apply_h2
// reads).  Keep bits [B/4][W/8] (byte j = row 4 g + j, bit c = column 8 (col/8) + c: bn_bf16.hip): a lane's 4 rows x 8
// columns are exactly one word; the stage stores keep AND [y > 0] (= [a != 0]: a = keep ? 2 y : 0 with y >= 0).
// The block of column block 0 also leaves xT[feature][row] (zero beyond the batch), the K-major image of x the
// backward's MFMAs want as their A operand.
typedef __bf16 enc_bf16x8 __attribute__((ext_vector_type(8)));
union EncFrag { uint4 u; enc_bf16x8 v; };

__global__ __launch_bounds__(256) void enc_fwd_h_kernel(const enc_bf16* __restrict__ x, const enc_bf16* __restrict__ W0,
                                                        const float* __restrict__ b0, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, enc_bf16* __restrict__ A,
                                                        uint32_t* __restrict__ keepbits, enc_bf16* __restrict__ xT,
                                                        int64_t xt_ld, int64_t batch, int W, int ncb, DropoutSrc drop,
                                                        int64_t* nbt) {
  int cb; int64_t rb;
  if (!enc_block(ncb, (batch + 31) >> 5, cb, rb)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int col = (cb * 4 + wave) * 128 + 8 * n;
  if (nbt && blockIdx.x == 0 && threadIdx.x == 0) nbt[0] += 1;
  const int64_t base = rb * 32;
  // operands: both tiles' x rows, the eight W0 rows of the lane's columns
  EncFrag xa[2], wb[8];
#pragma unroll
  for (int t = 0; t < 2; ++t)
    xa[t].u = *reinterpret_cast<const uint4*>(x + min(base + 16 * t + n, batch - 1) * ENC_IF + 8 * q);
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) wb[jj].u = *reinterpret_cast<const uint4*>(W0 + (int64_t)(col + jj) * ENC_IF + 8 * q);
  float bias[8], sc2[8], sh2[8];
  enc_ld_f8(b0 + col, bias);
  enc_ld_f8(scale + col, sc2);
  enc_ld_f8(shift + col, sh2);
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) { sc2[jj] *= 2.f; sh2[jj] *= 2.f; }      // a = 2 y: exact, folded into the FMA
  // dropout: the block is one 32-row patch of rows; the lane's columns are two 4-column patches
  Philox128 p0{}, p1{};
  if (!drop.keep) {
    p0 = dropout_patch(drop.seed, dropout_step(drop), drop.layer, base + drop.row_offset, col);
    p1 = dropout_patch(drop.seed, dropout_step(drop), drop.layer, base + drop.row_offset, col + 4);
  }
  if (cb == 0 && wave == 0) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int64_t row = base + 16 * t + n;
      const uint32_t w[4] = {xa[t].u.x, xa[t].u.y, xa[t].u.z, xa[t].u.w};
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const uint32_t v = (e & 1) ? (w[e >> 1] >> 16) : (w[e >> 1] & 0xFFFFu);
        xT[(int64_t)(8 * q + e) * xt_ld + row] = row < batch ? (enc_bf16)v : (enc_bf16)0;
      }
    }
  }
  // FULL: every row of the block exists (all blocks but the last of a ragged batch): no per-row predicates
  auto tiles = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      encf4 acc[8];
#pragma unroll
      for (int jj = 0; jj < 8; ++jj)
        acc[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xa[t].v, wb[jj].v, encf4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
      // keep words of the 8-row group the lane's rows belong to: word 2 t + (q >> 1) of the patch
      uint32_t kw0, kw1;
      if (drop.keep) {
        kw0 = f2_keep_word(drop, base, 2 * t + (q >> 1), col, W, batch);
        kw1 = f2_keep_word(drop, base, 2 * t + (q >> 1), col + 4, W, batch);
      } else {
        kw0 = (q >> 1) ? p0.w[2 * t + 1] : p0.w[2 * t];
        kw1 = (q >> 1) ? p1.w[2 * t + 1] : p1.w[2 * t];
      }
      // the lane's 16 + 16 keep bits: nibble reg of each
      kw0 >>= 16 * (q & 1); kw1 >>= 16 * (q & 1);
      uint32_t word = 0;
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int64_t row = base + 16 * t + 4 * q + reg;
        const uint32_t kb = ((kw0 >> (4 * reg)) & 0xFu) | (((kw1 >> (4 * reg)) & 0xFu) << 4);   // keep bits of the row's 8 columns
        float a[8];
        uint32_t rowbits = 0;
#pragma unroll
        for (int jj = 0; jj < 8; jj += 2) {
          // (two columns at a time: one v_cvt_pk_bf16_f32 rounds both)
          const uint32_t zz = enc_pack2(acc[jj][reg] + bias[jj], acc[jj + 1][reg] + bias[jj + 1]);
          const float z0 = __uint_as_float(zz << 16), z1 = __uint_as_float(zz & 0xffff0000u);
          const float y0 = fmaxf(fmaf(z0, sc2[jj], sh2[jj]), 0.f), y1 = fmaxf(fmaf(z1, sc2[jj + 1], sh2[jj + 1]), 0.f);
          const int m0 = __builtin_amdgcn_sbfe(kb, jj, 1), m1 = __builtin_amdgcn_sbfe(kb, jj + 1, 1);      // 0 or ~0
          const uint32_t a0 = __float_as_uint(y0) & (uint32_t)m0, a1 = __float_as_uint(y1) & (uint32_t)m1;
          a[jj] = __uint_as_float(a0); a[jj + 1] = __uint_as_float(a1);
          rowbits |= (min(a0, 1u) << jj) | (min(a1, 1u) << (jj + 1));
        }
        if (FULL || row < batch) {
          enc_st_row(A + row * W + col, a);
          word |= rowbits << (8 * reg);
        }
      }
      const int64_t rg4 = base + 16 * t + 4 * q;
      if (FULL || rg4 < batch) keepbits[(rg4 >> 2) * (W >> 3) + (col >> 3)] = word;
    }
  };
  if (base + 32 <= batch) tiles(std::true_type{});
  else tiles(std::false_type{});
}

// ---- backward: dA0 -> S2 = sum dY' and P = dY'^T X, partials per row block ------------------------------------------
// block = NW waves on the SAME 16 CPL columns, wave w takes the tiles w, w + NW, ... of the block's rows; blocks are
// numbered as in the forward (the column blocks of a row block on one XCD: x comes from HBM once).  The partials cost
// 2 x 33 floats per column and row block (written here, read by the finish kernel): rows_per_block is as large as the
// launch still fills the chip with (host side: enc_bwd_plan) — at 128 rows per block they were half of the kernel's
// traffic (PMC r05: 1.48 x its algorithmic bytes, 1.76 x with the finish kernel).
template <int NW>
__global__ __launch_bounds__(64 * NW) void enc_bwd_kernel(const float* __restrict__ dA, const float* __restrict__ x,
                                                          const uint32_t* __restrict__ gatebits,
                                                          float* __restrict__ s2part, float* __restrict__ ppart,
                                                          int64_t batch, int W, int rows_per_block, int ncb, int nrb) {
  constexpr int CPL = 4, BC = 16 * CPL;
  __builtin_amdgcn_s_setprio(3);
  // 9 KiB of LDS: the kernel has to fit BESIDE a workgroup of the side stream's weight-gradient GEMM
  // (128 KiB of the CU's 160) — with a 32 KiB buffer per block its workgroups waited for the GEMM's to retire: 75 us
  __shared__ __attribute__((aligned(16))) float red[BC * ENC_IF];        // [columns][32 features], waves add in turn
  __shared__ float sred[NW][BC];
  int cb; int64_t rb;
  if (!enc_block(ncb, nrb, cb, rb)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int c0 = cb * BC;
  const int col = c0 + CPL * n;
  const int64_t r0 = rb * rows_per_block;
  const int64_t r1 = min(batch, r0 + rows_per_block);
  encf4 pacc[CPL][2];
#pragma unroll
  for (int jj = 0; jj < CPL; ++jj) pacc[jj][0] = pacc[jj][1] = encf4{0.f, 0.f, 0.f, 0.f};
  float s2[CPL];
#pragma unroll
  for (int jj = 0; jj < CPL; ++jj) s2[jj] = 0.f;
  // one 16-row tile: the lane's 4 rows x 4 columns of dA0, its half keep-and-gate word, the x^T operand
  struct Tile { float g[4][CPL]; uint32_t bits; float xt[4][2]; };
  auto load = [&](int64_t base) {
    Tile t;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) enc_ld_row(dA + min(base + 4 * q + reg, batch - 1) * W + col, t.g[reg]);
    const int64_t rg = base + 8 * (q >> 1);
    const uint32_t kw = rg < batch ? gatebits[(rg >> 3) * (W >> 2) + (col >> 2)] : 0u;
    t.bits = (kw >> (16 * (q & 1))) & 0xFFFFu;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {      // feature 16 h + n of row base + 4 q + reg
      const float* xr = x + min(base + 4 * q + reg, batch - 1) * ENC_IF + n;
      t.xt[reg][0] = xr[0]; t.xt[reg][1] = xr[16];
    }
    return t;
  };
  int64_t base = r0 + 16 * wave;
  if (base < r1) {
    Tile cur = load(base);
    for (; base < r1; base += 16 * NW) {
      // (the next tile is requested before this one's 32 MFMAs: one memory round trip per tile was the kernel's time)
      const int64_t nb = base + 16 * NW;
      Tile nxt = load(nb < r1 ? nb : base);
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const uint32_t rbits = (base + 4 * q + reg < batch) ? (cur.bits >> (CPL * reg)) : 0u;
#pragma unroll
        for (int jj = 0; jj < CPL; ++jj) {
          const float dy = ((rbits >> jj) & 1u) ? cur.g[reg][jj] * 2.f : 0.f;     // dY' = 2 keep [y > 0] dA
          s2[jj] += dy;
          cur.g[reg][jj] = dy;
        }
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg)
#pragma unroll
        for (int jj = 0; jj < CPL; ++jj) {
          pacc[jj][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.xt[reg][0], cur.g[reg][jj], pacc[jj][0], 0, 0, 0);
          pacc[jj][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(cur.xt[reg][1], cur.g[reg][jj], pacc[jj][1], 0, 0, 0);
        }
      cur = nxt;
    }
  }
  // column sums over the lane groups q (rows), then over the waves
#pragma unroll
  for (int jj = 0; jj < CPL; ++jj) {
    s2[jj] += __shfl_xor(s2[jj], 16);
    s2[jj] += __shfl_xor(s2[jj], 32);
    if (q == 0) sred[wave][CPL * n + jj] = s2[jj];
  }
  // pacc[jj][h][reg'] = (dY'^T X)[column col + jj][feature 16 h + 4 q + reg']: the waves add their tiles into one
  // LDS image in a fixed order (every lane owns the same float4s of it in each wave)
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    if (wave == w) {
#pragma unroll
      for (int jj = 0; jj < CPL; ++jj)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float4* dst = reinterpret_cast<float4*>(&red[(CPL * n + jj) * ENC_IF + 16 * h + 4 * q]);
          float4 v = make_float4(pacc[jj][h][0], pacc[jj][h][1], pacc[jj][h][2], pacc[jj][h][3]);
          if (w > 0) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
          *dst = v;
        }
    }
    __syncthreads();
  }
  const int t = threadIdx.x;
  if (t < BC) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += sred[w][t];
    s2part[rb * W + c0 + t] = v;
  }
  float* pp = ppart + (rb * W + c0) * ENC_IF;
  for (int e = t; e < BC * ENC_IF / 4; e += 64 * NW)
    *reinterpret_cast<float4*>(pp + 4 * e) = *reinterpret_cast<const float4*>(&red[4 * e]);
}

// bf16 storage.  The contraction P = dY'^T X runs over the ROWS, 32 of them per v_mfma_f32_16x16x32_bf16: B operand =
// 8 rows of one column (k = 8 q + j), A operand = 8 rows of one feature of x — the K-major image xT[feature][row] the
// forward left (16 bytes per lane).  A lane loads rows 8 q .. 8 q + 7 x its 8 columns (eight 16-byte loads), gates them
// with the stage's keep-and-gate bits (v_pk_mul_lo_u16 by 0 / 1 per half word), transposes the 8 x 8 block of 16-bit
// values in its own registers (32 v_perm_b32) and feeds column jj's four registers to MFMA group jj.  dY' = 2 dA keep':
// the factor 2 is applied to the accumulators at the end (exact).  S2 = sum dY' comes from a third MFMA per column
// whose A operand is a row of ones (feature 32 of an augmented x): 24 MFMAs of 16 cycles per 32 rows x 128 columns,
// where the first form issued 128 exact-fp32 MFMAs of 32 cycles on the converted values.
template <int NW>
__global__ __launch_bounds__(64 * NW) void enc_bwd_h_kernel(const enc_bf16* __restrict__ dA, const enc_bf16* __restrict__ xT,
                                                            int64_t xt_ld, const uint32_t* __restrict__ gatebits,
                                                            float* __restrict__ s2part, float* __restrict__ ppart,
                                                            int64_t batch, int W, int rows_per_block, int ncb, int nrb) {
  constexpr int BC = 128;
  __builtin_amdgcn_s_setprio(3);
  __shared__ __attribute__((aligned(16))) float red[BC * ENC_IF];        // 16 KiB: [columns][32 features]
  __shared__ float sred[NW][BC];
  int cb; int64_t rb;
  if (!enc_block(ncb, nrb, cb, rb)) return;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int n = lane & 15, q = lane >> 4;
  const int c0 = cb * BC;
  const int col = c0 + 8 * n;
  const int64_t r0 = rb * rows_per_block;                       // a multiple of 32
  const int64_t r1 = min(batch, r0 + rows_per_block);
  encf4 pacc[8][3];
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) pacc[jj][0] = pacc[jj][1] = pacc[jj][2] = encf4{0.f, 0.f, 0.f, 0.f};
  EncFrag ones;
  ones.u = (n == 0) ? make_uint4(0x3F803F80u, 0x3F803F80u, 0x3F803F80u, 0x3F803F80u) : make_uint4(0, 0, 0, 0);
  struct Tile { uint4 g[8]; uint32_t w0, w1; EncFrag xa[2]; };
  auto load = [&](int64_t base) {
    Tile t;
    const int64_t rg = base + 8 * q;
#pragma unroll
    for (int j = 0; j < 8; ++j) t.g[j] = *reinterpret_cast<const uint4*>(dA + min(rg + j, batch - 1) * W + col);
    // keep-and-gate words of rows rg .. rg + 3 and rg + 4 .. rg + 7 (the forward wrote zero bits beyond the batch)
    t.w0 = rg < batch ? gatebits[(rg >> 2) * (W >> 3) + (col >> 3)] : 0u;
    t.w1 = rg + 4 < batch ? gatebits[((rg >> 2) + 1) * (W >> 3) + (col >> 3)] : 0u;
#pragma unroll
    for (int h = 0; h < 2; ++h) t.xa[h].u = *reinterpret_cast<const uint4*>(xT + (int64_t)(16 * h + n) * xt_ld + rg);
    return t;
  };
  int64_t base = r0 + 32 * wave;
  if (base < r1) {
    Tile cur = load(base);
    for (; base < r1; base += 32 * NW) {
      const int64_t nb = base + 32 * NW;
      Tile nxt = load(nb < r1 ? nb : base);
      // gate: row j's byte of the word -> 0 / 1 per 16-bit half of each of its four registers
      uint32_t r[8][4];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const uint32_t byte = ((j < 4 ? cur.w0 : cur.w1) >> (8 * (j & 3))) & 0xFFu;
        const uint32_t sp = byte | (byte << 15);                  // bit c at c (even c) and bit c at c + 15 (odd c)
        const uint32_t gw[4] = {cur.g[j].x, cur.g[j].y, cur.g[j].z, cur.g[j].w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          typedef uint16_t u16x2 __attribute__((ext_vector_type(2)));
          union { uint32_t u; u16x2 v; } a, m, o;
          a.u = gw[d];
          m.u = (sp >> (2 * d)) & 0x00010001u;
          o.v = a.v * m.v;                                        // v_pk_mul_lo_u16
          r[j][d] = o.u;
        }
      }
      // transpose the lane's 8 rows x 8 columns: column c = 2 d + e -> its rows (0,1), (2,3), (4,5), (6,7)
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        const int d = jj >> 1;
        const uint32_t sel = (jj & 1) ? 0x07060302u : 0x05040100u;
        EncFrag b;
        b.u.x = __builtin_amdgcn_perm(r[1][d], r[0][d], sel);
        b.u.y = __builtin_amdgcn_perm(r[3][d], r[2][d], sel);
        b.u.z = __builtin_amdgcn_perm(r[5][d], r[4][d], sel);
        b.u.w = __builtin_amdgcn_perm(r[7][d], r[6][d], sel);
        pacc[jj][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.xa[0].v, b.v, pacc[jj][0], 0, 0, 0);
        pacc[jj][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(cur.xa[1].v, b.v, pacc[jj][1], 0, 0, 0);
        pacc[jj][2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones.v, b.v, pacc[jj][2], 0, 0, 0);
      }
      cur = nxt;
    }
  }
  // S2: feature 32 = accumulator row 0 (q == 0, register 0) of the third group; dY' = 2 (gated dA)
#pragma unroll
  for (int jj = 0; jj < 8; ++jj)
    if (q == 0) sred[wave][8 * n + jj] = 2.f * pacc[jj][2][0];
#pragma unroll
  for (int w = 0; w < NW; ++w) {
    if (wave == w) {
#pragma unroll
      for (int jj = 0; jj < 8; ++jj)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          float4* dst = reinterpret_cast<float4*>(&red[(8 * n + jj) * ENC_IF + 16 * h + 4 * q]);
          float4 v = make_float4(2.f * pacc[jj][h][0], 2.f * pacc[jj][h][1], 2.f * pacc[jj][h][2], 2.f * pacc[jj][h][3]);
          if (w > 0) { const float4 o = *dst; v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
          *dst = v;
        }
    }
    __syncthreads();
  }
  const int t = threadIdx.x;
  if (t < BC) {
    float v = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) v += sred[w][t];
    s2part[rb * W + c0 + t] = v;
  }
  float* pp = ppart + (rb * W + c0) * ENC_IF;
  for (int e = t; e < BC * ENC_IF / 4; e += 64 * NW)
    *reinterpret_cast<float4*>(pp + 4 * e) = *reinterpret_cast<const float4*>(&red[4 * e]);
}

// ---- backward finish: dgamma, dbeta, dW0, db0 (+ the sums of squares of what it writes) ---------------------------
// block = 16 columns x 32 features x 2 halves of the row-block partials (1024 threads: each thread requests its share
// of the partials in one round where 512 threads went round and round; the halves meet in LDS in a fixed order).  It
// runs beside the side stream's weight-gradient GEMM, where every vector instruction waits for an issue slot the GEMM
// leaves: everything that does not depend on the gradient — z^T X, the sum of z — was left by the forward
// (enc_bn_finalize); the first form computed it here and took 50 us.
static constexpr int ENC_FIN_MAXRB = 64;      // row blocks the finish kernel sums without a second round per thread
template <typename TW>
__global__ __launch_bounds__(1024) void enc_bwd_finish_kernel(
    const float* __restrict__ ppart, const float* __restrict__ s2part, int nrb, const float* __restrict__ xs,
    const float* __restrict__ ttab, const float* __restrict__ zs, const TW* __restrict__ W0,
    const float* __restrict__ b0, const float* __restrict__ saved, int64_t batch, int W, float* __restrict__ dW0,
    float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ db_rows, int db_nrows,
    double* __restrict__ sq_w, double* __restrict__ sq_gb) {
  __builtin_amdgcn_s_setprio(3);
  __shared__ double sqred[2][8];
  __shared__ double hP[512], hS[512];
  const int half = threadIdx.x >> 9, t = threadIdx.x & 511;
  const int col = blockIdx.x * 16 + (t >> 5), f = t & 31;
  const bool ok = col < W;        // (whole 32-lane groups: the shuffles below stay inside a column)
  const int cc = ok ? col : 0;
  // what does not depend on the partials is requested first
  const float wf = enc_ld1(W0 + (int64_t)cc * ENC_IF + f), tf = ttab[(int64_t)cc * ENC_IF + f], xf = xs[f];
  const float bcol = b0[cc], zsc = zs[cc];
  const float mean_f = saved[cc], invstd_f = saved[W + cc], scale_f = saved[2 * W + cc];
  double P = 0.0, S2 = 0.0;
  {
    const int per = (nrb + 1) >> 1, lo = half * per, hi = min(nrb, lo + per);
    for (int rb0 = lo; rb0 < hi; rb0 += 16) {               // 2 x 16 loads in flight per round trip
      float v[16], u2[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int64_t rb = min(rb0 + u, nrb - 1);
        v[u] = ppart[(rb * W + cc) * ENC_IF + f];
        u2[u] = s2part[rb * W + cc];
      }
#pragma unroll
      for (int u = 0; u < 16; ++u)
        if (rb0 + u < hi) { P += (double)v[u]; S2 += (double)u2[u]; }
    }
  }
  if (half == 1) { hP[t] = P; hS[t] = S2; }
  __syncthreads();
  double q2w = 0.0, q2g = 0.0;
  if (half == 0 && ok) {
    P += hP[t]; S2 += hS[t];
    const double B = (double)batch, b = (double)bcol;
    double s1 = (double)wf * P;                           // S1 = sum dY' z = W0[col] . P[col] + b0 S2
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) s1 += __shfl_xor(s1, o);
    s1 += b * S2;
    const double mean = (double)mean_f, invstd = (double)invstd_f, scale = (double)scale_f;
    const double dg = invstd * (s1 - mean * S2), db = S2;
    const double c1 = db / B, c2 = dg / B;
    const double a = -scale * c2 * invstd, bp = scale * (c2 * invstd * mean - c1);
    const float g = (float)(scale * P + a * (double)tf + bp * (double)xf);
    dW0[(int64_t)col * ENC_IF + f] = g;
    q2w = (double)g * (double)g;
    if (f == 0) {
      const float dgf = (float)dg, dbf = (float)db;
      dgamma[col] = dgf; dbeta[col] = dbf;
      q2g = (double)dgf * (double)dgf + (double)dbf * (double)dbf;
      db_rows[col] = (float)(scale * S2 + a * (double)zsc + bp * B);
    }
  }
  // (db_rows: the stage's slot of the bias column-sum partials, [db_nrows][W]: row 0 carries db0, the others zero)
  for (int r = 1 + (threadIdx.x >> 4); r < db_nrows; r += 64) {
    const int c16 = blockIdx.x * 16 + (threadIdx.x & 15);
    if (c16 < W) db_rows[(int64_t)r * W + c16] = 0.f;
  }
  if (sq_w) {
    if (half == 0) {
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) { q2w += __shfl_xor(q2w, o); q2g += __shfl_xor(q2g, o); }
      if ((t & 63) == 0) { sqred[0][t >> 6] = q2w; sqred[1][t >> 6] = q2g; }
    }
    __syncthreads();
    if (threadIdx.x < 2) {
      double tt = 0.0;
#pragma unroll
      for (int k = 0; k < 8; ++k) tt += sqred[threadIdx.x][k];
      (threadIdx.x == 0 ? sq_w : sq_gb)[blockIdx.x] = tt;
    }
  }
}

// ---- host side ------------------------------------------------------------------------------------------------
// Rows per block of enc_bwd: whole Philox patches (multiples of 128), as many as leave `min_wgs` workgroups in the
// launch (the partials cost 2 x 33 floats per column and row block), at most ENC_FIN_MAXRB row blocks.
struct EncBwdPlan { int rows, nrb, waves; };
static EncBwdPlan enc_bwd_plan(int64_t batch, int W, bool h) {
  const int ncb = W / (h ? 128 : 64);
  // fp32 storage: the kernel runs beside the side stream's weight-gradient GEMM, which owns the CUs anyway — its
  // partials, not its occupancy, are what the launch is planned for; bf16 storage: alone on the chip, a full launch
  const int min_wgs = h ? 256 : 128;
  int rows = 128;
  while (rows < 1024 && (int64_t)ncb * ceil_div(batch, 2 * rows) >= min_wgs) rows *= 2;
  while (ceil_div(batch, rows) > ENC_FIN_MAXRB) rows *= 2;
  EncBwdPlan p;
  p.rows = rows;
  p.nrb = (int)ceil_div(batch, rows);
  const int tile = h ? 32 : 16;
  p.waves = rows / tile >= 8 ? 8 : 4;          // (each wave at least one tile)
  return p;
}
int enc_bwd_row_blocks(int64_t batch, int W, bool h) { return enc_bwd_plan(batch, W, h).nrb; }
int enc_bwd_finish_blocks(int W) { return (int)ceil_div(W, 16); }      // = bn_bwd_finalize_blocks(W)

// scratch inside the (unused) Z0 buffer of the stage, floats: [xs 32 | zs W | T W x 32 | xpart | ppart | s2part | xT]
struct EncScratch { float *xs, *zs, *ttab, *xpart, *ppart, *s2part; enc_bf16* xT; int64_t xt_ld; int64_t floats; };
static EncScratch enc_scratch(float* z0, int64_t batch, int W, bool h) {
  EncScratch e;
  int64_t off = 0;
  auto take = [&](int64_t n) { float* p = z0 ? z0 + off : nullptr; off += (n + 63) / 64 * 64; return p; };
  const int nrb = enc_bwd_row_blocks(batch, W, h);
  e.xs = take(ENC_IF);
  e.zs = take(W);
  e.ttab = take((int64_t)W * ENC_IF);
  e.xpart = take((int64_t)ENC_XBLOCKS * ENC_XN);
  e.ppart = take((int64_t)nrb * W * ENC_IF);
  e.s2part = take((int64_t)nrb * W);
  e.xt_ld = round_up(batch, 32);
  e.xT = h ? reinterpret_cast<enc_bf16*>(take(ENC_IF * e.xt_ld / 2)) : nullptr;     // (bf16 storage: x^T for the backward)
  e.floats = off;
  return e;
}
bool enc_fused_supported(int64_t batch, int W, int in_features) {
  return in_features == ENC_IF && W % 256 == 0 && batch >= 64 &&
         enc_scratch(nullptr, batch, W, false).floats <= batch * (int64_t)W;
}

static int enc_forward_f(hipStream_t s, const float* x, const float* W0, const float* b0, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, int64_t* nbt, float momentum,
                         float* saved, float* scratch, float* A, uint32_t* keepbits, int64_t batch, int W,
                         const DropoutSrc& drop) {
  const EncScratch e = enc_scratch(scratch, batch, W, false);
  const int xrows = (int)(ENC_XROWS * ceil_div(batch, (int64_t)ENC_XROWS * ENC_XBLOCKS));
  const int xblocks = (int)ceil_div(batch, xrows);
  hipLaunchKernelGGL(enc_xstats_kernel<float>, dim3(xblocks), dim3(64 * ENC_XWAVES), 0, s, x, batch, xrows, e.xpart,
                     (enc_bf16*)nullptr);
  hipLaunchKernelGGL(enc_bn_finalize_kernel<float>, dim3((unsigned)ceil_div(W, 16)), dim3(256), 0, s, x, e.xpart, xblocks,
                     e.xs, e.ttab, e.zs, W0, b0, batch, W, gamma, beta, running_mean, running_var, nbt, momentum, saved,
                     saved + W, saved + 2 * W, saved + 3 * W);
  const int ncb = W / 256;
  hipLaunchKernelGGL(enc_fwd_kernel, dim3(enc_grid(ncb, ceil_div(batch, 32))), dim3(256), 0, s, x, W0, b0, saved + 2 * W,
                     saved + 3 * W, A, keepbits, batch, W, ncb, drop, nbt);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

static int enc_forward_h(hipStream_t s, enc_bf16* x, const float* x_f32, const enc_bf16* W0, const float* b0,
                         const float* gamma, const float* beta, float* running_mean, float* running_var, int64_t* nbt,
                         float momentum, float* saved, float* scratch, enc_bf16* A, uint32_t* keepbits, int64_t batch,
                         int W, const DropoutSrc& drop) {
  const EncScratch e = enc_scratch(scratch, batch, W, true);
  const int xrows = (int)(ENC_XROWS * ceil_div(batch, (int64_t)ENC_XROWS * ENC_XBLOCKS));
  const int xblocks = (int)ceil_div(batch, xrows);
  // x_f32: the bf16 image `x` has not been written yet — the statistics kernel casts on the way
  if (x_f32)
    hipLaunchKernelGGL((enc_xstats_kernel<float, true>), dim3(xblocks), dim3(64 * ENC_XWAVES), 0, s, x_f32, batch, xrows,
                       e.xpart, x);
  else
    hipLaunchKernelGGL(enc_xstats_kernel<enc_bf16>, dim3(xblocks), dim3(64 * ENC_XWAVES), 0, s, (const enc_bf16*)x, batch,
                       xrows, e.xpart, (enc_bf16*)nullptr);
  hipLaunchKernelGGL(enc_bn_finalize_kernel<enc_bf16>, dim3((unsigned)ceil_div(W, 16)), dim3(256), 0, s, x, e.xpart, xblocks,
                     e.xs, e.ttab, e.zs, W0, b0, batch, W, gamma, beta, running_mean, running_var, nbt, momentum, saved,
                     saved + W, saved + 2 * W, saved + 3 * W);
  const int ncb = W / 512;
  hipLaunchKernelGGL(enc_fwd_h_kernel, dim3(enc_grid(ncb, ceil_div(batch, 32))), dim3(256), 0, s, x, W0, b0, saved + 2 * W,
                     saved + 3 * W, A, keepbits, e.xT, e.xt_ld, batch, W, ncb, drop, nbt);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

template <bool H>
static int enc_backward_t(hipStream_t s, const typename EncT<H>::T* dA, const typename EncT<H>::T* x,
                          const typename EncT<H>::T* W0, const float* b0, const float* saved, const uint32_t* gatebits,
                          float* scratch, int64_t batch, int W, float* dW0, float* dgamma, float* dbeta,
                          float* db_rows, int db_nrows, double* sq_w, double* sq_gb) {
  typedef typename EncT<H>::T T;
  const EncScratch e = enc_scratch(scratch, batch, W, H);
  const EncBwdPlan pl = enc_bwd_plan(batch, W, H);
  const int ncb = W / (H ? 128 : 64);
  const dim3 grid(enc_grid(ncb, pl.nrb));
  if constexpr (H) {
    (void)x;
    if (pl.waves == 8)
      hipLaunchKernelGGL(enc_bwd_h_kernel<8>, grid, dim3(512), 0, s, dA, e.xT, e.xt_ld, gatebits, e.s2part, e.ppart, batch, W,
                         pl.rows, ncb, pl.nrb);
    else
      hipLaunchKernelGGL(enc_bwd_h_kernel<4>, grid, dim3(256), 0, s, dA, e.xT, e.xt_ld, gatebits, e.s2part, e.ppart, batch, W,
                         pl.rows, ncb, pl.nrb);
  } else {
    if (pl.waves == 8)
      hipLaunchKernelGGL(enc_bwd_kernel<8>, grid, dim3(512), 0, s, dA, x, gatebits, e.s2part, e.ppart, batch, W, pl.rows, ncb,
                         pl.nrb);
    else
      hipLaunchKernelGGL(enc_bwd_kernel<4>, grid, dim3(256), 0, s, dA, x, gatebits, e.s2part, e.ppart, batch, W, pl.rows, ncb,
                         pl.nrb);
  }
  hipLaunchKernelGGL(enc_bwd_finish_kernel<T>, dim3((unsigned)enc_bwd_finish_blocks(W)), dim3(1024), 0, s, e.ppart,
                     e.s2part, pl.nrb, e.xs, e.ttab, e.zs, W0, b0, saved, batch, W, dW0, dgamma, dbeta,
                     db_rows, db_nrows, sq_w, sq_gb);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

int launch_enc_forward(hipStream_t s, const float* x, const float* W0, const float* b0, const float* gamma,
                       const float* beta, float* running_mean, float* running_var, int64_t* nbt, float momentum,
                       float* saved, float* z0_scratch, float* A, uint32_t* keepbits, int64_t batch, int W,
                       const DropoutSrc& drop) {
  if (!enc_fused_supported(batch, W, ENC_IF)) return BLH_ERR_SHAPE;
  return enc_forward_f(s, x, W0, b0, gamma, beta, running_mean, running_var, nbt, momentum, saved, z0_scratch, A,
                       keepbits, batch, W, drop);
}
int launch_enc_backward(hipStream_t s, const float* dA, const float* x, const float* W0, const float* b0,
                        const float* saved, const uint32_t* gatebits, float* z0_scratch, int64_t batch, int W,
                        float* dW0, float* dgamma, float* dbeta, float* db_rows, int db_nrows, double* sq_w,
                        double* sq_gb) {
  return enc_backward_t<false>(s, dA, x, W0, b0, saved, gatebits, z0_scratch, batch, W, dW0, dgamma, dbeta, db_rows,
                               db_nrows, sq_w, sq_gb);
}

// bf16 storage: the scratch is the stage's (unused) bf16 Z0 buffer, batch * W / 2 floats
bool enc_fused_supported_h(int64_t batch, int W, int in_features) {
  return in_features == ENC_IF && W % 512 == 0 && batch >= 64 &&
         enc_scratch(nullptr, batch, W, true).floats <= batch * (int64_t)W / 2;
}
int launch_enc_forward_h(hipStream_t s, uint16_t* xh, const float* x_f32, const uint16_t* W0h, const float* b0,
                         const float* gamma, const float* beta, float* running_mean, float* running_var, int64_t* nbt,
                         float momentum, float* saved, uint16_t* z0_scratch, uint16_t* A, uint32_t* keepbits,
                         int64_t batch, int W, const DropoutSrc& drop) {
  if (!enc_fused_supported_h(batch, W, ENC_IF)) return BLH_ERR_SHAPE;
  return enc_forward_h(s, xh, x_f32, W0h, b0, gamma, beta, running_mean, running_var, nbt, momentum, saved,
                       reinterpret_cast<float*>(z0_scratch), A, keepbits, batch, W, drop);
}
int launch_enc_backward_h(hipStream_t s, const uint16_t* dA, const uint16_t* xh, const uint16_t* W0h, const float* b0,
                          const float* saved, const uint32_t* gatebits, uint16_t* z0_scratch, int64_t batch, int W,
                          float* dW0, float* dgamma, float* dbeta, float* db_rows, int db_nrows) {
  return enc_backward_t<true>(s, dA, xh, W0h, b0, saved, gatebits, reinterpret_cast<float*>(z0_scratch), batch, W, dW0,
                              dgamma, dbeta, db_rows, db_nrows, nullptr, nullptr);
}

}  // namespace blh
