// Purpose-built kernels for the skinny projection at the end of the lifter
// (/root/reference/model/bilinear.py:29,39: nn.Linear(1024, 48); train_bilinear.py:78: MSELoss).
//
// decode forward  pred[B,48] = A[B,W] Wd^T + bd  fused with the loss: dpred = 2 (pred - t)/(B 48),
// per-block partial sums of the squared error and of dpred (the decode-bias gradient).
// HBM-bound: it reads A once (W * 4 bytes per pose) and writes 2 * 48 * 4 bytes; the generic
// MFMA GEMM needed a split reduction (8 slabs of [B,48]) plus a finishing kernel for it, because
// a 128-row tile with N = 48 leaves only B/128 workgroups.
//
// Structure: workgroup = 8 waves over 16*RT rows; the eight waves split the reduction index
// (K/8 each), so that 2048 waves — two per SIMD at B = 4096 — stream disjoint 16-byte pieces of A
// straight into registers (no LDS staging: nothing is shared between waves) and feed
// v_mfma_f32_16x16x4_f32 (exact fp32; 48 = 3 column tiles of 16, no padding waste).  A lane's
// float4 (4 consecutive k of its row) serves 4 MFMAs: MFMA j contracts k = k0 + 4 q + j over the
// four lane quarters q, the same map on both operands.  Wd (196 KB) is re-read by every workgroup
// from L2.  The eight partial accumulators meet in LDS, then 256 threads finish 16*RT x 48 outputs.
#include "common.h"

namespace blh {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

static constexpr int DEC_WAVES = 8;     // the reduction index is split over this many waves

template <int RT>
__global__ __launch_bounds__(64 * DEC_WAVES) void decode_fwd_mse_kernel(
    const float* __restrict__ A, const float* __restrict__ Wd, const float* __restrict__ bd,
    const float* __restrict__ target, float* __restrict__ pred, float* __restrict__ dpred,
    float* __restrict__ loss_part, float* __restrict__ dbias_part, int64_t batch, int W, int OF,
    float scale) {
  constexpr int ROWS = 16 * RT, NT = 4;             // up to 4 column tiles (OF <= 64)
  __shared__ __attribute__((aligned(16))) float red[DEC_WAVES][ROWS][64];
  __shared__ float colred[16][64];
  __shared__ float lossred[DEC_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int ntiles = (OF + 15) >> 4;
  const int kq = W / DEC_WAVES;                     // reduction range of one wave (multiple of 32)
  const int kbeg = wave * kq;

  const float* arow[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t)
    arow[t] = A + min(row0 + t * 16 + r, batch - 1) * (int64_t)W + kbeg + 4 * q;   // clamped rows are never stored
  const float* wrow[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wrow[t] = Wd + (int64_t)min(t * 16 + r, OF - 1) * W + kbeg + 4 * q;

  f32x4_t acc[RT][NT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // 32 k per chunk (two k-steps of 16), register double buffer: the loads of chunk c+1 are in
  // flight while the MFMAs of chunk c issue (one wave per SIMD: nothing else hides the latency)
  float4 a[2][2][RT], b[2][2][NT];
  auto load_chunk = [&](int buf, int k) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int i = 0; i < RT; ++i) a[buf][u][i] = *reinterpret_cast<const float4*>(arow[i] + k + 16 * u);
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (t < ntiles) b[buf][u][t] = *reinterpret_cast<const float4*>(wrow[t] + k + 16 * u);
    }
  };
  auto mfma_chunk = [&](int buf) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int i = 0; i < RT; ++i)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          if (t < ntiles) {
            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[buf][u][i].x, b[buf][u][t].x, acc[i][t], 0, 0, 0);
            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[buf][u][i].y, b[buf][u][t].y, acc[i][t], 0, 0, 0);
            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[buf][u][i].z, b[buf][u][t].z, acc[i][t], 0, 0, 0);
            acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[buf][u][i].w, b[buf][u][t].w, acc[i][t], 0, 0, 0);
          }
  };
  load_chunk(0, 0);
  for (int k = 0; k < kq; k += 64) {       // kq % 64 == 0 (W % 256 == 0) or the tail below
    if (k + 32 < kq) load_chunk(1, k + 32);
    mfma_chunk(0);
    if (k + 32 < kq) {
      if (k + 64 < kq) load_chunk(0, k + 64);
      mfma_chunk(1);
    }
  }
  // C layout of the 16x16 MFMA: column = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (t < ntiles)
#pragma unroll
        for (int g = 0; g < 4; ++g) red[wave][i * 16 + 4 * q + g][t * 16 + r] = acc[i][t][g];
  __syncthreads();

  // finish: thread = (row, 4 columns); 16 column groups x 16 rows per pass (the first 256 threads)
  const int cg = tid & 15, rr = (tid >> 4) & 15;
  const int col = cg * 4;
  float sq = 0.f;
  float4 dsum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < OF && tid < 256) {
    const float4 bv = *reinterpret_cast<const float4*>(bd + col);
#pragma unroll
    for (int pass = 0; pass < RT; ++pass) {
      const int lr = pass * 16 + rr;
      const int64_t row = row0 + lr;
      float4 v = bv;
#pragma unroll
      for (int w = 0; w < DEC_WAVES; ++w) {
        const float4 u = *reinterpret_cast<const float4*>(&red[w][lr][col]);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      if (row < batch) {
        *reinterpret_cast<float4*>(pred + row * OF + col) = v;
        if (target) {
          const float4 t = *reinterpret_cast<const float4*>(target + row * OF + col);
          float4 d = make_float4(v.x - t.x, v.y - t.y, v.z - t.z, v.w - t.w);
          sq += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
          d.x *= scale; d.y *= scale; d.z *= scale; d.w *= scale;
          dsum.x += d.x; dsum.y += d.y; dsum.z += d.z; dsum.w += d.w;
          *reinterpret_cast<float4*>(dpred + row * OF + col) = d;
        }
      }
    }
  }
  if (target) {
    // loss partial of the block (fixed order: deterministic)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sq += __shfl_xor(sq, o);
    if (lane == 0) lossred[wave] = sq;
    if (tid < 256) *reinterpret_cast<float4*>(&colred[rr][col]) = dsum;
    __syncthreads();
    if (tid == 0) loss_part[blockIdx.x] = (lossred[0] + lossred[1]) + (lossred[2] + lossred[3]);   // waves 4.. hold no rows
    if (dbias_part && tid < OF) {
      float s = 0.f;
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) s += colred[k2][tid];
      dbias_part[(int64_t)blockIdx.x * OF + tid] = s;
    }
  }
}

// ---- bf16 storage (gemm_dtype 4): the same kernel for bf16 A and the bf16 image of Wd ----------
// A lane's 16 bytes (8 consecutive k of its row) are exactly one operand of
// v_mfma_f32_16x16x32_bf16 (lane l: row l & 15, k = 8 (l >> 4) + j), for A and for Wd alike: the
// reduction range of a wave (W / 8) is walked in k-steps of 32 straight from global memory.
// Besides pred / dpred (fp32) it writes dpred as bf16, the operand of the decode backward GEMMs.
typedef __bf16 dec_bf16x8 __attribute__((ext_vector_type(8)));

template <int RT>
__global__ __launch_bounds__(64 * DEC_WAVES) void decode_fwd_mse_h_kernel(
    const uint16_t* __restrict__ A, const uint16_t* __restrict__ Wd, const float* __restrict__ bd,
    const float* __restrict__ target, float* __restrict__ pred, float* __restrict__ dpred,
    uint16_t* __restrict__ dpred_h, float* __restrict__ loss_part, float* __restrict__ dbias_part,
    int64_t batch, int W, int OF, float scale) {
  constexpr int ROWS = 16 * RT, NT = 4;
  __shared__ __attribute__((aligned(16))) float red[DEC_WAVES][ROWS][64];
  __shared__ float colred[16][64];
  __shared__ float lossred[DEC_WAVES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int r = lane & 15, q = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * ROWS;
  const int ntiles = (OF + 15) >> 4;
  const int kq = W / DEC_WAVES;                     // reduction range of one wave (multiple of 32)
  const int kbeg = wave * kq;

  const uint16_t* arow[RT];
#pragma unroll
  for (int t = 0; t < RT; ++t)
    arow[t] = A + min(row0 + t * 16 + r, batch - 1) * (int64_t)W + kbeg + 8 * q;   // clamped rows are never stored
  const uint16_t* wrow[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) wrow[t] = Wd + (int64_t)min(t * 16 + r, OF - 1) * W + kbeg + 8 * q;

  f32x4_t acc[RT][NT];
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[i][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // two k-steps (64 k) per chunk, register double buffer
  uint4 a[2][2][RT], b[2][2][NT];
  auto load_chunk = [&](int buf, int k) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int kk = min(k + 32 * u, kq - 32);      // (kq % 64 == 32: the second k-step of the last chunk is unused)
#pragma unroll
      for (int i = 0; i < RT; ++i) a[buf][u][i] = *reinterpret_cast<const uint4*>(arow[i] + kk);
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (t < ntiles) b[buf][u][t] = *reinterpret_cast<const uint4*>(wrow[t] + kk);
    }
  };
  auto mfma_chunk = [&](int buf, int k) {
#pragma unroll
    for (int u = 0; u < 2; ++u)
      if (k + 32 * u < kq)
#pragma unroll
        for (int i = 0; i < RT; ++i)
#pragma unroll
          for (int t = 0; t < NT; ++t)
            if (t < ntiles) {
              union { uint4 u4; dec_bf16x8 v; } ua, ub;
              ua.u4 = a[buf][u][i]; ub.u4 = b[buf][u][t];
              acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v, ub.v, acc[i][t], 0, 0, 0);
            }
  };
  load_chunk(0, 0);
  for (int k = 0; k < kq; k += 128) {
    if (k + 64 < kq) load_chunk(1, k + 64);
    mfma_chunk(0, k);
    if (k + 64 < kq) {
      if (k + 128 < kq) load_chunk(0, k + 128);
      mfma_chunk(1, k + 64);
    }
  }
#pragma unroll
  for (int i = 0; i < RT; ++i)
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (t < ntiles)
#pragma unroll
        for (int g = 0; g < 4; ++g) red[wave][i * 16 + 4 * q + g][t * 16 + r] = acc[i][t][g];
  __syncthreads();

  const int cg = tid & 15, rr = (tid >> 4) & 15;
  const int col = cg * 4;
  float sq = 0.f;
  float4 dsum = make_float4(0.f, 0.f, 0.f, 0.f);
  if (col < OF && tid < 256) {
    const float4 bv = *reinterpret_cast<const float4*>(bd + col);
#pragma unroll
    for (int pass = 0; pass < RT; ++pass) {
      const int lr = pass * 16 + rr;
      const int64_t row = row0 + lr;
      float4 v = bv;
#pragma unroll
      for (int w = 0; w < DEC_WAVES; ++w) {
        const float4 u = *reinterpret_cast<const float4*>(&red[w][lr][col]);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      if (row < batch) {
        *reinterpret_cast<float4*>(pred + row * OF + col) = v;
        if (target) {
          const float4 t = *reinterpret_cast<const float4*>(target + row * OF + col);
          float4 d = make_float4(v.x - t.x, v.y - t.y, v.z - t.z, v.w - t.w);
          sq += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
          d.x *= scale; d.y *= scale; d.z *= scale; d.w *= scale;
          dsum.x += d.x; dsum.y += d.y; dsum.z += d.z; dsum.w += d.w;
          *reinterpret_cast<float4*>(dpred + row * OF + col) = d;
          typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
          const bf2 lo = {(__bf16)d.x, (__bf16)d.y}, hi = {(__bf16)d.z, (__bf16)d.w};
          uint2 o;
          o.x = *reinterpret_cast<const uint32_t*>(&lo);
          o.y = *reinterpret_cast<const uint32_t*>(&hi);
          *reinterpret_cast<uint2*>(dpred_h + row * OF + col) = o;
        }
      }
    }
  }
  if (target) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sq += __shfl_xor(sq, o);
    if (lane == 0) lossred[wave] = sq;
    if (tid < 256) *reinterpret_cast<float4*>(&colred[rr][col]) = dsum;
    __syncthreads();
    if (tid == 0) loss_part[blockIdx.x] = (lossred[0] + lossred[1]) + (lossred[2] + lossred[3]);
    if (dbias_part && tid < OF) {
      float s = 0.f;
#pragma unroll
      for (int k2 = 0; k2 < 16; ++k2) s += colred[k2][tid];
      dbias_part[(int64_t)blockIdx.x * OF + tid] = s;
    }
  }
}

// rows per workgroup so that the block count (= number of loss / bias partials) stays <= 1024
int decode_fwd_rows_per_block(int64_t batch) {
  int rt = 1;
  while (rt < 4 && ceil_div(batch, 16 * rt) > 1024) rt *= 2;
  return 16 * rt;
}
// bf16 operands: a block re-reads the whole decode weight (96 KB at W = 1024) for its rows, three
// quarters of the kernel's load instructions at 16 rows per block; as many rows per block as still
// leave one block per CU (measured, B = 16384: 23.3 / 21.7 / 19.0 us at 16 / 32 / 64 rows; B = 8192:
// 13.8 / 12.8 / 16.2 us; the fp32 kernel at B = 4096 is fastest at 16 rows: 11.1 / 15.3 / 22.3 us)
int decode_fwd_rows_per_block_h(int64_t batch) {
  int rt = 1;
  while (rt < 4 && (ceil_div(batch, 16 * rt) > 1024 || ceil_div(batch, 32 * rt) >= 256)) rt *= 2;
  return 16 * rt;
}

bool decode_fwd_supported(int64_t batch, int W, int OF) {
  // (a wave's share of the reduction, W / 8, is walked in chunks of 32)
  return W % (32 * DEC_WAVES) == 0 && OF % 4 == 0 && OF >= 4 && OF <= 64 && ceil_div(batch, 64) <= 1024;
}

int launch_decode_fwd_mse(hipStream_t s, const float* A, const float* Wd, const float* bd,
                          const float* target, float* pred, float* dpred, float* loss_part,
                          float* dbias_part, int64_t batch, int W, int OF, float scale, int* nparts) {
  if (!decode_fwd_supported(batch, W, OF)) return BLH_ERR_SHAPE;
  const int rows = decode_fwd_rows_per_block(batch);
  const int blocks = (int)ceil_div(batch, rows);
  if (rows == 16)
    hipLaunchKernelGGL(decode_fwd_mse_kernel<1>, dim3(blocks), dim3(64 * DEC_WAVES), 0, s, A, Wd, bd, target, pred,
                       dpred, loss_part, dbias_part, batch, W, OF, scale);
  else if (rows == 32)
    hipLaunchKernelGGL(decode_fwd_mse_kernel<2>, dim3(blocks), dim3(64 * DEC_WAVES), 0, s, A, Wd, bd, target, pred,
                       dpred, loss_part, dbias_part, batch, W, OF, scale);
  else
    hipLaunchKernelGGL(decode_fwd_mse_kernel<4>, dim3(blocks), dim3(64 * DEC_WAVES), 0, s, A, Wd, bd, target, pred,
                       dpred, loss_part, dbias_part, batch, W, OF, scale);
  BLH_HIP_TRY(hipGetLastError());
  if (nparts) *nparts = blocks;
  return BLH_OK;
}

int launch_decode_fwd_mse_h(hipStream_t s, const uint16_t* A, const uint16_t* Wd, const float* bd,
                            const float* target, float* pred, float* dpred, uint16_t* dpred_h,
                            float* loss_part, float* dbias_part, int64_t batch, int W, int OF, float scale,
                            int* nparts) {
  if (!decode_fwd_supported(batch, W, OF) || W % 8 != 0 || OF % 4 != 0) return BLH_ERR_SHAPE;
  const int rows = decode_fwd_rows_per_block_h(batch);
  const int blocks = (int)ceil_div(batch, rows);
  if (rows == 16)
    hipLaunchKernelGGL(decode_fwd_mse_h_kernel<1>, dim3(blocks), dim3(64 * DEC_WAVES), 0, s, A, Wd, bd, target,
                       pred, dpred, dpred_h, loss_part, dbias_part, batch, W, OF, scale);
  else if (rows == 32)
    hipLaunchKernelGGL(decode_fwd_mse_h_kernel<2>, dim3(blocks), dim3(64 * DEC_WAVES), 0, s, A, Wd, bd, target,
                       pred, dpred, dpred_h, loss_part, dbias_part, batch, W, OF, scale);
  else
    hipLaunchKernelGGL(decode_fwd_mse_h_kernel<4>, dim3(blocks), dim3(64 * DEC_WAVES), 0, s, A, Wd, bd, target,
                       pred, dpred, dpred_h, loss_part, dbias_part, batch, W, OF, scale);
  BLH_HIP_TRY(hipGetLastError());
  if (nparts) *nparts = blocks;
  return BLH_OK;
}

}  // namespace blh
