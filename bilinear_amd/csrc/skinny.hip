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
#include <atomic>

#include "common.h"
#include "gemm_dma.h"

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

// FUSE (round 5, one-pass decode in bf16 storage): the block also forms dA = dP Wd for its rows — the decode data
// gradient the backward would otherwise compute with a K = 48 GEMM that re-reads nothing of A but costs a launch,
// a fork and 13 us (B = 16384) — from the dP tile it has just produced (bf16, in LDS: the A operand of
// v_mfma_f32_16x16x32_bf16, k = output feature padded to 64) and WdT[W][64], the decode weight's bf16 image with
// the output feature contiguous (written by the forward's cast launch).  A wave takes 128-column groups; its lane
// (n, q) feeds MFMA j with column 8 n + j, so that after 8 MFMAs it holds 8 consecutive columns of 4 rows: one
// 16-byte store per row.
template <int RT, bool FUSE>
__global__ __launch_bounds__(64 * DEC_WAVES) void decode_fwd_mse_h_kernel(
    const uint16_t* __restrict__ A, const uint16_t* __restrict__ Wd, const float* __restrict__ bd,
    const float* __restrict__ target, float* __restrict__ pred, float* __restrict__ dpred,
    uint16_t* __restrict__ dpred_h, float* __restrict__ loss_part, float* __restrict__ dbias_part,
    int64_t batch, int W, int OF, float scale, const uint16_t* __restrict__ WdT, uint16_t* __restrict__ dA) {
  constexpr int ROWS = 16 * RT, NT = 4;
  __shared__ __attribute__((aligned(16))) float red[DEC_WAVES][ROWS][64];
  __shared__ float colred[16][64];
  __shared__ float lossred[DEC_WAVES];
  __shared__ __attribute__((aligned(16))) uint16_t dps[FUSE ? ROWS : 1][FUSE ? 72 : 8];   // dP tile, bf16, rows 144 B apart
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
          if (FUSE) *reinterpret_cast<uint2*>(&dps[lr][col]) = o;
        }
      } else if (FUSE) {
        *reinterpret_cast<uint2*>(&dps[lr][col]) = make_uint2(0u, 0u);       // rows beyond the batch: zero gradient
      }
    }
  }
  if (FUSE && tid < 256 && col >= OF) {      // output features 48 .. 63 of the padded contraction: zero
#pragma unroll
    for (int pass = 0; pass < RT; ++pass) *reinterpret_cast<uint2*>(&dps[pass * 16 + rr][col]) = make_uint2(0u, 0u);
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
  if (FUSE) {
    // phase 2 (behind the barrier above: the dP tile is complete).  WdT image: for column group g, MFMA j, k-step ks
    // one 1 KiB block of 64 x 16 bytes, lane (q, n) -> Wd[32 ks + 8 q .. + 7][128 g + 8 n + j] (wdT_image_kernel)
    const int ngroups = W >> 7;
    for (int g = wave; g < ngroups; g += DEC_WAVES) {
      uint4 bw[8][2];
#pragma unroll
      for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
          bw[j][ks] = *reinterpret_cast<const uint4*>(WdT + ((((int64_t)g * 8 + j) * 2 + ks) * 64 + lane) * 8);
#pragma unroll
      for (int i = 0; i < RT; ++i) {
        union { uint4 u4; dec_bf16x8 v; } a0, a1;
        a0.u4 = *reinterpret_cast<const uint4*>(&dps[i * 16 + r][8 * q]);
        a1.u4 = *reinterpret_cast<const uint4*>(&dps[i * 16 + r][32 + 8 * q]);
        f32x4_t o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          union { uint4 u4; dec_bf16x8 v; } b0, b1;
          b0.u4 = bw[j][0]; b1.u4 = bw[j][1];
          o[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0.v, b0.v, f32x4_t{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
          o[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1.v, b1.v, o[j], 0, 0, 0);
        }
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          const int64_t row = row0 + i * 16 + 4 * q + gq;
          typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
          const bf2 p0 = {(__bf16)o[0][gq], (__bf16)o[1][gq]}, p1 = {(__bf16)o[2][gq], (__bf16)o[3][gq]},
                    p2 = {(__bf16)o[4][gq], (__bf16)o[5][gq]}, p3 = {(__bf16)o[6][gq], (__bf16)o[7][gq]};
          uint4 st;
          st.x = *reinterpret_cast<const uint32_t*>(&p0); st.y = *reinterpret_cast<const uint32_t*>(&p1);
          st.z = *reinterpret_cast<const uint32_t*>(&p2); st.w = *reinterpret_cast<const uint32_t*>(&p3);
          if (row < batch) *reinterpret_cast<uint4*>(dA + row * (int64_t)W + 128 * g + 8 * r) = st;
        }
      }
    }
  }
}

// ---- one-pass decode: forward + MSE + data gradient from ONE read of the last activation ----------------------
// (/root/reference/model/bilinear.py:29,39; train_bilinear.py:78-79.)  The loss is row-local, so a block that
// holds 16 rows of A can finish their predictions, form dP = 2 (P - T) / (B OF) and write dA = dP Wd for the same
// rows: the data gradient of the decode Linear leaves the forward kernel, and backward starts at the first
// BatchNorm-backward reduction (the decode WEIGHT gradient dWd = dP^T A stays a GEMM over batch slabs on the side
// stream: per-block partials of a [48][W] tensor would be 50 MB at 256 blocks).
//
// Block = 16 rows x 8 waves, wave w owns the SL = W / 8 columns [w SL, (w + 1) SL) of A, of Wd and of dA.
//   phase 1  pred partial of the wave = A[16][SL] Wd[OF][SL]^T on v_mfma_f32_16x16x4_f32 (exact fp32).  A comes
//            global -> LDS by LDS-DMA in pieces of 8 rows x 128 bytes (whole cache lines; the fragment-shaped
//            16 rows x 64 bytes loads of decode_fwd_mse_kernel use half of every line per instruction), 16-byte
//            chunks XOR-swizzled on the SOURCE address with (row >> 1) & 7 so that the ds_read_b128 of an MFMA
//            operand (16 rows, one chunk) is conflict-free; the Wd fragments (3 x SL / 16 float4 per lane, from
//            L2) are requested BEFORE the DMAs: the wave's memory queue retires in order, so the counted
//            `s_waitcnt vmcnt(2 (KR - 1 - kr))` in front of k-range kr covers them too and MFMAs start when the
//            first two pieces have landed.
//   finish   the eight partials meet in LDS (each wave reuses its own stage), 192 threads add the bias, write
//            pred / dpred, the loss and decode-bias partials of the block, and leave dP in LDS.
//   phase 2  dA[16][SL] = dP[16][OF] Wd[OF][SL]: A operand = dP from LDS (lane: row l & 15, o = 4 step + (l >> 4)),
//            B operand = a float4 of Wd row o (four consecutive columns of ONE o: MFMA j takes component j, so the
//            four MFMAs of a step produce columns 4 n + j — a column permutation that makes the accumulators of a
//            lane four consecutive columns of its rows: 16-byte stores, 256 contiguous bytes per row and wave).
//            The Wd rows of phase 2 are requested before the finish barrier.
// Algorithmic bytes per pose: W s (A) + W s (dA) + 3 OF s (target, pred, dpred).
// ABL (tools/decode_bench only; results are then wrong): 1 no phase 2, 2 no phase-1 MFMAs, 3 neither, 4 empty kernel
template <int SL, int ABL = 0>
__global__ __launch_bounds__(512, 2) void decode_fused_kernel(
    const float* __restrict__ A, const float* __restrict__ Wd, const float* __restrict__ bd,
    const float* __restrict__ target, float* __restrict__ pred, float* __restrict__ dpred,
    float* __restrict__ dA, float* __restrict__ loss_part, float* __restrict__ dbias_part, int64_t batch, float scale) {
  constexpr int OF = 48, NT = 3, W = 8 * SL, KR = SL / 32, NU = SL / 16, NG = SL / 64, DPP = 52;
  // one dynamic LDS object (more than 64 KiB): per wave its A slice (then its partial) | dP | column partials | loss
  extern __shared__ __attribute__((aligned(16))) float dec_lds[];
  float* stage0 = dec_lds;                                   // [8][16 * SL]
  float (*dps)[DPP] = reinterpret_cast<float (*)[DPP]>(dec_lds + 8 * 16 * SL);
  float (*colred)[OF] = reinterpret_cast<float (*)[OF]>(dec_lds + 8 * 16 * SL + 16 * DPP);
  float* lossred = dec_lds + 8 * 16 * SL + 16 * DPP + 16 * OF;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 15, q = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * 16;
  const int kbeg = wave * SL;
  if (ABL == 4) return;

  // what the finishing threads need (thread = row tid >> 4, 4 columns): requested first, long landed when used
  const int fcol = (tid & 15) * 4, frr = (tid >> 4) & 15;
  const bool fin = tid < 256 && fcol < OF;
  float4 tgt = make_float4(0.f, 0.f, 0.f, 0.f), bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (fin) {
    bias4 = *reinterpret_cast<const float4*>(bd + fcol);
    tgt = *reinterpret_cast<const float4*>(target + min(row0 + frr, batch - 1) * OF + fcol);
  }
  // Operands of phase 1, k-range by k-range (32 columns): the Wd fragments of the range — lane (o = 16 t + r,
  // k = kbeg + 16 u + 4 q .. + 3), 2 x NT float4 — then the two A pieces of the range by LDS-DMA: piece p =
  // (k-range p >> 1, rows 8 (p & 1) .. + 7), lane = (row l >> 3, slot l & 7).  8 operations per range and wave.
  float4 bw[NU][NT];
  const uint32_t stage_lds = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) float*)dec_lds) +
                             (uint32_t)wave * (uint32_t)(16 * SL * 4);
#pragma unroll
  for (int kr = 0; kr < KR; ++kr) {
#pragma unroll
    for (int uu = 0; uu < 2; ++uu)
#pragma unroll
      for (int t = 0; t < NT; ++t)
        bw[2 * kr + uu][t] =
            *reinterpret_cast<const float4*>(Wd + (int64_t)(t * 16 + r) * W + kbeg + 16 * (2 * kr + uu) + 4 * q);
    asm volatile("" ::: "memory");
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int p = 2 * kr + h;
      const int row = 8 * h + (lane >> 3);
      const int chunk = kr * 8 + ((lane & 7) ^ ((row >> 1) & 7));
      const float* src = A + min(row0 + row, batch - 1) * (int64_t)W + kbeg + chunk * 4;
      lds_dma16_asm(src, __builtin_amdgcn_readfirstlane(stage_lds + (uint32_t)p * 1024u));
    }
  }
  f32x4_t acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  const float* st = stage0 + (size_t)wave * (16 * SL);
#pragma unroll
  for (int kr = 0; kr < KR; ++kr) {
    // the 8 operations of range kr landed — and every older one (the queue retires in order)
    if (kr == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (KR - 1)) : "memory");
    else if (kr == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KR > 1 ? 8 * (KR - 2) : 0) : "memory");
    else if (kr == 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(KR > 2 ? 8 * (KR - 3) : 0) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int uu = 0; uu < 2; ++uu) {
      const int u = 2 * kr + uu, c = 4 * u + q;              // chunk of this lane's 4 k
      const float4 a = *reinterpret_cast<const float4*>(
          st + (((c >> 3) * 2 + (r >> 3)) * 256 + (r & 7) * 32 + (((c & 7) ^ ((r >> 1) & 7)) << 2)));
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        if (ABL == 2 || ABL == 3) { acc[t][0] += a.x * bw[u][t].x + a.y * bw[u][t].y + a.z * bw[u][t].z + a.w * bw[u][t].w; continue; }
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, bw[u][t].x, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, bw[u][t].y, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, bw[u][t].z, acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, bw[u][t].w, acc[t], 0, 0, 0);
      }
    }
  }
  // Wd rows of phase 2 (group gc = 64 columns): lane (o = 4 step + q, columns kbeg + 64 gc + 4 r .. + 3)
  float4 bw2[NG][OF / 4];
#pragma unroll
  for (int gc = 0; gc < NG; ++gc)
#pragma unroll
    for (int sp = 0; sp < OF / 4; ++sp)
      bw2[gc][sp] = *reinterpret_cast<const float4*>(Wd + (int64_t)(4 * sp + q) * W + kbeg + 64 * gc + 4 * r);
  // the wave's partial into its own (now dead) stage: [16 rows][64]; C layout: column l & 15, row 4 (l >> 4) + reg
  float* part = stage0 + (size_t)wave * (16 * SL);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int g = 0; g < 4; ++g) part[(4 * q + g) * 64 + t * 16 + r] = acc[t][g];
  __syncthreads();

  // finish: thread = (row tid >> 4, 4 columns 4 (tid & 15)); 12 column groups x 16 rows = 192 of the first 256 threads
  if (tid < 256) {
    const int col = fcol, rr = frr;
    const int64_t row = row0 + rr;
    float sq = 0.f;
    float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
    if (fin) {
      float4 v = bias4;
#pragma unroll
      for (int w = 0; w < 8; ++w) {
        const float4 u = *reinterpret_cast<const float4*>(stage0 + (size_t)w * (16 * SL) + rr * 64 + col);
        v.x += u.x; v.y += u.y; v.z += u.z; v.w += u.w;
      }
      if (row < batch) {
        *reinterpret_cast<float4*>(pred + row * OF + col) = v;
        d = make_float4(v.x - tgt.x, v.y - tgt.y, v.z - tgt.z, v.w - tgt.w);
        sq = d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w;
        d.x *= scale; d.y *= scale; d.z *= scale; d.w *= scale;
        *reinterpret_cast<float4*>(dpred + row * OF + col) = d;
      }
      *reinterpret_cast<float4*>(&dps[rr][col]) = d;          // (rows beyond the batch: zero gradient)
      *reinterpret_cast<float4*>(&colred[rr][col]) = d;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) sq += __shfl_xor(sq, o);
    if (lane == 0) lossred[wave] = sq;
  }
  __syncthreads();
  if (tid == 0) loss_part[blockIdx.x] = (lossred[0] + lossred[1]) + (lossred[2] + lossred[3]);
  if (tid < OF) {
    float s = 0.f;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) s += colred[k2][tid];
    dbias_part[(int64_t)blockIdx.x * OF + tid] = s;
  }

  // phase 2
  if (ABL == 1 || ABL == 3) return;
  float da[OF / 4];
#pragma unroll
  for (int sp = 0; sp < OF / 4; ++sp) da[sp] = dps[r][4 * sp + q];
  // (measured and not kept: groups of 32 columns with 8-byte loads / stores, so that stores leave earlier: 19.0
  //  against 17.4 us — profiles/r05_decode_fused.md)
#pragma unroll
  for (int gc = 0; gc < NG; ++gc) {
    f32x4_t o4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) o4[j] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int sp = 0; sp < OF / 4; ++sp) {
      if (ABL == 6) { o4[0][0] += da[sp] * bw2[gc][sp].x; o4[1][0] += bw2[gc][sp].y; o4[2][0] += bw2[gc][sp].z; o4[3][0] += bw2[gc][sp].w; continue; }
      o4[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(da[sp], bw2[gc][sp].x, o4[0], 0, 0, 0);
      o4[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(da[sp], bw2[gc][sp].y, o4[1], 0, 0, 0);
      o4[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(da[sp], bw2[gc][sp].z, o4[2], 0, 0, 0);
      o4[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(da[sp], bw2[gc][sp].w, o4[3], 0, 0, 0);
    }
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int64_t row = row0 + 4 * q + g;
      if (ABL == 5) { asm volatile("" :: "v"(o4[0][g]), "v"(o4[1][g]), "v"(o4[2][g]), "v"(o4[3][g])); continue; }
      if (row < batch)
        *reinterpret_cast<float4*>(dA + row * W + kbeg + 64 * gc + 4 * r) =
            make_float4(o4[0][g], o4[1][g], o4[2][g], o4[3][g]);
    }
  }
}

bool decode_fused_supported(int64_t batch, int W, int OF) {
  return OF == 48 && (W == 1024 || W == 512) && batch >= 1 && ceil_div(batch, 16) <= 1024;
}

int launch_decode_fused(hipStream_t s, const float* A, const float* Wd, const float* bd, const float* target,
                        float* pred, float* dpred, float* dA, float* loss_part, float* dbias_part, int64_t batch,
                        int W, int OF, float scale, int* nparts) {
  if (!decode_fused_supported(batch, W, OF) || !target || !dA || !loss_part || !dbias_part) return BLH_ERR_SHAPE;
  const int blocks = (int)ceil_div(batch, 16);
  const int SL = W / 8;
  const size_t lds = (size_t)(8 * 16 * SL + 16 * 52 + 16 * 48 + 4) * sizeof(float);
  static std::atomic<uint64_t> attr_done[2];
  int dev = 0;
  BLH_HIP_TRY(hipGetDevice(&dev));
  const uint64_t bit = 1ull << (dev & 63);
  std::atomic<uint64_t>& done = attr_done[W == 1024 ? 0 : 1];
  const void* kern = W == 1024 ? reinterpret_cast<const void*>(decode_fused_kernel<128>)
                               : reinterpret_cast<const void*>(decode_fused_kernel<64>);
  if (!(done.load(std::memory_order_acquire) & bit)) {
    BLH_HIP_TRY(hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    done.fetch_or(bit, std::memory_order_release);
  }
  if (W == 1024)
    launch_kernel(decode_fused_kernel<128>, dim3(blocks), dim3(512), lds, s, A, Wd, bd, target, pred, dpred, dA,
                       loss_part, dbias_part, batch, scale);
  else
    launch_kernel(decode_fused_kernel<64>, dim3(blocks), dim3(512), lds, s, A, Wd, bd, target, pred, dpred, dA,
                       loss_part, dbias_part, batch, scale);
  BLH_HIP_TRY(hipGetLastError());
  if (nparts) *nparts = blocks;
  return BLH_OK;
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
    hipLaunchKernelGGL((decode_fwd_mse_h_kernel<1, false>), dim3(blocks), dim3(64 * DEC_WAVES), 0, s, A, Wd, bd, target,
                       pred, dpred, dpred_h, loss_part, dbias_part, batch, W, OF, scale, nullptr, nullptr);
  else if (rows == 32)
    hipLaunchKernelGGL((decode_fwd_mse_h_kernel<2, false>), dim3(blocks), dim3(64 * DEC_WAVES), 0, s, A, Wd, bd, target,
                       pred, dpred, dpred_h, loss_part, dbias_part, batch, W, OF, scale, nullptr, nullptr);
  else
    hipLaunchKernelGGL((decode_fwd_mse_h_kernel<4, false>), dim3(blocks), dim3(64 * DEC_WAVES), 0, s, A, Wd, bd, target,
                       pred, dpred, dpred_h, loss_part, dbias_part, batch, W, OF, scale, nullptr, nullptr);
  BLH_HIP_TRY(hipGetLastError());
  if (nparts) *nparts = blocks;
  return BLH_OK;
}

// one-pass decode, bf16 storage: the kernel above with its phase 2 (dA = dP Wd into `dA`, bf16 [batch][W])
bool decode_fused_h_supported(int64_t batch, int W, int OF) {
  return OF == 48 && W % 128 == 0 && batch % 4 == 0 && decode_fwd_supported(batch, W, OF);
}
int64_t decode_wdT_elems(int W) { return (int64_t)W * 64; }

int launch_decode_fused_h(hipStream_t s, const uint16_t* A, const uint16_t* Wd, const uint16_t* WdT, const float* bd,
                          const float* target, float* pred, float* dpred, uint16_t* dpred_h, uint16_t* dA,
                          float* loss_part, float* dbias_part, int64_t batch, int W, int OF, float scale, int* nparts) {
  if (!decode_fused_h_supported(batch, W, OF) || !target || !WdT || !dA) return BLH_ERR_SHAPE;
  const int rows = decode_fwd_rows_per_block_h(batch);
  const int blocks = (int)ceil_div(batch, rows);
  if (rows == 16)
    launch_kernel(decode_fwd_mse_h_kernel<1, true>, dim3(blocks), dim3(64 * DEC_WAVES), 0, s, A, Wd, bd, target, pred,
                  dpred, dpred_h, loss_part, dbias_part, batch, W, OF, scale, WdT, dA);
  else if (rows == 32)
    launch_kernel(decode_fwd_mse_h_kernel<2, true>, dim3(blocks), dim3(64 * DEC_WAVES), 0, s, A, Wd, bd, target, pred,
                  dpred, dpred_h, loss_part, dbias_part, batch, W, OF, scale, WdT, dA);
  else
    launch_kernel(decode_fwd_mse_h_kernel<4, true>, dim3(blocks), dim3(64 * DEC_WAVES), 0, s, A, Wd, bd, target, pred,
                  dpred, dpred_h, loss_part, dbias_part, batch, W, OF, scale, WdT, dA);
  BLH_HIP_TRY(hipGetLastError());
  if (nparts) *nparts = blocks;
  return BLH_OK;
}

}  // namespace blh
