// Hidden stages of the fp32 step at 385 .. 2048 rows — the per-GPU shapes of the headline batch under strong scaling,
// where a stage is a chain of launches at the ~4.7 us launch floor (profiles/r05_colowner.md: one launch per stage on
// column-owner workgroups lost, 32 workgroups cannot stream what 2048 can).  Here launches are merged PAIRWISE, only
// where no workgroup needs another workgroup's result inside the launch, and every grid keeps its size:
//
//   forward   split-K GEMM -> [slab sum + bias + Z + 64-row column statistics]            (was sum_slabs_bias, col_stats_chunk)
//                          -> [statistics merge (every block, for its own 256 columns) + BatchNorm / ReLU / dropout]
//                                                                                          (was bn_fwd_finalize, bn_apply_f2)
//
// The arithmetic is the multi-launch path's, operation for operation and in its order: results are bit-identical
// (tests/test_gpu_timed_path.py::test_mid_batch_pair_fusions_are_bit_identical).  MEASURED, NOT ADOPTED (opt-in
// BLH_MID_PAIR=1 [BLH_MID_PAIR_APPLY=1]; profiles/r05_mid_pair.md): the merged launches take 7.8 and 9.4-10.2 us
// against 4.8 + 4.6 and 4.7 + 4.8 us for the pairs they replace, and the step is 1-2 % slower — a dependent memory
// round trip costs inside a kernel what it costs across a launch boundary on this stack.  Reference arithmetic:
// /root/reference/model/bilinear.py:7-13 (Linear -> BatchNorm1d -> ReLU -> Dropout).
#include "common.h"
#include "philox.h"
#include "bn_f32_dev.h"
#include "bn_stats_dev.h"

namespace blh {

namespace {
__device__ __forceinline__ float4 mid_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ float mid_finite_abs(float x) {
  const float a = fabsf(x);
  return a <= 3.402823466e+38f ? a : 0.f;
}
constexpr int MID_ROWS = 64;       // rows of a statistics chunk (= elementwise.hip: FWD_FINISH_ROWS)
}  // namespace

// ---- forward, first launch: Z = sum of the split-K slabs + bias, and the (mean, M2) of every column over the block's
// 64 rows.  block = 128 columns (32 lanes x float4) x 8 row lanes x 8 rows; grid (N / 128, ceil(M / 64)).
// Per column the operations are sum_slabs_bias_kernel's and col_stats_chunk_kernel's (elementwise.hip).
__global__ __launch_bounds__(256) void mid_fwd_finish_kernel(const float* __restrict__ slabs, int64_t count, int splits,
                                                             int64_t M, int N, const float* __restrict__ bias,
                                                             float* __restrict__ Z, float* __restrict__ stat_part) {
  __shared__ __attribute__((aligned(16))) float red[8][128];
  const int cl = threadIdx.x & 31, rl = threadIdx.x >> 5;
  const int col = blockIdx.x * 128 + cl * 4;
  const int64_t r0 = (int64_t)blockIdx.y * MID_ROWS;
  const int cnt = (int)min<int64_t>(MID_ROWS, M - r0);
  const float4 b = mid_ld4(bias + col);
  float4 x[8];
  // (the remainder loop of sum_slabs_bias_kernel — splits not a multiple of 4 — is kept: same operations)
#pragma unroll
  for (int jj = 0; jj < 8; jj += 4) {         // four rows' slabs in flight
    float4 a[4];
    const int s4 = splits & ~3;
#pragma unroll
    for (int h = 0; h < 4; ++h) a[h] = b;
    for (int s = 0; s < s4; s += 4) {
      float4 v[4][4];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int r = min(rl * 8 + jj + h, cnt - 1);
        const int64_t e = (r0 + r) * N + col;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[h][q] = mid_ld4(slabs + (int64_t)(s + q) * count + e);
      }
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        a[h].x += (v[h][0].x + v[h][1].x) + (v[h][2].x + v[h][3].x); a[h].y += (v[h][0].y + v[h][1].y) + (v[h][2].y + v[h][3].y);
        a[h].z += (v[h][0].z + v[h][1].z) + (v[h][2].z + v[h][3].z); a[h].w += (v[h][0].w + v[h][1].w) + (v[h][2].w + v[h][3].w);
      }
    }
    for (int s = s4; s < splits; ++s) {
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        const int r = min(rl * 8 + jj + h, cnt - 1);
        const float4 bb = mid_ld4(slabs + (int64_t)s * count + (r0 + r) * N + col);
        a[h].x += bb.x; a[h].y += bb.y; a[h].z += bb.z; a[h].w += bb.w;
      }
    }
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const int r = rl * 8 + jj + h;
      if (r < cnt) *reinterpret_cast<float4*>(Z + (r0 + r) * N + col) = a[h];
      else a[h] = make_float4(0.f, 0.f, 0.f, 0.f);
      x[jj + h] = a[h];
    }
  }
  if (!stat_part) return;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < 8; ++j) { s.x += x[j].x; s.y += x[j].y; s.z += x[j].z; s.w += x[j].w; }
  *reinterpret_cast<float4*>(&red[rl][cl * 4]) = s;
  __syncthreads();
  float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int w = 0; w < 8; ++w) {
    const float4 v = *reinterpret_cast<const float4*>(&red[w][cl * 4]);
    t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
  }
  const float fc = (float)cnt;
  const float4 mean = make_float4(t.x / fc, t.y / fc, t.z / fc, t.w / fc);
  __syncthreads();
  float4 d2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (rl * 8 + j < cnt) {
      const float dx = x[j].x - mean.x, dy = x[j].y - mean.y, dz = x[j].z - mean.z, dw = x[j].w - mean.w;
      d2.x = fmaf(dx, dx, d2.x); d2.y = fmaf(dy, dy, d2.y); d2.z = fmaf(dz, dz, d2.z); d2.w = fmaf(dw, dw, d2.w);
    }
  }
  *reinterpret_cast<float4*>(&red[rl][cl * 4]) = d2;
  __syncthreads();
  if (rl == 0) {
    float4 m2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int w = 0; w < 8; ++w) {
      const float4 v = *reinterpret_cast<const float4*>(&red[w][cl * 4]);
      m2.x += v.x; m2.y += v.y; m2.z += v.z; m2.w += v.w;
    }
    *reinterpret_cast<float4*>(stat_part + ((int64_t)blockIdx.y * 2 + 0) * N + col) = mean;
    *reinterpret_cast<float4*>(stat_part + ((int64_t)blockIdx.y * 2 + 1) * N + col) = m2;
  }
}

// ---- forward, second launch: every block merges the (mean, M2) chunk partials of ITS 256 columns (Chan et al., fp64,
// in bn_fwd_finalize_kernel's order: 8 slices of tiles s, s + 8, ..., added in slice order), then runs bn_apply_f2's
// body.  The blocks of the first row chunk also store what backward and the module read: saved mean / invstd / scale /
// shift and the running statistics.  tiles <= 64.  The partial array is tiles x 2 x W floats (128 KiB at 1024 rows):
// every block re-reads its 256-column share from L2, which is what a separate finalize launch costs less than at
// these sizes and more than at 4096 rows (profiles/r03_bn_finalize_merge.md: +0.9 % there).
__global__ __launch_bounds__(256) void mid_bn_apply_kernel(
    const float* __restrict__ Z, const float* __restrict__ part, int tiles, int tile_rows,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean, float* running_var,
    float momentum, float* __restrict__ saved, const float* __restrict__ skip, float* __restrict__ A,
    uint32_t* __restrict__ keepbits, int64_t batch, int W, int row_chunk, DropoutSrc drop, int64_t* nbt,
    float* __restrict__ amax_part) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = blockIdx.x * 256 + lane * 4;
  if (nbt && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) nbt[0] += 1;
  float am = 0.f;
  const bool okc = col < W;
  const int cc = okc ? col : 0;
  // the first rows of Z are requested before the merge: their latency hides behind it
  const int64_t r0 = (int64_t)blockIdx.y * row_chunk;
  const int64_t r1 = min(batch, r0 + row_chunk);
  float sc[4], sh[4];
  {
    // acc[sl][c]: slice sl = t & 7 takes tiles t = sl, sl + 8, ... in ascending order; the slices are then added in
    // slice order — bn_fwd_finalize_kernel's sums exactly.  Loads of 16 tiles are requested together (a load per
    // loop iteration made this prologue sixteen dependent round trips: 8 us per launch).
    double acc[8][4];
    double mean1[4], m2[4];
    auto clear = [&]() {
#pragma unroll
      for (int sl = 0; sl < 8; ++sl)
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[sl][c] = 0.0;
    };
    clear();
    for (int t0 = 0; t0 < tiles; t0 += 16) {
      float4 mu[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) mu[u] = mid_ld4(part + ((int64_t)min(t0 + u, tiles - 1) * 2 + 0) * W + cc);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int t = t0 + u;
        const double n = (double)min((int64_t)tile_rows, batch - (int64_t)t * tile_rows);
        if (t < tiles) {
          acc[u & 7][0] += n * (double)mu[u].x; acc[u & 7][1] += n * (double)mu[u].y;
          acc[u & 7][2] += n * (double)mu[u].z; acc[u & 7][3] += n * (double)mu[u].w;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double m = 0.0;
#pragma unroll
      for (int sl = 0; sl < 8; ++sl) m += acc[sl][c];
      mean1[c] = m / (double)batch;
    }
    clear();
    for (int t0 = 0; t0 < tiles; t0 += 16) {
      float4 mu[16], mt[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int64_t tt = min(t0 + u, tiles - 1);
        mu[u] = mid_ld4(part + (tt * 2 + 0) * W + cc);
        mt[u] = mid_ld4(part + (tt * 2 + 1) * W + cc);
      }
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const int t = t0 + u;
        const double n = (double)min((int64_t)tile_rows, batch - (int64_t)t * tile_rows);
        const double d0 = (double)mu[u].x - mean1[0], d1 = (double)mu[u].y - mean1[1], d2 = (double)mu[u].z - mean1[2],
                     d3 = (double)mu[u].w - mean1[3];
        if (t < tiles) {
          acc[u & 7][0] += (double)mt[u].x + n * d0 * d0; acc[u & 7][1] += (double)mt[u].y + n * d1 * d1;
          acc[u & 7][2] += (double)mt[u].z + n * d2 * d2; acc[u & 7][3] += (double)mt[u].w + n * d3 * d3;
        }
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      double m = 0.0;
#pragma unroll
      for (int sl = 0; sl < 8; ++sl) m += acc[sl][c];
      m2[c] = m;
    }
    const float4 g = mid_ld4(gamma + cc), b = mid_ld4(beta + cc);
    const float gv[4] = {g.x, g.y, g.z, g.w}, bv[4] = {b.x, b.y, b.z, b.w};
    BnColumn bc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      bc[c] = bn_finalize_values(mean1[c], m2[c], batch, gv[c], bv[c]);
      sc[c] = bc[c].sc; sh[c] = bc[c].sh;
    }
    if (blockIdx.y == 0 && w == 0 && okc) {
      *reinterpret_cast<float4*>(saved + 0 * (int64_t)W + col) = make_float4(bc[0].mu, bc[1].mu, bc[2].mu, bc[3].mu);
      *reinterpret_cast<float4*>(saved + 1 * (int64_t)W + col) =
          make_float4(bc[0].invstd, bc[1].invstd, bc[2].invstd, bc[3].invstd);
      *reinterpret_cast<float4*>(saved + 2 * (int64_t)W + col) = make_float4(sc[0], sc[1], sc[2], sc[3]);
      *reinterpret_cast<float4*>(saved + 3 * (int64_t)W + col) = make_float4(sh[0], sh[1], sh[2], sh[3]);
#pragma unroll
      for (int c = 0; c < 4; ++c)
        bn_running_update(mean1[c], m2[c], batch, col + c, running_mean, running_var, nullptr, momentum);
    }
  }
  const int W4 = W >> 2;
  for (int64_t base = r0; okc && base < r1; base += 32) {
    const int64_t rg = base + 8 * w;
    if (rg >= batch) break;
    float4 z[8], k[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int64_t r = min(rg + j, batch - 1);
      z[j] = mid_ld4(Z + r * W + col);
      if (skip) k[j] = mid_ld4(skip + r * W + col);
    }
    const uint32_t kw = f2_keep_word(drop, base, w, col, W, batch);
    if (keepbits) keepbits[(rg >> 3) * W4 + (col >> 2)] = kw;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const uint32_t nib = kw >> (4 * j);
      float4 a;
      a.x = fmaxf(fmaf(z[j].x, sc[0], sh[0]), 0.f); a.y = fmaxf(fmaf(z[j].y, sc[1], sh[1]), 0.f);
      a.z = fmaxf(fmaf(z[j].z, sc[2], sh[2]), 0.f); a.w = fmaxf(fmaf(z[j].w, sc[3], sh[3]), 0.f);
      a.x = (nib & 1u) ? a.x * 2.f : 0.f; a.y = (nib & 2u) ? a.y * 2.f : 0.f;
      a.z = (nib & 4u) ? a.z * 2.f : 0.f; a.w = (nib & 8u) ? a.w * 2.f : 0.f;
      if (skip) { a.x += k[j].x; a.y += k[j].y; a.z += k[j].z; a.w += k[j].w; }
      if (rg + j < batch) {
        *reinterpret_cast<float4*>(A + (rg + j) * W + col) = a;
        am = fmaxf(fmaxf(am, fmaxf(mid_finite_abs(a.x), mid_finite_abs(a.y))),
                   fmaxf(mid_finite_abs(a.z), mid_finite_abs(a.w)));
      }
    }
  }
  if (amax_part) {     // one max-|value| partial per wave (bn_f32.hip: f2_wave_amax_store)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) am = fmaxf(am, __shfl_xor(am, o));
    if (lane == 0) amax_part[((int64_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + w] = am;
  }
}

// ---- host ---------------------------------------------------------------------------------------------------------
bool mid_fwd_pair_supported(int64_t batch, int W, float momentum) {
  // (momentum < 0, the cumulative average, reads num_batches_tracked in the finalize step while the apply step
  //  advances it: two launches keep that ordered, one launch would not)
  return W % 256 == 0 && batch >= 1 && ceil_div(batch, MID_ROWS) <= 64 && momentum >= 0.f;
}

int launch_mid_fwd_finish(hipStream_t s, const float* slabs, int splits, int64_t M, int N, const float* bias, float* Z,
                          float* stat_part) {
  if (N % 128 != 0) return BLH_ERR_SHAPE;
  hipLaunchKernelGGL(mid_fwd_finish_kernel, dim3((unsigned)(N / 128), (unsigned)ceil_div(M, MID_ROWS)), dim3(256), 0, s,
                     slabs, M * (int64_t)N, splits, M, N, bias, Z, stat_part);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

int launch_mid_bn_apply(hipStream_t s, const float* Z, const float* stat_part, int tiles, int tile_rows,
                        const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                        float* saved, const float* skip, float* A, uint32_t* keepbits, int64_t batch, int W,
                        const DropoutSrc& drop, int64_t* nbt, float* amax_part) {
  if (W % 256 != 0 || tiles > 64) return BLH_ERR_SHAPE;
  hipLaunchKernelGGL(mid_bn_apply_kernel, dim3((unsigned)(W / 256), (unsigned)ew_num_row_chunks(batch)), dim3(256), 0, s, Z,
                     stat_part, tiles, tile_rows, gamma, beta, running_mean, running_var, momentum, saved, skip, A,
                     keepbits, batch, W, ew_row_chunk(batch), drop, nbt, amax_part);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

}  // namespace blh
