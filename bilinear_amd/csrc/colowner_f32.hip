// Column-owner BatchNorm kernels for the batches between the <= 384-row kernels (small_step.hip) and the full-chip
// path: 385 .. 1024 rows per GPU (the per-GPU shapes of the headline batch under strong scaling: 4096 rows over
// 4 or 8 GPUs; /root/reference/util/config.py:15-17, model/bilinear.py:7-13).  fp32 storage.
//
// At these sizes the multi-launch stage is launch-bound: the forward of a hidden stage was
//   split-K GEMM -> sum_slabs_bias -> col_stats_chunk -> bn_fwd_finalize -> bn_apply          (5 launches, 4 x ~4.7 us)
// and its backward  [sum of the data-gradient slabs ->] bn_bwd_reduce -> bn_bwd_finalize -> bn_bwd_apply (3-4 launches).
// A column of at most 1024 rows fits one workgroup, and BatchNorm couples nothing but the rows of a column: a
// workgroup that OWNS 32 columns for all rows needs no other workgroup's data, no partials and no second launch.
//   colowner_fwd   slabs (+ bias) -> Z, batch statistics (two passes over registers: mean, then sum (z - mean)^2),
//                  running statistics, scale / shift, A = 2 keep relu(z scale + shift) (+ skip), keep bits.
//   colowner_bwd   dA (a tensor, or the split-K slabs of the data gradient that produced it, + the block-skip
//                  gradient) -> S1, S2, dgamma, dbeta, dZ = scale dY' + a z + b', the column sums of dZ (bias
//                  gradient); where the summed dA is needed again (the block-boundary gradient) it is written out.
// Thread map: 1024 threads = 128 row groups x 8 column quads; a thread takes rows 8 rg + j (j = 0..7) and 4
// consecutive columns — the keep word of bn_f32.hip (8 rows x 4 columns) is one thread's, a wave's load touches
// 8 rows x 128 bytes = whole cache lines, and the stage's values stay in registers between the reduction and the
// apply pass.  32 workgroups at W = 1024, each streaming up to 8 split-K slabs of its columns: what bounds them is
// loads in flight, hence 16 waves per workgroup and the loads of a row group requested two slabs at a time before
// the first add (the first build — 256 threads, slab loads in a run-time loop — took 42-53 us per launch).
#include "common.h"
#include "philox.h"
#include "bn_f32_dev.h"
#include "bn_stats_dev.h"

namespace blh {

static constexpr int CO_COLS = 32;       // columns per workgroup
static constexpr int CO_THREADS = 1024;  // 128 row groups of 8 rows x 8 column quads: up to 1024 rows
static constexpr int CO_RG = CO_THREADS / 8;

__device__ __forceinline__ float4 co_ld(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void co_add(float4& a, const float4& b) { a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w; }

// v[j] = sum over the slabs of rows r[j] (+ the addend): the loads of two slabs x 8 rows are requested together
__device__ __forceinline__ void co_load_rows(float4 (&v)[8], const float* __restrict__ src, int splits,
                                             int64_t slab_stride, const float* __restrict__ addend, const int64_t (&off)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = co_ld(src + off[j]);
  int sp = 1;
  for (; sp + 1 < splits; sp += 2) {
    float4 a[8], b[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      a[j] = co_ld(src + sp * slab_stride + off[j]);
      b[j] = co_ld(src + (sp + 1) * slab_stride + off[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) { co_add(v[j], a[j]); co_add(v[j], b[j]); }
  }
  if (sp < splits) {
    float4 a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = co_ld(src + sp * slab_stride + off[j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) co_add(v[j], a[j]);
  }
  if (addend) {
    float4 a[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] = co_ld(addend + off[j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) co_add(v[j], a[j]);
  }
}

// sum over the 128 row-group threads that share a column quad, in a fixed order (deterministic), fp64 at the end:
// red[128][32] -> 8 partials of 16 row groups per column (the first 256 threads) -> every thread's 4 column totals
__device__ __forceinline__ void co_colsum(float4 v, float* red, float* red2, int rg, int c4, double (&tot)[4]) {
  __syncthreads();
  *reinterpret_cast<float4*>(&red[rg * CO_COLS + 4 * c4]) = v;
  __syncthreads();
  const int t = threadIdx.x;
  if (t < 256) {
    const int c = t & 31, part = t >> 5;
    float a = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) a += red[(part * 16 + r) * CO_COLS + c];
    red2[part * CO_COLS + c] = a;
  }
  __syncthreads();
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    double tt = 0.0;
#pragma unroll
    for (int part = 0; part < 8; ++part) tt += (double)red2[part * CO_COLS + 4 * c4 + c];
    tot[c] = tt;
  }
}

// every workgroup has read what it needs of the stage's counter before ANY of them bumps it: the last one to arrive
// at the ticket does (and resets the ticket for the next launch; launches of one stream follow each other)
__device__ __forceinline__ void co_bump_counter_last(int64_t* nbt, uint32_t* ticket) {
  __syncthreads();
  if (threadIdx.x == 0 && nbt) {
    const uint32_t t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
    if (t == gridDim.x - 1) {
      nbt[0] += 1;
      __hip_atomic_store(ticket, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// ---- forward ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CO_THREADS) void colowner_fwd_kernel(
    const float* __restrict__ slabs, int splits, int64_t slab_stride, const float* __restrict__ bias,
    const float* __restrict__ gamma, const float* __restrict__ beta, float* running_mean, float* running_var,
    int64_t* nbt, uint32_t* ticket, float momentum, float* __restrict__ saved, const float* __restrict__ skip,
    float* __restrict__ Z, float* __restrict__ A, uint32_t* __restrict__ keepbits, int64_t batch, int W,
    DropoutSrc drop) {
  __shared__ __attribute__((aligned(16))) float red[CO_RG * CO_COLS];
  __shared__ __attribute__((aligned(16))) float red2[8 * CO_COLS];
  __shared__ __attribute__((aligned(16))) float stat[2][CO_COLS];       // scale, shift of the workgroup's columns
  const int t = threadIdx.x, c4 = t & 7, rg = t >> 3;
  const int col = blockIdx.x * CO_COLS + 4 * c4;
  const int64_t rg0 = 8 * (int64_t)rg;
  const float4 bv = co_ld(bias + col);
  int64_t off[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) off[j] = min(rg0 + j, batch - 1) * W + col;
  float4 z[8];
  co_load_rows(z, slabs, splits, slab_stride, nullptr, off);
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    co_add(z[j], bv);
    if (rg0 + j < batch) { co_add(s, z[j]); *reinterpret_cast<float4*>(Z + off[j]) = z[j]; }
  }
  double tot[4];
  co_colsum(s, red, red2, rg, c4, tot);
  const double B = (double)batch;
  const float4 mean = make_float4((float)(tot[0] / B), (float)(tot[1] / B), (float)(tot[2] / B), (float)(tot[3] / B));
  float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (rg0 + j < batch) {
      const float dx = z[j].x - mean.x, dy = z[j].y - mean.y, dz = z[j].z - mean.z, dw = z[j].w - mean.w;
      q.x = fmaf(dx, dx, q.x); q.y = fmaf(dy, dy, q.y); q.z = fmaf(dz, dz, q.z); q.w = fmaf(dw, dw, q.w);
    }
  double m2[4];
  co_colsum(q, red, red2, rg, c4, m2);
  if (rg == 0) {      // 8 threads x 4 columns: the statistics of the workgroup's 32 columns
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      bn_finalize_store(tot[c] / B, m2[c], batch, col + c, gamma, beta, running_mean, running_var, nbt, momentum,
                        saved, saved + W, saved + 2 * W, saved + 3 * W);
      stat[0][4 * c4 + c] = saved[2 * W + col + c];      // (this thread's own stores: program order)
      stat[1][4 * c4 + c] = saved[3 * W + col + c];
    }
  }
  __syncthreads();
  const float4 sc = *reinterpret_cast<const float4*>(&stat[0][4 * c4]), sh = *reinterpret_cast<const float4*>(&stat[1][4 * c4]);
  if (rg0 < batch) {
    const uint32_t kw = f2_keep_word(drop, rg0 & ~(int64_t)31, rg & 3, col, W, batch);
    keepbits[(rg0 >> 3) * (W >> 2) + (col >> 2)] = kw;
    float4 sk[8];
    if (skip) {
#pragma unroll
      for (int j = 0; j < 8; ++j) sk[j] = co_ld(skip + off[j]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if (rg0 + j >= batch) break;
      const uint32_t nib = kw >> (4 * j);
      float4 a;
      a.x = fmaxf(fmaf(z[j].x, sc.x, sh.x), 0.f); a.y = fmaxf(fmaf(z[j].y, sc.y, sh.y), 0.f);
      a.z = fmaxf(fmaf(z[j].z, sc.z, sh.z), 0.f); a.w = fmaxf(fmaf(z[j].w, sc.w, sh.w), 0.f);
      a.x = (nib & 1u) ? a.x * 2.f : 0.f; a.y = (nib & 2u) ? a.y * 2.f : 0.f;
      a.z = (nib & 4u) ? a.z * 2.f : 0.f; a.w = (nib & 8u) ? a.w * 2.f : 0.f;
      if (skip) co_add(a, sk[j]);
      *reinterpret_cast<float4*>(A + off[j]) = a;
    }
  }
  co_bump_counter_last(nbt, ticket);
}

// ---- backward --------------------------------------------------------------------------------------------------------
// dA = sum of `splits` slabs (+ addend); written to dA_out when that is not null (the block-boundary gradient is an
// operand again two stages further down).  One pass over memory: dA and z of the thread's 8 rows stay in registers
// between the column reductions and the dZ pass.
__global__ __launch_bounds__(CO_THREADS) void colowner_bwd_kernel(
    const float* __restrict__ dA_src, int splits, int64_t slab_stride, const float* __restrict__ addend,
    float* __restrict__ dA_out, const float* __restrict__ Z, const float* __restrict__ saved,
    const uint32_t* __restrict__ keepbits, float* __restrict__ dZ, float* __restrict__ dgamma,
    float* __restrict__ dbeta, float* __restrict__ db_rows, int db_nrows, double* __restrict__ sq_gb, int64_t batch,
    int W) {
  __shared__ __attribute__((aligned(16))) float red[CO_RG * CO_COLS];
  __shared__ __attribute__((aligned(16))) float red2[8 * CO_COLS];
  const int t = threadIdx.x, c4 = t & 7, rg = t >> 3;
  const int col = blockIdx.x * CO_COLS + 4 * c4;
  const int64_t rg0 = 8 * (int64_t)rg;
  const float4 mu = co_ld(saved + col), is = co_ld(saved + W + col), sc = co_ld(saved + 2 * W + col),
               sh = co_ld(saved + 3 * W + col);
  int64_t off[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) off[j] = min(rg0 + j, batch - 1) * W + col;
  const uint32_t kw = rg0 < batch ? keepbits[(rg0 >> 3) * (W >> 2) + (col >> 2)] : 0u;
  float4 z[8], g[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) z[j] = co_ld(Z + off[j]);
  co_load_rows(g, dA_src, splits, slab_stride, addend, off);
  float4 s1 = make_float4(0.f, 0.f, 0.f, 0.f), s2 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const bool ok = rg0 + j < batch;
    if (dA_out && ok) *reinterpret_cast<float4*>(dA_out + off[j]) = g[j];
    const uint32_t nib = ok ? (kw >> (4 * j)) : 0u;
    // g becomes dY' = 2 keep [y > 0] dA
    g[j].x = ((nib & 1u) && (fmaf(z[j].x, sc.x, sh.x) > 0.f)) ? g[j].x * 2.f : 0.f;
    g[j].y = ((nib & 2u) && (fmaf(z[j].y, sc.y, sh.y) > 0.f)) ? g[j].y * 2.f : 0.f;
    g[j].z = ((nib & 4u) && (fmaf(z[j].z, sc.z, sh.z) > 0.f)) ? g[j].z * 2.f : 0.f;
    g[j].w = ((nib & 8u) && (fmaf(z[j].w, sc.w, sh.w) > 0.f)) ? g[j].w * 2.f : 0.f;
    co_add(s2, g[j]);
    s1.x = fmaf(g[j].x, z[j].x, s1.x); s1.y = fmaf(g[j].y, z[j].y, s1.y);
    s1.z = fmaf(g[j].z, z[j].z, s1.z); s1.w = fmaf(g[j].w, z[j].w, s1.w);
  }
  double t1[4], t2[4];
  co_colsum(s1, red, red2, rg, c4, t1);
  co_colsum(s2, red, red2, rg, c4, t2);
  // dgamma = invstd (S1 - mean S2), dbeta = S2 (bn_bwd_finalize_h2's arithmetic); a, b' as bn_bwd_apply_f2 forms them
  const double B = (double)batch;
  const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, isv[4] = {is.x, is.y, is.z, is.w}, scv[4] = {sc.x, sc.y, sc.z, sc.w};
  float dg[4], db[4], ca[4], cb[4];
  const float inv_b = (float)(1.0 / B);
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    dg[c] = (float)((double)isv[c] * (t1[c] - (double)muv[c] * t2[c]));
    db[c] = (float)t2[c];
    const float tx = scv[c] * (dg[c] * inv_b) * isv[c];
    ca[c] = -tx;
    cb[c] = fmaf(tx, muv[c], -scv[c] * (db[c] * inv_b));
  }
  double q2 = 0.0;
  if (rg == 0) {
    *reinterpret_cast<float4*>(dgamma + col) = make_float4(dg[0], dg[1], dg[2], dg[3]);
    *reinterpret_cast<float4*>(dbeta + col) = make_float4(db[0], db[1], db[2], db[3]);
#pragma unroll
    for (int c = 0; c < 4; ++c) q2 += (double)dg[c] * (double)dg[c] + (double)db[c] * (double)db[c];
  }
  float4 cs = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    if (rg0 + j >= batch) break;
    float4 o;
    o.x = fmaf(sc.x, g[j].x, fmaf(ca[0], z[j].x, cb[0])); o.y = fmaf(sc.y, g[j].y, fmaf(ca[1], z[j].y, cb[1]));
    o.z = fmaf(sc.z, g[j].z, fmaf(ca[2], z[j].z, cb[2])); o.w = fmaf(sc.w, g[j].w, fmaf(ca[3], z[j].w, cb[3]));
    co_add(cs, o);
    *reinterpret_cast<float4*>(dZ + off[j]) = o;
  }
  double tz[4];
  co_colsum(cs, red, red2, rg, c4, tz);
  if (rg == 0)
    *reinterpret_cast<float4*>(db_rows + col) = make_float4((float)tz[0], (float)tz[1], (float)tz[2], (float)tz[3]);
  // (db_rows: the stage's slot of the bias column-sum partials, [db_nrows][W]: row 0 carries the sums, the others zero)
  for (int r = 1 + rg; r < db_nrows; r += CO_RG)
    *reinterpret_cast<float4*>(db_rows + (int64_t)r * W + col) = make_float4(0.f, 0.f, 0.f, 0.f);
  if (sq_gb) {   // two slots of bn_bwd_finalize_h2's layout (16 columns each) per workgroup: the sum in the first
    q2 += __shfl_xor(q2, 1); q2 += __shfl_xor(q2, 2); q2 += __shfl_xor(q2, 4);     // threads 0 .. 7 (rg == 0)
    if (t == 0) { sq_gb[2 * blockIdx.x] = q2; sq_gb[2 * blockIdx.x + 1] = 0.0; }
  }
}

// ---- host side ---------------------------------------------------------------------------------------------------------
bool colowner_supported(int64_t batch, int W) {
  return W % CO_COLS == 0 && batch > 384 && batch <= 8 * CO_RG;
}

int launch_colowner_fwd(hipStream_t s, const float* slabs, int splits, int64_t slab_stride, const float* bias,
                        const float* gamma, const float* beta, float* running_mean, float* running_var, int64_t* nbt,
                        uint32_t* ticket, float momentum, float* saved, const float* skip, float* Z, float* A,
                        uint32_t* keepbits, int64_t batch, int W, const DropoutSrc& drop) {
  if (!colowner_supported(batch, W) || !ticket) return BLH_ERR_SHAPE;
  hipLaunchKernelGGL(colowner_fwd_kernel, dim3(W / CO_COLS), dim3(CO_THREADS), 0, s, slabs, splits, slab_stride, bias,
                     gamma, beta, running_mean, running_var, nbt, ticket, momentum, saved, skip, Z, A, keepbits, batch, W,
                     drop);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

int launch_colowner_bwd(hipStream_t s, const float* dA_src, int splits, int64_t slab_stride, const float* addend,
                        float* dA_out, const float* Z, const float* saved, const uint32_t* keepbits, float* dZ,
                        float* dgamma, float* dbeta, float* db_rows, int db_nrows, double* sq_gb, int64_t batch, int W) {
  if (!colowner_supported(batch, W)) return BLH_ERR_SHAPE;
  hipLaunchKernelGGL(colowner_bwd_kernel, dim3(W / CO_COLS), dim3(CO_THREADS), 0, s, dA_src, splits, slab_stride, addend, dA_out,
                     Z, saved, keepbits, dZ, dgamma, dbeta, db_rows, db_nrows, sq_gb, batch, W);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

}  // namespace blh
