// The training step, the drop-in forward / backward and the eval forward at the reference's own batch size
// (util/config.py:15: batch_size = 64; the stage kernels take up to 384 rows): zero_grad, forward
// (Linear -> BatchNorm1d -> ReLU -> Dropout stages with block skips, /root/reference/model/bilinear.py:7-13,31-41),
// MSE, backward, clip_grad_norm_(1) and Adam (/root/reference/train_bilinear.py:75-83).
//
// Why: at 64 rows the multi-launch step is ~50 kernels of 5 us each, every one of them nothing but launch latency
// and a cold first read (0.32 ms per step at 2 x 1024, hipGraph replay no faster).  Here the batch is small enough
// for a workgroup that owns FOUR columns of a stage to own them for all rows: BatchNorm statistics, the BatchNorm
// backward sums, the bias / gamma / beta gradients and the dropout patch (32 rows x 4 columns per Philox call,
// philox.h) are all local to it; only the activations (forward) and dZ (backward) of a stage cross workgroups.
//
// ONE LAUNCH PER STAGE (0.150 ms per step at 2 x 1024, batch 64): stage kernels, a decode kernel, one
// batched GEMM launch for the hidden weight gradients, the library's clip + Adam kernel.  (The same stage bodies as
// one PERSISTENT launch with a grid barrier per stage measured 0.200 ms and needed the whole grid resident; it left
// the library in round 6 and lives on as a developer tool: tools/small_step_persistent.h, profiles/r04_small_step.md.)
//
// Work split: workgroup g owns column group cg(g) (XCD-aware: the 32 workgroups of one XCD own 128 adjacent
// columns, so the strided weight columns the data gradient reads are fetched once per XCD L2).  256 threads =
// 4 waves x 16 rows; the 64 lanes of a wave split the reduction index four elements each (float4, coalesced
// rows of the activation), every lane accumulates 8 x 4 blocks and a shuffle butterfly leaves element
// (row, column) = (tid / 4, tid % 4) in thread tid.  That mapping is the same in every stage, forward and backward,
// so what backward needs of forward (x-hat, the ReLU/dropout gate, the BatchNorm scale) belongs to the same thread:
// between stage launches it waits in the workspace.
//
// Arithmetic is fp32 FMA on the vector ALU (0.5 MFLOP per workgroup and stage: the matrix cores have nothing to
// win at 64 x 4 tiles); column statistics in fp32 over <= 384 rows, the norm of the gradient in fp64 partials
// summed in a fixed order (deterministic).  Same Philox keep bits, same Adam arithmetic (clip_adam_body,
// elementwise.hip) as the multi-launch path.
#include <atomic>

#include "common.h"
#include "philox.h"
#include "small_step.h"

namespace blh {

namespace {

constexpr int SS_THREADS = 256;
constexpr int SS_STAGED_MAX_ROWS = 384;  // the stage kernels (8 waves per workgroup above 64 rows, row blocks of 128;
                                         // at 512 rows the multi-launch path is as fast: 0.466 against 0.477 ms)
constexpr int SS_PASSES = 4;            // reduction length <= 4 * 256
constexpr int SS_SMALLK = 64;           // encode fan-in up to which the staged stage-0 kernel keeps x in LDS
constexpr int SS_RB = 8;                // rows per batch of loads in ss_gemm (x 4 passes = 32 loads in flight per wave;
                                        // all 64 at once was tried: the allocator spills)

typedef float4 WBlock[SS_PASSES][4];

__device__ __forceinline__ float4 ss_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
// What other workgroups read behind the next barrier (activations, dZ, dpred) is stored write-through at agent
// scope (global_store ... sc1): the barrier then needs no L2 write-back on the arriving side (1.9 us of 7.3,
// tools/grid_barrier_bench.hip), only the drain of the stores that arrive_published() waits for.
__device__ __forceinline__ void ss_publish(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// (the gradients too: every workgroup reads the whole gradient arena behind the last barrier)
__device__ __forceinline__ void ss_publish4(float* p, float4 v) {
  typedef float f32x4_t __attribute__((ext_vector_type(4)));
  const f32x4_t q = {v.x, v.y, v.z, v.w};
  // (s_nop: a VMEM store of more than 64 bits must not be followed at once by a VALU write of the VGPRs that hold
  //  its data — the compiler inserts that wait state for its own stores, it cannot for inline asm; without it the
  //  split kernels stored 20 % wrong weight gradients, tools_dev/small_dropin_debug.py)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(q) : "memory");
}

// weights of one reduction pass, ROWS form (forward): wb[j][c] = Wrow[c][k .. k+3], k = 256 j + 4 lane
__device__ __forceinline__ void ss_load_w_rows(WBlock& wb, const float* __restrict__ Wp, int K, int lane) {
#pragma unroll
  for (int j = 0; j < SS_PASSES; ++j) {
    const int k = j * 256 + lane * 4;
    const int kk = min(k, K - 4);          // (loads are unconditional: a load under a lane mask costs a branch and
#pragma unroll                             //  a full wait each; out-of-range lanes get zero weights instead)
    for (int c = 0; c < 4; ++c) {
      const float4 w = ss_ld4(Wp + (int64_t)c * K + kk);
      wb[j][c] = (k < K) ? w : float4{0.f, 0.f, 0.f, 0.f};
    }
  }
}
// COLUMNS form (data gradient): wb[j][q] = W[m + q][n0 .. n0+3], m = 256 j + 4 lane; ld = row length of W
__device__ __forceinline__ void ss_load_w_cols(WBlock& wb, const float* __restrict__ Wp, int M, int ld, int lane) {
#pragma unroll
  for (int j = 0; j < SS_PASSES; ++j) {
    const int m = j * 256 + lane * 4;
    const int mm = min(m, M - 4);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 w = ss_ld4(Wp + (int64_t)(mm + q) * ld);
      wb[j][q] = (m < M) ? w : float4{0.f, 0.f, 0.f, 0.f};
    }
  }
}

// out(row, c) = sum_k In[row][k] * w(k, c) for the wave's 16 rows; returns the element (16 wave + lane / 4, lane % 4).
//  * Two halves of 8 rows: 32 accumulators next to the 64 weight registers.  Each half ends in a 32-value
//    butterfly (31 shuffles + 1) that leaves the sum of value (lane % 32) in both half-waves; lanes 0-31 keep the
//    first half's element, lanes 32-63 the second's, which is the (lane / 4, lane % 4) mapping.
//  * A batch of loads is SS_RB WHOLE ROWS (x 4 passes of 1 KB): rows are exactly 4 KB apart, so a batch of one
//    1 KB piece of many rows would sit on the same few L2 channels for every wave of every CU at once.
//  * Loads are unconditional and all four passes always run (clamped index, zero weights past K): a load under a
//    lane mask, or a pass under a uniform branch, costs a branch and a full wait each.  sched_barrier keeps the scheduler from hoisting every batch to the top (spills).
template <bool COLS>
__device__ __forceinline__ float ss_gemm(const WBlock& wb, const float* __restrict__ In, int ld, int K, int batch,
                                         int wave, int lane) {
  float out = 0.f;
  // (addresses as uniform row base + unsigned 32-bit lane offset: the saddr form of global_load, one VGPR per
  //  pass instead of a 64-bit address pair per load)
  uint32_t koff[SS_PASSES];
#pragma unroll
  for (int j = 0; j < SS_PASSES; ++j) koff[j] = (uint32_t)min(j * 256 + lane * 4, K - 4) * 4u;
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    float acc[8][4];
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[r][c] = 0.f;
#pragma unroll
    for (int rp = 0; rp < 8 / SS_RB; ++rp) {
      float4 a[SS_RB][SS_PASSES];
#pragma unroll
      for (int u = 0; u < SS_RB; ++u) {
        const int row = min(wave * 16 + half * 8 + rp * SS_RB + u, batch - 1);
        const char* base = reinterpret_cast<const char*>(In + (int64_t)row * ld);
#pragma unroll
        for (int j = 0; j < SS_PASSES; ++j) a[u][j] = *reinterpret_cast<const float4*>(base + koff[j]);
      }
#pragma unroll
      for (int u = 0; u < SS_RB; ++u) {
        const int r = rp * SS_RB + u;
#pragma unroll
        for (int j = 0; j < SS_PASSES; ++j) {
          {
            const float4 av = a[u][j];
            if (COLS) {
              const float4 w0 = wb[j][0], w1 = wb[j][1], w2 = wb[j][2], w3 = wb[j][3];
              acc[r][0] = fmaf(av.x, w0.x, fmaf(av.y, w1.x, fmaf(av.z, w2.x, fmaf(av.w, w3.x, acc[r][0]))));
              acc[r][1] = fmaf(av.x, w0.y, fmaf(av.y, w1.y, fmaf(av.z, w2.y, fmaf(av.w, w3.y, acc[r][1]))));
              acc[r][2] = fmaf(av.x, w0.z, fmaf(av.y, w1.z, fmaf(av.z, w2.z, fmaf(av.w, w3.z, acc[r][2]))));
              acc[r][3] = fmaf(av.x, w0.w, fmaf(av.y, w1.w, fmaf(av.z, w2.w, fmaf(av.w, w3.w, acc[r][3]))));
            } else {
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                const float4 w = wb[j][c];
                acc[r][c] = fmaf(av.x, w.x, fmaf(av.y, w.y, fmaf(av.z, w.z, fmaf(av.w, w.w, acc[r][c]))));
              }
            }
          }
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = acc[i >> 2][i & 3];
#pragma unroll
    for (int h = 16; h >= 1; h >>= 1) {
      const bool up = (lane & h) != 0;
#pragma unroll
      for (int i = 0; i < h; ++i) {
        // (the empty asm keeps the two operands opaque: otherwise the select of two array elements is folded
        //  into ONE dynamically indexed element, which is lowered to a compare/select chain over all 32)
        float lo = v[i], hi = v[i + h];
        asm volatile("" : "+v"(lo), "+v"(hi));
        const float keep = up ? hi : lo;
        const float send = up ? lo : hi;
        v[i] = keep + __shfl_xor(send, h);
      }
    }
    const float tot = v[0] + __shfl_xor(v[0], 32);
    if ((lane >> 5) == half) out = tot;
  }
  return out;
}

// sum over the rows of the workgroup's 64 x 4 tile: every thread gets the sum of its column
template <int NW = 4>
__device__ __forceinline__ float ss_colsum(float v, float* sh, int wave, int lane) {
  v += __shfl_xor(v, 4);
  v += __shfl_xor(v, 8);
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  __syncthreads();
  if (lane < 4) sh[wave * 4 + lane] = v;
  __syncthreads();
  const int c = lane & 3;
  float t = (sh[c] + sh[4 + c]) + (sh[8 + c] + sh[12 + c]);
  if (NW == 8) t += (sh[16 + c] + sh[20 + c]) + (sh[24 + c] + sh[28 + c]);
  return t;
}

__device__ __forceinline__ bool ss_keep(const DropoutSrc& d, int layer_index, int64_t layer_elems, int row, int col,
                                        int W) {
  if (d.keep) return d.keep[(int64_t)layer_index * layer_elems + (int64_t)row * W + col] != 0;
  const Philox128 p = dropout_patch(d.seed, dropout_step(d), d.layer + layer_index,
                                    (int64_t)(row & ~31) + d.row_offset, col);
  const int w = (row & 31) >> 3;
  const uint32_t word = w == 0 ? p.w[0] : (w == 1 ? p.w[1] : (w == 2 ? p.w[2] : p.w[3]));
  return ((word >> (4 * (row & 7) + (col & 3))) & 1u) != 0;
}

// dW rows of the workgroup: out[c][k] = sum_b dz[b][c] * In[b][k]; dz: LDS [64] float4 (4 columns of a row)
template <bool PUBLISH = false, int NT = SS_THREADS>
__device__ __forceinline__ double ss_wgrad(const float4* __restrict__ sh_dz, const float* __restrict__ In, int K,
                                           int batch, float* __restrict__ out) {
  double sq = 0.0;
  for (int kq = threadIdx.x; kq * 4 < K; kq += NT) {
    float4 acc[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = float4{0.f, 0.f, 0.f, 0.f};
    for (int b0 = 0; b0 < batch; b0 += 16) {      // (32 rows per batch: the allocator spills, 213 us instead of 200)
      float4 a[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) a[u] = ss_ld4(In + (int64_t)min(b0 + u, batch - 1) * K + kq * 4);
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const float4 dz = (b0 + u < batch) ? sh_dz[b0 + u] : float4{0.f, 0.f, 0.f, 0.f};
        acc[0].x = fmaf(dz.x, a[u].x, acc[0].x); acc[0].y = fmaf(dz.x, a[u].y, acc[0].y);
        acc[0].z = fmaf(dz.x, a[u].z, acc[0].z); acc[0].w = fmaf(dz.x, a[u].w, acc[0].w);
        acc[1].x = fmaf(dz.y, a[u].x, acc[1].x); acc[1].y = fmaf(dz.y, a[u].y, acc[1].y);
        acc[1].z = fmaf(dz.y, a[u].z, acc[1].z); acc[1].w = fmaf(dz.y, a[u].w, acc[1].w);
        acc[2].x = fmaf(dz.z, a[u].x, acc[2].x); acc[2].y = fmaf(dz.z, a[u].y, acc[2].y);
        acc[2].z = fmaf(dz.z, a[u].z, acc[2].z); acc[2].w = fmaf(dz.z, a[u].w, acc[2].w);
        acc[3].x = fmaf(dz.w, a[u].x, acc[3].x); acc[3].y = fmaf(dz.w, a[u].y, acc[3].y);
        acc[3].z = fmaf(dz.w, a[u].z, acc[3].z); acc[3].w = fmaf(dz.w, a[u].w, acc[3].w);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (PUBLISH) ss_publish4(out + (int64_t)c * K + kq * 4, acc[c]);
      else *reinterpret_cast<float4*>(out + (int64_t)c * K + kq * 4) = acc[c];
      sq += (double)acc[c].x * acc[c].x + (double)acc[c].y * acc[c].y + (double)acc[c].z * acc[c].z +
            (double)acc[c].w * acc[c].w;
    }
  }
  return sq;
}

template <int NT = SS_THREADS>
__device__ __forceinline__ double ss_block_sum(double v, double* sh) {
  __syncthreads();
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int o = NT / 2; o >= 1; o >>= 1) {
    if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
    __syncthreads();
  }
  return sh[0];
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------
// The same stage bodies, ONE LAUNCH PER STAGE (no grid barrier, no residency requirement): forward stages,
// decode (+ MSE), backward stages, then the library's clip + Adam kernel — 2 nh + 2 launches instead of ~52.
// A kernel boundary costs the next kernel's launch latency and a cold L2, about what a grid barrier's poll +
// invalidate costs (5.5 us) — but back-to-back launches overlap their launch latencies, the hardware's cache
// maintenance is cheaper than 256 workgroups' fences, and everything is stored with ordinary (write-back) stores.
// What backward needs of forward crosses in the workspace exactly as for SS_FWD / SS_BWD (x-hat in Z, the gate in dZ,
// gamma * invstd in the saved-statistics rows); the skip gradient of an even stage waits for stage i - 2 in G0 / G1.
struct SsIdx {
  int tid, lane, wave, g, n0, row, c, col;
};
__device__ __forceinline__ SsIdx ss_idx(int W) {
  SsIdx x;
  x.tid = threadIdx.x; x.lane = x.tid & 63; x.wave = x.tid >> 6; x.g = blockIdx.x;
  const int ngroups = W >> 2;
  x.n0 = ((x.g & 7) * (ngroups >> 3) + (x.g >> 3)) * 4;       // XCD-aware, as in the persistent kernel
  x.row = x.tid >> 2; x.c = x.tid & 3; x.col = x.n0 + x.c;    // (row within a row block)
  return x;
}

// Workgroups of NW waves walk RB row blocks of 16 NW rows (wave w takes rows 16 w .. 16 w + 15 of a block): NW = 4,
// RB = 1 serves up to 64 rows, NW = 8 up to 128 RB rows.  A thread holds element (row block r, row tid / 4, column
// tid % 4) for every r; the column sums run over all blocks (the per-thread sum over r goes into ONE ss_colsum).
// The GEMM core costs one pass over the whole activation per row block, so the form stops paying at ~400 rows.

// grid: W / 4 workgroups
template <int NW, int RB>
__global__ __launch_bounds__(64 * NW) void small_fwd_stage_kernel(const SmallStepParams p, const int i) {
  constexpr int RPB = 16 * NW;
  __shared__ float sh_cs[4 * NW];
  const int W = p.W, B = p.batch;
  const SsIdx x = ss_idx(W);
  const int K = (i == 0) ? p.in_f : W;
  const float* in = (i == 0) ? p.x : p.A[i - 1];
  const float inv_b = 1.0f / (float)B;
  WBlock wb;
  ss_load_w_rows(wb, p.params + p.w_off[i] + (int64_t)x.n0 * K, K, x.lane);
  const float bias = p.params[p.b_off[i] + x.col], gamma = p.params[p.g_off[i] + x.col], beta = p.params[p.be_off[i] + x.col];
  float* rm = p.bn_running + ((int64_t)i * 2 + 0) * W;
  float* rv = p.bn_running + ((int64_t)i * 2 + 1) * W;
  const float rm0 = rm[x.col], rv0 = rv[x.col];
  const int64_t nbt0 = p.nbt[i];
  const bool has_skip = i >= 2 && (i & 1) == 0;
  float z[RB], skipv[RB];
  bool valid[RB], kept[RB];
  float zs = 0.f;
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int row = r * RPB + x.row;
    valid[r] = row < B;
    skipv[r] = (has_skip && valid[r]) ? p.A[i - 2][(int64_t)row * W + x.col] : 0.f;
    kept[r] = valid[r] && ss_keep(p.drop, i, (int64_t)B * W, row, x.col, W);
    z[r] = ss_gemm<false>(wb, in, K, K, B, r * NW + x.wave, x.lane) + bias;
    if (!valid[r]) z[r] = 0.f;
    zs += z[r];
  }
  const float mean = ss_colsum<NW>(zs, sh_cs, x.wave, x.lane) * inv_b;
  float dlt[RB], d2 = 0.f;
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    dlt[r] = valid[r] ? z[r] - mean : 0.f;
    d2 = fmaf(dlt[r], dlt[r], d2);
  }
  const float m2 = ss_colsum<NW>(d2, sh_cs, x.wave, x.lane);
  const float invstd = (float)(1.0 / sqrt((double)m2 / (double)B + (double)1e-5f));
  const float sc = gamma * invstd;
  const float sh = beta - mean * sc;
  if (x.row == 0) {
    const double f = (p.momentum >= 0.f) ? (double)p.momentum : 1.0 / (double)(nbt0 + 1);
    const double unbiased = (double)m2 / (double)(B > 1 ? B - 1 : 1);
    rm[x.col] = (float)((1.0 - f) * (double)rm0 + f * (double)mean);
    rv[x.col] = (float)((1.0 - f) * (double)rv0 + f * unbiased);
    float* st = p.bn_saved[i];
    st[x.col] = mean; st[W + x.col] = invstd; st[2 * W + x.col] = sc; st[3 * W + x.col] = sh;
  }
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const float y = fmaf(z[r], sc, sh);
    const bool on = kept[r] && y > 0.f;
    if (valid[r]) {
      const int64_t e = (int64_t)(r * RPB + x.row) * W + x.col;
      p.A[i][e] = (on ? y * 2.f : 0.f) + skipv[r];
      p.Z[i][e] = dlt[r] * invstd;
      p.dZ[i][e] = on ? 2.f : 0.f;
    }
  }
}

// Eval mode (/root/reference/valid_bilinear.py:31,52): Linear + BatchNorm with the running statistics + ReLU (+ skip),
// no dropout, nothing saved.  Same operations in the same order as the eval epilogue of the big-batch path
// (gemm_epilogue.h, EPI_BN_RELU): sc = gamma / sqrt(var + eps), sh = beta - mean * sc, a = max(fma(z, sc, sh), 0).
template <int NW, int RB>
__global__ __launch_bounds__(64 * NW) void small_eval_stage_kernel(const SmallStepParams p, const int i) {
  constexpr int RPB = 16 * NW;
  const int W = p.W, B = p.batch;
  const SsIdx x = ss_idx(W);
  const int K = (i == 0) ? p.in_f : W;
  const float* in = (i == 0) ? p.x : p.A[i - 1];
  WBlock wb;
  ss_load_w_rows(wb, p.params + p.w_off[i] + (int64_t)x.n0 * K, K, x.lane);
  const float bias = p.params[p.b_off[i] + x.col], gamma = p.params[p.g_off[i] + x.col], beta = p.params[p.be_off[i] + x.col];
  const float rm = p.bn_running[((int64_t)i * 2 + 0) * W + x.col], rv = p.bn_running[((int64_t)i * 2 + 1) * W + x.col];
  const float sc = gamma * (1.0f / sqrtf(rv + 1e-5f));
  const float sh = beta - rm * sc;
  const bool has_skip = i >= 2 && (i & 1) == 0;
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int row = r * RPB + x.row;
    const bool valid = row < B;
    const int64_t e = (int64_t)row * W + x.col;
    const float skipv = (has_skip && valid) ? p.A[i - 2][e] : 0.f;
    const float z = ss_gemm<false>(wb, in, K, K, B, r * NW + x.wave, x.lane) + bias;
    if (valid) p.A[i][e] = fmaxf(fmaf(z, sc, sh), 0.f) + skipv;
  }
}

// grid: out_f / 4 workgroups.  MSE: + loss partials, d loss / d prediction, the decode bias gradient and its norm
// partial (slots [W / 4, W / 4 + out_f / 4) of sumsq_part); either way the BatchNorm counters (every forward stage has
// read them: the stage kernels are complete; eval: nbt == nullptr).  The decode WEIGHT gradient is formed by the top
// backward stage kernel, four columns per workgroup, from the activations its threads own: here 12 workgroups would
// each read the whole activation for it.
template <bool MSE, int NW, int RB>
__global__ __launch_bounds__(64 * NW) void small_decode_kernel(const SmallStepParams p) {
  constexpr int RPB = 16 * NW;
  __shared__ float sh_cs[4 * NW];
  __shared__ double sh_d[64 * NW];
  const int W = p.W, B = p.batch, OF = p.out_f, nh = p.nh;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = blockIdx.x;
  const int o0 = g * 4, c = tid & 3, oc = o0 + c;
  WBlock wb;
  ss_load_w_rows(wb, p.params + p.dec_w + (int64_t)o0 * W, W, lane);
  const float bd = p.params[p.dec_b + oc];
  float d2 = 0.f, dps = 0.f;
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int row = r * RPB + (tid >> 2);
    const bool valid = row < B;
    const float tg = (MSE && valid) ? p.target[(int64_t)row * OF + oc] : 0.f;
    const float pr = ss_gemm<false>(wb, p.A[nh - 1], W, W, B, r * NW + wave, lane) + bd;
    if (valid) p.pred[(int64_t)row * OF + oc] = pr;
    if (MSE) {
      const float diff = valid ? pr - tg : 0.f;
      const float dp = diff * p.mse_scale;
      if (valid) p.dpred[(int64_t)row * OF + oc] = dp;
      d2 = fmaf(diff, diff, d2);
      dps += dp;
    }
  }
  if (p.nbt && g == 0 && tid == 0) for (int i = 0; i < nh; ++i) p.nbt[i] += 1;
  if (!MSE) return;
  (void)ss_colsum<NW>(d2, sh_cs, wave, lane);
  if (tid == 0) {       // the per-wave column sums -> one partial per workgroup (fixed order)
    float l = 0.f;
    for (int k = 0; k < 4 * NW; ++k) l += sh_cs[k];
    p.loss_part[g] = l;
  }
  double sq = 0.0;
  const float db = ss_colsum<NW>(dps, sh_cs, wave, lane);
  if ((tid >> 2) == 0) { p.grads[p.dec_b + oc] = db; sq += (double)db * db; }
  const double wg_sq = ss_block_sum<64 * NW>(sq, sh_d);
  if (tid == 0) p.sumsq_part[(W >> 2) + g] = wg_sq;
}

// grid: W / 4 workgroups.  dec_here (top stage of the drop-in backward): d loss / d prediction comes from the
// caller and the decode bias gradient is formed here too (workgroup 0).  wgrad_here = 0: the weight gradients of
// the hidden stages come from ONE batched MFMA GEMM launch behind the stage kernels (the host's launch_gemm over
// the consecutive dZ / A buffers: 64 tiles of 128 x 128 per stage read 4 MB where the four-column ownership of this
// kernel reads the whole activation in every workgroup — 8 us per stage).
template <int NW, int RB>
__global__ __launch_bounds__(64 * NW) void small_bwd_stage_kernel(const SmallStepParams p, const int i,
                                                                   const int dec_here, const int wgrad_here) {
  constexpr int NT = 64 * NW, RPB = 16 * NW;
  __shared__ float sh_cs[4 * NW];
  __shared__ float4 sh_dz[RPB * RB];                           // dz of every row (the in-kernel weight gradient)
  __shared__ double sh_d[NT];
  __shared__ __align__(16) float sh_big[RPB * 64];             // one row block of dpred [.][out_f] or of x [.][in_f]
  const int W = p.W, B = p.batch, OF = p.out_f, nh = p.nh;
  const SsIdx x = ss_idx(W);
  const bool top = (i == nh - 1);
  const float inv_b = 1.0f / (float)B;
  double sq = 0.0;
  WBlock wb;
  if (top) ss_load_w_cols(wb, p.params + p.dec_w + x.n0, OF, W, x.lane);
  else ss_load_w_cols(wb, p.params + p.w_off[i + 1] + x.n0, W, W, x.lane);
  // what forward left in the workspace, and the skip gradient of stage i + 2
  const bool even = (i & 1) == 0;
  const float sc = p.bn_saved[i][2 * W + x.col];
  float xhat[RB], gate[RB], gin[RB];
  bool valid[RB];
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    const int row = r * RPB + x.row;
    valid[r] = row < B;
    const int64_t e = (int64_t)row * W + x.col;
    xhat[r] = valid[r] ? p.Z[i][e] : 0.f;
    gate[r] = valid[r] ? p.dZ[i][e] : 0.f;
    gin[r] = (even && !top && valid[r]) ? p.gskip[((i >> 1) + 1) & 1][e] : 0.f;
  }
  if (top) {
    // decode weight gradient, the four columns of this workgroup: dWd[o][col] = sum_b dpred[b][o] * A[b][col], one
    // row block at a time through LDS (dec_here, workgroup 0: + the decode bias gradient — the drop-in backward has
    // no decode kernel before it)
    float acc = 0.f, db = 0.f;
    const int o = x.tid >> 2, cc = x.tid & 3;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
      const int r0 = r * RPB, nrow = min(RPB, B - r0);
      if (nrow > 0) {                                            // (uniform)
        for (int k = x.tid; k < nrow * OF; k += NT) sh_big[k] = p.dpred[(int64_t)r0 * OF + k];
        reinterpret_cast<float*>(sh_dz)[x.tid] = valid[r] ? p.A[nh - 1][(int64_t)(r0 + x.row) * W + x.col] : 0.f;
        __syncthreads();
        if (x.tid < OF * 4)
          for (int b = 0; b < nrow; ++b) acc = fmaf(sh_big[b * OF + o], reinterpret_cast<const float*>(sh_dz)[b * 4 + cc], acc);
        if (dec_here && x.g == 0 && x.tid < OF)
          for (int b = 0; b < nrow; ++b) db += sh_big[b * OF + x.tid];
        __syncthreads();
      }
    }
    if (x.tid < OF * 4) {
      p.grads[p.dec_w + (int64_t)o * W + x.n0 + cc] = acc;
      sq += (double)acc * acc;
    }
    if (dec_here && x.g == 0 && x.tid < OF) {
      p.grads[p.dec_b + x.tid] = db;
      sq += (double)db * db;
    }
  }
  float ga[RB], dy[RB], dz[RB];
  float sb = 0.f, sg = 0.f;
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    ga[r] = top ? ss_gemm<true>(wb, p.dpred, OF, OF, B, r * NW + x.wave, x.lane)
                : ss_gemm<true>(wb, p.dZ[i + 1], W, W, B, r * NW + x.wave, x.lane);
    if (even) {
      ga[r] += gin[r];
      if (i >= 2 && valid[r]) p.gskip[(i >> 1) & 1][(int64_t)(r * RPB + x.row) * W + x.col] = ga[r];
    }
    dy[r] = valid[r] ? ga[r] * gate[r] : 0.f;
    sb += dy[r];
    sg = fmaf(dy[r], xhat[r], sg);
  }
  const float s_b = ss_colsum<NW>(sb, sh_cs, x.wave, x.lane);
  const float s_g = ss_colsum<NW>(sg, sh_cs, x.wave, x.lane);
  float dzs = 0.f;
#pragma unroll
  for (int r = 0; r < RB; ++r) {
    dz[r] = valid[r] ? sc * (dy[r] - (s_b + xhat[r] * s_g) * inv_b) : 0.f;
    dzs += dz[r];
    if (i >= 1 && valid[r]) p.dZ[i][(int64_t)(r * RPB + x.row) * W + x.col] = dz[r];
  }
  const float dbias = ss_colsum<NW>(dzs, sh_cs, x.wave, x.lane);
  if (x.row == 0) {
    p.grads[p.g_off[i] + x.col] = s_g;
    p.grads[p.be_off[i] + x.col] = s_b;
    p.grads[p.b_off[i] + x.col] = dbias;
    sq += (double)s_g * s_g + (double)s_b * s_b + (double)dbias * dbias;
  }
  if (i == 0 || wgrad_here) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < RB; ++r) reinterpret_cast<float*>(sh_dz)[r * NT + x.tid] = dz[r];      // [row][c], all rows
    __syncthreads();
    const int K = (i == 0) ? p.in_f : W;
    if (i == 0 && K <= SS_SMALLK) {
      // dW0[col][k] = sum_b dz[b][c] * x[b][k]: one (c, k) pair per thread, x one row block at a time through LDS
      float acc[(4 * SS_SMALLK + NT - 1) / NT];
#pragma unroll
      for (int u = 0; u < (4 * SS_SMALLK + NT - 1) / NT; ++u) acc[u] = 0.f;
#pragma unroll
      for (int r = 0; r < RB; ++r) {
        const int r0 = r * RPB, nrow = min(RPB, B - r0);
        if (nrow > 0) {
          for (int k = x.tid * 4; k < nrow * K; k += NT * 4)
            *reinterpret_cast<float4*>(sh_big + k) = ss_ld4(p.x + (int64_t)r0 * K + k);
          __syncthreads();
#pragma unroll
          for (int u = 0; u < (4 * SS_SMALLK + NT - 1) / NT; ++u) {
            const int t = u * NT + x.tid;
            if (t < 4 * K) {
              const int cc = t / K, k = t - cc * K;
              float a = acc[u];
              for (int b = 0; b < nrow; ++b)
                a = fmaf(reinterpret_cast<const float*>(sh_dz)[(r0 + b) * 4 + cc], sh_big[b * K + k], a);
              acc[u] = a;
            }
          }
          __syncthreads();
        }
      }
#pragma unroll
      for (int u = 0; u < (4 * SS_SMALLK + NT - 1) / NT; ++u) {
        const int t = u * NT + x.tid;
        if (t < 4 * K) {
          const int cc = t / K, k = t - cc * K;
          p.grads[p.w_off[0] + (int64_t)(x.n0 + cc) * K + k] = acc[u];
          sq += (double)acc[u] * acc[u];
        }
      }
    } else {
      sq += ss_wgrad<false, NT>(sh_dz, (i == 0) ? p.x : p.A[i - 1], K, B, p.grads + p.w_off[i] + (int64_t)x.n0 * K);
    }
  }
  // one norm partial per workgroup, accumulated over the stage launches in a fixed order (deterministic)
  const double wg_sq = ss_block_sum<NT>(sq, sh_d);
  if (x.tid == 0) p.sumsq_part[x.g] = (top ? 0.0 : p.sumsq_part[x.g]) + wg_sq;
}

bool small_staged_shape_ok(const SmallStepParams& p) {
  return p.batch <= SS_STAGED_MAX_ROWS && p.nh <= SS_MAX_STAGES && p.W <= 256 * SS_PASSES && p.W % 32 == 0 &&
         p.in_f <= 256 * SS_PASSES && p.in_f % 4 == 0 && p.out_f <= 64 && p.out_f % 4 == 0;
}

// (up to 64 rows: 4 waves per workgroup, one row block; above: 8 waves and ceil(batch / 128) row blocks)
#define SS_DISPATCH(FN, ...)                                   \
  do {                                                         \
    if (p.batch <= 64) FN<4, 1>(__VA_ARGS__);                  \
    else if (p.batch <= 128) FN<8, 1>(__VA_ARGS__);            \
    else if (p.batch <= 256) FN<8, 2>(__VA_ARGS__);            \
    else FN<8, 3>(__VA_ARGS__);                                \
  } while (0)

template <int NW, int RB>
static void small_forward_staged(hipStream_t s, const SmallStepParams& p, bool mse) {
  const dim3 grid((unsigned)(p.W / 4)), block(64 * NW), dgrid((unsigned)(p.out_f / 4));
  for (int i = 0; i < p.nh; ++i) hipLaunchKernelGGL((small_fwd_stage_kernel<NW, RB>), grid, block, 0, s, p, i);
  if (mse) hipLaunchKernelGGL((small_decode_kernel<true, NW, RB>), dgrid, block, 0, s, p);
  else hipLaunchKernelGGL((small_decode_kernel<false, NW, RB>), dgrid, block, 0, s, p);
}
int launch_small_forward_staged(hipStream_t s, const SmallStepParams& p, bool mse) {
  if (!small_staged_shape_ok(p)) return BLH_ERR_SHAPE;
  SS_DISPATCH(small_forward_staged, s, p, mse);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

template <int NW, int RB>
static void small_eval_staged(hipStream_t s, const SmallStepParams& p) {
  const dim3 grid((unsigned)(p.W / 4)), block(64 * NW);
  for (int i = 0; i < p.nh; ++i) hipLaunchKernelGGL((small_eval_stage_kernel<NW, RB>), grid, block, 0, s, p, i);
  hipLaunchKernelGGL((small_decode_kernel<false, NW, RB>), dim3((unsigned)(p.out_f / 4)), block, 0, s, p);
}
int launch_small_eval_staged(hipStream_t s, const SmallStepParams& p) {
  if (!small_staged_shape_ok(p) || p.nbt != nullptr) return BLH_ERR_INVALID_ARGUMENT;
  SS_DISPATCH(small_eval_staged, s, p);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

template <int NW, int RB>
static void small_backward_staged(hipStream_t s, const SmallStepParams& p, bool dec_here, bool wgrad_here) {
  const dim3 grid((unsigned)(p.W / 4)), block(64 * NW);
  for (int i = p.nh - 1; i >= 0; --i)
    hipLaunchKernelGGL((small_bwd_stage_kernel<NW, RB>), grid, block, 0, s, p, i, dec_here ? 1 : 0, wgrad_here ? 1 : 0);
}
int launch_small_backward_staged(hipStream_t s, const SmallStepParams& p, bool dec_here, bool wgrad_here) {
  if (!small_staged_shape_ok(p)) return BLH_ERR_SHAPE;
  SS_DISPATCH(small_backward_staged, s, p, dec_here, wgrad_here);
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}
#undef SS_DISPATCH

}  // namespace blh
