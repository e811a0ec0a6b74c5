// Shared declarations of the native library (host side).
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

#include "../../include/bilinear_hip.h"
#include "host_util.h"

namespace blh {

extern thread_local int g_last_hip_error;

#define BLH_HIP_TRY(expr)                          \
  do {                                             \
    hipError_t _e = (expr);                        \
    if (_e != hipSuccess) {                        \
      ::blh::g_last_hip_error = (int)_e;           \
      return BLH_ERR_HIP;                          \
    }                                              \
  } while (0)

#define BLH_TRY(expr)            \
  do {                           \
    int _s = (expr);             \
    if (_s != BLH_OK) return _s; \
  } while (0)

// Fork without a marker packet.  An event a side stream has to wait for is normally recorded
// with hipEventRecord BEHIND the producing kernel: an extra barrier packet in the queue, and the
// next kernel of the producing stream starts ~7 us late (measured: gap in front of every dgrad of
// the two-stream backward, profiles/r02_step_timeline.md).  Arming `tl_stop_event` makes the NEXT
// kernel launched by this thread through launch_kernel() carry the event as its own completion
// signal (hipExtLaunchKernelGGL stop event): no extra packet.  Set and consumed inside one call.
extern thread_local hipEvent_t tl_stop_event;

template <typename K, typename... Args>
static inline void launch_kernel(K kern, dim3 grid, dim3 block, size_t lds, hipStream_t s,
                                 Args... args) {
  hipEvent_t ev = tl_stop_event;
  tl_stop_event = nullptr;
  if (ev) hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)lds, s, nullptr, ev, 0, args...);
  else hipLaunchKernelGGL(kern, grid, block, lds, s, args...);
}


// ---------------------------------------------------------------- GEMM -----
// fp32 backward GEMMs: does a launch of `wgs` 128x128 workgroups take the one-workgroup-per-CU
// (128 KB LDS) form?  (gemm_f32.hip; the backward scheduler forks early when it does)
bool gemm_f32_backward_exclusive(int64_t wgs, int K, int k_per_launch_slice);
enum Layout : int { ROWK = 0, KROW = 1 };
enum Epilogue : int {
  EPI_STORE = 0,       // C = acc                         (split-K slabs, plain)
  EPI_BIAS = 1,        // C = acc + bias[n]
  EPI_BIAS_STATS = 2,  // C = acc + bias[n]; per-tile column (mean, M2) partials
  EPI_ADD = 3,         // C = acc + addend[m][n]
  EPI_BN_RELU = 4,     // eval forward: C = relu(bn_eval(acc + bias[n])) (+ addend[m][n], the block skip)
  EPI_BN_BWD = 5,      // bf16 storage: data gradient feeding a BatchNorm backward (gemm_bf16s_256.h): C = the gated
                       // gradient dY' = 2 keep [y > 0] acc, per-row-tile column sums of dY' z and dY'
  EPI_BN_BWD_ADD = 6,  // the same with acc + addend[m][n] (the block-skip gradient)
  // (7 was EPI_BN_FWD, the bf16 forward stage in one launch behind a grid barrier: measured slower and removed in
  //  round 6, profiles/r04_fused_forward.md)
  EPI_STORE_SQ = 8     // C = acc, and one fp64 sum of squares of the stored tile per workgroup (sq_part[z * tiles + x]):
                       // the gradient-norm partials of a weight gradient without another pass over it
};

struct GemmParams {
  const float* A;
  const float* B;
  float* C;
  int64_t lda, ldb, ldc;
  int M, N, K;
  int k_per_split;          // multiple of 32 (or == K when splits == 1)
  int64_t c_split_stride;   // floats between slabs
  const float* bias;        // [N]
  const float* addend;      // [M][ldadd]
  int64_t ldadd;
  float* stat_part;         // [tiles_m][2][N]  (mean, M2) of each BM-row tile
  float* loss_part;         // diagnostic builds only: STAMP output (tools/gemm_bench)
  // EPI_BN_RELU: BatchNorm1d in eval mode (running statistics), [N] each
  const float* bn_gamma; const float* bn_beta; const float* bn_mean; const float* bn_var;
  // gemm_dtype 3 (fp16 two-piece split): max |value| partials of each operand tensor
  const float* a_amax; int a_namax;
  const float* b_amax; int b_namax;
  double* sq_part;          // EPI_STORE_SQ: [splits][tiles]
};

enum GemmTile : int { TILE_128x128 = 0, TILE_128x64 = 1, TILE_64x128 = 2, TILE_128x32 = 3 };

// number of BM-row tiles the EPI_BIAS_STATS epilogue produces partials for
int gemm_stat_tile_rows(GemmTile tile);
int gemm_grid_blocks(GemmTile tile, int M, int N);
// dtype 0: exact fp32 MFMA; 1: operands rounded to bf16 on load, bf16 MFMA, fp32 accumulate;
// 2: bf16x3 split; 3: fp16x2 split (needs p.a_amax / p.b_amax, else falls back to dtype 2)
int launch_gemm(hipStream_t s, GemmTile tile, int la, int lb, int epi, const GemmParams& p,
                int splits, int dtype = 0);

// ---------------------------------------------------------- elementwise ----
struct DropoutSrc {
  const uint8_t* keep;   // [B][W] for this layer, or nullptr -> Philox
  uint64_t seed, step;
  int64_t row_offset;
  int layer;
  // hipGraph replay: the dropout step lives in device memory (blh_step_state.rng_step) and is
  // added to `step`, so that a captured launch draws a fresh mask on every replay
  const uint64_t* step_dev;
};
__host__ __device__ inline uint64_t dropout_step(const DropoutSrc& d) {
#if defined(__HIP_DEVICE_COMPILE__)
  return d.step + (d.step_dev ? *d.step_dev : 0ull);
#else
  return d.step;
#endif
}

// (row chunking of the streaming kernels: host_util.h)

// forward BN: merge per-tile (mean, M2) -> batch mean / invstd, scale/shift, running stats
int launch_bn_fwd_finalize(hipStream_t s, const float* stat_part, int tiles, int tile_rows,
                           int64_t batch, int W, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, int64_t* nbt,
                           float momentum, float* saved_mean, float* saved_invstd,
                           float* scale, float* shift);
// gemm_dtype 3: the streaming BatchNorm kernels can emit max |value| partials of what they write
// (one per wave: ew_num_amax_parts() floats), from which the fp16-split GEMM picks its scale
int launch_wamax(hipStream_t s, const float* W, int64_t w_stride, int layers, int64_t count,
                 float* part);
// bf16-storage path, second generation (bn_bf16.hip): the forward writes one keep bit per element
// (keepbits: [ceil(B/4)][W/8] words, bn_keepbits_words), the backward kernels read them
int64_t bn_keepbits_words(int64_t batch, int W);
int launch_bn_apply_h2(hipStream_t s, bool train, const uint16_t* Z, const float* scale, const float* shift,
                       const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, const uint16_t* skip, uint16_t* A, uint32_t* keepbits,
                       int64_t batch, int W, const DropoutSrc& drop, int64_t* nbt);
// part [chunks][2][W]: sum dY z, sum dY (chunks = ew_num_row_chunks_h(batch))
int launch_bn_bwd_reduce_h2(hipStream_t s, const uint16_t* dA, const uint16_t* Z, const float* scale,
                            const float* shift, const uint32_t* keepbits, float* part, int64_t batch, int W);
// sq (optional): one sum-of-squares partial of (dgamma, dbeta) per block, bn_bwd_finalize_blocks(W) of them
int launch_bn_bwd_finalize_h2(hipStream_t s, const float* part, int chunks, int W, const float* mean,
                              const float* invstd, float* dgamma, float* dbeta, double* sq = nullptr);
int bn_bwd_finalize_blocks(int W);
int launch_bn_bwd_apply_h2(hipStream_t s, const uint16_t* dA, const uint16_t* Z, const float* scale,
                           const float* shift, const float* mean, const float* invstd, const float* dgamma,
                           const float* dbeta, const uint32_t* keepbits, uint16_t* dZ, float* colsum_part,
                           int64_t batch, int W, int64_t norm_batch, bool pregated = false);
// fp32-storage path, second generation (bn_f32.hip): keepbits [ceil(B/8)][W/4] words
int64_t bn_keepbits_words_f32(int64_t batch, int W);
int launch_bn_apply_f2(hipStream_t s, bool train, const float* Z, const float* scale, const float* shift,
                       const float* gamma, const float* beta, const float* running_mean,
                       const float* running_var, const float* skip, float* A, uint32_t* keepbits,
                       int64_t batch, int W, const DropoutSrc& drop, int64_t* nbt, float* amax_part = nullptr);
// part [chunks][2][W]: sum dY z, sum dY (chunks = ew_num_row_chunks(batch)); finalize: launch_bn_bwd_finalize_h2
int launch_bn_bwd_reduce_f2(hipStream_t s, const float* dA, const float* Z, const float* scale,
                            const float* shift, const uint32_t* keepbits, float* part, int64_t batch, int W);
int launch_bn_bwd_apply_f2(hipStream_t s, const float* dA, const float* Z, const float* scale,
                           const float* shift, const float* mean, const float* invstd, const float* dgamma,
                           const float* dbeta, const uint32_t* keepbits, float* dZ, float* colsum_part,
                           int64_t batch, int W, int64_t norm_batch, float* amax_part = nullptr);
// SyncBN: local fp64 column sums [sum z | sum z^2] -> (host all-reduce) -> finalize
int launch_bn_fwd_local_sums(hipStream_t s, const float* stat_part, int tiles, int tile_rows,
                             int64_t batch, int W, double* sums);
int launch_bn_fwd_finalize_sums(hipStream_t s, const double* sums, int64_t n_global, int W,
                                const float* gamma, const float* beta, float* running_mean,
                                float* running_var, int64_t* nbt, float momentum,
                                float* saved_mean, float* saved_invstd, float* scale,
                                float* shift);
// out[c] = sum_s in[s][c] for c < ncols (rows of `ld` floats), fp64 accumulation
int launch_colreduce(hipStream_t s, const float* in, int S, int64_t ld, int ncols, float* out);
// Linear-bias gradients of all stages in one launch (partials [stage][S][W])
// sq (optional): one sum-of-squares partial of the written gradients per block, bias_colreduce_blocks() of them
int launch_bias_colreduce(hipStream_t s, const float* part, int64_t stage_stride, int S, int W,
                          int num_stages, const int64_t* out_offsets, float* grads,
                          const float* extra_part = nullptr, int extra_S = 0, int extra_cols = 0,
                          int64_t extra_off = 0, double* sq = nullptr);
int bias_colreduce_blocks(int W, int num_stages, bool extra);
// out[i] = sum_s in[s][i]   (slabs of `count` floats)
int launch_sum_slabs(hipStream_t s, const float* slabs, int64_t count, int splits, float* out);
// the same with one sum-of-squares partial of `out` per block: sq[0 .. sum_slabs_sq_blocks(count, max_blocks))
int launch_sum_slabs_sq(hipStream_t s, const float* slabs, int64_t count, int splits, float* out, double* sq,
                        int max_blocks);
int sum_slabs_sq_blocks(int64_t count, int max_blocks);
// the batched slab sum that also leaves the gradient norm's partials (elementwise.hip): `ranges` = every part of the
// gradient arena the items do NOT cover; *nparts partials in sq (at most max_parts, else BLH_ERR_SHAPE)
struct SqRanges { int n; int64_t total4; int64_t off[36]; int64_t cnt[36]; };   // total4: sum of cnt / 4 (set by the launcher)
int launch_sum_slabs_batched_sq(hipStream_t s, const float* slabs, int64_t count, int splits, int items,
                                int64_t slab_item_stride, float* out, int64_t out_item_stride, const float* arena,
                                const SqRanges& ranges, double* sq, int max_parts, int* nparts);
int launch_sum_slabs_batched(hipStream_t s, const float* slabs, int64_t count, int splits, int items,
                             int64_t slab_item_stride, float* out, int64_t out_item_stride);
int launch_sum_slabs_add(hipStream_t s, const float* slabs, int64_t count, int splits,
                         const float* addend, float* out);
// small-batch forward: Z = sum(slabs) + bias; stat_part (may be null) receives the column (mean, M2) of every
// fwd_finish_stat_rows()-row chunk: [ceil(M / rows)][2][N] (the tile form bn_fwd_finalize merges)
int fwd_finish_stat_rows();
int launch_fwd_finish(hipStream_t s, const float* slabs, int splits, int64_t M, int N,
                      const float* bias, float* Z, float* stat_part);
// out[c] = sum over rows of X[rows][ld] columns [0,cols)
int launch_colsum(hipStream_t s, const float* X, int64_t rows, int cols, int64_t ld, float* part,
                  float* out);
// loss = sum(part[0..n)) / denom
int launch_loss_finalize(hipStream_t s, const float* part, int n, double denom, float* loss_out);
// dpred = scale*(pred-target), loss partials (stand-alone MSE for the autograd path)
int launch_mse(hipStream_t s, const float* pred, const float* target, int64_t n, float scale,
               float* dpred, float* part, int* nparts);
// optimiser
// loss = sum(part[0..n)) / denom, finished by block 0 of the optimiser kernel (part == nullptr: off)
struct LossFinish { const float* part; int n; double denom; float* out; };
// scalars of one Adam step (torch.optim.Adam forms them in double and rounds each once: adam_consts)
struct AdamConsts { float one_minus_b1, b2, one_minus_b2, step_size, bc2_sqrt, eps, max_norm; };
AdamConsts adam_consts(const blh_adam_hyper& h);
int launch_sumsq(hipStream_t s, const float* g, int64_t count, double* part, int* nparts);
// shadow (optional): bf16 images of the updated parameters for the next bf16-storage forward — `plain`: the arena
// element for element (the GEMMs' weights); `wdT` (optional): the decode weight [OF][W] at arena offset dec_w in the
// one-pass decode's K-major layout (decode_wdT_dev.h; rows >= OF of that image are zero and never written here)
struct ShadowDst { uint16_t* plain; uint16_t* wdT; int64_t dec_w; int W, OF; };
static constexpr ShadowDst NO_SHADOW = {nullptr, nullptr, 0, 0, 0};
int launch_clip_adam(hipStream_t s, float* p, float* g, float* m, float* v, int64_t count,
                     const blh_adam_hyper& h, const double* sumsq_part, int nparts,
                     float* stats_out, LossFinish lf = LossFinish{nullptr, 0, 1.0, nullptr},
                     ShadowDst shadow = NO_SHADOW);
// the gradient arrives as bf16 (compressed data-parallel buckets) times gscale; gout (fp32 arena)
// receives the clipped gradient
int launch_sumsq_bf16(hipStream_t s, const uint16_t* g, int64_t count, float gscale, double* part, int* nparts);
int launch_clip_adam_bf16(hipStream_t s, float* p, const uint16_t* g_bf16, float gscale, float* gout, float* m,
                          float* v, int64_t count, const blh_adam_hyper& h, const double* sumsq_part,
                          int nparts, float* stats_out, ShadowDst shadow = NO_SHADOW);
// device-state variants (graph replay): hyper-parameters and step counters read on the device
int launch_step_state_advance(hipStream_t s, blh_step_state* st);
int launch_clip_adam_dev(hipStream_t s, float* p, float* g, float* m, float* v, int64_t count,
                         const blh_step_state* st, const double* sumsq_part, int nparts,
                         float* stats_out, LossFinish lf = LossFinish{nullptr, 0, 1.0, nullptr},
                         ShadowDst shadow = NO_SHADOW);
int launch_clip_scale(hipStream_t s, float* g, int64_t count, float max_norm,
                      const double* sumsq_part, int nparts, float* stats_out);
// pred = sum(slabs) + bias (+ fused MSE when target != nullptr)
int launch_decode_finish(hipStream_t s, const float* slabs, int splits, int64_t batch,
                         int out_features, const float* bias, float* pred, const float* target,
                         float scale, float* dpred, float* loss_part, int* nparts,
                         float* dbias_part = nullptr);
// skinny.hip: decode forward fused with the MSE loss (pred, dpred, loss / decode-bias partials)
bool decode_fwd_supported(int64_t batch, int W, int OF);
// ---- encode stage without its pre-BatchNorm tensor (encode_f32.hip) ----
bool enc_fused_supported(int64_t batch, int W, int in_features);
int enc_bwd_finish_blocks(int W);
int launch_enc_forward(hipStream_t s, const float* x, const float* W0, const float* b0, const float* gamma,
                       const float* beta, float* running_mean, float* running_var, int64_t* nbt, float momentum,
                       float* saved, float* z0_scratch, float* A, uint32_t* keepbits, int64_t batch, int W,
                       const DropoutSrc& drop);
// one pass over dA0 + the finish kernel: dW0, dgamma, dbeta, the bias column-sum rows, sums of squares (optional)
int launch_enc_backward(hipStream_t s, const float* dA, const float* x, const float* W0, const float* b0,
                        const float* saved, const uint32_t* gatebits, float* z0_scratch, int64_t batch, int W,
                        float* dW0, float* dgamma, float* dbeta, float* db_rows, int db_nrows, double* sq_w,
                        double* sq_gb);
bool enc_fused_supported_h(int64_t batch, int W, int in_features);
// x_f32 != nullptr: xh has not been written yet; the stage's statistics kernel casts x on the way
int launch_enc_forward_h(hipStream_t s, uint16_t* xh, const float* x_f32, const uint16_t* W0h, const float* b0, const float* gamma,
                         const float* beta, float* running_mean, float* running_var, int64_t* nbt, float momentum,
                         float* saved, uint16_t* z0_scratch, uint16_t* A, uint32_t* keepbits, int64_t batch, int W,
                         const DropoutSrc& drop);
int launch_enc_backward_h(hipStream_t s, const uint16_t* dA, const uint16_t* xh, const uint16_t* W0h, const float* b0,
                          const float* saved, const uint32_t* gatebits, uint16_t* z0_scratch, int64_t batch, int W,
                          float* dW0, float* dgamma, float* dbeta, float* db_rows, int db_nrows);
bool decode_fused_supported(int64_t batch, int W, int OF);
int launch_decode_fused(hipStream_t s, const float* A, const float* Wd, const float* bd, const float* target,
                        float* pred, float* dpred, float* dA, float* loss_part, float* dbias_part, int64_t batch,
                        int W, int OF, float scale, int* nparts);
int launch_decode_fwd_mse(hipStream_t s, const float* A, const float* Wd, const float* bd,
                          const float* target, float* pred, float* dpred, float* loss_part,
                          float* dbias_part, int64_t batch, int W, int OF, float scale, int* nparts);
// bf16 storage: A and the bf16 image of Wd; also writes dpred as bf16 (operand of the decode backward)
int launch_decode_fwd_mse_h(hipStream_t s, const uint16_t* A, const uint16_t* Wd, const float* bd,
                            const float* target, float* pred, float* dpred, uint16_t* dpred_h,
                            float* loss_part, float* dbias_part, int64_t batch, int W, int OF, float scale,
                            int* nparts);
// one-pass decode, bf16 storage (skinny.hip): the kernel above + dA = dP Wd (bf16 [batch][W]) from the dP tile and the
// decode weight's image WdT (decode_wdT_dev.h; written by launch_cast2_f32_bf16)
bool decode_fused_h_supported(int64_t batch, int W, int OF);
int launch_decode_fused_h(hipStream_t s, const uint16_t* A, const uint16_t* Wd, const uint16_t* WdT, const float* bd,
                          const float* target, float* pred, float* dpred, uint16_t* dpred_h, uint16_t* dA,
                          float* loss_part, float* dbias_part, int64_t batch, int W, int OF, float scale, int* nparts);
int launch_mpjpe(hipStream_t s, const float* pred, const float* target, const float* mean,
                 const float* stddev, int64_t batch, int joints, float* dist);
int launch_segment_sum(hipStream_t s, const float* dist, const int32_t* ids, int64_t batch,
                       int segments, double* sum, int64_t* count);
int launch_dropout_mask(hipStream_t s, uint8_t* out, int64_t batch, int W, const DropoutSrc& drop);

}  // namespace blh
