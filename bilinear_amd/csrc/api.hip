// C ABI of libbilinear_hip.so: arena layout, workspace carving and the fixed kernel
// DAG of the lifter's forward / backward / optimiser step.  Pure enqueue code: no
// allocation, no synchronisation, so every entry point is hipGraph-capturable.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include <mutex>
#include "common.h"
#include "gemm_bf16s_kernel.h"
#include "api_layout.h"

namespace blh {

thread_local int g_last_hip_error = 0;
thread_local hipEvent_t tl_stop_event = nullptr;

// SyncBN plumbing of the current call (data parallel): statistics over `global_batch` rows,
// exchanged by the host callback
struct SyncCtx { blh_sync_fn fn; void* user; int64_t global_batch; };

}  // namespace blh

// Caller-owned context (include/bilinear_hip.h): the side stream of the two-stream backward with
// its fork / join events, the option flags, and the per-call plumbing (SyncBN callback, device
// address of the captured step's dropout counter).  Bound to one device; one call at a time.
static constexpr int BLH_CTX_EVENTS = 34;      // 1 + 2*num_blocks <= 32 stages, + decode, + spare
struct blh_context {
  int device = -1;
  hipStream_t s2 = nullptr;
  hipEvent_t ev_dz[BLH_CTX_EVENTS], ev_w[BLH_CTX_EVENTS], ev_r[BLH_CTX_EVENTS];
  bool two_stream = true;
  bool defer_slabs = false;
  int late_fork = 2;        // BLH_OPT_LATE_FORK: 0 early, 1 late, 2 auto
  // per-call state (set by the entry point for the duration of the call)
  blh::SyncCtx sync = {nullptr, nullptr, 0};
  const uint64_t* step_dev = nullptr;
  // left by blh_forward_train_loss for blh_backward(dpred == NULL): rows of that forward and the
  // number of decode-bias / loss partial rows its decode kernel wrote (0: none)
  int64_t loss_batch = 0;
  int loss_nparts = 0;
  // BLH_OPT_PERSISTENT_SHADOW: the (params, workspace) whose bf16 parameter image the last fused
  // step's Adam kernel left up to date (nullptr: none)
  bool persistent_shadow = false;
  const void* shadow_params = nullptr;
  const void* shadow_ws = nullptr;
  // grid barrier of the fused forward stage (gemm_bf16s_bnfwd.h): arrivals, generation, timeouts; device memory
  // owned by the context, zeroed once here
  uint32_t* grid_bar = nullptr;
};

namespace blh {

static DropoutSrc layer_drop(const blh_context* ctx, const blh_dropout* drop, int layer,
                             int64_t batch, int W) {
  DropoutSrc d;
  d.step_dev = ctx->step_dev;
  d.keep = drop->keep_mask ? drop->keep_mask + (int64_t)layer * batch * W : nullptr;
  d.seed = drop->seed; d.step = drop->step; d.row_offset = drop->row_offset;
  d.layer = drop->layer_base + layer;
  return d;
}

// the process-wide side stream table (streams.hip)
hipStream_t side_stream_current(int device);
hipError_t side_stream_acquire(int device, hipStream_t* out);
void side_stream_release(int device);

// every network-level entry point: the context must belong to the device that is current; it picks up
// the device's current side stream (blh_side_stream_renew may have replaced it since the last call)
static int check_ctx(blh_context* ctx) {
  if (!ctx) return BLH_ERR_INVALID_ARGUMENT;
  int dev = -1;
  BLH_HIP_TRY(hipGetDevice(&dev));
  if (dev != ctx->device) return BLH_ERR_INVALID_ARGUMENT;
  ctx->s2 = side_stream_current(dev);
  return ctx->s2 ? BLH_OK : BLH_ERR_INVALID_ARGUMENT;
}

// ------------------------------------------------------------- forward -----
static int forward_impl(blh_context* ctx, const blh_model_desc* d, hipStream_t s, const float* params,
                        float* bn_running, int64_t* nbt, const float* x, const blh_dropout* drop,
                        float momentum, const Workspace& ws, float* pred, int64_t batch,
                        bool train, const float* target, float mse_scale, float* loss_part,
                        int* loss_nparts) {
  const ArenaLayout L = make_layout(d);
  const int nh = (int)L.heavy.size();
  const int W = d->width;
  const int tiles_m = (int)ceil_div(batch, 128);
  if (ws.amax_W)   // gemm_dtype 3: max |w| of every hidden Linear weight, once per forward
    for (int i = 1; i < nh; ++i)
      BLH_TRY(launch_wamax(s, params + L.heavy[i].w, 0, 1, (int64_t)W * W, ws.amax_W + (int64_t)i * WAMAX_PARTS));
  for (int i = 0; i < nh; ++i) {
    const HeavyOffsets& h = L.heavy[i];
    const float* in = (i == 0) ? x : ws.A[i - 1];
    GemmParams g{};
    g.A = in; g.lda = h.fan_in;
    g.B = params + h.w; g.ldb = h.fan_in;
    if (ws.amax_W && i >= 1) {
      g.a_amax = ws.amax_A[i - 1]; g.a_namax = ws.amax_parts;
      g.b_amax = ws.amax_W + (int64_t)i * WAMAX_PARTS; g.b_namax = WAMAX_PARTS;
    }
    g.C = ws.Z[i]; g.ldc = W;
    g.M = (int)batch; g.N = W; g.K = h.fan_in; g.k_per_split = h.fan_in;
    g.bias = params + h.b;
    g.stat_part = ws.stat_part;
    const Splits fs = small_m_splits(batch, W, h.fan_in);
    const bool enc64 = train && i == 0 && fs.splits == 1 && h.fan_in <= 32 && batch >= 2048;
    if (fs.splits > 1) {
      // small batch: too few 128x128 output tiles to fill the chip and each would walk the
      // whole reduction alone (latency-bound), so cut the reduction across workgroups and
      // finish (slab sum + bias + BN tile statistics) in a streaming kernel
      g.C = ws.slabs; g.c_split_stride = batch * (int64_t)W; g.k_per_split = fs.k_per;
      BLH_TRY(launch_gemm(s, TILE_128x128, ROWK, ROWK, EPI_STORE, g, fs.splits, d->gemm_dtype));
      BLH_TRY(launch_fwd_finish(s, ws.slabs, fs.splits, batch, W, params + h.b, ws.Z[i],
                                train ? ws.stat_part : nullptr));
    } else if (!train && d->gemm_dtype != 3) {
      // eval: the whole heavy_linear in one kernel — bias, BatchNorm with the running statistics,
      // ReLU and the block skip sit in the GEMM epilogue (the BN "folded into the Linear" of
      // SURVEY.md 8(f) rank 1); Z is not materialised.  (fp16x2 keeps the two-kernel form: its
      // next GEMM wants the maximum of A that bn_apply gathers.)
      g.C = ws.A[i];
      g.bn_gamma = params + h.gamma; g.bn_beta = params + h.beta;
      g.bn_mean = bn_running + ((int64_t)i * 2 + 0) * W;
      g.bn_var = bn_running + ((int64_t)i * 2 + 1) * W;
      g.addend = (i >= 2 && (i % 2) == 0) ? ws.A[i - 2] : nullptr; g.ldadd = W;
      BLH_TRY(launch_gemm(s, TILE_128x128, ROWK, ROWK, EPI_BN_RELU, g, 1, d->gemm_dtype));
      continue;
    } else if (enc64) {
      // encode (K = 32): one K tile, the kernel is all prologue + 16.8 MB of output; 64-row tiles
      // put two workgroups on every CU, so one's DMA wait overlaps the other's stores
      BLH_TRY(launch_gemm(s, TILE_64x128, ROWK, ROWK, EPI_BIAS_STATS, g, 1, d->gemm_dtype));
    } else {
      BLH_TRY(launch_gemm(s, TILE_128x128, ROWK, ROWK, train ? EPI_BIAS_STATS : EPI_BIAS, g, 1,
                          d->gemm_dtype));
    }
    // second stage of a block adds the block input (model/bilinear.py:36-38)
    const float* skip = (i >= 2 && (i % 2) == 0) ? ws.A[i - 2] : nullptr;
    float* rm = bn_running + ((int64_t)i * 2 + 0) * W;
    float* rv = bn_running + ((int64_t)i * 2 + 1) * W;
    if (train) {
      float* sv = ws.bn_saved[i];
      // (the small-batch path produced one statistics tile covering all rows)
      const int st_tiles = fs.splits > 1 ? 1 : (enc64 ? (int)ceil_div(batch, 64) : tiles_m);
      const int st_rows = fs.splits > 1 ? (int)batch : (enc64 ? 64 : 128);
      if (ctx->sync.fn) {
        BLH_TRY(launch_bn_fwd_local_sums(s, ws.stat_part, st_tiles, st_rows, batch, W, ws.sync_buf));
        ctx->sync.fn(ctx->sync.user, ws.sync_buf, 2 * (int64_t)W, 1);
        BLH_TRY(launch_bn_fwd_finalize_sums(s, ws.sync_buf, ctx->sync.global_batch, W,
                                            params + h.gamma, params + h.beta, rm, rv, nbt + i,
                                            momentum, sv, sv + W, sv + 2 * W, sv + 3 * W));
      } else {
        BLH_TRY(launch_bn_fwd_finalize(s, ws.stat_part, st_tiles, st_rows, batch, W,
                                       params + h.gamma, params + h.beta, rm, rv, nbt + i,
                                       momentum, sv, sv + W, sv + 2 * W, sv + 3 * W));
      }
      BLH_TRY(launch_bn_apply_f2(s, true, ws.Z[i], sv + 2 * W, sv + 3 * W, nullptr, nullptr, nullptr,
                                 nullptr, skip, ws.A[i], ws.keep[i], batch, W,
                                 layer_drop(ctx, drop, i, batch, W), nbt + i, ws.amax_A[i]));
    } else {
      DropoutSrc none{nullptr, 0, 0, 0, 0, nullptr};
      BLH_TRY(launch_bn_apply_f2(s, false, ws.Z[i], nullptr, nullptr, params + h.gamma, params + h.beta,
                                 rm, rv, skip, ws.A[i], nullptr, batch, W, none, nullptr, ws.amax_A[i]));
    }
  }
  // decode (model/bilinear.py:39): N = 48 gives only B/128 output tiles, so the reduction
  // over W is split across workgroups (slabs) and a small kernel adds the slabs, the bias and,
  // in the fused step, the MSE loss / gradient (train_bilinear.py:78).
  const int OF = d->out_features;
  if (decode_fwd_supported(batch, W, OF)) {
    // purpose-built kernel (skinny.hip): reads A once, no slabs, bias + MSE + dpred + the loss and
    // decode-bias partials in the same launch
    int np = 0;
    BLH_TRY(launch_decode_fwd_mse(s, ws.A[nh - 1], params + L.dec_w, params + L.dec_b, target, pred,
                                  target ? ws.dpred : nullptr, loss_part,
                                  target ? ws.dec_bias_part : nullptr, batch, W, OF, mse_scale, &np));
    if (loss_nparts) *loss_nparts = np;
    return BLH_OK;
  }
  const Splits sp = decode_fwd_splits(batch, W);
  GemmParams g{};
  g.A = ws.A[nh - 1]; g.lda = W;
  g.B = params + L.dec_w; g.ldb = W;
  g.C = ws.slabs; g.ldc = OF; g.c_split_stride = batch * OF;
  g.M = (int)batch; g.N = OF; g.K = W; g.k_per_split = sp.k_per;
  BLH_TRY(launch_gemm(s, TILE_128x64, ROWK, ROWK, EPI_STORE, g, sp.splits, d->gemm_dtype));
  return launch_decode_finish(s, ws.slabs, sp.splits, batch, OF, params + L.dec_b, pred, target,
                              mse_scale, target ? ws.dpred : nullptr, loss_part, loss_nparts,
                              target ? ws.dec_bias_part : nullptr);
}

// ------------------------------------------------------------ backward -----
// dW = dZ^T act, the reduction over the batch split across workgroups.  With `defer` the
// partial slabs stay in `slabs` (a per-stage buffer) and *region records them for the single
// grads_finish launch at the end of backward; otherwise they are summed right away.
static int wgrad(int dtype, hipStream_t s, GemmTile tile, const float* dZ, int64_t ld_dz, int M,
                 const float* act, int64_t ld_act, int N, int64_t batch, int64_t tiles,
                 float* slabs, float* out, GradRegion* region, const float* amax_dz = nullptr,
                 const float* amax_act = nullptr, int amax_parts = 0, double* sq = nullptr, int sq_blocks = 0) {
  const Splits sp = pick_splits(batch, tiles);
  GemmParams g{};
  g.A = dZ; g.lda = ld_dz;
  g.B = act; g.ldb = ld_act;
  if (amax_dz && amax_act) {   // gemm_dtype 3
    g.a_amax = amax_dz; g.a_namax = amax_parts;
    g.b_amax = amax_act; g.b_namax = amax_parts;
  }
  g.M = M; g.N = N; g.K = (int)batch; g.k_per_split = sp.k_per;
  g.ldc = N;
  if (region) { region->slabs = nullptr; region->splits = 0; }
  if (sp.splits == 1) {
    g.C = out; g.c_split_stride = 0;
    return launch_gemm(s, tile, KROW, KROW, EPI_STORE, g, 1, dtype);
  }
  g.C = slabs; g.c_split_stride = (int64_t)M * N;
  BLH_TRY(launch_gemm(s, tile, KROW, KROW, EPI_STORE, g, sp.splits, dtype));
  if (region) {
    region->slabs = slabs; region->splits = sp.splits;
    return BLH_OK;
  }
  if (sq) return launch_sum_slabs_sq(s, slabs, (int64_t)M * N, sp.splits, out, sq, sq_blocks);
  return launch_sum_slabs(s, slabs, (int64_t)M * N, sp.splits, out);
}

// Side stream for the weight-gradient GEMMs (owned by the context): nothing in the rest of
// backward depends on dW, so wgrad(l) (+ its slab sum) runs on a second stream concurrently
// with dgrad(l) and the HBM-bound BatchNorm-backward kernels of stage l-1, which leave the
// MFMA pipes idle.  Fork / join by events (capturable into a hipGraph); dZ is double-buffered
// so that stage l-2 does not overwrite what wgrad(l) is still reading.
// fused: the caller is the whole-step path: the decode-bias partials come from decode_finish
// (dec_bias_S rows) and the sum-of-squares partials of the arena are returned for clip+Adam.
// sumsq_src[0]: where the partials really are when backward_impl returns (sumsq_part, or the producers' array)
struct FusedBackward { int dec_bias_S; double* sumsq_part; int* sumsq_nparts; double** sumsq_src; };

static int backward_impl(blh_context* ctx, const blh_model_desc* d, hipStream_t s, const float* params,
                         const float* x, const blh_dropout* drop, const Workspace& ws,
                         const float* dpred, float* grads, int64_t batch,
                         blh_grad_ready_fn on_ready, void* user,
                         const FusedBackward* fused = nullptr) {
  const ArenaLayout L = make_layout(d);
  const int nh = (int)L.heavy.size();
  const int W = d->width;
  const int OF = d->out_features;
  const int chunks = ew_num_row_chunks(batch);

  // The split-K slabs of each stage are summed right after its wgrad, while they are still in
  // L2 / Infinity Cache.  Deferring all of them to the single grads_finish launch at the end
  // (BLH_DEFER_SLABS=1, kept for experiments) saves six launches but reads 68 MB of by then
  // cold slabs: measured 1.268 vs 1.253 ms/step, so it is off.
  tl_stop_event = nullptr;
  const bool defer = (on_ready == nullptr) && ctx->defer_slabs;
  std::vector<GradRegion> wreg(nh + 1);
  // Fused step, gradient norm (clip_grad_norm_, train_bilinear.py:81): the kernels that WRITE the gradient ranges
  // — slab sums of the weight gradients, the gamma / beta finalize, the bias reduction — also emit the
  // sum-of-squares partials of what they write (ws.sumsq_fold: a dense array, every launch fills exactly its
  // own slots, so the sum is deterministic), and no pass over the arena is left between the join and
  // clip + Adam (grads_finish: 9 us + its launch gap at configs[1]).  Needs every weight gradient to come out
  // of a slab sum, and the slots to fit.
  struct Fold { bool on = false; int per_w = 0; std::vector<int> w_off; int gb0 = 0, bias0 = 0, total = 0; } fold;
  if (fused && fused->sumsq_part && !on_ready && !defer && !getenv("BLH_NO_SUMSQ_FOLD") && W % 16 == 0 &&
      fused->dec_bias_S > 0) {
    const bool slabbed = pick_splits(batch, ceil_div(W, 128) * ceil_div(W, 128)).splits > 1 &&
                         pick_splits(batch, ceil_div(W, 128) * ceil_div(d->in_features, 32)).splits > 1 &&
                         pick_splits(batch, ceil_div(OF, 64) * ceil_div(W, 128)).splits > 1;
    fold.per_w = (int)std::min<int64_t>(256, std::max<int64_t>(8, 2048 / (nh + 1)));
    int off = 0;
    for (int i = 0; i <= nh; ++i) {
      fold.w_off.push_back(off);
      const int64_t cnt = i == nh ? (int64_t)OF * W : (int64_t)W * L.heavy[i].fan_in;
      off += sum_slabs_sq_blocks(cnt, fold.per_w);
    }
    fold.gb0 = off;
    fold.bias0 = fold.gb0 + nh * bn_bwd_finalize_blocks(W);
    fold.total = fold.bias0 + bias_colreduce_blocks(W, nh, true);
    fold.on = slabbed && nh >= 2 && fold.total <= SUMSQ_FOLD_PARTS;
  }
  auto fold_w = [&](int i) -> double* { return fold.on ? ws.sumsq_fold + fold.w_off[i] : nullptr; };
  // two streams: on by default (-3 % step)
  const bool two = ctx->two_stream && !ctx->sync.fn && !defer && small_m_splits(batch, W, W).splits == 1;
  hipStream_t s2 = two ? ctx->s2 : s;
  // auto (fp32 kernels): early.  The data-gradient launch fills the chip's LDS — 256 workgroups of the
  // 128 KB form (gemm_f32_backward_exclusive) or >= 512 of the 64 KB form — so the weight gradient
  // cannot become resident beside it: the dispatcher places its workgroups as the data gradient's
  // retire, nothing waits for a cross-queue signal, and the BatchNorm chain of the next stage runs
  // beside the weight gradient.  Measured early vs late: B 2048 0.80 / 0.87, B 4096 1.032 / 1.054,
  // B 8192 1.88 / 1.98, B 16384 3.59 / 3.62 ms (profiles/r03_backward_schedule.md).  The split-
  // precision modes keep the late fork they were measured with.
  bool late_policy = ctx->late_fork != 0;
  if (ctx->late_fork == 2 && d->gemm_dtype == 0) late_policy = false;
  SyncCtx& g_sync = ctx->sync;
  blh_context& g_side = *ctx;
  // Fork: s2 continues behind a kernel of s.  Outside stream capture the event rides on that
  // kernel's own completion signal (arm_fork before its launch, fork_wait after: common.h,
  // tl_stop_event) instead of a marker packet behind it; under capture (events are graph edges
  // there, not packets) it is recorded the ordinary way.  wdone: marks wgrad(idx) complete.
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cap);
  const bool attach = two && cap == hipStreamCaptureStatusNone;
  auto arm_fork = [&](int idx) {
    if (attach) tl_stop_event = g_side.ev_dz[idx];
  };
  auto fork_wait = [&](int idx) -> int {
    if (!two) return BLH_OK;
    if (!attach) BLH_HIP_TRY(hipEventRecord(g_side.ev_dz[idx], s));
    BLH_HIP_TRY(hipStreamWaitEvent(s2, g_side.ev_dz[idx], 0));
    return BLH_OK;
  };
  auto wdone = [&](int idx) -> int {
    if (!two) return BLH_OK;
    BLH_HIP_TRY(hipEventRecord(g_side.ev_w[idx], s2));
    return BLH_OK;
  };
  // Data-parallel hook.  A range's weight gradient is produced on the side stream, its bias /
  // gamma / beta gradients on the main stream: the side stream is made to wait for the main one
  // (everything enqueued so far), so that the range is complete ON THE SIDE STREAM when the
  // host callback runs — blh_backward_side_stream() tells the caller which stream that is.
  auto ready = [&](int idx, int64_t off, int64_t cnt) -> int {
    if (!on_ready) return BLH_OK;
    if (two) {
      // nothing to do: every part of the range was produced on the side stream behind the stage's
      // fork (weight gradient, bias reduction) or on the main stream in front of it (gamma / beta)
    } else if (ctx->two_stream) {   // small-batch / SyncBN call of a two-stream context: the
      // range is complete on `s`; keep the contract "complete on the side stream"
      BLH_HIP_TRY(hipEventRecord(g_side.ev_r[idx], s));
      BLH_HIP_TRY(hipStreamWaitEvent(g_side.s2, g_side.ev_r[idx], 0));
    }
    on_ready(user, off, cnt);
    return BLH_OK;
  };
  // decode: dA_last = dP W_d on the main stream first (it carries the fork event: the decode
  // weight gradient then starts when it completes, next to the first BatchNorm-backward kernels),
  // dW = dP^T A_last on the side stream, db = colsum(dP)
  {
    GemmParams g{};
    g.A = dpred; g.lda = OF;
    g.B = params + L.dec_w; g.ldb = W;
    g.C = ws.G0; g.ldc = W;
    g.M = (int)batch; g.N = W; g.K = OF; g.k_per_split = OF;
    arm_fork(nh);
    // (K = 48: two K tiles and 16.8 MB of output; 64-row tiles = two workgroups per CU)
    BLH_TRY(launch_gemm(s, batch >= 2048 ? TILE_64x128 : TILE_128x128, ROWK, KROW, EPI_STORE, g, 1,
                        d->gemm_dtype));
    tl_stop_event = nullptr;
  }
  BLH_TRY(fork_wait(nh));
  BLH_TRY(wgrad(d->gemm_dtype, s2, TILE_64x128, dpred, OF, OF, ws.A[nh - 1], W, W, batch,
                ceil_div(OF, 64) * ceil_div(W, 128), defer ? ws.stage_slabs[nh] : ws.slabs,
                grads + L.dec_w, defer ? &wreg[nh] : nullptr, nullptr, nullptr, 0, fold_w(nh), fold.per_w));
  BLH_TRY(wdone(nh));
  // (decode bias: on the side stream under a hook — behind the fork, its inputs are older than that)
  if (!fused) BLH_TRY(launch_colsum(on_ready ? s2 : s, dpred, batch, OF, OF, ws.colsum_part, grads + L.dec_b));
  else if (on_ready)   // (the hook wants the decode range complete now; else: one batched launch at the end)
    BLH_TRY(launch_colreduce(s2, ws.dec_bias_part, fused->dec_bias_S, OF, OF, grads + L.dec_b));
  BLH_TRY(ready(nh, L.dec_w, L.total - L.dec_w));

  for (int i = nh - 1; i >= 0; --i) {
    const HeavyOffsets& h = L.heavy[i];
    // gradient w.r.t. this stage's output: block boundaries live in G0, the middle of a
    // block in G1 (stage i odd = first of a block: its output feeds only stage i+1)
    const bool first_of_block = (i >= 1) && (i % 2 == 1);
    const float* dA = first_of_block ? ws.G1 : ws.G0;
    const float* sv = ws.bn_saved[i];
    // (dropout: the keep bits the forward wrote, ws.keep[i])
    BLH_TRY(launch_bn_bwd_reduce_f2(s, dA, ws.Z[i], sv + 2 * W, sv + 3 * W, ws.keep[i], ws.bn_part, batch, W));
    BLH_TRY(launch_bn_bwd_finalize_h2(s, ws.bn_part, chunks, W, sv, sv + W, grads + h.gamma, grads + h.beta,
                                      fold.on ? ws.sumsq_fold + fold.gb0 + i * bn_bwd_finalize_blocks(W) : nullptr));
    const float* dg = grads + h.gamma;
    const float* db = grads + h.beta;
    int64_t norm_batch = batch;
    if (g_sync.fn) {
      // the parameter gradients keep the LOCAL sums (averaged later with the rest of the
      // arena); the BN backward itself needs the sums over the global batch
      float* sb = reinterpret_cast<float*>(ws.sync_buf);
      BLH_HIP_TRY(hipMemcpyAsync(sb, grads + h.gamma, W * sizeof(float), hipMemcpyDeviceToDevice, s));
      BLH_HIP_TRY(hipMemcpyAsync(sb + W, grads + h.beta, W * sizeof(float), hipMemcpyDeviceToDevice, s));
      g_sync.fn(g_sync.user, sb, 2 * (int64_t)W, 0);
      dg = sb; db = sb + W; norm_batch = g_sync.global_batch;
    }
    float* dzbuf = ws.dZ[i];
    float* dz_amax = ws.amax_dZ[i & 1];
    // (the two amax partial buffers alternate: the fp16x2 wgrad(i+2) on the side stream may still
    //  be reading the one bn_bwd_apply(i) is about to write)
    if (two && dz_amax && i + 2 <= nh - 1)
      BLH_HIP_TRY(hipStreamWaitEvent(s, g_side.ev_w[i + 2], 0));
    // Schedule (profiles/r02_step_timeline.md).  The weight gradient of stage i forks behind the
    // data-gradient GEMM of the stage (BLH_OPT_LATE_FORK, default): that GEMM runs alone at full
    // speed, and the weight gradient then runs beside the HBM-bound BatchNorm-backward kernels of
    // stage i-1 (which raise their wave priority, elementwise.hip) and the first half of the next
    // data-gradient GEMM: 148 us per stage.  With the option off it forks behind bn_bwd_apply(i) and
    // starts together with the data-gradient GEMM (two workgroups per CU, 127 us per pair against
    // 134 us one after the other, but the BatchNorm chain then sits between two GEMM pairs): 158 us
    // per stage, step 1.091 against 1.078 ms.
    // Stage 0 has no data gradient: its weight gradient stays on the main stream (a fork + join
    // there only adds two cross-queue latencies at the very end of backward) unless the
    // data-parallel hook wants every range complete on the side stream.
    const bool side = two && (i > 0 || on_ready != nullptr);
    const bool fork_late = side && late_policy && i > 0 && small_m_splits(batch, W, W).splits == 1;
    hipStream_t sw = side ? s2 : s;
    if (side && !fork_late) arm_fork(i);
    BLH_TRY(launch_bn_bwd_apply_f2(s, dA, ws.Z[i], sv + 2 * W, sv + 3 * W, sv, sv + W, dg, db, ws.keep[i],
                                   dzbuf, ws.dz_colsum_part + (int64_t)i * chunks * W, batch, W,
                                   norm_batch, dz_amax));
    tl_stop_event = nullptr;
    if (side && !fork_late) BLH_TRY(fork_wait(i));
    if (fork_late) arm_fork(i);
    // Linear: db = colsum(dZ); dW = dZ^T a_in; d a_in = dZ W
    if (i > 0) {
      GemmParams g{};
      g.A = dzbuf; g.lda = W;
      g.B = params + h.w; g.ldb = W;
      g.M = (int)batch; g.N = W; g.K = W; g.k_per_split = W;
      g.ldc = W;
      if (dz_amax && ws.amax_W) {   // gemm_dtype 3
        g.a_amax = dz_amax; g.a_namax = ws.amax_parts;
        g.b_amax = ws.amax_W + (int64_t)i * WAMAX_PARTS; g.b_namax = WAMAX_PARTS;
      }
      const Splits ds2 = small_m_splits(batch, W, W);
      float* dst = first_of_block ? ws.G0 : ws.G1;
      if (ds2.splits > 1) {
        g.C = ws.slabs; g.c_split_stride = batch * (int64_t)W; g.k_per_split = ds2.k_per;
        BLH_TRY(launch_gemm(s, TILE_128x128, ROWK, KROW, EPI_STORE, g, ds2.splits, d->gemm_dtype));
        BLH_TRY(launch_sum_slabs_add(s, ws.slabs, batch * (int64_t)W, ds2.splits,
                                     first_of_block ? ws.G0 : nullptr, dst));
      } else {
        if (first_of_block) {
          // d(block input) = dZ W + d(block output)   (skip path), in place in G0
          g.C = ws.G0; g.addend = ws.G0; g.ldadd = W;
          BLH_TRY(launch_gemm(s, TILE_128x128, ROWK, KROW, EPI_ADD, g, 1, d->gemm_dtype));
        } else {
          g.C = ws.G1;
          BLH_TRY(launch_gemm(s, TILE_128x128, ROWK, KROW, EPI_STORE, g, 1, d->gemm_dtype));
        }
      }
    }
    tl_stop_event = nullptr;
    if (fork_late) BLH_TRY(fork_wait(i));
    // (data parallel: the bucket hook needs this stage's bias gradient now; otherwise all
    //  stages are reduced by one launch after the loop)
    if (on_ready)   // on the side stream (after the fork): nothing on the main stream waits for it
      BLH_TRY(launch_colreduce(sw, ws.dz_colsum_part + (int64_t)i * chunks * W, chunks, W, W,
                               grads + h.b));
    if (i == 0) {
      // (on the main stream it takes its own slab buffer: the shared one may still be in use by
      //  wgrad(1) on the side stream, and waiting for that costs a cross-queue latency of ~10 us
      //  at the very end of backward)
      BLH_TRY(wgrad(d->gemm_dtype, sw, TILE_128x32, dzbuf, W, W, x, d->in_features,
                    d->in_features, batch, ceil_div(W, 128) * ceil_div(d->in_features, 32),
                    (defer || (two && !side)) ? ws.stage_slabs[0] : ws.slabs, grads + h.w,
                    defer ? &wreg[0] : nullptr, nullptr, nullptr, 0, fold_w(0), fold.per_w));
      if (side) BLH_TRY(wdone(0));
    } else {
      BLH_TRY(wgrad(d->gemm_dtype, sw, TILE_128x128, dzbuf, W, W, ws.A[i - 1], W, W, batch,
                    ceil_div(W, 128) * ceil_div(W, 128), defer ? ws.stage_slabs[i] : ws.slabs,
                    grads + h.w, defer ? &wreg[i] : nullptr, dz_amax, ws.amax_A[i - 1],
                    ws.amax_parts, fold_w(i), fold.per_w));
      BLH_TRY(wdone(i));
    }
    if (on_ready) {
      const int64_t end = (i + 1 < nh) ? L.heavy[i + 1].w : L.dec_w;
      BLH_TRY(ready(i, h.w, end - h.w));
    }
  }
  // join: s2 is in order (stage 0 ran on the main stream unless the bucket hook wants the side one)
  // (the bias reduction needs nothing from the side stream: it goes in front of the join, whose
  //  cross-queue wait costs the main stream ~9 us even when the signal is already there)
  if (!on_ready) {
    int64_t offs[32];
    if (nh > 32) return BLH_ERR_SHAPE;
    for (int i = 0; i < nh; ++i) offs[i] = L.heavy[i].b;
    BLH_TRY(launch_bias_colreduce(s, ws.dz_colsum_part, (int64_t)chunks * W, chunks, W, nh, offs,
                                  grads, fused ? ws.dec_bias_part : nullptr,
                                  fused ? fused->dec_bias_S : 0, OF, L.dec_b,
                                  fold.on ? ws.sumsq_fold + fold.bias0 : nullptr));
  }
  if (two) BLH_HIP_TRY(hipStreamWaitEvent(s, g_side.ev_w[on_ready != nullptr ? 0 : 1], 0));
  if (fold.on) {     // the producers left the partials: hand them to clip + Adam
    fused->sumsq_src[0] = ws.sumsq_fold;
    *fused->sumsq_nparts = fold.total;
  } else if (!on_ready) {   // (covers the no_defer A/B mode too: its regions are all plain)
    // regions in arena order: [weight (slabs or plain)] [bias, gamma, beta (plain)] per stage,
    // then decode weight and decode bias (+ tail padding)
    GradRegions R{};
    auto push = [&](int64_t off, int64_t end, const GradRegion* w) {
      GradRegion& r = R.r[R.n++];
      r.off4 = off / 4; r.cnt4 = (end - off) / 4;
      r.slabs = w ? w->slabs : nullptr; r.splits = w ? w->splits : 0; r.first_block = 0;
    };
    for (int i = 0; i < nh; ++i) {
      const HeavyOffsets& h = L.heavy[i];
      const int64_t wend = h.w + (int64_t)W * h.fan_in;
      push(h.w, wend, &wreg[i]);
      push(wend, (i + 1 < nh) ? L.heavy[i + 1].w : L.dec_w, nullptr);
    }
    push(L.dec_w, L.dec_w + (int64_t)OF * W, &wreg[nh]);
    push(L.dec_w + (int64_t)OF * W, L.total, nullptr);
    BLH_TRY(launch_grads_finish(s, grads, R, L.total / 4, fused ? fused->sumsq_part : nullptr,
                                fused ? fused->sumsq_nparts : nullptr));
  }
  return BLH_OK;
}


// =================================================================================================
// gemm_dtype 4 — "bf16s": bf16 storage (BASELINE configs 3-5).  Every [B,W] tensor (pre-BN output
// Z, activation A, gradients G / dZ), the network input and a shadow of all parameters are bf16
// in HBM; every contraction runs on gemm_bf16s_kernel.h (bf16 MFMA, fp32 accumulate, operands fed
// by LDS-DMA without any conversion); BatchNorm statistics (from the fp32 accumulators, before
// rounding), the parameters, their gradients (fp32 slabs of the weight-gradient GEMM), Adam and
// the loss stay fp32.  Gradients need no loss scaling: bf16 keeps fp32's exponent range.
// =================================================================================================
static int forward_h(blh_context* ctx, const blh_model_desc* d, hipStream_t s, const float* params,
                     float* bn_running, int64_t* nbt, const float* x, const blh_dropout* drop,
                     float momentum, const WorkspaceH& ws, float* pred, int64_t batch, bool train,
                     bool shadow_valid = false, const float* target = nullptr, float mse_scale = 0.f,
                     int* loss_nparts = nullptr) {
  const ArenaLayout L = make_layout(d);
  const int nh = (int)L.heavy.size();
  const int W = d->width, OF = d->out_features, IF = d->in_features;
  // bf16 images of the parameters (the GEMMs read the weights from it) and of the input
  // (shadow_valid: the previous fused step's Adam kernel wrote it, BLH_OPT_PERSISTENT_SHADOW)
  ctx->shadow_params = ctx->shadow_ws = nullptr;
  if (!shadow_valid) BLH_TRY(launch_cast_f32_bf16(s, params, ws.wsh, L.total));
  BLH_TRY(launch_cast_f32_bf16(s, x, ws.xh, batch * IF));
  for (int i = 0; i < nh; ++i) {
    const HeavyOffsets& h = L.heavy[i];
    GemmParamsH g{};
    g.A = (i == 0) ? ws.xh : ws.A[i - 1]; g.lda = h.fan_in;
    g.B = ws.wsh + h.w; g.ldb = h.fan_in;
    g.C = ws.Z[i]; g.ldc = W;
    g.M = (int)batch; g.N = W; g.K = h.fan_in; g.k_per_split = h.fan_in;
    g.bias = params + h.b; g.stat_part = ws.stat_part;
    // (BatchNorm partials: one (mean, M2) pair per row tile of the kernel that ran, 128 or 256 rows)
    const int tile = gemm_bf16s_pick_tile(ROWK, ROWK, true, g, 1);
    const int st_rows = gemm_bf16s_tile_rows(tile);
    const int st_tiles = (int)ceil_div(batch, st_rows);
    const uint16_t* skip = (i >= 2 && (i % 2) == 0) ? ws.A[i - 2] : nullptr;
    float* rm = bn_running + ((int64_t)i * 2 + 0) * W;
    float* rv = bn_running + ((int64_t)i * 2 + 1) * W;
    // The whole stage in ONE launch (gemm_bf16s_bnfwd.h: statistics merged behind a grid barrier, BatchNorm +
    // ReLU + dropout + skip applied to the tile the workgroup still holds): big-tile kernels whose grid fits the
    // chip one workgroup per CU (configs[2]: 256 tiles of 256 x 256; configs[3] per GPU: 256 of 128 x 256), per-rank
    // statistics.  Bit-identical to the three-launch form and MEASURED SLOWER (configs[2] 1.59 against 1.47 ms,
    // configs[3] per-GPU shape 1.10 against 0.95: profiles/r04_fused_forward.md), so it is opt-in: BLH_FWD_FUSE=1.
    if (train && !ctx->sync.fn && tile != H_TILE_128 && ctx->grid_bar && getenv("BLH_FWD_FUSE") &&
        (int64_t)st_tiles * (W / 256) <= gemm_bf16s_fused_forward_max_wgs() && st_tiles <= 128) {
      g.fwd.gamma = params + h.gamma; g.fwd.beta = params + h.beta;
      g.fwd.running_mean = rm; g.fwd.running_var = rv; g.fwd.nbt = nbt + i; g.fwd.momentum = momentum;
      g.fwd.saved = ws.bn_saved[i];
      g.fwd.skip = skip; g.fwd.ldskip = W;
      g.fwd.A = ws.A[i]; g.fwd.lda_out = W;
      g.fwd.keepbits = ws.keep[i];
      g.fwd.drop = layer_drop(ctx, drop, i, batch, W);
      g.fwd.bar = ctx->grid_bar;
      g.fwd.tile_rows = st_rows;
      BLH_TRY(launch_gemm_bf16s(s, ROWK, ROWK, EPI_BN_FWD, true, g, 1));
      continue;
    }
    BLH_TRY(launch_gemm_bf16s(s, ROWK, ROWK, train ? EPI_BIAS_STATS : EPI_BIAS, true, g, 1));
    if (train) {
      float* sv = ws.bn_saved[i];
      if (ctx->sync.fn) {   // SyncBN: statistics over the global batch (fp64 sums exchanged by the host)
        BLH_TRY(launch_bn_fwd_local_sums(s, ws.stat_part, st_tiles, st_rows, batch, W, ws.sync_buf));
        ctx->sync.fn(ctx->sync.user, ws.sync_buf, 2 * (int64_t)W, 1);
        BLH_TRY(launch_bn_fwd_finalize_sums(s, ws.sync_buf, ctx->sync.global_batch, W, params + h.gamma,
                                            params + h.beta, rm, rv, nbt + i, momentum, sv, sv + W,
                                            sv + 2 * W, sv + 3 * W));
      } else {
        BLH_TRY(launch_bn_fwd_finalize(s, ws.stat_part, st_tiles, st_rows, batch, W, params + h.gamma,
                                       params + h.beta, rm, rv, nbt + i, momentum, sv, sv + W,
                                       sv + 2 * W, sv + 3 * W));
      }
      BLH_TRY(launch_bn_apply_h2(s, true, ws.Z[i], sv + 2 * W, sv + 3 * W, nullptr, nullptr, nullptr,
                                 nullptr, skip, ws.A[i], ws.keep[i], batch, W,
                                 layer_drop(ctx, drop, i, batch, W), nbt + i));
    } else {
      DropoutSrc none{nullptr, 0, 0, 0, 0, nullptr};
      BLH_TRY(launch_bn_apply_h2(s, false, ws.Z[i], nullptr, nullptr, params + h.gamma, params + h.beta,
                                 rm, rv, skip, ws.A[i], nullptr, batch, W, none, nullptr));
    }
  }
  // decode (model/bilinear.py:39) fused with nn.MSELoss (train_bilinear.py:78) when a target is given:
  // skinny.hip's purpose-built kernel reads A once and writes pred, dpred (fp32 and bf16), the loss
  // partials and the decode-bias partials; *loss_nparts = their row count (0: the generic path ran)
  if (loss_nparts) *loss_nparts = 0;
  if (decode_fwd_supported(batch, W, OF)) {
    int np = 0;
    BLH_TRY(launch_decode_fwd_mse_h(s, ws.A[nh - 1], ws.wsh + L.dec_w, params + L.dec_b, target, pred,
                                    target ? ws.dpred : nullptr, target ? ws.dpredh : nullptr,
                                    target ? ws.loss_part : nullptr, target ? ws.dec_bias_part : nullptr,
                                    batch, W, OF, mse_scale, &np));
    if (loss_nparts && target) *loss_nparts = np;
    return BLH_OK;
  }
  GemmParamsH g{};   // (shapes the skinny kernel does not take: N = 48 as one ragged column tile)
  g.A = ws.A[nh - 1]; g.lda = W;
  g.B = ws.wsh + L.dec_w; g.ldb = W;
  g.C = pred; g.ldc = OF;
  g.M = (int)batch; g.N = OF; g.K = W; g.k_per_split = W;
  g.bias = params + L.dec_b;
  return launch_gemm_bf16s(s, ROWK, ROWK, EPI_BIAS, false, g, 1);
}

// dW = dZ^T act (both bf16, reduction over the batch split into fp32 slabs), summed into `out`
static int wgrad_h(hipStream_t s, const uint16_t* dZ, int64_t ld_dz, int M, const uint16_t* act,
                   int64_t ld_act, int N, int64_t batch, float* slabs, float* out) {
  const Splits sp = wgrad_plan_h(M, N, batch);
  GemmParamsH g{};
  g.A = dZ; g.lda = ld_dz; g.B = act; g.ldb = ld_act;
  g.M = M; g.N = N; g.K = (int)batch; g.k_per_split = sp.k_per; g.ldc = N;
  if (sp.splits == 1) {
    g.C = out;
    return launch_gemm_bf16s(s, KROW, KROW, EPI_STORE, false, g, 1);
  }
  g.C = slabs; g.c_split_stride = (int64_t)M * N;
  BLH_TRY(launch_gemm_bf16s(s, KROW, KROW, EPI_STORE, false, g, sp.splits));
  return launch_sum_slabs(s, slabs, (int64_t)M * N, sp.splits, out);
}

// dec_bias_S > 0: the forward ran the fused decode + MSE kernel: ws.dpredh and the decode-bias
// partials (dec_bias_S rows of ws.dec_bias_part) are already there
static int backward_h(blh_context* ctx, const blh_model_desc* d, hipStream_t s, const float* params,
                      const blh_dropout* drop, const WorkspaceH& ws, const float* dpred,
                      float* grads, int64_t batch, blh_grad_ready_fn on_ready, void* user,
                      int dec_bias_S = 0) {
  const ArenaLayout L = make_layout(d);
  const int nh = (int)L.heavy.size();
  const int W = d->width, OF = d->out_features, IF = d->in_features;
  const int chunks = ew_num_row_chunks_h(batch);
  // BLH_OPT_LATE_FORK auto, as in backward_impl: early when the data-gradient launch is one round of
  // workgroups (256 of the 256x256 kernel, 512 of the 128x128 one; configs[2], configs[3] per GPU),
  // late when it is several (configs[4]: 7.78 against 7.91 ms)
  bool late_policy = ctx->late_fork != 0;
  if (ctx->late_fork == 2) {
    GemmParamsH gp{};
    gp.M = (int)batch; gp.N = W; gp.K = W; gp.k_per_split = W; gp.lda = gp.ldb = gp.ldc = W;
    const int tile = gemm_bf16s_pick_tile(ROWK, KROW, true, gp, 1);
    // (the big-tile kernels hold a CU alone; two workgroups of the 128 x 128 kernel share one)
    late_policy = ceil_div(batch, gemm_bf16s_tile_rows(tile)) * ceil_div(W, gemm_bf16s_tile_cols(tile)) >
                  (tile == H_TILE_128 ? 512 : 256);
  }
  // Two streams as in backward_impl: every weight-gradient GEMM (+ its slab sum) runs on the
  // context's side stream — in order there, so they share one slab buffer — forked behind the
  // data-gradient GEMM of its stage (BLH_OPT_LATE_FORK) or behind bn_bwd_apply; one join at the end.
  tl_stop_event = nullptr;
  const bool two = ctx->two_stream && !ctx->sync.fn;   // (SyncBN: the exchanges are enqueued on `s`)
  hipStream_t s2 = two ? ctx->s2 : s;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(s, &cap);
  const bool attach = two && cap == hipStreamCaptureStatusNone;
  auto arm_fork = [&](int idx) { if (attach) tl_stop_event = ctx->ev_dz[idx]; };
  auto fork_wait = [&](int idx, bool attached) -> int {   // s2 continues behind the last kernel of s
    if (!two) return BLH_OK;
    if (!(attached && attach)) BLH_HIP_TRY(hipEventRecord(ctx->ev_dz[idx], s));
    BLH_HIP_TRY(hipStreamWaitEvent(s2, ctx->ev_dz[idx], 0));
    return BLH_OK;
  };
  // a reported range: weight gradient on the side stream, bias / gamma / beta on the main one;
  // the side stream waits for the main one, so the range is complete ON THE SIDE STREAM
  auto ready = [&](int idx, int64_t off, int64_t cnt) -> int {
    if (!on_ready) return BLH_OK;
    if (!two && ctx->two_stream) {   // (SyncBN call of a two-stream context: produced on `s`)
      BLH_HIP_TRY(hipEventRecord(ctx->ev_r[idx], s));
      BLH_HIP_TRY(hipStreamWaitEvent(ctx->s2, ctx->ev_r[idx], 0));
    }
    // two streams: every part of the range was produced on the side stream behind the stage's fork
    // (weight gradient, bias reduction) or on the main stream in front of it (gamma / beta)
    on_ready(user, off, cnt);
    return BLH_OK;
  };
  // decode: dA_last = dP W_d (carries the first fork), dW = dP^T A_last, db = colsum(dP)
  if (dec_bias_S == 0) BLH_TRY(launch_cast_f32_bf16(s, dpred, ws.dpredh, batch * OF));
  {
    GemmParamsH g{};
    g.A = ws.dpredh; g.lda = OF;
    g.B = ws.wsh + L.dec_w; g.ldb = W;
    g.C = ws.G0; g.ldc = W;
    g.M = (int)batch; g.N = W; g.K = OF; g.k_per_split = OF;
    arm_fork(nh);
    BLH_TRY(launch_gemm_bf16s(s, ROWK, KROW, EPI_STORE, true, g, 1));
    tl_stop_event = nullptr;
  }
  BLH_TRY(fork_wait(nh, true));
  BLH_TRY(wgrad_h(s2, ws.dpredh, OF, OF, ws.A[nh - 1], W, W, batch, ws.slabs, grads + L.dec_w));
  if (dec_bias_S == 0) BLH_TRY(launch_colsum(on_ready ? s2 : s, dpred, batch, OF, OF, ws.colsum_part, grads + L.dec_b));
  else if (on_ready) BLH_TRY(launch_colreduce(s2, ws.dec_bias_part, dec_bias_S, OF, OF, grads + L.dec_b));
  BLH_TRY(ready(nh, L.dec_w, L.total - L.dec_w));
  // Weight gradients of the hidden stages: batched launches of the 256 x 256 kernel (api_layout.h:
  // wgrad_batched_plan_h) instead of one launch per stage.  Without a bucket hook: ONE group, all hidden
  // stages, on the main stream after the loop.  With a hook (data parallel): groups of four stages from
  // the top, each launched on the side stream when its lowest stage has its dZ, so that the first ranges
  // are reported — and their all-reduce starts — after half of a four-block backward.
  struct WGroup { int lo, hi; Splits plan; };
  std::vector<WGroup> wgroups;
  if (nh - 1 >= 2) {
    if (on_ready == nullptr) {
      wgroups.push_back(WGroup{1, nh - 1, wgrad_batched_plan_h(W, batch, nh - 1)});
    } else {
      for (int hi = nh - 1; hi >= 1; hi -= WGRAD_HOOK_GROUP) {
        const int lo = std::max(1, hi - (WGRAD_HOOK_GROUP - 1));
        wgroups.push_back(WGroup{lo, hi, wgrad_batched_plan_h(W, batch, hi - lo + 1)});
      }
    }
  }
  auto group_of = [&](int i) -> const WGroup* {
    for (const WGroup& g : wgroups)
      if (g.lo <= i && i <= g.hi && g.plan.splits > 0) return &g;
    return nullptr;
  };
  // dW_k = dZ_k^T A_{k-1} for k = lo .. hi in one launch: the stages' dZ, A and gradient tensors lie one
  // fixed stride apart (carve_h, make_layout)
  auto launch_group = [&](const WGroup& grp, hipStream_t st) -> int {
    const int items = grp.hi - grp.lo + 1;
    if (nh < 3 || (ws.dZ[2] - ws.dZ[1]) != (ws.A[1] - ws.A[0])) return BLH_ERR_SHAPE;
    const int64_t gstride = L.heavy[2].w - L.heavy[1].w;
    for (int k = 2; k < nh; ++k)
      if (L.heavy[k].w - L.heavy[k - 1].w != gstride) return BLH_ERR_SHAPE;
    GemmParamsH g{};
    g.A = ws.dZ[grp.lo]; g.lda = W; g.B = ws.A[grp.lo - 1]; g.ldb = W;
    g.M = W; g.N = W; g.K = (int)batch; g.k_per_split = grp.plan.k_per; g.ldc = W;
    g.batch_splits = grp.plan.splits;
    g.a_batch_stride = ws.dZ[2] - ws.dZ[1];
    g.b_batch_stride = ws.A[1] - ws.A[0];
    float* out = grads + L.heavy[grp.lo].w;
    if (grp.plan.splits == 1) {
      g.C = out; g.c_batch_stride = gstride; g.c_split_stride = 0;
      return launch_gemm_bf16s(st, KROW, KROW, EPI_STORE, false, g, items);
    }
    g.C = ws.bslabs; g.c_split_stride = (int64_t)W * W; g.c_batch_stride = (int64_t)grp.plan.splits * W * W;
    BLH_TRY(launch_gemm_bf16s(st, KROW, KROW, EPI_STORE, false, g, items * grp.plan.splits));
    return launch_sum_slabs_batched(st, ws.bslabs, (int64_t)W * W, grp.plan.splits, items, g.c_batch_stride, out,
                                    gstride);
  };
  // a data-gradient GEMM whose output only feeds the BatchNorm backward of the stage below
  // (the second stage of a block: its output is not a skip operand; and stage 1, whose block-input gradient
  // nothing below needs) forms that stage's gated gradient dY' and the (dY' z, dY') column sums in its
  // epilogue (EPI_BN_BWD, big-tile kernels only): the stage below then skips bn_bwd_reduce_h2 and its
  // bn_bwd_apply_h2 reads no keep bits.  k9_chunks > 0: stage i's dA arrived that way, with that many partial rows.
  const bool k9_enabled = getenv("BLH_NO_K9") == nullptr;
  int k9_chunks = 0;
  for (int i = nh - 1; i >= 0; --i) {
    const HeavyOffsets& h = L.heavy[i];
    const bool first_of_block = (i >= 1) && (i % 2 == 1);
    const uint16_t* dA = first_of_block ? ws.G1 : ws.G0;
    const float* sv = ws.bn_saved[i];
    const WGroup* grp = i > 0 ? group_of(i) : nullptr;
    const int k9_in = k9_chunks;      // how this stage's dA was produced
    k9_chunks = 0;
    const bool batched_w = grp != nullptr;
    // (a hidden stage of a batched group hands nothing to the side stream, except the group's lowest
    //  stage under a hook: the group's launch goes there, behind its bn_bwd_apply)
    const bool group_fork = batched_w && on_ready != nullptr && i == grp->lo;
    const bool forks = two && (!batched_w || group_fork);
    {   // dropout: the keep bits the forward wrote (bn_bf16.hip)
      if (k9_in == 0)
        BLH_TRY(launch_bn_bwd_reduce_h2(s, dA, ws.Z[i], sv + 2 * W, sv + 3 * W, ws.keep[i], ws.bn_part, batch, W));
      BLH_TRY(launch_bn_bwd_finalize_h2(s, ws.bn_part, k9_in ? k9_in : chunks, W, sv, sv + W, grads + h.gamma,
                                        grads + h.beta));
      const float* dg = grads + h.gamma;
      const float* db = grads + h.beta;
      int64_t norm_batch = batch;
      if (ctx->sync.fn) {
        // SyncBN: the parameter gradients keep the LOCAL sums (averaged later with the rest of the
        // arena); the BatchNorm backward itself needs the sums over the global batch
        float* sb = reinterpret_cast<float*>(ws.sync_buf);
        BLH_HIP_TRY(hipMemcpyAsync(sb, grads + h.gamma, W * sizeof(float), hipMemcpyDeviceToDevice, s));
        BLH_HIP_TRY(hipMemcpyAsync(sb + W, grads + h.beta, W * sizeof(float), hipMemcpyDeviceToDevice, s));
        ctx->sync.fn(ctx->sync.user, sb, 2 * (int64_t)W, 0);
        dg = sb; db = sb + W; norm_batch = ctx->sync.global_batch;
      }
      BLH_TRY(launch_bn_bwd_apply_h2(s, dA, ws.Z[i], sv + 2 * W, sv + 3 * W, sv, sv + W, dg, db, ws.keep[i],
                                     ws.dZ[i], ws.dz_colsum_part + (int64_t)i * chunks * W, batch, W,
                                     norm_batch, k9_in > 0));
    }
    const bool late = forks && late_policy && i > 0 && !group_fork;
    if (forks && !late) BLH_TRY(fork_wait(i, false));     // behind bn_bwd_apply (marker event)
    if (group_fork) {
      for (int k = grp->hi; k >= grp->lo; --k)
        BLH_TRY(launch_colreduce(s2, ws.dz_colsum_part + (int64_t)k * chunks * W, chunks, W, W,
                                 grads + L.heavy[k].b));
      BLH_TRY(launch_group(*grp, s2));
      for (int k = grp->hi; k >= grp->lo; --k) {
        const int64_t end = (k + 1 < nh) ? L.heavy[k + 1].w : L.dec_w;
        BLH_TRY(ready(k, L.heavy[k].w, end - L.heavy[k].w));
      }
    }
    // (data parallel: the bucket hook needs this stage's bias gradient now — on the side stream, in
    //  front of the stage's weight gradient: nothing on the main stream waits for it; otherwise all
    //  stages are reduced by one launch after the loop, as in backward_impl)
    auto bias_now = [&]() -> int {
      return on_ready ? launch_colreduce(s2, ws.dz_colsum_part + (int64_t)i * chunks * W, chunks, W, W,
                                         grads + h.b)
                      : BLH_OK;
    };
    if (!late && !batched_w) BLH_TRY(bias_now());
    if (i > 0) {
      GemmParamsH g{};
      g.A = ws.dZ[i]; g.lda = W;
      g.B = ws.wsh + h.w; g.ldb = W;
      g.M = (int)batch; g.N = W; g.K = W; g.k_per_split = W; g.ldc = W;
      if (late) arm_fork(i);
      // K9: this GEMM's output is only read by the BatchNorm backward of stage i - 1
      const int tile = gemm_bf16s_pick_tile(ROWK, KROW, true, g, 1);
      const int64_t k9_rows = ceil_div(batch, gemm_bf16s_tile_rows(tile));
      // (the addend form — stage 1 only — spills in the 256 x 256 kernel, which holds 128 accumulator registers
      //  through its epilogue: there stage 0 keeps the streaming reduction)
      const bool k9 = k9_enabled && tile != H_TILE_128 && k9_rows <= chunks &&
                      (!first_of_block || (i == 1 && tile == H_TILE_128x256));
      if (k9) {
        const float* svd = ws.bn_saved[i - 1];
        g.bn_z = ws.Z[i - 1]; g.ldz = W; g.bn_keep = ws.keep[i - 1];
        g.bn_scale = svd + 2 * W; g.bn_shift = svd + 3 * W; g.stat_part = ws.bn_part;
        k9_chunks = (int)k9_rows;
      }
      if (first_of_block) {   // d(block input) = dZ W + d(block output), in place in G0
        g.C = ws.G0; g.addend = ws.G0; g.ldadd = W;
        BLH_TRY(launch_gemm_bf16s(s, ROWK, KROW, k9 ? EPI_BN_BWD_ADD : EPI_ADD, true, g, 1));
      } else {
        g.C = ws.G1;
        BLH_TRY(launch_gemm_bf16s(s, ROWK, KROW, k9 ? EPI_BN_BWD : EPI_STORE, true, g, 1));
      }
      tl_stop_event = nullptr;
      if (late) {
        BLH_TRY(fork_wait(i, true));
        BLH_TRY(bias_now());
      }
      if (!batched_w) BLH_TRY(wgrad_h(s2, ws.dZ[i], W, W, ws.A[i - 1], W, W, batch, ws.slabs, grads + h.w));
    } else {
      BLH_TRY(wgrad_h(s2, ws.dZ[0], W, W, ws.xh, IF, IF, batch, ws.slabs, grads + h.w));
    }
    if (on_ready && !batched_w) {
      const int64_t end = (i + 1 < nh) ? L.heavy[i + 1].w : L.dec_w;
      BLH_TRY(ready(i, h.w, end - h.w));
    }
  }
  if (on_ready == nullptr && !wgroups.empty() && wgroups[0].plan.splits > 0) {
    BLH_TRY(launch_group(wgroups[0], s));   // (the stage-0 and decode weight gradients, side stream, run beside it)
  }
  if (!on_ready) {
    int64_t offs[32];
    if (nh > 32) return BLH_ERR_SHAPE;
    for (int i = 0; i < nh; ++i) offs[i] = L.heavy[i].b;
    BLH_TRY(launch_bias_colreduce(s, ws.dz_colsum_part, (int64_t)chunks * W, chunks, W, nh, offs, grads,
                                  dec_bias_S > 0 ? ws.dec_bias_part : nullptr, dec_bias_S, OF, L.dec_b));
  }
  if (two) {   // join: the side stream is in order, its last kernel is stage 0's slab sum
    BLH_HIP_TRY(hipEventRecord(ctx->ev_w[0], s2));
    BLH_HIP_TRY(hipStreamWaitEvent(s, ctx->ev_w[0], 0));
  }
  return BLH_OK;
}

}  // namespace blh

using namespace blh;

// =============================================================== C ABI =======
extern "C" {

const char* blh_status_string(int status) {
  switch (status) {
    case BLH_OK: return "ok";
    case BLH_ERR_INVALID_ARGUMENT: return "invalid argument";
    case BLH_ERR_SHAPE: return "unsupported shape";
    case BLH_ERR_HIP: return "HIP runtime error";
    case BLH_ERR_WORKSPACE: return "workspace too small";
  }
  return "unknown status";
}

int blh_last_hip_error(void) { return g_last_hip_error; }
int blh_abi_version(void) { return BLH_ABI_VERSION; }

int blh_context_create(blh_context** out) {
  if (!out) return BLH_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  blh_context* c = new (std::nothrow) blh_context();
  if (!c) return BLH_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < BLH_CTX_EVENTS; ++i) c->ev_dz[i] = c->ev_w[i] = c->ev_r[i] = nullptr;
  auto fail = [&](hipError_t e) {
    g_last_hip_error = (int)e;
    blh_context_destroy(c);
    return BLH_ERR_HIP;
  };
  hipError_t e = hipGetDevice(&c->device);
  if (e != hipSuccess) return fail(e);
  // (measured and not kept: a lowest-priority side stream, and s_setprio 2 in the data-gradient
  //  GEMM: neither moves the 50/50 split of a CU's matrix pipes between two co-resident GEMMs)
  // The side stream is created at the LOWEST priority — not for the priority (measured: it does not
  // move the split of a CU between two co-resident GEMMs, nor any step time of the fused path) but for
  // the hardware queue.  HIP maps the streams of one priority level onto a small pool of hardware
  // queues (4 by default) and streams that share one run in submission order.  A process that has
  // created an RCCL communicator through torch holds some 64 normal- and high-priority pool streams;
  // a normal-priority side stream then lands on the main stream's queue or on the collective stream's,
  // and the "two-stream" backward runs serialised: 1.26-1.40 ms against 1.07-1.14 ms for the
  // data-parallel step at configs[1] (profiles/r03_dp_overhead.md).  Nothing else uses the lowest level.
  // ... and it is ONE stream per device and process, shared by every context of that device
  // (reference-counted): each additional hardware queue a process keeps busy brings it closer to the
  // point where the queue scheduler starts time-slicing queues, and kernels of ALL of them then run
  // 1.5-5x longer (a second context + a high-priority compute stream + RCCL's stream was enough:
  // 2.27 against 1.14 ms per data-parallel step).  Contexts are used one call at a time; two threads
  // driving two contexts of one device merely take turns on the side stream.
  if ((e = side_stream_acquire(c->device, &c->s2)) != hipSuccess) return fail(e);
  for (int i = 0; i < BLH_CTX_EVENTS; ++i) {
    if ((e = hipEventCreateWithFlags(&c->ev_dz[i], hipEventDisableTiming)) != hipSuccess) return fail(e);
    if ((e = hipEventCreateWithFlags(&c->ev_w[i], hipEventDisableTiming)) != hipSuccess) return fail(e);
    if ((e = hipEventCreateWithFlags(&c->ev_r[i], hipEventDisableTiming)) != hipSuccess) return fail(e);
  }
  // A/B switches for experiments (the documented way is blh_context_set_option)
  if ((e = hipMalloc(reinterpret_cast<void**>(&c->grid_bar), 64)) != hipSuccess) return fail(e);
  if ((e = hipMemset(c->grid_bar, 0, 64)) != hipSuccess) return fail(e);
  c->two_stream = getenv("BLH_ONE_STREAM") == nullptr;
  c->defer_slabs = getenv("BLH_DEFER_SLABS") != nullptr;
  c->late_fork = getenv("BLH_EARLY_FORK") ? 0 : (getenv("BLH_LATE_FORK") ? 1 : 2);
  *out = c;
  return BLH_OK;
}

int blh_context_destroy(blh_context* c) {
  if (!c) return BLH_OK;
  for (int i = 0; i < BLH_CTX_EVENTS; ++i) {
    if (c->ev_dz[i]) (void)hipEventDestroy(c->ev_dz[i]);
    if (c->ev_w[i]) (void)hipEventDestroy(c->ev_w[i]);
    if (c->ev_r[i]) (void)hipEventDestroy(c->ev_r[i]);
  }
  if (c->s2) side_stream_release(c->device);
  if (c->grid_bar) (void)hipFree(c->grid_bar);
  delete c;
  return BLH_OK;
}

int blh_context_set_option(blh_context* c, int32_t option, int32_t value) {
  if (!c) return BLH_ERR_INVALID_ARGUMENT;
  switch (option) {
    case BLH_OPT_TWO_STREAM: c->two_stream = value != 0; return BLH_OK;
    case BLH_OPT_DEFER_SLABS: c->defer_slabs = value != 0; return BLH_OK;
    case BLH_OPT_LATE_FORK:
      if (value < 0 || value > 2) return BLH_ERR_INVALID_ARGUMENT;
      c->late_fork = value;
      return BLH_OK;
    case BLH_OPT_PERSISTENT_SHADOW:
      c->persistent_shadow = value != 0;
      c->shadow_params = c->shadow_ws = nullptr;
      return BLH_OK;
  }
  return BLH_ERR_INVALID_ARGUMENT;
}

int blh_context_get_option(const blh_context* c, int32_t option) {
  if (!c) return BLH_ERR_INVALID_ARGUMENT;
  switch (option) {
    case BLH_OPT_TWO_STREAM: return c->two_stream ? 1 : 0;
    case BLH_OPT_DEFER_SLABS: return c->defer_slabs ? 1 : 0;
    case BLH_OPT_LATE_FORK: return c->late_fork;
    case BLH_OPT_PERSISTENT_SHADOW: return c->persistent_shadow ? 1 : 0;
  }
  return BLH_ERR_INVALID_ARGUMENT;
}

int64_t blh_context_grid_barrier_timeouts(blh_context* c) {
  if (!c || !c->grid_bar) return BLH_ERR_INVALID_ARGUMENT;
  uint32_t words[3] = {0, 0, 0};
  BLH_HIP_TRY(hipMemcpy(words, c->grid_bar, sizeof(words), hipMemcpyDeviceToHost));
  return (int64_t)words[2];
}

void* blh_context_side_stream(blh_context* c) {
  if (!c || !c->two_stream) return nullptr;
  c->s2 = blh::side_stream_current(c->device);
  return (void*)c->s2;
}

int32_t blh_num_heavy(const blh_model_desc* d) { return d ? 1 + 2 * d->num_blocks : 0; }

int64_t blh_param_arena_floats(const blh_model_desc* d) {
  if (check_desc(d) != BLH_OK) return check_desc(d);
  return make_layout(d).total;
}

int32_t blh_num_param_tensors(const blh_model_desc* d) {
  if (check_desc(d) != BLH_OK) return check_desc(d);
  return (int32_t)make_layout(d).tensors.size();
}

int blh_param_tensor_info(const blh_model_desc* d, int32_t index, char* name, int32_t name_cap,
                          int64_t* offset_floats, int64_t* rows, int64_t* cols) {
  BLH_TRY(check_desc(d));
  const ArenaLayout L = make_layout(d);
  if (index < 0 || index >= (int32_t)L.tensors.size() || !name || name_cap <= 0)
    return BLH_ERR_INVALID_ARGUMENT;
  const TensorInfo& t = L.tensors[index];
  snprintf(name, (size_t)name_cap, "%s", t.name);
  if (offset_floats) *offset_floats = t.offset;
  if (rows) *rows = t.rows;
  if (cols) *cols = t.cols;
  return BLH_OK;
}

int64_t blh_bn_running_floats(const blh_model_desc* d) {
  if (check_desc(d) != BLH_OK) return check_desc(d);
  return (int64_t)(1 + 2 * d->num_blocks) * 2 * d->width;
}

int64_t blh_workspace_bytes(const blh_model_desc* d, int64_t batch) {
  if (check_desc(d) != BLH_OK) return check_desc(d);
  if (batch <= 0) return BLH_ERR_INVALID_ARGUMENT;
  return d->gemm_dtype == 4 ? carve_h(d, batch, nullptr).bytes : carve(d, batch, nullptr).bytes;
}

static int check_common(blh_context* ctx, const blh_model_desc* d, const void* ws,
                        int64_t ws_bytes, int64_t batch) {
  BLH_TRY(check_ctx(ctx));
  BLH_TRY(check_desc(d));
  if (batch <= 0 || batch > (1 << 30)) return BLH_ERR_INVALID_ARGUMENT;
  if (!ws || ((uintptr_t)ws % WS_ALIGN) != 0) return BLH_ERR_INVALID_ARGUMENT;
  const int64_t need = d->gemm_dtype == 4 ? carve_h(d, batch, nullptr).bytes : carve(d, batch, nullptr).bytes;
  if (ws_bytes < need) return BLH_ERR_WORKSPACE;
  if (d->gemm_dtype == 4 && (d->width % 128 != 0 || d->in_features % 8 != 0 || d->out_features % 8 != 0))
    return BLH_ERR_SHAPE;
  return BLH_OK;
}

static int check_drop(const blh_dropout* drop) {
  if (!drop) return BLH_ERR_INVALID_ARGUMENT;
  if (!drop->keep_mask && (drop->row_offset % 32) != 0) return BLH_ERR_SHAPE;
  return BLH_OK;
}

int blh_forward_train(blh_context* ctx, const blh_model_desc* d, void* stream, const float* params,
                      float* bn_running, int64_t* bn_nbt, const float* x,
                      const blh_dropout* drop, float momentum, void* workspace,
                      int64_t workspace_bytes, float* pred, int64_t batch) {
  BLH_TRY(check_common(ctx, d, workspace, workspace_bytes, batch));
  BLH_TRY(check_drop(drop));
  if (!params || !bn_running || !bn_nbt || !x || !pred) return BLH_ERR_INVALID_ARGUMENT;
  if (batch < 2) return BLH_ERR_SHAPE;   // BatchNorm1d needs > 1 value per channel in training
  if (d->gemm_dtype == 4)
    return forward_h(ctx, d, (hipStream_t)stream, params, bn_running, bn_nbt, x, drop, momentum,
                     carve_h(d, batch, workspace), pred, batch, true);
  const Workspace ws = carve(d, batch, workspace);
  return forward_impl(ctx, d, (hipStream_t)stream, params, bn_running, bn_nbt, x, drop, momentum, ws,
                      pred, batch, true, nullptr, 0.f, nullptr, nullptr);
}

int blh_forward_train_loss(blh_context* ctx, const blh_model_desc* d, void* stream, const float* params,
                           float* bn_running, int64_t* bn_nbt, const float* x, const float* target,
                           const blh_dropout* drop, float momentum, void* workspace,
                           int64_t workspace_bytes, float* pred, float* loss_out, int64_t batch) {
  BLH_TRY(check_common(ctx, d, workspace, workspace_bytes, batch));
  BLH_TRY(check_drop(drop));
  if (!params || !bn_running || !bn_nbt || !x || !target || !pred || !loss_out) return BLH_ERR_INVALID_ARGUMENT;
  if (batch < 2) return BLH_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const double denom = (double)batch * d->out_features;
  int nparts = 0;
  ctx->loss_batch = 0;
  if (d->gemm_dtype == 4) {
    const WorkspaceH wh = carve_h(d, batch, workspace);
    int dec_S = 0;
    BLH_TRY(forward_h(ctx, d, s, params, bn_running, bn_nbt, x, drop, momentum, wh, pred, batch, true, false,
                      target, (float)(2.0 / denom), &dec_S));
    if (dec_S > 0) nparts = dec_S;
    else BLH_TRY(launch_mse(s, pred, target, batch * d->out_features, (float)(2.0 / denom), wh.dpred,
                            wh.loss_part, &nparts));
    BLH_TRY(launch_loss_finalize(s, wh.loss_part, nparts, denom, loss_out));
    ctx->loss_batch = batch; ctx->loss_nparts = dec_S;
    return BLH_OK;
  }
  const Workspace ws = carve(d, batch, workspace);
  BLH_TRY(forward_impl(ctx, d, s, params, bn_running, bn_nbt, x, drop, momentum, ws, pred, batch, true,
                       target, (float)(2.0 / denom), ws.loss_part, &nparts));
  BLH_TRY(launch_loss_finalize(s, ws.loss_part, nparts, denom, loss_out));
  ctx->loss_batch = batch; ctx->loss_nparts = nparts;
  return BLH_OK;
}

int blh_forward_eval(blh_context* ctx, const blh_model_desc* d, void* stream, const float* params,
                     const float* bn_running, const float* x, void* workspace,
                     int64_t workspace_bytes, float* pred, int64_t batch) {
  BLH_TRY(check_common(ctx, d, workspace, workspace_bytes, batch));
  if (!params || !bn_running || !x || !pred) return BLH_ERR_INVALID_ARGUMENT;
  blh_dropout none{nullptr, 0, 0, 0, 0, 0};
  if (d->gemm_dtype == 4)
    return forward_h(ctx, d, (hipStream_t)stream, params, const_cast<float*>(bn_running), nullptr, x,
                     &none, 0.f, carve_h(d, batch, workspace), pred, batch, false);
  const Workspace ws = carve(d, batch, workspace);
  return forward_impl(ctx, d, (hipStream_t)stream, params, const_cast<float*>(bn_running), nullptr, x,
                      &none, 0.f, ws, pred, batch, false, nullptr, 0.f, nullptr, nullptr);
}

int blh_mse_loss_grad(void* stream, const float* pred, const float* target, int64_t batch,
                      int64_t out_features, double loss_denominator, float grad_scale,
                      float* loss_out, float* dpred, void* workspace, int64_t workspace_bytes) {
  if (!pred || !target || !loss_out || !dpred || !workspace) return BLH_ERR_INVALID_ARGUMENT;
  if (batch <= 0 || out_features <= 0 || loss_denominator <= 0) return BLH_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < SCRATCH_BYTES) return BLH_ERR_WORKSPACE;
  const Scratch sc = carve_scratch(workspace);
  int nparts = 0;
  const float scale = (float)(2.0 * (double)grad_scale / loss_denominator);
  BLH_TRY(launch_mse((hipStream_t)stream, pred, target, batch * out_features, scale, dpred,
                     sc.loss_part, &nparts));
  return launch_loss_finalize((hipStream_t)stream, sc.loss_part, nparts, loss_denominator,
                              loss_out);
}

int blh_backward(blh_context* ctx, const blh_model_desc* d, void* stream, const float* params, const float* x,
                 const blh_dropout* drop, void* workspace, int64_t workspace_bytes,
                 const float* dpred, float* grads, int64_t batch, blh_grad_ready_fn on_ready,
                 void* user) {
  BLH_TRY(check_common(ctx, d, workspace, workspace_bytes, batch));
  BLH_TRY(check_drop(drop));
  if (!params || !x || !grads) return BLH_ERR_INVALID_ARGUMENT;
  // dpred == NULL: the loss gradient blh_forward_train_loss left in the workspace
  const bool from_loss = dpred == nullptr;
  if (from_loss && ctx->loss_batch != batch) return BLH_ERR_INVALID_ARGUMENT;
  const int loss_nparts = ctx->loss_nparts;
  ctx->loss_batch = 0;
  if (d->gemm_dtype == 4) {
    const WorkspaceH wh = carve_h(d, batch, workspace);
    return backward_h(ctx, d, (hipStream_t)stream, params, drop, wh, from_loss ? wh.dpred : dpred, grads,
                      batch, on_ready, user, from_loss ? loss_nparts : 0);
  }
  const Workspace ws = carve(d, batch, workspace);
  if (from_loss && loss_nparts > 0) {   // decode-bias partials of the fused decode kernel
    const FusedBackward fb{loss_nparts, nullptr, nullptr, nullptr};
    return backward_impl(ctx, d, (hipStream_t)stream, params, x, drop, ws, ws.dpred, grads, batch,
                         on_ready, user, &fb);
  }
  return backward_impl(ctx, d, (hipStream_t)stream, params, x, drop, ws, from_loss ? ws.dpred : dpred,
                       grads, batch, on_ready, user);
}

int blh_clip_adam_step(void* stream, float* params, float* grads, float* exp_avg,
                       float* exp_avg_sq, int64_t count, const blh_adam_hyper* hyper,
                       void* workspace, int64_t workspace_bytes, float* stats_out) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !hyper || !workspace || count <= 0)
    return BLH_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < SCRATCH_BYTES) return BLH_ERR_WORKSPACE;
  const Scratch sc = carve_scratch(workspace);
  int nparts = 0;
  BLH_TRY(launch_sumsq((hipStream_t)stream, grads, count, sc.sumsq_part, &nparts));
  return launch_clip_adam((hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, count, *hyper,
                          sc.sumsq_part, nparts, stats_out);
}

int blh_clip_adam_step_bf16(void* stream, float* params, const uint16_t* grads_bf16, float grad_scale,
                            float* grads, float* exp_avg, float* exp_avg_sq, int64_t count,
                            const blh_adam_hyper* hyper, void* workspace, int64_t workspace_bytes,
                            float* stats_out) {
  if (!params || !grads_bf16 || !grads || !exp_avg || !exp_avg_sq || !hyper || !workspace || count <= 0)
    return BLH_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < SCRATCH_BYTES) return BLH_ERR_WORKSPACE;
  const Scratch sc = carve_scratch(workspace);
  int nparts = 0;
  BLH_TRY(launch_sumsq_bf16((hipStream_t)stream, grads_bf16, count, grad_scale, sc.sumsq_part, &nparts));
  return launch_clip_adam_bf16((hipStream_t)stream, params, grads_bf16, grad_scale, grads, exp_avg, exp_avg_sq,
                               count, *hyper, sc.sumsq_part, nparts, stats_out);
}

int blh_refresh_param_shadow(blh_context* ctx, const blh_model_desc* d, void* stream, const float* params,
                             void* workspace, int64_t workspace_bytes, int64_t batch) {
  BLH_TRY(check_common(ctx, d, workspace, workspace_bytes, batch));
  if (!params || d->gemm_dtype != 4) return BLH_ERR_INVALID_ARGUMENT;
  const WorkspaceH wh = carve_h(d, batch, workspace);
  BLH_TRY(launch_cast_f32_bf16((hipStream_t)stream, params, wh.wsh, make_layout(d).total));
  if (ctx->persistent_shadow) { ctx->shadow_params = params; ctx->shadow_ws = workspace; }
  return BLH_OK;
}

int blh_clip_grad_norm(void* stream, float* grads, int64_t count, float max_norm, void* workspace,
                       int64_t workspace_bytes, float* stats_out) {
  if (!grads || !workspace || count <= 0 || !(max_norm > 0.f)) return BLH_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < SCRATCH_BYTES) return BLH_ERR_WORKSPACE;
  const Scratch sc = carve_scratch(workspace);
  int nparts = 0;
  BLH_TRY(launch_sumsq((hipStream_t)stream, grads, count, sc.sumsq_part, &nparts));
  return launch_clip_scale((hipStream_t)stream, grads, count, max_norm, sc.sumsq_part, nparts,
                           stats_out);
}

int blh_train_step(blh_context* ctx, const blh_model_desc* d, void* stream, float* params, float* grads,
                   float* exp_avg, float* exp_avg_sq, float* bn_running, int64_t* bn_nbt,
                   const float* x, const float* target, const blh_dropout* drop, float momentum,
                   const blh_adam_hyper* hyper, void* workspace, int64_t workspace_bytes,
                   float* pred, float* loss_out, float* stats_out, int64_t batch) {
  BLH_TRY(check_common(ctx, d, workspace, workspace_bytes, batch));
  BLH_TRY(check_drop(drop));
  if (!params || !grads || !exp_avg || !exp_avg_sq || !bn_running || !bn_nbt || !x || !target ||
      !hyper || !pred || !loss_out)
    return BLH_ERR_INVALID_ARGUMENT;
  if (batch < 2) return BLH_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const double denom = (double)batch * d->out_features;
  int nparts = 0;
  if (d->gemm_dtype == 4) {   // bf16 storage: forward, MSE, backward, norm, clip + Adam
    const WorkspaceH wh = carve_h(d, batch, workspace);
    const int64_t count = make_layout(d).total;
    int np = 0;
    const bool keep = ctx->persistent_shadow;
    const bool valid = keep && ctx->shadow_params == params && ctx->shadow_ws == workspace;
    int dec_S = 0;
    BLH_TRY(forward_h(ctx, d, s, params, bn_running, bn_nbt, x, drop, momentum, wh, pred, batch, true, valid,
                      target, (float)(2.0 / denom), &dec_S));
    if (dec_S > 0) nparts = dec_S;
    else BLH_TRY(launch_mse(s, pred, target, batch * d->out_features, (float)(2.0 / denom), wh.dpred,
                            wh.loss_part, &nparts));
    BLH_TRY(backward_h(ctx, d, s, params, drop, wh, wh.dpred, grads, batch, nullptr, nullptr, dec_S));
    BLH_TRY(launch_sumsq(s, grads, count, wh.sumsq_part, &np));
    BLH_TRY(launch_clip_adam(s, params, grads, exp_avg, exp_avg_sq, count, *hyper, wh.sumsq_part, np,
                             stats_out, LossFinish{wh.loss_part, nparts, denom, loss_out},
                             keep ? wh.wsh : nullptr));
    if (keep) { ctx->shadow_params = params; ctx->shadow_ws = workspace; }
    return BLH_OK;
  }
  const Workspace ws = carve(d, batch, workspace);
  BLH_TRY(forward_impl(ctx, d, s, params, bn_running, bn_nbt, x, drop, momentum, ws, pred, batch, true,
                       target, (float)(2.0 / denom), ws.loss_part, &nparts));
  int np = 0;
  double* sq_src = ws.sumsq_part;
  const FusedBackward fb{nparts, ws.sumsq_part, &np, &sq_src};
  BLH_TRY(backward_impl(ctx, d, s, params, x, drop, ws, ws.dpred, grads, batch, nullptr, nullptr, &fb));
  const int64_t count = make_layout(d).total;
  return launch_clip_adam(s, params, grads, exp_avg, exp_avg_sq, count, *hyper, sq_src, np,
                          stats_out, LossFinish{ws.loss_part, nparts, denom, loss_out});
}

int blh_forward_train_sync(blh_context* ctx, const blh_model_desc* d, void* stream, const float* params,
                           float* bn_running, int64_t* bn_nbt, const float* x,
                           const blh_dropout* drop, float momentum, void* workspace,
                           int64_t workspace_bytes, float* pred, int64_t batch,
                           int64_t global_batch, blh_sync_fn sync, void* user) {
  if (!ctx || !sync || global_batch < batch) return BLH_ERR_INVALID_ARGUMENT;
  struct Guard {
    blh_context* c;
    Guard(blh_context* c_, blh_sync_fn f, void* u, int64_t g) : c(c_) { c->sync = SyncCtx{f, u, g}; }
    ~Guard() { c->sync = SyncCtx{nullptr, nullptr, 0}; }
  } guard(ctx, sync, user, global_batch);
  return blh_forward_train(ctx, d, stream, params, bn_running, bn_nbt, x, drop, momentum, workspace,
                           workspace_bytes, pred, batch);
}

int blh_forward_train_loss_sync(blh_context* ctx, const blh_model_desc* d, void* stream, const float* params,
                                float* bn_running, int64_t* bn_nbt, const float* x, const float* target,
                                const blh_dropout* drop, float momentum, void* workspace,
                                int64_t workspace_bytes, float* pred, float* loss_out, int64_t batch,
                                int64_t global_batch, blh_sync_fn sync, void* user) {
  if (!ctx || !sync || global_batch < batch) return BLH_ERR_INVALID_ARGUMENT;
  struct Guard {
    blh_context* c;
    Guard(blh_context* c_, blh_sync_fn f, void* u, int64_t g) : c(c_) { c->sync = SyncCtx{f, u, g}; }
    ~Guard() { c->sync = SyncCtx{nullptr, nullptr, 0}; }
  } guard(ctx, sync, user, global_batch);
  return blh_forward_train_loss(ctx, d, stream, params, bn_running, bn_nbt, x, target, drop, momentum,
                                workspace, workspace_bytes, pred, loss_out, batch);
}

int blh_backward_sync(blh_context* ctx, const blh_model_desc* d, void* stream, const float* params, const float* x,
                      const blh_dropout* drop, void* workspace, int64_t workspace_bytes,
                      const float* dpred, float* grads, int64_t batch, blh_grad_ready_fn on_ready,
                      void* user, int64_t global_batch, blh_sync_fn sync, void* sync_user) {
  if (!ctx || !sync || global_batch < batch) return BLH_ERR_INVALID_ARGUMENT;
  struct Guard {
    blh_context* c;
    Guard(blh_context* c_, blh_sync_fn f, void* u, int64_t g) : c(c_) { c->sync = SyncCtx{f, u, g}; }
    ~Guard() { c->sync = SyncCtx{nullptr, nullptr, 0}; }
  } guard(ctx, sync, sync_user, global_batch);
  return blh_backward(ctx, d, stream, params, x, drop, workspace, workspace_bytes, dpred, grads, batch,
                      on_ready, user);
}

// ---------------------------------------------------------------- single stage ----
// heavy_linear (model/bilinear.py:7-13) on its own: Linear -> BatchNorm1d -> ReLU -> Dropout.
struct HeavyWs {
  float* Z; float* dZ; float* saved; float* stat_part; float* bn_part; float* dz_part; float* slabs;
  uint32_t* keep;
  int64_t bytes;
};
static HeavyWs carve_heavy(int64_t batch, int in_f, int out_f, void* base) {
  HeavyWs w;
  char* p = (char*)base;
  int64_t off = 0;
  auto take = [&](int64_t bytes) {
    char* r = p ? p + off : nullptr;
    off += round_up(bytes, WS_ALIGN);
    return r;
  };
  const int64_t chunks = ew_num_row_chunks(batch);
  w.Z = (float*)take(batch * out_f * sizeof(float));
  w.dZ = (float*)take(batch * out_f * sizeof(float));
  w.saved = (float*)take(4 * (int64_t)out_f * sizeof(float));
  w.stat_part = (float*)take(ceil_div(batch, 64) * 2 * out_f * sizeof(float));
  w.bn_part = (float*)take(chunks * 2 * out_f * sizeof(float));
  w.dz_part = (float*)take(chunks * out_f * sizeof(float));
  const Splits sp = pick_splits(batch, ceil_div(out_f, 128) * ceil_div(in_f, 128));
  w.slabs = (float*)take((int64_t)sp.splits * out_f * in_f * sizeof(float));
  w.keep = (uint32_t*)take(ceil_div(batch, 8) * (out_f / 4) * 4);
  w.bytes = off;
  return w;
}

int64_t blh_heavy_workspace_bytes(int64_t batch, int32_t in_features, int32_t out_features) {
  if (batch <= 0 || in_features <= 0 || out_features <= 0) return BLH_ERR_INVALID_ARGUMENT;
  if (in_features % 4 != 0 || out_features % 4 != 0) return BLH_ERR_SHAPE;
  return carve_heavy(batch, in_features, out_features, nullptr).bytes;
}

int blh_heavy_forward(blh_context* ctx, void* stream, const float* a_in, const float* weight, const float* bias,
                      const float* gamma, const float* beta, float* running_mean,
                      float* running_var, int64_t* num_batches_tracked, const blh_dropout* drop,
                      float momentum, int32_t training, int32_t gemm_dtype, void* workspace,
                      int64_t workspace_bytes, float* a_out, int64_t batch, int32_t in_features,
                      int32_t out_features) {
  if (!a_in || !weight || !bias || !gamma || !beta || !running_mean || !running_var || !a_out ||
      !workspace || batch <= 0)
    return BLH_ERR_INVALID_ARGUMENT;
  if (in_features % 4 != 0 || out_features % 4 != 0) return BLH_ERR_SHAPE;
  if (training && (batch < 2 || !drop || !num_batches_tracked)) return BLH_ERR_INVALID_ARGUMENT;
  BLH_TRY(check_ctx(ctx));
  if (training) BLH_TRY(check_drop(drop));
  if (((uintptr_t)workspace % WS_ALIGN) != 0) return BLH_ERR_INVALID_ARGUMENT;
  const HeavyWs w = carve_heavy(batch, in_features, out_features, workspace);
  if (workspace_bytes < w.bytes) return BLH_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int W = out_features;
  GemmParams g{};
  g.A = a_in; g.lda = in_features; g.B = weight; g.ldb = in_features; g.C = w.Z; g.ldc = W;
  g.M = (int)batch; g.N = W; g.K = in_features; g.k_per_split = in_features;
  g.bias = bias; g.stat_part = w.stat_part;
  BLH_TRY(launch_gemm(s, TILE_128x128, ROWK, ROWK, training ? EPI_BIAS_STATS : EPI_BIAS, g, 1,
                      gemm_dtype));
  if (!training) {
    DropoutSrc none{nullptr, 0, 0, 0, 0, nullptr};
    return launch_bn_apply_f2(s, false, w.Z, nullptr, nullptr, gamma, beta, running_mean, running_var,
                              nullptr, a_out, nullptr, batch, W, none, nullptr);
  }
  float* sv = w.saved;
  BLH_TRY(launch_bn_fwd_finalize(s, w.stat_part, (int)ceil_div(batch, 128), 128, batch, W, gamma,
                                 beta, running_mean, running_var, num_batches_tracked, momentum, sv,
                                 sv + W, sv + 2 * W, sv + 3 * W));
  DropoutSrc ds{drop->keep_mask, drop->seed, drop->step, drop->row_offset, drop->layer_base, nullptr};
  return launch_bn_apply_f2(s, true, w.Z, sv + 2 * W, sv + 3 * W, nullptr, nullptr, nullptr, nullptr,
                            nullptr, a_out, w.keep, batch, W, ds, num_batches_tracked);
}

int blh_heavy_backward(blh_context* ctx, void* stream, const float* d_out, const float* a_in, const float* weight,
                       const float* gamma, const blh_dropout* drop, int32_t gemm_dtype,
                       void* workspace, int64_t workspace_bytes, float* d_weight, float* d_bias,
                       float* d_gamma, float* d_beta, float* d_in, int64_t batch,
                       int32_t in_features, int32_t out_features) {
  if (!d_out || !a_in || !weight || !gamma || !drop || !workspace || !d_weight || !d_bias ||
      !d_gamma || !d_beta || batch < 2)
    return BLH_ERR_INVALID_ARGUMENT;
  if (in_features % 4 != 0 || out_features % 4 != 0) return BLH_ERR_SHAPE;
  BLH_TRY(check_ctx(ctx));
  BLH_TRY(check_drop(drop));
  const HeavyWs w = carve_heavy(batch, in_features, out_features, workspace);
  if (workspace_bytes < w.bytes) return BLH_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int W = out_features;
  const int chunks = ew_num_row_chunks(batch);
  const float* sv = w.saved;
  // (the dropout mask of the stage is in w.keep, written by blh_heavy_forward)
  BLH_TRY(launch_bn_bwd_reduce_f2(s, d_out, w.Z, sv + 2 * W, sv + 3 * W, w.keep, w.bn_part, batch, W));
  BLH_TRY(launch_bn_bwd_finalize_h2(s, w.bn_part, chunks, W, sv, sv + W, d_gamma, d_beta));
  BLH_TRY(launch_bn_bwd_apply_f2(s, d_out, w.Z, sv + 2 * W, sv + 3 * W, sv, sv + W, d_gamma, d_beta, w.keep,
                                 w.dZ, w.dz_part, batch, W, batch));
  BLH_TRY(launch_colreduce(s, w.dz_part, chunks, W, W, d_bias));
  BLH_TRY(wgrad(gemm_dtype, s, TILE_128x128, w.dZ, W, W, a_in, in_features, in_features, batch,
                ceil_div(W, 128) * ceil_div(in_features, 128), w.slabs, d_weight, nullptr));
  if (d_in) {
    GemmParams g{};
    g.A = w.dZ; g.lda = W; g.B = weight; g.ldb = in_features; g.C = d_in; g.ldc = in_features;
    g.M = (int)batch; g.N = in_features; g.K = W; g.k_per_split = W;
    BLH_TRY(launch_gemm(s, TILE_128x128, ROWK, KROW, EPI_STORE, g, 1, gemm_dtype));
  }
  return BLH_OK;
}

int blh_mpjpe(void* stream, const float* pred, const float* target, const float* mean,
              const float* stddev, int64_t batch, int32_t joints, float* dist_out,
              const int32_t* action_ids, int32_t num_actions, double* action_sum,
              int64_t* action_count) {
  if (!pred || !target || !mean || !stddev || !dist_out || batch <= 0 || joints <= 0)
    return BLH_ERR_INVALID_ARGUMENT;
  BLH_TRY(launch_mpjpe((hipStream_t)stream, pred, target, mean, stddev, batch, joints, dist_out));
  if (action_ids) {
    if (num_actions <= 0 || !action_sum || !action_count) return BLH_ERR_INVALID_ARGUMENT;
    BLH_TRY(launch_segment_sum((hipStream_t)stream, dist_out, action_ids, batch, num_actions,
                               action_sum, action_count));
  }
  return BLH_OK;
}

int blh_context_set_step_state(blh_context* ctx, const blh_step_state* dev_state) {
  if (!ctx) return BLH_ERR_INVALID_ARGUMENT;
  ctx->step_dev = dev_state ? &dev_state->rng_step : nullptr;
  return BLH_OK;
}

int blh_clip_adam_step_captured(void* stream, float* params, float* grads, float* exp_avg,
                                float* exp_avg_sq, int64_t count, const blh_step_state* dev_state,
                                void* workspace, int64_t workspace_bytes, float* stats_out) {
  if (!params || !grads || !exp_avg || !exp_avg_sq || !dev_state || !workspace || count <= 0)
    return BLH_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < SCRATCH_BYTES) return BLH_ERR_WORKSPACE;
  const Scratch sc = carve_scratch(workspace);
  int nparts = 0;
  BLH_TRY(launch_sumsq((hipStream_t)stream, grads, count, sc.sumsq_part, &nparts));
  return launch_clip_adam_dev((hipStream_t)stream, params, grads, exp_avg, exp_avg_sq, count, dev_state,
                              sc.sumsq_part, nparts, stats_out);
}

int blh_step_state_advance(void* stream, blh_step_state* dev_state) {
  if (!dev_state) return BLH_ERR_INVALID_ARGUMENT;
  return launch_step_state_advance((hipStream_t)stream, dev_state);
}

int blh_train_step_captured(blh_context* ctx, const blh_model_desc* d, void* stream, float* params, float* grads,
                            float* exp_avg, float* exp_avg_sq, float* bn_running,
                            int64_t* bn_nbt, const float* x, const float* target,
                            const blh_dropout* drop, float momentum, blh_step_state* dev_state,
                            void* workspace, int64_t workspace_bytes, float* pred,
                            float* loss_out, float* stats_out, int64_t batch) {
  BLH_TRY(check_common(ctx, d, workspace, workspace_bytes, batch));
  BLH_TRY(check_drop(drop));
  if (!params || !grads || !exp_avg || !exp_avg_sq || !bn_running || !bn_nbt || !x || !target ||
      !dev_state || !pred || !loss_out)
    return BLH_ERR_INVALID_ARGUMENT;
  if (batch < 2) return BLH_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  const double denom = (double)batch * d->out_features;
  int nparts = 0, np = 0;
  BLH_TRY(launch_step_state_advance(s, dev_state));
  struct StepDevGuard {   // kernels of this call add dev_state->rng_step to the dropout step
    blh_context* c;
    StepDevGuard(blh_context* c_, const uint64_t* p) : c(c_) { c->step_dev = p; }
    ~StepDevGuard() { c->step_dev = nullptr; }
  } guard(ctx, &dev_state->rng_step);
  if (d->gemm_dtype == 4) {
    const WorkspaceH wh = carve_h(d, batch, workspace);
    const int64_t count = make_layout(d).total;
    // (a captured step is replayed as recorded: with BLH_OPT_PERSISTENT_SHADOW the capture holds
    //  no arena re-cast — the caller refreshes the image before the first replay and after any
    //  out-of-band parameter change with blh_refresh_param_shadow)
    const bool keep = ctx->persistent_shadow;
    int dec_S = 0;
    BLH_TRY(forward_h(ctx, d, s, params, bn_running, bn_nbt, x, drop, momentum, wh, pred, batch, true, keep,
                      target, (float)(2.0 / denom), &dec_S));
    if (dec_S > 0) nparts = dec_S;
    else BLH_TRY(launch_mse(s, pred, target, batch * d->out_features, (float)(2.0 / denom), wh.dpred,
                            wh.loss_part, &nparts));
    BLH_TRY(backward_h(ctx, d, s, params, drop, wh, wh.dpred, grads, batch, nullptr, nullptr, dec_S));
    BLH_TRY(launch_sumsq(s, grads, count, wh.sumsq_part, &np));
    return launch_clip_adam_dev(s, params, grads, exp_avg, exp_avg_sq, count, dev_state,
                                wh.sumsq_part, np, stats_out,
                                LossFinish{wh.loss_part, nparts, denom, loss_out}, keep ? wh.wsh : nullptr);
  }
  const Workspace ws = carve(d, batch, workspace);
  BLH_TRY(forward_impl(ctx, d, s, params, bn_running, bn_nbt, x, drop, momentum, ws, pred, batch, true,
                       target, (float)(2.0 / denom), ws.loss_part, &nparts));
  double* sq_src = ws.sumsq_part;
  const FusedBackward fb{nparts, ws.sumsq_part, &np, &sq_src};
  BLH_TRY(backward_impl(ctx, d, s, params, x, drop, ws, ws.dpred, grads, batch, nullptr, nullptr, &fb));
  const int64_t count = make_layout(d).total;
  return launch_clip_adam_dev(s, params, grads, exp_avg, exp_avg_sq, count, dev_state,
                              sq_src, np, stats_out,
                              LossFinish{ws.loss_part, nparts, denom, loss_out});
}

static int gemm_entry(int dtype, void* stream, const float* A, int64_t lda, int32_t a_kmajor,
                      const float* B, int64_t ldb, int32_t b_kmajor, float* C, int64_t ldc,
                      int64_t M, int64_t N, int64_t K, int32_t splits, const float* bias,
                      const float* addend, int64_t ldadd) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || splits < 1) return BLH_ERR_INVALID_ARGUMENT;
  if (bias && addend) return BLH_ERR_INVALID_ARGUMENT;
  if (splits > 1 && (bias || addend)) return BLH_ERR_INVALID_ARGUMENT;
  GemmParams g{};
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.k_per_split = (splits > 1) ? (int)round_up(ceil_div(K, splits), SPLIT_GRAIN) : (int)K;
  if (splits > 1 && (int64_t)g.k_per_split * (splits - 1) >= K) return BLH_ERR_SHAPE;
  g.c_split_stride = M * ldc;
  g.bias = bias; g.addend = addend; g.ldadd = ldadd;
  const int epi = bias ? EPI_BIAS : (addend ? EPI_ADD : EPI_STORE);
  GemmTile tile = TILE_128x128;
  if (N <= 32) tile = TILE_128x32;
  else if (N <= 64) tile = TILE_128x64;
  else if (M <= 64) tile = TILE_64x128;
  return launch_gemm((hipStream_t)stream, tile, a_kmajor ? KROW : ROWK, b_kmajor ? KROW : ROWK,
                     epi, g, splits, dtype);
}

int blh_gemm_f32(void* stream, const float* A, int64_t lda, int32_t a_kmajor, const float* B,
                 int64_t ldb, int32_t b_kmajor, float* C, int64_t ldc, int64_t M, int64_t N,
                 int64_t K, int32_t splits, const float* bias, const float* addend,
                 int64_t ldadd) {
  return gemm_entry(0, stream, A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, M, N, K, splits, bias,
                    addend, ldadd);
}

int blh_gemm_bf16x3(void* stream, const float* A, int64_t lda, int32_t a_kmajor, const float* B,
                    int64_t ldb, int32_t b_kmajor, float* C, int64_t ldc, int64_t M, int64_t N,
                    int64_t K, int32_t splits, const float* bias, const float* addend,
                    int64_t ldadd) {
  return gemm_entry(2, stream, A, lda, a_kmajor, B, ldb, b_kmajor, C, ldc, M, N, K, splits, bias,
                    addend, ldadd);
}

int64_t blh_gemm_fp16x2_workspace_bytes(void) { return 2 * WAMAX_PARTS * (int64_t)sizeof(float); }

int blh_gemm_fp16x2(void* stream, const float* A, int64_t lda, int32_t a_kmajor, const float* B,
                    int64_t ldb, int32_t b_kmajor, float* C, int64_t ldc, int64_t M, int64_t N,
                    int64_t K, int32_t splits, const float* bias, const float* addend,
                    int64_t ldadd, void* workspace, int32_t maxima_ready) {
  if (!A || !B || !C || !workspace || M <= 0 || N <= 0 || K <= 0 || splits < 1)
    return BLH_ERR_INVALID_ARGUMENT;
  if (bias && addend) return BLH_ERR_INVALID_ARGUMENT;
  if (splits > 1 && (bias || addend)) return BLH_ERR_INVALID_ARGUMENT;
  // dense operands only (the maxima are taken over M*K and N*K contiguous floats)
  if (lda != (a_kmajor ? M : K) || ldb != (b_kmajor ? N : K)) return BLH_ERR_SHAPE;
  if ((M * K) % 4 != 0 || (N * K) % 4 != 0) return BLH_ERR_SHAPE;
  hipStream_t s = (hipStream_t)stream;
  float* part = (float*)workspace;
  if (!maxima_ready) {
    BLH_TRY(launch_wamax(s, A, 0, 1, M * K, part));
    BLH_TRY(launch_wamax(s, B, 0, 1, N * K, part + WAMAX_PARTS));
  }
  GemmParams g{};
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.k_per_split = (splits > 1) ? (int)round_up(ceil_div(K, splits), SPLIT_GRAIN) : (int)K;
  if (splits > 1 && (int64_t)g.k_per_split * (splits - 1) >= K) return BLH_ERR_SHAPE;
  g.c_split_stride = M * ldc;
  g.bias = bias; g.addend = addend; g.ldadd = ldadd;
  g.a_amax = part; g.a_namax = WAMAX_PARTS;
  g.b_amax = part + WAMAX_PARTS; g.b_namax = WAMAX_PARTS;
  const int epi = bias ? EPI_BIAS : (addend ? EPI_ADD : EPI_STORE);
  GemmTile tile = TILE_128x128;
  if (N <= 32) tile = TILE_128x32;
  else if (N <= 64) tile = TILE_128x64;
  else if (M <= 64) tile = TILE_64x128;
  return launch_gemm(s, tile, a_kmajor ? KROW : ROWK, b_kmajor ? KROW : ROWK, epi, g, splits, 3);
}

int blh_linear_fwd_stats(void* stream, const float* A, const float* W, const float* bias, float* Z,
                         float* stat_part, int64_t M, int64_t N, int64_t K) {
  if (!A || !W || !bias || !Z || !stat_part || M <= 0 || N <= 0 || K <= 0)
    return BLH_ERR_INVALID_ARGUMENT;
  GemmParams g{};
  g.A = A; g.lda = K; g.B = W; g.ldb = K; g.C = Z; g.ldc = N;
  g.M = (int)M; g.N = (int)N; g.K = (int)K; g.k_per_split = (int)K;
  g.bias = bias; g.stat_part = stat_part;
  return launch_gemm((hipStream_t)stream, TILE_128x128, ROWK, ROWK, EPI_BIAS_STATS, g, 1);
}

int blh_dropout_mask(void* stream, const blh_dropout* drop, int32_t layer, int64_t batch,
                     int32_t width, uint8_t* keep_out) {
  if (!drop || !keep_out || batch <= 0 || width <= 0 || width % 4 != 0 || drop->keep_mask)
    return BLH_ERR_INVALID_ARGUMENT;
  BLH_TRY(check_drop(drop));
  DropoutSrc d{nullptr, drop->seed, drop->step, drop->row_offset, drop->layer_base + layer, nullptr};
  return launch_dropout_mask((hipStream_t)stream, keep_out, batch, width, d);
}

int blh_gemm_bf16s(void* stream, const uint16_t* A, int64_t lda, int32_t a_kmajor, const uint16_t* B,
                   int64_t ldb, int32_t b_kmajor, void* C, int64_t ldc, int32_t out_bf16, int64_t M,
                   int64_t N, int64_t K, int32_t splits, const float* bias, const uint16_t* addend,
                   int64_t ldadd, float* stat_part) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || splits < 1) return BLH_ERR_INVALID_ARGUMENT;
  if (bias && addend) return BLH_ERR_INVALID_ARGUMENT;
  if (splits > 1 && (bias || addend)) return BLH_ERR_INVALID_ARGUMENT;
  if (stat_part && (!bias || splits > 1)) return BLH_ERR_INVALID_ARGUMENT;
  GemmParamsH g{};
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.k_per_split = (splits > 1) ? (int)round_up(ceil_div(K, splits), 128) : (int)K;
  if (splits > 1 && (int64_t)g.k_per_split * (splits - 1) >= K) return BLH_ERR_SHAPE;
  g.c_split_stride = M * ldc;
  g.bias = bias; g.addend = addend; g.ldadd = ldadd; g.stat_part = stat_part;
  const int epi = bias ? (stat_part ? EPI_BIAS_STATS : EPI_BIAS) : (addend ? EPI_ADD : EPI_STORE);
  return launch_gemm_bf16s((hipStream_t)stream, a_kmajor ? KROW : ROWK, b_kmajor ? KROW : ROWK, epi,
                           out_bf16 != 0, g, splits);
}

int blh_gemm_bf16s_batched(void* stream, const uint16_t* A, int64_t lda, int32_t a_kmajor, int64_t a_item_stride,
                           const uint16_t* B, int64_t ldb, int32_t b_kmajor, int64_t b_item_stride, float* C,
                           int64_t ldc, int64_t c_item_stride, int64_t M, int64_t N, int64_t K, int32_t items,
                           int32_t splits) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || items < 1 || splits < 1) return BLH_ERR_INVALID_ARGUMENT;
  if (K % splits != 0 || a_item_stride % 8 != 0 || b_item_stride % 8 != 0) return BLH_ERR_SHAPE;
  if (splits > 1 && c_item_stride < (int64_t)splits * M * ldc) return BLH_ERR_INVALID_ARGUMENT;
  GemmParamsH g{};
  g.A = A; g.lda = lda; g.B = B; g.ldb = ldb; g.C = C; g.ldc = ldc;
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.k_per_split = (int)(K / splits);
  g.c_split_stride = M * ldc;
  g.batch_splits = splits;
  g.a_batch_stride = a_item_stride; g.b_batch_stride = b_item_stride; g.c_batch_stride = c_item_stride;
  return launch_gemm_bf16s((hipStream_t)stream, a_kmajor ? KROW : ROWK, b_kmajor ? KROW : ROWK, EPI_STORE, false, g,
                           items * splits);
}

int32_t blh_gemm_bf16s_tile(int64_t M, int64_t N, int64_t K, int32_t a_kmajor, int32_t b_kmajor,
                            int32_t out_bf16, int32_t splits) {
  if (M <= 0 || N <= 0 || K <= 0 || splits < 1) return BLH_ERR_INVALID_ARGUMENT;
  GemmParamsH g{};   // contiguous operands, aligned pointers
  g.lda = a_kmajor ? M : K; g.ldb = b_kmajor ? N : K; g.ldc = N;
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.k_per_split = (splits > 1) ? (int)round_up(ceil_div(K, splits), 128) : (int)K;
  g.c_split_stride = M * N;
  return gemm_bf16s_tile_rows(gemm_bf16s_pick_tile(a_kmajor ? KROW : ROWK, b_kmajor ? KROW : ROWK, out_bf16 != 0, g, splits));
}

int32_t blh_gemm_bf16s_tile_cols(int64_t M, int64_t N, int64_t K, int32_t a_kmajor, int32_t b_kmajor,
                                 int32_t out_bf16, int32_t splits) {
  if (M <= 0 || N <= 0 || K <= 0 || splits < 1) return BLH_ERR_INVALID_ARGUMENT;
  GemmParamsH g{};
  g.lda = a_kmajor ? M : K; g.ldb = b_kmajor ? N : K; g.ldc = N;
  g.M = (int)M; g.N = (int)N; g.K = (int)K;
  g.k_per_split = (splits > 1) ? (int)round_up(ceil_div(K, splits), 128) : (int)K;
  g.c_split_stride = M * N;
  return gemm_bf16s_tile_cols(gemm_bf16s_pick_tile(a_kmajor ? KROW : ROWK, b_kmajor ? KROW : ROWK, out_bf16 != 0, g, splits));
}

int blh_wgrad_plan_bf16s(int64_t width, int64_t batch, int32_t stages, int32_t* batched_slabs,
                         int32_t* stage_slabs) {
  if (width <= 0 || batch <= 0 || stages < 1) return BLH_ERR_INVALID_ARGUMENT;
  if (batched_slabs) *batched_slabs = wgrad_batched_plan_h(width, batch, stages).splits;
  if (stage_slabs) *stage_slabs = wgrad_plan_h(width, width, batch).splits;
  return BLH_OK;
}

int blh_gemm_bf16s_force_tile(int32_t tile) {
  if (tile != 0 && tile != -1 && tile != H_TILE_128 && tile != H_TILE_256 && tile != H_TILE_128x256)
    return BLH_ERR_INVALID_ARGUMENT;
  gemm_bf16s_force_tile(tile);
  return BLH_OK;
}

int blh_cast_f32_to_bf16(void* stream, const float* src, uint16_t* dst, int64_t count) {
  if (!src || !dst || count <= 0) return BLH_ERR_INVALID_ARGUMENT;
  return launch_cast_f32_bf16((hipStream_t)stream, src, dst, count);
}

int blh_cast_bf16_to_f32(void* stream, const uint16_t* src, float* dst, int64_t count) {
  if (!src || !dst || count <= 0) return BLH_ERR_INVALID_ARGUMENT;
  return launch_cast_bf16_f32((hipStream_t)stream, src, dst, count);
}

// ---- the skinny projections exactly as the step launches them (profiling / unit tests) -----
static int64_t skinny_slab_floats(int64_t batch, int W, int IF, int OF) {
  const Splits es = pick_splits(batch, ceil_div(W, 128) * ceil_div(IF, 32));
  const Splits ds = pick_splits(batch, ceil_div(OF, 64) * ceil_div(W, 128));
  int64_t m = std::max<int64_t>(es.splits * (int64_t)W * IF, ds.splits * (int64_t)OF * W);
  m = std::max<int64_t>(m, decode_fwd_splits(batch, W).splits * batch * OF);
  return m + 1026 * (int64_t)OF + 4096;     // + decode-bias partials + loss partials
}

int64_t blh_skinny_workspace_bytes(int64_t batch, int32_t width, int32_t in_features,
                                   int32_t out_features) {
  if (batch <= 0 || width <= 0 || in_features <= 0 || out_features <= 0) return BLH_ERR_INVALID_ARGUMENT;
  return skinny_slab_floats(batch, width, in_features, out_features) * (int64_t)sizeof(float);
}

int blh_skinny_encode_fwd(void* stream, const float* x, const float* W0, const float* b0, float* Z,
                          float* stat_part, int32_t* stat_tile_rows, int64_t batch, int32_t width,
                          int32_t in_features) {
  if (!x || !W0 || !b0 || !Z || !stat_part || batch <= 0) return BLH_ERR_INVALID_ARGUMENT;
  GemmParams g{};
  g.A = x; g.lda = in_features; g.B = W0; g.ldb = in_features; g.C = Z; g.ldc = width;
  g.M = (int)batch; g.N = width; g.K = in_features; g.k_per_split = in_features;
  g.bias = b0; g.stat_part = stat_part;
  const bool enc64 = in_features <= 32 && batch >= 2048;
  if (stat_tile_rows) *stat_tile_rows = enc64 ? 64 : 128;
  return launch_gemm((hipStream_t)stream, enc64 ? TILE_64x128 : TILE_128x128, ROWK, ROWK,
                     EPI_BIAS_STATS, g, 1, 0);
}

int blh_skinny_decode_fwd_mse(void* stream, const float* A, const float* Wd, const float* bd,
                              const float* target, float* pred, float* dpred, float* loss_out,
                              void* workspace, int64_t workspace_bytes, int64_t batch, int32_t width,
                              int32_t out_features) {
  if (!A || !Wd || !bd || !target || !pred || !dpred || !workspace || batch <= 0)
    return BLH_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < blh_skinny_workspace_bytes(batch, width, 32, out_features)) return BLH_ERR_WORKSPACE;
  if (!decode_fwd_supported(batch, width, out_features)) return BLH_ERR_SHAPE;
  float* part = (float*)workspace;              // [1026*OF] bias partials, then loss partials
  float* loss_part = part + 1026 * (int64_t)out_features;
  const double denom = (double)batch * out_features;
  int np = 0;
  BLH_TRY(launch_decode_fwd_mse((hipStream_t)stream, A, Wd, bd, target, pred, dpred, loss_part, part,
                                batch, width, out_features, (float)(2.0 / denom), &np));
  // (the training step folds this tiny reduction into the optimiser kernel; NULL skips it)
  return loss_out ? launch_loss_finalize((hipStream_t)stream, loss_part, np, denom, loss_out) : BLH_OK;
}

int blh_skinny_decode_bwd(void* stream, const float* dpred, const float* A, const float* Wd,
                          float* dWd, float* dA, void* workspace, int64_t workspace_bytes,
                          int64_t batch, int32_t width, int32_t out_features) {
  if (!dpred || !A || !Wd || !dWd || !dA || !workspace || batch <= 0) return BLH_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < blh_skinny_workspace_bytes(batch, width, 32, out_features)) return BLH_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  const int W = width, OF = out_features;
  BLH_TRY(wgrad(0, s, TILE_64x128, dpred, OF, OF, A, W, W, batch, ceil_div(OF, 64) * ceil_div(W, 128),
                (float*)workspace, dWd, nullptr));
  GemmParams g{};
  g.A = dpred; g.lda = OF; g.B = Wd; g.ldb = W; g.C = dA; g.ldc = W;
  g.M = (int)batch; g.N = W; g.K = OF; g.k_per_split = OF;
  return launch_gemm(s, batch >= 2048 ? TILE_64x128 : TILE_128x128, ROWK, KROW, EPI_STORE, g, 1, 0);
}

int blh_skinny_encode_wgrad(void* stream, const float* dZ, const float* x, float* dW0, void* workspace,
                            int64_t workspace_bytes, int64_t batch, int32_t width, int32_t in_features) {
  if (!dZ || !x || !dW0 || !workspace || batch <= 0) return BLH_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < blh_skinny_workspace_bytes(batch, width, in_features, 48)) return BLH_ERR_WORKSPACE;
  return wgrad(0, (hipStream_t)stream, TILE_128x32, dZ, width, width, x, in_features, in_features, batch,
               ceil_div(width, 128) * ceil_div(in_features, 32), (float*)workspace, dW0, nullptr);
}

int blh_sum_slabs(void* stream, const float* slabs, int64_t count, int32_t splits, float* out) {
  if (!slabs || !out || count <= 0 || splits < 1) return BLH_ERR_INVALID_ARGUMENT;
  return launch_sum_slabs((hipStream_t)stream, slabs, count, splits, out);
}

}  // extern "C"
