// The fp32-storage step (gemm_dtype 0 exact fp32 MFMA; 2 / 3: the split-precision GEMMs): the fixed kernel
// DAG of forward and backward (/root/reference/model/bilinear.py:31-41, train_bilinear.py:76-79).  Pure
// enqueue code: no allocation, no synchronisation, hipGraph-capturable.
#include "step.h"

namespace blh {

// ------------------------------------------------------------- forward -----
// The encode stage without its pre-BatchNorm tensor (encode_f32.hip): exact-fp32 mode, per-rank statistics (SyncBN
// exchanges tile sums of Z), the multi-launch path's batch sizes.  A train-mode forward that takes it records saved
// format 3 for the workspace; the backward that consumes that forward reads the record.
bool enc_fused_ok(const blh_context* ctx, const blh_model_desc* d, int64_t batch) {
  return d->gemm_dtype == 0 && !ctx->sync.fn && !ctx->knob(KNOB_NO_ENCODE_FUSE) && batch > 384 &&
         enc_fused_supported(batch, d->width, d->in_features);
}

int forward_impl(blh_context* ctx, const blh_model_desc* d, hipStream_t s, const float* params,
                        float* bn_running, int64_t* nbt, const float* x, const blh_dropout* drop,
                        float momentum, const Workspace& ws, float* pred, int64_t batch,
                        bool train, const float* target, float mse_scale, float* loss_part,
                        int* loss_nparts) {
  const ArenaLayout L = make_layout(d);
  const int nh = (int)L.heavy.size();
  const int W = d->width;
  const bool enc_fused = train && enc_fused_ok(ctx, d, batch);
  if (train)     // (ws.Z[0] is the workspace base, api_layout.h carve)
    ctx->note_saved(ws.Z[0], batch, enc_fused ? blh_context::SAVED_ENC_FUSED : blh_context::SAVED_MULTI);
  if (ws.amax_W)   // gemm_dtype 3: max |w| of every hidden Linear weight, once per forward
    for (int i = 1; i < nh; ++i)
      BLH_TRY(launch_wamax(s, params + L.heavy[i].w, 0, 1, (int64_t)W * W, ws.amax_W + (int64_t)i * WAMAX_PARTS));
  for (int i = 0; i < nh; ++i) {
    const HeavyOffsets& h = L.heavy[i];
    const float* in = (i == 0) ? x : ws.A[i - 1];
    if (i == 0 && enc_fused) {
      // x statistics -> BatchNorm statistics -> A0 and the keep bits; Z0's buffer serves as scratch
      BLH_TRY(launch_enc_forward(s, x, params + h.w, params + h.b, params + h.gamma, params + h.beta, bn_running,
                                 bn_running + W, nbt, momentum, ws.bn_saved[0], ws.Z[0], ws.A[0], ws.keep[0], batch, W,
                                 layer_drop(ctx, drop, 0, batch, W)));
      continue;
    }
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
    const Splits fs = small_m_splits(batch, W, h.fan_in, d->gemm_dtype);
    // 64-row tiles: the encode Linear at large batch (below), and the hidden Linears of a half-chip batch
    // (api_layout.h: mid_tile64)
    const bool mid64 = train && i >= 1 && fs.splits == 1 && mid_tile64(batch, W, d->gemm_dtype);
    const bool enc64 = (train && i == 0 && fs.splits == 1 && h.fan_in <= 32 && batch >= 2048) || mid64;
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
      const bool ev64 = i >= 1 && mid_tile64(batch, W, d->gemm_dtype);      // (half-chip batch: 64-row tiles)
      BLH_TRY(launch_gemm(s, ev64 ? TILE_64x128 : TILE_128x128, ROWK, ROWK, EPI_BN_RELU, g, 1, d->gemm_dtype));
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
      // (the small-batch path produced statistics tiles of fwd_finish_stat_rows() rows)
      const int st_rows = fs.splits > 1 ? fwd_finish_stat_rows() : (enc64 ? 64 : 128);
      const int st_tiles = (int)ceil_div(batch, st_rows);
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
  ctx->dec_da_ws = nullptr;
  if (train && target && d->gemm_dtype == 0 && !ctx->knob(KNOB_NO_DECODE_FUSE) && decode_fused_supported(batch, W, OF)) {
    // one pass over the last activation: prediction, MSE, dpred, the loss / decode-bias partials AND the decode data
    // gradient dA = dP Wd (into G0, where backward expects the gradient of the last stage's output)
    int np = 0;
    // (the two-stream backward forks its side stream behind this kernel: outside capture the fork event rides on
    //  the launch's own completion signal — 3.3 against 5.3 us on this stream for a marker packet behind it,
    //  tools/event_cost_bench, profiles/r05_fork_cost.md; an event nobody waits for costs nothing)
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    (void)hipStreamIsCapturing(s, &cap);
    ctx->dec_fork_attached = ctx->two_stream && cap == hipStreamCaptureStatusNone;
    if (ctx->dec_fork_attached) tl_stop_event = ctx->ev_dz[nh];
    const int rc = launch_decode_fused(s, ws.A[nh - 1], params + L.dec_w, params + L.dec_b, target, pred, ws.dpred, ws.G0,
                                       loss_part, ws.dec_bias_part, batch, W, OF, mse_scale, &np);
    tl_stop_event = nullptr;
    BLH_TRY(rc);
    if (loss_nparts) *loss_nparts = np;
    ctx->dec_da_ws = ws.Z[0]; ctx->dec_da_batch = batch;
    return BLH_OK;
  }
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
// dW = dZ^T act, the reduction over the batch split across workgroups; the partial slabs are summed right away, while
// they are still in L2 / Infinity Cache (one deferred sum of all stages' slabs at the end of backward saved six launches
// and read 68 MB of cold slabs: 1.268 against 1.253 ms per step in round 1, removed in round 6).
int wgrad(int dtype, hipStream_t s, GemmTile tile, const float* dZ, int64_t ld_dz, int M,
                 const float* act, int64_t ld_act, int N, int64_t batch, int64_t tiles,
          float* slabs, float* out, const float* amax_dz, const float* amax_act,
          int amax_parts, double* sq, int sq_blocks) {
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
  if (sp.splits == 1) {
    g.C = out; g.c_split_stride = 0;
    return launch_gemm(s, tile, KROW, KROW, EPI_STORE, g, 1, dtype);
  }
  g.C = slabs; g.c_split_stride = (int64_t)M * N;
  BLH_TRY(launch_gemm(s, tile, KROW, KROW, EPI_STORE, g, sp.splits, dtype));
  if (sq) return launch_sum_slabs_sq(s, slabs, (int64_t)M * N, sp.splits, out, sq, sq_blocks);
  return launch_sum_slabs(s, slabs, (int64_t)M * N, sp.splits, out);
}

// Side stream for the weight-gradient GEMMs (owned by the context): nothing in the rest of
// backward depends on dW, so wgrad(l) (+ its slab sum) runs on a second stream concurrently
// with dgrad(l) and the HBM-bound BatchNorm-backward kernels of stage l-1, which leave the
// MFMA pipes idle.  Fork / join by events (capturable into a hipGraph); dZ is double-buffered
// so that stage l-2 does not overwrite what wgrad(l) is still reading.

int backward_impl(blh_context* ctx, const blh_model_desc* d, hipStream_t s, const float* params,
                         const float* x, const blh_dropout* drop, const Workspace& ws,
                         const float* dpred, float* grads, int64_t batch,
                         blh_grad_ready_fn on_ready, void* user, int saved_mode,
                         const FusedBackward* fused) {
  const ArenaLayout L = make_layout(d);
  const int nh = (int)L.heavy.size();
  const int W = d->width;
  const int OF = d->out_features;
  const int chunks = ew_num_row_chunks(batch);

  tl_stop_event = nullptr;
  // Fused step, gradient norm (clip_grad_norm_, train_bilinear.py:81): the kernels that WRITE the gradient ranges
  // — slab sums of the weight gradients, the gamma / beta finalize, the bias reduction — also emit the
  // sum-of-squares partials of what they write (ws.sumsq_fold: a dense array, every launch fills exactly its
  // own slots, so the sum is deterministic), and no pass over the arena is left between the join and
  // clip + Adam (grads_finish: 9 us + its launch gap at configs[1]).  Needs every weight gradient to come out
  // of a slab sum, and the slots to fit.
  struct Fold { bool on = false; int per_w = 0; std::vector<int> w_off; int gb0 = 0, bias0 = 0, total = 0; } fold;
  if (fused && fused->sumsq_part && !on_ready && !ctx->knob(KNOB_NO_SUMSQ_FOLD) && W % 16 == 0 &&
      fused->dec_bias_S > 0) {
    const bool slabbed = pick_splits(batch, ceil_div(W, 128) * ceil_div(W, 128)).splits > 1 &&
                         pick_splits(batch, ceil_div(W, 128) * ceil_div(d->in_features, 32)).splits > 1 &&
                         pick_splits(batch, ceil_div(OF, 64) * ceil_div(W, 128)).splits > 1;
    fold.per_w = (int)std::min<int64_t>(256, std::max<int64_t>(8, 2048 / (nh + 1)));
    int off = 0;
    const bool enc0 = saved_mode == blh_context::SAVED_ENC_FUSED;      // (stage 0: enc_wgrad_finish writes dW0)
    for (int i = 0; i <= nh; ++i) {
      fold.w_off.push_back(off);
      const int64_t cnt = i == nh ? (int64_t)OF * W : (int64_t)W * L.heavy[i].fan_in;
      off += (i == 0 && enc0) ? enc_bwd_finish_blocks(W) : sum_slabs_sq_blocks(cnt, fold.per_w);
    }
    fold.gb0 = off;
    fold.bias0 = fold.gb0 + nh * bn_bwd_finalize_blocks(W);
    fold.total = fold.bias0 + bias_colreduce_blocks(W, nh, true);
    fold.on = slabbed && nh >= 2 && fold.total <= SUMSQ_FOLD_PARTS;
  }
  auto fold_w = [&](int i) -> double* { return fold.on ? ws.sumsq_fold + fold.w_off[i] : nullptr; };
  // two streams: on by default (-3 % step)
  const bool two = ctx->two_stream && !ctx->sync.fn && small_m_splits(batch, W, W, d->gemm_dtype).splits == 1;
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
  // (recorded only where something waits for it — the join at the end, and gemm_dtype 3's amax buffers: every
  //  record is a marker packet in the side stream's queue; no measurable step time either way, profiles/r04_batch_sweeps.md)
  const int join_idx = on_ready != nullptr ? 0 : 1;
  auto wdone = [&](int idx) -> int {
    if (!two) return BLH_OK;
    if (idx != join_idx && !ws.amax_W) return BLH_OK;
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
  // (one-pass decode: the forward that produced this dpred already left dA in G0 — the side stream forks at once,
  //  behind the forward's last kernel, and the main stream goes straight to the first BatchNorm backward)
  const bool have_da = fused && dpred == ws.dpred && ctx->dec_da_ws == ws.Z[0] && ctx->dec_da_batch == batch;
  ctx->dec_da_ws = nullptr;
  if (!have_da) {
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
  if (have_da && two) {
    if (!(ctx->dec_fork_attached && attach)) BLH_HIP_TRY(hipEventRecord(g_side.ev_dz[nh], s));
    BLH_HIP_TRY(hipStreamWaitEvent(s2, g_side.ev_dz[nh], 0));
  } else {
    BLH_TRY(fork_wait(nh));
  }
  BLH_TRY(wgrad(d->gemm_dtype, s2, TILE_64x128, dpred, OF, OF, ws.A[nh - 1], W, W, batch,
                ceil_div(OF, 64) * ceil_div(W, 128), ws.slabs, grads + L.dec_w, nullptr, nullptr, 0, fold_w(nh),
                fold.per_w));
  BLH_TRY(wdone(nh));
  // (decode bias: on the side stream under a hook — behind the fork, its inputs are older than that)
  if (!fused) BLH_TRY(launch_colsum(on_ready ? s2 : s, dpred, batch, OF, OF, ws.colsum_part, grads + L.dec_b));
  else if (on_ready)   // (the hook wants the decode range complete now; else: one batched launch at the end)
    BLH_TRY(launch_colreduce(s2, ws.dec_bias_part, fused->dec_bias_S, OF, OF, grads + L.dec_b));
  BLH_TRY(ready(nh, L.dec_w, L.total - L.dec_w));

  // (the BatchNorm-backward sums in the fp32 data-gradient epilogue — SURVEY K9 on the exact-fp32 path — were built,
  //  measured slower than the streaming reduce kernel beside the side stream's weight gradient and removed:
  //  profiles/r04_k9_f32.md; the bf16-storage step keeps its K9, step_bf16s.hip)
  const bool enc_fused = saved_mode == blh_context::SAVED_ENC_FUSED;
  for (int i = nh - 1; i >= 0; --i) {
    const HeavyOffsets& h = L.heavy[i];
    if (i == 0 && enc_fused) {
      // Encode stage without Z0 (encode_f32.hip): one pass over dA0 leaves sum dY' and dY'^T X per row block, the
      // finish kernel forms dgamma, dbeta, dW0 and db0 from them and the forward's sums of x.  Main stream; under
      // the bucket hook the side stream is made to wait for it (the range has to be complete there).
      BLH_TRY(launch_enc_backward(s, ws.G0, x, params + h.w, params + h.b, ws.bn_saved[0], ws.keep[0], ws.Z[0], batch, W,
                                  grads + h.w, grads + h.gamma, grads + h.beta,
                                  on_ready ? grads + h.b : ws.dz_colsum_part, on_ready ? 1 : chunks, fold_w(0),
                                  fold.on ? ws.sumsq_fold + fold.gb0 : nullptr));
      if (on_ready) {
        if (ctx->two_stream) {
          BLH_HIP_TRY(hipEventRecord(g_side.ev_r[0], s));
          BLH_HIP_TRY(hipStreamWaitEvent(g_side.s2, g_side.ev_r[0], 0));
        }
        if (two) BLH_TRY(wdone(0));
        on_ready(user, h.w, ((1 < nh) ? L.heavy[1].w : L.dec_w) - h.w);
      }
      continue;
    }
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
    const bool fork_late = side && late_policy && i > 0 && small_m_splits(batch, W, W, d->gemm_dtype).splits == 1;
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
      const Splits ds2 = small_m_splits(batch, W, W, d->gemm_dtype);
      float* dst = first_of_block ? ws.G0 : ws.G1;
      if (ds2.splits > 1) {
        g.C = ws.slabs; g.c_split_stride = batch * (int64_t)W; g.k_per_split = ds2.k_per;
        BLH_TRY(launch_gemm(s, TILE_128x128, ROWK, KROW, EPI_STORE, g, ds2.splits, d->gemm_dtype));
        BLH_TRY(launch_sum_slabs_add(s, ws.slabs, batch * (int64_t)W, ds2.splits,
                                     first_of_block ? ws.G0 : nullptr, dst));
      } else {
        const GemmTile dtile = mid_tile64(batch, W, d->gemm_dtype) ? TILE_64x128 : TILE_128x128;
        if (first_of_block) {
          // d(block input) = dZ W + d(block output)   (skip path), in place in G0
          g.C = ws.G0; g.addend = ws.G0; g.ldadd = W;
          BLH_TRY(launch_gemm(s, dtile, ROWK, KROW, EPI_ADD, g, 1, d->gemm_dtype));
        } else {
          g.C = ws.G1;
          BLH_TRY(launch_gemm(s, dtile, ROWK, KROW, EPI_STORE, g, 1, d->gemm_dtype));
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
                    (two && !side) ? ws.enc_slabs : ws.slabs, grads + h.w, nullptr, nullptr, 0, fold_w(0),
                    fold.per_w));
      if (side) BLH_TRY(wdone(0));
    } else {
      BLH_TRY(wgrad(d->gemm_dtype, sw, TILE_128x128, dzbuf, W, W, ws.A[i - 1], W, W, batch,
                    ceil_div(W, 128) * ceil_div(W, 128), ws.slabs, grads + h.w, dz_amax, ws.amax_A[i - 1],
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
  if (two) BLH_HIP_TRY(hipStreamWaitEvent(s, g_side.ev_w[join_idx], 0));
  if (fold.on) {     // the producers left the partials: hand them to clip + Adam
    fused->sumsq_src[0] = ws.sumsq_fold;
    *fused->sumsq_nparts = fold.total;
  } else if (!on_ready && fused && fused->sumsq_part) {
    // (the producers' partials were not available — small shapes, BLH_NO_SUMSQ_FOLD: one pass over the arena)
    BLH_TRY(launch_sumsq(s, grads, L.total, fused->sumsq_part, fused->sumsq_nparts));
  }
  return BLH_OK;
}


}  // namespace blh
