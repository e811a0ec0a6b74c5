// The bf16-storage step (gemm_dtype 4, BASELINE configs 3-5): forward and backward DAG.  Pure enqueue code.
#include "step.h"

namespace blh {

// =================================================================================================
// gemm_dtype 4 — "bf16s": bf16 storage (BASELINE configs 3-5).  Every [B,W] tensor (pre-BN output
// Z, activation A, gradients G / dZ), the network input and a shadow of all parameters are bf16
// in HBM; every contraction runs on gemm_bf16s_kernel.h (bf16 MFMA, fp32 accumulate, operands fed
// by LDS-DMA without any conversion); BatchNorm statistics (from the fp32 accumulators, before
// rounding), the parameters, their gradients (fp32 slabs of the weight-gradient GEMM), Adam and
// the loss stay fp32.  Gradients need no loss scaling: bf16 keeps fp32's exponent range.
// =================================================================================================
// The encode stage without its pre-BatchNorm tensor (encode_f32.hip, bf16-storage form): per-rank statistics (SyncBN
// exchanges tile sums of Z).  The forward records what it saved (blh_context::note_saved); the backward reads the
// record of its workspace, so a knob or SyncBN state that changed in between cannot pair them wrongly.
bool enc_fused_ok_h(const blh_context* ctx, const blh_model_desc* d, int64_t batch) {
  return !ctx->sync.fn && !ctx->knob(KNOB_NO_ENCODE_FUSE) && enc_fused_supported_h(batch, d->width, d->in_features);
}

int forward_h(blh_context* ctx, const blh_model_desc* d, hipStream_t s, const float* params,
                     float* bn_running, int64_t* nbt, const float* x, const blh_dropout* drop,
                     float momentum, const WorkspaceH& ws, float* pred, int64_t batch, bool train,
              bool shadow_valid, const float* target, float mse_scale, int* loss_nparts) {
  const ArenaLayout L = make_layout(d);
  const int nh = (int)L.heavy.size();
  const int W = d->width, OF = d->out_features, IF = d->in_features;
  const bool enc_fused_fwd = train && enc_fused_ok_h(ctx, d, batch);
  if (train)     // (ws.wsh is the workspace base, api_layout.h carve_h)
    ctx->note_saved(ws.wsh, batch, enc_fused_fwd ? blh_context::SAVED_ENC_FUSED : blh_context::SAVED_MULTI);
  // bf16 images of the parameters (the GEMMs read the weights from it) and of the input
  // (shadow_valid: the previous fused step's Adam kernel wrote it, BLH_OPT_PERSISTENT_SHADOW)
  const bool wdT_valid = shadow_valid && ctx->shadow_wdT;
  ctx->shadow_params = ctx->shadow_ws = nullptr;
  ctx->shadow_wdT = false;
  // one-pass decode (skinny.hip): forward + MSE + the decode data gradient from one read of the last activation; its
  // second phase reads the decode weight through an image the cast launch below writes — or, with a persistent
  // shadow, the previous step's Adam kernel wrote (elementwise.hip: ShadowDst.wdT)
  const bool dec_fused = train && target && (!shadow_valid || wdT_valid) && !ctx->knob(KNOB_NO_DECODE_FUSE) &&
                         decode_fused_h_supported(batch, W, OF);
  ctx->dec_da_ws = nullptr;
  if (!shadow_valid)
    BLH_TRY(launch_cast2_f32_bf16(s, params, ws.wsh, L.total, x, ws.xh, batch * IF, dec_fused ? params + L.dec_w : nullptr,
                                  dec_fused ? ws.wdT : nullptr, W, OF));
  // (weight image kept by Adam: only x is left to cast — the encode stage's statistics kernel does it on the way)
  const bool x_cast_in_stage = shadow_valid && enc_fused_fwd;
  if (shadow_valid && !x_cast_in_stage) BLH_TRY(launch_cast_f32_bf16(s, x, ws.xh, batch * IF));
  for (int i = 0; i < nh; ++i) {
    const HeavyOffsets& h = L.heavy[i];
    if (i == 0 && enc_fused_fwd) {
      // x statistics -> BatchNorm statistics -> A0 and the keep-and-gate bits; Z0's buffer serves as scratch
      BLH_TRY(launch_enc_forward_h(s, ws.xh, x_cast_in_stage ? x : nullptr, ws.wsh + h.w, params + h.b, params + h.gamma, params + h.beta, bn_running,
                                   bn_running + W, nbt, momentum, ws.bn_saved[0], ws.Z[0], ws.A[0], ws.keep[0], batch, W,
                                   layer_drop(ctx, drop, 0, batch, W)));
      continue;
    }
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
    // (the whole stage in ONE launch behind a grid barrier was built, bit-identical, measured slower and removed:
    //  profiles/r04_fused_forward.md)
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
  if (dec_fused) {
    int np = 0;
    BLH_TRY(launch_decode_fused_h(s, ws.A[nh - 1], ws.wsh + L.dec_w, ws.wdT, params + L.dec_b, target, pred, ws.dpred,
                                  ws.dpredh, ws.G0, ws.loss_part, ws.dec_bias_part, batch, W, OF, mse_scale, &np));
    if (loss_nparts) *loss_nparts = np;
    ctx->dec_da_ws = ws.wsh; ctx->dec_da_batch = batch;     // (the backward that follows finds dA in G0)
    return BLH_OK;
  }
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
// dW[M][N] = dZ^T act over `batch` rows.  The GEMM moves 16-byte chunks along the contraction index, i.e. whole groups
// of 8 rows; a batch that is not a multiple of 8 — the last batch of an epoch of the reference's DataLoader,
// /root/reference/train_bilinear.py:33-43 (drop_last unset) — runs the GEMM over the first batch - batch % 8 rows and a
// small kernel adds the outer products of the up to seven rows that are left (r06; such a batch used to be refused).
static int wgrad_h(hipStream_t s, const uint16_t* dZ, int64_t ld_dz, int M, const uint16_t* act,
                   int64_t ld_act, int N, int64_t batch, float* slabs, int64_t slab_cap, float* out) {
  const int64_t kmain = batch & ~(int64_t)7;
  const int tail = (int)(batch - kmain);
  if (kmain > 0) {
    const Splits sp = wgrad_plan_h(M, N, kmain);
    GemmParamsH g{};
    g.A = dZ; g.lda = ld_dz; g.B = act; g.ldb = ld_act;
    g.M = M; g.N = N; g.K = (int)kmain; g.k_per_split = sp.k_per; g.ldc = N;
    if (sp.splits == 1) {
      g.C = out;
      BLH_TRY(launch_gemm_bf16s(s, KROW, KROW, EPI_STORE, false, g, 1));
    } else {
      if ((int64_t)sp.splits * M * N > slab_cap) return BLH_ERR_WORKSPACE;     // (never: api_layout.h sizes for this plan)
      g.C = slabs; g.c_split_stride = (int64_t)M * N;
      BLH_TRY(launch_gemm_bf16s(s, KROW, KROW, EPI_STORE, false, g, sp.splits));
      BLH_TRY(launch_sum_slabs(s, slabs, (int64_t)M * N, sp.splits, out));
    }
  }
  if (tail) return launch_wgrad_tail_h(s, dZ + kmain * ld_dz, ld_dz, act + kmain * ld_act, ld_act, tail, M, N, out, kmain > 0);
  return BLH_OK;
}

// dec_bias_S > 0: the forward ran the fused decode + MSE kernel: ws.dpredh and the decode-bias
// partials (dec_bias_S rows of ws.dec_bias_part) are already there
int backward_h(blh_context* ctx, const blh_model_desc* d, hipStream_t s, const float* params,
                      const blh_dropout* drop, const WorkspaceH& ws, const float* dpred,
                      float* grads, int64_t batch, blh_grad_ready_fn on_ready, void* user, int saved_mode,
                      int dec_bias_S, int* fold_nparts) {
  // fold_nparts != NULL (the fused step): when the hidden weight gradients leave through ONE batched slab sum, that
  // kernel also takes the gradient norm's partials — of its own output and of every other range of the arena — into
  // ws.sumsq_part and reports how many; 0: the caller runs its pass over the arena (launch_sumsq)
  if (fold_nparts) *fold_nparts = 0;
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
  // One stream when the hidden weight gradients go out as ONE batched launch after the chain (no bucket hook): the
  // side stream would then carry only the decode and stage-0 weight gradients (22 + 17 us of work at configs[3]'s
  // shape) for two forks and a join (5 + 7 + 12 us of cross-queue latency), and the stage-0 one starves behind the
  // batched GEMM, which holds every CU's LDS.  Measured (profiles/r04_batch_sweeps.md): configs[2] 1.440 against
  // 1.458 ms, configs[3] shape 0.925 against 0.943, configs[4] shape 7.38 against 7.44.  Per-stage weight gradients
  // (smaller batches) keep the side stream: 0.755 against 0.830 ms at 4 x 1024, B = 4096.
  const bool batched_main = on_ready == nullptr && nh - 1 >= 2 && wgrad_batched_plan_h(W, batch, nh - 1).splits > 0 &&
                            !ctx->knob(KNOB_BF16_FORCE_TWO_STREAM);
  const bool two = ctx->two_stream && !ctx->sync.fn && !batched_main;   // (SyncBN: the exchanges are enqueued on `s`)
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
  // (one-pass decode: the forward that produced this dpred already left dA in G0)
  const bool have_da = dec_bias_S > 0 && dpred == ws.dpred && ctx->dec_da_ws == ws.wsh && ctx->dec_da_batch == batch;
  ctx->dec_da_ws = nullptr;
  if (dec_bias_S == 0) BLH_TRY(launch_cast_f32_bf16(s, dpred, ws.dpredh, batch * OF));
  if (have_da) {
    if (two) BLH_HIP_TRY(hipEventRecord(ctx->ev_dz[nh], s));     // (fork_wait below waits for it)
  } else {
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
  BLH_TRY(wgrad_h(s2, ws.dpredh, OF, OF, ws.A[nh - 1], W, W, batch, ws.slabs, ws.slab_cap, grads + L.dec_w));
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
        wgroups.push_back(WGroup{lo, hi, wgrad_batched_plan_h(W, batch, hi - lo + 1, true)});
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
  auto launch_group = [&](const WGroup& grp, hipStream_t st, bool fold = false) -> int {
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
    if (fold) {
      // every range of the arena outside the group's weights (the group is ALL hidden stages here)
      SqRanges rg{};
      int64_t at = 0;
      for (int k = grp.lo; k <= grp.hi; ++k) {
        if (L.heavy[k].w > at) { rg.off[rg.n] = at; rg.cnt[rg.n] = L.heavy[k].w - at; ++rg.n; }
        at = L.heavy[k].w + (int64_t)W * W;
      }
      if (L.total > at) { rg.off[rg.n] = at; rg.cnt[rg.n] = L.total - at; ++rg.n; }
      return launch_sum_slabs_batched_sq(st, ws.bslabs, (int64_t)W * W, grp.plan.splits, items, g.c_batch_stride, out,
                                         gstride, grads, rg, ws.sumsq_part, SUMSQ_FOLD_PARTS_H, fold_nparts);
    }
    return launch_sum_slabs_batched(st, ws.bslabs, (int64_t)W * W, grp.plan.splits, items, g.c_batch_stride, out,
                                    gstride);
  };
  // a data-gradient GEMM whose output only feeds the BatchNorm backward of the stage below
  // (the second stage of a block: its output is not a skip operand; and stage 1, whose block-input gradient
  // nothing below needs) forms that stage's gated gradient dY' and the (dY' z, dY') column sums in its
  // epilogue (EPI_BN_BWD, big-tile kernels only): the stage below then skips bn_bwd_reduce_h2 and its
  // bn_bwd_apply_h2 reads no keep bits.  k9_chunks > 0: stage i's dA arrived that way, with that many partial rows.
  const bool k9_enabled = !ctx->knob(KNOB_NO_K9);
  const bool enc_fused = saved_mode == blh_context::SAVED_ENC_FUSED;
  int k9_chunks = 0;
  for (int i = nh - 1; i >= 0; --i) {
    const HeavyOffsets& h = L.heavy[i];
    if (i == 0 && enc_fused) {
      // Encode stage without Z0 (encode_f32.hip): one pass over dA0 (bf16) + the finish kernel, on the main stream
      BLH_TRY(launch_enc_backward_h(s, ws.G0, ws.xh, ws.wsh + h.w, params + h.b, ws.bn_saved[0], ws.keep[0], ws.Z[0], batch,
                                    W, grads + h.w, grads + h.gamma, grads + h.beta,
                                    on_ready ? grads + h.b : ws.dz_colsum_part, on_ready ? 1 : chunks));
      if (on_ready) {
        if (ctx->two_stream) {     // (the range has to be complete on the side stream)
          BLH_HIP_TRY(hipEventRecord(ctx->ev_r[0], s));
          BLH_HIP_TRY(hipStreamWaitEvent(ctx->s2, ctx->ev_r[0], 0));
        }
        on_ready(user, h.w, ((1 < nh) ? L.heavy[1].w : L.dec_w) - h.w);
      }
      continue;
    }
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
      // the group's GEMM first, then the bias gradients of its stages in ONE launch: four column reductions in front
      // of the GEMM held it back by 67 us on the low-priority side stream beside the main stream's data gradient,
      // and the last group's GEMM is the tail of the data-parallel step (profiles/r05_dp_overhead.md)
      BLH_TRY(launch_group(*grp, s2));
      {
        int64_t offs[32];
        const int items = grp->hi - grp->lo + 1;
        if (items > 32) return BLH_ERR_SHAPE;
        for (int k = 0; k < items; ++k) offs[k] = L.heavy[grp->lo + k].b;
        BLH_TRY(launch_bias_colreduce(s2, ws.dz_colsum_part + (int64_t)grp->lo * chunks * W, (int64_t)chunks * W, chunks,
                                      W, items, offs, grads, nullptr, 0, OF, L.dec_b));
      }
      {   // the group's stages complete together: one range
        const int64_t end = (grp->hi + 1 < nh) ? L.heavy[grp->hi + 1].w : L.dec_w;
        BLH_TRY(ready(grp->lo, L.heavy[grp->lo].w, end - L.heavy[grp->lo].w));
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
      // (stage 0 without Z0: nothing for the epilogue to gate against)
      const bool k9 = k9_enabled && tile != H_TILE_128 && k9_rows <= chunks && !(i == 1 && enc_fused) &&
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
      if (!batched_w) BLH_TRY(wgrad_h(s2, ws.dZ[i], W, W, ws.A[i - 1], W, W, batch, ws.slabs, ws.slab_cap, grads + h.w));
    } else {
      BLH_TRY(wgrad_h(s2, ws.dZ[0], W, W, ws.xh, IF, IF, batch, ws.slabs, ws.slab_cap, grads + h.w));
    }
    if (on_ready && !batched_w) {
      const int64_t end = (i + 1 < nh) ? L.heavy[i + 1].w : L.dec_w;
      BLH_TRY(ready(i, h.w, end - h.w));
    }
  }
  // the norm's partials from the batched slab sum: the group must be all hidden stages through slabs, on `s`, with
  // every other gradient final in front of it — the bias reduction then goes FIRST (it needs nothing from the group)
  const bool fold = fold_nparts != nullptr && on_ready == nullptr && batched_main && !wgroups.empty() &&
                    wgroups[0].plan.splits > 1 && nh - 1 <= 34 && !ctx->knob(KNOB_NO_SUMSQ_FOLD);
  auto bias_all = [&]() -> int {
    int64_t offs[32];
    if (nh > 32) return BLH_ERR_SHAPE;
    for (int i = 0; i < nh; ++i) offs[i] = L.heavy[i].b;
    return launch_bias_colreduce(s, ws.dz_colsum_part, (int64_t)chunks * W, chunks, W, nh, offs, grads,
                                 dec_bias_S > 0 ? ws.dec_bias_part : nullptr, dec_bias_S, OF, L.dec_b);
  };
  if (!on_ready && fold) BLH_TRY(bias_all());
  if (on_ready == nullptr && !wgroups.empty() && wgroups[0].plan.splits > 0) {
    BLH_TRY(launch_group(wgroups[0], s, fold));   // (everything is on `s` in this plan: batched_main above)
  }
  if (!on_ready && !fold) BLH_TRY(bias_all());
  if (two) {   // join: the side stream is in order, its last kernel is stage 0's slab sum
    BLH_HIP_TRY(hipEventRecord(ctx->ev_w[0], s2));
    BLH_HIP_TRY(hipStreamWaitEvent(s, ctx->ev_w[0], 0));
  }
  return BLH_OK;
}

}  // namespace blh
