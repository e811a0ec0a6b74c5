// C ABI of libbilinear_hip.so: contexts, argument checks, workspace carving and the entry points; the
// kernel DAGs of the step live in step_f32.hip / step_bf16s.hip (step.h).  Pure enqueue code: no
// allocation, no synchronisation, so every entry point is hipGraph-capturable.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include <mutex>
#include "step.h"
#include "small_step.h"

namespace blh {

thread_local int g_last_hip_error = 0;
thread_local hipEvent_t tl_stop_event = nullptr;

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
    case BLH_ERR_COMM: return "RCCL unavailable or an RCCL call failed (blh_comm_last_error)";
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
  c->two_stream = getenv("BLH_ONE_STREAM") == nullptr;
  c->late_fork = getenv("BLH_EARLY_FORK") ? 0 : (getenv("BLH_LATE_FORK") ? 1 : 2);
  c->knobs = blh::dev_knobs_from_env();
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
  delete c;
  return BLH_OK;
}

int blh_context_set_option(blh_context* c, int32_t option, int32_t value) {
  if (!c) return BLH_ERR_INVALID_ARGUMENT;
  switch (option) {
    case BLH_OPT_TWO_STREAM: c->two_stream = value != 0; return BLH_OK;
    case BLH_OPT_LATE_FORK:
      if (value < 0 || value > 2) return BLH_ERR_INVALID_ARGUMENT;
      c->late_fork = value;
      return BLH_OK;
    case BLH_OPT_SMALL_STEP:
      if (value < 0 || value > 1) return BLH_ERR_INVALID_ARGUMENT;
      c->small_step = value;
      return BLH_OK;
    case BLH_OPT_PERSISTENT_SHADOW:
      c->persistent_shadow = value != 0;
      c->shadow_params = c->shadow_ws = nullptr;
      c->shadow_wdT = false;
      return BLH_OK;
    case BLH_OPT_DEV_KNOBS:
      if (value < 0 || (value & ~blh::KNOB_ALL) != 0) return BLH_ERR_INVALID_ARGUMENT;
      c->knobs = value;
      return BLH_OK;
    case BLH_OPT_BUCKET_FLOATS:
      if (value < 0) return BLH_ERR_INVALID_ARGUMENT;
      c->bucket_floats = value;
      return BLH_OK;
  }
  return BLH_ERR_INVALID_ARGUMENT;
}

int blh_context_get_option(const blh_context* c, int32_t option) {
  if (!c) return BLH_ERR_INVALID_ARGUMENT;
  switch (option) {
    case BLH_OPT_TWO_STREAM: return c->two_stream ? 1 : 0;
    case BLH_OPT_LATE_FORK: return c->late_fork;
    case BLH_OPT_PERSISTENT_SHADOW: return c->persistent_shadow ? 1 : 0;
    case BLH_OPT_SMALL_STEP: return c->small_step;
    case BLH_OPT_DEV_KNOBS: return c->knobs;
    case BLH_OPT_BUCKET_FLOATS: return c->bucket_floats;
  }
  return BLH_ERR_INVALID_ARGUMENT;
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

// Batches of at most 384 rows in exact fp32 take the small-batch kernels (small_step.hip): one launch per stage, no
// grid barrier, no residency requirement.  BLH_OPT_SMALL_STEP 0 or BLH_NO_SMALL_STEP=1 = the multi-launch path of every
// other batch size.  Returns 0 (not applicable) or 1 (staged).
static int small_step_mode(const blh_context* ctx, const blh_model_desc* d, int64_t batch, bool drop_in) {
  // (gemm_dtype 2 / 3 — fp32 accuracy on the 16-bit matrix cores — take the same exact-fp32 kernels here: at 64
  //  rows there is nothing for a matrix core to win, and exact fp32 is what those modes approximate)
  if (!ctx->small_step || d->gemm_dtype == 4 || batch > 384 || ctx->sync.fn) return 0;
  if (ctx->knob(blh::KNOB_NO_SMALL_STEP) || d->width > 1024 || d->in_features > 1024) return 0;
  (void)drop_in;
  return 1;
}

static int small_params(SmallStepParams& p, blh_context* ctx, const blh_model_desc* d, float* params, float* grads,
                        float* exp_avg, float* exp_avg_sq, float* bn_running, int64_t* bn_nbt, const float* x,
                        const float* target, const blh_dropout* drop, float momentum, const blh_adam_hyper* hyper,
                        const blh_step_state* dev_state, const Workspace& ws, float* pred, float* loss_out,
                        float* stats_out, int64_t batch, const float* dpred) {
  const ArenaLayout L = make_layout(d);
  p = SmallStepParams{};
  p.nh = (int)L.heavy.size(); p.W = d->width; p.in_f = d->in_features; p.out_f = d->out_features;
  p.batch = (int)batch;
  for (int i = 0; i < p.nh; ++i) {
    p.w_off[i] = L.heavy[i].w; p.b_off[i] = L.heavy[i].b; p.g_off[i] = L.heavy[i].gamma; p.be_off[i] = L.heavy[i].beta;
    p.A[i] = ws.A[i]; p.dZ[i] = ws.dZ[i]; p.Z[i] = ws.Z[i]; p.bn_saved[i] = ws.bn_saved[i];
  }
  p.gskip[0] = ws.G0; p.gskip[1] = ws.G1;
  p.dec_w = L.dec_w; p.dec_b = L.dec_b; p.count = L.total;
  p.params = params; p.grads = grads; p.m = exp_avg; p.v = exp_avg_sq;
  p.bn_running = bn_running; p.nbt = bn_nbt; p.x = x; p.target = target;
  p.dpred = dpred ? const_cast<float*>(dpred) : ws.dpred; p.pred = pred; p.loss_out = loss_out; p.stats_out = stats_out;
  p.loss_part = ws.loss_part; p.sumsq_part = ws.sumsq_fold;      // (SUMSQ_FOLD_PARTS slots)
  p.bar = nullptr;
  p.drop = layer_drop(ctx, drop, 0, batch, d->width);
  p.momentum = momentum;
  p.denom = (double)batch * d->out_features;
  p.mse_scale = (float)(2.0 / p.denom);
  if (hyper) {
    if (hyper->step < 1) return BLH_ERR_INVALID_ARGUMENT;
    p.adam = adam_consts(*hyper);
  }
  p.st = dev_state;
  return BLH_OK;
}

// The weight gradients of all hidden stages of the staged small-batch backward as ONE GEMM launch: the dZ buffers of
// stages 1 .. nh-1 are consecutive in the workspace, and so are the activations A[0 .. nh-2] and the weight slots of
// the gradient arena (equal strides), so dW_i = dZ_i^T A_{i-1} for all i is the "split-K" form of the ring GEMM
// with one slab per stage, each slab stored to its own output.  Needs whole 32-deep K tiles per stage.
static bool small_wgrad_batched_ok(const blh_model_desc* d, const ArenaLayout& L, const Workspace& ws, int64_t batch) {
  const int nh = (int)L.heavy.size();
  if (nh < 2 || batch % 32 != 0) return false;
  const int64_t act = batch * (int64_t)d->width;
  for (int i = 1; i + 1 < nh; ++i) {
    if (ws.dZ[i + 1] - ws.dZ[i] != act || ws.A[i] - ws.A[i - 1] != act) return false;
    if (L.heavy[i + 1].w - L.heavy[i].w != L.heavy[2].w - L.heavy[1].w) return false;
  }
  return true;
}
// sq (optional): one norm partial per workgroup of the launch, (nh - 1) * (W / 128)^2 of them
static int small_wgrad_batched(hipStream_t s, const blh_model_desc* d, const ArenaLayout& L, const Workspace& ws,
                               float* grads, int64_t batch, double* sq = nullptr) {
  const int nh = (int)L.heavy.size(), W = d->width;
  GemmParams g{};
  g.A = ws.dZ[1]; g.lda = W;
  g.B = ws.A[0]; g.ldb = W;
  g.C = grads + L.heavy[1].w; g.ldc = W;
  g.M = W; g.N = W; g.K = (int)((nh - 1) * batch); g.k_per_split = (int)batch;
  g.c_split_stride = nh > 2 ? L.heavy[2].w - L.heavy[1].w : 0;
  g.sq_part = sq;
  return launch_gemm(s, TILE_128x128, KROW, KROW, sq ? EPI_STORE_SQ : EPI_STORE, g, nh - 1, 0);
}

// the whole step (blh_train_step / blh_train_step_captured)
static int small_train_step(int mode, blh_context* ctx, const blh_model_desc* d, hipStream_t s, float* params,
                            float* grads, float* exp_avg, float* exp_avg_sq, float* bn_running, int64_t* bn_nbt,
                            const float* x, const float* target, const blh_dropout* drop, float momentum,
                            const blh_adam_hyper* hyper, const blh_step_state* dev_state, const Workspace& ws,
                            float* pred, float* loss_out, float* stats_out, int64_t batch) {
  SmallStepParams p;
  BLH_TRY(small_params(p, ctx, d, params, grads, exp_avg, exp_avg_sq, bn_running, bn_nbt, x, target, drop, momentum,
                       hyper, dev_state, ws, pred, loss_out, stats_out, batch, nullptr));
  ctx->note_saved(ws.Z[0], batch, blh_context::SAVED_NONE);     // (nothing a later blh_backward could use)
  BLH_TRY(launch_small_forward_staged(s, p, true));
  const ArenaLayout L = make_layout(d);
  // gradient-norm partials: [0, W/4) the stage kernels (accumulated over the stages), [W/4, W/4 + out/4) decode, then
  // one per workgroup of the batched weight-gradient GEMM
  int nparts = d->width / 4 + d->out_features / 4;
  const int tiles = (int)(ceil_div(d->width, 128) * ceil_div(d->width, 128));
  if (small_wgrad_batched_ok(d, L, ws, batch) && nparts + (p.nh - 1) * tiles <= SUMSQ_FOLD_PARTS) {
    BLH_TRY(launch_small_backward_staged(s, p, false, false));
    BLH_TRY(small_wgrad_batched(s, d, L, ws, grads, batch, p.sumsq_part + nparts));
    nparts += (p.nh - 1) * tiles;
  } else {
    BLH_TRY(launch_small_backward_staged(s, p, false, true));
  }
  const LossFinish lf{ws.loss_part, d->out_features / 4, p.denom, loss_out};
  if (dev_state)
    return launch_clip_adam_dev(s, params, grads, exp_avg, exp_avg_sq, p.count, dev_state, p.sumsq_part, nparts,
                                stats_out, lf);
  return launch_clip_adam(s, params, grads, exp_avg, exp_avg_sq, p.count, *hyper, p.sumsq_part, nparts, stats_out, lf);
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
  if (small_step_mode(ctx, d, batch, true)) {
    // the drop-in forward at small batch; what it saves for backward is in the small-batch format
    SmallStepParams p;
    BLH_TRY(small_params(p, ctx, d, const_cast<float*>(params), nullptr, nullptr, nullptr, bn_running, bn_nbt, x, nullptr,
                         drop, momentum, nullptr, nullptr, ws, pred, nullptr, nullptr, batch, nullptr));
    BLH_TRY(launch_small_forward_staged((hipStream_t)stream, p, false));
    ctx->note_saved(workspace, batch, blh_context::SAVED_SMALL);
    return BLH_OK;
  }
  return forward_impl(ctx, d, (hipStream_t)stream, params, bn_running, bn_nbt, x, drop, momentum, ws,
                      pred, batch, true, nullptr, 0.f, nullptr, nullptr);
}

}  // extern "C"

// blh_forward_train_loss with two switches for the step that owns the whole call sequence (comm.hip: blh_train_step_dp):
// shadow_valid — bf16 storage: the bf16 parameter image in the workspace is up to date (the previous step's Adam
// kernel wrote it), skip the re-cast; pending != NULL — do NOT launch loss_finalize: the partial sums, their count and
// the denominator are returned for the caller to finalise where it costs nothing (off the main stream).
int blh::forward_train_loss_core(blh_context* ctx, const blh_model_desc* d, hipStream_t s, const float* params,
                                 float* bn_running, int64_t* bn_nbt, const float* x, const float* target,
                                 const blh_dropout* drop, float momentum, void* workspace, int64_t workspace_bytes,
                                 float* pred, float* loss_out, int64_t batch, bool shadow_valid, PendingLoss* pending) {
  BLH_TRY(check_common(ctx, d, workspace, workspace_bytes, batch));
  BLH_TRY(check_drop(drop));
  if (!params || !bn_running || !bn_nbt || !x || !target || !pred || !loss_out) return BLH_ERR_INVALID_ARGUMENT;
  if (batch < 2) return BLH_ERR_SHAPE;
  const double denom = (double)batch * d->out_features;
  int nparts = 0;
  ctx->loss_batch = 0;
  if (d->gemm_dtype == 4) {
    const WorkspaceH wh = carve_h(d, batch, workspace);
    int dec_S = 0;
    BLH_TRY(forward_h(ctx, d, s, params, bn_running, bn_nbt, x, drop, momentum, wh, pred, batch, true, shadow_valid,
                      target, (float)(2.0 / denom), &dec_S));
    if (dec_S > 0) nparts = dec_S;
    else BLH_TRY(launch_mse(s, pred, target, batch * d->out_features, (float)(2.0 / denom), wh.dpred,
                            wh.loss_part, &nparts));
    if (pending) *pending = PendingLoss{wh.loss_part, nparts, denom};
    else BLH_TRY(launch_loss_finalize(s, wh.loss_part, nparts, denom, loss_out));
    ctx->loss_batch = batch; ctx->loss_nparts = dec_S;
    return BLH_OK;
  }
  const Workspace ws = carve(d, batch, workspace);
  BLH_TRY(forward_impl(ctx, d, s, params, bn_running, bn_nbt, x, drop, momentum, ws, pred, batch, true,
                       target, (float)(2.0 / denom), ws.loss_part, &nparts));
  if (pending) *pending = PendingLoss{ws.loss_part, nparts, denom};
  else BLH_TRY(launch_loss_finalize(s, ws.loss_part, nparts, denom, loss_out));
  ctx->loss_batch = batch; ctx->loss_nparts = nparts;
  return BLH_OK;
}

extern "C" {

int blh_forward_train_loss(blh_context* ctx, const blh_model_desc* d, void* stream, const float* params,
                           float* bn_running, int64_t* bn_nbt, const float* x, const float* target,
                           const blh_dropout* drop, float momentum, void* workspace,
                           int64_t workspace_bytes, float* pred, float* loss_out, int64_t batch) {
  return forward_train_loss_core(ctx, d, (hipStream_t)stream, params, bn_running, bn_nbt, x, target, drop, momentum,
                                 workspace, workspace_bytes, pred, loss_out, batch, false, nullptr);
}

int blh_forward_eval(blh_context* ctx, const blh_model_desc* d, void* stream, const float* params,
                     const float* bn_running, const float* x, void* workspace,
                     int64_t workspace_bytes, float* pred, int64_t batch) {
  BLH_TRY(check_common(ctx, d, workspace, workspace_bytes, batch));
  if (!params || !bn_running || !x || !pred) return BLH_ERR_INVALID_ARGUMENT;
  blh_dropout none{nullptr, 0, 0, 0, 0, 0};
  ctx->note_saved(workspace, batch, blh_context::SAVED_NONE);      // (the activation buffers are overwritten)
  if (d->gemm_dtype == 4)
    return forward_h(ctx, d, (hipStream_t)stream, params, const_cast<float*>(bn_running), nullptr, x,
                     &none, 0.f, carve_h(d, batch, workspace), pred, batch, false);
  const Workspace ws = carve(d, batch, workspace);
  if (small_step_mode(ctx, d, batch, false)) {
    // serving-sized batches: one launch per stage (7 launches instead of ~17)
    SmallStepParams p;
    BLH_TRY(small_params(p, ctx, d, const_cast<float*>(params), nullptr, nullptr, nullptr, const_cast<float*>(bn_running),
                         nullptr, x, nullptr, &none, 0.f, nullptr, nullptr, ws, pred, nullptr, nullptr, batch, nullptr));
    return launch_small_eval_staged((hipStream_t)stream, p);
  }
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

// BLH_OPT_BUCKET_FLOATS: adjacent ready ranges (backward walks the arena downwards) are merged until a bucket holds at
// least `want` elements; what is pending when backward returns is reported then.  Ranges that complete together (the
// stages of one batched weight-gradient launch) arrive as ONE range and are not cut into buckets: nothing could
// overlap the cut.
namespace {
struct BucketMerger {
  blh_grad_ready_fn fn; void* user; int64_t want; int64_t lo = 0, hi = 0; bool pending = false;
  void flush() { if (pending) { fn(user, lo, hi - lo); pending = false; } }
  static void thunk(void* self, int64_t off, int64_t cnt) {
    BucketMerger* m = static_cast<BucketMerger*>(self);
    const int64_t lo = off, hi = off + cnt;
    if (!m->pending) { m->lo = lo; m->hi = hi; m->pending = true; }
    else if (hi == m->lo) m->lo = lo;
    else if (lo == m->hi) m->hi = hi;
    else { m->flush(); m->lo = lo; m->hi = hi; m->pending = true; }
    // (a full bucket is held back while what is left below it — the arena starts with the encode stage — is a small
    //  fraction of a bucket: that remainder would be a collective of its own at the very end of backward, where
    //  every collective costs its launch and a queue hop: 22 us at configs[2], profiles/r05_dp_overhead.md)
    if (m->hi - m->lo >= m->want && !(m->lo > 0 && m->lo < m->want / 8)) m->flush();
  }
};
}  // namespace

static int backward_unmerged(blh_context* ctx, const blh_model_desc* d, void* stream, const float* params, const float* x,
                             const blh_dropout* drop, void* workspace, int64_t workspace_bytes,
                             const float* dpred, float* grads, int64_t batch, blh_grad_ready_fn on_ready,
                             void* user);

int blh_backward(blh_context* ctx, const blh_model_desc* d, void* stream, const float* params, const float* x,
                 const blh_dropout* drop, void* workspace, int64_t workspace_bytes,
                 const float* dpred, float* grads, int64_t batch, blh_grad_ready_fn on_ready,
                 void* user) {
  if (ctx && on_ready && ctx->bucket_floats > 0) {
    BucketMerger m{on_ready, user, ctx->bucket_floats};
    const int rc = backward_unmerged(ctx, d, stream, params, x, drop, workspace, workspace_bytes, dpred, grads, batch,
                                     &BucketMerger::thunk, &m);
    if (rc == BLH_OK) m.flush();
    return rc;
  }
  return backward_unmerged(ctx, d, stream, params, x, drop, workspace, workspace_bytes, dpred, grads, batch, on_ready,
                           user);
}

static int backward_unmerged(blh_context* ctx, const blh_model_desc* d, void* stream, const float* params, const float* x,
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
  // which forward saved what is in this workspace (step.h: the record of this workspace, else what a forward with the
  // context's present options would save — refused when that is ambiguous).  A format whose backward has no SyncBN
  // form (the encode stage without Z0, the small-batch kernels) is refused under a SyncBN call.
  if (d->gemm_dtype == 4) {
    const int fmt = ctx->saved_format(workspace, batch, enc_fused_ok_h(ctx, d, batch) ? blh_context::SAVED_ENC_FUSED
                                                                                     : blh_context::SAVED_MULTI);
    if (fmt == blh_context::SAVED_NONE || (fmt == blh_context::SAVED_ENC_FUSED && ctx->sync.fn))
      return BLH_ERR_INVALID_ARGUMENT;
    const WorkspaceH wh = carve_h(d, batch, workspace);
    return backward_h(ctx, d, (hipStream_t)stream, params, drop, wh, from_loss ? wh.dpred : dpred, grads,
                      batch, on_ready, user, fmt, from_loss ? loss_nparts : 0);
  }
  const Workspace ws = carve(d, batch, workspace);
  const int predicted = small_step_mode(ctx, d, batch, true) ? blh_context::SAVED_SMALL
                        : (enc_fused_ok(ctx, d, batch) ? blh_context::SAVED_ENC_FUSED : blh_context::SAVED_MULTI);
  const int fmt = ctx->saved_format(workspace, batch, predicted);
  if (fmt == blh_context::SAVED_NONE || (fmt == blh_context::SAVED_ENC_FUSED && ctx->sync.fn))
    return BLH_ERR_INVALID_ARGUMENT;
  if (fmt == blh_context::SAVED_SMALL) {
    // the activations in this workspace were saved by the small-batch forward: only its backward can read them
    if (from_loss || ctx->sync.fn) return BLH_ERR_INVALID_ARGUMENT;
    // (the small-batch backward works IN the buffers it reads — x-hat, the gates: a second backward of the same
    //  forward, e.g. retain_graph, would read its own output; it is refused instead)
    ctx->note_saved(workspace, batch, blh_context::SAVED_NONE);
    SmallStepParams p;
    BLH_TRY(small_params(p, ctx, d, const_cast<float*>(params), grads, nullptr, nullptr, nullptr, nullptr, x, nullptr, drop,
                         0.f, nullptr, nullptr, ws, nullptr, nullptr, nullptr, batch, dpred));
    const ArenaLayout L = make_layout(d);
    const bool batched = small_wgrad_batched_ok(d, L, ws, batch);
    BLH_TRY(launch_small_backward_staged((hipStream_t)stream, p, true, !batched));
    if (batched) BLH_TRY(small_wgrad_batched((hipStream_t)stream, d, L, ws, grads, batch));
    if (on_ready) on_ready(user, 0, make_layout(d).total);     // every range at once
    return BLH_OK;
  }
  if (from_loss && loss_nparts > 0) {   // decode-bias partials of the fused decode kernel
    const FusedBackward fb{loss_nparts, nullptr, nullptr, nullptr};
    return backward_impl(ctx, d, (hipStream_t)stream, params, x, drop, ws, ws.dpred, grads, batch,
                         on_ready, user, fmt, &fb);
  }
  return backward_impl(ctx, d, (hipStream_t)stream, params, x, drop, ws, from_loss ? ws.dpred : dpred,
                       grads, batch, on_ready, user, fmt);
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
  const ArenaLayout L = make_layout(d);
  const bool wdT = !ctx->knob(blh::KNOB_NO_DECODE_FUSE) && decode_fused_h_supported(batch, d->width, d->out_features);
  // (with the decode weight's K-major image when this shape takes the one-pass decode; the second source of the
  //  launch is a dummy: the first 4 parameters onto themselves)
  BLH_TRY(launch_cast2_f32_bf16((hipStream_t)stream, params, wh.wsh, L.total, params, wh.wsh, 4,
                                wdT ? params + L.dec_w : nullptr, wdT ? wh.wdT : nullptr, d->width, d->out_features));
  if (ctx->persistent_shadow) { ctx->shadow_params = params; ctx->shadow_ws = workspace; ctx->shadow_wdT = wdT; }
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
    const bool wdT_was = ctx->shadow_wdT;
    int dec_S = 0;
    BLH_TRY(forward_h(ctx, d, s, params, bn_running, bn_nbt, x, drop, momentum, wh, pred, batch, true, valid,
                      target, (float)(2.0 / denom), &dec_S));
    if (dec_S > 0) nparts = dec_S;
    else BLH_TRY(launch_mse(s, pred, target, batch * d->out_features, (float)(2.0 / denom), wh.dpred,
                            wh.loss_part, &nparts));
    BLH_TRY(backward_h(ctx, d, s, params, drop, wh, wh.dpred, grads, batch, nullptr, nullptr, ctx->fwd_mode, dec_S, &np));
    ctx->note_saved(workspace, batch, blh_context::SAVED_NONE);
    // (np > 0: the batched slab sum left the norm's partials — step_bf16s.hip; else one pass over the arena)
    if (np == 0) BLH_TRY(launch_sumsq(s, grads, count, wh.sumsq_part, &np));
    const bool wdT = keep && !ctx->knob(blh::KNOB_NO_DECODE_FUSE) &&
                     decode_fused_h_supported(batch, d->width, d->out_features);
    const ShadowDst sd = keep ? ShadowDst{wh.wsh, wdT ? wh.wdT : nullptr, make_layout(d).dec_w, d->width, d->out_features}
                              : NO_SHADOW;
    BLH_TRY(launch_clip_adam(s, params, grads, exp_avg, exp_avg_sq, count, *hyper, wh.sumsq_part, np,
                             stats_out, LossFinish{wh.loss_part, nparts, denom, loss_out}, sd));
    // (the K-major image is complete only if an earlier launch zeroed its padding rows: the first step's cast does)
    if (keep) { ctx->shadow_params = params; ctx->shadow_ws = workspace; ctx->shadow_wdT = wdT && (valid ? wdT_was : true); }
    return BLH_OK;
  }
  const Workspace ws = carve(d, batch, workspace);
  if (const int mode = small_step_mode(ctx, d, batch, false))
    return small_train_step(mode, ctx, d, s, params, grads, exp_avg, exp_avg_sq, bn_running, bn_nbt, x, target, drop,
                            momentum, hyper, nullptr, ws, pred, loss_out, stats_out, batch);
  BLH_TRY(forward_impl(ctx, d, s, params, bn_running, bn_nbt, x, drop, momentum, ws, pred, batch, true,
                       target, (float)(2.0 / denom), ws.loss_part, &nparts));
  int np = 0;
  double* sq_src = ws.sumsq_part;
  const FusedBackward fb{nparts, ws.sumsq_part, &np, &sq_src};
  BLH_TRY(backward_impl(ctx, d, s, params, x, drop, ws, ws.dpred, grads, batch, nullptr, nullptr, ctx->fwd_mode, &fb));
  ctx->note_saved(workspace, batch, blh_context::SAVED_NONE);
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
    const bool wdT_was = ctx->shadow_wdT && ctx->shadow_params == params && ctx->shadow_ws == workspace;
    if (keep && !wdT_was) ctx->shadow_wdT = false;
    int dec_S = 0;
    BLH_TRY(forward_h(ctx, d, s, params, bn_running, bn_nbt, x, drop, momentum, wh, pred, batch, true, keep,
                      target, (float)(2.0 / denom), &dec_S));
    if (dec_S > 0) nparts = dec_S;
    else BLH_TRY(launch_mse(s, pred, target, batch * d->out_features, (float)(2.0 / denom), wh.dpred,
                            wh.loss_part, &nparts));
    BLH_TRY(backward_h(ctx, d, s, params, drop, wh, wh.dpred, grads, batch, nullptr, nullptr, ctx->fwd_mode, dec_S, &np));
    ctx->note_saved(workspace, batch, blh_context::SAVED_NONE);
    // (np > 0: the batched slab sum left the norm's partials — step_bf16s.hip; else one pass over the arena)
    if (np == 0) BLH_TRY(launch_sumsq(s, grads, count, wh.sumsq_part, &np));
    // (captured: the image state is what blh_refresh_param_shadow left — plain + K-major decode weight — and every
    //  replay keeps both up to date)
    const bool wdT = keep && wdT_was;
    const ShadowDst sd = keep ? ShadowDst{wh.wsh, wdT ? wh.wdT : nullptr, make_layout(d).dec_w, d->width, d->out_features}
                              : NO_SHADOW;
    const int rc = launch_clip_adam_dev(s, params, grads, exp_avg, exp_avg_sq, count, dev_state, wh.sumsq_part, np,
                                        stats_out, LossFinish{wh.loss_part, nparts, denom, loss_out}, sd);
    if (keep) { ctx->shadow_params = params; ctx->shadow_ws = workspace; ctx->shadow_wdT = wdT; }
    return rc;
  }
  const Workspace ws = carve(d, batch, workspace);
  if (const int mode = small_step_mode(ctx, d, batch, false))
    return small_train_step(mode, ctx, d, s, params, grads, exp_avg, exp_avg_sq, bn_running, bn_nbt, x, target, drop,
                            momentum, nullptr, dev_state, ws, pred, loss_out, stats_out, batch);
  BLH_TRY(forward_impl(ctx, d, s, params, bn_running, bn_nbt, x, drop, momentum, ws, pred, batch, true,
                       target, (float)(2.0 / denom), ws.loss_part, &nparts));
  double* sq_src = ws.sumsq_part;
  const FusedBackward fb{nparts, ws.sumsq_part, &np, &sq_src};
  BLH_TRY(backward_impl(ctx, d, s, params, x, drop, ws, ws.dpred, grads, batch, nullptr, nullptr, ctx->fwd_mode, &fb));
  ctx->note_saved(workspace, batch, blh_context::SAVED_NONE);
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

// The encode stage as the exact-fp32 step runs it at more than 384 rows (encode_f32.hip): forward x -> A0 + the
// stage's keep AND gate bits + saved statistics, backward dA0 -> dW0, db0, dgamma, dbeta.  `scratch`: batch * width floats.
int blh_skinny_encode_fused_fwd(void* stream, const float* x, const float* W0, const float* b0, const float* gamma,
                                const float* beta, float* running_mean, float* running_var, int64_t* nbt,
                                float momentum, float* saved, float* scratch, float* A, uint32_t* keepbits,
                                const blh_dropout* drop, int64_t batch, int32_t width, int32_t in_features) {
  if (!x || !W0 || !b0 || !gamma || !beta || !running_mean || !running_var || !nbt || !saved || !scratch || !A ||
      !keepbits || !drop || batch < 2)
    return BLH_ERR_INVALID_ARGUMENT;
  if (!enc_fused_supported(batch, width, in_features)) return BLH_ERR_SHAPE;
  DropoutSrc d;
  d.step_dev = nullptr; d.keep = drop->keep_mask; d.seed = drop->seed; d.step = drop->step;
  d.row_offset = drop->row_offset; d.layer = drop->layer_base;
  return launch_enc_forward((hipStream_t)stream, x, W0, b0, gamma, beta, running_mean, running_var, nbt, momentum, saved,
                            scratch, A, keepbits, batch, width, d);
}

int blh_skinny_encode_fused_bwd(void* stream, const float* dA, const float* x, const float* W0, const float* b0,
                                const float* saved, const uint32_t* keepbits, float* scratch, float* dW0, float* db0,
                                float* dgamma, float* dbeta, int64_t batch, int32_t width, int32_t in_features) {
  if (!dA || !x || !W0 || !b0 || !saved || !keepbits || !scratch || !dW0 || !db0 || !dgamma || !dbeta || batch < 2)
    return BLH_ERR_INVALID_ARGUMENT;
  if (!enc_fused_supported(batch, width, in_features)) return BLH_ERR_SHAPE;
  return launch_enc_backward((hipStream_t)stream, dA, x, W0, b0, saved, keepbits, scratch, batch, width, dW0, dgamma,
                             dbeta, db0, 1, nullptr, nullptr);
}

int blh_skinny_encode_fused_fwd_bf16(void* stream, const uint16_t* x, const uint16_t* W0, const float* b0,
                                     const float* gamma, const float* beta, float* running_mean, float* running_var,
                                     int64_t* nbt, float momentum, float* saved, uint16_t* scratch, uint16_t* A,
                                     uint32_t* keepbits, const blh_dropout* drop, int64_t batch, int32_t width,
                                     int32_t in_features) {
  if (!x || !W0 || !b0 || !gamma || !beta || !running_mean || !running_var || !nbt || !saved || !scratch || !A ||
      !keepbits || !drop || batch < 2)
    return BLH_ERR_INVALID_ARGUMENT;
  if (!enc_fused_supported_h(batch, width, in_features)) return BLH_ERR_SHAPE;
  DropoutSrc d;
  d.step_dev = nullptr; d.keep = drop->keep_mask; d.seed = drop->seed; d.step = drop->step;
  d.row_offset = drop->row_offset; d.layer = drop->layer_base;
  return launch_enc_forward_h((hipStream_t)stream, const_cast<uint16_t*>(x), nullptr, W0, b0, gamma, beta, running_mean, running_var, nbt, momentum,
                              saved, scratch, A, keepbits, batch, width, d);
}

int blh_skinny_encode_fused_bwd_bf16(void* stream, const uint16_t* dA, const uint16_t* x, const uint16_t* W0,
                                     const float* b0, const float* saved, const uint32_t* keepbits, uint16_t* scratch,
                                     float* dW0, float* db0, float* dgamma, float* dbeta, int64_t batch, int32_t width,
                                     int32_t in_features) {
  if (!dA || !x || !W0 || !b0 || !saved || !keepbits || !scratch || !dW0 || !db0 || !dgamma || !dbeta || batch < 2)
    return BLH_ERR_INVALID_ARGUMENT;
  if (!enc_fused_supported_h(batch, width, in_features)) return BLH_ERR_SHAPE;
  return launch_enc_backward_h((hipStream_t)stream, dA, x, W0, b0, saved, keepbits, scratch, batch, width, dW0, dgamma,
                               dbeta, db0, 1);
}

int blh_skinny_decode_fused(void* stream, const float* A, const float* Wd, const float* bd,
                            const float* target, float* pred, float* dpred, float* dA, float* loss_out,
                            void* workspace, int64_t workspace_bytes, int64_t batch, int32_t width,
                            int32_t out_features) {
  if (!A || !Wd || !bd || !target || !pred || !dpred || !dA || !workspace || batch <= 0)
    return BLH_ERR_INVALID_ARGUMENT;
  if (workspace_bytes < blh_skinny_workspace_bytes(batch, width, 32, out_features)) return BLH_ERR_WORKSPACE;
  if (!decode_fused_supported(batch, width, out_features)) return BLH_ERR_SHAPE;
  float* part = (float*)workspace;              // [1026*OF] bias partials, then loss partials
  float* loss_part = part + 1026 * (int64_t)out_features;
  const double denom = (double)batch * out_features;
  int np = 0;
  BLH_TRY(launch_decode_fused((hipStream_t)stream, A, Wd, bd, target, pred, dpred, dA, loss_part, part, batch, width,
                              out_features, (float)(2.0 / denom), &np));
  return loss_out ? launch_loss_finalize((hipStream_t)stream, loss_part, np, denom, loss_out) : BLH_OK;
}

// bf16 storage: workspace = [Wd bf16 | 64 pad | WdT image | dpred bf16 | decode-bias partials | loss partials]
static int64_t dec_h_ws_off(int64_t batch, int W, int OF, int which) {
  int64_t off[6];
  off[0] = 0;                                                        // Wd image, OF * W bf16
  off[1] = off[0] + round_up((int64_t)OF * W * 2, 256);              // scratch for the second tensor of the cast launch
  off[2] = off[1] + 256;                                             // WdT image, W * 64 bf16
  off[3] = off[2] + round_up((int64_t)W * 64 * 2, 256);              // dpred bf16
  off[4] = off[3] + round_up(batch * OF * 2, 256);                   // decode-bias partials (1026 * OF floats)
  off[5] = off[4] + round_up((int64_t)1026 * OF * 4 + 4096 * 4, 256);   // end
  return off[which];
}

int64_t blh_skinny_decode_fused_bf16_workspace_bytes(int64_t batch, int32_t width, int32_t out_features) {
  if (batch <= 0 || width <= 0 || out_features <= 0) return BLH_ERR_INVALID_ARGUMENT;
  return dec_h_ws_off(batch, width, out_features, 5);
}

int blh_skinny_decode_fused_bf16(void* stream, const uint16_t* A, const float* Wd, const float* bd,
                                 const float* target, float* pred, float* dpred, uint16_t* dA, float* loss_out,
                                 void* workspace, int64_t workspace_bytes, int64_t batch, int32_t width,
                                 int32_t out_features) {
  if (!A || !Wd || !bd || !target || !pred || !dpred || !dA || !workspace || batch <= 0)
    return BLH_ERR_INVALID_ARGUMENT;
  if (!decode_fused_h_supported(batch, width, out_features)) return BLH_ERR_SHAPE;
  if (workspace_bytes < dec_h_ws_off(batch, width, out_features, 5)) return BLH_ERR_WORKSPACE;
  hipStream_t s = (hipStream_t)stream;
  char* w = (char*)workspace;
  const int W = width, OF = out_features;
  uint16_t* wdh = (uint16_t*)(w + dec_h_ws_off(batch, W, OF, 0));
  uint16_t* pad = (uint16_t*)(w + dec_h_ws_off(batch, W, OF, 1));
  uint16_t* wdT = (uint16_t*)(w + dec_h_ws_off(batch, W, OF, 2));
  uint16_t* dph = (uint16_t*)(w + dec_h_ws_off(batch, W, OF, 3));
  float* part = (float*)(w + dec_h_ws_off(batch, W, OF, 4));
  float* loss_part = part + 1026 * (int64_t)OF;
  // the decode weight's two bf16 images in one launch (the bias rides along as the launch's second tensor)
  BLH_TRY(launch_cast2_f32_bf16(s, Wd, wdh, (int64_t)OF * W, bd, pad, OF, Wd, wdT, W, OF));
  const double denom = (double)batch * OF;
  int np = 0;
  BLH_TRY(launch_decode_fused_h(s, A, wdh, wdT, bd, target, pred, dpred, dph, dA, loss_part, part, batch, W, OF,
                                (float)(2.0 / denom), &np));
  return loss_out ? launch_loss_finalize(s, loss_part, np, denom, loss_out) : BLH_OK;
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
