// Collectives the library issues itself (round 6, opt-in): the data-parallel step as ONE call.
//
// New work — the reference is single-device (/root/reference/util/config.py:17).  The default data-parallel driver
// (bilinear_amd/dp.py) launches every bucket all-reduce from a Python hook through torch.distributed's process group:
// its own pool stream, work.wait() events, three cross-queue hops in the tail (profiles/r05_dp_overhead.md).  Here the
// library talks to RCCL directly: librccl.so is resolved at run time (dlopen of the copy the process already holds —
// inside a PyTorch process that is torch's), one communicator per blh_comm, and blh_train_step_dp enqueues
//
//   forward + MSE -> bucket k all-reduce behind the kernel that completes it -> ... -> [loss | last bucket] (one RCCL group)
//   -> gradient norm + clip + Adam right behind it, on the stream that carries it -> one join
//
// with no return to the host language per bucket.  Bucket boundaries are the ones blh_backward reports (BucketMerger,
// BLH_OPT_BUCKET_FLOATS), the averaged gradients and therefore every result equal the torch-driven step bit for bit at
// equal world size (tests/test_gpu_dp.py).  N > 1 ranks have never been available to this repository: the path is
// executed at world size 1 with every collective issued, like the torch one.
#include <dlfcn.h>

#include <cstring>
#include <mutex>
#include <new>

#include "step.h"

using namespace blh;

namespace {

// ---- the slice of rccl.h this file needs (rccl.h itself is not included: the library must load without librccl) ----
typedef struct ncclComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
enum { NCCL_SUCCESS = 0 };
enum { NCCL_SUM = 0, NCCL_AVG = 4 };
enum { NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8, NCCL_BFLOAT16 = 9, NCCL_UINT8 = 1 };

struct Rccl {
  void* handle = nullptr;
  int (*GetVersion)(int*) = nullptr;
  int (*GetUniqueId)(ncclUniqueId*) = nullptr;
  int (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  int (*CommDestroy)(ncclComm_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t) = nullptr;
  int (*GroupStart)() = nullptr;
  int (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(int) = nullptr;
  bool ok = false;
};

std::once_flag g_rccl_once;
Rccl g_rccl;
char g_rccl_load_error[256] = "";        // why librccl could not be used (written once, inside call_once)
thread_local char tl_comm_error[256] = "";

void set_error(const char* what, int code) {
  const char* s = (g_rccl.GetErrorString && code >= 0) ? g_rccl.GetErrorString(code) : "";
  snprintf(tl_comm_error, sizeof(tl_comm_error), "%s%s%s (ncclResult_t %d)", what, s[0] ? ": " : "", s, code);
}

void load_rccl() {
  // the copy already in the process first (torch's librccl.so: one RCCL per process), then the loader's search path,
  // then ROCm's; BLH_RCCL_PATH names a file explicitly
  const char* env = getenv("BLH_RCCL_PATH");
  const char* names[] = {env, "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
  void* h = nullptr;
  for (int pass = 0; pass < 2 && !h; ++pass)
    for (const char* n : names) {
      if (!n) continue;
      h = dlopen(n, RTLD_NOW | RTLD_LOCAL | (pass == 0 ? RTLD_NOLOAD : 0));
      if (h) break;
    }
  if (!h) { snprintf(g_rccl_load_error, sizeof(g_rccl_load_error), "librccl.so not found (%s)", dlerror()); return; }
  Rccl r;
  r.handle = h;
  bool all = true;
  auto sym = [&](const char* n) { void* p = dlsym(h, n); if (!p) all = false; return p; };
  r.GetVersion = (int (*)(int*))sym("ncclGetVersion");
  r.GetUniqueId = (int (*)(ncclUniqueId*))sym("ncclGetUniqueId");
  r.CommInitRank = (int (*)(ncclComm_t*, int, ncclUniqueId, int))sym("ncclCommInitRank");
  r.CommDestroy = (int (*)(ncclComm_t))sym("ncclCommDestroy");
  r.AllReduce = (int (*)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t))sym("ncclAllReduce");
  r.Broadcast = (int (*)(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t))sym("ncclBroadcast");
  r.GroupStart = (int (*)())sym("ncclGroupStart");
  r.GroupEnd = (int (*)())sym("ncclGroupEnd");
  r.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
  if (!all) { snprintf(g_rccl_load_error, sizeof(g_rccl_load_error), "librccl.so lacks a required symbol"); return; }
  r.ok = true;
  g_rccl = r;
}

const Rccl* rccl() {
  std::call_once(g_rccl_once, load_rccl);
  if (g_rccl.ok) return &g_rccl;
  snprintf(tl_comm_error, sizeof(tl_comm_error), "%s", g_rccl_load_error);     // (every thread that asks gets the text)
  return nullptr;
}

#define BLH_NCCL_TRY(what, expr)                                  \
  do {                                                            \
    const int _r = (expr);                                        \
    if (_r != NCCL_SUCCESS) { set_error(what, _r); return BLH_ERR_COMM; } \
  } while (0)

int nccl_dtype(int32_t dtype) {
  switch (dtype) {
    case 0: return NCCL_FLOAT32;
    case 1: return NCCL_FLOAT64;
    case 2: return NCCL_BFLOAT16;
  }
  return -1;
}

constexpr int COMM_EVENTS = 40;   // one per bucket of a step (at most one per stage + decode), reused every step

}  // namespace

// One communicator on one device: the collective stream (normal priority: the driver's compute stream is high, the
// library's side stream lowest — three levels, three hardware queues) and the events that order a
// bucket behind its producer.
struct blh_comm {
  ncclComm_t comm = nullptr;
  int device = -1, world = 0, rank = 0, version = 0;
  hipStream_t cs = nullptr;
  hipEvent_t ev[COMM_EVENTS] = {};
  hipEvent_t ev_cs = nullptr, ev_tail = nullptr;
  int next_ev = 0;
  int64_t collectives = 0;      // issued so far (tests: every bucket went out)
  uint16_t* half = nullptr;     // blh_comm_set_bf16_buffer: the caller's bf16 image of the gradient arena (bf16 buckets)
  int64_t half_count = 0;
};

namespace {

// state of one blh_train_step_dp call, handed to blh_backward's hook
struct DpCall {
  blh_comm* c;
  hipStream_t producer;     // the stream a reported range is complete on
  float* grads;
  float* loss;              // goes out with the LAST bucket (one RCCL group = one launch): that one is ordered behind
                            // everything the main stream ran, the first is not (the one-pass decode carries the
                            // side stream's fork as its completion signal, loss_finalize runs behind it)
  bool tail_on_producer;    // the last bucket (offset 0) rides the producer stream itself
  uint16_t* half;           // bf16 buckets: a bucket is cast into this image and exchanged in bf16 (half the wire bytes)
  PendingLoss pend;         // the forward's loss partials, not yet finalised: the first bucket's turn on the producer
  float* loss_dst;          // stream does it (behind the decode kernel that wrote them; nothing waits for it there)
  bool used_cs = false;
  bool last_seen = false;
  int status = BLH_OK;
};

// half != NULL: the bucket is rounded to bf16 (round to nearest even, the library's cast kernel) on the stream that carries
// the collective and averaged in bf16; the fp32 values stay where they are (clip + Adam read the bf16 image)
int all_reduce_avg_f32(blh_comm* c, hipStream_t st, float* buf, int64_t count, float* extra, uint16_t* half = nullptr) {
  const Rccl* r = rccl();
  if (half) BLH_TRY(launch_cast_f32_bf16(st, buf, half, count));
  if (extra) {
    BLH_NCCL_TRY("ncclGroupStart", r->GroupStart());
    BLH_NCCL_TRY("ncclAllReduce", r->AllReduce(extra, extra, 1, NCCL_FLOAT32, NCCL_AVG, c->comm, st));
  }
  if (half) BLH_NCCL_TRY("ncclAllReduce", r->AllReduce(half, half, (size_t)count, NCCL_BFLOAT16, NCCL_AVG, c->comm, st));
  else
  BLH_NCCL_TRY("ncclAllReduce", r->AllReduce(buf, buf, (size_t)count, NCCL_FLOAT32, NCCL_AVG, c->comm, st));
  if (extra) BLH_NCCL_TRY("ncclGroupEnd", r->GroupEnd());
  c->collectives += extra ? 2 : 1;
  return BLH_OK;
}

int bucket_ready(DpCall* k, int64_t off, int64_t cnt) {
  blh_comm* c = k->c;
  const bool last = off == 0;            // backward walks the arena downwards: the encode stage closes it
  if (k->pend.part) {
    BLH_TRY(launch_loss_finalize(k->producer, k->pend.part, k->pend.n, k->pend.denom, k->loss_dst));
    k->pend.part = nullptr;
  }
  float* extra = last ? k->loss : nullptr;
  if (last) { k->last_seen = true; k->loss = nullptr; }
  if (last && k->tail_on_producer) {
    // nothing else will follow on the producer stream but the optimiser: the collective goes in line, no event hop
    return all_reduce_avg_f32(c, k->producer, k->grads + off, cnt, extra, k->half ? k->half + off : nullptr);
  }
  hipEvent_t e = c->ev[c->next_ev];
  c->next_ev = (c->next_ev + 1) % COMM_EVENTS;
  BLH_HIP_TRY(hipEventRecord(e, k->producer));
  BLH_HIP_TRY(hipStreamWaitEvent(c->cs, e, 0));
  k->used_cs = true;
  return all_reduce_avg_f32(c, c->cs, k->grads + off, cnt, extra, k->half ? k->half + off : nullptr);
}

void bucket_thunk(void* user, int64_t off, int64_t cnt) {
  DpCall* k = static_cast<DpCall*>(user);
  if (k->status != BLH_OK) return;       // (an error must not unwind through blh_backward: remembered, returned after)
  k->status = bucket_ready(k, off, cnt);
}

}  // namespace

extern "C" {

const char* blh_comm_last_error(void) { return tl_comm_error; }

int blh_rccl_version(void) {
  const Rccl* r = rccl();
  int v = 0;
  if (!r || r->GetVersion(&v) != NCCL_SUCCESS) return 0;
  return v;
}

int blh_rccl_unique_id(void* id_out, int64_t id_bytes) {
  if (!id_out || id_bytes != (int64_t)sizeof(ncclUniqueId)) return BLH_ERR_INVALID_ARGUMENT;
  const Rccl* r = rccl();
  if (!r) return BLH_ERR_COMM;
  ncclUniqueId id;
  BLH_NCCL_TRY("ncclGetUniqueId", r->GetUniqueId(&id));
  std::memcpy(id_out, &id, sizeof(id));
  return BLH_OK;
}

int blh_comm_destroy(blh_comm* c) {
  if (!c) return BLH_OK;
  int rc = BLH_OK;
  // (collectives of this communicator may still be in flight on its own stream, on a context's side stream — the tail
  //  of blh_train_step_dp — or on a caller's stream — blh_comm_all_reduce: wait for the device, destroying is rare)
  if (c->device >= 0) {
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur == c->device) (void)hipDeviceSynchronize();
    else if (c->cs) (void)hipStreamSynchronize(c->cs);
  }
  if (c->comm && rccl()) {
    const int r = rccl()->CommDestroy(c->comm);
    if (r != NCCL_SUCCESS) { set_error("ncclCommDestroy", r); rc = BLH_ERR_COMM; }
  }
  for (int i = 0; i < COMM_EVENTS; ++i) if (c->ev[i]) (void)hipEventDestroy(c->ev[i]);
  if (c->ev_cs) (void)hipEventDestroy(c->ev_cs);
  if (c->ev_tail) (void)hipEventDestroy(c->ev_tail);
  if (c->cs) (void)hipStreamDestroy(c->cs);
  delete c;
  return rc;
}

int blh_comm_create(blh_comm** out, const void* unique_id, int64_t id_bytes, int32_t world, int32_t rank) {
  if (!out) return BLH_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  if (!unique_id || id_bytes != (int64_t)sizeof(ncclUniqueId) || world < 1 || rank < 0 || rank >= world)
    return BLH_ERR_INVALID_ARGUMENT;
  const Rccl* r = rccl();
  if (!r) return BLH_ERR_COMM;
  blh_comm* c = new (std::nothrow) blh_comm();
  if (!c) return BLH_ERR_INVALID_ARGUMENT;
  auto fail_hip = [&](hipError_t e) { g_last_hip_error = (int)e; blh_comm_destroy(c); return BLH_ERR_HIP; };
  hipError_t e = hipGetDevice(&c->device);
  if (e != hipSuccess) return fail_hip(e);
  c->world = world; c->rank = rank;
  (void)r->GetVersion(&c->version);
  if ((e = hipStreamCreateWithFlags(&c->cs, hipStreamNonBlocking)) != hipSuccess) return fail_hip(e);
  for (int i = 0; i < COMM_EVENTS; ++i)
    if ((e = hipEventCreateWithFlags(&c->ev[i], hipEventDisableTiming)) != hipSuccess) return fail_hip(e);
  if ((e = hipEventCreateWithFlags(&c->ev_cs, hipEventDisableTiming)) != hipSuccess) return fail_hip(e);
  if ((e = hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming)) != hipSuccess) return fail_hip(e);
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof(id));
  const int rc = r->CommInitRank(&c->comm, world, id, rank);
  if (rc != NCCL_SUCCESS) { set_error("ncclCommInitRank", rc); c->comm = nullptr; blh_comm_destroy(c); return BLH_ERR_COMM; }
  *out = c;
  return BLH_OK;
}

int blh_comm_info(const blh_comm* c, int32_t* world, int32_t* rank, int32_t* rccl_version, int64_t* collectives) {
  if (!c) return BLH_ERR_INVALID_ARGUMENT;
  if (world) *world = c->world;
  if (rank) *rank = c->rank;
  if (rccl_version) *rccl_version = c->version;
  if (collectives) *collectives = c->collectives;
  return BLH_OK;
}

void* blh_comm_stream(blh_comm* c) { return c ? (void*)c->cs : nullptr; }

int blh_comm_set_bf16_buffer(blh_comm* c, uint16_t* buf, int64_t count) {
  if (!c || (buf && count <= 0)) return BLH_ERR_INVALID_ARGUMENT;
  c->half = buf; c->half_count = buf ? count : 0;
  return BLH_OK;
}

int blh_comm_all_reduce(blh_comm* c, void* stream, void* buf, int64_t count, int32_t dtype, int32_t average) {
  if (!c || !buf || count <= 0 || nccl_dtype(dtype) < 0) return BLH_ERR_INVALID_ARGUMENT;
  const Rccl* r = rccl();
  if (!r) return BLH_ERR_COMM;
  BLH_NCCL_TRY("ncclAllReduce", r->AllReduce(buf, buf, (size_t)count, nccl_dtype(dtype), average ? NCCL_AVG : NCCL_SUM,
                                             c->comm, (hipStream_t)stream));
  c->collectives += 1;
  return BLH_OK;
}

int blh_comm_broadcast(blh_comm* c, void* stream, void* buf, int64_t bytes, int32_t root) {
  if (!c || !buf || bytes <= 0 || root < 0 || root >= c->world) return BLH_ERR_INVALID_ARGUMENT;
  const Rccl* r = rccl();
  if (!r) return BLH_ERR_COMM;
  BLH_NCCL_TRY("ncclBroadcast", r->Broadcast(buf, buf, (size_t)bytes, NCCL_UINT8, root, c->comm, (hipStream_t)stream));
  c->collectives += 1;
  return BLH_OK;
}

int blh_train_step_dp(blh_context* ctx, blh_comm* comm, const blh_model_desc* d, void* stream, float* params,
                      float* grads, float* exp_avg, float* exp_avg_sq, float* bn_running, int64_t* bn_nbt,
                      const float* x, const float* target, const blh_dropout* drop, float momentum,
                      const blh_adam_hyper* hyper, blh_step_state* dev_state, void* workspace,
                      int64_t workspace_bytes, float* pred, float* loss_out, float* stats_out, int64_t batch,
                      int64_t global_batch, blh_sync_fn sync, void* sync_user, int32_t flags) {
  if (!ctx || !comm || !d || !params || !grads || !exp_avg || !exp_avg_sq || !loss_out || (!hyper == !dev_state))
    return BLH_ERR_INVALID_ARGUMENT;
  if (flags & ~(BLH_DP_TAIL_ON_COMM_STREAM | BLH_DP_BF16_BUCKETS)) return BLH_ERR_INVALID_ARGUMENT;
  const bool bf16_buckets = (flags & BLH_DP_BF16_BUCKETS) != 0;
  // (bf16 buckets: the caller's bf16 image must hold the arena; the capturable form has no bf16-reading optimiser)
  if (bf16_buckets && (!comm->half || comm->half_count < make_layout(d).total || dev_state)) return BLH_ERR_INVALID_ARGUMENT;
  if (global_batch != batch * (int64_t)comm->world) return BLH_ERR_INVALID_ARGUMENT;
  int dev = -1;
  BLH_HIP_TRY(hipGetDevice(&dev));
  if (dev != comm->device || dev != ctx->device) return BLH_ERR_INVALID_ARGUMENT;
  if (!rccl()) return BLH_ERR_COMM;
  hipStream_t s = (hipStream_t)stream;
  struct StepDevGuard {   // captured form: kernels of this call add dev_state->rng_step to the dropout step
    blh_context* c; bool on;
    StepDevGuard(blh_context* c_, const uint64_t* p) : c(c_), on(p != nullptr) { if (on) c->step_dev = p; }
    ~StepDevGuard() { if (on) c->step_dev = nullptr; }
  };
  if (dev_state) BLH_TRY(launch_step_state_advance(s, dev_state));
  StepDevGuard guard(ctx, dev_state ? &dev_state->rng_step : nullptr);
  // forward + MSE: the loss gradient and the loss partial sums stay in the workspace.  bf16 storage with
  // BLH_OPT_PERSISTENT_SHADOW: the bf16 parameter image the previous call's Adam kernel left is used as it is.
  const bool keep = d->gemm_dtype == 4 && ctx->persistent_shadow;
  const bool valid = keep && ctx->shadow_params == params && ctx->shadow_ws == workspace;
  const bool wdT_was = ctx->shadow_wdT;
  struct SyncGuard {
    blh_context* c; bool on;
    SyncGuard(blh_context* c_, blh_sync_fn f, void* u, int64_t g) : c(c_), on(f != nullptr) { if (on) c->sync = SyncCtx{f, u, g}; }
    ~SyncGuard() { if (on) c->sync = SyncCtx{nullptr, nullptr, 0}; }
  };
  PendingLoss pend{nullptr, 0, 0.0};
  {
    SyncGuard sg(ctx, sync, sync_user, global_batch);
    BLH_TRY(forward_train_loss_core(ctx, d, s, params, bn_running, bn_nbt, x, target, drop, momentum, workspace,
                                    workspace_bytes, pred, loss_out, batch, valid, &pend));
  }
  // backward: every bucket's all-reduce behind the kernel that completes it.  A range is complete on the side stream
  // of a two-stream context (include/bilinear_hip.h: blh_backward), else on `stream`.
  // (under stream capture every collective of the step goes through the communicator's ONE stream: RCCL launches of
  //  one communicator from two streams inside one capture end in a segmentation fault at hipStreamEndCapture —
  //  RCCL 2.26.6 / HIP 7.0, tools_dev/native_capture_debug.py)
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  BLH_HIP_TRY(hipStreamIsCapturing(s, &cap));
  const bool tail_on_producer = !(flags & BLH_DP_TAIL_ON_COMM_STREAM) && cap == hipStreamCaptureStatusNone;
  DpCall call{comm, ctx->two_stream ? ctx->s2 : s, grads, loss_out, tail_on_producer, bf16_buckets ? comm->half : nullptr,
              pend, loss_out};
  int rc = sync ? blh_backward_sync(ctx, d, s, params, x, drop, workspace, workspace_bytes, nullptr, grads, batch,
                                    bucket_thunk, &call, global_batch, sync, sync_user)
                : blh_backward(ctx, d, s, params, x, drop, workspace, workspace_bytes, nullptr, grads, batch,
                               bucket_thunk, &call);
  if (rc == BLH_OK) rc = call.status;
  if (rc == BLH_OK && !call.last_seen) rc = BLH_ERR_INVALID_ARGUMENT;   // (every backward closes with offset 0)
  // whatever happened, the caller's stream must end up behind everything this call enqueued elsewhere
  hipStream_t tail = call.tail_on_producer ? call.producer : comm->cs;
  if (rc == BLH_OK) {
    if (call.tail_on_producer && call.used_cs) {
      BLH_HIP_TRY(hipEventRecord(comm->ev_cs, comm->cs));
      BLH_HIP_TRY(hipStreamWaitEvent(tail, comm->ev_cs, 0));
    }
    // norm of the AVERAGED gradients (it cannot be taken before the exchange), clip, Adam — identical on every rank
    const int64_t count = make_layout(d).total;
    double* sumsq_part = d->gemm_dtype == 4 ? carve_h(d, batch, workspace).sumsq_part : carve(d, batch, workspace).sumsq_part;
    int np = 0;
    rc = bf16_buckets ? launch_sumsq_bf16(tail, comm->half, count, 1.0f, sumsq_part, &np)
                      : launch_sumsq(tail, grads, count, sumsq_part, &np);
    // bf16 storage under BLH_OPT_PERSISTENT_SHADOW: Adam also writes the bf16 image of the updated weights (and the
    // decode weight's K-major image when this shape takes the one-pass decode), as blh_train_step does
    const bool wdT = keep && !ctx->knob(KNOB_NO_DECODE_FUSE) && decode_fused_h_supported(batch, d->width, d->out_features);
    ShadowDst sd = NO_SHADOW;
    if (keep) {
      const WorkspaceH wh = carve_h(d, batch, workspace);
      sd = ShadowDst{wh.wsh, wdT ? wh.wdT : nullptr, make_layout(d).dec_w, d->width, d->out_features};
    }
    if (rc == BLH_OK) {
      if (bf16_buckets)      // norm, clip and Adam read the averaged bf16 buckets; the fp32 arena receives the clipped gradient
        rc = launch_clip_adam_bf16(tail, params, comm->half, 1.0f, grads, exp_avg, exp_avg_sq, count, *hyper, sumsq_part, np,
                                   stats_out, sd);
      else
        rc = dev_state ? launch_clip_adam_dev(tail, params, grads, exp_avg, exp_avg_sq, count, dev_state, sumsq_part, np,
                                              stats_out, LossFinish{nullptr, 0, 1.0, nullptr}, sd)
                       : launch_clip_adam(tail, params, grads, exp_avg, exp_avg_sq, count, *hyper, sumsq_part, np, stats_out,
                                          LossFinish{nullptr, 0, 1.0, nullptr}, sd);
    }
    if (keep && rc == BLH_OK) {
      // (the K-major image is complete only if an earlier launch zeroed its padding rows: a step's cast does)
      ctx->shadow_params = params; ctx->shadow_ws = workspace; ctx->shadow_wdT = wdT && (valid ? wdT_was : true);
    } else {
      ctx->shadow_params = nullptr; ctx->shadow_ws = nullptr; ctx->shadow_wdT = false;
    }
  }
  if (tail != s) {
    if (rc != BLH_OK && call.used_cs && tail != comm->cs) {   // (error path: the collective stream is still to be joined)
      BLH_HIP_TRY(hipEventRecord(comm->ev_cs, comm->cs));
      BLH_HIP_TRY(hipStreamWaitEvent(s, comm->ev_cs, 0));
    }
    BLH_HIP_TRY(hipEventRecord(comm->ev_tail, tail));
    BLH_HIP_TRY(hipStreamWaitEvent(s, comm->ev_tail, 0));
  }
  return rc;
}

}  // extern "C"
