// Internal interface between the C ABI (api.hip) and the two step implementations: step_f32.hip (fp32
// storage: gemm_dtype 0, 2, 3) and step_bf16s.hip (bf16 storage: gemm_dtype 4).  Not installed.
#pragma once
#include <cstdlib>
#include <vector>

#include "common.h"
#include "gemm_bf16s_kernel.h"
#include "api_layout.h"

namespace blh {
// SyncBN plumbing of the current call (data parallel): statistics over `global_batch` rows,
// exchanged by the host callback
struct SyncCtx { blh_sync_fn fn; void* user; int64_t global_batch; };

// Developer A/B switches ("Developer switches"): read from the environment ONCE, when the context is
// created (blh_context_create), or set through blh_context_set_option(BLH_OPT_DEV_KNOBS, mask).  Nothing on the
// per-step host path calls getenv.
// Every knob switches a SHIPPED path back to the form it replaced (A/B measurements, and the materialised forms are what
// other shapes / SyncBN take anyway).  The slower opt-in variants of rounds 4-5 (bits 2, 4, 256, 512, 1024, 2048:
// BLH_K9_F32, BLH_FWD_FUSE, BLH_MID_FUSE, BLH_NO_DEC_ATTACH, BLH_MID_PAIR, BLH_MID_PAIR_APPLY) left the library in
// round 6 (ABI 5); their records stay under profiles/.  The bit values of the remaining knobs did not move.
enum : int32_t {
  KNOB_NO_SUMSQ_FOLD = 1,          // BLH_NO_SUMSQ_FOLD: fp32 fused step, separate gradient-norm pass
  KNOB_BF16_FORCE_TWO_STREAM = 8,  // BLH_BF16_FORCE_TWO_STREAM: bf16 batched plan on two streams (the r04 order)
  KNOB_NO_K9 = 16,                 // BLH_NO_K9: bf16, streaming reduce kernel instead of the epilogue
  KNOB_NO_SMALL_STEP = 32,         // BLH_NO_SMALL_STEP: the <= 384-row kernels off
  KNOB_NO_DECODE_FUSE = 64,        // BLH_NO_DECODE_FUSE: decode forward and backward as separate launches
  KNOB_NO_ENCODE_FUSE = 128,       // BLH_NO_ENCODE_FUSE: encode stage through the materialised Z0
  KNOB_ALL = 1 | 8 | 16 | 32 | 64 | 128
};
static inline int32_t dev_knobs_from_env() {
  int32_t m = 0;
  if (std::getenv("BLH_NO_SUMSQ_FOLD")) m |= KNOB_NO_SUMSQ_FOLD;
  if (std::getenv("BLH_BF16_FORCE_TWO_STREAM")) m |= KNOB_BF16_FORCE_TWO_STREAM;
  if (std::getenv("BLH_NO_K9")) m |= KNOB_NO_K9;
  if (std::getenv("BLH_NO_SMALL_STEP")) m |= KNOB_NO_SMALL_STEP;
  if (std::getenv("BLH_NO_DECODE_FUSE")) m |= KNOB_NO_DECODE_FUSE;
  if (std::getenv("BLH_NO_ENCODE_FUSE")) m |= KNOB_NO_ENCODE_FUSE;
  return m;
}
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
  int late_fork = 2;        // BLH_OPT_LATE_FORK: 0 early, 1 late, 2 auto
  int32_t knobs = 0;        // BLH_OPT_DEV_KNOBS: blh::KNOB_* mask, latched from the environment at creation
  int32_t bucket_floats = 0;   // BLH_OPT_BUCKET_FLOATS: > 0: blh_backward merges adjacent ready ranges into buckets of
                               // at least this many elements before it calls the hook
  bool knob(int32_t k) const { return (knobs & k) != 0; }
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
  int small_step = 1;                 // BLH_OPT_SMALL_STEP: 0 off, 1 one launch per stage
  // Format of the activations each recent train-mode forward saved, by workspace (the last SAVED_SLOTS workspaces,
  // least recently noted evicted first): SAVED_MULTI the multi-launch layout with Z0 materialised, SAVED_SMALL the
  // small-batch layout (small_step.hip), SAVED_ENC_FUSED the multi-launch layout WITHOUT Z0 (stage 0 keeps keep + gate
  // bits and the moments of x: encode_f32.hip), SAVED_NONE nothing a backward could use (an eval forward or a whole
  // fused step overwrote the buffers).  blh_backward asks saved_format(): the record of its workspace when there is
  // one — with the batch it was saved for, anything else is refused.  A workspace WITHOUT a record (a traced program
  // may hand a functionalised copy — same contents, another address; or more than SAVED_SLOTS forwards are
  // outstanding) gets the format a forward with this context's present options would save for that batch
  // (`predicted`), unless a live record of the same batch says otherwise — then the call is ambiguous and refused
  // (the old fallback to "the last forward" read mode-3 activations as mode 0 without an error).
  enum : int { SAVED_NONE = -1, SAVED_MULTI = 0, SAVED_SMALL = 1, SAVED_ENC_FUSED = 3 };
  static constexpr int SAVED_SLOTS = 16;
  struct SavedFormat { const void* ws; int64_t batch; int mode; uint64_t stamp; };
  SavedFormat saved[SAVED_SLOTS] = {};
  uint64_t saved_clock = 0;
  int fwd_mode = SAVED_NONE;      // what the forward of the CURRENT call saved (read by the fused step's backward)
  void note_saved(const void* ws, int64_t batch, int mode) {
    int slot = 0;
    for (int i = 0; i < SAVED_SLOTS; ++i) {
      if (saved[i].ws == ws) { slot = i; break; }
      if (saved[i].stamp < saved[slot].stamp) slot = i;
    }
    saved[slot] = {ws, batch, mode, ++saved_clock};
    fwd_mode = mode;
  }
  int saved_format(const void* ws, int64_t batch, int predicted) const {
    if (!ws) return SAVED_NONE;
    for (int i = 0; i < SAVED_SLOTS; ++i)
      if (saved[i].ws == ws && saved[i].stamp) return saved[i].batch == batch ? saved[i].mode : SAVED_NONE;
    for (int i = 0; i < SAVED_SLOTS; ++i)
      if (saved[i].stamp && saved[i].batch == batch && saved[i].mode != SAVED_NONE && saved[i].mode != predicted)
        return SAVED_NONE;
    return predicted;
  }
  const void* shadow_params = nullptr;
  const void* shadow_ws = nullptr;
  bool shadow_wdT = false;            // ... and the decode weight's K-major image (one-pass decode) with it
  // one-pass decode (skinny.hip: decode_fused_kernel): the forward with a target also left the decode data gradient
  // dA = dP Wd in this workspace's G0 for this batch; the backward that consumes that forward skips the GEMM
  const void* dec_da_ws = nullptr;
  bool dec_fork_attached = false;   // the one-pass decode launch carried ev_dz[nh] as its completion signal
  int64_t dec_da_batch = 0;
};

namespace blh {

static inline DropoutSrc layer_drop(const blh_context* ctx, const blh_dropout* drop, int layer,
                             int64_t batch, int W) {
  DropoutSrc d;
  d.step_dev = ctx->step_dev;
  d.keep = drop->keep_mask ? drop->keep_mask + (int64_t)layer * batch * W : nullptr;
  d.seed = drop->seed; d.step = drop->step; d.row_offset = drop->row_offset;
  d.layer = drop->layer_base + layer;
  return d;
}

// fused: the caller is the whole-step path: the decode-bias partials come from the fused decode kernel
// (dec_bias_S rows) and the sum-of-squares partials of the arena are returned for clip + Adam;
// sumsq_src[0]: where the partials really are when backward_impl returns (sumsq_part, or the producers' array)
struct FusedBackward { int dec_bias_S; double* sumsq_part; int* sumsq_nparts; double** sumsq_src; };

// ---- step_f32.hip
int forward_impl(blh_context* ctx, const blh_model_desc* d, hipStream_t s, const float* params,
                 float* bn_running, int64_t* nbt, const float* x, const blh_dropout* drop,
                 float momentum, const Workspace& ws, float* pred, int64_t batch,
                 bool train, const float* target, float mse_scale, float* loss_part,
                 int* loss_nparts);
int wgrad(int dtype, hipStream_t s, GemmTile tile, const float* dZ, int64_t ld_dz, int M,
          const float* act, int64_t ld_act, int N, int64_t batch, int64_t tiles,
          float* slabs, float* out, const float* amax_dz = nullptr,
          const float* amax_act = nullptr, int amax_parts = 0, double* sq = nullptr, int sq_blocks = 0);
// saved_mode: blh_context::SAVED_MULTI or SAVED_ENC_FUSED — what the forward that this backward consumes saved
int backward_impl(blh_context* ctx, const blh_model_desc* d, hipStream_t s, const float* params,
                  const float* x, const blh_dropout* drop, const Workspace& ws,
                  const float* dpred, float* grads, int64_t batch,
                  blh_grad_ready_fn on_ready, void* user, int saved_mode,
                  const FusedBackward* fused = nullptr);
// would a train-mode forward of this batch on this context take the encode stage without Z0?
bool enc_fused_ok(const blh_context* ctx, const blh_model_desc* d, int64_t batch);
bool enc_fused_ok_h(const blh_context* ctx, const blh_model_desc* d, int64_t batch);
// ---- api.hip (for comm.hip)
struct PendingLoss { const float* part; int n; double denom; };
int forward_train_loss_core(blh_context* ctx, const blh_model_desc* d, hipStream_t s, const float* params,
                            float* bn_running, int64_t* bn_nbt, const float* x, const float* target,
                            const blh_dropout* drop, float momentum, void* workspace, int64_t workspace_bytes,
                            float* pred, float* loss_out, int64_t batch, bool shadow_valid, PendingLoss* pending);
// ---- step_bf16s.hip
int forward_h(blh_context* ctx, const blh_model_desc* d, hipStream_t s, const float* params,
              float* bn_running, int64_t* nbt, const float* x, const blh_dropout* drop,
              float momentum, const WorkspaceH& ws, float* pred, int64_t batch, bool train,
              bool shadow_valid = false, const float* target = nullptr, float mse_scale = 0.f,
              int* loss_nparts = nullptr);
int backward_h(blh_context* ctx, const blh_model_desc* d, hipStream_t s, const float* params,
               const blh_dropout* drop, const WorkspaceH& ws, const float* dpred,
               float* grads, int64_t batch, blh_grad_ready_fn on_ready, void* user, int saved_mode,
               int dec_bias_S = 0, int* fold_nparts = nullptr);

}  // namespace blh
