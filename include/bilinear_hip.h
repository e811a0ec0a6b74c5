/*
 * bilinear_hip.h — C ABI of libbilinear_hip.so, the MI355X (gfx950) native
 * implementation of the one hot path of nulledge/bilinear: forward / backward /
 * optimiser step of the 2D->3D pose-lifting MLP.
 *
 * The reference has no FFI for this path: its boundary is the Python
 * torch.nn.Module / torch.optim.Optimizer surface consumed by
 *   /root/reference/train_bilinear.py:75-83   (zero_grad, forward, MSELoss,
 *                                              backward, clip_grad_norm_, step)
 *   /root/reference/valid_bilinear.py:31,52   (eval-mode forward)
 * and produced by
 *   /root/reference/model/bilinear.py:7-13    heavy_linear
 *   /root/reference/model/bilinear.py:16-55   BilinearUnit
 *   /root/reference/model/bilinear.py:58-92   load()
 * Each entry point below names the reference call-site it replaces.  The
 * reference-side binding (a ctypes stub inside model/bilinear.py) is shown in
 * INTEGRATION.md; bilinear_amd/_native.py is that stub.
 *
 * Conventions
 *   - plain pointers and sizes; no torch types.  Every pointer marked "device"
 *     is a HIP device pointer owned by the caller (PyTorch allocations in the
 *     shipped host code); the library never frees or retains caller memory
 *     beyond the call, and allocates nothing on the device.
 *   - `stream` is a hipStream_t passed as void*; every entry point only
 *     enqueues work on it (no host synchronisation, no allocation), so calls
 *     are asynchronous and hipGraph-capturable.
 *   - all tensors are row-major fp32 unless stated; a "[B,W]" tensor has the
 *     batch as the slow index, exactly like the reference's torch tensors.
 *   - return value: BLH_OK (0) or a negative blh_status; never aborts.
 *     blh_status_string() gives the text; HIP errors are reported as
 *     BLH_ERR_HIP and the hipError_t is available via blh_last_hip_error().
 */
#ifndef BILINEAR_HIP_H
#define BILINEAR_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 5 (round 6): blh_context_grid_barrier_timeouts, BLH_OPT_DEFER_SLABS and BLH_OPT_SMALL_STEP = 2 (the persistent
 * small-batch launches) were removed with the slower opt-in variants they served.
 * 6 (round 6): blh_comm_* / blh_rccl_* / blh_train_step_dp (collectives issued by the library) and BLH_ERR_COMM added;
 * nothing removed or changed.                                                                              */
#define BLH_ABI_VERSION 6

typedef enum {
  BLH_OK = 0,
  BLH_ERR_INVALID_ARGUMENT = -1, /* NULL pointer, non-positive size, bad enum   */
  BLH_ERR_SHAPE = -2,            /* shape the kernels do not support            */
  BLH_ERR_HIP = -3,              /* a HIP runtime call failed                   */
  BLH_ERR_WORKSPACE = -4,        /* workspace smaller than blh_workspace_bytes  */
  BLH_ERR_COMM = -5              /* librccl.so missing or an RCCL call failed: blh_comm_last_error() */
} blh_status;

const char* blh_status_string(int status);
int blh_last_hip_error(void);
int blh_abi_version(void);

/* ------------------------------------------------------------------------
 * Context.  Everything the library owns on a device is reached through a caller-owned
 * blh_context: the fork/join events of the two-stream backward, the per-call SyncBN /
 * captured-step plumbing, the option flags, and a reference to the side stream.  The side
 * stream is the ONE process-wide object of the library: one lowest-priority stream per device,
 * shared (reference-counted) by all contexts of that device.  HIP maps the streams of one
 * priority level onto a few hardware queues, and streams on one queue run in submission
 * order; a per-context, normal-priority side stream shared its queue with the caller's
 * stream or with RCCL's in any process that had created a communicator, which serialised
 * the backward.  Nothing else uses the lowest level.
 *   - blh_context_create binds the context to the CURRENT HIP device (hipGetDevice) and
 *     acquires that device's side stream; every network-level entry point below takes the
 *     context first and returns BLH_ERR_INVALID_ARGUMENT when another device is current.
 *   - one context = one in-flight call: a context must not be used from two host threads
 *     at once (use one context per thread / per model replica; contexts are cheap).
 *     Several models may take turns on one context: it remembers per workspace (the last four)
 *     which forward saved the activations there, so each blh_backward reads them with the kernels
 *     that wrote them; the loss gradient of blh_forward_train_loss is remembered for the LAST
 *     such forward only.
 *   - the contexts of one device share the side stream, so they take turns at it, and STREAM
 *     CAPTURE EXCLUDES THEM ALL: while one context captures a two-stream step
 *     (blh_train_step_captured / a captured blh_backward with BLH_OPT_TWO_STREAM = 1) the side
 *     stream is in capture mode, and work another context (or thread) enqueues on it then is
 *     recorded into — or invalidates — that capture.  Capture with no other context of the device
 *     in use (the Python mirror captures the single-stream order by default: its graph does not
 *     touch the side stream at all).
 *   - kernel attributes (dynamic LDS size) are set once per device, thread-safely.
 * The reference has no counterpart: it runs on PyTorch's per-process CUDA state
 * (/root/reference/util/config.py:17 picks the one device).                              */
typedef struct blh_context blh_context;
int blh_context_create(blh_context** out);
int blh_context_destroy(blh_context* ctx);
typedef enum {
  BLH_OPT_TWO_STREAM = 0, /* 1 (default): weight-gradient GEMMs on the context's side stream;
                             0: single-stream order.  Results are bit-identical either way.  */
  /* (1 was BLH_OPT_DEFER_SLABS until ABI 4: one deferred slab sum at the end of backward, measured slower) */
  BLH_OPT_LATE_FORK = 2,  /* when a stage's weight-gradient GEMM is handed to the side stream.
                             1: behind its data-gradient GEMM (it runs beside the next stage's
                             BatchNorm backward); 0: behind bn_bwd_apply, together with the
                             data-gradient GEMM; 2 (default): 0 where both GEMMs are launches of
                             at most one 128 KB-LDS workgroup per CU — the dispatcher then runs
                             them one after the other without a cross-queue latency — else 1.
                             Scheduling only: results are bit-identical.                     */
  BLH_OPT_SMALL_STEP = 4, /* gemm_dtype 0 (and 2 / 3, which approximate it), at most 384 rows, width <= 1024 — the
                             reference's own batch size is 64, /root/reference/util/config.py:15 — run on
                             purpose-built kernels (small_step.hip) in which a workgroup owns four columns of a
                             stage for all rows, so BatchNorm, dropout and the BatchNorm backward are local to it,
                             instead of ~50 launch-bound launches.  Applies to blh_train_step(_captured),
                             blh_forward_train, blh_backward and blh_forward_eval:
                               1 (default) one launch per stage (the fused step: 2 nh + 3 launches, 0.150 ms at
                                 2 x 1024 against 0.316; no residency requirement);   0 the multi-launch path.
                             Same arithmetic up to the order of fp32 sums (all within the fp32 parity tolerance).
                             The saved activations of such a forward are in the small-batch format: the context
                             pairs blh_backward with them.  (Until ABI 4 the value 2 selected persistent launches
                             with grid barriers: slower, and wrong when not fully resident — removed.)          */
  BLH_OPT_BUCKET_FLOATS = 6, /* default 0.  > 0: blh_backward / blh_backward_sync merge the contiguous gradient ranges
                             they would report (one per stage, decode first) and call the hook once per BUCKET of at
                             least this many elements, plus once for what is left at the end — the merging the
                             data-parallel driver used to do in its Python callback, one return to the interpreter
                             per stage.  A range is reported when its last part is ready, so
                             a bucket is complete on the side stream exactly when its last range is.            */
  BLH_OPT_DEV_KNOBS = 5,  /* developer A/B switches as one bit mask (csrc/step.h: blh::KNOB_*; DESIGN.md lists
                             them).  The initial value is read from the BLH_* environment variables ONCE, in
                             blh_context_create; no entry point reads the environment afterwards.  Measurement
                             knobs, not API: every combination computes the same step.                       */
  BLH_OPT_PERSISTENT_SHADOW = 3
                          /* gemm_dtype 4 only, default 0.  1: the Adam kernel of blh_train_step /
                             blh_train_step_captured also writes the bf16 image of the updated
                             parameters (fused Adam -> bf16 weight re-cast), and the NEXT
                             fused step on the same context, parameter arena and workspace skips
                             its own re-cast of the arena.  The caller promises that nothing else
                             writes `params` between two such steps; setting the option (to any
                             value) or calling any other forward entry point discards the image.
                             Results are bit-identical (the same rounding of the same values).  */
} blh_option;
int blh_context_set_option(blh_context* ctx, int32_t option, int32_t value);
int blh_context_get_option(const blh_context* ctx, int32_t option);
/* The side stream (hipStream_t as void*) the weight-gradient GEMMs run on; NULL when
 * BLH_OPT_TWO_STREAM is 0.  See blh_backward.                                            */
void* blh_context_side_stream(blh_context* ctx);
/* Replace the current device's process-wide side stream by a freshly created one; every context of
 * the device uses the new one from its next call on (the old one is drained and destroyed: call it
 * between steps, never under stream capture).  Why it exists: on this stack the two streams of the
 * backward (the caller's compute stream and the side stream) run at full speed only when their
 * hardware queues were created on the same side of the last RCCL communicator creation — with one
 * older and one younger every kernel of the step runs 1.5-5x longer (DESIGN.md section 4,
 * profiles/r04_dp_setup_order.md).  blh_tune_streams is the measured form of the same repair. */
int blh_side_stream_renew(void);
/* Counts the replacements of side streams in this process (any device): a caller that caches the result
 * of blh_tune_streams for a compute stream re-tunes when the number has changed.                     */
int32_t blh_side_stream_generation(void);
/* Measure whether kernels on `stream` (the compute stream the step will be enqueued on) and on the
 * current device's side stream really run beside each other, and repair the pair if they do not: a
 * short probe (12 short kernels on `stream` while 6 longer ones keep the side stream busy) is timed
 * for the current side stream and, if they take more than 2.5x their solo time (measured: 1.6-1.8x
 * for a good pair, 3.8x and more for a bad one), for up to `max_candidates` (0..7) freshly created
 * side streams, stopping at the first good one; the best becomes the device's side stream, the
 * others are destroyed.  Synchronises, allocates 32 MB for the duration of the call: call it once
 * per compute stream at set-up, never under stream capture.  report (optional, 4 floats): ms of the
 * short kernels alone, beside the side stream found, beside the one kept, number of candidates tried.
 * Tuning never fails a step: when the probe itself cannot run (no memory for its buffers, a stream that
 * cannot be created) the pair stays as it was, the call returns BLH_OK and report[3] is -1; a device's side
 * stream is replaced at most 4 times per process (two compute streams must not undo each other's choice
 * for ever).  The caller judges the result: report[2] > 2.5 * report[0] means the pair kept is still bad. */
int blh_tune_streams(void* stream, int32_t max_candidates, float* report);

/* ------------------------------------------------------------------------
 * Model description.  The reference hard-codes num_blocks=2, width=1024,
 * in_features=32, out_features=48 (model/bilinear.py:20-29); the build
 * generalises (num_blocks, width).  width must be a multiple of 32.
 * ---------------------------------------------------------------------- */
typedef struct {
  int32_t num_blocks;   /* residual blocks, each = 2 heavy_linear            */
  int32_t width;        /* hidden width W                                    */
  int32_t in_features;  /* 2*16 = 32                                         */
  int32_t out_features; /* 3*16 = 48                                         */
  int32_t gemm_dtype;   /* 0: exact fp32 MFMA (the reference's arithmetic);
                           1: (removed in ABI 4: round 1's mixed mode — fp32 tensors, operands
                              rounded to bf16 on load; superseded by 4) -> BLH_ERR_INVALID_ARGUMENT;
                           2: "bf16x3" — fp32 accuracy on the bf16 matrix cores: every operand
                              value is split exactly into three bf16 pieces on load and the
                              product is accumulated in fp32 from six bf16 MFMAs (the dropped
                              terms are below 2^-25 |a b|, under the rounding of an fp32
                              multiply); the 1024-wide Linears only, the skinny contractions
                              stay on the exact fp32 MFMA;
                           3: "fp16x2" — the same idea with TWO fp16 pieces and three f16 MFMAs
                              per product; every operand tensor is scaled by a power of two taken
                              from its largest magnitude (gathered by the kernels that produce
                              it), so fp16's exponent range is never left; contractions whose
                              operands carry no such maximum (encode, decode, stand-alone
                              stages) run as in mode 2;
                           4: "bf16s" — bf16 STORAGE (BASELINE configs 3-5): every [B,W] tensor
                              (pre-BN outputs, activations, their gradients), the network input and
                              a shadow of the parameters are bf16 in device memory (inside the
                              workspace); all contractions run on bf16 MFMA with fp32 accumulation,
                              operands fed by LDS-DMA without conversion; BatchNorm statistics
                              (taken from the fp32 accumulators), parameters, their gradients,
                              Adam and the loss stay fp32.  width % 128 == 0.  SyncBN (blh_sync_fn)
                              is supported in this mode as in mode 0.                          */
} blh_model_desc;

/* Number of heavy_linear stages = 1 + 2*num_blocks (encode + hidden). */
int32_t blh_num_heavy(const blh_model_desc* d);

/* ---- flat parameter arena ------------------------------------------------
 * Parameters live in ONE flat fp32 arena in `module.parameters()` order
 * (model/bilinear.py:22-29): per heavy_linear {Linear.weight [out,in],
 * Linear.bias [out], BN.weight [out], BN.bias [out]}, then decode.weight
 * [48,W], decode.bias [48].  Gradients, Adam exp_avg and exp_avg_sq use arenas
 * of the same layout.  Every tensor starts on a 64-float boundary; padding is
 * zero and stays zero.                                                      */
int64_t blh_param_arena_floats(const blh_model_desc* d);
int32_t blh_num_param_tensors(const blh_model_desc* d);
/* index in [0, blh_num_param_tensors): name is the reference state_dict key. */
int blh_param_tensor_info(const blh_model_desc* d, int32_t index, char* name, int32_t name_cap,
                          int64_t* offset_floats, int64_t* rows, int64_t* cols);

/* BatchNorm buffers: running stats fp32 [num_heavy][2][W] (mean, var) and
 * num_batches_tracked int64 [num_heavy].                                    */
int64_t blh_bn_running_floats(const blh_model_desc* d);

/* Workspace (activations saved for backward, gradient staging, split-K slabs,
 * reduction partials) for a given batch; 256-byte aligned base required.    */
int64_t blh_workspace_bytes(const blh_model_desc* d, int64_t batch);

/* ---- dropout source --------------------------------------------------------
 * Either an explicit keep-mask (parity tests replay the reference's masks):
 *   keep_mask = device uint8 [num_heavy][B][W], 1 = keep;
 * or keep_mask == NULL: counter-based Philox4x32-10 keyed by (seed, step,
 * layer, global row, column); the mask is regenerated in backward, never
 * stored.  row_offset is the global index of local row 0 (data parallel:
 * rank * per-rank batch), and must be a multiple of 32.  The mask depends only on
 * (seed, step, layer_base + stage, global row, column).                     */
typedef struct {
  const uint8_t* keep_mask; /* device, or NULL for Philox                    */
  uint64_t seed;
  uint64_t step;
  int64_t row_offset;
  int32_t layer_base;       /* added to the stage index to form the Philox stream id: 0 for a
                               BilinearUnit (stage i draws stream i); a process-unique id for a
                               stand-alone heavy_linear, so that stacked stages of equal shape
                               never share a mask                                            */
  int32_t reserved;
} blh_dropout;

/* Device-resident step state for hipGraph replay.  A captured launch freezes by-value
 * arguments, so blh_train_step_captured reads the Adam step count t, the learning rate and
 * the dropout step from this struct in device memory; blh_step_state_advance (a one-thread
 * kernel, first node of the captured step) increments `step` and `rng_step` and derives
 * step_size = lr/(1-beta1^t) and bc2_sqrt = sqrt(1-beta2^t).  The host rewrites `lr`
 * (the lr-decay hook, train_bilinear.py:66-70) with a plain async copy between replays.   */
typedef struct {
  double lr, beta1, beta2, eps, max_norm; /* as blh_adam_hyper (doubles: see there)      */
  uint64_t rng_step; /* dropout step used by the NEXT forward minus one   */
  int32_t step;      /* number of completed Adam updates                  */
  int32_t reserved;
  float step_size, bc2_sqrt;              /* derived by blh_step_state_advance            */
} blh_step_state;

/* ---- forward ---------------------------------------------------------------
 * Replaces BilinearUnit.forward (model/bilinear.py:31-41) in train mode
 * (train_bilinear.py:54,76): Linear -> BatchNorm1d(batch stats; running stats
 * updated with `momentum`, unbiased variance; num_batches_tracked += 1) ->
 * ReLU -> Dropout(0.5), residual adds, decode.  momentum < 0 selects
 * PyTorch's cumulative average (reset_statistics, model/bilinear.py:43-55).
 * Saves pre-BN outputs, activations and batch statistics in `workspace` for
 * blh_backward.  x: device [B,32]; pred: device [B,48] (written).           */
int blh_forward_train(blh_context* ctx, const blh_model_desc* d, void* stream, const float* params,
                      float* bn_running, int64_t* bn_num_batches_tracked, const float* x,
                      const blh_dropout* drop, float momentum, void* workspace,
                      int64_t workspace_bytes, float* pred, int64_t batch);

/* Eval-mode forward (valid_bilinear.py:31,52): BN uses running stats, dropout
 * is the identity, nothing is saved.                                        */
int blh_forward_eval(blh_context* ctx, const blh_model_desc* d, void* stream, const float* params,
                     const float* bn_running, const float* x, void* workspace,
                     int64_t workspace_bytes, float* pred, int64_t batch);

/* ---- loss -------------------------------------------------------------------
 * nn.MSELoss() (train_bilinear.py:49,78) and its gradient:
 * loss = sum((pred-target)^2)/(B*48) -> *loss_out (device scalar);
 * dpred = grad_scale * 2 (pred-target)/(B*48) (device [B,48]).
 * `loss_denominator` is B*48 of the GLOBAL batch under data parallelism.    */
int blh_mse_loss_grad(void* stream, const float* pred, const float* target, int64_t batch,
                      int64_t out_features, double loss_denominator, float grad_scale,
                      float* loss_out, float* dpred, void* workspace, int64_t workspace_bytes);

/* ---- backward ----------------------------------------------------------------
 * Replaces loss.backward() (train_bilinear.py:79) from d(loss)/d(pred) down:
 * writes (not accumulates) every parameter gradient into `grads` (arena
 * layout).  No input gradient (train_bilinear.py:72).  `on_ready`, if not
 * NULL, is called on the host right after the kernels that complete a
 * contiguous arena range [offset, offset+count) have been enqueued, in
 * decode -> encode order: the data-parallel host code launches the RCCL
 * all-reduce of that bucket from it, overlapping the rest of backward.
 * The weight-gradient GEMMs run on a side stream owned by the library (next to the
 * data-gradient GEMM and the BatchNorm-backward kernels of the following stage); a reported
 * range is complete on THAT stream: blh_context_side_stream(ctx) returns it (NULL when the
 * library runs single-stream, BLH_ONE_STREAM=1, and the range is complete on `stream`), and
 * the collective must be ordered behind it (e.g. make it the current stream while launching
 * the all-reduce).  blh_backward itself returns with `stream` waiting for the side stream. */
typedef void (*blh_grad_ready_fn)(void* user, int64_t offset_floats, int64_t count_floats);
int blh_backward(blh_context* ctx, const blh_model_desc* d, void* stream, const float* params, const float* x,
                 const blh_dropout* drop, void* workspace, int64_t workspace_bytes,
                 const float* dpred, float* grads, int64_t batch, blh_grad_ready_fn on_ready,
                 void* user);

/* Forward (train mode) + nn.MSELoss (train_bilinear.py:76,78) as one enqueue — what the fused
 * blh_train_step runs, for callers that put their own work between forward and backward (the
 * data-parallel step launches RCCL all-reduces from blh_backward's hook).  As blh_forward_train,
 * plus: loss_out (device scalar) = mean((pred - target)^2) over this call's rows, and the
 * gradient of that loss with respect to the prediction is left in `workspace`: a following
 * blh_backward / blh_backward_sync on the same context, workspace and batch with dpred == NULL
 * uses it (and takes the decode-bias partial sums the fused decode kernel produced).        */
int blh_forward_train_loss(blh_context* ctx, const blh_model_desc* d, void* stream, const float* params,
                           float* bn_running, int64_t* bn_num_batches_tracked, const float* x,
                           const float* target, const blh_dropout* drop, float momentum,
                           void* workspace, int64_t workspace_bytes, float* pred, float* loss_out,
                           int64_t batch);

/* ---- SyncBN variants (data parallel with statistics over the GLOBAL batch) -------------
 * Same as blh_forward_train / blh_backward, except that every BatchNorm's statistics are
 * exchanged across ranks: `sync` is called on the host, once per stage, right after the
 * kernels that produce this rank's partial sums have been enqueued; it must enqueue a SUM
 * all-reduce of `count` elements at `device_buf` (dtype 0 = fp32, 1 = fp64) on `stream`
 * (RCCL in the shipped host code).  Forward exchanges [sum z | sum z^2] (fp64, 2W), backward
 * [sum dY*zhat | sum dY] (fp32, 2W).  With equal per-rank batches the result equals the
 * reference's single-device run on the concatenated batch (SURVEY.md hazard H5).        */
typedef void (*blh_sync_fn)(void* user, void* device_buf, int64_t count, int32_t dtype);
int blh_forward_train_sync(blh_context* ctx, const blh_model_desc* d, void* stream, const float* params,
                           float* bn_running, int64_t* bn_num_batches_tracked, const float* x,
                           const blh_dropout* drop, float momentum, void* workspace,
                           int64_t workspace_bytes, float* pred, int64_t batch,
                           int64_t global_batch, blh_sync_fn sync, void* user);
int blh_forward_train_loss_sync(blh_context* ctx, const blh_model_desc* d, void* stream, const float* params,
                                float* bn_running, int64_t* bn_num_batches_tracked, const float* x,
                                const float* target, const blh_dropout* drop, float momentum,
                                void* workspace, int64_t workspace_bytes, float* pred, float* loss_out,
                                int64_t batch, int64_t global_batch, blh_sync_fn sync, void* user);
int blh_backward_sync(blh_context* ctx, const blh_model_desc* d, void* stream, const float* params, const float* x,
                      const blh_dropout* drop, void* workspace, int64_t workspace_bytes,
                      const float* dpred, float* grads, int64_t batch, blh_grad_ready_fn on_ready,
                      void* user, int64_t global_batch, blh_sync_fn sync, void* sync_user);

/* ---- clip + Adam ---------------------------------------------------------------
 * nn.utils.clip_grad_norm_(params, max_norm) (train_bilinear.py:81) fused with
 * torch.optim.Adam.step() (model/bilinear.py:60, train_bilinear.py:83):
 *   total = ||grads||_2 ; coef = min(1, max_norm/(total+1e-6)) ; g *= coef
 *   m += (1-b1)(g-m) ; v = b2 v + (1-b2) g^2
 *   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * `step` is t (1-based, after increment).  max_norm <= 0 disables clipping.
 * grads are rewritten with the clipped values (as the reference's .grad is).
 * stats_out (device float[2], optional): total_norm, clip coefficient.
 * The hyper-parameters are doubles, as the Python floats torch.optim.Adam computes with: it forms
 * 1-b1, 1-b2, lr/(1-b1^t) and sqrt(1-b2^t) in double and rounds each ONCE to the tensors' fp32
 * (with a float b2 = 0.999f, 1-b2 would be 0.99998713e-3 instead of 1e-3: 1.3e-5 off in exp_avg_sq). */
typedef struct {
  double lr, beta1, beta2, eps, max_norm;
  int32_t step;
  int32_t reserved;
} blh_adam_hyper;
/* nn.utils.clip_grad_norm_ alone (train_bilinear.py:81): scales `grads` in place;
 * stats_out (device float[2], optional) = total_norm, coefficient.           */
int blh_clip_grad_norm(void* stream, float* grads, int64_t count, float max_norm, void* workspace,
                       int64_t workspace_bytes, float* stats_out);
int blh_clip_adam_step(void* stream, float* params, float* grads, float* exp_avg,
                       float* exp_avg_sq, int64_t count, const blh_adam_hyper* hyper,
                       void* workspace, int64_t workspace_bytes, float* stats_out);
/* The same with the gradient given as bf16 (the compressed buckets of the data-parallel exchange,
 * averaged on the wire in bf16) times grad_scale: norm, clip and Adam read the bf16 values
 * directly (no cast back to the fp32 arena); `grads` (fp32 arena) receives the clipped gradient. */
int blh_clip_adam_step_bf16(void* stream, float* params, const uint16_t* grads_bf16, float grad_scale,
                            float* grads, float* exp_avg, float* exp_avg_sq, int64_t count,
                            const blh_adam_hyper* hyper, void* workspace, int64_t workspace_bytes,
                            float* stats_out);

/* ---- whole training step (single GPU) --------------------------------------------
 * The step body of train_bilinear.py:75-83 as one enqueue: forward_train, MSE,
 * backward, clip, Adam.  loss_out: device scalar.                            */
int blh_train_step(blh_context* ctx, const blh_model_desc* d, void* stream, float* params, float* grads,
                   float* exp_avg, float* exp_avg_sq, float* bn_running,
                   int64_t* bn_num_batches_tracked, const float* x, const float* target,
                   const blh_dropout* drop, float momentum, const blh_adam_hyper* hyper,
                   void* workspace, int64_t workspace_bytes, float* pred, float* loss_out,
                   float* stats_out, int64_t batch);

/* The same step with every per-step scalar read from `dev_state` (device): safe to capture
 * into a hipGraph and replay; drop->step is added to dev_state->rng_step.  Enqueues
 * blh_step_state_advance first.                                                        */
int blh_step_state_advance(void* stream, blh_step_state* dev_state);
int blh_train_step_captured(blh_context* ctx, const blh_model_desc* d, void* stream, float* params, float* grads,
                            float* exp_avg, float* exp_avg_sq, float* bn_running,
                            int64_t* bn_num_batches_tracked, const float* x, const float* target,
                            const blh_dropout* drop, float momentum, blh_step_state* dev_state,
                            void* workspace, int64_t workspace_bytes, float* pred,
                            float* loss_out, float* stats_out, int64_t batch);

/* gemm_dtype 4: re-cast the fp32 parameter arena into the workspace's bf16 image now.  Needed only
 * with BLH_OPT_PERSISTENT_SHADOW: before the first replay of a captured step, and after anything
 * but a fused step wrote `params`.                                                            */
int blh_refresh_param_shadow(blh_context* ctx, const blh_model_desc* d, void* stream, const float* params,
                             void* workspace, int64_t workspace_bytes, int64_t batch);

/* Pieces of the captured step for callers that interleave their own work (the data-parallel
 * step puts the RCCL all-reduces between backward and the optimiser):
 *   blh_context_set_step_state(ctx, dev_state): while dev_state is not NULL every forward /
 *     backward entry point called with this context adds dev_state->rng_step (device memory) to
 *     drop->step, exactly as blh_train_step_captured does — set it around a stream capture;
 *   blh_clip_adam_step_captured: blh_clip_adam_step with the hyper-parameters, the step count and
 *     the bias corrections read from dev_state (advance it first: blh_step_state_advance).     */
int blh_context_set_step_state(blh_context* ctx, const blh_step_state* dev_state);
int blh_clip_adam_step_captured(void* stream, float* params, float* grads, float* exp_avg,
                                float* exp_avg_sq, int64_t count, const blh_step_state* dev_state,
                                void* workspace, int64_t workspace_bytes, float* stats_out);

/* ---- data parallel with collectives issued by the library (round 6, opt-in) -----------------------------
 * New work: the reference is single-device (util/config.py:17); the step body is train_bilinear.py:75-83 on this
 * rank's rows.  The default data-parallel driver launches the bucket all-reduces from blh_backward's host hook
 * through torch.distributed; these entry points let the library call RCCL itself (librccl.so is resolved with
 * dlopen at the first call — the copy the process already holds, i.e. torch's inside a PyTorch process; the library
 * loads and every other entry point works without it).
 *   blh_rccl_version: ncclGetVersion of the library found (0: none).
 *   blh_rccl_unique_id: ncclGetUniqueId into id_out (id_bytes must be 128); rank 0 calls it and hands the bytes
 *     to the other ranks by any means (the shipped host code: a broadcast over the existing process group).
 *   blh_comm_create: ncclCommInitRank on the CURRENT device + a collective stream and events of its own.  Collective
 *     call: every rank of the communicator.  blh_comm_destroy waits for the device (called with the communicator's
 *     device current), then destroys.  A communicator serves ONE call at a time (like a blh_context), and every rank
 *     must issue the same calls on it in the same order (RCCL's rule).
 *   blh_comm_all_reduce / blh_comm_broadcast: one in-place collective enqueued on `stream` (dtype 0 fp32, 1 fp64,
 *     2 bf16; average != 0: ncclAvg, else ncclSum; broadcast counts bytes).
 *   blh_comm_last_error: text of the last BLH_ERR_COMM on this thread.                                             */
typedef struct blh_comm blh_comm;
int blh_rccl_version(void);
int blh_rccl_unique_id(void* id_out, int64_t id_bytes);
int blh_comm_create(blh_comm** out, const void* unique_id, int64_t id_bytes, int32_t world, int32_t rank);
int blh_comm_destroy(blh_comm* comm);
int blh_comm_info(const blh_comm* comm, int32_t* world, int32_t* rank, int32_t* rccl_version,
                  int64_t* collectives_issued);
void* blh_comm_stream(blh_comm* comm);
/* bf16 buckets (BLH_DP_BF16_BUCKETS): a caller-owned device buffer of at least blh_param_arena_floats() bf16 values, in
 * arena layout; NULL detaches it.                                                                                    */
int blh_comm_set_bf16_buffer(blh_comm* comm, uint16_t* buf, int64_t count);
int blh_comm_all_reduce(blh_comm* comm, void* stream, void* buf, int64_t count, int32_t dtype, int32_t average);
int blh_comm_broadcast(blh_comm* comm, void* stream, void* buf, int64_t bytes, int32_t root);
const char* blh_comm_last_error(void);
/* The data-parallel step as ONE enqueue: blh_forward_train_loss on this rank's `batch` rows, blh_backward with every
 * bucket (the ranges blh_backward reports, merged as BLH_OPT_BUCKET_FLOATS says) all-reduced (ncclAvg, fp32, in place
 * in `grads`) behind the kernel that completes it — on the communicator's own stream, so the exchange of stage l
 * overlaps the backward GEMMs of the stages below — then the norm of the AVERAGED gradients, clip and Adam
 * (blh_clip_adam_step) right behind the last bucket, and `stream` joined once at the end.  The LAST bucket (it holds
 * the encode stage) and the optimiser run on the stream that produced the last gradient (no queue hop in the tail);
 * BLH_DP_TAIL_ON_COMM_STREAM puts them on the communicator's stream instead (always the case while `stream` is
 * being captured).  loss_out: the GLOBAL batch's loss (mean
 * of the per-rank means; exchanged together with the last bucket).  global_batch must be batch * world.  Exactly one
 * of `hyper` / `dev_state` is given: dev_state selects the capturable form (as blh_train_step_captured; RCCL calls
 * are captured like kernels).  sync != NULL: SyncBN through the caller's callback (blh_forward_train_loss_sync).
 * Results equal blh_forward_train_loss + blh_backward + all-reduce(avg) per bucket + blh_clip_adam_step.           */
#define BLH_DP_TAIL_ON_COMM_STREAM 1
#define BLH_DP_BF16_BUCKETS 2 /* every bucket is rounded to bf16 (the library's cast kernel), averaged on the wire in bf16 —
                                 half the bytes on every xGMI link — and norm, clip and Adam read the bf16 values directly
                                 (blh_clip_adam_step_bf16); needs blh_comm_set_bf16_buffer and `hyper` (not dev_state)  */
int blh_train_step_dp(blh_context* ctx, blh_comm* comm, const blh_model_desc* d, void* stream, float* params,
                      float* grads, float* exp_avg, float* exp_avg_sq, float* bn_running,
                      int64_t* bn_num_batches_tracked, const float* x, const float* target,
                      const blh_dropout* drop, float momentum, const blh_adam_hyper* hyper,
                      blh_step_state* dev_state, void* workspace, int64_t workspace_bytes, float* pred,
                      float* loss_out, float* stats_out, int64_t batch, int64_t global_batch, blh_sync_fn sync,
                      void* sync_user, int32_t flags);

/* ---- one heavy_linear stage on its own --------------------------------------------------
 * model/bilinear.py:7-13 as a stand-alone module: a_out = Dropout(ReLU(BN(a_in W^T + b))).
 * in/out features must be multiples of 4.  `workspace` (blh_heavy_workspace_bytes) keeps the
 * pre-BN output and the batch statistics between forward and backward.  training = 0: running
 * statistics, no dropout, nothing saved.  backward writes dW, db, dgamma, dbeta and, if d_in is
 * not NULL, the input gradient [B, in].  drop->keep_mask, if given, is [B, out].            */
int64_t blh_heavy_workspace_bytes(int64_t batch, int32_t in_features, int32_t out_features);
int blh_heavy_forward(blh_context* ctx, void* stream, const float* a_in, const float* weight, const float* bias,
                      const float* gamma, const float* beta, float* running_mean,
                      float* running_var, int64_t* num_batches_tracked, const blh_dropout* drop,
                      float momentum, int32_t training, int32_t gemm_dtype, void* workspace,
                      int64_t workspace_bytes, float* a_out, int64_t batch, int32_t in_features,
                      int32_t out_features);
int blh_heavy_backward(blh_context* ctx, void* stream, const float* d_out, const float* a_in, const float* weight,
                       const float* gamma, const blh_dropout* drop, int32_t gemm_dtype,
                       void* workspace, int64_t workspace_bytes, float* d_weight, float* d_bias,
                       float* d_gamma, float* d_beta, float* d_in, int64_t batch,
                       int32_t in_features, int32_t out_features);

/* ---- validation metric ----------------------------------------------------------------
 * valid_bilinear.py:53-70: pred/target [B, joints*3] are de-normalised with the train-set
 * mean/stddev [joints*3]; dist_out[b] = sum over joints of the Euclidean distance (mm).
 * If action_ids (device int32 [B], values in [0,num_actions)) is given, action_sum (fp64)
 * and action_count (int64) are ACCUMULATED per action (zero them before the first batch):
 * MPJPE(action) = action_sum / (action_count * joints), as valid_bilinear.py:76-83.   */
int blh_mpjpe(void* stream, const float* pred, const float* target, const float* mean,
              const float* stddev, int64_t batch, int32_t joints, float* dist_out,
              const int32_t* action_ids, int32_t num_actions, double* action_sum,
              int64_t* action_count);

/* ---- kernel-level entry points (unit tests, profiling) ---------------------------
 * C[M,N] = op(A) * op(B) with fp32 MFMA.  a_kmajor=0: A is [M,K] (K
 * contiguous); 1: A is [K,M].  b_kmajor=0: B is [N,K] (K contiguous, i.e. a
 * Linear weight); 1: B is [K,N].  K and every contiguous dimension must be
 * multiples of 4.  splits>1 writes `splits` partial slabs [splits][M][N]
 * (reduction dimension cut in equal parts) that blh_sum_slabs adds up.      */
int blh_gemm_f32(void* stream, const float* A, int64_t lda, int32_t a_kmajor, const float* B,
                 int64_t ldb, int32_t b_kmajor, float* C, int64_t ldc, int64_t M, int64_t N,
                 int64_t K, int32_t splits, const float* bias, const float* addend,
                 int64_t ldadd);
/* Same contraction in gemm_dtype = 2 arithmetic (three-way bf16 split of both operands, six bf16
 * MFMAs per product, fp32 accumulate).  Shapes the 128x128 split kernel does not cover (N <= 64,
 * or M <= 64) run on the exact fp32 kernel.                                                    */
int blh_gemm_bf16x3(void* stream, const float* A, int64_t lda, int32_t a_kmajor, const float* B,
                    int64_t ldb, int32_t b_kmajor, float* C, int64_t ldc, int64_t M, int64_t N,
                    int64_t K, int32_t splits, const float* bias, const float* addend,
                    int64_t ldadd);
/* Same contraction in gemm_dtype = 3 arithmetic (two-piece fp16 split with per-operand power-of-two
 * scales, three f16 MFMAs per product, fp32 accumulate).  The entry point takes the maxima of the
 * two (dense: lda / ldb = the contiguous extent) operands itself, into `workspace`
 * (blh_gemm_fp16x2_workspace_bytes() bytes); inside the network they come for free from the
 * kernels that produce the operands (maxima_ready != 0: `workspace` already holds them from an
 * earlier call on the same operands).  Shapes outside the 128x128 split kernel run as in
 * blh_gemm_bf16x3.                                                                             */
int64_t blh_gemm_fp16x2_workspace_bytes(void);
int blh_gemm_fp16x2(void* stream, const float* A, int64_t lda, int32_t a_kmajor, const float* B,
                    int64_t ldb, int32_t b_kmajor, float* C, int64_t ldc, int64_t M, int64_t N,
                    int64_t K, int32_t splits, const float* bias, const float* addend,
                    int64_t ldadd, void* workspace, int32_t maxima_ready);
/* The GEMM of gemm_dtype = 4 ("bf16s": bf16 storage): A and B are bf16 IN MEMORY (uint16 bit
 * patterns; lda / ldb / ldc in elements; 16-byte aligned bases; K and every contiguous extent a
 * multiple of 8), bf16 MFMA with fp32 accumulation, C bf16 (out_bf16 != 0) or fp32.  Operand
 * layouts as blh_gemm_f32.  bias fp32 [N] or NULL; addend bf16 [M][ldadd] or NULL (not both; not
 * with splits > 1); stat_part (optional, with bias, splits == 1): per-row-tile column (mean,
 * M2) of the fp32 values before rounding, [ceil(M/T)][2][N] with T = blh_gemm_bf16s_tile(...)
 * rows per tile.  splits > 1 writes slabs [splits][M][ldc].
 * Three kernels serve it: 128 x 128 tiles (any shape); for launches with enough tiles to fill
 * the chip (N % 256 == 0, reduction slabs % 128 == 0), 256 x 256 tiles with the 8-phase
 * schedule; and 128 x 256 tiles with the same schedule on a three-deep LDS ring where only
 * half-height tiles fill the chip (M = 8192 at N = 1024).  blh_gemm_bf16s_tile /
 * blh_gemm_bf16s_tile_cols say which rows x columns tile a contraction of contiguous operands
 * takes (the environment variable BLH_BF16S_TILE = 128 | 256 | 384 (= 128 x 256) forces one
 * where the shape allows it, for A/B measurements).                                          */
int blh_gemm_bf16s(void* stream, const uint16_t* A, int64_t lda, int32_t a_kmajor, const uint16_t* B,
                   int64_t ldb, int32_t b_kmajor, void* C, int64_t ldc, int32_t out_bf16, int64_t M,
                   int64_t N, int64_t K, int32_t splits, const float* bias, const uint16_t* addend,
                   int64_t ldadd, float* stat_part);
int32_t blh_gemm_bf16s_tile(int64_t M, int64_t N, int64_t K, int32_t a_kmajor, int32_t b_kmajor,
                            int32_t out_bf16, int32_t splits);
int32_t blh_gemm_bf16s_tile_cols(int64_t M, int64_t N, int64_t K, int32_t a_kmajor, int32_t b_kmajor,
                                 int32_t out_bf16, int32_t splits);
/* How the bf16-storage backward plans the weight gradients dW = dZ^T A of its hidden width x width
 * stages at this batch (api_layout.h): *batched_slabs = batch slabs per stage when `stages` of them
 * run as ONE batched launch (0: they do not; each stage then runs alone with *stage_slabs slabs).
 * For benchmarks and tools that time these contractions the way the step launches them.        */
int blh_wgrad_plan_bf16s(int64_t width, int64_t batch, int32_t stages, int32_t* batched_slabs,
                         int32_t* stage_slabs);
/* Developer / test knob (process-wide): force the tile of every following bf16-storage GEMM where
 * its shape allows it: 128, 256, 384 (= 128 x 256); 0 = automatic; -1 = re-read BLH_BF16S_TILE.  */
int blh_gemm_bf16s_force_tile(int32_t tile);
/* `items` independent contractions of ONE shape in one launch of the 256 x 256 kernel: the weight
 * gradients dW_l = dZ_l^T A_(l-1) of several hidden stages of the lifter at once
 * (train_bilinear.py:79 through model/bilinear.py:25-26), which one stage alone cannot fill the
 * chip with.  Item i reads A + i * a_item_stride and B + i * b_item_stride (elements) and writes
 * fp32 slabs C + i * c_item_stride + s * M * ldc for s < splits (splits == 1: the result itself).
 * Shapes: M, N % 256 == 0, K == splits * k with k % 128 == 0; BLH_ERR_SHAPE otherwise, and when
 * items * splits * (M/256) * (N/256) < 224 workgroups (use blh_gemm_bf16s per item then).       */
int blh_gemm_bf16s_batched(void* stream, const uint16_t* A, int64_t lda, int32_t a_kmajor, int64_t a_item_stride,
                           const uint16_t* B, int64_t ldb, int32_t b_kmajor, int64_t b_item_stride, float* C,
                           int64_t ldc, int64_t c_item_stride, int64_t M, int64_t N, int64_t K, int32_t items,
                           int32_t splits);
/* fp32 <-> bf16 (round to nearest even) over `count` elements (multiple of 4).              */
int blh_cast_f32_to_bf16(void* stream, const float* src, uint16_t* dst, int64_t count);
int blh_cast_bf16_to_f32(void* stream, const uint16_t* src, float* dst, int64_t count);
/* The four skinny (HBM-bound) projections exactly as the training step launches them, for
 * profiling and unit tests (fp32).  model/bilinear.py:22 (encode, 32 -> W) and :29 (decode,
 * W -> 48) with train_bilinear.py:78 (MSELoss) fused into the decode forward.
 *   encode_fwd      Z[B,W] = x[B,in] W0^T + b0, BatchNorm partials per `*stat_tile_rows`-row tile
 *   decode_fwd_mse  pred = A Wd^T + bd ; dpred = 2 (pred - target)/(B out) ; *loss_out = MSE
 *                   (loss_out may be NULL: the partial sums stay in the workspace)
 *   decode_bwd      dWd[out,W] = dpred^T A ; dA[B,W] = dpred Wd
 *   encode_wgrad    dW0[W,in] = dZ^T x
 * `workspace`: blh_skinny_workspace_bytes (split-reduction slabs, partials).               */
int64_t blh_skinny_workspace_bytes(int64_t batch, int32_t width, int32_t in_features,
                                   int32_t out_features);
int blh_skinny_encode_fwd(void* stream, const float* x, const float* W0, const float* b0, float* Z,
                          float* stat_part, int32_t* stat_tile_rows, int64_t batch, int32_t width,
                          int32_t in_features);
int blh_skinny_decode_fwd_mse(void* stream, const float* A, const float* Wd, const float* bd,
                              const float* target, float* pred, float* dpred, float* loss_out,
                              void* workspace, int64_t workspace_bytes, int64_t batch, int32_t width,
                              int32_t out_features);
/* The encode stage without its pre-BatchNorm tensor (r05; /root/reference/model/bilinear.py:22,34): what the exact-fp32
 * step launches at more than 384 rows.  Forward: the batch statistics of z = x W0^T + b0 follow from the 33 x 32 sums
 * of x (mean_j = w_j . xbar + b_j, var_j = w_j^T Cov(x) w_j), so x -> A0 = 2 keep relu(BN(z)) is computed without
 * writing Z0; `keepbits` ([ceil(B/8)][W/4] words) receives keep AND [y > 0], `saved` ([4][W]) mean, invstd, scale,
 * shift; running statistics and the counter are updated (momentum < 0: cumulative average).  Backward: from dA0, the
 * bits and x alone — dW0 [W][32], db0 [W], dgamma, dbeta.  `scratch`: batch * width floats, kept between the two
 * calls (the forward leaves the sums of x there).  width % 256 == 0, in_features == 32; BLH_ERR_SHAPE otherwise.   */
int blh_skinny_encode_fused_fwd(void* stream, const float* x, const float* W0, const float* b0, const float* gamma,
                                const float* beta, float* running_mean, float* running_var, int64_t* nbt,
                                float momentum, float* saved, float* scratch, float* A, uint32_t* keepbits,
                                const blh_dropout* drop, int64_t batch, int32_t width, int32_t in_features);
int blh_skinny_encode_fused_bwd(void* stream, const float* dA, const float* x, const float* W0, const float* b0,
                                const float* saved, const uint32_t* keepbits, float* scratch, float* dW0, float* db0,
                                float* dgamma, float* dbeta, int64_t batch, int32_t width, int32_t in_features);
/* The same stage in bf16 storage (gemm_dtype 4; r06: bf16 MFMAs, x^T left by the forward for the backward): `x` [B][32],
 * `W0` [W][32], `A` / `dA` [B][W] are bf16 bit patterns; `keepbits` [ceil(B/4)][W/8] words (bn_bf16.hip's layout);
 * `scratch`: batch * width bf16 values (the stage's unused Z0 buffer), kept between the two calls.  width % 512 == 0.
 * What the bf16-storage step launches for stage 0 (/root/reference/model/bilinear.py:22,34).                      */
int blh_skinny_encode_fused_fwd_bf16(void* stream, const uint16_t* x, const uint16_t* W0, const float* b0,
                                     const float* gamma, const float* beta, float* running_mean, float* running_var,
                                     int64_t* nbt, float momentum, float* saved, uint16_t* scratch, uint16_t* A,
                                     uint32_t* keepbits, const blh_dropout* drop, int64_t batch, int32_t width,
                                     int32_t in_features);
int blh_skinny_encode_fused_bwd_bf16(void* stream, const uint16_t* dA, const uint16_t* x, const uint16_t* W0,
                                     const float* b0, const float* saved, const uint32_t* keepbits, uint16_t* scratch,
                                     float* dW0, float* db0, float* dgamma, float* dbeta, int64_t batch, int32_t width,
                                     int32_t in_features);
/* One-pass decode (r05): blh_skinny_decode_fwd_mse AND the data gradient dA[B,W] = dpred Wd from one read of A
 * (/root/reference/model/bilinear.py:29,39; train_bilinear.py:78-79: the loss is row-local).  What the fused
 * training step launches at out_features == 48, width 512 / 1024, batch <= 16384; BLH_ERR_SHAPE otherwise.      */
int blh_skinny_decode_fused(void* stream, const float* A, const float* Wd, const float* bd,
                            const float* target, float* pred, float* dpred, float* dA, float* loss_out,
                            void* workspace, int64_t workspace_bytes, int64_t batch, int32_t width,
                            int32_t out_features);
/* The same for bf16 storage (r05; what the bf16-storage step launches: out_features == 48, width % 256 == 0):
 * A and dA are bf16 [B][W] (raw uint16 bit patterns), Wd / bd / target / pred / dpred fp32; Wd is rounded to bf16
 * inside (the step reads its bf16 parameter image), dpred is also rounded to bf16 before dA = dpred Wd is formed
 * (fp32 accumulation), as the bf16 backward GEMM it replaces did.  `workspace`: at least
 * blh_skinny_decode_fused_bf16_workspace_bytes(batch, width, out_features) bytes.                                   */
int64_t blh_skinny_decode_fused_bf16_workspace_bytes(int64_t batch, int32_t width, int32_t out_features);
int blh_skinny_decode_fused_bf16(void* stream, const uint16_t* A, const float* Wd, const float* bd,
                                 const float* target, float* pred, float* dpred, uint16_t* dA, float* loss_out,
                                 void* workspace, int64_t workspace_bytes, int64_t batch, int32_t width,
                                 int32_t out_features);
int blh_skinny_decode_bwd(void* stream, const float* dpred, const float* A, const float* Wd,
                          float* dWd, float* dA, void* workspace, int64_t workspace_bytes,
                          int64_t batch, int32_t width, int32_t out_features);
int blh_skinny_encode_wgrad(void* stream, const float* dZ, const float* x, float* dW0, void* workspace,
                            int64_t workspace_bytes, int64_t batch, int32_t width, int32_t in_features);
int blh_sum_slabs(void* stream, const float* slabs, int64_t count, int32_t splits, float* out);
/* The forward kernel of one heavy_linear exactly as blh_forward_train launches it:
 * Z[M,N] = A[M,K] W[N,K]^T + bias, plus per-128-row-tile column statistics
 * stat_part[ceil(M/128)][2][N] = (tile mean, tile sum of squared deviations).  */
int blh_linear_fwd_stats(void* stream, const float* A, const float* W, const float* bias, float* Z,
                         float* stat_part, int64_t M, int64_t N, int64_t K);
/* Materialise the Philox keep-mask of one stage (uint8 [B,W], 1 = keep) exactly as the
 * forward/backward kernels regenerate it; drop->keep_mask must be NULL.         */
int blh_dropout_mask(void* stream, const blh_dropout* drop, int32_t layer, int64_t batch,
                     int32_t width, uint8_t* keep_out);

#ifdef __cplusplus
}
#endif
#endif /* BILINEAR_HIP_H */
