// The small-batch kernels (small_step.hip): one launch per stage, up to 384 rows.
#pragma once
#include "common.h"

namespace blh {

constexpr int SS_MAX_STAGES = 33;        // 1 + 2 * num_blocks (api_layout.h: check_desc)

struct SmallStepParams {
  int nh, W, in_f, out_f, batch;
  int64_t w_off[SS_MAX_STAGES], b_off[SS_MAX_STAGES], g_off[SS_MAX_STAGES], be_off[SS_MAX_STAGES];
  int64_t dec_w, dec_b, count;
  float* params; float* grads; float* m; float* v;
  float* bn_running; int64_t* nbt;
  const float* x; const float* target;
  float* A[SS_MAX_STAGES];               // [batch][W] per stage (workspace)
  float* dZ[SS_MAX_STAGES];
  float* Z[SS_MAX_STAGES];               // x-hat crosses from the forward to the backward launches here (the gate in dZ)
  float* bn_saved[SS_MAX_STAGES];        // [4][W] mean, invstd, scale, shift
  float* gskip[2];                       // staged launches: gradient w.r.t. the output of an even stage, for stage i - 2
  float* dpred;                          // [batch][out_f]
  float* pred; float* loss_out; float* stats_out;
  float* loss_part;                      // [out_f / 4]
  double* sumsq_part;                    // [grid]
  uint32_t* bar;                         // (tools/small_step_persistent.h only: grid barrier words)
  DropoutSrc drop;                       // of stage 0 (keep: base of the [nh][batch][W] masks)
  float momentum, mse_scale;
  double denom;
  AdamConsts adam;
  const blh_step_state* st;              // captured step: Adam scalars from device memory instead of `adam`
  unsigned long long* stamps;            // developer tool only (tools/small_step_bench.hip): [grid][64] s_memrealtime
};

// one launch per stage (no grid barrier, no residency requirement): forward stages + decode (mse: + MSE, dpred,
// decode gradients, loss / norm partials), backward stages (dec_here: decode gradients from the caller's dpred);
// the gradient-norm partials are sumsq_part[0 .. W / 4 + out_f / 4)
int launch_small_forward_staged(hipStream_t s, const SmallStepParams& p, bool mse);
// eval-mode forward (running statistics, no dropout, nothing saved; p.nbt must be null): nh + 1 launches
int launch_small_eval_staged(hipStream_t s, const SmallStepParams& p);
// wgrad_here = false: the hidden stages' weight gradients are left to the caller (one batched GEMM launch)
int launch_small_backward_staged(hipStream_t s, const SmallStepParams& p, bool dec_here, bool wgrad_here);

}  // namespace blh
