// The store step of BatchNorm's forward statistics, shared by bn_fwd_finalize (elementwise.hip: per-tile partials)
// and enc_bn_finalize (encode_f32.hip: statistics derived from the 33 x 32 sums of x).
#pragma once
#include "common.h"

namespace blh {

static constexpr float BN_EPS = 1e-5f;

// batch mean / M2 of one column -> what BatchNorm saves and applies
struct BnColumn { float mu, invstd, sc, sh; };
__device__ __forceinline__ BnColumn bn_finalize_values(double mean, double m2, int64_t batch, float gamma, float beta) {
  const double var = m2 / (double)batch;
  BnColumn c;
  c.invstd = (float)(1.0 / sqrt(var + (double)BN_EPS));
  c.mu = (float)mean;
  c.sc = gamma * c.invstd;
  c.sh = beta - c.mu * c.sc;
  return c;
}
// running statistics (unbiased variance, PyTorch BatchNorm1d semantics; momentum < 0: cumulative average)
__device__ __forceinline__ void bn_running_update(double mean, double m2, int64_t batch, int col, float* running_mean,
                                                  float* running_var, const int64_t* nbt, float momentum) {
  const double f = (momentum >= 0.f) ? (double)momentum : 1.0 / (double)(nbt[0] + 1);
  const double unbiased = m2 / (double)(batch > 1 ? batch - 1 : 1);
  running_mean[col] = (float)((1.0 - f) * (double)running_mean[col] + f * mean);
  running_var[col] = (float)((1.0 - f) * (double)running_var[col] + f * unbiased);
}

// what bn_finalize_store reads, requested early (the finalize kernels are chains of dependent round trips: these four
// loads used to be a round trip of their own behind the merge)
struct BnColumnIn { float gamma, beta, running_mean, running_var; };
__device__ __forceinline__ BnColumnIn bn_finalize_prefetch(int col, const float* gamma, const float* beta,
                                                           const float* running_mean, const float* running_var) {
  return BnColumnIn{gamma[col], beta[col], running_mean[col], running_var[col]};
}
__device__ __forceinline__ void bn_finalize_store(double mean, double m2, int64_t batch, int col, const BnColumnIn& in,
                                                  float* running_mean, float* running_var, const int64_t* nbt,
                                                  float momentum, float* saved_mean, float* saved_invstd, float* scale,
                                                  float* shift) {
  const BnColumn c = bn_finalize_values(mean, m2, batch, in.gamma, in.beta);
  saved_mean[col] = c.mu;
  saved_invstd[col] = c.invstd;
  scale[col] = c.sc;
  shift[col] = c.sh;
  const double f = (momentum >= 0.f) ? (double)momentum : 1.0 / (double)(nbt[0] + 1);
  const double unbiased = m2 / (double)(batch > 1 ? batch - 1 : 1);
  running_mean[col] = (float)((1.0 - f) * (double)in.running_mean + f * mean);
  running_var[col] = (float)((1.0 - f) * (double)in.running_var + f * unbiased);
}

// batch mean / M2 of one column -> saved statistics, scale / shift, running statistics
__device__ __forceinline__ void bn_finalize_store(double mean, double m2, int64_t batch, int col,
                                                  const float* gamma, const float* beta,
                                                  float* running_mean, float* running_var,
                                                  const int64_t* nbt, float momentum, float* saved_mean,
                                                  float* saved_invstd, float* scale, float* shift) {
  const BnColumn c = bn_finalize_values(mean, m2, batch, gamma[col], beta[col]);
  saved_mean[col] = c.mu;
  saved_invstd[col] = c.invstd;
  scale[col] = c.sc;
  shift[col] = c.sh;
  bn_running_update(mean, m2, batch, col, running_mean, running_var, nbt, momentum);
}

}  // namespace blh
