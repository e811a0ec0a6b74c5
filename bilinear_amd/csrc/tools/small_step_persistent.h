// Developer tool, NOT part of libbilinear_hip.so (round 6): the small-batch step as ONE persistent launch with a grid
// barrier per stage (BLH_OPT_SMALL_STEP = 2 until ABI 4).  Measured slower than one launch per stage (0.200 against
// 0.150 ms per step at 2 x 1024, batch 64: profiles/r04_small_step.md) and its barrier could time out when the grid was
// not resident, so the library ships the staged form only.  Kept buildable for tools/small_step_bench.hip (phase
// timeline from s_memrealtime stamps).  Include AFTER ../small_step.hip (uses its stage helpers).
#pragma once
#include "grid_barrier.h"

namespace blh {

namespace { constexpr int SS_MAX_ROWS = 64; }
enum SmallStepPhase { SS_ALL = 0, SS_FWD = 1, SS_BWD = 2 };
int small_step_max_grid(int* num_cus_out);
int launch_small_step(hipStream_t s, const SmallStepParams& p, int phase = SS_ALL);

// PHASE SS_ALL: the whole step.  SS_FWD / SS_BWD: the two halves the drop-in surface calls separately
// (blh_forward_train, blh_backward: /root/reference/train_bilinear.py:76,79 with nn.MSELoss, clip_grad_norm_ and
// Adam.step as their own calls between / behind them); what backward needs of forward then crosses the two launches
// in the workspace (x-hat in Z, the gate in dZ, gamma * invstd in the saved-statistics rows) instead of in LDS.
// LDS: [0, 2 KB) scratch doubles for block sums | 64 float4 dz | 16 floats colsum | per stage: x-hat[256], gate[256], bn[8]
template <int PHASE>
__global__ __launch_bounds__(SS_THREADS) void small_step_kernel(const SmallStepParams p) {
  extern __shared__ __align__(16) unsigned char ss_smem[];
  double* sh_d = reinterpret_cast<double*>(ss_smem);                               // 256 doubles
  float4* sh_dz = reinterpret_cast<float4*>(ss_smem + 2048);                       // 64 float4
  float* sh_cs = reinterpret_cast<float*>(ss_smem + 2048 + 1024);                  // 16 floats
  float* sh_save = reinterpret_cast<float*>(ss_smem + 2048 + 1024 + 64);           // nh x (256 + 256 + 8) floats
  constexpr int SAVE_STRIDE = 256 + 256 + 8;

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = tid >> 6;
  const int g = blockIdx.x;
  const int W = p.W, B = p.batch, nh = p.nh, OF = p.out_f;
  const int ngroups = W >> 2;
  const bool own = g < ngroups;                                     // owns a column group of the hidden width
  const int cg = own ? (g & 7) * (ngroups >> 3) + (g >> 3) : 0;
  const int n0 = cg * 4;
  const bool own_dec = g < (OF >> 2);                               // owns 4 output columns of decode
  const int o0 = g * 4;
  const int row = tid >> 2, c = tid & 3;
  const bool valid = row < B;
  const int col = n0 + c;
  const float inv_b = 1.0f / (float)B;
  GridBarrier bar{p.bar, gridDim.x, 0u};
  bar.init();
  double sq = 0.0;                                                  // sum of squares of the gradients this thread wrote
  int stamp_i = 0;
  auto stamp = [&]() {
    if (p.stamps && tid == 0) p.stamps[(int64_t)g * 64 + (stamp_i < 64 ? stamp_i : 63)] = __builtin_amdgcn_s_memrealtime();
    ++stamp_i;
  };
  stamp();

  WBlock wb;
  if (PHASE != SS_BWD && own) ss_load_w_rows(wb, p.params + p.w_off[0] + (int64_t)n0 * p.in_f, p.in_f, lane);

  // ------------------------------------------------------------------ forward ----
  for (int i = 0; PHASE != SS_BWD && i < nh; ++i) {
    const int K = (i == 0) ? p.in_f : W;
    const float* in = (i == 0) ? p.x : p.A[i - 1];
    if (own) {
      // the stage's scalars first: their latency hides behind the GEMM
      const float bias = p.params[p.b_off[i] + col], gamma = p.params[p.g_off[i] + col], beta = p.params[p.be_off[i] + col];
      float* rm = p.bn_running + ((int64_t)i * 2 + 0) * W;
      float* rv = p.bn_running + ((int64_t)i * 2 + 1) * W;
      const float rm0 = rm[col], rv0 = rv[col];
      const int64_t nbt0 = p.nbt[i];
      const float skipv = (i >= 2 && (i & 1) == 0 && valid) ? p.A[i - 2][(int64_t)row * W + col] : 0.f;
      const bool kept = valid && ss_keep(p.drop, i, (int64_t)B * W, row, col, W);
      stamp();
      float z = ss_gemm<false>(wb, in, K, K, B, wave, lane) + bias;
      stamp();
      if (!valid) z = 0.f;
      const float mean = ss_colsum(z, sh_cs, wave, lane) * inv_b;
      const float dlt = valid ? z - mean : 0.f;
      const float m2 = ss_colsum(dlt * dlt, sh_cs, wave, lane);
      const float invstd = (float)(1.0 / sqrt((double)m2 / (double)B + (double)1e-5f));
      const float sc = gamma * invstd;
      const float sh = beta - mean * sc;
      if (row == 0) {            // running statistics (unbiased variance; momentum < 0: cumulative average)
        const double f = (p.momentum >= 0.f) ? (double)p.momentum : 1.0 / (double)(nbt0 + 1);
        const double unbiased = (double)m2 / (double)(B > 1 ? B - 1 : 1);
        rm[col] = (float)((1.0 - f) * (double)rm0 + f * (double)mean);
        rv[col] = (float)((1.0 - f) * (double)rv0 + f * unbiased);
      }
      const float y = fmaf(z, sc, sh);
      const bool on = kept && y > 0.f;
      const float a = (on ? y * 2.f : 0.f) + skipv;
      if (valid) ss_publish(&p.A[i][(int64_t)row * W + col], a);
      if (PHASE == SS_ALL) {
        float* sv = sh_save + i * SAVE_STRIDE;
        sv[tid] = dlt * invstd;                 // x-hat
        sv[256 + tid] = on ? 2.f : 0.f;         // d a / d y
        if (row == 0) { sv[512 + c] = sc; }
      } else {
        if (valid) {
          p.Z[i][(int64_t)row * W + col] = dlt * invstd;
          p.dZ[i][(int64_t)row * W + col] = on ? 2.f : 0.f;
        }
        if (row == 0) {
          float* st = p.bn_saved[i];
          st[col] = mean; st[W + col] = invstd; st[2 * W + col] = sc; st[3 * W + col] = sh;
        }
      }
    }
    stamp();
    // weights of the next stage (or of decode): nobody else writes them, so ask before waiting
    if (i + 1 < nh) {
      if (own) ss_load_w_rows(wb, p.params + p.w_off[i + 1] + (int64_t)n0 * W, W, lane);
    } else if (own_dec) {
      ss_load_w_rows(wb, p.params + p.dec_w + (int64_t)o0 * W, W, lane);
    }
    bar.arrive_published();
    bar.wait();
    stamp();
  }

  // ------------------------------------------------------------------ decode (+ MSE) ----
  float g_even = 0.f;      // gradient w.r.t. the output of the last even stage (the block skip source)
  if (PHASE == SS_FWD) {   // forward only: the prediction, and the BatchNorm counters once everybody has read them
    if (own_dec) {
      const int oc = o0 + c;
      const float pr = ss_gemm<false>(wb, p.A[nh - 1], W, W, B, wave, lane) + p.params[p.dec_b + oc];
      if (valid) p.pred[(int64_t)row * OF + oc] = pr;
    }
    if (g == 0) {
      if (tid == 0) for (int i = 0; i < nh; ++i) p.nbt[i] += 1;
      bar.finish();
    }
    stamp();
    return;
  }
  if (own_dec) {
    const int oc = o0 + c;
    float dp = 0.f;
    if (PHASE == SS_ALL) {
      const float pr = ss_gemm<false>(wb, p.A[nh - 1], W, W, B, wave, lane) + p.params[p.dec_b + oc];
      float diff = 0.f;
      if (valid) {
        p.pred[(int64_t)row * OF + oc] = pr;
        diff = pr - p.target[(int64_t)row * OF + oc];
        ss_publish(&p.dpred[(int64_t)row * OF + oc], diff * p.mse_scale);
      }
      (void)ss_colsum(diff * diff, sh_cs, wave, lane);
      // the four column sums -> one partial per workgroup
      if (tid == 0)
        ss_publish(&p.loss_part[g], (sh_cs[0] + sh_cs[4] + sh_cs[8] + sh_cs[12]) + (sh_cs[1] + sh_cs[5] + sh_cs[9] + sh_cs[13]) +
                                        (sh_cs[2] + sh_cs[6] + sh_cs[10] + sh_cs[14]) + (sh_cs[3] + sh_cs[7] + sh_cs[11] + sh_cs[15]));
      dp = diff * p.mse_scale;
    } else {               // backward only: d loss / d prediction comes from the caller
      dp = valid ? p.dpred[(int64_t)row * OF + oc] : 0.f;
    }
    const float db = ss_colsum(dp, sh_cs, wave, lane);
    if (row == 0) { ss_publish(&p.grads[p.dec_b + oc], db); sq += (double)db * db; }
    __syncthreads();
    reinterpret_cast<float*>(sh_dz)[tid] = dp;          // [row][c]
  }
  stamp();
  if (PHASE == SS_ALL) bar.arrive_published();          // (backward only: dpred is an input of the launch)
  if (own_dec) {
    __syncthreads();
    sq += ss_wgrad<true>(sh_dz, p.A[nh - 1], W, B, p.grads + p.dec_w + (int64_t)o0 * W);
  }
  // decode weights, COLUMNS form: [OF][W], reduction over the OF outputs
  if (own) ss_load_w_cols(wb, p.params + p.dec_w + n0, OF, W, lane);
  if (PHASE == SS_ALL) bar.wait();
  stamp();

  // ------------------------------------------------------------------ backward ----
  for (int i = nh - 1; i >= 0; --i) {
    if (own) {
      const bool top = (i == nh - 1);
      // (backward only: what forward left in the workspace, requested before the GEMM)
      float sx = 0.f, sg = 0.f, ss = 0.f;
      if (PHASE == SS_BWD) {
        if (valid) { sx = p.Z[i][(int64_t)row * W + col]; sg = p.dZ[i][(int64_t)row * W + col]; }
        ss = p.bn_saved[i][2 * W + col];
      }
      float ga = top ? ss_gemm<true>(wb, p.dpred, OF, OF, B, wave, lane)
                     : ss_gemm<true>(wb, p.dZ[i + 1], W, W, B, wave, lane);
      stamp();
      if ((i & 1) == 0) {
        if (!top) ga += g_even;
        g_even = ga;
      }
      float xhat, gate, sc;
      if (PHASE == SS_ALL) {
        const float* sv = sh_save + i * SAVE_STRIDE;
        xhat = sv[tid]; gate = sv[256 + tid]; sc = sv[512 + c];
      } else {
        xhat = sx; gate = sg; sc = ss;
      }
      const float dy = valid ? ga * gate : 0.f;
      const float s_b = ss_colsum(dy, sh_cs, wave, lane);             // d beta
      const float s_g = ss_colsum(dy * xhat, sh_cs, wave, lane);      // d gamma
      const float dz = valid ? sc * (dy - (s_b + xhat * s_g) * inv_b) : 0.f;
      const float dbias = ss_colsum(dz, sh_cs, wave, lane);
      if (row == 0) {
        ss_publish(&p.grads[p.g_off[i] + col], s_g);
        ss_publish(&p.grads[p.be_off[i] + col], s_b);
        ss_publish(&p.grads[p.b_off[i] + col], dbias);
        sq += (double)s_g * s_g + (double)s_b * s_b + (double)dbias * dbias;
      }
      if (i >= 1 && valid) ss_publish(&p.dZ[i][(int64_t)row * W + col], dz);
      __syncthreads();
      reinterpret_cast<float*>(sh_dz)[tid] = dz;
    }
    stamp();
    if (i >= 1) bar.arrive_published();
    if (own) {
      __syncthreads();
      const int K = (i == 0) ? p.in_f : W;
      sq += ss_wgrad<true>(sh_dz, (i == 0) ? p.x : p.A[i - 1], K, B, p.grads + p.w_off[i] + (int64_t)n0 * K);
      if (i >= 1) ss_load_w_cols(wb, p.params + p.w_off[i] + n0, W, W, lane);
    }
    stamp();
    if (i >= 1) bar.wait();
    stamp();
  }

  if (PHASE == SS_BWD) {           // the gradient arena is complete (clip_grad_norm_ and Adam.step are the caller's next calls)
    if (g == 0) bar.finish();
    stamp();
    return;
  }
  // ------------------------------------------------------------------ clip + Adam ----
  const double wg_sq = ss_block_sum(sq, sh_d);
  if (tid == 0) __hip_atomic_store(&p.sumsq_part[g], wg_sq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  bar.arrive_published();          // (gradients, loss and norm partials all went out write-through)
  bar.wait();
  stamp();
  double a = 0.0;
  for (int k = tid; k < (int)gridDim.x; k += SS_THREADS) a += p.sumsq_part[k];
  const double total_sq = ss_block_sum(a, sh_d);
  AdamConsts ac = p.adam;
  if (p.st) ac = AdamConsts{(float)(1.0 - p.st->beta1), (float)p.st->beta2, (float)(1.0 - p.st->beta2),
                            p.st->step_size, p.st->bc2_sqrt, (float)p.st->eps, (float)p.st->max_norm};
  const float total_norm = (float)sqrt(total_sq);
  float coef = 1.0f;
  if (ac.max_norm > 0.f) coef = fminf(ac.max_norm / (total_norm + 1e-6f), 1.0f);
  if (g == 0 && tid == 0) {
    if (p.stats_out) { p.stats_out[0] = total_norm; p.stats_out[1] = coef; }
    double l = 0.0;
    for (int k = 0; k < (OF >> 2); ++k) l += (double)p.loss_part[k];
    p.loss_out[0] = (float)(l / p.denom);
    for (int i = 0; i < nh; ++i) p.nbt[i] += 1;
  }
  if (g == 0) bar.finish();
  const int64_t n4 = p.count >> 2;
  for (int64_t i = (int64_t)g * SS_THREADS + tid; i < n4; i += (int64_t)gridDim.x * SS_THREADS) {
    float4 gv = ss_ld4(p.grads + i * 4), mv = ss_ld4(p.m + i * 4), vv = ss_ld4(p.v + i * 4), pv = ss_ld4(p.params + i * 4);
    float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x; float* pp = &pv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float gg = gp[k] * coef;
      gp[k] = gg;
      mp[k] = mp[k] + (gg - mp[k]) * ac.one_minus_b1;
      vp[k] = vp[k] * ac.b2 + (ac.one_minus_b2 * gg) * gg;
      const float denom = sqrtf(vp[k]) / ac.bc2_sqrt + ac.eps;
      pp[k] = pp[k] - ac.step_size * (mp[k] / denom);
    }
    *reinterpret_cast<float4*>(p.grads + i * 4) = gv;
    *reinterpret_cast<float4*>(p.m + i * 4) = mv;
    *reinterpret_cast<float4*>(p.v + i * 4) = vv;
    *reinterpret_cast<float4*>(p.params + i * 4) = pv;
  }
  stamp();
}

size_t small_step_lds_bytes(int nh) { return 2048 + 1024 + 64 + (size_t)nh * (256 + 256 + 8) * sizeof(float); }

int small_step_max_grid(int* num_cus_out) {
  // per device (one entry per ordinal; a process normally drives one GPU): the dynamic-LDS attribute of the three
  // persistent kernels is a per-device setting, and so is what fits
  constexpr int MAX_DEV = 64;
  static std::atomic<int> cache[MAX_DEV];          // 0 = not probed, -1 = does not fit, n = CUs
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MAX_DEV) return 0;
  int v = cache[dev].load(std::memory_order_acquire);
  if (v == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 0;
    const size_t lds = small_step_lds_bytes(SS_MAX_STAGES);
    int worst = 1 << 30;
    auto probe = [&](auto kern) {
      int per_cu = 0;
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)lds) != hipSuccess ||
          hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, SS_THREADS, lds) != hipSuccess)
        per_cu = 0;
      worst = per_cu < worst ? per_cu : worst;
    };
    probe(small_step_kernel<SS_ALL>);
    probe(small_step_kernel<SS_FWD>);
    probe(small_step_kernel<SS_BWD>);
    v = worst >= 1 ? n : -1;
    cache[dev].store(v, std::memory_order_release);      // (two threads racing here compute the same value)
  }
  if (num_cus_out) *num_cus_out = v > 0 ? v : 0;
  return v > 0 ? v : 0;
}

int launch_small_step(hipStream_t s, const SmallStepParams& p, int phase) {
  const int grid = small_step_max_grid(nullptr);
  if (grid <= 0 || p.W / 4 > grid || p.batch > SS_MAX_ROWS || p.nh > SS_MAX_STAGES || p.W > 256 * SS_PASSES ||
      p.in_f > 256 * SS_PASSES || p.out_f > 256 * SS_PASSES)
    return BLH_ERR_SHAPE;
  const size_t lds = small_step_lds_bytes(p.nh);
  if (phase == SS_ALL) hipLaunchKernelGGL(small_step_kernel<SS_ALL>, dim3((unsigned)grid), dim3(SS_THREADS), lds, s, p);
  else if (phase == SS_FWD) hipLaunchKernelGGL(small_step_kernel<SS_FWD>, dim3((unsigned)grid), dim3(SS_THREADS), lds, s, p);
  else if (phase == SS_BWD) hipLaunchKernelGGL(small_step_kernel<SS_BWD>, dim3((unsigned)grid), dim3(SS_THREADS), lds, s, p);
  else return BLH_ERR_INVALID_ARGUMENT;
  BLH_HIP_TRY(hipGetLastError());
  return BLH_OK;
}

}  // namespace blh
