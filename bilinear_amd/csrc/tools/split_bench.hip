// Developer tool: the bf16x3 split GEMM (gemm_split_kernel.h) — timing of the three Linear
// contractions and its error against an fp64 product, next to the exact-fp32 MFMA kernel's.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "../gemm_split_kernel.h"
#include "../gemm_f16x2_kernel.h"
using namespace blh;
thread_local int blh::g_last_hip_error = 0;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int BM, int BN, int WM, int WN, int LA, int LB, int EPI>
float run_split(const GemmParams& p, int splits, int reps) {
  constexpr int NT = 64 * WM * WN;
  constexpr size_t lds = gemm_split_lds_bytes<BM, BN>();
  auto kern = gemm_split_kernel<BM, BN, WM, WN, LA, LB, EPI>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int tiles = (int)(ceil_div(p.M, BM) * ceil_div(p.N, BN));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < (reps > 1 ? 100 : 0); ++i) hipLaunchKernelGGL(kern, dim3(tiles, 1, splits), dim3(NT), lds, 0, p);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(tiles, 1, splits), dim3(NT), lds, 0, p);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps * 1e3f;
}

template <int LA, int LB, int EPI>
float run_f16x2(const GemmParams& p, int splits, int reps) {
  constexpr size_t lds = gemm_f16x2_lds_bytes<128, 128>();
  auto kern = gemm_f16x2_kernel<LA, LB, EPI>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int tiles = (int)(ceil_div(p.M, 128) * ceil_div(p.N, 128));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < (reps > 1 ? 100 : 0); ++i) hipLaunchKernelGGL(kern, dim3(tiles, 1, splits), dim3(256), lds, 0, p);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(tiles, 1, splits), dim3(256), lds, 0, p);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps * 1e3f;
}

template <int LA, int LB, int EPI>
void run_f32(const GemmParams& p, int splits) {
  constexpr size_t lds = gemm_lds_bytes<128, 128, LA, LB, 3>();
  auto kern = gemm_f32_kernel<128, 128, 4, 2, LA, LB, EPI, 3>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int tiles = (int)(ceil_div(p.M, 128) * ceil_div(p.N, 128));
  hipLaunchKernelGGL(kern, dim3(tiles, 1, splits), dim3(512), lds, 0, p);
  CK(hipDeviceSynchronize());
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4096, W = argc > 2 ? atoi(argv[2]) : 1024;
  const int reps = argc > 3 ? atoi(argv[3]) : 300;
  float *A, *B, *C, *C2, *bias, *stat;
  const size_t act = (size_t)M * W;
  CK(hipMalloc(&A, act * 4)); CK(hipMalloc(&B, (size_t)W * W * 4));
  CK(hipMalloc(&C, std::max(act, (size_t)16 * W * W) * 4)); CK(hipMalloc(&C2, std::max(act, (size_t)16 * W * W) * 4));
  CK(hipMalloc(&bias, W * 4)); CK(hipMalloc(&stat, (size_t)(M / 32 + 1) * 2 * W * 4));
  std::vector<float> ha(act), hb((size_t)W * W), hbias(W, 0.f);
  srand(1);
  // activations ~ half-normal-ish positive/negative mix with a wide dynamic range, weights ~ N(0, 2/W)
  for (auto& v : ha) { double u = (double)rand() / RAND_MAX, s = (double)rand() / RAND_MAX; v = (float)((u - 0.3) * exp(4.0 * (s - 0.5))); }
  for (auto& v : hb) v = (float)(((double)rand() / RAND_MAX - 0.5) * 0.15);
  CK(hipMemcpy(A, ha.data(), act * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, hbias.data(), W * 4, hipMemcpyHostToDevice));
  // fp16x2 kernel: operand maxima (one partial each; in the library they come from the producers)
  float hmax[2] = {0.f, 0.f};
  for (float v : ha) hmax[0] = std::max(hmax[0], std::fabs(v));
  for (float v : hb) hmax[1] = std::max(hmax[1], std::fabs(v));
  float* dmax; CK(hipMalloc(&dmax, 8)); CK(hipMemcpy(dmax, hmax, 8, hipMemcpyHostToDevice));
  const double flop = 2.0 * M * W * (double)W;
  GemmParams f{}; f.A = A; f.lda = W; f.B = B; f.ldb = W; f.C = C; f.ldc = W; f.M = M; f.N = W; f.K = W; f.k_per_split = W;
  f.bias = bias; f.stat_part = stat;
  f.a_amax = dmax; f.a_namax = 1; f.b_amax = dmax + 1; f.b_namax = 1;
  GemmParams w{}; w.A = A; w.lda = W; w.B = A; w.ldb = W; w.C = C; w.ldc = W; w.M = W; w.N = W; w.K = M;
  w.a_amax = dmax; w.a_namax = 1; w.b_amax = dmax; w.b_namax = 1;
  const int splits = 4;
  w.k_per_split = (int)round_up(ceil_div(M, splits), 32); w.c_split_stride = (int64_t)W * W;

#define ROW(name, WM, WN)                                                                     \
  {                                                                                            \
    float t0 = run_split<128, 128, WM, WN, ROWK, ROWK, EPI_BIAS_STATS>(f, 1, 3 * reps);        \
    (void)t0; /* clock ramp: the first ~50 ms after idle run 10-15 % slow */                  \
    float t1 = run_split<128, 128, WM, WN, ROWK, ROWK, EPI_BIAS_STATS>(f, 1, reps);            \
    float t2 = run_split<128, 128, WM, WN, ROWK, KROW, EPI_STORE>(f, 1, reps);                 \
    float t3 = run_split<128, 128, WM, WN, KROW, KROW, EPI_STORE>(w, splits, reps);            \
    printf("split %-14s fwd %6.1f us %6.0f TF(fp32-equiv) | dgrad %6.1f us %6.0f TF | wgrad(x%d) %6.1f us %6.0f TF\n", \
           name, t1, flop / t1 / 1e6, t2, flop / t2 / 1e6, splits, t3, flop / t3 / 1e6);       \
  }
  ROW("128x128 w2x2", 2, 2)
  {
    float t1 = run_f16x2<ROWK, ROWK, EPI_BIAS_STATS>(f, 1, reps);
    float t2 = run_f16x2<ROWK, KROW, EPI_STORE>(f, 1, reps);
    float t3 = run_f16x2<KROW, KROW, EPI_STORE>(w, splits, reps);
    printf("fp16x2 (scaled)      fwd %6.1f us %6.0f TF(fp32-equiv) | dgrad %6.1f us %6.0f TF | wgrad(x%d) %6.1f us %6.0f TF\n",
           t1, flop / t1 / 1e6, t2, flop / t2 / 1e6, splits, t3, flop / t3 / 1e6);
  }

  {  // fixed part: the same launches with K = 64 (one loop round)
    GemmParams f1 = f; f1.K = 64; f1.k_per_split = 64;
    float a = run_split<128, 128, 2, 2, ROWK, ROWK, EPI_BIAS_STATS>(f1, 1, reps);
    float b = run_split<128, 128, 2, 2, ROWK, KROW, EPI_STORE>(f1, 1, reps);
    GemmParams f2 = f1; f2.M = 128;
    float c = run_split<128, 128, 2, 2, ROWK, KROW, EPI_STORE>(f2, 1, reps);
    printf("K=64: fwd(stats) %.1f us | dgrad(store) %.1f us | dgrad, 8 tiles only %.1f us\n", a, b, c);
  }
#ifdef BLH_SPLIT_STAMP
  {
    unsigned long long* st; CK(hipMalloc(&st, 4096 * 16)); CK(hipMemset(st, 0, 4096 * 16));
    GemmParams fs = f; fs.loss_part = reinterpret_cast<float*>(st); fs.C = C2;
    for (int i = 0; i < 50; ++i) run_split<128, 128, 2, 2, ROWK, KROW, EPI_STORE>(fs, 1, 1);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hs(2 * 256);
    CK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
    double cyc = 0, rt = 0; for (int i = 0; i < 256; ++i) { cyc += hs[2 * i]; rt += hs[2 * i + 1]; }
    cyc /= 256; rt /= 256;
    printf("K loop: %.0f shader cycles, %.2f us -> clock %.3f GHz; per MFMA slot %.1f cycles (ideal 32)\n", cyc, rt / 100.0, cyc / (rt * 10.0), cyc / (M ? (W / 16.0 * 24.0) : 1));
  }
#endif
  // ---- accuracy: sampled entries against an fp64 product ---------------------------------
  auto check = [&](const char* name, const GemmParams& gp, int la, int lb, int nsplit, bool split_kernel) {
    std::vector<float> hc((size_t)gp.M * gp.N * nsplit);
    CK(hipMemcpy(hc.data(), gp.C, hc.size() * 4, hipMemcpyDeviceToHost));
    const float* pa = (gp.A == A) ? ha.data() : hb.data();
    const float* pb = (gp.B == A) ? ha.data() : hb.data();
    double max_rel = 0, sum_rel2 = 0; int n = 0;
    for (int t = 0; t < 4000; ++t) {
      const int i = rand() % gp.M, j = rand() % gp.N;
      double ref = 0, mag = 0, got = 0;
      for (int k = 0; k < gp.K; ++k) {
        const double a = la == ROWK ? pa[(size_t)i * gp.lda + k] : pa[(size_t)k * gp.lda + i];
        const double b = lb == ROWK ? pb[(size_t)j * gp.ldb + k] : pb[(size_t)k * gp.ldb + j];
        ref += a * b; mag += fabs(a * b);
      }
      for (int s = 0; s < nsplit; ++s) got += hc[(size_t)s * gp.M * gp.N + (size_t)i * gp.ldc + j];
      const double rel = fabs(got - ref) / mag;    // error relative to sum |a b| (the fp32 bound's scale)
      max_rel = std::max(max_rel, rel); sum_rel2 += rel * rel; ++n;
    }
    printf("  %-6s %-22s max |err| / sum|ab| = %.3e   rms = %.3e\n", name, split_kernel ? "bf16x3 split" : "fp32 MFMA (exact)", max_rel, sqrt(sum_rel2 / n));
  };
  GemmParams g = f; g.C = C2;
  run_f16x2<ROWK, ROWK, EPI_BIAS_STATS>(g, 1, 1); CK(hipDeviceSynchronize());
  check("fwd", g, ROWK, ROWK, 1, true);
  printf("      ^ fp16x2 (scaled) \n");
  run_f16x2<ROWK, KROW, EPI_STORE>(g, 1, 1); CK(hipDeviceSynchronize());
  check("dgrad", g, ROWK, KROW, 1, true);
  printf("      ^ fp16x2 (scaled) \n");
  { GemmParams gw2 = w; gw2.C = C2; run_f16x2<KROW, KROW, EPI_STORE>(gw2, splits, 1); CK(hipDeviceSynchronize());
    check("wgrad", gw2, KROW, KROW, splits, true); printf("      ^ fp16x2 (scaled) \n"); }
  run_split<128, 128, 2, 2, ROWK, ROWK, EPI_BIAS_STATS>(g, 1, 1); CK(hipDeviceSynchronize());
  check("fwd", g, ROWK, ROWK, 1, true);
  run_f32<ROWK, ROWK, EPI_BIAS_STATS>(g, 1);
  check("fwd", g, ROWK, ROWK, 1, false);
  run_split<128, 128, 2, 2, ROWK, KROW, EPI_STORE>(g, 1, 1); CK(hipDeviceSynchronize());
  check("dgrad", g, ROWK, KROW, 1, true);
  run_f32<ROWK, KROW, EPI_STORE>(g, 1);
  check("dgrad", g, ROWK, KROW, 1, false);
  GemmParams gw = w; gw.C = C2;
  run_split<128, 128, 2, 2, KROW, KROW, EPI_STORE>(gw, splits, 1); CK(hipDeviceSynchronize());
  check("wgrad", gw, KROW, KROW, splits, true);
  run_f32<KROW, KROW, EPI_STORE>(gw, splits);
  check("wgrad", gw, KROW, KROW, splits, false);
  return 0;
}
