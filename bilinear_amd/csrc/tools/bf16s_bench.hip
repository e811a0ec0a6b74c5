// Developer tool (not part of the library): K-tile / ring-depth variants of the bf16-storage GEMM
// at the config-3 / config-5 shapes.  hipcc -O3 --offload-arch=gfx950 tools/bf16s_bench.hip -o bf16s_bench
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../gemm_bf16s_kernel.h"
#include "../gemm_bf16s_256.h"
#include "../gemm_bf16s_128x256.h"
#include "gemm_bf16s_p128x256.h"

using namespace blh;
thread_local int blh::g_last_hip_error = 0;
thread_local hipEvent_t blh::tl_stop_event = nullptr;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int LA, int LB, int EPI, bool OB, int BKE, int ST>
float run(const GemmParamsH& p, int splits, int reps) {
  constexpr size_t lds = gemm_bf16s_lds_bytes<BKE, ST>();
  auto kern = gemm_bf16s_kernel<LA, LB, EPI, OB, BKE, ST>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int tiles = (int)(ceil_div(p.M, 128) * ceil_div(p.N, 128));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(tiles, 1, splits), dim3(256), lds, 0, p);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(tiles, 1, splits), dim3(256), lds, 0, p);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

template <int LA, int LB, int EPI, bool OB>
float run256(const GemmParamsH& p, int splits, int reps) {
  auto kern = gemm_bf16s_256_kernel<LA, LB, EPI, OB>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)H256_LDS_BYTES));
  const int tiles = (int)(ceil_div(p.M, 256) * (p.N / 256));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(tiles, 1, splits), dim3(512), H256_LDS_BYTES, 0, p);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(tiles, 1, splits), dim3(512), H256_LDS_BYTES, 0, p);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

template <int LA, int LB, int EPI, bool OB, int ABL = 0, int SYNC = 0>
float run128x256(const GemmParamsH& p, int splits, int reps) {
  auto kern = gemm_bf16s_128x256_kernel<LA, LB, EPI, OB, ABL, SYNC>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)H128_LDS_BYTES));
  const int tiles = (int)(ceil_div(p.M, 128) * (p.N / 256));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(tiles, 1, splits), dim3(512), H128_LDS_BYTES, 0, p);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(tiles, 1, splits), dim3(512), H128_LDS_BYTES, 0, p);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

template <int LB, int EPI, int DBG = 0>
float runp128x256(const GemmParamsH& p, int reps, int grid_cap = 256) {
  auto kern = gemm_bf16s_p128x256_kernel<LB, EPI, DBG>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)HP128_LDS_BYTES));
  const int tiles = (int)((p.M / 128) * (p.N / 256));
  const int grid = std::min(tiles, grid_cap);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), HP128_LDS_BYTES, 0, p, tiles);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(grid), dim3(512), HP128_LDS_BYTES, 0, p, tiles);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 16384, W = argc > 2 ? atoi(argv[2]) : 1024;
  const int reps = argc > 3 ? atoi(argv[3]) : 100;
  uint16_t *A, *B, *C; float *bias, *stat, *slab;
  CK(hipMalloc(&A, (size_t)M * W * 2)); CK(hipMalloc(&B, (size_t)W * W * 2)); CK(hipMalloc(&C, (size_t)M * W * 2));
  CK(hipMalloc(&bias, W * 4)); CK(hipMalloc(&stat, (size_t)(M / 64 + 1) * 2 * W * 4)); CK(hipMalloc(&slab, (size_t)64 * W * W * 4));
  std::vector<uint16_t> h((size_t)M * W);
  for (auto& v : h) { float f = (float)rand() / RAND_MAX - 0.5f; uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
  CK(hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, h.data(), (size_t)W * W * 2, hipMemcpyHostToDevice));
  CK(hipMemset(bias, 0, W * 4));
  const double flop = 2.0 * M * W * (double)W;
  GemmParamsH f{};
  f.A = A; f.lda = W; f.B = B; f.ldb = W; f.C = C; f.ldc = W; f.M = M; f.N = W; f.K = W; f.k_per_split = W;
  f.bias = bias; f.stat_part = stat;
  GemmParamsH d = f;
  GemmParamsH w{};
  w.A = A; w.lda = W; w.B = A; w.ldb = W; w.C = slab; w.ldc = W; w.M = W; w.N = W; w.K = M;
  const int tiles = (W / 128) * (W / 128);
  const int splits = std::max(1, std::min(256 / tiles, M / 128));
  w.k_per_split = (int)round_up(ceil_div(M, splits), 128); w.c_split_stride = (int64_t)W * W;
#define ROW(BKE, ST)                                                                                    \
  {                                                                                                     \
    float t1 = run<ROWK, ROWK, EPI_BIAS_STATS, true, BKE, ST>(f, 1, reps);                              \
    float t2 = run<ROWK, KROW, EPI_STORE, true, BKE, ST>(d, 1, reps);                                   \
    float t3 = run<KROW, KROW, EPI_STORE, false, BKE, ST>(w, splits, reps);                             \
    printf("bk%-3d st%d (%3d KB LDS)  fwd %7.1f us %6.0f TF | dgrad %7.1f us %6.0f TF | wgrad(x%d) %7.1f us %6.0f TF\n", \
           BKE, ST, (int)(gemm_bf16s_lds_bytes<BKE, ST>() / 1024), t1 * 1e3, flop / t1 / 1e9, t2 * 1e3,     \
           flop / t2 / 1e9, splits, t3 * 1e3, flop / t3 / 1e9);                                        \
  }
  const int tiles256 = (W / 256) * (W / 256);
  const int splits256 = std::max(1, std::min(256 / tiles256, M / 128));
  GemmParamsH w2 = w;
  w2.k_per_split = (int)round_up(ceil_div(M, splits256), 128);
  const bool only256 = argc > 4;
  if (getenv("WGRAD_SWEEP")) {   // weight gradient only: slab counts x kernels
    for (int round = 0; round < 2; ++round)
      for (int sp : {2, 4, 8, 16}) {
        GemmParamsH ws = w;
        ws.k_per_split = (int)round_up(ceil_div(M, sp), 128);
        if ((int64_t)ws.k_per_split * sp != M) continue;
        float a = run<KROW, KROW, EPI_STORE, false, 64, 2>(ws, sp, reps);
        float b = run<KROW, KROW, EPI_STORE, false, 128, 2>(ws, sp, reps);
        float c = run256<KROW, KROW, EPI_STORE, false>(ws, sp, reps);
        printf("wgrad x%-2d  128x128 bk64 %6.1f us %5.0f TF | 128x128 bk128 %6.1f us %5.0f TF | 256x256 %6.1f us %5.0f TF\n",
               sp, a * 1e3, flop / a / 1e9, b * 1e3, flop / b / 1e9, c * 1e3, flop / c / 1e9);
      }
    return 0;
  }
  if (getenv("KSWEEP")) {   // forward at N = W, K = 128 .. W: fixed cost (intercept) and cost per K tile (slope)
    for (int K = 128; K <= W; K *= 2) {
      GemmParamsH fk = f;
      fk.K = K; fk.k_per_split = K;
      float a = run128x256<ROWK, ROWK, EPI_BIAS_STATS, true>(fk, 1, reps);
      float b = run256<ROWK, ROWK, EPI_BIAS_STATS, true>(fk, 1, reps);
      float c = run<ROWK, ROWK, EPI_BIAS_STATS, true, 64, 2>(fk, 1, reps);
      printf("K %5d (%3d K tiles)  128x256 %6.1f us | 256x256 %6.1f us | 128x128 %6.1f us\n", K, K / 64, a * 1e3, b * 1e3, c * 1e3);
    }
    return 0;
  }
  if (getenv("PERSIST")) {   // persistent 128 x 256 tiles (gemm_bf16s_p128x256.h) against the one-tile-per-workgroup kernels
    const size_t nstat = (size_t)(M / 64 + 1) * 2 * W;
    std::vector<uint16_t> c0((size_t)M * W), c1((size_t)M * W);
    std::vector<float> s0(nstat), s1(nstat);
    std::vector<float> hb(W);
    for (auto& v : hb) v = (float)rand() / RAND_MAX - 0.5f;
    CK(hipMemcpy(bias, hb.data(), W * 4, hipMemcpyHostToDevice));
    for (int round = 0; round < 3; ++round) {
      CK(hipMemset(C, 0, c0.size() * 2)); CK(hipMemset(stat, 0, nstat * 4));
      float a0 = run128x256<ROWK, ROWK, EPI_BIAS_STATS, true>(f, 1, reps);
      CK(hipMemcpy(c0.data(), C, c0.size() * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(s0.data(), stat, nstat * 4, hipMemcpyDeviceToHost));
      CK(hipMemset(C, 0, c0.size() * 2)); CK(hipMemset(stat, 0, nstat * 4));
      float a1 = runp128x256<ROWK, EPI_BIAS_STATS>(f, reps);
      CK(hipMemcpy(c1.data(), C, c1.size() * 2, hipMemcpyDeviceToHost));
      CK(hipMemcpy(s1.data(), stat, nstat * 4, hipMemcpyDeviceToHost));
      const bool same_c = !memcmp(c0.data(), c1.data(), c0.size() * 2);
      const bool same_s = !memcmp(s0.data(), s1.data(), (size_t)(M / 128) * 2 * W * 4);
      float a2 = run256<ROWK, ROWK, EPI_BIAS_STATS, true>(f, 1, reps);
      CK(hipMemset(C, 0, c0.size() * 2));
      float d0 = run128x256<ROWK, KROW, EPI_STORE, true>(d, 1, reps);
      CK(hipMemcpy(c0.data(), C, c0.size() * 2, hipMemcpyDeviceToHost));
      CK(hipMemset(C, 0, c0.size() * 2));
      float d1 = runp128x256<KROW, EPI_STORE>(d, reps);
      CK(hipMemcpy(c1.data(), C, c1.size() * 2, hipMemcpyDeviceToHost));
      const bool same_d = !memcmp(c0.data(), c1.data(), c0.size() * 2);
      float d2 = run256<ROWK, KROW, EPI_STORE, true>(d, 1, reps);
      if (round == 0) {
        float x1 = runp128x256<ROWK, EPI_BIAS_STATS, 1>(f, reps), x2 = runp128x256<ROWK, EPI_BIAS_STATS, 2>(f, reps);
        float x3 = runp128x256<ROWK, EPI_BIAS_STATS, 3>(f, reps);
        printf("persistent fwd, ablations: staged, not stored %6.1f us | no window traffic %6.1f us | stored, not staged %6.1f us\n",
               x1 * 1e3, x2 * 1e3, x3 * 1e3);
      }
      printf("fwd: 128x256 %6.1f us | persistent %6.1f us %5.0f TF (C %s, stats %s) | 256x256 %6.1f us || "
             "dgrad: 128x256 %6.1f | persistent %6.1f us %5.0f TF (C %s) | 256x256 %6.1f us\n",
             a0 * 1e3, a1 * 1e3, flop / a1 / 1e9, same_c ? "identical" : "DIFFER", same_s ? "identical" : "DIFFER",
             a2 * 1e3, d0 * 1e3, d1 * 1e3, flop / d1 / 1e9, same_d ? "identical" : "DIFFER", d2 * 1e3);
    }
    return 0;
  }
  if (getenv("SYNCAB")) {   // barrier structure of the 128 x 256 kernel: shipped (0) against one barrier per phase (1)
    std::vector<uint16_t> c0((size_t)M * W), c1((size_t)M * W);
    for (int round = 0; round < 3; ++round) {
      float a0 = run128x256<ROWK, ROWK, EPI_BIAS_STATS, true, 0, 0>(f, 1, reps);
      CK(hipMemcpy(c0.data(), C, c0.size() * 2, hipMemcpyDeviceToHost));
      float a1 = run128x256<ROWK, ROWK, EPI_BIAS_STATS, true, 0, 1>(f, 1, reps);
      CK(hipMemcpy(c1.data(), C, c1.size() * 2, hipMemcpyDeviceToHost));
      float d0 = run128x256<ROWK, KROW, EPI_STORE, true, 0, 0>(d, 1, reps);
      float d1 = run128x256<ROWK, KROW, EPI_STORE, true, 0, 1>(d, 1, reps);
      printf("128x256 fwd: two barriers + stagger %6.1f us | one barrier %6.1f us (outputs %s) || dgrad %6.1f | %6.1f us\n",
             a0 * 1e3, a1 * 1e3, memcmp(c0.data(), c1.data(), c0.size() * 2) ? "DIFFER" : "identical", d0 * 1e3, d1 * 1e3);
    }
    return 0;
  }
  if (getenv("STAMPS")) {   // where one launch of the 128 x 256 forward spends its time (wave 0 of every workgroup)
    const int wgs = (M / 128) * (W / 256);
    uint64_t* dbg; CK(hipMalloc(&dbg, (size_t)wgs * 64));
    GemmParamsH fs = f;
    fs.addend = reinterpret_cast<const bf16_bits*>(dbg);
    for (int r = 0; r < 3; ++r) {
      run128x256<ROWK, ROWK, EPI_BIAS_STATS, true, 4>(fs, 1, 20);
      CK(hipDeviceSynchronize());
      std::vector<uint64_t> hs((size_t)wgs * 8);
      CK(hipMemcpy(hs.data(), dbg, hs.size() * 8, hipMemcpyDeviceToHost));
      uint64_t t0 = ~0ull, t4 = 0;
      for (int w = 0; w < wgs; ++w) { t0 = std::min(t0, hs[w * 8]); t4 = std::max(t4, hs[w * 8 + 4]); }
      double seg[4] = {0, 0, 0, 0}, start = 0, end = 0;
      for (int w = 0; w < wgs; ++w) {
        for (int i = 0; i < 4; ++i) seg[i] += (double)(hs[w * 8 + i + 1] - hs[w * 8 + i]) * 0.01 / wgs;
        start += (double)(hs[w * 8] - t0) * 0.01 / wgs;
        end += (double)(t4 - hs[w * 8 + 4]) * 0.01 / wgs;
      }
      printf("first entry -> last drain %.2f us | mean per workgroup: entry skew %.2f, prologue %.2f, loop %.2f, epilogue (stores issued) %.2f, "
             "store drain %.2f, idle until the last one ends %.2f us\n", (double)(t4 - t0) * 0.01, start, seg[0], seg[1], seg[2], seg[3], end);
    }
    return 0;
  }
  if (getenv("ABLATE")) {   // the 128 x 256 kernel's loop with one ingredient removed (timing only)
    for (int round = 0; round < 2; ++round) {
      float a0 = run128x256<ROWK, ROWK, EPI_BIAS_STATS, true, 0>(f, 1, reps);
      float a1 = run128x256<ROWK, ROWK, EPI_BIAS_STATS, true, 1>(f, 1, reps);
      float a2 = run128x256<ROWK, ROWK, EPI_BIAS_STATS, true, 2>(f, 1, reps);
      float a3 = run128x256<ROWK, ROWK, EPI_BIAS_STATS, true, 3>(f, 1, reps);
      printf("128x256 fwd: full %6.1f us | no DMA %6.1f | no fragment reads %6.1f | no MFMAs %6.1f\n", a0 * 1e3, a1 * 1e3,
             a2 * 1e3, a3 * 1e3);
    }
    return 0;
  }
  for (int round = 0; round < 3; ++round) {
    if (!only256) {
      ROW(64, 2)
      if (round == 0) { ROW(64, 3) ROW(128, 2) }
    }
    float t1 = run256<ROWK, ROWK, EPI_BIAS_STATS, true>(f, 1, reps);
    float t2 = run256<ROWK, KROW, EPI_STORE, true>(d, 1, reps);
    float t3 = run256<KROW, KROW, EPI_STORE, false>(w2, splits256, reps);
    printf("256x256 8-phase (128 KB LDS)  fwd %7.1f us %6.0f TF | dgrad %7.1f us %6.0f TF | wgrad(x%d) %7.1f us %6.0f TF\n",
           t1 * 1e3, flop / t1 / 1e9, t2 * 1e3, flop / t2 / 1e9, splits256, t3 * 1e3, flop / t3 / 1e9);
    {
      // the 128 x 256 kernel (three-deep ring): weight gradient with twice the slabs of the 256 x 256 plan
      const int sp = std::min(8, splits256 * 2);
      GemmParamsH w3 = w;
      w3.k_per_split = (int)round_up(ceil_div(M, sp), 128);
      float u1 = run128x256<ROWK, ROWK, EPI_BIAS_STATS, true>(f, 1, reps);
      float u2 = run128x256<ROWK, KROW, EPI_STORE, true>(d, 1, reps);
      float u3 = ((int64_t)w3.k_per_split * sp == M) ? run128x256<KROW, KROW, EPI_STORE, false>(w3, sp, reps) : 0.f;
      printf("128x256 3-ring  (144 KB LDS)  fwd %7.1f us %6.0f TF | dgrad %7.1f us %6.0f TF | wgrad(x%d) %7.1f us %6.0f TF\n",
             u1 * 1e3, flop / u1 / 1e9, u2 * 1e3, flop / u2 / 1e9, sp, u3 * 1e3, u3 > 0 ? flop / u3 / 1e9 : 0.0);
    }
  }
  return 0;
}
