// Developer tool (not part of the library): time tile/wave variants of the fp32 MFMA GEMM
// at the hidden-layer shape.   hipcc -O3 --offload-arch=gfx950 tools/gemm_bench.hip -o gemm_bench
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "gemm_f32_kernel.h"
#include "../gemm_f32_ring.h"

using namespace blh;
thread_local int blh::g_last_hip_error = 0;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int BM, int BN, int WM, int WN, int LA, int LB, int EPI, int PIPE>
float run(const GemmParams& p, int splits, int reps) {
  constexpr int NT = 64 * WM * WN;
  constexpr size_t lds = gemm_lds_bytes<BM, BN, LA, LB, PIPE>();
  auto kern = gemm_f32_kernel<BM, BN, WM, WN, LA, LB, EPI, PIPE>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int tiles = (int)(ceil_div(p.M, BM) * ceil_div(p.N, BN));
  dim3 grid(tiles, 1, splits);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(NT), lds, 0, p);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(NT), lds, 0, p);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

template <int BM, int BN, int WM, int WN, int LA, int LB, int EPI, int BKT, int STAGES>
float run_ring(const GemmParams& p, int splits, int reps) {
  constexpr int NT = 64 * WM * WN;
  constexpr size_t lds = gemm_ring_lds_bytes<BM, BN, BKT, STAGES>();
  auto kern = gemm_f32_ring_kernel<BM, BN, WM, WN, LA, LB, EPI, BKT, STAGES>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int tiles = (int)(ceil_div(p.M, BM) * ceil_div(p.N, BN));
  dim3 grid(tiles, 1, splits);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kern, grid, dim3(NT), lds, 0, p);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(NT), lds, 0, p);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

// dgrad on one stream and wgrad on another, as the two-stream backward launches them: time per PAIR
template <int BM, int BN, int WM, int WN, int BKT, int STAGES>
float run_pair(const GemmParams& d, const GemmParams& w, int splits, int reps) {
  constexpr int NT = 64 * WM * WN;
  constexpr size_t lds = gemm_ring_lds_bytes<BM, BN, BKT, STAGES>();
  auto kd = gemm_f32_ring_kernel<BM, BN, WM, WN, ROWK, KROW, EPI_STORE, BKT, STAGES>;
  auto kw = gemm_f32_ring_kernel<BM, BN, WM, WN, KROW, KROW, EPI_STORE, BKT, STAGES>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kd), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kw), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  static hipStream_t sa = nullptr, sb = nullptr;
  if (!sa) { CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking)); }
  const int td = (int)(ceil_div(d.M, BM) * ceil_div(d.N, BN)), tw = (int)(ceil_div(w.M, BM) * ceil_div(w.N, BN));
  hipEvent_t e0, e1, eb;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&eb));
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, sa));
  CK(hipStreamWaitEvent(sb, e0, 0));
  for (int i = 0; i < reps; ++i) {
    hipLaunchKernelGGL(kd, dim3(td), dim3(NT), lds, sa, d);
    hipLaunchKernelGGL(kw, dim3(tw, 1, splits), dim3(NT), lds, sb, w);
  }
  CK(hipEventRecord(eb, sb));
  CK(hipStreamWaitEvent(sa, eb, 0));
  CK(hipEventRecord(e1, sa));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

// spot-check 64 output elements against a host fp64 dot product
static double g_maxerr;
void verify(const char* what, const std::vector<float>& hA, const std::vector<float>& hB, const std::vector<float>& hbias,
            const float* dC, int M, int N, int K, int W, int kind, int splits) {
  std::vector<float> c((size_t)M * N * splits);
  CK(hipMemcpy(c.data(), dC, c.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0;
  for (int t = 0; t < 64; ++t) {
    int i = (t * 7919 + 13) % M, j = (t * 104729 + 7) % N;
    double ref = 0;
    for (int k = 0; k < K; ++k) {
      double a = (kind == 2) ? hA[(size_t)k * W + i] : hA[(size_t)i * W + k];
      double b = (kind == 0) ? hB[(size_t)j * W + k] : (kind == 1 ? hB[(size_t)k * W + j] : hA[(size_t)k * W + j]);
      ref += a * b;
    }
    if (kind == 0) ref += hbias[j];
    double got = 0;
    for (int s = 0; s < splits; ++s) got += c[(size_t)s * M * N + (size_t)i * N + j];
    maxerr = std::max(maxerr, std::abs(got - ref));
  }
  if (maxerr > 1e-3) printf("   !! %s MISMATCH max err %g\n", what, maxerr);
  g_maxerr = std::max(g_maxerr, maxerr);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4096, W = argc > 2 ? atoi(argv[2]) : 1024;
  const char* filter = argc > 3 ? argv[3] : nullptr;
  const int reps = argc > 4 ? atoi(argv[4]) : 20;
  float *A, *B, *C, *bias, *stat;
  const size_t act = (size_t)M * W;
  CK(hipMalloc(&A, act * 4)); CK(hipMalloc(&B, (size_t)W * W * 4)); CK(hipMalloc(&C, std::max(act, (size_t)16 * W * W) * 4));
  float* C2; CK(hipMalloc(&C2, (size_t)16 * W * W * 4));
  CK(hipMalloc(&bias, W * 4)); CK(hipMalloc(&stat, (size_t)(M / 32 + 1) * 2 * W * 4));
  std::vector<float> h(act);
  const bool normal = getenv("BENCH_NORMAL") != nullptr;
  for (size_t i = 0; i < act; ++i) {
    if (normal) { double u = 0; for (int k = 0; k < 12; ++k) u += (double)rand() / RAND_MAX; h[i] = (float)(u - 6.0); }
    else h[i] = (float)((double)rand() / RAND_MAX - 0.5);
  }
  CK(hipMemcpy(A, h.data(), act * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, h.data(), (size_t)W * W * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, h.data(), W * 4, hipMemcpyHostToDevice));
  const double flop = 2.0 * M * W * (double)W;

  GemmParams f{};   // forward: Z = A W^T + b, stats
  f.A = A; f.lda = W; f.B = B; f.ldb = W; f.C = C; f.ldc = W; f.M = M; f.N = W; f.K = W; f.k_per_split = W;
  f.bias = bias; f.stat_part = stat;
  GemmParams d = f;  // dgrad: dA = dZ W
  GemmParams w{};    // wgrad: dW = dZ^T A, split over the batch
  w.A = A; w.lda = W; w.B = A; w.ldb = W; w.C = C; w.ldc = W; w.M = W; w.N = W; w.K = M;

#define ROW(name, BM, BN, WM, WN, PIPE) if (!filter || strstr(name, filter))                                                              \
  {                                                                                            \
    float t1 = run<BM, BN, WM, WN, ROWK, ROWK, EPI_BIAS_STATS, PIPE>(f, 1, reps);                    \
    verify("fwd", h, h, h, C, M, W, W, W, 0, 1);                                               \
    float t2 = run<BM, BN, WM, WN, ROWK, KROW, EPI_STORE, PIPE>(d, 1, reps);                         \
    verify("dgrad", h, h, h, C, M, W, W, W, 1, 1);                                             \
    int tiles = (int)(ceil_div(W, BM) * ceil_div(W, BN));                                      \
    int splits = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(256, tiles), 16));      \
    if (BM * BN <= 64 * 128) splits = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(512, tiles), 16)); \
    w.k_per_split = (int)round_up(ceil_div(M, splits), 32); w.c_split_stride = (int64_t)W * W; \
    float t3 = run<BM, BN, WM, WN, KROW, KROW, EPI_STORE, PIPE>(w, splits, reps);                    \
    verify("wgrad", h, h, h, C, W, W, M, W, 2, splits);                                        \
    printf("%-22s fwd %7.1f us %6.1f TF | dgrad %7.1f us %6.1f TF | wgrad(x%d) %7.1f us %6.1f TF\n", name, \
           t1 * 1e3, flop / t1 / 1e9, t2 * 1e3, flop / t2 / 1e9, splits, t3 * 1e3, flop / t3 / 1e9); \
  }

#define RROW(name, BM, BN, WM, WN, BKT, ST) if (!filter || strstr(name, filter))                                                              \
  {                                                                                            \
    float t1 = run_ring<BM, BN, WM, WN, ROWK, ROWK, EPI_BIAS_STATS, BKT, ST>(f, 1, reps);                    \
    verify("fwd", h, h, h, C, M, W, W, W, 0, 1);                                               \
    float t2 = run_ring<BM, BN, WM, WN, ROWK, KROW, EPI_STORE, BKT, ST>(d, 1, reps);                         \
    verify("dgrad", h, h, h, C, M, W, W, W, 1, 1);                                             \
    int tiles = (int)(ceil_div(W, BM) * ceil_div(W, BN));                                      \
    int splits = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div((BM * BN <= 64 * 128) ? 512 : 256, tiles), 16));      \
    w.k_per_split = (int)round_up(ceil_div(M, splits), 64); w.c_split_stride = (int64_t)W * W; \
    float t3 = run_ring<BM, BN, WM, WN, KROW, KROW, EPI_STORE, BKT, ST>(w, splits, reps);                    \
    verify("wgrad", h, h, h, C, W, W, M, W, 2, splits);                                        \
    printf("%-22s fwd %7.1f us %6.1f TF | dgrad %7.1f us %6.1f TF | wgrad(x%d) %7.1f us %6.1f TF\n", name, \
           t1 * 1e3, flop / t1 / 1e9, t2 * 1e3, flop / t2 / 1e9, splits, t3 * 1e3, flop / t3 / 1e9); \
    if (getenv("BENCH_PAIR")) {                                                                \
      GemmParams w2 = w; w2.C = C2;                                                            \
      float tp = run_pair<BM, BN, WM, WN, BKT, ST>(d, w2, splits, reps);                         \
      printf("%-22s dgrad || wgrad on two streams: %7.1f us per pair (sum of the two alone %7.1f)\n", name, tp * 1e3, (t2 + t3) * 1e3); \
    }                                                                                          \
  }
  // interleaved rounds in one process (guide rule 24): shipped r01 kernel vs the ring variants
  for (int round = 0; round < (getenv("BENCH_ROUNDS") ? atoi(getenv("BENCH_ROUNDS")) : 1); ++round) {
  ROW("128x128 w4x2 pipe3", 128, 128, 4, 2, 3)
  RROW("ring 4x2 bk32 st3", 128, 128, 4, 2, 32, 3)
  RROW("ring 4x2 bk32 st4", 128, 128, 4, 2, 32, 4)
  RROW("ring 4x2 bk64 st2", 128, 128, 4, 2, 64, 2)
  RROW("ring 4x2 bk32 st2", 128, 128, 4, 2, 32, 2)
  RROW("ring 2x2 bk32 st2", 128, 128, 2, 2, 32, 2)
  RROW("ring 64x128 2x2 bk32 st2", 64, 128, 2, 2, 32, 2)
  RROW("ring 64x128 2x2 bk32 st3", 64, 128, 2, 2, 32, 3)
  RROW("ring 128x64 2x2 bk32 st3", 128, 64, 2, 2, 32, 3)
  RROW("ring 64x128 2x2 bk64 st2", 64, 128, 2, 2, 64, 2)
  }
  {  // where the time of the shipped ring kernel goes: in-kernel stamps of the forward kernel
    unsigned long long* st; CK(hipMalloc(&st, 4096 * 64));
    auto stamped = [&](const char* name, auto kern, size_t lds, const GemmParams& base) {
      GemmParams fs = base; fs.loss_part = reinterpret_cast<float*>(st);
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      const int tiles = (int)(ceil_div(fs.M, 128) * ceil_div(fs.N, 128));
      hipEvent_t e0v, e1v; CK(hipEventCreate(&e0v)); CK(hipEventCreate(&e1v));
      for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, 0, fs);
      CK(hipEventRecord(e0v, 0));
      for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, 0, fs);
      CK(hipEventRecord(e1v, 0));
      CK(hipDeviceSynchronize());
      float msv; CK(hipEventElapsedTime(&msv, e0v, e1v));
      std::vector<unsigned long long> hs(8 * tiles);
      CK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
      double cs = 0, rs = 0, pro = 0, epi = 0; unsigned long long e0 = ~0ull, e1 = 0, l1max = 0, emax = 0;
      for (int i = 0; i < tiles; ++i) {
        const unsigned long long* o = &hs[8 * i];
        cs += o[0]; rs += o[1]; pro += o[3] - o[2]; epi += o[5] - o[4];
        e0 = std::min(e0, o[2]); e1 = std::max(e1, o[5]); l1max = std::max(l1max, o[4]); emax = std::max(emax, o[2]);
      }
      const double ideal = 2.0 * fs.K * 128 * 128 / 2048.0 / 4.0 * 64.0 / 2.0;   // (2 FLOP per multiply-add)   // cycles: MFMAs per SIMD x 64
      printf("%s: events %.1f us/launch | per WG: entry->loop %.2f us, main loop %.1f us (%.0f cycles, %.3f GHz, MFMA-ideal %.0f), epilogue %.2f us | span first entry -> last exit %.1f us, last entry +%.2f us, last loop end +%.1f us\n",
             name, msv / 200 * 1e3, pro / tiles / 100.0, rs / tiles / 100.0, cs / tiles, cs / rs * 0.1, ideal,
             epi / tiles / 100.0, (e1 - e0) / 100.0, (emax - e0) / 100.0, (l1max - e0) / 100.0);
    };
    stamped("STAMP ring bk64 st2 fwd  ", gemm_f32_ring_kernel<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 64, 2, 1>, gemm_ring_lds_bytes<128, 128, 64, 2>(), f);
    stamped("STAMP ring bk64 st2 dgrad", gemm_f32_ring_kernel<128, 128, 4, 2, ROWK, KROW, EPI_STORE, 64, 2, 1>, gemm_ring_lds_bytes<128, 128, 64, 2>(), d);
    stamped("ABL no barrier  bk64 st2 ", gemm_f32_ring_kernel<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 64, 2, 2>, gemm_ring_lds_bytes<128, 128, 64, 2>(), f);
    stamped("ABL no DMA      bk64 st2 ", gemm_f32_ring_kernel<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 64, 2, 3>, gemm_ring_lds_bytes<128, 128, 64, 2>(), f);
    stamped("ABL no ds_read  bk64 st2 ", gemm_f32_ring_kernel<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 64, 2, 4>, gemm_ring_lds_bytes<128, 128, 64, 2>(), f);
    stamped("ABL no wait+bar bk64 st2 ", gemm_f32_ring_kernel<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 64, 2, 5>, gemm_ring_lds_bytes<128, 128, 64, 2>(), f);
    stamped("ABL MFMAs only  bk64 st2 ", gemm_f32_ring_kernel<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 64, 2, 6>, gemm_ring_lds_bytes<128, 128, 64, 2>(), f);
    stamped("STAMP ring bk32 st2 fwd  ", gemm_f32_ring_kernel<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 32, 2, 1>, gemm_ring_lds_bytes<128, 128, 32, 2>(), f);
    GemmParams f1 = f; f1.K = 64; f1.k_per_split = 64;   // fixed per-launch cost: one K tile
    float tk = run_ring<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 64, 2>(f1, 1, 200);
    GemmParams f2 = f; f2.K = 512; f2.k_per_split = 512;
    float tk2 = run_ring<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 64, 2>(f2, 1, 200);
    printf("ring bk64 st2 fwd at K=64: %.1f us ; K=512: %.1f us ; K=%d: see table (slope = main loop)\n", tk * 1e3, tk2 * 1e3, W);
  }
  if (getenv("BENCH_RING_ONLY")) { printf("max spot-check error %g\n", g_maxerr); return 0; }
  ROW("128x128 w4x2 pipe1", 128, 128, 4, 2, 1)
  ROW("128x128 w4x2 pipe3", 128, 128, 4, 2, 3)
  ROW("128x128 w2x4 pipe3", 128, 128, 2, 4, 3)
  ROW("128x128 w2x2 pipe3", 128, 128, 2, 2, 3)
  if (getenv("BENCH_COLD")) {  // cold-cache timing: evict L2 + Infinity Cache before every launch
    char* big; const size_t bigsz = (size_t)768 << 20; CK(hipMalloc(&big, bigsz));
    auto cold = [&](const char* name, auto kern, const GemmParams& pp, int splits, size_t lds, int nt) {
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      const int tiles = (int)(ceil_div(pp.M, 128) * ceil_div(pp.N, 128));
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      double tot_cold = 0, tot_warm = 0; const int n = 40;
      for (int i = 0; i < n; ++i) {
        CK(hipMemsetAsync(big, i, bigsz, 0));
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(kern, dim3(tiles, 1, splits), dim3(nt), lds, 0, pp);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); tot_cold += ms;
        CK(hipEventRecord(a, 0));
        hipLaunchKernelGGL(kern, dim3(tiles, 1, splits), dim3(nt), lds, 0, pp);
        CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b)); tot_warm += ms;
      }
      printf("  %-12s cold %.1f us   warm (2nd launch) %.1f us\n", name, tot_cold / n * 1e3, tot_warm / n * 1e3);
    };
    w.k_per_split = (int)round_up(ceil_div(M, 4), 32); w.c_split_stride = (int64_t)W * W;
    cold("fwd", gemm_f32_kernel<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 3>, f, 1, gemm_lds_bytes<128, 128, ROWK, ROWK, 3>(), 512);
    cold("dgrad", gemm_f32_kernel<128, 128, 4, 2, ROWK, KROW, EPI_STORE, 3>, d, 1, gemm_lds_bytes<128, 128, ROWK, KROW, 3>(), 512);
    cold("wgrad", gemm_f32_kernel<128, 128, 4, 2, KROW, KROW, EPI_STORE, 3>, w, 4, gemm_lds_bytes<128, 128, KROW, KROW, 3>(), 512);
    return 0;
  }
  {  // ablations of the PIPE=1 main loop (timing only; results are wrong)
    auto abl = [&](const char* name, auto kern) {
      constexpr size_t lds = gemm_lds_bytes<128, 128, ROWK, ROWK, 3>();
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      const int tiles = (int)(ceil_div(M, 128) * ceil_div(W, 128));
      hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
      for (int i = 0; i < 300; ++i) hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, 0, f);
      CK(hipEventRecord(a, 0));
      for (int i = 0; i < 500; ++i) hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, 0, f);
      CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      printf("  ablate %-34s %.1f us\n", name, ms / 500 * 1e3);
    };
    abl("PIPE 3 (shipped)", gemm_f32_kernel<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 3, 0, 0>);
    abl("PIPE 3 no in-loop DMA", gemm_f32_kernel<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 3, 0, 8>);
    abl("PIPE 3 no wait/barrier", gemm_f32_kernel<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 3, 0, 16>);
    abl("PIPE 3 no DMA, no wait/barrier", gemm_f32_kernel<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 3, 0, 24>);
    abl("none", gemm_f32_kernel<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 1, 0, 0>);
    abl("no global loads", gemm_f32_kernel<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 1, 0, 1>);
    abl("no ds_write (loads unused)", gemm_f32_kernel<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 1, 0, 2>);
    abl("no loads, no ds_write", gemm_f32_kernel<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 1, 0, 3>);
    abl("no loads/ds_write/barrier", gemm_f32_kernel<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 1, 0, 7>);
  }
  {  // fixed per-launch cost: same kernel, K = 32 (one K tile), and without the output store traffic
    GemmParams f1 = f; f1.K = 32; f1.k_per_split = 32;
    float t1 = run<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 1>(f1, 1, 50);
    GemmParams f2 = f1; f2.M = 128;   // a single row of tiles: 8 workgroups
    float t2 = run<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 1>(f2, 1, 50);
    printf("K=32 launch: %d tiles %.1f us ; 8 tiles %.1f us\n", (int)(ceil_div(M, 128) * ceil_div(W, 128)), t1 * 1e3, t2 * 1e3);
  }
  {  // in-kernel clock of the main loop (diagnostic STAMP build of the forward kernel)
    unsigned long long* st; CK(hipMalloc(&st, 4096 * 64));
    GemmParams fs = f; fs.loss_part = reinterpret_cast<float*>(st);
    auto kern = gemm_f32_kernel<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 3, 1>;
    constexpr size_t lds = gemm_lds_bytes<128, 128, ROWK, ROWK, 3>();
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const int tiles = (int)(ceil_div(M, 128) * ceil_div(W, 128));
    hipEvent_t e0v, e1v; CK(hipEventCreate(&e0v)); CK(hipEventCreate(&e1v));
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, 0, fs);
    CK(hipEventRecord(e0v, 0));
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, 0, fs);
    CK(hipEventRecord(e1v, 0));
    CK(hipDeviceSynchronize());
    float msv; CK(hipEventElapsedTime(&msv, e0v, e1v));
    printf("stamped kernel, events: %.1f us per launch\n", msv / 200 * 1e3);
    std::vector<unsigned long long> hs(8 * tiles);
    CK(hipMemcpy(hs.data(), st, hs.size() * 8, hipMemcpyDeviceToHost));
    double cs = 0, rs = 0, pro = 0, epi = 0; unsigned long long e0 = ~0ull, e1 = 0, l1max = 0, emax = 0;
    for (int i = 0; i < tiles; ++i) {
      const unsigned long long* o = &hs[8 * i];
      cs += o[0]; rs += o[1]; pro += o[3] - o[2]; epi += o[5] - o[4];
      e0 = std::min(e0, o[2]); e1 = std::max(e1, o[5]); l1max = std::max(l1max, o[4]); emax = std::max(emax, o[2]);
    }
    printf("per WG: entry->loop %.2f us | main loop %.1f us (%.0f cycles, clock %.3f GHz, MFMA-ideal 131072) | epilogue %.2f us\n",
           pro / tiles / 100.0, rs / tiles / 100.0, cs / tiles, cs / rs * 0.1, epi / tiles / 100.0);
    printf("kernel span first entry -> last exit %.1f us; last WG entry at +%.2f us; last loop end at +%.1f us\n",
           (e1 - e0) / 100.0, (emax - e0) / 100.0, (l1max - e0) / 100.0);
  }
  printf("max spot-check error %g\n", g_maxerr);
  return 0;
}
