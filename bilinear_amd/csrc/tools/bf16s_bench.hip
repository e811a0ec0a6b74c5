// Developer tool (not part of the library): K-tile / ring-depth variants of the bf16-storage GEMM
// at the config-3 / config-5 shapes.  hipcc -O3 --offload-arch=gfx950 tools/bf16s_bench.hip -o bf16s_bench
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../gemm_bf16s_kernel.h"

using namespace blh;
thread_local int blh::g_last_hip_error = 0;
thread_local hipEvent_t blh::tl_stop_event = nullptr;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int LA, int LB, int EPI, bool OB, int BKE, int ST, int TM = 2, int TN = 2, int ABL = 0>
float run(const GemmParamsH& p, int splits, int reps) {
  constexpr size_t lds = gemm_bf16s_lds_bytes<BKE, ST, TM, TN>();
  auto kern = gemm_bf16s_kernel<LA, LB, EPI, OB, BKE, ST, TM, TN, ABL>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int tiles = (int)(ceil_div(p.M, 64 * TM) * ceil_div(p.N, 64 * TN));
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

static float bf(uint16_t v) { uint32_t u = (uint32_t)v << 16; float f; memcpy(&f, &u, 4); return f; }
// spot check of the bf16 output against the host dot product (kind 0: fwd A W^T, 1: dgrad A W)
static void verify(const char* what, const std::vector<uint16_t>& h, const uint16_t* dC, int M, int W, int kind) {
  std::vector<uint16_t> c((size_t)M * W);
  CK(hipMemcpy(c.data(), dC, c.size() * 2, hipMemcpyDeviceToHost));
  double worst = 0;
  for (int t = 0; t < 96; ++t) {
    const int i = (int)(((long long)t * 7919 + 13) % M), j = (t * 104729 + 7) % W;
    double ref = 0;
    for (int k = 0; k < W; ++k) ref += (double)bf(h[(size_t)i * W + k]) * (kind == 0 ? bf(h[(size_t)j * W + k]) : bf(h[(size_t)k * W + j]));
    worst = std::max(worst, std::abs(bf(c[(size_t)i * W + j]) - ref) / (std::abs(ref) + 0.05));
  }
  if (worst > 2e-2) printf("   !! %s MISMATCH rel err %g\n", what, worst);
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 16384, W = argc > 2 ? atoi(argv[2]) : 1024;
  const int reps = argc > 3 ? atoi(argv[3]) : 100;
  uint16_t *A, *B, *C; float *bias, *stat, *slab;
  CK(hipMalloc(&A, (size_t)M * W * 2)); CK(hipMalloc(&B, (size_t)W * W * 2)); CK(hipMalloc(&C, (size_t)M * W * 2));
  CK(hipMalloc(&bias, W * 4)); CK(hipMalloc(&stat, (size_t)(M / 64 + 1) * 2 * W * 4)); CK(hipMalloc(&slab, (size_t)16 * W * W * 4));
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
#define BIG(TM, TN, BKE, ST)                                                                            \
  {                                                                                                     \
    CK(hipMemset(C, 0, (size_t)M * W * 2));                                                             \
    float t1 = run<ROWK, ROWK, EPI_BIAS_STATS, true, BKE, ST, TM, TN>(f, 1, reps);                      \
    verify("fwd", h, C, M, W, 0);                                                                       \
    CK(hipMemset(C, 0, (size_t)M * W * 2));                                                             \
    float t2 = (TN == 2) ? run<ROWK, KROW, EPI_STORE, true, BKE, ST, TM, 2>(d, 1, reps) : 0.f;          \
    if (TN == 2) verify("dgrad", h, C, M, W, 1);                                                        \
    printf("tile %dx%d bk%-3d st%d (%3d KB LDS)  fwd %7.1f us %6.0f TF | dgrad %7.1f us %6.0f TF\n", 64 * TM, 64 * TN, \
           BKE, ST, (int)(gemm_bf16s_lds_bytes<BKE, ST, TM, TN>() / 1024), t1 * 1e3, flop / t1 / 1e9, t2 * 1e3, \
           t2 > 0 ? flop / t2 / 1e9 : 0.0);                                                             \
  }
#define ABLROW(TM, TN, BKE, ST)                                                                         \
  {                                                                                                     \
    float t0 = run<ROWK, ROWK, EPI_BIAS_STATS, true, BKE, ST, TM, TN, 0>(f, 1, reps);                   \
    float t1 = run<ROWK, ROWK, EPI_BIAS_STATS, true, BKE, ST, TM, TN, 1>(f, 1, reps);                   \
    float t2 = run<ROWK, ROWK, EPI_BIAS_STATS, true, BKE, ST, TM, TN, 2>(f, 1, reps);                   \
    float t3 = run<ROWK, ROWK, EPI_BIAS_STATS, true, BKE, ST, TM, TN, 3>(f, 1, reps);                   \
    float t4 = run<ROWK, ROWK, EPI_BIAS_STATS, true, BKE, ST, TM, TN, 4>(f, 1, reps);                   \
    printf("ablation fwd tile %dx%d bk%d st%d: full %.1f us | no DMA %.1f | no wait+barrier %.1f | no frag reads %.1f | no MFMA %.1f\n", \
           64 * TM, 64 * TN, BKE, ST, t0 * 1e3, t1 * 1e3, t2 * 1e3, t3 * 1e3, t4 * 1e3);                \
  }
  ABLROW(2, 2, 64, 2)
  ABLROW(4, 4, 64, 2)
  ABLROW(4, 4, 32, 4)
  for (int round = 0; round < 1; ++round) {
    BIG(4, 4, 32, 4)
    BIG(4, 4, 32, 3)
    BIG(4, 2, 32, 4)
    BIG(2, 2, 32, 4)
    BIG(2, 2, 32, 3)
    BIG(4, 2, 64, 2)
    BIG(4, 4, 64, 2)
    ROW(64, 2)
    ROW(128, 2)
  }
  return 0;
}
