// Developer tool: tile / wave variants of the bf16-MFMA (fp32 storage) GEMM.
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "gemm_bf16_kernel.h"
using namespace blh;
thread_local int blh::g_last_hip_error = 0;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int BM, int BN, int WM, int WN, int LA, int LB, int EPI>
float run(const GemmParams& p, int splits, int reps) {
  constexpr int NT = 64 * WM * WN;
  constexpr size_t lds = gemm_bf16_lds_bytes<BM, BN>();
  auto kern = gemm_bf16_kernel<BM, BN, WM, WN, LA, LB, EPI>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int tiles = (int)(ceil_div(p.M, BM) * ceil_div(p.N, BN));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(tiles, 1, splits), dim3(NT), lds, 0, p);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(tiles, 1, splits), dim3(NT), lds, 0, p);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4096, W = argc > 2 ? atoi(argv[2]) : 1024, reps = 300;
  float *A, *B, *C, *bias, *stat;
  const size_t act = (size_t)M * W;
  CK(hipMalloc(&A, act * 4)); CK(hipMalloc(&B, (size_t)W * W * 4)); CK(hipMalloc(&C, std::max(act, (size_t)16 * W * W) * 4));
  CK(hipMalloc(&bias, W * 4)); CK(hipMalloc(&stat, (size_t)(M / 32 + 1) * 2 * W * 4));
  std::vector<float> h(act);
  for (size_t i = 0; i < act; ++i) h[i] = (float)((double)rand() / RAND_MAX - 0.5);
  CK(hipMemcpy(A, h.data(), act * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, h.data(), (size_t)W * W * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bias, h.data(), W * 4, hipMemcpyHostToDevice));
  const double flop = 2.0 * M * W * (double)W;
  GemmParams f{}; f.A = A; f.lda = W; f.B = B; f.ldb = W; f.C = C; f.ldc = W; f.M = M; f.N = W; f.K = W; f.k_per_split = W;
  f.bias = bias; f.stat_part = stat;
  GemmParams w{}; w.A = A; w.lda = W; w.B = A; w.ldb = W; w.C = C; w.ldc = W; w.M = W; w.N = W; w.K = M;
#define ROW(name, BM, BN, WM, WN)                                                             \
  {                                                                                            \
    float t1 = run<BM, BN, WM, WN, ROWK, ROWK, EPI_BIAS_STATS>(f, 1, reps);                    \
    float t2 = run<BM, BN, WM, WN, ROWK, KROW, EPI_STORE>(f, 1, reps);                         \
    int tiles = (int)(ceil_div(W, BM) * ceil_div(W, BN));                                      \
    int splits = (int)std::max<int64_t>(1, std::min<int64_t>(ceil_div(256, tiles), 16));      \
    w.k_per_split = (int)round_up(ceil_div(M, splits), 32); w.c_split_stride = (int64_t)W * W; \
    float t3 = run<BM, BN, WM, WN, KROW, KROW, EPI_STORE>(w, splits, reps);                    \
    printf("BK=%d %-16s fwd %6.1f us %6.0f TF | dgrad %6.1f us %6.0f TF | wgrad(x%d) %6.1f us %6.0f TF\n", BKH, name, \
           t1, flop / t1 / 1e6, t2, flop / t2 / 1e6, splits, t3, flop / t3 / 1e6);             \
  }
  ROW("128x128 w2x2", 128, 128, 2, 2)
  ROW("128x128 w4x2", 128, 128, 4, 2)
#if BLH_BKH <= 64
  ROW("128x64  w2x2", 128, 64, 2, 2)
  ROW("64x128  w2x2", 64, 128, 2, 2)
  ROW("64x64   w2x2", 64, 64, 2, 2)
#endif
  return 0;
}
