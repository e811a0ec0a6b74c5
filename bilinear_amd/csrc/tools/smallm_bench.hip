// Developer tool: the split-K hidden-layer GEMMs at small batch (512 / 1024 rows: the per-GPU shapes of the headline
// batch split over 8 / 4 GPUs) — tile shape x split count x ring depth, forward (ROWK, ROWK) and data gradient
// (ROWK, KROW), EPI_STORE into slabs.  usage: smallm_bench [M=512] [W=1024] [reps=200]
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "gemm_f32_kernel.h"
#include "../gemm_f32_ring.h"

using namespace blh;
thread_local int blh::g_last_hip_error = 0;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int BM, int BN, int WM, int WN, int LA, int LB, int BKT, int STAGES>
float run_ring(GemmParams p, int splits, int reps) {
  constexpr int NT = 64 * WM * WN;
  constexpr size_t lds = gemm_ring_lds_bytes<BM, BN, BKT, STAGES>();
  auto kern = gemm_f32_ring_kernel<BM, BN, WM, WN, LA, LB, EPI_STORE, BKT, STAGES>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  p.k_per_split = (int)round_up(ceil_div(p.K, splits), 64);
  splits = (int)ceil_div(p.K, p.k_per_split);
  p.c_split_stride = (int64_t)p.M * p.N;
  const int tiles = (int)(ceil_div(p.M, BM) * ceil_div(p.N, BN));
  dim3 grid(tiles, 1, splits);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, grid, dim3(NT), lds, 0, p);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, grid, dim3(NT), lds, 0, p);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("    %3dx%-3d bk%d st%d: %3d tiles x %2d slabs = %4d workgroups, %2d-deep K per workgroup: %6.1f us\n", BM, BN, BKT, STAGES,
         tiles, splits, tiles * splits, p.k_per_split, ms / reps * 1e3);
  return ms / reps;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 512, W = argc > 2 ? atoi(argv[2]) : 1024, reps = argc > 3 ? atoi(argv[3]) : 200;
  float *A, *B, *C;
  CK(hipMalloc(&A, (size_t)M * W * 4)); CK(hipMalloc(&B, (size_t)W * W * 4)); CK(hipMalloc(&C, (size_t)32 * M * W * 4));
  std::vector<float> h((size_t)W * W);
  for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  CK(hipMemcpy(A, h.data(), (size_t)M * W * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, h.data(), (size_t)W * W * 4, hipMemcpyHostToDevice));
  GemmParams f{};
  f.A = A; f.lda = W; f.B = B; f.ldb = W; f.C = C; f.ldc = W; f.M = M; f.N = W; f.K = W;
  for (int pass = 0; pass < 2; ++pass) {
    printf("M = %d, N = K = %d, %s:\n", M, W, pass == 0 ? "forward (ROWK, ROWK)" : "data gradient (ROWK, KROW)");
#define BOTH(BM, BN, WM, WN, BKT, ST, SPL)                                                     \
    if (pass == 0) run_ring<BM, BN, WM, WN, ROWK, ROWK, BKT, ST>(f, SPL, reps);                 \
    else run_ring<BM, BN, WM, WN, ROWK, KROW, BKT, ST>(f, SPL, reps);
    const int t128 = (int)(ceil_div(M, 128) * ceil_div(W, 128)), t64 = (int)(ceil_div(M, 64) * ceil_div(W, 128));
    BOTH(128, 128, 4, 2, 64, 2, (int)ceil_div(256, t128))       // shipped
    BOTH(128, 128, 4, 2, 64, 2, (int)ceil_div(128, t128))
    BOTH(128, 128, 4, 2, 32, 2, (int)ceil_div(256, t128))
    BOTH(128, 128, 4, 2, 32, 2, (int)ceil_div(512, t128))
    BOTH(64, 128, 2, 2, 32, 3, (int)ceil_div(256, t64))
    BOTH(64, 128, 2, 2, 32, 3, (int)ceil_div(512, t64))
    BOTH(64, 128, 2, 2, 32, 3, (int)ceil_div(128, t64))
    BOTH(64, 128, 2, 2, 64, 2, (int)ceil_div(256, t64))
    BOTH(128, 64, 2, 2, 32, 3, (int)ceil_div(256, t64))
  }
  return 0;
}
