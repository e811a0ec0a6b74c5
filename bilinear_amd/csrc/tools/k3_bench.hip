// Developer tool: SURVEY K3 "normalise on load" measured on the shipped fp32 ring GEMM.
// Forward Linear at M = batch, N = K = width, BK 64 x 2 stages (the shipped forward configuration):
//   (a) plain: A = post-activation tensor (what the library does: bn_apply materialises it);
//   (b) PRE  : A = pre-BatchNorm Z, BN-apply + ReLU (x2) on the A fragments inside the GEMM.
// The dropout keep bit is not applied here (it would add one 1 KiB LDS-DMA piece per K tile and one
// v_cndmask per element); the point is the cost of the extra VALU work between the fragment reads
// and the MFMAs.   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/k3_bench.hip -o ../lib/k3_bench
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../gemm_f32_ring.h"
using namespace blh;
thread_local int blh::g_last_hip_error = 0;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int PRE>
float run(const GemmParams& p, int reps) {
  constexpr size_t lds = gemm_ring_lds_bytes<128, 128, 64, 2>() + (PRE ? 2 * 2048 * 4 : 0);
  auto kern = gemm_f32_ring_kernel<128, 128, 4, 2, ROWK, ROWK, EPI_BIAS_STATS, 64, 2, 0, PRE>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int tiles = (int)(ceil_div(p.M, 128) * ceil_div(p.N, 128));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, 0, p);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(tiles), dim3(512), lds, 0, p);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4096, W = argc > 2 ? atoi(argv[2]) : 1024;
  const int reps = argc > 3 ? atoi(argv[3]) : 300;
  float *A, *Z, *B, *C, *bias, *stat, *sc, *sh;
  CK(hipMalloc(&A, (size_t)M * W * 4)); CK(hipMalloc(&Z, (size_t)M * W * 4)); CK(hipMalloc(&B, (size_t)W * W * 4));
  CK(hipMalloc(&C, (size_t)M * W * 4)); CK(hipMalloc(&bias, W * 4)); CK(hipMalloc(&stat, (size_t)(M / 64 + 1) * 2 * W * 4));
  CK(hipMalloc(&sc, W * 4)); CK(hipMalloc(&sh, W * 4));
  std::vector<float> hz((size_t)M * W), ha((size_t)M * W), hb((size_t)W * W), hs(W), ht(W), h0(W, 0.f);
  for (auto& v : hz) v = (float)rand() / RAND_MAX * 2.f - 1.f;
  for (auto& v : hb) v = ((float)rand() / RAND_MAX - 0.5f) * 0.06f;
  for (int k = 0; k < W; ++k) { hs[k] = 2.f * (0.8f + 0.4f * rand() / RAND_MAX); ht[k] = 2.f * (0.2f * rand() / RAND_MAX - 0.1f); }
  for (size_t i = 0; i < hz.size(); ++i) ha[i] = fmaxf(fmaf(hz[i], hs[i % W], ht[i % W]), 0.f);
  CK(hipMemcpy(Z, hz.data(), hz.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(A, ha.data(), ha.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(B, hb.data(), hb.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bias, h0.data(), W * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(sc, hs.data(), W * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(sh, ht.data(), W * 4, hipMemcpyHostToDevice));
  GemmParams p{};
  p.lda = W; p.B = B; p.ldb = W; p.C = C; p.ldc = W; p.M = M; p.N = W; p.K = W; p.k_per_split = W;
  p.bias = bias; p.stat_part = stat; p.bn_gamma = sc; p.bn_beta = sh;
  const double flop = 2.0 * M * W * (double)W;
  std::vector<float> c0((size_t)M * W), c1((size_t)M * W);
  for (int round = 0; round < 3; ++round) {
    p.A = A;
    const float t0 = run<0>(p, reps);
    CK(hipMemcpy(c0.data(), C, c0.size() * 4, hipMemcpyDeviceToHost));
    p.A = Z;
    const float t1 = run<1>(p, reps);
    CK(hipMemcpy(c1.data(), C, c1.size() * 4, hipMemcpyDeviceToHost));
    double md = 0;
    for (size_t i = 0; i < c0.size(); i += 97) md = std::max(md, (double)fabsf(c0[i] - c1[i]));
    printf("M=%d W=%d  plain (A materialised) %6.1f us %6.1f TF | normalise-on-load %6.1f us %6.1f TF | max |diff| %.2e\n",
           M, W, t0, flop / t0 / 1e6, t1, flop / t1 / 1e6, md);
  }
  return 0;
}
