// Developer tool: the one-pass decode kernel (skinny.hip: decode_fused_kernel) alone, with phases removed
// (timing only).  hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/decode_bench.hip -o ../lib/decode_bench
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../skinny.hip"

using namespace blh;
thread_local int blh::g_last_hip_error = 0;
thread_local hipEvent_t blh::tl_stop_event = nullptr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int ABL>
float run(const float* A, const float* Wd, const float* bd, const float* t, float* pred, float* dpred, float* dA,
          float* lp, float* dbp, int64_t batch, int reps) {
  auto kern = decode_fused_kernel<128, ABL>;
  const size_t lds = (size_t)(8 * 16 * 128 + 16 * 52 + 16 * 48 + 4) * sizeof(float);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  const int blocks = (int)((batch + 15) / 16);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, A, Wd, bd, t, pred, dpred, dA, lp, dbp, batch, 1e-5f);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), lds, 0, A, Wd, bd, t, pred, dpred, dA, lp, dbp, batch, 1e-5f);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
  const int64_t B = argc > 1 ? atoll(argv[1]) : 4096;
  const int W = 1024, OF = 48, reps = 300;
  float *A, *Wd, *bd, *t, *pred, *dpred, *dA, *lp, *dbp;
  CK(hipMalloc(&A, B * W * 4)); CK(hipMalloc(&Wd, OF * W * 4)); CK(hipMalloc(&bd, OF * 4)); CK(hipMalloc(&t, B * OF * 4));
  CK(hipMalloc(&pred, B * OF * 4)); CK(hipMalloc(&dpred, B * OF * 4)); CK(hipMalloc(&dA, B * W * 4));
  CK(hipMalloc(&lp, 4096 * 4)); CK(hipMalloc(&dbp, 4096 * OF * 4));
  std::vector<float> h((size_t)B * W);
  for (auto& v : h) v = (float)rand() / RAND_MAX - 0.5f;
  CK(hipMemcpy(A, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(Wd, h.data(), (size_t)OF * W * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(bd, h.data(), OF * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(t, h.data(), (size_t)B * OF * 4, hipMemcpyHostToDevice));
  for (int round = 0; round < 2; ++round)
    printf("B %lld: full %5.1f us | no phase 2 %5.1f | no phase-1 MFMAs %5.1f | neither %5.1f | empty kernel %5.1f\n", (long long)B,
           run<0>(A, Wd, bd, t, pred, dpred, dA, lp, dbp, B, reps), run<1>(A, Wd, bd, t, pred, dpred, dA, lp, dbp, B, reps),
           run<2>(A, Wd, bd, t, pred, dpred, dA, lp, dbp, B, reps), run<3>(A, Wd, bd, t, pred, dpred, dA, lp, dbp, B, reps),
           run<4>(A, Wd, bd, t, pred, dpred, dA, lp, dbp, B, reps));
  printf("   phase 2 without its stores %5.1f us | without its MFMAs %5.1f us\n", run<5>(A, Wd, bd, t, pred, dpred, dA, lp, dbp, B, reps),
         run<6>(A, Wd, bd, t, pred, dpred, dA, lp, dbp, B, reps));
  return 0;
}
