// Developer tool: bandwidth of the streaming BN kernels vs a plain float4 copy.
#include <cstdio>
#include <cstdlib>
#include "../elementwise.hip"
#include "../bn_f32.hip"
#include "../bn_bf16.hip"
using namespace blh;
thread_local int blh::g_last_hip_error = 0;
thread_local hipEvent_t blh::tl_stop_event = nullptr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

__global__ void copy4(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i];
}

template <class F> float timeit(F f, int reps = 20) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) f();
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
  const int64_t B = argc > 1 ? atoi(argv[1]) : 4096; const int W = argc > 2 ? atoi(argv[2]) : 1024;
  const size_t n = (size_t)B * W;
  float *Z, *A, *S, *G, *vec, *part;
  CK(hipMalloc(&Z, n * 4)); CK(hipMalloc(&A, n * 4)); CK(hipMalloc(&S, n * 4)); CK(hipMalloc(&G, n * 4));
  CK(hipMalloc(&vec, 8 * W * 4)); CK(hipMalloc(&part, (size_t)512 * 2 * W * 4));
  CK(hipMemset(Z, 0, n * 4)); CK(hipMemset(S, 0, n * 4)); CK(hipMemset(G, 0, n * 4)); CK(hipMemset(vec, 0, 8 * W * 4));
  DropoutSrc d{nullptr, 1, 2, 0, 1, nullptr};
  uint32_t* kb; CK(hipMalloc(&kb, bn_keepbits_words_f32(B, W) * 4));
  const double mb = n * 4 / 1e6;
  float t;
  t = timeit([&] { hipLaunchKernelGGL(copy4, dim3(2048), dim3(256), 0, 0, (const float4*)Z, (float4*)A, n / 4); });
  printf("copy4 (2048 blocks)        %7.1f us  %6.2f TB/s\n", t, 2 * mb / t);
  t = timeit([&] { hipLaunchKernelGGL(copy4, dim3(512), dim3(256), 0, 0, (const float4*)Z, (float4*)A, n / 4); });
  printf("copy4 (512 blocks)         %7.1f us  %6.2f TB/s\n", t, 2 * mb / t);
  t = timeit([&] { launch_bn_apply_f2(0, true, Z, vec, vec + W, nullptr, nullptr, nullptr, nullptr, nullptr, A, kb, B, W, d, nullptr); });
  printf("bn_apply philox            %7.1f us  %6.2f TB/s\n", t, 2 * mb / t);
  uint8_t* keep; CK(hipMalloc(&keep, n)); CK(hipMemset(keep, 1, n));
  DropoutSrc dk{keep, 0, 0, 0, 1, nullptr};
  t = timeit([&] { launch_bn_apply_f2(0, true, Z, vec, vec + W, nullptr, nullptr, nullptr, nullptr, nullptr, A, kb, B, W, dk, nullptr); });
  printf("bn_apply explicit mask     %7.1f us  %6.2f TB/s\n", t, 2.25 * mb / t);
  for (int rep = 0; rep < 3; ++rep) {
    t = timeit([&] { launch_bn_apply_f2(0, true, Z, vec, vec + W, nullptr, nullptr, nullptr, nullptr, nullptr, A, kb, B, W, d, nullptr); }, 200);
    printf("bn_apply philox x200       %7.1f us\n", t);
  }
  t = timeit([&] { launch_dropout_mask(0, keep, B, W, d); });
  printf("dropout_mask kernel        %7.1f us\n", t);
  t = timeit([&] { launch_bn_apply_f2(0, true, Z, vec, vec + W, nullptr, nullptr, nullptr, nullptr, S, A, kb, B, W, d, nullptr); });
  printf("bn_apply philox + skip     %7.1f us  %6.2f TB/s\n", t, 3 * mb / t);
  t = timeit([&] { launch_bn_apply_f2(0, false, Z, nullptr, nullptr, vec, vec + W, vec + 2 * W, vec + 3 * W, nullptr, A, nullptr, B, W, d, nullptr); });
  printf("bn_apply eval (no philox)  %7.1f us  %6.2f TB/s\n", t, 2 * mb / t);
  t = timeit([&] { launch_bn_bwd_reduce_f2(0, G, Z, vec, vec + W, kb, part, B, W); });
  printf("bn_bwd_reduce              %7.1f us  %6.2f TB/s\n", t, 2 * mb / t);
  t = timeit([&] { launch_bn_bwd_apply_f2(0, G, Z, vec, vec + W, vec + 2 * W, vec + 3 * W, vec + 4 * W, vec + 5 * W, kb, A, part, B, W, B); });
  printf("bn_bwd_apply               %7.1f us  %6.2f TB/s\n", t, 3 * mb / t);
  t = timeit([&] { launch_colreduce(0, part, 128, 2 * W, 2 * W, vec + 6 * W); });
  printf("colreduce 128x%d          %7.1f us\n", 2 * W, t);
  return 0;
}
