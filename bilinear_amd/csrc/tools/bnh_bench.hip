// Developer tool: the bf16-storage BatchNorm streaming kernels, first generation (elementwise.hip)
// against second generation (bn_bf16.hip), alone on the chip.  BLH_EW_H_CHUNKS sets the row chunks.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/bnh_bench.hip -o ../lib/bnh_bench
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "../elementwise.hip"
#include "../bn_bf16.hip"
using namespace blh;
thread_local int blh::g_last_hip_error = 0;
thread_local hipEvent_t blh::tl_stop_event = nullptr;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <class F> float timeit(F f, int reps = 100) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 10; ++i) f();
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) f();
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps * 1e3f;
}

int main(int argc, char** argv) {
  const int64_t B = argc > 1 ? atoi(argv[1]) : 16384; const int W = argc > 2 ? atoi(argv[2]) : 1024;
  const size_t n = (size_t)B * W;
  uint16_t *Z, *A, *S, *G, *D; float *vec, *part, *cpart; uint32_t* keep;
  CK(hipMalloc(&Z, n * 2)); CK(hipMalloc(&A, n * 2)); CK(hipMalloc(&S, n * 2)); CK(hipMalloc(&G, n * 2)); CK(hipMalloc(&D, n * 2));
  CK(hipMalloc(&vec, 8 * W * 4)); CK(hipMalloc(&part, (size_t)4096 * 2 * W * 4)); CK(hipMalloc(&cpart, (size_t)4096 * W * 4));
  CK(hipMalloc(&keep, bn_keepbits_words(B, W) * 4));
  std::vector<uint16_t> h(n);
  for (auto& v : h) { float f = (float)rand() / RAND_MAX - 0.5f; uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
  CK(hipMemcpy(Z, h.data(), n * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(S, h.data(), n * 2, hipMemcpyHostToDevice));
  CK(hipMemcpy(G, h.data(), n * 2, hipMemcpyHostToDevice));
  std::vector<float> hv(8 * W, 1.0f);
  CK(hipMemcpy(vec, hv.data(), 8 * W * 4, hipMemcpyHostToDevice));
  DropoutSrc d{nullptr, 1, 2, 0, 1, nullptr};
  const double mb = n * 2 / 1e6;
  const int chunks = ew_num_row_chunks_h(B);
  printf("B=%lld W=%d row chunks %d (blocks %d)\n", (long long)B, W, chunks, chunks * (int)ceil_div(W, 512));
  for (int round = 0; round < 2; ++round) {
    float t;
    t = timeit([&] { launch_bn_apply_t(0, true, Z, ET_BF16, vec, vec + W, nullptr, nullptr, nullptr, nullptr, nullptr, ET_BF16, A, ET_BF16, B, W, d, nullptr); });
    printf("gen1 bn_apply            %6.1f us %5.2f TB/s |", t, 2 * mb / t);
    t = timeit([&] { launch_bn_apply_h2(0, true, Z, vec, vec + W, nullptr, nullptr, nullptr, nullptr, nullptr, A, keep, B, W, d, nullptr); });
    printf(" gen2 %6.1f us %5.2f TB/s\n", t, 2 * mb / t);
    t = timeit([&] { launch_bn_apply_t(0, true, Z, ET_BF16, vec, vec + W, nullptr, nullptr, nullptr, nullptr, S, ET_BF16, A, ET_BF16, B, W, d, nullptr); });
    printf("gen1 bn_apply + skip     %6.1f us %5.2f TB/s |", t, 3 * mb / t);
    t = timeit([&] { launch_bn_apply_h2(0, true, Z, vec, vec + W, nullptr, nullptr, nullptr, nullptr, S, A, keep, B, W, d, nullptr); });
    printf(" gen2 %6.1f us %5.2f TB/s\n", t, 3 * mb / t);
    t = timeit([&] { launch_bn_bwd_reduce_t(0, G, ET_BF16, Z, ET_BF16, vec, vec + W, vec + 2 * W, vec + 3 * W, part, B, W, d); });
    printf("gen1 bn_bwd_reduce       %6.1f us %5.2f TB/s |", t, 2 * mb / t);
    t = timeit([&] { launch_bn_bwd_reduce_h2(0, G, Z, vec, vec + W, keep, part, B, W); });
    printf(" gen2 %6.1f us %5.2f TB/s\n", t, 2 * mb / t);
    t = timeit([&] { launch_bn_bwd_finalize(0, part, chunks, W, vec + 4 * W, vec + 5 * W); });
    printf("gen1 finalize (colreduce) %5.1f us            |", t);
    t = timeit([&] { launch_bn_bwd_finalize_h2(0, part, chunks, W, vec + 2 * W, vec + 3 * W, vec + 4 * W, vec + 5 * W); });
    printf(" gen2 %6.1f us\n", t);
    t = timeit([&] { launch_bn_bwd_apply_t(0, G, ET_BF16, Z, ET_BF16, vec, vec + W, vec + 2 * W, vec + 3 * W, vec + 4 * W, vec + 5 * W, D, ET_BF16, cpart, B, W, d, B); });
    printf("gen1 bn_bwd_apply        %6.1f us %5.2f TB/s |", t, 3 * mb / t);
    t = timeit([&] { launch_bn_bwd_apply_h2(0, G, Z, vec, vec + W, vec + 2 * W, vec + 3 * W, vec + 4 * W, vec + 5 * W, keep, D, cpart, B, W, B); });
    printf(" gen2 %6.1f us %5.2f TB/s\n", t, 3 * mb / t);
  }
  return 0;
}
