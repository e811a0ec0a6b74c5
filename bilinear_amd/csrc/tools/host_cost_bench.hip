// Developer tool: HOST time of enqueuing the small-batch fused step through the C-ABI (blh_train_step, batch 64),
// next to the raw cost of launching an empty kernel with a 64-byte and a 3 KB by-value argument block.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/host_cost_bench.hip -I../../include -L../lib -lbilinear_hip
//         -Wl,-rpath,'$ORIGIN' -o ../lib/host_cost_bench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "bilinear_hip.h"
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
struct Big { char b[3072]; };
struct Small { char b[64]; };
__global__ void k_big(Big) {}
__global__ void k_small(Small) {}
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 64;
  blh_model_desc d{2, 1024, 32, 48, 0};
  blh_context* ctx; if (blh_context_create(&ctx) != BLH_OK) return 1;
  const int64_t n = blh_param_arena_floats(&d), nr = blh_bn_running_floats(&d), wsb = blh_workspace_bytes(&d, B);
  auto dev = [&](size_t bytes) { void* q; CK(hipMalloc(&q, bytes)); CK(hipMemset(q, 0, bytes)); return q; };
  std::vector<float> h(n);
  for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.06f;
  float* params = (float*)dev(n * 4); CK(hipMemcpy(params, h.data(), n * 4, hipMemcpyHostToDevice));
  float* grads = (float*)dev(n * 4); float* m = (float*)dev(n * 4); float* v = (float*)dev(n * 4);
  float* run = (float*)dev(nr * 4); int64_t* nbt = (int64_t*)dev(64 * 8);
  float* x = (float*)dev(B * 32 * 4); float* t = (float*)dev(B * 48 * 4); float* pred = (float*)dev(B * 48 * 4);
  float* loss = (float*)dev(16); float* stats = (float*)dev(16);
  void* ws = dev(wsb);
  blh_dropout drop{}; drop.seed = 7;
  blh_adam_hyper hy{1e-3, 0.9, 0.999, 1e-8, 1.0, 0, 0};
  auto step = [&](int i) {
    hy.step = i + 1; drop.step = i;
    return blh_train_step(ctx, &d, nullptr, params, grads, m, v, run, nbt, x, t, &drop, 0.1f, &hy, ws, wsb, pred, loss,
                          stats, B);
  };
  for (int i = 0; i < 50; ++i) if (step(i) != BLH_OK) { printf("step failed\n"); return 1; }
  CK(hipDeviceSynchronize());
  for (int rep = 0; rep < 3; ++rep) {
    const int N = 40;
    const double t0 = now();
    for (int i = 0; i < N; ++i) step(50 + i);
    const double t1 = now();
    CK(hipDeviceSynchronize());
    const double t2 = now();
    printf("blh_train_step B=%d: host enqueue %.1f us/step, until done %.1f us/step\n", B, (t1 - t0) / N, (t2 - t0) / N);
  }
  for (int rep = 0; rep < 2; ++rep) {
    const int N = 500;
    Big big{}; Small sm{};
    double t0 = now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_small, dim3(1), dim3(64), 0, 0, sm);
    double t1 = now(); CK(hipDeviceSynchronize());
    printf("empty kernel, 64 B args: host %.2f us/launch\n", (t1 - t0) / N);
    t0 = now();
    for (int i = 0; i < N; ++i) hipLaunchKernelGGL(k_big, dim3(1), dim3(64), 0, 0, big);
    t1 = now(); CK(hipDeviceSynchronize());
    printf("empty kernel, 3 KB args: host %.2f us/launch\n", (t1 - t0) / N);
  }
  return 0;
}
