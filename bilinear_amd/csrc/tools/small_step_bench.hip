// Developer tool: phase timeline of the one-launch small-batch step (small_step.hip) from s_memrealtime stamps
// (100 MHz) taken by thread 0 of every workgroup.  Random data; timing only.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/small_step_bench.hip -o ../lib/small_step_bench
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#include "../small_step.hip"
#include "small_step_persistent.h"
#include "../api_layout.h"
using namespace blh;
thread_local int blh::g_last_hip_error = 0;
AdamConsts blh::adam_consts(const blh_adam_hyper&) { return AdamConsts{0.1f, 0.999f, 0.001f, 1e-3f, 0.03f, 1e-8f, 1.f}; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
  const int nb = argc > 1 ? atoi(argv[1]) : 2, W = argc > 2 ? atoi(argv[2]) : 1024, B = argc > 3 ? atoi(argv[3]) : 64;
  const int reps = argc > 4 ? atoi(argv[4]) : 200;
  blh_model_desc d{nb, W, 32, 48, 0};
  const ArenaLayout L = make_layout(&d);
  SmallStepParams p{};
  p.nh = (int)L.heavy.size(); p.W = W; p.in_f = 32; p.out_f = 48; p.batch = B;
  std::vector<float> h(L.total);
  for (auto& v : h) v = (rand() / (float)RAND_MAX - 0.5f) * 0.06f;
  auto dev = [&](size_t n, const float* init) { float* q; CK(hipMalloc(&q, n * 4)); if (init) CK(hipMemcpy(q, init, n * 4, hipMemcpyHostToDevice)); else CK(hipMemset(q, 0, n * 4)); return q; };
  p.params = dev(L.total, h.data()); p.grads = dev(L.total, nullptr); p.m = dev(L.total, nullptr); p.v = dev(L.total, nullptr);
  p.bn_running = dev((size_t)p.nh * 2 * W, nullptr);
  CK(hipMalloc(&p.nbt, 64 * 8)); CK(hipMemset(p.nbt, 0, 64 * 8));
  std::vector<float> hx((size_t)B * 32), ht((size_t)B * 48);
  for (auto& v : hx) v = rand() / (float)RAND_MAX - 0.5f;
  for (auto& v : ht) v = rand() / (float)RAND_MAX - 0.5f;
  p.x = dev(hx.size(), hx.data()); p.target = dev(ht.size(), ht.data());
  for (int i = 0; i < p.nh; ++i) {
    p.w_off[i] = L.heavy[i].w; p.b_off[i] = L.heavy[i].b; p.g_off[i] = L.heavy[i].gamma; p.be_off[i] = L.heavy[i].beta;
    p.A[i] = dev((size_t)B * W, nullptr); p.dZ[i] = dev((size_t)B * W, nullptr);
  }
  p.dec_w = L.dec_w; p.dec_b = L.dec_b; p.count = L.total;
  p.dpred = dev((size_t)B * 48, nullptr); p.pred = dev((size_t)B * 48, nullptr); p.loss_out = dev(4, nullptr);
  p.stats_out = dev(4, nullptr); p.loss_part = dev(64, nullptr);
  CK(hipMalloc(&p.sumsq_part, 1024 * 8));
  CK(hipMalloc(&p.bar, 64)); CK(hipMemset(p.bar, 0, 64));
  p.drop = DropoutSrc{nullptr, 1234, 0, 0, 0, nullptr};
  p.momentum = 0.1f; p.denom = (double)B * 48; p.mse_scale = (float)(2.0 / p.denom);
  p.adam = AdamConsts{0.1f, 0.999f, 0.001f, 1e-3f, 0.0316f, 1e-8f, 1.f};
  int cus = 0;
  const int grid = small_step_max_grid(&cus);
  printf("grid %d (CUs %d), nh %d, W %d, B %d, LDS %zu\n", grid, cus, p.nh, W, B, (size_t)0);
  if (grid <= 0) return 1;
  CK(hipMalloc(&p.stamps, (size_t)grid * 64 * 8)); CK(hipMemset(p.stamps, 0, (size_t)grid * 64 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < 20; ++i) if (launch_small_step(0, p) != BLH_OK) { printf("launch failed\n"); return 1; }
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) launch_small_step(0, p);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipGetLastError());
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("step %.1f us\n", ms / reps * 1e3f);
  std::vector<unsigned long long> st((size_t)grid * 64);
  CK(hipMemcpy(st.data(), p.stamps, st.size() * 8, hipMemcpyDeviceToHost));
  uint32_t bw[3]; CK(hipMemcpy(bw, p.bar, 12, hipMemcpyDeviceToHost));
  printf("barrier timeouts %u\n", bw[2]);
  const int n = 1 + 4 * p.nh + 2 + 4 * p.nh + 2;
  unsigned long long t0 = st[0];
  for (int g = 0; g < grid; ++g) t0 = std::min(t0, st[(size_t)g * 64]);
  printf("stamp  wg0_us  min_us  max_us   (since the earliest start)\n");
  for (int k = 0; k < n && k < 64; ++k) {
    unsigned long long lo = ~0ull, hi = 0;
    for (int g = 0; g < grid; ++g) { lo = std::min(lo, st[(size_t)g * 64 + k]); hi = std::max(hi, st[(size_t)g * 64 + k]); }
    printf("%3d  %7.2f %7.2f %7.2f\n", k, (st[k] - t0) * 0.01, (lo - t0) * 0.01, (hi - t0) * 0.01);
  }
  return 0;
}
