// Developer tool: cost of one grid barrier (grid_barrier.h) on 256 resident workgroups, with and without the
// agent-scope release / acquire fences, and with some dirty data to write back.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/grid_barrier_bench.hip -o ../lib/grid_barrier_bench
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "grid_barrier.h"
using namespace blh;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE, int SLEEP = 1>   // 0: full barrier; 1: no fences (atomics only); 2: full + each workgroup dirties `kb` KB before
__global__ __launch_bounds__(256) void bench(uint32_t* bar, int n, float* buf, int kb, unsigned long long* out) {
  GridBarrier b{bar, gridDim.x, 0u};
  b.init();
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < n; ++i) {
    if (MODE == 2)
      for (int k = threadIdx.x; k < kb * 256; k += 256) buf[(size_t)blockIdx.x * kb * 256 + k] = (float)(i + k);
    if (MODE == 1 || MODE == 3 || MODE == 4 || MODE == 5) {
      __syncthreads();
      if (threadIdx.x == 0) {
        if (MODE == 3 || MODE == 5) asm volatile("buffer_wbl2 sc1\n s_waitcnt vmcnt(0)" ::: "memory");
        (void)__hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while ((int32_t)(__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - b.target) < 0)
          if (SLEEP > 0) __builtin_amdgcn_s_sleep(SLEEP);
        if (MODE == 4 || MODE == 5) asm volatile("buffer_inv sc1\n s_waitcnt vmcnt(0)" ::: "memory");
        b.target += b.nwg;
      }
      __syncthreads();
    } else {
      b.sync();
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memrealtime();
  if (blockIdx.x == 0) b.finish();
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
}

int main() {
  uint32_t* bar; CK(hipMalloc(&bar, 64)); CK(hipMemset(bar, 0, 64));
  float* buf; CK(hipMalloc(&buf, (size_t)256 * 64 * 1024)); 
  unsigned long long* out; CK(hipMalloc(&out, 256 * 8));
  const int n = 200;
  auto report = [&](const char* name) {
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> h(256);
    CK(hipMemcpy(h.data(), out, 256 * 8, hipMemcpyDeviceToHost));
    unsigned long long mx = 0; for (auto v : h) mx = v > mx ? v : mx;
    printf("%-40s %.2f us per barrier\n", name, mx * 0.01 / n);
  };
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(bench<0>, dim3(256), dim3(256), 0, 0, bar, n, buf, 0, out); report("release + add + poll + acquire");
    hipLaunchKernelGGL(bench<1>, dim3(256), dim3(256), 0, 0, bar, n, buf, 0, out); report("add + poll only (no fences)");
    hipLaunchKernelGGL((bench<1, 0>), dim3(256), dim3(256), 0, 0, bar, n, buf, 0, out); report("add + poll, no sleep");
    hipLaunchKernelGGL((bench<1, 4>), dim3(256), dim3(256), 0, 0, bar, n, buf, 0, out); report("add + poll, s_sleep 4");
    hipLaunchKernelGGL((bench<1, 16>), dim3(256), dim3(256), 0, 0, bar, n, buf, 0, out); report("add + poll, s_sleep 16");
    hipLaunchKernelGGL((bench<1, 64>), dim3(256), dim3(256), 0, 0, bar, n, buf, 0, out); report("add + poll, s_sleep 64");
    hipLaunchKernelGGL(bench<3>, dim3(256), dim3(256), 0, 0, bar, n, buf, 0, out); report("wbl2 + add + poll");
    hipLaunchKernelGGL(bench<4>, dim3(256), dim3(256), 0, 0, bar, n, buf, 0, out); report("add + poll + inv");
    hipLaunchKernelGGL(bench<5>, dim3(256), dim3(256), 0, 0, bar, n, buf, 0, out); report("wbl2 + add + poll + inv");
    hipLaunchKernelGGL(bench<2>, dim3(256), dim3(256), 0, 0, bar, n, buf, 1, out); report("full, 1 KB dirtied per workgroup");
    hipLaunchKernelGGL(bench<2>, dim3(256), dim3(256), 0, 0, bar, n, buf, 16, out); report("full, 16 KB dirtied per workgroup");
  }
  return 0;
}
