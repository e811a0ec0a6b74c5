// Developer tool (not part of the library): how fast can a CU move L2-resident data into LDS?
//   path A: LDS-DMA (global_load_lds_dwordx4, the path of the GEMM kernels)
//   path B: global_load_dwordx4 into registers + ds_write_b128
//   path C: global_load_dwordx4 into registers only (L2 -> L1 -> VGPR ceiling)
// Every workgroup streams the same `span` bytes (default 1 MB: L2-resident) `iters` times.
//   hipcc -O3 --offload-arch=gfx950 tools/dma_bw.hip -o ../lib/dma_bw
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

constexpr int NT = 256;
constexpr int CHUNKS = 8;                 // 16-byte chunks per thread per step: 32 KB per workgroup-step

__global__ __launch_bounds__(NT) void dma_kernel(const float* __restrict__ src, int span_floats, int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const uint32_t lds0 = (uint32_t)(uintptr_t)((__attribute__((address_space(3))) float*)smem);
  const uint32_t wave_off = __builtin_amdgcn_readfirstlane((uint32_t)(threadIdx.x & ~63) * 16u);
  const int step_floats = NT * CHUNKS * 4;
  int pos = (blockIdx.x * 4096) % span_floats;
  for (int it = 0; it < iters; ++it) {
    const uint32_t stage = (it & 1) * (step_floats * 4);
#pragma unroll
    for (int p = 0; p < CHUNKS; ++p) {
      const float* g = src + pos + (p * NT + threadIdx.x) * 4;
      const uint32_t l = lds0 + stage + wave_off + (uint32_t)(p * NT * 16);
      asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(g), "s"(l) : "memory", "m0");
    }
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(CHUNKS) : "memory");   // the previous step's DMAs
    pos += step_floats;
    if (pos + step_floats > span_floats) pos = 0;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = smem[5];
}

template <bool WRITE_LDS>
__global__ __launch_bounds__(NT) void reg_kernel(const float* __restrict__ src, int span_floats, int iters, float* out) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int step_floats = NT * CHUNKS * 4;
  int pos = (blockIdx.x * 4096) % span_floats;
  float4 acc = make_float4(0, 0, 0, 0);
  float4 v[CHUNKS];
#pragma unroll
  for (int p = 0; p < CHUNKS; ++p) v[p] = *reinterpret_cast<const float4*>(src + pos + (p * NT + threadIdx.x) * 4);
  for (int it = 0; it < iters; ++it) {
    pos += step_floats;
    if (pos + step_floats > span_floats) pos = 0;
    float4 n[CHUNKS];
#pragma unroll
    for (int p = 0; p < CHUNKS; ++p) n[p] = *reinterpret_cast<const float4*>(src + pos + (p * NT + threadIdx.x) * 4);
    const int stage = (it & 1) * step_floats;
#pragma unroll
    for (int p = 0; p < CHUNKS; ++p) {
      if (WRITE_LDS) *reinterpret_cast<float4*>(smem + stage + (p * NT + threadIdx.x) * 4) = v[p];
      else { acc.x += v[p].x; acc.y += v[p].y; acc.z += v[p].z; acc.w += v[p].w; }
    }
#pragma unroll
    for (int p = 0; p < CHUNKS; ++p) v[p] = n[p];
  }
  __syncthreads();
  if (out && threadIdx.x == 0 && blockIdx.x == 0) out[0] = WRITE_LDS ? smem[5] : acc.x + acc.y + acc.z + acc.w + v[0].x;
}

template <typename K>
void time_it(const char* name, K kern, int wgs, size_t lds, const float* src, int span_floats, int iters, float* out) {
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(kern, dim3(wgs), dim3(NT), lds, 0, src, span_floats, iters, out);
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(kern, dim3(wgs), dim3(NT), lds, 0, src, span_floats, iters, out);
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  CK(hipGetLastError());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= 5;
  const double bytes = (double)wgs * iters * NT * CHUNKS * 16;
  printf("%-44s %4d WGs  %7.1f us  %7.2f TB/s  = %5.1f B/clk/CU at 2.1 GHz (256 CUs)\n", name, wgs, ms * 1e3,
         bytes / ms / 1e9, bytes / (ms * 1e-3) / 256 / 2.1e9);
}

int main(int argc, char** argv) {
  const int span = (argc > 1 ? atoi(argv[1]) : 1024) * 1024 / 4;   // floats
  const int iters = argc > 2 ? atoi(argv[2]) : 400;
  float *src, *out;
  CK(hipMalloc(&src, (size_t)span * 4 + (1 << 20))); CK(hipMalloc(&out, 64));
  CK(hipMemset(src, 0, (size_t)span * 4 + (1 << 20)));
  const size_t lds = 2 * NT * CHUNKS * 16;   // 64 KB: two workgroups per CU
  for (int wgs : {256, 512, 1024}) {
    time_it("A  LDS-DMA global_load_lds_dwordx4", dma_kernel, wgs, lds, src, span, iters, out);
    time_it("B  global_load_dwordx4 + ds_write_b128", reg_kernel<true>, wgs, lds, src, span, iters, out);
    time_it("C  global_load_dwordx4 only", reg_kernel<false>, wgs, lds, src, span, iters, out);
  }
  return 0;
}
