// Developer tool: device ceiling of v_mfma_f32_32x32x2_f32 (register operands, no memory),
// and the in-kernel clock (s_memtime / s_memrealtime) under that load.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int LDS_READS>
__global__ __launch_bounds__(256) void peak(float* out, int iters, unsigned long long* clk) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = (float)(i % 7) * 0.25f - 0.5f;
  __syncthreads();
  f32x16 acc[4];
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  float a0 = 0.5f + lane * 0.01f, a1 = -0.25f + lane * 0.02f, b0 = 0.75f - lane * 0.01f, b1 = 0.3f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if (LDS_READS) {
      const float4 v = *reinterpret_cast<const float4*>(&lds[((it * 64 + lane) * 4) & 8188]);
      const float4 u = *reinterpret_cast<const float4*>(&lds[((it * 64 + lane) * 4 + 2048) & 8188]);
      a0 = v.x; a1 = v.y; b0 = u.x; b1 = u.y;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
      acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
      acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) s += acc[a][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

// random operands held in registers (16 per lane for A and B), rotated every MFMA group:
// the ceiling with realistic data toggling (power), still without any memory traffic
__global__ __launch_bounds__(256) void peak_rand(float* out, const float* rnd, int iters, unsigned long long* clk) {
  const int gid = blockIdx.x * blockDim.x + threadIdx.x;
  float a[16], b[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { a[i] = rnd[(gid * 16 + i) & 0xFFFFF]; b[i] = rnd[(gid * 16 + i + 7777) & 0xFFFFF]; }
  f32x16 acc[4];
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) acc[q][r] = 0.f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; it += 4) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u * 4 + j], b[u * 4 + j], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u * 4 + j], b[(u * 4 + j + 5) & 15], acc[1], 0, 0, 0);
        acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u * 4 + j + 9) & 15], b[u * 4 + j], acc[2], 0, 0, 0);
        acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[(u * 4 + j + 9) & 15], b[(u * 4 + j + 5) & 15], acc[3], 0, 0, 0);
      }
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int q = 0; q < 4; ++q) for (int r = 0; r < 16; ++r) s += acc[q][r];
  out[gid] = s;
  if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

void go_rand(const char* name, int blocks, int iters, int launches) {
  float* out; float* rnd; unsigned long long* clk;
  CK(hipMalloc(&out, blocks * 256 * 4)); CK(hipMalloc(&clk, blocks * 16)); CK(hipMalloc(&rnd, (1 << 20) * 4));
  float* h = (float*)malloc((1 << 20) * 4);
  for (int i = 0; i < (1 << 20); ++i) { float u = 0; for (int k = 0; k < 12; ++k) u += (float)rand() / RAND_MAX; h[i] = u - 6.f; }
  CK(hipMemcpy(rnd, h, (1 << 20) * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(peak_rand, dim3(blocks), dim3(256), 0, 0, out, rnd, iters, clk);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < launches; ++i) hipLaunchKernelGGL(peak_rand, dim3(blocks), dim3(256), 0, 0, out, rnd, iters, clk);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= launches;
  unsigned long long hc[2]; CK(hipMemcpy(hc, clk, 16, hipMemcpyDeviceToHost));
  double flop = (double)blocks * 4 * iters * 16 * 4096.0;
  printf("%-28s blocks %4d: %8.1f us  %6.1f TF   in-kernel clock %.3f GHz\n", name, blocks, ms * 1e3, flop / ms / 1e9,
         (double)hc[0] / (double)hc[1] * 0.1);
}

template <int L>
void go(const char* name, int blocks, int iters) {
  float* out; unsigned long long* clk;
  CK(hipMalloc(&out, blocks * 256 * 4)); CK(hipMalloc(&clk, blocks * 16));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(peak<L>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(peak<L>, dim3(blocks), dim3(256), 0, 0, out, iters, clk);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  unsigned long long h[2]; CK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
  double flop = (double)blocks * 4 * iters * 16 * 4096.0;
  printf("%-28s blocks %4d: %8.1f us  %6.1f TF   in-kernel clock %.3f GHz\n", name, blocks, ms * 1e3, flop / ms / 1e9,
         (double)h[0] / (double)h[1] * 0.1);
  CK(hipFree(out)); CK(hipFree(clk));
}

int main() {
  go<0>("mfma only, 1 wave/SIMD", 256, 2000);
  go<0>("mfma only, 2 waves/SIMD", 512, 2000);
  go<1>("mfma + 2 ds_read_b128/16", 256, 2000);
  go<1>("mfma + lds, 2 waves/SIMD", 512, 2000);
  go_rand("mfma random regs, 1 w/SIMD", 256, 2000, 5);
  go_rand("mfma random regs, 2 w/SIMD", 512, 2000, 5);
  go_rand("mfma random regs, sustained", 512, 2000, 400);
  return 0;
}
