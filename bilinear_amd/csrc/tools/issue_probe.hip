// Developer probe: how many independent VALU fit beside one v_mfma_f32_32x32x16_bf16, at one and
// at two waves per SIMD (cycles per MFMA slot, shader clock).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int NV>
__global__ void probe(unsigned long long* out, float* sink, int iters) {
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  bf16x8 a, b;
  for (int j = 0; j < 8; ++j) { a[j] = (__bf16)(threadIdx.x * 0.001f + j); b[j] = (__bf16)(j * 0.5f); }
  float v[8];
  for (int j = 0; j < 8; ++j) v[j] = threadIdx.x + j;
  __syncthreads();
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[s & 3]) : "v"(a), "v"(b));
#pragma unroll
      for (int n = 0; n < NV; ++n) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[n & 7]) : "v"(v[(n + 3) & 7]));
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int j = 0; j < 8; ++j) s += v[j];
  sink[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) atomicMax(&out[blockIdx.x], c1 - c0);   // slowest wave of the block
}

template <int NV>
void run(int threads) {
  unsigned long long* out; float* sink; const int blocks = 256, iters = 2000;
  CK(hipMalloc(&out, blocks * 8)); CK(hipMalloc(&sink, blocks * threads * 4));
  hipLaunchKernelGGL(probe<NV>, dim3(blocks), dim3(threads), 0, 0, out, sink, iters);
  CK(hipMemset(out, 0, blocks * 8));
  hipLaunchKernelGGL(probe<NV>, dim3(blocks), dim3(threads), 0, 0, out, sink, iters);
  CK(hipDeviceSynchronize());
  unsigned long long h[256]; CK(hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost));
  double c = 0; for (int i = 0; i < blocks; ++i) c += h[i];
  c /= blocks;
  const double per_simd_mfma = c / (iters * 8.0) / (threads / 256.0);
  printf("  %2d VALU/MFMA, %d waves/SIMD: slowest wave %.1f cycles per own MFMA = %.1f cycles per MFMA on the SIMD\n", NV, threads / 256, c / (iters * 8.0), per_simd_mfma);
  CK(hipFree(out)); CK(hipFree(sink));
}

int main() {
  for (int t : {256, 512}) {
    run<0>(t); run<2>(t); run<3>(t); run<4>(t); run<5>(t); run<6>(t); run<8>(t); run<11>(t);
  }
  return 0;
}
