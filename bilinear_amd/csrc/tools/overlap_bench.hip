// Developer tool (not part of the library): how long do the HBM-bound BatchNorm-backward kernels
// take while a weight-gradient GEMM owns the matrix pipes of every CU?  Links the library's objects:
//   hipcc -O3 --offload-arch=gfx950 tools/overlap_bench.hip ../lib/obj/{gemm_f32,gemm_bf16s,elementwise,skinny}.o -o ../lib/overlap_bench
// Probes on stream B while stream A runs wgrad GEMMs back to back:
//   copy   : hipMemcpyAsync D2D of one [B,W] tensor (pure memory traffic)
//   apply/m: bn_bwd_apply with an explicit keep mask (no Philox arithmetic)
//   apply/p: bn_bwd_apply with Philox regeneration
//   reduce : bn_bwd_reduce (Philox)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../common.h"
#include "../gemm_f32_ring.h"

using namespace blh;
thread_local int blh::g_last_hip_error = 0;
thread_local hipEvent_t blh::tl_stop_event = nullptr;

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 4096, W = argc > 2 ? atoi(argv[2]) : 1024;
  const int reps = 40;
  const size_t act = (size_t)M * W;
  float *dA, *Z, *dZ, *A, *vec, *part, *colsum, *slabs, *copy_dst;
  uint8_t* keep;
  CK(hipMalloc(&dA, act * 4)); CK(hipMalloc(&Z, act * 4)); CK(hipMalloc(&dZ, act * 4)); CK(hipMalloc(&A, act * 4));
  CK(hipMalloc(&copy_dst, act * 4));
  CK(hipMalloc(&vec, 8 * W * 4)); CK(hipMalloc(&part, (size_t)1024 * 2 * W * 4)); CK(hipMalloc(&colsum, (size_t)1024 * W * 4));
  CK(hipMalloc(&slabs, (size_t)16 * W * W * 4)); CK(hipMalloc(&keep, act));
  std::vector<float> h(act);
  for (size_t i = 0; i < act; ++i) h[i] = (float)((double)rand() / RAND_MAX - 0.5);
  CK(hipMemcpy(dA, h.data(), act * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(Z, h.data(), act * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(A, h.data(), act * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dZ, h.data(), act * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(vec, h.data(), 8 * W * 4, hipMemcpyHostToDevice));
  CK(hipMemset(keep, 1, act));
  hipStream_t sa, sb;
  CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));

  GemmParams w{};   // wgrad: dW = dZ^T A, 4 slabs over the batch
  w.A = dZ; w.lda = W; w.B = A; w.ldb = W; w.C = slabs; w.ldc = W; w.M = W; w.N = W; w.K = M;
  w.k_per_split = (int)round_up(ceil_div(M, 4), 64); w.c_split_stride = (int64_t)W * W;
  GemmParams d{};   // dgrad: dA = dZ W
  d.A = dZ; d.lda = W; d.B = A; d.ldb = W; d.C = copy_dst; d.ldc = W; d.M = M; d.N = W; d.K = W; d.k_per_split = W;

  DropoutSrc philox{}; philox.keep = nullptr; philox.seed = 7; philox.step = 3; philox.layer = 2;
  DropoutSrc mask = philox; mask.keep = keep; (void)mask;
  uint32_t* keepbits; CK(hipMalloc(&keepbits, bn_keepbits_words_f32(M, W) * 4)); CK(hipMemset(keepbits, 0x5a, bn_keepbits_words_f32(M, W) * 4));
  const float *scale = vec, *shift = vec + W, *mean = vec + 2 * W, *invstd = vec + 3 * W, *gamma = vec + 4 * W,
              *dgamma = vec + 5 * W, *dbeta = vec + 6 * W;

  auto probe = [&](int which) {
    switch (which) {
      case 0: CK(hipMemcpyAsync(copy_dst, dA, act * 4, hipMemcpyDeviceToDevice, sb)); break;
      case 1: launch_bn_bwd_apply_f2(sb, dA, Z, scale, shift, mean, invstd, dgamma, dbeta, keepbits, copy_dst, colsum, M, W, M); break;
      case 2: launch_bn_bwd_apply_f2(sb, dA, Z, scale, shift, mean, invstd, dgamma, dbeta, keepbits, copy_dst, colsum, M, W, M); break;
      case 3: launch_bn_bwd_reduce_f2(sb, dA, Z, scale, shift, keepbits, part, M, W); break;
      case 4: launch_bn_apply_f2(sb, true, Z, scale, shift, nullptr, nullptr, nullptr, nullptr, nullptr, copy_dst, keepbits, M, W, philox, nullptr); break;
    }
  };
  const char* names[] = {"copy [B,W] D2D", "bn_bwd_apply (mask)", "bn_bwd_apply (Philox)", "bn_bwd_reduce (Philox)", "bn_apply train (Philox)"};
  GemmParams dh = d; dh.M = M / 2;            // half the tiles: half of the CUs stay free of GEMM workgroups
  GemmParams f = d; f.B = A;                   // forward layout (ROWK, ROWK): bk64 x 2 stages, 128 KB LDS
  auto gemm = [&](int kind) {
    if (kind == 1) launch_gemm(sa, TILE_128x128, KROW, KROW, EPI_STORE, w, 4, 0);
    if (kind == 2) launch_gemm(sa, TILE_128x128, ROWK, KROW, EPI_STORE, d, 1, 0);
    if (kind == 3) launch_gemm(sa, TILE_128x128, ROWK, KROW, EPI_STORE, dh, 1, 0);
    if (kind == 4) launch_gemm(sa, TILE_128x128, ROWK, ROWK, EPI_STORE, f, 1, 0);
  };
  // ablated interferers (dgrad shape): which half of the GEMM hurts a co-resident streaming kernel?
  unsigned long long* stamps; CK(hipMalloc(&stamps, 4096 * 64));
  GemmParams ds = d; ds.loss_part = reinterpret_cast<float*>(stamps);
  auto abl = [&](auto kern) {
    constexpr size_t lds = gemm_ring_lds_bytes<128, 128, 32, 2>();
    static bool once = false;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    (void)once;
    hipLaunchKernelGGL(kern, dim3((M / 128) * (W / 128)), dim3(512), lds, sa, ds);
  };
  auto gemm_full = gemm;
  auto gemm2 = [&](int kind) {
    if (kind <= 4) gemm_full(kind);
    if (kind == 5) abl(gemm_f32_ring_kernel<128, 128, 4, 2, ROWK, KROW, EPI_STORE, 32, 2, 6>);   // MFMAs only
    if (kind == 6) abl(gemm_f32_ring_kernel<128, 128, 4, 2, ROWK, KROW, EPI_STORE, 32, 2, 7>);   // no MFMAs
    if (kind == 7) abl(gemm_f32_ring_kernel<128, 128, 4, 2, ROWK, KROW, EPI_STORE, 32, 2, 3>);   // no DMA
  };
  const char* gnames[] = {"alone", "beside wgrad", "beside dgrad", "beside dgrad M/2", "beside fwd bk64",
                          "beside MFMA-only", "beside no-MFMA", "beside no-DMA"};
  hipEvent_t e0, e1, ea0, ea1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1)); CK(hipEventCreate(&ea0)); CK(hipEventCreate(&ea1));
  for (int round = 0; round < 2; ++round)
    for (int kind = 0; kind < 8; ++kind)
      for (int which = 0; which < 5; ++which) {
        if (kind >= 5 && which != 0 && which != 2) continue;
        CK(hipDeviceSynchronize());
        // stream A: enough GEMMs to outlast the probes (70 us each); stream B: the probes
        const int ngemm = kind ? reps : 0;
        CK(hipEventRecord(ea0, sa));
        for (int i = 0; i < ngemm; ++i) gemm2(kind);
        CK(hipEventRecord(ea1, sa));
        CK(hipEventRecord(e0, sb));
        for (int i = 0; i < reps; ++i) probe(which);
        CK(hipEventRecord(e1, sb));
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        float tb = 0, ta = 0;
        CK(hipEventElapsedTime(&tb, e0, e1)); CK(hipEventElapsedTime(&ta, ea0, ea1));
        if (round == 1)
          printf("%-26s %-14s probe %7.1f us each | GEMM stream %7.1f us per GEMM (%d GEMMs, probes cover %.0f%% of it)\n",
                 names[which], gnames[kind], tb / reps * 1e3, ngemm ? ta / ngemm * 1e3 : 0.f, ngemm,
                 ngemm ? 100.0 * tb / ta : 0.0);
      }
  return 0;
}
