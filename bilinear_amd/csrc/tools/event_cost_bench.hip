// What does it cost stream A to tell stream B "kernel K1 is done" between two of its own kernels?
// (the fork of the two-stream backward, step_f32.hip: one per stage; profiles/r05_fork_cost.md)
//
//   mode 0  A: K1 K2                                              (no signal: the floor)
//   mode 1  A: K1 record(ev) K2                                   (nobody waits)
//   mode 2  A: K1 record(ev) K2          B: wait(ev) Kb           (the shipped fork)
//   mode 3  A: K1+stop event (hipExtLaunchKernelGGL) K2   B: wait(ev) Kb
//   mode 4  A: K1 (its last workgroup stores a flag) K2   B: hipStreamWaitValue32(flag) Kb
//   mode 5  A: K1 hipStreamWriteValue32(flag) K2          B: hipStreamWaitValue32(flag) Kb
//   mode 6  A: K1 K2                                      B: Kb   (no dependency at all: what two busy queues cost A)
//
// Every kernel spins for a fixed time on s_memrealtime (100 MHz); Kb checks that K1's data arrived.
// usage: event_cost_bench [us_per_kernel=10] [reps=200]
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); std::exit(1); } } while (0)

__device__ __forceinline__ void spin_ticks(long ticks) {
  const long t0 = (long)__builtin_amdgcn_s_memrealtime();
  while ((long)__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(2);
}

// K1 / K2: spin; K1 also leaves `seq` in data[] (what Kb checks) and, with flag != nullptr, its last workgroup
// publishes `seq` in *flag (system scope: the command processor of another queue polls it)
__global__ void busy_kernel(long ticks, unsigned* data, unsigned seq, unsigned* ticket, unsigned* flag) {
  spin_ticks(ticks);
  if (threadIdx.x == 0) {
    if (data) __hip_atomic_store(&data[blockIdx.x], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (flag) {
      __threadfence();
      const unsigned t = atomicAdd(ticket, 1u);
      if (t == gridDim.x * seq - 1u) __hip_atomic_store(flag, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

__global__ void check_kernel(long ticks, const unsigned* data, int n, unsigned seq, unsigned* errors) {
  if (threadIdx.x == 0 && (int)blockIdx.x < n) {
    const unsigned v = __hip_atomic_load(&data[blockIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (v < seq) atomicAdd(errors, 1u);
  }
  spin_ticks(ticks);
}

int main(int argc, char** argv) {
  const double us = argc > 1 ? std::atof(argv[1]) : 10.0;
  const int reps = argc > 2 ? std::atoi(argv[2]) : 200;
  const long ticks = (long)(us * 100.0);
  const int G = 256, T = 256, NEV = 32;
  int can = 0;
  CK(hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, 0));
  std::printf("kernel %.1f us, %d pairs per mode; hipDeviceAttributeCanUseStreamWaitValue = %d\n", us, reps, can);
  int lo = 0, hi = 0;
  CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  hipStream_t A, B, R;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking));
  CK(hipStreamCreateWithPriority(&B, hipStreamNonBlocking, lo));
  CK(hipStreamCreateWithFlags(&R, hipStreamNonBlocking));
  unsigned *data, *ticket, *errors, *flag = nullptr;
  CK(hipMalloc(&data, G * sizeof(unsigned)));
  CK(hipMalloc(&ticket, sizeof(unsigned)));
  CK(hipMalloc(&errors, sizeof(unsigned)));
  if (can) CK(hipExtMallocWithFlags((void**)&flag, 8, hipMallocSignalMemory));
  std::vector<hipEvent_t> ev(NEV);
  for (auto& e : ev) CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  hipEvent_t t0, t1, joinB;
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  CK(hipEventCreateWithFlags(&joinB, hipEventDisableTiming));

  const char* names[7] = {"no signal", "record, nobody waits", "record + wait on B (shipped)", "stop event of K1 (hipExtLaunchKernelGGL) + wait on B",
                          "flag stored by K1's last workgroup + hipStreamWaitValue32 on B", "hipStreamWriteValue32 on A + hipStreamWaitValue32 on B",
                          "no signal, an independent kernel per pair on B"};
  double base = 0;
  for (int mode = 0; mode < 7; ++mode) {
    if ((mode == 4 || mode == 5) && !can) { std::printf("mode %d: stream wait-value not supported\n", mode); continue; }
    for (int pass = 0; pass < 2; ++pass) {   // pass 0 warms up
      CK(hipMemsetAsync(data, 0, G * sizeof(unsigned), A));
      CK(hipMemsetAsync(ticket, 0, sizeof(unsigned), A));
      CK(hipMemsetAsync(errors, 0, sizeof(unsigned), A));
      if (flag) CK(hipStreamWriteValue32(A, flag, 0, 0));
      CK(hipStreamSynchronize(A));
      CK(hipEventRecord(t0, A));
      for (int i = 0; i < reps; ++i) {
        const unsigned seq = (unsigned)i + 1u;
        hipEvent_t e = ev[i % NEV];
        if (mode == 3) {
          hipExtLaunchKernelGGL(busy_kernel, dim3(G), dim3(T), 0, A, nullptr, e, 0, ticks, data, seq, ticket, (unsigned*)nullptr);
        } else {
          hipLaunchKernelGGL(busy_kernel, dim3(G), dim3(T), 0, A, ticks, data, seq, ticket, mode == 4 ? flag : (unsigned*)nullptr);
        }
        if (mode == 1 || mode == 2) CK(hipEventRecord(e, A));
        if (mode == 5) CK(hipStreamWriteValue32(A, flag, seq, 0));
        hipLaunchKernelGGL(busy_kernel, dim3(G), dim3(T), 0, A, ticks, (unsigned*)nullptr, seq, ticket, (unsigned*)nullptr);
        if (mode == 2 || mode == 3) CK(hipStreamWaitEvent(B, e, 0));
        if (mode == 4 || mode == 5) CK(hipStreamWaitValue32(B, flag, seq, hipStreamWaitValueGte, 0xFFFFFFFFu));
        if (mode >= 2) hipLaunchKernelGGL(check_kernel, dim3(G), dim3(T), 0, B, ticks, data, mode == 6 ? 0 : G, seq, errors);
      }
      CK(hipEventRecord(t1, A));
      // B must never be left waiting for a value that does not come: poll, and release it by hand after 5 s
      bool released = false;
      const auto start = std::chrono::steady_clock::now();
      while (hipStreamQuery(B) == hipErrorNotReady || hipStreamQuery(A) == hipErrorNotReady) {
        std::this_thread::sleep_for(std::chrono::milliseconds(1));
        if (!released && flag && std::chrono::steady_clock::now() - start > std::chrono::seconds(5)) {
          CK(hipStreamWriteValue32(R, flag, 0xFFFFFFFFu, 0));
          released = true;
        }
        if (std::chrono::steady_clock::now() - start > std::chrono::seconds(20)) { std::printf("mode %d: stuck\n", mode); std::exit(2); }
      }
      CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
      float ms = 0;
      CK(hipEventElapsedTime(&ms, t0, t1));
      unsigned err = 0;
      CK(hipMemcpy(&err, errors, sizeof(err), hipMemcpyDeviceToHost));
      if (pass == 1) {
        const double per = ms * 1000.0 / reps;
        if (mode == 0) base = per;
        std::printf("mode %d  %-72s A: %7.2f us per pair (+%5.2f)  order errors %u%s\n", mode, names[mode], per, per - base, err,
                    released ? "  [B HAD TO BE RELEASED BY HAND]" : "");
      }
    }
  }
  return 0;
}
