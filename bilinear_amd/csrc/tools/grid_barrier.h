// Grid-wide barriers for launches whose workgroups are all resident (at most one per CU; the hosts that use them
// check the grid against the device), on words in device memory zeroed once when the context is created.  A spin
// is bounded — a launch whose workgroups are not all resident must not hang the device: its results are then
// wrong, the timeout word says why and blh_context_grid_barrier_timeouts() reports it.  (The one-barrier-per-
// launch sense-reversal form of the bf16 fused forward stage is grid_barrier_once, gemm_bf16s_bnfwd.h.)
//
// Visibility across the 8 XCDs (one L2 each): the arriving side releases at agent scope (writes the dirty L2
// lines back), the leaving side acquires at agent scope (drops its stale L1 / L2 lines).  The fences are issued by
// one thread; they act on the caches, so the caller must have drained every wave's stores first
// (s_waitcnt vmcnt(0) + workgroup barrier), which arrive() does.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace blh {

// GridBarrier is the many-barriers-per-launch form: ONE monotonic arrival counter (word 0 of its three
// words), never reset — barrier k of a launch is complete when the counter has reached base + (k + 1) * nwg, where
// base (word 1) is what the previous launch on the stream left behind; workgroup 0 stores the new base when it
// leaves the last barrier.  Arriving is one fire-and-forget atomic add behind the release; there is no last-arriver
// work and no generation word, so the dependent chain of a barrier is: release, add, the poll that sees the count,
// acquire.  Comparisons are on the signed difference (the counter may wrap).
struct GridBarrier {
  uint32_t* bar;       // [0] arrivals (monotonic), [1] base of the current launch, [2] timeouts
  uint32_t nwg;
  uint32_t target;     // count at which the next barrier is complete (thread 0 only)

  // once per launch, before the first arrive (thread 0 reads the base; the load is not waited for here)
  __device__ __forceinline__ void init() {
    if (threadIdx.x == 0) target = __hip_atomic_load(&bar[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + nwg;
  }
  // every thread of the workgroup calls arrive(); work that does not depend on other workgroups may follow
  __device__ __forceinline__ void arrive() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      (void)__hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // arrive() for workgroups whose data for the others went out with write-through stores at agent scope
  // (__hip_atomic_store relaxed / agent = global_store sc1): the stores only have to have completed — no L2
  // write-back.  Ordinary stores made so far are NOT visible to other XCDs behind this barrier.
  __device__ __forceinline__ void arrive_published() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) (void)__hip_atomic_fetch_add(&bar[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  // ... and wait() before the first read of another workgroup's data
  __device__ __forceinline__ void wait() {
    if (threadIdx.x == 0) {
      int spins = 0;
      while ((int32_t)(__hip_atomic_load(&bar[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) < 0) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > (1 << 22)) {      // ~0.3 s: give up instead of hanging
          __hip_atomic_fetch_add(&bar[2], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      target += nwg;
    }
    __syncthreads();
  }
  __device__ __forceinline__ void sync() { arrive(); wait(); }
  // after the last wait() of the launch, by ONE workgroup: the next launch's base
  __device__ __forceinline__ void finish() {
    if (threadIdx.x == 0) __hip_atomic_store(&bar[1], target - nwg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
};

}  // namespace blh
