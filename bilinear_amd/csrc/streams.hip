// The library's process-wide side stream (one per device) and the stream-pair tuner.
//
// The backward runs its weight-gradient GEMMs on a second stream next to the caller's compute stream.
// On this stack (ROCm 7.0 HIP under PyTorch) whether the two streams' kernels really run beside each
// other depends on which hardware queues the runtime gave the two streams — an accident of how many
// streams the process had created before each of them (profiles/r04_dp_setup_order.md): with an
// unlucky pair every kernel that has work on the other stream beside it runs 1.5-5x longer (the
// fused step 2.09 instead of 1.03 ms; single-stream work is unaffected), and creating an RCCL
// communicator between the two streams is one way to get such a pair (round 3's "model before the
// process group").  Nothing in the API tells a good pair from a bad one, so the library measures:
// blh_tune_streams() times a short two-stream probe for the current side stream and, when that pair
// is a bad one, for freshly created candidates, and keeps the first good one (the others are destroyed
// again).  Measured (tools_dev/dp_order_probe.py tensors_first_tune): the probe's short kernels take
// 1.58-1.76x their solo time beside a busy side stream in a good pair, 3.8-3.9x in a bad one.
#include <algorithm>
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "common.h"

namespace blh {
namespace {
constexpr int kMaxDevices = 64;
std::mutex g_side_mu;
hipStream_t g_side_stream[kMaxDevices] = {};
int g_side_refs[kMaxDevices] = {};
std::atomic<int> g_side_generation{0};

hipError_t side_stream_create(hipStream_t* out) {
  // LOWEST priority: a level nothing else in a PyTorch + RCCL process uses, so the stream never shares
  // a hardware queue with torch's default stream, its pool streams or the collective stream (HIP maps
  // the streams of one priority level onto a small pool of hardware queues; streams that share one run
  // in submission order: profiles/r03_dp_overhead.md)
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  const char* pr = getenv("BLH_SIDE_PRIORITY");   // "normal": the round-2 behaviour (A/B knob)
  return (pr && !strcmp(pr, "normal")) ? hipStreamCreateWithFlags(out, hipStreamNonBlocking)
                                       : hipStreamCreateWithPriority(out, hipStreamNonBlocking, least);
}

// ---- probe: how long do short kernels on `main` take while `side` is busy? --------------------------
__global__ __launch_bounds__(256) void stream_probe_kernel(float* __restrict__ buf, int n, int iters) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float v = buf[i];
    for (int k = 0; k < iters; ++k) v = fmaf(v, 1.0000001f, 1e-9f);
    buf[i] = v;
  }
}

struct Probe {
  float* a = nullptr;     // main-stream buffer
  float* b = nullptr;     // side-stream buffer
  hipEvent_t e0 = nullptr, e1 = nullptr, go = nullptr, done = nullptr;
  static constexpr int N = 4 << 20;       // 16 MB per buffer
  int init() {
    BLH_HIP_TRY(hipMalloc(&a, (size_t)N * 4));
    BLH_HIP_TRY(hipMalloc(&b, (size_t)N * 4));
    BLH_HIP_TRY(hipMemset(a, 0, (size_t)N * 4));
    BLH_HIP_TRY(hipMemset(b, 0, (size_t)N * 4));
    BLH_HIP_TRY(hipEventCreate(&e0));
    BLH_HIP_TRY(hipEventCreate(&e1));
    BLH_HIP_TRY(hipEventCreateWithFlags(&go, hipEventDisableTiming));
    BLH_HIP_TRY(hipEventCreateWithFlags(&done, hipEventDisableTiming));
    return BLH_OK;
  }
  ~Probe() {
    if (a) (void)hipFree(a);
    if (b) (void)hipFree(b);
    for (hipEvent_t e : {e0, e1, go, done})
      if (e) (void)hipEventDestroy(e);
  }
  // ms for 12 short kernels on `main`; with `side` != null, 6 longer kernels run on it meanwhile
  int run(hipStream_t main, hipStream_t side, float* ms) {
    BLH_HIP_TRY(hipEventRecord(go, main));
    if (side) {
      BLH_HIP_TRY(hipStreamWaitEvent(side, go, 0));
      for (int i = 0; i < 6; ++i) hipLaunchKernelGGL(stream_probe_kernel, dim3(2048), dim3(256), 0, side, b, N, 48);
      BLH_HIP_TRY(hipEventRecord(done, side));
    }
    BLH_HIP_TRY(hipEventRecord(e0, main));
    for (int i = 0; i < 12; ++i) hipLaunchKernelGGL(stream_probe_kernel, dim3(2048), dim3(256), 0, main, a, N, 4);
    BLH_HIP_TRY(hipEventRecord(e1, main));
    if (side) BLH_HIP_TRY(hipStreamWaitEvent(main, done, 0));
    BLH_HIP_TRY(hipEventSynchronize(e1));
    if (side) BLH_HIP_TRY(hipStreamSynchronize(side));
    BLH_HIP_TRY(hipGetLastError());
    BLH_HIP_TRY(hipEventElapsedTime(ms, e0, e1));
    return BLH_OK;
  }
  // best of three (after one warm-up run)
  int measure(hipStream_t main, hipStream_t side, float* ms) {
    float v = 0.f, best = 1e30f;
    BLH_TRY(run(main, side, &v));
    for (int r = 0; r < 3; ++r) {
      BLH_TRY(run(main, side, &v));
      best = std::min(best, v);
    }
    *ms = best;
    return BLH_OK;
  }
};
}  // namespace

hipError_t side_stream_acquire(int device, hipStream_t* out) {
  if (device < 0 || device >= kMaxDevices) return hipErrorInvalidDevice;
  std::lock_guard<std::mutex> lk(g_side_mu);
  if (g_side_refs[device] == 0) {
    const hipError_t e = side_stream_create(&g_side_stream[device]);
    if (e != hipSuccess) return e;
    g_side_generation.fetch_add(1);
  }
  ++g_side_refs[device];
  *out = g_side_stream[device];
  return hipSuccess;
}

void side_stream_release(int device) {
  std::lock_guard<std::mutex> lk(g_side_mu);
  if (device < 0 || device >= kMaxDevices || g_side_refs[device] == 0) return;
  if (--g_side_refs[device] == 0) {
    (void)hipStreamDestroy(g_side_stream[device]);
    g_side_stream[device] = nullptr;
  }
}

hipStream_t side_stream_current(int device) {
  if (device < 0 || device >= kMaxDevices) return nullptr;
  std::lock_guard<std::mutex> lk(g_side_mu);
  return g_side_stream[device];
}

// Replace the device's side stream (every context of the device picks the new one up at its next call).
// The old stream is drained and destroyed: no library call may be in flight on it.
static hipError_t side_stream_replace(int device, hipStream_t fresh) {
  hipStream_t old = g_side_stream[device];
  g_side_stream[device] = fresh;
  g_side_generation.fetch_add(1);
  if (!old) return hipSuccess;
  const hipError_t e = hipStreamSynchronize(old);
  if (e != hipSuccess) return e;
  return hipStreamDestroy(old);
}

}  // namespace blh

using namespace blh;

extern "C" {

int blh_side_stream_renew(void) {
  int dev = -1;
  BLH_HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return BLH_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(g_side_mu);
  if (g_side_refs[dev] == 0) return BLH_OK;        // nothing to replace: the next context creates one
  hipStream_t fresh = nullptr;
  BLH_HIP_TRY(side_stream_create(&fresh));
  BLH_HIP_TRY(side_stream_replace(dev, fresh));
  return BLH_OK;
}

int32_t blh_side_stream_generation(void) { return g_side_generation.load(); }

// How often a device's side stream may be replaced by the tuner in one process: a caller that alternates two
// compute streams (the default stream for the fused step, DataParallel.stream for the data-parallel one) could
// otherwise make each tuning undo the other's choice for ever.
static constexpr int kMaxTunerReplacements = 4;
static int g_tuner_replacements[kMaxDevices] = {};

// The probe proper.  Every failure inside it (no memory for the probe buffers beside a full caching allocator, a
// stream that cannot be created, ...) leaves the pair as it was: tuning is an optimisation and never fails a step.
static void tune_locked(hipStream_t main, int dev, int max_candidates, float* report) {
  hipStream_t cand[8] = {};
  int tried = 0;
  hipStream_t best_stream = nullptr;                // nullptr: keep the current one
  float alone = 0.f, cur = 0.f, best = 0.f;
  bool ok = false;
  {
    Probe pr;
    do {
      if (pr.init() != BLH_OK) break;
      if (pr.measure(main, nullptr, &alone) != BLH_OK) break;
      if (pr.measure(main, g_side_stream[dev], &cur) != BLH_OK) break;
      best = cur;
      ok = true;
      // (a pair is good when the short kernels take at most kGood x their solo time beside the busy side
      //  stream — 1.58-1.76x measured for good pairs, 3.8-3.9x for bad ones; candidates are created only
      //  when the current pair is bad, and the search stops at the first good one)
      constexpr float kGood = 2.5f;
      if (cur <= kGood * alone || g_tuner_replacements[dev] >= kMaxTunerReplacements) break;
      while (tried < max_candidates) {
        if (side_stream_create(&cand[tried]) != hipSuccess) break;
        float v = 0.f;
        const int rc = pr.measure(main, cand[tried], &v);
        ++tried;
        if (rc != BLH_OK) break;
        if (v < best) { best = v; best_stream = cand[tried - 1]; }
        if (v <= kGood * alone) break;
      }
    } while (false);
  }
  (void)hipGetLastError();                          // (a failed probe must not poison the caller's next check)
  for (int i = 0; i < 8; ++i)
    if (cand[i] && cand[i] != best_stream) (void)hipStreamDestroy(cand[i]);
  if (best_stream) {
    if (side_stream_replace(dev, best_stream) == hipSuccess) ++g_tuner_replacements[dev];
  }
  if (report) {
    report[0] = ok ? alone : 0.f; report[1] = ok ? cur : 0.f; report[2] = ok ? best : 0.f;
    report[3] = ok ? (float)tried : -1.f;
  }
}

int blh_tune_streams(void* stream, int32_t max_candidates, float* report) {
  hipStream_t main = (hipStream_t)stream;
  int dev = -1;
  BLH_HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices || max_candidates < 0 || max_candidates > 7) return BLH_ERR_INVALID_ARGUMENT;
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(main, &cap);
  if (cap != hipStreamCaptureStatusNone) return BLH_ERR_INVALID_ARGUMENT;
  std::lock_guard<std::mutex> lk(g_side_mu);
  if (g_side_refs[dev] == 0) return BLH_ERR_INVALID_ARGUMENT;     // no context on this device yet
  tune_locked(main, dev, max_candidates, report);
  return BLH_OK;
}

}  // extern "C"
