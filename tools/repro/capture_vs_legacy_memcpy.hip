// Minimal reproducer of round 3's "operation failed due to a previous error during capture" (profiles/r04_threads_root_cause.md):
// thread A captures a few launches on ITS OWN hipStreamNonBlocking stream in hipStreamCaptureModeThreadLocal; thread B does
// nothing but a plain synchronous hipMemcpy of its own buffer.  On this runtime B's copy returns hipErrorStreamCaptureImplicit,
// A's capture is invalidated, hipStreamEndCapture returns hipErrorStreamCaptureInvalidated and leaves the stream in capture
// mode (the next hipStreamSynchronize on it fails too).  The same graph built with hipGraphAddKernelNode is untouched.
//   hipcc --offload-arch=gfx950 -O2 -pthread tools/repro/capture_vs_legacy_memcpy.hip -o /tmp/capture_repro && /tmp/capture_repro
#include <hip/hip_runtime.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>
#include <vector>

__global__ void bump(float* p) { p[threadIdx.x] += 1.0f; }

static std::atomic<int> phase{0};

int main() {
  float *da = nullptr, *db = nullptr;
  hipMalloc(&da, 256), hipMalloc(&db, 1 << 20);
  hipMemset(da, 0, 256);
  hipStream_t st;
  hipStreamCreateWithFlags(&st, hipStreamNonBlocking);
  std::vector<char> host(1 << 20);
  hipError_t copy_result = hipSuccess;
  std::thread b([&] {
    while (phase.load() != 1) std::this_thread::yield();
    copy_result = hipMemcpy(host.data(), db, host.size(), hipMemcpyDeviceToHost);  // another thread, another buffer, legacy stream
    phase.store(2);
  });
  hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal);
  printf("A: hipStreamBeginCapture(ThreadLocal, non-blocking stream) -> %s\n", hipGetErrorName(e));
  hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, st, da);
  phase.store(1);
  while (phase.load() != 2) std::this_thread::yield();
  b.join();
  printf("B: hipMemcpy(D2H) while A captures            -> %s\n", hipGetErrorName(copy_result));
  hipLaunchKernelGGL(bump, dim3(1), dim3(64), 0, st, da);
  printf("A: launch into the capture afterwards          -> %s\n", hipGetErrorName(hipGetLastError()));
  hipGraph_t g = nullptr;
  e = hipStreamEndCapture(st, &g);
  printf("A: hipStreamEndCapture                         -> %s (graph %p)\n", hipGetErrorName(e), (void*)g);
  e = hipStreamSynchronize(st);
  printf("A: hipStreamSynchronize on the same stream     -> %s\n", hipGetErrorName(e));
  (void)hipGetLastError();

  // the same two launches as an explicitly built graph, with B copying in the middle of the construction
  hipStream_t st2;
  hipStreamCreateWithFlags(&st2, hipStreamNonBlocking);
  hipGraph_t g2;
  hipGraphCreate(&g2, 0);
  void* params[] = {&da};
  hipKernelNodeParams np = {};
  np.func = (void*)bump, np.gridDim = dim3(1), np.blockDim = dim3(64), np.kernelParams = params;
  hipGraphNode_t n1, n2;
  e = hipGraphAddKernelNode(&n1, g2, nullptr, 0, &np);
  hipError_t c2 = hipMemcpy(host.data(), db, host.size(), hipMemcpyDeviceToHost);
  hipError_t e2 = hipGraphAddKernelNode(&n2, g2, &n1, 1, &np);
  hipGraphExec_t x;
  hipError_t e3 = hipGraphInstantiate(&x, g2, nullptr, nullptr, 0);
  hipError_t e4 = hipGraphLaunch(x, st2);
  hipError_t e5 = hipStreamSynchronize(st2);
  float out[64];
  hipMemcpy(out, da, sizeof(out), hipMemcpyDeviceToHost);
  printf("explicit graph: add %s, memcpy in between %s, add %s, instantiate %s, launch %s, sync %s, value %g (expected 2)\n",
         hipGetErrorName(e), hipGetErrorName(c2), hipGetErrorName(e2), hipGetErrorName(e3), hipGetErrorName(e4), hipGetErrorName(e5), out[0]);
  return 0;
}
