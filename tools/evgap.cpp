// gap a cross-stream event costs the RECORDING stream: main: K K [signal] K K ..., side: wait + small kernel
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <chrono>
__global__ void spin(float* p, int n) {
  float x = p[threadIdx.x];
  for (int i = 0; i < n; ++i) x = x * 1.0001f + 0.5f;
  p[threadIdx.x + blockIdx.x * blockDim.x] = x;
}
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("err %d at %d\n", (int)err_, __LINE__); return 1; } } while (0)
int main() {
  float* d; CK(hipMalloc(&d, 1 << 24));
  hipStream_t m, s; CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int N = 40, iters = 20000;   // kernel ~ tens of us
  for (int mode = 0; mode < 5; ++mode) {
    double best = 1e9;
    for (int rep = 0; rep < 5; ++rep) {
      hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(t0, m));
      for (int i = 0; i < N; ++i) {
        if (mode == 4) {
          hipEvent_t e; CK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
          hipExtLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, m, nullptr, e, 0, d, iters);
          CK(hipStreamWaitEvent(s, e, 0));
          hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d + (1 << 20), 100);
          CK(hipEventDestroy(e));
        } else {
          hipLaunchKernelGGL(spin, dim3(1024), dim3(256), 0, m, d, iters);
          if (mode >= 1) {
            unsigned fl = hipEventDisableTiming;
            if (mode == 2) fl |= hipEventDisableSystemFence;
            if (mode == 3) fl |= hipEventReleaseToDevice;
            hipEvent_t e; CK(hipEventCreateWithFlags(&e, fl));
            CK(hipEventRecord(e, m));
            CK(hipStreamWaitEvent(s, e, 0));
            hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d + (1 << 20), 100);
            CK(hipEventDestroy(e));
          }
        }
      }
      CK(hipEventRecord(t1, m));
      CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, t0, t1));
      if (ms < best) best = ms;
    }
    const char* names[] = {"no event", "event default", "event DisableSystemFence", "event ReleaseToDevice", "hipExtLaunch stopEvent"};
    printf("%-28s %8.2f us per kernel\n", names[mode], best * 1e3 / N);
  }
  return 0;
}
