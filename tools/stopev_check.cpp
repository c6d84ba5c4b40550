// Does a consumer on a second stream that waits on a kernel's hipExtLaunchKernelGGL stop event see that kernel's stores?
// main: producer (spins, then writes `tag` to buf[0..n)), launched with the event as its stop event; a second producer behind it
// overwrites buf with tag + 1000 (what a consumer that waited for too much - or too little, after the fact - would see mixed).
// side: waits on the event, copies buf to out.  The check: every out[i] is `tag` or `tag + 1000`, never the previous round's value
// (= the consumer started before the first producer finished).  Build: hipcc --offload-arch=gfx950 -O2 tools/stopev_check.cpp -o build/stopev_check
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdio.h>
#include <vector>
__global__ void produce(float* buf, int n, float tag, int spin) {
  float x = tag;
  for (int i = 0; i < spin; ++i) x = x * 1.0000001f + 0.0f;
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) buf[i] = (x == x) ? tag : 0.f;
}
__global__ void consume(const float* buf, float* out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = buf[i];
}
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("err %d at %d\n", (int)err_, __LINE__); return 1; } } while (0)
int main() {
  const int n = 1 << 22;
  float *buf, *out; CK(hipMalloc(&buf, n * 4)); CK(hipMalloc(&out, n * 4));
  CK(hipMemset(buf, 0, n * 4));
  hipStream_t m, s; CK(hipStreamCreateWithFlags(&m, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  std::vector<float> h(n);
  int bad_early = 0, saw_first = 0, saw_second = 0;
  for (unsigned flags : {unsigned(hipEventDisableTiming), unsigned(hipEventDisableTiming | hipEventDisableSystemFence)}) {
    for (int round = 1; round <= 40; ++round) {
      const float tag = static_cast<float>(round);
      hipEvent_t e; CK(hipEventCreateWithFlags(&e, flags));
      hipExtLaunchKernelGGL(produce, dim3(n / 256), dim3(256), 0, m, nullptr, e, 0, buf, n, tag, 20000 + 500 * round);
      CK(hipStreamWaitEvent(s, e, 0));
      hipLaunchKernelGGL(consume, dim3(n / 256), dim3(256), 0, s, buf, out, n);
      CK(hipEventDestroy(e));
      hipEvent_t back; CK(hipEventCreateWithFlags(&back, hipEventDisableTiming));      // the second producer must not overtake the consumer
      CK(hipEventRecord(back, s)); CK(hipStreamWaitEvent(m, back, 0)); CK(hipEventDestroy(back));
      hipLaunchKernelGGL(produce, dim3(n / 256), dim3(256), 0, m, buf, n, tag + 1000.f, 10);
      CK(hipDeviceSynchronize());
      CK(hipMemcpy(h.data(), out, n * 4, hipMemcpyDeviceToHost));
      for (int i = 0; i < n; ++i) {
        if (h[i] == tag) ++saw_first;
        else if (h[i] == tag + 1000.f) ++saw_second;
        else ++bad_early;
      }
    }
  }
  printf("elements seen: first producer %d, second producer %d, stale (consumer ran early) %d -> %s\n", saw_first, saw_second, bad_early,
         bad_early == 0 && saw_second == 0 ? "OK" : "FAIL");
  return bad_early == 0 && saw_second == 0 ? 0 : 1;
}
