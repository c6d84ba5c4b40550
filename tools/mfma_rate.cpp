// cycles per MFMA of the bf16 shapes a split GEMM could use for a short last k-step (one wave per SIMD, 4 accumulators)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, int n, unsigned long long* cyc) {
  f32x4 acc[4];
  for (int i = 0; i < 4; ++i) acc[i] = f32x4(0.f);
  bf16x8 a8, b8;
  s16x4 a4, b4;
  for (int i = 0; i < 8; ++i) { a8[i] = (__bf16)(0.5f + threadIdx.x * 1e-3f); b8[i] = (__bf16)(0.25f); }
  for (int i = 0; i < 4; ++i) { a4[i] = (short)(0x3f00 + threadIdx.x); b4[i] = (short)0x3e80; }
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < n; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (KIND == 0) acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a8, b8, acc[j], 0, 0, 0);
      else acc[j] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a4, b4, acc[j], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 4; ++j) s += acc[j][0] + acc[j][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[KIND] = t1 - t0;
}
int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 16);
  const int n = 20000;
  for (int rep = 0; rep < 2; ++rep) {
    hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, n, cyc);
    hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, n, cyc);
  }
  hipDeviceSynchronize();
  unsigned long long h[2]; hipMemcpy(h, cyc, 16, hipMemcpyDeviceToHost);
  printf("v_mfma_f32_16x16x32_bf16: %.2f cycles per MFMA (s_memtime ticks)\n", (double)h[0] / (4.0 * n));
  printf("v_mfma_f32_16x16x16_bf16: %.2f cycles per MFMA\n", (double)h[1] / (4.0 * n));
  return 0;
}
