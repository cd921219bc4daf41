// Diagnostic (not part of the product): at what rate does s_memtime tick, idle vs under fp32-MFMA load?
// k_idle: one wave per CU sleeping; k_mfma: 8 waves per CU issuing back-to-back v_mfma_f32_16x16x4_f32.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k_idle(unsigned long long* out, int iters) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) __builtin_amdgcn_s_sleep(127);
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
}
__global__ __launch_bounds__(512) void k_mfma(unsigned long long* out, float* sink, int iters) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float x = 1.0f + threadIdx.x * 1e-3f, y = 0.5f + blockIdx.x * 1e-4f;
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, x, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, x, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(y, y, a3, 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
  if (a0[0] + a1[1] + a2[2] + a3[3] == 12345.f) sink[0] = 1.f;
}
int main() {
  unsigned long long* d; float* s; hipMalloc(&d, 8); hipMalloc(&s, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    unsigned long long h; float ms;
    hipEventRecord(e0); hipLaunchKernelGGL(k_idle, dim3(256), dim3(64), 0, 0, d, 20000); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("idle : %.2f ms, %llu ticks -> %.3f GHz\n", ms, h, h / (ms * 1e6));
    for (int it : {200000, 2000000}) {
      hipEventRecord(e0); hipLaunchKernelGGL(k_mfma, dim3(256), dim3(512), 0, 0, d, s, it); hipEventRecord(e1); hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
      double flop = 256.0 * 8 * it * 64.0 * 2048.0;
      printf("mfma : %.2f ms, %llu ticks -> %.3f GHz, %.1f TFLOP/s fp32 MFMA (nominal peak 157.3)\n", ms, h, h / (ms * 1e6), flop / (ms * 1e-3) / 1e12);
    }
  }
  return 0;
}
