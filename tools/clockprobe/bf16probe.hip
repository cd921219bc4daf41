// Diagnostic (not part of the product): throughput of v_mfma_f32_16x16x32_bf16 on gfx950 and whether fp32 VALU work overlaps
// with it (it does not with the fp32 MFMA: mixprobe.hip).  Each wave issues, per loop trip, 16 bf16 MFMAs (4 independent
// chains) and NV independent v_fma_f32; 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int NV>
__global__ __launch_bounds__(512) void k_mix(float* sink, int iters) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  float x = 1.0f + threadIdx.x * 1e-3f, y = 0.5f + blockIdx.x * 1e-4f;
  bf16x8 p, q8;
#pragma unroll
  for (int q = 0; q < 8; ++q) { p[q] = (__bf16)(x + q); q8[q] = (__bf16)(y - q); }
  float v[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) v[q] = x + q;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, q8, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q8, p, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(p, p, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(q8, q8, a3, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < NV / 4; ++q) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[q & 7]) : "v"(x), "v"(y));
    }
  }
  float s = 0;
#pragma unroll
  for (int q = 0; q < 8; ++q) s += v[q];
  if (a0[0] + a1[1] + a2[2] + a3[3] + s == 12345.f) sink[0] = 1.f;
}
template <int NV>
void run(float* s, int threads) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int it = 400000; float ms;
  hipLaunchKernelGGL(k_mix<NV>, dim3(256), dim3(threads), 0, 0, s, 1000);
  hipEventRecord(e0); hipLaunchKernelGGL(k_mix<NV>, dim3(256), dim3(threads), 0, 0, s, it); hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  const double waves = 256.0 * threads / 64, mf = waves * it * 16.0;
  const double cyc = ms * 1e-3 * 2.4e9 / it / (threads / 256.0);
  printf("waves/SIMD %d  VALU per 16 MFMA %3d : %7.2f ms  %7.1f TFLOP/s bf16 MFMA  %6.1f cycles per (16 MFMA + %d VALU) per wave-slot\n",
         threads / 256, NV, ms, mf * 16384 / (ms * 1e-3) / 1e12, cyc, NV);
}
int main() {
  float* s; hipMalloc(&s, 4);
  for (int threads : {256, 512}) {
    run<0>(s, threads); run<16>(s, threads); run<32>(s, threads); run<64>(s, threads); run<128>(s, threads);
  }
  return 0;
}
