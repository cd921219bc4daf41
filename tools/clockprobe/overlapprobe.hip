// Diagnostic (not part of the product): do a matrix-instruction wave and a vector-instruction wave that SHARE a SIMD run side by side?
// Workgroups of 8 waves (waves w and w + 4 share SIMD w): waves 0-3 run dependent v_mfma_f32_16x16x32_f16 chains (the pointer MLP's
// shape), waves 4-7 run fp32 vector work (fma chains, optionally every 8th instruction a v_exp_f32: the attention phase's mix).
//   overlapprobe <mode>   0: only the matrix waves work, 1: only the vector waves, 2: both;  prints ms per launch.
// T(2) ~ max(T(0), T(1)): the two pipes overlap across waves;  T(2) ~ T(0) + T(1): they do not (the "sum" the rollout's phases show).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int TRANS>
__global__ __launch_bounds__(512) void k_overlap(float* sink, int iters, int mode) {
  const int wave = threadIdx.x >> 6;
  if (wave < 4) {
    if (mode == 1) return;
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0;
    f16x8 a, b;
    for (int q = 0; q < 8; ++q) { a[q] = (_Float16)(0.01f * (threadIdx.x % 7) + q * 0.1f); b[q] = (_Float16)(0.5f - 0.1f * q); }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 8; ++u) {        // two dependent chains, as the MLP's hidden tiles
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(b, a, c1, 0, 0, 0);
      }
      if ((i & 63) == 63) { for (int r = 0; r < 4; ++r) { c0[r] *= 1e-3f; c1[r] *= 1e-3f; } }
    }
    if (c0[0] + c1[1] == 12345.f) sink[0] = 1.f;
  } else {
    if (mode == 0) return;
    float v0 = threadIdx.x * 1e-3f, v1 = 0.5f, v2 = 0.25f, v3 = 0.125f;
    const float m = 0.999f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
      for (int u = 0; u < 16; ++u) {       // 64 vector instructions per iteration (4 independent chains) = the issue time of 16 matrix instructions
        v0 = __builtin_fmaf(v0, m, 0.001f); v1 = __builtin_fmaf(v1, m, 0.002f);
        v2 = __builtin_fmaf(v2, m, 0.003f);
        if (TRANS && (u & 1)) v3 = __builtin_amdgcn_exp2f(v3 * 0.5f); else v3 = __builtin_fmaf(v3, m, 0.004f);
      }
    }
    if (v0 + v1 + v2 + v3 == 12345.f) sink[0] = 1.f;
  }
}

int main(int argc, char** argv) {
  float* d; (void)hipMalloc(&d, 4096);
  const int iters = 40000;
  for (int trans = 0; trans < 2; ++trans)
    for (int mode = 0; mode < 3; ++mode) {
      hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
      auto launch = [&]() { if (trans) hipLaunchKernelGGL(k_overlap<1>, dim3(256), dim3(512), 0, 0, d, iters, mode); else hipLaunchKernelGGL(k_overlap<0>, dim3(256), dim3(512), 0, 0, d, iters, mode); };
      launch(); (void)hipDeviceSynchronize();
      (void)hipEventRecord(e0); launch(); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1);
      printf("vector mix %s, mode %d (%s): %.2f ms per launch\n", trans ? "fma + exp" : "fma only", mode,
             mode == 0 ? "matrix waves only" : mode == 1 ? "vector waves only" : "both", ms / 2);
    }
  return 0;
}
