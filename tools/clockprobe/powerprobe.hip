// Diagnostic (not part of the product): what the fp16 matrix pipe sustains under the board's power cap when its operands are
// DATA (pseudo-random halves that change from instruction to instruction) rather than the constant registers of f16probe's rate
// loop.  Run under tools/clock_power_sample.py "cmd:..." to read clock and socket power beside the rate.
//   powerprobe <seconds> <mode>   mode 0: constant operands (f16probe's loop), 1: 16 rotating random operand sets per wave,
//                                 2: mode 1 with one VALU fma per MFMA beside it (a crude stand-in for split / softmax work)
//                                 3: mode 1 with v_mfma_f32_32x32x16_f16 (the same flop per cycle at peak, half the operand bytes per flop)
//                                 4: mode 1 plus one ds_read_b128 per matrix instruction (operands streamed from LDS as the MLP kernels do)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <chrono>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ __launch_bounds__(512) void k_power32(float* sink, int iters) {
  f32x16 a0, a1;
  for (int r = 0; r < 16; ++r) { a0[r] = 0.f; a1[r] = 0.f; }
  f16x8 op[16];
#pragma unroll
  for (int s = 0; s < 16; ++s)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const unsigned h = (threadIdx.x * 131u + blockIdx.x * 7919u + s * 17u + q) * 2654435761u;
      op[s][q] = (_Float16)(((int)((h >> 8) & 0xffff) - 32768) * (1.0f / 16384.0f));
    }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(op[(4 * u) & 15], op[(4 * u + 5) & 15], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(op[(4 * u + 1) & 15], op[(4 * u + 7) & 15], a1, 0, 0, 0);
    }
    if ((i & 255) == 255) {
#pragma unroll
      for (int r = 0; r < 16; ++r) { a0[r] *= 1e-3f; a1[r] *= 1e-3f; }
    }
  }
  if (a0[0] + a1[1] == 12345.f) sink[0] = 1.f;
}

__global__ __launch_bounds__(512) void k_power_lds(float* sink, int iters) {
  __shared__ __attribute__((aligned(16))) char lds[32 * 1024];
  for (int i = threadIdx.x; i < 32 * 1024 / 4; i += 512) {
    const unsigned h = (i * 131u + blockIdx.x * 7919u) * 2654435761u;
    reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u ^ (h & 0x03ff03ffu);
  }
  __syncthreads();
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  f16x8 b[4];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int q = 0; q < 8; ++q) b[s][q] = (_Float16)(0.01f * (float)((threadIdx.x + s * 7 + q) & 63) - 0.3f);
  const char* base = lds + (threadIdx.x & 63) * 16;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const f16x8 x0 = *reinterpret_cast<const f16x8*>(base + ((4 * u + (i & 3) * 16) & 31) * 1024);
      const f16x8 x1 = *reinterpret_cast<const f16x8*>(base + ((4 * u + 1 + (i & 3) * 16) & 31) * 1024);
      const f16x8 x2 = *reinterpret_cast<const f16x8*>(base + ((4 * u + 2 + (i & 3) * 16) & 31) * 1024);
      const f16x8 x3 = *reinterpret_cast<const f16x8*>(base + ((4 * u + 3 + (i & 3) * 16) & 31) * 1024);
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x0, b[0], a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x1, b[1], a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x2, b[2], a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x3, b[3], a3, 0, 0, 0);
    }
    if ((i & 255) == 255) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { a0[r] *= 1e-3f; a1[r] *= 1e-3f; a2[r] *= 1e-3f; a3[r] *= 1e-3f; }
    }
  }
  if (a0[0] + a1[1] + a2[2] + a3[3] == 12345.f) sink[0] = 1.f;
}

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

template <int MODE>
__global__ __launch_bounds__(512) void k_power(float* sink, int iters) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  f16x8 op[16];
#pragma unroll
  for (int s = 0; s < 16; ++s)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const unsigned h = hash32(threadIdx.x * 131u + blockIdx.x * 7919u + s * 17u + q);
      op[s][q] = MODE == 0 ? (_Float16)(1.0f + q) : (_Float16)(((int)(h & 0xffff) - 32768) * (1.0f / 16384.0f));
    }
  float v0 = threadIdx.x * 1e-3f, v1 = 1.0001f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(op[(4 * u) & 15], op[(4 * u + 5) & 15], a0, 0, 0, 0);
      if (MODE == 2) v0 = __builtin_fmaf(v0, v1, 0.25f);
      a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(op[(4 * u + 1) & 15], op[(4 * u + 7) & 15], a1, 0, 0, 0);
      if (MODE == 2) v0 = __builtin_fmaf(v0, v1, 0.5f);
      a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(op[(4 * u + 2) & 15], op[(4 * u + 9) & 15], a2, 0, 0, 0);
      if (MODE == 2) v0 = __builtin_fmaf(v0, v1, 0.75f);
      a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(op[(4 * u + 3) & 15], op[(4 * u + 11) & 15], a3, 0, 0, 0);
      if (MODE == 2) v0 = __builtin_fmaf(v0, v1, 1.0f);
    }
    if ((i & 255) == 255) {      // keep the accumulators finite: results shrink back (cheap, rare)
#pragma unroll
      for (int r = 0; r < 4; ++r) { a0[r] *= 1e-3f; a1[r] *= 1e-3f; a2[r] *= 1e-3f; a3[r] *= 1e-3f; }
    }
  }
  if (a0[0] + a1[1] + a2[2] + a3[3] + v0 == 12345.f) sink[0] = 1.f;
}

int main(int argc, char** argv) {
  const double secs = argc > 1 ? atof(argv[1]) : 2.0;
  const int mode = argc > 2 ? atoi(argv[2]) : 1;
  float* d; hipMalloc(&d, 4096);
  const int iters = 20000, blocks = 256 * 2;
  auto launch = [&]() {
    if (mode == 0) hipLaunchKernelGGL(k_power<0>, dim3(blocks), dim3(512), 0, 0, d, iters);
    else if (mode == 1) hipLaunchKernelGGL(k_power<1>, dim3(blocks), dim3(512), 0, 0, d, iters);
    else if (mode == 2) hipLaunchKernelGGL(k_power<2>, dim3(blocks), dim3(512), 0, 0, d, iters);
    else if (mode == 3) hipLaunchKernelGGL(k_power32, dim3(blocks), dim3(512), 0, 0, d, iters);
    else hipLaunchKernelGGL(k_power_lds, dim3(blocks), dim3(512), 0, 0, d, iters);
  };
  launch(); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double total_ms = 0; long n = 0;
  auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); total_ms += ms; ++n;
  }
  const double flop = (double)n * blocks * 8 * (double)iters * (mode == 3 ? 8 * 32768.0 : 16 * 16384.0);
  printf("powerprobe mode %d: %ld launches, %.1f ms each, %.0f TFLOP/s of fp16 matrix instructions (16x16x32)\n", mode, n, total_ms / n, flop / (total_ms * 1e-3) / 1e12);
  return 0;
}
