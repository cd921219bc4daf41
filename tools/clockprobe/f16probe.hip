// Diagnostic (not part of the product): is a two-piece fp16 split (x = hi + 2^-11 lo', both fp16) on v_mfma_f32_16x16x32_f16 as
// accurate as the fp32 MFMA?  Checks (1) whether the f16 MFMA keeps subnormal inputs, (2) the error of a K-long dot product
// computed as hi*hi + 2^-11 (hi*lo' + lo'*hi) against float64, beside the fp32 MFMA and the three-piece bf16 split,
// (3) the issue rate of the f16 MFMA.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

// A[16][K], B[16][K] row-major fp32; C[16][16] = A B^T by three methods
__device__ __forceinline__ void split_f16(const float (&x)[8], f16x8& hi, f16x8& lo) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2 v = {x[2 * q], x[2 * q + 1]};
    const f16x2 h = __builtin_convertvector(v, f16x2);
    const f32x2 r = (v - __builtin_convertvector(h, f32x2)) * 2048.0f;
    const f16x2 l = __builtin_convertvector(r, f16x2);
    hi[2 * q] = h[0]; hi[2 * q + 1] = h[1]; lo[2 * q] = l[0]; lo[2 * q + 1] = l[1];
  }
}
__device__ __forceinline__ void split_bf16(const float (&x)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2 v = {x[2 * q], x[2 * q + 1]};
    const bf16x2 h = __builtin_convertvector(v, bf16x2);
    const f32x2 r1 = v - __builtin_convertvector(h, f32x2);
    const bf16x2 m = __builtin_convertvector(r1, bf16x2);
    const f32x2 r2 = r1 - __builtin_convertvector(m, f32x2);
    const bf16x2 l = __builtin_convertvector(r2, bf16x2);
    hi[2 * q] = h[0]; hi[2 * q + 1] = h[1]; mid[2 * q] = m[0]; mid[2 * q + 1] = m[1]; lo[2 * q] = l[0]; lo[2 * q + 1] = l[1];
  }
}
__global__ void k_gemm(const float* A, const float* B, float* C, int K) {   // one wave
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  f32x4 c32 = {0, 0, 0, 0}, cb = {0, 0, 0, 0}, cs = {0, 0, 0, 0}, c3 = {0, 0, 0, 0}, c3s = {0, 0, 0, 0};
  for (int k0 = 0; k0 < K; k0 += 32) {
    float a[8], b[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) { a[q] = A[i * K + k0 + 8 * g + q]; b[q] = B[i * K + k0 + 8 * g + q]; }
    // fp32 MFMA: 16x16x4, lane (i, g) holds k = g: 8 instructions cover k0 .. k0+31 in the order k = 8g + q  <->  (q, g)
#pragma unroll
    for (int q = 0; q < 8; ++q) c32 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[q], b[q], c32, 0, 0, 0);
    f16x8 ah, al, bh, bl;
    split_f16(a, ah, al); split_f16(b, bh, bl);
    cb = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, cb, 0, 0, 0);
    cs = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, cs, 0, 0, 0);
    cs = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, cs, 0, 0, 0);
    bf16x8 a0, a1, a2, b0, b1, b2;
    split_bf16(a, a0, a1, a2); split_bf16(b, b0, b1, b2);
    c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, c3, 0, 0, 0);
    c3s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b1, c3s, 0, 0, 0);
    c3s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b0, c3s, 0, 0, 0);
    c3s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b2, c3s, 0, 0, 0);
    c3s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a2, b0, c3s, 0, 0, 0);
    c3s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, c3s, 0, 0, 0);
  }
#pragma unroll
  for (int r = 0; r < 4; ++r) {     // C row = 4g + r (A row), col = i (B row)
    const int o = (4 * g + r) * 16 + i;
    C[o] = c32[r];
    C[256 + o] = cb[r] + cs[r] * (1.0f / 2048.0f);
    C[512 + o] = c3[r] + c3s[r];
  }
}
__global__ void k_denorm(float* out) {   // one wave: A = 2^-20 (fp16 subnormal), B = 2^10: each of the 32 products is 2^-10
  f16x8 a, b;
#pragma unroll
  for (int q = 0; q < 8; ++q) { a[q] = (_Float16)9.5367431640625e-07f; b[q] = (_Float16)1024.0f; }
  f32x4 c = {0, 0, 0, 0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
}
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(512) void k_rate16(float* sink, int iters) {     // v_mfma_f32_16x16x16_f16 (k = 16)
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  f16x4 p, q4;
#pragma unroll
  for (int q = 0; q < 4; ++q) { p[q] = (_Float16)(1.0f + threadIdx.x * 1e-3f + q); q4[q] = (_Float16)(0.5f - q); }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      a0 = __builtin_amdgcn_mfma_f32_16x16x16f16(p, q4, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x16f16(q4, p, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f32_16x16x16f16(p, p, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x16f16(q4, q4, a3, 0, 0, 0);
    }
  }
  if (a0[0] + a1[1] + a2[2] + a3[3] == 12345.f) sink[0] = 1.f;
}
template <bool F16>
__global__ __launch_bounds__(512) void k_rate(float* sink, int iters) {
  f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  f16x8 p, q8; bf16x8 pb, qb;
#pragma unroll
  for (int q = 0; q < 8; ++q) { p[q] = (_Float16)(1.0f + threadIdx.x * 1e-3f + q); q8[q] = (_Float16)(0.5f - q); pb[q] = (__bf16)(float)p[q]; qb[q] = (__bf16)(float)q8[q]; }
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (F16) {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(p, q8, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(q8, p, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(p, p, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(q8, q8, a3, 0, 0, 0);
      } else {
        a0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pb, qb, a0, 0, 0, 0); a1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qb, pb, a1, 0, 0, 0);
        a2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(pb, pb, a2, 0, 0, 0); a3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qb, qb, a3, 0, 0, 0);
      }
    }
  }
  if (a0[0] + a1[1] + a2[2] + a3[3] == 12345.f) sink[0] = 1.f;
}
int main() {
  float* d; hipMalloc(&d, 4096);
  hipLaunchKernelGGL(k_denorm, dim3(1), dim3(64), 0, 0, d);
  float h[2]; hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
  printf("subnormal inputs: 32 x (2^-20 * 2^10) = %.9g (expect %.9g; 0 means flushed); a as float %.9g\n", h[0], 32.0 / 1024.0, h[1]);
  for (int K : {128, 512}) {
    for (double scale : {1.0, 1e-3, 30.0}) {
      std::vector<float> A(16 * K), B(16 * K);
      srand(7);
      for (auto& v : A) v = (float)(scale * (2.0 * rand() / RAND_MAX - 1.0));
      for (auto& v : B) v = (float)(2.0 * rand() / RAND_MAX - 1.0) * (rand() % 7 == 0 ? 1e-4f : 1.0f);
      float *dA, *dB, *dC; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 768 * 4);
      hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k_gemm, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
      std::vector<float> C(768); hipMemcpy(C.data(), dC, 768 * 4, hipMemcpyDeviceToHost);
      double e[3] = {0, 0, 0}, ref_abs = 0;
      for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double r = 0, ra = 0;
        for (int k = 0; k < K; ++k) { r += (double)A[i * K + k] * B[j * K + k]; ra += fabs((double)A[i * K + k] * B[j * K + k]); }
        ref_abs = fmax(ref_abs, ra);
        for (int m = 0; m < 3; ++m) e[m] = fmax(e[m], fabs(C[m * 256 + i * 16 + j] - r) / ra);
      }
      printf("K %3d scale %g: max |err| / sum|a b|:  fp32 MFMA %.3e   f16x2 %.3e   bf16x3 %.3e\n", K, scale, e[0], e[1], e[2]);
    }
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int f = 0; f < 2; ++f) {
    const int it = 200000; float ms;
    if (f) hipLaunchKernelGGL(k_rate<true>, dim3(256), dim3(512), 0, 0, d, 1000); else hipLaunchKernelGGL(k_rate<false>, dim3(256), dim3(512), 0, 0, d, 1000);
    hipEventRecord(e0);
    if (f) hipLaunchKernelGGL(k_rate<true>, dim3(256), dim3(512), 0, 0, d, it); else hipLaunchKernelGGL(k_rate<false>, dim3(256), dim3(512), 0, 0, d, it);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    printf("%s 16x16x32 MFMA: %.1f TFLOP/s\n", f ? "f16 " : "bf16", 256.0 * 8 * it * 16.0 * 16384 / (ms * 1e-3) / 1e12);
  }
  {
    const int it = 200000; float ms;
    hipLaunchKernelGGL(k_rate16, dim3(256), dim3(512), 0, 0, d, 1000);
    hipEventRecord(e0); hipLaunchKernelGGL(k_rate16, dim3(256), dim3(512), 0, 0, d, it); hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
    printf("f16  16x16x16 MFMA: %.1f TFLOP/s (%.2f x the time of a 16x16x32 per instruction)\n", 256.0 * 8 * it * 16.0 * 8192 / (ms * 1e-3) / 1e12, 0.0);
  }
  return 0;
}
