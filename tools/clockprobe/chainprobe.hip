// Diagnostic (not part of the product): does an accumulate chain survive ONE independent matrix instruction between its links?
//   c0 = mfma(a0, b0, c0); c1 = mfma(a1, b1, c1); c0 = mfma(a2, b2, c0); c1 = mfma(a3, b3, c1); ...
// (hipcc emits this order without wait states, e.g. in k_init_embed's gate GEMM; the hand-written k_mlp_rows stage body with it gave
// wrong sums, with the links adjacent or padded by s_nop the right ones.)  Three builds of the same sums: builtins in the interleaved
// order pinned by sched_barrier, inline asm interleaved, inline asm adjacent; compared element by element.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ f16x8 mk(unsigned seed, int lane) {
  f16x8 v;
  for (int q = 0; q < 8; ++q) {
    unsigned h = (seed * 9781u + lane * 131u + q * 17u) * 2654435761u;
    v[q] = (_Float16)(((int)((h >> 8) & 0xff) - 128) * (1.0f / 64.0f));
  }
  return v;
}

template <int MODE>
__global__ __launch_bounds__(64) void k_chain(float* out, int nlinks, int busy) {
  const int lane = threadIdx.x;
  f16x8 a[8], b[8];
  for (int s = 0; s < 8; ++s) { a[s] = mk(2 * s + 1, lane); b[s] = mk(2 * s + 2, lane); }
  f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  for (int it = 0; it < nlinks; ++it) {
    if (MODE == 0) {            // builtins, interleaved, order pinned
#pragma unroll
      for (int s = 0; s < 8; s += 2) {
        c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s], b[s], c0, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[s + 1], b[s + 1], c1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    } else if (MODE == 1) {     // asm, interleaved
      asm volatile(
          "v_mfma_f32_16x16x32_f16 %0, %2, %10, %0\n\t" "v_mfma_f32_16x16x32_f16 %1, %3, %11, %1\n\t"
          "v_mfma_f32_16x16x32_f16 %0, %4, %12, %0\n\t" "v_mfma_f32_16x16x32_f16 %1, %5, %13, %1\n\t"
          "v_mfma_f32_16x16x32_f16 %0, %6, %14, %0\n\t" "v_mfma_f32_16x16x32_f16 %1, %7, %15, %1\n\t"
          "v_mfma_f32_16x16x32_f16 %0, %8, %16, %0\n\t" "v_mfma_f32_16x16x32_f16 %1, %9, %17, %1\n\t"
          "s_nop 15\n\ts_nop 15"
          : "+v"(c0), "+v"(c1)
          : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
            "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
    } else {                    // asm, adjacent links
      asm volatile(
          "v_mfma_f32_16x16x32_f16 %0, %2, %10, %0\n\t" "v_mfma_f32_16x16x32_f16 %0, %4, %12, %0\n\t"
          "v_mfma_f32_16x16x32_f16 %0, %6, %14, %0\n\t" "v_mfma_f32_16x16x32_f16 %0, %8, %16, %0\n\t"
          "v_mfma_f32_16x16x32_f16 %1, %3, %11, %1\n\t" "v_mfma_f32_16x16x32_f16 %1, %5, %13, %1\n\t"
          "v_mfma_f32_16x16x32_f16 %1, %7, %15, %1\n\t" "v_mfma_f32_16x16x32_f16 %1, %9, %17, %1\n\t"
          "s_nop 15\n\ts_nop 15"
          : "+v"(c0), "+v"(c1)
          : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
            "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
    }
  }
  for (int r = 0; r < 4; ++r) { out[(blockIdx.x * 64 + lane) * 8 + r] = c0[r]; out[(blockIdx.x * 64 + lane) * 8 + 4 + r] = c1[r]; }
}

// a second wave on the same SIMD keeping the matrix pipe busy (WG of 8 waves: waves w and w + 4 share a SIMD)
template <int MODE>
__global__ __launch_bounds__(512) void k_chain_busy(float* out, int nlinks) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  f16x8 a[8], b[8];
  for (int s = 0; s < 8; ++s) { a[s] = mk(2 * s + 1, lane); b[s] = mk(2 * s + 2, lane); }
  f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
  for (int it = 0; it < nlinks; ++it) {
    if (MODE == 1) {
      asm volatile(
          "v_mfma_f32_16x16x32_f16 %0, %2, %10, %0\n\t" "v_mfma_f32_16x16x32_f16 %1, %3, %11, %1\n\t"
          "v_mfma_f32_16x16x32_f16 %0, %4, %12, %0\n\t" "v_mfma_f32_16x16x32_f16 %1, %5, %13, %1\n\t"
          "v_mfma_f32_16x16x32_f16 %0, %6, %14, %0\n\t" "v_mfma_f32_16x16x32_f16 %1, %7, %15, %1\n\t"
          "v_mfma_f32_16x16x32_f16 %0, %8, %16, %0\n\t" "v_mfma_f32_16x16x32_f16 %1, %9, %17, %1\n\t"
          "s_nop 15\n\ts_nop 15"
          : "+v"(c0), "+v"(c1)
          : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
            "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
    } else {
      asm volatile(
          "v_mfma_f32_16x16x32_f16 %0, %2, %10, %0\n\t" "v_mfma_f32_16x16x32_f16 %0, %4, %12, %0\n\t"
          "v_mfma_f32_16x16x32_f16 %0, %6, %14, %0\n\t" "v_mfma_f32_16x16x32_f16 %0, %8, %16, %0\n\t"
          "v_mfma_f32_16x16x32_f16 %1, %3, %11, %1\n\t" "v_mfma_f32_16x16x32_f16 %1, %5, %13, %1\n\t"
          "v_mfma_f32_16x16x32_f16 %1, %7, %15, %1\n\t" "v_mfma_f32_16x16x32_f16 %1, %9, %17, %1\n\t"
          "s_nop 15\n\ts_nop 15"
          : "+v"(c0), "+v"(c1)
          : "v"(a[0]), "v"(a[1]), "v"(a[2]), "v"(a[3]), "v"(a[4]), "v"(a[5]), "v"(a[6]), "v"(a[7]),
            "v"(b[0]), "v"(b[1]), "v"(b[2]), "v"(b[3]), "v"(b[4]), "v"(b[5]), "v"(b[6]), "v"(b[7]));
    }
  }
  for (int r = 0; r < 4; ++r) { out[((blockIdx.x * 8 + wave) * 64 + lane) * 8 + r] = c0[r]; out[((blockIdx.x * 8 + wave) * 64 + lane) * 8 + 4 + r] = c1[r]; }
}

static int compare(const float* x, const float* y, int n, const char* what) {
  int bad = 0; double worst = 0;
  for (int i = 0; i < n; ++i) { if (x[i] != y[i]) { ++bad; double d = fabs((double)x[i] - y[i]) / (fabs((double)y[i]) + 1e-30); if (d > worst) worst = d; } }
  printf("%-60s %d of %d values differ (largest relative difference %.2e)\n", what, bad, n, worst);
  return bad;
}

int main() {
  const int blocks = 256, n1 = blocks * 64 * 8, n8 = blocks * 8 * 64 * 8;
  float *d0, *d1, *d2, *e1, *e2;
  hipMalloc(&d0, n1 * 4); hipMalloc(&d1, n1 * 4); hipMalloc(&d2, n1 * 4); hipMalloc(&e1, n8 * 4); hipMalloc(&e2, n8 * 4);
  float *h0 = new float[n1], *h1 = new float[n1], *h2 = new float[n1], *g1 = new float[n8], *g2 = new float[n8];
  for (int links : {1, 4}) {
    hipLaunchKernelGGL(k_chain<0>, dim3(blocks), dim3(64), 0, 0, d0, links, 0);
    hipLaunchKernelGGL(k_chain<1>, dim3(blocks), dim3(64), 0, 0, d1, links, 0);
    hipLaunchKernelGGL(k_chain<2>, dim3(blocks), dim3(64), 0, 0, d2, links, 0);
    hipLaunchKernelGGL(k_chain_busy<1>, dim3(blocks), dim3(512), 0, 0, e1, links);
    hipLaunchKernelGGL(k_chain_busy<2>, dim3(blocks), dim3(512), 0, 0, e2, links);
    hipDeviceSynchronize();
    hipMemcpy(h0, d0, n1 * 4, hipMemcpyDeviceToHost); hipMemcpy(h1, d1, n1 * 4, hipMemcpyDeviceToHost); hipMemcpy(h2, d2, n1 * 4, hipMemcpyDeviceToHost);
    hipMemcpy(g1, e1, n8 * 4, hipMemcpyDeviceToHost); hipMemcpy(g2, e2, n8 * 4, hipMemcpyDeviceToHost);
    printf("%d block(s) of 8 links per chain:\n", links);
    compare(h0, h2, n1, "  builtins interleaved (pinned) vs asm adjacent, 1 wave/SIMD");
    compare(h1, h2, n1, "  asm interleaved vs asm adjacent, one wave per CU");
    compare(g1, g2, n8, "  asm interleaved vs asm adjacent, two waves per SIMD");
  }
  return 0;
}
