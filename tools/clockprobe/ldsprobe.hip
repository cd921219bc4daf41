// Diagnostic (not part of the product): LDS read bandwidth of one CU on gfx950.  NW waves of a 512-thread workgroup (one workgroup
// per CU, 160 KB declared) read 1 KB fragments (lane l: 16 bytes at 16 l, conflict-free) with ds_read_b128 / b64 / b32 in a loop,
// 16 reads in flight per wave.  Prints cycles per wave-instruction and bytes per clock and CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));

template <int W>
__global__ __launch_bounds__(512) void k_lds(int niter, unsigned wmask, unsigned long long* cyc, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 160 * 1024 / 4; i += 512) reinterpret_cast<float*>(lds)[i] = (float)i;
  __syncthreads();
  if (!((wmask >> wave) & 1u)) return;
  const unsigned a = (unsigned)lane * (unsigned)W + (unsigned)wave * 16384u;
  f4 acc = {0, 0, 0, 0};
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < niter; ++it) {
    if (W == 16) {
      f4 r[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[q]) : "v"(a), "n"(q * 1024));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < 16; ++q) acc += r[q];
    } else if (W == 8) {
      f2 r[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(r[q]) : "v"(a), "n"(q * 512));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < 16; ++q) { acc[0] += r[q][0]; acc[1] += r[q][1]; }
    } else {
      float r[16];
#pragma unroll
      for (int q = 0; q < 16; ++q) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(r[q]) : "v"(a), "n"(q * 256));
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[0] += r[q];
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) atomicAdd(cyc, t1 - t0);
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) sink[0] = 1.f;
}

template <int W>
static void run(unsigned wmask, unsigned long long* cyc, float* sink) {
  const int grid = 512, niter = 2000;
  hipFuncSetAttribute((const void*)k_lds<W>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((k_lds<W>), dim3(256), dim3(512), 160 * 1024, 0, 10, wmask, cyc, sink);
  hipMemset(cyc, 0, 8);
  hipLaunchKernelGGL((k_lds<W>), dim3(grid), dim3(512), 160 * 1024, 0, niter, wmask, cyc, sink);
  hipDeviceSynchronize();
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const int nw = __builtin_popcount(wmask);
  const double per_instr = (double)c / grid / nw / niter / 16.0;           // cycles per wave-instruction as seen by one wave
  printf("ds_read_b%-3d waves=0x%02x (%d): %6.1f cycles per wave-instruction per wave -> %6.1f B/clk per CU\n", W * 8, wmask, nw, per_instr,
         nw * 64.0 * W / per_instr);
  fflush(stdout);
}

int main() {
  unsigned long long* cyc; float* sink;
  hipMalloc(&cyc, 8); hipMalloc(&sink, 4);
  for (unsigned m : {0x01u, 0x03u, 0x0fu, 0x7fu, 0xffu}) { run<16>(m, cyc, sink); run<8>(m, cyc, sink); run<4>(m, cyc, sink); }
  return 0;
}
