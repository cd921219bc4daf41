// Diagnostic (not part of the product): how fast can a CU stream an L2-resident 512 KB weight image into LDS by LDS-DMA
// (global_load_lds_dwordx4, 1 KB per wave-instruction) when every CU streams the same image — the weight ring of the
// rollout's pointer MLP?  Variables: loader waves per workgroup (NL), pieces in flight per loader wave (D, the vmcnt the
// wave waits for after each request), ring size.  One workgroup per CU (160 KB of LDS declared), no consumers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void glds16(const void* g, void* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int D, int BUF>
__global__ __launch_bounds__(512) void k_dma(const char* w, int npass, int nl, int ring_pieces, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave >= nl) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned lane16 = lane * 16u;
  const __amdgpu_buffer_rsrc_t rw = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(w), 0, 512 * 1024, 0x00020000);
  int slot = wave;
  for (int pass = 0; pass < npass; ++pass) {
#pragma unroll 4
    for (int p = wave; p < 512; p += nl) {
      if (BUF) __builtin_amdgcn_raw_ptr_buffer_load_lds(rw, (__attribute__((address_space(3))) void*)(lds + slot * 1024), 16, lane16, p * 1024, 0, 0);
      else glds16(w + (size_t)p * 1024 + lane16, lds + slot * 1024);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(D) : "memory");
      slot += nl; if (slot >= ring_pieces) slot -= ring_pieces;
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0 && wave == 0) atomicAdd(cyc, __builtin_amdgcn_s_memtime() - t0);
}

// register-staged variant: a loader wave fetches a 16 KB stage with 16 plain 16-byte loads (64 VGPRs), requests the next stage
// into a second register set, then parks the first in LDS with 16 ds_write_b128
__global__ __launch_bounds__(512) void k_stage(const char* w, int npass, int nl, int ring_pieces, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (wave >= nl) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned lane16 = lane * 16u;
  const int nst = 32 / nl;                       // stages of 16 pieces per pass and wave
  float4 a[16], b[16];
  const char* src = w + (size_t)wave * nst * 16384 + lane16;
#pragma unroll
  for (int f = 0; f < 16; ++f) a[f] = *reinterpret_cast<const float4*>(src + f * 1024);
  int slot = wave * 16;
  for (int it = 0; it < npass * nst; it += 2) {
    const char* s1 = src + (size_t)((it + 1) % nst) * 16384;
#pragma unroll
    for (int f = 0; f < 16; ++f) b[f] = *reinterpret_cast<const float4*>(s1 + f * 1024);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#pragma unroll
    for (int f = 0; f < 16; ++f) *reinterpret_cast<float4*>(lds + (slot + f) * 1024 + lane16) = a[f];
    slot += 16 * nl; if (slot >= ring_pieces) slot -= ring_pieces;
    const char* s2 = src + (size_t)((it + 2) % nst) * 16384;
#pragma unroll
    for (int f = 0; f < 16; ++f) a[f] = *reinterpret_cast<const float4*>(s2 + f * 1024);
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
#pragma unroll
    for (int f = 0; f < 16; ++f) *reinterpret_cast<float4*>(lds + (slot + f) * 1024 + lane16) = b[f];
    slot += 16 * nl; if (slot >= ring_pieces) slot -= ring_pieces;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  if (lane == 0 && wave == 0) atomicAdd(cyc, __builtin_amdgcn_s_memtime() - t0);
}
static void run_stage(const char* w, int nl, unsigned long long* cyc) {
  const int grid = 1024, npass = 20;
  hipFuncSetAttribute((const void*)k_stage, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k_stage, dim3(256), dim3(512), 160 * 1024, 0, w, 2, nl, 144, cyc);
  hipMemset(cyc, 0, 8);
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_stage, dim3(grid), dim3(512), 160 * 1024, 0, w, npass, nl, 144, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double per_wg_s = ms * 1e-3 / (grid / 256.0);
  const double gbs = npass * 512.0 * 1024 / per_wg_s / 1e9;
  printf("register-staged loaders=%d : %7.3f ms  %6.1f GB/s per CU  %5.2f TB/s chip  %6.0f cycles per 512 KB pass\n", nl, ms, gbs, gbs * 256 / 1e3,
         (double)c / grid / npass);
}

template <int D, int BUF>
static void run(const char* w, int nl, int ring_kb, unsigned long long* cyc) {
  const int grid = 1024, npass = 20;
  hipFuncSetAttribute((const void*)k_dma<D, BUF>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipMemset(cyc, 0, 8);
  hipLaunchKernelGGL((k_dma<D, BUF>), dim3(256), dim3(512), 160 * 1024, 0, w, 2, nl, ring_kb, cyc);   // warm-up
  hipMemset(cyc, 0, 8);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_dma<D, BUF>), dim3(grid), dim3(512), 160 * 1024, 0, w, npass, nl, ring_kb, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double per_wg_s = ms * 1e-3 / (grid / 256.0);
  const double gbs = npass * 512.0 * 1024 / per_wg_s / 1e9;
  printf("%s loaders=%d in-flight/wave=%2d ring=%3d KB : %7.3f ms  %6.1f GB/s per CU  %5.2f TB/s chip  %6.0f cycles per 512 KB pass (in-kernel clock)\n",
         BUF ? "buffer_load_lds" : "global_load_lds", nl, D + 1, ring_kb, ms, gbs, gbs * 256 / 1e3, (double)c / grid / npass);
}

int main() {
  char* w; unsigned long long* cyc;
  hipMalloc(&w, 512 * 1024); hipMemset(w, 1, 512 * 1024); hipMalloc(&cyc, 8);
  for (int nl : {1, 2, 4, 8}) {
    run<31, 0>(w, nl, 48, cyc);
    run<62, 0>(w, nl, 144, cyc);
  }
  for (int nl : {1, 2, 4}) run<31, 1>(w, nl, 48, cyc);
  for (int nl : {1, 2, 4}) run_stage(w, nl, cyc);
  return 0;
}
