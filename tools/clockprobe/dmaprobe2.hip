// Diagnostic (not part of the product): what paces ONE wave's L2 -> LDS stream?  dmaprobe.hip found one 1 KB
// global_load_lds_dwordx4 per 68 cycles and wave whatever is in flight.  Variants here, all with every CU streaming the same
// L2-resident 512 KB image (grid 1024 x 512 threads, 160 KB of LDS declared, only the named waves work):
//   0 LDS-DMA dwordx4, 64-bit vector address          1 LDS-DMA dwordx4, scalar base + 32-bit vector offset
//   2 LDS-DMA dword (256 B per instruction)           3 plain global_load_dwordx4 into registers (no LDS write)
//   4 one LDS-DMA dwordx4 + one plain dwordx4 load per iteration      5 ds_write_b128 alone
//   6 one LDS-DMA dwordx4 + one plain load + one ds_write_b128 per iteration (the hybrid loader)
// `wmask` selects the working waves (bit w = wave w; waves w and w+4 share a SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

#define PROBE_BODY(u, d) {\
      const unsigned o = (off + (unsigned)u * 1024u) & (512u * 1024u - 1u);\
      const unsigned l = slot + (((unsigned)(it + u) & 15u) << 10);\
      if (V == 0) {\
        const char* p = w + o;\
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(l), "v"(p) : "memory");\
      } else if (V == 1) {\
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(o), "s"(w) : "memory");\
      } else if (V == 2) {\
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, %2" ::"s"(l), "v"(o >> 2), "s"(w) : "memory");\
      } else if (V == 3) {\
        asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(d) : "v"(o), "s"(w) : "memory");\
      } else if (V == 4) {\
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(o), "s"(w) : "memory");\
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(d) : "v"(o), "s"(w) : "memory");\
      } else if (V == 5) {\
        asm volatile("ds_write_b128 %0, %1" ::"v"(ldsw + (((unsigned)(it + u) & 7u) << 10)), "v"(r0) : "memory");\
      } else {\
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(l), "v"(o), "s"(w) : "memory");\
        asm volatile("ds_write_b128 %0, %1" ::"v"(ldsw + 8192u + (((unsigned)(it + u) & 7u) << 10)), "v"(d) : "memory");\
        asm volatile("global_load_dwordx4 %0, %1, %2 offset:2048" : "=v"(d) : "v"(o), "s"(w) : "memory");\
      }\
    }
template <int V>
__global__ __launch_bounds__(512) void k_probe(const char* w, int niter, unsigned wmask, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (!((wmask >> wave) & 1u)) return;
  const unsigned lane16 = lane * 16u;
  const char* base = w + lane16;
  unsigned off = lane16;                    // byte offset inside the 512 KB image
  unsigned slot = (unsigned)wave * 16384u;  // LDS byte offset of this wave's 16 KB strip
  unsigned ldsw = slot + lane16;
  f4 r0 = {0, 0, 0, 0}, r1 = r0, r2 = r0, r3 = r0;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < niter; it += 4) {
    PROBE_BODY(0, r0) PROBE_BODY(1, r1) PROBE_BODY(2, r2) PROBE_BODY(3, r3)
    if (V == 5) asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    else if (V == 4 || V == 6) asm volatile("s_waitcnt vmcnt(56)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(60)" ::: "memory");
    off += 4096u;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) {
    atomicAdd(cyc, t1 - t0);
    if (r0.x + r1.x + r2.x + r3.x == 12345.f) cyc[1] = 1;
  }
}

static int g_only = -1;
template <int V>
static void run(const char* name, const char* w, unsigned wmask, unsigned long long* cyc) {
  if (g_only >= 0 && g_only != V) return;
  printf("[%d] ", V); fflush(stdout);
  const int grid = 1024, niter = 4096;
  hipFuncSetAttribute((const void*)k_probe<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((k_probe<V>), dim3(256), dim3(512), 160 * 1024, 0, w, 64, wmask, cyc);
  hipMemset(cyc, 0, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_probe<V>), dim3(grid), dim3(512), 160 * 1024, 0, w, niter, wmask, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const int nw = __builtin_popcount(wmask);
  printf("%-52s waves=0x%02x : %8.3f ms  %7.1f cycles per iteration and wave\n", name, wmask, ms, (double)c / grid / nw / niter);
  fflush(stdout);
}

// 7: one M0 write per four requests, the instruction offset (0 / 1024 / 2048 / 3072) moves both addresses
// 8: s_mov m0 + request, nothing between       9: v7 with NO M0 write inside the loop (same LDS strip over and over)
template <int V>
__global__ __launch_bounds__(512) void k_probe2(const char* w, int niter, unsigned wmask, unsigned long long* cyc) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (!((wmask >> wave) & 1u)) return;
  const unsigned lane16 = lane * 16u;
  unsigned off = lane16;
  const unsigned slot = (unsigned)wave * 16384u;
  if (V == 9) asm volatile("s_mov_b32 m0, %0" ::"s"(slot) : "memory");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < niter; it += 4) {
    const unsigned o = off & (512u * 1024u - 1u);
    const unsigned l = slot + (((unsigned)it & 12u) << 10);
    if (V == 7) {
      asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                   "global_load_lds_dwordx4 %1, %2 offset:2048\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072" ::"s"(l), "v"(o), "s"(w) : "memory");
    } else if (V == 8) {
      asm volatile("s_mov_b32 m0, %0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_add_i32 m0, %0, 0x400\n\tglobal_load_lds_dwordx4 %1, %2 offset:1024\n\t"
                   "s_add_i32 m0, %0, 0x800\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\ts_add_i32 m0, %0, 0xc00\n\tglobal_load_lds_dwordx4 %1, %2 offset:3072" ::"s"(l), "v"(o), "s"(w) : "memory", "scc");
    } else {
      asm volatile("global_load_lds_dwordx4 %0, %1\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024\n\t"
                   "global_load_lds_dwordx4 %0, %1 offset:2048\n\tglobal_load_lds_dwordx4 %0, %1 offset:3072" ::"v"(o), "s"(w) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(60)" ::: "memory");
    off += 4096u;
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) atomicAdd(cyc, t1 - t0);
}
template <int V>
static void run2(const char* name, const char* w, unsigned wmask, unsigned long long* cyc) {
  if (g_only >= 0 && g_only != V) return;
  const int grid = 1024, niter = 4096;
  hipFuncSetAttribute((const void*)k_probe2<V>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((k_probe2<V>), dim3(256), dim3(512), 160 * 1024, 0, w, 64, wmask, cyc);
  hipMemset(cyc, 0, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_probe2<V>), dim3(grid), dim3(512), 160 * 1024, 0, w, niter, wmask, cyc);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const int nw = __builtin_popcount(wmask);
  printf("[%d] %-52s waves=0x%02x : %8.3f ms  %7.1f cycles per request and wave\n", V, name, wmask, ms, (double)c / grid / nw / niter);
  fflush(stdout);
}

int main(int argc, char** argv) {
  if (argc > 1) g_only = atoi(argv[1]);
  char* w; unsigned long long* cyc;
  hipMalloc(&w, 512 * 1024 + 8192); hipMemset(w, 1, 512 * 1024 + 8192); hipMalloc(&cyc, 16);
  for (unsigned m : {0x01u, 0x03u, 0x11u, 0x0fu}) {
    run<0>("LDS-DMA x4, 64-bit vaddr", w, m, cyc);
    run<1>("LDS-DMA x4, saddr + voffset", w, m, cyc);
  }
  run<2>("LDS-DMA x1 (256 B), saddr + voffset", w, 0x01, cyc);
  run<3>("plain dwordx4 load (no LDS)", w, 0x01, cyc);
  run<3>("plain dwordx4 load (no LDS)", w, 0x11, cyc);
  run<4>("LDS-DMA x4 + plain dwordx4 per iteration", w, 0x01, cyc);
  run<5>("ds_write_b128", w, 0x01, cyc);
  run<6>("LDS-DMA x4 + plain dwordx4 + ds_write_b128", w, 0x01, cyc);
  run<6>("LDS-DMA x4 + plain dwordx4 + ds_write_b128", w, 0x11, cyc);
  run2<7>("LDS-DMA x4, one M0 write per 4 (inst offsets)", w, 0x01, cyc);
  run2<7>("LDS-DMA x4, one M0 write per 4 (inst offsets)", w, 0x11, cyc);
  run2<8>("LDS-DMA x4, s_mov/s_add m0 + request", w, 0x01, cyc);
  run2<9>("LDS-DMA x4, no M0 write in the loop", w, 0x01, cyc);
  return 0;
}
