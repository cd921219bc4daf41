// Diagnostic (not part of the product): the rollout's MLP REGION (rr_rollout_w.inc, instance mode) as a stand-alone loop, to price
// the two-tiles-per-wave form before it is built.  A region = 16 fragment reads of 1 KB from a three-buffer LDS ring (static
// content here), per rollout tile 24 v_mfma_f32_16x16x32_f16 (12 in four output chains + 12 in one hidden chain, as in the kernel),
// relu + two-piece fp16 split of a hidden pair (4 + 12 vector instructions), one workgroup barrier.
//   TILES = 1: the product's form (a wave owns one 16-rollout tile; seven or eight waves per workgroup, two per SIMD)
//   TILES = 2: a wave owns two tiles; every fragment read feeds both (one wave per SIMD for the same eight tiles)
// Prints cycles per region and the matrix pipe's share of them.  Build: hipcc --offload-arch=gfx950 -O3 -I../../real-routing-nco_amd/csrc
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "rr_common.h"

template <int TILES, int SCHED>
__global__ __launch_bounds__(512) void k_region(int nreg, unsigned wmask, unsigned long long* cyc, float* out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 48 * 1024 / 4; i += 512) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + (unsigned)(i & 7);     // fp16 values near 1
  __syncthreads();
  if (!((wmask >> wave) & 1u)) return;
  rr_f16x8 Gs[TILES][4][2], Hh[TILES], Hl[TILES];
  f32x4 Fa[TILES][8], cA[TILES], cB[TILES];
  float hx[TILES][8];
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
#pragma unroll
    for (int sl = 0; sl < 4; ++sl)
#pragma unroll
      for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int e = 0; e < 8; ++e) Gs[t][sl][p][e] = (_Float16)(0.001f * (float)(lane + e + sl + t));
#pragma unroll
    for (int u = 0; u < 8; ++u) Fa[t][u] = rr_zero4();
    cA[t] = rr_zero4(); cB[t] = rr_zero4();
#pragma unroll
    for (int e = 0; e < 8; ++e) { Hh[t][e] = (_Float16)0.5f; Hl[t][e] = (_Float16)0.001f; hx[t][e] = 0.f; }
  }
  rr_f16x8 Xa[8], Xb[8];
  auto s_read = [&](rr_f16x8 (&B)[8], int buf, int grp) {
#pragma unroll
    for (int q = 0; q < 8; ++q) B[q] = *reinterpret_cast<const rr_f16x8*>(lds + buf * 16384 + (grp * 8 + q) * 1024 + lane * 16);
  };
  auto hid = [&](const rr_f16x8 (&B)[8], f32x4 (&c)[TILES]) {
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
#pragma unroll
      for (int t = 0; t < TILES; ++t) c[t] = rr_mfma_f16(B[2 * sl], Gs[t][sl][0], c[t]);
#pragma unroll
      for (int t = 0; t < TILES; ++t) c[t] = rr_mfma_f16(B[2 * sl], Gs[t][sl][1], c[t]);
#pragma unroll
      for (int t = 0; t < TILES; ++t) c[t] = rr_mfma_f16(B[2 * sl + 1], Gs[t][sl][0], c[t]);
    }
  };
  auto output4 = [&](const rr_f16x8 (&B)[8], int u0) {
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) Fa[t][u0 + u] = rr_mfma_f16(B[2 * u], Hh[t], Fa[t][u0 + u]);
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) Fa[t][u0 + u] = rr_mfma_f16(B[2 * u], Hl[t], Fa[t][u0 + u]);
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
      for (int u = 0; u < 4; ++u) Fa[t][u0 + u] = rr_mfma_f16(B[2 * u + 1], Hh[t], Fa[t][u0 + u]);
  };
  auto relu4 = [&](const f32x4& c, float* h) {
#pragma unroll
    for (int r = 0; r < 4; ++r) h[r] = rr_relu(c[r]);
  };
  int gq = 0;
  auto bar = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    gq = gq == 2 ? 0 : gq + 1;
    __builtin_amdgcn_sched_barrier(0);
  };
  s_read(Xb, gq, 1);
  bar();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#define SB() do { if (SCHED) __builtin_amdgcn_sched_barrier(0); } while (0)
  const bool upper = SCHED == 2 && wave >= 4;               // SCHED 2: the second wave of a SIMD starts a region with matrix work
  if (!upper) {
#pragma unroll 1
  for (int r = 0; r < nreg; r += 2) {
    // region 2q + 1: O2(q - 1), H(2q + 2); relu of tile 2q + 1, split of pair q
    s_read(Xa, gq, 0);
    SB();
    output4(Xb, 4);
#pragma unroll
    for (int t = 0; t < TILES; ++t) { relu4(cB[t], hx[t] + 4); rr_usplit8(hx[t], Hh[t], Hl[t]); }
    SB();
    s_read(Xb, gq, 1);
    SB();
    hid(Xa, cA);
    bar();
    // region 2q + 2: H(2q + 3), O1(q); relu of tile 2q + 2
    s_read(Xa, gq, 0);
    SB();
    hid(Xb, cB);
#pragma unroll
    for (int t = 0; t < TILES; ++t) relu4(cA[t], hx[t]);
    SB();
    s_read(Xb, gq, 1);
    SB();
    output4(Xa, 0);
    bar();
  }
  } else {
#pragma unroll 1
  for (int r = 0; r < nreg; r += 2) {
    output4(Xb, 4);
#pragma unroll
    for (int t = 0; t < TILES; ++t) { relu4(cB[t], hx[t] + 4); rr_usplit8(hx[t], Hh[t], Hl[t]); }
    SB();
    s_read(Xa, gq, 0);
    s_read(Xb, gq, 1);
    SB();
    hid(Xa, cA);
    bar();
    hid(Xb, cB);
#pragma unroll
    for (int t = 0; t < TILES; ++t) relu4(cA[t], hx[t]);
    SB();
    s_read(Xa, gq, 0);
    s_read(Xb, gq, 1);
    SB();
    output4(Xa, 0);
    bar();
  }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) atomicAdd(cyc, t1 - t0);
  float acc = 0.f;
#pragma unroll
  for (int t = 0; t < TILES; ++t) {
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += Fa[t][u][0] + Fa[t][u][1] + Fa[t][u][2] + Fa[t][u][3];
    acc += cA[t][0] + cB[t][1];
  }
  if (acc == 12345.f) out[0] = acc;
}

template <int TILES, int SCHED>
static void run(unsigned wmask, unsigned long long* cyc, float* out) {
  const int grid = 1024, nreg = 3200;
  hipFuncSetAttribute((const void*)k_region<TILES, SCHED>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((k_region<TILES, SCHED>), dim3(256), dim3(512), 160 * 1024, 0, 32, wmask, cyc, out);
  hipMemset(cyc, 0, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_region<TILES, SCHED>), dim3(grid), dim3(512), 160 * 1024, 0, nreg, wmask, cyc, out);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const int nw = __builtin_popcount(wmask);
  const double per_region = (double)c / grid / nw / nreg;
  int per_simd[4] = {0, 0, 0, 0};
  for (int w = 0; w < 8; ++w) if ((wmask >> w) & 1u) per_simd[w & 3]++;
  const int wmax = per_simd[0] > per_simd[3] ? per_simd[0] : per_simd[3];
  printf("sched %d tiles per wave %d, waves 0x%02x (%d, up to %d per SIMD): %7.1f cycles per region, matrix pipe %4.1f %% busy, %6.2f tile-regions per k-cycle and CU, %7.2f ms\n",
         SCHED, TILES, wmask, nw, wmax, per_region, 100.0 * 24 * TILES * wmax * 16 / per_region, 1000.0 * nw * TILES / per_region, ms);
  fflush(stdout);
}

int main() {
  unsigned long long* cyc; float* out;
  hipMalloc(&cyc, 8); hipMalloc(&out, 4);
  run<1, 0>(0x01, cyc, out); run<1, 0>(0x0f, cyc, out); run<1, 0>(0x7f, cyc, out); run<1, 0>(0xff, cyc, out);
  run<1, 1>(0x01, cyc, out); run<1, 1>(0x7f, cyc, out);
  run<1, 2>(0x7f, cyc, out); run<1, 2>(0xff, cyc, out);
  run<2, 0>(0x01, cyc, out); run<2, 0>(0x0f, cyc, out); run<2, 0>(0xff, cyc, out);
  run<2, 1>(0x0f, cyc, out);
  return 0;
}
