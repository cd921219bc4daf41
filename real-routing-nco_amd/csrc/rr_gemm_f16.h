// rr_gemm_wx (rr_common.h) on the fp16 matrix pipe with two-piece split operands (rr_common.h, "fp16 matrix pipe"): for kernels
// that keep their activations as an LDS image.  Both operands are stored in the k = 16 pair form — per group of four fp32 values
// one 16-byte unit: the weights as [hi x4 | lo' x4] (packing.f16x2_image of a pack_a fragment), the activations as
// [lo' x4 | hi x4] (rr_to_lohi below, applied where the LDS image is written) — so that a k-group costs two instructions per
// node tile, v_mfma_f32_16x16x16_f16(W.hi, X.hi) and v_mfma_f32_16x16x32_f16(W, X) = W.hi X.lo' + W.lo' X.hi, instead of
// four fp32 MFMAs of twice the cycles each.  acc / accs: the large and the small terms (add accs * 2^-11 at the end).
#pragma once
#include "rr_common.h"

// four fp32 values -> the [lo' | hi] unit, as a float4 to store in their place
__device__ __forceinline__ float4 rr_to_lohi(float4 v) {
  const float x[4] = {v.x, v.y, v.z, v.w};
  return __builtin_bit_cast(float4, rr_split4s(x));
}

template <int NT>
__device__ __forceinline__ void rr_gemm_wx_h(f32x4 (&acc)[NT], f32x4 (&accs)[NT], const float4* __restrict__ wp, int kk0, int nkk,
                                             const float* X, int ldx, int xk0, int n_valid, int lane) {
  const int j = lane & 15, g = lane >> 4;
  int rowoff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int node = nt * 16 + j;
    node = node < n_valid ? node : n_valid - 1;
    rowoff[nt] = node * ldx + 4 * g + xk0;
  }
  // weight fragments up to eight k-groups ahead (a k-group is 2 NT matrix instructions: shorter than their L2 round trip)
  constexpr int PF = 8;
  float4 a[PF];
#pragma unroll
  for (int q = 0; q < PF; ++q) a[q] = wp[(size_t)(kk0 + (q < nkk ? q : nkk - 1)) * 64 + lane];
  float4 b[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) b[nt] = rr_ld4(X + rowoff[nt]);
  for (int k0 = 0; k0 < nkk; k0 += PF) {
    float4 an[PF];
    const bool more = k0 + PF < nkk;
#pragma unroll
    for (int q = 0; q < PF; ++q) { const int kq = k0 + PF + q; an[q] = more ? wp[(size_t)(kk0 + (kq < nkk ? kq : nkk - 1)) * 64 + lane] : a[q]; }
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int kk = k0 + q;
      if (kk < nkk) {
        float4 bn[NT];
        const int kn = kk + 1 < nkk ? kk + 1 : kk;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bn[nt] = rr_ld4(X + rowoff[nt] + kn * 16);
        __builtin_amdgcn_sched_barrier(0);
        const rr_f16x8 af = rr_as_f16x8(a[q]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[nt] = rr_mfma_f16k16(rr_lo4(af), rr_hi4(rr_as_f16x8(b[nt])), acc[nt]);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) accs[nt] = rr_mfma_f16(af, rr_as_f16x8(b[nt]), accs[nt]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) b[nt] = bn[nt];
      }
    }
#pragma unroll
    for (int q = 0; q < PF; ++q) a[q] = an[q];
  }
}
