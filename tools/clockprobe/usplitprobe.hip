// Diagnostic (not part of the product): the second-form two-piece product (rr_common.h: rr_usplit4s / rr_usplit8 + one accumulator)
// against float64: C[i][j] = sum_k A[i][k] B[j][k], K = 16 per k16+k32 pair (as the attention scores) and K = 128 (as the MLP).
#include "../../real-routing-nco_amd/csrc/rr_common.h"
#include <cstdio>
#include <cmath>
#include <vector>
__global__ void k(const float* A, const float* B, float* C16, float* C128) {   // A[16][128], B[16][128] row-major
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  // K = 16 (first 16 columns): A as a stored image [hi | lo] of 2^RR_KS A, B as the register tuple [lo | hi]
  {
    float ax[4], bx[4];
    for (int q = 0; q < 4; ++q) { ax[q] = A[i * 128 + 4 * g + q] * (float)(1 << RR_KS); bx[q] = B[i * 128 + 4 * g + q]; }
    const rr_f16x8 as = rr_usplit4s(ax), bs = rr_usplit4s(bx);
    const rr_f16x8 aimg = rr_cat4(rr_hi4(as), rr_lo4(as));            // [hi | lo]
    f32x4 c = rr_zero4();
    c = rr_mfma_f16k16(rr_lo4(aimg), rr_hi4(bs), c);
#ifdef PADIT
    __builtin_amdgcn_sched_barrier(0);
    asm volatile(PADIT ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
#endif
    c = rr_mfma_f16(aimg, bs, c);
    for (int r = 0; r < 4; ++r) C16[(4 * g + r) * 16 + i] = c[r] / (float)(1 << RR_KS);
    if (lane == 5) { for (int q = 0; q < 8; ++q) { C128[256 + q] = (float)bs[q]; C128[264 + q] = (float)as[q]; } for (int q = 0; q < 4; ++q) { C128[272 + q] = bx[q]; C128[276 + q] = ax[q]; } }
  }
  // K = 128: k = 32 per instruction, permuted k (any permutation, the same for A and B)
  {
    f32x4 c = rr_zero4();
    for (int s = 0; s < 4; ++s) {
      float ax[8], bx[8];
      for (int q = 0; q < 8; ++q) { ax[q] = A[i * 128 + 32 * s + 8 * g + q] * (float)(1 << RR_WS); bx[q] = B[i * 128 + 32 * s + 8 * g + q]; }
      rr_f16x8 ah, al, bh, bl;
      rr_usplit8(ax, ah, al); rr_usplit8(bx, bh, bl);
      c = rr_mfma_f16(ah, bh, c); c = rr_mfma_f16(ah, bl, c); c = rr_mfma_f16(al, bh, c);
    }
    for (int r = 0; r < 4; ++r) C128[(4 * g + r) * 16 + i] = c[r] / (float)(1 << RR_WS);
  }
}
int main() {
  std::vector<float> A(16 * 128), B(16 * 128), c16(256), c128(256);
  srand(5);
  for (auto& v : A) v = (rand() % 20001 - 10000) / 10000.0f * 0.8f;
  for (auto& v : B) v = (rand() % 20001 - 10000) / 10000.0f * 3.0f;
  float *dA, *dB, *d16, *d128;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&d16, 1024); hipMalloc(&d128, 2048);
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, d16, d128);
  hipMemcpy(c16.data(), d16, 1024, hipMemcpyDeviceToHost); hipMemcpy(c128.data(), d128, 1024, hipMemcpyDeviceToHost);
  double e16 = 0, e128 = 0, s16 = 0, s128 = 0;
  for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
    double r16 = 0, r128 = 0, a16 = 0, a128 = 0;
    for (int k2 = 0; k2 < 128; ++k2) { const double p = (double)A[i * 128 + k2] * B[j * 128 + k2]; r128 += p; a128 += std::fabs(p); if (k2 < 16) { r16 += p; a16 += std::fabs(p); } }
    if (j == 0) printf("row %2d col 0: gpu %.5f ref %.5f\n", i, c16[i * 16 + j], r16);
    e16 = std::fmax(e16, std::fabs(c16[i * 16 + j] - r16) / a16); e128 = std::fmax(e128, std::fabs(c128[i * 16 + j] - r128) / a128);
  }
  { std::vector<float> dbg(24); hipMemcpy(dbg.data(), d128 + 256, 96, hipMemcpyDeviceToHost);
    printf("lane 5: b = %g %g %g %g -> [lo|hi] = %g %g %g %g | %g %g %g %g\n", dbg[16], dbg[17], dbg[18], dbg[19], dbg[0], dbg[1], dbg[2], dbg[3], dbg[4], dbg[5], dbg[6], dbg[7]);
    printf("lane 5: a = %g %g %g %g -> [lo|hi] = %g %g %g %g | %g %g %g %g\n", dbg[20], dbg[21], dbg[22], dbg[23], dbg[8], dbg[9], dbg[10], dbg[11], dbg[12], dbg[13], dbg[14], dbg[15]); }
  { double hh = 0, full = 0; for (int k2 = 0; k2 < 16; ++k2) { const double a = A[k2], b = B[k2]; full += a * b; const double ah = (double)(float)(_Float16)(float)(a * 16), bh = (double)(float)(_Float16)(float)b; hh += ah * bh / 16; }
    printf("C[0][0]: gpu %.6f  float64 %.6f  hi*hi only %.6f | C[0][1] gpu %.6f C[1][0] gpu %.6f\n", c16[0], full, hh, c16[1], c16[16]);
    double r01 = 0, r10 = 0; for (int k2 = 0; k2 < 16; ++k2) { r01 += (double)A[k2] * B[128 + k2]; r10 += (double)A[128 + k2] * B[k2]; } printf("ref C[0][1] %.6f C[1][0] %.6f\n", r01, r10); }
  printf("second-form split products vs float64: K = 16 (k16 + k32 pair) worst error / sum|ab| = %.3e; K = 128 (12 k32) = %.3e\n", e16, e128);
  return 0;
}
