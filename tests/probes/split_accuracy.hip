// Test probe (compiled by tests/test_gpu_split_accuracy.py with hipcc on the GPU box; not part of the product): the accuracy of the
// two-piece fp16 products the default kernels are built on (csrc/rr_common.h, second form: x~ = 2^s x, hi = fp16(x~), lo = fp16(x~ - hi),
// three partial products into one fp32 accumulator) against float64, beside the fp32 MFMA on the same operands.
// For K in {16, 128, 512} and operand magnitudes 2^-12 .. 2^12 it prints one line
//     K <K> mag <m> split <err> fp32mfma <err> single <err>
// with err = max over a 16 x 16 tile of |C - float64| / sum_k |a b|;  "single" = one fp16 piece per operand (the 16-mixed variant).
// K = 16 uses the k16 + k32 instruction pair of the attention scores / P.V / logits (operand image [hi | lo], register tuple [lo | hi]);
// K >= 128 the k = 32 form of the MLP (separate hi and lo fragments).  The weight-like operand A carries the image scale (2^RR_KS for
// K = 16, 2^RR_WS otherwise) exactly as the kernels' images do; below |x~| = 2^-3 the lo piece is a subnormal fp16 number — the
// small-magnitude rows of the sweep probe exactly that.
#include "../../real-routing-nco_amd/csrc/rr_common.h"
#include <cstdio>
#include <cmath>
#include <vector>

__global__ void k_probe(const float* A, const float* B, float* C, int K) {   // A[16][K], B[16][K] row-major; C[3][16][16]
  const int lane = threadIdx.x, i = lane & 15, g = lane >> 4;
  f32x4 cs = rr_zero4(), cf = rr_zero4(), ch = rr_zero4();
  if (K == 16) {
    float ax[4], bx[4];
    for (int q = 0; q < 4; ++q) { ax[q] = A[i * K + 4 * g + q] * (float)(1 << RR_KS); bx[q] = B[i * K + 4 * g + q]; }
    const rr_f16x8 as = rr_usplit4s(ax), bs = rr_usplit4s(bx);
    const rr_f16x8 aimg = rr_cat4(rr_hi4(as), rr_lo4(as));            // stored image [hi | lo]
    f32x4 c0 = rr_mfma_f16k16(rr_lo4(aimg), rr_hi4(bs), rr_zero4());   // (two chains: the k = 32 instruction never continues the k = 16 one)
    f32x4 c1 = rr_mfma_f16(aimg, bs, rr_zero4());
    for (int r = 0; r < 4; ++r) { cs[r] = (c0[r] + c1[r]) / (float)(1 << RR_KS); ch[r] = c0[r] / (float)(1 << RR_KS); }
    for (int q = 0; q < 4; ++q) cf = rr_mfma(A[i * K + 4 * g + q], B[i * K + 4 * g + q], cf);
  } else {
    for (int s = 0; s < K / 32; ++s) {
      float ax[8], bx[8];
      for (int q = 0; q < 8; ++q) { ax[q] = A[i * K + 32 * s + 8 * g + q] * (float)(1 << RR_WS); bx[q] = B[i * K + 32 * s + 8 * g + q]; }
      rr_f16x8 ah, al, bh, bl;
      rr_usplit8(ax, ah, al); rr_usplit8(bx, bh, bl);
      cs = rr_mfma_f16(ah, bh, cs); cs = rr_mfma_f16(ah, bl, cs); cs = rr_mfma_f16(al, bh, cs);
      ch = rr_mfma_f16(ah, bh, ch);
      for (int q = 0; q < 8; ++q) cf = rr_mfma(A[i * K + 32 * s + 8 * g + q], B[i * K + 32 * s + 8 * g + q], cf);
    }
    for (int r = 0; r < 4; ++r) { cs[r] /= (float)(1 << RR_WS); ch[r] /= (float)(1 << RR_WS); }
  }
  for (int r = 0; r < 4; ++r) { const int o = (4 * g + r) * 16 + i; C[o] = cs[r]; C[256 + o] = cf[r]; C[512 + o] = ch[r]; }
}

int main() {
  float *dA, *dB, *dC; hipMalloc(&dA, 16 * 512 * 4); hipMalloc(&dB, 16 * 512 * 4); hipMalloc(&dC, 768 * 4);
  for (int K : {16, 128, 512}) {
    for (int e = -12; e <= 12; e += 3) {
      // B: activation-like operand of magnitude 2^e (the kernels leave activations unscaled); A: weight-like, |a| <= 1 (scaled by its image scale)
      std::vector<float> A(16 * K), B(16 * K);
      srand(11 + K + e);
      for (auto& v : A) v = (float)((2.0 * rand() / RAND_MAX - 1.0) * (rand() % 5 == 0 ? 1e-3 : 1.0));
      for (auto& v : B) v = (float)(std::ldexp(2.0 * rand() / RAND_MAX - 1.0, e));
      hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
      hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, K);
      std::vector<float> C(768); hipMemcpy(C.data(), dC, 768 * 4, hipMemcpyDeviceToHost);
      double err[3] = {0, 0, 0};
      for (int i = 0; i < 16; ++i) for (int j = 0; j < 16; ++j) {
        double r = 0, ra = 0;
        for (int k = 0; k < K; ++k) { const double p = (double)A[i * K + k] * B[j * K + k]; r += p; ra += std::fabs(p); }
        for (int m = 0; m < 3; ++m) err[m] = std::fmax(err[m], std::fabs(C[m * 256 + i * 16 + j] - r) / ra);
      }
      printf("K %d mag %d split %.4e fp32mfma %.4e single %.4e\n", K, e, err[0], err[1], err[2]);
    }
  }
  return 0;
}
