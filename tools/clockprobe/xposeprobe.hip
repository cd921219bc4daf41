// Diagnostic (not part of the product): the exact transposition of a 16x16 tile of bf16 values by one v_mfma_f32_16x16x16_bf16
// with the identity (csrc/rr_train_dec.hip: td_xpose16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out) {
  const int lane = threadIdx.x, j = lane & 15, g = lane >> 4;
  bf16x4 p, ident;
  for (int m = 0; m < 4; ++m) { p[m] = (__bf16)(float)(j * 16 + 4 * g + m); ident[m] = (__bf16)((j == 4 * g + m) ? 1.0f : 0.f); }
  f32x4 z = {0, 0, 0, 0};
  const f32x4 d = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(p, ident, z, 0, 0, 0);
  const float d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3];     // (__builtin_bit_cast on a vector ELEMENT reads element 0 every time: hipcc 7.2)
  const uint2 o = make_uint2(__builtin_amdgcn_perm(__float_as_uint(d1), __float_as_uint(d0), 0x07060302u),
                             __builtin_amdgcn_perm(__float_as_uint(d3), __float_as_uint(d2), 0x07060302u));
  const bf16x4 t = __builtin_bit_cast(bf16x4, o);
  for (int r = 0; r < 4; ++r) { out[lane * 8 + r] = (float)t[r]; out[lane * 8 + 4 + r] = d[r]; }
}
int main() {
  float* o; hipMalloc(&o, 64 * 8 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
  float h[512]; hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
    const int j = l & 15, g = l >> 4; const float want = (float)((4 * g + r) * 16 + j);
    if (h[l * 8 + r] != want || h[l * 8 + 4 + r] != want) { if (bad < 8) printf("lane %d r %d: got %g / %g want %g\n", l, r, h[l * 8 + r], h[l * 8 + 4 + r], want); ++bad; }
  }
  printf("xpose16: %d mismatches\n", bad);
  return 0;
}
