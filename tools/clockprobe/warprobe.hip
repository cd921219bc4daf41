// Diagnostic (not part of the product): does gfx950 interlock a vector WRITE to a register that a just-issued
// v_mfma_f32_16x16x32_f16 reads as SrcA / SrcB / SrcC?  Half of the waves of every SIMD run a pure matrix stream (contention), the others issue
//   [K back-to-back matrix instructions on operands of 1.0]  [D filler instructions]  [a write into the last one's SrcA, SrcB or SrcC]
// (v_mov_b32, or v_fma_mixlo_f16 = the partial write of the split's inline asm) and check the last result: 32.0 when the instruction read its
// operands before the write landed.  Measured on MI355X (profiles/r06/NOTES.md section 7): 0 wrong values in all 18 configurations.
//   warprobe [iters]   prints mismatching lanes per (write kind, operand, K, D)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int WHICH, int K, int D, int HOW>
__global__ __launch_bounds__(512) void k_war(unsigned long long* bad, int iters) {
  const int wave = threadIdx.x >> 6;
  f16x8 one, three;
  for (int q = 0; q < 8; ++q) { one[q] = (_Float16)1.0f; three[q] = (_Float16)3.0f; }
  if (wave >= 4) {                       // waves 4 .. 7 share the SIMDs of waves 0 .. 3: a dense matrix stream
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters * 8; ++i) {
      c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(one, one, c0, 0, 0, 0);
      c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(one, one, c1, 0, 0, 0);
      c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(one, one, c2, 0, 0, 0);
      c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(one, one, c3, 0, 0, 0);
      if ((i & 63) == 63) { c0 *= 0.f; c1 *= 0.f; c2 *= 0.f; c3 *= 0.f; }
    }
    if (c0[0] + c1[0] + c2[0] + c3[0] == 12345.f) bad[63] = 1;
    return;
  }
  unsigned long long nbad = 0;
  for (int i = 0; i < iters; ++i) {
    f32x4 r;
    // fixed registers: A = v[100:103], B = v[104:107] (1.0 everywhere), results v[108:123]; the write is a v_mov of 3.0 (two halves) into
    // the FIRST register of the operand (k = 0, 1 of every row / column)
    asm volatile(
        "v_mov_b32 v100, %1\n\t v_mov_b32 v101, %1\n\t v_mov_b32 v102, %1\n\t v_mov_b32 v103, %1\n\t"
        "v_mov_b32 v104, %1\n\t v_mov_b32 v105, %1\n\t v_mov_b32 v106, %1\n\t v_mov_b32 v107, %1\n\t"
        "v_mov_b32 v126, 0\n\t v_mov_b32 v127, 0\n\t v_mov_b32 v128, 0\n\t v_mov_b32 v129, 0\n\t"
        "s_nop 7\n\t"
        "v_mfma_f32_16x16x32_f16 v[108:111], v[100:103], v[104:107], 0\n\t"
        ".if %3 > 1\n\t v_mfma_f32_16x16x32_f16 v[112:115], v[100:103], v[104:107], 0\n\t .endif\n\t"
        ".if %3 > 2\n\t v_mfma_f32_16x16x32_f16 v[116:119], v[100:103], v[104:107], 0\n\t .endif\n\t"
        ".if %3 > 3\n\t .if %5 == 2\n\t v_mfma_f32_16x16x32_f16 v[120:123], v[100:103], v[104:107], v[126:129]\n\t .else\n\t v_mfma_f32_16x16x32_f16 v[120:123], v[100:103], v[104:107], 0\n\t .endif\n\t .endif\n\t"
        ".rept %4\n\t v_mov_b32 v124, v124\n\t .endr\n\t"
        ".if %6 == 0\n\t"
        ".if %5 == 0\n\t v_mov_b32 v100, %2\n\t .endif\n\t .if %5 == 1\n\t v_mov_b32 v104, %2\n\t .endif\n\t .if %5 == 2\n\t v_mov_b32 v126, %2\n\t .endif\n\t"
        ".else\n\t"
        ".if %5 == 0\n\t v_fma_mixlo_f16 v100, %2, 1.0, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t .endif\n\t"
        ".if %5 == 1\n\t v_fma_mixlo_f16 v104, %2, 1.0, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t .endif\n\t"
        ".if %5 == 2\n\t v_fma_mixlo_f16 v126, %2, 1.0, 0 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t .endif\n\t"
        ".endif\n\t"
        "s_nop 15\n\t s_nop 15\n\t"
        ".if %3 == 1\n\t v_mov_b32 %0, v108\n\t .endif\n\t"
        ".if %3 == 2\n\t v_mov_b32 %0, v112\n\t .endif\n\t"
        ".if %3 == 3\n\t v_mov_b32 %0, v116\n\t .endif\n\t"
        ".if %3 == 4\n\t v_mov_b32 %0, v120\n\t .endif"
        : "=v"(r[0])
        : "v"(0x3c003c00u), "v"(HOW ? 0x447a0000u : 0x42004200u), "n"(K), "n"(D), "n"(WHICH), "n"(HOW)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",
          "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129");
    nbad += r[0] != 32.0f;
  }
  if (nbad) atomicAdd(&bad[0], nbad);
}

template <int WHICH, int K, int D, int HOW>
static void run(unsigned long long* d, int iters) {
  hipMemset(d, 0, 512);
  hipLaunchKernelGGL((k_war<WHICH, K, D, HOW>), dim3(1024), dim3(512), 0, 0, d, iters);
  unsigned long long h = 0;
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("%s of Src%c, %d matrix instruction(s) back to back, %d filler(s) before the write: %llu wrong values\n", HOW ? "v_fma_mixlo_f16 (partial write)" : "v_mov_b32", "ABC"[WHICH], K, D, h);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 2000;
  unsigned long long* d; hipMalloc(&d, 512);
  run<0, 1, 0, 0>(d, iters); run<1, 1, 0, 0>(d, iters); run<0, 4, 0, 0>(d, iters); run<1, 4, 0, 0>(d, iters); run<2, 4, 0, 0>(d, iters);
  run<0, 4, 2, 0>(d, iters); run<1, 4, 2, 0>(d, iters); run<2, 4, 2, 0>(d, iters); run<2, 4, 8, 0>(d, iters);
  run<0, 1, 0, 1>(d, iters); run<1, 1, 0, 1>(d, iters); run<0, 4, 0, 1>(d, iters); run<1, 4, 0, 1>(d, iters); run<2, 4, 0, 1>(d, iters);
  run<0, 4, 2, 1>(d, iters); run<1, 4, 2, 1>(d, iters); run<2, 4, 2, 1>(d, iters); run<2, 4, 8, 1>(d, iters);
  return 0;
}
