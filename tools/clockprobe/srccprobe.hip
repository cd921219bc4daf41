// Diagnostic (not part of the product): is an LDS load into the SrcC register of a just-issued, still queued matrix instruction interlocked?
//   chain of K dependent v_mfma_f32_16x16x32_f16 (each link writes a NEW register: c1 = A B + c0, c2 = A B + c1, ...), D filler
//   instructions, then ds_read_b128 INTO the register that the LAST link reads as SrcC; the last link's result must be 32 K.
// Half of the waves of every SIMD run a dense matrix stream (contention).  srccprobe [iters]
// Measured on MI355X (profiles/r06/NOTES.md section 7): 0 wrong values of 1.3e8 in all 8 configurations — the load IS interlocked.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int K, int D>
__global__ __launch_bounds__(512) void k_srcc(unsigned long long* bad, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[512 * 4];
  for (int i = threadIdx.x; i < 2048; i += 512) lds[i] = 1000.0f;          // what the load brings: 1000 in every component
  __syncthreads();
  const int wave = threadIdx.x >> 6;
  f16x8 one;
  for (int q = 0; q < 8; ++q) one[q] = (_Float16)1.0f;
  if (wave >= 4) {
    f32x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    for (int i = 0; i < iters * 6; ++i) {
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
  const unsigned addr = (unsigned)(size_t)(&lds[(threadIdx.x & 63) * 4]) ;       // LDS byte address of this lane's 16 bytes (low 32 bits of the generic pointer's LDS offset)
  for (int i = 0; i < iters; ++i) {
    float r;
    // fixed registers: A = B = v[100:103] (1.0), chain registers v[104:107] (zero), v[108:111], v[112:115], v[116:119], v[120:123]
    asm volatile(
        "v_mov_b32 v100, %1\n\t v_mov_b32 v101, %1\n\t v_mov_b32 v102, %1\n\t v_mov_b32 v103, %1\n\t"
        "v_mov_b32 v104, 0\n\t v_mov_b32 v105, 0\n\t v_mov_b32 v106, 0\n\t v_mov_b32 v107, 0\n\t"
        "s_nop 7\n\t"
        "v_mfma_f32_16x16x32_f16 v[108:111], v[100:103], v[100:103], v[104:107]\n\t"
        ".if %3 > 1\n\t v_mfma_f32_16x16x32_f16 v[112:115], v[100:103], v[100:103], v[108:111]\n\t .endif\n\t"
        ".if %3 > 2\n\t v_mfma_f32_16x16x32_f16 v[116:119], v[100:103], v[100:103], v[112:115]\n\t .endif\n\t"
        ".if %3 > 3\n\t v_mfma_f32_16x16x32_f16 v[120:123], v[100:103], v[100:103], v[116:119]\n\t .endif\n\t"
        ".rept %4\n\t v_mov_b32 v124, v124\n\t .endr\n\t"
        ".if %3 == 1\n\t ds_read_b128 v[104:107], %2\n\t .endif\n\t"
        ".if %3 == 2\n\t ds_read_b128 v[108:111], %2\n\t .endif\n\t"
        ".if %3 == 3\n\t ds_read_b128 v[112:115], %2\n\t .endif\n\t"
        ".if %3 == 4\n\t ds_read_b128 v[116:119], %2\n\t .endif\n\t"
        "s_waitcnt lgkmcnt(0)\n\t s_nop 15\n\t s_nop 15\n\t s_nop 15\n\t"
        ".if %3 == 1\n\t v_mov_b32 %0, v108\n\t .endif\n\t"
        ".if %3 == 2\n\t v_mov_b32 %0, v112\n\t .endif\n\t"
        ".if %3 == 3\n\t v_mov_b32 %0, v116\n\t .endif\n\t"
        ".if %3 == 4\n\t v_mov_b32 %0, v120\n\t .endif"
        : "=v"(r)
        : "v"(0x3c003c00u), "v"(addr), "n"(K), "n"(D)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115",
          "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "memory");
    nbad += r != 32.0f * K;
  }
  if (nbad) atomicAdd(&bad[0], nbad);
}

template <int K, int D>
static void run(unsigned long long* d, int iters) {
  hipMemset(d, 0, 512);
  hipLaunchKernelGGL((k_srcc<K, D>), dim3(1024), dim3(512), 0, 0, d, iters);
  unsigned long long h = 0;
  hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
  printf("chain of %d dependent matrix instruction(s), %d filler(s), then ds_read_b128 into the last link's SrcC: %llu wrong values of %d\n", K, D, h,
         1024 * 256 * iters);
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 500;
  unsigned long long* d; hipMalloc(&d, 512);
  run<1, 0>(d, iters); run<2, 0>(d, iters); run<3, 0>(d, iters); run<4, 0>(d, iters);
  run<4, 4>(d, iters); run<4, 8>(d, iters); run<4, 16>(d, iters); run<4, 32>(d, iters);
  return 0;
}
