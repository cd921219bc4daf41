// Shared device helpers for the RRNCO construction-rollout kernels (gfx950 / CDNA4 only).
//
// Numeric layout convention used by every MFMA kernel in this directory ("transposed tiles"):
//   we always compute  Y^T = W * X^T  with v_mfma_f32_16x16x4_f32, i.e.
//     A operand  = a 16(features) x 4(k) slice of a weight-like matrix      lane l: A[i = l&15][k = l>>4]
//     B operand  = a 4(k) x 16(nodes|rollouts) slice of an activation^T      lane l: B[k = l>>4][j = l&15]
//     C/D        = 16(features) x 16(nodes) tile                              lane l: D[row = 4*(l>>4)+reg][col = l&15]
//   so "which node / rollout" always lives on the low 4 lane bits (j) and the feature index lives on
//   (g = lane>>4, reg).  A lane therefore owns 4 CONSECUTIVE features of one node, which is exactly one
//   float4 of a row-major [node][feature] activation tile: C tiles are written back to LDS with one
//   ds_write_b128, and re-read as the next GEMM's B operand with one ds_read_b128 per 4 MFMAs.
//   The K index is consumed in the permuted order  k = 16*kk + 4*g + m  (m = 0..3 = the 4 MFMAs fed by one
//   float4); A and B use the same permutation so the sum is over every k exactly once.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define RR_WAVE 64
#define RR_E 128          // embed dim (fixed by the architecture: configs/experiment/rrnet.yaml:22)
#define RR_FF 512         // feed-forward hidden (encoder FFN and decoder pointer MLP)
#define RR_HEADS 8
#define RR_HD 16          // head dim
#define RR_MAXN 103       // max nodes for the LDS-resident per-instance kernels
#define RR_NT 7           // node / rollout tiles of 16 per workgroup (covers up to 112)

#define RR_OK 0
#define RR_EINVAL -1
#define RR_ELAUNCH -2

__device__ __forceinline__ f32x4 rr_mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

__device__ __forceinline__ f32x4 rr_zero4() { f32x4 z = {0.f, 0.f, 0.f, 0.f}; return z; }

__device__ __forceinline__ float4 rr_ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void rr_st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// acc[nt] (features 16t.., nodes 16nt..) += Wp(tile t, k-groups [kk0,kk0+nkk)) * X^T
//   Wp   : packed A operand, float4 index [(t*KK + kk)*64 + lane]  (see rrnco_amd/packing.py)
//   X    : LDS (or global) row-major activation [node][ldx]; rows clamped to n_valid-1
//   xk0  : column offset in X of k-group kk0 (X may be a chunk of the full K range)
template <int NT>
__device__ __forceinline__ void rr_gemm_wx(f32x4 (&acc)[NT], const float4* __restrict__ wp, int kk0, int nkk,
                                           const float* X, int ldx, int xk0, int n_valid, int lane) {
  const int j = lane & 15, g = lane >> 4;
  int rowoff[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int node = nt * 16 + j;
    node = node < n_valid ? node : n_valid - 1;
    rowoff[nt] = node * ldx + 4 * g + xk0;
  }
  float4 a = wp[(size_t)kk0 * 64 + lane];
  float4 b[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) b[nt] = rr_ld4(X + rowoff[nt]);
#pragma unroll 1
  for (int kk = 0; kk < nkk; ++kk) {
    // operands of the next k-group (A from L2, B from LDS) are requested before this group's 4*NT MFMAs issue
    float4 an = a, bn[NT];
    const int kn = kk + 1 < nkk ? kk + 1 : kk;
    an = wp[(size_t)(kk0 + kn) * 64 + lane];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) bn[nt] = rr_ld4(X + rowoff[nt] + kn * 16);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = rr_mfma(a.x, b[nt].x, acc[nt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = rr_mfma(a.y, b[nt].y, acc[nt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = rr_mfma(a.z, b[nt].z, acc[nt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = rr_mfma(a.w, b[nt].w, acc[nt]);
    __builtin_amdgcn_sched_barrier(0);
    a = an;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) b[nt] = bn[nt];
  }
}

// ---- raw buffer loads: 128-bit SGPR descriptor + 32-bit per-lane byte offset + SCALAR byte offset.  The uniform part of
// every operand address (fragment index, head, k-group) then advances with SALU adds and costs no VALU slot — which matters
// because the fp32 MFMA shares the SIMD's fp32 datapath with the VALU (measured: 2 waves/SIMD gain only 11 % over 1).
typedef unsigned int rr_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rr_make_buf(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
__device__ __forceinline__ float4 rr_bld4(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
  rr_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff_bytes, soff_bytes, 0);
  return make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
}
__device__ __forceinline__ float2 rr_bld2(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
  typedef unsigned rr_u32x2_ __attribute__((ext_vector_type(2)));
  const rr_u32x2_ v = __builtin_amdgcn_raw_buffer_load_b64(r, voff_bytes, soff_bytes, 0);   // out of range -> 0
  return make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));
}
__device__ __forceinline__ float rr_bld1(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
  return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff_bytes, soff_bytes, 0));   // out of range -> 0
}
// ---- bf16 matrix pipe with 3-way split fp32 operands (opt-in pointer MLP of the rollout, RR_MLP_SPLIT=1)
typedef __bf16 rr_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 rr_bf16x2 __attribute__((ext_vector_type(2)));
typedef float rr_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x4 rr_mfma_bf16(rr_bf16x8 a, rr_bf16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ rr_bf16x8 rr_bldh(__amdgpu_buffer_rsrc_t r, unsigned voff_bytes, unsigned soff_bytes) {
  rr_u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff_bytes, soff_bytes, 0);
  return __builtin_bit_cast(rr_bf16x8, v);
}
// ---- fp16 matrix pipe with two-piece split fp32 operands: x = hi + 2^-11 lo', hi = fp16(x), lo' = fp16(2^11 (x - hi)).
// x - hi is exact in fp32 and lo' keeps 11 of its <= 13 bits, so |x - hi - 2^-11 lo'| <= 2^-23 |x| (down to 2^-36 absolute: the
// f16 MFMA keeps subnormal inputs, tools/clockprobe/f16probe.hip); a product keeps hi*hi + 2^-11 (hi*lo' + lo'*hi), each term
// exact in the fp32 accumulator, the dropped lo'*lo' term is <= 2^-22 |x w|.  Measured error of a K = 128 .. 512 dot product:
// 4e-8 of sum |a b|, against 1.1e-7 .. 1.6e-7 for v_mfma_f32_16x16x4_f32 (same probe) — three MFMAs at 16x the fp32 rate.
typedef _Float16 rr_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 rr_f16x2 __attribute__((ext_vector_type(2)));
#define RR_LO_SCALE 2048.0f
#define RR_LO_INV (1.0f / 2048.0f)
__device__ __forceinline__ f32x4 rr_mfma_f16(rr_f16x8 a, rr_f16x8 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ void rr_split8h(const float (&x)[8], rr_f16x8& hi, rr_f16x8& lo) {
#ifdef RR_KO_SPLIT   // diagnostic knock-out: the price of the split's vector instructions (results are wrong)
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const rr_f32x2 v = {x[2 * q], x[2 * q + 1]};
    const rr_f16x2 h = __builtin_convertvector(v, rr_f16x2);
    hi[2 * q] = h[0]; hi[2 * q + 1] = h[1]; lo[2 * q] = h[0]; lo[2 * q + 1] = h[1];
  }
  return;
#endif
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const rr_f32x2 v = {x[2 * q], x[2 * q + 1]};
    const rr_f16x2 h = __builtin_convertvector(v, rr_f16x2);
    const rr_f32x2 r = (v - __builtin_convertvector(h, rr_f32x2)) * RR_LO_SCALE;
    const rr_f16x2 l = __builtin_convertvector(r, rr_f16x2);
    hi[2 * q] = h[0]; hi[2 * q + 1] = h[1]; lo[2 * q] = l[0]; lo[2 * q + 1] = l[1];
  }
}
typedef _Float16 rr_f16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void rr_split4h(const float (&x)[4], rr_f16x4& hi, rr_f16x4& lo) {
#ifdef RR_KO_SPLIT
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const rr_f32x2 v = {x[2 * q], x[2 * q + 1]};
    const rr_f16x2 h = __builtin_convertvector(v, rr_f16x2);
    hi[2 * q] = h[0]; hi[2 * q + 1] = h[1]; lo[2 * q] = h[0]; lo[2 * q + 1] = h[1];
  }
  return;
#endif
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const rr_f32x2 v = {x[2 * q], x[2 * q + 1]};
    const rr_f16x2 h = __builtin_convertvector(v, rr_f16x2);
    const rr_f32x2 r = (v - __builtin_convertvector(h, rr_f32x2)) * RR_LO_SCALE;
    const rr_f16x2 l = __builtin_convertvector(r, rr_f16x2);
    hi[2 * q] = h[0]; hi[2 * q + 1] = h[1]; lo[2 * q] = l[0]; lo[2 * q + 1] = l[1];
  }
}
// A k = 16 product on two-piece operands takes two instructions.  With A = [hi | lo'] of one operand in ONE 16-byte fragment
// (the footprint of its four fp32 values) and the other operand kept as the tuple S = [lo' | hi]:
//   large term:  v_mfma_f32_16x16x16_f16(A.hi, S.hi)            (the low 8 bytes of A, the high 8 bytes of S)
//   small terms: v_mfma_f32_16x16x32_f16(A, S) = hi*lo' + lo'*hi (scaled 2^11)
// so no operand is ever assembled by register moves.  (Both instructions take 16 cycles: tools/clockprobe/f16probe.hip.)
__device__ __forceinline__ rr_f16x8 rr_cat4(rr_f16x4 a, rr_f16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ rr_f16x4 rr_lo4(rr_f16x8 v) { return __builtin_shufflevector(v, v, 0, 1, 2, 3); }
__device__ __forceinline__ rr_f16x4 rr_hi4(rr_f16x8 v) { return __builtin_shufflevector(v, v, 4, 5, 6, 7); }
__device__ __forceinline__ rr_f16x8 rr_as_f16x8(float4 v) { return __builtin_bit_cast(rr_f16x8, v); }
// one fp16 piece per value (the 16-mixed rollout, rr_rollout_w.inc HALF): v_cvt_pk_f16_f32, round to nearest even
__device__ __forceinline__ rr_f16x4 rr_cvt4h(const float (&x)[4]) {
  typedef float f4_ __attribute__((ext_vector_type(4)));
  const f4_ v = {x[0], x[1], x[2], x[3]};
  return __builtin_convertvector(v, rr_f16x4);
}
__device__ __forceinline__ rr_f16x8 rr_cvt8h(const float (&x)[8]) {
  typedef float f8_ __attribute__((ext_vector_type(8)));
  const f8_ v = {x[0], x[1], x[2], x[3], x[4], x[5], x[6], x[7]};
  return __builtin_convertvector(v, rr_f16x8);
}
__device__ __forceinline__ f32x4 rr_mfma_f16k16(rr_f16x4 a, rr_f16x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
}
// x[0..3] -> the tuple [lo' | hi]
__device__ __forceinline__ rr_f16x8 rr_split4s(const float (&x)[4]) {
  rr_f16x4 hi, lo;
  rr_split4h(x, hi, lo);
  return rr_cat4(lo, hi);
}
// ---- Second form of the two-piece operands (the rollout kernel): NO scale between the pieces.  x~ = 2^s x (s = 0 for
// activations, RR_WS for MLP weights, RR_KS for the K / V^T / L images: exact), hi = fp16(x~), lo = fp16(x~ - hi).  All three
// partial products hi*hi + hi*lo + lo*hi then have the SAME scale 2^(sa + sb) and go into ONE fp32 accumulator: no second
// accumulator, no 2^-11 recombination, and the scale folds into a constant the consumer multiplies by anyway.  x~ - hi is exact in
// fp32 and lo keeps 11 of its bits as long as it is a normal fp16 number (|x~| >= 2^-3); below that lo is a subnormal with an
// absolute error <= 2^-25 — which is why weights (|w| ~ 1e-2) are pre-scaled and O(1) activations need not be.  The split is
// v_cvt_pk_f16_f32 for two hi halves + one v_fma_mixlo/hi_f16 per lo half (fp32 fma on the fp16 half, rounded into the half of
// the destination): 1.5 instructions per value against 3 for the scaled form (cvt, cvt back, subtract, scale, cvt).
// Measured (tools/clockprobe/mixsplit): |hi + lo - x| <= 2^-22 |x| for |x| >= 1/8, <= 2^-25 absolute below.
#define RR_WS 6            // log2 scale of the MLP weight images W1s / W2s (packing.pack_a_f16u)
#define RR_KS 4            // log2 scale of the K / V^T / L images (rr_pack_f16x2)
#define RR_F16_LIMIT 65504.0f
// The lo halves of 2 / 4 value pairs in ONE asm block.  hipcc cannot see that an asm statement is a VALU write, so it would not
// keep the 2 wait states gfx950 needs between a VALU write of a register and a matrix instruction reading it as SrcA / SrcB
// (it pads its own v_cvt_pk_f16_f32 -> v_mfma with s_nop 1); without them the MFMA reads the register's OLD contents — seen as
// garbage attention outputs and run-to-run differences.  The block ends with that s_nop 1; the lo / hi writes of one
// destination sit four (two) instructions apart.
// (diagnostics for profiles/r06/NOTES.md section 7: -DRR_ULO_VOLATILE pins the block where the source has it, -DRR_ULO_TAIL="s_nop 7" lengthens
// the wait behind its last write)
#ifdef RR_ULO_VOLATILE
#define RR_ULO_ASM asm volatile
#else
#define RR_ULO_ASM asm
#endif
#ifndef RR_ULO_TAIL
#define RR_ULO_TAIL "s_nop 1"
#endif
__device__ __forceinline__ void rr_ulo8(const rr_f16x2 (&h)[4], const float (&x)[8], rr_f16x2 (&l)[4]) {
  RR_ULO_ASM("s_nop 0\n\t"                     // (the hi halves come from a v_cvt_pk right before: hipcc pads its own VALU pairs where gfx950 needs it, not ours)
      "v_fma_mixlo_f16 %0, %4, -1.0, %8 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %1, %5, -1.0, %10 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %2, %6, -1.0, %12 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %3, %7, -1.0, %14 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %4, -1.0, %9 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %1, %5, -1.0, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %2, %6, -1.0, %13 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %3, %7, -1.0, %15 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      RR_ULO_TAIL
      : "=&v"(l[0]), "=&v"(l[1]), "=&v"(l[2]), "=&v"(l[3])
      : "v"(h[0]), "v"(h[1]), "v"(h[2]), "v"(h[3]), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]), "v"(x[4]), "v"(x[5]), "v"(x[6]), "v"(x[7]));
}
__device__ __forceinline__ void rr_ulo4(const rr_f16x2 (&h)[2], const float (&x)[4], rr_f16x2 (&l)[2]) {
  asm("s_nop 0\n\t"
      "v_fma_mixlo_f16 %0, %2, -1.0, %4 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixlo_f16 %1, %3, -1.0, %6 op_sel:[0,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %0, %2, -1.0, %5 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "v_fma_mixhi_f16 %1, %3, -1.0, %7 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
      "s_nop 1"
      : "=&v"(l[0]), "=&v"(l[1])
      : "v"(h[0]), "v"(h[1]), "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]));
}
__device__ __forceinline__ void rr_usplit8(const float (&x)[8], rr_f16x8& hi, rr_f16x8& lo) {
  rr_f16x2 h[4], l[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { const rr_f32x2 v = {x[2 * q], x[2 * q + 1]}; h[q] = __builtin_convertvector(v, rr_f16x2); }
  rr_ulo8(h, x, l);
#pragma unroll
  for (int q = 0; q < 4; ++q) { hi[2 * q] = h[q][0]; hi[2 * q + 1] = h[q][1]; lo[2 * q] = l[q][0]; lo[2 * q + 1] = l[q][1]; }
}
// x[0..3] -> the tuple [lo | hi] (the register operand of a k = 16 product pair, see rr_split4s)
__device__ __forceinline__ rr_f16x8 rr_usplit4s(const float (&x)[4]) {
  rr_f16x2 h[2], l[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) { const rr_f32x2 v = {x[2 * q], x[2 * q + 1]}; h[q] = __builtin_convertvector(v, rr_f16x2); }
  rr_ulo4(h, x, l);
  rr_f16x8 r;
#pragma unroll
  for (int q = 0; q < 2; ++q) { r[2 * q] = l[q][0]; r[2 * q + 1] = l[q][1]; r[4 + 2 * q] = h[q][0]; r[4 + 2 * q + 1] = h[q][1]; }
  return r;
}
// MI355X, ROCm 7.2: a VALU write to a source register of a v_mfma_f32_16x16x32_f16 issued just before it can reach the
// register file before the MFMA has read it when the other wave of the SIMD keeps the matrix pipe busy (seen in round 2 as
// run-to-run differences of the rollout's P.V, whose B operand is rebuilt for every key tile; hipcc inserts no wait states for
// this write-after-read).  The rollout keeps a rebuilt operand's registers allocated until the NEXT pair has issued
// (rr_rollout_w.inc, P.V: software-pipelined tuples) instead of idling an instruction time behind every pair.
// LDS-DMA: 16 bytes per lane from global memory straight into LDS at ldst + 16 * lane (ldst wave-uniform); completes on vmcnt
__device__ __forceinline__ void rr_glds16(const void* gsrc, void* ldst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
}
// max(x, 0) as ONE instruction: fmaxf() on a matrix-instruction result costs two (hipcc canonicalises the operand first:
// v_max_f32 x, x).  The integer maximum of the bit pattern with 0 is the same function for every non-NaN float (negative floats
// and -0 are negative integers) and stays visible to the hazard recognizer — an `asm("v_max_f32")` here is not: it read matrix
// results before they were written (tours changing from call to call, test_full_size_properties_n100_b64_aug8).
__device__ __forceinline__ float rr_relu(float x) {
#ifdef RR_RELU_FMAXF
  return fmaxf(x, 0.f);
#endif
  const int i = __float_as_int(x);
  return __int_as_float(i > 0 ? i : 0);
}
// ---- Hand-scheduled half regions of the pointer MLP (rr_rollout_w.inc, instance mode; round 6).  A half = 12 v_mfma_f32_16x16x32_f16 on
// the eight weight fragments Xp a wave read EARLIER, and beside them the eight ds_read_b128 of the fragments Xn the NEXT half needs: one
// read behind each of the first eight matrix instructions, none waited for before the end of the half.  hipcc's own schedule of the same
// work sinks every read to its use (an `s_waitcnt lgkmcnt` that exposes an LDS round trip in front of most matrix instructions:
// 1 250 cycles per region in the kernel, 1 136 in tools/clockprobe/mlpprobe.hip); this order measured 880 cycles per region for seven
// waves in tools/clockprobe/pipeprobe.hip (profiles/r06/NOTES.md §1).  Same products into the same accumulators in the same order as
// hid() / output4(): results are bit-identical.  Inline asm is invisible to hipcc's hazard recognizer and to its waitcnt pass, so each
// block (1) starts with s_nop 1 (a VALU write of a source right before it), (2) its reads are waited for (lgkmcnt(0)) before anything
// multiplies by them: a region's FIRST half leaves them in flight across the region's vector work (nothing touches Xn there: checked in
// the listing) and the SECOND half (SECOND = true) opens with the wait and closes with the wait for its own, in front of the region's barrier,
// (3) must not be followed within a few cycles by a VALU read of its accumulators: callers keep a sched_barrier + other work behind it.
// rr_half_hid: c = seed + sum_sl (Xp[2 sl] Gs[sl][0] + Xp[2 sl] Gs[sl][1] + Xp[2 sl + 1] Gs[sl][0])      (one dependent chain)
template <bool SECOND>
__device__ __forceinline__ void rr_half_hid(f32x4& c, f32x4 seed, const rr_f16x8 (&Xp)[8], rr_f16x8 (&Xn)[8], const rr_f16x8 (&Gs)[4][2], unsigned addr) {
  if constexpr (SECOND) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the first half's reads (its Xn = this half's Xp) have landed
  asm volatile(
      "s_nop 1\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %10, %18, %9\n\t"
      "ds_read_b128 %1, %26 offset:0\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %10, %19, %0\n\t"
      "ds_read_b128 %2, %26 offset:1024\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %11, %18, %0\n\t"
      "ds_read_b128 %3, %26 offset:2048\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %12, %20, %0\n\t"
      "ds_read_b128 %4, %26 offset:3072\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %12, %21, %0\n\t"
      "ds_read_b128 %5, %26 offset:4096\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %13, %20, %0\n\t"
      "ds_read_b128 %6, %26 offset:5120\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %14, %22, %0\n\t"
      "ds_read_b128 %7, %26 offset:6144\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %14, %23, %0\n\t"
      "ds_read_b128 %8, %26 offset:7168\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %15, %22, %0\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %16, %24, %0\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %16, %25, %0\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %17, %24, %0\n\t"
      "s_nop 0"
      : "=&v"(c), "=&v"(Xn[0]), "=&v"(Xn[1]), "=&v"(Xn[2]), "=&v"(Xn[3]), "=&v"(Xn[4]), "=&v"(Xn[5]), "=&v"(Xn[6]), "=&v"(Xn[7])
      : "v"(seed), "v"(Xp[0]), "v"(Xp[1]), "v"(Xp[2]), "v"(Xp[3]), "v"(Xp[4]), "v"(Xp[5]), "v"(Xp[6]), "v"(Xp[7]),
        "v"(Gs[0][0]), "v"(Gs[0][1]), "v"(Gs[1][0]), "v"(Gs[1][1]), "v"(Gs[2][0]), "v"(Gs[2][1]), "v"(Gs[3][0]), "v"(Gs[3][1]), "v"(addr)
      : "memory");
#ifdef RR_HALF_WAIT_END      // diagnostic: the first half waits for its reads at its own end as well (the first form of these blocks)
  if constexpr (!SECOND) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  if constexpr (SECOND) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the region's barrier follows: the next region multiplies by Xn)
}
// rr_half_hid_inplace: the same with c carrying the seed in (4 registers fewer; k_enc_tail is at the register limit)
template <bool SECOND>
__device__ __forceinline__ void rr_half_hid_inplace(f32x4& c, const rr_f16x8 (&Xp)[8], rr_f16x8 (&Xn)[8], const rr_f16x8 (&Gs)[4][2], unsigned addr) {
  if constexpr (SECOND) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the first half's reads (its Xn = this half's Xp) have landed
  asm volatile(
      "s_nop 1\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %10, %18, %0\n\t"
      "ds_read_b128 %1, %26 offset:0\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %10, %19, %0\n\t"
      "ds_read_b128 %2, %26 offset:1024\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %11, %18, %0\n\t"
      "ds_read_b128 %3, %26 offset:2048\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %12, %20, %0\n\t"
      "ds_read_b128 %4, %26 offset:3072\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %12, %21, %0\n\t"
      "ds_read_b128 %5, %26 offset:4096\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %13, %20, %0\n\t"
      "ds_read_b128 %6, %26 offset:5120\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %14, %22, %0\n\t"
      "ds_read_b128 %7, %26 offset:6144\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %14, %23, %0\n\t"
      "ds_read_b128 %8, %26 offset:7168\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %15, %22, %0\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %16, %24, %0\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %16, %25, %0\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %17, %24, %0\n\t"
      "s_nop 0"
      : "+v"(c), "=&v"(Xn[0]), "=&v"(Xn[1]), "=&v"(Xn[2]), "=&v"(Xn[3]), "=&v"(Xn[4]), "=&v"(Xn[5]), "=&v"(Xn[6]), "=&v"(Xn[7])
      : "v"(c), "v"(Xp[0]), "v"(Xp[1]), "v"(Xp[2]), "v"(Xp[3]), "v"(Xp[4]), "v"(Xp[5]), "v"(Xp[6]), "v"(Xp[7]),
        "v"(Gs[0][0]), "v"(Gs[0][1]), "v"(Gs[1][0]), "v"(Gs[1][1]), "v"(Gs[2][0]), "v"(Gs[2][1]), "v"(Gs[3][0]), "v"(Gs[3][1]), "v"(addr)
      : "memory");
#ifdef RR_HALF_WAIT_END      // diagnostic: the first half waits for its reads at its own end as well (the first form of these blocks)
  if constexpr (!SECOND) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  if constexpr (SECOND) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the region's barrier follows: the next region multiplies by Xn)
}
// rr_half_out: F[u] += Xp[2 u] Hh + Xp[2 u] Hl + Xp[2 u + 1] Hh, u = 0 .. 3      (four chains, each product pass over the four before the next)
template <bool SECOND>
__device__ __forceinline__ void rr_half_out(f32x4& F0, f32x4& F1, f32x4& F2, f32x4& F3, const rr_f16x8 (&Xp)[8], rr_f16x8 (&Xn)[8],
                                            rr_f16x8 Hh, rr_f16x8 Hl, unsigned addr) {
  if constexpr (SECOND) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  asm volatile(
      "s_nop 1\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %12, %20, %0\n\t"
      "ds_read_b128 %4, %22 offset:0\n\t"
      "v_mfma_f32_16x16x32_f16 %1, %14, %20, %1\n\t"
      "ds_read_b128 %5, %22 offset:1024\n\t"
      "v_mfma_f32_16x16x32_f16 %2, %16, %20, %2\n\t"
      "ds_read_b128 %6, %22 offset:2048\n\t"
      "v_mfma_f32_16x16x32_f16 %3, %18, %20, %3\n\t"
      "ds_read_b128 %7, %22 offset:3072\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %12, %21, %0\n\t"
      "ds_read_b128 %8, %22 offset:4096\n\t"
      "v_mfma_f32_16x16x32_f16 %1, %14, %21, %1\n\t"
      "ds_read_b128 %9, %22 offset:5120\n\t"
      "v_mfma_f32_16x16x32_f16 %2, %16, %21, %2\n\t"
      "ds_read_b128 %10, %22 offset:6144\n\t"
      "v_mfma_f32_16x16x32_f16 %3, %18, %21, %3\n\t"
      "ds_read_b128 %11, %22 offset:7168\n\t"
      "v_mfma_f32_16x16x32_f16 %0, %13, %20, %0\n\t"
      "v_mfma_f32_16x16x32_f16 %1, %15, %20, %1\n\t"
      "v_mfma_f32_16x16x32_f16 %2, %17, %20, %2\n\t"
      "v_mfma_f32_16x16x32_f16 %3, %19, %20, %3\n\t"
      "s_nop 0"
      : "+v"(F0), "+v"(F1), "+v"(F2), "+v"(F3), "=&v"(Xn[0]), "=&v"(Xn[1]), "=&v"(Xn[2]), "=&v"(Xn[3]), "=&v"(Xn[4]), "=&v"(Xn[5]), "=&v"(Xn[6]), "=&v"(Xn[7])
      : "v"(Xp[0]), "v"(Xp[1]), "v"(Xp[2]), "v"(Xp[3]), "v"(Xp[4]), "v"(Xp[5]), "v"(Xp[6]), "v"(Xp[7]), "v"(Hh), "v"(Hl), "v"(addr)
      : "memory");
#ifdef RR_HALF_WAIT_END
  if constexpr (!SECOND) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  if constexpr (SECOND) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}
// One 16 KB weight stage = 16 LDS-DMA requests of 1 KB to consecutive LDS slots.  What a request costs its wave is what its
// instructions cost: with per-fragment pointer arithmetic (a 64-bit vector add and a dozen scalar instructions, as the builtin form
// compiles) 68 cycles; with the lane offsets of the fragments precomputed in registers (`vo`), the stage's global base in a scalar
// pair and M0 stepped by 1 KB, 25 cycles (tools/clockprobe/dmaprobe2.hip; bare requests without an M0 write issue every 17).
__device__ __forceinline__ void rr_dma_stage16(unsigned lds_base, const void* gbase, const unsigned (&vo)[16]) {
#define RR_DMA_NEXT(K) "s_add_i32 m0, m0, 0x400\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %" #K ", %1\n\t"
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %1\n\t"
               RR_DMA_NEXT(3) RR_DMA_NEXT(4) RR_DMA_NEXT(5) RR_DMA_NEXT(6) RR_DMA_NEXT(7) RR_DMA_NEXT(8) RR_DMA_NEXT(9) RR_DMA_NEXT(10)
               RR_DMA_NEXT(11) RR_DMA_NEXT(12) RR_DMA_NEXT(13) RR_DMA_NEXT(14) RR_DMA_NEXT(15) RR_DMA_NEXT(16) RR_DMA_NEXT(17)
               ::"s"(lds_base), "s"(gbase), "v"(vo[0]), "v"(vo[1]), "v"(vo[2]), "v"(vo[3]), "v"(vo[4]), "v"(vo[5]), "v"(vo[6]), "v"(vo[7]),
               "v"(vo[8]), "v"(vo[9]), "v"(vo[10]), "v"(vo[11]), "v"(vo[12]), "v"(vo[13]), "v"(vo[14]), "v"(vo[15])
               : "memory", "scc", "m0");
#undef RR_DMA_NEXT
}
// one LDS-DMA request of 1 KB in the cheap form: LDS byte address in a scalar, global base in a scalar pair, lane offset in a register
__device__ __forceinline__ void rr_dma1(unsigned lds_addr, const void* gbase, unsigned voff) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(lds_addr), "v"(voff), "s"(gbase) : "memory", "m0");
}
__device__ __forceinline__ unsigned rr_lds_offset(const void* p) {
  return (unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)p;
}
// x[0..7] -> hi + mid + lo, each bf16 with round-to-nearest (v_cvt_pk_bf16_f32): x - hi and (x - hi) - mid are exact in fp32
__device__ __forceinline__ void rr_split8(const float (&x)[8], rr_bf16x8& hi, rr_bf16x8& mid, rr_bf16x8& lo) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const rr_f32x2 v = {x[2 * q], x[2 * q + 1]};
    const rr_bf16x2 h = __builtin_convertvector(v, rr_bf16x2);
    const rr_f32x2 r1 = v - __builtin_convertvector(h, rr_f32x2);
    const rr_bf16x2 m = __builtin_convertvector(r1, rr_bf16x2);
    const rr_f32x2 r2 = r1 - __builtin_convertvector(m, rr_f32x2);
    const rr_bf16x2 l = __builtin_convertvector(r2, rr_bf16x2);
    hi[2 * q] = h[0]; hi[2 * q + 1] = h[1]; mid[2 * q] = m[0]; mid[2 * q + 1] = m[1]; lo[2 * q] = l[0]; lo[2 * q + 1] = l[1];
  }
}
// exp without the low-order correction of rr_exp: |rel err| <= 6e-8 * |x| * log2(e); used where x is O(10)
__device__ __forceinline__ float rr_exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896341f); }

// add a per-feature bias (feature = fbase + 4g + reg) to every node tile
template <int NT>
__device__ __forceinline__ void rr_add_bias(f32x4 (&acc)[NT], const float* __restrict__ bias, int fbase, int lane) {
  const int g = lane >> 4;
  float4 b = rr_ld4(bias + fbase + 4 * g);
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    acc[nt][0] += b.x; acc[nt][1] += b.y; acc[nt][2] += b.z; acc[nt][3] += b.w;
  }
}

// write C tiles to a row-major [node][ld] buffer at columns fbase+4g..+3 (rows >= n_valid skipped)
template <int NT>
__device__ __forceinline__ void rr_store_tiles(const f32x4 (&acc)[NT], float* X, int ld, int fbase, int n_valid, int lane) {
  const int j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int node = nt * 16 + j;
    if (node < n_valid) rr_st4(X + node * ld + fbase + 4 * g, make_float4(acc[nt][0], acc[nt][1], acc[nt][2], acc[nt][3]));
  }
}

template <int NT>
__device__ __forceinline__ void rr_load_tiles(f32x4 (&acc)[NT], const float* X, int ld, int fbase, int n_valid, int lane) {
  const int j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    int node = nt * 16 + j;
    node = node < n_valid ? node : n_valid - 1;
    float4 v = rr_ld4(X + node * ld + fbase + 4 * g);
    acc[nt][0] = v.x; acc[nt][1] = v.y; acc[nt][2] = v.z; acc[nt][3] = v.w;
  }
}

// Lane permutations inside a row of 16 lanes by DPP (dpp_ctrl: quad_perm 0x00-0xFF, row_mirror 0x140, row_half_mirror 0x141): one vector
// instruction, no LDS round trip.  hipcc turns EVERY __shfl_xor into a ds_bpermute_b32 and waits for each (~100 cycles, serialised in
// a reduction): profiles/r06/NOTES.md section 7.
template <int CTRL>
__device__ __forceinline__ float rr_dpp(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float rr_row_xor1(float v) { return rr_dpp<0xB1>(v); }                    // quad_perm [1,0,3,2]
__device__ __forceinline__ float rr_row_xor2(float v) { return rr_dpp<0x4E>(v); }                    // quad_perm [2,3,0,1]
__device__ __forceinline__ float rr_row_xor4(float v) { return rr_dpp<0x1B>(rr_dpp<0x141>(v)); }     // mirror within 8, then within 4
__device__ __forceinline__ float rr_row_xor8(float v) { return rr_dpp<0x141>(rr_dpp<0x140>(v)); }    // mirror within 16, then within 8
// reductions over the 16 lanes that share g (= over the nodes of one tile column set), the result in every lane.  The xor butterfly's
// tree (1, 2, 4, 8): after the first two steps the four lanes of a quad hold the same value, so the partner at distance 4 (8) may be
// any lane of the other quad (half row) — the mirrors, one instruction each; sums and maxima are bit-identical to the __shfl_xor form.
__device__ __forceinline__ float rr_sum16(float v) {
  v += rr_dpp<0xB1>(v); v += rr_dpp<0x4E>(v); v += rr_dpp<0x141>(v); v += rr_dpp<0x140>(v);
  return v;
}
__device__ __forceinline__ float rr_max16(float v) {
  v = fmaxf(v, rr_dpp<0xB1>(v)); v = fmaxf(v, rr_dpp<0x4E>(v));
  v = fmaxf(v, rr_dpp<0x141>(v)); v = fmaxf(v, rr_dpp<0x140>(v));
  return v;
}
// lane l <-> l^16 and l <-> l^32 exchanges on the gfx950 VALU (v_permlane16_swap / v_permlane32_swap) instead of
// ds_bpermute through the LDS crossbar: with both operands equal to v, the two results hold (own, partner) in every lane
__device__ __forceinline__ void rr_pair16(float v, float& a, float& b) {
  auto p = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  a = __uint_as_float(p[0]); b = __uint_as_float(p[1]);
}
__device__ __forceinline__ void rr_pair32(float v, float& a, float& b) {
  auto p = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  a = __uint_as_float(p[0]); b = __uint_as_float(p[1]);
}
// reductions over g (the 4 lane groups that share j = one node / rollout)
__device__ __forceinline__ float rr_sum_g(float v) {
  float a, b;
  rr_pair16(v, a, b); v = a + b;
  rr_pair32(v, a, b); return a + b;
}
// bitwise OR over the 4 lane groups that share j (lane ^ 16, lane ^ 32) on the permlane swaps: as __shfl_xor these were two ds_bpermute
// round trips per word in every decode step of the VRP rollouts' mask update
__device__ __forceinline__ uint32_t rr_or_g(uint32_t v) {
  auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);
  v = p[0] | p[1];
  auto q = __builtin_amdgcn_permlane32_swap(v, v, false, false);
  return q[0] | q[1];
}
__device__ __forceinline__ float rr_max_g(float v) {
  // v_med3_f32(a, b, +inf) = max(a, b) (NaN-free inputs: scores and clipped logits; fmaxf on the two halves of a permlane swap costs
  // three v_max_f32: hipcc canonicalises both inputs first)
  float a, b;
  rr_pair16(v, a, b); v = __builtin_amdgcn_fmed3f(a, b, INFINITY);
  rr_pair32(v, a, b); return __builtin_amdgcn_fmed3f(a, b, INFINITY);
}
// wave reductions in the xor butterfly's order 32, 16, 8, 4, 2, 1 (the same tree as before, bit for bit): the two cross-row steps on
// v_permlane32_swap / v_permlane16_swap, the rest on DPP (true lane ^ 8 and lane ^ 4: two instructions each)
__device__ __forceinline__ float rr_wave_sum(float v) {
  float a, b;
  rr_pair32(v, a, b); v = a + b;
  rr_pair16(v, a, b); v = a + b;
  v += rr_row_xor8(v); v += rr_row_xor4(v); v += rr_row_xor2(v); v += rr_row_xor1(v);
  return v;
}
__device__ __forceinline__ float rr_wave_max(float v) {
  float a, b;
  rr_pair32(v, a, b); v = fmaxf(a, b);
  rr_pair16(v, a, b); v = fmaxf(a, b);
  v = fmaxf(v, rr_row_xor8(v)); v = fmaxf(v, rr_row_xor4(v)); v = fmaxf(v, rr_row_xor2(v)); v = fmaxf(v, rr_row_xor1(v));
  return v;
}
__device__ __forceinline__ float rr_wave_min(float v) {
  float a, b;
  rr_pair32(v, a, b); v = fminf(a, b);
  rr_pair16(v, a, b); v = fminf(a, b);
  v = fminf(v, rr_row_xor8(v)); v = fminf(v, rr_row_xor4(v)); v = fminf(v, rr_row_xor2(v)); v = fminf(v, rr_row_xor1(v));
  return v;
}

// InstanceNorm1d(E, affine) over the node axis on C-layout tiles (attn_freenet.py:84,104-105):
// per feature: mean over nodes, biased variance, eps 1e-5.  Lanes with node >= n_valid are ignored.
template <int NT>
__device__ __forceinline__ void rr_instnorm_tiles(f32x4 (&x)[NT], const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, int fbase, int n_valid, int lane) {
  const int j = lane & 15, g = lane >> 4;
  const float inv_n = 1.0f / (float)n_valid;
  float4 gm = rr_ld4(gamma + fbase + 4 * g), bt = rr_ld4(beta + fbase + 4 * g);
  float gmv[4] = {gm.x, gm.y, gm.z, gm.w}, btv[4] = {bt.x, bt.y, bt.z, bt.w};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    float s = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) s += (nt * 16 + j < n_valid) ? x[nt][r] : 0.f;
    float mean = rr_sum16(s) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) { float d = x[nt][r] - mean; q += (nt * 16 + j < n_valid) ? d * d : 0.f; }
    float var = rr_sum16(q) * inv_n;
    float rstd = 1.0f / sqrtf(var + 1e-5f);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) x[nt][r] = (x[nt][r] - mean) * rstd * gmv[r] + btv[r];
  }
}

// ---- transcendental helpers on the hardware v_exp_f32 / v_log_f32 / v_rcp_f32 (1 ulp each).
// exp keeps full fp32 accuracy by carrying the rounding error of x*log2(e) into a first-order correction, without
// the range / denormal handling of the libm version (our arguments are <= 0 after max-subtraction, or |x| < 30).
__device__ __forceinline__ float rr_exp(float x) {
  const float L2E = 1.44269504088896341f, L2E_LO = 1.92596299112661746e-08f, LN2 = 0.693147180559945309f;
  float xc = fmaxf(x, -104.0f);          // exp(-inf) must be 0, not NaN: clamp, then v_exp_f32 underflows to 0
  float v = xc * L2E;
  float lo = fmaf(xc, L2E, -v);
  lo = fmaf(xc, L2E_LO, lo);
  float e = __builtin_amdgcn_exp2f(v);   // 2^-150 -> 0 (v_exp_f32 flushes)
  return fmaf(e, lo * LN2, e);
}
__device__ __forceinline__ float rr_log(float x) {
  const float LN2_HI = 0.693147180559945309f, LN2_LO = -1.90465429995776804e-09f;
  float l2 = __builtin_amdgcn_logf(x);
  return fmaf(l2, LN2_LO, l2 * LN2_HI);
}
__device__ __forceinline__ float rr_tanh(float x) {          // 1 - 2/(e^{2x}+1); abs error ~1e-7
  float t = rr_exp(2.0f * x);
  return 1.0f - 2.0f * __builtin_amdgcn_rcpf(t + 1.0f);
}
__device__ __forceinline__ float rr_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + rr_exp(-x)); }

// counter-based uniform / Gumbel noise for the sampling decode (keyed by seed, rollout, step, key)
__device__ __forceinline__ uint32_t rr_hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
// (seed, rollout) part of the noise key: constant over a rollout's keys and decode steps — kernels hoist it out of their loops
__device__ __forceinline__ uint32_t rr_noise_key(uint64_t seed, uint32_t r) {
  return rr_hash32((uint32_t)seed ^ rr_hash32(r + 0x9e3779b9U * (uint32_t)(seed >> 32)));
}
__device__ __forceinline__ float rr_uniform_k(uint32_t h1, uint32_t step, uint32_t key) {
  const uint32_t h = rr_hash32(h1 ^ ((step * 131u + key + 0x85ebca6bU) * 0x9E3779B1u));   // one multiply mixes (step, key) in, one full hash finishes
  return ((float)(h >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0,1)
}
__device__ __forceinline__ float rr_gumbel_k(uint32_t h1, uint32_t step, uint32_t key) {
  // -ln(-ln u) on the hardware base-2 logarithms (1 ulp each; noise needs no correction term): 2 transcendentals + 2 multiplies
  const float LN2 = 0.693147180559945309f;
  return -LN2 * __builtin_amdgcn_logf(-LN2 * __builtin_amdgcn_logf(rr_uniform_k(h1, step, key)));
}
// ---- sampling = inverse CDF over the keys in ascending order, one uniform per (seed, rollout, step) (key slot RR_CDF_SLOT): the fused
// rollout (rr_rollout_w.inc, its own lane layout) and the selection kernels draw the same way.  A draw is the LAST key that has mass and
// whose exclusive prefix is <= target; the target stays strictly below the total, and the first key with mass has prefix 0 — so a key
// is always found, and a key without mass never is, whatever the rounding of the prefix sums.
#define RR_CDF_SLOT 0xffffu
__device__ __forceinline__ float rr_cdf_target(float u, float total) {
  return fminf(u * total, __uint_as_float(__float_as_uint(total) - 1u));
}
// exclusive prefix sum over the 64 lanes of a wave (lane 0: 0) and the wave's total
__device__ __forceinline__ float rr_wave_excl_scan(float v, float& total) {
  const int lane = (int)(threadIdx.x & 63);
  float inc = v;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) { const float t = __shfl_up(inc, d); inc += lane >= d ? t : 0.f; }
  total = __shfl(inc, 63);
  const float ex = __shfl_up(inc, 1);
  return lane == 0 ? 0.f : ex;
}
// the same over each 16-lane row of a wave, on DPP (row_shr:n with bound_ctrl: lanes without a source add 0; row_newbcast:15 = the row's
// last lane to all of the row): no LDS crossbar round trips as with __shfl_up
template <int CTRL> __device__ __forceinline__ float rr_dpp0(float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float rr_row16_excl_scan(float v, float& total) {
  float inc = v;
  inc += rr_dpp0<0x111>(inc); inc += rr_dpp0<0x112>(inc); inc += rr_dpp0<0x114>(inc); inc += rr_dpp0<0x118>(inc);
  total = rr_dpp0<0x15F>(inc);
  return rr_dpp0<0x111>(inc);
}
__device__ __forceinline__ float rr_uniform(uint64_t seed, uint32_t r, uint32_t step, uint32_t key) {
  return rr_uniform_k(rr_noise_key(seed, r), step, key);
}
__device__ __forceinline__ float rr_gumbel(uint64_t seed, uint32_t r, uint32_t step, uint32_t key) {
  return rr_gumbel_k(rr_noise_key(seed, r), step, key);
}

static inline int rr_check(hipError_t e) { return e == hipSuccess ? RR_OK : RR_ELAUNCH; }
