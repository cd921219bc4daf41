#!/usr/bin/env python3
"""Generates tools/clockprobe/pipeprobe.hip (diagnostic, not part of the product): VERDICT r05 Next #1(a).

What do vector instructions cost when ONE wave issues them beside the matrix instructions of the rollout's MLP region (an intra-wave
software pipeline: tile A in the pointer MLP, tile B's softmax / selection arithmetic in the gaps)?  The loop body is ONE inline-asm
block with fixed registers, so the instruction order is exactly the one listed here (hipcc's scheduler, SLP vectoriser and register
allocator are out of the picture; `sched_group_barrier` on builtins did not pin the order: the first version of this probe came out
with the vector instructions packed and behind the matrix ones).

region = 24 v_mfma_f32_16x16x32_f16 (G1: 12 on the fragments read in the previous region, G2: 12 on the fragments read at the start of
this one; one dependent hidden chain + four output chains as in rr_rollout_w.inc) + 16 ds_read_b128 (static LDS content)
+ K vector instructions on 8 independent chains (every 8th a v_exp_f32) + optionally one workgroup barrier.
  place 0: the K vector instructions spread evenly behind the matrix instructions      place 1: all K behind the 24 matrix instructions
"""
import sys

VA, VB = 210, 211          # filler constants
VF = 200                   # 8 filler chains v200..v207
XA, XB = 64, 96            # fragment registers: Xa v[64:95], Xb v[96:127] (8 x 4 each)
GS = 32                    # B operands v[32:63] (8 x 4)
ACC = 128                  # accumulators: hidden chain v[128:131], outputs v[132:147]
ADDR = 212                 # LDS address


def region(K, place, lds, bar, agpr=0, indep=0, fop='fma', rd=0):
    ins = []
    nfill = [0]

    def fill(n):
        for _ in range(n):
            j = nfill[0]
            c = j % 8
            if c == 7 and fop != 'add_noexp' and fop != 'mov':
                ins.append(f"v_exp_f32 v{VF + c}, v{VF + c}")
            elif fop == 'fma':
                ins.append(f"v_fma_f32 v{VF + c}, v{VF + c}, v{VA}, v{VB}")
            elif fop in ('add', 'add_noexp'):
                ins.append(f"v_add_f32 v{VF + c}, v{VF + c}, v{VA}")
            elif fop == 'mov':
                ins.append(f"v_mov_b32 v{VF + c}, v{VA}")
            nfill[0] += 1

    def mfma(m):
        R = "a" if agpr else "v"
        base = 0 if agpr else ACC
        if m < 12:      # G1: hidden chain on Xb (read last region)
            a = XB + 4 * (m % 8)
            b = GS + 4 * (m % 8)
            c = base + (4 * (m % 5) if indep else 0)
        else:           # G2: four output chains on Xa (read at the start of this region)
            i = m - 12
            a = XA + 4 * (i % 8)
            b = GS + 4 * (i % 8)
            c = base + (4 * (m % 5) if indep else 4 + 4 * (i % 4))
        ins.append(f"v_mfma_f32_16x16x32_f16 {R}[{c}:{c + 3}], v[{a}:{a + 3}], v[{b}:{b + 3}], {R}[{c}:{c + 3}]")

    done = 0
    for m in range(24):
        if lds and rd == 1 and m in (0, 12):
            for q in range(8):
                if m == 0:
                    ins.append(f"ds_read_b128 v[{XA + 4 * q}:{XA + 4 * q + 3}], v{ADDR} offset:{q * 1024}")
                else:
                    ins.append(f"ds_read_b128 v[{XB + 4 * q}:{XB + 4 * q + 3}], v{ADDR} offset:{(8 + q) * 1024}")
            if m == 12:
                ins.append("s_waitcnt lgkmcnt(8)")
        if lds and rd == 0 and m >= 12:
            i = m - 12
            # every A read (8, issued behind G1's first 8) must have landed; the B reads issued so far may be in flight
            ins.append(f"s_waitcnt lgkmcnt({min(i, 8)})")
        mfma(m)
        if lds and rd == 0:
            if m < 8:
                ins.append(f"ds_read_b128 v[{XA + 4 * m}:{XA + 4 * m + 3}], v{ADDR} offset:{m * 1024}")
            elif 12 <= m < 20:
                i = m - 12
                ins.append(f"ds_read_b128 v[{XB + 4 * i}:{XB + 4 * i + 3}], v{ADDR} offset:{(8 + i) * 1024}")
        if place == 0:
            want = (m + 1) * K // 24
            fill(want - done)
            done = want
    if place == 1:
        fill(K)
    if lds:
        ins.append("s_waitcnt lgkmcnt(0)")
    if bar:
        ins.append("s_barrier")
    return ins


def kernel(name, K, place, lds, bar, wg, agpr=0, indep=0, fop='fma', rd=0):
    body = region(K, place, lds, bar, agpr, indep, fop, rd)
    asm = "\n".join(f'      "{i}\\n\\t"' for i in body)
    clob = ", ".join([f'"v{r}"' for r in list(range(GS, ACC + 20)) + list(range(VF, VF + 8))] + [f'"a{r}"' for r in range(20)])
    init = "\n".join(f'      "v_mov_b32 v{r}, 0x3c003c00\\n\\t"' for r in range(GS, ACC)) + "\n" + \
        "\n".join(f'      "v_mov_b32 v{r}, 0\\n\\t"' for r in range(ACC, ACC + 20)) + "\n" + \
        "\n".join(f'      "v_accvgpr_write_b32 a{r}, 0\\n\\t"' for r in range(20)) + "\n" + \
        "\n".join(f'      "v_mov_b32 v{VF + c}, 1.0\\n\\t"' for c in range(8))
    return f"""
__global__ __launch_bounds__({wg}) void {name}(int nreg, unsigned long long* cyc, float* out) {{
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 48 * 1024 / 4; i += {wg}) reinterpret_cast<unsigned*>(lds)[i] = 0x3c003c00u + (unsigned)(i & 7);
  __syncthreads();
  asm volatile(
{init}
      "v_mov_b32 v{VA}, 0.5\\n\\t"
      "v_mov_b32 v{VB}, 1.0\\n\\t"
      "v_mov_b32 v{ADDR}, %0\\n\\t"
      :: "v"((unsigned)(lane * 16)) : {clob}, "v{VA}", "v{VB}", "v{ADDR}");
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int r = 0; r < nreg; ++r) {{
    asm volatile(
{asm}
      ::: {clob}, "memory");
  }}
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (lane == 0) atomicAdd(cyc, t1 - t0);
  float acc;
  asm volatile("v_add_f32 %0, v{ACC}, v{VF}\\n\\tv_add_f32 %0, %0, v{ACC + 4}" : "=v"(acc) :: {clob});
  if (acc == 12345.f) out[0] = acc;
}}
"""


def main():
    variants = []
    for bar in (0, 1):
        for K in (0, 24, 48, 72, 96, 144):
            variants.append((K, 0, 1, bar, 256))
    for K in (48, 96):
        variants.append((K, 1, 1, 1, 256))          # vector instructions behind the matrix ones (what a phase-separated wave does)
    for K in (0, 48, 96):
        variants.append((K, 0, 0, 1, 256))          # without the fragment reads
    for K in (0, 24, 48, 96):
        variants.append((K, 0, 1, 1, 512))          # two waves per SIMD running the same stream (each with its own K)
    variants.append((48, 1, 1, 1, 512))
    out = ["""// GENERATED by pipeprobe_gen.py — do not edit.  Diagnostic (not part of the product); see the generator's docstring.
// Build: hipcc --offload-arch=gfx950 -O3 pipeprobe.hip -o pipeprobe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
"""]
    names = []
    variants = [v + (0, 0, 'fma') for v in variants]
    for agpr, indep, fop in ((1, 0, 'fma'), (0, 1, 'fma'), (1, 1, 'fma'), (1, 1, 'add'), (1, 1, 'add_noexp'), (1, 1, 'mov'), (0, 0, 'add_noexp')):
        for K in (0, 24, 48, 96):
            variants.append((K, 0, 0, 0, 256, agpr, indep, fop))
    variants = [v + (0,) for v in variants]
    for wg in (448, 512):
        for rd in (0, 1):
            for K in (0, 16, 24):
                variants.append((K, 0, 1, 1, wg, 0, 0, 'add_noexp', rd))
    for (K, place, lds, bar, wg, agpr, indep, fop, rd) in variants:
        name = f"k_pipe_k{K}_p{place}_l{lds}_b{bar}_w{wg}_a{agpr}_i{indep}_{fop}_r{rd}"
        names.append((name, K, place, lds, bar, wg, agpr, indep, fop, rd))
        out.append(kernel(name, K, place, lds, bar, wg, agpr, indep, fop, rd))
    out.append("""
typedef void (*kern_t)(int, unsigned long long*, float*);
static void run(kern_t k, int wg, const char* what, int K, unsigned long long* cyc, float* out) {
  const int grid = 1024, nreg = 3200;
  (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k, dim3(256), dim3(wg), 160 * 1024, 0, 32, cyc, out);
  (void)hipMemset(cyc, 0, 8);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k, dim3(grid), dim3(wg), 160 * 1024, 0, nreg, cyc, out);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long c; (void)hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double per_region = (double)c / grid / (wg / 64) / nreg;
  const int wps = (wg + 255) / 256;                       // waves on the fullest SIMD: they run side by side
  printf("%-62s K %3d (%.1f per matrix instr): %7.1f cycles per region and wave = %.3f x %d (the SIMD's matrix time for its %d wave(s)), %7.2f ms\\n",
         what, K, K / 24.0, per_region, per_region / (384.0 * wps), 384 * wps, wps, ms);
  fflush(stdout);
}
int main() {
  unsigned long long* cyc; float* out;
  (void)hipMalloc(&cyc, 8); (void)hipMalloc(&out, 4);
""")
    for (name, K, place, lds, bar, wg, agpr, indep, fop, rd) in names:
        what = f"place {place} reads {lds}{'b' if rd else 'i'} bar {bar} waves {wg // 64} agpr {agpr} indep {indep} {fop}"
        out.append(f'  run({name}, {wg}, "{what}", {K}, cyc, out);\n')
    out.append("  return 0;\n}\n")
    open(sys.argv[1] if len(sys.argv) > 1 else "pipeprobe.hip", "w").write("".join(out))


if __name__ == "__main__":
    main()
