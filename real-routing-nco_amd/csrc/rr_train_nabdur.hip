// Backward of the Neural Adaptive Bias WITH the duration matrix (RCVRPTW; rrnco/models/nn/attn_freenet.py:226-237, 265-286) in
// its folded form (packing.fold_nab_dur / models/grad_replay._nab_duration):
//   h_f = relu(a_f x_f + b_f)  (f = distance, angle, duration; 128 units each; x_f one scalar per edge)
//   z = Mcat [h_0; h_1; h_2] + cg (128),  s = silu(z),  l = Wg2 s + bg2 (3),  g = softmax(l / tau)
//   po_f = co_f . h_f + ko_f,  bias = sum_f g_f po_f + bo,  out = alpha * bias
// Given d loss / d out per edge (from k_aft_bwd) it returns the gradients of every folded parameter; autograd chains them to
// the module's.  Two kernels:
//   k_nabdur_bwd_edges : one wave = 16 edges.  H^T (384 x 16) is generated in MFMA B-operand registers, Z^T = Mcat H^T and
//       dH^T = Mcat^T dZ^T run on the fp32 MFMA with the packed weights streamed from L2, everything in between is per-edge
//       arithmetic in registers; the per-unit sums (d a, d b, d co, d cg, d Wg2) are reduced over the 16 edges by DPP row adds,
//       accumulated in LDS and flushed with one atomic per entry and workgroup.  dZ^T is written in MFMA A-operand fragment
//       order for the second kernel.
//   k_nabdur_bwd_mcat  : d Mcat (128 x 384) = dZ^T H, contraction over ALL edges: wave u of a workgroup owns rows 16u..16u+15,
//       recomputes H from the three scalars of an edge (nothing but dZ is read back), accumulates 24 output tiles in registers
//       over a grid-strided set of 16-edge tiles, one atomic add per entry and workgroup at the end.
#include "rr_common.h"
#include <cstdlib>

struct NabDurBwdW {
  const float *a, *b, *co;       // [384] each (family-major)
  const float *cg, *wg2;         // [128], [3][128]
  const float* scal;             // bg2[3], ko[3], inv_tau, bo, alpha
  const float4 *mcat, *mcatT;    // pack_a(Mcat [128][384]) = [8][24][64], pack_a(Mcat^T [384][128]) = [24][8][64]
  const void *mcat_s, *mcatT_s;  // optional: packing.pack_bf16x2 of the same two matrices ([8][12][2][64][8], [24][4][2][64][8]): bf16 pipe
};
// gradient buffer layout (floats)
#define ND_DA 0
#define ND_DB 384
#define ND_DCO 768
#define ND_DCG 1152
#define ND_DWG2 1280
#define ND_DSC 1664            // d bg2[3], d ko[3], d inv_tau, d bo, d alpha
#define ND_GRADS 1680

#define ND_DPP4(R)                                                              \
  "v_add_f32_dpp %0, %0, %0 row_ror:" R " row_mask:0xf bank_mask:0xf\n"         \
  "v_add_f32_dpp %1, %1, %1 row_ror:" R " row_mask:0xf bank_mask:0xf\n"         \
  "v_add_f32_dpp %2, %2, %2 row_ror:" R " row_mask:0xf bank_mask:0xf\n"         \
  "v_add_f32_dpp %3, %3, %3 row_ror:" R " row_mask:0xf bank_mask:0xf\n"
// sum over the 16 lanes of a DPP row (= the 16 edges of the tile, same g), four registers at once (rr_enc_w.inc:ew_rowsum4)
__device__ __forceinline__ void nd_rowsum4(f32x4& v) {
  float a = v[0], b = v[1], c = v[2], d = v[3];
  asm volatile("s_nop 1\n" ND_DPP4("8") ND_DPP4("4") ND_DPP4("2") ND_DPP4("1") : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
  v[0] = a; v[1] = b; v[2] = c; v[3] = d;
}
__device__ __forceinline__ float nd_rowsum1(float v) {
  v += rr_row_xor8(v); v += rr_row_xor4(v); v += rr_row_xor2(v); v += rr_row_xor1(v);      // (DPP: rr_common.h)
  return v;
}

__global__ __launch_bounds__(256, 2) void k_nabdur_bwd_edges(NabDurBwdW w, const float* __restrict__ xd, const float* __restrict__ xa,
                                                             const float* __restrict__ xt, const float* __restrict__ gout,
                                                             float4* __restrict__ dzf, float* __restrict__ grads, long long M) {
  __shared__ __attribute__((aligned(16))) float par[3 * 384 + 128 + 384 + 16];   // a, b, co, cg, wg2, scal
  __shared__ __attribute__((aligned(16))) float acc[ND_GRADS];
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 384; i += 256) { par[i] = w.a[i]; par[384 + i] = w.b[i]; par[768 + i] = w.co[i]; par[1280 + i] = w.wg2[i]; }
  for (int i = tid; i < 128; i += 256) par[1152 + i] = w.cg[i];
  if (tid < 9) par[1664 + tid] = w.scal[tid];
  for (int i = tid; i < ND_GRADS; i += 256) acc[i] = 0.f;
  __syncthreads();
  const float *pa = par, *pb = par + 384, *pco = par + 768, *pcg = par + 1152, *pw = par + 1280, *ps = par + 1664;
  const float inv_tau = ps[6], bo = ps[7], alpha = ps[8];
  const long long ntile = (M + 15) / 16;
  // packed weights through buffer loads: wave-uniform descriptor + per-lane offset + SCALAR fragment offset (no address registers)
  const __amdgpu_buffer_rsrc_t rM = rr_make_buf(w.mcat, 128 * 384 * 4), rMT = rr_make_buf(w.mcatT, 128 * 384 * 4);
  const unsigned lane16 = (unsigned)lane * 16u;
  float s_bg[3] = {0.f, 0.f, 0.f}, s_ko[3] = {0.f, 0.f, 0.f}, s_tau = 0.f, s_bo = 0.f, s_al = 0.f;
  for (long long tile = (long long)blockIdx.x * 4 + wave; tile < ntile; tile += (long long)gridDim.x * 4) {
    const long long e = tile * 16 + j;
    const bool valid = e < M;
    const float x[3] = {valid ? xd[e] : 0.f, valid ? xa[e] : 0.f, valid ? xt[e] : 0.f};
    const float go = valid ? gout[e] : 0.f;
    // ---- H^T in B-operand registers: H[t][r] = unit 16t + 4g + r (family t / 8) of edge j
    f32x4 H[24];
#pragma unroll
    for (int t = 0; t < 24; ++t) {
      if ((t & 3) == 0) __builtin_amdgcn_sched_barrier(0);          // (keeps hipcc from hoisting all 48 parameter reads at once)
      const float4 a4 = rr_ld4(pa + 16 * t + 4 * g), b4 = rr_ld4(pb + 16 * t + 4 * g);
      const float xx = x[t >> 3];
      H[t][0] = fmaxf(fmaf(a4.x, xx, b4.x), 0.f); H[t][1] = fmaxf(fmaf(a4.y, xx, b4.y), 0.f);
      H[t][2] = fmaxf(fmaf(a4.z, xx, b4.z), 0.f); H[t][3] = fmaxf(fmaf(a4.w, xx, b4.w), 0.f);
    }
    // ---- Z^T = Mcat H^T + cg (8 output tiles, 24 k-groups), two accumulation chains per tile
    f32x4 Z[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float4 c4 = rr_ld4(pcg + 16 * u + 4 * g);
      f32x4 c0 = {c4.x, c4.y, c4.z, c4.w}, c1 = rr_zero4();
#pragma unroll
      for (int kk = 0; kk < 24; kk += 2) {
        if ((kk & 7) == 0) __builtin_amdgcn_sched_barrier(0);      // at most 8 fragments requested ahead (registers)
        const float4 f0 = rr_bld4(rM, lane16, (unsigned)(u * 24 + kk) * 1024u), f1 = rr_bld4(rM, lane16, (unsigned)(u * 24 + kk + 1) * 1024u);
        c0 = rr_mfma(f0.x, H[kk][0], c0); c1 = rr_mfma(f1.x, H[kk + 1][0], c1);
        c0 = rr_mfma(f0.y, H[kk][1], c0); c1 = rr_mfma(f1.y, H[kk + 1][1], c1);
        c0 = rr_mfma(f0.z, H[kk][2], c0); c1 = rr_mfma(f1.z, H[kk + 1][2], c1);
        c0 = rr_mfma(f0.w, H[kk][3], c0); c1 = rr_mfma(f1.w, H[kk + 1][3], c1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) Z[u][r] = c0[r] + c1[r];
    }
    // ---- gate and output (per edge; a unit's values sit in the 4 lane groups of the edge's column)
    float l[3] = {0.f, 0.f, 0.f}, po[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if ((u & 1) == 0) __builtin_amdgcn_sched_barrier(0);
      const float4 w0 = rr_ld4(pw + 16 * u + 4 * g), w1 = rr_ld4(pw + 128 + 16 * u + 4 * g), w2 = rr_ld4(pw + 256 + 16 * u + 4 * g);
      const float ww0[4] = {w0.x, w0.y, w0.z, w0.w}, ww1[4] = {w1.x, w1.y, w1.z, w1.w}, ww2[4] = {w2.x, w2.y, w2.z, w2.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = Z[u][r] / (1.0f + __expf(-Z[u][r]));          // silu (recomputed in the backward pass below: not kept)
        l[0] = fmaf(ww0[r], s, l[0]); l[1] = fmaf(ww1[r], s, l[1]); l[2] = fmaf(ww2[r], s, l[2]);
      }
    }
#pragma unroll
    for (int t = 0; t < 24; ++t) {
      if ((t & 3) == 0) __builtin_amdgcn_sched_barrier(0);
      const float4 c4 = rr_ld4(pco + 16 * t + 4 * g);
      po[t >> 3] += H[t][0] * c4.x + H[t][1] * c4.y + H[t][2] * c4.z + H[t][3] * c4.w;
    }
    float gt[3], lr[3];
#pragma unroll
    for (int f = 0; f < 3; ++f) { lr[f] = rr_sum_g(l[f]) + ps[f]; po[f] = rr_sum_g(po[f]) + ps[3 + f]; }
    const float mx = fmaxf(fmaxf(lr[0], lr[1]), lr[2]) * inv_tau;
    float es = 0.f;
#pragma unroll
    for (int f = 0; f < 3; ++f) { gt[f] = __expf(lr[f] * inv_tau - mx); es += gt[f]; }
    float bias = bo;
#pragma unroll
    for (int f = 0; f < 3; ++f) { gt[f] /= es; bias = fmaf(gt[f], po[f], bias); }
    // ---- backward scalars
    const float dbias = go * alpha;
    float dpo[3], dg[3], gdg = 0.f, dlr[3];
#pragma unroll
    for (int f = 0; f < 3; ++f) { dpo[f] = dbias * gt[f]; dg[f] = dbias * po[f]; gdg = fmaf(gt[f], dg[f], gdg); }
    float dtau = 0.f;
#pragma unroll
    for (int f = 0; f < 3; ++f) { const float dl = gt[f] * (dg[f] - gdg); dtau = fmaf(dl, lr[f], dtau); dlr[f] = dl * inv_tau; }
    if (g == 0) {             // one lane per edge carries the per-edge scalars
#pragma unroll
      for (int f = 0; f < 3; ++f) { s_bg[f] += dlr[f]; s_ko[f] += dpo[f]; }
      s_tau += dtau; s_bo += dbias; s_al += go * bias;
    }
    // ---- dZ = (Wg2^T dl) silu'(z); d cg, d Wg2 row sums; dZ^T stored as MFMA A fragments for k_nabdur_bwd_mcat
    f32x4 (&DZ)[8] = Z;           // in place
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      __builtin_amdgcn_sched_barrier(0);
      const float4 w0 = rr_ld4(pw + 16 * u + 4 * g), w1 = rr_ld4(pw + 128 + 16 * u + 4 * g), w2 = rr_ld4(pw + 256 + 16 * u + 4 * g);
      const float ww0[4] = {w0.x, w0.y, w0.z, w0.w}, ww1[4] = {w1.x, w1.y, w1.z, w1.w}, ww2[4] = {w2.x, w2.y, w2.z, w2.w};
      f32x4 r0, r1, r2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float z = Z[u][r];
        const float sg = 1.0f / (1.0f + __expf(-z)), sv = z * sg;
        const float ds = ww0[r] * dlr[0] + ww1[r] * dlr[1] + ww2[r] * dlr[2];
        DZ[u][r] = ds * (sg * (1.0f + z * (1.0f - sg)));
        r0[r] = dlr[0] * sv; r1[r] = dlr[1] * sv; r2[r] = dlr[2] * sv;
      }
      f32x4 dc = DZ[u];
      nd_rowsum4(dc); nd_rowsum4(r0); nd_rowsum4(r1); nd_rowsum4(r2);
      if (j == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          atomicAdd(&acc[ND_DCG + 16 * u + 4 * g + r], dc[r]);
          atomicAdd(&acc[ND_DWG2 + 16 * u + 4 * g + r], r0[r]);
          atomicAdd(&acc[ND_DWG2 + 128 + 16 * u + 4 * g + r], r1[r]);
          atomicAdd(&acc[ND_DWG2 + 256 + 16 * u + 4 * g + r], r2[r]);
        }
      }
      // A fragment (tile, u): lane' (i = 4g + r, g' = j & 3), element j >> 2  <-  dZ^T[16u + 4g + r][edge j]
      float* dst = reinterpret_cast<float*>(dzf + ((size_t)tile * 8 + u) * 64);
#pragma unroll
      for (int r = 0; r < 4; ++r) dst[(((j & 3) * 16 + 4 * g + r) << 2) + (j >> 2)] = DZ[u][r];
    }
    // ---- dH^T = Mcat^T dZ^T tile by tile; d pre-activation; d a, d b, d co row sums
#pragma unroll 1
    for (int t = 0; t < 24; ++t) {
      f32x4 c0 = rr_zero4(), c1 = rr_zero4();
#pragma unroll
      for (int u = 0; u < 8; u += 2) {
        const float4 f0 = rr_bld4(rMT, lane16, (unsigned)(t * 8 + u) * 1024u), f1 = rr_bld4(rMT, lane16, (unsigned)(t * 8 + u + 1) * 1024u);
        c0 = rr_mfma(f0.x, DZ[u][0], c0); c1 = rr_mfma(f1.x, DZ[u + 1][0], c1);
        c0 = rr_mfma(f0.y, DZ[u][1], c0); c1 = rr_mfma(f1.y, DZ[u + 1][1], c1);
        c0 = rr_mfma(f0.z, DZ[u][2], c0); c1 = rr_mfma(f1.z, DZ[u + 1][2], c1);
        c0 = rr_mfma(f0.w, DZ[u][3], c0); c1 = rr_mfma(f1.w, DZ[u + 1][3], c1);
      }
      const int f = t >> 3;
      const float4 c4 = rr_ld4(pco + 16 * t + 4 * g);
      const float cc[4] = {c4.x, c4.y, c4.z, c4.w};
      // H[t] with a runtime t: re-derive the unit's activation from its parameters (cheaper than indexing the register array)
      const float4 a4 = rr_ld4(pa + 16 * t + 4 * g), b4 = rr_ld4(pb + 16 * t + 4 * g);
      const float aa[4] = {a4.x, a4.y, a4.z, a4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
      const float xx = f == 0 ? x[0] : f == 1 ? x[1] : x[2];
      const float dpf = f == 0 ? dpo[0] : f == 1 ? dpo[1] : dpo[2];
      f32x4 da, db, dco;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float h = fmaxf(fmaf(aa[r], xx, bb[r]), 0.f);
        const float dpre = h > 0.f ? (c0[r] + c1[r]) + cc[r] * dpf : 0.f;
        da[r] = dpre * xx; db[r] = dpre; dco[r] = h * dpf;
      }
      nd_rowsum4(da); nd_rowsum4(db); nd_rowsum4(dco);
      if (j == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          atomicAdd(&acc[ND_DA + 16 * t + 4 * g + r], da[r]);
          atomicAdd(&acc[ND_DB + 16 * t + 4 * g + r], db[r]);
          atomicAdd(&acc[ND_DCO + 16 * t + 4 * g + r], dco[r]);
        }
      }
    }
  }
  // per-edge scalars: over the 16 edges of the row (lanes with g == 0 hold them), then LDS
  {
    float v[9] = {s_bg[0], s_bg[1], s_bg[2], s_ko[0], s_ko[1], s_ko[2], s_tau, s_bo, s_al};
#pragma unroll
    for (int q = 0; q < 9; ++q) { const float r = nd_rowsum1(v[q]); if (lane == 0) atomicAdd(&acc[ND_DSC + q], r); }
  }
  __syncthreads();
  for (int i = tid; i < ND_GRADS; i += 256) { const float v = acc[i]; if (v != 0.f) atomicAdd(&grads[i], v); }
}

__global__ __launch_bounds__(512, 2) void k_nabdur_bwd_mcat(NabDurBwdW w, const float* __restrict__ xd, const float* __restrict__ xa,
                                                            const float* __restrict__ xt, const float4* __restrict__ dzf,
                                                            float* __restrict__ dmcat, long long M) {
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int u = __builtin_amdgcn_readfirstlane(tid >> 6);         // output rows 16u .. 16u + 15
  // this lane's column of every output tile: unit 16t + j of family t / 8
  float ua[24], ub[24];
#pragma unroll
  for (int t = 0; t < 24; ++t) { ua[t] = w.a[16 * t + j]; ub[t] = w.b[16 * t + j]; }
  f32x4 acc[24];
#pragma unroll
  for (int t = 0; t < 24; ++t) acc[t] = rr_zero4();
  const long long ntile = (M + 15) / 16;
  for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    const float4 fz = dzf[((size_t)tile * 8 + u) * 64 + lane];      // dZ^T[16u + i][edges 4m + g], m = 0..3 (zero rows for edges >= M)
    float xs[3][4];
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const long long e = tile * 16 + 4 * m + g;
      const bool ok = e < M;
      xs[0][m] = ok ? xd[e] : 0.f; xs[1][m] = ok ? xa[e] : 0.f; xs[2][m] = ok ? xt[e] : 0.f;
    }
    const float fzm[4] = {fz.x, fz.y, fz.z, fz.w};
#pragma unroll
    for (int t = 0; t < 24; ++t) {
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const float h = fmaxf(fmaf(ua[t], xs[t >> 3][m], ub[t]), 0.f);      // B operand: H[edge 4m + g][unit 16t + j]
        acc[t] = rr_mfma(fzm[m], h, acc[t]);
      }
    }
  }
#pragma unroll
  for (int t = 0; t < 24; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (acc[t][r] != 0.f) atomicAdd(&dmcat[(size_t)(16 * u + 4 * g + r) * 384 + 16 * t + j], acc[t][r]);
}

// ------------------------------------------------------------------------------------------------
// The same two kernels on the bf16 matrix pipe with two-piece split operands (x = hi + lo, three partial products, error
// 2^-16 of a product: the scheme of csrc/rr_train_dec.hip): 576 MFMAs of 16 cycles per 16 edges instead of 1 536 of 32, and a
// d Mcat kernel that shares the dZ fragments of a 32-edge tile through LDS (LDS-DMA, double buffered) among eight waves that
// each own three of the 24 unit tiles.
typedef rr_bf16x8 ndfrag;
__device__ __forceinline__ void nd_split8(const float (&x)[8], ndfrag& hi, ndfrag& lo) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const rr_f32x2 v = {x[2 * q], x[2 * q + 1]};
    const rr_bf16x2 h = __builtin_convertvector(v, rr_bf16x2);
    const rr_f32x2 r1 = v - __builtin_convertvector(h, rr_f32x2);
    const rr_bf16x2 l = __builtin_convertvector(r1, rr_bf16x2);
    hi[2 * q] = h[0]; hi[2 * q + 1] = h[1]; lo[2 * q] = l[0]; lo[2 * q + 1] = l[1];
  }
}
__device__ __forceinline__ f32x4 nd_mfma3(ndfrag ah, ndfrag al, ndfrag bh, ndfrag bl, f32x4 c) {
  c = rr_mfma_bf16(ah, bl, c);
  c = rr_mfma_bf16(al, bh, c);
  return rr_mfma_bf16(ah, bh, c);
}

// The two weight images (Mcat as A operands of Z^T = Mcat H^T, Mcat^T as A operands of dH^T = Mcat^T dZ^T: 192 KB each as [hi | lo]
// bf16 fragments) go through LDS in 16 stages of 24 KB per round of tiles — 8 stages = the 8 unit tiles u of Mcat, 8 stages = three
// hidden tiles t of Mcat^T each — fetched ONCE per workgroup and round by LDS-DMA (three 1 KB requests per wave and stage, double
// buffered, one barrier per stage) and read by its eight waves.  With every wave streaming its own fragments from L2 (round 3) a
// 16-edge tile cost 384 KB of L2 reads: 63 GB per call at configs-4 size, 6.3 ms = the L2's bandwidth, ten times the matrix work.
#define ND_STAGE (24 * 1024)
__global__ __launch_bounds__(512, 1) void k_nabdur_bwd_edges2(NabDurBwdW w, const float* __restrict__ xd, const float* __restrict__ xa,
                                                              const float* __restrict__ xt, const float* __restrict__ gout,
                                                              char* __restrict__ dzf, float* __restrict__ grads, long long M) {
  __shared__ __attribute__((aligned(16))) float par[3 * 384 + 128 + 384 + 16];
  __shared__ __attribute__((aligned(16))) float acc[ND_GRADS];
  __shared__ __attribute__((aligned(1024))) char stage[2][ND_STAGE];
  const int tid = threadIdx.x, lane = tid & 63, j = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < 384; i += 512) { par[i] = w.a[i]; par[384 + i] = w.b[i]; par[768 + i] = w.co[i]; par[1280 + i] = w.wg2[i]; }
  for (int i = tid; i < 128; i += 512) par[1152 + i] = w.cg[i];
  if (tid < 9) par[1664 + tid] = w.scal[tid];
  for (int i = tid; i < ND_GRADS; i += 512) acc[i] = 0.f;
  __syncthreads();
  const float *pa = par, *pb = par + 384, *pco = par + 768, *pcg = par + 1152, *pw = par + 1280, *ps = par + 1664;
  const float inv_tau = ps[6], bo = ps[7], alpha = ps[8];
  const long long ntile = ((M + 31) / 32) * 2;                // 16-edge tiles, an even number: both halves of every 32-edge dZ tile are written
  const unsigned lane16 = (unsigned)lane * 16u;
  const unsigned sbase = rr_lds_offset(&stage[0][0]) + (unsigned)(3 * wave) * 1024u;
  // stage q of a round (0..7: Mcat unit tile q, 8..15: Mcat^T hidden tiles 3 (q - 8) ..) into buffer `buf`: this wave's three fragments
  auto issue = [&](int q, int buf) {
    const char* src = (q < 8 ? (const char*)w.mcat_s + (size_t)q * ND_STAGE : (const char*)w.mcatT_s + (size_t)(q - 8) * ND_STAGE) + (size_t)(3 * wave) * 1024;
#pragma unroll
    for (int f = 0; f < 3; ++f) rr_dma1(sbase + (unsigned)buf * (unsigned)ND_STAGE + (unsigned)f * 1024u, src + f * 1024, lane16);
  };
  auto frag = [&](int buf, int f) { return *reinterpret_cast<const ndfrag*>(&stage[buf][f * 1024 + lane * 16]); };
  // every wave of the workgroup walks the same number of rounds (the stages are workgroup-wide); a wave past the last tile computes on zeros and writes nothing
  const long long per_round = (long long)gridDim.x * 8;
  const long long rounds = (ntile + per_round - 1) / per_round;
  float s_bg[3] = {0.f, 0.f, 0.f}, s_ko[3] = {0.f, 0.f, 0.f}, s_tau = 0.f, s_bo = 0.f, s_al = 0.f;
  // Per-unit sums over the wave's edges in REGISTERS: after the row reduction all 16 lanes of a row hold the tile's sum, so lane j keeps
  // the running totals of unit tile u = j (d cg, d Wg2: aZ) and of hidden tiles t = j and t = 16 + j (d a, d b, d co: aH) — 40 registers,
  // flushed to LDS once per kernel.  The 416 four-lane ds_add_f32 per tile they replace were 1.5 of the kernel's 5.8 ms (LDS float
  // atomics are served lane after lane on this chip, see k_nab_hist_bwd).
  f32x4 aZ[4], aH[2][3];
#pragma unroll
  for (int i = 0; i < 4; ++i) aZ[i] = rr_zero4();
#pragma unroll
  for (int i = 0; i < 3; ++i) { aH[0][i] = rr_zero4(); aH[1][i] = rr_zero4(); }
  issue(0, 0);
  for (long long rd = 0; rd < rounds; ++rd) {
    const long long tile = (rd * gridDim.x + blockIdx.x) * 8 + wave;
    const bool tvalid = tile < ntile;
    const long long e = tile * 16 + j;
    const bool valid = tvalid && e < M;
    const float x[3] = {valid ? xd[e] : 0.f, valid ? xa[e] : 0.f, valid ? xt[e] : 0.f};
    const float go = valid ? gout[e] : 0.f;
    // ---- H^T (unit 16t + 4g + r of edge j), po = co . h per family, then H as bf16 pieces in k = 32 operand order
    float po[3] = {0.f, 0.f, 0.f};
    ndfrag Hh[12], Hl[12];
#pragma unroll
    for (int sl = 0; sl < 12; ++sl) {
      if ((sl & 1) == 0) __builtin_amdgcn_sched_barrier(0);
      float hv[8];
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int t = 2 * sl + q;
        const float4 a4 = rr_ld4(pa + 16 * t + 4 * g), b4 = rr_ld4(pb + 16 * t + 4 * g), c4 = rr_ld4(pco + 16 * t + 4 * g);
        const float xx = x[t >> 3];
        hv[4 * q] = fmaxf(fmaf(a4.x, xx, b4.x), 0.f); hv[4 * q + 1] = fmaxf(fmaf(a4.y, xx, b4.y), 0.f);
        hv[4 * q + 2] = fmaxf(fmaf(a4.z, xx, b4.z), 0.f); hv[4 * q + 3] = fmaxf(fmaf(a4.w, xx, b4.w), 0.f);
        po[t >> 3] += hv[4 * q] * c4.x + hv[4 * q + 1] * c4.y + hv[4 * q + 2] * c4.z + hv[4 * q + 3] * c4.w;
      }
      nd_split8(hv, Hh[sl], Hl[sl]);
    }
    // ---- Z^T = Mcat H^T + cg
    f32x4 Z[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const float4 c4 = rr_ld4(pcg + 16 * u + 4 * g);
      f32x4 c0 = {c4.x, c4.y, c4.z, c4.w}, c1 = rr_zero4();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();                      // stage u landed (every wave's requests); the stage before it is consumed by every wave
      issue(u + 1, (u + 1) & 1);
#pragma unroll
      for (int sl = 0; sl < 12; sl += 2) {
        if ((sl & 3) == 0) __builtin_amdgcn_sched_barrier(0);
        const ndfrag a0h = frag(u & 1, sl * 2), a0l = frag(u & 1, sl * 2 + 1);
        const ndfrag a1h = frag(u & 1, (sl + 1) * 2), a1l = frag(u & 1, (sl + 1) * 2 + 1);
        c0 = nd_mfma3(a0h, a0l, Hh[sl], Hl[sl], c0);
        c1 = nd_mfma3(a1h, a1l, Hh[sl + 1], Hl[sl + 1], c1);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int r = 0; r < 4; ++r) Z[u][r] = c0[r] + c1[r];
    }
    // ---- gate and output
    float l[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if ((u & 1) == 0) __builtin_amdgcn_sched_barrier(0);
      const float4 w0 = rr_ld4(pw + 16 * u + 4 * g), w1 = rr_ld4(pw + 128 + 16 * u + 4 * g), w2 = rr_ld4(pw + 256 + 16 * u + 4 * g);
      const float ww0[4] = {w0.x, w0.y, w0.z, w0.w}, ww1[4] = {w1.x, w1.y, w1.z, w1.w}, ww2[4] = {w2.x, w2.y, w2.z, w2.w};
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float sv = Z[u][r] * __builtin_amdgcn_rcpf(1.0f + __expf(-Z[u][r]));      // (v_rcp_f32, 1 ulp: a division is ~10 instructions, 64 of them per tile)
        l[0] = fmaf(ww0[r], sv, l[0]); l[1] = fmaf(ww1[r], sv, l[1]); l[2] = fmaf(ww2[r], sv, l[2]);
      }
    }
    float gt[3], lr[3];
#pragma unroll
    for (int f = 0; f < 3; ++f) { lr[f] = rr_sum_g(l[f]) + ps[f]; po[f] = rr_sum_g(po[f]) + ps[3 + f]; }
    const float mx = fmaxf(fmaxf(lr[0], lr[1]), lr[2]) * inv_tau;
    float es = 0.f;
#pragma unroll
    for (int f = 0; f < 3; ++f) { gt[f] = __expf(lr[f] * inv_tau - mx); es += gt[f]; }
    float bias = bo;
#pragma unroll
    for (int f = 0; f < 3; ++f) { gt[f] /= es; bias = fmaf(gt[f], po[f], bias); }
    const float dbias = go * alpha;
    float dpo[3], dg[3], gdg = 0.f, dlr[3];
#pragma unroll
    for (int f = 0; f < 3; ++f) { dpo[f] = dbias * gt[f]; dg[f] = dbias * po[f]; gdg = fmaf(gt[f], dg[f], gdg); }
    float dtau = 0.f;
#pragma unroll
    for (int f = 0; f < 3; ++f) { const float dl = gt[f] * (dg[f] - gdg); dtau = fmaf(dl, lr[f], dtau); dlr[f] = dl * inv_tau; }
    if (g == 0) {
#pragma unroll
      for (int f = 0; f < 3; ++f) { s_bg[f] += dlr[f]; s_ko[f] += dpo[f]; }
      s_tau += dtau; s_bo += dbias; s_al += go * bias;
    }
    // ---- dZ (in place), d cg / d Wg2 row sums, dZ as bf16 pieces: B fragments of the 32-edge tile for k_nabdur_bwd_mcat2
    f32x4 (&DZ)[8] = Z;
    const long long t32 = tile >> 1;
    const int eo = (int)(tile & 1) * 16 + j;                  // edge within the 32-edge tile
    char* dst0 = dzf + (size_t)t32 * 16 * 1024 + ((eo >> 3) * 16) * 16 + (eo & 7) * 2;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      __builtin_amdgcn_sched_barrier(0);
      const float4 w0 = rr_ld4(pw + 16 * u + 4 * g), w1 = rr_ld4(pw + 128 + 16 * u + 4 * g), w2 = rr_ld4(pw + 256 + 16 * u + 4 * g);
      const float ww0[4] = {w0.x, w0.y, w0.z, w0.w}, ww1[4] = {w1.x, w1.y, w1.z, w1.w}, ww2[4] = {w2.x, w2.y, w2.z, w2.w};
      f32x4 r0, r1, r2;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float z = Z[u][r];
        const float sg = __builtin_amdgcn_rcpf(1.0f + __expf(-z)), sv = z * sg;
        const float ds = ww0[r] * dlr[0] + ww1[r] * dlr[1] + ww2[r] * dlr[2];
        DZ[u][r] = ds * (sg * (1.0f + z * (1.0f - sg)));
        r0[r] = dlr[0] * sv; r1[r] = dlr[1] * sv; r2[r] = dlr[2] * sv;
        // fragment (t32, u, piece): lane' = g' * 16 + (4g + r) with g' = eo >> 3, element eo & 7
        const __bf16 hi = (__bf16)DZ[u][r];
        const __bf16 lo = (__bf16)(DZ[u][r] - (float)hi);
        char* d = dst0 + (size_t)(u * 2) * 1024 + (4 * g + r) * 16;
        if (tvalid) {
          *reinterpret_cast<__bf16*>(d) = hi;
          *reinterpret_cast<__bf16*>(d + 1024) = lo;
        }
      }
      f32x4 dc = DZ[u];
      nd_rowsum4(dc); nd_rowsum4(r0); nd_rowsum4(r1); nd_rowsum4(r2);
      {
        const float mine = j == u ? 1.0f : 0.0f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          aZ[0][r] = fmaf(mine, dc[r], aZ[0][r]); aZ[1][r] = fmaf(mine, r0[r], aZ[1][r]);
          aZ[2][r] = fmaf(mine, r1[r], aZ[2][r]); aZ[3][r] = fmaf(mine, r2[r], aZ[3][r]);
        }
      }
    }
    ndfrag Dh[4], Dl[4];
#pragma unroll
    for (int sl = 0; sl < 4; ++sl) {
      const float dv[8] = {DZ[2 * sl][0], DZ[2 * sl][1], DZ[2 * sl][2], DZ[2 * sl][3], DZ[2 * sl + 1][0], DZ[2 * sl + 1][1], DZ[2 * sl + 1][2], DZ[2 * sl + 1][3]};
      nd_split8(dv, Dh[sl], Dl[sl]);
    }
    // ---- dH^T = Mcat^T dZ^T tile by tile; d pre-activation; d a, d b, d co row sums
    auto dh_tile = [&](int t, f32x4 (&A)[3]) {
      const int q = 8 + t / 3, tt = t - (t / 3) * 3;
      if (tt == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (q + 1 < 16) issue(q + 1, (q + 1) & 1);
        else if (rd + 1 < rounds) issue(0, 0);            // the next round's first stage
      }
      f32x4 c0 = rr_zero4(), c1 = rr_zero4();
#pragma unroll
      for (int sl = 0; sl < 4; sl += 2) {
        const ndfrag a0h = frag(q & 1, (tt * 4 + sl) * 2), a0l = frag(q & 1, (tt * 4 + sl) * 2 + 1);
        const ndfrag a1h = frag(q & 1, (tt * 4 + sl + 1) * 2), a1l = frag(q & 1, (tt * 4 + sl + 1) * 2 + 1);
        c0 = nd_mfma3(a0h, a0l, Dh[sl], Dl[sl], c0);
        c1 = nd_mfma3(a1h, a1l, Dh[sl + 1], Dl[sl + 1], c1);
      }
      const int f = t >> 3;
      const float4 c4 = rr_ld4(pco + 16 * t + 4 * g);
      const float cc[4] = {c4.x, c4.y, c4.z, c4.w};
      const float4 a4 = rr_ld4(pa + 16 * t + 4 * g), b4 = rr_ld4(pb + 16 * t + 4 * g);
      const float aa[4] = {a4.x, a4.y, a4.z, a4.w}, bb[4] = {b4.x, b4.y, b4.z, b4.w};
      const float xx = f == 0 ? x[0] : f == 1 ? x[1] : x[2];
      const float dpf = f == 0 ? dpo[0] : f == 1 ? dpo[1] : dpo[2];
      f32x4 da, db, dco;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float h = fmaxf(fmaf(aa[r], xx, bb[r]), 0.f);
        const float dpre = h > 0.f ? (c0[r] + c1[r]) + cc[r] * dpf : 0.f;
        da[r] = dpre * xx; db[r] = dpre; dco[r] = h * dpf;
      }
      nd_rowsum4(da); nd_rowsum4(db); nd_rowsum4(dco);
      const float mine = j == (t & 15) ? 1.0f : 0.0f;
#pragma unroll
      for (int r = 0; r < 4; ++r) { A[0][r] = fmaf(mine, da[r], A[0][r]); A[1][r] = fmaf(mine, db[r], A[1][r]); A[2][r] = fmaf(mine, dco[r], A[2][r]); }
    };
#pragma unroll 1
    for (int t = 0; t < 16; ++t) dh_tile(t, aH[0]);
#pragma unroll 1
    for (int t = 16; t < 24; ++t) dh_tile(t, aH[1]);
  }
  // (0 * inf: a non-finite sum of a tile this lane does not own would turn into NaN through the multiply by `mine` = 0 — and that is
  // the right answer: the gradient buffer is poisoned either way, as it was with the atomics)
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    if (j < 8) {
      atomicAdd(&acc[ND_DCG + 16 * j + 4 * g + r], aZ[0][r]); atomicAdd(&acc[ND_DWG2 + 16 * j + 4 * g + r], aZ[1][r]);
      atomicAdd(&acc[ND_DWG2 + 128 + 16 * j + 4 * g + r], aZ[2][r]); atomicAdd(&acc[ND_DWG2 + 256 + 16 * j + 4 * g + r], aZ[3][r]);
      atomicAdd(&acc[ND_DA + 16 * (16 + j) + 4 * g + r], aH[1][0][r]); atomicAdd(&acc[ND_DB + 16 * (16 + j) + 4 * g + r], aH[1][1][r]);
      atomicAdd(&acc[ND_DCO + 16 * (16 + j) + 4 * g + r], aH[1][2][r]);
    }
    atomicAdd(&acc[ND_DA + 16 * j + 4 * g + r], aH[0][0][r]); atomicAdd(&acc[ND_DB + 16 * j + 4 * g + r], aH[0][1][r]);
    atomicAdd(&acc[ND_DCO + 16 * j + 4 * g + r], aH[0][2][r]);
  }
  {
    float v[9] = {s_bg[0], s_bg[1], s_bg[2], s_ko[0], s_ko[1], s_ko[2], s_tau, s_bo, s_al};
#pragma unroll
    for (int q = 0; q < 9; ++q) { const float r = nd_rowsum1(v[q]); if (lane == 0) atomicAdd(&acc[ND_DSC + q], r); }
  }
  __syncthreads();
  for (int i = tid; i < ND_GRADS; i += 512) { const float v = acc[i]; if (v != 0.f) atomicAdd(&grads[i], v); }
}

__global__ __launch_bounds__(512, 2) void k_nabdur_bwd_mcat2(NabDurBwdW w, const float* __restrict__ xd, const float* __restrict__ xa,
                                                             const float* __restrict__ xt, const char* __restrict__ dzf,
                                                             float* __restrict__ dmcat, long long M) {
  __shared__ __attribute__((aligned(16))) char stage[2][16 * 1024];      // dZ fragments of a 32-edge tile: (u, piece) x 1 KB
  const int tid = threadIdx.x, lane = tid & 63, i = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  // this wave's unit tiles t = 3 wave .. 3 wave + 2; lane (i, g) = unit 16t + i, edges 8g .. 8g + 7 of the tile
  float ua[3], ub[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) { ua[q] = w.a[16 * (3 * wave + q) + i]; ub[q] = w.b[16 * (3 * wave + q) + i]; }
  f32x4 acc[3][8];
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int u = 0; u < 8; ++u) acc[q][u] = rr_zero4();
  const long long ntile = (M + 31) / 32;
  // this wave's two fragments of a tile (rr_dma1: the tile's base in scalars, lane offsets and LDS slots computed once)
  const unsigned fvo0 = (unsigned)lane * 16u + (unsigned)(2 * wave) * 1024u, fdst0 = rr_lds_offset(&stage[0][0]) + (unsigned)(2 * wave) * 1024u;
  auto issue = [&](long long tile, int buf) {
    const char* gb = dzf + (size_t)tile * 16 * 1024;
    rr_dma1(fdst0 + (unsigned)buf * 16384u, gb, fvo0);
    rr_dma1(fdst0 + (unsigned)buf * 16384u + 1024u, gb, fvo0 + 1024u);
  };
  int buf = 0;
  // the three scalars of a tile's edges, requested one tile ahead like the dZ fragments (they used to be loaded behind the barrier: a
  // round trip to L2 exposed in front of every tile's 72 matrix instructions)
  float xn[3][8];
  auto load_x = [&](long long tile) {
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const long long e = tile * 32 + 8 * g + q;
      const bool ok = e < M;
      xn[0][q] = ok ? xd[e] : 0.f; xn[1][q] = ok ? xa[e] : 0.f; xn[2][q] = ok ? xt[e] : 0.f;
    }
  };
  if ((long long)blockIdx.x < ntile) { issue(blockIdx.x, 0); load_x(blockIdx.x); }
  for (long long tile = blockIdx.x; tile < ntile; tile += gridDim.x) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float xs[3][8];
#pragma unroll
    for (int f = 0; f < 3; ++f)
#pragma unroll
      for (int q = 0; q < 8; ++q) xs[f][q] = xn[f][q];
    if (tile + gridDim.x < ntile) { issue(tile + gridDim.x, buf ^ 1); load_x(tile + gridDim.x); }
    ndfrag Bh[8], Bl[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      Bh[u] = *reinterpret_cast<const ndfrag*>(stage[buf] + (u * 2) * 1024 + lane * 16);
      Bl[u] = *reinterpret_cast<const ndfrag*>(stage[buf] + (u * 2 + 1) * 1024 + lane * 16);
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
      const int f = (3 * wave + q) >> 3;                     // wave-uniform
      float hv[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float xx = f == 0 ? xs[0][k] : f == 1 ? xs[1][k] : xs[2][k];
        const long long e = tile * 32 + 8 * g + k;
        hv[k] = e < M ? fmaxf(fmaf(ua[q], xx, ub[q]), 0.f) : 0.f;
      }
      ndfrag Ah, Al;
      nd_split8(hv, Ah, Al);
#pragma unroll
      for (int u = 0; u < 8; ++u) acc[q][u] = nd_mfma3(Ah, Al, Bh[u], Bl[u], acc[q][u]);
    }
    buf ^= 1;
  }
  // acc[q][u]: rows = units 16 (3 wave + q) + 4g + r, columns = z 16u + i
#pragma unroll
  for (int q = 0; q < 3; ++q)
#pragma unroll
    for (int u = 0; u < 8; ++u)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (acc[q][u][r] != 0.f) atomicAdd(&dmcat[(size_t)(16 * u + i) * 384 + 16 * (3 * wave + q) + 4 * g + r], acc[q][u][r]);
}

// grads [ND_GRADS] and dmcat [128 * 384] must be zero-filled by the caller (the kernels add); dzf: M rounded up to 32, x 128 floats
extern "C" int rr_nabdur_bwd(const NabDurBwdW* w, const float* xd, const float* xa, const float* xt, const float* gout, float* dzf,
                             float* grads, float* dmcat, long long M, hipStream_t st) {
  if (M <= 0) return RR_OK;
  if (w == nullptr || xd == nullptr || xa == nullptr || xt == nullptr || gout == nullptr || dzf == nullptr) return RR_EINVAL;
  static const int f32only = getenv("RR_NABDUR_F32") ? atoi(getenv("RR_NABDUR_F32")) : 0;
  if (w->mcat_s != nullptr && w->mcatT_s != nullptr && !f32only) {      // bf16 pipe, two-piece operands
    const long long nt16 = ((M + 31) / 32) * 2, nt32 = (M + 31) / 32;
    const int g1 = (int)((nt16 + 7) / 8 < 256 ? (nt16 + 7) / 8 : 256);          // one workgroup of eight waves per CU
    hipLaunchKernelGGL(k_nabdur_bwd_edges2, dim3(g1), dim3(512), 0, st, *w, xd, xa, xt, gout, (char*)dzf, grads, M);
    const int g2 = (int)(nt32 < 512 ? nt32 : 512);
    hipLaunchKernelGGL(k_nabdur_bwd_mcat2, dim3(g2), dim3(512), 0, st, *w, xd, xa, xt, (const char*)dzf, dmcat, M);
    return rr_check(hipGetLastError());
  }
  const long long ntile = (M + 15) / 16;
  const int g1 = (int)((ntile + 3) / 4 < 2048 ? (ntile + 3) / 4 : 2048);
  hipLaunchKernelGGL(k_nabdur_bwd_edges, dim3(g1), dim3(256), 0, st, *w, xd, xa, xt, gout, (float4*)dzf, grads, M);
  const int g2 = (int)(ntile < 1024 ? ntile : 1024);
  hipLaunchKernelGGL(k_nabdur_bwd_mcat, dim3(g2), dim3(512), 0, st, *w, xd, xa, xt, (const float4*)dzf, dmcat, M);
  return rr_check(hipGetLastError());
}
