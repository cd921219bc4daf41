// Hand-written backward of the REINFORCE step's decoder (BASELINE configs[4]): everything
// rrnco/models/rl.py:118-128 differentiates through rrnco/models/decoder.py:151-329 (RRNetDecoder.forward,
// RRNet_PointerAttention) and rrnco/models/decoding.py:311-361 (process_logits) for the sampled tours.
//
// The sampling rollout (csrc/rr_rollout_w.inc) leaves, per decoder evaluation m = (instance b, step t, start s), what it
// had in registers anyway: the pointer-MLP input g0[m][128] (glimpse + query, decoder.py:294), its output g[m][128]
// (:296), the action-mask words the decision saw, the node it was taken at, the node chosen (RolloutIO::dump_*).  With those
// the backward is five row-parallel kernels, none of which replays the environment and none of which is problem specific:
//
//   k_dec_logit_bwd   per 16-row tile: logits = g L^T / sqrt(E) (fp32 MFMA), inductive bias, log / 10 tanh / mask /
//                     log-softmax exactly as the rollout computed them, the row's log-probability (the replayed
//                     log-likelihood), d logits (-> dlg[m][112]) and dg = dlg L (fp32 MFMA); d alpha / d beta.
//   k_gemm_tn         dL[b] = dlg_b^T g_b: batched "TN" GEMM with the row index as the MFMA k dimension (also the weight
//                     gradient of every Linear layer: dW = dY^T X).
//   k_mlp_rows<1>     dg0 = dg + W1^T [ (W2^T dg) . 1(W1 g0 + b1 > 0) ]: the pointer MLP's input gradient, on the bf16 matrix
//                     pipe with every fp32 operand split in two bf16 pieces (hi + lo, three partial products, error 2^-16 of
//                     a product; the fp32 MFMA is 1/16 of that rate), weights shared by the workgroup through LDS-DMA stages.
//   k_mlp_wgrad       dW1, db1, dW2, db2 from (g0, dg): recomputes the hidden tile it owns, products with the ROW index on the
//                     MFMA k axis from transposed LDS images.
//   k_dec_attn_bwd    masked multi-head attention backward, one (instance, head) per workgroup so that K_h, V_h, K_h^T and
//                     the dK_h / dV_h accumulators stay in registers over all rows of the instance; d query scattered to the
//                     step-context tables (ctxA[first], ctxB[current]) through LDS float atomics.
#include "rr_common.h"

#define TD_NT 7
#define TD_LDK 112          // padded key count: dlg rows, K^T / L^T rows
#define TD_AS 20           // row stride (floats) of k_dec_attn_bwd's d query accumulators in LDS

// rows of one launch: nseg segments (instances) of seg_rows live rows each, seg_stride rows apart
struct RowSegs { int nseg; int seg_rows; long long seg_stride; };
__device__ __forceinline__ long long td_row(const RowSegs& rs, long long i) {
  if (rs.nseg == 1) return i;                          // wave-uniform
  const unsigned iu = (unsigned)i, sg = iu / (unsigned)rs.seg_rows;       // row counts stay far below 2^31: 32-bit divide
  return (long long)sg * rs.seg_stride + (long long)(iu - sg * (unsigned)rs.seg_rows);
}

// ------------------------------------------------------------------------------------------------ logits backward
struct DecLogitIO {
  float* g;                   // [rows][128] pointer-MLP output (decoder.py:296); dead rows are zeroed in place (the d L product reads them)
  const uint32_t* meta;       // [rows][8]
  const float *L, *Lt;        // logit keys [Bp][N][128] and transposed, zero padded [Bp][128][112]
  const float *D, *Dur;       // [Bp][N][N] (Dur NULL unless rcvrptw)
  const float* gll;           // [S*Bp] d loss / d log-likelihood, r = s*Bp + b
  float *dlg;                 // [rows][112] d loss / d (L g) (before the 1/sqrt(E))
  float *dg;                  // [rows][128]
  float *logp;                // [rows] log-probability of the chosen node (0 for dead rows)
  float *dscal;               // [2] += d alpha, d beta (decoder.py:187-190)
  int Bp, N, S, T;            // T = decode steps that ran; rows of instance b: b*seg_stride + t*S + s
  long long seg_stride;
  float alpha, beta, tanh_clip, temperature;
  const void* Ls;             // fp16 two-piece image of L (PrecomputedCache.split_images / rr_pack_f16x2) or NULL: fp32-MFMA kernel
};

__global__ __launch_bounds__(256, 2) void k_dec_logit_bwd(DecLogitIO io, int tiles_per_inst, int wgs_per_inst) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / wgs_per_inst;
  const int tile = (blockIdx.x - b * wgs_per_inst) * 4 + wave;
  if (tile >= tiles_per_inst) return;
  const int N = io.N, S = io.S, rows_b = io.T * S;
  int q = tile * 16 + j;
  const bool vrow = q < rows_b;
  q = vrow ? q : rows_b - 1;
  const size_t m = (size_t)b * (size_t)io.seg_stride + (size_t)q;
  const int t = q / S, s = q - t * S;
  const uint4 m0 = *reinterpret_cast<const uint4*>(io.meta + m * 8);
  const uint4 m1 = *reinterpret_cast<const uint4*>(io.meta + m * 8 + 4);
  const int prev = min((int)m1.x, N - 1), target = (int)m1.y;
  const bool live = vrow && m1.z != 0u;
  const float gl = live ? io.gll[(size_t)s * io.Bp + b] : 0.f;
  const uint32_t mw[4] = {m0.x, m0.y, m0.z, m0.w};

  f32x4 F[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) {
    const float4 v = rr_ld4(io.g + m * RR_E + 16 * kk + 4 * g);
    F[kk][0] = live ? v.x : 0.f; F[kk][1] = live ? v.y : 0.f; F[kk][2] = live ? v.z : 0.f; F[kk][3] = live ? v.w : 0.f;
    if (vrow && !live) rr_st4(io.g + m * RR_E + 16 * kk + 4 * g, make_float4(0.f, 0.f, 0.f, 0.f));
  }
  // ---- logits^T[key][row] = L F^T (decoder.py:300-302)
  const size_t nE4 = (size_t)N * RR_E * 4;
  const __amdgpu_buffer_rsrc_t rL = rr_make_buf((const char*)io.L + (size_t)b * nE4, (unsigned)nE4);
  unsigned koff[TD_NT];
#pragma unroll
  for (int kt = 0; kt < TD_NT; ++kt) {
    int key = kt * 16 + j; key = key < N ? key : N - 1;
    koff[kt] = (unsigned)(key * RR_E + 4 * g) * 4u;
  }
  f32x4 la[TD_NT];
#pragma unroll
  for (int kt = 0; kt < TD_NT; ++kt) la[kt] = rr_zero4();
  {
    float4 fa[TD_NT], fb[TD_NT];
#pragma unroll
    for (int kt = 0; kt < TD_NT; ++kt) fa[kt] = rr_bld4(rL, koff[kt], 0);
#pragma unroll
    for (int kk = 0; kk < 8; kk += 2) {
#pragma unroll
      for (int kt = 0; kt < TD_NT; ++kt) fb[kt] = rr_bld4(rL, koff[kt], 64u * (kk + 1));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kt = 0; kt < TD_NT; ++kt) {
        la[kt] = rr_mfma(fa[kt].x, F[kk][0], la[kt]); la[kt] = rr_mfma(fa[kt].y, F[kk][1], la[kt]);
        la[kt] = rr_mfma(fa[kt].z, F[kk][2], la[kt]); la[kt] = rr_mfma(fa[kt].w, F[kk][3], la[kt]);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kk + 2 < 8) {
#pragma unroll
        for (int kt = 0; kt < TD_NT; ++kt) fa[kt] = rr_bld4(rL, koff[kt], 64u * (kk + 2));
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kt = 0; kt < TD_NT; ++kt) {
        la[kt] = rr_mfma(fb[kt].x, F[kk + 1][0], la[kt]); la[kt] = rr_mfma(fb[kt].y, F[kk + 1][1], la[kt]);
        la[kt] = rr_mfma(fb[kt].z, F[kk + 1][2], la[kt]); la[kt] = rr_mfma(fb[kt].w, F[kk + 1][3], la[kt]);
      }
    }
  }
  // ---- bias, log(exp(.) + 1e-6) (decoder.py:187-198), 10 tanh, mask, temperature, log-softmax (decoding.py:341-361)
  const float inv_sqe = 1.0f / sqrtf((float)RR_E), inv_temp = 1.0f / io.temperature;
  const bool clip = io.tanh_clip > 0.f;
  const float* Drow = io.D + ((size_t)b * N + prev) * N;
  const float* Trow = io.Dur ? io.Dur + ((size_t)b * N + prev) * N : nullptr;
  float uu[TD_NT][4], vv[TD_NT][4], dd[TD_NT][4], tt[TD_NT][4];
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < TD_NT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kt * 16 + 4 * g + r;
      const int kc = key < N ? key : N - 1;
      dd[kt][r] = Drow[kc];
      tt[kt][r] = Trow ? Trow[kc] : 0.f;
      const float x = la[kt][r] * inv_sqe - (io.alpha * dd[kt][r] + io.beta * tt[kt][r]);
      const float u = rr_exp(x) + 1e-6f;
      uu[kt][r] = u;
      float v;
      if (clip) v = (1.0f - 2.0f / fmaf(u, u, 1.0f)) * io.tanh_clip * inv_temp;      // tanh(log u) = (u^2-1)/(u^2+1)
      else v = rr_log(u) * inv_temp;
      const bool ok = key < N && ((mw[kt >> 1] >> (16 * (kt & 1) + 4 * g + r)) & 1u);
      v = ok ? v : -INFINITY;
      vv[kt][r] = v;
      mx = fmaxf(mx, v);
    }
  mx = rr_max_g(mx);
  if (mx == -INFINITY) mx = 0.f;
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < TD_NT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) sum += rr_exp(vv[kt][r] - mx);
  sum = rr_sum_g(sum);
  const float lse = rr_log(sum);
  float lpt = 0.f, da = 0.f, db = 0.f;
  f32x4 dla[TD_NT];
#pragma unroll
  for (int kt = 0; kt < TD_NT; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kt * 16 + 4 * g + r;
      const float v = vv[kt][r];
      const bool ok = v > -INFINITY;
      const float lp = v - mx - lse;
      if (key == target) lpt += ok ? lp : 0.f;
      const float p = ok ? rr_exp(lp) : 0.f;
      float dv = gl * ((key == target ? 1.0f : 0.f) - p);          // d loss / d v (log-softmax picked at the target)
      dv = (ok && live) ? dv : 0.f;
      const float u = uu[kt][r];
      float dx;                                                    // through v(u), u = exp(x) + 1e-6
      if (clip) { const float w = fmaf(u, u, 1.0f); dx = dv * io.tanh_clip * inv_temp * (4.0f * u / (w * w)) * (u - 1e-6f); }
      else dx = dv * inv_temp * (u - 1e-6f) / u;
      dla[kt][r] = dx * inv_sqe;
      da -= dx * dd[kt][r];
      db -= dx * tt[kt][r];
    }
  lpt = rr_sum_g(lpt);
  if (vrow && g == 0) io.logp[m] = live ? lpt : 0.f;
  if (vrow) {
#pragma unroll
    for (int kt = 0; kt < TD_NT; ++kt)
      rr_st4(io.dlg + m * TD_LDK + 16 * kt + 4 * g, make_float4(dla[kt][0], dla[kt][1], dla[kt][2], dla[kt][3]));
  }
  da = rr_wave_sum(vrow ? da : 0.f); db = rr_wave_sum(vrow ? db : 0.f);
  if (lane == 0) { atomicAdd(io.dscal, da); if (io.Dur) atomicAdd(io.dscal + 1, db); }
  // ---- dg^T[feat][row] = L^T dla  (A operand: L^T rows = features, k = key)
  const __amdgpu_buffer_rsrc_t rT = rr_make_buf((const char*)io.Lt + (size_t)b * (RR_E * TD_LDK * 4), RR_E * TD_LDK * 4);
  const unsigned voff = (unsigned)(j * TD_LDK + 4 * g) * 4u;
  float4 lf[TD_NT], ln[TD_NT];
#pragma unroll
  for (int kt = 0; kt < TD_NT; ++kt) lf[kt] = rr_bld4(rT, voff, 64u * kt);
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    if (u + 1 < 8) {
#pragma unroll
      for (int kt = 0; kt < TD_NT; ++kt) ln[kt] = rr_bld4(rT, voff, (unsigned)((u + 1) * 16 * TD_LDK * 4) + 64u * kt);
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 c0 = rr_zero4(), c1 = rr_zero4();
#pragma unroll
    for (int kt = 0; kt < TD_NT; ++kt) {
      c0 = rr_mfma(lf[kt].x, dla[kt][0], c0); c1 = rr_mfma(lf[kt].y, dla[kt][1], c1);
      c0 = rr_mfma(lf[kt].z, dla[kt][2], c0); c1 = rr_mfma(lf[kt].w, dla[kt][3], c1);
    }
    if (vrow) rr_st4(io.dg + m * RR_E + 16 * u + 4 * g, make_float4(c0[0] + c1[0], c0[1] + c1[1], c0[2] + c1[2], c0[3] + c1[3]));
    if (u + 1 < 8) {
#pragma unroll
      for (int kt = 0; kt < TD_NT; ++kt) lf[kt] = ln[kt];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

static int rr_dec_logit_bwd_split(const DecLogitIO* io, int tiles, hipStream_t st);      // (k_dec_logit_bwd_s, below)
extern "C" int rr_dec_logit_bwd(const DecLogitIO* io, hipStream_t st) {
  if (io == nullptr || io->g == nullptr || io->meta == nullptr || io->L == nullptr || io->Lt == nullptr || io->D == nullptr ||
      io->gll == nullptr || io->dlg == nullptr || io->dg == nullptr || io->logp == nullptr || io->dscal == nullptr)
    return RR_EINVAL;
  if (io->Bp <= 0 || io->N < 2 || io->N > TD_LDK || io->S < 1 || io->T < 1) return RR_EINVAL;
  const int tiles = (io->T * io->S + 15) / 16, wgs = (tiles + 3) / 4;
  if (io->Ls != nullptr) return rr_dec_logit_bwd_split(io, tiles, st);
  hipLaunchKernelGGL(k_dec_logit_bwd, dim3((unsigned)io->Bp * wgs), dim3(256), 0, st, *io, tiles, wgs);
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ C = A^T B (k = rows)
// C[b][p][q] (+)= sum_m A[b][m][p] B[b][m][q],  p < P (<= PT*16 per p-block), q < 128.
// A workgroup (4 waves) owns a PT*16 x 128 block of one batch element and one M split; wave w owns q tiles 2w, 2w+1.
// Rows go through LDS in chunks of 32 (row-major, leading dimensions = 16 mod 32 words so that the four k rows a
// v_mfma_f32_16x16x4_f32 operand takes from one ds_read_b32 fall on distinct banks).
template <int PT>
__global__ __launch_bounds__(256, 2) void k_gemm_tn(const float* __restrict__ A, const float* __restrict__ B, float* __restrict__ C,
                                                    int Mb, int lda, int ldb, int ldc, int P, long long sA, long long sB, long long sC,
                                                    int msplit, int accumulate, float* __restrict__ ws) {
  constexpr int LA = PT * 16 + ((PT * 16) % 32 == 16 ? 0 : 16), LB = 144;
  __shared__ __attribute__((aligned(16))) float As[32 * LA];
  __shared__ __attribute__((aligned(16))) float Bs[32 * LB];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int pb = blockIdx.x, ms = blockIdx.y, b = blockIdx.z;
  const int p0 = pb * PT * 16;
  const int chunks = (Mb + 31) / 32;
  const int cper = (chunks + msplit - 1) / msplit;
  const int c_lo = ms * cper, c_hi = min(chunks, c_lo + cper);
  const float* Ab = A + (size_t)b * sA;
  const float* Bb = B + (size_t)b * sB;
  constexpr int NA4 = 32 * PT * 4, NB4 = 32 * 32;          // float4 units per chunk
  constexpr int IA = (NA4 + 255) / 256, IB = NB4 / 256;
  float4 ra[IA], rb[IB];
  auto g_load = [&](int c) {
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const int e = tid + 256 * i;
      const int row = e / (PT * 4), c4 = e - row * (PT * 4);
      const int mrow = c * 32 + row;
      const bool ok = e < NA4 && mrow < Mb && p0 + 4 * c4 < lda;
      ra[i] = ok ? rr_ld4(Ab + (size_t)mrow * lda + p0 + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const int e = tid + 256 * i;
      const int row = e >> 5, c4 = e & 31;
      const int mrow = c * 32 + row;
      rb[i] = mrow < Mb ? rr_ld4(Bb + (size_t)mrow * ldb + 4 * c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto s_store = [&]() {
#pragma unroll
    for (int i = 0; i < IA; ++i) {
      const int e = tid + 256 * i;
      const int row = e / (PT * 4), c4 = e - row * (PT * 4);
      if (e < NA4) rr_st4(As + row * LA + 4 * c4, ra[i]);
    }
#pragma unroll
    for (int i = 0; i < IB; ++i) {
      const int e = tid + 256 * i;
      rr_st4(Bs + (e >> 5) * LB + 4 * (e & 31), rb[i]);
    }
  };
  f32x4 acc[PT][2];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) { acc[pt][0] = rr_zero4(); acc[pt][1] = rr_zero4(); }
  if (c_lo < c_hi) g_load(c_lo);
  for (int c = c_lo; c < c_hi; ++c) {
    __syncthreads();                 // previous chunk fully consumed
    s_store();
    __syncthreads();
    if (c + 1 < c_hi) g_load(c + 1);
#pragma unroll
    for (int kb = 0; kb < 8; ++kb) {
      const float* ar = As + (4 * kb + g) * LA + j;
      const float* br = Bs + (4 * kb + g) * LB + 32 * wave + j;
      const float b0 = br[0], b1 = br[16];
      float a[PT];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) a[pt] = ar[16 * pt];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) { acc[pt][0] = rr_mfma(a[pt], b0, acc[pt][0]); acc[pt][1] = rr_mfma(a[pt], b1, acc[pt][1]); }
    }
  }
  float* Cb = C + (size_t)b * sC;
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
#pragma unroll
    for (int qt = 0; qt < 2; ++qt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int p = p0 + 16 * pt + 4 * g + r, qq = 32 * wave + 16 * qt + j;
        if (p < P) {
          float* dst = Cb + (size_t)p * ldc + qq;
          if (ws != nullptr) ws[(((size_t)b * msplit + ms) * P + p) * 128 + qq] = acc[pt][qt][r];      // this split's partial: k_split_reduce adds them up
          else if (msplit > 1 || accumulate) atomicAdd(dst, acc[pt][qt][r]);
          else *dst = acc[pt][qt][r];
        }
      }
}

// C[b][p][q] = (accumulate ? C : 0) + sum over the msplit partials, in a fixed order.  The chip retires ~1e11 global float atomics
// per second: 256 row splits of a 128 x 128 block are 4.2 M of them, 42 of the 50 us such a product took with the atomic epilogue.
__global__ __launch_bounds__(1024) void k_split_reduce(const float* __restrict__ ws, float* __restrict__ C, int P, int ldc, long long sC,
                                                       int msplit, int accumulate) {
  // 256 elements per workgroup, FOUR threads per element (a quarter of the splits each, then a fixed-order sum through LDS): with one
  // thread per element the 64 workgroups of a 128 x 128 block had too few loads in flight (16 MB in 19 us)
  __shared__ float part[4][256];
  const int el = threadIdx.x & 255, qd = threadIdx.x >> 8;
  const int e = blockIdx.x * 256 + el, b = blockIdx.y;
  const bool ok = e < P * 128;
  const int per = (msplit + 3) >> 2, m0 = qd * per, m1 = min(m0 + per, msplit);
  const float* src = ws + (size_t)b * msplit * P * 128 + (ok ? e : 0);
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int m = m0;
  for (; m + 4 <= m1; m += 4) {
    s0 += src[(size_t)m * P * 128]; s1 += src[(size_t)(m + 1) * P * 128]; s2 += src[(size_t)(m + 2) * P * 128]; s3 += src[(size_t)(m + 3) * P * 128];
  }
  for (; m < m1; ++m) s0 += src[(size_t)m * P * 128];
  part[qd][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (qd == 0 && ok) {
    const int p = e >> 7, q = e & 127;
    float* dst = C + (size_t)b * sC + (size_t)p * ldc + q;
    const float tot = (part[0][el] + part[1][el]) + (part[2][el] + part[3][el]);
    *dst = accumulate ? *dst + tot : tot;
  }
}

// C[b] = A[b]^T B[b] with B 128 columns wide.  msplit > 1 (or accumulate) adds into C: with a workspace `ws` of batch * msplit * P * 128
// floats through per-split partials and a fixed-order reduction (deterministic), without one with float atomics (the caller zeroes C).
extern "C" int rr_gemm_tn(const float* A, const float* B, float* C, int batch, int Mb, int P, int lda, int ldb, int ldc,
                          long long strideA, long long strideB, long long strideC, int msplit, int accumulate, float* ws, hipStream_t st) {
  if (A == nullptr || B == nullptr || C == nullptr || batch <= 0 || Mb <= 0 || P <= 0 || msplit < 1) return RR_EINVAL;
  if ((lda & 3) || (ldb & 3) || ldb < 128 || lda < P) return RR_EINVAL;
  if (msplit == 1 && !accumulate) ws = nullptr;             // plain stores already
  if (P <= 112 || (P % 128) != 0) {
    if (P > 112 && (P % 112) != 0) return RR_EINVAL;
    hipLaunchKernelGGL((k_gemm_tn<7>), dim3((P + 111) / 112, msplit, batch), dim3(256), 0, st, A, B, C, Mb, lda, ldb, ldc, P,
                       strideA, strideB, strideC, msplit, accumulate, ws);
  } else {
    hipLaunchKernelGGL((k_gemm_tn<8>), dim3(P / 128, msplit, batch), dim3(256), 0, st, A, B, C, Mb, lda, ldb, ldc, P,
                       strideA, strideB, strideC, msplit, accumulate, ws);
  }
  if (ws != nullptr)
    hipLaunchKernelGGL(k_split_reduce, dim3((P * 128 + 255) / 256, batch), dim3(1024), 0, st, ws, C, P, ldc, strideC, msplit, accumulate);
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ pointer MLP on rows
// bf16 matrix pipe, two-piece split operands.  x = hi + lo with hi = bf16(x), lo = bf16(x - hi): x - (hi + lo) <= 2^-17 |x|;
// a product keeps hi*hi + hi*lo + lo*hi (fp32 accumulate), dropping lo*lo <= 2^-16 |x w|.
typedef rr_bf16x8 bfrag;
__device__ __forceinline__ void td_split8(const float (&x)[8], bfrag& hi, bfrag& lo) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const rr_f32x2 v = {x[2 * q], x[2 * q + 1]};
    const rr_bf16x2 h = __builtin_convertvector(v, rr_bf16x2);
    const rr_f32x2 r1 = v - __builtin_convertvector(h, rr_f32x2);
    const rr_bf16x2 l = __builtin_convertvector(r1, rr_bf16x2);
    hi[2 * q] = h[0]; hi[2 * q + 1] = h[1]; lo[2 * q] = l[0]; lo[2 * q + 1] = l[1];
  }
}
// HALF (the opt-in "16-mixed" training step, configs/trainer/default.yaml:8): ONE bf16 piece per operand — what the reference's own
// training precision multiplies (autocast: half-precision matmuls, fp32 accumulate and master weights) — a third of the matrix
// instructions.  Never the default; tolerance: tests/test_gpu_mixed.py against the reference's gradients under autocast.
template <bool HALF = false>
__device__ __forceinline__ f32x4 td_mfma3(bfrag ah, bfrag al, bfrag bh, bfrag bl, f32x4 c) {
  if constexpr (!HALF) {
    c = rr_mfma_bf16(ah, bl, c);
    c = rr_mfma_bf16(al, bh, c);
  }
  return rr_mfma_bf16(ah, bh, c);
}
__device__ __forceinline__ void td_glds16(const void* gsrc, void* ldst) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
}

#ifndef RR_ROWS_ASM
#define RR_ROWS_ASM 0          // 3: the stage body of k_mlp_rows<1> as hand-scheduled asm (rr_mlp_rows_asm.h) — bit-identical, and NOT faster: the kernel
                               // runs at the board's power limit (1 340 W, 2.14 GHz; profiles/r06/NOTES.md section 6), cycles saved come back as a lower clock
#endif
#include "rr_mlp_rows_asm.h"

struct MlpRowsW {
  const void* wa1;   // W1   [512][128] as A operands, tile-major  [32][4][2][64][8] bf16 (packing.pack_bf16x2)
  const void* wa2;   // MODE 1: W2^T [512][128] likewise
  const void* wb;    // MODE 0: W2 [128][512], MODE 1: W1^T [128][512]; k-major [16][8][2][64][8]
  const float *b1, *b2;
};

// MODE 0: out = x + W2 relu(W1 x + b1) + b2        (TransformerFFN / pointer MLP forward, decoder.py:296)
// MODE 1: out = dy + W1^T [ (W2^T dy) . 1(W1 x + b1 > 0) ]   (its input gradient)
// A workgroup = 8 waves x 16 rows (two waves per SIMD); the weight fragments of one hidden pair (32 units) are one LDS stage,
// filled by LDS-DMA one stage ahead, one barrier per stage.  The rows of the workgroup's next block are requested during the
// first stage of the current one and converted when it is done.
template <int MODE, bool HALF = false>
__global__ __launch_bounds__(512, 1) void k_mlp_rows(MlpRowsW w, const float* __restrict__ X, const float* __restrict__ dY,
                                                     float* __restrict__ out, RowSegs rs, const uint32_t* __restrict__ meta) {
  constexpr int NF = MODE == 1 ? 48 : 32;                 // fragments (1 KB each) per stage
  extern __shared__ __attribute__((aligned(16))) char td_lds[];
  char* stage = td_lds;                                   // [2][NF * 1024]
  float* b1s = reinterpret_cast<float*>(td_lds + 2 * NF * 1024);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  for (int i = tid; i < RR_FF; i += 512) b1s[i] = w.b1[i];
  const long long total = (long long)rs.nseg * rs.seg_rows;
  const long long nblk = (total + 127) / 128;
  // this wave's NF / 8 fragments of a stage: global base and LDS slot of each as scalars, computed once; a request is then
  // s_mov m0 / global_load_lds (rr_dma1: 25 instead of 68 cycles of the wave's issue time, tools/clockprobe/dmaprobe2.hip)
  const char* fsrc[NF / 8];
  unsigned fdst[NF / 8];
#pragma unroll
  for (int q = 0; q < NF / 8; ++q) {
    const int f = wave * (NF / 8) + q;
    if (f < 16) fsrc[q] = (const char*)w.wa1 + (size_t)f * 1024;
    else if (MODE == 1 && f < 32) fsrc[q] = (const char*)w.wa2 + (size_t)(f - 16) * 1024;
    else fsrc[q] = (const char*)w.wb + (size_t)(f - (MODE == 1 ? 32 : 16)) * 1024;
    fdst[q] = rr_lds_offset(stage) + (unsigned)f * 1024u;
  }
  const unsigned lane16 = (unsigned)lane * 16u;
  auto issue = [&](int p, int buf) {
    const unsigned vo = lane16 + (unsigned)p * 16384u;
#pragma unroll
    for (int q = 0; q < NF / 8; ++q) rr_dma1(fdst[q] + (unsigned)buf * (unsigned)(NF * 1024), fsrc[q], vo);
  };
  auto frag = [&](int buf, int f) { return *reinterpret_cast<const bfrag*>(stage + buf * (NF * 1024) + f * 1024 + lane * 16); };
  // a lane's row of a block: 8 + 8 float4 in flight (every load unconditional, zeroed by the flag afterwards: no dependent loads)
  long long nrow; bool nvr; unsigned nlv;
  float4 xa4[4], xb4[4], ya4[4], yb4[4];
  auto load_rows = [&](long long blk) {
    long long i = blk * 128 + wave * 16 + j;
    nvr = i < total;
    i = nvr ? i : total - 1;
    nrow = td_row(rs, i);
    nlv = meta == nullptr ? 1u : meta[nrow * 8 + 6];      // rows the rollout never reached (finished routes) read as zero rows
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      xa4[s] = rr_ld4(X + nrow * RR_E + 32 * s + 4 * g); xb4[s] = rr_ld4(X + nrow * RR_E + 32 * s + 16 + 4 * g);
      if (MODE == 1) { ya4[s] = rr_ld4(dY + nrow * RR_E + 32 * s + 4 * g); yb4[s] = rr_ld4(dY + nrow * RR_E + 32 * s + 16 + 4 * g); }
    }
  };
  if ((long long)blockIdx.x < nblk) load_rows(blockIdx.x);
  for (long long blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const long long mrow = nrow; const bool vr = nvr;
    bfrag Xh[4], Xl[4], Yh[4], Yl[4];
    f32x4 acc[8];
    {
      const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
      const bool lv = nlv != 0u;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const float4 xa = lv ? xa4[s] : z4, xb = lv ? xb4[s] : z4;
        const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
        td_split8(xv, Xh[s], Xl[s]);
        if (MODE == 1) {
          const float4 ya = lv ? ya4[s] : z4, yb = lv ? yb4[s] : z4;
          const float yv[8] = {ya.x, ya.y, ya.z, ya.w, yb.x, yb.y, yb.z, yb.w};
          td_split8(yv, Yh[s], Yl[s]);
          acc[2 * s] = f32x4{ya.x, ya.y, ya.z, ya.w}; acc[2 * s + 1] = f32x4{yb.x, yb.y, yb.z, yb.w};
        } else {
          const float4 ba = rr_ld4(w.b2 + 32 * s + 4 * g), bb = rr_ld4(w.b2 + 32 * s + 16 + 4 * g);
          acc[2 * s] = f32x4{xa.x + ba.x, xa.y + ba.y, xa.z + ba.z, xa.w + ba.w};
          acc[2 * s + 1] = f32x4{xb.x + bb.x, xb.y + bb.y, xb.z + bb.z, xb.w + bb.w};
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    issue(0, 0);
#pragma unroll 1
    for (int p = 0; p < RR_FF / 32; ++p) {
      const int buf = p & 1;
#ifndef RR_ROWS_KO
#define RR_ROWS_KO 0             // diagnostic knock-outs (results are wrong): 1 no weight DMA, 2 no stage barrier
#endif
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (!(RR_ROWS_KO & 2)) __syncthreads();                    // stage p landed (every wave's DMA) and stage p-1 is consumed by every wave
      if (!(RR_ROWS_KO & 1) && p + 1 < RR_FF / 32) issue(p + 1, buf ^ 1);
      if (p == 0 && blk + gridDim.x < nblk) load_rows(blk + gridDim.x);      // (behind the DMA of stage 1: the wait of stage 1 covers both)
      f32x4 pre[2], dpre[2];
      constexpr bool ASM_BODY = (RR_ROWS_ASM & 1) && MODE == 1 && !HALF, ASM_BODY2 = (RR_ROWS_ASM & 2) && MODE == 1 && !HALF;     // hand-scheduled stage body (rr_mlp_rows_asm.h); -DRR_ROWS_ASM=0: hipcc's
      const unsigned la = rr_lds_offset(stage) + (unsigned)buf * (unsigned)(NF * 1024) + lane16;
      if constexpr (ASM_BODY) {
        const float4 bv0 = rr_ld4(b1s + 32 * p + 4 * g), bv1 = rr_ld4(b1s + 32 * p + 16 + 4 * g);
        pre[0] = f32x4{bv0.x, bv0.y, bv0.z, bv0.w}; pre[1] = f32x4{bv1.x, bv1.y, bv1.z, bv1.w};
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // (the bias reads: the blocks count their own reads only)
        td_rows_p1<false>(pre[0], dpre[0], Xh, Xl, Yh, Yl, la);
        td_rows_p1<true>(pre[1], dpre[1], Xh, Xl, Yh, Yl, la + 8u * 1024u);
      } else
#pragma unroll
      for (int tl = 0; tl < 2; ++tl) {
        const float4 bv = rr_ld4(b1s + 32 * p + 16 * tl + 4 * g);
        pre[tl] = f32x4{bv.x, bv.y, bv.z, bv.w}; dpre[tl] = rr_zero4();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const bfrag ah = frag(buf, tl * 8 + s * 2), al = frag(buf, tl * 8 + s * 2 + 1);
          pre[tl] = td_mfma3<HALF>(ah, al, Xh[s], Xl[s], pre[tl]);
          if (MODE == 1) {
            const bfrag ch = frag(buf, 16 + tl * 8 + s * 2), cl = frag(buf, 16 + tl * 8 + s * 2 + 1);
            dpre[tl] = td_mfma3<HALF>(ch, cl, Yh[s], Yl[s], dpre[tl]);
          }
        }
      }
      bfrag Hh, Hl;
      {
        float hx[8];
#pragma unroll
        for (int tl = 0; tl < 2; ++tl)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            hx[4 * tl + r] = MODE == 1 ? (pre[tl][r] > 0.f ? dpre[tl][r] : 0.f) : fmaxf(pre[tl][r], 0.f);
        td_split8(hx, Hh, Hl);
      }
      constexpr int FB = MODE == 1 ? 32 : 16;
      if constexpr (ASM_BODY2) {
        td_rows_p2(acc, Hh, Hl, la + (unsigned)FB * 1024u);
      } else
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const bfrag bh = frag(buf, FB + u * 2), bl = frag(buf, FB + u * 2 + 1);
        acc[u] = td_mfma3<HALF>(bh, bl, Hh, Hl, acc[u]);
      }
    }
    if constexpr (RR_ROWS_ASM && MODE == 1 && !HALF) asm volatile("s_nop 15" ::: "memory");      // (the asm blocks' accumulators, before the stores read them)
    if (vr) {
#pragma unroll
      for (int u = 0; u < 8; ++u)
        rr_st4(out + mrow * RR_E + 16 * u + 4 * g, make_float4(acc[u][0], acc[u][1], acc[u][2], acc[u][3]));
    }
    __syncthreads();                      // the next block's first stage overwrites buffer 0
  }
}

extern "C" int rr_mlp_rows(const MlpRowsW* w, int mode, const float* X, const float* dY, float* out, const uint32_t* meta,
                           int nseg, int seg_rows, long long seg_stride, hipStream_t st) {
  if (w == nullptr || X == nullptr || out == nullptr || w->wa1 == nullptr || w->wb == nullptr || w->b1 == nullptr) return RR_EINVAL;
  if (mode == 1 && (dY == nullptr || w->wa2 == nullptr)) return RR_EINVAL;
  if (mode == 0 && w->b2 == nullptr) return RR_EINVAL;
  const bool half = (mode & 2) != 0;                     // modes 2 / 3: modes 0 / 1 with one bf16 piece per operand ("16-mixed")
  mode &= ~2;
  if (mode < 0 || mode > 1 || nseg <= 0 || seg_rows <= 0 || seg_stride < seg_rows) return RR_EINVAL;
  RowSegs rs{nseg, seg_rows, seg_stride};
  const long long nblk = ((long long)nseg * seg_rows + 127) / 128;
  const unsigned grid = (unsigned)(nblk < 256 * 8 ? nblk : 256 * 8);
  if (half) {
    const int shm = 2 * (mode == 1 ? 48 : 32) * 1024 + RR_FF * 4;
    if (mode == 1) {
      (void)hipFuncSetAttribute((const void*)k_mlp_rows<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
      hipLaunchKernelGGL((k_mlp_rows<1, true>), dim3(grid), dim3(512), shm, st, *w, X, dY, out, rs, meta);
    } else {
      (void)hipFuncSetAttribute((const void*)k_mlp_rows<0, true>, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
      hipLaunchKernelGGL((k_mlp_rows<0, true>), dim3(grid), dim3(512), shm, st, *w, X, dY, out, rs, meta);
    }
    return rr_check(hipGetLastError());
  }
  if (mode == 1) {
    const int shm = 2 * 48 * 1024 + RR_FF * 4;
    (void)hipFuncSetAttribute((const void*)k_mlp_rows<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
    hipLaunchKernelGGL((k_mlp_rows<1, false>), dim3(grid), dim3(512), shm, st, *w, X, dY, out, rs, meta);
  } else {
    const int shm = 2 * 32 * 1024 + RR_FF * 4;
    (void)hipFuncSetAttribute((const void*)k_mlp_rows<0, false>, hipFuncAttributeMaxDynamicSharedMemorySize, shm);
    hipLaunchKernelGGL((k_mlp_rows<0, false>), dim3(grid), dim3(512), shm, st, *w, X, dY, out, rs, meta);
  }
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ pointer MLP weight gradients
struct MlpWgradW {
  const void* w1n;    // W1   [512][128], natural-k B operands [32][4][2][64][8] bf16 (packing.pack_bf16x2_nat)
  const void* w2tn;   // W2^T [512][128] likewise
  const float* b1;
};
#define WG_RM 288     // bytes per row of the row-major bf16 images (128 values + 32 B: conflict-free ds_read_b128 by row)
#define WG_TQ 36      // 16-byte slots per feature residue block of the transposed images (32 + 4: residue blocks 16 banks apart)
#define WG_TG (4 * WG_TQ * 16)   // bytes per row-group block of a transposed image

// dW1[hid][feat] += sum_m dH[m][hid] x[m][feat], db1 += sum_m dH, dW2[feat][hid] += sum_m dy[m][feat] H[m][hid], db2 += sum_m dy
// with H = relu(W1 x + b1), dH = (W2^T dy) . 1(H > 0).  Workgroup (slab, split): hidden units [128 slab, +128), wave w the
// tile w of them with its W1 / W2^T fragments resident in registers; rows in chunks of 32 through LDS: a row-major
// image (A operand of the recomputation, k = feature) and a transposed one (A operand of the two outer products, k = row).
// Transposed image: [row group gg][feature slot][8 rows = 16 B], row 16 rt + 4 gg + r at position 4 rt + r, feature f in slot
// (f & 3) * WG_TQ + (f >> 2).  A staging thread owns rows 16 rt + 4 gg + (0..3) of four consecutive features: its 8-byte
// writes run along the slots (lanes = consecutive slots, the two half-waves = the two halves of a slot) and the 16-byte
// operand reads of a 16-feature tile hit 16 different bank quads: neither side conflicts.
template <bool HALF>
__global__ __launch_bounds__(512, 1) void k_mlp_wgrad(MlpWgradW w, const float* __restrict__ X, const float* __restrict__ dY,
                                                      float* __restrict__ dW1, float* __restrict__ db1, float* __restrict__ dW2,
                                                      float* __restrict__ db2, RowSegs rs, int nsplit, const uint32_t* __restrict__ meta,
                                                      float* __restrict__ ws) {
  __shared__ __attribute__((aligned(16))) char rm[2][4][32 * WG_RM];     // [buffer][x hi, x lo, dy hi, dy lo]
  __shared__ __attribute__((aligned(16))) char tr[2][4][4 * WG_TG];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // eight waves: two per SIMD, one hidden tile each
  const int j = lane & 15, g = lane >> 4;
  // the four slabs of one row split share an XCD (blocks idx, idx+8, ...): its L2 serves the re-reads of the rows
  const int idx = blockIdx.x, xcd = idx & 7, kq = idx >> 3;
  const int slab = kq & 3, split = (kq >> 2) * 8 + xcd;
  if (split >= nsplit) return;
  // with a workspace the (slab, split) block of dW1 / dW2 is STORED to the split's partial image [dW1 512 x 128 | dW2 128 x 512] and
  // k_wgrad_reduce adds the splits up in a fixed order (64 splits x 4 slabs x 32 K float atomics were 84 of the 240 us of an encoder FFN's call)
  float* wsp = ws ? ws + (size_t)split * (2 * RR_FF * RR_E) : nullptr;
  const long long total = (long long)rs.nseg * rs.seg_rows;
  const long long chunks = (total + 31) / 32;
  const long long cper = (chunks + nsplit - 1) / nsplit;
  const long long c_lo = split * cper, c_hi = (c_lo + cper < chunks) ? c_lo + cper : chunks;
  const int T0 = slab * 8 + wave;                         // this wave's hidden tile
  bfrag W1h[4], W1l[4], W2h[4], W2l[4];
  const float b1v = w.b1[16 * T0 + j];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const size_t f = ((size_t)T0 * 4 + s) * 2;
    W1h[s] = *reinterpret_cast<const bfrag*>((const char*)w.w1n + f * 1024 + lane * 16);
    W1l[s] = *reinterpret_cast<const bfrag*>((const char*)w.w1n + (f + 1) * 1024 + lane * 16);
    W2h[s] = *reinterpret_cast<const bfrag*>((const char*)w.w2tn + f * 1024 + lane * 16);
    W2l[s] = *reinterpret_cast<const bfrag*>((const char*)w.w2tn + (f + 1) * 1024 + lane * 16);
  }
  f32x4 aW2[8], aW1[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) { aW2[u] = rr_zero4(); aW1[u] = rr_zero4(); }
  float ab1 = 0.f;
  float4 sb2 = make_float4(0.f, 0.f, 0.f, 0.f);
  // staging: thread -> column group c4 = tid & 31 (4 features), rows r0, r0 + 1 with r0 = 16 rt + 4 gg + 2 sh, (gg, rt, sh) = tid >> 5
  const int c4 = tid & 31, rg = tid >> 5;
  const int sh = rg & 1, srt = (rg >> 1) & 1, sgg = rg >> 2, r0 = 16 * srt + 4 * sgg + 2 * sh;
  float4 px[2], py[2];                   // the next chunk's rows of this thread (requested a chunk ahead)
  unsigned plv[2];
  // every load unconditional (clamped row), zeroed by the flag when staged: no dependent loads.  !real: a chunk past the end.
  // Row -> address without a divide: segment = (i * magic) >> 40, magic = 2^40 / seg_rows + 1 (exact while i * seg_rows < 2^40:
  // checked by the launcher); one segment: magic 0.
  const unsigned long long magic = rs.nseg == 1 ? 0ull : (1ull << 40) / (unsigned)rs.seg_rows + 1ull;
  long long mrow[2];
  auto g_load_x = [&](long long c) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const long long i0 = c * 32 + r0 + k, i = i0 < total ? i0 : total - 1;
      const long long sg = (long long)(((unsigned long long)i * magic) >> 40);
      mrow[k] = sg * rs.seg_stride + (i - sg * rs.seg_rows);
      px[k] = rr_ld4(X + mrow[k] * RR_E + 4 * c4);
    }
  };
  auto g_load_y = [&](long long c, bool real) {      // (after g_load_x of the same chunk)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      py[k] = rr_ld4(dY + mrow[k] * RR_E + 4 * c4);
      plv[k] = (real && c * 32 + r0 + k < total) ? (meta == nullptr ? 1u : meta[mrow[k] * 8 + 6]) : 0u;       // dead rows (finished routes) read as zero rows
    }
  };
  uint32_t ch[2][2], cl[2][2];           // the operand being staged as packed bf16 pieces: [row][feature pair]
  // row r0 + k, features 4c4 .. 4c4 + 3 -> pieces, and the row-major images (dead rows as zeros: selects, NaN-safe)
  auto put_row = [&](float4 v, int k, char* rmh, char* rml) {
    const bool lv = plv[k] != 0u;
    const float xs[4] = {lv ? v.x : 0.f, lv ? v.y : 0.f, lv ? v.z : 0.f, lv ? v.w : 0.f};
    uint16_t h[4], l[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const __bf16 hb = (__bf16)xs[q];
      const __bf16 lb = (__bf16)(xs[q] - (float)hb);
      h[q] = __builtin_bit_cast(uint16_t, hb); l[q] = __builtin_bit_cast(uint16_t, lb);
    }
    ch[k][0] = h[0] | ((uint32_t)h[1] << 16); ch[k][1] = h[2] | ((uint32_t)h[3] << 16);
    cl[k][0] = l[0] | ((uint32_t)l[1] << 16); cl[k][1] = l[2] | ((uint32_t)l[3] << 16);
    *reinterpret_cast<uint2*>(rmh + (r0 + k) * WG_RM + c4 * 8) = make_uint2(ch[k][0], ch[k][1]);
    *reinterpret_cast<uint2*>(rml + (r0 + k) * WG_RM + c4 * 8) = make_uint2(cl[k][0], cl[k][1]);
  };
  // the transposed images: per feature the two rows as one 4-byte write (positions 4 rt + 2 sh + 0..1 of row group gg)
  auto put_tr = [&](int q, char* trh, char* trl) {
    const unsigned sel = (q & 1) ? 0x07060302u : 0x05040100u;       // the high / low halves of two packed pairs
    const int off = sgg * WG_TG + (q * WG_TQ + c4) * 16 + 8 * srt + 4 * sh;
    *reinterpret_cast<uint32_t*>(trh + off) = __builtin_amdgcn_perm(ch[1][q >> 1], ch[0][q >> 1], sel);
    *reinterpret_cast<uint32_t*>(trl + off) = __builtin_amdgcn_perm(cl[1][q >> 1], cl[0][q >> 1], sel);
  };
  // Double-buffered images: chunk c is consumed from buffer c & 1 while chunk c + 1 (in registers since the previous chunk) is
  // converted and written to the other one in eight slices placed between the matrix-instruction groups of chunk c (VALU work
  // under the matrix pipe), and chunk c + 2 is requested as soon as a slice has freed its registers.  One barrier per chunk.
  auto stage = [&](int n, int buf, long long cn, bool real) {      // slices 0-3: x, 4-7: dy; cn: the chunk to request
    if (n == 0) put_row(px[0], 0, rm[buf][0], rm[buf][1]);
    else if (n == 1) put_row(px[1], 1, rm[buf][0], rm[buf][1]);
    else if (n == 2) { put_tr(0, tr[buf][0], tr[buf][1]); put_tr(1, tr[buf][0], tr[buf][1]); }
    else if (n == 3) { put_tr(2, tr[buf][0], tr[buf][1]); put_tr(3, tr[buf][0], tr[buf][1]); g_load_x(cn); }
    else if (n == 4 || n == 5) {
      const int k = n - 4;
      put_row(py[k], k, rm[buf][2], rm[buf][3]);
      const float k0 = (slab == 0 && plv[k] != 0u) ? 1.0f : 0.f;       // branch-free: the staging stays in the matrix instructions' basic block
      sb2.x = fmaf(k0, py[k].x, sb2.x); sb2.y = fmaf(k0, py[k].y, sb2.y); sb2.z = fmaf(k0, py[k].z, sb2.z); sb2.w = fmaf(k0, py[k].w, sb2.w);
    } else if (n == 6) { put_tr(0, tr[buf][2], tr[buf][3]); put_tr(1, tr[buf][2], tr[buf][3]); }
    else { put_tr(2, tr[buf][2], tr[buf][3]); put_tr(3, tr[buf][2], tr[buf][3]); g_load_y(cn, real); }
  };
  // operand fragments of one matrix-instruction group: [x hi, x lo, dy hi, dy lo]
  auto ld_rm = [&](int buf, int s, int rt, bfrag (&f)[4]) {
    const int off = (16 * rt + j) * WG_RM + (32 * s + 8 * g) * 2;
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const bfrag*>(rm[buf][i] + off);
  };
  auto ld_tr = [&](int buf, int u, bfrag (&f)[4]) {
    const int off = g * WG_TG + ((j & 3) * WG_TQ + 4 * u + (j >> 2)) * 16;
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = *reinterpret_cast<const bfrag*>(tr[buf][i] + off);
  };
  if (c_lo < c_hi) {
    g_load_x(c_lo); g_load_y(c_lo, true);
#pragma unroll
    for (int n = 0; n < 8; ++n) stage(n, 0, c_lo + 1 < c_hi ? c_lo + 1 : c_lo, c_lo + 1 < c_hi);
  }
  __syncthreads();
  for (long long c = c_lo; c < c_hi; ++c) {
    const int buf = (int)((c - c_lo) & 1);
    const long long cn = c + 2 < c_hi ? c + 2 : c;
    const bool rn = c + 2 < c_hi;
    bfrag fa[2][4];
    ld_rm(buf, 0, 0, fa[0]);
    // ---- recompute pre[row][hid] = x W1^T, dpre = dy W2 for this wave's hidden tile (A = activations, k = feature); the
    // fragments of group i + 1 are requested before the matrix instructions of group i
    f32x4 pre[2] = {rr_zero4(), rr_zero4()}, dpre[2] = {rr_zero4(), rr_zero4()};
#pragma unroll
    for (int i = 0; i < 8; ++i) {                              // i = 2 s + rt
      if (i + 1 < 8) ld_rm(buf, (i + 1) >> 1, (i + 1) & 1, fa[(i + 1) & 1]);
      else ld_tr(buf, 0, fa[0]);
      const int s = i >> 1, rt = i & 1;
      pre[rt] = td_mfma3<HALF>(fa[i & 1][0], fa[i & 1][1], W1h[s], W1l[s], pre[rt]);
      dpre[rt] = td_mfma3<HALF>(fa[i & 1][2], fa[i & 1][3], W2h[s], W2l[s], dpre[rt]);
      if ((i & 1) == 0) stage(i >> 1, buf ^ 1, cn, rn);       // (also for the last chunk: the rows past the end stage as zeros)
      __builtin_amdgcn_sched_barrier(0);
    }
    // C layout here: lane (hid = j, g) holds rows 16 rt + 4g + r.  As the B operand of the outer products (k = row) a lane's
    // eight values are rows {4g + r} u {16 + 4g + r}: positions 4 rt + r of row group g in the transposed images.
    bfrag Hh, Hl, Gh, Gl;
    {
      float hx[8], gx[8];
#pragma unroll
      for (int rt = 0; rt < 2; ++rt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pv = pre[rt][r] + b1v;
          hx[4 * rt + r] = fmaxf(pv, 0.f);
          gx[4 * rt + r] = pv > 0.f ? dpre[rt][r] : 0.f;
          ab1 += gx[4 * rt + r];
        }
      td_split8(hx, Hh, Hl);
      td_split8(gx, Gh, Gl);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (u + 1 < 8) ld_tr(buf, u + 1, fa[(u + 1) & 1]);
      aW2[u] = td_mfma3<HALF>(fa[u & 1][2], fa[u & 1][3], Hh, Hl, aW2[u]);      // [feat][hid] += dy^T H
      aW1[u] = td_mfma3<HALF>(fa[u & 1][0], fa[u & 1][1], Gh, Gl, aW1[u]);      // [feat][hid] += x^T dH  (= dW1^T)
      if ((u & 1) == 0) stage(4 + (u >> 1), buf ^ 1, cn, rn);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();                      // buffer buf is consumed by every wave, buffer buf ^ 1 is complete
  }
  // ---- epilogue: float atomics into the (zeroed) gradients.  (125 us per call when every wave-instruction of the dW1 half touched
  // 16 cache lines: lanes along the hidden unit = along dW1's ROWS.  Its tile goes through LDS (the row images are free: the loop's
  // last barrier is behind us; a wave only reads what it wrote) and out with the lanes along the features: 4 lines per instruction,
  // like dW2's, whose lanes already run along its rows.)
  {
    float* ep = reinterpret_cast<float*>(&rm[0][0][0]) + wave * (16 * 132);      // [16 hidden units][128 features + 4]
#pragma unroll
    for (int u = 0; u < 8; ++u) rr_st4(ep + j * 132 + 16 * u + 4 * g, make_float4(aW1[u][0], aW1[u][1], aW1[u][2], aW1[u][3]));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int h = 0; h < 16; ++h)
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const size_t o = (size_t)(16 * T0 + h) * RR_E + 64 * half + lane;
        if (wsp) wsp[o] = ep[h * 132 + 64 * half + lane];
        else atomicAdd(dW1 + o, ep[h * 132 + 64 * half + lane]);
      }
  }
#pragma unroll
  for (int u = 0; u < 8; ++u)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const size_t o = (size_t)(16 * u + 4 * g + r) * RR_FF + 16 * T0 + j;
      if (wsp) wsp[RR_FF * RR_E + o] = aW2[u][r];
      else atomicAdd(dW2 + o, aW2[u][r]);
    }
  {
    float v = ab1;
    v = rr_sum_g(v);      // (lane ^ 16, then lane ^ 32: v_permlane swaps)
    if (g == 0) atomicAdd(db1 + 16 * T0 + j, v);
  }
  if (slab == 0 && db2 != nullptr) {
    atomicAdd(db2 + 4 * c4 + 0, sb2.x); atomicAdd(db2 + 4 * c4 + 1, sb2.y);
    atomicAdd(db2 + 4 * c4 + 2, sb2.z); atomicAdd(db2 + 4 * c4 + 3, sb2.w);
  }
}

__global__ __launch_bounds__(256) void k_wgrad_reduce(const float* __restrict__ ws, float* __restrict__ dW1, float* __restrict__ dW2, int nsplit) {
  const int e = blockIdx.x * 256 + threadIdx.x;           // < 2 * 512 * 128
  const float* src = ws + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int m = 0;
  for (; m + 4 <= nsplit; m += 4) {
    s0 += src[(size_t)m * (2 * RR_FF * RR_E)]; s1 += src[(size_t)(m + 1) * (2 * RR_FF * RR_E)];
    s2 += src[(size_t)(m + 2) * (2 * RR_FF * RR_E)]; s3 += src[(size_t)(m + 3) * (2 * RR_FF * RR_E)];
  }
  for (; m < nsplit; ++m) s0 += src[(size_t)m * (2 * RR_FF * RR_E)];
  float* dst = e < RR_FF * RR_E ? dW1 + e : dW2 + (e - RR_FF * RR_E);
  *dst += (s0 + s1) + (s2 + s3);
}

// dW1 [512][128], db1 [512], dW2 [128][512], db2 [128]: ADDED to (caller zeroes them).  ws: NULL (float atomics) or 64 * 2 * 512 * 128
// floats for the row splits' partials of dW1 / dW2 (added up in a fixed order).
static int rr_mlp_wgrad_impl(bool half, const MlpWgradW* w, const float* X, const float* dY, float* dW1, float* db1, float* dW2, float* db2,
                             const uint32_t* meta, int nseg, int seg_rows, long long seg_stride, float* ws, hipStream_t st) {
  if (w == nullptr || w->w1n == nullptr || w->w2tn == nullptr || w->b1 == nullptr || X == nullptr || dY == nullptr ||
      dW1 == nullptr || db1 == nullptr || dW2 == nullptr)
    return RR_EINVAL;
  if (nseg <= 0 || seg_rows <= 0 || seg_stride < seg_rows) return RR_EINVAL;
  if (nseg > 1 && (long long)nseg * seg_rows >= (1ll << 40) / seg_rows) return RR_EINVAL;      // (the kernel's divide-free row map)
  RowSegs rs{nseg, seg_rows, seg_stride};
  const long long chunks = ((long long)nseg * seg_rows + 31) / 32;
  int nsplit = chunks >= 64 * 8 ? 64 : (chunks >= 64 ? 16 : 8);
  if (half) hipLaunchKernelGGL(k_mlp_wgrad<true>, dim3(4 * nsplit), dim3(512), 0, st, *w, X, dY, dW1, db1, dW2, db2, rs, nsplit, meta, ws);
  else hipLaunchKernelGGL(k_mlp_wgrad<false>, dim3(4 * nsplit), dim3(512), 0, st, *w, X, dY, dW1, db1, dW2, db2, rs, nsplit, meta, ws);
  if (ws != nullptr) hipLaunchKernelGGL(k_wgrad_reduce, dim3(2 * RR_FF * RR_E / 256), dim3(256), 0, st, ws, dW1, dW2, nsplit);
  return rr_check(hipGetLastError());
}
extern "C" int rr_mlp_wgrad(const MlpWgradW* w, const float* X, const float* dY, float* dW1, float* db1, float* dW2, float* db2,
                            const uint32_t* meta, int nseg, int seg_rows, long long seg_stride, float* ws, hipStream_t st) {
  return rr_mlp_wgrad_impl(false, w, X, dY, dW1, db1, dW2, db2, meta, nseg, seg_rows, seg_stride, ws, st);
}
// the same with ONE bf16 piece per operand (the opt-in 16-mixed training step; see td_mfma3)
extern "C" int rr_mlp_wgrad16(const MlpWgradW* w, const float* X, const float* dY, float* dW1, float* db1, float* dW2, float* db2,
                              const uint32_t* meta, int nseg, int seg_rows, long long seg_stride, float* ws, hipStream_t st) {
  return rr_mlp_wgrad_impl(true, w, X, dY, dW1, db1, dW2, db2, meta, nseg, seg_rows, seg_stride, ws, st);
}

// ------------------------------------------------------------------------------------------------ attention backward
struct DecAttnIO {
  const float* dg0;            // [rows][128] d loss / d (glimpse + query)
  const uint32_t* meta;        // [rows][8]
  const float* scal;           // [rows][4] VRP state scalars (NULL: none)
  const int64_t* first;        // [S*Bp] first node per rollout (ATSP TSPContext) or NULL
  const float *K, *V, *Kt;     // glimpse keys / values [Bp][N][128], keys transposed zero padded [Bp][128][112]
  const float *ctxA, *ctxB;    // step-context tables [Bp][N][128] (ctxA NULL unless `first`)
  const float* wstate;         // [nscal][128] state columns of project_context (NULL: none)
  float *dK, *dV, *dctxA, *dctxB;   // [Bp][N][128], written (not added)
  float* dwstate;              // [nscal][128], ADDED to with atomics
  int Bp, N, S, T, nscal;
  long long seg_stride;
};

// 16x16 tile held as C layout (lane (c = j, g) register r = X[4g + r][c])  ->  lane (a = j, g) register m = X[a][4g + m]
// (the A / B operand layout with the former lane index on the k axis), on the matrix pipe: register r as the A operand is
// A_r[i][k] = X[4k + r][i]; with the constant selector B_r[k][col] = [col == 4k + r] the four products sum to X[col][i], exact
// in fp32 (one non-zero term per output).  No LDS round trip, and sixteen of these per (tile, head) pipeline like any MFMA.
__device__ __forceinline__ f32x4 td_xpose(f32x4 v, const float (&sel)[4]) {
  f32x4 d = rr_zero4();
  d = rr_mfma(v[0], sel[0], d); d = rr_mfma(v[1], sel[1], d); d = rr_mfma(v[2], sel[2], d); d = rr_mfma(v[3], sel[3], d);
  return d;
}

// The same on the bf16 matrix pipe (BF = true, the default): every fp32 operand as two bf16 pieces (td_split4: x = hi + lo to 2^-17,
// a product keeps hi*hi + hi*lo + lo*hi like the pointer-MLP kernels above), one v_mfma_f32_16x16x16_bf16 per partial product:
// 24 matrix-pipe cycles per 16x16x16 product against 128 on the fp32 MFMA.  A piece is transposed exactly by ONE product with the
// identity (a bf16 value times 1.0 accumulates exactly in fp32 and converts back exactly).
typedef __bf16 rr_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x4 td_mfma16(rr_bf16x4 a, rr_bf16x4 b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0);
}
// (one asm block: hipcc scalarises the two-wide conversion — a convert per value plus packing, 22 instructions for four values
// against these 12.  Leading s_nop: an operand may come straight from a transcendental; trailing s_nop 1: the pieces feed a
// matrix instruction, and hipcc does not see the vector writes inside the block — profiles/r03/NOTES.md §2.)
__device__ __forceinline__ void td_split4(const float (&x)[4], rr_bf16x4& hi, rr_bf16x4& lo) {
  uint32_t h01, h23, l01, l23;
  float t0, t1, t2, t3;
  asm("s_nop 0\n\t"
      "v_cvt_pk_bf16_f32 %0, %8, %9\n\t"
      "v_cvt_pk_bf16_f32 %1, %10, %11\n\t"
      "v_lshlrev_b32 %4, 16, %0\n\t"
      "v_and_b32 %5, 0xffff0000, %0\n\t"
      "v_lshlrev_b32 %6, 16, %1\n\t"
      "v_and_b32 %7, 0xffff0000, %1\n\t"
      "v_sub_f32 %4, %8, %4\n\t"
      "v_sub_f32 %5, %9, %5\n\t"
      "v_sub_f32 %6, %10, %6\n\t"
      "v_sub_f32 %7, %11, %7\n\t"
      "v_cvt_pk_bf16_f32 %2, %4, %5\n\t"
      "v_cvt_pk_bf16_f32 %3, %6, %7\n\t"
      "s_nop 1"
      : "=&v"(h01), "=&v"(h23), "=&v"(l01), "=&v"(l23), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
      : "v"(x[0]), "v"(x[1]), "v"(x[2]), "v"(x[3]));
  const uint2 hv = make_uint2(h01, h23), lv = make_uint2(l01, l23);
  hi = __builtin_bit_cast(rr_bf16x4, hv); lo = __builtin_bit_cast(rr_bf16x4, lv);
}
__device__ __forceinline__ f32x4 td_mfma16x3(rr_bf16x4 ah, rr_bf16x4 al, rr_bf16x4 bh, rr_bf16x4 bl, f32x4 c) {
  c = td_mfma16(ah, bl, c);
  c = td_mfma16(al, bh, c);
  return td_mfma16(ah, bh, c);
}
// piece at lane (a = j, g), element m = X[a][4g + m]  ->  lane (c = j, g), element r = X[4g + r][c]
__device__ __forceinline__ rr_bf16x4 td_xpose16(rr_bf16x4 p, rr_bf16x4 ident) {
  const f32x4 d = td_mfma16(p, ident, rr_zero4());            // every element IS a bf16 value: packing = taking the high halves
  const float d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3];     // (__builtin_bit_cast on a vector ELEMENT reads element 0 every time: hipcc 7.2)
  const uint2 o = make_uint2(__builtin_amdgcn_perm(__float_as_uint(d1), __float_as_uint(d0), 0x07060302u),
                             __builtin_amdgcn_perm(__float_as_uint(d3), __float_as_uint(d2), 0x07060302u));
  return __builtin_bit_cast(rr_bf16x4, o);
}

// ------------------------------------------------------------------------------------------------ logits backward, split operands
// k_dec_logit_bwd on the fp16 / bf16 matrix pipe.  The fp32-MFMA kernel above spends its time on 448 matrix instructions of 32 cycles
// per tile and on 112 KB of L / L^T fragments every wave fetches from L2 for its ONE tile.  Here a workgroup (8 waves) keeps one
// instance's operands in LDS for all the tiles it runs: the rollout's own two-piece fp16 image of L (same products, same order
// as rr_rollout_w.inc's logits phase: the replayed logits are the rollout's) and L^T as [hi | lo] bf16 pieces for d g = d logits L
// (gradients leave the fp16 range; error 2^-16 of a product like the pointer-MLP kernels).  Per tile 112 + 112 matrix
// instructions of 8 / 16 cycles; what is left is the row traffic (g in, d g and d logits out).
__device__ __forceinline__ rr_bf16x4 td_lo4(rr_bf16x8 v) { return __builtin_shufflevector(v, v, 0, 1, 2, 3); }
__device__ __forceinline__ rr_bf16x4 td_hi4(rr_bf16x8 v) { return __builtin_shufflevector(v, v, 4, 5, 6, 7); }

template <bool DUR>
__global__ __launch_bounds__(512, 1) void k_dec_logit_bwd_s(DecLogitIO io, int tiles_per_inst, int chunks) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  char* limg = lds;                                  // fragment (kt, kk): A = [L~ hi | L~ lo] of keys 16kt + j, features 16kk + 4g ..
  char* timg = lds + TD_NT * 8 * 1024;               // fragment (u, kt):  A = [L^T hi | L^T lo] of features 16u + j, keys 16kt + 4g ..
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int b = blockIdx.x / chunks, ch = blockIdx.x - b * chunks;
  const int N = io.N, S = io.S, rows_b = io.T * S;
  for (int f = wave; f < TD_NT * 8; f += 8) {
    const int kt = f >> 3, kk = f & 7;
    int key = kt * 16 + j; key = key < N ? key : N - 1;
    rr_glds16((const char*)io.Ls + ((size_t)b * N + key) * (RR_E * 4) + (size_t)(16 * kk + 4 * g) * 4, limg + f * 1024);
    const int u = f / TD_NT, kq = f - u * TD_NT;
    const float4 v = rr_ld4(io.Lt + ((size_t)b * RR_E + 16 * u + j) * TD_LDK + 16 * kq + 4 * g);
    const float x4[4] = {v.x, v.y, v.z, v.w};
    rr_bf16x4 hi, lo;
    td_split4(x4, hi, lo);
    *reinterpret_cast<rr_bf16x8*>(timg + (f * 64 + lane) * 16) = rr_bf16x8{hi[0], hi[1], hi[2], hi[3], lo[0], lo[1], lo[2], lo[3]};
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  const int per = (tiles_per_inst + chunks - 1) / chunks;
  const int t_lo = ch * per, t_hi = min(t_lo + per, tiles_per_inst);
  const float inv_sqe = 1.0f / ((float)(1 << RR_KS) * sqrtf((float)RR_E)), inv_temp = 1.0f / io.temperature;     // (the L image carries 2^RR_KS)
  const float inv_sqg = 1.0f / sqrtf((float)RR_E);
  const bool clip = io.tanh_clip > 0.f;
  const float cs_clip = io.tanh_clip * inv_temp;
  float da_acc = 0.f, db_acc = 0.f;
  uint32_t nmw = 0;                         // nibble kt: which of this lane's four keys of key tile kt exist (key < N)
#pragma unroll
  for (int kt = 0; kt < TD_NT; ++kt) {
    const int left = N - (16 * kt + 4 * g);
    nmw |= (left >= 4 ? 15u : left <= 0 ? 0u : ((1u << left) - 1u)) << (4 * kt);
  }
  const __amdgpu_buffer_rsrc_t rD = rr_make_buf(io.D + (size_t)b * N * N, (unsigned)(N * N) * 4u);
  const __amdgpu_buffer_rsrc_t rT = rr_make_buf((DUR ? io.Dur : io.D) + (size_t)b * N * N, (unsigned)(N * N) * 4u);
  struct In { uint4 m0, m1; float4 f[8]; float gl; size_t m; bool vrow; };
  auto ld_in = [&](int tile) {
    In r;
    int q = tile * 16 + j;
    r.vrow = q < rows_b;
    q = r.vrow ? q : rows_b - 1;
    r.m = (size_t)b * (size_t)io.seg_stride + (size_t)q;
    const int t = q / S;
    r.m0 = *reinterpret_cast<const uint4*>(io.meta + r.m * 8);
    r.m1 = *reinterpret_cast<const uint4*>(io.meta + r.m * 8 + 4);
    r.gl = io.gll[(size_t)(q - t * S) * io.Bp + b];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) r.f[kk] = rr_ld4(io.g + r.m * RR_E + 16 * kk + 4 * g);
    return r;
  };
  In nx = ld_in(t_lo + wave < t_hi ? t_lo + wave : t_lo);
#pragma unroll 1
  for (int tile = t_lo + wave; tile < t_hi; tile += 8) {
    const In in = nx;
    const size_t m = in.m;
    const bool vrow = in.vrow;
    const int prev = min((int)in.m1.x, N - 1), target = (int)in.m1.y;
    const bool live = vrow && in.m1.z != 0u;
    const float gl = live ? in.gl : 0.f;
    // the bias rows of this tile, then the next tile's inputs: both land under the matrix instructions
    // (buffer loads: one 32-bit offset for the 28 values; keys past N read a neighbour row or, past the matrix, zero: they are masked)
    const unsigned drow = (unsigned)(prev * N + 4 * g) * 4u;
    float dd[TD_NT][4], tt[TD_NT][4];
#pragma unroll
    for (int kt = 0; kt < TD_NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dd[kt][r] = rr_bld1(rD, drow, (unsigned)(16 * kt + r) * 4u);
        tt[kt][r] = DUR ? rr_bld1(rT, drow, (unsigned)(16 * kt + r) * 4u) : 0.f;
      }
    rr_f16x8 FS[8];                                   // [g lo | g hi] per feature group (dead rows: zeros, and zeroed in place)
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const float fx[4] = {live ? in.f[kk].x : 0.f, live ? in.f[kk].y : 0.f, live ? in.f[kk].z : 0.f, live ? in.f[kk].w : 0.f};
      FS[kk] = rr_usplit4s(fx);
      if (vrow && !live) rr_st4(io.g + m * RR_E + 16 * kk + 4 * g, make_float4(0.f, 0.f, 0.f, 0.f));
    }
    // ---- logits^T[key][row] = L g^T (decoder.py:300-302): two chains per key tile (k = 16: hi hi; k = 32: hi lo + lo hi)
    f32x4 la[TD_NT];
    {
      const char* lb = limg + lane * 16;
      rr_f16x8 A[2][4];                              // half a key tile ahead
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) A[0][kk] = *reinterpret_cast<const rr_f16x8*>(lb + kk * 1024);
#pragma unroll
      for (int kt = 0; kt < TD_NT; ++kt) {
        f32x4 lt = rr_zero4(), ls = rr_zero4();
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int nxt = 2 * kt + hf + 1;
          if (nxt < 2 * TD_NT) {
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) A[nxt & 1][kk] = *reinterpret_cast<const rr_f16x8*>(lb + (nxt * 4 + kk) * 1024);
          }
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            lt = rr_mfma_f16k16(rr_lo4(A[hf][kk]), rr_hi4(FS[4 * hf + kk]), lt);
            ls = rr_mfma_f16(A[hf][kk], FS[4 * hf + kk], ls);
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) la[kt][q] = lt[q] + ls[q];
      }
    }
    nx = ld_in(tile + 8 < t_hi ? tile + 8 : tile);      // the next tile's rows land under the epilogue and the d g products
    __builtin_amdgcn_sched_barrier(0);
    // ---- bias, log(exp(.) + 1e-6) (decoder.py:187-198), 10 tanh, mask, temperature, log-softmax (decoding.py:341-361)
    const uint32_t mw[4] = {in.m0.x >> (4 * g), in.m0.y >> (4 * g), in.m0.z >> (4 * g), in.m0.w >> (4 * g)};
    float uu[TD_NT][4], vv[TD_NT][4];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < TD_NT; ++kt) {
      const uint32_t bits = (mw[kt >> 1] >> (16 * (kt & 1))) & (nmw >> (4 * kt));      // this lane's four keys of the tile
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // (the rollout's own forms, rr_rollout_w.inc: one v_exp_f32 on the base-2 argument, v_rcp_f32 — the replayed log-likelihood then
        //  repeats the rollout's; the corrected exponential and the IEEE divisions this used to spend were ~900 of the tile's vector instructions)
        const float x = la[kt][r] * inv_sqe - (io.alpha * dd[kt][r] + io.beta * tt[kt][r]);
        const float u = rr_exp_fast(x) + 1e-6f;
        uu[kt][r] = u;
        float v;
        if (clip) v = fmaf(__builtin_amdgcn_rcpf(fmaf(u, u, 1.0f)), -2.0f * cs_clip, cs_clip);      // tanh(log u) = (u^2-1)/(u^2+1)
        else v = rr_log(u) * inv_temp;
        v = (bits & (1u << r)) ? v : -INFINITY;
        vv[kt][r] = v;
        mx = fmaxf(mx, v);
      }
    }
    mx = rr_max_g(mx);
    if (mx == -INFINITY) mx = 0.f;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < TD_NT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) sum += rr_exp_fast(vv[kt][r] - mx);
    sum = rr_sum_g(sum);
    const float lse = rr_log(sum);
    float lpt = 0.f, da = 0.f, db = 0.f;
    rr_bf16x8 DT[TD_NT];                              // d logits / sqrt(E) as [lo | hi] bf16 pieces
#pragma unroll
    for (int kt = 0; kt < TD_NT; ++kt) {
      float dl4[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kt * 16 + 4 * g + r;
        const float v = vv[kt][r];
        const bool ok = v > -INFINITY;
        const float lp = v - mx - lse;
        if (key == target) lpt += ok ? lp : 0.f;
        const float p = ok ? rr_exp_fast(lp) : 0.f;
        float dv = gl * ((key == target ? 1.0f : 0.f) - p);          // d loss / d v (log-softmax picked at the target)
        dv = (ok && live) ? dv : 0.f;
        const float u = uu[kt][r];
        float dx;                                                    // through v(u), u = exp(x) + 1e-6
        if (clip) { const float rw = __builtin_amdgcn_rcpf(fmaf(u, u, 1.0f)); dx = dv * cs_clip * (4.0f * u * rw * rw) * (u - 1e-6f); }
        else dx = dv * inv_temp * (u - 1e-6f) * __builtin_amdgcn_rcpf(u);
        dl4[r] = dx * inv_sqg;
        da -= dx * dd[kt][r];
        db -= dx * tt[kt][r];
      }
      if (vrow) rr_st4(io.dlg + m * TD_LDK + 16 * kt + 4 * g, make_float4(dl4[0], dl4[1], dl4[2], dl4[3]));
      rr_bf16x4 hi, lo;
      td_split4(dl4, hi, lo);
      DT[kt] = rr_bf16x8{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
    }
    lpt = rr_sum_g(lpt);
    if (vrow && g == 0) io.logp[m] = live ? lpt : 0.f;
    da_acc += vrow ? da : 0.f; db_acc += vrow ? db : 0.f;
    // ---- dg^T[feat][row] = L^T dla  (A = [L^T hi | L^T lo], k = key): two chains per feature tile
    {
      const char* tb = timg + lane * 16;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        f32x4 c0 = rr_zero4(), c1 = rr_zero4();
#pragma unroll
        for (int kt = 0; kt < TD_NT; ++kt) {
          const rr_bf16x8 A = *reinterpret_cast<const rr_bf16x8*>(tb + (u * TD_NT + kt) * 1024);
          c0 = td_mfma16(td_lo4(A), td_hi4(DT[kt]), c0);
          c1 = rr_mfma_bf16(A, DT[kt], c1);
        }
        if (vrow) rr_st4(io.dg + m * RR_E + 16 * u + 4 * g, make_float4(c0[0] + c1[0], c0[1] + c1[1], c0[2] + c1[2], c0[3] + c1[3]));
      }
    }
  }
  da_acc = rr_wave_sum(da_acc); db_acc = rr_wave_sum(db_acc);
  if (lane == 0) { atomicAdd(io.dscal, da_acc); if (DUR) atomicAdd(io.dscal + 1, db_acc); }
}

static int rr_dec_logit_bwd_split(const DecLogitIO* io, int tiles, hipStream_t st) {
  // one workgroup per instance when there are enough instances to fill the chip twice over; otherwise an instance's tiles are cut
  // into chunks (each pays the 112 KB of operand staging again)
  int chunks = io->Bp >= 512 ? 1 : (512 + io->Bp - 1) / io->Bp;
  chunks = chunks < 1 ? 1 : chunks;
  if (chunks > (tiles + 7) / 8) chunks = (tiles + 7) / 8;
  const size_t shm = (size_t)2 * TD_NT * 8 * 1024;
  if (io->Dur) {
    (void)hipFuncSetAttribute((const void*)k_dec_logit_bwd_s<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(k_dec_logit_bwd_s<true>, dim3((unsigned)io->Bp * chunks), dim3(512), shm, st, *io, tiles, chunks);
  } else {
    (void)hipFuncSetAttribute((const void*)k_dec_logit_bwd_s<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(k_dec_logit_bwd_s<false>, dim3((unsigned)io->Bp * chunks), dim3(512), shm, st, *io, tiles, chunks);
  }
  return rr_check(hipGetLastError());
}

template <bool BF>
__global__ __launch_bounds__(256, BF ? 2 : 1) void k_dec_attn_bwd(DecAttnIO io) {
  // d query accumulators [node][16 dims of this head] (row stride TD_AS floats: 16-byte rows, banks spread): accA for ctxA[first]
  // shared by the waves (float atomics), accB for ctxB[current] ONE PER WAVE.  LDS float atomics retire ~0.7 lanes per cycle and CU
  // (they were 3.2 of this kernel's 7.6 ms), so the rows are tiled to need almost none: a tile = 16 consecutive decode steps of ONE
  // rollout, whose current nodes are all different (a node is visited once; the VRPs' depot is the exception and keeps its
  // atomic) — every lane adds its four values to its own 16 bytes of the wave's table with a plain read-modify-write, and the
  // rollout's first node is common to the tile: one 16-lane sum, four lanes' atomics.  The T % 16 left-over steps of the rollouts
  // are packed 16 / (T % 16) rollouts to a tile and take the atomic path.
  __shared__ __attribute__((aligned(16))) float accA[TD_LDK * TD_AS], accB[4][TD_LDK * TD_AS];
  // fp32: the four waves' dK / dV partials.  BF: during the loop the head's K / V / K^T tiles as [hi | lo] bf16 operands (21 KB, one
  // 16-byte entry per (operand, key tile, lane): two workgroups per CU need them out of the registers), afterwards two partial
  // images (the fold runs in two rounds).
  constexpr int RED_W = BF ? 2 : 4;
  __shared__ __attribute__((aligned(16))) float red[RED_W][2 * TD_NT][64 * 4];
  char* opnd = reinterpret_cast<char*>(&red[0][0][0]);      // BF: entry (o, kt, lane) at ((o * TD_NT + kt) * 64 + lane) * 16
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int idx = blockIdx.x, xcd = idx & 7, kq = idx >> 3;
  const int h = kq & 7, b = (kq >> 3) * 8 + xcd;          // the eight heads of an instance share an XCD (its L2 holds the rows)
  if (b >= io.Bp) return;
  const int N = io.N, S = io.S, rows_b = io.T * S;
  for (int i = tid; i < TD_LDK * TD_AS; i += 256) { accA[i] = 0.f; accB[0][i] = 0.f; accB[1][i] = 0.f; accB[2][i] = 0.f; accB[3][i] = 0.f; }
  __syncthreads();
  float sel[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) sel[r] = (j == 4 * g + r) ? 1.0f : 0.f;
  // per-head operands, resident for the whole instance
  float4 kf[TD_NT], vf[TD_NT], ktf[TD_NT];                 // fp32: resident in registers
  rr_bf16x4 ident;
#pragma unroll
  for (int r = 0; r < 4; ++r) ident[r] = (__bf16)sel[r];
  uint32_t nmw = 0;                         // nibble kt: which of this lane's four keys of key tile kt exist (key < N)
#pragma unroll
  for (int kt = 0; kt < TD_NT; ++kt) {
    const int left = N - (16 * kt + 4 * g);
    nmw |= (left >= 4 ? 15u : left <= 0 ? 0u : ((1u << left) - 1u)) << (4 * kt);
  }
  auto ld_opnd = [&](int kt, float4& kq4, float4& vq4, float4& tq4) {
    int key = kt * 16 + j; key = key < N ? key : N - 1;
    kq4 = rr_ld4(io.K + ((size_t)b * N + key) * RR_E + 16 * h + 4 * g);       // A[i = key][k = dim 4g+m]
    vq4 = rr_ld4(io.V + ((size_t)b * N + key) * RR_E + 16 * h + 4 * g);
    tq4 = rr_ld4(io.Kt + ((size_t)b * RR_E + 16 * h + j) * TD_LDK + 16 * kt + 4 * g);   // A[i = dim j][k = key 16kt+4g+m]
  };
  if constexpr (BF) {
    for (int kt = wave; kt < TD_NT; kt += 4) {
      float4 q3[3];
      ld_opnd(kt, q3[0], q3[1], q3[2]);
#pragma unroll
      for (int o = 0; o < 3; ++o) {
        const float x4[4] = {q3[o].x, q3[o].y, q3[o].z, q3[o].w};
        rr_bf16x4 hi, lo;
        td_split4(x4, hi, lo);
        const rr_bf16x8 e = {hi[0], hi[1], hi[2], hi[3], lo[0], lo[1], lo[2], lo[3]};
        *reinterpret_cast<rr_bf16x8*>(opnd + ((o * TD_NT + kt) * 64 + lane) * 16) = e;
      }
    }
    __syncthreads();
  } else {
#pragma unroll
    for (int kt = 0; kt < TD_NT; ++kt) ld_opnd(kt, kf[kt], vf[kt], ktf[kt]);
  }
  auto opnd_hi = [&](int o, int kt, rr_bf16x4& hi, rr_bf16x4& lo) {
    const rr_bf16x8 e = *reinterpret_cast<const rr_bf16x8*>(opnd + ((o * TD_NT + kt) * 64 + lane) * 16);
    hi = rr_bf16x4{e[0], e[1], e[2], e[3]}; lo = rr_bf16x4{e[4], e[5], e[6], e[7]};
  };
  f32x4 aK[TD_NT], aV[TD_NT];
#pragma unroll
  for (int kt = 0; kt < TD_NT; ++kt) { aK[kt] = rr_zero4(); aV[kt] = rr_zero4(); }
  float dws[4][4];
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int r = 0; r < 4; ++r) dws[k][r] = 0.f;
  const int nfull = io.T >> 4, rem = io.T & 15;          // per rollout: full 16-step tiles, left-over steps
  const int nft = S * nfull;                               // tile < nft: rollout tile % S, steps 16 (tile / S) + j
  const int lper = rem ? 16 / rem : 1;                     // rollouts per left-over tile
  const int ntile = nft + (rem ? (S + lper - 1) / lper : 0);
  const int lr = rem ? j / rem : 0, lst = rem ? j - lr * rem : 0;       // this lane's (rollout, step) slot in a left-over tile
  // Inputs of a tile are two dependent global round trips (meta -> gathers of the context rows): they are requested two /
  // one tiles ahead, so that a tile's arithmetic runs under the next tiles' loads.
  struct TMeta { uint4 m0, m1; size_t m; int fst; bool vrow; };      // (the rollout's first node rides with the meta words: its context row is a dependent gather)
  struct TVals { float4 qb, qa, dh, sv; int fst; };
  auto ld_meta = [&](int tile) {
    TMeta r;
    int q, sr;
    if (tile < nft) {                                      // (wave-uniform)
      const int tb = tile / S;
      sr = tile - tb * S;
      q = (16 * tb + j) * S + sr;
      r.vrow = true;
    } else {
      sr = (tile - nft) * lper + lr;
      r.vrow = lr < lper && sr < S;
      sr = sr < S ? sr : S - 1;
      q = (16 * nfull + lst) * S + sr;
    }
    r.m = (size_t)b * (size_t)io.seg_stride + (size_t)q;
    r.fst = io.first ? (int)io.first[(size_t)sr * io.Bp + b] : 0;
    r.m0 = *reinterpret_cast<const uint4*>(io.meta + r.m * 8);
    r.m1 = *reinterpret_cast<const uint4*>(io.meta + r.m * 8 + 4);
    return r;
  };
  auto ld_vals = [&](const TMeta& tm) {
    TVals v;
    const int prev = min((int)tm.m1.x, N - 1);
    v.qb = rr_ld4(io.ctxB + ((size_t)b * N + prev) * RR_E + 16 * h + 4 * g);
    v.fst = 0;
    v.qa = make_float4(0.f, 0.f, 0.f, 0.f);
    if (io.first) {
      v.fst = min(tm.fst, N - 1);
      v.qa = rr_ld4(io.ctxA + ((size_t)b * N + v.fst) * RR_E + 16 * h + 4 * g);
    }
    v.sv = io.nscal > 0 ? rr_ld4(io.scal + tm.m * 4) : make_float4(0.f, 0.f, 0.f, 0.f);
    v.dh = rr_ld4(io.dg0 + tm.m * RR_E + 16 * h + 4 * g);
    return v;
  };
  float4 wst[4];                            // state columns of project_context for this head (VRPs)
#pragma unroll
  for (int k = 0; k < 4; ++k) wst[k] = k < io.nscal ? rr_ld4(io.wstate + k * RR_E + 16 * h + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
  float* myB = &accB[wave][0];
  auto scatter_dq = [&](const float (&dq)[4], bool live, int prev, int fst, bool fulltile) {
    if (fulltile) {
      if (live && prev != 0) {                             // distinct nodes within the tile: plain read-modify-write
        float4 v = rr_ld4(myB + prev * TD_AS + 4 * g);
        v.x += dq[0]; v.y += dq[1]; v.z += dq[2]; v.w += dq[3];
        rr_st4(myB + prev * TD_AS + 4 * g, v);
      } else if (live) {
#pragma unroll
        for (int r = 0; r < 4; ++r) atomicAdd(myB + prev * TD_AS + 4 * g + r, dq[r]);
      }
      if (io.first) {                                      // one rollout: one first node
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = live ? dq[r] : 0.f;
          v = rr_sum16(v);
          if (j == 0) atomicAdd(&accA[fst * TD_AS + 4 * g + r], v);
        }
      }
    } else if (live) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        atomicAdd(myB + prev * TD_AS + 4 * g + r, dq[r]);
        if (io.first) atomicAdd(&accA[fst * TD_AS + 4 * g + r], dq[r]);
      }
    }
  };
  TMeta mA = ld_meta(wave < ntile ? wave : 0), mB = ld_meta(wave + 4 < ntile ? wave + 4 : 0);
  TVals vA = ld_vals(mA);
#pragma unroll 1
  for (int tile = wave; tile < ntile; tile += 4) {
    const TMeta tm = mA;
    const TVals tv = vA;
    {   // requests for the tiles after this one
      const TMeta mC = ld_meta(tile + 8 < ntile ? tile + 8 : 0);
      vA = ld_vals(mB);
      mA = mB; mB = mC;
    }
    __builtin_amdgcn_sched_barrier(0);
    const bool vrow = tm.vrow;
    const size_t m = tm.m;
    const uint4 m0 = tm.m0, m1 = tm.m1;
    const int prev = min((int)m1.x, N - 1);
    const bool live = vrow && m1.z != 0u;
    const uint32_t mw[4] = {m0.x, m0.y, m0.z, m0.w};
    const int fst = tv.fst;
    // ---- query slice of this head (TSPContext / VRPContext / MTVRPContext as table gathers, see rr_rollout_w.inc)
    float4 qv = tv.qb;
    qv.x += tv.qa.x; qv.y += tv.qa.y; qv.z += tv.qa.z; qv.w += tv.qa.w;
    float sc4[4] = {tv.sv.x, tv.sv.y, tv.sv.z, tv.sv.w};
    // Rows the rollout never reached (finished routes: live flag 0) were never written: their state scalars are whatever the
    // allocator left there, NaN patterns included, and a NaN query poisons dK of the whole instance through 0 x NaN (found by
    // training RCVRPTW for 1 600 steps: profiles/r04/NOTES.md).  Dead rows get a finite query; their dh is zeroed below.
    if (!live) { sc4[0] = 0.f; sc4[1] = 0.f; sc4[2] = 0.f; sc4[3] = 0.f; }
    if (io.nscal > 0) {
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < io.nscal) {
          const float4 wv = wst[k];
          qv.x = fmaf(wv.x, sc4[k], qv.x); qv.y = fmaf(wv.y, sc4[k], qv.y); qv.z = fmaf(wv.z, sc4[k], qv.z); qv.w = fmaf(wv.w, sc4[k], qv.w);
        }
    }
    float4 dh = tv.dh;
    if (!live) dh = make_float4(0.f, 0.f, 0.f, 0.f);
    if constexpr (BF) {
      const float LOG2E = 1.44269504088896341f;
      // scores in base 2: the query carries log2(e) / sqrt(head_dim); dK is scaled back by ln 2 when it is stored
      const float qs[4] = {qv.x * (0.25f * LOG2E), qv.y * (0.25f * LOG2E), qv.z * (0.25f * LOG2E), qv.w * (0.25f * LOG2E)};
      const float dhv[4] = {dh.x, dh.y, dh.z, dh.w};
      rr_bf16x4 qh, ql, hh, hl;
      td_split4(qs, qh, ql); td_split4(dhv, hh, hl);
      const uint32_t mwg[4] = {mw[0] >> (4 * g), mw[1] >> (4 * g), mw[2] >> (4 * g), mw[3] >> (4 * g)};
      // ---- scores, masked softmax (decoder.py:308-323), d weights = V dhv
      f32x4 a[TD_NT], ds[TD_NT];
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < TD_NT; ++kt) {
        rr_bf16x4 Kh, Kl, Vh, Vl;
        opnd_hi(0, kt, Kh, Kl); opnd_hi(1, kt, Vh, Vl);
        f32x4 c = td_mfma16x3(Kh, Kl, qh, ql, rr_zero4());
        ds[kt] = td_mfma16x3(Vh, Vl, hh, hl, rr_zero4());
        const uint32_t bits = (mwg[kt >> 1] >> (16 * (kt & 1))) & (nmw >> (4 * kt));       // this lane's four keys of the tile
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          c[r] = (bits & (1u << r)) ? c[r] : -INFINITY;
          mx = fmaxf(mx, c[r]);
        }
        a[kt] = c;
      }
      mx = rr_max_g(mx);
      if (mx == -INFINITY) mx = 0.f;
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < TD_NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { a[kt][r] = __builtin_amdgcn_exp2f(a[kt][r] - mx); sum += a[kt][r]; }
      sum = rr_sum_g(sum);
      const float inv = sum > 0.f ? 1.0f / sum : 0.f;
      float dot = 0.f;
#pragma unroll
      for (int kt = 0; kt < TD_NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { a[kt][r] *= inv; dot = fmaf(a[kt][r], ds[kt][r], dot); }
      dot = rr_sum_g(dot);
      // ---- d scores (softmax backward) as pieces; d query = dhv (residual, decoder.py:294) + K^T ds / sqrt(d)
      rr_bf16x4 sh[TD_NT], sl[TD_NT];
      f32x4 dq0 = rr_zero4();
#pragma unroll
      for (int kt = 0; kt < TD_NT; ++kt) {
        const float d4[4] = {a[kt][0] * (ds[kt][0] - dot), a[kt][1] * (ds[kt][1] - dot), a[kt][2] * (ds[kt][2] - dot),
                             a[kt][3] * (ds[kt][3] - dot)};
        td_split4(d4, sh[kt], sl[kt]);
        rr_bf16x4 Th, Tl;
        opnd_hi(2, kt, Th, Tl);
        dq0 = td_mfma16x3(Th, Tl, sh[kt], sl[kt], dq0);
      }
      float dq[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) dq[r] = fmaf(dq0[r], 0.25f, dhv[r]);
      scatter_dq(dq, live, prev, fst, tile < nft);
      if (live) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int r = 0; r < 4; ++r) dws[k][r] = fmaf(dq[r], sc4[k], dws[k][r]);
      }
      // ---- dK_h[key][dim] += ds[row][key] q[row][dim] / sqrt(d),  dV_h[key][dim] += a[row][key] dhv[row][dim]  (k = row)
      const rr_bf16x4 qTh = td_xpose16(qh, ident), qTl = td_xpose16(ql, ident);
      const rr_bf16x4 hTh = td_xpose16(hh, ident), hTl = td_xpose16(hl, ident);
#pragma unroll
      for (int kt = 0; kt < TD_NT; ++kt) {
        const float p4[4] = {a[kt][0], a[kt][1], a[kt][2], a[kt][3]};
        rr_bf16x4 ph, pl;
        td_split4(p4, ph, pl);
        const rr_bf16x4 dTh = td_xpose16(sh[kt], ident), dTl = td_xpose16(sl[kt], ident);
        const rr_bf16x4 pTh = td_xpose16(ph, ident), pTl = td_xpose16(pl, ident);
        aK[kt] = td_mfma16x3(qTh, qTl, dTh, dTl, aK[kt]);
        aV[kt] = td_mfma16x3(hTh, hTl, pTh, pTl, aV[kt]);
      }
    } else {
      const float qs[4] = {qv.x * 0.25f, qv.y * 0.25f, qv.z * 0.25f, qv.w * 0.25f};          // 1/sqrt(head_dim)
      const float dhv[4] = {dh.x, dh.y, dh.z, dh.w};
      // ---- scores, masked softmax (decoder.py:308-323), d weights = V dhv
      f32x4 a[TD_NT], ds[TD_NT];
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < TD_NT; ++kt) {
        f32x4 c = rr_zero4(), d = rr_zero4();
        c = rr_mfma(kf[kt].x, qs[0], c); c = rr_mfma(kf[kt].y, qs[1], c); c = rr_mfma(kf[kt].z, qs[2], c); c = rr_mfma(kf[kt].w, qs[3], c);
        d = rr_mfma(vf[kt].x, dhv[0], d); d = rr_mfma(vf[kt].y, dhv[1], d); d = rr_mfma(vf[kt].z, dhv[2], d); d = rr_mfma(vf[kt].w, dhv[3], d);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kt * 16 + 4 * g + r;
          const bool ok = key < N && ((mw[kt >> 1] >> (16 * (kt & 1) + 4 * g + r)) & 1u);
          c[r] = ok ? c[r] : -INFINITY;
          mx = fmaxf(mx, c[r]);
        }
        a[kt] = c; ds[kt] = d;
      }
      mx = rr_max_g(mx);
      if (mx == -INFINITY) mx = 0.f;
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < TD_NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { a[kt][r] = rr_exp(a[kt][r] - mx); sum += a[kt][r]; }
      sum = rr_sum_g(sum);
      const float inv = sum > 0.f ? 1.0f / sum : 0.f;
      float dot = 0.f;
#pragma unroll
      for (int kt = 0; kt < TD_NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { a[kt][r] *= inv; dot = fmaf(a[kt][r], ds[kt][r], dot); }
      dot = rr_sum_g(dot);
#pragma unroll
      for (int kt = 0; kt < TD_NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) ds[kt][r] = a[kt][r] * (ds[kt][r] - dot);          // d scores (softmax backward)
      // ---- d query = dhv (residual, decoder.py:294) + K^T ds / sqrt(d)
      f32x4 dq0 = rr_zero4(), dq1 = rr_zero4();
#pragma unroll
      for (int kt = 0; kt < TD_NT; ++kt) {
        dq0 = rr_mfma(ktf[kt].x, ds[kt][0], dq0); dq1 = rr_mfma(ktf[kt].y, ds[kt][1], dq1);
        dq0 = rr_mfma(ktf[kt].z, ds[kt][2], dq0); dq1 = rr_mfma(ktf[kt].w, ds[kt][3], dq1);
      }
      float dq[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) dq[r] = fmaf(dq0[r] + dq1[r], 0.25f, dhv[r]);
      scatter_dq(dq, live, prev, fst, tile < nft);
      if (live) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
#pragma unroll
          for (int r = 0; r < 4; ++r) dws[k][r] = fmaf(dq[r], sc4[k], dws[k][r]);
      }
      // ---- dK_h[key][dim] += ds[row][key] q[row][dim] / sqrt(d),  dV_h[key][dim] += a[row][key] dhv[row][dim]  (k = row)
      const f32x4 qT = td_xpose(f32x4{qs[0], qs[1], qs[2], qs[3]}, sel);
      const f32x4 hT = td_xpose(f32x4{dhv[0], dhv[1], dhv[2], dhv[3]}, sel);
#pragma unroll
      for (int kt = 0; kt < TD_NT; ++kt) {
        const f32x4 dT = td_xpose(ds[kt], sel);
        const f32x4 pT = td_xpose(a[kt], sel);
        aK[kt] = rr_mfma(qT[0], dT[0], aK[kt]); aV[kt] = rr_mfma(hT[0], pT[0], aV[kt]);
        aK[kt] = rr_mfma(qT[1], dT[1], aK[kt]); aV[kt] = rr_mfma(hT[1], pT[1], aV[kt]);
        aK[kt] = rr_mfma(qT[2], dT[2], aK[kt]); aV[kt] = rr_mfma(hT[2], pT[2], aV[kt]);
        aK[kt] = rr_mfma(qT[3], dT[3], aK[kt]); aV[kt] = rr_mfma(hT[3], pT[3], aV[kt]);
      }
  
    }
  }
  // ---- fold the four waves' dK_h / dV_h (fixed order), write the head's slice of every table
  if constexpr (BF) {
    __syncthreads();                      // every wave is done with the operand entries
    if (wave >= 2) {
#pragma unroll
      for (int kt = 0; kt < TD_NT; ++kt) {
        rr_st4(&red[wave - 2][kt][lane * 4], make_float4(aK[kt][0], aK[kt][1], aK[kt][2], aK[kt][3]));
        rr_st4(&red[wave - 2][TD_NT + kt][lane * 4], make_float4(aV[kt][0], aV[kt][1], aV[kt][2], aV[kt][3]));
      }
    }
    __syncthreads();
    if (wave < 2) {                       // (w0 + w2), (w1 + w3): a lane rewrites the 16 bytes it read
#pragma unroll
      for (int kt = 0; kt < TD_NT; ++kt) {
        const float4 o = rr_ld4(&red[wave][kt][lane * 4]), p = rr_ld4(&red[wave][TD_NT + kt][lane * 4]);
        rr_st4(&red[wave][kt][lane * 4], make_float4(aK[kt][0] + o.x, aK[kt][1] + o.y, aK[kt][2] + o.z, aK[kt][3] + o.w));
        rr_st4(&red[wave][TD_NT + kt][lane * 4], make_float4(aV[kt][0] + p.x, aV[kt][1] + p.y, aV[kt][2] + p.z, aV[kt][3] + p.w));
      }
    }
  } else {
#pragma unroll
    for (int kt = 0; kt < TD_NT; ++kt) {
      rr_st4(&red[wave][kt][lane * 4], make_float4(aK[kt][0], aK[kt][1], aK[kt][2], aK[kt][3]));
      rr_st4(&red[wave][TD_NT + kt][lane * 4], make_float4(aV[kt][0], aV[kt][1], aV[kt][2], aV[kt][3]));
    }
  }
  __syncthreads();
  for (int e = tid; e < 2 * TD_NT * 64; e += 256) {
    const int ti = e >> 6, ln = e & 63;
    float4 sacc = rr_ld4(&red[0][ti][ln * 4]);
#pragma unroll
    for (int wv = 1; wv < RED_W; ++wv) {
      const float4 o = rr_ld4(&red[wv][ti][ln * 4]);
      sacc.x += o.x; sacc.y += o.y; sacc.z += o.z; sacc.w += o.w;
    }
    const int kt = ti % TD_NT, key = 16 * kt + (ln & 15), gg = ln >> 4;
    if (BF && ti < TD_NT) { sacc.x *= 0.693147180559945309f; sacc.y *= 0.693147180559945309f; sacc.z *= 0.693147180559945309f; sacc.w *= 0.693147180559945309f; }
    if (key < N) rr_st4((ti < TD_NT ? io.dK : io.dV) + ((size_t)b * N + key) * RR_E + 16 * h + 4 * gg, sacc);
  }
  for (int e = tid; e < N * 16; e += 256) {
    const int n = e >> 4, d = e & 15;
    io.dctxB[((size_t)b * N + n) * RR_E + 16 * h + d] =
        (accB[0][n * TD_AS + d] + accB[1][n * TD_AS + d]) + (accB[2][n * TD_AS + d] + accB[3][n * TD_AS + d]);
    if (io.dctxA) io.dctxA[((size_t)b * N + n) * RR_E + 16 * h + d] = accA[n * TD_AS + d];
  }
  if (io.nscal > 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int r = 0; r < 4; ++r) if (k < io.nscal) {
        float v = dws[k][r];
        v = rr_sum16(v);
        if (j == 0) atomicAdd(io.dwstate + k * RR_E + 16 * h + 4 * g + r, v);
      }
  }
}

extern "C" int rr_dec_attn_bwd(const DecAttnIO* io, hipStream_t st) {
  if (io == nullptr || io->dg0 == nullptr || io->meta == nullptr || io->K == nullptr || io->V == nullptr || io->Kt == nullptr ||
      io->ctxB == nullptr || io->dK == nullptr || io->dV == nullptr || io->dctxB == nullptr)
    return RR_EINVAL;
  if ((io->first != nullptr) != (io->ctxA != nullptr) || (io->first != nullptr) != (io->dctxA != nullptr)) return RR_EINVAL;
  if (io->nscal < 0 || io->nscal > 4 || (io->nscal > 0 && (io->scal == nullptr || io->wstate == nullptr || io->dwstate == nullptr))) return RR_EINVAL;
  if (io->Bp <= 0 || io->N < 2 || io->N > TD_LDK || io->S < 1 || io->T < 1) return RR_EINVAL;
  const unsigned grid = (unsigned)((io->Bp + 7) / 8) * 64u;
  const char* ev = getenv("RR_ATTN_BWD_F32");                    // diagnostic: the fp32-MFMA kernel (read per call: tests switch it)
  const int f32only = ev ? atoi(ev) : 0;
  if (f32only) hipLaunchKernelGGL(k_dec_attn_bwd<false>, dim3(grid), dim3(256), 0, st, *io);
  else hipLaunchKernelGGL(k_dec_attn_bwd<true>, dim3(grid), dim3(256), 0, st, *io);
  return rr_check(hipGetLastError());
}
