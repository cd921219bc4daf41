// Hand-written backward of one AttnFree_Block of the encoder (rrnco/models/nn/attn_freenet.py:417-441: AFTFull :309-327,
// Normalization :78-116 (instance), TransformerFFN :330-357) for the REINFORCE step (BASELINE configs[4]) — what Lightning
// autograd does behind rrnco/models/rl.py:118-128 for the encoder.
//
// The training forward is the inference block kernel (csrc/rr_enc_w.inc) with EncSave stores; the backward of a block is a
// short chain of row-parallel kernels over the [Bp*N, 128] activations (everything fits the 256 MB Infinity Cache):
//   k_inorm_bwd    InstanceNorm1d backward per instance (statistics recomputed from the saved input), d gamma / d beta
//   k_linear_rows  out = x W^T (+ bias) (+ out): Linear forward / input gradient (W^T packed) on the fp32 MFMA, optional column
//                  sums of x (the bias gradient of the layer whose output gradient x is)
//   k_aft_bwd      AFTFull mixing backward per instance: d q, d k, d v and d (NAB bias) [N][N]
// plus csrc/rr_train_dec.hip's k_mlp_rows / k_mlp_wgrad (the FFN is the pointer MLP's shape) and k_gemm_tn (dW = dY^T X), and
// csrc/rr_train.hip's NAB backward.
#include "rr_common.h"

#define TE_LDI 116                      // row stride (floats) of the [feature][node] LDS images (as EW_LDN)
#define TE_NP 112                       // padded node count of the transposed exp(softmax(bias)) image

// ------------------------------------------------------------------------------------------------ InstanceNorm1d backward
// y = gamma * (x - mean) * rstd + beta over the node axis, per instance and feature (biased variance, eps 1e-5).
// dx = gamma rstd (dy - mean_n(dy) - xhat mean_n(dy xhat)); dgamma += sum dy xhat; dbeta += sum dy.
// dy = dy1 (+ dy2).  `accumulate`: dx is added to what dx_out holds.
__global__ __launch_bounds__(1024) void k_inorm_bwd(const float* __restrict__ x, const float* __restrict__ dy1, const float* __restrict__ dy2,
                                                   const float* __restrict__ gamma, float* __restrict__ dx_out,
                                                   float* __restrict__ dgamma, float* __restrict__ dbeta, int N, int accumulate) {
  __shared__ float red[3][1024];
  const int b = blockIdx.x, tid = threadIdx.x, f = tid & 127, half = tid >> 7;      // eight node residues (`half` keeps its name from the two-residue form): 1 024 threads per
  // instance — at 256 the launch had 8 waves per CU in flight and ran at 1.6 TB/s (48 us per call; 512 threads: 36, 1 024: 30)
  const size_t base = (size_t)b * N * RR_E + f;
  constexpr int MAXR = 14;               // nodes per thread (N <= 112)
  float xv[MAXR], dv[MAXR], old[MAXR];
  float s0 = 0.f;
  // every load unconditional on a clamped node, the test applied to the VALUE: with the loads under `if (n < N)` hipcc gave each node its
  // own branch with a full vmcnt wait at the join — 14 serialised round trips per thread (profiles/r06/NOTES.md section 7)
  const float* d2 = dy2 ? dy2 : dy1;
#pragma unroll
  for (int i = 0; i < MAXR; ++i) {
    const int n = 8 * i + half;
    const size_t at = base + (size_t)(n < N ? n : N - 1) * RR_E;
    const float xa = x[at], da = dy1[at], db = d2[at];
    old[i] = accumulate ? dx_out[at] : 0.f;          // (read before the first barrier; the stores come after the last)
    xv[i] = n < N ? xa : 0.f;
    dv[i] = n < N ? da + (dy2 ? db : 0.f) : 0.f;
    s0 += xv[i];
  }
  red[0][tid] = s0;
  __syncthreads();
  const float inv_n = 1.0f / (float)N;
  const float mean = (((red[0][f] + red[0][128 + f]) + (red[0][256 + f] + red[0][384 + f])) + ((red[0][512 + f] + red[0][640 + f]) + (red[0][768 + f] + red[0][896 + f]))) * inv_n;
  float q = 0.f, s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < MAXR; ++i) {
    const int n = 8 * i + half;
    const float d = n < N ? xv[i] - mean : 0.f;
    xv[i] = d;
    q = fmaf(d, d, q); s1 += dv[i]; s2 = fmaf(dv[i], d, s2);
  }
  __syncthreads();
  red[0][tid] = q; red[1][tid] = s1; red[2][tid] = s2;
  __syncthreads();
  const float var = (((red[0][f] + red[0][128 + f]) + (red[0][256 + f] + red[0][384 + f])) + ((red[0][512 + f] + red[0][640 + f]) + (red[0][768 + f] + red[0][896 + f]))) * inv_n;
  const float rstd = 1.0f / sqrtf(var + 1e-5f);
  const float t1 = (((red[1][f] + red[1][128 + f]) + (red[1][256 + f] + red[1][384 + f])) + ((red[1][512 + f] + red[1][640 + f]) + (red[1][768 + f] + red[1][896 + f])));
  const float t2 = (((red[2][f] + red[2][128 + f]) + (red[2][256 + f] + red[2][384 + f])) + ((red[2][512 + f] + red[2][640 + f]) + (red[2][768 + f] + red[2][896 + f]))) * rstd;             // sum dy xhat
  const float gm = gamma[f];
  const float m1 = t1 * inv_n, m2 = t2 * inv_n;
#pragma unroll
  for (int i = 0; i < MAXR; ++i) {
    const int n = 8 * i + half;
    if (n < N) {
      const float xh = xv[i] * rstd;
      float v = gm * rstd * (dv[i] - m1 - xh * m2);
      if (accumulate) v += old[i];
      dx_out[base + (size_t)n * RR_E] = v;
    }
  }
  if (half == 0) { atomicAdd(dgamma + f, t2); atomicAdd(dbeta + f, t1); }
}

extern "C" int rr_inorm_bwd(const float* x, const float* dy1, const float* dy2, const float* gamma, float* dx, float* dgamma,
                            float* dbeta, int Bp, int N, int accumulate, hipStream_t st) {
  if (x == nullptr || dy1 == nullptr || gamma == nullptr || dx == nullptr || dgamma == nullptr || dbeta == nullptr) return RR_EINVAL;
  if (Bp <= 0 || N < 1 || N > 112) return RR_EINVAL;
  hipLaunchKernelGGL(k_inorm_bwd, dim3(Bp), dim3(1024), 0, st, x, dy1, dy2, gamma, dx, dgamma, dbeta, N, accumulate);
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ batch norm backward (TRAINING statistics)
// nn.BatchNorm1d over the M = Bp * N rows in train mode (attn_freenet.py:82-83, 102-103), x = the norm's input, dy = dy1 (+ dy2):
//   xh = (x - mean) rstd,  d beta = sum dy,  d gamma = sum dy xh,  dx = gamma rstd (dy - d beta / M - xh d gamma / M)   (+ dx when accumulating)
// Two launches: the four per-feature sums (x, x^2, dy, dy x) in float64 atomics (ws[512], zeroed here), then the map.
__global__ __launch_bounds__(256) void k_bn_bwd_stats(const float* __restrict__ x, const float* __restrict__ dy1, const float* __restrict__ dy2,
                                                      double* __restrict__ ws, long long M) {
  __shared__ double red[4][256];
  const int tid = threadIdx.x, f = tid & 127, half = tid >> 7;
  double a = 0.0, b = 0.0, c = 0.0, d = 0.0;
  for (long long m = (long long)blockIdx.x * 2 + half; m < M; m += (long long)gridDim.x * 2) {
    const double xv = (double)x[m * RR_E + f];
    const double g = (double)(dy1[m * RR_E + f] + (dy2 ? dy2[m * RR_E + f] : 0.f));
    a += xv; b += xv * xv; c += g; d += g * xv;
  }
  red[0][tid] = a; red[1][tid] = b; red[2][tid] = c; red[3][tid] = d;
  __syncthreads();
  if (half == 0) {
#pragma unroll
    for (int q = 0; q < 4; ++q) atomicAdd(ws + 128 * q + f, red[q][f] + red[q][128 + f]);
  }
}
__global__ __launch_bounds__(256) void k_bn_bwd_apply(const float* __restrict__ x, const float* __restrict__ dy1, const float* __restrict__ dy2,
                                                      const float* __restrict__ gamma, float* __restrict__ dx, float* __restrict__ dgamma,
                                                      float* __restrict__ dbeta, const double* __restrict__ ws, long long M, int accumulate) {
  const int tid = threadIdx.x, f = tid & 127, half = tid >> 7;
  const double inv = 1.0 / (double)M;
  const double mean = ws[f] * inv;
  const double var = fmax(ws[128 + f] * inv - mean * mean, 0.0);
  const float mu = (float)mean, rstd = 1.0f / sqrtf((float)var + 1e-5f);
  const double sdy = ws[256 + f], sdyx = ws[384 + f];
  const float dbt = (float)sdy, dgm = (float)((sdyx - mean * sdy) * (double)rstd);
  const float k1 = (float)(sdy * inv), k2 = (float)((sdyx - mean * sdy) * (double)rstd * inv);
  const float gr = gamma[f] * rstd;
  for (long long m = (long long)blockIdx.x * 2 + half; m < M; m += (long long)gridDim.x * 2) {
    const float g = dy1[m * RR_E + f] + (dy2 ? dy2[m * RR_E + f] : 0.f);
    const float xh = (x[m * RR_E + f] - mu) * rstd;
    const float v = gr * (g - k1 - xh * k2);
    dx[m * RR_E + f] = accumulate ? dx[m * RR_E + f] + v : v;
  }
  if (blockIdx.x == 0 && half == 0) { atomicAdd(dgamma + f, dgm); atomicAdd(dbeta + f, dbt); }
}
extern "C" int rr_bnorm_bwd(const float* x, const float* dy1, const float* dy2, const float* gamma, float* dx, float* dgamma,
                            float* dbeta, double* ws, long long M, int accumulate, hipStream_t st) {
  if (x == nullptr || dy1 == nullptr || gamma == nullptr || dx == nullptr || dgamma == nullptr || dbeta == nullptr || ws == nullptr || M <= 0)
    return RR_EINVAL;
  if (hipMemsetAsync(ws, 0, 512 * sizeof(double), st) != hipSuccess) return RR_ELAUNCH;
  const long long want = (M + 1) / 2;
  const unsigned grid = (unsigned)(want < 2048 ? want : 2048);
  hipLaunchKernelGGL(k_bn_bwd_stats, dim3(grid), dim3(256), 0, st, x, dy1, dy2, ws, M);
  hipLaunchKernelGGL(k_bn_bwd_apply, dim3(grid), dim3(256), 0, st, x, dy1, dy2, gamma, dx, dgamma, dbeta, ws, M, accumulate);
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ Linear on rows (128 -> 128)
// out[m][n] = sum_k x[m][k] W[n][k] (+ bias[n]) (+ out[m][n]);  W as an fp32 MFMA A operand pack [8][8][64][4] (packing.pack_a).
// One wave = 64 rows (four 16-row tiles as B operands in registers), the eight output tiles in turn.
// colsum (optional) += sum_m x[m][:]  (the bias gradient when x is an output gradient).
__global__ __launch_bounds__(256, 2) void k_linear_rows(const float4* __restrict__ Wp, const float* __restrict__ bias,
                                                        const float* __restrict__ X, float* __restrict__ out, long long M,
                                                        int accumulate, float* __restrict__ colsum) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  __shared__ float cs_s[RR_E];
  const long long r0 = ((long long)blockIdx.x * 4 + wave) * 64;
  if (colsum != nullptr) {                    // wave-uniform
    if (threadIdx.x < RR_E) cs_s[threadIdx.x] = 0.f;
    __syncthreads();
  }
  const bool wvalid = r0 < M;
  const __amdgpu_buffer_rsrc_t rW = rr_make_buf(Wp, RR_E * RR_E * 4);
  const unsigned lane16 = (unsigned)lane * 16u;
  f32x4 x[4][8];
  long long row[4]; bool vr[4];
  f32x4 cs[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) cs[kk] = rr_zero4();
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
    row[rt] = r0 + 16 * rt + j;
    vr[rt] = wvalid && row[rt] < M;
    const long long rc = vr[rt] ? row[rt] : M - 1;
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
      const float4 v = rr_ld4(X + rc * RR_E + 16 * kk + 4 * g);
      x[rt][kk] = f32x4{v.x, v.y, v.z, v.w};
      if (vr[rt]) cs[kk] += x[rt][kk];
    }
  }
  if (colsum != nullptr) {
#pragma unroll
    for (int kk = 0; kk < 8; ++kk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = cs[kk][r];
        v = rr_sum16(v);
        if (j == 0) atomicAdd(&cs_s[16 * kk + 4 * g + r], v);
      }
    __syncthreads();                          // one global atomic per workgroup and column (800 waves on 128 addresses serialise)
    if (threadIdx.x < RR_E) atomicAdd(colsum + threadIdx.x, cs_s[threadIdx.x]);
  }
  if (!wvalid) return;
  float4 a[8], an[8];
#pragma unroll
  for (int kk = 0; kk < 8; ++kk) a[kk] = rr_bld4(rW, lane16, kk * 1024u);
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    if (t + 1 < 8) {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) an[kk] = rr_bld4(rW, lane16, (unsigned)((t + 1) * 8 + kk) * 1024u);
    }
    float4 bb = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias != nullptr) bb = rr_ld4(bias + 16 * t + 4 * g);
    __builtin_amdgcn_sched_barrier(0);
    f32x4 c[4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) c[rt] = f32x4{bb.x, bb.y, bb.z, bb.w};
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) c[rt] = rr_mfma(a[kk].x, x[rt][kk][0], c[rt]);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) c[rt] = rr_mfma(a[kk].y, x[rt][kk][1], c[rt]);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) c[rt] = rr_mfma(a[kk].z, x[rt][kk][2], c[rt]);
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) c[rt] = rr_mfma(a[kk].w, x[rt][kk][3], c[rt]);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
      if (vr[rt]) {
        float* dst = out + row[rt] * RR_E + 16 * t + 4 * g;
        float4 v = make_float4(c[rt][0], c[rt][1], c[rt][2], c[rt][3]);
        if (accumulate) { const float4 o = rr_ld4(dst); v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w; }
        rr_st4(dst, v);
      }
    if (t + 1 < 8) {
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) a[kk] = an[kk];
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

extern "C" int rr_linear_rows(const void* Wp, const float* bias, const float* X, float* out, long long M, int accumulate,
                              float* colsum, hipStream_t st) {
  if (Wp == nullptr || X == nullptr || out == nullptr || M <= 0) return RR_EINVAL;
  const unsigned grid = (unsigned)((M + 255) / 256);
  hipLaunchKernelGGL(k_linear_rows, dim3(grid), dim3(256), 0, st, (const float4*)Wp, bias, X, out, M, accumulate, colsum);
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ AFTFull backward
struct AftBwdIO {
  const float *dy;                        // [Bp][N][128] d loss / d (sigmoid(q) num / den)
  const float *q, *ek, *v, *num, *den;    // saved by the training forward (EncSave)
  const float *eaT;                       // [Bp][112][112] exp(softmax(bias)) transposed
  float *dq, *dk, *dv;                    // [Bp][N][128]
  float *dbias;                           // [Bp][N][N] d loss / d (alpha * NAB), row i = query node
  int N;
};

// sum over the 16 lanes of a row (same g) in every lane
__device__ __forceinline__ float te_rowsum(float v) {
  return rr_sum16(v);      // (DPP: rr_common.h)
}

// One workgroup = one instance, wave w = node tile w (rows i of the first half, columns j of the second).
//   mix = num / den, y = sigmoid(q) mix:  dq = dy mix s (1 - s);  dnum = dy s / den;  dden = -dy s mix / den
//   ea = exp(softmax_j(bias)):  dea[i][j] = dnum[i] . (ek v)[j] + dden[i] . ek[j];  dbias = sa (dea ea - sum_j sa dea ea), sa = log ea
//   d(ek v)[j] = sum_i ea[i][j] dnum[i];  dek = sum_i ea[i][j] dden[i] + d(ek v) v;  dv = d(ek v) ek
//   ek = exp(softmax_nodes(k)):  dk = sk (dek ek - sum_nodes sk dek ek), sk = log ek
template <int NT>
__global__ __launch_bounds__(64 * NT, 1) void k_aft_bwd(AftBwdIO io) {
  __shared__ __attribute__((aligned(16))) float img[2 * RR_E * TE_LDI];          // dnum^T, dden^T as [feature][node i]
  __shared__ float part[NT][RR_E];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int N = io.N;
  const int node = 16 * wave + j;
  const bool nvalid = node < N;
  const int nc = nvalid ? node : N - 1;
  const size_t roff = ((size_t)b * N + nc) * RR_E + 4 * g;
  const float* eaT = io.eaT + (size_t)b * TE_NP * TE_NP;
  // ---- phase 0: elementwise part for this wave's rows i; images of dnum, dden
  f32x4 dnum[8], dden[8];
  {
    float* pn = img + (4 * g) * TE_LDI + node;
    float* pd = pn + RR_E * TE_LDI;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const float4 dy = rr_ld4(io.dy + roff + 16 * t), qq = rr_ld4(io.q + roff + 16 * t);
      const float4 nm = rr_ld4(io.num + roff + 16 * t), dn = rr_ld4(io.den + roff + 16 * t);
      const float dyv[4] = {dy.x, dy.y, dy.z, dy.w}, qv[4] = {qq.x, qq.y, qq.z, qq.w};
      const float nv[4] = {nm.x, nm.y, nm.z, nm.w}, dv_[4] = {dn.x, dn.y, dn.z, dn.w};
      float dqv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = 1.0f / (1.0f + rr_exp(-qv[r]));
        const float rd = 1.0f / dv_[r];
        const float mix = nv[r] * rd;
        const float dm = dyv[r] * s;
        dqv[r] = dyv[r] * mix * s * (1.0f - s);
        dnum[t][r] = nvalid ? dm * rd : 0.f;
        dden[t][r] = nvalid ? -dm * mix * rd : 0.f;
        pn[(16 * t + r) * TE_LDI] = dnum[t][r];
        pd[(16 * t + r) * TE_LDI] = dden[t][r];
      }
      if (nvalid) rr_st4(io.dq + roff + 16 * t, make_float4(dqv[0], dqv[1], dqv[2], dqv[3]));
    }
  }
  // ---- phase 1: dea^T[j][i] for this wave's rows i, all j; row-softmax backward -> dbias[i][:]
  {
    float sa[NT][4], dsa[NT][4];
    float dot = 0.f;
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
      int key = 16 * jt + j; key = key < N ? key : N - 1;
      const float* pe = io.ek + ((size_t)b * N + key) * RR_E + 4 * g;
      const float* pv = io.v + ((size_t)b * N + key) * RR_E + 4 * g;
      f32x4 c0 = rr_zero4(), c1 = rr_zero4();
#pragma unroll
      for (int kk = 0; kk < 8; ++kk) {
        const float4 e = rr_ld4(pe + 16 * kk), vv = rr_ld4(pv + 16 * kk);
        c0 = rr_mfma(e.x * vv.x, dnum[kk][0], c0); c1 = rr_mfma(e.x, dden[kk][0], c1);
        c0 = rr_mfma(e.y * vv.y, dnum[kk][1], c0); c1 = rr_mfma(e.y, dden[kk][1], c1);
        c0 = rr_mfma(e.z * vv.z, dnum[kk][2], c0); c1 = rr_mfma(e.z, dden[kk][2], c1);
        c0 = rr_mfma(e.w * vv.w, dnum[kk][3], c0); c1 = rr_mfma(e.w, dden[kk][3], c1);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jj = 16 * jt + 4 * g + r;
        const float ea = eaT[(size_t)jj * TE_NP + node];                // zero outside N x N
        const bool ok = jj < N && nvalid;
        sa[jt][r] = ok ? rr_log(ea) : 0.f;
        dsa[jt][r] = ok ? (c0[r] + c1[r]) * ea : 0.f;
        dot = fmaf(sa[jt][r], dsa[jt][r], dot);
      }
    }
    dot = rr_sum_g(dot);
    if (nvalid) {
      float* dst = io.dbias + ((size_t)b * N + node) * N;
#pragma unroll
      for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int jj = 16 * jt + 4 * g + r;
          if (jj < N) dst[jj] = sa[jt][r] * (dsa[jt][r] - dot);
        }
    }
  }
  __syncthreads();                      // images complete
  // ---- phase 2: this wave's nodes j: d(ek v), dek, dv, dk
  {
    float4 ef[NT];                      // B operand: ea[i][j] for k = i = 16 it + 4g + m, col j = this lane's node
#pragma unroll
    for (int it = 0; it < NT; ++it) {
      ef[it] = rr_ld4(eaT + (size_t)node * TE_NP + 16 * it + 4 * g);     // node < 112: inside the padded image
      if (!nvalid) ef[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    f32x4 dsk[8], skv[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      f32x4 c0 = rr_zero4(), c1 = rr_zero4();
      const float* pa = img + (16 * t + j) * TE_LDI + 4 * g;
#pragma unroll
      for (int it = 0; it < NT; ++it) {
        const float4 an = rr_ld4(pa + 16 * it), ad = rr_ld4(pa + RR_E * TE_LDI + 16 * it);
        c0 = rr_mfma(an.x, ef[it].x, c0); c1 = rr_mfma(ad.x, ef[it].x, c1);
        c0 = rr_mfma(an.y, ef[it].y, c0); c1 = rr_mfma(ad.y, ef[it].y, c1);
        c0 = rr_mfma(an.z, ef[it].z, c0); c1 = rr_mfma(ad.z, ef[it].z, c1);
        c0 = rr_mfma(an.w, ef[it].w, c0); c1 = rr_mfma(ad.w, ef[it].w, c1);
      }
      const float4 e = rr_ld4(io.ek + roff + 16 * t), vv = rr_ld4(io.v + roff + 16 * t);
      const float ev[4] = {e.x, e.y, e.z, e.w}, vx[4] = {vv.x, vv.y, vv.z, vv.w};
      float dvv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dvv[r] = c0[r] * ev[r];
        const float dek = fmaf(c0[r], vx[r], c1[r]);
        skv[t][r] = nvalid ? rr_log(ev[r]) : 0.f;
        dsk[t][r] = nvalid ? dek * ev[r] : 0.f;
      }
      if (nvalid) rr_st4(io.dv + roff + 16 * t, make_float4(dvv[0], dvv[1], dvv[2], dvv[3]));
    }
    // per-feature sum over ALL nodes of sk dsk: 16 lanes of the wave, then the waves (fixed order)
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float s = te_rowsum(skv[t][r] * dsk[t][r]);
        if (j == 0) part[wave][16 * t + 4 * g + r] = s;
      }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float dkv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float tot = 0.f;
#pragma unroll
        for (int wv = 0; wv < NT; ++wv) tot += part[wv][16 * t + 4 * g + r];
        dkv[r] = skv[t][r] * (dsk[t][r] - tot);
      }
      if (nvalid) rr_st4(io.dk + roff + 16 * t, make_float4(dkv[0], dkv[1], dkv[2], dkv[3]));
    }
  }
}

extern "C" int rr_aft_bwd(const AftBwdIO* io, int Bp, hipStream_t st) {
  if (io == nullptr || io->dy == nullptr || io->q == nullptr || io->ek == nullptr || io->v == nullptr || io->num == nullptr ||
      io->den == nullptr || io->eaT == nullptr || io->dq == nullptr || io->dk == nullptr || io->dv == nullptr || io->dbias == nullptr)
    return RR_EINVAL;
  const int N = io->N;
  if (Bp <= 0 || N < 2 || N > RR_MAXN) return RR_EINVAL;
  if (N <= 32) hipLaunchKernelGGL((k_aft_bwd<2>), dim3(Bp), dim3(128), 0, st, *io);
  else if (N <= 64) hipLaunchKernelGGL((k_aft_bwd<4>), dim3(Bp), dim3(256), 0, st, *io);
  else hipLaunchKernelGGL((k_aft_bwd<7>), dim3(Bp), dim3(448), 0, st, *io);
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ init embedding backward (ATSP)
// rrnco/models/env_embeddings/atsp.py:69-121 differentiated on kernels: the narrow Linear maps of the coordinates / the sorted sampled
// distances (k_linear_smallk; their weight gradients are rr_gemm_tn products), the 256 -> 256 layer of ContextualGating as four
// 128 x 128 rr_linear_rows blocks each way, and the scalar gate in between (k_gate_bwd).  Host side: models/init_backward.py.

// out[m][0..127] = bias + sum_{k < K} X[m][k] W[n][k],  K <= 32, X rows ldx floats apart.  A workgroup = 8 rows x 32 lanes of four
// output features; W^T in LDS.
__global__ __launch_bounds__(256) void k_linear_smallk(const float* __restrict__ X, int ldx, int K, const float* __restrict__ W,
                                                       const float* __restrict__ bias, float* __restrict__ out, long long M) {
  __shared__ __attribute__((aligned(16))) float Wt[32 * RR_E];
  const int tid = threadIdx.x, l = tid & 31, slot = tid >> 5;
  for (int i = tid; i < K * RR_E; i += 256) { const int n = i / K, k = i - n * K; Wt[k * RR_E + n] = W[i]; }
  __syncthreads();
  const float4 b4 = bias ? rr_ld4(bias + 4 * l) : make_float4(0.f, 0.f, 0.f, 0.f);
  for (long long m = (long long)blockIdx.x * 8 + slot; m < M; m += (long long)gridDim.x * 8) {
    const float* x = X + m * ldx;
    float4 a = b4;
    for (int k = 0; k < K; ++k) {
      const float xv = x[k];
      const float4 w = rr_ld4(Wt + k * RR_E + 4 * l);
      a.x = fmaf(xv, w.x, a.x); a.y = fmaf(xv, w.y, a.y); a.z = fmaf(xv, w.z, a.z); a.w = fmaf(xv, w.w, a.w);
    }
    rr_st4(out + m * RR_E + 4 * l, a);
  }
}
extern "C" int rr_linear_smallk(const float* X, int ldx, int K, const float* W, const float* bias, float* out, long long M, hipStream_t st) {
  if (X == nullptr || W == nullptr || out == nullptr || M <= 0 || K < 1 || K > 32 || ldx < K) return RR_EINVAL;
  const long long want = (M + 7) / 8;
  hipLaunchKernelGGL(k_linear_smallk, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(256), 0, st, X, ldx, K, W, bias, out, M);
  return rr_check(hipGetLastError());
}

// ContextualGating (atsp.py:108-121) around its scalar gate, forward recomputed and differentiated in one pass over the rows:
//   h = relu(hA | hB)  (the 256 pre-activations, bias included),  g = sigmoid(w2 . h + b2),  out = g node + (1 - g) dist
//   d gate = dout . (node - dist),  d pre = d gate g (1 - g),  dh = d pre w2 . 1(h > 0)  -> written over hA | hB
//   dnode (+)= g dout,  ddist = (1 - g) dout,  dw2 += sum_m d pre h,  db2 += sum_m d pre.
// A workgroup = 8 rows x 32 lanes (four features of each 128-vector per lane); the parameter gradients are folded over the
// workgroup's rows in LDS and added with one float atomic per feature and workgroup.
struct GateBwdIO {
  float *hA, *hB;
  const float *w2, *b2, *node, *dist, *dout;
  float *dnode, *ddist, *dw2, *db2;
  long long M;
  int acc_node;
  float* mix;          // optional: the forward value g node + (1 - g) dist (the VRPs' combine layer reads it: rcvrp.py:96-101)
};
__device__ __forceinline__ float gb_sum32(float v) {
  v = rr_sum16(v);
  float a, b;
  rr_pair16(v, a, b);
  return a + b;
}
__global__ __launch_bounds__(256) void k_gate_bwd(GateBwdIO io) {
  __shared__ float part[8][264];
  const int tid = threadIdx.x, l = tid & 31, slot = tid >> 5;
  const float4 wa = rr_ld4(io.w2 + 4 * l), wb = rr_ld4(io.w2 + RR_E + 4 * l);
  const float b2 = io.b2[0];
  float4 ga = make_float4(0.f, 0.f, 0.f, 0.f), gb = ga;
  float gb2 = 0.f;
  for (long long m = (long long)blockIdx.x * 8 + slot; m < io.M; m += (long long)gridDim.x * 8) {
    const size_t o = (size_t)m * RR_E + 4 * l;
    const float4 a = rr_ld4(io.hA + o), b = rr_ld4(io.hB + o), nd = rr_ld4(io.node + o), ds = rr_ld4(io.dist + o), dq_ = rr_ld4(io.dout + o);
    const float4 ra = make_float4(fmaxf(a.x, 0.f), fmaxf(a.y, 0.f), fmaxf(a.z, 0.f), fmaxf(a.w, 0.f));
    const float4 rb = make_float4(fmaxf(b.x, 0.f), fmaxf(b.y, 0.f), fmaxf(b.z, 0.f), fmaxf(b.w, 0.f));
    float s = wa.x * ra.x + wa.y * ra.y + wa.z * ra.z + wa.w * ra.w + wb.x * rb.x + wb.y * rb.y + wb.z * rb.z + wb.w * rb.w;
    float dg = dq_.x * (nd.x - ds.x) + dq_.y * (nd.y - ds.y) + dq_.z * (nd.z - ds.z) + dq_.w * (nd.w - ds.w);
    s = gb_sum32(s); dg = gb_sum32(dg);
    const float g = 1.0f / (1.0f + expf(-(s + b2)));
    const float dp = dg * g * (1.0f - g);
    rr_st4(io.hA + o, make_float4(a.x > 0.f ? dp * wa.x : 0.f, a.y > 0.f ? dp * wa.y : 0.f, a.z > 0.f ? dp * wa.z : 0.f, a.w > 0.f ? dp * wa.w : 0.f));
    rr_st4(io.hB + o, make_float4(b.x > 0.f ? dp * wb.x : 0.f, b.y > 0.f ? dp * wb.y : 0.f, b.z > 0.f ? dp * wb.z : 0.f, b.w > 0.f ? dp * wb.w : 0.f));
    float4 dn = make_float4(g * dq_.x, g * dq_.y, g * dq_.z, g * dq_.w);
    if (io.acc_node) { const float4 p = rr_ld4(io.dnode + o); dn.x += p.x; dn.y += p.y; dn.z += p.z; dn.w += p.w; }
    rr_st4(io.dnode + o, dn);
    const float h1 = 1.0f - g;
    rr_st4(io.ddist + o, make_float4(h1 * dq_.x, h1 * dq_.y, h1 * dq_.z, h1 * dq_.w));
    if (io.mix != nullptr) rr_st4(io.mix + o, make_float4(fmaf(g, nd.x, h1 * ds.x), fmaf(g, nd.y, h1 * ds.y), fmaf(g, nd.z, h1 * ds.z), fmaf(g, nd.w, h1 * ds.w)));
    ga.x = fmaf(dp, ra.x, ga.x); ga.y = fmaf(dp, ra.y, ga.y); ga.z = fmaf(dp, ra.z, ga.z); ga.w = fmaf(dp, ra.w, ga.w);
    gb.x = fmaf(dp, rb.x, gb.x); gb.y = fmaf(dp, rb.y, gb.y); gb.z = fmaf(dp, rb.z, gb.z); gb.w = fmaf(dp, rb.w, gb.w);
    gb2 += dp;
  }
  float* p = &part[slot][0];
  p[4 * l] = ga.x; p[4 * l + 1] = ga.y; p[4 * l + 2] = ga.z; p[4 * l + 3] = ga.w;
  p[RR_E + 4 * l] = gb.x; p[RR_E + 4 * l + 1] = gb.y; p[RR_E + 4 * l + 2] = gb.z; p[RR_E + 4 * l + 3] = gb.w;
  if (l == 0) p[256] = gb2;
  __syncthreads();
  for (int i = tid; i < 257; i += 256) {
    float tot = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) tot += part[s][i];
    atomicAdd(i < 256 ? io.dw2 + i : io.db2, tot);
  }
}
extern "C" int rr_gate_bwd(const GateBwdIO* io, hipStream_t st) {
  if (io == nullptr || io->hA == nullptr || io->hB == nullptr || io->w2 == nullptr || io->b2 == nullptr || io->node == nullptr ||
      io->dist == nullptr || io->dout == nullptr || io->dnode == nullptr || io->ddist == nullptr || io->dw2 == nullptr ||
      io->db2 == nullptr || io->M <= 0)
    return RR_EINVAL;
  const long long want = (io->M + 7) / 8;
  hipLaunchKernelGGL(k_gate_bwd, dim3((unsigned)(want < 1024 ? want : 1024)), dim3(256), 0, st, *io);
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ small dense products of the weight folds
// C[b] = op(A[b]) op(B[b]) for the handful of 128 x 128 (x 12 blocks) products of the host-side folds — project o multi_head_combine
// (attn_freenet.py:325, 435) and their chain rule in the training step — in float or double (the inference pack folds in float64).
// Plain 16 x 16 LDS tiling, one thread per output element: these are ~50 MFLOP per call; the point is that no BLAS library sits in the
// per-step repack / backward, not speed.
template <typename T>
__global__ __launch_bounds__(256) void k_small_gemm(const T* __restrict__ A, const T* __restrict__ B, T* __restrict__ C, int M, int N, int K,
                                                    int ta, int tb) {
  __shared__ T As[16][17], Bs[16][17];
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;
  const int m = blockIdx.y * 16 + ty, n = blockIdx.x * 16 + tx;
  const T* Ab = A + (size_t)blockIdx.z * M * K;
  const T* Bb = B + (size_t)blockIdx.z * K * N;
  T acc = (T)0;
  for (int k0 = 0; k0 < K; k0 += 16) {
    const int ka = k0 + tx, kb = k0 + ty;
    As[ty][tx] = (m < M && ka < K) ? (ta ? Ab[(size_t)ka * M + m] : Ab[(size_t)m * K + ka]) : (T)0;
    Bs[ty][tx] = (kb < K && n < N) ? (tb ? Bb[(size_t)n * K + kb] : Bb[(size_t)kb * N + n]) : (T)0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = fma(As[ty][k], Bs[k][tx], acc);
    __syncthreads();
  }
  if (m < M && n < N) C[(size_t)blockIdx.z * M * N + (size_t)m * N + n] = acc;
}
extern "C" int rr_small_gemm(const void* A, const void* B, void* C, int batch, int M, int N, int K, int transA, int transB, int f64,
                             hipStream_t st) {
  if (A == nullptr || B == nullptr || C == nullptr || batch <= 0 || M <= 0 || N <= 0 || K <= 0 || batch > 65535) return RR_EINVAL;
  const dim3 grid((N + 15) / 16, (M + 15) / 16, batch);
  if (f64) hipLaunchKernelGGL(k_small_gemm<double>, grid, dim3(256), 0, st, (const double*)A, (const double*)B, (double*)C, M, N, K, transA, transB);
  else hipLaunchKernelGGL(k_small_gemm<float>, grid, dim3(256), 0, st, (const float*)A, (const float*)B, (float*)C, M, N, K, transA, transB);
  return rr_check(hipGetLastError());
}
