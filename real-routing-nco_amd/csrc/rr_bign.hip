// Instances with more than 103 nodes (up to 1 024; rows of up to 208 keys in registers, longer ones streamed): the reference's generators go to 1000+ nodes
// (rrnco/envs/rcvrp/generator.py:22-37), while the register / LDS-resident kernels of rr_encoder.hip and rr_decode.hip hold one
// instance's [N, 128] activations, its N x N mixing weights and 7 key tiles per wave on chip (N <= 103).  Here the same
// operators run as row-parallel kernels over HBM / L2-resident tensors — every wave autonomous, no LDS images, key tiles as a
// template parameter — and the decode loop runs step by step (decoder.forward -> select -> env.step), the reference's own shape:
//   k_inorm_fwd      Normalization "instance" (attn_freenet.py:84, 104-105), optional residual input
//   k_nab_pwl_fwd    alpha * DistAngleFusion (:242-289) per edge from the folded piecewise-linear table
//   k_colsoftmax_exp exp(softmax_nodes(K)) and its product with V, transposed and zero padded ([feature][node]: A operands)
//   k_aft_mix_big    AFTFull mixing (:318-324): exp(softmax(bias)) of a 16-row tile in registers, num / den on the fp32 MFMA
//   k_dec_fwd_big    RRNetDecoder.forward (decoder.py:151-206, 281-323): context, masked 8-head attention, pointer MLP, logits
//   k_select_big     process_logits + greedy / sampling / evaluate (decoding.py:311-361) for rows of up to 256 keys
#include "rr_common.h"
#include <stdlib.h>

// ------------------------------------------------------------------------------------------------ instance norm forward
__global__ __launch_bounds__(256) void k_inorm_fwd(const float* __restrict__ x, const float* __restrict__ res,
                                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                                   float* __restrict__ out, int N) {
  __shared__ float red[2][256];
  const int b = blockIdx.x, tid = threadIdx.x, f = tid & 127, half = tid >> 7;
  const size_t base = (size_t)b * N * RR_E + f;
  float s = 0.f;
  for (int n = half; n < N; n += 2) s += x[base + (size_t)n * RR_E] + (res ? res[base + (size_t)n * RR_E] : 0.f);
  red[0][tid] = s;
  __syncthreads();
  const float inv_n = 1.0f / (float)N;
  const float mean = (red[0][f] + red[0][128 + f]) * inv_n;
  float q = 0.f;
  for (int n = half; n < N; n += 2) {
    const float d = x[base + (size_t)n * RR_E] + (res ? res[base + (size_t)n * RR_E] : 0.f) - mean;
    q = fmaf(d, d, q);
  }
  red[1][tid] = q;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((red[1][f] + red[1][128 + f]) * inv_n + 1e-5f);
  const float gm = gamma[f] * rstd, bt = beta[f];
  for (int n = half; n < N; n += 2) {
    const float v = x[base + (size_t)n * RR_E] + (res ? res[base + (size_t)n * RR_E] : 0.f);
    out[base + (size_t)n * RR_E] = fmaf(v - mean, gm, bt);
  }
}
extern "C" int rr_inorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* out, int Bp, int N,
                            hipStream_t st) {
  if (x == nullptr || gamma == nullptr || beta == nullptr || out == nullptr || Bp <= 0 || N < 1) return RR_EINVAL;
  hipLaunchKernelGGL(k_inorm_fwd, dim3(Bp), dim3(256), 0, st, x, res, gamma, beta, out, N);
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ batch norm forward, TRAINING mode
// Normalization("batch") with module.train() (attn_freenet.py:82-83, 102-103): nn.BatchNorm1d over the flattened M = Bp * N rows — batch
// mean and BIASED variance per feature normalise, the running statistics move by `momentum` (the variance UNBIASED, M / (M - 1)), eps 1e-5.
// Two launches: per-feature sums of v = x (+ res) and v^2 in float64 (atomics into ws[256], zeroed here), then the affine map.
// sum_out (optional) receives v (the hand-written backward wants ffn.norm1's input r + norm3(o)).
__global__ __launch_bounds__(256) void k_bn_stats(const float* __restrict__ x, const float* __restrict__ res, double* __restrict__ ws, long long M) {
  __shared__ double red[2][256];
  const int tid = threadIdx.x, f = tid & 127, half = tid >> 7;
  double s = 0.0, q = 0.0;
  for (long long m = (long long)blockIdx.x * 2 + half; m < M; m += (long long)gridDim.x * 2) {
    const float v = x[m * RR_E + f] + (res ? res[m * RR_E + f] : 0.f);
    s += (double)v; q += (double)v * (double)v;
  }
  red[0][tid] = s; red[1][tid] = q;
  __syncthreads();
  if (half == 0) { atomicAdd(ws + f, red[0][f] + red[0][128 + f]); atomicAdd(ws + 128 + f, red[1][f] + red[1][128 + f]); }
}
__global__ __launch_bounds__(256) void k_bn_apply(const float* __restrict__ x, const float* __restrict__ res, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, float* __restrict__ out, float* __restrict__ sum_out,
                                                  const double* __restrict__ ws, float* __restrict__ running_mean,
                                                  float* __restrict__ running_var, float momentum, long long M) {
  const int tid = threadIdx.x, f = tid & 127, half = tid >> 7;
  const double mean = ws[f] / (double)M;
  const double var = fmax(ws[128 + f] / (double)M - mean * mean, 0.0);
  const float mu = (float)mean, rstd = 1.0f / sqrtf((float)var + 1e-5f);
  const float gm = gamma[f] * rstd, bt = beta[f];
  for (long long m = (long long)blockIdx.x * 2 + half; m < M; m += (long long)gridDim.x * 2) {
    const float v = x[m * RR_E + f] + (res ? res[m * RR_E + f] : 0.f);
    if (sum_out) sum_out[m * RR_E + f] = v;
    out[m * RR_E + f] = fmaf(v - mu, gm, bt);
  }
  if (blockIdx.x == 0 && half == 0 && running_mean != nullptr && momentum > 0.f) {
    running_mean[f] = fmaf(momentum, mu - running_mean[f], running_mean[f]);
    const float unb = (float)(var * ((double)M / (double)(M > 1 ? M - 1 : 1)));
    running_var[f] = fmaf(momentum, unb - running_var[f], running_var[f]);
  }
}
extern "C" int rr_bnorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* out, float* sum_out,
                            double* ws, float* running_mean, float* running_var, float momentum, long long M, hipStream_t st) {
  if (x == nullptr || gamma == nullptr || beta == nullptr || out == nullptr || ws == nullptr || M <= 0) return RR_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return RR_EINVAL;
  if (hipMemsetAsync(ws, 0, 256 * sizeof(double), st) != hipSuccess) return RR_ELAUNCH;
  const long long want = (M + 1) / 2;
  const unsigned grid = (unsigned)(want < 2048 ? want : 2048);
  hipLaunchKernelGGL(k_bn_stats, dim3(grid), dim3(256), 0, st, x, res, ws, M);
  hipLaunchKernelGGL(k_bn_apply, dim3(grid), dim3(256), 0, st, x, res, gamma, beta, out, sum_out, ws, running_mean, running_var, momentum, M);
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ NAB forward per edge
#define BN_TAB (256 + 2 * 129 * 4 + 8)
__device__ __forceinline__ int bn_segment(const float* t, float x) {
  int m = 0;
#pragma unroll
  for (int s = 128; s >= 1; s >>= 1) {
    const int idx = m + s - 1;
    const float tv = t[idx < 128 ? idx : 127];
    m += (idx < 128 && tv <= x) ? s : 0;
  }
  return m;
}
// out[b][i][j] = alpha * NAB(d, theta[b][i][j]) with d = D[b][i][j] (row block) or D[b][j][i] (col block: D^T, attn_freenet.py:480-486)
__global__ __launch_bounds__(256) void k_nab_pwl_fwd(const float* __restrict__ pwl, const float* __restrict__ D,
                                                     const float* __restrict__ theta, float* __restrict__ out, int N, int transpose_d) {
  __shared__ __attribute__((aligned(16))) float tab[BN_TAB];
  for (int i = threadIdx.x; i < BN_TAB; i += 256) tab[i] = pwl[i];
  __syncthreads();
  const float bg = tab[256 + 1032], bo = tab[256 + 1033], alpha = tab[256 + 1034];
  const int b = blockIdx.y;
  const size_t base = (size_t)b * N * N;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < N * N; e += gridDim.x * 256) {
    const int i = e / N, j = e - i * N;
    const float x[2] = {fminf(D[base + (transpose_d ? j * N + i : e)], 3.0e38f), fminf(theta[base + e], 3.0e38f)};
    float fo[2], fg[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      const int m = bn_segment(tab + 128 * f, x[f]);
      float anchor = tab[128 * f + (m > 0 ? m - 1 : 0)];
      anchor = anchor < INFINITY ? anchor : 0.f;
      const float4 sg = rr_ld4(tab + 256 + 516 * f + 4 * m);
      const float dx = x[f] - anchor;
      fo[f] = fmaf(sg.x, dx, sg.y); fg[f] = fmaf(sg.z, dx, sg.w);
    }
    const float gt = rr_sigmoid(fg[0] + fg[1] + bg);
    out[base + e] = (gt * fo[0] + (1.0f - gt) * fo[1] + bo) * alpha;
  }
}
extern "C" int rr_nab_pwl_fwd(const float* pwl, const float* D, const float* theta, float* out, int Bp, int N, int transpose_d,
                              hipStream_t st) {
  if (pwl == nullptr || D == nullptr || theta == nullptr || out == nullptr || Bp <= 0 || N < 2) return RR_EINVAL;
  const int gx = min((N * N + 255) / 256, 64);
  hipLaunchKernelGGL(k_nab_pwl_fwd, dim3(gx, Bp), dim3(256), 0, st, pwl, D, theta, out, N, transpose_d);
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ exp(softmax_nodes(K)), times V
// ekT / kvT [Bp][128][NP] (NP = 16 * ceil(N / 16), zero beyond N): eK = exp(softmax over the NODE axis of K) (attn_freenet.py:319, 321)
__global__ __launch_bounds__(256) void k_colsoftmax_exp(const float* __restrict__ K, const float* __restrict__ V,
                                                        float* __restrict__ ekT, float* __restrict__ kvT, int N, int NP,
                                                        float* __restrict__ ek_rows) {
  __shared__ float red[2][256];
  const int b = blockIdx.x, tid = threadIdx.x, f = tid & 127, half = tid >> 7;
  const size_t base = (size_t)b * N * RR_E + f;
  float mx = -INFINITY;
  for (int n = half; n < N; n += 2) mx = fmaxf(mx, K[base + (size_t)n * RR_E]);
  red[0][tid] = mx;
  __syncthreads();
  mx = fmaxf(red[0][f], red[0][128 + f]);
  float s = 0.f;
  for (int n = half; n < N; n += 2) s += rr_exp(K[base + (size_t)n * RR_E] - mx);
  red[1][tid] = s;
  __syncthreads();
  const float inv = 1.0f / (red[1][f] + red[1][128 + f]);
  float* pe = ekT + ((size_t)b * RR_E + f) * NP;
  float* pk = kvT + ((size_t)b * RR_E + f) * NP;
  for (int n = half; n < NP; n += 2) {
    float e = 0.f, kv = 0.f;
    if (n < N) {
      e = rr_exp(rr_exp(K[base + (size_t)n * RR_E] - mx) * inv);
      kv = e * V[base + (size_t)n * RR_E];
      if (ek_rows) ek_rows[base + (size_t)n * RR_E] = e;     // [Bp][N][128]: what the block backward reads (EncSave.ek)
    }
    pe[n] = e; pk[n] = kv;
  }
}
extern "C" int rr_colsoftmax_exp(const float* K, const float* V, float* ekT, float* kvT, float* ek_rows, int Bp, int N, int NP, hipStream_t st) {
  if (K == nullptr || V == nullptr || ekT == nullptr || kvT == nullptr || Bp <= 0 || N < 1 || NP < N || (NP & 15)) return RR_EINVAL;
  hipLaunchKernelGGL(k_colsoftmax_exp, dim3(Bp), dim3(256), 0, st, K, V, ekT, kvT, N, NP, ek_rows);
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ AFT mixing
// y[i][f] = sigmoid(q[i][f]) * (sum_j ea[i][j] kv[j][f]) / (sum_j ea[i][j] ek[j][f]),  ea = exp(softmax_j(bias[i][:]))
// One wave = 16 rows i of one instance; ea as the B operand (k = j) in registers; A operands = rows of kvT / ekT from L2.
struct AftMixSave { float *num, *den, *eaT; };   // optional (all or none): [Bp][N][128] x 2 and exp(softmax(bias))^T [Bp][112][112] (zero-filled by the caller; N <= 112): EncSave's fields
template <int NTK>
__global__ __launch_bounds__(256, 2) void k_aft_mix_big(const float* __restrict__ bias, const float* __restrict__ q,
                                                        const float* __restrict__ ekT, const float* __restrict__ kvT,
                                                        float* __restrict__ y, int N, int NP, int tiles_per_inst, int ntask, AftMixSave sv) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int task = blockIdx.x * 4 + wave;
  if (task >= ntask) return;
  const int b = task / tiles_per_inst, tile = task - b * tiles_per_inst;
  const int node = 16 * tile + j;
  const bool nvalid = node < N;
  const int nc = nvalid ? node : N - 1;
  const float* brow = bias + ((size_t)b * N + nc) * N;
  f32x4 ea[NTK];
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jj = 16 * kt + 4 * g + r;
      const float v = jj < N ? brow[jj] : -INFINITY;
      ea[kt][r] = v; mx = fmaxf(mx, v);
    }
  mx = rr_max_g(mx);
  float sum = 0.f;
#pragma unroll
  for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) { ea[kt][r] = rr_exp(ea[kt][r] - mx); sum += ea[kt][r]; }
  sum = rr_sum_g(sum);
  const float is = 1.0f / sum;
#pragma unroll
  for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) ea[kt][r] = (16 * kt + 4 * g + r) < N ? rr_exp(ea[kt][r] * is) : 0.f;
  if (sv.eaT != nullptr && nvalid) {                         // [j][i]: lanes j = consecutive nodes i -> contiguous stores
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jj = 16 * kt + 4 * g + r;
        if (jj < N) sv.eaT[((size_t)b * 112 + jj) * 112 + node] = ea[kt][r];
      }
  }
  const float* pe = ekT + (size_t)b * RR_E * NP + 4 * g;
  const float* pk = kvT + (size_t)b * RR_E * NP + 4 * g;
  const size_t roff = ((size_t)b * N + nc) * RR_E + 4 * g;
#pragma unroll 1
  for (int t = 0; t < 8; ++t) {
    f32x4 dn = rr_zero4(), nm = rr_zero4();
    const size_t ao = (size_t)(16 * t + j) * NP;
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt) {
      if (16 * kt < NP) {                           // wave-uniform
        const float4 ae = rr_ld4(pe + ao + 16 * kt), az = rr_ld4(pk + ao + 16 * kt);
        dn = rr_mfma(ae.x, ea[kt][0], dn); nm = rr_mfma(az.x, ea[kt][0], nm);
        dn = rr_mfma(ae.y, ea[kt][1], dn); nm = rr_mfma(az.y, ea[kt][1], nm);
        dn = rr_mfma(ae.z, ea[kt][2], dn); nm = rr_mfma(az.z, ea[kt][2], nm);
        dn = rr_mfma(ae.w, ea[kt][3], dn); nm = rr_mfma(az.w, ea[kt][3], nm);
      }
    }
    const float4 qq = rr_ld4(q + roff + 16 * t);
    const float qv[4] = {qq.x, qq.y, qq.z, qq.w};
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = rr_sigmoid(qv[r]) * nm[r] / dn[r];
    if (nvalid) rr_st4(y + roff + 16 * t, make_float4(o[0], o[1], o[2], o[3]));
    if (sv.num != nullptr && nvalid) {
      rr_st4(sv.num + roff + 16 * t, make_float4(nm[0], nm[1], nm[2], nm[3]));
      rr_st4(sv.den + roff + 16 * t, make_float4(dn[0], dn[1], dn[2], dn[3]));
    }
  }
}
// The same mixing for any N (rows of more than 208 keys do not fit a lane's registers): three sweeps over the bias row — maximum,
// denominator, then per key tile the four weights of the lane rebuilt and the two products of all eight feature tiles accumulated.
// Every accumulator receives its key tiles in the same order as in k_aft_mix_big: bit-identical results where both apply.
__global__ __launch_bounds__(256, 2) void k_aft_mix_stream(const float* __restrict__ bias, const float* __restrict__ q,
                                                           const float* __restrict__ ekT, const float* __restrict__ kvT,
                                                           float* __restrict__ y, int N, int NP, int tiles_per_inst, int ntask) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int task = blockIdx.x * 4 + wave;
  if (task >= ntask) return;
  const int b = task / tiles_per_inst, tile = task - b * tiles_per_inst;
  const int node = 16 * tile + j;
  const bool nvalid = node < N;
  const int nc = nvalid ? node : N - 1;
  const float* brow = bias + ((size_t)b * N + nc) * N;
  const int nkt = NP >> 4;
  float mx = -INFINITY;
#pragma unroll 1
  for (int kt = 0; kt < nkt; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jj = 16 * kt + 4 * g + r;
      mx = fmaxf(mx, jj < N ? brow[jj] : -INFINITY);
    }
  mx = rr_max_g(mx);
  float sum = 0.f;
#pragma unroll 1
  for (int kt = 0; kt < nkt; ++kt)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jj = 16 * kt + 4 * g + r;
      sum += rr_exp((jj < N ? brow[jj] : -INFINITY) - mx);
    }
  sum = rr_sum_g(sum);
  const float is = 1.0f / sum;
  const float* pe = ekT + (size_t)b * RR_E * NP + 4 * g;
  const float* pk = kvT + (size_t)b * RR_E * NP + 4 * g;
  const size_t roff = ((size_t)b * N + nc) * RR_E + 4 * g;
  f32x4 dn[8], nm[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) { dn[t] = rr_zero4(); nm[t] = rr_zero4(); }
#pragma unroll 1
  for (int kt = 0; kt < nkt; ++kt) {
    float ea[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int jj = 16 * kt + 4 * g + r;
      ea[r] = jj < N ? rr_exp(rr_exp(brow[jj] - mx) * is) : 0.f;
    }
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const size_t ao = (size_t)(16 * t + j) * NP;
      const float4 ae = rr_ld4(pe + ao + 16 * kt), az = rr_ld4(pk + ao + 16 * kt);
      dn[t] = rr_mfma(ae.x, ea[0], dn[t]); nm[t] = rr_mfma(az.x, ea[0], nm[t]);
      dn[t] = rr_mfma(ae.y, ea[1], dn[t]); nm[t] = rr_mfma(az.y, ea[1], nm[t]);
      dn[t] = rr_mfma(ae.z, ea[2], dn[t]); nm[t] = rr_mfma(az.z, ea[2], nm[t]);
      dn[t] = rr_mfma(ae.w, ea[3], dn[t]); nm[t] = rr_mfma(az.w, ea[3], nm[t]);
    }
  }
#pragma unroll
  for (int t = 0; t < 8; ++t) {
    const float4 qq = rr_ld4(q + roff + 16 * t);
    const float qv[4] = {qq.x, qq.y, qq.z, qq.w};
    float o[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) o[r] = rr_sigmoid(qv[r]) * nm[t][r] / dn[t][r];
    if (nvalid) rr_st4(y + roff + 16 * t, make_float4(o[0], o[1], o[2], o[3]));
  }
}
#define RR_BIGN_MAX 1024     // nodes of the row-parallel path (the reference's generators go to 1 000)
static bool rr_bign_force_stream() {      // tests: the any-N kernels at every N (read per call: a test flips it inside one process)
  const char* e = getenv("RR_BIGN_STREAM");
  return e != nullptr && atoi(e) != 0;
}
extern "C" int rr_aft_mix_big(const float* bias, const float* q, const float* ekT, const float* kvT, float* y, float* num_out,
                              float* den_out, float* eaT_out, int Bp, int N, int NP, hipStream_t st) {
  if (bias == nullptr || q == nullptr || ekT == nullptr || kvT == nullptr || y == nullptr || Bp <= 0 || N < 2 || N > RR_BIGN_MAX || NP < N || (NP & 15))
    return RR_EINVAL;
  const bool saving = num_out != nullptr;
  if (saving != (den_out != nullptr) || saving != (eaT_out != nullptr) || (saving && N > 112)) return RR_EINVAL;
  const AftMixSave sv{num_out, den_out, eaT_out};
  const int tpi = (N + 15) / 16, ntask = Bp * tpi;
  const dim3 grid((ntask + 3) / 4);
  if ((N > 208 || rr_bign_force_stream()) && !saving) hipLaunchKernelGGL(k_aft_mix_stream, grid, dim3(256), 0, st, bias, q, ekT, kvT, y, N, NP, tpi, ntask);
  else if (N <= 128) hipLaunchKernelGGL((k_aft_mix_big<8>), grid, dim3(256), 0, st, bias, q, ekT, kvT, y, N, NP, tpi, ntask, sv);
  else hipLaunchKernelGGL((k_aft_mix_big<13>), grid, dim3(256), 0, st, bias, q, ekT, kvT, y, N, NP, tpi, ntask, sv);
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ decoder.forward
struct DecBigIO {
  const float *K, *Vt, *L;         // glimpse keys [Bp][N][128], values transposed [Bp][128][NP], logit keys [Bp][N][128]
  const float *ctxA, *ctxB;        // step-context tables [Bp][N][128] (ctxA NULL unless `first`)
  const float *D, *Dur;            // [Bp][N][N]
  const int64_t *cur, *first;      // [R] (first NULL for the VRPs)
  const float *scal, *wstate;      // [R][4] state scalars and [nscal][128] (NULL: none)
  const uint8_t* mask;             // [R][N]
  const float4 *w1, *w2;           // pointer MLP packs (packing.pack_a)
  const float *b1, *b2;
  float* logits;                   // [R][N]
  int Bp, N, NP, S, nscal;
  float alpha, beta;
};

template <int NTK>
__global__ __launch_bounds__(256, 1) void k_dec_fwd_big(DecBigIO io, int tiles_per_inst, int ntask) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int j = lane & 15, g = lane >> 4;
  const int task = blockIdx.x * 4 + wave;
  if (task >= ntask) return;
  const int N = io.N, NP = io.NP, Bp = io.Bp;
  const int b = task / tiles_per_inst, tile = task - b * tiles_per_inst;
  const int srow = 16 * tile + j;
  const bool rvalid = srow < io.S;
  const size_t r = (size_t)(rvalid ? srow : 0) * Bp + b;
  const int cur = min((int)io.cur[r], N - 1);
  const uint8_t* mrow = io.mask + r * N;
  // ---- step context (TSPContext / VRPContext / MTVRPContext as table gathers, as the rollout kernel does)
  f32x4 G[8];
  {
    const float* pc = io.ctxB + ((size_t)b * N + cur) * RR_E + 4 * g;
    const float* pa = io.first ? io.ctxA + ((size_t)b * N + min((int)io.first[r], N - 1)) * RR_E + 4 * g : nullptr;
    float sc[4] = {0.f, 0.f, 0.f, 0.f};
    if (io.nscal > 0) { const float4 s4 = rr_ld4(io.scal + r * 4); sc[0] = s4.x; sc[1] = s4.y; sc[2] = s4.z; sc[3] = s4.w; }
#pragma unroll
    for (int h = 0; h < 8; ++h) {
      float4 c = rr_ld4(pc + 16 * h);
      if (pa) { const float4 a = rr_ld4(pa + 16 * h); c.x += a.x; c.y += a.y; c.z += a.z; c.w += a.w; }
#pragma unroll
      for (int k = 0; k < 4; ++k)
        if (k < io.nscal) {
          const float4 wv = rr_ld4(io.wstate + k * RR_E + 16 * h + 4 * g);
          c.x = fmaf(wv.x, sc[k], c.x); c.y = fmaf(wv.y, sc[k], c.y); c.z = fmaf(wv.z, sc[k], c.z); c.w = fmaf(wv.w, sc[k], c.w);
        }
      G[h] = f32x4{c.x, c.y, c.z, c.w};
    }
  }
  // ---- masked multi-head attention (decoder.py:308-323): all key tiles of a head in registers; NTK == 0 (any N): two sweeps per head,
  // the scores rebuilt in the second one (same matrix instructions, same order of every sum: bit-identical where both forms apply)
  constexpr int NA = NTK > 0 ? NTK : 1;
  const int nkt = NP >> 4;
  f32x4 mneg[NA];
  auto mask4 = [&](int kt) {
    f32x4 m;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int key = 16 * kt + 4 * g + q;
      m[q] = (key < N && mrow[key]) ? 0.f : -INFINITY;
    }
    return m;
  };
  if constexpr (NTK > 0) {
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt) mneg[kt] = mask4(kt);
  }
  const float* Kb = io.K + (size_t)b * N * RR_E;
  const float* Vb = io.Vt + (size_t)b * RR_E * NP;
#pragma unroll 1
  for (int h = 0; h < 8; ++h) {
    const float g0 = G[0][0] * 0.25f, g1 = G[0][1] * 0.25f, g2 = G[0][2] * 0.25f, g3 = G[0][3] * 0.25f;
    if constexpr (NTK == 0) {
      auto score = [&](int kt) {
        int key = 16 * kt + j; key = key < N ? key : N - 1;
        const float4 kf = rr_ld4(Kb + (size_t)key * RR_E + 16 * h + 4 * g);
        f32x4 a = rr_mfma(kf.x, g0, mask4(kt));
        a = rr_mfma(kf.y, g1, a); a = rr_mfma(kf.z, g2, a); a = rr_mfma(kf.w, g3, a);
        return a;
      };
      float mx = -INFINITY;
#pragma unroll 1
      for (int kt = 0; kt < nkt; ++kt) {
        const f32x4 a = score(kt);
#pragma unroll
        for (int q = 0; q < 4; ++q) mx = fmaxf(mx, a[q]);
      }
      mx = rr_max_g(mx);
      if (mx == -INFINITY) mx = 0.f;
      float sum = 0.f;
      f32x4 o0 = rr_zero4(), o1 = rr_zero4();
#pragma unroll 1
      for (int kt = 0; kt < nkt; ++kt) {
        f32x4 a = score(kt);
#pragma unroll
        for (int q = 0; q < 4; ++q) { a[q] = rr_exp(a[q] - mx); sum += a[q]; }
        const float4 vf = rr_ld4(Vb + (size_t)(16 * h + j) * NP + 16 * kt + 4 * g);
        o0 = rr_mfma(vf.x, a[0], o0); o1 = rr_mfma(vf.y, a[1], o1);
        o0 = rr_mfma(vf.z, a[2], o0); o1 = rr_mfma(vf.w, a[3], o1);
      }
      sum = rr_sum_g(sum);
      const float inv = sum > 0.f ? 1.0f / sum : 0.f;
      f32x4 gsel;
#pragma unroll
      for (int q = 0; q < 4; ++q) gsel[q] = fmaf(o0[q] + o1[q], inv, G[0][q]);
#pragma unroll
      for (int q = 0; q < 7; ++q) G[q] = G[q + 1];
      G[7] = gsel;
      continue;
    }
    f32x4 sc[NA];
    float mx = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt) {
      int key = 16 * kt + j; key = key < N ? key : N - 1;
      const float4 kf = rr_ld4(Kb + (size_t)key * RR_E + 16 * h + 4 * g);
      f32x4 a = rr_mfma(kf.x, g0, mneg[kt]);
      a = rr_mfma(kf.y, g1, a); a = rr_mfma(kf.z, g2, a); a = rr_mfma(kf.w, g3, a);
      sc[kt] = a;
#pragma unroll
      for (int q = 0; q < 4; ++q) mx = fmaxf(mx, a[q]);
    }
    mx = rr_max_g(mx);
    if (mx == -INFINITY) mx = 0.f;
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt)
#pragma unroll
      for (int q = 0; q < 4; ++q) { const float e = rr_exp(sc[kt][q] - mx); sc[kt][q] = e; sum += e; }
    sum = rr_sum_g(sum);
    const float inv = sum > 0.f ? 1.0f / sum : 0.f;
    f32x4 o0 = rr_zero4(), o1 = rr_zero4();
#pragma unroll
    for (int kt = 0; kt < NTK; ++kt) {
      if (16 * kt < NP) {
        const float4 vf = rr_ld4(Vb + (size_t)(16 * h + j) * NP + 16 * kt + 4 * g);
        o0 = rr_mfma(vf.x, sc[kt][0], o0); o1 = rr_mfma(vf.y, sc[kt][1], o1);
        o0 = rr_mfma(vf.z, sc[kt][2], o0); o1 = rr_mfma(vf.w, sc[kt][3], o1);
      }
    }
    f32x4 gsel;
#pragma unroll
    for (int q = 0; q < 4; ++q) gsel[q] = fmaf(o0[q] + o1[q], inv, G[0][q]);          // glimpse = heads + query (decoder.py:294)
#pragma unroll
    for (int q = 0; q < 7; ++q) G[q] = G[q + 1];
    G[7] = gsel;
  }
  // ---- pointer MLP with residual (decoder.py:296)
  f32x4 F[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    const float4 b2 = rr_ld4(io.b2 + 16 * u + 4 * g);
    F[u] = f32x4{G[u][0] + b2.x, G[u][1] + b2.y, G[u][2] + b2.z, G[u][3] + b2.w};
  }
  {
    const __amdgpu_buffer_rsrc_t rW1 = rr_make_buf(io.w1, RR_FF * RR_E * 4), rW2 = rr_make_buf(io.w2, RR_FF * RR_E * 4);
    const unsigned lane16 = (unsigned)lane * 16u;
    float4 a1[8], a2[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) a1[kk] = rr_bld4(rW1, lane16, kk * 1024u);
    float4 bb = rr_ld4(io.b1 + 4 * g);
#pragma unroll 1
    for (int tt = 0; tt < RR_FF / 16; ++tt) {
#pragma unroll
      for (int u = 0; u < 8; ++u) a2[u] = rr_bld4(rW2, lane16, (unsigned)(u * 32 + tt) * 1024u);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 he = {bb.x, bb.y, bb.z, bb.w}, ho = rr_zero4();
#pragma unroll
      for (int kk = 0; kk < 8; kk += 2) {
        he = rr_mfma(a1[kk].x, G[kk][0], he); ho = rr_mfma(a1[kk + 1].x, G[kk + 1][0], ho);
        he = rr_mfma(a1[kk].y, G[kk][1], he); ho = rr_mfma(a1[kk + 1].y, G[kk + 1][1], ho);
        he = rr_mfma(a1[kk].z, G[kk][2], he); ho = rr_mfma(a1[kk + 1].z, G[kk + 1][2], ho);
        he = rr_mfma(a1[kk].w, G[kk][3], he); ho = rr_mfma(a1[kk + 1].w, G[kk + 1][3], ho);
      }
      __builtin_amdgcn_sched_barrier(0);
      {
        const int tn = (tt + 1 < RR_FF / 16) ? tt + 1 : 0;
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) a1[kk] = rr_bld4(rW1, lane16, (unsigned)(tn * 8 + kk) * 1024u);
        bb = rr_ld4(io.b1 + tn * 16 + 4 * g);
      }
      __builtin_amdgcn_sched_barrier(0);
      f32x4 hh;
#pragma unroll
      for (int q = 0; q < 4; ++q) hh[q] = fmaxf(he[q] + ho[q], 0.f);
#pragma unroll
      for (int u = 0; u < 8; ++u) F[u] = rr_mfma(a2[u].x, hh[0], F[u]);
#pragma unroll
      for (int u = 0; u < 8; ++u) F[u] = rr_mfma(a2[u].y, hh[1], F[u]);
#pragma unroll
      for (int u = 0; u < 8; ++u) F[u] = rr_mfma(a2[u].z, hh[2], F[u]);
#pragma unroll
      for (int u = 0; u < 8; ++u) F[u] = rr_mfma(a2[u].w, hh[3], F[u]);
    }
  }
  // ---- logits = L F / sqrt(E), inductive bias, log(exp + 1e-6) (decoder.py:186-198, 300-302); tanh / mask happen in the selection
  const float* Lb = io.L + (size_t)b * N * RR_E;
  const float inv_sqe = 1.0f / sqrtf((float)RR_E);
  const float* Drow = io.D + ((size_t)b * N + cur) * N;
  const float* Trow = io.Dur ? io.Dur + ((size_t)b * N + cur) * N : nullptr;
  float* lout = io.logits + r * N;
#pragma unroll 1
  for (int kt = 0; kt < (NTK > 0 ? NTK : nkt); ++kt) {
    if (16 * kt >= N) break;                         // wave-uniform
    int key = 16 * kt + j; key = key < N ? key : N - 1;
    const float* lp = Lb + (size_t)key * RR_E + 4 * g;
    f32x4 c0 = rr_zero4(), c1 = rr_zero4();
#pragma unroll
    for (int kk = 0; kk < 8; kk += 2) {
      const float4 fa = rr_ld4(lp + 16 * kk), fb = rr_ld4(lp + 16 * (kk + 1));
      c0 = rr_mfma(fa.x, F[kk][0], c0); c1 = rr_mfma(fb.x, F[kk + 1][0], c1);
      c0 = rr_mfma(fa.y, F[kk][1], c0); c1 = rr_mfma(fb.y, F[kk + 1][1], c1);
      c0 = rr_mfma(fa.z, F[kk][2], c0); c1 = rr_mfma(fb.z, F[kk + 1][2], c1);
      c0 = rr_mfma(fa.w, F[kk][3], c0); c1 = rr_mfma(fb.w, F[kk + 1][3], c1);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int kq = 16 * kt + 4 * g + q;
      if (kq < N && rvalid) {
        const float bias = io.alpha * Drow[kq] + (Trow ? io.beta * Trow[kq] : 0.f);
        lout[kq] = rr_log(rr_exp((c0[q] + c1[q]) * inv_sqe - bias) + 1e-6f);
      }
    }
  }
}
extern "C" int rr_dec_fwd_big(const DecBigIO* io, hipStream_t st) {
  if (io == nullptr || io->K == nullptr || io->Vt == nullptr || io->L == nullptr || io->ctxB == nullptr || io->D == nullptr ||
      io->cur == nullptr || io->mask == nullptr || io->w1 == nullptr || io->w2 == nullptr || io->b1 == nullptr || io->b2 == nullptr ||
      io->logits == nullptr)
    return RR_EINVAL;
  if ((io->first != nullptr) != (io->ctxA != nullptr)) return RR_EINVAL;
  if (io->nscal < 0 || io->nscal > 4 || (io->nscal > 0 && (io->scal == nullptr || io->wstate == nullptr))) return RR_EINVAL;
  if (io->Bp <= 0 || io->N < 2 || io->N > RR_BIGN_MAX || io->S < 1 || io->NP < io->N || (io->NP & 15)) return RR_EINVAL;
  const int tpi = (io->S + 15) / 16, ntask = io->Bp * tpi;
  const dim3 grid((ntask + 3) / 4);
  if (io->N > 208 || rr_bign_force_stream()) hipLaunchKernelGGL((k_dec_fwd_big<0>), grid, dim3(256), 0, st, *io, tpi, ntask);
  else if (io->N <= 128) hipLaunchKernelGGL((k_dec_fwd_big<8>), grid, dim3(256), 0, st, *io, tpi, ntask);
  else hipLaunchKernelGGL((k_dec_fwd_big<13>), grid, dim3(256), 0, st, *io, tpi, ntask);
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ selection, N <= 64 Q
// One wave per row, keys lane + 64 q (Q = 4: rows of up to 256 keys, 16: up to 1 024).  mode 0 greedy (first index on ties), 1 sampling
// (inverse CDF, keyed uniform: rr_common.h), 2 evaluate.
template <int Q>
__global__ __launch_bounds__(256) void k_select_big(const float* __restrict__ logits, const uint8_t* __restrict__ mask,
                                                    const int64_t* __restrict__ action_in, int64_t* __restrict__ action_out,
                                                    float* __restrict__ logp_out, float* __restrict__ logp_all, int R, int N,
                                                    float tanh_clip, float temperature, int mode, uint64_t seed, uint32_t step,
                                                    int top_k, float top_p) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float inv_temp = 1.0f / temperature;
  float x[Q];
  float m = -INFINITY;
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int jx = lane + 64 * q;
    float v = -INFINITY;
    if (jx < N) {
      v = logits[(size_t)r * N + jx];
      if (tanh_clip > 0.f) v = rr_tanh(v) * tanh_clip;
      if (mask != nullptr && mask[(size_t)r * N + jx] == 0) v = -INFINITY;
      v *= inv_temp;
    }
    x[q] = v; m = fmaxf(m, v);
  }
  m = rr_wave_max(m);
  // process_logits' filters (decoding.py:37-63, 352-358), top-k first; the row spans the wave (the same rules as rr_select / the fused
  // rollout's rr_filter_*: the k-th largest value with multiplicity; the ascending cumulative sum by bisection over the floats' ordered
  // bit patterns, ties in key order, the last one stays)
  if (top_k > 0 && top_k < N) {
    float bound = INFINITY, thr = -INFINITY;
    int covered = 0;
    for (int it = 0; it < top_k; ++it) {
      float cur = -INFINITY;
#pragma unroll
      for (int q = 0; q < Q; ++q) cur = fmaxf(cur, x[q] < bound ? x[q] : -INFINITY);
      cur = rr_wave_max(cur);
#pragma unroll
      for (int q = 0; q < Q; ++q) covered += __popcll(__ballot(lane + 64 * q < N && x[q] == cur));
      thr = cur; bound = cur;
      if (covered >= top_k || cur == -INFINITY) break;
    }
#pragma unroll
    for (int q = 0; q < Q; ++q) x[q] = x[q] < thr ? -INFINITY : x[q];
  }
  if (top_p > 0.f && top_p < 1.f) {
    const float mm = m == -INFINITY ? 0.f : m;
    float pe[Q], z = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) { pe[q] = (lane + 64 * q < N) ? rr_exp(x[q] - mm) : 0.f; z += pe[q]; }
    z = rr_wave_sum(z);
    const float lim = (1.0f - top_p) * z;
    auto okey = [](float v) { const uint32_t b = __float_as_uint(v); return b ^ ((uint32_t)((int32_t)b >> 31) | 0x80000000u); };
    auto okey_inv = [](uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k ^ 0x80000000u) : ~k); };
    uint32_t lo = 0x007fffffu, hi = okey(mm);
    float tlo = 0.f;
    while (hi - lo > 1u) {                                 // (wave-uniform: one row per wave)
      const uint32_t mid = lo + ((hi - lo) >> 1);
      const float fm = okey_inv(mid);
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < Q; ++q) t += x[q] <= fm ? pe[q] : 0.f;
      t = rr_wave_sum(t);
      if (t <= lim) { lo = mid; tlo = t; } else hi = mid;
    }
    const float flo = okey_inv(lo), fhi = okey_inv(hi);
    const float pstar = rr_exp(fhi - mm);
    int size = 0, before = 0;
    unsigned long long tb[Q];
#pragma unroll
    for (int q = 0; q < Q; ++q) { tb[q] = __ballot(lane + 64 * q < N && x[q] == fhi); size += __popcll(tb[q]); }
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      const bool tied = (tb[q] >> lane) & 1ull;
      const int rank = before + __popcll(tb[q] & ((2ull << lane) - 1ull));
      const bool kill = tied && rank < size && fmaf((float)rank, pstar, tlo) <= lim;
      x[q] = (x[q] <= flo || kill) ? -INFINITY : x[q];
      before += __popcll(tb[q]);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < Q; ++q) s += (lane + 64 * q < N) ? rr_exp(x[q] - m) : 0.f;
  const float lse = rr_log(rr_wave_sum(s));
  const int want = mode == 2 ? (int)action_in[r] : -1;
  float bv = -INFINITY, blp = 0.f;
  int bi = 0x7fffffff;
  // sampling: inverse CDF over the keys in ascending order (rr_common.h): a key's value is its index where it is eligible — it has mass
  // and its exclusive prefix is <= target — so the maximum below is the last eligible key
  float en[Q], cpre[Q], target = 0.f;
#pragma unroll
  for (int q = 0; q < Q; ++q) { en[q] = 0.f; cpre[q] = 0.f; }
  if (mode == 1) {
    float base = 0.f;
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      en[q] = (lane + 64 * q < N) ? rr_exp(x[q] - m) : 0.f;
      float tq;
      cpre[q] = base + rr_wave_excl_scan(en[q], tq);
      base += tq;
    }
    target = rr_cdf_target(rr_uniform(seed, (uint32_t)r, step, RR_CDF_SLOT), base);
  }
#pragma unroll
  for (int q = 0; q < Q; ++q) {
    const int jx = lane + 64 * q;
    const float lp = x[q] - m - lse;
    if (logp_all != nullptr && jx < N) logp_all[(size_t)r * N + jx] = lp;
    float sv;
    if (mode == 2) sv = jx == want ? 1.f : -INFINITY;
    else if (mode == 1) sv = (en[q] > 0.f && cpre[q] <= target) ? (float)jx : -INFINITY;
    else sv = jx < N ? lp : -INFINITY;
    const bool better = sv > bv || (bi == 0x7fffffff && jx < N && mode == 0);
    bv = better ? sv : bv; bi = better ? jx : bi; blp = better ? lp : blp;
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const float ov = __shfl_xor(bv, o), ol = __shfl_xor(blp, o);
    const int oi = __shfl_xor(bi, o);
    const bool take = ov > bv || (ov == bv && oi < bi);
    bv = take ? ov : bv; bi = take ? oi : bi; blp = take ? ol : blp;
  }
  if (lane == 0) { action_out[r] = bi; logp_out[r] = blp; }
}
extern "C" int rr_select_big(const float* logits, const uint8_t* mask, const int64_t* action_in, int64_t* action_out,
                             float* logp_out, float* logp_all, int R, int N, float tanh_clip, float temperature, int mode,
                             uint64_t seed, uint32_t step, int top_k, float top_p, hipStream_t st) {
  if (R <= 0 || N <= 0 || N > RR_BIGN_MAX || temperature <= 0.f || mode < 0 || mode > 2 || top_k < 0 || top_p < 0.f || top_p > 1.f) return RR_EINVAL;
  if (logits == nullptr || action_out == nullptr || logp_out == nullptr || (mode == 2 && action_in == nullptr)) return RR_EINVAL;
  if (N <= 256 && !rr_bign_force_stream())
    hipLaunchKernelGGL(k_select_big<4>, dim3((R + 3) / 4), dim3(256), 0, st, logits, mask, action_in, action_out, logp_out, logp_all, R, N,
                       tanh_clip, temperature, mode, seed, step, top_k, top_p);
  else
    hipLaunchKernelGGL(k_select_big<16>, dim3((R + 3) / 4), dim3(256), 0, st, logits, mask, action_in, action_out, logp_out, logp_all, R, N,
                       tanh_clip, temperature, mode, seed, step, top_k, top_p);
  return rr_check(hipGetLastError());
}
