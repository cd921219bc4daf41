// MatNet baseline encoder for gfx950: MatNetLayer = mixed-score cross attention of the row block on the
// column embeddings and of the column block on the row embeddings, each followed by TransformerFFN
// (rrnco/baselines/MatNet/encoder.py:14-92 MixedScoresSDPA, :95-145 MatNetCrossMHA / MatNetMHA, :148-172 MatNetLayer;
// TransformerFFN: in-tree copy rrnco/models/nn/attn_freenet.py:330-357).  configs/experiment/matnet.yaml: E = 256, 16 heads
// (head dim 16), 5 layers, FF 512 — twice the width of RRNet, so the register-resident block of rr_enc_w.inc does not apply;
// the layer is a pipeline of three kernels over a caller-provided workspace in HBM (sized for 288 GB: 7.5 KB per node):
//   k_mn_lin   Y = X W^T (+bias, +ReLU | +residual, InstanceNorm): one workgroup = one instance side, the instance's
//              [N, 256] input chunk staged in LDS once, each wave owns whole output features (so the per-feature norm
//              statistics over nodes are wave-local), fp32 MFMA in the transposed-tile convention of rr_common.h
//   k_mn_attn  per (instance, side): wave = head; S^T = K_h Q_h^T on MFMA, the 2 -> 16 -> 1 score mixer with the distance
//              entry on the VALU (this is the bound: 64 flop per (query, key, head)), softmax over keys across the four
//              lane groups, O^T = V_h^T P^T on MFMA with P fed straight from the C layout
// Launch order per layer (both sides per launch): Q, KV, attention, out_proj + norm1, W1 + ReLU, W2 + norm2.
#include "rr_common.h"

#define MN_LDX 260            // LDS row stride (floats) of the staged [N][256] input chunk: 4-bank shift per node
#define MN_KC 256             // K chunk staged at once
#define MN_THREADS 512
#define MN_WAVES 8

struct MatNetSideW {
  const float4 *wq, *wkv, *wo, *w1, *w2;       // packed A operands (rrnco_amd/packing.pack_a): [Nout/16][K/16][64]
  const float *b1, *b2;                        // FeedForward biases [FF], [E]
  const float *n1g, *n1b, *n2g, *n2b;          // TransformerFFN norm1 / norm2 affine [E]
  const float* mix;                            // [heads][68]: W1 score row x 1/sqrt(dk) [16] | W1 distance row [16] | b1 [16] | W2 [16] | b2, 0, 0, 0
};

struct MnLinArgs {
  const float4* wp[2];        // packed weight per side
  const float* bias[2];       // [Nout] or nullptr
  const float* x[2];          // input [Bp][N][K] per side
  float* y[2];                // output [Bp][N][Nout] per side
  const float* resid[2];      // EPI 2: residual input [Bp][N][Nout]
  const float* gamma[2];      // EPI 2
  const float* beta[2];
  int K, Nout;
};

// EPI 0: store (+bias); 1: +bias, ReLU; 2: InstanceNorm(resid + acc + bias) over the instance's nodes
template <int NT, int TPW, int EPI>
__global__ __launch_bounds__(MN_THREADS) void k_mn_lin(MnLinArgs a, int N) {
  extern __shared__ __attribute__((aligned(16))) float mn_x[];
  const int b = blockIdx.x, side = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int K = a.K, Nout = a.Nout, KK = K >> 4;
  const int tile0 = blockIdx.y * (MN_WAVES * TPW) + wave * TPW;        // this wave's first output feature tile
  const float* X = a.x[side] + (size_t)b * N * K;
  f32x4 acc[TPW][NT];
#pragma unroll
  for (int t = 0; t < TPW; ++t)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = rr_zero4();
  for (int kc = 0; kc < K; kc += MN_KC) {
    __syncthreads();
    for (int i = tid; i < N * (MN_KC / 4); i += MN_THREADS) {
      const int node = i / (MN_KC / 4), c4 = i - node * (MN_KC / 4);
      rr_st4(mn_x + node * MN_LDX + 4 * c4, rr_ld4(X + (size_t)node * K + kc + 4 * c4));
    }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < TPW; ++t)
      rr_gemm_wx<NT>(acc[t], a.wp[side] + (size_t)(tile0 + t) * KK * 64, kc >> 4, MN_KC >> 4, mn_x, MN_LDX, 0, N, lane);
  }
  const int j = lane & 15, g = lane >> 4;
#pragma unroll
  for (int t = 0; t < TPW; ++t) {
    const int fbase = (tile0 + t) * 16;
    if (fbase >= Nout) continue;
    if (a.bias[side] != nullptr) rr_add_bias<NT>(acc[t], a.bias[side], fbase, lane);
    if (EPI == 1) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t][nt][r] = fmaxf(acc[t][nt][r], 0.f);
    }
    if (EPI == 2) {
      const float* R = a.resid[side] + (size_t)b * N * Nout;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        int node = nt * 16 + j; node = node < N ? node : N - 1;
        const float4 v = rr_ld4(R + (size_t)node * Nout + fbase + 4 * g);
        acc[t][nt][0] += v.x; acc[t][nt][1] += v.y; acc[t][nt][2] += v.z; acc[t][nt][3] += v.w;
      }
      rr_instnorm_tiles<NT>(acc[t], a.gamma[side], a.beta[side], fbase, N, lane);
    }
    rr_store_tiles<NT>(acc[t], a.y[side] + (size_t)b * N * Nout, Nout, fbase, N, lane);
  }
}

struct MnAttnArgs {
  const float* q[2];          // [Bp][N][E]
  const float* kv[2];         // [Bp][N][2E]: K | V
  float* o[2];                // [Bp][N][E]
  const float* mix[2];        // [heads][68]
  const float* D;             // [Bp][N][N]; side 1 reads it transposed
  int E, heads;
};

typedef float mn_f32x2 __attribute__((ext_vector_type(2)));

#define MN_ATTN_WAVES 4       // one wave per SIMD: the head's K / V^T fragments, the score tiles and the mixer need ~300 VGPRs
template <int NT>
__global__ __launch_bounds__(64 * MN_ATTN_WAVES) void k_mn_attn(MnAttnArgs a, int N) {
  const int b = blockIdx.x, side = blockIdx.z;
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int E = a.E;
  const float* Q = a.q[side] + (size_t)b * N * E;
  const float* KV = a.kv[side] + (size_t)b * N * 2 * E;
  float* O = a.o[side] + (size_t)b * N * E;
  // the instance's distance matrix, as this side reads it (side 1: transposed), staged once: every head re-reads all of it
  extern __shared__ __attribute__((aligned(16))) float mn_d[];
  {
    const float* Dg = a.D + (size_t)b * N * N;
    for (int i = threadIdx.x; i < N * N; i += 64 * MN_ATTN_WAVES) {
      const int q = i / N, k = i - q * N;
      mn_d[i] = side == 0 ? Dg[i] : Dg[(size_t)k * N + q];
    }
    __syncthreads();
  }
  for (int h = wave + blockIdx.y * MN_ATTN_WAVES; h < a.heads; h += MN_ATTN_WAVES * gridDim.y) {
    const float* mx = a.mix[side] + h * 68;
    // K_h as A operand of S^T (keys x dk), V_h^T as A operand of O^T (dk x keys, permuted k = 16 kt + 4 g + m)
    float4 kf[NT];
    float vf[NT][4];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      int key = kt * 16 + j; key = key < N ? key : N - 1;
      kf[kt] = rr_ld4(KV + (size_t)key * 2 * E + 16 * h + 4 * g);
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        int kv = kt * 16 + 4 * g + m; kv = kv < N ? kv : N - 1;
        vf[kt][m] = KV[(size_t)kv * 2 * E + E + 16 * h + j];
      }
    }
#pragma unroll 1
    for (int qt = 0; qt < NT; ++qt) {
      int qn = qt * 16 + j;
      const bool qok = qn < N;
      qn = qok ? qn : N - 1;
      const float4 qf = rr_ld4(Q + (size_t)qn * E + 16 * h + 4 * g);
      f32x4 s[NT];
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        s[kt] = rr_zero4();
        s[kt] = rr_mfma(kf[kt].x, qf.x, s[kt]); s[kt] = rr_mfma(kf[kt].y, qf.y, s[kt]);
        s[kt] = rr_mfma(kf[kt].z, qf.z, s[kt]); s[kt] = rr_mfma(kf[kt].w, qf.w, s[kt]);
      }
      // score mixer (encoder.py:54-75): lane holds keys 16 kt + 4 g + r of query qn; two keys per packed instruction
      float mxv = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        float d[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          int key = kt * 16 + 4 * g + r; key = key < N ? key : N - 1;
          d[r] = mn_d[qn * N + key];
        }
#pragma unroll
        for (int rp = 0; rp < 4; rp += 2) {
          const mn_f32x2 sv = {s[kt][rp], s[kt][rp + 1]}, dv = {d[rp], d[rp + 1]};
          mn_f32x2 out = {mx[64], mx[64]};
#pragma unroll
          for (int u = 0; u < 16; ++u) {
            mn_f32x2 hid = sv * mx[u] + (dv * mx[16 + u] + mx[32 + u]);
            hid.x = fmaxf(hid.x, 0.f); hid.y = fmaxf(hid.y, 0.f);
            out = hid * mx[48 + u] + out;
          }
          s[kt][rp] = out.x; s[kt][rp + 1] = out.y;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (kt * 16 + 4 * g + r >= N) s[kt][r] = -INFINITY;
          mxv = fmaxf(mxv, s[kt][r]);
        }
      }
      mxv = rr_max_g(mxv);
      float sum = 0.f;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) { s[kt][r] = rr_exp(s[kt][r] - mxv); sum += s[kt][r]; }
      sum = rr_sum_g(sum);
      const float inv = 1.0f / sum;
      // O^T (dk x queries) = V_h^T P^T: the C layout of S^T is the permuted-k B operand
      f32x4 o = rr_zero4();
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        o = rr_mfma(vf[kt][0], s[kt][0], o); o = rr_mfma(vf[kt][1], s[kt][1], o);
        o = rr_mfma(vf[kt][2], s[kt][2], o); o = rr_mfma(vf[kt][3], s[kt][3], o);
      }
      if (qok) rr_st4(O + (size_t)qn * E + 16 * h + 4 * g, make_float4(o[0] * inv, o[1] * inv, o[2] * inv, o[3] * inv));
    }
  }
}

// ---- init embeddings (env_embeddings/atsp.py:21-34, rcvrp.py:37-81 with use_coords=False), folded on the host:
//   row[n] = rowv[kind] + rowv[2] * demand      col[n] = slot_t[rand_idx[n]] (= W_col[:, slot], or e_slot for ATSP) + colv[kind] + colv[2] * demand
// kind 0 depot, 1 customer; ATSP: every vector null (row = 0, col = one-hot).
__global__ __launch_bounds__(256) void k_mn_init(const int64_t* __restrict__ rand_idx, const float* __restrict__ demand,
                                                 const float* __restrict__ rowv, const float* __restrict__ colv,
                                                 const float* __restrict__ slot_t, float* __restrict__ row,
                                                 float* __restrict__ col, long total, int N, int E) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int f = (int)(i % E);
  const long bn = i / E;
  const int n = (int)(bn % N);
  const int slot = (int)rand_idx[bn];
  float r = 0.f, c = 0.f;
  if (rowv != nullptr) {
    const float dem = n == 0 ? 0.f : demand[(bn / N) * (N - 1) + n - 1];
    const int kind = n == 0 ? 0 : 1;
    r = rowv[kind * E + f] + (n == 0 ? 0.f : rowv[2 * E + f] * dem);
    c = colv[kind * E + f] + (n == 0 ? 0.f : colv[2 * E + f] * dem) + slot_t[(size_t)slot * E + f];
  } else {
    c = slot == f ? 1.0f : 0.f;
  }
  row[i] = r; col[i] = c;
}

extern "C" int rr_matnet_init(const int64_t* rand_idx, const float* demand, const float* rowv, const float* colv,
                              const float* slot_t, float* row, float* col, int Bp, int N, int E, hipStream_t st) {
  if (Bp <= 0 || N <= 0 || E <= 0 || rand_idx == nullptr) return RR_EINVAL;
  if ((rowv == nullptr) != (colv == nullptr) || (rowv != nullptr && (demand == nullptr || slot_t == nullptr))) return RR_EINVAL;
  const long total = (long)Bp * N * E;
  hipLaunchKernelGGL(k_mn_init, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, rand_idx, demand, rowv, colv, slot_t, row, col,
                     total, N, E);
  return rr_check(hipGetLastError());
}

extern "C" size_t rr_matnet_workspace_bytes(int Bp, int N, int E, int ff) {
  // per side: Q [E] + KV [2E] + O [E] + X1 [E] + H [ff] floats per node
  return (size_t)2 * Bp * N * (5 * (size_t)E + ff) * sizeof(float);
}

template <int NT>
static int mn_layer(const MatNetSideW* ws, const float* row_in, const float* col_in, float* row_out, float* col_out, const float* D,
                    float* wsp, int Bp, int N, int E, int heads, int ff, hipStream_t st) {
  const size_t M = (size_t)Bp * N;
  float* Qb[2] = {wsp, wsp + M * E};
  float* KVb[2] = {wsp + 2 * M * E, wsp + 4 * M * E};
  float* Ob[2] = {wsp + 6 * M * E, wsp + 7 * M * E};
  float* X1b[2] = {wsp + 8 * M * E, wsp + 9 * M * E};
  float* Hb[2] = {wsp + 10 * M * E, wsp + 10 * M * E + M * ff};
  const float* xin[2] = {row_in, col_in};
  float* xout[2] = {row_out, col_out};
  const int lds = N * MN_LDX * (int)sizeof(float);
  const dim3 blk(MN_THREADS);
  MnLinArgs la;
  auto launch = [&](int epi, int nout) {
    // TPW: output feature tiles per wave; 2 tiles -> 256 features per workgroup pass, 4 tiles -> 512
    const int tpw = nout % 512 == 0 ? 4 : 2;
    const dim3 grid(Bp, nout / (MN_WAVES * tpw * 16), 2);
#define MN_LAUNCH(TPWV, EPIV)                                                                                          \
  do {                                                                                                                 \
    (void)hipFuncSetAttribute((const void*)k_mn_lin<NT, TPWV, EPIV>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);      \
    hipLaunchKernelGGL((k_mn_lin<NT, TPWV, EPIV>), grid, blk, lds, st, la, N);                                         \
  } while (0)
    if (tpw == 4) { if (epi == 0) MN_LAUNCH(4, 0); else if (epi == 1) MN_LAUNCH(4, 1); else MN_LAUNCH(4, 2); }
    else { if (epi == 0) MN_LAUNCH(2, 0); else if (epi == 1) MN_LAUNCH(2, 1); else MN_LAUNCH(2, 2); }
#undef MN_LAUNCH
  };
  // Q = x_q Wq^T
  for (int s = 0; s < 2; ++s) { la.wp[s] = ws[s].wq; la.bias[s] = nullptr; la.x[s] = xin[s]; la.y[s] = Qb[s]; la.resid[s] = nullptr; la.gamma[s] = la.beta[s] = nullptr; }
  la.K = E; la.Nout = E; launch(0, E);
  // K | V = x_kv Wkv^T (the other side's embeddings)
  for (int s = 0; s < 2; ++s) { la.wp[s] = ws[s].wkv; la.x[s] = xin[1 - s]; la.y[s] = KVb[s]; }
  la.Nout = 2 * E; launch(0, 2 * E);
  // mixed-score attention
  MnAttnArgs aa;
  for (int s = 0; s < 2; ++s) { aa.q[s] = Qb[s]; aa.kv[s] = KVb[s]; aa.o[s] = Ob[s]; aa.mix[s] = ws[s].mix; }
  aa.D = D; aa.E = E; aa.heads = heads;
  const int lds_d = N * N * (int)sizeof(float);
  (void)hipFuncSetAttribute((const void*)k_mn_attn<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_d);
  hipLaunchKernelGGL(k_mn_attn<NT>, dim3(Bp, (heads + 4 * MN_ATTN_WAVES - 1) / (4 * MN_ATTN_WAVES), 2), dim3(64 * MN_ATTN_WAVES), lds_d, st, aa, N);
  // x1 = norm1(x_old + O Wo^T)
  for (int s = 0; s < 2; ++s) { la.wp[s] = ws[s].wo; la.bias[s] = nullptr; la.x[s] = Ob[s]; la.y[s] = X1b[s]; la.resid[s] = xin[s]; la.gamma[s] = ws[s].n1g; la.beta[s] = ws[s].n1b; }
  la.K = E; la.Nout = E; launch(2, E);
  // h = relu(x1 W1^T + b1)
  for (int s = 0; s < 2; ++s) { la.wp[s] = ws[s].w1; la.bias[s] = ws[s].b1; la.x[s] = X1b[s]; la.y[s] = Hb[s]; la.resid[s] = nullptr; }
  la.K = E; la.Nout = ff; launch(1, ff);
  // x2 = norm2(x1 + h W2^T + b2)
  for (int s = 0; s < 2; ++s) { la.wp[s] = ws[s].w2; la.bias[s] = ws[s].b2; la.x[s] = Hb[s]; la.y[s] = xout[s]; la.resid[s] = X1b[s]; la.gamma[s] = ws[s].n2g; la.beta[s] = ws[s].n2b; }
  la.K = ff; la.Nout = E; launch(2, E);
  return rr_check(hipGetLastError());
}

extern "C" int rr_matnet_layer(const MatNetSideW* row_side, const MatNetSideW* col_side, const float* row_in, const float* col_in,
                               float* row_out, float* col_out, const float* D, float* workspace, size_t workspace_bytes,
                               int Bp, int N, int E, int heads, int ff, hipStream_t st) {
  if (Bp <= 0 || N < 1 || N > 16 * RR_NT || row_side == nullptr || col_side == nullptr || workspace == nullptr) return RR_EINVAL;
  if (E % 256 != 0 || heads * 16 != E || ff % 256 != 0) return RR_EINVAL;       // head dim 16; tiles of 256 output features
  if (workspace_bytes < rr_matnet_workspace_bytes(Bp, N, E, ff)) return RR_EINVAL;
  if (row_in == row_out || col_in == col_out) return RR_EINVAL;
  if (!row_in || !col_in || !row_out || !col_out || !D) return RR_EINVAL;
  const MatNetSideW ws[2] = {*row_side, *col_side};
  if (N <= 32) return mn_layer<2>(ws, row_in, col_in, row_out, col_out, D, workspace, Bp, N, E, heads, ff, st);
  if (N <= 64) return mn_layer<4>(ws, row_in, col_in, row_out, col_out, D, workspace, Bp, N, E, heads, ff, st);
  return mn_layer<RR_NT>(ws, row_in, col_in, row_out, col_out, D, workspace, Bp, N, E, heads, ff, st);
}

// ------------------------------------------------------------------------------------------------
// MatNet baseline decoder (rrnco/baselines/MatNet/decoder.py; rl4co AttentionModelDecoder + PointerAttention at 256 / 16
// heads): one decoder.forward for every rollout = step context -> masked multi-head glimpse -> project_out -> pointer logits.
//   rr_matnet_linear    y = x W^T per instance (the cache: K | V | L = col_emb W_node^T; context tables row_emb W_ctx^T)
//   rr_matnet_dec_step  wave = 16 rollouts of one instance (r = s * Bp + b): q^T gathered from the context tables straight into
//                       B-operand registers, per head S^T = K_h q_h on MFMA, mask, softmax, heads^T = V_h^T P^T; glimpse^T =
//                       W_out heads^T and logits^T = L glimpse^T / sqrt(E) chain in registers (C layout = next B operand)
// The selection (MatNet's own process_logits: shift by the row maximum, clamp to [-50, -1e-4]) is rr_select's job.
// ------------------------------------------------------------------------------------------------
extern "C" int rr_matnet_linear(const void* w_packed, const float* x, float* y, int Bp, int N, int K, int Nout, hipStream_t st) {
  if (w_packed == nullptr || x == nullptr || y == nullptr || Bp <= 0 || N < 1 || N > 16 * RR_NT) return RR_EINVAL;
  if (K % 256 != 0 || Nout % 256 != 0) return RR_EINVAL;
  MnLinArgs la;
  for (int s = 0; s < 2; ++s) { la.wp[s] = static_cast<const float4*>(w_packed); la.bias[s] = nullptr; la.x[s] = x; la.y[s] = y; la.resid[s] = nullptr; la.gamma[s] = la.beta[s] = nullptr; }
  la.K = K; la.Nout = Nout;
  const int lds = N * MN_LDX * (int)sizeof(float);
  const int tpw = Nout % 512 == 0 ? 4 : 2;
  const dim3 grid(Bp, Nout / (MN_WAVES * tpw * 16), 1), blk(MN_THREADS);
#define MN_LIN1(NTV, TPWV)                                                                                            \
  do {                                                                                                                 \
    (void)hipFuncSetAttribute((const void*)k_mn_lin<NTV, TPWV, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);        \
    hipLaunchKernelGGL((k_mn_lin<NTV, TPWV, 0>), grid, blk, lds, st, la, N);                                           \
  } while (0)
  if (N <= 32) { if (tpw == 4) MN_LIN1(2, 4); else MN_LIN1(2, 2); }
  else if (N <= 64) { if (tpw == 4) MN_LIN1(4, 4); else MN_LIN1(4, 2); }
  else { if (tpw == 4) MN_LIN1(RR_NT, 4); else MN_LIN1(RR_NT, 2); }
#undef MN_LIN1
  return rr_check(hipGetLastError());
}

struct MnDecArgs {
  const float4* wo;            // pointer.project_out.weight, packed A operand [E/16][E/16][64]
  const float* kvl;            // [Bp][N][3E]: glimpse key | glimpse value | logit key
  const float* vt;             // [Bp][E][112]: the glimpse values transposed (keys along the row, zero-padded)
  const float *ctxA, *ctxB;    // [Bp][N][E]: W_ctx[:, :E] row_emb (first node), W_ctx[:, E:] row_emb (current node)
  const float* q0;             // [E] project_context(W_placeholder), used when first == nullptr (no node visited yet)
  const float *state, *wstate; // VRP: [R] vehicle_capacity - used_capacity and the state column of W_ctx [E] (then ctxA / first unused)
  const int64_t *first, *cur;  // [R]
  const uint8_t* mask;         // [R][N] action mask
  float* logits;               // [R][N]
  int Bp, N, S, E, heads;
};

template <int NT>
__global__ __launch_bounds__(256) void k_mn_dec_step(MnDecArgs a) {
  const int lane = threadIdx.x & 63, j = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int N = a.N, E = a.E, S = a.S, Bp = a.Bp;
  const int TPI = (S + 15) >> 4;
  const int tile = blockIdx.x * 4 + wave;
  if (tile >= Bp * TPI) return;
  const int b = tile / TPI, s = (tile - b * TPI) * 16 + j;
  const bool valid = s < S;
  const size_t r = (size_t)(valid ? s : 0) * Bp + b;
  const float* KVL = a.kvl + (size_t)b * N * 3 * E;
  // ---- step context q^T [E x 16 rollouts] (rl4co TSPContext: W_ctx [emb_first; emb_cur]; VRPContext: W_ctx [emb_cur; free
  // capacity]), gathered one head (16 features = one B-operand register quad) at a time, one head ahead of its use
  const float* pa = nullptr;
  const float* pb = nullptr;
  float st = 0.f;
  if (a.state != nullptr) { pb = a.ctxB + ((size_t)b * N + (int)a.cur[r]) * E + 4 * g; st = a.state[r]; }
  else if (a.first != nullptr) {
    pa = a.ctxA + ((size_t)b * N + (int)a.first[r]) * E + 4 * g;
    pb = a.ctxB + ((size_t)b * N + (int)a.cur[r]) * E + 4 * g;
  }
  auto load_q = [&](int h) -> f32x4 {
    f32x4 qh;
    if (pb == nullptr) { const float4 u = rr_ld4(a.q0 + 16 * h + 4 * g); qh[0] = u.x; qh[1] = u.y; qh[2] = u.z; qh[3] = u.w; return qh; }
    const float4 v = rr_ld4(pb + 16 * h);
    if (pa != nullptr) {
      const float4 u = rr_ld4(pa + 16 * h);
      qh[0] = u.x + v.x; qh[1] = u.y + v.y; qh[2] = u.z + v.z; qh[3] = u.w + v.w;
    } else {
      const float4 ws = rr_ld4(a.wstate + 16 * h + 4 * g);
      qh[0] = fmaf(ws.x, st, v.x); qh[1] = fmaf(ws.y, st, v.y); qh[2] = fmaf(ws.z, st, v.z); qh[3] = fmaf(ws.w, st, v.w);
    }
    return qh;
  };
  // action mask of this lane's keys (4 per key tile), once
  const uint8_t* mk = a.mask + r * N;
  uint32_t okbits = 0;
#pragma unroll
  for (int kt = 0; kt < NT; ++kt)
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int k2 = kt * 16 + 4 * g + rr;
      if (k2 < N && mk[k2 < N ? k2 : 0] != 0) okbits |= 1u << (4 * kt + rr);
    }
  int keyoff[NT];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) { int key = kt * 16 + j; key = key < N ? key : N - 1; keyoff[kt] = key * 3 * E + 4 * g; }
  const float* pv0 = a.vt + (size_t)b * E * 112 + (size_t)j * 112 + 4 * g;       // V^T [E][112]: four keys per 16-byte load
  // ---- masked multi-head glimpse, one head at a time, the next head's operands (q, K fragments, V^T fragments) in flight;
  // heads^T stays in registers (B operand of project_out)
  f32x4 H[16];
  f32x4 qn = load_q(0);
  float4 kn[NT], vn[NT];
#pragma unroll
  for (int kt = 0; kt < NT; ++kt) { kn[kt] = rr_ld4(KVL + keyoff[kt]); vn[kt] = rr_ld4(pv0 + 16 * kt); }
#pragma unroll
  for (int h = 0; h < 16; ++h) {      // unrolled: H[h] must be register-indexed
    const f32x4 qh = qn;
    float4 kf[NT], vv[NT];
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) { kf[kt] = kn[kt]; vv[kt] = vn[kt]; }
    if (h + 1 < 16) {
      qn = load_q(h + 1);
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) { kn[kt] = rr_ld4(KVL + keyoff[kt] + 16 * (h + 1)); vn[kt] = rr_ld4(pv0 + (size_t)(16 * (h + 1)) * 112 + 16 * kt); }
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 sc[NT];
    float mxv = -INFINITY;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {
      f32x4 c = rr_zero4();
      c = rr_mfma(kf[kt].x, qh[0], c); c = rr_mfma(kf[kt].y, qh[1], c); c = rr_mfma(kf[kt].z, qh[2], c); c = rr_mfma(kf[kt].w, qh[3], c);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        c[rr] = ((okbits >> (4 * kt + rr)) & 1u) ? c[rr] * 0.25f : -INFINITY;
        mxv = fmaxf(mxv, c[rr]);
      }
      sc[kt] = c;
    }
    mxv = rr_max_g(mxv);
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) { sc[kt][rr] = rr_exp(sc[kt][rr] - mxv); sum += sc[kt][rr]; }
    const float inv = 1.0f / rr_sum_g(sum);
    f32x4 o = rr_zero4();
#pragma unroll
    for (int kt = 0; kt < NT; ++kt) {                                              // keys >= N carry zero probability
      o = rr_mfma(vv[kt].x, sc[kt][0], o); o = rr_mfma(vv[kt].y, sc[kt][1], o); o = rr_mfma(vv[kt].z, sc[kt][2], o); o = rr_mfma(vv[kt].w, sc[kt][3], o);
    }
    H[h][0] = o[0] * inv; H[h][1] = o[1] * inv; H[h][2] = o[2] * inv; H[h][3] = o[3] * inv;
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- glimpse^T = W_out heads^T (project_out, no bias)
  f32x4 G[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    f32x4 c0 = rr_zero4(), c1 = rr_zero4();
#pragma unroll
    for (int kk = 0; kk < 16; kk += 2) {
      const float4 w0 = a.wo[(size_t)(t * 16 + kk) * 64 + lane], w1 = a.wo[(size_t)(t * 16 + kk + 1) * 64 + lane];
      c0 = rr_mfma(w0.x, H[kk][0], c0); c1 = rr_mfma(w1.x, H[kk + 1][0], c1);
      c0 = rr_mfma(w0.y, H[kk][1], c0); c1 = rr_mfma(w1.y, H[kk + 1][1], c1);
      c0 = rr_mfma(w0.z, H[kk][2], c0); c1 = rr_mfma(w1.z, H[kk + 1][2], c1);
      c0 = rr_mfma(w0.w, H[kk][3], c0); c1 = rr_mfma(w1.w, H[kk + 1][3], c1);
    }
    G[t] = c0 + c1;
  }
  // ---- logits^T = L glimpse^T / sqrt(E)
  const float isq = 1.0f / sqrtf((float)E);
#pragma unroll 1
  for (int kt = 0; kt < NT; ++kt) {
    int key = kt * 16 + j; key = key < N ? key : N - 1;
    const float* pl = KVL + (size_t)key * 3 * E + 2 * E + 4 * g;
    f32x4 c0 = rr_zero4(), c1 = rr_zero4();
#pragma unroll
    for (int kk = 0; kk < 16; kk += 2) {
      const float4 l0 = rr_ld4(pl + 16 * kk), l1 = rr_ld4(pl + 16 * (kk + 1));
      c0 = rr_mfma(l0.x, G[kk][0], c0); c1 = rr_mfma(l1.x, G[kk + 1][0], c1);
      c0 = rr_mfma(l0.y, G[kk][1], c0); c1 = rr_mfma(l1.y, G[kk + 1][1], c1);
      c0 = rr_mfma(l0.z, G[kk][2], c0); c1 = rr_mfma(l1.z, G[kk + 1][2], c1);
      c0 = rr_mfma(l0.w, G[kk][3], c0); c1 = rr_mfma(l1.w, G[kk + 1][3], c1);
    }
    if (valid) {
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int k2 = kt * 16 + 4 * g + rr;
        if (k2 < N) a.logits[r * N + k2] = (c0[rr] + c1[rr]) * isq;
      }
    }
  }
}

extern "C" int rr_matnet_dec_step(const void* wo_packed, const float* kvl, const float* vt, const float* ctxA, const float* ctxB, const float* q0,
                                  const float* state, const float* wstate,
                                  const int64_t* first, const int64_t* cur, const uint8_t* mask, float* logits,
                                  int Bp, int N, int S, int E, int heads, hipStream_t st) {
  if (!wo_packed || !kvl || !vt || !mask || !logits || Bp <= 0 || S < 1 || N < 2 || N > 16 * RR_NT) return RR_EINVAL;
  if (E != 256 || heads != 16) return RR_EINVAL;                       // registers are sized for the matnet.yaml configuration
  if (state != nullptr) { if (wstate == nullptr || cur == nullptr || ctxB == nullptr) return RR_EINVAL; }
  else if (first != nullptr ? (cur == nullptr || ctxA == nullptr || ctxB == nullptr) : q0 == nullptr) return RR_EINVAL;
  MnDecArgs a;
  a.wo = static_cast<const float4*>(wo_packed); a.kvl = kvl; a.vt = vt; a.ctxA = ctxA; a.ctxB = ctxB; a.q0 = q0; a.state = state; a.wstate = wstate; a.first = first; a.cur = cur;
  a.mask = mask; a.logits = logits; a.Bp = Bp; a.N = N; a.S = S; a.E = E; a.heads = heads;
  const int tiles = Bp * ((S + 15) / 16);
  const dim3 grid((tiles + 3) / 4), blk(256);
  if (N <= 32) hipLaunchKernelGGL(k_mn_dec_step<2>, grid, blk, 0, st, a);
  else if (N <= 64) hipLaunchKernelGGL(k_mn_dec_step<4>, grid, blk, 0, st, a);
  else hipLaunchKernelGGL(k_mn_dec_step<RR_NT>, grid, blk, 0, st, a);
  return rr_check(hipGetLastError());
}

