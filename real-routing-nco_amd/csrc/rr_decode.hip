// Persistent construction-rollout kernel: the POMO multi-start decode loop of RRNetPolicy.forward
// (rrnco/models/policy.py:208-228) fused with RRNetDecoder.forward (decoder.py:151-206), the decoding
// strategy step (decoding.py:219-298, 311-361) and env.step (envs/*/env.py) for one instance per
// workgroup.  All S starts of an instance advance in lock-step inside the workgroup; instances are
// independent, so there is no inter-workgroup communication and no host round trip per step.
//
// Per step, per workgroup (8 waves):
//   1. q   = ctxA[first] + ctxB[cur] (+ state columns)            gather from L2 into LDS  [rollout][E]
//   2. MHA : wave = head.  S^T = K_h q_h^T (MFMA), masked softmax over keys in registers
//            (wavefront shuffles over the 4 lane groups), O^T = V_h^T P^T (MFMA), G = O + q in place
//   3. FFN : F = G + W2 relu(W1 G + b1) + b2, hidden in 4 chunks of 128 through LDS (MFMA)
//   4. logits^T = L F^T / sqrt(E) (MFMA); inductive bias from the LDS-staged distance tile
//            logits = log(exp(l - alpha*D[cur,:] - beta*Dur[cur,:]) + 1e-6); 10*tanh; mask; log-softmax;
//            greedy argmax (lowest index on ties) | Gumbel-max sample | given action; env.step
#include "rr_common.h"
#include <stdlib.h>

#define DEC_THREADS 512
#define ROWS 112               // rollouts per workgroup (7 tiles of 16)
#define QLD 128

struct DecW {
  const float4 *w1, *w2;       // pointer.ffn.lins.{0,1} packed A operands
  const float *b1, *b2;
  const float* q0;             // project_context(W_placeholder) [E] (ATSP first step without multistart)
  const float* wstate;         // VRP: step-context state columns of project_context [nstate][E]
  float alpha, beta;
  const void *w1s, *w2s;       // optional: the same two matrices x 2^RR_WS as two-piece fp16 images (packing.pack_a_f16u) for the split rollout
  const float* b1s;            // 2^RR_WS b1 (accumulator seeds of the split rollout's hidden tiles)
};

struct RolloutIO {
  // per-instance caches (rr_dec_cache)
  const float *K, *Vt, *L, *ctxA, *ctxB;
  const float *D, *Dur;                  // normalised distance (and duration) matrices [Bp][N][N]
  // problem data
  const float* demand;                   // RCVRP: [Bp][N-1] customer demands (already / capacity); RCVRPTW: demand_linehaul [Bp][N]
  const float *tw, *service;             // RCVRPTW: time_windows [Bp][N][2], service_time [Bp][N]
  // rollout state, r = s*Bp + b
  int64_t *cur, *first;                  // [R]
  uint8_t *mask;                         // [R][N] action mask, 1 = feasible
  uint8_t *visited;                      // [R][N] (VRP)
  float *used, *vcap;                    // [R] (VRP)
  float *ctime, *rlen;                   // [R] (RCVRPTW) current_time, current_route_length
  uint8_t *done;                         // [R]
  // outputs
  int64_t* actions; float* logp;         // [R][T]
  float* logits_out;                     // optional [R][N]: decoder.forward output of the FIRST step
  const int64_t* actions_in;             // evaluate mode
  int* steps_out;                        // [1] max steps any workgroup executed (atomicMax)
  int Bp, N, S, T, t0, nsteps;
  int mode;                              // 0 greedy, 1 sampling, 2 evaluate
  int use_placeholder;                   // first step uses DecW.q0
  int set_first;                         // ATSP: first_node := action of step t0==0
  int write_state;                       // write cur/first/mask/... back (0 for a pure decoder.forward)
  int logits_only;                       // stop after writing logits_out (no selection)
  int stagger;                           // k_rollout_w: initial delay of waves 4-7, in units of ~8k cycles
  float tanh_clip, temperature;
  unsigned long long seed;
  // MTVRP variants (NULL = vrptw preset): used_capacity_backhaul [R] (state), open_route [Bp], distance_limit [Bp];
  // demand_backhaul [Bp][N] and backhaul_class [Bp] are needed by the in-kernel env.step (not by a logits_only launch)
  float* used_b;
  const uint8_t* open_route;
  const float* dist_limit;
  const float* demand_b;
  const int32_t* bclass;
  // Training dump (all NULL / 0 outside a training step): per decoder evaluation (instance b, step, start s), row
  // m = (b*dumpT + step)*S + s, what the hand-written backward (csrc/rr_train_dec.hip) needs and the rollout has in registers
  // anyway: the glimpse input of the pointer MLP g0 [m][128], its output g [m][128], and meta [m][8] = the 4 action-mask
  // words the decision saw, the node it was taken at, the chosen node, 1 if the row is live (not a finished / padding
  // rollout), 0; VRP: the step-context state scalars scal [m][4] (available load, current time, open route, remaining distance).
  float* dump_g0; float* dump_g; uint32_t* dump_meta; float* dump_scal;
  int dumpT;
  int use_split;                         // 1: this launch runs on the fp16 matrix pipe with two-piece split operands (rr_common.h), if DecW has w1s / w2s and Ks / Vts / Ls are given
  const void *Ks, *Vts, *Ls;             // fp16 two-piece images of K / Vt / L (rr_pack_f16x2), same shapes and byte offsets
  int* status;                           // optional: bit 2 <- a split launch met a non-finite log-probability (an operand left the fp16 range)
};

template <int NT, int PROB>  // PROB 0 = ATSP, 1 = RCVRP
__global__ __launch_bounds__(DEC_THREADS, 2) void k_rollout(DecW w, RolloutIO io) {
  __shared__ __attribute__((aligned(16))) float Qs[ROWS * QLD];   // q -> G -> F
  __shared__ __attribute__((aligned(16))) float Hs[ROWS * QLD];   // FFN hidden chunk
  __shared__ __attribute__((aligned(16))) float Ds[RR_MAXN * RR_MAXN];
  __shared__ int s_cur[ROWS], s_first[ROWS];
  __shared__ uint32_t s_av[ROWS][4];      // availability bitmask (action_mask)
  __shared__ uint32_t s_vis[ROWS][4];     // visited bitmask (VRP)
  __shared__ float s_used[ROWS], s_cap[ROWS];
  __shared__ int s_done[ROWS];
  __shared__ int s_alldone;

  const int b = blockIdx.x, chunk = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int j = lane & 15, g = lane >> 4;
  const int N = io.N, Bp = io.Bp;
  const int s0 = chunk * ROWS;
  const int nrows = min(ROWS, io.S - s0);
  const int fb = 16 * wave;
  const size_t eoff = (size_t)b * N * RR_E;
  const float* Kb = io.K + eoff;
  const float* Lb = io.L + eoff;
  const float* Vtb = io.Vt + (size_t)b * RR_E * 112;
  const float* cA = io.ctxA ? io.ctxA + eoff : nullptr;
  const float* cB = io.ctxB + eoff;

  // ---- load state
  for (int i = tid; i < N * N; i += DEC_THREADS) Ds[i] = io.D[(size_t)b * N * N + i];
  for (int i = tid; i < ROWS * QLD; i += DEC_THREADS) { Qs[i] = 0.f; Hs[i] = 0.f; }
  if (tid < ROWS) {
    int row = tid;
    int cur = 0, first = 0, dn = 1;
    uint32_t av[4] = {0, 0, 0, 0}, vs[4] = {0, 0, 0, 0};
    float used = 0.f, cap = 1.f;
    if (row < nrows) {
      size_t r = (size_t)(s0 + row) * Bp + b;
      cur = (int)io.cur[r];
      first = io.first ? (int)io.first[r] : 0;
      dn = io.done ? io.done[r] : 0;
      for (int k = 0; k < N; ++k) {
        if (io.mask[r * N + k]) av[k >> 5] |= 1u << (k & 31);
        if (PROB == 1 && io.visited[r * N + k]) vs[k >> 5] |= 1u << (k & 31);
      }
      if (PROB == 1) { used = io.used[r]; cap = io.vcap[r]; }
    }
    s_cur[row] = cur; s_first[row] = first; s_done[row] = dn;
    s_used[row] = used; s_cap[row] = cap;
#pragma unroll
    for (int q = 0; q < 4; ++q) { s_av[row][q] = av[q]; s_vis[row][q] = vs[q]; }
  }
  __syncthreads();

  int step = 0;
  for (;; ++step) {
    if (io.nsteps > 0) { if (step >= io.nsteps) break; }
    else {
      // data-dependent length (VRP): run until every rollout of this instance is done
      if (tid == 0) { int ad = 1; for (int r2 = 0; r2 < nrows; ++r2) ad &= s_done[r2]; s_alldone = ad; }
      __syncthreads();
      if (s_alldone || step >= io.T - io.t0) break;
    }
    const int t = io.t0 + step;

    // ---- 1. step context q -> Qs
    for (int e = tid; e < nrows * (RR_E / 4); e += DEC_THREADS) {
      int row = e >> 5, c4 = (e & 31) * 4;
      float4 q;
      if (PROB == 0) {
        if (io.use_placeholder && step == 0) q = rr_ld4(w.q0 + c4);
        else {
          float4 a = rr_ld4(cA + s_first[row] * RR_E + c4), c = rr_ld4(cB + s_cur[row] * RR_E + c4);
          q = make_float4(a.x + c.x, a.y + c.y, a.z + c.z, a.w + c.w);
        }
      } else {
        float4 c = rr_ld4(cB + s_cur[row] * RR_E + c4), ws = rr_ld4(w.wstate + c4);
        float rem = s_cap[row] - s_used[row];   // VRPContext: vehicle_capacity - used_capacity
        q = make_float4(fmaf(ws.x, rem, c.x), fmaf(ws.y, rem, c.y), fmaf(ws.z, rem, c.z), fmaf(ws.w, rem, c.w));
      }
      rr_st4(Qs + row * QLD + c4, q);
    }
    __syncthreads();

    // ---- 2. masked multi-head attention, wave = head (decoder.py:308-323)
    {
      const int h = wave;
      float4 kf[NT], vf[NT];
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) {
        int key = kt * 16 + j; key = key < N ? key : N - 1;
        kf[kt] = rr_ld4(Kb + key * RR_E + 16 * h + 4 * g);          // A[i=key][k=dim]
        vf[kt] = rr_ld4(Vtb + (16 * h + j) * 112 + 16 * kt + 4 * g);  // A[i=dim][k=key], 4 regs = 4 keys
      }
      const int ntr = (nrows + 15) >> 4;
      for (int nt = 0; nt < ntr; ++nt) {
        const int row = nt * 16 + j;
        float4 qf = rr_ld4(Qs + row * QLD + 16 * h + 4 * g);
        uint32_t av0 = s_av[row][0], av1 = s_av[row][1], av2 = s_av[row][2], av3 = s_av[row][3];
        f32x4 sc[NT];
        float mx = -INFINITY;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          f32x4 a = rr_zero4();
          a = rr_mfma(kf[kt].x, qf.x, a); a = rr_mfma(kf[kt].y, qf.y, a);
          a = rr_mfma(kf[kt].z, qf.z, a); a = rr_mfma(kf[kt].w, qf.w, a);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            int key = kt * 16 + 4 * g + r;
            uint32_t wsel = (key >> 5) == 0 ? av0 : (key >> 5) == 1 ? av1 : (key >> 5) == 2 ? av2 : av3;
            bool ok = key < N && ((wsel >> (key & 31)) & 1u);
            float v = ok ? a[r] * 0.25f : -INFINITY;   // 1/sqrt(head_dim)
            a[r] = v; mx = fmaxf(mx, v);
          }
          sc[kt] = a;
        }
        mx = rr_max_g(mx);
        if (mx == -INFINITY) mx = 0.f;   // padding rollouts (no feasible key): keep everything finite
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
          for (int r = 0; r < 4; ++r) { float e = rr_exp(sc[kt][r] - mx); sc[kt][r] = e; sum += e; }
        sum = rr_sum_g(sum);
        const float inv = sum > 0.f ? __builtin_amdgcn_rcpf(sum) : 0.f;
        f32x4 o0 = rr_zero4(), o1 = rr_zero4();
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          o0 = rr_mfma(vf[kt].x, sc[kt][0] * inv, o0);
          o1 = rr_mfma(vf[kt].y, sc[kt][1] * inv, o1);
          o0 = rr_mfma(vf[kt].z, sc[kt][2] * inv, o0);
          o1 = rr_mfma(vf[kt].w, sc[kt][3] * inv, o1);
        }
        // glimpse = heads + query (decoder.py:294): O^T[dim 4g+r][rollout j] -> Qs[row][16h+4g..]
        float4 qo = qf;
        qo.x += o0[0] + o1[0]; qo.y += o0[1] + o1[1]; qo.z += o0[2] + o1[2]; qo.w += o0[3] + o1[3];
        rr_st4(Qs + row * QLD + 16 * h + 4 * g, qo);
      }
    }
    __syncthreads();

    // ---- 3. pointer MLP with residual (decoder.py:296): F = G + W2 relu(W1 G + b1) + b2
    {
      f32x4 fa[NT];
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) fa[nt] = rr_zero4();
      for (int c = 0; c < RR_FF / 128; ++c) {
        f32x4 ha[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) ha[nt] = rr_zero4();
        rr_gemm_wx<NT>(ha, w.w1 + (size_t)(c * 8 + wave) * 8 * 64, 0, 8, Qs, QLD, 0, ROWS, lane);
        rr_add_bias<NT>(ha, w.b1, c * 128 + fb, lane);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
          for (int r = 0; r < 4; ++r) ha[nt][r] = fmaxf(ha[nt][r], 0.f);
        if (c > 0) __syncthreads();   // previous chunk's W2 pass has consumed Hs
        rr_store_tiles<NT>(ha, Hs, QLD, fb, ROWS, lane);
        __syncthreads();
        rr_gemm_wx<NT>(fa, w.w2 + (size_t)wave * 32 * 64, c * 8, 8, Hs, QLD, 0, ROWS, lane);
      }
      rr_add_bias<NT>(fa, w.b2, fb, lane);
      f32x4 gt[NT];
      rr_load_tiles<NT>(gt, Qs, QLD, fb, ROWS, lane);
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) fa[nt] = gt[nt] + fa[nt];
      __syncthreads();   // every wave has finished reading G (W1 passes)
      rr_store_tiles<NT>(fa, Qs, QLD, fb, ROWS, lane);
    }
    __syncthreads();

    // ---- 4. logits, inductive bias, selection, env.step: wave = rollout tile
    if (wave < ((nrows + 15) >> 4)) {
      const int nt = wave;
      const int row = nt * 16 + j;
      const bool rvalid = row < nrows;
      const size_t r = (size_t)(s0 + (rvalid ? row : 0)) * Bp + b;
      f32x4 la[NT];
#pragma unroll
      for (int kt = 0; kt < NT; ++kt) la[kt] = rr_zero4();
#pragma unroll 1
      for (int kk = 0; kk < 8; ++kk) {
        float4 bf = rr_ld4(Qs + row * QLD + kk * 16 + 4 * g);
#pragma unroll
        for (int kt = 0; kt < NT; ++kt) {
          int key = kt * 16 + j; key = key < N ? key : N - 1;
          float4 af = rr_ld4(Lb + key * RR_E + kk * 16 + 4 * g);
          la[kt] = rr_mfma(af.x, bf.x, la[kt]); la[kt] = rr_mfma(af.y, bf.y, la[kt]);
          la[kt] = rr_mfma(af.z, bf.z, la[kt]); la[kt] = rr_mfma(af.w, bf.w, la[kt]);
        }
      }
      const int cur = s_cur[row];
      const uint32_t av0 = s_av[row][0], av1 = s_av[row][1], av2 = s_av[row][2], av3 = s_av[row][3];
      const float sqe = sqrtf((float)RR_E);
      float mx = -INFINITY;
#pragma unroll
      for (int kt = 0; kt < NT; ++kt)
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          int key = kt * 16 + 4 * g + rg;
          float v = -INFINITY;
          if (key < N) {
            float l = la[kt][rg] * (1.0f / sqe);                      // decoder.py:300-302
            float bias = w.alpha * Ds[cur * N + key];                  // decoder.py:191-193
            l = rr_log(rr_exp(l - bias) + 1e-6f);                      // decoder.py:198
            if (io.logits_out && step == 0 && rvalid) io.logits_out[r * N + key] = l;
            uint32_t wsel = (key >> 5) == 0 ? av0 : (key >> 5) == 1 ? av1 : (key >> 5) == 2 ? av2 : av3;
            if ((wsel >> (key & 31)) & 1u) {
              v = io.tanh_clip > 0.f ? rr_tanh(l) * io.tanh_clip : l;   // decoding.py:342-343
              v = v * (1.0f / io.temperature);                         // decoding.py:350
            }
          }
          la[kt][rg] = v;
          mx = fmaxf(mx, v);
        }
      if (!io.logits_only) {
        mx = rr_max_g(mx);
        if (mx == -INFINITY) mx = 0.f;
        float sum = 0.f;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) sum += rr_exp(la[kt][rg] - mx);
        sum = rr_sum_g(sum);
        const float lse = rr_log(sum);
        // log-probs; pick
        float bv = -INFINITY, blp = 0.f; int bi = 0x7fffffff;
        const int want = (io.mode == 2 && rvalid) ? (int)io.actions_in[r * io.T + t] : -1;
#pragma unroll
        for (int kt = 0; kt < NT; ++kt)
#pragma unroll
          for (int rg = 0; rg < 4; ++rg) {
            int key = kt * 16 + 4 * g + rg;
            float lp = la[kt][rg] - mx - lse;
            if (io.mode == 2) { if (key == want) { bv = 1.f; bi = key; blp = lp; } }
            else if (la[kt][rg] > -INFINITY) {
              float sv = io.mode == 1 ? lp + rr_gumbel(io.seed, (uint32_t)r, (uint32_t)t, (uint32_t)key) : lp;
              if (sv > bv) { bv = sv; bi = key; blp = lp; }   // ascending keys within the lane: first max kept
            }
          }
#pragma unroll
        for (int o = 16; o <= 32; o <<= 1) {
          float ov = __shfl_xor(bv, o), olp = __shfl_xor(blp, o); int oi = __shfl_xor(bi, o);
          if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; blp = olp; }
        }
        // ---- env.step
        if (rvalid && g == 0) {
          const int a = bi;
          const int was_done = s_done[row];
          if (!(io.nsteps <= 0 && was_done)) {   // finished VRP rollouts keep their zero padding
            io.actions[r * io.T + t] = a;
            io.logp[r * io.T + t] = blp;
          }
          s_cur[row] = a;
          if (PROB == 0) {
            if (io.set_first && t == 0) s_first[row] = a;
            uint32_t nw[4] = {av0, av1, av2, av3};
            nw[a >> 5] &= ~(1u << (a & 31));
            s_av[row][0] = nw[0]; s_av[row][1] = nw[1]; s_av[row][2] = nw[2]; s_av[row][3] = nw[3];
            s_done[row] = (nw[0] | nw[1] | nw[2] | nw[3]) == 0;      // atsp/env.py:90
          }
        }
        if (PROB == 1) {
          // rcvrp/env.py:90-122 + get_action_mask :183-195, all 4 lane groups of a rollout cooperate
          const int a = bi;
          const float* dem = io.demand + (size_t)b * (N - 1);
          int ci = a - 1; ci = ci < 0 ? 0 : (ci > N - 2 ? N - 2 : ci);
          const float u = (s_used[row] + dem[ci]) * (a != 0 ? 1.0f : 0.0f);
          const float cap = s_cap[row];
          uint32_t vs[4] = {s_vis[row][0], s_vis[row][1], s_vis[row][2], s_vis[row][3]};
          vs[a >> 5] |= 1u << (a & 31);
          uint32_t nw[4] = {0, 0, 0, 0};
          int nvis = 0;
          for (int key = g; key < N; key += 4) {     // keys strided over the 4 lane groups
            bool v = (vs[key >> 5] >> (key & 31)) & 1u;
            nvis += v;
            if (key >= 1) {
              bool exceeds = dem[key - 1] + u > cap;
              if (!(v || exceeds)) nw[key >> 5] |= 1u << (key & 31);
            }
          }
#pragma unroll
          for (int q = 0; q < 4; ++q) { nw[q] |= __shfl_xor(nw[q], 16); nw[q] |= __shfl_xor(nw[q], 32); }
          nvis += __shfl_xor(nvis, 16); nvis += __shfl_xor(nvis, 32);
          const bool anyfree = (nw[0] | nw[1] | nw[2] | nw[3]) != 0;
          const bool mask_depot = (a == 0) && anyfree;
          if (!mask_depot) nw[0] |= 1u;
          if (rvalid && g == 0) {
            s_used[row] = u;
#pragma unroll
            for (int q = 0; q < 4; ++q) { s_av[row][q] = nw[q]; s_vis[row][q] = vs[q]; }
            s_done[row] = nvis == N;
          }
        }
      }
    }
    __syncthreads();
    if (io.logits_only) { ++step; break; }
  }

  // ---- write state back
  if (io.write_state && tid < nrows) {
    int row = tid;
    size_t r = (size_t)(s0 + row) * Bp + b;
    io.cur[r] = s_cur[row];
    if (io.first) io.first[r] = s_first[row];
    if (io.done) io.done[r] = (uint8_t)s_done[row];
    for (int k = 0; k < N; ++k) {
      io.mask[r * N + k] = (s_av[row][k >> 5] >> (k & 31)) & 1u;
      if (PROB == 1) io.visited[r * N + k] = (s_vis[row][k >> 5] >> (k & 31)) & 1u;
    }
    if (PROB == 1) io.used[r] = s_used[row];
  }
  if (io.steps_out && tid == 0) atomicMax(io.steps_out, step);
}

#include "rr_rollout_w.inc"

// fp32 -> two-piece fp16 image of 2^RR_KS x (rr_common.h, second form): every group of four values becomes its four hi and four
// lo halves (16 bytes in, 16 bytes out, same offset): K / Vt / L of the decoder cache for the split rollout's attention and
// logits.  Range guard: a value that is not finite or leaves the fp16 range after the scale sets bit 0 of *status.
__global__ __launch_bounds__(256) void k_pack_f16x2(const float4* __restrict__ src, float4* __restrict__ dst, long long n4, int* status) {
  bool bad = false;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = src[i];
    const float sc = (float)(1 << RR_KS);
    const float x[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};
#pragma unroll
    for (int q = 0; q < 4; ++q) bad = bad || !(fabsf(x[q]) < RR_F16_LIMIT);
    const rr_f16x8 s = rr_usplit4s(x);                          // [lo | hi]
    dst[i] = __builtin_bit_cast(float4, rr_cat4(rr_hi4(s), rr_lo4(s)));
  }
  if (status != nullptr && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(status, 1);
}
extern "C" int rr_pack_f16x2(const float* src, void* dst, long long n_floats, int* status, hipStream_t st) {
  if (n_floats <= 0) return RR_OK;
  if (n_floats % 4 != 0) return RR_EINVAL;
  const long long n4 = n_floats / 4;
  const long long blocks = (n4 + 255) / 256;
  hipLaunchKernelGGL(k_pack_f16x2, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, st, (const float4*)src, (float4*)dst, n4, status);
  return rr_check(hipGetLastError());
}

extern "C" int rr_rollout(const DecW* w, const RolloutIO* io, int prob, hipStream_t st) {
  if (w == nullptr || io == nullptr) return RR_EINVAL;
  const int N = io->N, S = io->S;
  if (io->Bp <= 0 || N < 2 || N > RR_MAXN || S < 1 || io->T < 1 || prob < 0 || prob > 2) return RR_EINVAL;
  if (prob == 2 && (io->Dur == nullptr || io->tw == nullptr || io->service == nullptr || io->ctime == nullptr)) return RR_EINVAL;
  // backhauls / open routes / distance limits (rmtvrp/env.py:343-428): RCVRPTW launches only, all variant inputs together
  {
    const bool any_var = io->used_b || io->open_route || io->dist_limit || io->demand_b || io->bclass;
    const bool ctx_var = io->used_b && io->open_route && io->dist_limit;
    if (any_var && (prob != 2 || !ctx_var)) return RR_EINVAL;
    if (any_var && !io->logits_only && !(io->demand_b && io->bclass)) return RR_EINVAL;
  }
  if (io->mode == 2 && io->actions_in == nullptr) return RR_EINVAL;
  static const int variant = getenv("RR_ROLLOUT_VARIANT") ? atoi(getenv("RR_ROLLOUT_VARIANT")) : 1;
  const char* es = getenv("RR_MLP_SPLIT");
  // RolloutIO.use_split decides; the environment variable only forces it on for callers that leave the field 0
  const bool mlp_split = (io->use_split != 0 || (es != nullptr && atoi(es) != 0 && io->use_split < 0)) && w->w1s != nullptr && w->w2s != nullptr && w->b1s != nullptr &&
                         io->Ks != nullptr && io->Vts != nullptr && io->Ls != nullptr;
  if (variant == 0 && prob < 2) {   // workgroup-per-instance variant (kept for A/B measurements; ATSP / RCVRP only)
    dim3 grid(io->Bp, (S + ROWS - 1) / ROWS), blk(DEC_THREADS);
    const int need = N > (S < ROWS ? S : ROWS) ? N : (S < ROWS ? S : ROWS);
#define RR_LAUNCH(NTV)                                                                       \
  do {                                                                                       \
    if (prob == 0) hipLaunchKernelGGL((k_rollout<NTV, 0>), grid, blk, 0, st, *w, *io);       \
    else hipLaunchKernelGGL((k_rollout<NTV, 1>), grid, blk, 0, st, *w, *io);                 \
  } while (0)
    if (need <= 32) RR_LAUNCH(2);
    else if (need <= 64) RR_LAUNCH(4);
    else RR_LAUNCH(7);
#undef RR_LAUNCH
    return rr_check(hipGetLastError());
  }
  // tail packing (rr_rollout_w.inc): the S % 16 left-over rollouts of 16 / (S % 16) consecutive instances share one tile
  // Off by default since the logit keys of ordinary tiles live in LDS: a packed tile runs its attention and logits once per
  // instance it spans, with every operand from L2, and a workgroup of such tiles takes 3 (a 2-round launch) to ~6 (the 14-round
  // headline launch) ordinary workgroup times for the work of 4 quarter-full tiles.  Measured: headline rollout 73.9 ms packed,
  // 69.2 ms not (14 full rounds of 256 workgroups); 512-instance training rollout 18.4 / 12.2 ms.  RR_TAIL_PACK=1 turns it on.
  static const int pack = getenv("RR_TAIL_PACK") ? atoi(getenv("RR_TAIL_PACK")) : 0;
  const int tail_m = S & 15;
  const int tail_g = (pack && S > 16 && tail_m > 0 && tail_m <= 8 && io->Bp > 1) ? 16 / tail_m : 0;
  const int ntask = tail_g ? (io->Bp + tail_g - 1) / tail_g + io->Bp * (S / 16) : io->Bp * ((S + 15) / 16);
  dim3 grid((ntask + WWAVES - 1) / WWAVES), blk(WTHREADS);
  // LDS-staged distance tiles: as many instances (<= 2) as leave room for 160 KB / WTHREADS-sized workgroups per CU
  const size_t tile = (size_t)N * N * sizeof(float);
  const size_t per_wg = (size_t)(160 * 1024) / (512 / WTHREADS) - 512;
  const int lds_inst = 2 * tile <= per_wg ? 2 : (tile <= per_wg ? 1 : 0);
  const size_t shmem = (size_t)lds_inst * tile;
  // split variant: two 16 KB weight stage buffers, then the logit-key images of two instances (NT x 8 KB each) if they fit
  const size_t wstg = 2 * 16384;
  // split variant: XCD-aware workgroup order (rr_rollout_w.inc), grid padded to (heavy rounded up to 8) + 8 * per
  const int ntail_l = tail_g ? (io->Bp + tail_g - 1) / tail_g : 0;
  const int theavy = ((ntail_l + WWAVES - 1) / WWAVES + 7) & ~7;
  const int nwg = (ntask + WWAVES - 1) / WWAVES;
  const int per_x = nwg > theavy ? (nwg - theavy + 7) / 8 : 0;
  const dim3 grid_s(theavy + 8 * per_x), blk_s = blk;
  const int mode = io->logits_only ? 3 : io->mode;
  static const bool inst_on = getenv("RR_ROLLOUT_INST") == nullptr || atoi(getenv("RR_ROLLOUT_INST")) != 0;
#define RR_LAUNCHW4(NTV, P, M, SP)                                                                            \
  do {                                                                                                       \
    const int lds_inst_s = wstg + 2 * (size_t)(NTV) * 8192 <= per_wg ? 2 : 0;                               \
    const size_t shm = (SP) ? wstg + (size_t)lds_inst_s * (NTV) * 8192 : shmem;              \
    (void)hipFuncSetAttribute((const void*)k_rollout_w<NTV, P, M, SP>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
    hipLaunchKernelGGL((k_rollout_w<NTV, P, M, SP>), (SP) ? grid_s : grid, (SP) ? blk_s : blk, shm, st, *w, *io, tail_g, \
                       (SP) ? lds_inst_s : lds_inst);                                                        \
  } while (0)
  /* instance mode (rr_rollout_w.inc): one instance per workgroup, K + L in LDS, the eighth wave loads the weight ring: */ \
  /* three 16 KB weight stage buffers + the L and K images = all 160 KB                                                 */
#define RR_LAUNCHW4I(NTV, P, M)                                                                              \
  do {                                                                                                       \
    if constexpr ((NTV) == 7) {                                                                              \
      const size_t shm_i = 3 * 16384 + 2 * (size_t)(NTV) * 8192;                                             \
      if (io->use_split == 2 && (P) <= 2) {        /* 16-mixed: one fp16 piece per operand (rr_rollout_w.inc, HALF) */ \
        (void)hipFuncSetAttribute((const void*)k_rollout_w<NTV, ((P) <= 2 ? (P) : 0), M, true, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_i); \
        hipLaunchKernelGGL((k_rollout_w<NTV, ((P) <= 2 ? (P) : 0), M, true, true, true>), dim3(io->Bp), blk_s, shm_i, st, *w, *io, 0, 0); \
      } else {                                                                                               \
      (void)hipFuncSetAttribute((const void*)k_rollout_w<NTV, P, M, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_i); \
      hipLaunchKernelGGL((k_rollout_w<NTV, P, M, true, true>), dim3(io->Bp), blk_s, shm_i, st, *w, *io, 0, 0); \
      }                                                                                                      \
    }                                                                                                        \
  } while (0)
  // the split-bf16 pointer MLP is opt-in (RR_MLP_SPLIT=1), for the greedy / sampling rollouts of every problem
#define RR_LAUNCHW3(NTV, P, M)                                                                               \
  do {                                                                                                       \
    if ((M) <= 1 && mlp_split && (NTV) == 7 && inst_on && tail_g == 0 && (S + 15) / 16 == WWAVES - 1) RR_LAUNCHW4I(NTV, P, ((M) <= 1 ? (M) : 0)); \
    else if ((M) <= 1 && mlp_split) RR_LAUNCHW4(NTV, P, ((M) <= 1 ? (M) : 0), true);                          \
    else RR_LAUNCHW4(NTV, P, M, false);                                                                      \
  } while (0)
#define RR_LAUNCHW2(NTV, P)                                                                                  \
  do {                                                                                                       \
    if (mode == 0) RR_LAUNCHW3(NTV, P, 0);                                                                   \
    else if (mode == 1) RR_LAUNCHW3(NTV, P, 1);                                                              \
    else if (mode == 2) RR_LAUNCHW3(NTV, P, 2);                                                              \
    else RR_LAUNCHW3(NTV, P, 3);                                                                             \
  } while (0)
#define RR_LAUNCHW(NTV)                                                                      \
  do {                                                                                       \
    if (prob == 0) RR_LAUNCHW2(NTV, 0);                                                      \
    else if (prob == 1) RR_LAUNCHW2(NTV, 1);                                                 \
    else if (io->used_b == nullptr) RR_LAUNCHW2(NTV, 2);                                     \
    else RR_LAUNCHW2(NTV, 3);                                                                \
  } while (0)
#ifdef RR_DEV_HEADLINE_ONLY   // diagnostic builds: only the headline instantiation (ATSP greedy, instance mode) is compiled
  if (!(prob == 0 && mode == 0 && mlp_split && N > 64 && tail_g == 0 && (S + 15) / 16 == WWAVES - 1)) return RR_EINVAL;
  RR_LAUNCHW4I(7, 0, 0);
#else
  if (N <= 32) RR_LAUNCHW(2);
  else if (N <= 64) RR_LAUNCHW(4);
  else RR_LAUNCHW(7);
#endif
#undef RR_LAUNCHW2
#undef RR_LAUNCHW3
#undef RR_LAUNCHW4
#undef RR_LAUNCHW
  return rr_check(hipGetLastError());
}

#ifdef RR_STAMP
extern "C" int rr_debug_wave_cycles(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rr_wave_cycles), sizeof(unsigned long long) * 8) != hipSuccess) return RR_ELAUNCH;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rr_wave_cycles), z, sizeof(z)) != hipSuccess) return RR_ELAUNCH; }
  return RR_OK;
}
extern "C" int rr_debug_stamps(unsigned long long* out, int reset) {
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(rr_stamps), sizeof(unsigned long long) * 8) != hipSuccess) return RR_ELAUNCH;
  if (reset) { unsigned long long z[8] = {0}; if (hipMemcpyToSymbol(HIP_SYMBOL(rr_stamps), z, sizeof(z)) != hipSuccess) return RR_ELAUNCH; }
  return RR_OK;
}
#endif
