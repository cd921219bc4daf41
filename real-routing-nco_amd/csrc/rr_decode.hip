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
  // g0 and g hold Bp*dumpT*S + 1 rows: lanes without a live rollout store into the last one (a trash row nobody reads).
  float* dump_g0; float* dump_g; uint32_t* dump_meta; float* dump_scal;
  int dumpT;
  int use_split;                         // 1: this launch runs on the fp16 matrix pipe with two-piece split operands (rr_common.h), if DecW has w1s / w2s and Ks / Vts / Ls are given
  const void *Ks, *Vts, *Ls;             // fp16 two-piece images of K / Vt / L (rr_pack_f16x2), same shapes and byte offsets
  int* status;                           // optional: bit 2 <- a split launch met a non-finite log-probability (an operand left the fp16 range)
  int top_k; float top_p;                // process_logits' filters (decoding.py:352-358) inside the rollout: split greedy / sampling launches only (0 / 0.0: off)
  int tail_pack;                         // 1: pack the S % 16 left-over rollouts of 16 / (S % 16) instances into one tile (off by default: see rr_rollout)
  int no_inst;                           // 1: never launch the instance-mode kernel (A/B measurements)
};

// (The first-generation workgroup-per-instance rollout kernel — activations through LDS, ten barriers per decode step; rounds 1-3 kept
// it behind RR_ROLLOUT_VARIANT=0 for A/B timing — left the library in round 4: nothing exercised it any more.)

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
  // RolloutIO.use_split decides (the launcher reads no environment variable: the host side passes its switches as fields)
  const bool mlp_split = io->use_split != 0 && w->w1s != nullptr && w->w2s != nullptr && w->b1s != nullptr &&
                         io->Ks != nullptr && io->Vts != nullptr && io->Ls != nullptr;
  // tail packing (rr_rollout_w.inc): the S % 16 left-over rollouts of 16 / (S % 16) consecutive instances share one tile
  // Off by default since the logit keys of ordinary tiles live in LDS: a packed tile runs its attention and logits once per
  // instance it spans, with every operand from L2, and a workgroup of such tiles takes 3 (a 2-round launch) to ~6 (the 14-round
  // headline launch) ordinary workgroup times for the work of 4 quarter-full tiles.  Measured: headline rollout 73.9 ms packed,
  // 69.2 ms not (14 full rounds of 256 workgroups); 512-instance training rollout 18.4 / 12.2 ms.  RolloutIO.tail_pack = 1 turns it on
  // (rrnco_amd.models.rollout reads RR_TAIL_PACK once at import).
  const int pack = io->tail_pack;
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
  // top-k / top-p inside the rollout: FILT builds of the split greedy / sampling kernels (two-piece operands); anything else is the caller's
  // per-step loop (rr_select carries the same filters)
  if (io->top_k < 0 || io->top_p < 0.f || io->top_p > 1.f) return RR_EINVAL;
  const bool filt = !io->logits_only && ((io->top_k > 0 && io->top_k < N) || (io->top_p > 0.f && io->top_p < 1.f));
  if (filt && !(mlp_split && mode <= 1 && io->use_split != 2)) return RR_EINVAL;
  const bool inst_on = io->no_inst == 0;
#define RR_LAUNCHW4(NTV, P, M, SP)                                                                            \
  do {                                                                                                       \
    const int lds_inst_s = wstg + 2 * (size_t)(NTV) * 8192 <= per_wg ? 2 : 0;                               \
    const size_t shm = (SP) ? wstg + (size_t)lds_inst_s * (NTV) * 8192 : shmem;              \
    if constexpr (SP) {                                                                                      \
      if (filt) {                                                                                            \
        (void)hipFuncSetAttribute((const void*)k_rollout_w<NTV, P, M, SP, false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); \
        hipLaunchKernelGGL((k_rollout_w<NTV, P, M, SP, false, false, true>), grid_s, blk_s, shm, st, *w, *io, tail_g, lds_inst_s); \
        break;                                                                                               \
      }                                                                                                      \
    }                                                                                                        \
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
      } else if (filt) {                                                                                     \
      (void)hipFuncSetAttribute((const void*)k_rollout_w<NTV, P, M, true, true, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_i); \
      hipLaunchKernelGGL((k_rollout_w<NTV, P, M, true, true, false, true>), dim3(io->Bp), blk_s, shm_i, st, *w, *io, 0, 0); \
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

// The rollout's in-register top-k / top-p filters (rr_rollout_w.inc: rr_filter_top_k / rr_filter_top_p) on rows of processed logits
// (masked keys -inf, already / temperature): out = the rows with the removed keys at -inf (decoding.py:37-63, 352-358, top-k first).
// Same device functions, same lane layout as the rollout's selection stage (a wave = 16 rows, the four lanes that share j hold a
// row): what tests/test_gpu_filters_fused.py pins against process_logits key by key, ties included.  N <= 112.
__global__ __launch_bounds__(256) void k_filter_rows(const float* __restrict__ in, float* __restrict__ out, int R, int N, int top_k, float top_p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 15, gs = lane >> 4;
  const long long r = ((long long)blockIdx.x * 4 + wave) * 16 + j;
  f32x4 la[7];
  float mx = -INFINITY;
#pragma unroll
  for (int kt = 0; kt < 7; ++kt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int key = 16 * kt + 4 * gs + q;
      const float v = (r < R && key < N) ? in[r * N + key] : -INFINITY;
      la[kt][q] = v; mx = fmaxf(mx, v);
    }
  mx = rr_max_g(mx);
  if (mx == -INFINITY) mx = 0.f;
  if (top_k > 0 && top_k < N) rr_filter_top_k<7>(la, top_k);
  if (top_p > 0.f && top_p < 1.f) rr_filter_top_p<7>(la, mx, gs, top_p);
#pragma unroll
  for (int kt = 0; kt < 7; ++kt)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int key = 16 * kt + 4 * gs + q;
      if (r < R && key < N) out[r * N + key] = la[kt][q];
    }
}
extern "C" int rr_filter_rows(const float* logits, float* out, int R, int N, int top_k, float top_p, hipStream_t st) {
  if (logits == nullptr || out == nullptr || R <= 0 || N < 1 || N > 112 || top_k < 0 || top_p < 0.f || top_p > 1.f) return RR_EINVAL;
  hipLaunchKernelGGL(k_filter_rows, dim3((R + 63) / 64), dim3(256), 0, st, logits, out, R, N, top_k, top_p);
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
