// Environment-side kernels of the RRNCO rollout: reset normalisation, masked env.step for
// ATSP / RCVRP / RCVRPTW, tour cost (reward) and the logits -> action selection step.
// All integer / boolean outputs are bit-exact restatements of the reference; see include/rrnco_hip.h
// for the reference file:line each entry point replaces.
#include "rr_common.h"

// ------------------------------------------------------------------------------------------------
// min-max normalisation of a [B, M] batch of matrices (rrnco/envs/atsp/env.py:113-120)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_minmax_normalize(const float* __restrict__ in, float* __restrict__ out,
                                                          float* __restrict__ mn_out, float* __restrict__ mx_out, int M) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const float* src = in + (size_t)b * M;
  float* dst = out + (size_t)b * M;
  float mn = INFINITY, mx = -INFINITY;
  for (int i = tid; i < M; i += 256) { float v = src[i]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
  mn = rr_wave_min(mn); mx = rr_wave_max(mx);
  __shared__ float smn[4], smx[4];
  if ((tid & 63) == 0) { smn[tid >> 6] = mn; smx[tid >> 6] = mx; }
  __syncthreads();
  mn = fminf(fminf(smn[0], smn[1]), fminf(smn[2], smn[3]));
  mx = fmaxf(fmaxf(smx[0], smx[1]), fmaxf(smx[2], smx[3]));
  const float den = (mx - mn) + 1e-6f;
  for (int i = tid; i < M; i += 256) dst[i] = (src[i] - mn) / den;
  if (tid == 0) { mn_out[b] = mn; mx_out[b] = mx; }
}

extern "C" int rr_minmax_normalize(const float* in, float* out, float* mn, float* mx, int B, int M, hipStream_t st) {
  if (B <= 0 || M <= 0 || in == nullptr || out == nullptr || mn == nullptr || mx == nullptr) return RR_EINVAL;
  hipLaunchKernelGGL(k_minmax_normalize, dim3(B), dim3(256), 0, st, in, out, mn, mx, M);
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// ATSP step (rrnco/envs/atsp/env.py:80-105): available[r, action[r]] = 0; done = no node left
// one wave per rollout row; mask is uint8 (torch.bool storage)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_atsp_step(const int64_t* __restrict__ action, const uint8_t* __restrict__ mask_in,
                                                   uint8_t* __restrict__ mask_out, uint8_t* __restrict__ done, int R, int N) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int a = (int)action[r];
  int cnt = 0;
  for (int j = lane; j < N; j += 64) {
    uint8_t m = mask_in[(size_t)r * N + j];
    if (j == a) m = 0;
    mask_out[(size_t)r * N + j] = m;
    cnt += (m != 0);
  }
  cnt = (int)rr_wave_sum((float)cnt);
  if (lane == 0) done[r] = (cnt <= 0);
}

// First vectorised form (kept for A/B, RR_STEP_VARIANT=0), N % 2 == 0 and 8 N <= 1024: a wave owns 8 consecutive rows = 8 N contiguous bytes (a multiple of
// 16), each lane moves one 16-byte chunk.  The byte-per-lane kernel above keeps 64 B per load instruction in flight and
// reaches 13 % of the HBM roofline; this one issues 8N B per wave with one load and one store per lane.
__global__ __launch_bounds__(256) void k_atsp_step_v0(const int64_t* __restrict__ action, const uint8_t* __restrict__ mask_in,
                                                     uint8_t* __restrict__ mask_out, uint8_t* __restrict__ done, int R, int N) {
  const int lane = threadIdx.x & 63;
  const long r0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 8;
  if (r0 >= R) return;
  const int rows = (int)(R - r0 < 8 ? R - r0 : 8);
  const int nbytes = rows * N, base = lane * 16;
  const int myact = lane < rows ? (int)action[r0 + lane] : -1;
  uint4 v = make_uint4(0, 0, 0, 0);
  const bool active = base < nbytes;
  const bool full = base + 16 <= nbytes;
  const uint8_t* src = mask_in + r0 * N + base;
  if (full) v = *reinterpret_cast<const uint4*>(src);
  else if (active) { uint8_t* pv = reinterpret_cast<uint8_t*>(&v); for (int q = 0; q < nbytes - base; ++q) pv[q] = src[q]; }
  // the chunk [base, base+16) touches rows ra = base / N and (possibly) ra + 1
  const int ra = base / N, split = (ra + 1) * N - base;          // bytes [0, split) belong to row ra
  const int act_a = __shfl(myact, ra & 63), act_b = __shfl(myact, (ra + 1) & 63);
  const int clr_a = act_a - (base - ra * N);                      // byte position of row ra's action inside this chunk
  const int clr_b = act_b + split;
  uint32_t w[4] = {v.x, v.y, v.z, v.w};
  int cnt_a = 0, cnt_b = 0;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const bool in_a = q < split;
    const bool clear = in_a ? (q == clr_a) : (q == clr_b);
    uint32_t byte = (w[q >> 2] >> (8 * (q & 3))) & 0xffu;
    if (clear) { byte = 0; w[q >> 2] &= ~(0xffu << (8 * (q & 3))); }
    const bool on = byte != 0 && base + q < nbytes;
    cnt_a += (on && in_a); cnt_b += (on && !in_a);
  }
  uint8_t* dst = mask_out + r0 * N + base;
  if (full) *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
  else if (active) { const uint8_t* pw = reinterpret_cast<const uint8_t*>(w); for (int q = 0; q < nbytes - base; ++q) dst[q] = pw[q]; }
  // per-row counts: every lane contributes to rows ra and ra + 1
#pragma unroll
  for (int rr = 0; rr < 8; ++rr) {
    const int c = (active && ra == rr ? cnt_a : 0) + (active && ra + 1 == rr ? cnt_b : 0);
    const int tot = (int)rr_wave_sum((float)c);
    if (lane == 0 && rr < rows) done[r0 + rr] = (tot <= 0);
  }
}

// Vectorised form for N % 2 == 0, 16 <= N <= 128: a wave owns 8 consecutive rows = 8 N contiguous bytes (a multiple of
// 16), each lane moves one 16-byte chunk.  The
// byte-per-lane kernel above keeps 64 B per load instruction in flight and reaches 13 % of the HBM roofline.  The action
// byte of a row is cleared word-wise, `done` comes from one ballot per row (no per-byte loop, no shuffle reductions).
template <int STEPV_CHUNKS>          // 16-byte chunks per lane, 8 rows per chunk
__global__ __launch_bounds__(256) void k_atsp_step_v(const int64_t* __restrict__ action, const uint8_t* __restrict__ mask_in,
                                                     uint8_t* __restrict__ mask_out, uint8_t* __restrict__ done, int R, int N) {
  const int lane = threadIdx.x & 63;
  constexpr int STEPV_ROWS = 8 * STEPV_CHUNKS;
  const long r0 = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * STEPV_ROWS;
  if (r0 >= R) return;
  const int rows = (int)(R - r0 < STEPV_ROWS ? R - r0 : STEPV_ROWS);
  const int nbytes = rows * N;
  // both loads unconditional on always-valid addresses: guarded loads made hipcc wait for the action before issuing the
  // mask load (two memory latencies per wave)
  uint4 v[STEPV_CHUNKS];
#pragma unroll
  for (int h = 0; h < STEPV_CHUNKS; ++h) {
    const int base = (lane + 64 * h) * 16;
    v[h] = *reinterpret_cast<const uint4*>(mask_in + r0 * N + (base + 16 <= nbytes ? base : 0));
  }
  int myact = (int)action[r0 + (lane < rows ? lane : 0)];
  myact = lane < rows ? myact : -1;
#pragma unroll
  for (int h = 0; h < STEPV_CHUNKS; ++h) {
    const int base = (lane + 64 * h) * 16;
    if (base + 16 > nbytes) {                                     // last wave only: a partial or absent chunk
      v[h] = make_uint4(0, 0, 0, 0);
      if (base < nbytes) {
        const uint8_t* src = mask_in + r0 * N + base;
        uint32_t t[4] = {0, 0, 0, 0};
        for (int q = 0; q < nbytes - base; ++q) t[q >> 2] |= (uint32_t)src[q] << (8 * (q & 3));
        v[h] = make_uint4(t[0], t[1], t[2], t[3]);
      }
    }
  }
  uint32_t left = 0;
#pragma unroll
  for (int h = 0; h < STEPV_CHUNKS; ++h) {
    const int base = (lane + 64 * h) * 16;
    const bool active = base < nbytes;
    // the chunk [base, base+16) touches rows ra = base / N and (possibly) ra + 1
    const int ra = base / N, split = (ra + 1) * N - base;          // bytes [0, split) belong to row ra
    const int act_a = __shfl(myact, ra & 63), act_b = __shfl(myact, (ra + 1) & 63);
    const int clr_a = act_a - (base - ra * N);                      // byte position of row ra's action inside this chunk
    const int clr_b = act_b + split;
    uint32_t w[4] = {v[h].x, v[h].y, v[h].z, v[h].w};
    const int valid = min(max(nbytes - base, 0), 16);             // bytes of this chunk that exist
    const int sa = min(split, valid);                             // bytes [0, sa) belong to row ra, [sa, valid) to row ra + 1
    const int ca = (clr_a >= 0 && clr_a < sa) ? clr_a : -1;
    const int cb = (clr_b >= sa && clr_b < valid) ? clr_b : -1;
    uint32_t any_a = 0, any_b = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      uint32_t keepm = 0xffffffffu;
      if ((ca >> 2) == k) keepm &= ~(0xffu << (8 * (ca & 3)));
      if ((cb >> 2) == k) keepm &= ~(0xffu << (8 * (cb & 3)));
      w[k] &= keepm;
      // byte-range masks of word k: bytes with chunk index < n
      const int ta = min(max(sa - 4 * k, 0), 4), tv = min(max(valid - 4 * k, 0), 4);
      const uint32_t ma = ta == 4 ? 0xffffffffu : ((1u << (8 * ta)) - 1u), mv = tv == 4 ? 0xffffffffu : ((1u << (8 * tv)) - 1u);
      any_a |= w[k] & ma; any_b |= w[k] & mv & ~ma;
    }
    uint8_t* dst = mask_out + r0 * N + base;
    if (base + 16 <= nbytes) *reinterpret_cast<uint4*>(dst) = make_uint4(w[0], w[1], w[2], w[3]);
    else if (active) { const uint8_t* pw = reinterpret_cast<const uint8_t*>(w); for (int q = 0; q < nbytes - base; ++q) dst[q] = pw[q]; }
    // rows with a byte left set: one ballot per row
#pragma unroll
    for (int rr = 0; rr < STEPV_ROWS; ++rr) {
      const bool mine = active && ((ra == rr && any_a != 0) || (ra + 1 == rr && any_b != 0));
      left |= (__ballot(mine) != 0ull ? 1u : 0u) << rr;
    }
  }
  if (lane < rows) done[r0 + lane] = ((left >> lane) & 1u) == 0;
}

// Third form, N % 4 == 0 and 16 <= N <= 128: eight lanes per row, lane s of a row moves the 16-byte window starting at
// byte min(16 s, N - 16) with one dword-aligned access — the last window is pulled back to end at the row's end, so it
// overlaps its neighbour and both write the same updated bytes.  No window straddles two rows: no row arithmetic, no
// shuffle, no per-byte work, no divergent load (hipcc waits for all outstanding loads at a divergent one): ~50
// instructions per wave against ~450 in k_atsp_step_v, which was VALU-bound at a third of the plain-copy rate.  `done` is
// one ballot per pass.  STEPR_PASSES x 8 rows per wave, every load issued before the first use.
typedef uint32_t rr_u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
#define STEPR_PASSES 2
__global__ __launch_bounds__(256) void k_atsp_step_r(const int64_t* __restrict__ action, const uint8_t* __restrict__ mask_in,
                                                     uint8_t* __restrict__ mask_out, uint8_t* __restrict__ done, int R, int N) {
  const int lane = threadIdx.x & 63, s = lane & 7, rl = lane >> 3;
  const long rbase = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (8 * STEPR_PASSES);
  const int ws = min(16 * s, N - 16);
  const bool col_ok = 16 * s < N;
  rr_u32x4_a4 w[STEPR_PASSES];
  int act[STEPR_PASSES];
#pragma unroll
  for (int i = 0; i < STEPR_PASSES; ++i) {
    const long r = rbase + 8 * i + rl;
    const long rc = r < R ? r : 0;                       // rows past the end read row 0 and store nothing
    w[i] = *reinterpret_cast<const rr_u32x4_a4*>(mask_in + rc * N + ws);
    act[i] = (int)action[rc];
  }
#pragma unroll
  for (int i = 0; i < STEPR_PASSES; ++i) {
    const long r = rbase + 8 * i + rl;
    const bool ok = r < R && col_ok;
    const int c = act[i] - ws;                           // the action's byte inside this lane's window, if 0 <= c < 16
    const uint32_t clr = ~(0xffu << (8 * (c & 3)));
    w[i].x &= (c >> 2) == 0 ? clr : 0xffffffffu; w[i].y &= (c >> 2) == 1 ? clr : 0xffffffffu;
    w[i].z &= (c >> 2) == 2 ? clr : 0xffffffffu; w[i].w &= (c >> 2) == 3 ? clr : 0xffffffffu;
    if (ok) *reinterpret_cast<rr_u32x4_a4*>(mask_out + r * N + ws) = w[i];
    const unsigned long long left = __ballot(ok && (w[i].x | w[i].y | w[i].z | w[i].w) != 0);
    if (s == 0 && r < R) done[r] = ((left >> (8 * rl)) & 0xffull) == 0;
  }
}

extern "C" int rr_atsp_step(const int64_t* action, const uint8_t* mask_in, uint8_t* mask_out, uint8_t* done,
                            int R, int N, hipStream_t st) {
  if (R <= 0 || N <= 0 || action == nullptr || mask_in == nullptr || mask_out == nullptr || done == nullptr) return RR_EINVAL;
  const bool aligned = ((reinterpret_cast<uintptr_t>(mask_in) | reinterpret_cast<uintptr_t>(mask_out)) & 15) == 0;
  static const int variant = [] { const char* e = getenv("RR_STEP_VARIANT"); return e ? atoi(e) : 1; }();
  if (variant == 1 && N % 4 == 0 && N >= 16 && N <= 128 && ((reinterpret_cast<uintptr_t>(mask_in) | reinterpret_cast<uintptr_t>(mask_out)) & 3) == 0)
    hipLaunchKernelGGL(k_atsp_step_r, dim3((R + 32 * STEPR_PASSES - 1) / (32 * STEPR_PASSES)), dim3(256), 0, st, action, mask_in, mask_out, done, R, N);
  else if (N % 2 == 0 && N <= 128 && N >= 16 && aligned) {          // N >= 16: a 16-byte chunk spans at most two rows
    if (variant == 0) hipLaunchKernelGGL(k_atsp_step_v0, dim3((R + 31) / 32), dim3(256), 0, st, action, mask_in, mask_out, done, R, N);
    else if (variant == 3) hipLaunchKernelGGL(k_atsp_step_v<2>, dim3((R + 63) / 64), dim3(256), 0, st, action, mask_in, mask_out, done, R, N);
    else hipLaunchKernelGGL(k_atsp_step_v<1>, dim3((R + 31) / 32), dim3(256), 0, st, action, mask_in, mask_out, done, R, N);
  }
  else
    hipLaunchKernelGGL(k_atsp_step, dim3((R + 3) / 4), dim3(256), 0, st, action, mask_in, mask_out, done, R, N);
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// RCVRP step + action mask (rrnco/envs/rcvrp/env.py:90-122, 183-195).  One wave per rollout.
//   demand [Bp, N] (customers, shared by the S starts: rollout r uses instance r % Bp)
//   used_capacity / vehicle_capacity [R] f32, visited [R, N+1] u8, action_mask [R, N+1] u8
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rcvrp_step(const int64_t* __restrict__ action, const float* __restrict__ demand,
                                                    const float* __restrict__ vcap, float* __restrict__ used,
                                                    uint8_t* __restrict__ visited, uint8_t* __restrict__ mask,
                                                    int64_t* __restrict__ cur_out, uint8_t* __restrict__ done,
                                                    int R, int Bp, int N /*customers*/) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int b = r % Bp;
  const int a = (int)action[r];
  const float* dem = demand + (size_t)b * N;
  int ci = a - 1; ci = ci < 0 ? 0 : (ci > N - 1 ? N - 1 : ci);
  const float sel = dem[ci];
  const float u = (used[r] + sel) * (a != 0 ? 1.0f : 0.0f);
  const float cap = vcap[r];
  uint8_t* vis = visited + (size_t)r * (N + 1);
  uint8_t* mk = mask + (size_t)r * (N + 1);
  int nvis = 0, nfree = 0;
  for (int j = lane; j <= N; j += 64) {
    uint8_t v = vis[j];
    if (j == a) v = 1;
    vis[j] = v;
    nvis += (v != 0);
    if (j >= 1) {
      bool exceeds = dem[j - 1] + u > cap;
      bool mloc = (v != 0) || exceeds;
      mk[j] = !mloc;
      nfree += !mloc;
    }
  }
  nvis = (int)rr_wave_sum((float)nvis);
  nfree = (int)rr_wave_sum((float)nfree);
  if (lane == 0) {
    bool mask_depot = (a == 0) && (nfree > 0);
    mk[0] = !mask_depot;
    used[r] = u;
    cur_out[r] = a;
    done[r] = (nvis == N + 1);
  }
}

// Second form (4 <= N + 1 <= 128): 32 lanes per rollout, lane l owns the four nodes of the window starting at
// min(4 l, N + 1 - 4) (the last window is pulled back and overlaps its neighbour, which computes and stores the same bytes):
// 16-byte demand loads, one 4-byte visited load and mask store per lane, `done` / "any free customer" by ballot, `visited`
// updated by one byte store.  Same fp32 expressions as k_rcvrp_step: bit-identical masks.  RCV_PASSES x 2 rollouts per wave.
#define RCV_PASSES 2
__global__ __launch_bounds__(256) void k_rcvrp_step_v(const int64_t* __restrict__ action, const float* __restrict__ demand,
                                                      const float* __restrict__ vcap, float* __restrict__ used,
                                                      uint8_t* __restrict__ visited, uint8_t* __restrict__ mask,
                                                      int64_t* __restrict__ cur_out, uint8_t* __restrict__ done,
                                                      int R, int Bp, int N /*customers*/) {
  typedef uint32_t u32_a1 __attribute__((aligned(1)));
  const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
  const long rbase = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (2 * RCV_PASSES);
  const int N1 = N + 1;
  const int ws = min(4 * hl, N1 - 4);
  const bool col_ok = 4 * hl < N1;
  int a[RCV_PASSES];
  float u0[RCV_PASSES], cap[RCV_PASSES], sel[RCV_PASSES], dem[RCV_PASSES][4];
  uint32_t visw[RCV_PASSES];
#pragma unroll
  for (int i = 0; i < RCV_PASSES; ++i) {
    const long r = rbase + 2 * i + half;
    const long rc = r < R ? r : R - 1;
    a[i] = (int)action[rc]; u0[i] = used[rc]; cap[i] = vcap[rc];
    visw[i] = *reinterpret_cast<const u32_a1*>(visited + rc * N1 + ws);
    const float* db = demand + (size_t)(rc % Bp) * N;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int k = ws + q; dem[i][q] = db[k >= 1 ? k - 1 : 0]; }   // node k <-> demand[k-1]
    int ci = a[i] - 1; ci = ci < 0 ? 0 : (ci > N - 1 ? N - 1 : ci);
    sel[i] = db[ci];
  }
#pragma unroll
  for (int i = 0; i < RCV_PASSES; ++i) {
    const long r = rbase + 2 * i + half;
    const bool row_ok = r < R;
    const int av = a[i];
    const float u = (u0[i] + sel[i]) * (av != 0 ? 1.0f : 0.0f);                        // rcvrp/env.py:96-101
    bool unvis = false, free_here = false;
    bool ml[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int k = ws + q;
      const bool v = ((visw[i] >> (8 * q)) & 0xffu) != 0 || k == av;
      unvis |= !v;
      const bool exceeds = dem[i][q] + u > cap[i];                                     // :185-187
      ml[q] = v || exceeds;
      free_here |= k >= 1 && !ml[q];
    }
    const int sh = 32 * half;
    const bool anyfree = (uint32_t)(__ballot(col_ok && free_here) >> sh) != 0;
    const bool all_visited = (uint32_t)(__ballot(col_ok && unvis) >> sh) == 0;
    uint32_t mw = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool bit = ws + q == 0 ? !(av == 0 && anyfree) : !ml[q];                   // :189-195
      mw |= (bit ? 1u : 0u) << (8 * q);
    }
    if (row_ok && col_ok) *reinterpret_cast<u32_a1*>(mask + r * N1 + ws) = mw;
    if (row_ok && hl == 0) {
      visited[r * N1 + av] = 1;
      used[r] = u; cur_out[r] = av; done[r] = all_visited;
    }
  }
}

extern "C" int rr_rcvrp_step(const int64_t* action, const float* demand, const float* vcap, float* used,
                             uint8_t* visited, uint8_t* mask, int64_t* cur_out, uint8_t* done,
                             int R, int Bp, int N, hipStream_t st) {
  if (R <= 0 || N <= 0 || Bp <= 0 || !action || !demand || !vcap || !used || !visited || !mask || !cur_out || !done) return RR_EINVAL;
  static const int variant = [] { const char* e = getenv("RR_STEP_VARIANT"); return e ? atoi(e) : 1; }();
  if (variant != 0 && N + 1 >= 4 && N + 1 <= 128)
    hipLaunchKernelGGL(k_rcvrp_step_v, dim3((R + 8 * RCV_PASSES - 1) / (8 * RCV_PASSES)), dim3(256), 0, st, action, demand, vcap, used,
                       visited, mask, cur_out, done, R, Bp, N);
  else
    hipLaunchKernelGGL(k_rcvrp_step, dim3((R + 3) / 4), dim3(256), 0, st, action, demand, vcap, used, visited, mask,
                       cur_out, done, R, Bp, N);
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// Tour cost (reward): mode 0 closed tour (atsp/env.py:192-211), mode 1 depot-prefixed open list that
// returns to the depot (rcvrp/env.py:197-219, rmtvrp/env.py:430-455).  One wave per rollout.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_tour_cost(const float* __restrict__ D, const int64_t* __restrict__ actions,
                                                   const float* __restrict__ mn, const float* __restrict__ mx,
                                                   float* __restrict__ norm_out, float* __restrict__ real_out,
                                                   int R, int Bp, int N, int T, int mode, const uint8_t* __restrict__ open_route) {
  const int lane = threadIdx.x & 63;
  // Rollout r = s * Bp + b belongs to instance b.  The S rollouts of an instance read the same 4 N^2 bytes of D: they are kept
  // together AND on one XCD (workgroup i runs on XCD i % 8), so that an instance's matrix is fetched into one L2 once instead
  // of being re-fetched from the Infinity Cache by rollouts 4 096 instances apart.
  int r;
  if (R % Bp == 0) {
    const int S = R / Bp, G = (S + 3) >> 2;                  // G workgroups of 4 waves per instance
    const int x = blockIdx.x & 7, y = blockIdx.x >> 3;
    const int bb = (y / G) * 8 + x, sg = y % G;
    const int s = sg * 4 + (threadIdx.x >> 6);
    if (bb >= Bp || s >= S) return;
    r = s * Bp + bb;
  } else {
    r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= R) return;
  }
  const int b = r % Bp;
  const float* Db = D + (size_t)b * N * N;
  const int64_t* act = actions + (size_t)r * T;
  float s = 0.f;
  if (mode == 0) {
    for (int t = lane; t < T; t += 64) {
      int u = (int)act[t], v = (int)act[t + 1 < T ? t + 1 : 0];
      s += Db[u * N + v];
    }
  } else {
    // edges: (0 -> a_0), (a_t -> a_{t+1}), (a_{T-1} -> 0)
    // rmtvrp/env.py:430-434: on an open route every arc INTO the depot is free (column 0 of the cost matrix is zeroed)
    const bool open = open_route != nullptr && open_route[b] != 0;
    for (int t = lane; t <= T; t += 64) {
      int u = t == 0 ? 0 : (int)act[t - 1];
      int v = t == T ? 0 : (int)act[t];
      s += (open && v == 0) ? 0.f : Db[u * N + v];
    }
  }
  s = rr_wave_sum(s);
  if (lane == 0) {
    float nd = -s;
    norm_out[r] = nd;
    if (mn != nullptr) real_out[r] = nd * ((mx[b] - mn[b]) + 1e-6f) + mn[b];
    else real_out[r] = nd;
  }
}

extern "C" int rr_tour_cost(const float* D, const int64_t* actions, const float* mn, const float* mx,
                            float* norm_out, float* real_out, int R, int Bp, int N, int T, int mode,
                            const uint8_t* open_route, hipStream_t st) {
  if (R <= 0 || N <= 0 || T <= 0 || Bp <= 0 || !D || !actions || !norm_out || !real_out || ((mn == nullptr) != (mx == nullptr))) return RR_EINVAL;
  const unsigned grid = R % Bp == 0 ? (unsigned)(((Bp + 7) / 8) * 8) * (unsigned)((R / Bp + 3) / 4) : (unsigned)((R + 3) / 4);
  hipLaunchKernelGGL(k_tour_cost, dim3(grid), dim3(256), 0, st, D, actions, mn, mx, norm_out, real_out, R, Bp, N, T, mode,
                     open_route);
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// Real-world instance sampling (rrnco/envs/*/sampler.py:78-94): out[b][i][j] = city[idx[b][i]][idx[b][j]] for the
// distance and (optionally) the duration matrix of one city in a single pass.  One workgroup per (instance, row):
// the row's source address is wave-uniform, the column indices are read once, the output row is written coalesced.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_submatrix_gather(const float* __restrict__ dist, const float* __restrict__ dur,
                                                          const int64_t* __restrict__ idx, float* __restrict__ out_dist,
                                                          float* __restrict__ out_dur, int M, int n) {
  const int b = blockIdx.y, i = blockIdx.x;
  const int64_t* ib = idx + (size_t)b * n;
  const size_t src = (size_t)ib[i] * M;
  const size_t dst = ((size_t)b * n + i) * n;
  for (int j = threadIdx.x; j < n; j += 128) {
    const size_t c = (size_t)ib[j];
    out_dist[dst + j] = dist[src + c];
    if (dur != nullptr) out_dur[dst + j] = dur[src + c];
  }
}

extern "C" int rr_submatrix_gather(const float* dist, const float* dur, const int64_t* idx, float* out_dist, float* out_dur,
                                   int B, int M, int n, hipStream_t st) {
  if (B <= 0 || M <= 0 || n <= 0 || n > M || dist == nullptr || idx == nullptr || out_dist == nullptr) return RR_EINVAL;
  if ((dur == nullptr) != (out_dur == nullptr)) return RR_EINVAL;
  hipLaunchKernelGGL(k_submatrix_gather, dim3(n, B), dim3(128), 0, st, dist, dur, idx, out_dist, out_dur, M, n);
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// logits -> (action, logp): process_logits + greedy / sampling / evaluate
// (rrnco/models/decoding.py:311-361, 272-298, 266).  One wave per row, N <= 128.
// mode: 0 greedy (first index on ties), 1 sampling (inverse CDF over ascending keys on a counter-based uniform: rr_common.h),
//       2 evaluate (action given).  logp_all (optional) receives the full log-softmax row.
// ------------------------------------------------------------------------------------------------
#define SEL_ROWS 4   // rows per wave: all their loads are issued before the first row is reduced (bytes in flight)
__device__ __forceinline__ void rr_select_row(const float (&raw)[2], const uint8_t (&keep)[2], int r, int lane,
                                              const int64_t* __restrict__ action_in, int64_t* __restrict__ action_out,
                                              float* __restrict__ logp_out, float* __restrict__ logp_all,
                                              int N, float tanh_clip, float temperature, int mode,
                                              uint64_t seed, uint32_t step, int top_k, float top_p) {
  float x[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    int j = lane + 64 * q;
    float v = -INFINITY;
    if (j < N) {
      v = raw[q];
      if (tanh_clip > 0.f) v = rr_tanh(v) * tanh_clip;
      if (keep[q] == 0) v = -INFINITY;
      v = v * (1.0f / temperature);
    }
    x[q] = v;
  }
  if (top_k > 0 && top_k < N) {
    // modify_logits_for_top_k_filtering (decoding.py:37-42): drop everything below the k-th largest value, counted with
    // multiplicity as torch.topk does.  Peel off the current maximum and its copies until k values are covered.
    float bound = INFINITY, thr = -INFINITY;
    int covered = 0;
    for (int it = 0; it < top_k; ++it) {
      const float c0 = x[0] < bound ? x[0] : -INFINITY, c1 = x[1] < bound ? x[1] : -INFINITY;
      const float cur = rr_wave_max(fmaxf(c0, c1));
      covered += __popcll(__ballot(lane < N && x[0] == cur)) + __popcll(__ballot(lane + 64 < N && x[1] == cur));
      thr = cur; bound = cur;
      if (covered >= top_k || cur == -INFINITY) break;
    }
    if (x[0] < thr) x[0] = -INFINITY;
    if (x[1] < thr) x[1] = -INFINITY;
  }
  if (top_p > 0.f && top_p < 1.f) {
    // modify_logits_for_top_p_filtering (decoding.py:45-63): remove the lower tail whose cumulative probability, summed
    // from the smallest logit upwards, is <= 1 - top_p.  Without sorting: cum_i = sum of p_j over the elements that a
    // stable ascending sort places at or before i (x_j < x_i, or x_j == x_i and j <= i; ties are common once tanh saturates).
    const float mm = rr_wave_max(fmaxf(x[0], x[1]));
    const float p0 = (lane < N) ? rr_exp(x[0] - mm) : 0.f, p1 = (lane + 64 < N) ? rr_exp(x[1] - mm) : 0.f;
    const float inv = 1.0f / rr_wave_sum(p0 + p1);
    float c0 = 0.f, c1 = 0.f;
    for (int j = 0; j < N; ++j) {
      const float xj = (j & 64) ? __shfl(x[1], j & 63) : __shfl(x[0], j & 63);
      const float pj = ((j & 64) ? __shfl(p1, j & 63) : __shfl(p0, j & 63)) * inv;
      c0 += (xj < x[0] || (xj == x[0] && j <= lane)) ? pj : 0.f;
      c1 += (xj < x[1] || (xj == x[1] && j <= lane + 64)) ? pj : 0.f;
    }
    if (c0 <= 1.0f - top_p) x[0] = -INFINITY;
    if (c1 <= 1.0f - top_p) x[1] = -INFINITY;
  }
  float m = rr_wave_max(fmaxf(x[0], x[1]));
  float e0 = (lane < N) ? rr_exp(x[0] - m) : 0.f;
  float e1 = (lane + 64 < N) ? rr_exp(x[1] - m) : 0.f;
  float ssum = rr_wave_sum(e0 + e1);
  float lse = rr_log(ssum);
  float lp0 = x[0] - m - lse, lp1 = x[1] - m - lse;
  if (logp_all != nullptr) {
    if (lane < N) logp_all[(size_t)r * N + lane] = lp0;
    if (lane + 64 < N) logp_all[(size_t)r * N + lane + 64] = lp1;
  }
  int sel;
  if (mode == 2) {
    sel = (int)action_in[r];
  } else if (mode == 0) {
    // argmax, lowest index on ties (torch.argmax on CPU returns the first maximal index)
    float bv = lp0; int bi = lane;
    if (lane + 64 < N && lp1 > bv) { bv = lp1; bi = lane + 64; }
    if (lane >= N) { bv = -INFINITY; bi = 0x7fffffff; }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
      float ov = __shfl_xor(bv, o); int oi = __shfl_xor(bi, o);
      if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
    }
    sel = bi;
  } else {
    // inverse CDF over the keys in ascending order (rr_common.h: same uniform, same rule as the fused rollout kernel, so both paths draw
    // the same action for a given (seed, rollout, step) unless fp32 noise moves a boundary across the target)
    float t0, t1;
    const float x0 = rr_wave_excl_scan(e0, t0), x1 = rr_wave_excl_scan(e1, t1);
    const float target = rr_cdf_target(rr_uniform(seed, (uint32_t)r, step, RR_CDF_SLOT), t0 + t1);
    int cand = -1;
    if (e0 > 0.f && x0 <= target) cand = lane;
    if (e1 > 0.f && t0 + x1 <= target) cand = lane + 64;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) cand = max(cand, __shfl_xor(cand, o));
    sel = cand < 0 ? 0x7fffffff : cand;
  }
  float lpsel = (sel & 64) ? __shfl(lp1, sel & 63) : __shfl(lp0, sel & 63);
  if (lane == 0) { action_out[r] = sel; logp_out[r] = lpsel; }
}

__global__ __launch_bounds__(256) void k_select(const float* __restrict__ logits, const uint8_t* __restrict__ mask,
                                                const int64_t* __restrict__ action_in, int64_t* __restrict__ action_out,
                                                float* __restrict__ logp_out, float* __restrict__ logp_all,
                                                int R, int N, float tanh_clip, float temperature, int mode,
                                                uint64_t seed, uint32_t step, int top_k, float top_p) {
  const int lane = threadIdx.x & 63;
  const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * SEL_ROWS;
  if (r0 >= R) return;
  float raw[SEL_ROWS][2];
  uint8_t keep[SEL_ROWS][2];
#pragma unroll
  for (int i = 0; i < SEL_ROWS; ++i)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int j = lane + 64 * q;
      const bool ok = r0 + i < R && j < N;
      const size_t off = (size_t)(r0 + i) * N + j;
      raw[i][q] = ok ? logits[off] : 0.f;
      keep[i][q] = (ok && mask != nullptr) ? mask[off] : (uint8_t)1;
    }
#pragma unroll
  for (int i = 0; i < SEL_ROWS; ++i)
    if (r0 + i < R)
      rr_select_row(raw[i], keep[i], r0 + i, lane, action_in, action_out, logp_out, logp_all, N, tanh_clip, temperature, mode,
                    seed, step, top_k, top_p);
}

// Second generation for the plain strategies (no top-k / top-p): 16 lanes per row, 4 rows per wave pass, SEL16_PASSES
// passes per wave with every load issued up front.  A lane owns two groups of four consecutive keys (16-byte logits load,
// 4-byte mask load when N % 4 == 0), so a row's reductions are 7 in-lane steps + 4 DPP steps inside the 16-lane row
// (quad_perm xor 1 / xor 2, row_half_mirror, row_mirror: every lane ends with the row's result) instead of 6 ds_bpermute
// round trips per reduction; elementwise formulas are those of rr_select_row.
#define SEL16_PASSES 4
// (rr_dpp<CTRL>: rr_common.h)
template <int CTRL> __device__ __forceinline__ int rr_dppi(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
__device__ __forceinline__ float rr_row16_max(float v) {
  v = fmaxf(v, rr_dpp<0xB1>(v)); v = fmaxf(v, rr_dpp<0x4E>(v)); v = fmaxf(v, rr_dpp<0x141>(v)); return fmaxf(v, rr_dpp<0x140>(v));
}
__device__ __forceinline__ float rr_row16_sum(float v) {
  v += rr_dpp<0xB1>(v); v += rr_dpp<0x4E>(v); v += rr_dpp<0x141>(v); return v + rr_dpp<0x140>(v);
}
// (value, key, logp) candidates: larger value wins, then the lower key — symmetric, so both lanes of a pair agree
template <int CTRL> __device__ __forceinline__ void rr_row16_best(float& bv, int& bi, float& blp) {
  const float ov = rr_dpp<CTRL>(bv), ol = rr_dpp<CTRL>(blp);
  const int oi = rr_dppi<CTRL>(bi);
  const bool take = ov > bv || (ov == bv && oi < bi);
  bv = take ? ov : bv; bi = take ? oi : bi; blp = take ? ol : blp;
}

// MODE is a compile-time copy of `mode` (0 greedy, 1 sampling, 2 evaluate; -1: decided at run time, the MatNet variant): the
// greedy instantiation carries no Gumbel / evaluate code, and without a full log-probability row to write it needs no third
// pass over the keys at all — the winner is the first key equal to the row maximum and its log-probability is -lse.
template <bool VEC, int MODE = -1>
__global__ __launch_bounds__(256) void k_select16(const float* __restrict__ logits, const uint8_t* __restrict__ mask,
                                                  const int64_t* __restrict__ action_in, int64_t* __restrict__ action_out,
                                                  float* __restrict__ logp_out, float* __restrict__ logp_all,
                                                  int R, int N, float tanh_clip, float temperature, int mode_rt,
                                                  uint64_t seed, uint32_t step, int shift_clamp) {
  const int mode = MODE >= 0 ? MODE : mode_rt;
  const int lane = threadIdx.x & 63, p = lane & 15, rw = lane >> 4;
  const int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * (4 * SEL16_PASSES) + rw;
  float raw[SEL16_PASSES][8];
  uint32_t keep[SEL16_PASSES][2];
#pragma unroll
  for (int i = 0; i < SEL16_PASSES; ++i) {
    const int r = r0 + 4 * i;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int c = 4 * (p + 16 * q);
      const size_t off = (size_t)r * N + c;
      keep[i][q] = 0x01010101u;
      if (VEC) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (r < R && c < N) {
          // read once, never again: non-temporal loads keep the 210 MB stream from evicting what L2 holds for the next kernel
          const f32x4 nv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(logits + off));
          v = make_float4(nv[0], nv[1], nv[2], nv[3]);
          if (mask != nullptr) keep[i][q] = __builtin_nontemporal_load(reinterpret_cast<const uint32_t*>(mask + off));
        }
        raw[i][4 * q] = v.x; raw[i][4 * q + 1] = v.y; raw[i][4 * q + 2] = v.z; raw[i][4 * q + 3] = v.w;
      } else {
        uint32_t kw = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const bool ok = r < R && c + k < N;
          raw[i][4 * q + k] = ok ? logits[off + k] : 0.f;
          kw |= (uint32_t)((ok && mask != nullptr) ? (mask[off + k] != 0) : 1) << (8 * k);
        }
        keep[i][q] = kw;
      }
    }
  }
  const float inv_temp = 1.0f / temperature;
#pragma unroll
  for (int i = 0; i < SEL16_PASSES; ++i) {
    const int r = r0 + 4 * i;            // rows >= R run on zeros and store nothing: the DPP steps need the whole wave
    float x[8];
    float m = -INFINITY;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int j = 4 * (p + 16 * (e >> 2)) + (e & 3);
      float v = raw[i][e];
      // tanh through one plain v_exp_f32: |error| <= 4e-8 on tanh (4e-7 after the x10 clip), largest near |v| = 0.75
      if (tanh_clip > 0.f) v = fmaf(__builtin_amdgcn_rcpf(rr_exp_fast(2.0f * v) + 1.0f), -2.0f * tanh_clip, tanh_clip);
      if (((keep[i][e >> 2] >> (8 * (e & 3))) & 0xffu) == 0) v = -INFINITY;
      v = v * inv_temp;
      x[e] = j < N ? v : -INFINITY;
      m = fmaxf(m, x[e]);
    }
    m = rr_row16_max(m);
    if (shift_clamp) {     // MatNet's process_logits (rrnco/baselines/MatNet/decoding.py:357-359): x - max, clamped to [-50, -1e-4]:
      const float m0 = m;  // actions within 1e-4 of the best tie with it, masked ones keep the finite logit -50
      m = -INFINITY;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int j = 4 * (p + 16 * (e >> 2)) + (e & 3);
        if (j < N) { x[e] = fminf(fmaxf(x[e] - m0, -50.0f), -1e-4f); m = fmaxf(m, x[e]); }
      }
      m = rr_row16_max(m);
    }
    float ssum = 0.f, en[8];                            // the softmax numerators (sampling reads them again for its CDF)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int j = 4 * (p + 16 * (e >> 2)) + (e & 3);
      en[e] = j < N ? rr_exp_fast(x[e] - m) : 0.f;       // arguments in [-2 clip / T, 0]; same form as the fused rollout
      ssum += en[e];
    }
    const float lse = rr_log(rr_row16_sum(ssum));
    if (MODE == 0 && logp_all == nullptr) {          // greedy: argmax by equality with the row maximum, first index on ties
      int bi = 0x7fffffff;
#pragma unroll
      for (int e = 7; e >= 0; --e) {
        const int j = 4 * (p + 16 * (e >> 2)) + (e & 3);
        bi = (j < N && x[e] == m) ? j : bi;             // descending e: the lowest key of the lane survives
      }
      bi = min(bi, rr_dppi<0xB1>(bi)); bi = min(bi, rr_dppi<0x4E>(bi)); bi = min(bi, rr_dppi<0x141>(bi)); bi = min(bi, rr_dppi<0x140>(bi));
      if (p == 0 && r < R) { action_out[r] = bi; logp_out[r] = -lse; }      // x[best] - m - lse with x[best] == m
      continue;
    }
    if (MODE == 1 && logp_all == nullptr) {
      // sampling without the full log-probability row: inverse CDF over the keys in ascending order (rr_common.h; keys 4p .. 4p+3 of the
      // row's first 64, then of its second 64).  The draw is the LAST key with mass whose exclusive prefix is <= target: ascending keys
      // in the lane (a later eligible key overwrites), then the largest candidate of the row; only the winner's logit is carried along.
      const float s0 = ((en[0] + en[1]) + en[2]) + en[3], s1 = ((en[4] + en[5]) + en[6]) + en[7];
      float t0, t1;
      const float c0 = rr_row16_excl_scan(s0, t0), c1 = t0 + rr_row16_excl_scan(s1, t1);
      const float target = rr_cdf_target(rr_uniform(seed, (uint32_t)r, step, RR_CDF_SLOT), t0 + t1);
      int cand = -1;
      float xs = 0.f, crun = c0;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (e == 4) crun = c1;
        const bool el = en[e] > 0.f && crun <= target;
        cand = el ? 4 * (p + 16 * (e >> 2)) + (e & 3) : cand;
        xs = el ? x[e] : xs;
        crun += en[e];
      }
#define RR_CAND_STEP(CTRL) { const int oc = rr_dppi<CTRL>(cand); const float ox = __int_as_float(rr_dppi<CTRL>(__float_as_int(xs))); \
                             const bool take = oc > cand; cand = take ? oc : cand; xs = take ? ox : xs; }
      RR_CAND_STEP(0xB1) RR_CAND_STEP(0x4E) RR_CAND_STEP(0x141) RR_CAND_STEP(0x140)
#undef RR_CAND_STEP
      if (p == 0 && r < R) { action_out[r] = cand < 0 ? 0x7fffffff : cand; logp_out[r] = xs - m - lse; }
      continue;
    }
    const int want = (mode == 2 && r < R) ? (int)action_in[r] : -1;
    float bv = -INFINITY, blp = 0.f;
    int bi = 0x7fffffff;
    float lp[8];
    // sampling: inverse CDF over the keys in ascending order (rr_common.h; keys 4p .. 4p+3 of the row's first 64, then of its second 64):
    // a key's value is its index where it is eligible, so the maximum below is the last eligible key
    float cpre[2] = {0.f, 0.f}, target = 0.f;
    if (mode == 1) {
      const float s0 = ((en[0] + en[1]) + en[2]) + en[3], s1 = ((en[4] + en[5]) + en[6]) + en[7];
      float t0, t1;
      cpre[0] = rr_row16_excl_scan(s0, t0);
      cpre[1] = t0 + rr_row16_excl_scan(s1, t1);
      target = rr_cdf_target(rr_uniform(seed, (uint32_t)r, step, RR_CDF_SLOT), t0 + t1);
    }
    float crun = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int j = 4 * (p + 16 * (e >> 2)) + (e & 3);
      lp[e] = x[e] - m - lse;
      float sv;
      if (mode == 2) sv = j == want ? 1.f : -INFINITY;
      else if (mode == 1) {
        if ((e & 3) == 0) crun = cpre[e >> 2];
        sv = (en[e] > 0.f && crun <= target) ? (float)j : -INFINITY;
        crun += en[e];
      }
      else sv = j < N ? lp[e] : -INFINITY;
      const bool better = sv > bv || (bi == 0x7fffffff && j < N && mode == 0);   // ascending keys in the lane: first maximum kept
      bv = better ? sv : bv; bi = better ? j : bi; blp = better ? lp[e] : blp;
    }
    if (logp_all != nullptr && r < R) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int c = 4 * (p + 16 * q);
        float* dst = logp_all + (size_t)r * N + c;
        if (VEC) { if (c < N) *reinterpret_cast<float4*>(dst) = make_float4(lp[4 * q], lp[4 * q + 1], lp[4 * q + 2], lp[4 * q + 3]); }
        else {
#pragma unroll
          for (int k = 0; k < 4; ++k) if (c + k < N) dst[k] = lp[4 * q + k];
        }
      }
    }
    rr_row16_best<0xB1>(bv, bi, blp); rr_row16_best<0x4E>(bv, bi, blp);
    rr_row16_best<0x141>(bv, bi, blp); rr_row16_best<0x140>(bv, bi, blp);
    if (p == 0 && r < R) { action_out[r] = bi; logp_out[r] = blp; }
  }
}

extern "C" int rr_select(const float* logits, const uint8_t* mask, const int64_t* action_in, int64_t* action_out,
                         float* logp_out, float* logp_all, int R, int N, float tanh_clip, float temperature,
                         int mode, uint64_t seed, uint32_t step, int top_k, float top_p, hipStream_t st) {
  if (R <= 0 || N <= 0 || N > 128 || temperature <= 0.f || top_k < 0 || top_p < 0.f || top_p > 1.f) return RR_EINVAL;
  if (mode == 2 && action_in == nullptr) return RR_EINVAL;
  if (logits == nullptr || action_out == nullptr || logp_out == nullptr || mode < 0 || mode > 2) return RR_EINVAL;
  static const int gen = [] { const char* e = getenv("RR_SELECT_VARIANT"); return e ? atoi(e) : 1; }();
  const bool filters = (top_k > 0 && top_k < N) || (top_p > 0.f && top_p < 1.f);
  if (gen == 1 && !filters) {
    const uintptr_t al = reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(logp_all);
    const bool vec = N % 4 == 0 && (al & 15) == 0 && (reinterpret_cast<uintptr_t>(mask) & 3) == 0;
    const dim3 grid((R + 16 * SEL16_PASSES - 1) / (16 * SEL16_PASSES));
#define RR_SEL16(V, M) hipLaunchKernelGGL((k_select16<V, M>), grid, dim3(256), 0, st, logits, mask, action_in, action_out, logp_out, logp_all, R, N, \
                                         tanh_clip, temperature, mode, seed, step, 0)
    if (vec) { if (mode == 0) RR_SEL16(true, 0); else if (mode == 1) RR_SEL16(true, 1); else RR_SEL16(true, 2); }
    else { if (mode == 0) RR_SEL16(false, 0); else if (mode == 1) RR_SEL16(false, 1); else RR_SEL16(false, 2); }
#undef RR_SEL16
    return rr_check(hipGetLastError());
  }
  hipLaunchKernelGGL(k_select, dim3((R + 4 * SEL_ROWS - 1) / (4 * SEL_ROWS)), dim3(256), 0, st, logits, mask, action_in, action_out, logp_out,
                     logp_all, R, N, tanh_clip, temperature, mode, seed, step, top_k, top_p);
  return rr_check(hipGetLastError());
}

// MatNet's own process_logits + selection (rrnco/baselines/MatNet/decoding.py:316-372, 219-298): as rr_select without the
// top-k / top-p filters, with the row shifted by its maximum and clamped to [-50, -1e-4] before the log-softmax.
extern "C" int rr_select_matnet(const float* logits, const uint8_t* mask, const int64_t* action_in, int64_t* action_out,
                                float* logp_out, float* logp_all, int R, int N, float tanh_clip, float temperature,
                                int mode, uint64_t seed, uint32_t step, hipStream_t st) {
  if (R <= 0 || N <= 0 || N > 128 || temperature <= 0.f || mode < 0 || mode > 2) return RR_EINVAL;
  if (logits == nullptr || mask == nullptr || action_out == nullptr || logp_out == nullptr || (mode == 2 && action_in == nullptr)) return RR_EINVAL;
  const dim3 grid((R + 16 * SEL16_PASSES - 1) / (16 * SEL16_PASSES));
  hipLaunchKernelGGL(k_select16<false>, grid, dim3(256), 0, st, logits, mask, action_in, action_out, logp_out, logp_all, R, N,
                     tanh_clip, temperature, mode, seed, step, 1);
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// RMTVRPEnv._step + get_action_mask under the vrptw preset (rmtvrp/env.py:155-215, 343-428): backhaul demands are
// zero, routes closed, no distance limit, backhaul class 1 — those terms are inert and not evaluated.
// One wave per rollout; rollout r uses instance r % Bp; N counts the depot.
// ------------------------------------------------------------------------------------------------
// Optional per-variant data of the multi-task VRP (backhauls B / MB, open routes O, distance limits L); NULL = vrptw preset.
struct MtvrpExtra {
  const float* demand_b;      // [Bp][N] demand_backhaul incl. the depot zero
  float* used_b;              // [R] used_capacity_backhaul (updated)
  const uint8_t* open_route;  // [Bp]
  const float* dist_limit;    // [Bp] (+inf = none)
  const int32_t* bclass;      // [Bp] backhaul class 1 (classical) / 2 (mixed)
};

__global__ __launch_bounds__(256) void k_rmtvrp_step(const int64_t* __restrict__ action, const float* __restrict__ D,
                                                     const float* __restrict__ T, const float* __restrict__ dem_l,
                                                     const float* __restrict__ tw, const float* __restrict__ service,
                                                     const float* __restrict__ vcap, int64_t* __restrict__ cur_io,
                                                     float* __restrict__ ctime, float* __restrict__ rlen,
                                                     float* __restrict__ used_l, uint8_t* __restrict__ visited,
                                                     uint8_t* __restrict__ mask, uint8_t* __restrict__ done,
                                                     int R, int Bp, int N, MtvrpExtra ex, int has_ex) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int b = r % Bp;
  const float* Db = D + (size_t)b * N * N;
  const float* Tb = T + (size_t)b * N * N;
  const float* twb = tw + (size_t)b * N * 2;
  const float* svb = service + (size_t)b * N;
  const float* dl = dem_l + (size_t)b * N;
  const float* db = has_ex ? ex.demand_b + (size_t)b * N : nullptr;
  const int prev = (int)cur_io[r], a = (int)action[r];
  const float nz = a != 0 ? 1.0f : 0.0f;
  const float t1 = nz * (fmaxf(ctime[r] + Tb[prev * N + a], twb[a * 2]) + svb[a]);      // env.py:170-172
  const float len1 = nz * (rlen[r] + Db[prev * N + a]);                                // :175-177
  const float u1 = nz * (used_l[r] + dl[a]);                                           // :189-191
  const float ub1 = has_ex ? nz * (ex.used_b[r] + db[a]) : 0.f;                        // :192-194
  const float cap = vcap[r], late0 = twb[1];
  const float closed = (has_ex && ex.open_route[b]) ? 0.f : 1.f;                       // `* ~open_route`
  const float limit = has_ex ? ex.dist_limit[b] : INFINITY;
  const int bclass = has_ex ? ex.bclass[b] : 1;
  const bool carrying_b = has_ex && db[a] > 0.f;                                       // :388-396
  uint8_t* vis = visited + (size_t)r * N;
  uint8_t* mk = mask + (size_t)r * N;
  int nvis = 0, missing = 0;
  for (int k = lane; k < N; k += 64) {
    uint8_t v = vis[k];
    if (k == a) v = 1;
    vis[k] = v;
    nvis += (v != 0);
    missing |= (v == 0 && dl[k] > 0.f);                                                // linehauls_missing :384-386
  }
  nvis = (int)rr_wave_sum((float)nvis);
  missing = __any(missing);
  int nfree = 0;
  for (int k = lane; k < N; k += 64) {
    const bool v = vis[k] != 0;
    const float arrival = t1 + Tb[a * N + k];
    const bool reach = arrival < twb[k * 2 + 1];                                       // :361
    const bool back = ((fmaxf(arrival, twb[k * 2]) + svb[k] + Tb[k * N]) * closed) < late0;          // :364-366
    const bool far = (len1 + Db[a * N + k] + Db[k * N] * closed) > limit;              // :369-372
    const float dlk = dl[k], dbk = has_ex ? db[k] : 0.f;
    const bool ex_l = dlk + u1 > cap, ex_b = dbk + ub1 > cap;                          // :375-380
    const bool ok1 = (missing && !ex_l && !carrying_b && dlk > 0.f) || (!ex_b && dbk > 0.f);         // :397-402
    const bool ok2 = !ex_l && !ex_b && !(dlk > cap - ub1);                             // :407-412
    const bool ok = bclass == 1 ? ok1 : (bclass == 2 ? ok2 : false);                   // :415-417
    const bool can = reach && back && ok && !far && !v;                                // :420-426
    if (k >= 1) { mk[k] = can; nfree += can; }
  }
  nfree = (int)rr_wave_sum((float)nfree);
  if (lane == 0) {
    mk[0] = !((a == 0) && nfree > 0);                                                  // :429
    cur_io[r] = a; ctime[r] = t1; rlen[r] = len1; used_l[r] = u1;
    if (has_ex) ex.used_b[r] = ub1;
    done[r] = (nvis == N);
  }
}

// Second form (4 <= N <= 128): 32 lanes per rollout, lane l owns the four nodes of the window starting at min(4 l, N - 4)
// (the last window is pulled back to end at N - 1 and overlaps its neighbour, which computes and stores the same bytes),
// so every per-node operand is one 16-byte load: the rows D[a,:], T[a,:], the depot columns as contiguous vectors
// (to_depot_D / to_depot_T, [Bp][N], prepared once per instance by the caller: read through D[k*N] they were 2 x N
// separate cache lines per rollout, 80 % of the kernel's L2 traffic), time windows, service times, demands.  The counts of
// the first form (visited == N, linehauls missing, any free customer) are existence tests: three ballots, no shuffle
// reduction, and `visited` is updated by one byte store instead of being rewritten.  Same fp32 expressions as
// k_rmtvrp_step: masks are bit-identical.  MTV_PASSES x 2 rollouts per wave.
typedef float rr_f32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t rr_u32_a1 __attribute__((aligned(1)));
#define MTV_PASSES 2
__global__ __launch_bounds__(256) void k_rmtvrp_step_v(const int64_t* __restrict__ action, const float* __restrict__ D,
                                                       const float* __restrict__ T, const float* __restrict__ D0,
                                                       const float* __restrict__ T0, const float* __restrict__ dem_l,
                                                       const float* __restrict__ tw, const float* __restrict__ service,
                                                       const float* __restrict__ vcap, int64_t* __restrict__ cur_io,
                                                       float* __restrict__ ctime, float* __restrict__ rlen,
                                                       float* __restrict__ used_l, uint8_t* __restrict__ visited,
                                                       uint8_t* __restrict__ mask, uint8_t* __restrict__ done,
                                                       int R, int Bp, int N, MtvrpExtra ex, int has_ex) {
  const int lane = threadIdx.x & 63, hl = lane & 31, half = lane >> 5;
  const long rbase = ((long)blockIdx.x * 4 + (threadIdx.x >> 6)) * (2 * MTV_PASSES);
  const int ws = min(4 * hl, N - 4);
  const bool col_ok = 4 * hl < N;
  int a[MTV_PASSES], prev[MTV_PASSES], b[MTV_PASSES];
  float ct[MTV_PASSES], rl[MTV_PASSES], ul[MTV_PASSES], ub[MTV_PASSES], cap[MTV_PASSES];
#pragma unroll
  for (int i = 0; i < MTV_PASSES; ++i) {
    const long r = rbase + 2 * i + half;
    const long rc = r < R ? r : R - 1;                    // rows past the end recompute the last row and store nothing
    b[i] = (int)(rc % Bp);
    a[i] = (int)action[rc]; prev[i] = (int)cur_io[rc];
    ct[i] = ctime[rc]; rl[i] = rlen[rc]; ul[i] = used_l[rc]; cap[i] = vcap[rc];
    ub[i] = has_ex ? ex.used_b[rc] : 0.f;
  }
  float Tpa[MTV_PASSES], Dpa[MTV_PASSES], twa[MTV_PASSES], sva[MTV_PASSES], dla[MTV_PASSES], dba[MTV_PASSES], late0[MTV_PASSES];
  float closed[MTV_PASSES], limit[MTV_PASSES];
  int bclass[MTV_PASSES];
  uint32_t visw[MTV_PASSES];
  rr_f32x4_a4 Trow[MTV_PASSES], Drow[MTV_PASSES], T0v[MTV_PASSES], D0v[MTV_PASSES], twlo[MTV_PASSES], twhi[MTV_PASSES];
  rr_f32x4_a4 svv[MTV_PASSES], dlv[MTV_PASSES], dbv[MTV_PASSES];
#pragma unroll
  for (int i = 0; i < MTV_PASSES; ++i) {
    const long r = rbase + 2 * i + half;
    const long rc = r < R ? r : R - 1;
    const size_t bo = (size_t)b[i] * N;
    const float* Db = D + bo * N;
    const float* Tb = T + bo * N;
    const float* twb = tw + bo * 2;
    Tpa[i] = Tb[prev[i] * N + a[i]]; Dpa[i] = Db[prev[i] * N + a[i]];
    twa[i] = twb[a[i] * 2]; sva[i] = service[bo + a[i]]; dla[i] = dem_l[bo + a[i]]; late0[i] = twb[1];
    dba[i] = has_ex ? ex.demand_b[bo + a[i]] : 0.f;
    closed[i] = (has_ex && ex.open_route[b[i]]) ? 0.f : 1.f;
    limit[i] = has_ex ? ex.dist_limit[b[i]] : INFINITY;
    bclass[i] = has_ex ? ex.bclass[b[i]] : 1;
    visw[i] = *reinterpret_cast<const rr_u32_a1*>(visited + rc * N + ws);
    Trow[i] = *reinterpret_cast<const rr_f32x4_a4*>(Tb + a[i] * N + ws);
    Drow[i] = *reinterpret_cast<const rr_f32x4_a4*>(Db + a[i] * N + ws);
    if (T0 != nullptr) {
      T0v[i] = *reinterpret_cast<const rr_f32x4_a4*>(T0 + bo + ws);
      D0v[i] = *reinterpret_cast<const rr_f32x4_a4*>(D0 + bo + ws);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) { T0v[i][q] = Tb[(ws + q) * N]; D0v[i][q] = Db[(ws + q) * N]; }
    }
    twlo[i] = *reinterpret_cast<const rr_f32x4_a4*>(twb + 2 * ws);
    twhi[i] = *reinterpret_cast<const rr_f32x4_a4*>(twb + 2 * ws + 4);
    svv[i] = *reinterpret_cast<const rr_f32x4_a4*>(service + bo + ws);
    dlv[i] = *reinterpret_cast<const rr_f32x4_a4*>(dem_l + bo + ws);
    if (has_ex) dbv[i] = *reinterpret_cast<const rr_f32x4_a4*>(ex.demand_b + bo + ws);
    else dbv[i] = rr_f32x4_a4{0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int i = 0; i < MTV_PASSES; ++i) {
    const long r = rbase + 2 * i + half;
    const bool row_ok = r < R;
    const int av = a[i];
    const float nz = av != 0 ? 1.0f : 0.0f;
    const float t1 = nz * (fmaxf(ct[i] + Tpa[i], twa[i]) + sva[i]);                     // env.py:170-172
    const float len1 = nz * (rl[i] + Dpa[i]);                                          // :175-177
    const float u1 = nz * (ul[i] + dla[i]);                                            // :189-191
    const float ub1 = has_ex ? nz * (ub[i] + dba[i]) : 0.f;                            // :192-194
    const bool carrying_b = has_ex && dba[i] > 0.f;                                    // :388-396
    bool vq[4];
    bool unvis = false, miss = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      vq[q] = ((visw[i] >> (8 * q)) & 0xffu) != 0 || ws + q == av;
      unvis |= !vq[q];
      miss |= !vq[q] && dlv[i][q] > 0.f;                                               // linehauls_missing :384-386
    }
    const int sh = 32 * half;
    const bool missing = (uint32_t)(__ballot(col_ok && miss) >> sh) != 0;
    const bool all_visited = (uint32_t)(__ballot(col_ok && unvis) >> sh) == 0;
    bool can[4];
    bool free_here = false;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float early = q < 2 ? twlo[i][2 * q] : twhi[i][2 * q - 4], late = q < 2 ? twlo[i][2 * q + 1] : twhi[i][2 * q - 3];
      const float arrival = t1 + Trow[i][q];
      const bool reach = arrival < late;                                               // :361
      const bool back = ((fmaxf(arrival, early) + svv[i][q] + T0v[i][q]) * closed[i]) < late0[i];      // :364-366
      const bool far = (len1 + Drow[i][q] + D0v[i][q] * closed[i]) > limit[i];         // :369-372
      const float dlk = dlv[i][q], dbk = dbv[i][q];
      const bool ex_l = dlk + u1 > cap[i], ex_b = dbk + ub1 > cap[i];                  // :375-380
      const bool ok1 = (missing && !ex_l && !carrying_b && dlk > 0.f) || (!ex_b && dbk > 0.f);       // :397-402
      const bool ok2 = !ex_l && !ex_b && !(dlk > cap[i] - ub1);                        // :407-412
      const bool ok = bclass[i] == 1 ? ok1 : (bclass[i] == 2 ? ok2 : false);           // :415-417
      can[q] = reach && back && ok && !far && !vq[q];                                  // :420-426
      free_here |= can[q] && ws + q >= 1;
    }
    const bool anyfree = (uint32_t)(__ballot(col_ok && free_here) >> sh) != 0;
    uint32_t mw = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const bool bit = ws + q == 0 ? !(av == 0 && anyfree) : can[q];                   // depot rule :429
      mw |= (bit ? 1u : 0u) << (8 * q);
    }
    if (row_ok && col_ok) *reinterpret_cast<rr_u32_a1*>(mask + r * N + ws) = mw;
    if (row_ok && hl == 0) {
      visited[r * N + av] = 1;
      cur_io[r] = av; ctime[r] = t1; rlen[r] = len1; used_l[r] = u1;
      if (has_ex) ex.used_b[r] = ub1;
      done[r] = all_visited;
    }
  }
}

extern "C" int rr_rmtvrp_step(const int64_t* action, const float* D, const float* T, const float* to_depot_D,
                              const float* to_depot_T, const float* demand_l,
                              const float* tw, const float* service, const float* vcap, int64_t* cur, float* ctime,
                              float* rlen, float* used_l, uint8_t* visited, uint8_t* mask, uint8_t* done,
                              int R, int Bp, int N, const MtvrpExtra* extra, hipStream_t st) {
  if (R <= 0 || N < 2 || Bp <= 0) return RR_EINVAL;
  if (!action || !D || !T || !demand_l || !tw || !service || !vcap || !cur || !ctime || !rlen || !used_l || !visited || !mask || !done) return RR_EINVAL;
  MtvrpExtra ex = {nullptr, nullptr, nullptr, nullptr, nullptr};
  if (extra != nullptr) {
    ex = *extra;
    if (!ex.demand_b || !ex.used_b || !ex.open_route || !ex.dist_limit || !ex.bclass) return RR_EINVAL;
  }
  if ((to_depot_D == nullptr) != (to_depot_T == nullptr)) return RR_EINVAL;
  static const int variant = [] { const char* e = getenv("RR_STEP_VARIANT"); return e ? atoi(e) : 1; }();
  if (variant != 0 && N >= 4 && N <= 128)
    hipLaunchKernelGGL(k_rmtvrp_step_v, dim3((R + 8 * MTV_PASSES - 1) / (8 * MTV_PASSES)), dim3(256), 0, st, action, D, T, to_depot_D,
                       to_depot_T, demand_l, tw, service, vcap, cur, ctime, rlen, used_l, visited, mask, done, R, Bp, N, ex,
                       extra != nullptr ? 1 : 0);
  else
    hipLaunchKernelGGL(k_rmtvrp_step, dim3((R + 3) / 4), dim3(256), 0, st, action, D, T, demand_l, tw, service, vcap, cur,
                       ctime, rlen, used_l, visited, mask, done, R, Bp, N, ex, extra != nullptr ? 1 : 0);
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// POMO shared-baseline REINFORCE loss (rrnco/models/rl.py:112-128; formula in-tree at
// rrnco/baselines/routefinder/model.py:182-202; rl4co SharedBaseline = mean over the S starts of an instance):
//   adv[b,s] = R[b,s] - mean_s R[b,s];  loss = -mean_{b,s}(adv * ll);  d loss / d ll = -adv / (B*S)
// rollout index r = s*B + b.  Stage 1: one wave per instance (partial sum per instance, fixed order);
// stage 2: one workgroup folds the B partials in a fixed order -> bit-reproducible loss.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_reinforce_stage1(const float* __restrict__ reward, const float* __restrict__ ll,
                                                          float* __restrict__ adv, float* __restrict__ grad_ll,
                                                          float* __restrict__ bl, float* __restrict__ partial, int B, int S) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (b >= B) return;
  float s = 0.f;
  for (int k = lane; k < S; k += 64) s += reward[(size_t)k * B + b];
  const float mean = rr_wave_sum(s) / (float)S;
  float acc = 0.f;
  const float scale = -1.0f / ((float)B * (float)S);
  for (int k = lane; k < S; k += 64) {
    const size_t r = (size_t)k * B + b;
    const float a = reward[r] - mean;
    adv[r] = a;
    grad_ll[r] = a * scale;
    acc += a * ll[r];
  }
  acc = rr_wave_sum(acc);
  if (lane == 0) { bl[b] = mean; partial[b] = acc; }
}

__global__ __launch_bounds__(256) void k_reinforce_stage2(const float* __restrict__ partial, float* __restrict__ loss, int B, int S) {
  __shared__ float sm[256];
  float s = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) s += partial[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o >= 1; o >>= 1) { if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o]; __syncthreads(); }
  if (threadIdx.x == 0) loss[0] = -sm[0] / ((float)B * (float)S);
}

extern "C" int rr_reinforce_loss(const float* reward, const float* ll, float* adv, float* grad_ll, float* bl,
                                 float* partial, float* loss, int B, int S, hipStream_t st) {
  if (B <= 0 || S <= 1) return RR_EINVAL;
  hipLaunchKernelGGL(k_reinforce_stage1, dim3((B + 3) / 4), dim3(256), 0, st, reward, ll, adv, grad_ll, bl, partial, B, S);
  hipLaunchKernelGGL(k_reinforce_stage2, dim3(1), dim3(256), 0, st, partial, loss, B, S);
  return rr_check(hipGetLastError());
}
