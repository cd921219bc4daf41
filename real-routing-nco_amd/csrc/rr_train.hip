// Training-side kernels of the Neural Adaptive Bias (gating DistAngleFusion, attn_freenet.py:242-289): the forward in the
// folded 128-unit form and its backward with respect to the folded parameters.  The NAB is 90 % of the reference's
// arithmetic and, differentiated through torch ops, 73 % of the config-5 training step (rocBLAS gemv on
// [B,N,N,128] tensors); here an edge costs ~1.3 kflop forward and ~3 kflop backward of plain VALU work and no [.,128] tensor
// is ever materialised.  Used by rrnco_amd/models/grad_replay.py through a torch.autograd.Function; the chain rule from the
// folded table back to the module parameters (W2^T wo etc.) stays in torch (tiny tensors).
//
// Table layout (same as nab_edge in rr_encoder.hip): rows a_d, b_d, co_d, cg_d, a_a, b_a, co_a, cg_a [8][E], then
// s = (ko_d, kg_d, ko_a, kg_a, bg, bo, alpha, -).   out = alpha * (g O_d + (1-g) O_a + bo),
//   O_x = co_x . relu(a_x x + b_x) + ko_x,  g = sigmoid(cg_d . h_d + kg_d + cg_a . h_a + kg_a + bg).
#include "rr_common.h"

#define TR_THREADS 256
#define TR_TAB (8 * RR_E + 8)

struct NabFwd { float od, oa, g; };

__device__ __forceinline__ NabFwd nab_fwd_edge(const float* __restrict__ tab, float d, float th) {
  float pod = 0.f, pgd = 0.f, poa = 0.f, pga = 0.f;
#pragma unroll 8
  for (int k = 0; k < RR_E; ++k) {
    const float hd = fmaxf(fmaf(tab[0 * RR_E + k], d, tab[1 * RR_E + k]), 0.f);
    pod = fmaf(tab[2 * RR_E + k], hd, pod);
    pgd = fmaf(tab[3 * RR_E + k], hd, pgd);
    const float ha = fmaxf(fmaf(tab[4 * RR_E + k], th, tab[5 * RR_E + k]), 0.f);
    poa = fmaf(tab[6 * RR_E + k], ha, poa);
    pga = fmaf(tab[7 * RR_E + k], ha, pga);
  }
  const float* s = tab + 8 * RR_E;
  NabFwd r;
  r.od = pod + s[0]; r.oa = poa + s[2];
  r.g = 1.0f / (1.0f + expf(-((pgd + s[1]) + (pga + s[3]) + s[4])));
  return r;
}

__global__ __launch_bounds__(TR_THREADS) void k_nab_train_fwd(const float* __restrict__ tab_g, const float* __restrict__ xd,
                                                              const float* __restrict__ xa, float* __restrict__ out, long M) {
  __shared__ float tab[TR_TAB];
  for (int i = threadIdx.x; i < TR_TAB; i += TR_THREADS) tab[i] = tab_g[i];
  __syncthreads();
  const float bo = tab[8 * RR_E + 5], alpha = tab[8 * RR_E + 6];
  for (long e = (long)blockIdx.x * TR_THREADS + threadIdx.x; e < M; e += (long)gridDim.x * TR_THREADS) {
    const NabFwd f = nab_fwd_edge(tab, xd[e], xa[e]);
    out[e] = (f.g * f.od + (1.0f - f.g) * f.oa + bo) * alpha;
  }
}

// grad_tab (TR_TAB floats, zeroed by the caller) += d loss / d table, given gout = d loss / d out per edge
__global__ __launch_bounds__(TR_THREADS) void k_nab_train_bwd(const float* __restrict__ tab_g, const float* __restrict__ xd,
                                                              const float* __restrict__ xa, const float* __restrict__ gout,
                                                              float* __restrict__ grad_tab, long M) {
  __shared__ float tab[TR_TAB];
  __shared__ float exd[TR_THREADS], exa[TR_THREADS], edod[TR_THREADS], edoa[TR_THREADS], edz[TR_THREADS];
  const int t = threadIdx.x;
  for (int i = t; i < TR_TAB; i += TR_THREADS) tab[i] = tab_g[i];
  __syncthreads();
  const float bo = tab[8 * RR_E + 5], alpha = tab[8 * RR_E + 6];
  // phase-B role of this thread: hidden unit k of family f
  const int f = t >> 7, k = t & 127;
  const float ak = tab[(4 * f + 0) * RR_E + k], bk = tab[(4 * f + 1) * RR_E + k];
  const float cok = tab[(4 * f + 2) * RR_E + k], cgk = tab[(4 * f + 3) * RR_E + k];
  float acc_a = 0.f, acc_b = 0.f, acc_co = 0.f, acc_cg = 0.f;
  float s_alpha = 0.f, s_bo = 0.f, s_od = 0.f, s_oa = 0.f, s_z = 0.f;     // phase-A scalar sums of this thread's edges
  const long nchunk = (M + TR_THREADS - 1) / TR_THREADS;
  for (long c = blockIdx.x; c < nchunk; c += gridDim.x) {
    const long e = c * TR_THREADS + t;
    float d = 0.f, th = 0.f, dod = 0.f, doa = 0.f, dz = 0.f;
    if (e < M) {
      d = xd[e]; th = xa[e];
      const NabFwd r = nab_fwd_edge(tab, d, th);
      const float G = gout[e];
      const float val = r.g * r.od + (1.0f - r.g) * r.oa + bo;
      const float Gv = G * alpha;
      dod = Gv * r.g; doa = Gv * (1.0f - r.g); dz = Gv * (r.od - r.oa) * r.g * (1.0f - r.g);
      s_alpha += G * val; s_bo += Gv; s_od += dod; s_oa += doa; s_z += dz;
    }
    __syncthreads();                       // the previous chunk's phase B is done with the edge arrays
    exd[t] = d; exa[t] = th; edod[t] = dod; edoa[t] = doa; edz[t] = dz;      // edges past M contribute zeros
    __syncthreads();
    const float* ex = f ? exa : exd;
    const float* eo = f ? edoa : edod;
#pragma unroll 4
    for (int i = 0; i < TR_THREADS; ++i) {
      const float x = ex[i], dO = eo[i], dzv = edz[i];
      const float pre = fmaf(ak, x, bk);
      const float h = fmaxf(pre, 0.f);
      acc_co = fmaf(dO, h, acc_co);
      acc_cg = fmaf(dzv, h, acc_cg);
      const float dh = pre > 0.f ? fmaf(dO, cok, dzv * cgk) : 0.f;
      acc_a = fmaf(dh, x, acc_a);
      acc_b += dh;
    }
  }
  atomicAdd(&grad_tab[(4 * f + 0) * RR_E + k], acc_a);
  atomicAdd(&grad_tab[(4 * f + 1) * RR_E + k], acc_b);
  atomicAdd(&grad_tab[(4 * f + 2) * RR_E + k], acc_co);
  atomicAdd(&grad_tab[(4 * f + 3) * RR_E + k], acc_cg);
  s_alpha = rr_wave_sum(s_alpha); s_bo = rr_wave_sum(s_bo); s_od = rr_wave_sum(s_od); s_oa = rr_wave_sum(s_oa); s_z = rr_wave_sum(s_z);
  if ((t & 63) == 0) {
    float* gs = grad_tab + 8 * RR_E;
    atomicAdd(&gs[0], s_od); atomicAdd(&gs[1], s_z); atomicAdd(&gs[2], s_oa); atomicAdd(&gs[3], s_z);
    atomicAdd(&gs[4], s_z); atomicAdd(&gs[5], s_bo); atomicAdd(&gs[6], s_alpha);
  }
}

// ------------------------------------------------------------------------------------------------
// The same backward in O(1) per edge instead of O(128).  Each of the four scalar functions of the folded NAB,
//   f(x) = sum_k c_k relu(a_k x + b_k) + const,
// is piecewise linear between the sorted breakpoints t_k = -b_k / a_k (the table the encoder kernel evaluates,
// packing.fold_nab_pwl), and unit k is active on a prefix or a suffix of the segments.  So
//   d loss / d c_k = sum_e w_e relu(a_k x_e + b_k) = a_k S1_k + b_k S0_k,   S0_k = sum_{e: k active} w_e,  S1_k = sum_{..} w_e x_e
// (and likewise d a_k, d b_k) only need, per family and SEGMENT, the four moments sum w_out, sum w_out x, sum w_gate, sum w_gate x
// of the edges that fall into it: eight LDS atomics per edge here, a prefix sum over 129 segments on the host
// (models/enc_backward.py:nab_grad_from_hist).  hist: [2][129][4] moments + [1] d alpha, ADDED to (caller zeroes).
#define NH_TAB (256 + 2 * 129 * 4 + 8)
#define NH_HIST (2 * 129 * 4)
__device__ __forceinline__ int nh_segment(const float* t, float x) {        // number of breakpoints <= x (bisection, as nab_family)
  int m = 0;
#pragma unroll
  for (int s = 128; s >= 1; s >>= 1) {
    const int idx = m + s - 1;
    const float tv = t[idx < 128 ? idx : 127];
    m += (idx < 128 && tv <= x) ? s : 0;
  }
  return m;
}
// The per-segment sums are kept in LDS as 64-bit FIXED-POINT integers: `ds_add_f32` retires about one lane per two cycles on this chip
// whatever the addresses (150 cycles per wave instruction: 165 of the kernel's 276 us at 5 M edges, private copies per lane residue
// changed nothing), `ds_add_u64` costs next to nothing (116 us with, 111 us without any atomics).  Scale per workgroup: a first pass over
// its edges takes max |d bias|; max * |alpha| maps to 2^36, which leaves 2^26 of headroom for (the other factors of a moment: x, f_o
// differences) x (5 120 edges per workgroup) — values beyond are clamped, a non-finite d bias poisons every sum of the workgroup.
// Integer sums are exact and order-independent; the resolution is 2^-36 of the workgroup's largest term (a float accumulator keeps 2^-24 of the running sum).
__device__ __forceinline__ void nh_add(unsigned long long* h, float v, float scale) {
  const float lim = 9.0e14f;                                    // (2^62 / 5 120 edges)
  // float -> 64-bit integer by hand, on the MAGNITUDE: high word = floor(|x| / 2^32), low word = the remainder, both exactly
  // representable (|x| has 24 significant bits: the remainder is either |x| itself or a multiple of the ulp of |x| >= 2^32, below 2^32)
  // — two conversions and an fma instead of the dozen instructions of the generic (long long) cast, eight times per edge; a negative
  // value is added as the two's complement of its magnitude.  (Until round 5 the split ran on the signed value: for x < 0 the
  // remainder lay in [2^31, 2^32) or rounded to 2^32 itself — inexact by up to 2^-28 of the term, and the conversion of 2^32 to
  // unsigned relied on v_cvt_u32_f32 saturating.)
  const float x = fminf(fmaxf(v * scale, -lim), lim);
  const float ax = fabsf(x);
  const float hi = floorf(ax * 2.3283064365386963e-10f);
  const float lo = fmaf(hi, -4294967296.0f, ax);                // in [0, 2^32), exact
  const unsigned long long mag = ((unsigned long long)(unsigned)hi << 32) | (unsigned long long)(unsigned)lo;
  atomicAdd(h, x < 0.f ? 0ull - mag : mag);
}
__global__ __launch_bounds__(256) void k_nab_hist_bwd(const float* __restrict__ pwl, const float* __restrict__ xd,
                                                      const float* __restrict__ xa, const float* __restrict__ gout,
                                                      float* __restrict__ hist, long M) {
  __shared__ __attribute__((aligned(16))) float tab[NH_TAB];
  __shared__ unsigned long long hs[NH_HIST];                    // [moment][family][segment]: the lanes of one atomic spread over the banks by segment
  __shared__ float wmax[4];
  const int t = threadIdx.x;
  for (int i = t; i < NH_TAB; i += 256) tab[i] = pwl[i];
  for (int i = t; i < NH_HIST; i += 256) hs[i] = 0ull;
  float mx = 0.f;
  bool bad = false;
  for (long e = (long)blockIdx.x * 256 + t; e < M; e += (long)gridDim.x * 256) {
    const float a = fabsf(gout[e]);
    bad |= !(a < INFINITY);
    mx = fmaxf(mx, a);
  }
  mx = rr_wave_max(bad ? INFINITY : mx);
  if ((t & 63) == 0) wmax[t >> 6] = mx;
  __syncthreads();
  const float bg = tab[256 + 1032], bo = tab[256 + 1033], alpha = tab[256 + 1034];
  mx = fmaxf(fmaxf(wmax[0], wmax[1]), fmaxf(wmax[2], wmax[3]));
  const bool poison = !(mx < INFINITY) || !(fabsf(alpha) < INFINITY);
  int ex = 0;
  (void)frexpf(poison ? 0.f : mx * fabsf(alpha), &ex);          // mx |alpha| < 2^ex
  const int sh = min(36 - ex, 120);
  const float scale = ldexpf(1.0f, sh), inv = ldexpf(1.0f, -sh);
  float s_alpha = 0.f;
  for (long e = (long)blockIdx.x * 256 + t; e < M; e += (long)gridDim.x * 256) {
    const float x[2] = {fminf(xd[e], 3.0e38f), fminf(xa[e], 3.0e38f)};
    const float G = gout[e];
    int m[2]; float fo[2], fg[2];
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      m[f] = nh_segment(tab + 128 * f, x[f]);
      float anchor = tab[128 * f + (m[f] > 0 ? m[f] - 1 : 0)];
      anchor = anchor < INFINITY ? anchor : 0.f;
      const float4 sg = rr_ld4(tab + 256 + 516 * f + 4 * m[f]);
      const float dx = x[f] - anchor;
      fo[f] = fmaf(sg.x, dx, sg.y); fg[f] = fmaf(sg.z, dx, sg.w);
    }
    const float gt = rr_sigmoid(fg[0] + fg[1] + bg);             // (the encoder kernel's own form: nab_edge4_grid)
    const float val = gt * fo[0] + (1.0f - gt) * fo[1] + bo;
    const float Gv = G * alpha;
    const float wo[2] = {Gv * gt, Gv * (1.0f - gt)};
    const float wg = Gv * (fo[0] - fo[1]) * gt * (1.0f - gt);
    s_alpha = fmaf(G, val, s_alpha);
#pragma unroll
    for (int f = 0; f < 2; ++f) {
      unsigned long long* h = hs + 129 * f + m[f];
      nh_add(h, wo[f], scale); nh_add(h + 258, wo[f] * x[f], scale); nh_add(h + 516, wg, scale); nh_add(h + 774, wg * x[f], scale);
    }
  }
  __syncthreads();
  for (int i = t; i < NH_HIST; i += 256) {
    const float tot = (float)(long long)hs[(i & 3) * 258 + (i >> 2)] * inv;
    atomicAdd(hist + i, poison ? NAN : tot);
  }
  s_alpha = rr_wave_sum(s_alpha);
  if ((t & 63) == 0) atomicAdd(hist + NH_HIST, s_alpha);
}

extern "C" int rr_nab_hist_bwd(const float* pwl, const float* xd, const float* xa, const float* gout, float* hist, long M,
                               hipStream_t st) {
  if (pwl == nullptr || xd == nullptr || xa == nullptr || gout == nullptr || hist == nullptr || M <= 0) return RR_EINVAL;
  const long want = (M + 255) / 256;
  // (1 024 workgroups: 295 -> 258 us at 5 M edges against 512; 4 096 pay for their histograms' global atomics: 375 us)
  hipLaunchKernelGGL(k_nab_hist_bwd, dim3((unsigned)(want < 1024 ? want : 1024)), dim3(256), 0, st, pwl, xd, xa, gout, hist, M);
  return rr_check(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// Chain rule from the per-segment moments of k_nab_hist_bwd to the PARAMETERS of DistAngleFusion (attn_freenet.py:201-289), for
// every block of the encoder in one launch: what `nab_grad_from_hist` (d loss / d folded table) followed by torch autograd through the
// fold did until round 5.  The folded table of a family: a_k = W0[k], b_k = b0[k], co_k = (W2^T wo)_k, cg_k = (W2^T wg_f)_k, scalars
// wo . b2, wg_f . b2, bg, bo, alpha (models/grad_replay._nab_table).  Unit k is active behind its breakpoint t_k = -b_k / a_k when
// a_k > 0 and up to it when a_k < 0 (rank of t_k among the family's sorted breakpoints = the segment index the histogram kernel
// counted in): A_k = the four moments summed over the unit's active segments (float64 prefix sums over 129 segments).
//   d a = co A1 + cg A3,  d b = co A0 + cg A2,  d co = a A1 + b A0,  d cg = a A3 + b A2          (A0..3 = sum w_o, w_o x, w_g, w_g x)
//   d W2[i][j] = wo_i d co_j + wg_i d cg_j,  d wo_i += (W2 d co)_i + b2_i T0,  d wg_i = (W2 d cg)_i + b2_i T2,
//   d b2_i = wo_i T0 + wg_i T2,  d bg = T2 (distance family),  d bo = T0_d + T0_a,  d alpha = the moment the histogram kernel summed.
// tbl [nb][26] int64: 13 parameter addresses (per family .0.weight, .0.bias, .2.weight, .2.bias; out_lin.weight, out_lin.bias,
// gate.0.weight, gate.0.bias, alpha) and 13 offsets (floats) of their gradients in `gflat`; grid (block, family) x 128 threads.
// The gradient buffers are the step's zero-filled accumulators; the two families of a block meet in d wo / d bo only (two addends:
// order-independent).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(128) void k_nab_tab_bwd(const long long* __restrict__ tbl, const float* __restrict__ hist,
                                                     float* __restrict__ gflat) {
  const int blk = blockIdx.x, f = blockIdx.y, k = threadIdx.x;
  const long long* T = tbl + (size_t)blk * 26;
  const float* W0 = reinterpret_cast<const float*>(T[4 * f + 0]);
  const float* b0 = reinterpret_cast<const float*>(T[4 * f + 1]);
  const float* W2 = reinterpret_cast<const float*>(T[4 * f + 2]);
  const float* b2 = reinterpret_cast<const float*>(T[4 * f + 3]);
  const float* wo = reinterpret_cast<const float*>(T[8]);
  const float* wg = reinterpret_cast<const float*>(T[10]) + RR_E * f;
  float *gW0 = gflat + T[13 + 4 * f + 0], *gb0 = gflat + T[13 + 4 * f + 1], *gW2 = gflat + T[13 + 4 * f + 2], *gb2 = gflat + T[13 + 4 * f + 3];
  float *gwo = gflat + T[13 + 8], *gbo = gflat + T[13 + 9], *gwg = gflat + T[13 + 10] + RR_E * f, *gbg = gflat + T[13 + 11], *galpha = gflat + T[13 + 12];
  __shared__ double ts[RR_E];
  __shared__ double C[129][4];
  __shared__ float wos[RR_E], wgs[RR_E], gco[RR_E], gcg[RR_E];
  const float a = W0[k], b = b0[k];
  wos[k] = wo[k]; wgs[k] = wg[k];
  ts[k] = a != 0.f ? -(double)b / (double)a : (double)INFINITY;
  if (k < 4) {                                       // prefix sums of the family's 129 x 4 moments, float64
    const float* H = hist + (size_t)blk * (2 * 129 * 4 + 1) + (size_t)f * 129 * 4;
    double c = 0.0;
    for (int s = 0; s < 129; ++s) { c += (double)H[s * 4 + k]; C[s][k] = c; }
  }
  __syncthreads();
  float co = 0.f, cg = 0.f;
  for (int i = 0; i < RR_E; ++i) { const float w = W2[i * RR_E + k]; co = fmaf(w, wos[i], co); cg = fmaf(w, wgs[i], cg); }
  int rank = 0;
  const double tk = ts[k];
  for (int j = 0; j < RR_E; ++j) rank += (ts[j] < tk || (ts[j] == tk && j < k)) ? 1 : 0;      // position in a stable ascending sort
  double A[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const double tot = C[128][q], cr = C[rank][q];
    A[q] = a > 0.f ? tot - cr : a < 0.f ? cr : (b > 0.f ? tot : 0.0);
  }
  const double da = (double)co * A[1] + (double)cg * A[3], db = (double)co * A[0] + (double)cg * A[2];
  const double dco = (double)a * A[1] + (double)b * A[0], dcg = (double)a * A[3] + (double)b * A[2];
  gW0[k] += (float)da; gb0[k] += (float)db;
  gco[k] = (float)dco; gcg[k] = (float)dcg;
  __syncthreads();
  const float gck = gco[k], ggk = gcg[k];
  for (int i = 0; i < RR_E; ++i) gW2[i * RR_E + k] += fmaf(wos[i], gck, wgs[i] * ggk);
  float so = 0.f, sg = 0.f;
  for (int j = 0; j < RR_E; ++j) { const float w = W2[k * RR_E + j]; so = fmaf(w, gco[j], so); sg = fmaf(w, gcg[j], sg); }
  const float T0 = (float)C[128][0], T2 = (float)C[128][2];
  atomicAdd(gwo + k, fmaf(b2[k], T0, so));
  gwg[k] += fmaf(b2[k], T2, sg);
  gb2[k] += fmaf(wos[k], T0, wgs[k] * T2);
  if (k == 0) {
    atomicAdd(gbo, T0);
    if (f == 0) { *gbg += T2; *galpha += hist[(size_t)blk * (2 * 129 * 4 + 1) + 2 * 129 * 4]; }
  }
}
extern "C" int rr_nab_tab_bwd(const long long* tbl, const float* hist, float* gflat, int nb, hipStream_t st) {
  if (tbl == nullptr || hist == nullptr || gflat == nullptr || nb <= 0) return RR_EINVAL;
  hipLaunchKernelGGL(k_nab_tab_bwd, dim3(nb, 2), dim3(128), 0, st, tbl, hist, gflat);
  return rr_check(hipGetLastError());
}

extern "C" int rr_nab_train_fwd(const float* tab, const float* xd, const float* xa, float* out, long M, hipStream_t st) {
  if (tab == nullptr || xd == nullptr || xa == nullptr || out == nullptr || M <= 0) return RR_EINVAL;
  const long want = (M + TR_THREADS - 1) / TR_THREADS;
  hipLaunchKernelGGL(k_nab_train_fwd, dim3((unsigned)(want < 4096 ? want : 4096)), dim3(TR_THREADS), 0, st, tab, xd, xa, out, M);
  return rr_check(hipGetLastError());
}

extern "C" int rr_nab_train_bwd(const float* tab, const float* xd, const float* xa, const float* gout, float* grad_tab, long M,
                                hipStream_t st) {
  if (tab == nullptr || xd == nullptr || xa == nullptr || gout == nullptr || grad_tab == nullptr || M <= 0) return RR_EINVAL;
  const long want = (M + TR_THREADS - 1) / TR_THREADS;
  hipLaunchKernelGGL(k_nab_train_bwd, dim3((unsigned)(want < 2048 ? want : 2048)), dim3(TR_THREADS), 0, st, tab, xd, xa, gout, grad_tab, M);
  return rr_check(hipGetLastError());
}
