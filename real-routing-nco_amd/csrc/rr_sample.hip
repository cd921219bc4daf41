// Neighbour sampling of the init embeddings (rrnco/models/env_embeddings/atsp.py:55-67, rcvrp.py:170-182): per node i,
// `sample_size` distinct neighbours j drawn without replacement with probability proportional to 1 / (d_ij + 1e-6) (the
// diagonal counted as d = 1e6) — torch.multinomial(prob, k, replacement=False) on [B*N, N] in the reference, on EVERY
// forward.  Sequential sampling without replacement from weights w is the Plackett-Luce law, which is also the law of the k
// largest of  log w_j + G_j  with independent Gumbel noise G_j ("Gumbel top-k"): one pass over the row, keyed counter-based
// noise (seed, row, j), no sort.  Only the SET matters downstream (the sampled distances are sorted, atsp.py:76-80); the
// indices come out in draw order all the same.
#include "rr_common.h"

// 16 lanes per row (4 rows per wave), keys j = lane16 + 16 q
__global__ __launch_bounds__(256) void k_sample_neighbors(const float* __restrict__ D, int64_t* __restrict__ out, long rows,
                                                          int N, int K, unsigned long long seed) {
  const int lane16 = threadIdx.x & 15;
  const long row = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
  const bool rv = row < rows;
  const long rc = rv ? row : rows - 1;
  const int i = (int)(rc % N);
  const float* d = D + rc * N;
  float key[7];
#pragma unroll
  for (int q = 0; q < 7; ++q) {
    const int j = lane16 + 16 * q;
    if (j < N) {
      const float pd = (j == i) ? 1e6f : d[j];
      key[q] = -rr_log(pd + 1e-6f) + rr_gumbel(seed, (uint32_t)rc, (uint32_t)(rc >> 32), (uint32_t)j);
    } else key[q] = -INFINITY;
  }
  for (int t = 0; t < K; ++t) {
    float bv = key[0]; int bi = lane16;
#pragma unroll
    for (int q = 1; q < 7; ++q) { const bool b = key[q] > bv; bv = b ? key[q] : bv; bi = b ? lane16 + 16 * q : bi; }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      const float ov = __shfl_xor(bv, o); const int oi = __shfl_xor(bi, o);
      const bool take = ov > bv || (ov == bv && oi < bi);
      bv = take ? ov : bv; bi = take ? oi : bi;
    }
    if (rv && lane16 == 0) out[row * K + t] = bi;
#pragma unroll
    for (int q = 0; q < 7; ++q) key[q] = (lane16 + 16 * q == bi) ? -INFINITY : key[q];      // drawn: out of the urn
  }
}

// out [Bp][N][K] int64.  D [Bp][N][N] (normalised distances).  K <= N - 1.
extern "C" int rr_sample_neighbors(const float* D, int64_t* out, int Bp, int N, int K, unsigned long long seed, hipStream_t st) {
  if (D == nullptr || out == nullptr || Bp <= 0 || N < 2 || N > 112 || K < 1 || K > N) return RR_EINVAL;
  const long rows = (long)Bp * N;
  hipLaunchKernelGGL(k_sample_neighbors, dim3((unsigned)((rows * 16 + 255) / 256)), dim3(256), 0, st, D, out, rows, N, K, seed);
  return rr_check(hipGetLastError());
}
