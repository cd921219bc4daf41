/* C ABI of the MI355X (gfx950) construction-rollout hot path — librrnco_hip.so.
 *
 * Plain pointers and sizes only; every pointer is a DEVICE pointer unless marked host.  Launchers never
 * allocate, never synchronise, enqueue on `stream` and return 0 (RR_OK), -1 (invalid argument) or -2 (HIP launch
 * error).  Each entry point cites the reference code (relative to the reference repo root) it replaces.
 *
 * Layout conventions: all tensors row-major contiguous, float32 / int64 / uint8(bool) exactly as the reference's
 * torch tensors.  Rollout index r = s*Bp + b (start-major over the Bp encoder instances; rl4co `batchify`).
 * Per-instance data (distance matrices, demands) is never replicated per start: rollout r reads instance r % Bp.
 */
#ifndef RRNCO_HIP_H
#define RRNCO_HIP_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t;

/* ---- packed-weight descriptors (built by rrnco_amd/packing.py; host structs holding device pointers) ---- */
typedef struct {                /* one AttnFree_Block: rrnco/models/nn/attn_freenet.py:360-441 */
  const float *n1g, *n1b, *n2g, *n2b, *n3g, *n3b, *f1g, *f1b, *f2g, *f2b;   /* InstanceNorm1d affine [128] */
  const void *wq, *wk, *wv, *wp, *wc, *w1, *w2;   /* MFMA A-operand packs [M/16][K/16][64 lanes][4]; wp = Wc Wp (project and multi_head_combine folded), wc unused */
  const float *bq, *bk, *bv, *bp, *bc, *b1, *b2;
  const float *nab;             /* folded DistAngleFusion (:201-289): piecewise-linear tables, packing.fold_nab_pwl */
  const void *w1s, *w2s;        /* optional two-piece fp16 splits of w1 / w2 (packing.pack_a_f16x2): FFN on the fp16 pipe (default; RR_MLP_SPLIT=0 turns it off) */
  const void *wqs, *wks, *wvs, *wps;   /* likewise for the four 128 x 128 projections (all six or none) */
  const float *muk;             /* [128] mean over the nodes of to_k(norm2(y)) = Wk n2.beta + bk (norm2's output has mean beta exactly): the shift
                                 * of the node softmax in rr_enc_layer_split (softmax is shift-invariant; :319-321); may be NULL for rr_enc_layer */
} EncBlockW;

typedef struct {                /* ATSPInitEmbedding (rrnco/models/env_embeddings/atsp.py:5-121) and
                                 * RVRPInitEmbedding / RVRPTWInitEmbedding (rcvrp.py:5-200, rcvrptw.py) */
  const float *wi, *bi, *wr, *br, *wcl, *bcl;   /* wr / wcl: row_embed / col_embed weights transposed to [SS][E] */
  const void *g0r, *g0c;        /* gating_fc.0 packs [16][16][64][4] */
  const float *g0rb, *g0cb, *g2r, *g2c;
  const float *wdep, *bdep, *wdm, *bdm;   /* VRP: depot Linear(2,E), demand_init Linear(F,E) */
  const void *cmr, *cmc;        /* VRP: combine_{row,col}_embed packs [8][16][64][4] */
  const float *cmrb, *cmcb;
  float g2rb, g2cb;
  int nfeat;
  /* gating_fc.0 folded through the two embeddings it reads (packing.fold_init_gate), all six or none (none: the fp32 kernels):
     gf = (W0[:, E:2E] W_dist)^T zero-padded [32][2E]; gn = (W0[:, 0:E] W_node | constants) [2E][4]; gd = gn for the VRP depot (NULL for ATSP) */
  const float *gfr, *gfc, *gnr, *gnc, *gdr, *gdc;
} InitW;

typedef struct {                /* folded DistAngleFusion(use_duration_matrix=True): attn_freenet.py:226-237, 265-286 */
  const void *mp;               /* pack [9][24][64][4]: [M_d|M_a|M_t] (128x384) + rows co_d/co_a/co_t */
  const float *ab, *cg, *wg2;   /* first-layer slopes/offsets [2][384], gate constant [128], gate.2 weight [3][128] */
  float bg2[3], ko[3], inv_tau, bo, alpha;
  const float *pwl;             /* vector-valued piecewise-linear tables of the gate pre-activation (packing.fold_nab_dur_pwl) */
} NabDurW;

typedef struct {                /* rrnco/models/decoder.py:214-232 + the step-context tables */
  const void *wk, *wv, *wl, *wca, *wcb;       /* pack_a fragments */
  const void *wks, *wvs, *wls, *wcas, *wcbs;  /* optional: the same fragments as [hi | lo'] fp16 pairs (packing.f16x2_image): fp16 pipe */
} CacheW;

typedef struct {                /* pointer MLP + inductive-bias scalars: rrnco/models/decoder.py:186-198, 272-277 */
  const void *w1, *w2; const float *b1, *b2, *q0, *wstate; float alpha, beta;
  const void *w1s, *w2s;        /* optional two-piece fp16 images of 2^6 w1 / 2^6 w2 (packing.pack_a_f16u) for the split rollout (RolloutIO.use_split) */
  const float* b1s;             /* 2^6 b1: the hidden layer of the split rollout lives at the scale of the w1s image */
} DecW;

typedef struct {                /* arguments of the persistent rollout: see csrc/rr_decode.hip */
  const float *K, *Vt, *L, *ctxA, *ctxB, *D, *Dur, *demand, *tw, *service;
  int64_t *cur, *first; uint8_t *mask, *visited; float *used, *vcap, *ctime, *rlen; uint8_t *done;
  int64_t *actions; float *logp, *logits_out; const int64_t *actions_in; int *steps_out;
  int Bp, N, S, T, t0, nsteps, mode, use_placeholder, set_first, write_state, logits_only, stagger;
  float tanh_clip, temperature; unsigned long long seed;
  /* MTVRP variants (backhauls, open routes, distance limits; rmtvrp/env.py:343-428, env_embeddings/context.py:51-70);
   * all NULL = vrptw preset.  used_b / open_route / dist_limit feed the decoder context; the in-kernel env.step of a
   * multi-step launch also needs demand_b and bclass (a logits_only launch does not). */
  float *used_b;                /* [R] used_capacity_backhaul (state, written back with write_state) */
  const uint8_t *open_route;    /* [Bp] */
  const float *dist_limit;      /* [Bp] (+inf = none) */
  const float *demand_b;        /* [Bp][N] demand_backhaul incl. the depot zero */
  const int32_t *bclass;        /* [Bp] backhaul class 1 / 2 */
  /* training dump (NULL / 0 otherwise), row m = (b*dumpT + step)*S + s per decoder evaluation: pointer-MLP input g0 and
   * output g [m][128], meta [m][8] = 4 action-mask words seen, node decided at, node chosen, live flag, 0; VRP state
   * scalars scal [m][4] (context.py:51-70).  Consumed by rr_dec_* below (the REINFORCE backward of decoder.py:151-329).
   * g0 and g must hold Bp*dumpT*S + 1 rows (the last one is a trash row for lanes without a live rollout). */
  float *dump_g0, *dump_g; uint32_t *dump_meta; float *dump_scal; int dumpT;
  int use_split;                /* 1: greedy / sampling launch on the fp16 matrix pipe with two-piece split fp32 operands (x~ = hi + lo,
                                 * three f16 MFMAs per product, error of a dot product 4e-8 of sum |a b|: csrc/rr_common.h); needs DecW.w1s /
                                 * w2s and the three images below, otherwise the fp32 MFMA kernel runs.
                                 * 2: "16-mixed" — the reference's own GPU arithmetic mode (torch.autocast in test.py:183, Lightning
                                 * precision 16-mixed in configs/trainer/default.yaml:8, fp32 logits per rrnco/models/decoder.py:195-196):
                                 * ONE fp16 piece per operand (the hi halves of the same images / packs), fp32 accumulation; instance-mode
                                 * launches (7 rollout tiles per instance) only, other shapes run as use_split = 1 */
  const void *Ks, *Vts, *Ls;    /* rr_pack_f16x2 images of K / Vt / L (same shapes) */
  int* status;                  /* optional device word: bit 2 is set when a split launch meets a non-finite log-probability — an
                                 * operand left the fp16 range somewhere upstream (|x| >= 65504 after its image's scale); the caller
                                 * then repeats the call on the fp32 kernels (use_split = 0) */
  int top_k; float top_p;       /* process_logits' filters (rrnco/models/decoding.py:37-63, 352-358) applied inside the rollout, top-k first:
                                 * 0 / 0.0 (or top_k >= N, top_p >= 1) = off.  Served by the two-piece greedy / sampling kernels
                                 * (use_split = 1, mode 0 / 1); any other launch with a filter set returns RR_EINVAL — the per-step
                                 * loop (rr_select) carries the same filters */
  int tail_pack;                /* 1: the S % 16 left-over rollouts of 16 / (S % 16) consecutive instances share one tile (off by default) */
  int no_inst;                  /* 1: never the instance-mode kernel (A/B measurements); the launcher itself reads no environment variable */
} RolloutIO;

/* Backward of the Neural Adaptive Bias with the duration matrix (rrnco/models/nn/attn_freenet.py:226-237, 265-286) in its folded
 * form: h_f = relu(a_f x_f + b_f) (f = distance, angle, duration; 128 units each), z = Mcat h + cg, gate = softmax((Wg2 silu(z) +
 * bg2) / tau), bias = sum_f gate_f (co_f . h_f + ko_f) + bo, out = alpha bias.  Given d loss / d out per edge: grads [1680] =
 * d a [384] | d b [384] | d co [384] | d cg [128] | d Wg2 [3][128] | d bg2 [3] | d ko [3] | d (1/tau) | d bo | d alpha, and
 * dmcat [128][384]; both are ADDED to (zero them first).  dzf: scratch of ceil(M / 32) * 32 * 128 floats. */
typedef struct {
  const float *a, *b, *co, *cg, *wg2, *scal;   /* scal = bg2[3], ko[3], 1/tau, bo, alpha */
  const void *mcat, *mcatT;                    /* pack_a(Mcat [128][384]), pack_a(Mcat^T [384][128]) */
  const void *mcat_s, *mcatT_s;                /* optional packing.pack_bf16x2 of the same two: the kernels then run on the bf16 pipe with
                                                * two-piece split operands (error 2^-16 of a product); NULL: fp32 MFMA */
} NabDurBwdW;
int rr_nabdur_bwd(const NabDurBwdW* w, const float* xd, const float* xa, const float* xt, const float* gout, float* dzf,
                  float* grads, float* dmcat, long long M, hipStream_t stream);

/* fp32 -> two-piece fp16 image of 2^4 x: every group of four values becomes the four hi = fp16(2^4 x) and the four
 * lo = fp16(2^4 x - hi) halves at the same byte offset (n_floats % 4 == 0; csrc/rr_common.h, second form).  Used for the K / Vt / L
 * operands of the split rollout (rrnco/models/decoder.py:214-232 products).  status (optional device word): bit 0 is set when a
 * value is non-finite or leaves the fp16 range (|2^4 x| >= 65504). */
int rr_pack_f16x2(const float* src, void* dst, long long n_floats, int* status, hipStream_t stream);

/* The top-k / top-p filters of process_logits (rrnco/models/decoding.py:37-63, 352-358; top-k first) on rows of processed logits
 * [R][N] (masked keys -inf, already divided by the temperature): out = the rows with every removed key at -inf.  The same device
 * functions the fused rollout applies in registers when RolloutIO.top_k / top_p are set (N <= 112). */
int rr_filter_rows(const float* logits, float* out, int R, int N, int top_k, float top_p, hipStream_t stream);

/* ATSPEnv._reset / RCVRPEnv._reset / RMTVRPEnv._reset min-max normalisation
 * (rrnco/envs/atsp/env.py:113-120, rcvrp/env.py:137-146, rmtvrp/env.py:289-300): out = (in-min)/(max-min+1e-6)
 * per matrix of M elements; mn/mx [B]. */
int rr_minmax_normalize(const float* in, float* out, float* mn, float* mx, int B, int M, hipStream_t stream);

/* Neighbour sampling of the init embeddings (rrnco/models/env_embeddings/atsp.py:55-67, rcvrp.py:170-182):
 * torch.multinomial(1 / (d + 1e-6), K, replacement=False) per node (diagonal as d = 1e6) as a Gumbel top-K with counter-based
 * noise keyed by (seed, row, j).  out [Bp][N][K] int64, in draw order. */
int rr_sample_neighbors(const float* D, int64_t* out, int Bp, int N, int K, unsigned long long seed, hipStream_t stream);

/* ATSPEnv._step (rrnco/envs/atsp/env.py:80-105): mask_out[r, action[r]] = 0, done[r] = no node left. */
int rr_atsp_step(const int64_t* action, const uint8_t* mask_in, uint8_t* mask_out, uint8_t* done,
                 int R, int N, hipStream_t stream);

/* RCVRPEnv._step + get_action_mask (rrnco/envs/rcvrp/env.py:90-122, 183-195); N = customers. */
int rr_rcvrp_step(const int64_t* action, const float* demand, const float* vcap, float* used, uint8_t* visited,
                  uint8_t* mask, int64_t* cur_out, uint8_t* done, int R, int Bp, int N, hipStream_t stream);

/* _get_reward: mode 0 closed tour (rrnco/envs/atsp/env.py:192-211), mode 1 depot-prefixed route list
 * (rrnco/envs/rcvrp/env.py:197-219, rmtvrp/env.py:430-455).  norm_out = -sum, real_out de-normalised. */
int rr_tour_cost(const float* D, const int64_t* actions, const float* mn, const float* mx, float* norm_out,
                 float* real_out, int R, int Bp, int N, int T, int mode, const uint8_t* open_route /* [Bp] or NULL:
                 rmtvrp/env.py:433 zeroes the arcs into the depot of open routes */, hipStream_t stream);

/* process_logits + Greedy/Sampling/Evaluate._step + logp gather (rrnco/models/decoding.py:311-361, 272-298, 266).
 * mode 0 greedy, 1 sampling (inverse CDF over the keys in ascending order on one counter-based uniform per (seed, row, step)), 2 evaluate (action_in).
 * top_k > 0 / 0 < top_p < 1: the filters of decoding.py:37-63 applied after masking and temperature (0 = off). */
int rr_select(const float* logits, const uint8_t* mask, const int64_t* action_in, int64_t* action_out,
              float* logp_out, float* logp_all, int R, int N, float tanh_clip, float temperature, int mode,
              uint64_t seed, uint32_t step, int top_k, float top_p, hipStream_t stream);

/* One Attn_Free_Layer = row block + col block (rrnco/models/nn/attn_freenet.py:472-488). */
/* theta [Bp][N][N] = rr_edge_angles(locs), or bias_pre [Bp][2][N*N] = the rr_nab_dur / rr_nab_simple output when the bias is
 * evaluated by a kernel of its own (duration matrix, ablation biases): one of the two is required (RR_EINVAL otherwise).
 * dbg: reserved, must be NULL (the stage dumps of the first-generation kernel left the library in round 4; the stage tensors of
 * rr_enc_layer_train serve the per-stage parity test). */
int rr_enc_layer(const EncBlockW* wrow, const EncBlockW* wcol, const float* row_in, const float* col_in,
                 float* row_out, float* col_out, const float* D, const float* locs, const float* theta,
                 const float* bias_pre, int Bp, int N, int norm_affine_only /* Normalization (attn_freenet.py:78-116) of the five norms:
                 0 InstanceNorm1d; 1 per-feature affine maps (BatchNorm1d in eval mode, running statistics folded into n*g / n*b
                 by the caller); 2 "layer" (one mean / unbiased variance per instance, no affine); 3 RMSNorm (n*g = weight) */,
                 float* dbg, hipStream_t stream);

/* The same layer (inference, InstanceNorm1d, two-piece weight images, 64 < N <= 103) re-cut for occupancy into three
 * launches (csrc/rr_enc_split.inc): K / V of both blocks as one row-parallel GEMM over all rows (attn_freenet.py:314-315, 422),
 * the AFT mixing per (instance, block) with two workgroups per CU (:318-323), and the rest of the block on a loader-fed FFN
 * kernel (:313, 324-325, 421, 435-441, 355-356).  Results are bit-identical to rr_enc_layer's.
 * stats_in [2][Bp][2][128]: per (tensor 0 row | 1 col, instance): mean[128], rsqrt(var + 1e-5)[128] of the layer's INPUT over the
 * node axis (what Normalization("instance"), :84, 104-105, derives first): rr_enc_stats for the init embedding, stats_out of the
 * previous layer afterwards (NULL: not written).  work: 6 * Bp * N * 128 floats of scratch (K, V, mixing ratio of both blocks).
 * dist_family / n_base (optional): for a batch of Bp / n_base augmentation copies of n_base base instances (instance copy * n_base + b;
 * StateAugmentation, transforms.py:142-154: the matrices are replicated, only the coordinates differ) the distance family of the folded
 * NAB, looked up once per base instance by rr_nab_dist_family -> [n_base][2 blocks][N*N][2]; NULL: every instance looks it up itself.
 * status (optional device word, the policy's range guard): the K projection leaves as exp(K - mean over the nodes) — the reference's
 * softmax subtracts the MAXIMUM (:319-321) — so a K more than ~88 above its node mean overflows; with a status word the exponent is
 * clamped at 80 and bit 0 is raised (the caller repeats the call on rr_enc_layer, which subtracts the maximum); NULL: inf / NaN results. */
int rr_enc_stats(const float* row, const float* col, float* stats, int Bp, int N, hipStream_t stream);
int rr_enc_layer_split(const EncBlockW* wrow, const EncBlockW* wcol, const float* row_in, const float* col_in,
                       float* row_out, float* col_out, const float* D, const float* theta, const float* bias_pre,
                       const float* stats_in, float* stats_out, float* work, const float* dist_family, int n_base,
                       int Bp, int N, int* status, hipStream_t stream);
int rr_nab_dist_family(const EncBlockW* wrow, const EncBlockW* wcol, const float* D, float* out, int B, int N, hipStream_t stream);

/* theta[b][i][j] = atan2(y_i - y_j, x_i - x_j): the angle input of the Neural Adaptive Bias
 * (rrnco/models/nn/attn_freenet.py:262-264), computed once per instance and shared by every encoder block. */
int rr_edge_angles(const float* locs, float* theta, int Bp, int N, hipStream_t stream);

/* Neural Adaptive Bias with the duration matrix for the row and col block of one layer
 * (rrnco/models/nn/attn_freenet.py:226-237, 265-286, x alpha :427-429) -> bias_out [Bp][2][N*N], fed to rr_enc_layer. */
int rr_nab_dur(const NabDurW* wrow, const NabDurW* wcol, const float* D, const float* T, const float* locs,
               float* bias_out, int Bp, int N, hipStream_t stream);

/* rr_nab_dur for a batch of n_aug (= 8) augmentations of Bp / n_aug base instances, augmentation-major as
 * StateAugmentation builds it (rrnco/models/utils/transforms.py:142-154: instance a B + b holds the matrices of base instance b
 * and its own reflected coordinates): distance and duration are read from the first block and their share of the evaluation is
 * done once per edge for all copies.  RR_EINVAL for anything else (callers then use rr_nab_dur).  Same output layout. */
int rr_nab_dur_aug(const NabDurW* wrow, const NabDurW* wcol, const float* D, const float* T, const float* locs,
                   float* bias_out, int Bp, int N, int n_aug, hipStream_t stream);

/* Ablation bias modules, nab_type "heuristic" (kind 0, rrnco/models/nn/attn_freenet.py:119-167) and "naive" (kind 1,
 * :170-199; needs T and locs), for the row and col block of one layer -> bias_out [Bp][2][N*N] (x alpha), fed to
 * rr_enc_layer as bias_pre. */
typedef struct {
  const float *w0, *b0, *w2;    /* naive: mlp.0 weight [E][3], bias [E]; mlp.2 weight [E] */
  float b2, alpha, dw, tw;      /* naive: mlp.2 bias; block alpha; heuristic: distance_weight, duration_weight */
} NabSimpleW;
int rr_nab_simple(const NabSimpleW* wrow, const NabSimpleW* wcol, int kind, const float* D, const float* T,
                  const float* locs, float* bias_out, int Bp, int N, hipStream_t stream);

/* RMTVRPEnv._step + get_action_mask (rrnco/envs/rmtvrp/env.py:155-215, 343-428).  extra == NULL evaluates the vrptw
 * preset (linehaul demands, time windows, closed routes); with extra the backhaul (classes 1 / 2), open-route and
 * distance-limit terms of the multi-task env are active. */
typedef struct {
  const float* demand_b;        /* [Bp][N] demand_backhaul incl. the depot zero */
  float* used_b;                /* [R] used_capacity_backhaul, updated */
  const uint8_t* open_route;    /* [Bp] */
  const float* dist_limit;      /* [Bp], +inf = none */
  const int32_t* bclass;        /* [Bp] backhaul class 1 or 2 */
} MtvrpExtra;
int rr_rmtvrp_step(const int64_t* action, const float* D, const float* T,
                   const float* to_depot_D /* [Bp][N] = D[:, :, 0] contiguous, or NULL (read strided from D) */,
                   const float* to_depot_T /* [Bp][N] = T[:, :, 0], NULL together with to_depot_D */,
                   const float* demand_l, const float* tw,
                   const float* service, const float* vcap, int64_t* cur, float* ctime, float* rlen, float* used_l,
                   uint8_t* visited, uint8_t* mask, uint8_t* done, int R, int Bp, int N, const MtvrpExtra* extra,
                   hipStream_t stream);

/* kind 0: ATSPInitEmbedding.forward (rrnco/models/env_embeddings/atsp.py:69-91);
 * kind 1: RVRPInitEmbedding._embed_with_distance (rcvrp.py:88-102; rcvrptw.py with F=4), node 0 = depot,
 * vfeat [Bp,N,F] = (demand with 0 at the depot[, tw0, tw1, service]).  sidx [Bp,N,SS] int64 is an input. */
int rr_init_embed(const InitW* w, int kind, const float* D, const float* locs, const int64_t* sidx,
                  const float* vfeat, float* row_out, float* col_out, int Bp, int N, int SS, hipStream_t stream);

/* The non-default branches of ATSPInitEmbedding.forward (rrnco/models/env_embeddings/atsp.py:92 and :94-104; no reference config uses them):
 * mode 1 (use_coords, not use_dist): row = col = init_embed(locs) — needs InitW.wi / bi;
 * mode 2 (not use_coords): row = row_embed(D[n, sidx[n, :]]), col = col_embed(D[sidx[n, :], n]), gathers in sample order (unsorted) —
 * needs InitW.wr / br / wcl / bcl ([SS][E] transposed weights) and sidx [Bp,N,SS]. */
int rr_init_embed_plain(const InitW* w, int mode, const float* D, const float* locs, const int64_t* sidx,
                        float* row_out, float* col_out, int Bp, int N, int SS, hipStream_t stream);

/* RRNetDecoder._precompute_cache (rrnco/models/decoder.py:214-232) + per-node step-context tables.
 * Ks / Vts / Ls (optional, all three or none; same shapes and byte offsets as K / Vt / L): the split rollout's two-piece fp16 images
 * (what rr_pack_f16x2 makes of K / Vt / L, bit for bit) written from the accumulators in the same launch; status (optional device
 * word): bit 0 is set when an image value is non-finite or leaves the fp16 range, as rr_pack_f16x2 does. */
int rr_dec_cache(const CacheW* w, const float* row_emb, const float* col_emb, float* K, float* Vt, float* L,
                 float* ctxA, float* ctxB, void* Ks, void* Vts, void* Ls, int* status, int Bp, int N, hipStream_t stream);

/* The decode loop of RRNetPolicy.forward (rrnco/models/policy.py:210-228) = RRNetDecoder.forward
 * (decoder.py:151-206) + DecodingStrategy.step (decoding.py:219-270) + env.step, `nsteps` steps in one launch;
 * logits_only = a single pure RRNetDecoder.forward.  prob 0 = ATSP, 1 = RCVRP, 2 = RCVRPTW. */
int rr_rollout(const DecW* w, const RolloutIO* io, int prob, hipStream_t stream);

/* Real-world instance sampling (rrnco/envs/atsp/sampler.py:78-94, rmtvrp/sampler.py:80): B sub-matrices
 * out[b][i][j] = city[idx[b][i]][idx[b][j]] of one city's [M][M] distance (and, if dur != NULL, duration) matrix. */
int rr_submatrix_gather(const float* dist, const float* dur, const int64_t* idx, float* out_dist, float* out_dur,
                        int B, int M, int n, hipStream_t stream);

/* Training side of the gating Neural Adaptive Bias (rrnco/models/nn/attn_freenet.py:242-289) in its folded 128-unit
 * form: tab = rows a_d, b_d, co_d, cg_d, a_a, b_a, co_a, cg_a [8][128] + (ko_d, kg_d, ko_a, kg_a, bg, bo, alpha, 0);
 * xd / xa = distance / angle per edge [M]; out[M] = alpha * bias.  The backward adds d loss / d tab into grad_tab
 * [8*128+8] (caller zeroes it) given gout = d loss / d out; the chain rule back to the module parameters is the caller's. */
int rr_nab_train_fwd(const float* tab, const float* xd, const float* xa, float* out, long M, hipStream_t stream);
int rr_nab_train_bwd(const float* tab, const float* xd, const float* xa, const float* gout, float* grad_tab, long M,
                     hipStream_t stream);

/* ---- hand-written backward of the REINFORCE step's decoder (csrc/rr_train_dec.hip) ------------------------------------
 * Replaces what Lightning autograd does for rrnco/models/rl.py:118-128 through rrnco/models/decoder.py:151-329
 * (RRNetDecoder.forward, RRNet_PointerAttention) and rrnco/models/decoding.py:311-361 (process_logits).  Rows are the
 * decoder evaluations the sampling rollout dumped (RolloutIO::dump_*): instance b owns rows b*seg_stride + t*S + s. */
typedef struct {
  float *g; const uint32_t *meta; const float *L, *Lt, *D, *Dur, *gll;   /* Lt: [Bp][128][112] zero padded; gll [S*Bp]; dead rows of g are zeroed */
  float *dlg, *dg, *logp, *dscal;     /* dlg [rows][112], dg [rows][128], logp [rows], dscal[2] += d alpha, d beta */
  int Bp, N, S, T; long long seg_stride;
  float alpha, beta, tanh_clip, temperature;
  const void* Ls;                     /* rr_pack_f16x2 image of L [Bp][N][128] or NULL */
} DecLogitIO;
/* logits = g L^T / sqrt(E), inductive bias, log(exp + 1e-6), 10 tanh, mask, log-softmax (decoder.py:186-198, 300-302;
 * decoding.py:341-361): the chosen node's log-probability per row, d logits and d g.  With Ls the products run on the fp16 /
 * bf16 matrix pipe from LDS-resident operands (the logits exactly as the split rollout computed them), without it on the fp32 MFMA. */
int rr_dec_logit_bwd(const DecLogitIO* io, hipStream_t stream);

/* C[b][p][q] (+)= sum_m A[b][m][p] B[b][m][q], q < 128: d logit keys (dlg^T g per instance) and the weight gradient of a
 * Linear layer (dY^T X).  msplit == 1, accumulate == 0: plain stores, C is overwritten (ws ignored).  Otherwise, with `ws`
 * (batch * msplit * P * 128 floats, owned by this call until the stream has passed it): the row splits' partials go through ws and
 * one fixed-order reduction that OVERWRITES C when accumulate == 0 and ADDS to C when accumulate != 0 (bit-reproducible); with
 * ws == NULL: float atomics into C for either value of accumulate (caller zeroes C when it wants a plain product). */
int rr_gemm_tn(const float* A, const float* B, float* C, int batch, int Mb, int P, int lda, int ldb, int ldc,
               long long strideA, long long strideB, long long strideC, int msplit, int accumulate, float* ws, hipStream_t stream);

typedef struct { const void *wa1, *wa2, *wb; const float *b1, *b2; } MlpRowsW;   /* packing.pack_mlp_train */
typedef struct { const void *w1n, *w2tn; const float *b1; } MlpWgradW;
/* The 128 -> 512 -> 128 ReLU MLP with residual on rows (pointer MLP decoder.py:272-277, 296; TransformerFFN
 * attn_freenet.py:330-357) on the bf16 matrix pipe with two-piece split fp32 operands:
 * mode 0: out = x + W2 relu(W1 x + b1) + b2;  mode 1: out = dy + W1^T[(W2^T dy) . 1(W1 x + b1 > 0)];
 * modes 2 / 3 = 0 / 1 with ONE bf16 piece per operand: the opt-in "16-mixed" training step (configs/trainer/default.yaml:8).
 * Rows: nseg segments of seg_rows rows, seg_stride rows apart; meta (optional, the rollout's [rows][8] dump): rows whose
 * live flag meta[m][6] is 0 hold no data and are read as zero rows. */
int rr_mlp_rows(const MlpRowsW* w, int mode, const float* X, const float* dY, float* out, const uint32_t* meta,
                int nseg, int seg_rows, long long seg_stride, hipStream_t stream);
/* dW1 [512][128], db1 [512], dW2 [128][512], db2 [128] of that MLP from (x, dy); ADDED to (caller zeroes).  ws: NULL (float atomics)
 * or 64 * 2 * 512 * 128 floats: the row splits' partials of dW1 / dW2, added up in a fixed order. */
int rr_mlp_wgrad(const MlpWgradW* w, const float* X, const float* dY, float* dW1, float* db1, float* dW2, float* db2,
                 const uint32_t* meta, int nseg, int seg_rows, long long seg_stride, float* ws, hipStream_t stream);
/* the same with ONE bf16 piece per operand (16-mixed; fp32 accumulation, fp32 gradients out) */
int rr_mlp_wgrad16(const MlpWgradW* w, const float* X, const float* dY, float* dW1, float* db1, float* dW2, float* db2,
                   const uint32_t* meta, int nseg, int seg_rows, long long seg_stride, float* ws, hipStream_t stream);

typedef struct {
  const float *dg0; const uint32_t *meta; const float *scal; const int64_t *first;
  const float *K, *V, *Kt, *ctxA, *ctxB, *wstate;     /* Kt [Bp][128][112] zero padded; wstate [nscal][128] */
  float *dK, *dV, *dctxA, *dctxB, *dwstate;           /* [Bp][N][128] written; dwstate [nscal][128] added to */
  int Bp, N, S, T, nscal; long long seg_stride;
} DecAttnIO;
/* Masked multi-head attention of the pointer (decoder.py:281-323) backward: d keys, d values, and d query scattered into
 * the step-context tables (rl4co TSPContext / VRPContext, env_embeddings/context.py:34-70). */
int rr_dec_attn_bwd(const DecAttnIO* io, hipStream_t stream);

/* ---- hand-written backward of the encoder blocks (csrc/rr_train_enc.hip, csrc/rr_enc_w.inc) ---------------------------
 * rr_enc_layer_train = rr_enc_layer (instance norm, gating NAB in-kernel or bias_pre) that also stores, per block, what
 * the backward reads back (attn_freenet.py:417-441): r = norm1(x), c = norm2(y), q, ek = exp(softmax_nodes(k)), v,
 * num = ea @ (ek v), den = ea @ ek, y = sigmoid(q) num / den, o (input of norm3), u1 (input of ffn.norm1), x1 = ffn.norm1(u1)
 * [Bp][N][128] each, and eaT [Bp][112][112] = exp(softmax(alpha * NAB)) transposed. */
typedef struct { float *r, *c, *q, *ek, *v, *num, *den, *y, *o, *u1, *x1, *eaT; } EncSave;
int rr_enc_layer_train(const EncBlockW* wrow, const EncBlockW* wcol, const float* row_in, const float* col_in, float* row_out,
                       float* col_out, const float* D, const float* theta, const float* bias_pre, int Bp, int N,
                       const EncSave* save_row, const EncSave* save_col, hipStream_t stream);
/* InstanceNorm1d backward (attn_freenet.py:84, 104-105) per instance over the node axis; dy = dy1 (+ dy2); statistics are
 * recomputed from x; dgamma / dbeta [128] are ADDED to; accumulate != 0 adds dx to what dx holds. */
int rr_inorm_bwd(const float* x, const float* dy1, const float* dy2, const float* gamma, float* dx, float* dgamma, float* dbeta,
                 int Bp, int N, int accumulate, hipStream_t stream);
/* out[m][:] = x[m][:] W^T (+ bias) (+ out), 128 -> 128, W as packing.pack_a(W) ([8][8][64][4]): Linear forward, or its input
 * gradient with pack_a(W^T); colsum [128] (optional) += sum_m x[m][:] (the bias gradient when x is an output gradient). */
int rr_linear_rows(const void* Wp, const float* bias, const float* X, float* out, long long M, int accumulate, float* colsum,
                   hipStream_t stream);
/* out[m][0..127] = bias + sum_{k<K} X[m][k] W[n][k], K <= 32, X rows ldx floats apart, W [128][K] as nn.Linear stores it: the narrow
 * Linear maps of the init embeddings (coordinates, sorted sampled distances: rrnco/models/env_embeddings/atsp.py:69-91) recomputed
 * for their backward (their weight gradients are rr_gemm_tn products). */
int rr_linear_smallk(const float* X, int ldx, int K, const float* W, const float* bias, float* out, long long M, hipStream_t stream);
/* ContextualGating (atsp.py:108-121) around its scalar gate, forward recomputed and differentiated in one pass over M rows.
 * In: hA | hB = the 256 pre-activations of gating_fc.0 (bias included), w2 [256] / b2 [1] = gating_fc.2, node / dist = the two
 * embeddings the gate mixes, dout = d loss / d (g node + (1 - g) dist).  Out: dh over hA | hB, dnode (+= when acc_node) = g dout,
 * ddist = (1 - g) dout, dw2 [256] and db2 [1] ADDED to; mix (optional) = the forward value g node + (1 - g) dist, which the VRPs'
 * combine layer reads (rcvrp.py:96-101). */
typedef struct { float *hA, *hB; const float *w2, *b2, *node, *dist, *dout; float *dnode, *ddist, *dw2, *db2; long long M; int acc_node; float* mix; } GateBwdIO;
int rr_gate_bwd(const GateBwdIO* io, hipStream_t stream);
/* C[b] = op(A[b]) op(B[b]), row-major [batch][M][K] x [batch][K][N] (transX: the operand is stored transposed), float or double (f64 != 0):
 * the small products of the host-side weight folds (project o multi_head_combine, attn_freenet.py:325, 435) and of their chain rule in
 * the training step — so that no BLAS library is involved in a repack or a REINFORCE step. */
int rr_small_gemm(const void* A, const void* B, void* C, int batch, int M, int N, int K, int transA, int transB, int f64, hipStream_t stream);
typedef struct { const float *dy, *q, *ek, *v, *num, *den, *eaT; float *dq, *dk, *dv, *dbias; int N; } AftBwdIO;
/* AFTFull (attn_freenet.py:309-324) backward per instance from the saved forward tensors: d q, d k, d v [Bp][N][128] and
 * d loss / d (alpha * NAB bias) [Bp][N][N]. */
int rr_aft_bwd(const AftBwdIO* io, int Bp, hipStream_t stream);

/* The same NAB backward in O(1) per edge: pwl = the piecewise-linear table the encoder evaluates (EncBlockW.nab,
 * packing.fold_nab_pwl); hist [2][129][4] (+1) += per family and segment (sum w_out, sum w_out x, sum w_gate, sum w_gate x)
 * and d alpha; the prefix sums that turn the moments into d (folded table) are the caller's (tiny). */
int rr_nab_hist_bwd(const float* pwl, const float* xd, const float* xa, const float* gout, float* hist, long M,
                    hipStream_t stream);
/* From those moments to the gradients of DistAngleFusion's parameters (attn_freenet.py:201-289: dist_emb / angle_emb .0 / .2, out_lin,
 * gate.0, the block's alpha), all nb blocks in one launch: float64 prefix sums over the 129 segments, each unit's active range by the
 * rank of its breakpoint, then the chain rule through the fold (co = W2^T wo, cg = W2^T wg, wo . b2, wg . b2).  tbl [nb][26] int64 on
 * the device: 13 parameter addresses (per family .0.weight, .0.bias, .2.weight, .2.bias; then out_lin.weight, out_lin.bias,
 * gate.0.weight, gate.0.bias, alpha) and the 13 offsets, in floats, of their gradient accumulators in gflat (added to);
 * hist [nb][2 * 129 * 4 + 1] as rr_nab_hist_bwd wrote it. */
int rr_nab_tab_bwd(const long long* tbl, const float* hist, float* gflat, int nb, hipStream_t stream);

/* ---- instances with 104 .. 1 024 nodes (csrc/rr_bign.hip; the reference's generators, rrnco/envs/rcvrp/generator.py:21-37, go to
 * 1 000): the same operators as row-parallel kernels over HBM-resident tensors, the decode loop step by step.  rr_enc_layer /
 * rr_rollout keep one instance's activations on chip (N <= 103).  rr_aft_mix_big / rr_dec_fwd_big hold a row of up to 208 keys in
 * registers and stream longer rows in two or three sweeps (same order of every sum: bit-identical where both forms apply). */
/* Normalization "instance" (attn_freenet.py:84, 104-105) of x (+ res) over the node axis. */
int rr_inorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* out, int Bp, int N, hipStream_t stream);
/* alpha * DistAngleFusion (attn_freenet.py:242-289) per edge from the folded piecewise-linear table (EncBlockW.nab);
 * transpose_d: the col block's D^T (:480-486). out [Bp][N][N]. */
int rr_nab_pwl_fwd(const float* pwl, const float* D, const float* theta, float* out, int Bp, int N, int transpose_d, hipStream_t stream);
/* ekT = exp(softmax_nodes(K))^T, kvT = (ek * V)^T as [Bp][128][NP] (NP = 16 ceil(N/16), zero padded): attn_freenet.py:319-321. */
int rr_colsoftmax_exp(const float* K, const float* V, float* ekT, float* kvT, float* ek_rows, int Bp, int N, int NP, hipStream_t stream);
/* y = sigmoid(q) * (exp(softmax(bias)) @ kv) / (exp(softmax(bias)) @ ek): AFTFull mixing (attn_freenet.py:318-324).
 * ek_rows (above) and num_out / den_out / eaT_out (all three or none; eaT [Bp][112][112] zero-filled by the caller, N <= 112) are the
 * tensors the hand-written block backward reads (csrc/rr_enc_w.inc: EncSave): optional, NULL for inference. */
int rr_aft_mix_big(const float* bias, const float* q, const float* ekT, const float* kvT, float* y, float* num_out, float* den_out,
                   float* eaT_out, int Bp, int N, int NP, hipStream_t stream);
/* Normalization("batch") in TRAIN mode (attn_freenet.py:82-83, 102-103: nn.BatchNorm1d over the flattened M = Bp * N rows, batch
 * statistics): out = (x (+ res) - mean) rstd gamma + beta with the biased batch variance, eps 1e-5; running_mean / running_var (both or
 * neither) move by `momentum` (variance unbiased); sum_out (optional) receives x + res; ws: 256 doubles of scratch.  The backward:
 * dx (+)= gamma rstd (dy - mean(dy) - xh mean(dy xh)) with dy = dy1 (+ dy2), d gamma / d beta added; ws: 512 doubles. */
int rr_bnorm_fwd(const float* x, const float* res, const float* gamma, const float* beta, float* out, float* sum_out, double* ws,
                 float* running_mean, float* running_var, float momentum, long long M, hipStream_t stream);
int rr_bnorm_bwd(const float* x, const float* dy1, const float* dy2, const float* gamma, float* dx, float* dgamma, float* dbeta,
                 double* ws, long long M, int accumulate, hipStream_t stream);
typedef struct {
  const float *K, *Vt, *L, *ctxA, *ctxB, *D, *Dur; const int64_t *cur, *first; const float *scal, *wstate; const uint8_t *mask;
  const void *w1, *w2; const float *b1, *b2; float *logits; int Bp, N, NP, S, nscal; float alpha, beta;
} DecBigIO;
/* RRNetDecoder.forward (decoder.py:151-206, 281-323) for all S*Bp rollouts: logits [R][N] after the inductive-bias transform. */
int rr_dec_fwd_big(const DecBigIO* io, hipStream_t stream);
/* process_logits (incl. the top-k / top-p filters, decoding.py:37-63, 352-358; 0 / 0.0 = off) + greedy / sampling / evaluate
 * (decoding.py:311-361) for rows of up to 1 024 keys. */
int rr_select_big(const float* logits, const uint8_t* mask, const int64_t* action_in, int64_t* action_out, float* logp_out,
                  float* logp_all, int R, int N, float tanh_clip, float temperature, int mode, unsigned long long seed,
                  unsigned int step, int top_k, float top_p, hipStream_t stream);

/* POMO shared-baseline REINFORCE loss, forward half + d loss / d log-likelihood
 * (rrnco/models/rl.py:112-128; in-tree formula rrnco/baselines/routefinder/model.py:182-202). reward / ll / adv /
 * grad_ll are [S*B] with r = s*B + b; bl and partial are [B] workspaces; loss is one float. */
int rr_reinforce_loss(const float* reward, const float* ll, float* adv, float* grad_ll, float* bl, float* partial,
                      float* loss, int B, int S, hipStream_t stream);

/* ---- MatNet baseline encoder ------------------------------------------------------------------------
 * One MatNetLayer (rrnco/baselines/MatNet/encoder.py:148-172): mixed-score cross attention (MixedScoresSDPA :14-92 inside
 * rl4co's MultiHeadCrossAttention: Wq, Wkv = K | V, out_proj, no biases) of the row block on the column embeddings with
 * dmat = D and of the column block on the row embeddings with dmat = D^T (:133-145), each followed by TransformerFFN
 * (norm1(x_old + x), norm2(x + W2 relu(W1 x + b1) + b2), InstanceNorm1d affine).  Head dim 16 (E = 16 heads), E and ff
 * multiples of 256 (configs/experiment/matnet.yaml: 256 / 16 / 512), N <= 112.  Weight matrices are packed A operands
 * (rrnco_amd/packing.pack_a); mix [heads][68] = W1 score row / sqrt(16) | W1 distance row | b1 | W2 | b2, 0, 0, 0. */
typedef struct {
  const void *wq, *wkv, *wo, *w1, *w2;
  const float *b1, *b2;
  const float *n1g, *n1b, *n2g, *n2b;
  const float* mix;
} MatNetSideW;
size_t rr_matnet_workspace_bytes(int Bp, int N, int E, int ff);
int rr_matnet_layer(const MatNetSideW* row_side, const MatNetSideW* col_side, const float* row_in, const float* col_in,
                    float* row_out, float* col_out, const float* D, float* workspace, size_t workspace_bytes,
                    int Bp, int N, int E, int heads, int ff, hipStream_t stream);
/* MatNet init embeddings (env_embeddings/atsp.py:21-34; rcvrp.py:37-81 with use_coords=False) in host-folded form:
 * row[n] = rowv[kind] + rowv[2] * demand, col[n] = slot_t[rand_idx[n]] + colv[kind] + colv[2] * demand (kind 0 depot,
 * 1 customer; rowv / colv [3][E], slot_t [E][E] = col_combine_embed.weight[:, :E]^T, demand [Bp][N-1]); with rowv = colv =
 * NULL: row = 0, col = one-hot at rand_idx (ATSP).  rand_idx [Bp][N] int64 = the reference's rand.argsort(dim=1). */
int rr_matnet_init(const int64_t* rand_idx, const float* demand, const float* rowv, const float* colv, const float* slot_t,
                   float* row, float* col, int Bp, int N, int E, hipStream_t stream);

/* MatNet baseline decoder (rrnco/baselines/MatNet/decoder.py = rl4co AttentionModelDecoder + PointerAttention, 256 wide / 16
 * heads, no graph context): rr_matnet_linear = y[b] = x[b] W^T per instance (cache K | V | L = col_emb W_node^T [Bp][N][3E];
 * context tables ctxA / ctxB = row_emb W_ctx[:, :E]^T / W_ctx[:, E:]^T), w_packed = packing.pack_a(W);
 * rr_matnet_dec_step = one decoder.forward for all R = S * Bp rollouts (r = s * Bp + b): logits [R][N], before process_logits;
 * first == NULL: the placeholder context q0 = W_ctx W_placeholder (nothing visited yet).
 * rr_select_matnet = the baseline's own process_logits (MatNet/decoding.py:316-372: shift by the row maximum, clamp to
 * [-50, -1e-4], log-softmax) + greedy / sampling / evaluate selection; arguments as rr_select. */
int rr_matnet_linear(const void* w_packed, const float* x, float* y, int Bp, int N, int K, int Nout, hipStream_t stream);
int rr_matnet_dec_step(const void* wo_packed, const float* kvl, const float* vt /* [Bp][E][112] = V^T, zero-padded */,
                       const float* ctxA, const float* ctxB, const float* q0,
                       const float* state /* VRP: [R] vehicle_capacity - used_capacity, or NULL */, const float* wstate /* [E] */,
                       const int64_t* first, const int64_t* cur, const uint8_t* mask, float* logits,
                       int Bp, int N, int S, int E, int heads, hipStream_t stream);
int rr_select_matnet(const float* logits, const uint8_t* mask, const int64_t* action_in, int64_t* action_out,
                     float* logp_out, float* logp_all, int R, int N, float tanh_clipping, float temperature,
                     int mode, uint64_t seed, uint32_t step, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
