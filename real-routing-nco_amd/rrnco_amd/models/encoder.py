"""RRNetEncoder — drop-in for rrnco.models.encoder.RRNetEncoder (rrnco/models/encoder.py:80-112).

The module tree exists to own the parameters under the reference's state_dict names (so the published
checkpoints load with `load_state_dict`); `forward` runs the HIP kernels of csrc/rr_encoder.hip.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import _lib as L


class _RMSWeight(nn.Module):     # RMSNorm attn_freenet.py:13-26 -> `.normalizer.weight`
    def __init__(self, E):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(E))


class _Norm(nn.Module):          # Normalization attn_freenet.py:78-116 -> `.normalizer.{weight,bias}` (+ running stats for "batch")
    def __init__(self, E, normalization="instance"):
        super().__init__()
        if normalization == "layer":            # attn_freenet.py:92-93, 106-109: no parameters (the reference stores the string)
            self.normalizer = None
        elif normalization == "rms":            # :13-26: weight only
            self.normalizer = _RMSWeight(E)
        else:
            self.normalizer = nn.BatchNorm1d(E, affine=True) if normalization == "batch" else nn.InstanceNorm1d(E, affine=True)


class _MLP1(nn.Sequential):      # nn.Sequential(Linear(1,E), ReLU, Linear(E,E))  attn_freenet.py:216-225
    def __init__(self, E):
        super().__init__(nn.Linear(1, E), nn.ReLU(), nn.Linear(E, E))


class _DistAngleFusion(nn.Module):   # attn_freenet.py:201-240 (no duration)
    def __init__(self, E):
        super().__init__()
        self.dist_emb, self.angle_emb = _MLP1(E), _MLP1(E)
        self.gate = nn.Sequential(nn.Linear(2 * E, 1), nn.Sigmoid())
        self.out_lin = nn.Linear(E, 1)


class _AFT(nn.Module):           # AFTFull attn_freenet.py:292-307
    def __init__(self, E):
        super().__init__()
        self.to_q, self.to_k, self.to_v, self.project = (nn.Linear(E, E) for _ in range(4))


class _FFN(nn.Module):           # FeedForward attn_freenet.py:524-536
    def __init__(self, E, ff):
        super().__init__()
        self.W1, self.W2 = nn.Linear(E, ff), nn.Linear(ff, E)


class _TransformerFFN(nn.Module):  # attn_freenet.py:330-357
    def __init__(self, E, ff, normalization="instance"):
        super().__init__()
        self.ops = nn.ModuleDict({"norm1": _Norm(E, normalization), "ffn": _FFN(E, ff), "norm2": _Norm(E, normalization)})


class _DistAngleDurFusion(nn.Module):   # attn_freenet.py:201-240 with use_duration_matrix=True
    def __init__(self, E):
        super().__init__()
        self.dist_emb, self.angle_emb, self.dur_emb = _MLP1(E), _MLP1(E), _MLP1(E)
        self.gate = nn.Sequential(nn.Linear(3 * E, E), nn.SiLU(), nn.Linear(E, 3))
        self.gate_temperature = nn.Parameter(torch.tensor(5.0))
        self.out_lin = nn.Linear(E, 1)


class _HeuristicNAB(nn.Module):  # HeuristicNeuralAdaptiveBias attn_freenet.py:119-167 (its own alpha is never used in forward)
    def __init__(self, use_duration):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(1))
        if use_duration:
            self.distance_weight, self.duration_weight = nn.Parameter(torch.ones(1)), nn.Parameter(torch.ones(1))


class _NaiveNAB(nn.Module):      # NaiveNeuralAdaptiveBias attn_freenet.py:170-199
    def __init__(self, E, use_duration):
        super().__init__()
        self.mlp = nn.Sequential(nn.Linear(3 if use_duration else 2, E), nn.SiLU(), nn.Linear(E, 1))


class _Block(nn.Module):         # AttnFree_Block attn_freenet.py:360-415
    def __init__(self, E, ff, use_duration=False, nab_type="gating", normalization="instance"):
        super().__init__()
        self.alpha = nn.Parameter(torch.ones(1))
        self.attn_free = _AFT(E)
        self.multi_head_combine = nn.Linear(E, E)
        if nab_type == "naive":
            self.neural_adaptive_bias = _NaiveNAB(E, use_duration)  # :391-395
        elif nab_type == "heuristic":
            self.neural_adaptive_bias = _HeuristicNAB(use_duration)  # :396-400
        elif nab_type != "gating":
            raise ValueError(f"Unknown nab_type: {nab_type}. Supported types: 'gating', 'naive', 'heuristic'")
        elif use_duration:
            self.neural_adaptive_bias = _DistAngleDurFusion(E)     # :379-383
        else:
            self.angle_distance_fusion = _DistAngleFusion(E)       # :384-389
        self.feed_forward = _TransformerFFN(E, ff, normalization)
        self.norm1, self.norm2, self.norm3 = _Norm(E, normalization), _Norm(E, normalization), _Norm(E, normalization)


class _Layer(nn.Module):         # Attn_Free_Layer attn_freenet.py:444-470
    def __init__(self, E, ff, use_duration=False, nab_type="gating", normalization="instance"):
        super().__init__()
        self.row_encoding_block = _Block(E, ff, use_duration, nab_type, normalization)
        self.col_encoding_block = _Block(E, ff, use_duration, nab_type, normalization)


class AttnFreeNet(nn.Module):    # attn_freenet.py:491-515
    def __init__(self, embed_dim=128, feedforward_hidden=512, num_layers=3, use_duration_matrix=False, nab_type="gating",
                 normalization="instance", **unused):
        super().__init__()
        self.layers = nn.ModuleList([_Layer(embed_dim, feedforward_hidden, use_duration_matrix, nab_type, normalization)
                                     for _ in range(num_layers)])


class _Gating(nn.Module):        # ContextualGating env_embeddings/atsp.py:108-121
    def __init__(self, E):
        super().__init__()
        self.gating_fc = nn.Sequential(nn.Linear(2 * E, 2 * E), nn.ReLU(), nn.Linear(2 * E, 1))


def rank_noise_seed(seed: int, rank: int) -> int:
    """Seed of the device-side neighbour sampler on data-parallel rank `rank`: the ranks share torch.manual_seed (same initial
    weights, hence the same CPU draw), so every rank offsets it by a multiple of the 64-bit golden ratio — distinct noise fields,
    reproducible from (seed, rank) alone (a resumed run restores the CPU generator and continues the same per-rank streams)."""
    return (int(seed) + 0x9E3779B97F4A7C15 * int(rank)) % (2 ** 62)


def shared_random_indices(phase: str, B: int, N: int, sample_size: int, device) -> torch.Tensor:
    """sample_type="random" (env_embeddings/atsp.py:38-54, rcvrp.py:153-169): ONE index row (drawn with replacement) shared by every
    node of every instance in training, one per block of B / 8 instances (the 8 augmentation copies) otherwise.  -> [B, N, sample_size]."""
    if phase == "train":
        r = torch.randint(0, N, (1, sample_size), device=device)
        return r.unsqueeze(1).expand(B, N, sample_size).contiguous()
    r = torch.randint(0, N, (8, 1, sample_size), device=device)
    return r.unsqueeze(1).expand(8, B // 8, N, sample_size).reshape(8 * B // 8, N, sample_size).contiguous()      # (fails as the reference does unless 8 | B)


class ATSPInitEmbedding(nn.Module):   # env_embeddings/atsp.py:5-35
    def __init__(self, embed_dim, linear_bias=True, use_coords=True, use_polar_feats=False, use_dist=True,
                 use_matnet_init=True, sample_type="prob", sample_size=25):
        super().__init__()
        if sample_type not in ("prob", "random"):
            raise ValueError(f"sample_type {sample_type!r}: the reference knows 'prob' and 'random' (atsp.py:38-67)")
        if not linear_bias:
            raise NotImplementedError("rrnco_amd: ATSPInitEmbedding without linear_bias (no reference config)")
        self.use_coords, self.use_dist, self.sample_type, self.sample_size = use_coords, use_dist, sample_type, sample_size
        # same parameters as the reference builds (atsp.py:29-35): init_embed only with use_coords, the gates only with both
        self.init_embed = nn.Linear(2, embed_dim, linear_bias) if use_coords else None
        self.row_embed = nn.Linear(sample_size, embed_dim, linear_bias)
        self.col_embed = nn.Linear(sample_size, embed_dim, linear_bias)
        if use_coords and use_dist:
            self.gating_network_row, self.gating_network_col = _Gating(embed_dim), _Gating(embed_dim)

    def indices_for(self, distance, phase):
        """The neighbour index tensor of one forward (atsp.py:37-67) when the caller supplies none in td["sample_idx"]."""
        if self.sample_type == "random":
            return shared_random_indices(phase, distance.shape[0], distance.shape[1], self.sample_size, distance.device)
        return ATSPInitEmbedding.sample_indices(distance, self.sample_size)

    @staticmethod
    def sample_indices(distance, sample_size):
        """env_embeddings/atsp.py:55-67: multinomial without replacement on 1/(d+1e-6), diagonal 1e6."""
        B, N, _ = distance.shape
        if distance.is_cuda and N <= 112:
            # on the device: Gumbel top-k (the same Plackett-Luce law as multinomial without replacement), one pass, keyed
            # counter-based noise; the seed advances torch's CPU generator like any draw would (csrc/rr_sample.hip)
            out = torch.empty(B, N, sample_size, dtype=torch.int64, device=distance.device)
            import os
            seed = rank_noise_seed(int(torch.randint(0, 2 ** 62, (1,)).item()), int(os.environ.get("RANK", "0")))
            L.check(L.lib().rr_sample_neighbors(L.ptr(distance.float().contiguous()), L.ptr(out), B, N, int(sample_size), seed,
                                                L.stream()), "rr_sample_neighbors")
            return out
        ar = torch.arange(N, device=distance.device)
        pd = distance.clone()
        pd[:, ar, ar] = 1e6
        inv = 1 / (pd + 1e-6)
        prob = (inv / inv.sum(dim=-1, keepdim=True)).reshape(B * N, -1)
        return torch.multinomial(prob, sample_size, replacement=False).reshape(B, N, sample_size)


def draw_sample_indices(init_embedding, distance, phase="val"):
    """The neighbour index tensor a forward draws when td carries no "sample_idx": the init embedding's own law (sample_type "prob":
    multinomial without replacement per node, atsp.py:55-67; "random": shared index rows, atsp.py:38-54)."""
    fn = getattr(init_embedding, "indices_for", None)
    if fn is not None:
        return fn(distance, phase)
    return ATSPInitEmbedding.sample_indices(distance, init_embedding.sample_size)


class RRNetEncoder(nn.Module):
    def __init__(self, embed_dim=128, init_embedding=None, init_embedding_kwargs=None, env_name="rcvrp",
                 num_heads=8, num_layers=3, normalization="batch", feedforward_hidden=512, net=None,
                 sdpa_fn=None, moe_kwargs=None, use_coords=False, use_polar_feats=False, nab_type="gating"):
        super().__init__()
        if embed_dim != 128 or feedforward_hidden != 512:
            raise NotImplementedError("rrnco_amd kernels are specialised for embed_dim=128, feedforward_hidden=512")
        if normalization not in ("instance", "batch", "layer", "rms"):
            raise NotImplementedError(f"Normalization type {normalization} not found")      # attn_freenet.py:94-99 logs and skips
        self.normalization = normalization
        self.env_name = getattr(env_name, "name", env_name)
        kw = dict(init_embedding_kwargs or {})
        if init_embedding is not None:
            self.init_embedding = init_embedding
        elif self.env_name == "atsp":
            self.init_embedding = ATSPInitEmbedding(embed_dim, **kw)
        else:
            from .vrp_embeddings import make_vrp_init_embedding
            self.init_embedding = make_vrp_init_embedding(self.env_name, embed_dim, **kw)
        self.net = AttnFreeNet(embed_dim, feedforward_hidden, num_layers, nab_type=nab_type, normalization=normalization,
                               use_duration_matrix=self.env_name not in ("atsp", "rcvrp")) if net is None else net   # encoder.py:63-66

    def supports_hip_backward(self, packed) -> bool:
        """The hand-written block backward (csrc/rr_train_enc.hip) covers the published configuration: instance norm and the
        gating NAB — without duration (ATSP, RCVRP: moment-histogram kernel) or with it (RCVRPTW: the bias gradient d loss / d bias
        [Bp,N,N] of every block comes from the kernels, the duration NAB itself is differentiated by grad_replay._NabDurationFolded)."""
        return self.normalization == "instance" and packed.get("nab_kind", "gating") == "gating"

    def forward(self, td, phase: str = "val", mask=None, packed=None, train_saves=None, status=None):
        """-> (row_emb, col_emb) [B,N,E].  `packed` = packing.pack_policy(...) (the policy caches it).  `train_saves` (a list):
        training forward — every layer's inputs and the per-block tensors of _lib.EncSave are appended to it.  `status`: the policy's
        range-guard word (int32 device scalar) — the re-cut layer raises bit 0 when exp(K - mean K) would overflow (rr_enc_layer_split)."""
        assert packed is not None, "RRNetEncoder.forward needs packed weights (call through RRNetPolicy or pass packed=)"
        bn = self.normalization == "batch"
        norm_mode = {"instance": 0, "batch": 1, "layer": 2, "rms": 3}[self.normalization]        # rr_enc_layer norm_affine_only
        if bn and self.training:
            raise NotImplementedError("the encoder kernels evaluate normalization='batch' with running statistics (module.eval()); "
                                      "train mode (batch statistics across instances) goes through RRNetPolicy, which runs the "
                                      "encoder with torch ops in that mode (models/grad_replay.encode_for_policy)")
        D = td["distance_matrix"].contiguous()
        L.require_gpu(D)
        import os as _os
        if D.shape[-1] > 103 or _os.environ.get("RR_FORCE_BIGN", "0") == "1":   # more nodes than the on-chip kernels hold: row-parallel kernels (csrc/rr_bign.hip); RR_FORCE_BIGN=1: diagnostic, any N
            from . import bign
            if D.shape[-1] > bign.MAX_N_BIG or not bign.supported(self.env_name, packed, self.normalization) or train_saves is not None:
                raise NotImplementedError(f"{D.shape[-1]} nodes: the encoder kernels cover N <= 103 for every configuration and "
                                          f"N <= {bign.MAX_N_BIG} with instance norm and the gating NAB (inference)")
            return bign.encode(self, td, packed)
        locs = td["locs"].float().contiguous()
        Bp, N = D.shape[0], D.shape[-1]
        dev = D.device
        lib = L.lib()
        row = torch.empty(Bp, N, 128, device=dev, dtype=torch.float32)
        col = torch.empty_like(row)
        ie = self.init_embedding
        draw = lambda d, ph: draw_sample_indices(ie, d, ph)
        if self.env_name == "atsp":
            mode = packed.get("init_mode", 0)           # 0: coordinates + sorted distances + gate; 1: coordinates only; 2: unsorted distances only
            sidx = td.get("sample_idx", None)
            if sidx is None and mode != 1:
                sidx = draw(D, phase)
            sidx = sidx.contiguous() if sidx is not None else None
            if mode == 0:
                L.check(lib.rr_init_embed(packed["init"], 0, L.ptr(D), L.ptr(locs), L.ptr(sidx), None, L.ptr(row), L.ptr(col),
                                          Bp, N, sidx.shape[-1], L.stream()), "rr_init_embed")
            else:
                if train_saves is not None:
                    raise NotImplementedError("training with use_coords=False / use_dist=False init embeddings (inference branches only)")
                L.check(lib.rr_init_embed_plain(packed["init"], mode, L.ptr(D), L.ptr(locs), L.ptr(sidx), L.ptr(row), L.ptr(col),
                                                Bp, N, sidx.shape[-1] if sidx is not None else 0, L.stream()), "rr_init_embed_plain")
        else:
            sidx = td.get("sample_idx", None)
            if sidx is None:
                sidx = draw(D, phase)
            sidx = sidx.contiguous()
            vfeat = self.init_embedding.node_features(td).contiguous()
            L.check(lib.rr_init_embed(packed["init"], 1, L.ptr(D), L.ptr(locs), L.ptr(sidx), L.ptr(vfeat), L.ptr(row),
                                      L.ptr(col), Bp, N, sidx.shape[-1], L.stream()), "rr_init_embed")
        self._last_init = (row, col)
        row2, col2 = torch.empty_like(row), torch.empty_like(col)
        has_dur = self.env_name not in ("atsp", "rcvrp")   # encoder.py:98-106: duration matrix only for rcvrptw
        simple = packed.get("nab_kind", "gating") != "gating"
        use_dur = len(packed["nabdur"]) > 0 or simple         # "bias_pre" path: NAB evaluated by a kernel of its own
        theta = None
        T = td["duration_matrix"].float().contiguous() if has_dur else None
        if use_dur:
            bias = torch.empty(Bp, 2, N * N, device=dev, dtype=torch.float32)
        else:   # the angle matrix only depends on the coordinates: once per instance, shared by all twelve blocks
            theta = torch.empty(Bp, N, N, device=dev, dtype=torch.float32)
            L.check(lib.rr_edge_angles(L.ptr(locs), L.ptr(theta), Bp, N, L.stream()), "rr_edge_angles")
        # headline shape (inference, instance norm, two-piece weight images, 64 < N <= 103): the layer as three launches with less
        # state each (csrc/rr_enc_split.inc); RR_ENC_SPLIT=0 keeps the one-workgroup-per-block kernels (A/B, bit-identical results)
        from .. import packing as _P
        resplit = (train_saves is None and norm_mode == 0 and 64 < N <= 103 and _P.mlp_split_enabled()
                   and bool(packed["blocks"][0][0].wqs) and _os.environ.get("RR_ENC_SPLIT", "1") != "0")
        if resplit:
            stats = torch.empty(2, 2, Bp, 2, 128, device=dev, dtype=torch.float32)       # [ping-pong][tensor][b][mean | rstd][f]
            work = torch.empty(6, Bp, N, 128, device=dev, dtype=torch.float32)
            L.check(lib.rr_enc_stats(L.ptr(row), L.ptr(col), L.ptr(stats[0]), Bp, N, L.stream()), "rr_enc_stats")
            # x8-augmented batch (StateAugmentation's note in td.meta): the matrices of the 8 copies are the base instance's, so the
            # distance family of the folded NAB is looked up once per base instance and layer (rr_nab_dist_family), the angle per copy
            n_aug = getattr(td, "meta", {}).get("num_augment", 1) if hasattr(td, "meta") else 1
            dfam = None
            if (not use_dur and n_aug and n_aug > 1 and Bp % n_aug == 0 and _os.environ.get("RR_ENC_AUGSHARE", "1") != "0"):
                dfam = torch.empty(Bp // n_aug, 2, N * N, 2, device=dev, dtype=torch.float32)
        nl = len(packed["blocks"])
        for l, (wr, wc) in enumerate(packed["blocks"]):
            if simple:
                nr, nc = packed["nabsimple"][l]
                L.check(lib.rr_nab_simple(nr, nc, 1 if packed["nab_kind"] == "naive" else 0, L.ptr(D), L.ptr(T), L.ptr(locs),
                                          L.ptr(bias), Bp, N, L.stream()), "rr_nab_simple")
            elif use_dur:
                nr, nc = packed["nabdur"][l]
                n_aug = getattr(td, "meta", {}).get("num_augment", 1) if hasattr(td, "meta") else 1
                if (n_aug == 8 and Bp % 8 == 0 and N * N >= 2048 and train_saves is None and nr.pwl and nc.pwl
                        and _os.environ.get("RR_NABDUR_AUG", "1") != "0"):
                    # x8 augmentation (StateAugmentation's note): distance and duration are those of the base instance in all 8 copies —
                    # their table rows and piecewise-linear evaluations once per edge (csrc/rr_encoder.hip: k_nab_dur_aug)
                    L.check(lib.rr_nab_dur_aug(nr, nc, L.ptr(D), L.ptr(T), L.ptr(locs), L.ptr(bias), Bp, N, 8, L.stream()), "rr_nab_dur_aug")
                else:
                    L.check(lib.rr_nab_dur(nr, nc, L.ptr(D), L.ptr(T), L.ptr(locs), L.ptr(bias), Bp, N, L.stream()), "rr_nab_dur")
            if train_saves is not None:
                sv = []
                for _ in range(2):
                    d = {n: torch.empty(Bp, N, 128, device=dev, dtype=torch.float32) for n in L.EncSave.NAMES[:-1]}
                    d["eaT"] = torch.empty(Bp, 112, 112, device=dev, dtype=torch.float32)
                    st_ = L.EncSave()
                    for n in L.EncSave.NAMES:
                        setattr(st_, n, L.ptr(d[n]))
                    sv.append((d, st_))
                L.check(lib.rr_enc_layer_train(wr, wc, L.ptr(row), L.ptr(col), L.ptr(row2), L.ptr(col2), L.ptr(D),
                                               L.ptr(theta) if theta is not None else None, L.ptr(bias) if use_dur else None,
                                               Bp, N, sv[0][1], sv[1][1], L.stream()), "rr_enc_layer_train")
                train_saves.append({"row_in": row, "col_in": col, "row": sv[0][0], "col": sv[1][0]})
                row, col, row2, col2 = row2, col2, torch.empty_like(row), torch.empty_like(col)
                continue
            if resplit:
                if dfam is not None:
                    L.check(lib.rr_nab_dist_family(wr, wc, L.ptr(D), L.ptr(dfam), Bp // n_aug, N, L.stream()), "rr_nab_dist_family")
                L.check(lib.rr_enc_layer_split(wr, wc, L.ptr(row), L.ptr(col), L.ptr(row2), L.ptr(col2), L.ptr(D),
                                               L.ptr(theta) if theta is not None else None, L.ptr(bias) if use_dur else None,
                                               L.ptr(stats[l & 1]), L.ptr(stats[1 - (l & 1)]) if l + 1 < nl else None, L.ptr(work),
                                               L.ptr(dfam), Bp // n_aug if dfam is not None else 0, Bp, N, L.ptr(status), L.stream()), "rr_enc_layer_split")
                row, col, row2, col2 = row2, col2, row, col
                continue
            L.check(lib.rr_enc_layer(wr, wc, L.ptr(row), L.ptr(col), L.ptr(row2), L.ptr(col2), L.ptr(D), L.ptr(locs),
                                     L.ptr(theta) if theta is not None else None, L.ptr(bias) if use_dur else None,
                                     Bp, N, norm_mode, None, L.stream()),
                    "rr_enc_layer")
            row, col, row2, col2 = row2, col2, row, col
        if train_saves is not None:
            if theta is None:      # duration NAB: the kernels take the angles from the coordinates; the backward wants the matrix
                theta = torch.empty(Bp, N, N, device=dev, dtype=torch.float32)
                L.check(lib.rr_edge_angles(L.ptr(locs), L.ptr(theta), Bp, N, L.stream()), "rr_edge_angles")
            train_saves.append({"theta": theta})
        return row, col
