"""RRNetPolicy — drop-in for rrnco.models.policy.RRNetPolicy (rrnco/models/policy.py:138-255)."""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import packing
from ..tensordict_lite import TensorDict
from ..ops import calculate_entropy, get_log_likelihood
from .decoder import RRNetDecoder
from .decoding import get_decoding_strategy
from .encoder import RRNetEncoder
from .rollout import PROB_ID, fused_filters_ok, launch_rollout


class RRNetPolicy(nn.Module):
    def __init__(self, encoder=None, decoder=None, embed_dim=128, num_encoder_layers=3, num_heads=8,
                 normalization="batch", feedforward_hidden=512, env_name="rcvrp", encoder_network=None,
                 init_embedding=None, init_embedding_kwargs={}, context_embedding=None, dynamic_embedding=None,
                 use_graph_context=True, linear_bias_decoder=False, sdpa_fn=None, sdpa_fn_encoder=None,
                 sdpa_fn_decoder=None, mask_inner=True, out_bias_pointer_attn=False, check_nan=True,
                 temperature=1.0, tanh_clipping=10.0, mask_logits=True, train_decode_type="sampling",
                 val_decode_type="greedy", test_decode_type="greedy", moe_kwargs={"encoder": None, "decoder": None},
                 nab_type="gating", precision="32", **unused_kwargs):
        super().__init__()
        # "32" (default): fp32-equivalent arithmetic everywhere (two-piece fp16 operands, three partial products).  "16-mixed": the
        # fused inference rollout multiplies ONE fp16 piece per operand with fp32 accumulation, as the reference does under
        # torch.autocast (test.py:183) / Lightning 16-mixed (configs/trainer/default.yaml:8); the encoder and the decoder cache stay
        # fp32-equivalent.  Opt-in: its results are NOT held to the fp32 golden tolerances (tests/test_gpu_mixed.py states its own).
        if precision not in ("32", "32-true", "16-mixed"):
            raise ValueError(f"precision {precision!r}: '32' or '16-mixed'")
        self.precision = "16-mixed" if precision == "16-mixed" else "32"
        self.encoder = encoder if encoder is not None else RRNetEncoder(
            embed_dim=embed_dim, num_heads=num_heads, num_layers=num_encoder_layers, env_name=env_name,
            normalization=normalization, feedforward_hidden=feedforward_hidden, net=encoder_network,
            init_embedding=init_embedding, init_embedding_kwargs=init_embedding_kwargs, nab_type=nab_type)
        self.decoder = decoder if decoder is not None else RRNetDecoder(
            embed_dim=embed_dim, num_heads=num_heads, env_name=env_name, mask_inner=mask_inner,
            out_bias_pointer_attn=out_bias_pointer_attn, linear_bias=linear_bias_decoder,
            use_graph_context=use_graph_context, check_nan=check_nan)
        self.env_name = getattr(env_name, "name", env_name)
        self.temperature, self.tanh_clipping, self.mask_logits = temperature, tanh_clipping, mask_logits
        self.train_decode_type, self.val_decode_type, self.test_decode_type = \
            train_decode_type, val_decode_type, test_decode_type
        self._pack_cache = None
        # Throughput knob for the VRPs (not a constructor argument of the reference): their rollouts end after a data-dependent number
        # of steps, and trimming `actions` / the log-probabilities to it costs one host read per call.  lazy_trim = True leaves them at
        # the allocated length 2 N + 2 (depot / 0.0 behind each route's end: neutral for the reward, the log-likelihood and the
        # validity check), returns the step count as a DEVICE scalar in out["steps"] and lets the range guard run deferred.
        self.lazy_trim = False

    # ---- packed (MFMA-ordered / folded) weights, rebuilt when any parameter changes (versions + packing.weights_fingerprint)
    def invalidate_pack(self) -> None:
        self._pack_cache = None
        self._pack_verified = False
        self._mlp_train_pack = None          # (models/dec_backward.py: the pointer MLP's training packs follow the same rule)
        self._enc_train_pack = None          # (models/enc_backward.py: transposed projections, FFN packs of the encoder backward)
        # Generation counter: part of the "in-scope" pack key below.  A fused optimizer updates in place without bumping the version
        # counters, so (device, versions, pointers) repeat step after step; without the counter a pack derived from the policy's key
        # (the encoder's training packs) would be taken for current after every optimizer step but the first.
        self._pack_gen = getattr(self, "_pack_gen", 0) + 1
        self._range_sticky_fp32 = False      # new weights: the split kernels get another chance (a raised word sets it again)
        # words of calls made with the OLD weights say nothing about the new ones (their outputs were NaN-marked on the device): without
        # this a word raised before load_state_dict would throw at the first call afterwards and send the new weights straight back to fp32
        self._range_pending = None

    def load_state_dict(self, *args, **kwargs):
        out = super().load_state_dict(*args, **kwargs)
        self.invalidate_pack()               # new weights: every derived pack is stale; the fp32 exile of the range guard ends
        return out

    def param_index(self) -> dict:
        """{"named": [(name, parameter)], "params": [...], "P": {name: parameter}, "named_buffers": [...], "buffers": [...]} of this module.
        Inside pack_scope (one training step: nobody registers or replaces a parameter there) the module tree is walked ONCE per step
        instead of once per consumer — pack key, pack, decoder / encoder / init backward, gradient list: ten walks over ~565 parameters
        were ~3 ms of a small-batch step's host time (profiles/r06/NOTES.md section 9).  Outside a scope every call walks, as before."""
        idx = getattr(self, "_pidx", None)
        if idx is not None and getattr(self, "_pack_scope", False):
            return idx
        named, bufs = list(self.named_parameters()), list(self.named_buffers())
        idx = {"named": named, "params": [q for _, q in named], "P": dict(named), "named_buffers": bufs, "buffers": [q for _, q in bufs]}
        self._pidx = idx if getattr(self, "_pack_scope", False) else None
        return idx

    def pack_scope(self):
        """Context: the weights do not change inside (one training step up to its optimizer step) — packed() verifies once."""
        pol = self

        class _Scope:
            def __enter__(self_):
                pol._pack_scope, pol._pack_verified, pol._pidx = True, False, None

            def __exit__(self_, *exc):
                pol._pack_scope, pol._pack_verified, pol._pidx = False, False, None
                return False
        return _Scope()

    def train(self, mode: bool = True):
        self._pack_dirty = True          # any train() / eval() switch: re-check the weights' fingerprint once (see packed())
        self._pack_verified = False
        return super().train(mode)

    def packed(self, device):
        # Inside RRNet.training_step (pack_scope) the pack is checked ONCE per step: the first packed() verifies (fingerprint: one host
        # read, ~470 detached views) or rebuilds, the others — the encoder, the decoder backward, the encoder backward — take the
        # cache.  Outside such a scope every call verifies, as before.
        if getattr(self, "_pack_scope", False) and getattr(self, "_pack_verified", False) and self._pack_cache is not None \
                and self._pack_cache[0][0] == str(device) \
                and self._pack_cache[0][1] == packing.mlp_split_enabled():
            return self._pack_cache[1]
        # buffers too: BatchNorm running statistics are folded into the pack (packing.py), and a buffer-only load must repack
        pidx = self.param_index()
        ts = pidx["params"] + pidx["buffers"]
        key = (str(device), packing.mlp_split_enabled(), tuple(p._version for p in ts), tuple(p.data_ptr() for p in ts))
        # The norm fingerprint (one multi-tensor launch + ONE HOST READ) catches in-place updates that do not bump the version
        # counters (fused optimizers).  Those only happen around training-mode calls: an eval-mode module that has not been in
        # training mode since its last pack is checked by versions and pointers alone — no host synchronisation per inference call.
        dirty = self.training or getattr(self, "_pack_dirty", True)
        if getattr(self, "_pack_scope", False) and self._pack_cache is None:
            # a training step right after invalidate_pack(): there is nothing to compare a fingerprint with (one host read saved: the
            # host may run ahead into this step while the device finishes the last optimizer step); the key without it never
            # matches a later out-of-scope key, so the next unscoped call verifies from scratch
            key = key + (("in-scope", getattr(self, "_pack_gen", 0)),)
            self._pack_dirty = True
        elif dirty:
            key = key + (packing.weights_fingerprint(self, ts),)
            self._pack_dirty = self.training
        elif self._pack_cache is not None:
            key = key + (self._pack_cache[0][-1],)
        if self._pack_cache is None or self._pack_cache[0] != key:
            if not dirty:
                key = key[:4] + (packing.weights_fingerprint(self, ts),)
            # (names -> tensors like state_dict(), without its ~470 detached views: this runs once per training step)
            sd = dict(pidx["P"])
            sd.update(pidx["named_buffers"])
            self._pack_cache = (key, packing.pack_policy(sd, self.env_name, device))
        self._pack_verified = True
        return self._pack_cache[1]

    def forward(self, td, env=None, phase="train", *args, capture=None, **kwargs) -> dict:
        """rrnco/models/policy.py:138-255.  The kernels run without an autograd graph; when gradients are enabled, the module
        is in training mode and the call is a training forward (phase="train", sampled, fused rollout), the returned
        `log_likelihood` is attached to the parameters through `_PolicyLogLikelihood`, whose backward is the hand-written
        decoder backward + the encoder replay (models/grad_replay.py:replay_backward_hip): `loss.backward()` of the
        reference's training step (rrnco/models/rl.py:118-128) works unchanged."""
        want = (capture is None and phase == "train" and self.training and torch.is_grad_enabled()
                and kwargs.get("actions", None) is None and self.env_name in PROB_ID
                and any(p.requires_grad for p in self.parameters()))
        if not want:
            return self._forward_impl(td, env, phase, *args, capture=capture, **kwargs)
        from .encoder import ATSPInitEmbedding, draw_sample_indices
        if td.get("sample_idx", None) is None:       # forward and backward must see the same neighbour sample (atsp.py:55-67)
            td.set("sample_idx", draw_sample_indices(self.encoder.init_embedding, td["distance_matrix"], phase))
        keys = ("distance_matrix", "locs", "demand", "duration_matrix", "demand_linehaul", "time_windows", "service_time")
        state = {k: td[k] for k in keys if k in td.keys()}
        sidx = td["sample_idx"]
        cap = {}
        out = self._forward_impl(td, env, phase, *args, capture=cap, **kwargs)
        if "dump" not in cap:        # step-wise decode paths (top-k / top-p, entropy rows, beam search) leave no dump
            return out
        params = [p for p in self.parameters() if p.requires_grad]
        out["log_likelihood"] = _PolicyLogLikelihood.apply(out["log_likelihood"], self, state, cap, sidx, *params)
        return out

    @torch.no_grad()
    def _forward_impl(self, td, env=None, *args, capture=None, range_guard=None, **kwargs) -> dict:
        """Range guard of the fp16 two-piece arithmetic (csrc/rr_common.h).  The default kernels convert fp32 operands to fp16
        pairs: a weight, an embedding or an activation of magnitude >= 65 504 (after its image's scale) becomes inf.  One device
        word collects: bit 0 a K / V^T / L image out of range or non-finite (rr_pack_f16x2: that includes anything non-finite the
        encoder produced), bit 1 a pointer-MLP weight image out of range (packing.f16_range_status), bit 2 a non-finite
        log-probability in the fused rollout (RolloutIO::status; the kernel also writes NaN into that rollout's log-probability,
        so the call's log-likelihood is NaN, never a plausible number).  Modes (`range_guard=` or RR_RANGE_GUARD):
          "sync"     the word is read once per call, before anything looks at the tours; a call that raised it is repeated on the
                     fp32-MFMA kernels, which have no such limit — costs one host synchronisation per call (~2 ms of pipeline bubble
                     on the 80 ms headline step);
          "deferred" no host read in the call and none that waits for it later: every float output of a flagged call (reward,
                     normalized_reward, log_likelihood) is NaN-marked ON THE DEVICE (one scalar `where` + three adds), so a caller that
                     only reads the rewards (test.py:204-213) cannot consume a bad batch as a plausible number; the word itself is copied
                     into a pinned host word behind the call (asynchronous copy + event).  Later calls POLL the events of earlier ones —
                     the host never waits for the device: it runs one call ahead, so a word is typically seen two calls later — and
                     check_range() waits for all of them: a raised word raises FloatingPointError there and the policy runs in fp32
                     until its weights change (invalidate_pack / load_state_dict);
          "auto"     (default) "sync" where the call synchronises anyway (the VRPs read their step count; training steps),
                     "deferred" otherwise (ATSP inference);  "off": no guard.
        The step-wise decode paths (fused=False, N > 103, top-k / top-p, beam search) run the same split encoder / cache kernels: "sync"
        reads the word after their loop as well, "deferred" NaN-marks their outputs the same way."""
        import os
        mode = range_guard or os.environ.get("RR_RANGE_GUARD", "auto")
        mode = {"1": "sync", "0": "off"}.get(mode, mode)
        self._range_poll(wait=False)                             # deferred words of earlier calls that have landed (no wait: round 4 read the
        #                                                          previous call's word here and so serialised the host behind the device)
        if getattr(self, "_range_sticky_fp32", False):
            with packing.force_fp32():
                return self._forward_core(td, env, *args, capture=capture, **kwargs)
        if not packing.mlp_split_enabled() or mode == "off" or td.device.type != "cuda":
            self._range_status = None
            return self._forward_core(td, env, *args, capture=capture, **kwargs)
        if mode == "auto":
            syncs_anyway = capture is not None or (self.env_name != "atsp" and not getattr(self, "lazy_trim", False))
            mode = "sync" if syncs_anyway else "deferred"
        self._range_status = self.packed(td.device)["range_status"].clone()
        self._range_sync = mode == "sync"
        self.last_range_flags = 0
        if mode == "deferred":
            status = self._range_status
            try:
                out = self._forward_core(td, env, *args, capture=capture, **kwargs)
            finally:
                self._range_status = None
                self._range_push(status)
            # device-side poison: 0 for a clean word, NaN for a raised one — added to every float output of the call
            poison = torch.where(status != 0, torch.full((), float("nan"), device=status.device), torch.zeros((), device=status.device))
            for k in ("reward", "normalized_reward", "log_likelihood"):
                v = out.get(k, None)
                if torch.is_tensor(v) and v.is_floating_point():
                    out[k] = v + poison.to(v.dtype)
            return out
        td_in = TensorDict(td, batch_size=td.batch_size)         # shallow copy: the second pass starts from the same state
        try:
            self._range_read = False
            out = self._forward_core(td, env, *args, capture=capture, **kwargs)
            if not self._range_read:                             # step-wise decode path: nobody has read the word yet
                flags = int(self._range_status.item())
                if flags != 0:
                    raise _RangeRetry(flags)
            return out
        except _RangeRetry as r:
            import warnings
            self.last_range_flags = r.flags
            warnings.warn(f"rrnco_amd: an operand left the fp16 range of the split kernels (flags {r.flags:#x}: 1 = K/V/L image, "
                          "2 = weight image, 4 = non-finite log-probability); repeating the call on the fp32 MFMA kernels")
            if capture is not None:
                capture.clear()
            self._range_status = None
            with packing.force_fp32():
                return self._forward_core(td_in, env, *args, capture=capture, **kwargs)
        finally:
            self._range_status = None

    def _range_push(self, status) -> None:
        """A deferred call's guard word -> pending list.  Outside a graph capture: an asynchronous copy into a pinned host word (a small
        ring of them) and an event behind it, both on the call's stream.  Inside a capture nothing may be read back or recorded against
        the host: the device word itself stays pending for check_range()."""
        pend = getattr(self, "_range_pending", None) or []
        if status.is_cuda and not torch.cuda.is_current_stream_capturing():
            ring = getattr(self, "_range_ring", None)
            if ring is None:
                ring = self._range_ring = [torch.zeros(1, dtype=torch.int32).pin_memory() for _ in range(8)]
                self._range_ring_next = 0
            if len(pend) >= len(ring):                           # a ring slot is only reused after its word was read
                self._range_pending = pend
                self._range_poll(wait=True)
                pend = self._range_pending or []
            host = ring[self._range_ring_next]
            self._range_ring_next = (self._range_ring_next + 1) % len(ring)
            host.copy_(status.reshape(1), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            pend.append((host, ev))
        elif status.is_cuda and getattr(self, "_range_graph_word", None) is not None:
            # inside a hipGraph capture (prepare_graph_capture() was called first): the call's word is created and zeroed INSIDE the captured
            # call, so every replay starts it from 0 — queueing it would only ever show the LAST replay.  The captured call ORs
            # it into a persistent word allocated outside the capture instead; that word is what check_range() reads (and clears).
            acc = self._range_graph_word
            acc.bitwise_or_(status.reshape(()).to(acc.dtype))
            if not any(w is acc for w, _ in pend):
                pend.append((acc, None))
        else:
            pend.append((status, None))
        self._range_pending = pend

    def prepare_graph_capture(self, device) -> None:
        """Call once BEFORE capturing a policy call into a hipGraph: allocates (outside the capture) the persistent range-guard word the
        captured call ORs its own status word into on every replay; check_range() reads and clears it."""
        if getattr(self, "_range_graph_word", None) is None:
            self._range_graph_word = torch.zeros((), dtype=torch.int32, device=device)

    def _range_poll(self, wait: bool) -> None:
        """Reads the pending guard words whose call has finished (wait=False: event query, never blocks) or all of them (wait=True)."""
        pend = getattr(self, "_range_pending", None)
        if not pend:
            return
        capturing = torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()
        flags, keep, dev_words = 0, [], []
        for word, ev in pend:
            if ev is None:                                       # a device word (captured call, CPU tensor): only an explicit check reads it
                if wait and not capturing:
                    dev_words.append(word)
                else:
                    keep.append((word, ev))
            elif wait and not capturing:
                ev.synchronize()
                flags |= int(word[0])
            elif not capturing and ev.query():
                flags |= int(word[0])
            else:
                keep.append((word, ev))
        if dev_words:
            for f in torch.stack([p.reshape(()) for p in dev_words]).tolist():      # ONE host read for all of them
                flags |= int(f)
            acc = getattr(self, "_range_graph_word", None)
            if acc is not None and any(w is acc for w in dev_words):                # the graph replays' accumulator: read, now cleared; it stays
                acc.zero_()                                                          # pending (later replays OR into it without a new push)
                keep.append((acc, None))
        self._range_pending = keep or None
        self.last_range_flags = flags
        if flags != 0:
            self._range_sticky_fp32 = True
            raise FloatingPointError(
                f"rrnco_amd: an earlier policy call left the fp16 range of the split kernels (flags {flags:#x}: 1 = K/V/L image, 2 = weight "
                "image, 4 = non-finite log-probability); its outputs were NaN-marked (reward, log-likelihood).  This policy now runs on the "
                "fp32 MFMA kernels until its weights change: repeat the call (range_guard='sync' repeats such calls by itself).")

    def check_range(self) -> None:
        """Waits for and reads every range-guard word "deferred" calls left behind.  If one is raised: FloatingPointError — that call's
        log-likelihoods were NaN-marked and its tours are not to be trusted — and the policy runs on the fp32-MFMA kernels from now on."""
        self._range_poll(wait=True)

    def _forward_core(self, td, env=None, phase="train", calc_reward=True, return_actions=True, return_entropy=False,
                      return_hidden=False, return_init_embeds=False, return_sum_log_likelihood=True, actions=None,
                      max_steps=1_000_000, fused=True, capture=None, **decoding_kwargs) -> dict:
        if env is None or isinstance(env, str):
            raise ValueError("pass an instantiated rrnco_amd env")
        packed = self.packed(td.device)
        saves = None
        if capture is not None and self.encoder.supports_hip_backward(packed) and capture.get("enc_saves", True):
            saves = capture["enc"] = []          # training forward: the encoder keeps what its hand-written backward reads
        from . import grad_replay as GR
        if GR.uses_batch_statistics(self):
            # normalization='batch' in train mode: statistics over all instances of the call — the one configuration the fused
            # block kernel (one instance per workgroup) does not serve; the encoder runs through torch ops on the device here
            # (forward: running statistics updated once, momentum 0.1; its backward recomputes with the same batch statistics)
            if td.get("sample_idx", None) is None:
                from .encoder import ATSPInitEmbedding, draw_sample_indices
                td.set("sample_idx", draw_sample_indices(self.encoder.init_embedding, td["distance_matrix"], phase))
            from . import bign
            import os as _os
            if (td["distance_matrix"].shape[-1] <= bign.MAX_N_ONCHIP and bign.supported(self.env_name, packed, "instance")
                    and _os.environ.get("RR_BN_TORCH", "0") != "1"):
                # round 5: on kernels too — the block as the row-parallel composition of models/bign.py with rr_bnorm_fwd (batch statistics,
                # running statistics moved once), saving what the hand-written block backward reads (enc_backward: rr_bnorm_bwd)
                bn_saves = []
                row_emb, col_emb = bign.encode_bn_train(self.encoder, td, packed, GR.params_and_buffers(self, 0.1), bn_saves, momentum=0.1)
                if capture is not None and capture.get("enc_saves", True):
                    capture["enc"] = bn_saves
            else:       # (RR_BN_TORCH=1, the ablation NABs: the model restated in torch ops, models/grad_replay.py)
                row_emb, col_emb = GR.encode_for_policy(self, td, td["sample_idx"], bn_momentum=0.1)
            row_emb, col_emb = row_emb.contiguous(), col_emb.contiguous()
            self._pack_dirty = True
            self._pack_verified = False      # (the running statistics just moved)
        else:
            row_emb, col_emb = self.encoder(td, phase=phase, packed=packed, train_saves=saves, status=getattr(self, "_range_status", None))
        if capture is not None:
            capture["emb"] = (row_emb, col_emb)

        decode_type = decoding_kwargs.pop("decode_type", None)
        if actions is not None:
            decode_type = "evaluate"
        elif decode_type is None:
            decode_type = getattr(self, f"{phase}_decode_type")
        strategy = get_decoding_strategy(
            decode_type, temperature=decoding_kwargs.pop("temperature", self.temperature),
            tanh_clipping=decoding_kwargs.pop("tanh_clipping", self.tanh_clipping),
            mask_logits=decoding_kwargs.pop("mask_logits", self.mask_logits),
            store_all_logp=decoding_kwargs.pop("store_all_logp", return_entropy), **decoding_kwargs)

        td, env, num_starts = strategy.pre_decoder_hook(td, env)
        td, env, cache = self.decoder.pre_decoder_hook(td, env, (row_emb, col_emb), num_starts, packed=packed,
                                                       status=getattr(self, "_range_status", None))

        # the fused rollout keeps one log-probability per step; full rows (store_all_logp / return_entropy) come from the step-wise loop
        if td["action_mask"].shape[-1] > 103:
            fused = False          # N > 103: the reference's own step-by-step loop on the row-parallel kernels (models/bign.py)
        # top-k / top-p (decoding.py:352-358) run inside the rollout on the two-piece greedy / sampling kernels (csrc/rr_rollout_w.inc, FILT);
        # with the fp32-MFMA or 16-mixed kernels, in evaluate mode and in a training step they take the per-step loop (rr_select filters there)
        filters = strategy.top_k > 0 or 0.0 < strategy.top_p < 1.0
        filters_fused = (not filters) or (fused_filters_ok(getattr(self, "precision", "32"), capture is not None)
                                          and strategy.mode in ("greedy", "sampling"))
        if (fused and not strategy.store_all_logp and self.env_name in PROB_ID and strategy.mask_logits and filters_fused
                and not getattr(strategy, "is_beam_search", False)):
            dump = None
            if capture is not None:      # training: keep what the hand-written backward needs (models/dec_backward.py)
                dump = capture.setdefault("dump", {})
                capture["cache"] = cache
            logprobs, actions_out, td = self._fused_rollout(td, env, cache, packed, strategy, actions, dump=dump)
            if strategy.num_starts > 0 and getattr(strategy, "select_best", False) and dump is None:
                logprobs, actions_out, td, env = strategy._select_best(logprobs, actions_out, td, env)
        else:
            step = 0
            while not td["done"].all():
                logits, mask = self.decoder(td, cache, num_starts, packed=packed)
                td = strategy.step(logits, mask, td, action=actions[..., step] if actions is not None else None)
                td = env.step(td)["next"]
                step += 1
                if step > max_steps:
                    break
            logprobs, actions_out, td, env = strategy.post_decoder_hook(td, env)

        if calc_reward:
            if env.normalize:
                real, normd = env.get_reward(td, actions_out)
                td.set("reward", real)
            else:
                td.set("reward", env.get_reward(td, actions_out))
        out = {"reward": td["reward"],
               "log_likelihood": get_log_likelihood(logprobs, actions_out, td.get("mask", None), return_sum_log_likelihood)}
        if calc_reward and env.normalize:
            out["normalized_reward"] = normd
        if return_actions:
            out["actions"] = actions_out
        if getattr(self, "_last_steps", None) is not None:
            out["steps"] = self._last_steps                                         # lazy_trim: see __init__
            self._last_steps = None
        if return_entropy:
            out["entropy"] = calculate_entropy(logprobs)                            # policy.py:248-249
        if return_hidden:
            out["hidden"] = (row_emb, col_emb)                                      # policy.py:250-251: the encoder output
        return out

    def _fused_rollout(self, td, env, cache, packed, strategy, actions_in, dump=None):
        """The whole `while not done` loop of policy.py:210-228 as ONE kernel launch."""
        R, N = td["action_mask"].shape
        dev = td.device
        t0 = len(strategy.actions)                      # 1 after a multistart hook, else 0
        if self.env_name == "atsp":
            T, nsteps = N, N - t0
        else:
            T, nsteps = 2 * N + 2, 0                    # data-dependent; trimmed to the longest rollout below
        # ATSP: every (rollout, step) of the N - t0 decode steps is written by the kernel — no 490 MB of zero fill at the headline shape;
        # the VRPs stop at a data-dependent step and leave depot / 0.0 behind each route's end
        alloc = torch.empty if (self.env_name == "atsp" and dump is None) else torch.zeros
        acts = alloc(R, T, dtype=torch.int64, device=dev)
        logp = alloc(R, T, dtype=torch.float32, device=dev)
        if t0:
            acts[:, 0] = strategy.actions[0]
            logp[:, 0] = 0.0
        steps_out = torch.zeros(1, dtype=torch.int32, device=dev)
        if dump is not None:      # training dump: one row per decoder evaluation (instance, step, start); see rollout.launch_rollout
            Bp = td["distance_matrix"].shape[0]
            Sd = R // Bp
            dT = (N - t0 - 1) if self.env_name == "atsp" else (T - t0)      # ATSP: the forced last move is not evaluated
            rows = Bp * dT * Sd
            # (rows a finished instance never reaches stay unwritten: the backward must not let them through — tests poison them with NaN)
            alloc = (lambda *sh: torch.full(sh, float("nan"), device=dev)) if getattr(self, "_debug_poison_dump", False) \
                else (lambda *sh: torch.empty(*sh, device=dev))
            dump.update({"T": dT, "S": Sd, "Bp": Bp, "N": N, "t0": t0,
                         # (+ 1: the trash row lanes without a live rollout store into, csrc/rr_rollout_w.inc)
                         "g0": alloc(rows + 1, 128)[:rows], "g": alloc(rows + 1, 128)[:rows],
                         # zeroed: rows a finished tile of rollouts never reaches must read as "not live"
                         "meta": torch.zeros(rows, 8, dtype=torch.int32, device=dev),
                         "scal": None if self.env_name == "atsp" else alloc(rows, 4)})
        ain = None
        if actions_in is not None:                      # evaluate: actions[..., step] feeds decode step `step`
            ain = torch.zeros(R, T, dtype=torch.int64, device=dev)
            ain[:, t0:t0 + actions_in.shape[1]] = actions_in
        st = launch_rollout(self.env_name, packed, cache, td, strategy.num_starts, actions=acts, logp=logp, t0=t0,
                            nsteps=nsteps, mode=strategy.mode, actions_in=ain, write_state=True,
                            tanh_clip=strategy.tanh_clipping, temperature=strategy.temperature, seed=strategy.seed,
                            steps_out=steps_out, dump=dump, status=getattr(self, "_range_status", None),
                            precision=getattr(self, "precision", "32"), top_k=strategy.top_k, top_p=strategy.top_p)
        if dump is not None:
            dump.update({"first": st["first"], "tanh_clip": strategy.tanh_clipping, "temperature": strategy.temperature})
        status = getattr(self, "_range_status", None)
        if status is not None and getattr(self, "_range_sync", True):      # the range guard's one host read (before anything looks at the tours)
            self._range_read = True
            flags = int(status.item())
            if flags != 0:
                raise _RangeRetry(flags)
        self._last_steps = None
        if self.env_name != "atsp":
            if getattr(self, "lazy_trim", False) and dump is None:
                self._last_steps = steps_out                 # device scalar: decode steps of the longest rollout (t0 not included)
            else:
                T_used = t0 + int(steps_out.item())
                if dump is not None:
                    dump["T_used"] = T_used - t0
                acts, logp = acts[:, :T_used].contiguous(), logp[:, :T_used].contiguous()
        td.update({"current_node": st["cur"], "action_mask": st["mask"].bool(), "action": acts[:, -1]})
        if st["first"] is not None:
            td.set("first_node", st["first"])
        td.set("done", torch.ones(R, dtype=torch.bool, device=dev))
        return logp, acts, td


class _RangeRetry(Exception):
    """Raised inside a split-kernel forward whose range-guard word is set (RRNetPolicy._forward_impl repeats the call in fp32)."""

    def __init__(self, flags):
        super().__init__(f"fp16 range guard flags {flags:#x}")
        self.flags = flags


class _PolicyLogLikelihood(torch.autograd.Function):
    """log-likelihood of the sampled tours as a differentiable function of the policy parameters: the forward value is the
    rollout's; the backward turns d loss / d ll into parameter gradients with the hand-written decoder backward
    (csrc/rr_train_dec.hip) and the encoder replay, exactly what RRNet.training_step does."""

    @staticmethod
    def forward(ctx, ll, policy, state, cap, sidx, *params):
        ctx.policy, ctx.state, ctx.cap, ctx.sidx, ctx.params = policy, state, cap, sidx, params
        return ll.detach().clone()

    @staticmethod
    def backward(ctx, gll):
        from .grad_replay import replay_backward_hip
        params = ctx.params
        held = [p.grad for p in params]
        for p in params:
            p.grad = None
        replay_backward_hip(ctx.policy, ctx.state, ctx.cap, ctx.cap["dump"]["S"], gll.contiguous(), ctx.sidx)
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
        for p, h in zip(params, held):
            p.grad = h
        ctx.cap = ctx.state = None            # the dump is several GB: release it with the graph
        return (None, None, None, None, None) + tuple(grads)
