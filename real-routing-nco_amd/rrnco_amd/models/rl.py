"""RRNet RL module — mirror of rrnco.models.rl.RRNet.shared_step (rrnco/models/rl.py:96-166) without Lightning.

val / test phases are complete (augmentation, multistart, best-of metrics).  The train phase runs the sampling rollout (with the
training dump) and the POMO shared-baseline REINFORCE loss on HIP kernels and returns d loss / d log-likelihood; `training_step` turns
that into parameter gradients on the hand-written backward kernels (models/grad_replay.replay_backward_hip: decoder logits / attention /
pointer MLP in csrc/rr_train_dec.hip, encoder blocks in rr_train_enc.hip, init embeddings, NAB tables), combines them across ranks with
ONE flat RCCL all-reduce (parallel.allreduce_flat_gradients) and steps the optimizer: BASELINE configs[4].  `replay="torch"` keeps the
teacher-forced autograd replay of rounds 1-3 (models/grad_replay.replay_backward) for the step-wise decode paths and as an A/B."""
from __future__ import annotations

import torch

from .. import _lib as L
from ..ops import gather_by_index, unbatchify
from ..parallel import allreduce_flat_gradients
from .grad_replay import replay_backward, replay_backward_hip
from .policy import RRNetPolicy
from .transforms import StateAugmentation


def reinforce_loss(reward: torch.Tensor, log_likelihood: torch.Tensor, num_starts: int) -> dict:
    """reward, log_likelihood: [S*B] (r = s*B + b).  -> loss, bl_val [B], advantage [S*B], grad_ll [S*B]."""
    L.require_gpu(reward)
    R = reward.numel()
    B = R // num_starts
    reward, ll = reward.contiguous().float(), log_likelihood.contiguous().float()
    adv, grad = torch.empty_like(reward), torch.empty_like(reward)
    bl, part = torch.empty(B, device=reward.device), torch.empty(B, device=reward.device)
    loss = torch.empty(1, device=reward.device)
    L.check(L.lib().rr_reinforce_loss(L.ptr(reward), L.ptr(ll), L.ptr(adv), L.ptr(grad), L.ptr(bl), L.ptr(part), L.ptr(loss),
                                      B, num_starts, L.stream()), "rr_reinforce_loss")
    return {"loss": loss[0], "reinforce_loss": loss[0], "bl_loss": 0, "bl_val": bl, "advantage": adv, "grad_log_likelihood": grad}


class RRNet:
    def __init__(self, env, policy: RRNetPolicy = None, baseline: str = "shared", policy_kwargs: dict = {}, num_augment: int = 8,
                 augment_fn="dihedral8", first_aug_identity: bool = True, feats=None, num_starts: int = None,
                 no_aug_coords: bool = True, **unused):
        assert baseline == "shared", "RRNet only supports shared baseline"          # rl.py:74
        self.env, self.env_name = env, env.name
        kw = {"num_encoder_layers": 6, "normalization": "instance", "use_graph_context": False, **policy_kwargs}
        self.policy = policy if policy is not None else RRNetPolicy(env_name=env.name, **kw)
        self.num_starts, self.num_augment = num_starts, num_augment
        self.augment = StateAugmentation(num_augment=num_augment, augment_fn=augment_fn, first_aug_identity=first_aug_identity,
                                         feats=feats, no_aug_coords=no_aug_coords) if num_augment > 1 else None
        for phase in ("train", "val", "test"):                                      # rl.py:93-94
            attr = f"{phase}_decode_type"
            if "multistart" not in getattr(self.policy, attr):
                setattr(self.policy, attr, "multistart_" + getattr(self.policy, attr))

    def training_step(self, batch, optimizer=None, world: int = 1, enc_chunk: int = 512, dec_chunk: int = None,
                      replay: str = "hip", grad_clip: float = None, **policy_kw) -> dict:
        """One REINFORCE step on this rank's shard of instances (rl.py:96-128 + Lightning's DDP mean-reduction):
        sampling rollout, reward, shared-baseline loss and d loss / d ll on the HIP kernels; parameter gradients by the
        teacher-forced replay; one flat all-reduce (mean over ranks); optimizer step.  Returns the shared_step dict plus
        `replay_log_likelihood` (must agree with the rollout's) and `grad_norm`."""
        if self.env_name not in ("atsp", "rcvrp", "rcvrptw"):
            raise NotImplementedError(f"training_step for env '{self.env_name}'")
        from .encoder import ATSPInitEmbedding, draw_sample_indices
        td = self.env.reset(batch)
        if td.get("sample_idx", None) is None:            # the rollout and the replay must see the same neighbour sample
            td.set("sample_idx", draw_sample_indices(self.policy.encoder.init_embedding, td["distance_matrix"], "train"))
        state = {"distance_matrix": td["distance_matrix"], "locs": td["locs"]}
        if self.env_name == "rcvrp":
            state["demand"] = td["demand"]
        elif self.env_name == "rcvrptw":
            state.update({k: td[k] for k in ("duration_matrix", "demand_linehaul", "time_windows", "service_time")})
            if td.meta.get("mtvrp_variant", False):      # backhauls / open routes / distance limits: replayed with their terms
                state.update({k: td[k] for k in ("demand_backhaul", "open_route", "distance_limit", "backhaul_class")})
        sidx = td["sample_idx"]
        n_start = self.env.get_num_starts(td) if self.num_starts is None else self.num_starts
        cap = {} if replay == "hip" else None
        with self.policy.pack_scope():      # forward and backward of this step see the same weights: one pack verification
            return self._training_step_body(td, state, sidx, n_start, cap, optimizer, world, enc_chunk, dec_chunk, grad_clip, policy_kw)

    def _training_step_body(self, td, state, sidx, n_start, cap, optimizer, world, enc_chunk, dec_chunk, grad_clip, policy_kw) -> dict:
        out = self.policy(td, self.env, phase="train", num_starts=n_start, capture=cap, **policy_kw)
        r = out["normalized_reward"] if self.env.normalize else out["reward"]
        out.update(reinforce_loss(r, out["log_likelihood"], n_start))
        out["max_reward"] = unbatchify(out["reward"], (0, n_start)).max(dim=-1).values
        params = self.policy.param_index()["params"] if hasattr(self.policy, "param_index") else list(self.policy.parameters())
        for p in params:
            p.grad = None
        if cap is not None and "dump" in cap:     # decoder backward on the hand-written kernels (csrc/rr_train_dec.hip)
            out["replay_log_likelihood"] = replay_backward_hip(self.policy, state, cap, n_start, out["grad_log_likelihood"],
                                                               sidx, enc_chunk=enc_chunk)
        else:                                     # teacher-forced torch replay (the step-wise decode paths, A/B)
            gen = getattr(self.env, "generator", None)
            out["replay_log_likelihood"] = replay_backward(
                self.policy, state, out["actions"], n_start, out["grad_log_likelihood"], sidx, enc_chunk=enc_chunk, dec_chunk=dec_chunk,
                tanh_clipping=policy_kw.get("tanh_clipping"), temperature=policy_kw.get("temperature"),
                vehicle_capacity=float(getattr(gen, "vehicle_capacity", 1.0) or 1.0))
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
        grads = allreduce_flat_gradients(grads, world)
        for p, g in zip(params, grads):
            p.grad = g
        # one multi-tensor launch, not two per parameter (565 parameters: ~90 ms of launches per step on the host)
        out["grad_norm"] = torch.linalg.vector_norm(torch.stack(torch._foreach_norm([g.float() for g in grads])))
        if grad_clip is not None and grad_clip > 0:      # Lightning's gradient_clip_val (configs/trainer/default.yaml:6 = 1.0): clip_grad_norm_
            torch._foreach_mul_(grads, torch.clamp(grad_clip / (out["grad_norm"] + 1e-6), max=1.0))       # on the device, no host sync
        if optimizer is not None:
            optimizer.step()
            self.policy.invalidate_pack()                # the packed (MFMA-ordered, folded) weights are stale by construction
        return out

    def shared_step(self, batch, batch_idx: int = 0, phase: str = "val", **policy_kw) -> dict:
        td = self.env.reset(batch)
        n_aug, n_start = self.num_augment, self.num_starts
        n_start = self.env.get_num_starts(td) if n_start is None else n_start
        if phase == "train":
            n_aug = 0
        elif n_aug > 1:
            td = self.augment(td)
        out = self.policy(td, self.env, phase=phase, num_starts=n_start, **policy_kw)
        reward = unbatchify(out["reward"], (n_aug, n_start))
        if phase == "train":
            assert n_start > 1, "num_starts must be > 1 during training"
            r = out["normalized_reward"] if self.env.normalize else out["reward"]
            ll = out["log_likelihood"]
            out.update(reinforce_loss(r, ll.detach(), n_start))
            if ll.requires_grad:      # rl.py:123-128: loss = -(advantage * ll).mean(), here with the kernel's d loss / d ll as weights
                out["loss"] = out["reinforce_loss"] = (out["grad_log_likelihood"] * ll).sum()
            out["max_reward"] = reward.max(dim=-1).values
            return out
        out.update({"reward": reward, "no_aug_reward": reward[:, [0], :] if n_aug > 1 else reward})
        max_reward, max_idxs = reward.max(dim=-1)
        out.update({"max_reward": max_reward})
        if n_aug > 1:
            out["no_aug_max_reward"] = max_reward[:, [0]]
        if out.get("actions", None) is not None:
            actions = unbatchify(out["actions"], (n_aug, n_start))
            out["best_multistart_actions"] = gather_by_index(actions, max_idxs, dim=max_idxs.dim())
            out["actions"] = actions
        if n_aug > 1:
            max_aug_reward, aug_idx = max_reward.max(dim=1)
            out["max_aug_reward"] = max_aug_reward
            if "best_multistart_actions" in out:
                out["best_aug_actions"] = gather_by_index(out["best_multistart_actions"], aug_idx)
        return out
