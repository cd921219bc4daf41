"""Host launcher of the persistent rollout kernel (csrc/rr_decode.hip)."""
from __future__ import annotations

import torch

from .. import _lib as L

PROB_ID = {"atsp": 0, "rcvrp": 1, "rcvrptw": 2}
MODE_ID = {"greedy": 0, "sampling": 1, "evaluate": 2}
import os
# Fused greedy / sampling rollouts on the fp16 matrix pipe with two-piece split fp32 operands (x = hi + 2^-11 lo'; per product
# hi*hi + 2^-11 (hi*lo' + lo'*hi), fp32 accumulate; measured error of a dot product 4e-8 of sum |a b|, below the fp32 MFMA's own
# 1.1e-7: tools/clockprobe/f16probe.hip): attention scores, P.V, pointer MLP and logits.  The default since round 2 — tours
# identical to the fp32-MFMA build on the golden fixtures and on >= 99.9 % of the full-size rollouts with every divergence at
# a decision gap < 1e-3 (tests/test_gpu_fullsize.py).  RR_MLP_SPLIT=0 (or SPLIT_MLP = False) runs the fp32 MFMA kernel.
SPLIT_MLP = True      # False: the fp32 MFMA kernel whatever the pack holds (A/B measurements); packing.mlp_split_enabled() decides otherwise
STAGGER = int(os.environ.get("RR_STAGGER", "0"))   # initial delay of the second wave of every SIMD (units of ~8k cycles)
# process-wide A/B switches, read ONCE here and handed to the launcher as RolloutIO fields (csrc/rr_decode.hip reads no environment)
TAIL_PACK = int(os.environ.get("RR_TAIL_PACK", "0") != "0")          # pack the S % 16 left-over rollouts of several instances into one tile
NO_INST = int(os.environ.get("RR_ROLLOUT_INST", "1") == "0")         # never the instance-mode kernel
TIMING = None   # bench.py sets this to a list to collect (start, end) HIP events around each full rollout launch


def fused_filters_ok(precision="32", training=False):
    """top-k / top-p inside the fused rollout: the FILT builds of the two-piece greedy / sampling kernels only (csrc/rr_decode.hip)."""
    from .. import packing
    return bool(SPLIT_MLP and packing.mlp_split_enabled()) and precision != "16-mixed" and not training


def launch_rollout(env_name, packed, cache, td, num_starts, *, actions=None, logp=None, t0=0, nsteps=1,
                   mode="greedy", actions_in=None, logits_out=None, logits_only=False, write_state=False,
                   tanh_clip=10.0, temperature=1.0, seed=0, steps_out=None, state=None, dump=None, status=None, precision="32",
                   top_k=0, top_p=0.0):
    """Runs `nsteps` decode steps (nsteps <= 0: until every rollout is done) for all rollouts of `td`.
    `td` is the batchified rollout state (R = S*Bp rows, per-instance keys left at Bp rows)."""
    if env_name not in PROB_ID:
        raise NotImplementedError(f"fused rollout for env '{env_name}'")
    D = td["distance_matrix"]
    Bp, N = D.shape[0], D.shape[-1]
    mask = td["action_mask"]
    R = mask.shape[0]
    S = max(R // Bp, 1)
    assert S * Bp == R
    io = L.RolloutIO()
    st = state if state is not None else {}
    cur = st.get("cur")
    if cur is None:
        cur = td["current_node"].reshape(-1).contiguous()
    first = st.get("first")
    if first is None and "first_node" in td:
        first = td["first_node"].reshape(-1).contiguous()
    io.K, io.Vt, io.L = L.ptr(cache.glimpse_key), L.ptr(cache.glimpse_val_t), L.ptr(cache.logit_key)
    io.ctxA, io.ctxB = L.ptr(cache.ctx_a), L.ptr(cache.ctx_b)
    io.D, io.Dur = L.ptr(D.contiguous()), None
    io.cur, io.first = L.ptr(cur), L.ptr(first)
    mask = mask.contiguous()
    io.mask = L.ptr(mask)
    keep = [cur, first, mask]
    if env_name == "rcvrp":
        dem, vis = td["demand"].contiguous(), td["visited"].contiguous()
        used, vcap = td["used_capacity"].reshape(-1).contiguous(), td["vehicle_capacity"].reshape(-1).contiguous()
        io.demand, io.visited, io.used, io.vcap = L.ptr(dem), L.ptr(vis), L.ptr(used), L.ptr(vcap)
        keep += [dem, vis, used, vcap]
    if env_name == "rcvrptw":
        T_ = td["duration_matrix"].float().contiguous()
        dem, vis = td["demand_linehaul"].contiguous(), td["visited"].contiguous()
        tw, sv = td["time_windows"].contiguous(), td["service_time"].contiguous()
        used, vcap = td["used_capacity_linehaul"].reshape(-1).contiguous(), td["vehicle_capacity"].reshape(-1).contiguous()
        ctime, rlen = td["current_time"].reshape(-1).contiguous(), td["current_route_length"].reshape(-1).contiguous()
        if vcap.shape[0] != R:
            vcap = vcap.repeat(R // vcap.shape[0])
        io.Dur, io.demand, io.tw, io.service = L.ptr(T_), L.ptr(dem), L.ptr(tw), L.ptr(sv)
        io.visited, io.used, io.vcap, io.ctime, io.rlen = L.ptr(vis), L.ptr(used), L.ptr(vcap), L.ptr(ctime), L.ptr(rlen)
        keep += [T_, dem, vis, tw, sv, used, vcap, ctime, rlen]
        if td.meta.get("mtvrp_variant", False):        # multi-task variants: context inputs + what the in-kernel env.step needs
            ub = td["used_capacity_backhaul"].reshape(-1).float().contiguous()
            opn = td["open_route"].reshape(-1).to(torch.uint8).contiguous()
            lim = td["distance_limit"].reshape(-1).float().contiguous()
            dmb = td["demand_backhaul"].float().contiguous()
            bcl = td["backhaul_class"].reshape(-1).to(torch.int32).contiguous()
            if ub.shape[0] != R:
                ub = ub.repeat(R // ub.shape[0])
            io.used_b, io.open_route, io.dist_limit = L.ptr(ub), L.ptr(opn), L.ptr(lim)
            io.demand_b, io.bclass = L.ptr(dmb), L.ptr(bcl)
            keep += [ub, opn, lim, dmb, bcl]
    done = td["done"].reshape(-1).contiguous() if "done" in td and td["done"].numel() == R else None
    io.done = L.ptr(done)
    io.actions, io.logp = L.ptr(actions), L.ptr(logp)
    io.logits_out, io.actions_in = L.ptr(logits_out), L.ptr(actions_in)
    io.steps_out = L.ptr(steps_out)
    io.Bp, io.N, io.S = Bp, N, S
    io.T = actions.shape[1] if actions is not None else 1
    io.t0, io.nsteps = t0, nsteps
    io.mode = MODE_ID[mode]
    io.use_placeholder = int(env_name == "atsp" and td.meta.get("i", 1) == 0)
    io.set_first = int(env_name == "atsp" and td.meta.get("i", 1) == 0)
    io.write_state, io.logits_only = int(write_state), int(logits_only)
    io.stagger = STAGGER
    io.tail_pack, io.no_inst = TAIL_PACK, NO_INST
    from .. import packing
    split_on = SPLIT_MLP and packing.mlp_split_enabled()
    io.use_split = int(split_on)
    if dump is not None:      # training: per decoder evaluation, what csrc/rr_train_dec.hip differentiates (see RolloutIO::dump_*)
        io.dump_g0, io.dump_g, io.dump_meta = L.ptr(dump["g0"]), L.ptr(dump["g"]), L.ptr(dump["meta"])
        io.dump_scal = L.ptr(dump.get("scal"))
        io.dumpT = int(dump["T"])
        # training rollouts run on the same split-operand kernel (fp32-level accuracy; the reference trains in 16-bit mixed
        # precision, configs/trainer/default.yaml:8)
        io.use_split = int(os.environ.get("RR_TRAIN_SPLIT", "1") != "0" and split_on)
    if precision == "16-mixed" and io.use_split and dump is None:
        # the reference's own GPU arithmetic mode (torch.autocast, test.py:183; Lightning "16-mixed", configs/trainer/default.yaml:8):
        # one fp16 piece per operand, fp32 accumulation, fp32 softmax / logits (decoder.py:195-196) — csrc/rr_rollout_w.inc, HALF.
        # Inference launches in instance mode only (7 tiles per instance); other shapes fall back to the two-piece kernels.
        io.use_split = 2
    if io.use_split and not logits_only and mode != "evaluate":
        ks, vts, ls = cache.split_images(status)
        io.Ks, io.Vts, io.Ls = L.ptr(ks), L.ptr(vts), L.ptr(ls)
        io.status = L.ptr(status)
    io.tanh_clip, io.temperature, io.seed = float(tanh_clip), float(temperature), int(seed)
    io.top_k, io.top_p = (0, 0.0) if logits_only else (int(top_k), float(top_p))
    timed = TIMING is not None and not logits_only
    if timed:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
    L.check(L.lib().rr_rollout(packed["dec"], io, PROB_ID[env_name], L.stream()), "rr_rollout")
    if timed:
        e1.record()
        TIMING.append((e0, e1))
    return {"cur": cur, "first": first, "mask": mask, "done": done, "keep": keep}
