"""ctypes binding of the C-ABI hot-path library (csrc/librrnco_hip.so, declared in include/rrnco_hip.h).

The product path has NO fallback: if the HIP library is missing or a launch fails, we raise.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.path.dirname(_HERE), "csrc", "librrnco_hip.so")

vp, i32, f32, u64, u32 = C.c_void_p, C.c_int, C.c_float, C.c_uint64, C.c_uint32


class EncBlockW(C.Structure):
    _fields_ = [(n, vp) for n in (
        "n1g", "n1b", "n2g", "n2b", "n3g", "n3b", "f1g", "f1b", "f2g", "f2b",
        "wq", "wk", "wv", "wp", "wc", "w1", "w2",
        "bq", "bk", "bv", "bp", "bc", "b1", "b2", "nab", "w1s", "w2s", "wqs", "wks", "wvs", "wps", "muk")]


class InitW(C.Structure):
    _fields_ = [(n, vp) for n in ("wi", "bi", "wr", "br", "wcl", "bcl", "g0r", "g0c", "g0rb", "g0cb", "g2r", "g2c",
                                  "wdep", "bdep", "wdm", "bdm", "cmr", "cmc", "cmrb", "cmcb")] + \
               [("g2rb", f32), ("g2cb", f32), ("nfeat", i32)] + [(n, vp) for n in ("gfr", "gfc", "gnr", "gnc", "gdr", "gdc")]


class CacheW(C.Structure):
    _fields_ = [(n, vp) for n in ("wk", "wv", "wl", "wca", "wcb", "wks", "wvs", "wls", "wcas", "wcbs")]


class DecW(C.Structure):
    _fields_ = [(n, vp) for n in ("w1", "w2", "b1", "b2", "q0", "wstate")] + [("alpha", f32), ("beta", f32)] + \
        [("w1s", vp), ("w2s", vp), ("b1s", vp)]


class NabDurW(C.Structure):
    _fields_ = [(n, vp) for n in ("mp", "ab", "cg", "wg2")] + [("bg2", f32 * 3), ("ko", f32 * 3),
                                                                ("inv_tau", f32), ("bo", f32), ("alpha", f32),
                                                                ("pwl", vp)]


class NabSimpleW(C.Structure):
    _fields_ = [(n, vp) for n in ("w0", "b0", "w2")] + [(n, f32) for n in ("b2", "alpha", "dw", "tw")]


class MatNetSideW(C.Structure):
    _fields_ = [(n, vp) for n in ("wq", "wkv", "wo", "w1", "w2", "b1", "b2", "n1g", "n1b", "n2g", "n2b", "mix")]


class RolloutIO(C.Structure):
    _fields_ = [(n, vp) for n in (
        "K", "Vt", "L", "ctxA", "ctxB", "D", "Dur", "demand", "tw", "service", "cur", "first", "mask", "visited", "used",
        "vcap", "ctime", "rlen", "done", "actions", "logp", "logits_out", "actions_in", "steps_out")] + \
        [(n, i32) for n in ("Bp", "N", "S", "T", "t0", "nsteps", "mode", "use_placeholder", "set_first",
                            "write_state", "logits_only", "stagger")] + \
        [("tanh_clip", f32), ("temperature", f32), ("seed", u64)] + [(n, vp) for n in ("used_b", "open_route", "dist_limit", "demand_b", "bclass")] + \
        [(n, vp) for n in ("dump_g0", "dump_g", "dump_meta", "dump_scal")] + [("dumpT", i32), ("use_split", i32), ("Ks", vp), ("Vts", vp), ("Ls", vp), ("status", vp), ("top_k", i32), ("top_p", f32),
                                                                                     ("tail_pack", i32), ("no_inst", i32)]


class DecLogitIO(C.Structure):          # csrc/rr_train_dec.hip
    _fields_ = [(n, vp) for n in ("g", "meta", "L", "Lt", "D", "Dur", "gll", "dlg", "dg", "logp", "dscal")] + \
        [(n, i32) for n in ("Bp", "N", "S", "T")] + [("seg_stride", C.c_longlong)] + \
        [(n, f32) for n in ("alpha", "beta", "tanh_clip", "temperature")] + [("Ls", vp)]


class MlpRowsW(C.Structure):
    _fields_ = [(n, vp) for n in ("wa1", "wa2", "wb", "b1", "b2")]


class MlpWgradW(C.Structure):
    _fields_ = [(n, vp) for n in ("w1n", "w2tn", "b1")]


class DecAttnIO(C.Structure):
    _fields_ = [(n, vp) for n in ("dg0", "meta", "scal", "first", "K", "V", "Kt", "ctxA", "ctxB", "wstate", "dK", "dV",
                                  "dctxA", "dctxB", "dwstate")] + \
        [(n, i32) for n in ("Bp", "N", "S", "T", "nscal")] + [("seg_stride", C.c_longlong)]


class EncSave(C.Structure):             # csrc/rr_enc_w.inc: what the training forward of a block keeps for its backward
    NAMES = ("r", "c", "q", "ek", "v", "num", "den", "y", "o", "u1", "x1", "eaT")
    _fields_ = [(n, vp) for n in NAMES]


class AftBwdIO(C.Structure):            # csrc/rr_train_enc.hip
    _fields_ = [(n, vp) for n in ("dy", "q", "ek", "v", "num", "den", "eaT", "dq", "dk", "dv", "dbias")] + [("N", i32)]


class NabDurBwdW(C.Structure):
    """csrc/rr_train_nabdur.hip: folded parameters of the duration NAB (models/grad_replay._nab_duration_params)."""
    _fields_ = [(n, vp) for n in ("a", "b", "co", "cg", "wg2", "scal", "mcat", "mcatT", "mcat_s", "mcatT_s")]


class DecBigIO(C.Structure):            # csrc/rr_bign.hip
    _fields_ = [(n, vp) for n in ("K", "Vt", "L", "ctxA", "ctxB", "D", "Dur", "cur", "first", "scal", "wstate", "mask", "w1", "w2",
                                  "b1", "b2", "logits")] + [(n, i32) for n in ("Bp", "N", "NP", "S", "nscal")] + [("alpha", f32), ("beta", f32)]


class GateBwdIO(C.Structure):
    _fields_ = [(n, vp) for n in ("hA", "hB", "w2", "b2", "node", "dist", "dout", "dnode", "ddist", "dw2", "db2")] + [("M", C.c_longlong), ("acc_node", i32), ("mix", vp)]


class MtvrpExtra(C.Structure):
    _fields_ = [(n, vp) for n in ("demand_b", "used_b", "open_route", "dist_limit", "bclass")]


_SIGS = {
    "rr_minmax_normalize": [vp, vp, vp, vp, i32, i32, vp],
    "rr_atsp_step": [vp, vp, vp, vp, i32, i32, vp],
    "rr_sample_neighbors": [vp, vp, i32, i32, i32, u64, vp],
    "rr_pack_f16x2": [vp, vp, C.c_longlong, vp, vp],
    "rr_nabdur_bwd": [C.POINTER(NabDurBwdW), vp, vp, vp, vp, vp, vp, vp, C.c_longlong, vp],
    "rr_rcvrp_step": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "rr_tour_cost": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp],
    "rr_select": [vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, i32, u64, u32, i32, f32, vp],
    "rr_enc_layer": [C.POINTER(EncBlockW), C.POINTER(EncBlockW), vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp],
    "rr_enc_stats": [vp, vp, vp, i32, i32, vp],
    "rr_enc_layer_split": [C.POINTER(EncBlockW), C.POINTER(EncBlockW), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp, vp],
    "rr_nab_dist_family": [C.POINTER(EncBlockW), C.POINTER(EncBlockW), vp, vp, i32, i32, vp],
    "rr_filter_rows": [vp, vp, i32, i32, i32, f32, vp],
    "rr_nab_tab_bwd": [vp, vp, vp, i32, vp],
    "rr_edge_angles": [vp, vp, i32, i32, vp],
    "rr_nab_dur": [C.POINTER(NabDurW), C.POINTER(NabDurW), vp, vp, vp, vp, i32, i32, vp],
    "rr_nab_dur_aug": [C.POINTER(NabDurW), C.POINTER(NabDurW), vp, vp, vp, vp, i32, i32, i32, vp],
    "rr_nab_simple": [C.POINTER(NabSimpleW), C.POINTER(NabSimpleW), i32, vp, vp, vp, vp, i32, i32, vp],
    "rr_rmtvrp_step": [vp] * 16 + [i32, i32, i32, C.POINTER(MtvrpExtra), vp],
    "rr_reinforce_loss": [vp, vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "rr_init_embed": [C.POINTER(InitW), i32, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "rr_init_embed_plain": [C.POINTER(InitW), i32, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "rr_dec_cache": [C.POINTER(CacheW), vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "rr_rollout": [C.POINTER(DecW), C.POINTER(RolloutIO), i32, vp],
    "rr_submatrix_gather": [vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "rr_matnet_layer": [C.POINTER(MatNetSideW), C.POINTER(MatNetSideW), vp, vp, vp, vp, vp, vp, C.c_size_t] + [i32] * 5 + [vp],
    "rr_matnet_init": [vp] * 7 + [i32, i32, i32, vp],
    "rr_matnet_linear": [vp, vp, vp, i32, i32, i32, i32, vp],
    "rr_matnet_dec_step": [vp] * 12 + [i32] * 5 + [vp],
    "rr_select_matnet": [vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, i32, u64, u32, vp],
    "rr_nab_train_fwd": [vp, vp, vp, vp, C.c_long, vp],
    "rr_nab_train_bwd": [vp, vp, vp, vp, vp, C.c_long, vp],
    "rr_nab_hist_bwd": [vp, vp, vp, vp, vp, C.c_long, vp],
    "rr_enc_layer_train": [C.POINTER(EncBlockW), C.POINTER(EncBlockW), vp, vp, vp, vp, vp, vp, vp, i32, i32,
                           C.POINTER(EncSave), C.POINTER(EncSave), vp],
    "rr_inorm_bwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "rr_linear_rows": [vp, vp, vp, vp, C.c_longlong, i32, vp, vp],
    "rr_aft_bwd": [C.POINTER(AftBwdIO), i32, vp],
    "rr_linear_smallk": [vp, i32, i32, vp, vp, vp, C.c_longlong, vp],
    "rr_gate_bwd": [C.POINTER(GateBwdIO), vp],
    "rr_small_gemm": [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, vp],
    "rr_inorm_fwd": [vp, vp, vp, vp, vp, i32, i32, vp],
    "rr_nab_pwl_fwd": [vp, vp, vp, vp, i32, i32, i32, vp],
    "rr_colsoftmax_exp": [vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "rr_aft_mix_big": [vp, vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, vp],
    "rr_bnorm_fwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, f32, C.c_longlong, vp],
    "rr_bnorm_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, C.c_longlong, i32, vp],
    "rr_dec_fwd_big": [C.POINTER(DecBigIO), vp],
    "rr_select_big": [vp, vp, vp, vp, vp, vp, i32, i32, f32, f32, i32, u64, u32, i32, f32, vp],
    "rr_dec_logit_bwd": [C.POINTER(DecLogitIO), vp],
    "rr_gemm_tn": [vp, vp, vp, i32, i32, i32, i32, i32, i32, C.c_longlong, C.c_longlong, C.c_longlong, i32, i32, vp, vp],
    "rr_mlp_rows": [C.POINTER(MlpRowsW), i32, vp, vp, vp, vp, i32, i32, C.c_longlong, vp],
    "rr_mlp_wgrad": [C.POINTER(MlpWgradW), vp, vp, vp, vp, vp, vp, vp, i32, i32, C.c_longlong, vp, vp],
    "rr_mlp_wgrad16": [C.POINTER(MlpWgradW), vp, vp, vp, vp, vp, vp, vp, i32, i32, C.c_longlong, vp, vp],
    "rr_dec_attn_bwd": [C.POINTER(DecAttnIO), vp],
}

_lib = None


def lib():
    """Load (once) and return the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"rrnco_amd: HIP library not built: {LIB_PATH} (run `python -c 'import __graft_entry__ as g; g.build()'` "
                "or `make -C real-routing-nco_amd/csrc`). There is no CPU fallback.")
        _lib = C.CDLL(LIB_PATH)
        for name, args in _SIGS.items():
            fn = getattr(_lib, name)
            fn.argtypes = args
            fn.restype = i32
        _lib.rr_matnet_workspace_bytes.argtypes = [i32] * 4
        _lib.rr_matnet_workspace_bytes.restype = C.c_size_t
        if os.environ.get("RR_MARKERS", "0") == "1":
            _wrap_with_markers(_lib)
    return _lib


# rocprofv3 range markers per kernel group of SURVEY.md section 3 (K1 .. K10), opt-in (RR_MARKERS=1; `rocprofv3 --marker-trace
# --kernel-trace`): every launcher call is bracketed by a roctx range named after the reference op sequence it replaces.
MARKER_LABELS = {
    "rr_edge_angles": "K1 NAB angles (attn_freenet.py:262-264)",
    "rr_nab_dist_family": "K1 NAB distance family, shared by the augmentation copies (attn_freenet.py:242-289)",
    "rr_nab_dur": "K1' NAB with duration (attn_freenet.py:226-237, 265-286)",
    "rr_nab_dur_aug": "K1' NAB with duration, x8 augmentation (attn_freenet.py:226-237, 265-286)",
    "rr_nab_simple": "K1 naive / heuristic NAB (attn_freenet.py:119-199)",
    "rr_enc_stats": "K3 instance-norm statistics of the init embedding (attn_freenet.py:84, 104-105)",
    "rr_enc_layer": "K1+K2+K3 encoder layer: NAB, AFT, norms, FFN (attn_freenet.py:292-327, 360-488)",
    "rr_enc_layer_split": "K1+K2+K3 encoder layer, three launches (attn_freenet.py:292-327, 360-488)",
    "rr_enc_layer_train": "K1+K2+K3 encoder layer, training forward (attn_freenet.py:292-327, 360-488)",
    "rr_init_embed": "K4 init embedding (env_embeddings/atsp.py:69-91, rcvrp.py:88-124)",
    "rr_init_embed_plain": "K4 init embedding, coordinates-only / distances-only branches (env_embeddings/atsp.py:92-104)",
    "rr_dec_cache": "K5 decoder cache (decoder.py:214-232)",
    "rr_rollout": "K6-K10 decode loop: context, pointer, inductive bias, select, env.step (policy.py:210-228)",
    "rr_select": "K9 select (decoding.py:341-361, 272-298)",
    "rr_atsp_step": "K10 env.step ATSP (atsp/env.py:72-101)",
    "rr_rcvrp_step": "K10 env.step RCVRP (rcvrp/env.py:90-122, 183-195)",
    "rr_rmtvrp_step": "K10 env.step RMTVRP (rmtvrp/env.py:343-428)",
    "rr_tour_cost": "reward (atsp/env.py:103-121, rcvrp/env.py:124-150)",
}


def _wrap_with_markers(lib_):
    rx = None
    for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
        try:
            rx = C.CDLL(name)
            break
        except OSError:
            continue
    if rx is None:
        raise ImportError("RR_MARKERS=1 but neither librocprofiler-sdk-roctx.so nor libroctx64.so can be loaded")
    rx.roctxRangePushA.argtypes = [C.c_char_p]
    rx.roctxRangePushA.restype = i32
    rx.roctxRangePop.restype = i32

    def marked(fn, label):
        text = label.encode()

        def call(*a):
            rx.roctxRangePushA(text)
            try:
                return fn(*a)
            finally:
                rx.roctxRangePop()
        call.__wrapped__ = fn
        return call
    for name in _SIGS:
        if hasattr(lib_, name):
            setattr(lib_, name, marked(getattr(lib_, name), MARKER_LABELS.get(name, name)))


def exported_symbols():
    return list(_SIGS)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    assert t.is_contiguous(), "rrnco_amd kernels need contiguous tensors"
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


def check(rc, what):
    if rc != 0:
        raise RuntimeError(f"rrnco_amd: {what} failed with status {rc} (-1 invalid argument, -2 HIP launch error)")


def require_gpu(t):
    if not t.is_cuda:
        raise RuntimeError("rrnco_amd: the MI355X hot path needs tensors on a ROCm device (no CPU fallback)")
