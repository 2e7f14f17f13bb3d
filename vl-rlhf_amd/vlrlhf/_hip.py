"""ctypes binding of libvlr_hip.so (include/vlr.h).  The product path has NO fallback: if the library is missing
or a call fails this module raises - nothing is silently routed to PyTorch or to the CPU oracle."""
import ctypes as C
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("VLR_LIB") or os.path.join(os.path.dirname(_HERE), "libvlr_hip.so")   # VLR_LIB: A/B another build of the library

_lib = None

P, I, L, F, U64 = C.c_void_p, C.c_int, C.c_long, C.c_float, C.c_uint64


class LlamaCfg(C.Structure):
    _fields_ = [("hidden", I), ("inter", I), ("heads", I), ("head_dim", I), ("rms_eps", F), ("max_pos", I),
                ("rope_cos", P), ("rope_sin", P), ("kv_heads", I), ("resid_f32", I)]


class LayerWeights(C.Structure):
    _fields_ = [(n, P) for n in ("ln1", "wqkv", "wo", "ln2", "wgu", "wdown", "bqkv")]


class LayerGrads(C.Structure):
    _fields_ = [(n, P) for n in ("ln1", "wqkv", "wo", "ln2", "wgu", "wdown")]


class LayerActs(C.Structure):
    _fields_ = [(n, P) for n in ("xn1", "rstd1", "qkv", "attn", "lse", "x_mid", "xn2", "rstd2", "gu", "act", "x_out")]


class LayerBwdWs(C.Structure):
    _fields_ = [(n, P) for n in ("dact", "dxn", "dattn", "dqkv", "dx_mid", "delta", "norm_ws")]


class LoraWeights(C.Structure):
    _fields_ = [("r", I), ("scale", F), ("dropout", F)] + [(n, P) for n in ("a_qkv", "b_qkv", "a_o", "b_o", "a_gu", "b_gu", "a_down", "b_down")] + [("qkv_targets", I), ("mask_bits", P)]


class LoraGrads(C.Structure):
    _fields_ = [(n, P) for n in ("a_qkv", "b_qkv", "a_o", "b_o", "a_gu", "b_gu", "a_down", "b_down")]


class LoraBcomb(C.Structure):      # vlr_lora_bcomb: the groups' [B_lora | B_plora] of the two-adapter layer passes (ABI v7)
    _fields_ = [(n, P) for n in ("qkv", "o", "gu", "down")]


class VitCfg(C.Structure):
    _fields_ = [("hidden", I), ("mlp", I), ("heads", I), ("head_dim", I), ("ln_eps", F), ("act", I), ("head_dim_pad", I), ("attn_scale", F)]


class VitLayerWeights(C.Structure):
    _fields_ = [(n, P) for n in ("ln1_w", "ln1_b", "wqkv", "bqkv", "wo", "bo", "ln2_w", "ln2_b", "w1", "b1", "w2", "b2")]


class VitWs(C.Structure):
    _fields_ = [(n, P) for n in ("xn", "qkv", "attn", "h")]


# name -> argtypes (every function returns int status, except the *_bytes helpers and vlr_last_error)
_SIGS = {
    "vlr_gemm_bf16": [I, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, P],
    "vlr_gemm_bf16_scaled": [I, P, P, P, P, P, I, I, I, I, I, I, I, I, I, I, F, P],
    "vlr_gemm_bf16_tn_pair": [P, P, P, I, I, I, I, I, P, P, P, I, I, I, I, I, I, I, P],
    "vlr_rmsnorm_fwd": [P, P, P, P, I, I, F, P],
    "vlr_rmsnorm_bwd": [P, P, P, P, P, P, P, I, P, I, I, P],
    "vlr_rmsnorm_fwd_f32": [P, P, P, P, I, I, F, P],
    "vlr_rmsnorm_bwd_f32": [P, P, P, P, P, P, P, I, P, I, I, P],
    "vlr_gemm_bf16_f32res": [I, P, P, P, P, I, I, I, I, I, I, I, P],
    "vlr_gemm_lora_f32res": [P, I, P, P, I, P, I, I, I, I, P, I, P, I, P],
    "vlr_layernorm_fwd": [P, P, P, P, I, I, F, P],
    "vlr_layernorm_bwd": [P, P, P, F, P, P, P, I, P, I, I, P],
    "vlr_vit_embed_ln": [P, P, P, P, P, P, I, I, I, F, P],
    "vlr_im2col": [P, P, I, I, I, I, P],
    "vlr_rope_table": [P, P, I, I, F, P],
    "vlr_rope": [P, P, P, P, I, I, I, I, I, I, P],
    "vlr_rope_heads": [P, P, P, P, I, I, I, I, I, I, P],
    "vlr_swiglu_fwd": [P, P, I, I, P],
    "vlr_swiglu_bwd": [P, P, I, I, P],
    "vlr_gelu_fwd": [P, P, L, P],
    "vlr_gelu_bwd": [P, P, P, L, P],
    "vlr_colsum": [P, I, I, I, P, I, P, P],
    "vlr_colsum_f32": [P, I, I, I, P, P, P],
    "vlr_gather_rows": [P, P, P, I, I, P],
    "vlr_scatter_rows": [P, P, P, I, I, P],
    "vlr_rows_gather": [P, I, P, P, I, I, P],
    "vlr_rows_add": [P, P, P, I, I, I, P],
    "vlr_cast_f32_to_bf16": [P, P, L, P],
    "vlr_cast_bf16_to_f32": [P, P, L, P],
    "vlr_rowdot": [P, P, P, I, I, P],
    "vlr_attn_fwd": [P, P, P, I, P, I, P, P, I, I, I, I, I, F, P],
    "vlr_attn_bwd": [P, P, P, I, P, P, I, P, P, P, P, P, P, I, I, I, I, I, I, F, P],
    "vlr_attn_fwd_gqa": [P, P, P, I, P, I, P, P, I, I, I, I, I, I, F, P],
    "vlr_attn_bwd_gqa": [P, P, P, I, P, P, I, P, P, P, P, P, P, I, I, I, I, I, I, I, F, P],
    "vlr_merge_index": [P, P, P, I, I, I, I, I, I, I, I, P, P, P, P, P, P, P, P],
    "vlr_merge_fwd": [P, P, P, P, P, I, I, I, I, P],
    "vlr_merge_bwd": [P, P, P, P, P, P, I, I, I, I, I, I, P],
    "vlr_merge_fwd_f32": [P, P, P, P, I, P, I, I, I, I, P],
    "vlr_build_rows": [P, P, I, I, I, P, P, P, P],
    "vlr_logp_rows": [P, P, P, I, I, L, P, P, P],
    "vlr_dlogits_rows": [P, P, P, P, I, P, I, I, I, L, P, L, P],
    "vlr_seq_sum": [P, P, I, I, P, P],
    "vlr_lmhead_logps_fwd": [P, P, P, P, P, P, P, I, I, I, P],
    "vlr_lmhead_logps_bwd": [P, P, P, P, P, I, P, I, P, P, P, I, I, I, P],
    "vlr_dpo_loss": [P, P, P, P, I, F, F, I, I, P, P, P, P, P, P, P, P],
    "vlr_grad_sqnorm": [P, L, F, F, F, P, P, P],
    "vlr_adamw_step": [P, P, P, P, P, L, F, F, F, F, F, I, P, P],
    "vlr_decoder_layer_fwd": [P, P, P, P, P, P, I, I, P],
    "vlr_decoder_layer_fwd_ex": [P, P, P, P, P, P, I, I, I, P],
    "vlr_gemm_swiglu": [P, P, P, P, I, I, I, I, I, P],
    "vlr_gemm_swiglu_bwd": [P, P, P, P, I, I, I, P],
    "vlr_gemm_qkv_rope": [P, P, P, P, P, P, I, I, I, I, I, I, I, P],
    "vlr_gemm_qkv_rope_bias": [P, P, P, P, P, P, P, I, I, I, I, I, I, I, P],
    "vlr_gemm_lora": [P, I, P, P, I, P, I, I, I, I, P, I, P, I, P],
    "vlr_gemm_dropout_acc": [P, I, P, P, P, I, I, I, F, U64, F, P],
    "vlr_gemm_grouped": [I, P, P, P, I, I, I, I, I, I, I, L, L, L, F, I, I, U64, F, I, P],
    "vlr_lora_rows_u": [I, P, I, P, P, I, I, I, I, I, F, P, L, P, P],
    "vlr_lora_rows_v": [I, P, I, P, P, P, I, I, I, P, P],
    "vlr_gemm_dropout_acc_multi": [I, P, I, P, P, I, I, I, F, U64, F, I, P],
    "vlr_gemm_grouped_bits": [I, P, P, P, I, I, I, I, I, I, I, L, L, L, F, I, I, U64, F, I, P, L, P],
    "vlr_gemm_dropout_acc_bits": [P, I, P, P, P, I, I, I, F, U64, F, P, P],
    "vlr_gemm_dropout_acc_multi_bits": [I, P, I, P, P, I, I, I, F, U64, F, I, P, L, P],
    "vlr_gemm_grouped_bits_rows": [I, P, P, P, I, I, I, I, I, I, I, L, L, L, F, I, I, U64, F, I, P, L, P, P],
    "vlr_rows_tile_list": [P, I, P, P],
    "vlr_rows_tile_flags": [P, I, I, P, P],
    "vlr_gemm_seg_rowskip": [P, I],
    "vlr_gemm_grouped_bits_ktiles": [I, P, P, P, I, I, I, I, I, I, I, L, L, L, F, I, I, U64, F, I, P, L, P, P],
    "vlr_gemm_dropout_acc_multi_rows": [I, P, I, P, P, I, I, I, F, U64, F, I, P, L, P, P],
    "vlr_gemm_swiglu_bwd_add": [P, P, P, P, P, I, I, I, P],
    "vlr_gemm_swiglu_lora": [P, P, P, P, I, I, I, I, P, I, P, I, P],
    "vlr_gemm_qkv_rope_lora": [P, P, P, P, P, P, P, I, I, I, I, I, I, I, P, I, P, I, I, I, P],
    "vlr_decoder_layer_bwd": [P, P, P, I, P, P, P, P, P, P, P, I, I, P],
    "vlr_vit_layer_fwd": [P, P, P, P, I, I, P],
    "vlr_decoder_layer_fwd_lora": [P, P, P, P, P, P, U64, P, P, P, I, I, P],
    "vlr_decoder_layer_bwd_lora": [P, P, P, P, I, P, P, P, P, P, U64, P, P, P, P, P, I, I, P],
    "vlr_decoder_layer_fwd_lora_ex": [P, P, P, P, P, P, U64, P, P, P, P, I, I, P],
    "vlr_decoder_layer_bwd_lora_ex": [P, P, P, P, P, I, P, P, P, P, P, U64, P, P, P, P, P, P, I, I, P],
    "vlr_lora_concat_b": [P, I, P, I, P, L, P],
    "vlr_decoder_layer_fwd_lora2": [P, P, P, P, P, P, P, U64, U64, P, P, P, P, I, I, P],
    "vlr_decoder_layer_bwd_lora2": [P, P, P, P, P, P, I, P, P, P, P, U64, U64, P, P, P, P, P, P, I, I, P],
    "vlr_rows_mask": [P, I, I, P, I, P],
    "vlr_dropout": [P, P, L, F, U64, F, I, P],
    "vlr_dropout_mask": [P, L, F, U64, P],
    "vlr_dropout_bits": [P, L, F, U64, P],
    "vlr_dropout_bits2": [P, P, I, I, F, U64, P],
    "vlr_layers_join": [P],
    "vlr_allreduce_bucket": [P, P, L, I, P],
    "vlr_comm_probe": [P, P, L, I, P],
}
_INT_HELPERS = {
    "vlr_rmsnorm_bwd_workspace_bytes": [I],
    "vlr_layernorm_bwd_workspace_bytes": [I],
    "vlr_colsum_workspace_bytes": [I],
    "vlr_grad_sqnorm_workspace_bytes": [],
    "vlr_abi_version": [],
    "vlr_prof_enable": [I],
    "vlr_prof_collect": [P, I],
    "vlr_gemm_set_splitk_workspace": [P, L],
    "vlr_gemm_set_sched": [I],
    "vlr_gemm_set_trace": [P, L],
    "vlr_set_comm_cus": [I],
    "vlr_compute_cus": [],
    "vlr_lmhead_is_fused": [I, I, I],
    "vlr_comm_unique_id_bytes": [],
    "vlr_comm_unique_id": [P],
    "vlr_comm_init": [P, I, I, P],
    "vlr_comm_init_cfg": [P, I, I, I, I, P],
    "vlr_comm_rccl_version": [],
    "vlr_comm_has_config": [],
    "vlr_comm_destroy": [P],
}


class VlrError(RuntimeError):
    pass


def lib():
    """Loads libvlr_hip.so once.  Raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise VlrError(f"{LIB_PATH} not found: build it with `python vl-rlhf_amd/build_hip.py` "
                           "(the MI355X DPO path has no PyTorch/CPU fallback)")
        l = C.CDLL(LIB_PATH)
        l.vlr_last_error.restype = C.c_char_p
        l.vlr_last_error.argtypes = []
        l.vlr_comm_library.restype = C.c_char_p
        l.vlr_comm_library.argtypes = []
        l.vlr_lmhead_workspace_bytes.restype = C.c_long
        l.vlr_lmhead_workspace_bytes.argtypes = [I, I]
        l.vlr_lora_mask_bytes.restype = C.c_long
        l.vlr_lora_mask_bytes.argtypes = [I, I, I]
        l.vlr_dropout_bits_kt_bytes.restype = C.c_long
        l.vlr_dropout_bits_kt_bytes.argtypes = [I, I]
        for name, sig in _SIGS.items():
            if os.environ.get("VLR_LIB") and not hasattr(l, name):      # A/B against an older build of the library: newer entries may be absent
                continue
            fn = getattr(l, name)
            fn.restype = I
            fn.argtypes = sig
        for name, sig in _INT_HELPERS.items():
            if os.environ.get("VLR_LIB") and not hasattr(l, name):      # A/B against an older build of the library: newer diagnostics entries may be absent
                continue
            fn = getattr(l, name)
            fn.restype = I
            fn.argtypes = sig
        _lib = l
    return _lib


def exported_symbols():
    return list(_SIGS) + list(_INT_HELPERS) + ["vlr_last_error", "vlr_comm_library", "vlr_lmhead_workspace_bytes", "vlr_lora_mask_bytes", "vlr_dropout_bits_kt_bytes"]


def ptr(t):
    """device pointer of a tensor (or None)."""
    if t is None:
        return None
    if isinstance(t, int):
        return t
    return t.data_ptr()


def stream():
    return torch.cuda.current_stream().cuda_stream


_prof = None   # {entry point: [(start_event, end_event, args)]} while bench.py times kernels with HIP events


PROF_KERNELS = ["gemm_nt", "gemm_nn", "gemm_tn", "attn_fwd", "attn_bwd", "gemm256p"]


def lib_profile_start(sample=1):
    """brackets one launch in `sample` with HIP events (api.cpp); counts and work stay exact, time is the scaled sampled mean"""
    lib().vlr_prof_enable(max(1, int(sample)))


def lib_profile_stop():
    """-> {kernel: (launches, total_ms, flops)} from the in-library HIP-event timers (api.cpp)."""
    buf = (C.c_double * (3 * len(PROF_KERNELS)))()
    lib().vlr_prof_collect(buf, len(PROF_KERNELS))
    lib().vlr_prof_enable(0)
    return {k: (int(buf[3 * i]), buf[3 * i + 1], buf[3 * i + 2]) for i, k in enumerate(PROF_KERNELS)}


def profile_start(names):
    global _prof
    _prof = {n: [] for n in names}


def profile_stop():
    """-> {name: [(milliseconds, args), ...]}; call after torch.cuda.synchronize()."""
    global _prof
    out = {n: [(s.elapsed_time(e), a) for s, e, a in v] for n, v in (_prof or {}).items()}
    _prof = None
    return out


def call(name, *args):
    """Invoke a vlr_* entry point on torch's current stream; argument errors -> ValueError (reference behaviour),
    launch failures -> VlrError."""
    l = lib()
    conv = [ptr(a) if isinstance(a, torch.Tensor) else (C.byref(a) if isinstance(a, C.Structure) else a) for a in args]
    if _prof is not None and name in _prof:
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        rc = getattr(l, name)(*conv, stream())
        e.record()
        _prof[name].append((s, e, tuple(a for a in args if isinstance(a, (int, float)))))
    else:
        rc = getattr(l, name)(*conv, stream())
    if rc != 0:
        msg = l.vlr_last_error().decode()
        raise (ValueError if rc == 1 else VlrError)(msg)
    return rc


_splitk_ws = None


def ensure_splitk_workspace(device="cuda", force=False):
    """process-wide fp32 scratch of the GEMM dispatcher (include/vlr.h: split-K partials and the stream-K / rotation slabs of the
    persistent kernels, one 128 MiB slot per stream);
    allocated once and never freed, so the pointer registered in the library cannot dangle."""
    global _splitk_ws
    if _splitk_ws is None:
        _splitk_ws = torch.empty(1 << 30, dtype=torch.uint8, device=device)        # eight 128 MiB slots, one per stream
        force = True
    if force:
        rc = lib().vlr_gemm_set_splitk_workspace(_splitk_ws.data_ptr(), _splitk_ws.numel())
        if rc != 0:
            raise VlrError(lib().vlr_last_error().decode())
    return _splitk_ws


def helper(name, *args):
    return getattr(lib(), name)(*args)
