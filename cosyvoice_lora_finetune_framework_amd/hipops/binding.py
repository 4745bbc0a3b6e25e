"""ctypes binding of libcvft.so (C ABI: include/cvft.h).

The HIP library is the *only* compute backend of the product path: if it is missing or
a tensor is not on a ROCm device, the ops raise -- there is no CPU / eager fallback.
``import torch`` must precede loading so the library binds to torch's HIP runtime
(same libamdhip64.so.7 SONAME), which makes torch's current stream usable as hipStream_t.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.environ.get("CVFT_LIB_PATH") or os.path.join(_HERE, "libcvft.so")      # (override: A/B builds of the library)

F32, BF16 = 0, 1
ACT = {None: 0, "none": 0, "relu": 1, "silu": 2, "swish": 2, "gelu": 3, "gelu_erf": 3, "gelu_tanh": 4, "mish": 5}


GN_SPLIT = 8          # CVFT_GN_SPLIT (include/cvft.h)


class RankProb(C.Structure):
    """mirror of cvft_rank_prob (include/cvft.h)"""
    _fields_ = [("C", C.c_int), ("Wd", C.c_void_p), ("ldw", C.c_int), ("Rk", C.c_void_p), ("ldr", C.c_int),
                ("part", C.c_void_p), ("transpose_out", C.c_int), ("rows_per_block", C.c_int)]


class RankProbM(C.Structure):
    """mirror of cvft_rank_prob_m (include/cvft.h)"""
    _fields_ = [("M", C.c_int), ("C", C.c_int), ("Wd", C.c_void_p), ("ldw", C.c_int), ("Rk", C.c_void_p), ("ldr", C.c_int),
                ("part", C.c_void_p), ("transpose_out", C.c_int), ("rows_per_block", C.c_int)]


class BlockTailArgs(C.Structure):
    """mirror of cvft_block_tail_args (include/cvft.h)"""
    _fields_ = [("M", C.c_int), ("o", C.c_void_p), ("ldo", C.c_int), ("DI", C.c_int), ("x0", C.c_void_p),
                ("W_fwd", C.c_void_p), ("bo", C.c_void_p), ("x1", C.c_void_p),
                ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float),
                ("b1", C.c_void_p), ("F", C.c_int), ("b2", C.c_void_p),
                ("act", C.c_int), ("z", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("out", C.c_void_p),
                ("lean", C.c_int)]


class BlockTailBwdArgs(C.Structure):
    """mirror of cvft_block_tail_bwd_args (include/cvft.h)"""
    _fields_ = [("M", C.c_int), ("x1", C.c_void_p), ("dy", C.c_void_p), ("gamma", C.c_void_p), ("mean", C.c_void_p),
                ("rstd", C.c_void_p), ("z", C.c_void_p), ("W_bwd", C.c_void_p), ("F", C.c_int), ("DI", C.c_int),
                ("act", C.c_int), ("dx1", C.c_void_p), ("dout", C.c_void_p), ("lddo", C.c_int),
                ("attn_o", C.c_void_p), ("attn_o_lo", C.c_void_p), ("ldao", C.c_int), ("delta", C.c_void_p), ("T", C.c_int),
                ("lean", C.c_int)]


class BlockQkvArgs(C.Structure):
    """mirror of cvft_block_qkv_args (include/cvft.h)"""
    _fields_ = [("M", C.c_int), ("x", C.c_void_p), ("gamma", C.c_void_p), ("beta", C.c_void_p), ("eps", C.c_float),
                ("mean", C.c_void_p), ("rstd", C.c_void_p), ("W_fwd", C.c_void_p), ("bias", C.c_void_p), ("N3", C.c_int),
                ("A", C.c_void_p), ("lda", C.c_int), ("Bb", C.c_void_p), ("ldb", C.c_int),
                ("alpha", C.c_float), ("p", C.c_float), ("seed", C.c_void_p), ("sites", C.c_uint * 3),
                ("U", C.c_void_p), ("ldu", C.c_int), ("xd", C.c_void_p * 3), ("y_out", C.c_void_p), ("Y", C.c_void_p), ("ldy", C.c_int),
                ("wide", C.c_int)]


class BlockQkvBwdArgs(C.Structure):
    """mirror of cvft_block_qkv_bwd_args (include/cvft.h)"""
    _fields_ = [("M", C.c_int), ("dY", C.c_void_p), ("lddy", C.c_int), ("dres", C.c_void_p), ("x", C.c_void_p),
                ("gamma", C.c_void_p), ("mean", C.c_void_p), ("rstd", C.c_void_p), ("W_bwd", C.c_void_p), ("N3", C.c_int),
                ("At", C.c_void_p), ("ldat", C.c_int), ("Bbt", C.c_void_p), ("ldbt", C.c_int),
                ("alpha", C.c_float), ("p", C.c_float), ("seed", C.c_void_p), ("sites", C.c_uint * 3),
                ("V", C.c_void_p), ("ldv", C.c_int), ("dx", C.c_void_p), ("wide", C.c_int)]


class GemmArgs(C.Structure):
    _fields_ = [
        ("dtype", C.c_int), ("M", C.c_int), ("N", C.c_int), ("K", C.c_int),
        ("Tm", C.c_int), ("Tin", C.c_int), ("Tout", C.c_int), ("in_stride", C.c_int),
        ("out_stride", C.c_int), ("out_off", C.c_int), ("ntaps", C.c_int), ("tap_off", C.c_int * 4),
        ("in_len", C.c_void_p), ("out_len", C.c_void_p),
        ("A", C.c_void_p), ("lda", C.c_int),
        ("W", C.c_void_p), ("ldw", C.c_int),
        ("U", C.c_void_p), ("ldu", C.c_int), ("R", C.c_int),
        ("Bl", C.c_void_p), ("ldbl", C.c_int),
        ("bias", C.c_void_p), ("alpha", C.c_float), ("act", C.c_int),
        ("preact", C.c_void_p), ("ldp", C.c_int),
        ("dact_src", C.c_void_p), ("ldd", C.c_int), ("dact", C.c_int),
        ("residual", C.c_void_p), ("ldr", C.c_int),
        ("C", C.c_void_p), ("ldc", C.c_int),
        ("La", C.c_void_p), ("ldla", C.c_int), ("lora_scale", C.c_float), ("Uout", C.c_void_p),
        ("xdrop_p", C.c_float), ("xdrop_seed", C.c_void_p), ("xdrop_sites", C.c_uint * 4),
        ("odrop_p", C.c_float), ("odrop_site", C.c_uint),
    ]


_i, _f, _p, _i64 = C.c_int, C.c_float, C.c_void_p, C.c_int64

# name -> argtypes (restype is int unless noted).  Must list every symbol of include/cvft.h.
SIGNATURES = {
    "cvft_version": [],
    "cvft_set_concurrent_chains": [_i],
    "cvft_concurrent_chains": [],
    "cvft_last_error": [],
    "cvft_gemm": [C.POINTER(GemmArgs), _p],
    "cvft_gemm_last_kernel": [],
    "cvft_tn_accum": [_i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _p],
    "cvft_lora_rank_accum": [_i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _i, _p],
    "cvft_lora_rank_partial": [_i, _i, _i, _i, _p, _i, _p, _i, _p, _i, _i, _p],
    "cvft_lora_rank_partial_pair": [_i, _i, _i, _p, _i, _p, _i, _p, _i, _i, _p, _i, _p, _i, _p, _i, _p],
    "cvft_lora_rank_partial_multi": [_i, _i, _i, _p, _p],
    "cvft_lora_rank_partial_batch": [_i, _i, _p, _p],
    "cvft_debug_glds_stamps": [_p],
    "cvft_debug_stamp": [_p, C.c_int, _p],
    "cvft_debug_mfma_fp8_probe": [_p, _p, _p, _p],
    "cvft_quant_fp8_rows": [_i, _i, _p, _i, _p, _i, _p, _p],
    "cvft_gemm_fp8": [_p, _p, _i, _p, _p, _i, _p, _p],
    "cvft_lora_grad_reduce": [_i, _p, _i, _p],
    "cvft_lora_shadow": [_i, _p, _p, _p],
    "cvft_layernorm_fwd": [_i, _i, _i, _p, _p, _p, _f, _i, _f, _p, _p, _p, _p],
    "cvft_layernorm_bwd": [_i, _i, _i, _p, _p, _p, _p, _p, _i, _f, _p, _p, _p, _p],
    "cvft_layernorm_bwd_mask": [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, C.c_uint, _p, _p],
    "cvft_layernorm_bwd_mask_side": [_i, _i, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, C.c_uint, _p, _p, _i, _f, _p, _p],
    "cvft_groupnorm_mish_fwd": [_i, _i, _i, _i, _i, _p, _p, _p, _f, _p, _p, _i, _p, _p, _p, _p, _p],
    "cvft_groupnorm_mish_bwd": [_i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _i, _p, _p, _p, _p, _p],
    "cvft_attn_bias_fwd": [_i, _i, _i, _i, _p, _p, _p, _i, _p, _f, _i, _p, _i, _p, _p, _p],
    "cvft_attn_bias_bwd": [_i, _i, _i, _i, _p, _p, _p, _i, _p, _f, _i, _p, _p, _i, _p, _p, _p, _p, _p, _p, _i, _p],
    "cvft_attn_relpos_fwd": [_i, _i, _i, _i, _p, _p, _p, _i, _p, _i, _p, _p, _p, _i, _f, _p, _i, _p, _p, _f, _p, C.c_uint, _p],
    "cvft_attn_relpos_bwd": [_i, _i, _i, _i, _p, _p, _p, _i, _p, _i, _p, _p, _p, _i, _f, _p, _p, _i, _p, _p, _p, _p,
                             _p, _p, _i, _p, _f, _p, C.c_uint, _p],
    "cvft_embed_gather": [_i, _i, _i, _i, _p, _p, _p, _p, _p],
    "cvft_gather_rows": [_i, _i, _i, _p, _p, _f, _p, _p],
    "cvft_scatter_rows": [_i, _i, _i, _p, _p, _p, _p],
    "cvft_l2norm_rows": [_i, _i, _i, _p, _p, _p],
    "cvft_time_embed": [_i, _i, _i, _p, _p, _f, _p, _p],
    "cvft_act_fwd": [_i, _i64, _i, _p, _p, _p],
    "cvft_act_bwd": [_i, _i64, _i, _p, _p, _p, _p],
    "cvft_dropout_add": [_i, _i64, _p, _p, _p, _f, _p, C.c_uint, _p],
    "cvft_act_dropout": [_i, _i64, _i, _p, _p, _p, _f, _p, C.c_uint, _p],
    "cvft_skinny_dropout": [_i, _i, _i, _p, _i, _p, _i, _f, _p, _i, _f, _p, _p, _p, _p],
    "cvft_ln_skinny_dropout": [_i, _i, _i, _p, _p, _p, _f, _p, _p, _p, _p, _i, _f, _p, _i, _f, _p, _p, _p, _p],
    "cvft_lora_side_dgrad": [_i, _i, _i, _p, _i, _p, _i, _p, _i, _p, _i, _f, _p, _p, _p],
    "cvft_cfm_prepare": [_i, _i, _i, _p, _p, _p, _p, _p, _p, _p, _f, _f, _f, _i, _p, _p, _p, _p],
    "cvft_masked_mse_fwd": [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p],
    "cvft_masked_mse_bwd": [_i, _i, _i, _i, _p, _p, _p, _p, _p, _p, _p],
    "cvft_interp_linear_fwd": [_i, _i, _i, _i, _i, _p, _p, _p, _p],
    "cvft_interp_linear_bwd": [_i, _i, _i, _i, _i, _p, _p, _p, _p],
    "cvft_ce_fwd": [_i, _i, _i, _p, _i, _p, _p, _p, _f, _p],
    "cvft_ce_bwd": [_i, _i, _i, _p, _i, _p, _p, _p, _p, _i, _f, _p],
    "cvft_dwconv1d_fwd": [_i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p, _p],
    "cvft_dwconv1d_bwd": [_i, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p],
    "cvft_sumsq": [_i64, _p, _p, _p],
    "cvft_sumsq_ordered": [_i64, _p, _p, _p, _p],
    "cvft_adamw_flat": [_i64, _p, _p, _p, _p, _p, _f, _f, _f, _f, _p, _p, _f, _f, _p],
    "cvft_cast_f32_to_bf16": [_i64, _p, _p, _p],
    "cvft_block_tail_fwd": [C.POINTER(BlockTailArgs), _p],
    "cvft_block_tail_bwd": [C.POINTER(BlockTailBwdArgs), _p],
    "cvft_block_qkv_fwd": [C.POINTER(BlockQkvArgs), _p],
    "cvft_block_link_fwd": [C.POINTER(BlockTailArgs), C.POINTER(BlockQkvArgs), _p, _p],
    "cvft_block_link_bwd": [C.POINTER(BlockQkvBwdArgs), C.POINTER(BlockTailBwdArgs), _p, _p],
    "cvft_block_qkv_bwd": [C.POINTER(BlockQkvBwdArgs), _p],
}

_lib: Optional[C.CDLL] = None


class CvftError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load libcvft.so (once).  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise CvftError(f"HIP extension not built: {LIB_PATH} is missing "
                            f"(run `python -c 'import __graft_entry__ as g; g.build()'`); there is no CPU fallback")
        l = C.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(l, name)          # AttributeError if a declared symbol is not exported
            fn.argtypes = argtypes
            fn.restype = C.c_char_p if name in ("cvft_last_error", "cvft_gemm_last_kernel") else C.c_int
        _lib = l
    return _lib


CALLS = 0          # C-ABI calls that went through check() (bench.py: roofline.step.cabi_calls_per_step)


def check(rc: int, what: str = "") -> None:
    global CALLS
    CALLS += 1
    if rc != 0:
        msg = lib().cvft_last_error()
        raise CvftError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    raise CvftError(f"unsupported dtype {t.dtype} (fp32 / bf16 only)")


def ptr(t: Optional[torch.Tensor]):
    if t is None:
        return None
    if not t.is_cuda:
        raise CvftError("libcvft ops need ROCm device tensors; there is no CPU fallback")
    return C.c_void_p(t.data_ptr())


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)
