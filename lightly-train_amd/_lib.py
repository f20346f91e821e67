"""ctypes binding of liblt_amd.so (C ABI declared in include/lt_amd.h).

There is deliberately NO fallback: if the HIP library is missing or a call fails, we raise.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Any

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("LT_AMD_LIB") or os.path.join(HERE, "lib", "liblt_amd.so")   # LT_AMD_LIB: another build of the same ABI (A/B runs)

vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float


class GemmDesc(C.Structure):
    _fields_ = [
        ("A", vp), ("B", vp), ("M", i32), ("N", i32), ("K", i32), ("lda", i32), ("ldb", i32),
        ("trans_a", i32), ("trans_b", i32), ("epilogue", i32), ("C", vp), ("ldc", i32), ("C2", vp), ("ldc2", i32),
        ("bias", vp), ("gamma", vp), ("resid", vp), ("ldr", i32), ("aux", vp), ("ldaux", i32),
        ("alpha", f32), ("split_k", i32), ("force_kernel", i32), ("rowscale", vp), ("branch_scale", f32), ("workspace", vp), ("workspace_bytes", C.c_size_t),
        ("batch", i32), ("stride_a", i64), ("stride_b", i64), ("stride_c", i64), ("colsum", vp),
        ("ln_weight", vp), ("ln_bias", vp), ("ln_out", vp), ("ln_mean", vp), ("ln_rstd", vp), ("ln_eps", f32),
    ]


EPI_BF16, EPI_BF16_GELU, EPI_RESID, EPI_F32, EPI_BF16_GELUGRAD, EPI_F32_ACCUM = range(6)
ABI_VERSION = 5   # lt_abi_version() of include/lt_amd.h this binding was written against

# name -> argtypes (every function returns int status except lt_last_error)
SIGNATURES: dict[str, list[Any]] = {
    "lt_abi_version": [],
    "lt_device_info": [C.c_char_p, i32, C.POINTER(i32), C.POINTER(i32)],
    "lt_gemm_bf16": [C.POINTER(GemmDesc), vp],
    "lt_gemm_bf16_naive": [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "lt_matmul_f32": [vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "lt_resize_4tap": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "lt_im2col_bf16": [vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "lt_kl_fwd_bwd": [vp, vp, i32, f32, f32, vp, vp, i32, i32, i32, vp],
    "lt_symmetrize_bf16": [vp, vp, i32, i32, i32, vp],
    "lt_mixup": [vp, vp, f32, vp, i32, i64, vp],
    "lt_resample_tokens": [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp],
    "lt_rope_apply": [vp, vp, vp, i32, i32, i32, i32, i32, i32, vp],
    "lt_gelu_fwd_bf16": [vp, vp, i64, vp],
    "lt_gelu_bwd_bf16": [vp, vp, vp, i64, vp],
    "lt_swiglu_fwd": [vp, vp, i64, i32, vp],
    "lt_swiglu_bwd": [vp, vp, vp, i64, i32, vp],
    "lt_assemble_tokens": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "lt_assemble_tokens_bwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, vp],
    "lt_layernorm_fwd": [vp, vp, vp, vp, vp, vp, vp, i32, i32, f32, vp],
    "lt_layernorm_bwd": [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, i64, i32, i32, vp],
    "lt_layernorm_bwd_fused": [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, i64, vp, vp, vp, f32, vp, i32, i32, vp],
    "lt_layerscale_dgamma": [vp, vp, vp, vp, vp, vp, i32, i32, vp],
    "lt_layerscale_dgamma_batched": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i64, vp],
    "lt_layerscale_bwd": [vp, vp, vp, vp, vp, vp, vp, f32, i32, i32, vp],
    "lt_layernorm_bwd_rows": [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, vp, i32, i32, vp],
    "lt_layerscale_bwd_rows": [vp, vp, vp, vp, vp, vp, vp, vp, f32, i32, i32, vp],
    "lt_colsum_bf16": [vp, vp, i32, i32, vp],
    "lt_colsum_f32": [vp, vp, i32, i32, i32, vp],
    "lt_gather_rows": [vp, i32, vp, vp, vp, i32, i32, vp],
    "lt_scatter_add_rows": [vp, vp, vp, i32, i32, i32, vp],
    "lt_cast_f32_to_bf16": [vp, vp, i64, vp],
    "lt_cast_pad_rows": [vp, vp, i32, i32, i32, vp],
    "lt_unpad_accumulate": [vp, vp, i32, i32, i32, vp],
    "lt_fill_f32": [vp, f32, i64, vp],
    "lt_scale_f32": [vp, f32, i64, vp],
    "lt_attention_fwd": [vp, vp, vp, i32, i32, i32, i32, f32, vp],
    "lt_attention_bwd": [vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, f32, vp],
    "lt_l2norm_fwd": [vp, vp, vp, i32, i32, f32, vp],
    "lt_l2norm_bwd": [vp, vp, vp, vp, i32, i32, vp],
    "lt_weightnorm_fwd": [vp, vp, vp, i32, i32, vp],
    "lt_weightnorm_bwd": [vp, vp, vp, vp, vp, i32, i32, vp],
    "lt_softmax_center": [vp, vp, vp, i32, i32, f32, vp],
    "lt_center_ema": [vp, vp, f32, f32, i32, vp],
    "lt_softmax_stats_colsum": [vp, vp, vp, vp, i32, i32, f32, vp, i64, vp],
    "lt_ce_fwd_bwd_logits": [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, f32, f32, f32, vp, vp, i32, i32, vp],
    "lt_softmax_stats_colsum_bf16": [vp, vp, vp, vp, i32, i32, f32, vp, i64, vp],
    "lt_ce_fwd_bwd_logits_bf16": [vp, vp, vp, vp, vp, i32, vp, vp, vp, vp, f32, f32, f32, vp, vp, i32, i32, vp],
    "lt_ce_fwd_bwd": [vp, vp, vp, vp, vp, vp, f32, f32, vp, vp, i32, i32, vp],
    "lt_sk_exp": [vp, vp, i64, f32, vp],
    "lt_sk_iter": [vp, vp, i32, i32, f32, f32, vp],
    "lt_mse_fwd_bwd": [vp, vp, vp, i64, f32, vp, vp],
    "lt_koleo_fwd_bwd": [vp, i32, vp, vp, i32, i32, i32, f32, f32, vp, vp, vp],
    "lt_sumsq_f32": [vp, vp, i64, vp],
    "lt_adamw_flat": [vp, vp, vp, vp, vp, i64, vp, vp, vp, vp, i32, f32, f32, C.c_double, C.c_double, f32, i32, vp, f32, vp],
    "lt_ema_flat": [vp, vp, vp, i64, C.c_double, vp],
    "lt_lars_norms": [vp, vp, i64, vp, i32, vp, vp, vp],
    "lt_lars_flat": [vp, vp, vp, vp, i64, vp, vp, vp, vp, f32, f32, f32, f32, i32, f32, f32, i32, vp, f32, vp],
    "lt_comm_unique_id": [vp, i32],
    "lt_comm_init": [i32, i32, vp, i32],
    "lt_comm_allreduce_f32": [vp, i64, vp],
    "lt_comm_wait": [vp],
    "lt_comm_size": [],
    "lt_comm_destroy": [],
    "lt_reduce_begin": [vp, i64],
    "lt_reduce_begin_at": [vp, i64, C.c_int],
    "lt_reduce_flush": [vp],
    "lt_reduce_end": [vp],
    "lt_sgd_flat": [vp, vp, vp, vp, i64, vp, vp, vp, f32, f32, f32, f32, i32, i32, vp, f32, vp],
    "lt_im2col_nhwc_bf16": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "lt_col2im_nhwc_bf16": [vp, vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "lt_im2col_nchw_f32": [vp, vp, i32, i32, i32, i32, i32, i32, i32, i32, i32, vp],
    "lt_batchnorm_fwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, f32, i32, vp, vp],
    "lt_batchnorm_apply": [vp, vp, vp, vp, vp, vp, vp, i64, i32, i32, vp],
    "lt_batchnorm_bwd": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp],
    "lt_maxpool3x3s2_fwd": [vp, vp, vp, i32, i32, i32, i32, vp],
    "lt_maxpool3x3s2_bwd": [vp, vp, vp, i32, i32, i32, i32, vp],
    "lt_token_mean_bf16": [vp, vp, i32, i32, i32, vp],
    "lt_pool_bwd_add": [vp, vp, vp, i32, i32, i32, vp],
    "lt_add_bf16": [vp, vp, vp, i64, vp],
    "lt_batchnorm_stats": [vp, i64, i32, vp, vp, vp],
    "lt_batchnorm_fwd_from_sums": [vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, f32, f32, i32, vp],
    "lt_batchnorm_bwd_sums": [vp, vp, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp, vp],
    "lt_batchnorm_bwd_from_sums": [vp, vp, vp, vp, vp, vp, vp, i64, i32, vp, vp],
    "lt_aug_crop_resize": [vp, vp, vp, i32, i32, vp],
    "lt_aug_color": [vp, vp, i32, i32, vp],
    "lt_aug_finish": [vp, vp, vp, i32, i32, vp, vp, vp],
    "lt_roi_resample_tokens": [vp, vp, vp, vp, vp, vp, i32, i64, i32, i32, vp],
    "lt_roi_resample_tokens_bwd": [vp, vp, vp, vp, i32, i64, i32, i32, i32, vp],
    "lt_center_tokens": [vp, vp, vp, i32, i32, i32, vp],
    "lt_cka_fwd_bwd": [vp, vp, vp, vp, vp, i32, i32, i32, f32, vp],
    "lt_sample_block_masks": [vp, vp, vp, i32, i32, i32, i32, i32, i32, C.c_double, C.c_double, vp],
}

_lib: C.CDLL | None = None
_recording: Any = None   # ops.record_plan(): a proxy of the library that also logs every launch it forwards (LaunchPlan)


class LtAmdError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load liblt_amd.so; raise loudly when it has not been built (python __graft_entry__.py build)."""
    global _lib
    if _recording is not None:
        return _recording
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LtAmdError(
            f"HIP extension not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'`. "
            "There is no CPU fallback for the MI355X kernels."
        )
    lib = C.CDLL(LIB_PATH)
    lib.lt_last_error.restype = C.c_char_p
    lib.lt_last_error.argtypes = []
    lib.lt_attention_bwd_ws_floats.restype = C.c_int64
    lib.lt_attention_bwd_ws_floats.argtypes = [i32, i32, i32, i32]
    lib.lt_batchnorm_ws_floats.restype = C.c_int64
    lib.lt_batchnorm_ws_floats.argtypes = [i32]
    lib.lt_reduce_overflows.restype = C.c_int64
    lib.lt_reduce_overflows.argtypes = []
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = ABI mismatch, also loud
        fn.argtypes = argtypes
        fn.restype = C.c_int
    if lib.lt_abi_version() != ABI_VERSION:   # a stale build of another signature set would corrupt arguments silently
        raise LtAmdError(f"{LIB_PATH} has ABI version {lib.lt_abi_version()}, this package binds version {ABI_VERSION}: rebuild it")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().lt_last_error().decode("utf-8", "replace")
        raise LtAmdError(f"{what} failed with code {rc}: {msg}")
