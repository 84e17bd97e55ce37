"""ctypes binding of libdrakegpt_hip.so (include/drakegpt_hip.h).

There is NO fallback: if the library is missing or a symbol cannot be bound, importing this module
raises.  `import torch` happens first so that the HIP runtime the kernels bind to
(DT_NEEDED libamdhip64.so.7, no rpath) is the one PyTorch-ROCm already loaded.
"""
from __future__ import annotations

import ctypes as C
import os

import torch  # noqa: F401  (must precede the CDLL below: shares torch's libamdhip64)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libdrakegpt_hip.so")

DG_F32 = 0
DG_BF16 = 1
DG_FP8_E4M3 = 2
DG_FP8_E5M2 = 3
ABI_VERSION = 20


class GemmNtArgs(C.Structure):
    """struct dg_gemm_nt_args"""
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("B", C.c_void_p), ("ldb", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
        ("in_dtype", C.c_int32), ("out_dtype", C.c_int32),
        ("bias", C.c_void_p),
        ("relu", C.c_int32),
        ("relu_mask", C.c_void_p), ("ldmask", C.c_int64),
        ("residual", C.c_void_p), ("ldr", C.c_int64),
        ("dropout_p", C.c_float),
        ("rng_state", C.c_void_p),
        ("site", C.c_uint32),
        ("sign_bits_out", C.c_void_p),
        ("sign_bits", C.c_void_p),
        ("sign_bits_bytes", C.c_int64),
        ("colsum_part", C.c_void_p),
        ("colsum_ld", C.c_int64),
        ("colsum_rows", C.c_int32),
        ("b_dtype", C.c_int32),
        ("scale_a", C.c_void_p),
        ("scale_b", C.c_void_p),
        ("fp8_out", C.c_void_p), ("ld_fp8_out", C.c_int64),
        ("fp8_out_parts2", C.c_void_p),
        ("fp8_out_step", C.c_void_p),
        ("fp8_out_scale_inv", C.c_void_p),
        ("fp8_out_only", C.c_int32),
    ]


class BlockChainArgs(C.Structure):
    """struct dg_block_chain_args"""
    _fields_ = [
        ("mode", C.c_int32), ("M", C.c_int32), ("C", C.c_int32), ("eps", C.c_float),
        ("o", C.c_void_p), ("x", C.c_void_p),
        ("wproj", C.c_void_p), ("bproj", C.c_void_p), ("x1", C.c_void_p),
        ("ln2w", C.c_void_p), ("ln2b", C.c_void_p), ("mean2", C.c_void_p), ("rstd2", C.c_void_p), ("h2", C.c_void_p),
        ("w1", C.c_void_p), ("b1", C.c_void_p), ("f", C.c_void_p), ("sign_bits", C.c_void_p), ("sign_bits_bytes", C.c_int64),
        ("w2", C.c_void_p), ("b2", C.c_void_p), ("x2", C.c_void_p), ("x2_bf16", C.c_void_p),
        ("ln1w", C.c_void_p), ("ln1b", C.c_void_p), ("mean1", C.c_void_p), ("rstd1", C.c_void_p), ("h1", C.c_void_p),
        ("wqkv", C.c_void_p), ("qkv", C.c_void_p),
        ("dropout_p", C.c_float), ("rng_state", C.c_void_p), ("site_proj", C.c_uint32), ("site_ffn", C.c_uint32),
    ]


class BlockChainBwdArgs(C.Structure):
    """struct dg_block_chain_bwd_args"""
    _fields_ = [
        ("mode", C.c_int32), ("M", C.c_int32), ("C", C.c_int32), ("reserved", C.c_int32),
        ("dqkv", C.c_void_p), ("wqkvT", C.c_void_p), ("x", C.c_void_p), ("mean1", C.c_void_p), ("rstd1", C.c_void_p), ("ln1w", C.c_void_p),
        ("dresid1", C.c_void_p), ("dx1", C.c_void_p), ("g1", C.c_void_p),
        ("dln1w_part", C.c_void_p), ("dln1b_part", C.c_void_p), ("gbias1_part", C.c_void_p),
        ("g_in", C.c_void_p),
        ("w2T", C.c_void_p), ("sign_bits", C.c_void_p), ("sign_bits_bytes", C.c_int64), ("df", C.c_void_p), ("db1_part", C.c_void_p),
        ("w1T", C.c_void_p), ("x1", C.c_void_p), ("mean2", C.c_void_p), ("rstd2", C.c_void_p), ("ln2w", C.c_void_p),
        ("dresid2", C.c_void_p), ("dx2", C.c_void_p), ("g2", C.c_void_p),
        ("dln2w_part", C.c_void_p), ("dln2b_part", C.c_void_p), ("gbias2_part", C.c_void_p),
        ("wprojT", C.c_void_p), ("dout", C.c_void_p),
        ("part_stride", C.c_int64),
        ("dropout_p", C.c_float), ("rng_state", C.c_void_p), ("site_ffn_below", C.c_uint32), ("site_proj", C.c_uint32),
    ]


class AttnFp8Out(C.Structure):
    """dg_attn_fp8_out (include/drakegpt_hip.h)"""
    _fields_ = [("q8", C.c_void_p), ("hist3", C.c_void_p), ("step_state", C.c_void_p), ("scale_inv", C.c_void_p), ("only8", C.c_int)]


class TnProblem(C.Structure):
    """struct dg_tn_problem"""
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("B", C.c_void_p), ("ldb", C.c_int64),
        ("out", C.c_void_p), ("ldo", C.c_int64),
        ("R", C.c_int32), ("P", C.c_int32), ("Q", C.c_int32), ("reserved", C.c_int32),
        ("scale_a", C.c_void_p), ("scale_b", C.c_void_p),
    ]


_vp, _i, _i64, _f, _u32 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_uint32

# name -> argtypes; every entry of include/drakegpt_hip.h (tests/test_abi.py checks the two agree)
SIGNATURES = {
    "dg_version": [],
    "dg_state_advance": [_vp, _vp],
    "dg_batch_gather": [_vp, _i64, _vp, _vp, _vp, _i, _i, _vp],
    "dg_embed_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i64, _vp],
    "dg_batch_embed_fwd": [_vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _i64, _vp],
    "dg_embed_bwd": [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp],
    "dg_layernorm_fwd": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _f, _vp],
    "dg_layernorm_fwd_fp8_parts": [_i],
    "dg_layernorm_fwd_fp8": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _f, _vp, _i, _vp, _vp, _vp],
    "dg_layernorm_bwd": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _vp],
    "dg_layernorm_bwd_fused": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i64, _i, _i, _i, _vp, _i, _f, _vp, _u32, _vp, _vp],
    "dg_layernorm_bwd_fused_fp8": [_vp, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp, _vp, _i64, _i, _i, _i, _vp, _f, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "dg_gemm_nt": [C.POINTER(GemmNtArgs), _vp],
    "dg_gemm_nt_sign_bits_supported": [C.POINTER(GemmNtArgs)],
    "dg_gemm_nt_fp8_out_supported": [C.POINTER(GemmNtArgs)],
    "dg_gemm_nt_colsum_supported": [C.POINTER(GemmNtArgs)],
    "dg_gemm_nt_colsum_rows": [C.POINTER(GemmNtArgs)],
    "dg_gemm_nt_sign_bits_bytes": [_i, _i],
    "dg_fp8_amax": [_vp, _i, _i64, _vp, _i, _vp, _vp],
    "dg_fp8_quantize": [_vp, _i, _vp, _i, _i64, _vp, _i, _vp, _vp, _vp],
    "dg_fp8_quantize_delayed": [_vp, _i, _vp, _i, _i64, _vp, _vp, _vp, _vp],
    "dg_gemm_tn": [_vp, _i64, _vp, _i64, _vp, _i64, _i64, _i, _i, _i, _i, _i, _vp],
    "dg_gemm_tn_grouped": [C.POINTER(TnProblem), _i, _i, _vp, _i64, _vp],
    "dg_gemm_tn_grouped_workspace_bytes": [C.POINTER(TnProblem), _i],
    "dg_reduce_partials": [_vp, _i64, _i, _vp, _i64, _vp],
    "dg_colsum": [_vp, _i64, _i, _vp, _i64, _i, _i, _i, _vp],
    "dg_dropout_bwd_cast": [_vp, _i, _i64, _vp, _i64, _i, _i, _i, _f, _vp, _u32, _vp, _i64, _vp, _i64, _i, _vp],
    "dg_cast": [_vp, _i, _vp, _i, _i64, _vp],
    "dg_transpose_cast": [_vp, _i64, _vp, _i64, _i, _i, _i, _vp],
    "dg_transpose_cast_batched": [_vp, _i, _i, _i, _i, _vp],
    "dg_transpose_u8_batched": [_vp, _i, _i, _vp],
    "dg_attn_keep_bits_bytes": [_i, _i, _i, _i, _i],
    "dg_attn_fwd": [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _u32, _i, _vp, _i64, _vp],
    "dg_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _f, _f, _vp, _u32, _i, _vp, _i64, _vp],
    "dg_attn_fp8_out_supported": [_i, _i, _i, _i, _i],
    "dg_attn_fwd_fp8": [_vp, _vp, _vp, _i, _i, _i, _i, _f, _f, _vp, _u32, _i, _vp, _i64, C.POINTER(AttnFp8Out), _vp],
    "dg_attn_bwd_fp8": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _f, _f, _vp, _u32, _i, _vp, _i64, C.POINTER(AttnFp8Out), _vp],
    "dg_attn_bwd_workspace_bytes": [_i, _i, _i, _i, _i],
    "dg_attn_decode": [_vp, _vp, _i, _i, _i, _i, _i, _f, _i, _vp],
    "dg_cross_entropy": [_vp, _i, _i64, _vp, _vp, _vp, _i64, _i, _f, _vp, _i, _i, _vp],
    "dg_cross_entropy_fp8": [_vp, _i64, _vp, _vp, _vp, _i64, _f, _i, _i, _vp, _i64, _vp],
    "dg_cross_entropy_fused": [_vp, _i64, _vp, _vp, _vp, _i64, _i, _f, _i, _i, _vp, _i64, _i, _vp, _vp, _vp, _f, _vp],
    "dg_reduce_sum": [_vp, _i64, _f, _vp, _vp],
    "dg_softmax_rows": [_vp, _i64, _vp, _i64, _i, _i, _vp],
    "dg_adamw_step": [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _f, _vp, _i, _vp],
    "dg_block_chain_supported": [_i, _i],
    "dg_block_chain_fwd": [C.POINTER(BlockChainArgs), _vp],
    "dg_l2_warm": [_vp, _i64, _vp],
    "dg_pack_chain_weights": [_vp, _i64, _vp, _i, _i, _vp],
    "dg_pack_chain_weights_batched": [_vp, _i, _i, _vp],
    "dg_block_chain_bwd_supported": [_i, _i],
    "dg_block_chain_bwd": [C.POINTER(BlockChainBwdArgs), _vp],
}


def _load() -> C.CDLL:
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"drakegpt_amd: {LIB_PATH} is missing. Build it with `python -m drakegpt_amd.build` "
            "(needs hipcc; cross-compiles for gfx950 without a GPU). There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise RuntimeError(f"drakegpt_amd: {LIB_PATH} does not export {name}; rebuild it") from e
        fn.argtypes = argtypes
        fn.restype = C.c_int
    lib.dg_gemm_nt_sign_bits_bytes.restype = C.c_int64
    lib.dg_gemm_tn_grouped_workspace_bytes.restype = C.c_int64
    lib.dg_attn_bwd_workspace_bytes.restype = C.c_int64
    lib.dg_attn_keep_bits_bytes.restype = C.c_int64
    lib.dg_error_string.argtypes = [C.c_int]
    lib.dg_error_string.restype = C.c_char_p
    v = lib.dg_version()
    if v != ABI_VERSION:
        raise RuntimeError(f"drakegpt_amd: ABI version mismatch: library {v}, python {ABI_VERSION}; rebuild")
    return lib


lib = _load()


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib.dg_error_string(rc)
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")
