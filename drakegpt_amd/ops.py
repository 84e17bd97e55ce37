"""Tensor-level wrappers over the C ABI (one function per entry point of include/drakegpt_hip.h).

PyTorch is plumbing here: it owns device memory (caching allocator) and the stream; every
function extracts raw device pointers, enqueues the HIP kernel on torch's CURRENT stream and
returns (so the calls can be captured into a hipGraph via torch.cuda.graph).  Arguments are
validated on the host before a launch -- a kernel that faults can reset the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _lib
from ._lib import DG_BF16, DG_F32, DG_FP8_E4M3, DG_FP8_E5M2, GemmNtArgs, check, lib

Tensor = torch.Tensor

_DT = {torch.float32: DG_F32, torch.bfloat16: DG_BF16, torch.float8_e4m3fn: DG_FP8_E4M3, torch.float8_e5m2: DG_FP8_E5M2}
FP8_DTYPES = (torch.float8_e4m3fn, torch.float8_e5m2)
FP8_AMAX_PARTS = 256            # DG_FP8_AMAX_PARTS
ATTN_FP8_HIST = 3 * 64 * 32     # DG_ATTN_FP8_HIST


def dt_code(dtype: torch.dtype) -> int:
    try:
        return _DT[dtype]
    except KeyError:
        raise TypeError(f"drakegpt_amd: unsupported dtype {dtype} (float32 / bfloat16 / OCP float8 only)") from None


def _stream() -> int:
    return torch.cuda.current_stream().cuda_stream


def _p(t: Optional[Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk(t: Tensor, name: str, dtype=None, contiguous: bool = True) -> None:
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise RuntimeError(f"drakegpt_amd: {name} must be a tensor on the GPU (got {getattr(t, 'device', type(t))}); "
                           "the HIP path has no CPU fallback")
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f"drakegpt_amd: {name} must be {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise RuntimeError(f"drakegpt_amd: {name} must be contiguous")


def _ld(t: Tensor) -> int:
    """leading dimension (elements) of a 2-D row-major view with unit column stride"""
    if t.dim() != 2 or t.stride(1) != 1:
        raise RuntimeError("drakegpt_amd: expected a 2-D tensor with unit column stride")
    return t.stride(0) if t.shape[0] > 1 else max(t.stride(0), t.shape[1])


def check_ids(ids: Tensor, n: int, what: str) -> None:
    """raise IndexError unless 0 <= ids < n -- what nn.Embedding / F.cross_entropy do in the reference (ref: src/model.py:595,
    606).  The kernels clamp ids so that a bad one can never fault the GPU, which would otherwise turn a tokenizer / vocabulary
    mismatch into a plausible loss on aliased ids.  One device round trip: call it where ids ENTER (module forward, set_corpus,
    set_batch), never inside a captured step."""
    if ids.numel() == 0:
        return
    lo, hi = torch.aminmax(ids)
    lo, hi = int(lo), int(hi)
    if lo < 0 or hi >= n:
        raise IndexError(f"index out of range in self: {what} holds ids in [{lo}, {hi}], valid range is [0, {n})")


# ------------------------------------------------------------------------------------------
def new_rng_state(seed: int, device, step: int = 0) -> Tensor:
    """device uint32[4] = {seed_lo, seed_hi, step, 0} (stored as int32 bit patterns)."""
    vals = [seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF, step & 0xFFFFFFFF, 0]
    vals = [v - (1 << 32) if v >= (1 << 31) else v for v in vals]
    return torch.tensor(vals, dtype=torch.int32, device=device)


def state_advance(rng_state: Tensor) -> None:
    _chk(rng_state, "rng_state", torch.int32)
    check(lib.dg_state_advance(_p(rng_state), _stream()), "dg_state_advance")


def batch_gather(corpus: Tensor, offsets: Tensor, T: int, x: Optional[Tensor] = None, y: Optional[Tensor] = None):
    _chk(corpus, "corpus", torch.int64)
    _chk(offsets, "offsets", torch.int64)
    B = offsets.numel()
    if x is None:
        x = torch.empty((B, T), dtype=torch.int64, device=corpus.device)
    if y is None:
        y = torch.empty((B, T), dtype=torch.int64, device=corpus.device)
    check(lib.dg_batch_gather(_p(corpus), corpus.numel(), _p(offsets), _p(x), _p(y), B, T, _stream()), "dg_batch_gather")
    return x, y


def embed_fwd(idx: Tensor, tok: Tensor, pos: Optional[Tensor], out: Optional[Tensor] = None,
              onehot: Optional[Tensor] = None) -> Tensor:
    """x = tok[idx] + pos; `onehot` (bf16 [B*T, >= V]) additionally receives one-hot rows of idx"""
    _chk(idx, "idx", torch.int64)
    _chk(tok, "tok", torch.float32)
    B, T = idx.shape
    V, Cd = tok.shape
    if pos is not None:
        _chk(pos, "pos", torch.float32)
        if T > pos.shape[0]:
            raise IndexError(f"index out of range in self: sequence length {T} exceeds context_length {pos.shape[0]}")
    if out is None:
        out = torch.empty((B, T, Cd), dtype=torch.float32, device=idx.device)
    if onehot is not None:
        _chk(onehot, "onehot", torch.bfloat16, contiguous=False)
    check(lib.dg_embed_fwd(_p(idx), _p(tok), _p(pos), _p(out), B, T, Cd, V, _p(onehot), _ld(onehot) if onehot is not None else 0,
                           _stream()), "dg_embed_fwd")
    return out


def batch_embed_fwd(corpus: Tensor, offsets: Tensor, step_state: Optional[Tensor], ctl: Optional[Tensor], x_ids: Tensor, y_ids: Tensor,
                    tok: Tensor, pos: Optional[Tensor], onehot: Optional[Tensor] = None) -> Tensor:
    """get_batch + embedding in one launch: gathers row (step - ctl[0]) of the staged offset block [n_rows, B] from the resident
    corpus, writes the ids / targets into x_ids / y_ids [B, T] and returns x = tok[ids] + pos  [B, T, C]"""
    _chk(corpus, "corpus", torch.int64)
    _chk(offsets, "offsets", torch.int64)
    _chk(x_ids, "x_ids", torch.int64)
    _chk(y_ids, "y_ids", torch.int64)
    _chk(tok, "tok", torch.float32)
    B, T = x_ids.shape
    V, Cd = tok.shape
    if offsets.dim() != 2 or offsets.shape[1] != B or y_ids.shape != x_ids.shape:
        raise ValueError("batch_embed_fwd: offsets must be [n_rows, B] and y_ids shaped like x_ids")
    if (ctl is None) != (step_state is None):
        raise ValueError("batch_embed_fwd: ctl and step_state go together")
    if pos is not None:
        _chk(pos, "pos", torch.float32)
        if T > pos.shape[0]:
            raise IndexError(f"index out of range in self: sequence length {T} exceeds context_length {pos.shape[0]}")
    out = torch.empty((B, T, Cd), dtype=torch.float32, device=tok.device)
    if onehot is not None:
        _chk(onehot, "onehot", torch.bfloat16, contiguous=False)
    check(lib.dg_batch_embed_fwd(_p(corpus), corpus.numel(), _p(offsets), _p(step_state), _p(ctl), _p(x_ids), _p(y_ids), _p(tok), _p(pos),
                                 _p(out), B, T, Cd, V, _p(onehot), _ld(onehot) if onehot is not None else 0, _stream()), "dg_batch_embed_fwd")
    return out


def embed_bwd(idx: Tensor, dx: Tensor, dtok: Optional[Tensor], dpos: Optional[Tensor], V: Optional[int] = None) -> None:
    _chk(idx, "idx", torch.int64)
    _chk(dx, "dx")                         # fp32, or the engine's bf16 gradient stream
    B, T = idx.shape
    if dtok is not None:
        _chk(dtok, "dtok", torch.float32)
        V, Cd = dtok.shape
    else:
        Cd = dx.shape[-1]
    if dpos is not None:
        _chk(dpos, "dpos", torch.float32)
        if dpos.shape[0] != T:
            raise RuntimeError("dpos must be the [T, C] slice of the position gradient")
    check(lib.dg_embed_bwd(_p(idx), _p(dx), dt_code(dx.dtype), _p(dtok), _p(dpos), B, T, Cd, V, _stream()), "dg_embed_bwd")


def layernorm_fwd(x: Tensor, gamma: Tensor, beta: Tensor, out_dtype: torch.dtype, eps: float = 1e-5):
    _chk(x, "x", torch.float32)
    _chk(gamma, "gamma", torch.float32)
    _chk(beta, "beta", torch.float32)
    Cd = x.shape[-1]
    M = x.numel() // Cd
    y = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    mean = torch.empty((M,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((M,), dtype=torch.float32, device=x.device)
    check(lib.dg_layernorm_fwd(_p(x), _p(gamma), _p(beta), _p(y), dt_code(out_dtype), _p(mean), _p(rstd), M, Cd, eps, _stream()),
          "dg_layernorm_fwd")
    return y, mean, rstd


def layernorm_fwd_fp8_parts(M: int) -> int:
    """partial maxima per history slot of layernorm_fwd_fp8 (one per workgroup of its launch)"""
    return int(lib.dg_layernorm_fwd_fp8_parts(int(M)))


def layernorm_fwd_fp8(x: Tensor, gamma: Tensor, beta: Tensor, parts2: Tensor, step_state: Tensor, want_bf16: bool = True, eps: float = 1e-5):
    """LayerNorm whose output leaves as e4m3 with delayed scaling (parts2: fp32 [2 * layernorm_fwd_fp8_parts(M)], this call site's
    history).  Returns (y bf16 -- unwritten and marked dg_unwritten when want_bf16 is False --, mean, rstd, q8, scale_inv [1])."""
    _chk(x, "x", torch.float32)
    _chk(gamma, "gamma", torch.float32)
    _chk(beta, "beta", torch.float32)
    _chk(parts2, "parts2", torch.float32)
    _chk(step_state, "step_state", torch.int32)
    Cd = x.shape[-1]
    M = x.numel() // Cd
    n = layernorm_fwd_fp8_parts(M)
    if parts2.numel() != 2 * n:
        raise RuntimeError(f"layernorm_fwd_fp8: parts2 must hold 2 x {n} floats")
    y = torch.empty(x.shape, dtype=torch.bfloat16, device=x.device)
    q8 = torch.empty(x.shape, dtype=torch.float8_e4m3fn, device=x.device)
    mean = torch.empty((M,), dtype=torch.float32, device=x.device)
    rstd = torch.empty((M,), dtype=torch.float32, device=x.device)
    sinv = torch.empty((1,), dtype=torch.float32, device=x.device)
    check(lib.dg_layernorm_fwd_fp8(_p(x), _p(gamma), _p(beta), _p(y) if want_bf16 else None, _p(q8), _p(mean), _p(rstd), M, Cd, eps, _p(parts2), n,
                                   _p(step_state), _p(sinv), _stream()), "dg_layernorm_fwd_fp8")
    if not want_bf16:
        y.dg_unwritten = True
    return y, mean, rstd, q8, sinv


def layernorm_bwd(dy: Tensor, x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor, dresid: Optional[Tensor],
                  dgamma_part: Tensor, dbeta_part: Tensor, part_stride: int, n_partials: int,
                  dx: Optional[Tensor] = None) -> Tensor:
    _chk(dy, "dy")                        # fp32, or bf16 straight from a dX GEMM
    _chk(x, "x", torch.float32)
    if dresid is not None:
        _chk(dresid, "dresid", torch.float32)
    Cd = x.shape[-1]
    M = x.numel() // Cd
    if dx is None:
        dx = torch.empty_like(x)
    check(lib.dg_layernorm_bwd(_p(dy), dt_code(dy.dtype), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dresid), _p(dx), _p(dgamma_part), _p(dbeta_part),
                               part_stride, n_partials, M, Cd, _stream()), "dg_layernorm_bwd")
    return dx


def layernorm_bwd_fused(dy: Tensor, x: Tensor, gamma: Tensor, mean: Tensor, rstd: Tensor, dresid: Optional[Tensor],
                        dgamma_part: Tensor, dbeta_part: Tensor, part_stride: int, n_partials: int,
                        g_dtype: torch.dtype, p: float, rng_state: Optional[Tensor], site: int, gbias_part: Optional[Tensor],
                        stream_dtype: torch.dtype = torch.float32, fp8_out=None, fp8_out_only: bool = False):
    """layernorm_bwd that also emits g = dropout_bwd(dx) in g_dtype and its column-sum partials; returns (dx, g).
    stream_dtype: type of dresid (in) and dx (out), fp32 or bf16 (the engine's bf16 gradient stream: bf16 dy and g only).
    fp8_out = (parts2 fp32 [2 * FP8_AMAX_PARTS], step_state): g (bf16) also leaves as e5m2 with delayed scaling -- the operand of
    the fp8 dX GEMM that runs next; n_partials must be FP8_AMAX_PARTS.  Returns (dx, g, g8, scale_inv [1]) then."""
    _chk(dy, "dy")
    _chk(x, "x", torch.float32)
    if dresid is not None:
        _chk(dresid, "dresid", stream_dtype)
    Cd = x.shape[-1]
    M = x.numel() // Cd
    dx = torch.empty(x.shape, dtype=stream_dtype, device=x.device)
    g = torch.empty(x.shape, dtype=g_dtype, device=x.device)
    if fp8_out is not None:
        parts2, step_state = fp8_out
        _chk(parts2, "fp8_out parts2", torch.float32)
        _chk(step_state, "fp8_out step_state", torch.int32)
        if g_dtype != torch.bfloat16 or parts2.numel() != 2 * FP8_AMAX_PARTS or n_partials != FP8_AMAX_PARTS:
            raise RuntimeError("layernorm_bwd_fused: fp8_out needs a bf16 g, a [2 * FP8_AMAX_PARTS] history and n_partials == FP8_AMAX_PARTS")
        g8 = torch.empty(x.shape, dtype=torch.float8_e5m2, device=x.device)
        sinv = torch.empty((1,), dtype=torch.float32, device=x.device)
        check(lib.dg_layernorm_bwd_fused_fp8(_p(dy), dt_code(dy.dtype), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dresid), _p(dx), dt_code(stream_dtype),
                                             _p(dgamma_part), _p(dbeta_part), part_stride, n_partials, M, Cd, _p(g), float(p),
                                             _p(rng_state) if p > 0.0 else None, site, _p(gbias_part), _p(g8), _p(parts2), _p(step_state), _p(sinv),
                                             1 if fp8_out_only else 0, _stream()), "dg_layernorm_bwd_fused_fp8")
        if fp8_out_only:
            g.dg_unwritten = True           # (its consumers read g8)
        return dx, g, g8, sinv
    check(lib.dg_layernorm_bwd_fused(_p(dy), dt_code(dy.dtype), _p(x), _p(gamma), _p(mean), _p(rstd), _p(dresid), _p(dx), dt_code(stream_dtype), _p(dgamma_part),
                                     _p(dbeta_part), part_stride, n_partials, M, Cd, _p(g), dt_code(g_dtype), float(p),
                                     _p(rng_state) if p > 0.0 else None, site, _p(gbias_part), _stream()), "dg_layernorm_bwd_fused")
    return dx, g


def layernorm_bwd_fused_supported(C: int) -> bool:
    return C % 4 == 0 and C <= 1024


def gemm_nt(A: Tensor, Bm: Tensor, out_dtype: torch.dtype, *, N: Optional[int] = None, K: Optional[int] = None,
            bias: Optional[Tensor] = None, relu: bool = False, relu_mask: Optional[Tensor] = None,
            residual: Optional[Tensor] = None, dropout_p: float = 0.0, rng_state: Optional[Tensor] = None,
            site: int = 0, out: Optional[Tensor] = None, sign_bits_out: Optional[Tensor] = None,
            sign_bits: Optional[Tensor] = None, colsum_part: Optional[Tensor] = None,
            scale_a: Optional[Tensor] = None, scale_b: Optional[Tensor] = None, fp8_out=None, fp8_out_only: bool = False) -> Tensor:
    """out[M,N] = epilogue(A[M,K] @ Bm[N,K]^T).  A/Bm may carry padding columns beyond K (ld > K).
    fp8_out_only: with fp8_out, do not write `out` (nobody reads the bf16 form); the returned tensor is marked dg_unwritten.
    fp8_out: (q8 [M, N] float8_e4m3fn -- float8_e5m2 in the dX direction --, parts2 fp32 [2 * FP8_AMAX_PARTS], step_state, scale_inv fp32 [1]) -- the epilogue also
    writes the output as e4m3 with delayed scaling (what fp8_quantize_delayed would make of it); only where
    gemm_nt_fp8_out_supported(M, N, K).
    fp8: A float8_e4m3fn (activations) or float8_e5m2 (gradients), Bm float8_e4m3fn, scale_a / scale_b the device scalars
    fp8_quantize returned (out = epilogue(scale_a * scale_b * A @ Bm^T)); K % 128 == 0.
    sign_bits_out / sign_bits: opaque uint8 buffer (new_sign_bits) holding one bit per element, out > 0.
    colsum_part: fp32 [gemm_nt_colsum_rows(...), N] (rows may be strided): partial rows of the column sums of out."""
    _chk(A, "A", contiguous=False)
    _chk(Bm, "B", contiguous=False)
    fp8 = A.dtype in FP8_DTYPES
    if fp8:
        if Bm.dtype != torch.float8_e4m3fn or scale_a is None or scale_b is None:
            raise TypeError("gemm_nt: fp8 operands need a float8_e4m3fn B operand and both dequantisation scales")
        _chk(scale_a, "scale_a", torch.float32)
        _chk(scale_b, "scale_b", torch.float32)
    elif A.dtype != Bm.dtype:
        raise TypeError(f"gemm_nt: operand dtypes differ ({A.dtype} vs {Bm.dtype})")
    M = A.shape[0]
    K = A.shape[1] if K is None else K
    N = Bm.shape[0] if N is None else N
    if out is None:
        out = torch.empty((M, N), dtype=out_dtype, device=A.device)
    a = GemmNtArgs()
    a.A, a.lda = _p(A), _ld(A)
    a.B, a.ldb = _p(Bm), _ld(Bm)
    a.C, a.ldc = _p(out), _ld(out)
    a.M, a.N, a.K = M, N, K
    a.in_dtype, a.out_dtype = dt_code(A.dtype), dt_code(out.dtype)
    if fp8:
        a.b_dtype, a.scale_a, a.scale_b = dt_code(Bm.dtype), _p(scale_a), _p(scale_b)
    if bias is not None:
        _chk(bias, "bias", torch.float32)
        if bias.numel() != N:
            raise RuntimeError("gemm_nt: bias size mismatch")
    a.bias = _p(bias)
    a.relu = 1 if relu else 0
    if relu_mask is not None:
        if fp8:
            raise TypeError("gemm_nt: the tensor-valued relu_mask is not offered for fp8 operands (use sign_bits)")
        _chk(relu_mask, "relu_mask", A.dtype, contiguous=False)
        a.relu_mask, a.ldmask = _p(relu_mask), _ld(relu_mask)
    if residual is not None:
        _chk(residual, "residual", torch.float32, contiguous=False)
        a.residual, a.ldr = _p(residual), _ld(residual)
    a.dropout_p = float(dropout_p)
    a.rng_state = _p(rng_state) if dropout_p > 0.0 else None
    a.site = site
    for name, t in (("sign_bits_out", sign_bits_out), ("sign_bits", sign_bits)):
        if t is not None:
            _chk(t, name, torch.uint8)
            setattr(a, name, _p(t))
            a.sign_bits_bytes = t.numel()
    if colsum_part is not None:
        _chk(colsum_part, "colsum_part", torch.float32, contiguous=False)
        if colsum_part.shape[1] < N:
            raise RuntimeError("gemm_nt: colsum_part must have N columns")
        a.colsum_part, a.colsum_ld, a.colsum_rows = _p(colsum_part), _ld(colsum_part), colsum_part.shape[0]
    if fp8_out is not None:
        q8, parts2, step_state, q_scale_inv = fp8_out
        _chk(q8, "fp8_out", torch.float8_e5m2 if A.dtype == torch.float8_e5m2 else torch.float8_e4m3fn, contiguous=False)
        _chk(parts2, "fp8_out parts2", torch.float32)
        _chk(q_scale_inv, "fp8_out scale_inv", torch.float32)
        if q8.shape != (M, N) or parts2.numel() < 2 * FP8_AMAX_PARTS:
            raise RuntimeError("gemm_nt: fp8_out must be [M, N] with a [2 * FP8_AMAX_PARTS] history")
        a.fp8_out, a.ld_fp8_out = _p(q8), _ld(q8)
        a.fp8_out_parts2, a.fp8_out_step, a.fp8_out_scale_inv = _p(parts2), _p(step_state), _p(q_scale_inv)
        if fp8_out_only:
            # the bf16 form stays unwritten (the tensor is returned as the carrier of the fp8 copy's attributes only); a consumer
            # that would read it must refuse: dg_unwritten marks it
            a.fp8_out_only = 1
            out.dg_unwritten = True
    elif fp8_out_only:
        raise RuntimeError("gemm_nt: fp8_out_only needs fp8_out")
    check(lib.dg_gemm_nt(C.byref(a), _stream()), "dg_gemm_nt")
    return out


def gemm_nt_fp8_out_supported(M: int, N: int, K: int, grad: bool = False) -> bool:
    """can dg_gemm_nt also emit its output as fp8 (fp8_out)?  grad = False: the bias + ReLU + sign-bit form on e4m3 operands
    (e4m3 copy); grad = True: the sign-bit-masked dX form with column sums on e5m2 gradients (e5m2 copy)"""
    a = GemmNtArgs()
    a.M, a.N, a.K = M, N, K
    a.in_dtype, a.out_dtype = dt_code(torch.float8_e5m2 if grad else torch.float8_e4m3fn), dt_code(torch.bfloat16)
    a.ldc = N
    a.C = 16                                   # any non-null, aligned pointer value: the query only looks at which operands are present
    if grad:
        a.sign_bits = a.colsum_part = 16
    else:
        a.bias = a.sign_bits_out = 16
        a.relu = 1
    return bool(lib.dg_gemm_nt_fp8_out_supported(C.byref(a)))


def gemm_nt_sign_bits_supported(dtype: torch.dtype, N: int, K: int, in_dtype: Optional[torch.dtype] = None) -> bool:
    """can dg_gemm_nt emit / consume the one-bit-per-element ReLU mask for this problem?  (dtype: output / activation type;
    in_dtype: operand type when it differs -- the fp8 forms)"""
    a = GemmNtArgs()
    a.M, a.N, a.K = 128, N, K
    a.out_dtype = dt_code(dtype)
    a.in_dtype = dt_code(in_dtype or dtype)
    return bool(lib.dg_gemm_nt_sign_bits_supported(C.byref(a)))


def gemm_nt_colsum_rows(dtype: torch.dtype, M: int, N: int, K: int, in_dtype: Optional[torch.dtype] = None) -> int:
    """partial rows the sign_bits-consuming dg_gemm_nt of this shape writes into colsum_part (0: column sums not offered)"""
    a = GemmNtArgs()
    a.M, a.N, a.K = M, N, K
    a.out_dtype = dt_code(dtype)
    a.in_dtype = dt_code(in_dtype or dtype)
    a.ldc = N
    a.sign_bits = 16        # any non-null, aligned pointer value: the query only looks at which operands are present
    return int(lib.dg_gemm_nt_colsum_rows(C.byref(a)))


def fp8_quantize(x: Tensor, fmt: torch.dtype, seg: Optional[Tensor] = None, n_seg: int = 0, out: Optional[Tensor] = None,
                 scale_inv: Optional[Tensor] = None, amax: Optional[Tensor] = None, reuse_amax: bool = False):
    """per-tensor (per-segment) just-in-time scaling: q = fp8(x * FMAX / amax(x)) and the dequantisation factor amax / FMAX.
    x: contiguous bf16 / fp32, numel % 8 == 0.  seg: int64 device table [n_seg, 2] {first element, count} over x.view(-1)
    (all weight matrices of a step in two launches).  Returns (q with x's shape and dtype `fmt`, scale_inv [n_seg or 1])."""
    _chk(x, "x")
    if fmt not in FP8_DTYPES:
        raise TypeError("fp8_quantize: fmt must be torch.float8_e4m3fn or torch.float8_e5m2")
    ns = n_seg if seg is not None else 1
    if seg is not None:
        _chk(seg, "seg", torch.int64)
    if out is None:
        out = torch.empty(x.shape, dtype=fmt, device=x.device)
    if scale_inv is None:
        scale_inv = torch.empty((ns,), dtype=torch.float32, device=x.device)
    if amax is None:
        amax = torch.empty((ns * FP8_AMAX_PARTS,), dtype=torch.float32, device=x.device)      # partial maxima, reduced by the cast kernel
    n = x.numel()
    if not reuse_amax:          # reuse_amax: `amax` already holds the partial maxima of the same values (W^T after W)
        check(lib.dg_fp8_amax(_p(x), dt_code(x.dtype), n, _p(seg), ns, _p(amax), _stream()), "dg_fp8_amax")
    check(lib.dg_fp8_quantize(_p(x), dt_code(x.dtype), _p(out), dt_code(fmt), n, _p(seg), ns, _p(amax), _p(scale_inv), _stream()),
          "dg_fp8_quantize")
    return out, scale_inv


def fp8_quantize_delayed(x: Tensor, fmt: torch.dtype, parts2: Tensor, rng_state: Tensor):
    """one-pass delayed scaling: scale from the amax this call site recorded one step ago (parts2 [2 * FP8_AMAX_PARTS] fp32, the
    slot chosen by the device-side step word rng_state[2]); records this tensor's amax for the next step.
    Returns (q, scale_inv [1])."""
    _chk(x, "x")
    _chk(parts2, "parts2", torch.float32)
    _chk(rng_state, "rng_state", torch.int32)
    if parts2.numel() != 2 * FP8_AMAX_PARTS:
        raise RuntimeError("fp8_quantize_delayed: parts2 must hold 2 x FP8_AMAX_PARTS floats")
    out = torch.empty(x.shape, dtype=fmt, device=x.device)
    scale_inv = torch.empty((1,), dtype=torch.float32, device=x.device)
    check(lib.dg_fp8_quantize_delayed(_p(x), dt_code(x.dtype), _p(out), dt_code(fmt), x.numel(), _p(parts2), _p(rng_state), _p(scale_inv),
                                      _stream()), "dg_fp8_quantize_delayed")
    return out, scale_inv


def new_sign_bits(M: int, N: int, device) -> Tensor:
    """buffer for gemm_nt(sign_bits_out=...) of an [M, N] output"""
    return torch.empty(int(lib.dg_gemm_nt_sign_bits_bytes(M, N)), dtype=torch.uint8, device=device)


def gemm_tn(A: Tensor, Bm: Tensor, out_part: Tensor, split_stride: int, n_splits: int, P: int, Q: int, ldo: Optional[int] = None) -> None:
    """partials of dW[P,Q] = sum_r A[r,:P]^T B[r,:Q] into out_part (+ s*split_stride)."""
    _chk(A, "A", contiguous=False)
    _chk(Bm, "B", contiguous=False)
    _chk(out_part, "out_part", torch.float32, contiguous=False)
    if A.dtype != Bm.dtype or A.shape[0] != Bm.shape[0]:
        raise RuntimeError("gemm_tn: operand mismatch")
    check(lib.dg_gemm_tn(_p(A), _ld(A), _p(Bm), _ld(Bm), _p(out_part), Q if ldo is None else ldo, split_stride, n_splits,
                         A.shape[0], P, Q, dt_code(A.dtype), _stream()), "dg_gemm_tn")


def _tn_problem_array(problems):
    """problems: (A, B, out, P, Q) with bf16 operands, or (A e5m2, B e4m3, out, P, Q, scale_a, scale_b).  Returns (array, dtype code)"""
    from ._lib import TnProblem
    arr = (TnProblem * len(problems))()
    f8 = len(problems[0]) == 7
    for t, pr in zip(arr, problems):
        if (len(pr) == 7) != f8:
            raise RuntimeError("gemm_tn_grouped: bf16 and fp8 problems go into separate calls")
        A, Bm, out, P, Q = pr[:5]
        if f8:
            _chk(A, "A", torch.float8_e5m2, contiguous=False)
            _chk(Bm, "B", torch.float8_e4m3fn, contiguous=False)
            _chk(pr[5], "scale_a", torch.float32)
            _chk(pr[6], "scale_b", torch.float32)
            t.scale_a, t.scale_b = _p(pr[5]), _p(pr[6])
        else:
            _chk(A, "A", torch.bfloat16, contiguous=False)
            _chk(Bm, "B", torch.bfloat16, contiguous=False)
        _chk(out, "out", torch.float32, contiguous=False)
        if A.shape[0] != Bm.shape[0] or out.numel() < P * Q:
            raise RuntimeError("gemm_tn_grouped: operand mismatch")
        t.A, t.lda, t.B, t.ldb, t.out, t.ldo = _p(A), _ld(A), _p(Bm), _ld(Bm), _p(out), Q
        t.R, t.P, t.Q, t.reserved = A.shape[0], P, Q, 0
    return arr, (DG_FP8_E5M2 if f8 else DG_BF16)


def gemm_tn_grouped_workspace(problems, device) -> Tensor:
    """zero-filled split-K workspace for gemm_tn_grouped on these problems (allocate once, pass to every call)"""
    arr, _ = _tn_problem_array(problems)
    return torch.zeros(max(16, int(lib.dg_gemm_tn_grouped_workspace_bytes(arr, len(problems)))), dtype=torch.uint8, device=device)


def gemm_tn_grouped(problems, workspace: Optional[Tensor] = None) -> None:
    """every dW of a backward pass in one launch: problems = [(A [R,>=P], B [R,>=Q], out [P,Q] fp32, P, Q), ...];
    out_i = A_i[:, :P]^T B_i[:, :Q] over all R rows (bf16 operands, R % 64 == 0).  fp8 form: [(A e5m2, B e4m3, out, P, Q,
    scale_a [1], scale_b [1]), ...] with R % 128 == 0: out_i = scale_a * scale_b * A_i^T B_i.  The caller keeps the operands alive.
    `workspace` (gemm_tn_grouped_workspace) lets the kernel split the contraction of every tile in two."""
    arr, code = _tn_problem_array(problems)
    if workspace is not None:
        _chk(workspace, "workspace", torch.uint8)
    check(lib.dg_gemm_tn_grouped(arr, len(problems), code, _p(workspace) if workspace is not None else None,
                                 workspace.numel() if workspace is not None else 0, _stream()), "dg_gemm_tn_grouped")


def reduce_partials(part: Tensor, stride: int, n_partials: int, out: Tensor, n: int) -> None:
    _chk(part, "partials", torch.float32, contiguous=False)
    _chk(out, "out", torch.float32, contiguous=False)
    check(lib.dg_reduce_partials(_p(part), stride, n_partials, _p(out), n, _stream()), "dg_reduce_partials")


def colsum(A: Tensor, part: Tensor, part_stride: int, n_partials: int, N: Optional[int] = None) -> None:
    _chk(A, "A", contiguous=False)
    M = A.shape[0]
    N = A.shape[1] if N is None else N
    check(lib.dg_colsum(_p(A), _ld(A), dt_code(A.dtype), _p(part), part_stride, n_partials, M, N, _stream()), "dg_colsum")


def dropout_bwd_cast(dy: Tensor, out_dtype: Optional[torch.dtype], p: float, rng_state: Optional[Tensor], site: int,
                     relu_mask: Optional[Tensor] = None, colsum_part: Optional[Tensor] = None, part_stride: int = 0,
                     n_partials: int = 0, want_g: bool = True) -> Optional[Tensor]:
    _chk(dy, "dy", contiguous=False)          # fp32, or bf16 (the engine's gradient stream)
    M, N = dy.shape
    g = torch.empty((M, N), dtype=out_dtype, device=dy.device) if want_g else None
    if relu_mask is not None:
        _chk(relu_mask, "relu_mask", torch.float32, contiguous=False)
    check(lib.dg_dropout_bwd_cast(_p(dy), dt_code(dy.dtype), _ld(dy), _p(g), N, dt_code(out_dtype) if want_g else DG_F32, M, N, float(p),
                                  _p(rng_state) if p > 0.0 else None, site,
                                  _p(relu_mask), _ld(relu_mask) if relu_mask is not None else 0,
                                  _p(colsum_part), part_stride, n_partials, _stream()), "dg_dropout_bwd_cast")
    return g


def cast(x: Tensor, out_dtype: torch.dtype, out: Optional[Tensor] = None) -> Tensor:
    _chk(x, "x")
    if out is None:
        out = torch.empty(x.shape, dtype=out_dtype, device=x.device)
    check(lib.dg_cast(_p(x), dt_code(x.dtype), _p(out), dt_code(out.dtype), x.numel(), _stream()), "dg_cast")
    return out


def transpose_cast(W: Tensor, out_dtype: torch.dtype, ldo: Optional[int] = None, out: Optional[Tensor] = None) -> Tensor:
    """W [R,C] fp32 -> W^T [C, ldo] (ldo >= R, zero padded) in out_dtype."""
    _chk(W, "W", torch.float32, contiguous=False)
    R, Cc = W.shape
    if ldo is None:
        g = 8 if out_dtype == torch.bfloat16 else 4
        ldo = (R + g - 1) // g * g
    if out is None:
        out = torch.empty((Cc, ldo), dtype=out_dtype, device=W.device)
    check(lib.dg_transpose_cast(_p(W), _ld(W), _p(out), ldo, dt_code(out.dtype), R, Cc, _stream()), "dg_transpose_cast")
    return out


def make_transpose_table(pairs, device) -> tuple:
    """descriptor table for transpose_cast_batched: pairs = [(W [R,C] fp32, Wt [C, ldo])]"""
    rows, first = [], 0
    for W, Wt in pairs:
        R, Cc = W.shape
        ldo = Wt.shape[1]
        tiles_x, tiles_y = (Cc + 63) // 64, (ldo + 63) // 64
        rows.append([W.data_ptr(), Wt.data_ptr(), _ld(W), ldo, R, Cc, first, tiles_x])
        first += tiles_x * tiles_y
    return torch.tensor(rows, dtype=torch.int64, device=device), len(rows), first


def make_transpose_u8_table(pairs, device) -> tuple:
    """descriptor table for transpose_u8_batched: pairs = [(W8 [R, C] one-byte elements, Wt8 [C, ldo >= R])]"""
    rows, first = [], 0
    for W, Wt in pairs:
        R, Cc = W.shape
        ldo = Wt.shape[1]
        tiles_x, tiles_y = (Cc + 127) // 128, (ldo + 127) // 128
        rows.append([W.data_ptr(), Wt.data_ptr(), _ld(W), ldo, R, Cc, first, tiles_x])
        first += tiles_x * tiles_y
    return torch.tensor(rows, dtype=torch.int64, device=device), len(rows), first


def transpose_u8_batched(table: Tensor, n_desc: int, total_tiles: int) -> None:
    _chk(table, "table", torch.int64)
    check(lib.dg_transpose_u8_batched(_p(table), n_desc, total_tiles, _stream()), "dg_transpose_u8_batched")


def transpose_cast_batched(table: Tensor, n_desc: int, total_tiles: int, dtype: torch.dtype, in_dtype: torch.dtype = torch.float32) -> None:
    _chk(table, "table", torch.int64)
    check(lib.dg_transpose_cast_batched(_p(table), n_desc, total_tiles, dt_code(in_dtype), dt_code(dtype), _stream()), "dg_transpose_cast_batched")


def attn_fp8_out_supported(B: int, T: int, NH: int, H: int, dtype: torch.dtype) -> bool:
    """can attn_fwd / attn_bwd leave their outputs as fp8 too (fp8_out=...)?"""
    return bool(lib.dg_attn_fp8_out_supported(B, T, NH, H, dt_code(dtype)))


def new_attn_fp8_history(amax: Tensor) -> Tensor:
    """history of an attention fp8 call site (DG_ATTN_FP8_HIST floats: three slots of 64 partial maxima, one 128-byte line each --
    view(3, 64, 32)[s, i, 0]), every word seeded with `amax` (a device scalar)"""
    return amax.detach().float().reshape(1).expand(ATTN_FP8_HIST).contiguous()


def _attn_fp8_arg(fp8_out, shape, fmt: torch.dtype, dev, only8: bool = False):
    """fp8_out = (hist3, step_state) -> (struct, q8, scale_inv)"""
    from ._lib import AttnFp8Out
    hist3, step_state = fp8_out
    _chk(hist3, "hist3", torch.float32)
    if hist3.numel() != ATTN_FP8_HIST:
        raise RuntimeError("attention fp8 history must hold ATTN_FP8_HIST floats (ops.new_attn_fp8_history)")
    q8 = torch.empty(shape, dtype=fmt, device=dev)
    sinv = torch.empty((1,), dtype=torch.float32, device=dev)
    return AttnFp8Out(_p(q8), _p(hist3), _p(step_state), _p(sinv), 1 if only8 else 0), q8, sinv


def attn_fwd(qkv: Tensor, B: int, T: int, NH: int, H: int, scale: float, p: float, rng_state: Optional[Tensor], site: int,
             keep: bool = False, fp8_out=None):
    """fp8_out = (hist3, step_state) (precision fp8, attn_fp8_out_supported): the output also as e4m3 with delayed scaling -- the
    attribute `dg_fp8` = (e4m3 copy, scale_inv) travels with it.
    keep=True (training with dropout, a backward pass follows): the forward pass also leaves its dropout keep decisions as
    wave masks (dg_attn_keep_bits_bytes; 128 bytes per unmasked 32 x 32 tile) -- they travel with the output as its attribute
    `dg_keep` and attn_bwd(..., keep_bits=out.dg_keep) selects with them instead of hashing again.  Shapes on the generic
    kernels have no such path (no attribute is set)."""
    _chk(qkv, "qkv")
    if qkv.shape != (B * T, 3 * NH * H):
        raise RuntimeError(f"attn_fwd: qkv shape {tuple(qkv.shape)} != {(B * T, 3 * NH * H)}")
    out = torch.empty((B * T, NH * H), dtype=qkv.dtype, device=qkv.device)
    lse = torch.empty((B, NH, T), dtype=torch.float32, device=qkv.device)
    kb = None
    if keep and p > 0.0 and rng_state is not None:
        n = int(lib.dg_attn_keep_bits_bytes(B, T, NH, H, dt_code(qkv.dtype)))
        if n > 0:
            kb = torch.empty(n, dtype=torch.uint8, device=qkv.device)
    if fp8_out is not None:
        import ctypes
        arg, q8, sinv = _attn_fp8_arg(fp8_out, (B * T, NH * H), torch.float8_e4m3fn, qkv.device)
        check(lib.dg_attn_fwd_fp8(_p(qkv), _p(out), _p(lse), B, T, NH, H, float(scale), float(p), _p(rng_state) if p > 0.0 else None,
                                  site, dt_code(qkv.dtype), _p(kb), kb.numel() if kb is not None else 0, ctypes.byref(arg), _stream()),
              "dg_attn_fwd_fp8")
        out.dg_fp8 = (q8, sinv)
    else:
        check(lib.dg_attn_fwd(_p(qkv), _p(out), _p(lse), B, T, NH, H, float(scale), float(p), _p(rng_state) if p > 0.0 else None,
                              site, dt_code(qkv.dtype), _p(kb), kb.numel() if kb is not None else 0, _stream()), "dg_attn_fwd")
    if kb is not None:
        out.dg_keep = kb
    return out, lse


def attn_bwd(qkv: Tensor, out: Tensor, dout: Tensor, lse: Tensor, B: int, T: int, NH: int, H: int, scale: float, p: float,
             rng_state: Optional[Tensor], site: int, keep_bits: Optional[Tensor] = None, fp8_out=None, fp8_out_only: bool = False) -> Tensor:
    """keep_bits: the forward pass's keep masks (attn_fwd(..., keep=True) leaves them as out.dg_keep); default: taken from `out`.
    fp8_out = (hist3, step_state): dqkv also as e5m2 (attribute `dg_fp8` = (e5m2 copy, scale_inv)); fp8_out_only: the bf16 dqkv is
    not written (marked `dg_unwritten`)."""
    _chk(qkv, "qkv")
    if keep_bits is None:
        keep_bits = getattr(out, "dg_keep", None)
    if keep_bits is not None:
        _chk(keep_bits, "keep_bits", torch.uint8)
    _chk(out, "out", qkv.dtype)
    _chk(dout, "dout", qkv.dtype)
    _chk(lse, "lse", torch.float32)
    dqkv = torch.empty_like(qkv)
    ws = torch.empty(int(lib.dg_attn_bwd_workspace_bytes(B, T, NH, H, dt_code(qkv.dtype))), dtype=torch.uint8, device=qkv.device)
    if fp8_out is not None:
        import ctypes
        arg, q8, sinv = _attn_fp8_arg(fp8_out, tuple(qkv.shape), torch.float8_e5m2, qkv.device, only8=fp8_out_only)
        check(lib.dg_attn_bwd_fp8(_p(qkv), _p(out), _p(dout), _p(lse), _p(dqkv), _p(ws), ws.numel(), B, T, NH, H, float(scale), float(p),
                                  _p(rng_state) if p > 0.0 else None, site, dt_code(qkv.dtype), _p(keep_bits),
                                  keep_bits.numel() if keep_bits is not None else 0, ctypes.byref(arg), _stream()), "dg_attn_bwd_fp8")
        dqkv.dg_fp8 = (q8, sinv)
        if fp8_out_only:
            dqkv.dg_unwritten = True
        return dqkv
    check(lib.dg_attn_bwd(_p(qkv), _p(out), _p(dout), _p(lse), _p(dqkv), _p(ws), ws.numel(), B, T, NH, H, float(scale), float(p),
                          _p(rng_state) if p > 0.0 else None, site, dt_code(qkv.dtype), _p(keep_bits),
                          keep_bits.numel() if keep_bits is not None else 0, _stream()), "dg_attn_bwd")
    return dqkv


def attn_decode(cache: Tensor, t: int, NH: int, H: int, scale: float) -> Tensor:
    """cache [B, Tcap, 3*NH*H]; returns the attention output of position t, [B, NH*H]."""
    _chk(cache, "cache")
    B, Tcap, W = cache.shape
    if W != 3 * NH * H:
        raise RuntimeError("attn_decode: cache width != 3*NH*H")
    out = torch.empty((B, NH * H), dtype=cache.dtype, device=cache.device)
    check(lib.dg_attn_decode(_p(cache), _p(out), B, Tcap, t, NH, H, float(scale), dt_code(cache.dtype), _stream()), "dg_attn_decode")
    return out


def cross_entropy(logits: Tensor, targets: Tensor, V: int, dlogits: Optional[Tensor] = None, grad_scale: float = 1.0,
                  grad_scale_dev: Optional[Tensor] = None, loss_rows: Optional[Tensor] = None) -> Tensor:
    _chk(logits, "logits", contiguous=False)              # fp32, or bf16 (large vocabularies: dlogits may then BE logits)
    if logits.dtype not in (torch.float32, torch.bfloat16):
        raise TypeError("cross_entropy: logits must be float32 or bfloat16")
    _chk(targets, "targets", torch.int64)
    M = logits.shape[0]
    if loss_rows is None:
        loss_rows = torch.empty((M,), dtype=torch.float32, device=logits.device)
    ldd, dcode = 0, DG_F32
    if dlogits is not None:
        _chk(dlogits, "dlogits", contiguous=False)
        ldd, dcode = _ld(dlogits), dt_code(dlogits.dtype)
    check(lib.dg_cross_entropy(_p(logits), dt_code(logits.dtype), _ld(logits), _p(targets), _p(loss_rows), _p(dlogits), ldd, dcode, float(grad_scale),
                               _p(grad_scale_dev), M, V, _stream()), "dg_cross_entropy")
    return loss_rows


def cross_entropy_fp8(logits: Tensor, targets: Tensor, V: int, dlogits: Tensor, grad_scale: float, dlogits_fp8: Tensor) -> Tensor:
    """cross_entropy on bf16 logits (gradient in bf16, possibly in place) that also writes the gradient as e5m2 with the a-priori
    scale 57344 / grad_scale into dlogits_fp8 [M, ld8 >= V] (pad columns zeroed); dequantisation factor: grad_scale / 57344"""
    _chk(logits, "logits", torch.bfloat16, contiguous=False)
    _chk(targets, "targets", torch.int64)
    _chk(dlogits, "dlogits", torch.bfloat16, contiguous=False)
    _chk(dlogits_fp8, "dlogits_fp8", torch.float8_e5m2, contiguous=False)
    M = logits.shape[0]
    loss_rows = torch.empty((M,), dtype=torch.float32, device=logits.device)
    check(lib.dg_cross_entropy_fp8(_p(logits), _ld(logits), _p(targets), _p(loss_rows), _p(dlogits), _ld(dlogits), float(grad_scale), M, V,
                                   _p(dlogits_fp8), _ld(dlogits_fp8), _stream()), "dg_cross_entropy_fp8")
    return loss_rows


def cross_entropy_fused_supported(logits: Tensor, dlogits: Tensor, n_partials: int) -> bool:
    return logits.dtype == torch.float32 and _ld(dlogits) <= 128 and 0 < n_partials <= 2048


def cross_entropy_fused(logits: Tensor, targets: Tensor, V: int, dlogits: Tensor, grad_scale: float, colsum_part: Optional[Tensor],
                        part_stride: int, n_partials: int, loss_scratch: Optional[Tensor], loss_out: Optional[Tensor], loss_scale: float,
                        loss_rows: Optional[Tensor] = None) -> Tensor:
    """cross_entropy + column-sum partials of dlogits (n_partials rows of part_stride floats) + loss_out = loss_scale * sum(rows)
    in one launch (V <= 128).  loss_scratch: fp32 [n_partials + 1], its LAST word the arrival counter (zero before the first
    launch; the kernel leaves it at zero).  Returns the per-row losses."""
    _chk(logits, "logits", torch.float32, contiguous=False)
    _chk(targets, "targets", torch.int64)
    _chk(dlogits, "dlogits", contiguous=False)
    M = logits.shape[0]
    if loss_rows is None:
        loss_rows = torch.empty((M,), dtype=torch.float32, device=logits.device)
    if colsum_part is not None:
        _chk(colsum_part, "colsum_part", torch.float32, contiguous=False)
    cnt = None
    if loss_out is not None:
        _chk(loss_out, "loss_out", torch.float32)
        _chk(loss_scratch, "loss_scratch", torch.float32)
        if loss_scratch.numel() < n_partials + 1:
            raise ValueError("cross_entropy_fused: loss_scratch needs n_partials + 1 words")
        cnt = loss_scratch.data_ptr() + 4 * n_partials
    check(lib.dg_cross_entropy_fused(_p(logits), _ld(logits), _p(targets), _p(loss_rows), _p(dlogits), _ld(dlogits), dt_code(dlogits.dtype),
                                     float(grad_scale), M, V, _p(colsum_part), part_stride, n_partials,
                                     _p(loss_scratch) if loss_out is not None else None, cnt, _p(loss_out), float(loss_scale), _stream()),
          "dg_cross_entropy_fused")
    return loss_rows


def reduce_sum(x: Tensor, scale: float, out: Optional[Tensor] = None) -> Tensor:
    _chk(x, "x", torch.float32)
    if out is None:
        out = torch.empty((), dtype=torch.float32, device=x.device)
    check(lib.dg_reduce_sum(_p(x), x.numel(), float(scale), _p(out), _stream()), "dg_reduce_sum")
    return out


def softmax_rows(logits: Tensor) -> Tensor:
    _chk(logits, "logits", torch.float32, contiguous=False)
    M, V = logits.shape
    probs = torch.empty((M, V), dtype=torch.float32, device=logits.device)
    check(lib.dg_softmax_rows(_p(logits), _ld(logits), _p(probs), V, M, V, _stream()), "dg_softmax_rows")
    return probs


def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, hyper: Tensor, rng_state: Tensor, grad_scale: float = 1.0,
               shadow_bf16: Optional[Tensor] = None, n: Optional[int] = None, advance: bool = False) -> None:
    """advance: the launch also moves the step word of rng_state on (what state_advance does, without its launch)"""
    for t, nm in ((p, "p"), (g, "g"), (m, "m"), (v, "v"), (hyper, "hyper")):
        _chk(t, nm, torch.float32)
    n = p.numel() if n is None else n
    check(lib.dg_adamw_step(_p(p), _p(g), _p(m), _p(v), n, _p(hyper), _p(rng_state), float(grad_scale), _p(shadow_bf16), int(advance),
                            _stream()), "dg_adamw_step")


def block_chain_supported(M: int, C: int, dtype: torch.dtype) -> bool:
    """can dg_block_chain_fwd run this shape?  (bf16 operands, C = 384, M % 64 == 0)"""
    return dtype == torch.bfloat16 and bool(lib.dg_block_chain_supported(int(M), int(C)))


def l2_warm(t: Tensor) -> None:
    """touch every 128-byte line of the contiguous tensor t from every XCD (its storage must start on a 128-byte boundary)"""
    _chk(t, "t")
    check(lib.dg_l2_warm(_p(t), t.numel() * t.element_size(), _stream()), "dg_l2_warm")


def pack_chain_weights(W: Tensor, out: Optional[Tensor] = None) -> Tensor:
    """W [N, K] bf16 (N % 384 == 0, K % 32 == 0) -> the packed operand dg_block_chain_fwd streams (same shape / byte count)"""
    _chk(W, "W", torch.bfloat16, contiguous=False)
    N, K = W.shape
    if out is None:
        out = torch.empty((N, K), dtype=torch.bfloat16, device=W.device)
    _chk(out, "out", torch.bfloat16)
    check(lib.dg_pack_chain_weights(_p(W), _ld(W), _p(out), N, K, _stream()), "dg_pack_chain_weights")
    return out


def make_pack_table(pairs, device) -> tuple:
    """descriptor table for pack_chain_weights_batched: pairs = [(W [N, K] bf16, packed [N, K] bf16)]"""
    rows, first = [], 0
    for W, Pk in pairs:
        N, K = W.shape
        if N % 384 or K % 32:
            raise RuntimeError("pack_chain_weights: N % 384 == 0 and K % 32 == 0 required")
        rows.append([W.data_ptr(), Pk.data_ptr(), N, K, first, _ld(W)])
        first += (N // 384) * (K // 32)
    return torch.tensor(rows, dtype=torch.int64, device=device), len(rows), first


def pack_chain_weights_batched(table: Tensor, n_desc: int, total_stages: int) -> None:
    _chk(table, "table", torch.int64)
    check(lib.dg_pack_chain_weights_batched(_p(table), n_desc, total_stages, _stream()), "dg_pack_chain_weights_batched")


def block_chain_fwd(mode: int, M: int, Cd: int, *, o: Optional[Tensor] = None, x: Optional[Tensor] = None, wproj=None, bproj=None, ln2w=None, ln2b=None,
                    w1=None, b1=None, w2=None, b2=None, ln1w=None, ln1b=None, wqkv=None, f: Optional[Tensor] = None, x1: Optional[Tensor] = None,
                    dropout_p: float = 0.0, rng_state: Optional[Tensor] = None, site_proj: int = 0, site_ffn: int = 0, eps: float = 1e-5) -> dict:
    """row-local pieces of a residual block with the LayerNorm inside the producing GEMM's epilogue (dg_block_chain_fwd).
    mode 0: proj .. the next block's QKV in one launch; 1: last block (x2 comes back as bf16); 2: head (LayerNorm 1 + QKV of the
    first block on x); 3: proj + residual + LayerNorm 2 (o, x -> x1, h2, mean2, rstd2); 4: FFN2 + residual + the next block's
    LayerNorm 1 (f, x1 -> x2, h1, mean1, rstd1).  Weights are the PACKED bf16 operands (pack_chain_weights of the [out, in]
    matrices).  Returns the tensors the separate launches would have produced, by name."""
    from ._lib import BlockChainArgs
    dev = next(t for t in (x, o, f) if t is not None).device
    a = BlockChainArgs()
    a.mode, a.M, a.C, a.eps = mode, M, Cd, eps
    out = {}
    has_proj, has_ffn1, has_ffn2 = mode in (0, 1, 3), mode in (0, 1), mode in (0, 1, 4)
    has_qkv, ln1 = mode in (0, 2), mode in (0, 2, 4)

    def new(name, shape, dtype):
        t = out[name] = torch.empty(shape, dtype=dtype, device=dev)
        return t.data_ptr()

    def inp(t, name, dtype, n=None):
        _chk(t, name, dtype)
        if n is not None and t.numel() != n:
            raise RuntimeError(f"block_chain_fwd: {name} must hold {n} values, got {t.numel()}")
        return t.data_ptr()
    bf, f32 = torch.bfloat16, torch.float32
    if mode != 4:
        a.x = inp(x, "x", f32, M * Cd)
    if has_proj:
        a.o = inp(o, "o", bf, M * Cd)
        a.wproj, a.bproj = inp(wproj, "wproj", bf, Cd * Cd), inp(bproj, "bproj", f32, Cd)
        a.ln2w, a.ln2b = inp(ln2w, "ln2w", f32, Cd), inp(ln2b, "ln2b", f32, Cd)
        a.x1 = new("x1", (M, Cd), f32)
        a.mean2, a.rstd2 = new("mean2", (M,), f32), new("rstd2", (M,), f32)
        a.h2 = new("h2", (M, Cd), bf)
    if has_ffn1:
        a.w1, a.b1 = inp(w1, "w1", bf, 4 * Cd * Cd), inp(b1, "b1", f32, 4 * Cd)
        a.f = new("f", (M, 4 * Cd), bf)
        bits = out["bits"] = new_sign_bits(M, 4 * Cd, dev)
        a.sign_bits, a.sign_bits_bytes = bits.data_ptr(), bits.numel()
    if has_ffn2:
        if mode == 4:
            a.f, a.x1 = inp(f, "f", bf, M * 4 * Cd), inp(x1, "x1", f32, M * Cd)
        a.w2, a.b2 = inp(w2, "w2", bf, 4 * Cd * Cd), inp(b2, "b2", f32, Cd)
        if mode == 1:
            a.x2_bf16 = new("x2", (M, Cd), bf)
        else:
            a.x2 = new("x2", (M, Cd), f32)
    if ln1:
        a.ln1w, a.ln1b = inp(ln1w, "ln1w", f32, Cd), inp(ln1b, "ln1b", f32, Cd)
        a.mean1, a.rstd1 = new("mean1", (M,), f32), new("rstd1", (M,), f32)
        a.h1 = new("h1", (M, Cd), bf)
    if has_qkv:
        a.wqkv = inp(wqkv, "wqkv", bf, 3 * Cd * Cd)
        a.qkv = new("qkv", (M, 3 * Cd), bf)
    a.dropout_p = float(dropout_p)
    a.rng_state = _p(rng_state) if dropout_p > 0.0 else None
    a.site_proj, a.site_ffn = site_proj, site_ffn
    check(lib.dg_block_chain_fwd(C.byref(a), _stream()), "dg_block_chain_fwd")
    return out


def block_chain_bwd_supported(M: int, C: int, dtype: torch.dtype) -> bool:
    """can dg_block_chain_bwd run this shape?  (bf16 operands and gradient stream, C = 384, M % 64 == 0, M / 64 <= #CUs)"""
    return dtype == torch.bfloat16 and bool(lib.dg_block_chain_bwd_supported(int(M), int(C)))


def block_chain_bwd(mode: int, M: int, Cd: int, *, part_stride: int, dqkv=None, wqkvT=None, x=None, mean1=None, rstd1=None, ln1w=None, dresid1=None,
                    dln1w_part=None, dln1b_part=None, gbias1_part=None, g_in=None, w2T=None, bits=None, db1_part=None, w1T=None, x1=None,
                    mean2=None, rstd2=None, ln2w=None, dresid2=None, dln2w_part=None, dln2b_part=None, gbias2_part=None, wprojT=None,
                    dropout_p: float = 0.0, rng_state: Optional[Tensor] = None, site_ffn_below: int = 0, site_proj: int = 0) -> dict:
    """the backward pass's row-local chain between two attention-backward calls in one launch (dg_block_chain_bwd).
    mode 0: dX-QKV + LayerNorm-1 backward of block l, then dX-FFN2 / dX-FFN1 / LayerNorm-2 backward / dX-proj of block l - 1;
    1: the second half only (g_in = the operand of dX-FFN2); 2: the first half only.  Weights are the PACKED W^T operands
    (pack_chain_weights of the [in, out] shadows).  The *_part arguments are views of row 0 of a partial buffer with
    `part_stride` floats per row and at least 2 * M / 64 rows.  Returns the tensors the separate launches would have produced."""
    from ._lib import BlockChainBwdArgs
    a = BlockChainBwdArgs()
    a.mode, a.M, a.C = mode, M, Cd
    bf, f32 = torch.bfloat16, torch.float32
    has_q, has_2 = mode in (0, 2), mode in (0, 1)
    dev = (dqkv if has_q else g_in).device
    out = {}

    def new(name, shape, dtype=bf):
        t = out[name] = torch.empty(shape, dtype=dtype, device=dev)
        return t.data_ptr()

    def inp(t, name, dtype, n=None):
        _chk(t, name, dtype)
        if n is not None and t.numel() != n:
            raise RuntimeError(f"block_chain_bwd: {name} must hold {n} values, got {t.numel()}")
        return t.data_ptr()

    def part(t, name, n):
        _chk(t, name, f32, contiguous=False)
        if t.numel() != n:
            raise RuntimeError(f"block_chain_bwd: {name} must be a row of {n} partial sums, got {t.numel()}")
        return t.data_ptr()
    if has_q:
        a.dqkv, a.wqkvT = inp(dqkv, "dqkv", bf, M * 3 * Cd), inp(wqkvT, "wqkvT", bf, 3 * Cd * Cd)
        a.x, a.mean1, a.rstd1, a.ln1w = inp(x, "x", f32, M * Cd), inp(mean1, "mean1", f32, M), inp(rstd1, "rstd1", f32, M), inp(ln1w, "ln1w", f32, Cd)
        a.dresid1 = inp(dresid1, "dresid1", bf, M * Cd)
        a.dx1, a.g1 = new("dx1", (M, Cd)), new("g1", (M, Cd))
        a.dln1w_part, a.dln1b_part = part(dln1w_part, "dln1w_part", Cd), part(dln1b_part, "dln1b_part", Cd)
        a.gbias1_part = part(gbias1_part, "gbias1_part", Cd) if gbias1_part is not None else None
    if has_2:
        if mode == 1:
            a.g_in = inp(g_in, "g_in", bf, M * Cd)
        a.w2T, a.w1T, a.wprojT = inp(w2T, "w2T", bf, 4 * Cd * Cd), inp(w1T, "w1T", bf, 4 * Cd * Cd), inp(wprojT, "wprojT", bf, Cd * Cd)
        _chk(bits, "bits", torch.uint8)
        a.sign_bits, a.sign_bits_bytes = bits.data_ptr(), bits.numel()
        a.df = new("df", (M, 4 * Cd))
        a.db1_part = part(db1_part, "db1_part", 4 * Cd)
        a.x1, a.mean2, a.rstd2, a.ln2w = inp(x1, "x1", f32, M * Cd), inp(mean2, "mean2", f32, M), inp(rstd2, "rstd2", f32, M), inp(ln2w, "ln2w", f32, Cd)
        a.dresid2 = out["dx1"].data_ptr() if mode == 0 else inp(dresid2, "dresid2", bf, M * Cd)
        a.dx2, a.g2, a.dout = new("dx2", (M, Cd)), new("g2", (M, Cd)), new("dout", (M, Cd))
        a.dln2w_part, a.dln2b_part = part(dln2w_part, "dln2w_part", Cd), part(dln2b_part, "dln2b_part", Cd)
        a.gbias2_part = part(gbias2_part, "gbias2_part", Cd)
    a.part_stride = int(part_stride)
    a.dropout_p = float(dropout_p)
    a.rng_state = _p(rng_state) if dropout_p > 0.0 else None
    a.site_ffn_below, a.site_proj = site_ffn_below, site_proj
    check(lib.dg_block_chain_bwd(C.byref(a), _stream()), "dg_block_chain_bwd")
    return out
