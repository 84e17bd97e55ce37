"""Kernel sequences of the hot path: forward and hand-written backward of each sub-layer.

Each *_fwd returns (output, saved) and each *_bwd consumes `saved`; none of them uses autograd.
They are shared by the two callers:
  - drakegpt_amd/functional.py wraps them in torch.autograd.Function (the drop-in nn.Module path);
  - drakegpt_amd/engine.py runs them back to back inside one captured hipGraph (the training path).

Weight gradients are emitted as partial sums through a `sink` (see GradSink below) and weight
operands (bf16 copies / transposes) come from a `weights` provider, so that the engine can keep
flat, persistent buffers while the autograd path allocates on the fly.

Reference call sites restated here (paths relative to the reference repository root):
  attention sub-layer  x + dropout(proj(cat_h softmax(mask(q k^T * s)) v))   src/model_component.py:378-407,436-455,505
  feed-forward         x + dropout(W2 relu(W1 ln(x) + b1) + b2)              src/model_component.py:320-325,506
  embedding            tok[idx] + pos[arange(T)]                             src/model.py:595-597
  lm head + CE         cross_entropy(x W^T + b, targets)                     src/model.py:599,604-607
"""
from __future__ import annotations

import os

from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import ops

Tensor = torch.Tensor
E4M3, E5M2 = torch.float8_e4m3fn, torch.float8_e5m2       # OCP encodings: forward operands / gradients


def site_attn(layer: int) -> int:
    return 4 * layer + 0


def site_proj(layer: int) -> int:
    return 4 * layer + 1


def site_ffn(layer: int) -> int:
    return 4 * layer + 2


def granule(dtype: torch.dtype) -> int:
    """elements per 16 bytes: the K / leading-dimension granule of the MFMA GEMMs"""
    return 8 if dtype == torch.bfloat16 else 4


def k_pad(n: int, dtype: torch.dtype) -> int:
    """padded contraction length of a dX GEMM whose K is a Linear's out_features (lm_head: K = V): a multiple of 64 in bf16
    so that the LDS-DMA kernel takes it (V = 80 -> 128, 50257 -> 50304), of the 16-byte granule in fp32"""
    return pad_to(n, 64) if dtype == torch.bfloat16 else pad_to(n, granule(dtype))


def pad_to(n: int, g: int) -> int:
    return (n + g - 1) // g * g


def n_splits_for(rows: int) -> int:
    """how many ways the dW GEMMs split their contraction over the B*T rows"""
    return max(1, min(8, rows // 2048))


_NCU = {}


def splits_for_matrix(P: int, Q: int, rows: int, s_max: int, device=None) -> int:
    """Per-matrix split count of a dW GEMM: as many splits as still give ONE round of workgroups
    (tiles * splits <= #CUs; the LDS-DMA kernel owns a CU), never more than the slab count s_max.
    FFN weights at the scaled config: 36 tiles -> 7 splits (30 us) instead of 8 (two rounds, 46 us)."""
    key = str(device)
    if key not in _NCU:
        _NCU[key] = torch.cuda.get_device_properties(device).multi_processor_count if torch.cuda.is_available() else 256
    tiles = ((P + 127) // 128) * ((Q + 127) // 128)
    return max(1, min(s_max, _NCU[key] // tiles, max(1, rows // 256)))


FUSED_LN_FP8 = os.environ.get("DG_FP8_FUSED_LN", "1") != "0"      # fp8 mode: the fused LayerNorm backward also emits the e5m2 operand (A/B switch)


def n_partials_for(rows: int) -> int:
    """row-chunk partials of the column-sum style reductions (bias / LayerNorm gradients)"""
    return max(1, min(256, rows // 32))


# ------------------------------------------------------------------------------------------------
class OnTheFlyWeights:
    """Weight operands computed per call (autograd path): bf16 copy for forward, W^T for dX."""

    def __init__(self, act: torch.dtype):
        self.act = act

    def fwd(self, W: Tensor) -> Tensor:
        if self.act == torch.float32:
            return W if W.is_contiguous() else W.contiguous()
        return ops.cast(W.contiguous(), self.act)

    def bwd(self, W: Tensor) -> Tensor:
        return ops.transpose_cast(W, self.act, ldo=k_pad(W.shape[0], self.act))

    def fwd8(self, W: Tensor):
        """(e4m3 copy of W, dequantisation scale) -- quantised per call here; the engine keeps persistent fp8 shadows"""
        return ops.fp8_quantize(self.fwd(W), E4M3)

    def bwd8(self, W: Tensor):
        return ops.fp8_quantize(self.bwd(W), E4M3)


class LocalSink:
    """Gradient sink of the autograd path: allocates partial buffers, reduces on `finish`."""

    def __init__(self, rows: int, device):
        self.rows = rows
        self.S = n_splits_for(rows)
        self.G = n_partials_for(rows)
        self.device = device
        self._pending: Dict[str, Tuple[Tensor, int, int, Tuple[int, ...]]] = {}
        self._direct: Dict[str, Tensor] = {}

    def matrix(self, key: str, P: int, Q: int):
        n = splits_for_matrix(P, Q, self.rows, self.S, self.device)
        part = torch.empty((n, P, Q), dtype=torch.float32, device=self.device)
        self._pending[key] = (part, P * Q, n, (P, Q))
        return part, P * Q, n

    def vector(self, key: str, N: int):
        part = torch.empty((self.G, N), dtype=torch.float32, device=self.device)
        self._pending[key] = (part, N, self.G, (N,))
        return part, N, self.G

    def vector_rows(self, key: str, N: int, rows: int) -> Tensor:
        """[rows, N] partial rows for a producer with its own partial count (GEMM epilogue column sums)"""
        part = torch.empty((rows, N), dtype=torch.float32, device=self.device)
        self._pending[key] = (part, N, rows, (N,))
        return part

    def direct(self, key: str, shape) -> Tensor:
        t = torch.zeros(shape, dtype=torch.float32, device=self.device)
        self._direct[key] = t
        return t

    def finish(self) -> Dict[str, Tensor]:
        out = dict(self._direct)
        for key, (part, stride, n, shape) in self._pending.items():
            g = torch.empty(shape, dtype=torch.float32, device=self.device)
            ops.reduce_partials(part, stride, n, g, g.numel())
            out[key] = g
        return out


def weight_grad(sink, key: str, dy: Tensor, x: Tensor, P: int, Q: int) -> None:
    """dW[P,Q] = dy[:, :P]^T x[:, :Q].  A sink with `defer` (engine, bf16) only records the operands and computes
    every dW of the step in one grouped launch at the end of backward; otherwise split-K partials now."""
    defer = getattr(sink, "defer", None)
    if defer is not None and defer(key, dy, x, P, Q):
        return
    part, stride, n = sink.matrix(key, P, Q)
    ops.gemm_tn(dy, x, part, stride, n, P, Q)


@dataclass
class Run:
    """per-call runtime configuration"""
    act: torch.dtype                 # GEMM operand / stored activation type
    rng: Optional[Tensor]            # device rng state snapshot; None => no dropout (eval)
    weights: object                  # OnTheFlyWeights | engine.ShadowWeights
    fp8: bool = False                # precision = "fp8": the block Linears (forward and dX) run on fp8 operands
    fp8_sites: Optional[dict] = None # engine only: per call site [2 x FP8_AMAX_PARTS] amax history => one-pass delayed scaling
    fp8_seed: bool = False           # engine warm-up: quantise just in time and seed the sites' history with this batch's amax
    step_word: Optional[Tensor] = None   # device {seed, step} words: the step parity selects the history slot
    stream: torch.dtype = torch.float32  # type of the residual-branch gradient stream (engine, bf16 / fp8 modes: bf16)
    fp8_only: bool = False               # engine, precision fp8 with the fp8 dW: the FFN hidden layer and its gradient are read as fp8
                                         # only (by the next GEMM and by the grouped dW launch), so their bf16 form is not written

    def p(self, p: float) -> float:
        return p if (self.rng is not None and p > 0.0) else 0.0


def fp8_k_ok(k: int) -> bool:
    """contraction lengths the fp8 GEMM takes (128-element K steps, at least two)"""
    return k % 128 == 0 and k >= 256


def _quantize_operand(run: Run, x: Tensor, fmt: torch.dtype, site: Optional[str]):
    """an activation / gradient operand as fp8.  Module path and evaluation: just in time (amax pass + cast pass).  Training
    engine: the call site `site` keeps the amax of the previous step and the tensor is cast in one pass (delayed scaling);
    the engine's eager warm-up step seeds that history with a just-in-time amax."""
    if run.fp8_sites is None or site is None:
        return ops.fp8_quantize(x, fmt)
    parts2 = run.fp8_sites.get(site)
    if run.fp8_seed or parts2 is None:
        if parts2 is None:
            parts2 = run.fp8_sites[site] = torch.zeros(2 * ops.FP8_AMAX_PARTS, dtype=torch.float32, device=x.device)
        xq, xs = ops.fp8_quantize(x, fmt, amax=parts2[:ops.FP8_AMAX_PARTS])
        parts2[ops.FP8_AMAX_PARTS:].copy_(parts2[:ops.FP8_AMAX_PARTS])
        return xq, xs
    return ops.fp8_quantize_delayed(x, fmt, parts2, run.step_word)


def _refuse_unwritten(t: Tensor) -> None:
    """a tensor whose bf16 form was skipped (gemm_nt fp8_out_only) must never be read as bf16"""
    if getattr(t, "dg_unwritten", False):
        raise RuntimeError("drakegpt_amd: this operand exists as fp8 only (fp8_out_only) and a bf16 consumer asked for it")


def linear_nt(run: Run, x: Tensor, W: Tensor, out_dtype: torch.dtype, fp8_site: Optional[str] = None, x8=None, **epi) -> Tensor:
    """forward of a block Linear: epilogue(x W^T), x [M, in] in the activation dtype, W [out, in] the fp32 master.
    fp8 mode: x is quantised to e4m3 (see _quantize_operand) -- or arrives quantised already as x8 = (e4m3 copy, scale) from
    the epilogue that produced it -- and W comes as its persistent e4m3 shadow."""
    if run.fp8 and fp8_k_ok(W.shape[1]) and x.is_contiguous() and x.dtype == torch.bfloat16:
        wq, ws = run.weights.fwd8(W)
        xq, xs = x8 if x8 is not None else _quantize_operand(run, x, E4M3, fp8_site)
        x.dg_fp8x = (xq, xs)        # the e4m3 copy travels with the activation: it is also the X operand of this Linear's fp8 dW
        return ops.gemm_nt(xq, wq, out_dtype, scale_a=xs, scale_b=ws, **epi)
    _refuse_unwritten(x)
    return ops.gemm_nt(x, run.weights.fwd(W), out_dtype, **epi)


def _fused_fp8_out(run: Run, site: str, M: int, N: int, K: int, dev, grad: bool = False):
    """fp8_out argument for the GEMM that PRODUCES the operand of call site `site` (training engine, history seeded): the e4m3
    copy leaves that GEMM's epilogue and the site's cast launch -- which would re-read the whole tensor -- disappears"""
    if not run.fp8 or run.fp8_sites is None or run.fp8_seed or site not in run.fp8_sites or run.step_word is None:
        return None
    if not (fp8_k_ok(K) and fp8_k_ok(N) and ops.gemm_nt_fp8_out_supported(M, N, K, grad=grad)):
        return None
    return (torch.empty((M, N), dtype=E5M2 if grad else E4M3, device=dev), run.fp8_sites[site], run.step_word,
            torch.empty((1,), dtype=torch.float32, device=dev))


def linear_dx(run: Run, g: Tensor, W: Tensor, out_dtype: torch.dtype, fp8_site: Optional[str] = None, g8=None, **epi) -> Tensor:
    """dX of a block Linear: epilogue(g W), g [M, out] in the activation dtype (a gradient: e5m2 in fp8 mode -- quantised here,
    or already by the epilogue that produced it: g8 = (e5m2 copy, scale)), W^T shadow [in, out padded]"""
    K = W.shape[0]
    if run.fp8 and fp8_k_ok(K) and g.is_contiguous() and g.dtype == torch.bfloat16 and g.shape[1] == K:
        wq, ws = run.weights.bwd8(W)
        if wq.shape[1] == K:
            if g8 is None:
                g8 = getattr(g, "dg_fp8", None)         # left by the fused LayerNorm backward that produced g (see _ln_tail)
            gq, gs = g8 if g8 is not None else _quantize_operand(run, g, E5M2, fp8_site)
            g.dg_fp8 = (gq, gs)     # (also the dY operand of this Linear's fp8 dW, looked up when the grouped launch is assembled)
            return ops.gemm_nt(gq, wq, out_dtype, K=K, scale_a=gs, scale_b=ws, **epi)
    _refuse_unwritten(g)
    return ops.gemm_nt(g, run.weights.bwd(W), out_dtype, K=K, **epi)


def _op_dtype(run: Run, k: int, grad: bool = False):
    """operand dtype linear_nt / linear_dx will use for a contraction of length k (the sign-bit / column-sum support queries)"""
    if run.fp8 and fp8_k_ok(k):
        return E5M2 if grad else E4M3
    return None


LN_FWD_FP8 = os.environ.get("DG_FP8_FUSED_LNF", "1") != "0"     # fp8 training: LayerNorm outputs leave their launch as e4m3 (A/B: 0 = cast launches)


def _ln_fwd(run: Run, x2d: Tensor, ln_w: Tensor, ln_b: Tensor, site: str):
    """LayerNorm forward -> (h, mean, rstd, x8).  Training engine in precision fp8 (history of call site `site` seeded): the output
    leaves the LayerNorm launch as e4m3 with delayed scaling (x8 = (e4m3 copy, scale): what the cast launch in front of the next
    GEMM would have produced one launch later), and where nobody reads the bf16 form (run.fp8_only) that form is not written."""
    C = ln_w.numel()
    ok = (run.fp8 and run.fp8_sites is not None and run.step_word is not None and run.act == torch.bfloat16 and LN_FWD_FP8
          and C % 4 == 0 and C <= 1024 and fp8_k_ok(C) and x2d.dtype == torch.float32)
    key = site + "#ln"
    if ok and not run.fp8_seed and key in run.fp8_sites:
        h, mean, rstd, q8, sinv = ops.layernorm_fwd_fp8(x2d, ln_w, ln_b, run.fp8_sites[key], run.step_word, want_bf16=not run.fp8_only)
        return h, mean, rstd, (q8, sinv)
    h, mean, rstd = ops.layernorm_fwd(x2d, ln_w, ln_b, run.act)
    if ok and run.fp8_seed:
        # (eager warm-up step) both history slots start at this batch's maximum
        n = ops.layernorm_fwd_fp8_parts(x2d.shape[0])
        run.fp8_sites[key] = h.detach().abs().max().float().reshape(1).expand(2 * n).contiguous()
    return h, mean, rstd, None


ATTN_FP8_OUT = os.environ.get("DG_FP8_FUSED_ATTN", "1") != "0"     # fp8 training: o / dqkv leave the attention kernels as e4m3 / e5m2 too (A/B: 0 = cast launches)


def _attn_fp8_out(run: Run, key: str, t: Optional[Tensor], B: int, T: int, NH: int, H: int, k: int):
    """fp8_out argument of ops.attn_fwd / ops.attn_bwd for call site `key` (training engine, precision fp8, the consumer's
    contraction length k takes the fp8 GEMM): (history, step words) once the history is seeded, else None.  During the engine's
    eager warm-up step (`t` = the tensor the unfused path just produced) the history is seeded with that tensor's maximum."""
    if not (ATTN_FP8_OUT and run.fp8 and run.fp8_sites is not None and run.step_word is not None and run.act == torch.bfloat16
            and fp8_k_ok(k) and ops.attn_fp8_out_supported(B, T, NH, H, run.act)):
        return None
    if run.fp8_seed:
        if t is not None:
            run.fp8_sites[key] = ops.new_attn_fp8_history(t.detach().abs().max())
        return None
    hist = run.fp8_sites.get(key)
    return None if hist is None else (hist, run.step_word)


def _as_act(run: Run, x2d: Tensor) -> Tensor:
    return x2d if x2d.dtype == run.act else ops.cast(x2d, run.act)


def _dh_dtype(run: Run, ln_w: Tensor) -> torch.dtype:
    """dtype of the gradient a dX GEMM hands to the LayerNorm backward: the activation type when the vector kernel can take it
    (half the store of the GEMM and half the read of the LayerNorm backward; it is rounded once, like every other activation
    gradient of the bf16 mode)"""
    return run.act if (run.act == torch.bfloat16 and ln_w.numel() % 4 == 0 and ln_w.numel() <= 2048) else torch.float32


def _ln_tail(run: Run, dh: Tensor, x2d: Tensor, ln_w: Tensor, mean, rstd, dresid, sink, keys, emit):
    """LayerNorm backward (+ residual-branch gradient).  With `emit = (p, site, bias_key, N[, fp8_site])` the kernel also
    produces g = dropout_bwd(dx) for the sub-layer that runs next in backward, and that sub-layer's bias partials; in fp8
    training (history of call site `fp8_site` seeded) g leaves a second time as the e5m2 operand of that sub-layer's dX GEMM
    and travels with g as its attribute `dg_fp8` = (e5m2 copy, scale) -- linear_dx picks it up instead of casting."""
    pg, sg, ng = sink.vector(keys["ln_w"], ln_w.numel())
    pb, _, _ = sink.vector(keys["ln_b"], ln_w.numel())
    if emit is not None and ops.layernorm_bwd_fused_supported(ln_w.numel()):
        p, site, bias_key, N = emit[:4]
        f8site = emit[4] if len(emit) > 4 else None
        pq = sink.vector(bias_key, N)[0] if bias_key is not None else None       # None: g only (no sub-layer bias behind it)
        if (f8site is not None and run.fp8 and run.fp8_sites is not None and not run.fp8_seed and f8site in run.fp8_sites
                and run.step_word is not None and run.act == torch.bfloat16 and ng == ops.FP8_AMAX_PARTS and fp8_k_ok(N) and FUSED_LN_FP8):
            dx, g, g8, gs = ops.layernorm_bwd_fused(dh, x2d, ln_w, mean, rstd, dresid, pg, pb, sg, ng, run.act, run.p(p), run.rng, site, pq,
                                                    stream_dtype=run.stream, fp8_out=(run.fp8_sites[f8site], run.step_word),
                                                    fp8_out_only=run.fp8_only)
            g.dg_fp8 = (g8, gs)
            return dx, g
        return ops.layernorm_bwd_fused(dh, x2d, ln_w, mean, rstd, dresid, pg, pb, sg, ng, run.act, run.p(p), run.rng, site, pq,
                                       stream_dtype=run.stream)
    if run.stream != torch.float32:
        raise RuntimeError("the bf16 gradient stream needs the fused LayerNorm backward (C % 4 == 0, C <= 1024)")
    return ops.layernorm_bwd(dh, x2d, ln_w, mean, rstd, dresid, pg, pb, sg, ng), None


# ------------------------------------------------------------------------------------------------
# attention sub-layer
def _chain_ok(run: Run, x2d: Tensor, W: Optional[Tensor]) -> bool:
    """can the producing GEMM also run the LayerNorm that follows it (dg_block_chain_fwd modes 3 / 4)?"""
    pack = getattr(run.weights, "pack", None)
    return (W is not None and pack is not None and not run.fp8 and x2d.dtype == torch.float32
            and ops.block_chain_supported(x2d.shape[0], x2d.shape[1], run.act) and pack(W) is not None)


def attn_fwd(run: Run, x2d: Tensor, ln_w: Optional[Tensor], ln_b: Optional[Tensor], wqkv: Tensor,
             wproj: Optional[Tensor], bproj: Optional[Tensor], residual: bool,
             B: int, T: int, NH: int, H: int, p_attn: float, p_proj: float, layer: int,
             pre=None, fuse_ln=None, nxt: Optional[list] = None):
    """x2d [B*T, C].  Returns y [B*T, C_out] fp32 and the tensors backward needs.
    pre = (h, mean, rstd): the LayerNorm of x2d, already computed by the GEMM that produced x2d.  fuse_ln = (gamma, beta) of the
    LayerNorm that FOLLOWS this sub-layer: when the row-complete form is available the projection's epilogue also runs it and
    (h_next, mean, rstd) is appended to `nxt` (else nothing is appended and the caller runs the LayerNorm itself)."""
    x8 = None
    if pre is not None:
        h, mean, rstd = pre
    elif ln_w is not None:
        h, mean, rstd, x8 = _ln_fwd(run, x2d, ln_w, ln_b, f"{layer}.h1")
    else:
        h, mean, rstd = _as_act(run, x2d), None, None
    qkv = linear_nt(run, h, wqkv, run.act, fp8_site=f"{layer}.h1", x8=x8)
    okey = f"{layer}.o#attn"
    f8 = _attn_fp8_out(run, okey, None, B, T, NH, H, NH * H) if wproj is not None else None
    o, lse = ops.attn_fwd(qkv, B, T, NH, H, H ** -0.5, run.p(p_attn), run.rng, site_attn(layer), keep=True,     # (keep masks: only with dropout on)
                          fp8_out=f8)
    if wproj is not None and run.fp8_seed:
        _attn_fp8_out(run, okey, o, B, T, NH, H, NH * H)
    if wproj is not None:
        if fuse_ln is not None and nxt is not None and residual and o.shape[1] == x2d.shape[1] and _chain_ok(run, x2d, wproj):
            r = ops.block_chain_fwd(3, x2d.shape[0], x2d.shape[1], o=o, x=x2d, wproj=run.weights.pack(wproj), bproj=bproj, ln2w=fuse_ln[0],
                                    ln2b=fuse_ln[1], dropout_p=run.p(p_proj), rng_state=run.rng, site_proj=site_proj(layer))
            y = r["x1"]
            nxt.append((r["h2"], r["mean2"], r["rstd2"]))
        else:
            y = linear_nt(run, o, wproj, torch.float32, fp8_site=f"{layer}.o", x8=getattr(o, "dg_fp8", None), bias=bproj, dropout_p=run.p(p_proj),
                          rng_state=run.rng, site=site_proj(layer), residual=x2d if residual else None)
    else:
        if residual:
            raise RuntimeError("residual attention without a projection is not a reference configuration")
        y = o if o.dtype == torch.float32 else ops.cast(o, torch.float32)
    return y, (x2d, h, mean, rstd, qkv, o, lse)


def attn_bwd(run: Run, saved, dy: Tensor, ln_w: Optional[Tensor], wqkv: Tensor, wproj: Optional[Tensor],
             residual: bool, B: int, T: int, NH: int, H: int, p_attn: float, p_proj: float, layer: int,
             sink, keys: Dict[str, str], need_dx: bool = True, g_in: Optional[Tensor] = None, emit=None):
    """Returns dx, or (dx, g_next) when `emit` is given (engine path; see _ln_tail).  `g_in`: dropout-backward of dy
    already produced (with the proj bias partials) by the previous LayerNorm backward."""
    x2d, h, mean, rstd, qkv, o, lse = saved
    M = x2d.shape[0]
    if wproj is not None:
        if g_in is not None:
            g = g_in
        else:
            part, stride, n = sink.vector(keys["bproj"], wproj.shape[0])
            g = ops.dropout_bwd_cast(dy, run.act, run.p(p_proj), run.rng, site_proj(layer), colsum_part=part,
                                     part_stride=stride, n_partials=n)
        weight_grad(sink, keys["wproj"], g, o, wproj.shape[0], wproj.shape[1])
        do = linear_dx(run, g, wproj, run.act, fp8_site=f"{layer}.g_proj")
    else:
        do = _as_act(run, dy)
    gkey = f"{layer}.dqkv#attn"
    f8 = _attn_fp8_out(run, gkey, None, B, T, NH, H, wqkv.shape[0]) if need_dx else None
    if f8 is not None and run.weights.bwd8(wqkv)[0].shape[1] != wqkv.shape[0]:
        f8 = None                                  # (linear_dx would not take the fp8 GEMM)
    dqkv = ops.attn_bwd(qkv, o, do, lse, B, T, NH, H, H ** -0.5, run.p(p_attn), run.rng, site_attn(layer), fp8_out=f8,
                        fp8_out_only=f8 is not None and run.fp8_only)
    if need_dx and run.fp8_seed:
        _attn_fp8_out(run, gkey, dqkv, B, T, NH, H, wqkv.shape[0])
    weight_grad(sink, keys["wqkv"], dqkv, h, wqkv.shape[0], wqkv.shape[1])
    if not need_dx:
        return None
    if ln_w is not None:
        dh = linear_dx(run, dqkv, wqkv, _dh_dtype(run, ln_w), fp8_site=f"{layer}.dqkv")
        dx, g_next = _ln_tail(run, dh, x2d, ln_w, mean, rstd, dy if residual else None, sink, keys, emit)
        return (dx, g_next) if emit is not None else dx
    dx = linear_dx(run, dqkv, wqkv, torch.float32, fp8_site=f"{layer}.dqkv", residual=dy if residual else None)
    return (dx, None) if emit is not None else dx


# ------------------------------------------------------------------------------------------------
# feed-forward sub-layer
def ffn_fwd(run: Run, x2d: Tensor, ln_w: Optional[Tensor], ln_b: Optional[Tensor], w1: Tensor, b1: Tensor,
            w2: Optional[Tensor], b2: Optional[Tensor], residual: bool, p: float, layer: int,
            out_dtype: torch.dtype = torch.float32, pre=None, fuse_ln=None, nxt: Optional[list] = None):
    """out_dtype: the engine asks for the activation type from the LAST block -- its output only feeds lm_head, which would
    cast it anyway (same rounding), so the fp32 copy and the cast launch disappear.  pre / fuse_ln / nxt: see attn_fwd (here
    the LayerNorm that follows is the NEXT block's first one, run by the second Linear's epilogue)."""
    x8 = None
    if pre is not None:
        h, mean, rstd = pre
    elif ln_w is not None:
        h, mean, rstd, x8 = _ln_fwd(run, x2d, ln_w, ln_b, f"{layer}.h2")
    else:
        h, mean, rstd = _as_act(run, x2d), None, None
    if w2 is None:
        # FeedForward: Linear(C,C) + ReLU (ref: src/model_component.py:118-121)
        y = linear_nt(run, h, w1, torch.float32, x8=x8, bias=b1, relu=True)
        return y, (x2d, h, mean, rstd, y, None)
    # the ReLU mask for backward travels as one bit per element next to f (1/16 of the bytes the dX GEMM would re-read)
    bits = None
    if (ops.gemm_nt_sign_bits_supported(run.act, w1.shape[0], w1.shape[1], in_dtype=_op_dtype(run, w1.shape[1]))
            and ops.gemm_nt_sign_bits_supported(run.act, w1.shape[0], w2.shape[0], in_dtype=_op_dtype(run, w2.shape[0], grad=True))):
        bits = ops.new_sign_bits(h.shape[0], w1.shape[0], h.device)
    f8 = _fused_fp8_out(run, f"{layer}.f", h.shape[0], w1.shape[0], w1.shape[1], h.device) if bits is not None else None
    f = linear_nt(run, h, w1, run.act, fp8_site=f"{layer}.h2", x8=x8, bias=b1, relu=True, sign_bits_out=bits, fp8_out=f8,
                  fp8_out_only=bool(f8 is not None and run.fp8_only and w2 is not None))
    if (fuse_ln is not None and nxt is not None and residual and out_dtype == torch.float32 and w1.shape[0] == 4 * x2d.shape[1]
            and _chain_ok(run, x2d, w2)):
        r = ops.block_chain_fwd(4, x2d.shape[0], x2d.shape[1], f=f, x1=x2d, w2=run.weights.pack(w2), b2=b2, ln1w=fuse_ln[0], ln1b=fuse_ln[1],
                                dropout_p=run.p(p), rng_state=run.rng, site_ffn=site_ffn(layer))
        y = r["x2"]
        nxt.append((r["h1"], r["mean1"], r["rstd1"]))
    else:
        y = linear_nt(run, f, w2, out_dtype, fp8_site=f"{layer}.f", x8=(f8[0], f8[3]) if f8 is not None else None, bias=b2,
                      dropout_p=run.p(p), rng_state=run.rng, site=site_ffn(layer), residual=x2d if residual else None)
    return y, (x2d, h, mean, rstd, f, bits)


def ffn_bwd(run: Run, saved, dy: Tensor, ln_w: Optional[Tensor], w1: Tensor, w2: Optional[Tensor], residual: bool,
            p: float, layer: int, sink, keys: Dict[str, str], need_dx: bool = True, g_in: Optional[Tensor] = None,
            emit=None):
    x2d, h, mean, rstd, f, bits = saved
    if w2 is None:
        part, stride, n = sink.vector(keys["b1"], w1.shape[0])
        df = ops.dropout_bwd_cast(dy, run.act, 0.0, None, 0, relu_mask=f, colsum_part=part, part_stride=stride, n_partials=n)
    else:
        if g_in is not None:
            g = g_in
        else:
            part, stride, n = sink.vector(keys["b2"], w2.shape[0])
            g = ops.dropout_bwd_cast(dy, run.act, run.p(p), run.rng, site_ffn(layer), colsum_part=part, part_stride=stride,
                                     n_partials=n)
        weight_grad(sink, keys["w2"], g, f, w2.shape[0], w2.shape[1])
        cs_part = None
        vector_rows = getattr(sink, "vector_rows", None)
        if bits is not None and vector_rows is not None:
            rows = ops.gemm_nt_colsum_rows(run.act, g.shape[0], w1.shape[0], w2.shape[0], in_dtype=_op_dtype(run, w2.shape[0], grad=True))
            if rows:
                cs_part = vector_rows(keys["b1"], w1.shape[0], rows)
        df8 = None
        if cs_part is not None:
            # the b1 gradient (column sums of df) leaves the dX GEMM's epilogue as partial rows -- and, in fp8 mode, df itself a
            # second time as the e5m2 operand of the first Linear's dX GEMM below
            if need_dx:
                df8 = _fused_fp8_out(run, f"{layer}.df", g.shape[0], w1.shape[0], w2.shape[0], g.device, grad=True)
            df = linear_dx(run, g, w2, run.act, fp8_site=f"{layer}.g_ffn", sign_bits=bits, colsum_part=cs_part, fp8_out=df8,
                           fp8_out_only=bool(df8 is not None and run.fp8_only and ln_w is not None))
        else:
            if bits is not None:
                df = linear_dx(run, g, w2, run.act, fp8_site=f"{layer}.g_ffn", sign_bits=bits)
            else:
                df = ops.gemm_nt(g, run.weights.bwd(w2), run.act, K=w2.shape[0], relu_mask=f)
            part, stride, n = sink.vector(keys["b1"], w1.shape[0])
            ops.colsum(df, part, stride, n)
    weight_grad(sink, keys["w1"], df, h, w1.shape[0], w1.shape[1])
    if not need_dx:
        return None
    d8 = (df8[0], df8[3]) if (w2 is not None and df8 is not None) else None
    if ln_w is not None:
        dh = linear_dx(run, df, w1, _dh_dtype(run, ln_w), fp8_site=f"{layer}.df", g8=d8)
        dx, g_next = _ln_tail(run, dh, x2d, ln_w, mean, rstd, dy if residual else None, sink, keys, emit)
        return (dx, g_next) if emit is not None else dx
    dx = linear_dx(run, df, w1, torch.float32, fp8_site=f"{layer}.df", g8=d8, residual=dy if residual else None)
    return (dx, None) if emit is not None else dx


# ------------------------------------------------------------------------------------------------
# plain Linear (lm_head): y = x W^T + b, fp32 out
def linear_fwd(run: Run, x2d: Tensor, w: Tensor, b: Optional[Tensor], pad_rows: bool = False, out: Optional[Tensor] = None):
    """pad_rows: give the fp32 output a leading dimension that is a multiple of 4 (a strided [M, N] view): with N = 50257 the
    rows of a packed buffer are only 4-byte aligned and the GEMM has to store element by element (2.0 ms instead of 1.0 ms
    for the GPT-2 logits).  Only for callers that consume the result through an `ld`-aware kernel (the engine's cross entropy)."""
    xa = _as_act(run, x2d)
    N = w.shape[0]
    if out is None and pad_rows and N % 4:
        out = torch.empty((x2d.shape[0], pad_to(N, 4)), dtype=torch.float32, device=x2d.device)[:, :N]
    y = ops.gemm_nt(xa, run.weights.fwd(w), out.dtype if out is not None else torch.float32, bias=b, out=out)
    return y, (xa,)


def linear_bwd_from_act(run: Run, saved, g: Tensor, w: Tensor, has_bias: bool, sink, keys, need_dx: bool = True,
                        bias_done: bool = False):
    """g: dY already in the activation dtype, leading dim padded to the granule with zeros."""
    (xa,) = saved
    N, K = w.shape
    if has_bias and not bias_done:
        part, stride, n = sink.vector(keys["b"], N)
        ops.colsum(g, part, stride, n, N=N)
    weight_grad(sink, keys["w"], g, xa, N, K)
    if not need_dx:
        return None
    d8 = getattr(g, "dg_fp8", None)
    w8 = getattr(run.weights, "bwd8_map", {}).get(w.data_ptr()) if run.fp8 else None
    wt_cols = w8[0].shape[1] if w8 is not None else run.weights.bwd(w).shape[1]
    Kp = k_pad(N, run.act)
    if ops._ld(g) < Kp or wt_cols < Kp:                           # caller padded to the granule only
        Kp = pad_to(N, granule(run.act))
    if d8 is not None and w8 is not None and fp8_k_ok(Kp) and ops._ld(d8[0]) >= Kp and w8[0].shape[1] >= Kp:
        # fp8 head: e5m2 dlogits (a-priori scale, from the loss kernel) x the e4m3 W^T shadow, contraction over the padded vocabulary
        return ops.gemm_nt(d8[0], w8[0], run.stream, K=Kp, scale_a=d8[1], scale_b=w8[1])
    return ops.gemm_nt(g, run.weights.bwd(w), run.stream, K=Kp)   # the head of the gradient stream


def linear_bwd(run: Run, saved, dy: Tensor, w: Tensor, has_bias: bool, sink, keys, need_dx: bool = True):
    """dy fp32 [M,N] (autograd path)."""
    N = w.shape[0]
    Np = k_pad(N, run.act)
    M = dy.shape[0]
    part = stride = n = None
    if has_bias:
        part, stride, n = sink.vector(keys["b"], N)
    if Np == N:
        g = ops.dropout_bwd_cast(dy, run.act, 0.0, None, 0, colsum_part=part, part_stride=stride or 0, n_partials=n or 0)
    else:
        # pad the contraction dim of the dX GEMM with zeros (V = 50257 is not a multiple of 8)
        gp = torch.zeros((M, Np), dtype=run.act, device=dy.device)
        g = gp[:, :N]
        ops.check(ops.lib.dg_dropout_bwd_cast(dy.data_ptr(), ops.dt_code(dy.dtype), ops._ld(dy), gp.data_ptr(), Np, ops.dt_code(run.act), M, N, 0.0, None, 0,
                                              None, 0, ops._p(part), stride or 0, n or 0, ops._stream()), "dg_dropout_bwd_cast")
    return linear_bwd_from_act(run, saved, g, w, has_bias, sink, keys, need_dx, bias_done=True)
