"""torch.autograd.Function shells over drakegpt_amd.sublayers -- the glue that lets the HIP
kernels sit behind the reference's nn.Module surface (forward on the caller's thread, backward
on autograd's worker thread; both only enqueue kernels on the current stream)."""
from __future__ import annotations

from typing import Optional

import torch

from . import ops
from . import sublayers as S

Tensor = torch.Tensor


def _run(act, rng: Optional[Tensor]) -> S.Run:
    """act: the activation dtype, or "fp8" (HipModule.run_mode): bf16 activations, fp8 operands for the block Linears"""
    fp8 = act == "fp8"
    dt = torch.bfloat16 if fp8 else act
    return S.Run(act=dt, rng=rng, weights=S.OnTheFlyWeights(dt), fp8=fp8)


class EmbedFn(torch.autograd.Function):
    """tok[idx] + pos[arange(T)]  (ref: src/model.py:595-597; BigramLM :96 with pos=None)"""

    @staticmethod
    def forward(ctx, idx: Tensor, tok: Tensor, pos: Optional[Tensor]):
        ctx.save_for_backward(idx)
        ctx.tok_shape = tok.shape
        ctx.pos_shape = None if pos is None else pos.shape
        return ops.embed_fwd(idx, tok.contiguous(), None if pos is None else pos.contiguous())

    @staticmethod
    def backward(ctx, dx: Tensor):
        (idx,) = ctx.saved_tensors
        dx = dx.contiguous()
        dtok = torch.empty(ctx.tok_shape, dtype=torch.float32, device=dx.device)
        dpos = None
        if ctx.pos_shape is not None:
            dpos = torch.zeros(ctx.pos_shape, dtype=torch.float32, device=dx.device)
        ops.embed_bwd(idx, dx, dtok, None if dpos is None else dpos[: idx.shape[1]])
        return None, dtok, dpos


class AttnFn(torch.autograd.Function):
    """[LayerNorm ->] packed QKV -> causal attention [-> proj -> dropout] [+ residual]"""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, wqkv, wproj, bproj, rng, act, residual, NH, H, p_attn, p_proj, layer):
        B, T, Cd = x.shape
        run = _run(act, rng)
        x2d = x.contiguous().view(B * T, Cd)
        y, saved = S.attn_fwd(run, x2d, ln_w, ln_b, wqkv, wproj, bproj, residual, B, T, NH, H, p_attn, p_proj, layer)
        x2d_s, h, mean, rstd, qkv, o, lse = saved
        ctx.save_for_backward(x2d_s, h, mean, rstd, qkv, o, lse, ln_w, wqkv, wproj, rng)
        ctx.cfg = (act, residual, B, T, NH, H, p_attn, p_proj, layer, bproj is not None)
        return y.view(B, T, -1)

    @staticmethod
    def backward(ctx, dy):
        x2d, h, mean, rstd, qkv, o, lse, ln_w, wqkv, wproj, rng = ctx.saved_tensors
        act, residual, B, T, NH, H, p_attn, p_proj, layer, has_bproj = ctx.cfg
        run = _run(act, rng)
        dy2 = dy.contiguous().view(B * T, -1)
        sink = S.LocalSink(B * T, dy.device)
        keys = {"wqkv": "wqkv", "wproj": "wproj", "bproj": "bproj", "ln_w": "ln_w", "ln_b": "ln_b"}
        dx = S.attn_bwd(run, (x2d, h, mean, rstd, qkv, o, lse), dy2, ln_w, wqkv, wproj, residual, B, T, NH, H,
                        p_attn, p_proj, layer, sink, keys, need_dx=ctx.needs_input_grad[0])
        g = sink.finish()
        return (None if dx is None else dx.view(B, T, -1), g.get("ln_w"), g.get("ln_b"), g["wqkv"], g.get("wproj"),
                g.get("bproj") if has_bproj else None, None, None, None, None, None, None, None, None)


class FFNFn(torch.autograd.Function):
    """[LayerNorm ->] Linear -> ReLU [-> Linear -> dropout] [+ residual]"""

    @staticmethod
    def forward(ctx, x, ln_w, ln_b, w1, b1, w2, b2, rng, act, residual, p, layer):
        B, T, Cd = x.shape
        run = _run(act, rng)
        x2d = x.contiguous().view(B * T, Cd)
        y, saved = S.ffn_fwd(run, x2d, ln_w, ln_b, w1, b1, w2, b2, residual, p, layer)
        x2d_s, h, mean, rstd, f, bits = saved
        ctx.save_for_backward(x2d_s, h, mean, rstd, f, ln_w, w1, w2, rng, bits)
        ctx.cfg = (act, residual, B, T, p, layer)
        return y.view(B, T, -1)

    @staticmethod
    def backward(ctx, dy):
        x2d, h, mean, rstd, f, ln_w, w1, w2, rng, bits = ctx.saved_tensors
        act, residual, B, T, p, layer = ctx.cfg
        run = _run(act, rng)
        dy2 = dy.contiguous().view(B * T, -1)
        sink = S.LocalSink(B * T, dy.device)
        keys = {"w1": "w1", "b1": "b1", "w2": "w2", "b2": "b2", "ln_w": "ln_w", "ln_b": "ln_b"}
        dx = S.ffn_bwd(run, (x2d, h, mean, rstd, f, bits), dy2, ln_w, w1, w2, residual, p, layer, sink, keys,
                       need_dx=ctx.needs_input_grad[0])
        g = sink.finish()
        return (None if dx is None else dx.view(B, T, -1), g.get("ln_w"), g.get("ln_b"), g["w1"], g["b1"], g.get("w2"),
                g.get("b2"), None, None, None, None, None)


class LinearFn(torch.autograd.Function):
    """y = x W^T + b with fp32 output (lm_head, ref: src/model.py:599)"""

    @staticmethod
    def forward(ctx, x, w, b, act):
        lead = x.shape[:-1]
        x2d = x.contiguous().view(-1, x.shape[-1])
        run = _run(act, None)
        y, (xa,) = S.linear_fwd(run, x2d, w, b)
        ctx.save_for_backward(xa, w)
        ctx.cfg = (act, b is not None, lead)
        return y.view(*lead, w.shape[0])

    @staticmethod
    def backward(ctx, dy):
        xa, w = ctx.saved_tensors
        act, has_bias, lead = ctx.cfg
        run = _run(act, None)
        dy2 = dy.contiguous().view(-1, w.shape[0])
        sink = S.LocalSink(dy2.shape[0], dy.device)
        dx = S.linear_bwd(run, (xa,), dy2, w, has_bias, sink, {"w": "w", "b": "b"}, need_dx=ctx.needs_input_grad[0])
        g = sink.finish()
        return (None if dx is None else dx.view(*lead, w.shape[1]), g["w"], g.get("b"), None)


class CrossEntropyFn(torch.autograd.Function):
    """mean cross entropy over the rows of logits [M,V] (ref: src/model.py:606-607)"""

    @staticmethod
    def forward(ctx, logits: Tensor, targets: Tensor):
        M, V = logits.shape
        logits = logits.contiguous()
        rows = ops.cross_entropy(logits, targets, V)
        loss = ops.reduce_sum(rows, 1.0 / M)
        ctx.save_for_backward(logits, targets)
        return loss

    @staticmethod
    def backward(ctx, dloss: Tensor):
        logits, targets = ctx.saved_tensors
        M, V = logits.shape
        dlogits = torch.empty_like(logits)
        rows = torch.empty((M,), dtype=torch.float32, device=logits.device)
        ops.cross_entropy(logits, targets, V, dlogits=dlogits, grad_scale=1.0 / M,
                          grad_scale_dev=dloss.contiguous().to(torch.float32), loss_rows=rows)
        return dlogits, None


def embed(idx, tok, pos):
    return EmbedFn.apply(idx, tok, pos)


def attention(x, ln_w, ln_b, wqkv, wproj, bproj, rng, act, residual, NH, H, p_attn, p_proj, layer):
    return AttnFn.apply(x, ln_w, ln_b, wqkv, wproj, bproj, rng, act, residual, NH, H, p_attn, p_proj, layer)


def feed_forward(x, ln_w, ln_b, w1, b1, w2, b2, rng, act, residual, p, layer):
    return FFNFn.apply(x, ln_w, ln_b, w1, b1, w2, b2, rng, act, residual, p, layer)


def linear(x, w, b, act):
    return LinearFn.apply(x, w, b, act)


def cross_entropy(logits, targets):
    return CrossEntropyFn.apply(logits, targets)
