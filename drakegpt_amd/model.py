"""The six DrakeGPT language models on the gfx950 HIP kernels.

Mirror of the reference's src/model.py: same class names, constructor signatures, attribute and
state_dict layout, ``forward(idx, targets=None) -> (logits, loss)`` (logits flattened to (B*T, V)
when targets are given, src/model.py:601-609) and ``generate(idx, max_new_tokens)``.

Reference quirks kept on purpose:
  * TransformerLM owns ``ln_f`` but never applies it (src/model.py:572,598-599): the parameters
    exist in the state_dict, take no part in compute and never receive a gradient;
  * MultiHeadAttentionLM's per-head size is head_size // num_heads (src/model.py:264);
  * generate() crops to context_length (src/model.py:625) and samples from torch's CPU generator so that, given
    matching logits, the sampled indices are bit-identical to the reference run on CPU.  The five earlier-stage
    models re-run the full forward per token as the reference does; TransformerLM.generate keeps a K/V cache while
    the sequence fits the context window (same logits, see its docstring) and falls back to the reference
    algorithm once the window slides;
  * token / target ids outside [0, vocab_size) raise IndexError at the module boundary, as nn.Embedding and
    F.cross_entropy do in the reference (the kernels themselves clamp and never fault).
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Optional

import torch
import torch.nn as nn

from . import functional as HF
from . import ops
from .model_component import (Block, Head, HipModule, MultiHeadAttention, ResidualBlock, ResidualBlock2)


def model_params(params: dict, model_type: str, vocab_size: int) -> int:
    """The reference's parameter-count ESTIMATE (src/model.py:8-63), reproduced because train.py
    prints it (src/train.py:112-113).  It is not the true count; use
    ``sum(p.numel() for p in model.parameters())`` for that."""
    C, T, L = params["embedding_dim"], params["context_length"], params["num_layers"]
    kqv = 3 * C * C
    attention = C + kqv + C * C
    ffw = C * 4 * C
    mlp = C + 2 * ffw
    total = C * vocab_size
    if model_type != "BigramLM":
        total += C * T + kqv + C * vocab_size
    if model_type in ("SingleHeadAttentionLM", "MultiHeadAttentionLM"):
        total += C * T + C * vocab_size
    if model_type == "BlocksLM":
        total += (kqv + ffw) * L
    if model_type == "ResidualBlocksLM":
        total += (kqv + 2 * ffw) * L
    if model_type == "TransformerLM":
        total += (attention + mlp) * L + vocab_size
    return total


class _LM(HipModule):
    """forward/generate scaffolding shared by the five position-aware models."""

    context_length: Optional[int]

    def _embed(self, idx):
        return HF.embed(idx, self.token_embedding_table.weight, self.position_embedding_table.weight)

    def _body(self, x, rng):          # overridden
        raise NotImplementedError

    def _any_dropout(self) -> bool:
        return False

    def _head(self, x):
        return HF.linear(x, self.lm_head.weight, self.lm_head.bias, self.act_dtype)

    check_ids = True      # False: skip the id range check (one or two device-to-host syncs per forward that nn.Embedding does not
                          # have) once a data source has been validated -- ids out of range are then CLAMPED by the kernels

    def _check_ids(self, idx, targets):
        ops._chk(idx, "idx", torch.int64, contiguous=False)
        if targets is not None:
            ops._chk(targets, "targets", torch.int64, contiguous=False)
        if not self.check_ids:
            return
        V = self.token_embedding_table.weight.shape[0]
        ops.check_ids(idx, V, "idx")
        if targets is not None:
            ops.check_ids(targets, V if not hasattr(self, "lm_head") else self.lm_head.weight.shape[0], "targets")

    def forward(self, idx, targets=None):
        if idx.dim() != 2:
            raise ValueError("idx must be (B, T)")
        self._check_ids(idx, targets)
        rng = self._rng_snapshot(idx.device, self._any_dropout())
        logits = self._head(self._body(self._embed(idx), rng))
        if targets is None:
            return logits, None
        B, T, V = logits.shape
        logits = logits.view(B * T, V)
        loss = HF.cross_entropy(logits, targets.reshape(B * T))
        return logits, loss

    @torch.no_grad()
    def _last_probs(self, idx):
        logits, _ = self(idx)
        return ops.softmax_rows(logits[:, -1, :])

    def generate(self, idx, max_new_tokens, generator: Optional[torch.Generator] = None):
        """ref: src/model.py:611-636.  softmax runs on the GPU; torch.multinomial runs on the host CPU
        generator (the global one unless `generator` is given), as it does in the reference on CPU."""
        for _ in range(max_new_tokens):
            cond = idx if self.context_length is None else idx[:, -self.context_length:]
            probs = self._last_probs(cond.contiguous())
            nxt = torch.multinomial(probs.cpu(), num_samples=1, generator=generator)
            idx = torch.cat((idx, nxt.to(idx.device)), dim=1)
        return idx


class BigramLM(_LM):
    """ref: src/model.py:65-130: logits are a (V,V) table lookup."""

    def __init__(self, vocab_size, *, precision: str = "fp32"):
        super().__init__(precision)
        self.context_length = None
        self.token_embedding_table = nn.Embedding(vocab_size, vocab_size)

    def forward(self, idx, targets=None):
        self._check_ids(idx, targets)
        logits = HF.embed(idx, self.token_embedding_table.weight, None)
        if targets is None:
            return logits, None
        B, T, V = logits.shape
        logits = logits.view(B * T, V)
        return logits, HF.cross_entropy(logits, targets.reshape(B * T))


class SingleHeadAttentionLM(_LM):
    """ref: src/model.py:133-227."""

    def __init__(self, vocab_size, embedding_dim, context_length, head_size, *, precision: str = "fp32"):
        super().__init__(precision)
        self.context_length = context_length
        self.token_embedding_table = nn.Embedding(vocab_size, embedding_dim)
        self.position_embedding_table = nn.Embedding(context_length, embedding_dim)
        self.sa_head = Head(head_size, embedding_dim, context_length, precision=precision)
        self.lm_head = nn.Linear(embedding_dim, vocab_size)

    def _body(self, x, rng):
        return self.sa_head(x)


class MultiHeadAttentionLM(_LM):
    """ref: src/model.py:230-331."""

    def __init__(self, vocab_size, embedding_dim, context_length, head_size, num_heads, *, precision: str = "fp32"):
        super().__init__(precision)
        self.context_length = context_length
        self.token_embedding_table = nn.Embedding(vocab_size, embedding_dim)
        self.position_embedding_table = nn.Embedding(context_length, embedding_dim)
        self.sa_head = MultiHeadAttention(num_heads, head_size // num_heads, embedding_dim, context_length, precision=precision)
        self.lm_head = nn.Linear(embedding_dim, vocab_size)

    def _body(self, x, rng):
        return self.sa_head(x)


class _BlocksLM(_LM):
    def _body(self, x, rng):
        for blk in self.blocks:
            x = blk(x, rng)
        return x


class BlocksLM(_BlocksLM):
    """ref: src/model.py:334-432."""

    def __init__(self, vocab_size, embedding_dim, context_length, num_heads, num_layers, *, precision: str = "fp32"):
        super().__init__(precision)
        self.context_length = context_length
        self.token_embedding_table = nn.Embedding(vocab_size, embedding_dim)
        self.position_embedding_table = nn.Embedding(context_length, embedding_dim)
        self.blocks = nn.Sequential(*[Block(embedding_dim, context_length, num_heads, precision=precision) for _ in range(num_layers)])
        self.lm_head = nn.Linear(embedding_dim, vocab_size)


class ResidualBlocksLM(_BlocksLM):
    """ref: src/model.py:435-533."""

    def __init__(self, vocab_size, embedding_dim, context_length, num_heads, num_layers, *, precision: str = "fp32"):
        super().__init__(precision)
        self.context_length = context_length
        self.token_embedding_table = nn.Embedding(vocab_size, embedding_dim)
        self.position_embedding_table = nn.Embedding(context_length, embedding_dim)
        self.blocks = nn.Sequential(*[ResidualBlock(embedding_dim, num_heads, context_length, precision=precision) for _ in range(num_layers)])
        self.lm_head = nn.Linear(embedding_dim, vocab_size)


class TransformerLM(_BlocksLM):
    """ref: src/model.py:535-636."""

    def __init__(self, vocab_size, embedding_dim, context_length, num_heads, num_layers, dropout, *, precision: str = "fp32"):
        super().__init__(precision)
        self.context_length = context_length
        self.token_embedding_table = nn.Embedding(vocab_size, embedding_dim)
        self.position_embedding_table = nn.Embedding(context_length, embedding_dim)
        self.blocks = nn.Sequential(
            *[ResidualBlock2(embedding_dim, num_heads, context_length, dropout, precision=precision) for _ in range(num_layers)])
        for l, blk in enumerate(self.blocks):
            blk.set_layer_index(l)
        self.ln_f = nn.LayerNorm(embedding_dim)      # never applied: src/model.py:598-599
        self.lm_head = nn.Linear(embedding_dim, vocab_size)
        self._p = float(dropout)

    def _any_dropout(self) -> bool:
        return self._p > 0.0


    # ------------------------------------------------------------------ KV-cached decoding
    @torch.no_grad()
    def _decode_weights(self):
        from . import sublayers as S
        wp = S.OnTheFlyWeights(self.act_dtype)
        ws = []
        for blk in self.blocks:
            heads = list(blk.sa_head.heads)
            wqkv = torch.cat([h.query.weight for h in heads] + [h.key.weight for h in heads] + [h.value.weight for h in heads], 0)
            ws.append(dict(ln1w=blk.ln1.weight, ln1b=blk.ln1.bias, wqkv=wp.fwd(wqkv), wproj=wp.fwd(blk.sa_head.proj.weight),
                           bproj=blk.sa_head.proj.bias, ln2w=blk.ln2.weight, ln2b=blk.ln2.bias, w1=wp.fwd(blk.ffwd.net[0].weight),
                           b1=blk.ffwd.net[0].bias, w2=wp.fwd(blk.ffwd.net[2].weight), b2=blk.ffwd.net[2].bias))
        return ws, wp.fwd(self.lm_head.weight)

    @torch.no_grad()
    def _decode_step(self, tok_col, t, caches, ws, w_lm):
        """logits (B, V) of the token at position t; K/V of position t are appended to the caches."""
        act = self.act_dtype
        B = tok_col.shape[0]
        NH = len(self.blocks[0].sa_head.heads)
        H = self.blocks[0].sa_head.heads[0].head_size
        x = ops.embed_fwd(tok_col, self.token_embedding_table.weight, self.position_embedding_table.weight[t:t + 1]).view(B, -1)
        for W, cache in zip(ws, caches):
            h, _, _ = ops.layernorm_fwd(x, W["ln1w"], W["ln1b"], act)
            ops.gemm_nt(h, W["wqkv"], act, out=cache[:, t])                       # q/k/v row t straight into the cache
            o = ops.attn_decode(cache, t, NH, H, H ** -0.5)
            x = ops.gemm_nt(o, W["wproj"], torch.float32, bias=W["bproj"], residual=x)
            h, _, _ = ops.layernorm_fwd(x, W["ln2w"], W["ln2b"], act)
            f = ops.gemm_nt(h, W["w1"], act, bias=W["b1"], relu=True)
            x = ops.gemm_nt(f, W["w2"], torch.float32, bias=W["b2"], residual=x)
        xa = x if act == torch.float32 else ops.cast(x, act)
        return ops.gemm_nt(xa, w_lm, torch.float32, bias=self.lm_head.bias)

    @torch.no_grad()
    def _prefill(self, idx, caches, ws, w_lm):
        """the prompt in ONE pass through the training-forward kernels (not one decode step per position): fills every layer's
        K/V cache rows [0, t0) and returns the logits (B, V) of the last prompt position -- the values the reference's full
        forward produces for it (the uncached path runs exactly these kernels)."""
        act = self.act_dtype
        B, t0 = idx.shape
        NH = len(self.blocks[0].sa_head.heads)
        H = self.blocks[0].sa_head.heads[0].head_size
        x = ops.embed_fwd(idx, self.token_embedding_table.weight, self.position_embedding_table.weight).view(B * t0, -1)
        for W, cache in zip(ws, caches):
            h, _, _ = ops.layernorm_fwd(x, W["ln1w"], W["ln1b"], act)
            qkv = ops.gemm_nt(h, W["wqkv"], act)
            cache[:, :t0].copy_(qkv.view(B, t0, -1))
            o, _ = ops.attn_fwd(qkv, B, t0, NH, H, H ** -0.5, 0.0, None, 0)
            x = ops.gemm_nt(o, W["wproj"], torch.float32, bias=W["bproj"], residual=x)
            h, _, _ = ops.layernorm_fwd(x, W["ln2w"], W["ln2b"], act)
            f = ops.gemm_nt(h, W["w1"], act, bias=W["b1"], relu=True)
            x = ops.gemm_nt(f, W["w2"], torch.float32, bias=W["b2"], residual=x)
        last = x.view(B, t0, -1)[:, -1].contiguous()
        xa = last if act == torch.float32 else ops.cast(last, act)
        return ops.gemm_nt(xa, w_lm, torch.float32, bias=self.lm_head.bias)

    def generate(self, idx, max_new_tokens, generator: Optional[torch.Generator] = None, use_cache: bool = True):
        """ref: src/model.py:611-636.  While the sequence still fits the context window the per-layer K/V of the
        tokens seen so far are kept (training layout, [B, ctx, 3C]) and only the new position is computed; once the
        window starts to slide every position embedding shifts, the cache is void, and decoding continues exactly
        as the reference does (full forward on the cropped window).  Dropout is off in both paths only in eval()
        mode -- like the reference, train() mode samples with dropout through the uncached path."""
        if not use_cache or self.training or idx.shape[1] >= self.context_length:
            return super().generate(idx, max_new_tokens, generator)
        B, t0 = idx.shape
        self._check_ids(idx, None)
        C3 = 3 * self.token_embedding_table.weight.shape[1]
        ws, w_lm = self._decode_weights()
        caches = [torch.zeros((B, self.context_length, C3), dtype=self.act_dtype, device=idx.device) for _ in self.blocks]
        logits = self._prefill(idx.contiguous(), caches, ws, w_lm)      # the whole prompt in one pass
        produced = 0
        while produced < max_new_tokens:
            probs = ops.softmax_rows(logits)
            nxt = torch.multinomial(probs.cpu(), num_samples=1, generator=generator).to(idx.device)
            idx = torch.cat((idx, nxt), dim=1)
            produced += 1
            if produced == max_new_tokens:
                break
            t = idx.shape[1] - 1
            if t >= self.context_length:                         # window slides: fall back to the reference algorithm
                return super().generate(idx, max_new_tokens - produced, generator)
            logits = self._decode_step(nxt.contiguous(), t, caches, ws, w_lm)
        return idx


MODEL_CLASSES = OrderedDict(
    BigramLM=BigramLM,
    SingleHeadAttentionLM=SingleHeadAttentionLM,
    MultiHeadAttentionLM=MultiHeadAttentionLM,
    BlocksLM=BlocksLM,
    ResidualBlocksLM=ResidualBlocksLM,
    TransformerLM=TransformerLM,
)
