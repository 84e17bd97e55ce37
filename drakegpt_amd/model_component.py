"""Building blocks with DrakeGPT's nn.Module surface, computed by the gfx950 HIP kernels.

Mirror of the reference's src/model_component.py: same class names, same positional constructor
arguments, same attribute / state_dict layout (per-head ``key/query/value`` Linears without bias,
a persistent ``tril`` buffer per head, ``proj``, ``net.0`` / ``net.2``, ``ln1`` / ``ln2``), same
default initialisers drawn in the same order -- so ``torch.manual_seed(s)`` followed by construction
gives the reference's weights, and the reference's checkpoints load unchanged.

What differs is only HOW forward/backward are computed: the torch sub-modules below are parameter
containers; the arithmetic is one packed-QKV MFMA GEMM, a fused causal attention kernel, MFMA
GEMMs with fused bias/ReLU/dropout/residual epilogues and a wave-per-row LayerNorm, with
hand-written backward (drakegpt_amd/sublayers.py).  Inputs must live on the GPU: there is no CPU
path in this package.

precision: "fp32" (default; exact-fp32 MFMA, tracks the reference to ~1e-6), "bf16" (bf16 MFMA
operands and stored activations, fp32 accumulation / residual stream / master weights) or "fp8" (as
"bf16", with the Linears of the residual blocks on OCP fp8 operands: e4m3 forward, e5m2 gradients).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn as nn

from . import functional as HF
from . import ops

# "fp8": activations are stored as in "bf16"; the operands of the residual blocks' Linears (forward and dX) are quantised to OCP
# fp8 (e4m3 forward, e5m2 gradients) with per-tensor just-in-time scales for the block-scaled MFMA (BASELINE.json configs[4])
_PRECISIONS = {"fp32": torch.float32, "bf16": torch.bfloat16, "fp8": torch.bfloat16}


class HipModule(nn.Module):
    """Shared plumbing: precision switch and the dropout counter state."""

    def __init__(self, precision: str = "fp32"):
        super().__init__()
        if precision not in _PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}, got {precision!r}")
        self.precision = precision
        self.layer_index = 0
        self._rng_state: Optional[torch.Tensor] = None
        self._dropout_seed: Optional[int] = None

    @property
    def act_dtype(self) -> torch.dtype:
        return _PRECISIONS[self.precision]

    @property
    def fp8(self) -> bool:
        return self.precision == "fp8"

    @property
    def run_mode(self):
        """what the autograd shells hand to sublayers.Run: the activation dtype, or "fp8" (bf16 activations + fp8 GEMM operands)"""
        return "fp8" if self.fp8 else self.act_dtype

    def set_precision(self, precision: str) -> "HipModule":
        if precision not in _PRECISIONS:
            raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}, got {precision!r}")
        for m in self.modules():
            if isinstance(m, HipModule):
                m.precision = precision
        return self

    def seed_dropout(self, seed: int) -> "HipModule":
        """Fix the (seed, step) counter the kernels hash for dropout masks."""
        self._dropout_seed = int(seed)
        self._rng_state = None
        return self

    def _rng_snapshot(self, device, any_dropout: bool) -> Optional[torch.Tensor]:
        """device {seed, step} for this forward (None in eval mode); the live counter advances."""
        if not (self.training and any_dropout):
            return None
        if self._rng_state is None or self._rng_state.device != device:
            seed = self._dropout_seed if self._dropout_seed is not None else torch.initial_seed()
            self._rng_state = ops.new_rng_state(seed, device)
        snap = self._rng_state.clone()
        ops.state_advance(self._rng_state)
        return snap


def _check_width(C: int, act: torch.dtype, what: str) -> None:
    g = 8 if act == torch.bfloat16 else 4
    if C % g:
        raise ValueError(f"{what}={C} must be a multiple of {g} for the {act} MFMA GEMMs")


def _pack_qkv(heads) -> torch.Tensor:
    """[Q heads | K heads | V heads] rows, each head (H, C): the packed operand of the one QKV GEMM
    that replaces the 3*NH per-head Linears (ref: src/model_component.py:392-393,404)."""
    return torch.cat([h.query.weight for h in heads] + [h.key.weight for h in heads] + [h.value.weight for h in heads], dim=0)


# ------------------------------------------------------------------------------------------------
class Head(HipModule):
    """Single causal self-attention head. ref: src/model_component.py:5-66 (README: SingleHeadAttention)."""

    def __init__(self, head_size, embedding_dim, context_length, *, precision: str = "fp32"):
        super().__init__(precision)
        self.head_size = head_size
        self.scale = head_size ** -0.5
        self.key = nn.Linear(embedding_dim, head_size, bias=False)
        self.query = nn.Linear(embedding_dim, head_size, bias=False)
        self.value = nn.Linear(embedding_dim, head_size, bias=False)
        # kept for checkpoint compatibility only: the kernels mask by index (j > i), never read it
        self.register_buffer("tril", torch.tril(torch.ones(context_length, context_length)))
        self._p = 0.0

    def forward(self, x):
        B, T, C = x.shape
        _check_width(C, self.act_dtype, "embedding_dim")
        _check_width(3 * self.head_size, self.act_dtype, "3*head_size")
        rng = self._rng_snapshot(x.device, self._p > 0.0)
        return HF.attention(x, None, None, _pack_qkv([self]), None, None, rng, self.run_mode, False, 1,
                            self.head_size, self._p, 0.0, self.layer_index)


class Head2(Head):
    """Head with dropout on the attention probabilities. ref: src/model_component.py:343-407."""

    def __init__(self, head_size, embedding_dim, context_length, dropout, *, precision: str = "fp32"):
        super().__init__(head_size, embedding_dim, context_length, precision=precision)
        self.dropout = nn.Dropout(dropout)
        self._p = float(dropout)


class _MultiHeadBase(HipModule):
    def _attend(self, x, ln_w=None, ln_b=None, residual=False, rng="auto"):
        B, T, C = x.shape
        heads = list(self.heads)
        H = heads[0].head_size
        NH = len(heads)
        act = self.act_dtype
        _check_width(C, act, "embedding_dim")
        _check_width(NH * H, act, "num_heads*head_size")
        proj = getattr(self, "proj", None)
        p = float(getattr(self, "_p", 0.0))
        if rng == "auto":
            rng = self._rng_snapshot(x.device, p > 0.0)
        return HF.attention(x, ln_w, ln_b, _pack_qkv(heads), None if proj is None else proj.weight,
                            None if proj is None else proj.bias, rng, self.run_mode, residual, NH, H, p, p, self.layer_index)

    def forward(self, x):
        return self._attend(x)


class MultiHeadAttention(_MultiHeadBase):
    """Concatenated heads. ref: src/model_component.py:69-103."""

    def __init__(self, num_heads, head_size, embedding_dim, context_length, *, precision: str = "fp32"):
        super().__init__(precision)
        self.heads = nn.ModuleList([Head(head_size, embedding_dim, context_length, precision=precision) for _ in range(num_heads)])


class MultiHeadAttention2(_MultiHeadBase):
    """Heads + output projection. ref: src/model_component.py:220-261."""

    def __init__(self, num_heads, head_size, embedding_dim, context_length, *, precision: str = "fp32"):
        super().__init__(precision)
        self.heads = nn.ModuleList([Head(head_size, embedding_dim, context_length, precision=precision) for _ in range(num_heads)])
        self.proj = nn.Linear(embedding_dim, embedding_dim)


class MultiHeadAttention3(_MultiHeadBase):
    """Heads (prob-dropout) + projection + dropout. ref: src/model_component.py:409-455."""

    def __init__(self, num_heads, head_size, embedding_dim, context_length, dropout, *, precision: str = "fp32"):
        super().__init__(precision)
        self.heads = nn.ModuleList(
            [Head2(head_size, embedding_dim, context_length, dropout, precision=precision) for _ in range(num_heads)])
        self.proj = nn.Linear(embedding_dim, embedding_dim)
        self.dropout = nn.Dropout(dropout)
        self._p = float(dropout)


# ------------------------------------------------------------------------------------------------
class _FeedForwardBase(HipModule):
    def _ffn(self, x, ln_w=None, ln_b=None, residual=False, rng="auto"):
        act = self.act_dtype
        _check_width(x.shape[-1], act, "embedding_dim")
        lin1 = self.net[0]
        lin2 = self.net[2] if len(self.net) > 2 else None
        p = float(getattr(self, "_p", 0.0))
        if rng == "auto":
            rng = self._rng_snapshot(x.device, p > 0.0)
        return HF.feed_forward(x, ln_w, ln_b, lin1.weight, lin1.bias, None if lin2 is None else lin2.weight,
                               None if lin2 is None else lin2.bias, rng, self.run_mode, residual, p, self.layer_index)

    def forward(self, x):
        return self._ffn(x)


class FeedForward(_FeedForwardBase):
    """Linear(C,C) + ReLU. ref: src/model_component.py:106-137."""

    def __init__(self, embedding_dim, *, precision: str = "fp32"):
        super().__init__(precision)
        self.net = nn.Sequential(nn.Linear(embedding_dim, embedding_dim), nn.ReLU())


class FeedForward2(_FeedForwardBase):
    """Linear(C,4C) + ReLU + Linear(4C,C). ref: src/model_component.py:184-217."""

    def __init__(self, embedding_dim, *, precision: str = "fp32"):
        super().__init__(precision)
        self.net = nn.Sequential(nn.Linear(embedding_dim, 4 * embedding_dim), nn.ReLU(),
                                 nn.Linear(4 * embedding_dim, embedding_dim))


class FeedForward3(_FeedForwardBase):
    """FeedForward2 + Dropout. ref: src/model_component.py:308-340."""

    def __init__(self, embedding_dim, dropout, *, precision: str = "fp32"):
        super().__init__(precision)
        self.net = nn.Sequential(nn.Linear(embedding_dim, 4 * embedding_dim), nn.ReLU(),
                                 nn.Linear(4 * embedding_dim, embedding_dim), nn.Dropout(dropout))
        self._p = float(dropout)


# ------------------------------------------------------------------------------------------------
class Block(HipModule):
    """sa_head then ffwd, no residual. ref: src/model_component.py:140-181 (note the argument order)."""

    def __init__(self, embedding_dim, context_length, num_heads, *, precision: str = "fp32"):
        super().__init__(precision)
        head_size = embedding_dim // num_heads
        self.sa_head = MultiHeadAttention(num_heads, head_size, embedding_dim, context_length, precision=precision)
        self.ffwd = FeedForward(embedding_dim, precision=precision)

    def forward(self, x, rng=None):
        return self.ffwd._ffn(self.sa_head._attend(x, rng=None), rng=None)


class ResidualBlock(HipModule):
    """x + sa_head(x); x + ffwd(x). ref: src/model_component.py:264-306."""

    def __init__(self, embedding_dim, num_heads, context_length, *, precision: str = "fp32"):
        super().__init__(precision)
        head_size = embedding_dim // num_heads
        self.sa_head = MultiHeadAttention2(num_heads, head_size, embedding_dim, context_length, precision=precision)
        self.ffwd = FeedForward2(embedding_dim, precision=precision)

    def forward(self, x, rng=None):
        x = self.sa_head._attend(x, residual=True, rng=None)     # residual add fused in the proj epilogue
        return self.ffwd._ffn(x, residual=True, rng=None)


class ResidualBlock2(HipModule):
    """Pre-LN block: x + sa_head(ln1(x)); x + ffwd(ln2(x)). ref: src/model_component.py:458-507."""

    def __init__(self, embedding_dim, num_heads, context_length, dropout, *, precision: str = "fp32"):
        super().__init__(precision)
        head_size = embedding_dim // num_heads
        self.sa_head = MultiHeadAttention3(num_heads, head_size, embedding_dim, context_length, dropout, precision=precision)
        self.ffwd = FeedForward3(embedding_dim, dropout, precision=precision)
        self.ln1 = nn.LayerNorm(embedding_dim)
        self.ln2 = nn.LayerNorm(embedding_dim)
        self._p = float(dropout)

    def set_layer_index(self, layer: int) -> None:
        for m in self.modules():
            if isinstance(m, HipModule):
                m.layer_index = layer

    def forward(self, x, rng="auto"):
        if rng == "auto":
            rng = self._rng_snapshot(x.device, self._p > 0.0)
        # LayerNorm is fused in front of each sub-layer and the residual add into its last GEMM
        x = self.sa_head._attend(x, self.ln1.weight, self.ln1.bias, residual=True, rng=rng)
        return self.ffwd._ffn(x, self.ln2.weight, self.ln2.bias, residual=True, rng=rng)
