"""CPU oracle for the DrakeGPT training hot path -- TEST INFRASTRUCTURE ONLY.

This file restates, as pure functions over a reference-layout ``state_dict``, the
arithmetic of the reference's six language models and of its training step, using
stock fp32 torch CPU ops in the same order as the reference issues them.  Nothing
under ``drakegpt_amd/`` may import it; only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg do, and there only as the checker / the
timed CPU baseline.

Pinning: ``oracle/make_golden.py`` imports the real reference from
``/root/reference/src`` (possible only in the build container), loads the six
shipped checkpoints and asserts that every function here is BIT-IDENTICAL to the
reference (logits, loss, every gradient, sampled tokens, train-mode dropout under
the same seed, 5-step AdamW trajectory).  It then writes the fixtures in
``tests/golden/``.  The reference itself publishes no tests or golden vectors
(SURVEY.md section 8c), so the goldens are outputs of the reference run here.

All ``ref:`` citations are relative to /root/reference/.
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

MODEL_NAMES = (
    "BigramLM",
    "SingleHeadAttentionLM",
    "MultiHeadAttentionLM",
    "BlocksLM",
    "ResidualBlocksLM",
    "TransformerLM",
)

# ref: src/config.py:14-25 (PARAMS) and :27-38 (SCALE_PARAMS)
TINY = dict(context_length=8, batch_size=32, base_lr=1e-3, max_lr=5e-3, betas=(0.9, 0.95),
            embedding_dim=32, head_size=32, num_heads=4, num_layers=3, dropout=0.1)
SCALED = dict(context_length=256, batch_size=64, base_lr=3e-4, max_lr=6e-4, betas=(0.9, 0.95),
              embedding_dim=384, head_size=64, num_heads=6, num_layers=6, dropout=0.2)


# --------------------------------------------------------------------------------------
# components
# --------------------------------------------------------------------------------------
def _drop(x: Tensor, p: float, training: bool, mask: Optional[Tensor]) -> Tensor:
    """nn.Dropout semantics (ref: src/model_component.py:376,401,433,454,324).

    With ``mask`` given (a {0,1} keep-mask of x's shape) the dropout is made explicit:
    ``x * mask / (1-p)`` -- used to compare against the HIP kernels' own RNG stream.
    """
    if mask is not None:
        return x * mask * (1.0 / (1.0 - p))
    return F.dropout(x, p, training)


# --------------------------------------------------------------------------------------
# rounding model of the kernels' bf16 mode (NOT reference arithmetic: the reference is fp32 only)
# --------------------------------------------------------------------------------------
class _RoundGrad(torch.autograd.Function):
    """identity in forward; the gradient is rounded to bf16 on its way back -- the places where the HIP path
    stores a gradient as a bf16 GEMM operand while the forward value stays fp32 (dropout-backward of the proj /
    FeedForward outputs, dlogits, dS)."""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.bfloat16().float()


def _rb(x: Tensor, on: bool) -> Tensor:
    """value AND gradient rounded to bf16 (autograd of the cast pair rounds the gradient too): the places where the
    HIP path stores an activation as bf16 (LayerNorm output, q/k/v, dropped-out probabilities, attention output,
    FeedForward hidden layer, lm_head input, bf16 weight copies)."""
    return x.bfloat16().float() if on else x


def _rg(x: Tensor, on: bool) -> Tensor:
    return _RoundGrad.apply(x) if on else x


def head_forward(sd: SD, prefix: str, x: Tensor, p: float = 0.0, training: bool = False,
                 mask: Optional[Tensor] = None, bf16: bool = False) -> Tensor:
    """One causal attention head. ref: Head.forward src/model_component.py:40-66,
    Head2.forward :378-407 (same math; Head2 adds dropout on the probabilities)."""
    wk, wq, wv = sd[prefix + "key.weight"], sd[prefix + "query.weight"], sd[prefix + "value.weight"]
    T = x.shape[1]
    hs = wk.shape[0]
    k = _rb(F.linear(x, _rb(wk, bf16)), bf16)              # mc:392
    q = _rb(F.linear(x, _rb(wq, bf16)), bf16)              # mc:393
    w = _rg(q @ k.transpose(-2, -1) * hs ** -0.5, bf16)    # mc:396 (scale after the matmul)
    tril = torch.tril(torch.ones(T, T))                    # mc:372-375, sliced [:T,:T] at :398
    w = w.masked_fill(tril == 0, float("-inf"))            # mc:397-399
    w = F.softmax(w, dim=-1)                               # mc:400
    if p > 0.0 or mask is not None:
        w = _drop(w, p, training, mask)                    # mc:401 (no renormalisation)
    v = _rb(F.linear(x, _rb(wv, bf16)), bf16)              # mc:404
    return _rb(_rb(w, bf16) @ v, bf16)                     # mc:405


def _num_heads(sd: SD, prefix: str) -> int:
    n = 0
    while f"{prefix}heads.{n}.key.weight" in sd:
        n += 1
    return n


def mha_forward(sd: SD, prefix: str, x: Tensor, p: float = 0.0, training: bool = False,
                masks: Optional[dict] = None, bf16: bool = False) -> Tensor:
    """MultiHeadAttention / 2 / 3. ref: src/model_component.py:86-103 (cat only),
    :241-261 (+proj), :436-455 (+proj +dropout)."""
    nh = _num_heads(sd, prefix)
    outs = []
    for h in range(nh):
        m = None if masks is None else masks.get(f"{prefix}heads.{h}")
        outs.append(head_forward(sd, f"{prefix}heads.{h}.", x, p, training, m, bf16))
    out = torch.cat(outs, dim=-1)                          # mc:453
    if prefix + "proj.weight" in sd:
        out = _rg(F.linear(out, _rb(sd[prefix + "proj.weight"], bf16), sd[prefix + "proj.bias"]), bf16)   # mc:454
        if p > 0.0 or masks is not None:
            m = None if masks is None else masks.get(f"{prefix}proj")
            out = _drop(out, p, training, m)
    return out


def ffn_forward(sd: SD, prefix: str, x: Tensor, p: float = 0.0, training: bool = False,
                mask: Optional[Tensor] = None, bf16: bool = False) -> Tensor:
    """FeedForward (Linear(C,C)+ReLU, mc:118-121), FeedForward2 (C->4C->C, mc:197-201),
    FeedForward3 (+Dropout, mc:320-325)."""
    h = F.relu(F.linear(x, _rb(sd[prefix + "net.0.weight"], bf16), sd[prefix + "net.0.bias"]))
    if prefix + "net.2.weight" in sd:
        h = _rg(F.linear(_rb(h, bf16), _rb(sd[prefix + "net.2.weight"], bf16), sd[prefix + "net.2.bias"]), bf16)
        if p > 0.0 or mask is not None:
            h = _drop(h, p, training, mask)
    return h


def block_forward(sd: SD, prefix: str, kind: str, x: Tensor, p: float = 0.0,
                  training: bool = False, masks: Optional[dict] = None, bf16: bool = False, stream_bf16: bool = False) -> Tensor:
    """Block (mc:179-181), ResidualBlock (mc:303-305), ResidualBlock2 (mc:505-507)."""
    if kind == "Block":
        return ffn_forward(sd, prefix + "ffwd.", mha_forward(sd, prefix + "sa_head.", x))
    if kind == "ResidualBlock":
        x = x + mha_forward(sd, prefix + "sa_head.", x)
        return x + ffn_forward(sd, prefix + "ffwd.", x)
    if kind == "ResidualBlock2":
        C = x.shape[-1]
        # stream_bf16 (rounding model of the engine's bf16 / fp8 modes): the gradient that arrives at the residual stream after
        # each sub-layer is stored in bf16 (the forward stream itself stays fp32)
        x = _rg(x, stream_bf16)
        h = _rb(F.layer_norm(x, (C,), sd[prefix + "ln1.weight"], sd[prefix + "ln1.bias"], 1e-5), bf16)
        x = _rg(x + mha_forward(sd, prefix + "sa_head.", h, p, training, masks, bf16), stream_bf16)
        h = _rb(F.layer_norm(x, (C,), sd[prefix + "ln2.weight"], sd[prefix + "ln2.bias"], 1e-5), bf16)
        m = None if masks is None else masks.get(f"{prefix}ffwd")
        return x + ffn_forward(sd, prefix + "ffwd.", h, p, training, m, bf16)
    raise ValueError(kind)


_BLOCK_KIND = {"BlocksLM": "Block", "ResidualBlocksLM": "ResidualBlock", "TransformerLM": "ResidualBlock2"}


def _num_layers(sd: SD) -> int:
    n = 0
    while any(k.startswith(f"blocks.{n}.") for k in sd):
        n += 1
    return n


# --------------------------------------------------------------------------------------
# the six LMs
# --------------------------------------------------------------------------------------
def lm_forward(model_name: str, sd: SD, idx: Tensor, targets: Optional[Tensor] = None,
               p: float = 0.0, training: bool = False, masks: Optional[dict] = None, bf16: bool = False,
               stream_bf16: bool = False) -> Tuple[Tensor, Optional[Tensor]]:
    """forward(idx, targets) of any of the six LMs.

    ``bf16=True`` (TransformerLM only) is NOT the reference's arithmetic: it inserts a bf16 rounding wherever the HIP
    path's precision="bf16" mode stores an operand as bf16 (see _rb / _rg), so that the bf16 kernels can be held to a
    tight per-tensor bound against "the reference computed with the same roundings" instead of a loose one against fp32.

    ref: BigramLM src/model.py:80-105; SingleHeadAttentionLM :176-203;
    MultiHeadAttentionLM :280-307; BlocksLM :381-408; ResidualBlocksLM :482-509;
    TransformerLM :578-609 (ln_f exists at :572 but is never applied, :598-599).
    """
    if model_name == "BigramLM":
        logits = F.embedding(idx, sd["token_embedding_table.weight"])
    else:
        T = idx.shape[1]
        tok = F.embedding(idx, sd["token_embedding_table.weight"])                    # model.py:595
        pos = F.embedding(torch.arange(T), sd["position_embedding_table.weight"])     # :596
        x = tok + pos                                                                  # :597
        if model_name == "SingleHeadAttentionLM":
            x = head_forward(sd, "sa_head.", x)
        elif model_name == "MultiHeadAttentionLM":
            x = mha_forward(sd, "sa_head.", x)
        else:
            kind = _BLOCK_KIND[model_name]
            for l in range(_num_layers(sd)):
                x = block_forward(sd, f"blocks.{l}.", kind, x, p, training, masks, bf16, stream_bf16)
        logits = F.linear(_rb(_rg(x, stream_bf16), bf16), _rb(sd["lm_head.weight"], bf16), sd["lm_head.bias"])   # :599
    if targets is None:
        return logits, None
    B, T, V = logits.shape
    logits = logits.view(B * T, V)                                                     # :605
    loss = F.cross_entropy(_rg(logits, bf16), targets.view(B * T))                     # :606-607
    return logits, loss


def context_length_of(model_name: str, sd: SD) -> Optional[int]:
    if model_name == "BigramLM":
        return None
    return sd["position_embedding_table.weight"].shape[0]


def lm_generate(model_name: str, sd: SD, idx: Tensor, max_new_tokens: int,
                generator: Optional[torch.Generator] = None) -> Tensor:
    """ref: src/model.py:611-636 (and :107-130 for BigramLM, which never crops).
    Sampling uses the CPU generator (global one when ``generator`` is None)."""
    ctx = context_length_of(model_name, sd)
    for _ in range(max_new_tokens):
        cond = idx if ctx is None else idx[:, -ctx:]
        logits, _ = lm_forward(model_name, sd, cond)
        probs = F.softmax(logits[:, -1, :], dim=-1)
        nxt = torch.multinomial(probs, num_samples=1, generator=generator)
        idx = torch.cat((idx, nxt), dim=1)
    return idx


# --------------------------------------------------------------------------------------
# parameter bookkeeping
# --------------------------------------------------------------------------------------
def param_keys(sd: SD) -> List[str]:
    """state_dict keys that are parameters (``tril`` entries are buffers)."""
    return [k for k in sd if not k.endswith(".tril")]


def trainable_keys(model_name: str, sd: SD) -> List[str]:
    """Parameters that receive a gradient. ``ln_f.*`` never does."""
    return [k for k in param_keys(sd) if not k.startswith("ln_f.")]


def init_state_dict(model_name: str, vocab_size: int, cfg: dict, seed: int = 42) -> SD:
    """A reference-layout state_dict with torch's default initialisers, created in the
    reference's module construction order so that ``torch.manual_seed(seed)`` gives the
    same numbers as ``build_model`` would (ref: src/train.py:58; src/model.py:558-576;
    src/model_component.py:365-376,428-433,318-325,477-489)."""
    import torch.nn as nn

    torch.manual_seed(seed)
    C, T = cfg["embedding_dim"], cfg["context_length"]
    sd: SD = {}

    def emb(name, n, d):
        sd[name + ".weight"] = nn.Embedding(n, d).weight.detach().clone()

    def lin(name, i, o, bias=True):
        m = nn.Linear(i, o, bias=bias)
        sd[name + ".weight"] = m.weight.detach().clone()
        if bias:
            sd[name + ".bias"] = m.bias.detach().clone()

    def head(prefix, hs):
        sd[prefix + "tril"] = torch.tril(torch.ones(T, T))   # own buffer precedes child params in state_dict order
        lin(prefix + "key", C, hs, False)
        lin(prefix + "query", C, hs, False)
        lin(prefix + "value", C, hs, False)

    def ln(name):
        sd[name + ".weight"] = torch.ones(C)
        sd[name + ".bias"] = torch.zeros(C)

    if model_name == "BigramLM":
        emb("token_embedding_table", vocab_size, vocab_size)
        return sd
    emb("token_embedding_table", vocab_size, C)
    emb("position_embedding_table", T, C)
    if model_name == "SingleHeadAttentionLM":
        head("sa_head.", cfg["head_size"])
    elif model_name == "MultiHeadAttentionLM":
        for h in range(cfg["num_heads"]):
            head(f"sa_head.heads.{h}.", cfg["head_size"] // cfg["num_heads"])   # model.py:264
    else:
        nh = cfg["num_heads"]
        for l in range(cfg["num_layers"]):
            for h in range(nh):
                head(f"blocks.{l}.sa_head.heads.{h}.", C // nh)
            if model_name != "BlocksLM":
                lin(f"blocks.{l}.sa_head.proj", C, C)
            if model_name == "BlocksLM":
                lin(f"blocks.{l}.ffwd.net.0", C, C)
            else:
                lin(f"blocks.{l}.ffwd.net.0", C, 4 * C)
                lin(f"blocks.{l}.ffwd.net.2", 4 * C, C)
            if model_name == "TransformerLM":
                ln(f"blocks.{l}.ln1")
                ln(f"blocks.{l}.ln2")
        if model_name == "TransformerLM":
            ln("ln_f")
    lin("lm_head", C, vocab_size)
    return sd


# --------------------------------------------------------------------------------------
# data + training step
# --------------------------------------------------------------------------------------
def get_batch(data: Tensor, context_length: int, batch_size: int,
              generator: Optional[torch.Generator] = None) -> Tuple[Tensor, Tensor]:
    """ref: src/preprocessing.py:28-46 (offsets from the CPU generator, y = x shifted by 1)."""
    ix = torch.randint(len(data) - context_length, (batch_size,), generator=generator)
    x = torch.stack([data[i:i + context_length] for i in ix])
    y = torch.stack([data[i + 1:i + context_length + 1] for i in ix])
    return x, y


def encode_corpus(text: str) -> Tuple[Tensor, Tensor, int, List[str]]:
    """char tokenizer + 90/10 split: (train_data, val_data, vocab_size, vocab) as int64 streams -- the tensors
    get_train_val_data saves to train_data.pt / val_data.pt (ref: src/preprocessing.py:3-26,68-79)."""
    vocab = sorted(list(set(text)))                         # pp:14
    stoi = {ch: i for i, ch in enumerate(vocab)}            # pp:19
    data = torch.tensor([stoi[c] for c in text], dtype=torch.long)    # pp:20,69
    n = len(data)
    return data[:int(0.9 * n)], data[int(0.9 * n):], len(vocab), vocab   # pp:76-79


def cyclic_lr(step_count: int, base_lr: float, max_lr: float, step_size_up: int = 5) -> float:
    """CyclicLR(mode='triangular', step_size_up=5, cycle_momentum=False) after
    ``step_count`` calls of scheduler.step(). ref: src/train.py:122-126,162."""
    total = 2.0 * step_size_up
    cycle = math.floor(1 + step_count / total)
    x = 1.0 + step_count / total - cycle
    ratio = step_size_up / total
    scale = x / ratio if x <= ratio else (x - 1) / (ratio - 1)
    return base_lr + (max_lr - base_lr) * scale


class AdamWState:
    """torch.optim.AdamW(params, lr, betas) with its defaults eps=1e-8, weight_decay=1e-2,
    amsgrad=False, restated per tensor (ref: src/train.py:121).  Parameters whose grad is
    None are skipped entirely -- no decay, no step count (ln_f)."""

    def __init__(self, keys: Sequence[str], lr: float, betas=(0.9, 0.95), eps=1e-8, weight_decay=1e-2):
        self.lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.m: SD = {}
        self.v: SD = {}
        self.t: Dict[str, int] = {k: 0 for k in keys}

    def step(self, params: SD, grads: Dict[str, Optional[Tensor]]) -> None:
        b1, b2 = self.betas
        for k, g in grads.items():
            if g is None:
                continue
            p = params[k]
            if k not in self.m:
                self.m[k] = torch.zeros_like(p)
                self.v[k] = torch.zeros_like(p)
            self.t[k] += 1
            t = self.t[k]
            p.mul_(1 - self.lr * self.wd)
            self.m[k].lerp_(g, 1 - b1)
            self.v[k].mul_(b2).addcmul_(g, g, value=1 - b2)
            bc1 = 1 - b1 ** t
            bc2 = 1 - b2 ** t
            step_size = self.lr / bc1
            denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
            p.addcdiv_(self.m[k], denom, value=-step_size)


def loss_and_grads(model_name: str, sd: SD, idx: Tensor, targets: Tensor, p: float = 0.0,
                   training: bool = False, masks: Optional[dict] = None, bf16: bool = False, stream_bf16: bool = False):
    """logits, loss and d(loss)/d(param) for every trainable key (autograd over the
    restatement; ref: loss.backward() at src/train.py:150)."""
    keys = trainable_keys(model_name, sd)
    work = dict(sd)
    leaves = []
    for k in keys:
        t = sd[k].detach().clone().requires_grad_(True)
        work[k] = t
        leaves.append(t)
    logits, loss = lm_forward(model_name, work, idx, targets, p, training, masks, bf16, stream_bf16)
    gs = torch.autograd.grad(loss, leaves, allow_unused=True)
    grads = {k: g for k, g in zip(keys, gs)}
    return logits.detach(), loss.detach(), grads


def train_step(model_name: str, sd: SD, opt: AdamWState, idx: Tensor, targets: Tensor,
               p: float = 0.0, training: bool = True, masks: Optional[dict] = None) -> float:
    """One iteration of the reference loop body (ref: src/train.py:146-151)."""
    _, loss, grads = loss_and_grads(model_name, sd, idx, targets, p, training, masks)
    opt.step(sd, grads)
    return float(loss)
