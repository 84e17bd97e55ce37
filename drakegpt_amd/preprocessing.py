"""Data side of the hot path (mirror of the reference's src/preprocessing.py).

get_batch keeps the reference's contract -- B window offsets from torch's CPU generator, x and
y = x shifted by one, int64 (src/preprocessing.py:28-46) -- but when the corpus already lives in
HBM the 2*B Python slices + 2 stacks + 2 H2D copies become one offsets copy and one gather kernel.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import ops


def get_mapper(text: str):
    """char-level tokenizer (ref: src/preprocessing.py:3-26): vocab = sorted(set(text))."""
    vocab = sorted(set(text))
    stoi = {ch: i for i, ch in enumerate(vocab)}
    itos = dict(enumerate(vocab))

    def encode(s):
        return [stoi[c] for c in s]

    def decode(ids):
        return "".join(itos[i] for i in ids)

    return encode, decode, len(vocab)


def draw_offsets(n_data: int, context_length: int, batch_size: int, generator: Optional[torch.Generator] = None) -> torch.Tensor:
    """the reference's ``torch.randint(len(data) - context_length, (batch_size,))`` on the CPU generator"""
    return torch.randint(n_data - context_length, (batch_size,), generator=generator)


def get_batch(data: torch.Tensor, context_length: int, batch_size: int, device, generator: Optional[torch.Generator] = None):
    """ref: src/preprocessing.py:28-46.  `data` may be a CPU tensor (moved once per call, like the
    reference) or, preferably, already resident on `device`."""
    device = torch.device(device)
    if device.type != "cuda":
        raise RuntimeError("drakegpt_amd.get_batch gathers on the GPU; there is no CPU path in this package")
    ix = draw_offsets(len(data), context_length, batch_size, generator)
    if not data.is_cuda:
        data = data.to(device)
    return ops.batch_gather(data.contiguous(), ix.to(device, non_blocking=True), context_length)


def split_train_val(data: torch.Tensor, frac: float = 0.9):
    """90/10 split (ref: src/preprocessing.py:76-79)."""
    n = len(data)
    return data[: int(frac * n)], data[int(frac * n):]


def encode_text(text: str):
    """text -> (int64 tensor, decode, vocab_size), the tensor format of train_data.pt / val_data.pt
    (ref: src/preprocessing.py:68-73)."""
    encode, decode, vocab_size = get_mapper(text)
    return torch.tensor(encode(text), dtype=torch.long), decode, vocab_size
