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


def get_train_val_data(input_path: str, train_path: str, val_path: str, verbose: bool = True):
    """ref: src/preprocessing.py:48-86 -- read the text, build the char mapper, encode to int64, split 90/10 and save the two
    token streams with ``torch.save`` as bare tensors: the on-disk ``train_data.pt`` / ``val_data.pt`` format that
    src/train.py:94-98 loads (and ``load_train_val_data`` below reads back with ``weights_only=True``).  Also prints the
    reference's example batch (4 windows of 8 tokens, seed 42) -- drawn by plain host indexing: this is file preparation, it
    needs no GPU.  Returns (train_data, val_data, vocab_size)."""
    torch.manual_seed(42)
    with open(input_path, "r", encoding="utf-8") as f:
        text = f.read()
    encode, decode, vocab_size = get_mapper(text)
    data = torch.tensor(encode(text), dtype=torch.long)
    train_data, val_data = split_train_val(data)
    if verbose:
        print(f"Vocab size of the text: {vocab_size}")
        if len(train_data) > 9:
            ix = draw_offsets(len(train_data), 8, 4)
            x0, y0 = train_data[ix[0]:ix[0] + 8].tolist(), train_data[ix[0] + 1:ix[0] + 9].tolist()
            print(f"Input (encoded):\n{x0}\nInput (decoded):\n{decode(x0)}\nOutput (encoded):\n{y0}\nOutput (decoded):\n{decode(y0)}\n")
    torch.save(train_data.clone(), train_path)        # clone: a view would drag the whole corpus into each file
    torch.save(val_data.clone(), val_path)
    return train_data, val_data, vocab_size


def load_train_val_data(train_path: str, val_path: str):
    """the two token streams get_train_val_data (or the reference's src/preprocessing.py) wrote -- ref: src/train.py:97-98.
    Loaded with ``weights_only=True`` (nothing in the file is executed); must be 1-D int64."""
    out = []
    for path in (train_path, val_path):
        t = torch.load(path, map_location="cpu", weights_only=True)
        if not isinstance(t, torch.Tensor) or t.dim() != 1 or t.dtype != torch.int64:
            raise ValueError(f"{path}: expected a 1-D int64 token tensor (train_data.pt / val_data.pt format), got "
                             f"{type(t).__name__} {getattr(t, 'dtype', '')} {tuple(getattr(t, 'shape', ()))}")
        out.append(t.contiguous())
    return out[0], out[1]
