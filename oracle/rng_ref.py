"""numpy restatement of the HIP kernels' counter-based dropout stream -- TEST INFRASTRUCTURE.

The reference draws dropout masks with ``aten::bernoulli_`` from torch's CPU generator
(ref: src/model_component.py:401,454,324); that stream cannot be reproduced on a GPU, so the
kernels use a stateless hash of (seed, step, site, element index) instead
(drakegpt_amd/csrc/common.h: dg_keep).  This file recomputes exactly that hash on the host so
the tests can hand the SAME keep-masks to the CPU oracle (``drake_ref`` functions take explicit
masks) and compare dropout'd forward/backward results with the kernels' exactly.
"""
from __future__ import annotations

import numpy as np

_M32 = np.uint64(0xFFFFFFFF)


def _mix32(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint64) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x


def site_key(seed: int, step: int, site: int) -> int:
    """dg_site_key in common.h."""
    lo, hi = seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF
    a = int(_mix32(np.array([(step * 0x9E3779B9 + site) & 0xFFFFFFFF]))[0])
    b = int(_mix32(np.array([hi ^ a]))[0])
    return int(_mix32(np.array([lo ^ b]))[0])


def threshold(p: float) -> int:
    """drop iff the element's 16-bit field < thr16; thr16 = floor(p * 2^16) clamped (dg_drop_threshold)."""
    return min(int(p * 65536.0), 0xFFFF)


def element_hash(key: int, idx: np.ndarray) -> np.ndarray:
    """dg_hash_w in common.h on the PAIR index idx: Weyl step, one xorshift32 round, one 24 x 24-bit multiply (low 32 bits)."""
    x = np.uint64(key) ^ ((idx.astype(np.uint64) * np.uint64(0x9E3779B1)) & _M32)
    x ^= x >> np.uint64(17)
    x ^= (x << np.uint64(11)) & _M32
    x ^= x >> np.uint64(13)
    return ((x & np.uint64(0xFFFFFF)) * np.uint64(0xEB352D)) & _M32


def keep_mask(seed: int, step: int, site: int, p: float, n: int) -> np.ndarray:
    """keep[i] for linear element indices i in [0, n) (n < 2^32) as float32 {0,1}: elements 2j and 2j + 1 share
    hash(key, j); the even one takes its low 16 bits, the odd one its high 16 bits (dg_keep in common.h)."""
    idx = np.arange(n, dtype=np.uint64)
    r = element_hash(site_key(seed, step, site), idx >> np.uint64(1))
    field = np.where((idx & np.uint64(1)) == 1, r >> np.uint64(16), r & np.uint64(0xFFFF))
    return (field >= np.uint64(threshold(p))).astype(np.float32)


# site numbering shared with drakegpt_amd/functional.py
def site_attn(layer: int) -> int:
    return 4 * layer + 0


def site_proj(layer: int) -> int:
    return 4 * layer + 1


def site_ffn(layer: int) -> int:
    return 4 * layer + 2


def transformer_masks(seed: int, step: int, p: float, B: int, T: int, C: int, NH: int, L: int):
    """Explicit keep-masks for ``drake_ref.lm_forward('TransformerLM', ..., masks=...)``.

    Attention element index = ((b*NH + h)*T + i)*T + j; proj / ffn index = row*C + col."""
    import torch

    masks = {}
    for l in range(L):
        a = keep_mask(seed, step, site_attn(l), p, B * NH * T * T).reshape(B, NH, T, T)
        for h in range(NH):
            masks[f"blocks.{l}.sa_head.heads.{h}"] = torch.from_numpy(np.ascontiguousarray(a[:, h]))
        masks[f"blocks.{l}.sa_head.proj"] = torch.from_numpy(
            keep_mask(seed, step, site_proj(l), p, B * T * C).reshape(B, T, C))
        masks[f"blocks.{l}.ffwd"] = torch.from_numpy(
            keep_mask(seed, step, site_ffn(l), p, B * T * C).reshape(B, T, C))
    return masks
