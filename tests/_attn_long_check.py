"""Attention forward + backward at sequence lengths beyond 256 (the GPT-2 shapes' 32-tile key chains, the balanced block
order, the XCD ranges, the P | dS tile workspace) against an fp64 restatement of Head2.forward (ref:
src/model_component.py:392-405) with the kernels' own keep-masks.  A module of its own so that the same check can run in a
child process under DG_ATTN_TILES=0 (the dK/dV pass that recomputes scores instead of reading the dQ pass's tiles): the
library reads its A/B switches once per process.

    python tests/_attn_long_check.py        # runs every case on cuda:0, prints one line per case, exits non-zero on failure
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# (B, T, NH, H, p): 32 key tiles with dropout; 20 tiles, one head, no dropout; a ragged last tile (T % 32 != 0) with dropout
# round 3: 448 (batch, head) pairs x 2 workgroups = 896 workgroups, more than one residency (768 on 256 CUs): the heavy-first
# block order (attn_item, balance == 2) instead of the equal-cost one
CASES = [(1, 1024, 2, 64, 0.1), (2, 640, 1, 64, 0.0), (1, 1000, 1, 64, 0.1), (14, 256, 32, 64, 0.1)]
# measured on MI355X (round 3), both dK/dV forms alike: bf16 forward 1.9e-3 .. 2.3e-3, backward 2.4e-3 .. 2.5e-3 (dQ, dK and dV
# each 2.3e-3 .. 2.6e-3), lse 9e-7 absolute -- bf16 operand rounding, the same level as the T <= 256 cases of
# tests/test_gpu_ops.py::test_attention (whose bounds these are): nothing grows with the chain length
TOL_FWD, TOL_BWD = 8e-3, 2e-2


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def attn_ref(qkv, B, T, NH, H, keep=None, p=0.0):
    """fp64, differentiable: softmax(mask(q k^T * H^-1/2)) [* keep / (1 - p)] v over packed qkv [B*T, 3*NH*H]"""
    C = NH * H
    q, k, v = qkv.view(B, T, 3, NH, H).permute(2, 0, 3, 1, 4)
    w = q @ k.transpose(-2, -1) * H ** -0.5
    tril = torch.tril(torch.ones(T, T, dtype=torch.bool))
    w = w.masked_fill(~tril, float("-inf")).softmax(-1)
    if keep is not None:
        w = w * keep / (1 - p)
    return (w @ v).permute(0, 2, 1, 3).reshape(B * T, C)


def check_case(dev, B, T, NH, H, p):
    from drakegpt_amd import ops
    from oracle import rng_ref
    g = torch.Generator().manual_seed(T * 7 + H)
    C = NH * H
    qkv = torch.randn(B * T, 3 * C, generator=g).bfloat16()
    dout = torch.randn(B * T, C, generator=g).bfloat16()
    seed, step, site = 77, 3, 4
    keep = rng = None
    if p > 0:
        keep = torch.from_numpy(rng_ref.keep_mask(seed, step, site, p, B * NH * T * T).reshape(B, NH, T, T)).double()
        rng = ops.new_rng_state(seed, dev, step)
    qd = qkv.double().requires_grad_(True)
    ref = attn_ref(qd, B, T, NH, H, keep, p)
    ref.backward(dout.double())
    out, lse = ops.attn_fwd(qkv.to(dev), B, T, NH, H, H ** -0.5, p, rng, site, keep=os.environ.get("DG_ATTN_KEEPBITS", "1") != "0")
    dqkv = ops.attn_bwd(qkv.to(dev), out, dout.to(dev), lse, B, T, NH, H, H ** -0.5, p, rng, site)
    torch.cuda.synchronize()
    ef, eb = rel(out, ref.detach()), rel(dqkv, qd.grad)
    # the three gradient blocks separately: a wrong dK or dV term must not hide behind a correct dQ
    parts = {n: rel(dqkv.view(B * T, 3, C)[:, i], qd.grad.view(B * T, 3, C)[:, i]) for i, n in enumerate(("dq", "dk", "dv"))}
    lse_ref = torch.logsumexp((qd.detach().view(B, T, 3, NH, H)[:, :, 0].permute(0, 2, 1, 3) @
                               qd.detach().view(B, T, 3, NH, H)[:, :, 1].permute(0, 2, 3, 1) * H ** -0.5)
                              .masked_fill(~torch.tril(torch.ones(T, T, dtype=torch.bool)), float("-inf")), -1)
    el = (lse.double().cpu() - lse_ref).abs().max().item()
    return ef, eb, parts, el


def run_all(dev, report=False):
    bad = []
    for case in CASES:
        ef, eb, parts, el = check_case(dev, *case)
        if report:
            print(f"[parity] attention {case} tiles={os.environ.get('DG_ATTN_TILES', '1')}: fwd {ef:.2e} bwd {eb:.2e} "
                  + " ".join(f"{k} {v:.2e}" for k, v in parts.items()) + f" lse {el:.2e}", flush=True)
        if not (ef < TOL_FWD and eb < TOL_BWD and max(parts.values()) < TOL_BWD and el < 2e-2):
            bad.append((case, ef, eb, parts, el))
    return bad


if __name__ == "__main__":
    bad = run_all(torch.device("cuda:0"), report=True)
    if bad:
        print("FAILED", bad)
        sys.exit(1)
    print("ok")
