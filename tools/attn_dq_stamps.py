"""Where a wave of the attention dQ pass spends its time: s_memtime deltas of wave 0 of every workgroup, summed per phase over its
tiles (tile staging incl. the wait for the prefetched K / V; S and dP MFMAs; score arithmetic + LDS staging of P | dS; tile store;
dQ MFMAs).    python tools/attn_dq_stamps.py [--B 64 --T 256 --NH 6 --p 0.2]"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--T", type=int, default=256)
    ap.add_argument("--NH", type=int, default=6)
    ap.add_argument("--p", type=float, default=0.2)
    a = ap.parse_args()
    from drakegpt_amd import ops, _lib
    dev = torch.device("cuda:0")
    B, T, NH, H = a.B, a.T, a.NH, 64
    C = NH * H
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B * T, 3 * C, generator=g).bfloat16().to(dev)
    do = torch.randn(B * T, C, generator=g).bfloat16().to(dev)
    rng = ops.new_rng_state(7, dev, 3) if a.p > 0 else None
    o, lse = ops.attn_fwd(qkv, B, T, NH, H, H ** -0.5, a.p, rng, 5, keep=True)
    for _ in range(3):
        ops.attn_bwd(qkv, o, do, lse, B, T, NH, H, H ** -0.5, a.p, rng, 5)
    torch.cuda.synchronize()
    n_wg = B * NH * ((T + 31) // 32) // 4
    buf = torch.zeros((n_wg, 8), dtype=torch.int64, device=dev)
    fn = _lib.lib.dg_debug_set_attn_stamps
    fn.argtypes = [ctypes.c_void_p]
    fn.restype = None
    fn(buf.data_ptr())
    ops.attn_bwd(qkv, o, do, lse, B, T, NH, H, H ** -0.5, a.p, rng, 5)
    torch.cuda.synchronize()
    fn(None)
    t = buf.cpu().double()
    tiles = t[:, 5]
    ok = tiles > 0
    names = ["stage K/V into LDS (incl. wait for the prefetch)", "S, dP: 8 MFMAs + fragment reads", "score arithmetic + P|dS staging writes",
             "tile store (LDS read-back + global store)", "dQ: 4 MFMAs + transposed reads"]
    tot = 0.0
    print(f"B={B} T={T} NH={NH} p={a.p}: {int(ok.sum())} workgroups, wave 0 walks {tiles[ok].mean():.1f} tiles on average; cycles per tile (s_memtime = shader clocks):")
    for k, nm in enumerate(names):
        per = (t[ok, k] / tiles[ok])
        print(f"  {nm:52s} mean {per.mean():8.0f}   min {per.min():8.0f}   max {per.max():8.0f}")
        tot += per.mean().item()
    print(f"  sum {tot:.0f} cycles per tile")


if __name__ == "__main__":
    main()
