"""Phase timing inside dg_block_chain_fwd (mode 0): s_memtime stamps of MFMA wave 0 of every workgroup at the phase boundaries
(s_memtime counts shader clocks; medians over the 256 workgroups of one launch behind three warm-up launches).

    python tools/chain_fwd_stamps.py [--p 0.2]
"""
import argparse
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--p", type=float, default=0.2)
    a = ap.parse_args()
    from drakegpt_amd import ops, _lib
    from drakegpt_amd import sublayers as S
    dev = torch.device("cuda:0")
    M, C, p = 16384, 384, a.p
    g = torch.Generator().manual_seed(1)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    o = rn(M, C).bfloat16().to(dev)
    x = rn(M, C, sc=2.0).to(dev)
    W = dict(wproj=rn(C, C, sc=C ** -0.5), w1=rn(4 * C, C, sc=C ** -0.5), w2=rn(C, 4 * C, sc=(4 * C) ** -0.5), wqkv=rn(3 * C, C, sc=C ** -0.5))
    Wp = {k: ops.pack_chain_weights(v.bfloat16().to(dev)) for k, v in W.items()}
    V = dict(bproj=rn(C, sc=0.1), b1=rn(4 * C, sc=0.1), b2=rn(C, sc=0.1), ln2w=1 + rn(C, sc=0.1), ln2b=rn(C, sc=0.1), ln1w=1 + rn(C, sc=0.1), ln1b=rn(C, sc=0.1))
    V = {k: v.to(dev) for k, v in V.items()}
    rng = ops.new_rng_state(1234, dev, 7) if p > 0 else None
    call = lambda: ops.block_chain_fwd(0, M, C, o=o, x=x, dropout_p=p, rng_state=rng, site_proj=S.site_proj(3), site_ffn=S.site_ffn(3), **Wp, **V)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    buf = torch.zeros((M // 64, 24), dtype=torch.int64, device=dev)
    fn = _lib.lib.dg_debug_set_chain_fwd_stamps
    fn.argtypes = [ctypes.c_void_p]
    fn.restype = None
    fn(buf.data_ptr())
    call()
    torch.cuda.synchronize()
    fn(None)
    t = buf.cpu().double()
    names = {1: "proj K loop (12 steps)", 2: "proj epilogue: residual + LayerNorm 2", 3: "FFN1 c0 K", 4: "FFN1 c0 epi", 5: "FFN1 c1 K", 6: "FFN1 c1 epi",
             7: "FFN1 c2 K", 8: "FFN1 c2 epi", 9: "FFN1 c3 K", 10: "FFN1 c3 epi", 11: "FFN2 K loop (48 steps, f through the ring)",
             12: "FFN2 epilogue: residual + LayerNorm 1'", 13: "QKV c0 K", 14: "QKV c0 epi", 15: "QKV c1 K", 16: "QKV c1 epi", 17: "QKV c2 K", 18: "QKV c2 epi",
             19: "-", 20: "END barrier"}
    tot = (t[:, 20] - t[:, 0]).median().item()
    print(f"cycles per phase (median over {t.shape[0]} workgroups); whole block {tot:.0f} cycles")
    k_sum = e_sum = 0.0
    for k in range(1, 21):
        d = t[:, k] - t[:, k - 1]
        med = d.median().item()
        print(f"  {names[k]:44s} {med:8.0f}   ({100 * med / tot:4.1f} %)   min {d.min().item():7.0f} max {d.max().item():7.0f}")
        if " K" in names[k]:
            k_sum += med
        elif k < 19:
            e_sum += med
    print(f"K loops {k_sum:.0f} cycles ({100 * k_sum / tot:.0f} %), epilogues {e_sum:.0f} ({100 * e_sum / tot:.0f} %)")


if __name__ == "__main__":
    main()
