"""Phase timing inside dg_block_chain_bwd (mode 0): s_memtime stamps of MFMA wave 0 of every workgroup at the phase boundaries.

    python tools/chain_bwd_stamps.py [--p 0.2]

s_memtime counts at a constant 100 MHz on this part (10 ns units); medians over the workgroups of the last of several launches.
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
    args = ap.parse_args()
    from drakegpt_amd import ops, _lib
    from drakegpt_amd import sublayers as S
    dev = torch.device("cuda:0")
    M, C, p = 16384, 384, args.p
    gen = torch.Generator().manual_seed(2)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc)
    bf, f32 = torch.bfloat16, torch.float32
    WT = dict(wqkvT=rn(C, 3 * C, sc=(3 * C) ** -0.5), w2T=rn(4 * C, C, sc=C ** -0.5), w1T=rn(C, 4 * C, sc=(4 * C) ** -0.5), wprojT=rn(C, C, sc=C ** -0.5))
    WTp = {k: ops.pack_chain_weights(v.to(bf).to(dev)) for k, v in WT.items()}
    dqkv = rn(M, 3 * C).to(bf).to(dev)
    x, x1 = rn(M, C, sc=2.0).to(dev), rn(M, C, sc=2.0).to(dev)
    ln1w, ln2w = (1 + rn(C, sc=0.1)).to(dev), (1 + rn(C, sc=0.1)).to(dev)
    _, mean1, rstd1 = ops.layernorm_fwd(x, ln1w, torch.zeros_like(ln1w), bf)
    h2, mean2, rstd2 = ops.layernorm_fwd(x1, ln2w, torch.zeros_like(ln2w), bf)
    dresid = rn(M, C).to(bf).to(dev)
    bits = ops.new_sign_bits(M, 4 * C, dev)
    ops.gemm_nt(h2, rn(4 * C, C, sc=C ** -0.5).to(bf).to(dev), bf, relu=True, sign_bits_out=bits)
    rng = ops.new_rng_state(4321, dev, 5) if p > 0 else None
    stride = 12 * C
    P = torch.zeros((2 * (M // 64), stride), dtype=f32, device=dev)
    kw = dict(part_stride=stride, dropout_p=p, rng_state=rng, site_ffn_below=S.site_ffn(2), site_proj=S.site_proj(2),
              dqkv=dqkv, wqkvT=WTp["wqkvT"], x=x, mean1=mean1, rstd1=rstd1, ln1w=ln1w, dresid1=dresid, dln1w_part=P[0, 0:C], dln1b_part=P[0, C:2 * C],
              gbias1_part=P[0, 2 * C:3 * C], w2T=WTp["w2T"], bits=bits, db1_part=P[0, 8 * C:12 * C], w1T=WTp["w1T"], x1=x1, mean2=mean2, rstd2=rstd2,
              ln2w=ln2w, dln2w_part=P[0, 3 * C:4 * C], dln2b_part=P[0, 4 * C:5 * C], gbias2_part=P[0, 5 * C:6 * C], wprojT=WTp["wprojT"])
    nb = M // 64
    buf = torch.zeros((nb, 16), dtype=torch.int64, device=dev)
    fn = _lib.lib.dg_debug_set_chain_bwd_stamps
    fn.argtypes = [ctypes.c_void_p]
    fn.restype = None
    for _ in range(3):
        ops.block_chain_bwd(0, M, C, **kw)
    torch.cuda.synchronize()
    fn(buf.data_ptr())
    ops.block_chain_bwd(0, M, C, **kw)
    torch.cuda.synchronize()
    fn(None)
    t = buf.cpu().double()
    t0 = t[:, 0].min()
    names = ["start", "dX-QKV K loop", "LN1 backward", "FFN2 c0 K", "FFN2 c0 epi", "FFN2 c1 K", "FFN2 c1 epi", "FFN2 c2 K", "FFN2 c2 epi",
             "FFN2 c3 K", "FFN2 c3 epi", "dX-FFN1 K loop", "LN2 backward", "dX-proj K loop", "dX-proj epi", "END barrier"]
    unit = 0.01     # us per tick (100 MHz)
    print(f"first workgroup starts at 0; start spread {(t[:, 0].max() - t0) * unit:.1f} us; last END {(t[:, 15].max() - t0) * unit:.1f} us")
    tot_k = tot_e = 0.0
    for k in range(1, 16):
        d = (t[:, k] - t[:, k - 1]) * unit
        med, lo, hi = d.median().item(), d.min().item(), d.max().item()
        print(f"{names[k]:16s} median {med:6.2f} us   min {lo:6.2f}  max {hi:6.2f}")
        if "K" in names[k]:
            tot_k += med
        else:
            tot_e += med
    print(f"K loops {tot_k:.1f} us, epilogues {tot_e:.1f} us (medians)")


if __name__ == "__main__":
    main()
