"""dg_block_chain_fwd against the launches it replaces (same kernels' arithmetic: NT GEMM epilogues + dg_layernorm_fwd), and both
timed on one box.

    python tools/chain_check.py [--M 16384] [--p 0.2] [--mode 0|1|2] [--reps 20] [--no-time]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def rel(a, b):
    a, b = a.double(), b.double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=16384)
    ap.add_argument("--p", type=float, default=0.2)
    ap.add_argument("--mode", type=int, default=0)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--no-time", action="store_true")
    args = ap.parse_args()
    from drakegpt_amd import ops
    from drakegpt_amd import sublayers as S
    dev = torch.device("cuda:0")
    M, C, p, mode = args.M, 384, args.p, args.mode
    g = torch.Generator().manual_seed(1)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)
    o = rn(M, C).bfloat16().to(dev)
    x = rn(M, C, sc=2.0).to(dev)
    W = dict(wproj=rn(C, C, sc=C ** -0.5), w1=rn(4 * C, C, sc=C ** -0.5), w2=rn(C, 4 * C, sc=(4 * C) ** -0.5), wqkv=rn(3 * C, C, sc=C ** -0.5))
    W = {k: v.bfloat16().to(dev) for k, v in W.items()}
    V = dict(bproj=rn(C, sc=0.1), b1=rn(4 * C, sc=0.1), b2=rn(C, sc=0.1), ln2w=1 + rn(C, sc=0.1), ln2b=rn(C, sc=0.1), ln1w=1 + rn(C, sc=0.1), ln1b=rn(C, sc=0.1))
    V = {k: v.to(dev) for k, v in V.items()}
    rng = ops.new_rng_state(1234, dev, 7) if p > 0 else None
    sp, sf = S.site_proj(3), S.site_ffn(3)
    bf = torch.bfloat16

    def reference():
        r = {}
        if mode in (0, 1, 3):
            r["x1"] = ops.gemm_nt(o, W["wproj"], torch.float32, bias=V["bproj"], dropout_p=p, rng_state=rng, site=sp, residual=x)
            r["h2"], r["mean2"], r["rstd2"] = ops.layernorm_fwd(r["x1"], V["ln2w"], V["ln2b"], bf)
        if mode in (0, 1):
            r["bits"] = ops.new_sign_bits(M, 4 * C, dev)
            r["f"] = ops.gemm_nt(r["h2"], W["w1"], bf, bias=V["b1"], relu=True, sign_bits_out=r["bits"])
        if mode in (0, 1, 4):
            fin, x1in = (f_in, x) if mode == 4 else (r["f"], r["x1"])
            r["x2"] = ops.gemm_nt(fin, W["w2"], bf if mode == 1 else torch.float32, bias=V["b2"], dropout_p=p, rng_state=rng, site=sf, residual=x1in)
        if mode in (0, 2, 4):
            r["h1"], r["mean1"], r["rstd1"] = ops.layernorm_fwd(x if mode == 2 else r["x2"], V["ln1w"], V["ln1b"], bf)
        if mode in (0, 2):
            r["qkv"] = ops.gemm_nt(r["h1"], W["wqkv"], bf)
        return r

    Wp = {k: ops.pack_chain_weights(v) for k, v in W.items()}
    f_in = (rn(M, 4 * C).clamp_min(0)).bfloat16().to(dev) if mode == 4 else None

    def chain():
        if mode == 4:
            return ops.block_chain_fwd(4, M, C, f=f_in, x1=x, dropout_p=p, rng_state=rng, site_proj=sp, site_ffn=sf, **Wp, **V)
        return ops.block_chain_fwd(mode, M, C, o=o, x=x, dropout_p=p, rng_state=rng, site_proj=sp, site_ffn=sf, **Wp, **V)

    ref = reference()
    torch.cuda.synchronize()
    got = chain()
    torch.cuda.synchronize()
    bad = False
    for k in ref:
        if k == "bits":
            same = torch.equal(got[k], ref[k])
            diff = int((got[k] != ref[k]).sum())
            print(f"{k:6s} equal={same} differing bytes={diff} of {ref[k].numel()}")
            bad |= diff > ref[k].numel() * 1e-4
            continue
        e = rel(got[k].float(), ref[k].float())
        mx = (got[k].float() - ref[k].float()).abs().max().item()
        print(f"{k:6s} rel {e:.3e} maxabs {mx:.3e} finite={bool(torch.isfinite(got[k].float()).all())}")
        bad |= not (e < 2e-3)
        if not (e < 2e-3) and got[k].dim() == 2:
            d = (got[k].float() - ref[k].float())[:64]
            nc = d.shape[1] // 96
            print("   error energy by 96-column strip:", [f"{d[:, i * 96:(i + 1) * 96].norm().item():.2f}" for i in range(nc)])
            print("   error energy by 16-row strip:", [f"{d[i * 16:(i + 1) * 16].norm().item():.2f}" for i in range(4)])
            print("   ref energy per 96-column strip ~", f"{ref[k].float()[:64, :96].norm().item():.2f}")
    if "qkv" in got:
        tq = (got["h1"].float() @ W["wqkv"].float().t())
        print("qkv vs torch matmul of chain h1: chain", rel(got["qkv"].float(), tq), " separate", rel(ref["qkv"].float(), tq))
        d = (got["qkv"].float() - tq)
        print("   chain-vs-torch error by column chunk of 384:", [f"{d[:, i * 384:(i + 1) * 384].norm().item():.2f}" for i in range(3)])
        # which K range is off?  project the error on per-K-tile contributions
        for kt in range(6):
            part = got["h1"].float()[:, kt * 64:(kt + 1) * 64] @ W["wqkv"].float()[:, kt * 64:(kt + 1) * 64].t()
            print(f"   K tile {kt}: <err, part>/<part, part> = {((d * part).sum() / (part * part).sum()).item():+.3f}")
    if bad and not os.environ.get("DG_CHAIN_DBG"):
        print("MISMATCH")
        sys.exit(1)
    print("chain == separate launches (within rounding)")
    if args.no_time:
        return

    def timeit(fn):
        fn(); fn()
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.cuda.graph(gr):
            for _ in range(args.reps):
                fn()
        gr.replay()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        gr.replay()
        e.record()
        e.synchronize()
        return s.elapsed_time(e) * 1e3 / args.reps
    for _ in range(2):
        t_ref, t_chain = timeit(reference), timeit(chain)
        print(f"M={M} mode={mode} p={p}: separate launches {t_ref:.1f} us, chain {t_chain:.1f} us")


if __name__ == "__main__":
    main()
