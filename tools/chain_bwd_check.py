"""dg_block_chain_bwd against the launches it replaces (dg_gemm_nt dX forms + dg_layernorm_bwd_fused), and both timed on one box.

    python tools/chain_bwd_check.py [--M 16384] [--p 0.2] [--mode 0|1|2] [--reps 20] [--no-time]

The separate launches round the dX GEMM's output to bf16 before the LayerNorm backward reads it; the chain hands over fp32
accumulators, so dx / g agree within bf16 rounding (not bit for bit).
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
    gen = torch.Generator().manual_seed(2)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=gen) * sc)
    bf, f32 = torch.bfloat16, torch.float32
    has_q, has_2 = mode in (0, 2), mode in (0, 1)
    # W^T shadows, [in, out]: the B operands of the dX GEMMs
    WT = dict(wqkvT=rn(C, 3 * C, sc=(3 * C) ** -0.5), w2T=rn(4 * C, C, sc=C ** -0.5), w1T=rn(C, 4 * C, sc=(4 * C) ** -0.5), wprojT=rn(C, C, sc=C ** -0.5))
    WT = {k: v.to(bf).to(dev) for k, v in WT.items()}
    WTp = {k: ops.pack_chain_weights(v) for k, v in WT.items()}
    dqkv = rn(M, 3 * C).to(bf).to(dev)
    x, x1 = rn(M, C, sc=2.0).to(dev), rn(M, C, sc=2.0).to(dev)
    ln1w, ln2w = (1 + rn(C, sc=0.1)).to(dev), (1 + rn(C, sc=0.1)).to(dev)
    h1, mean1, rstd1 = ops.layernorm_fwd(x, ln1w, torch.zeros_like(ln1w), bf)
    h2, mean2, rstd2 = ops.layernorm_fwd(x1, ln2w, torch.zeros_like(ln2w), bf)
    dresid = rn(M, C).to(bf).to(dev)
    g_in = rn(M, C).to(bf).to(dev)
    # ReLU sign bits as the forward GEMM leaves them
    bits = ops.new_sign_bits(M, 4 * C, dev)
    w1 = rn(4 * C, C, sc=C ** -0.5).to(bf).to(dev)
    ops.gemm_nt(h2, w1, bf, relu=True, sign_bits_out=bits)
    rng = ops.new_rng_state(4321, dev, 5) if p > 0 else None
    s_ffn, s_proj = S.site_ffn(2), S.site_proj(2)
    G = 256
    rows_cs = ops.gemm_nt_colsum_rows(bf, M, 4 * C, C)
    stride = 8 * C + 4 * C

    def parts(n):
        return torch.zeros((n, stride), dtype=f32, device=dev)

    Pref, Pch = parts(max(G, rows_cs)), parts(2 * (M // 64))

    def reference(check=True):
        r, P = {}, Pref
        if has_q:
            dh = ops.gemm_nt(dqkv, WT["wqkvT"], bf, K=3 * C)
            r["dx1"], r["g1"] = ops.layernorm_bwd_fused(dh, x, ln1w, mean1, rstd1, dresid, P[0, 0:C], P[0, C:2 * C], stride, G, bf, p, rng, s_ffn,
                                                        P[0, 2 * C:3 * C], stream_dtype=bf)
        if has_2:
            g = r["g1"] if mode == 0 else g_in
            r["df"] = ops.gemm_nt(g, WT["w2T"], bf, K=C, sign_bits=bits, colsum_part=P[:rows_cs, 8 * C:12 * C])
            dh2 = ops.gemm_nt(r["df"], WT["w1T"], bf, K=4 * C)
            r["dx2"], r["g2"] = ops.layernorm_bwd_fused(dh2, x1, ln2w, mean2, rstd2, r["dx1"] if mode == 0 else dresid, P[0, 3 * C:4 * C], P[0, 4 * C:5 * C],
                                                        stride, G, bf, p, rng, s_proj, P[0, 5 * C:6 * C], stream_dtype=bf)
            r["dout"] = ops.gemm_nt(r["g2"], WT["wprojT"], bf, K=C)
        if check:
            r["sums"] = P.sum(0)
        return r

    def chain(check=True):
        P = Pch
        kw = dict(part_stride=stride, dropout_p=p, rng_state=rng, site_ffn_below=s_ffn, site_proj=s_proj)
        if has_q:
            kw.update(dqkv=dqkv, wqkvT=WTp["wqkvT"], x=x, mean1=mean1, rstd1=rstd1, ln1w=ln1w, dresid1=dresid, dln1w_part=P[0, 0:C], dln1b_part=P[0, C:2 * C],
                      gbias1_part=P[0, 2 * C:3 * C])
        if has_2:
            kw.update(w2T=WTp["w2T"], bits=bits, db1_part=P[0, 8 * C:12 * C], w1T=WTp["w1T"], x1=x1, mean2=mean2, rstd2=rstd2, ln2w=ln2w,
                      dln2w_part=P[0, 3 * C:4 * C], dln2b_part=P[0, 4 * C:5 * C], gbias2_part=P[0, 5 * C:6 * C], wprojT=WTp["wprojT"])
            if mode == 1:
                kw.update(g_in=g_in, dresid2=dresid)
        r = ops.block_chain_bwd(mode, M, C, **kw)
        if check:
            r["sums"] = P.sum(0)
        return r

    ref = reference()
    torch.cuda.synchronize()
    got = chain()
    torch.cuda.synchronize()
    bad = False
    names = {0: "dln1w", 1: "dln1b", 2: "gbias1", 3: "dln2w", 4: "dln2b", 5: "gbias2"}
    for k in ref:
        if k == "sums":
            for i, nm in names.items():
                a, b = got[k][i * C:(i + 1) * C], ref[k][i * C:(i + 1) * C]
                if b.abs().sum() == 0 and a.abs().sum() == 0:
                    continue
                e = rel(a, b)
                print(f"{nm:7s} rel {e:.3e}")
                bad |= not (e < 1e-2)
            a, b = got[k][8 * C:12 * C], ref[k][8 * C:12 * C]
            if b.abs().sum() > 0 or a.abs().sum() > 0:
                e = rel(a, b)
                print(f"db1     rel {e:.3e}")
                bad |= not (e < 1e-2)
            continue
        e = rel(got[k].float(), ref[k].float())
        mx = (got[k].float() - ref[k].float()).abs().max().item()
        print(f"{k:6s} rel {e:.3e} maxabs {mx:.3e} finite={bool(torch.isfinite(got[k].float()).all())}")
        tol = 1e-2
        bad |= not (e < tol)
        if not (e < tol) and got[k].dim() == 2:
            d = (got[k].float() - ref[k].float())[:64]
            nc = d.shape[1] // 96
            print("   error energy by 96-column strip:", [f"{d[:, i * 96:(i + 1) * 96].norm().item():.2f}" for i in range(nc)])
            print("   error energy by 16-row strip:", [f"{d[i * 16:(i + 1) * 16].norm().item():.2f}" for i in range(4)])
            print("   ref energy per 96-column strip ~", f"{ref[k].float()[:64, :96].norm().item():.2f}")
    # an independent check of the whole chain in fp64 (no intermediate rounding): both paths must sit at bf16 distance from it
    def exact():
        r = {}
        d = lambda t: t.double()
        def ln_bwd(dh, xx, mu, rs, gam, dres):
            xh = (d(xx) - d(mu)[:, None]) * d(rs)[:, None]
            t = dh * d(gam)[None, :]
            return d(rs)[:, None] * (t - t.mean(1, keepdim=True) - xh * (t * xh).mean(1, keepdim=True)) + d(dres)
        if has_q:
            dh = d(dqkv) @ d(WT["wqkvT"]).t()
            r["dx1"] = ln_bwd(dh, x, mean1, rstd1, ln1w, dresid)
        return r
    ex = exact()
    for k in ex:
        print(f"{k:6s} vs fp64: chain {rel(got[k].float(), ex[k]):.3e}  separate {rel(ref[k].float(), ex[k]):.3e}")
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
        t_ref, t_chain = timeit(lambda: reference(False)), timeit(lambda: chain(False))
        print(f"M={M} mode={mode} p={p}: separate launches {t_ref:.1f} us, chain {t_chain:.1f} us")


if __name__ == "__main__":
    main()
