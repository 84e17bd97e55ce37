"""Does chunking the loss head over rows keep the logits cache-resident?  lm_head forward + cross entropy (in place) + column sums
+ lm_head dX at the GPT-2 vocabulary, whole batch against row chunks (same kernels, same bytes; only the order of the launches
changes: the bf16 logits of a chunk are consumed before the next chunk's are written).

    python tools/head_chunk_bench.py [--M 8192] [--C 768] [--chunks 1 2 4 8 16]
"""
import argparse
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--M", type=int, default=8192)
    ap.add_argument("--C", type=int, default=768)
    ap.add_argument("--V", type=int, default=50257)
    ap.add_argument("--chunks", type=int, nargs="+", default=[1, 2, 4, 8, 16])
    args = ap.parse_args()
    from drakegpt_amd import ops
    from drakegpt_amd import sublayers as S
    dev = torch.device("cuda:0")
    M, C, V = args.M, args.C, args.V
    bf = torch.bfloat16
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(M, C, generator=g)).to(bf).to(dev)
    W = (torch.randn(V, C, generator=g) * C ** -0.5).to(bf).to(dev)
    b = torch.zeros(V, device=dev)
    y = torch.randint(0, V, (M,), generator=g).to(dev)
    Vp = S.k_pad(V, bf)
    Wt = torch.zeros((C, Vp), dtype=bf, device=dev)
    Wt[:, :V] = W.t()
    buf = torch.empty((M, Vp), dtype=bf, device=dev)
    G = 256
    parts = torch.zeros((G, Vp), dtype=torch.float32, device=dev)
    rows = torch.empty((M,), dtype=torch.float32, device=dev)
    dx = torch.empty((M, C), dtype=bf, device=dev)

    def head(nch):
        R = M // nch
        for c in range(nch):
            sl = slice(c * R, (c + 1) * R)
            logits = ops.gemm_nt(x[sl], W, bf, bias=b, out=buf[sl, :V])
            ops.cross_entropy(logits, y[sl], V, dlogits=buf[sl], grad_scale=1.0 / M, loss_rows=rows[sl])
            ops.colsum(buf[sl, :V], parts[0], Vp, G, N=V)
            ops.gemm_nt(buf[sl], Wt, bf, K=Vp, out=dx[sl])

    def timeit(fn, reps=3):
        fn(); torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(reps):
                fn()
        gr.replay(); torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); gr.replay(); e.record(); e.synchronize()
        return s.elapsed_time(e) * 1e3 / reps
    for nch in args.chunks:
        if M % nch or (M // nch) % 128:
            continue
        t = timeit(lambda: head(nch))
        print(f"M={M} C={C} V={V}: {nch:2d} chunk(s) of {M // nch:5d} rows ({(M // nch) * Vp * 2 / 1e6:6.1f} MB of logits each): {t:8.1f} us")


if __name__ == "__main__":
    main()
