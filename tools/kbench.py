#!/usr/bin/env python3
"""Per-kernel micro-benchmarks on the shapes of the training step (run on the GPU box).

    python tools/kbench.py [gemm|tn|attn|ln|all] [--cfg scaled|gpt2_small] [--batch B]
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drakegpt_amd import ops  # noqa: E402
from drakegpt_amd.config import PRESETS  # noqa: E402


def timeit(fn, reps=30, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e-3 / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", nargs="?", default="all")
    ap.add_argument("--cfg", default="scaled")
    ap.add_argument("--batch", type=int, default=None)
    a = ap.parse_args()
    cfg = PRESETS[a.cfg]
    dev = torch.device("cuda:0")
    C, T, NH = cfg["embedding_dim"], cfg["context_length"], cfg["num_heads"]
    B = a.batch or cfg["batch_size"]
    V = cfg.get("vocab_size", 80)
    M = B * T
    bf = torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(0)

    def rnd(*shape, dtype=bf):
        return torch.randn(*shape, generator=g).to(dtype).to(dev)

    if a.what in ("gemm", "all"):
        print(f"-- gemm_nt (bf16), M={M}")
        for name, N, K, od, extra in (("qkv", 3 * C, C, bf, {}), ("ffn1+bias+relu", 4 * C, C, bf, {"bias": True, "relu": True}),
                                      ("ffn2+bias+drop+res", C, 4 * C, torch.float32, {"bias": True, "drop": True, "res": True}),
                                      ("proj+bias+drop+res", C, C, torch.float32, {"bias": True, "drop": True, "res": True}),
                                      ("dffn(hidden)+mask", 4 * C, C, bf, {"mask": True}), ("dh f32", C, 3 * C, torch.float32, {}),
                                      ("dh2 f32", C, 4 * C, torch.float32, {}), ("lm_head", V, C, torch.float32, {"bias": True}),
                                      ("square4096", 4096, 4096, bf, {"M": 4096})):
            Mm = extra.get("M", M)
            A, Bm = rnd(Mm, K), rnd(N, K)
            kw = {}
            if extra.get("bias"):
                kw["bias"] = rnd(N, dtype=torch.float32)
            if extra.get("relu"):
                kw["relu"] = True
            if extra.get("mask"):
                kw["relu_mask"] = rnd(Mm, N)
            if extra.get("res"):
                kw["residual"] = rnd(Mm, N, dtype=torch.float32)
            if extra.get("drop"):
                kw.update(dropout_p=0.2, rng_state=ops.new_rng_state(1, dev), site=1)
            out = torch.empty(Mm, N, dtype=od, device=dev)
            t = timeit(lambda: ops.gemm_nt(A, Bm, od, out=out, **kw))
            print(f"  {name:22s} M={Mm} N={N} K={K}: {t * 1e6:8.1f} us  {2 * Mm * N * K / t / 1e12:7.1f} TF/s")
    if a.what in ("tn", "all"):
        print(f"-- gemm_tn (bf16), R={M}")
        for name, P, Q in (("dWqkv", 3 * C, C), ("dW1", 4 * C, C), ("dW2", C, 4 * C), ("dWproj", C, C), ("dWlm", V, C)):
            for S in (1, 2, 4, 5, 6, 7, 8):
                A, Bm = rnd(M, P), rnd(M, Q)
                part = torch.empty(S, P, Q, device=dev)
                t = timeit(lambda: ops.gemm_tn(A, Bm, part, P * Q, S, P, Q))
                print(f"  {name:8s} P={P} Q={Q} S={S}: {t * 1e6:8.1f} us  {2 * M * P * Q / t / 1e12:7.1f} TF/s")
    if a.what in ("attn", "all"):
        H = C // NH
        qkv = rnd(M, 3 * C)
        rng = ops.new_rng_state(1, dev)
        for p in (0.0, 0.2):
            out, lse = ops.attn_fwd(qkv, B, T, NH, H, H ** -0.5, p, rng, 0)
            dout = rnd(M, C)
            t1 = timeit(lambda: ops.attn_fwd(qkv, B, T, NH, H, H ** -0.5, p, rng, 0))
            t2 = timeit(lambda: ops.attn_bwd(qkv, out, dout, lse, B, T, NH, H, H ** -0.5, p, rng, 0))
            sc = B * NH * T * (T + 1) / 2
            print(f"-- attention p={p}: fwd {t1 * 1e6:7.1f} us ({4 * H * sc / t1 / 1e12:6.1f} TF/s)  bwd {t2 * 1e6:7.1f} us ({10 * H * sc / t2 / 1e12:6.1f} TF/s)")
    if a.what in ("ln", "all"):
        x = rnd(M, C, dtype=torch.float32)
        w, b_ = rnd(C, dtype=torch.float32), rnd(C, dtype=torch.float32)
        y, mean, rstd = ops.layernorm_fwd(x, w, b_, bf)
        t = timeit(lambda: ops.layernorm_fwd(x, w, b_, bf))
        print(f"-- ln_fwd: {t * 1e6:7.1f} us  {M * C * 6 / t / 1e9:7.0f} GB/s")
        for G in (256, 512, 1024):
            pg, pb = torch.empty(G, C, device=dev), torch.empty(G, C, device=dev)
            dy, dres = rnd(M, C, dtype=torch.float32), rnd(M, C, dtype=torch.float32)
            dx = torch.empty_like(x)
            t = timeit(lambda: ops.layernorm_bwd(dy, x, w, mean, rstd, dres, pg, pb, C, G, dx=dx))
            print(f"-- ln_bwd G={G}: {t * 1e6:7.1f} us  {M * C * 16 / t / 1e9:7.0f} GB/s")
        n = 10_800_000
        p_, g_, m_, v_ = (torch.zeros(n, device=dev) for _ in range(4))
        hyper = torch.tensor([1e-3, 0.9, 0.95, 1e-8, 1e-2], device=dev)
        st = ops.new_rng_state(0, dev)
        sh = torch.empty(n, dtype=bf, device=dev)
        t = timeit(lambda: ops.adamw_step(p_, g_, m_, v_, hyper, st, shadow_bf16=sh))
        print(f"-- adamw: {t * 1e6:7.1f} us  {n * 30 / t / 1e9:7.0f} GB/s")
        dy = rnd(M, C, dtype=torch.float32)
        part = torch.empty(256, C, device=dev)
        t = timeit(lambda: ops.dropout_bwd_cast(dy, bf, 0.2, st, 1, colsum_part=part, part_stride=C, n_partials=256))
        print(f"-- dropbwd_cast: {t * 1e6:7.1f} us  {M * C * 6 / t / 1e9:7.0f} GB/s")


if __name__ == "__main__":
    main()
