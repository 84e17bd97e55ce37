#!/usr/bin/env python3
"""micro-benchmark: the block Linears of a model shape as bf16 and as fp8 NT GEMMs, and the just-in-time quantisation passes.
    python tools/fp8_bench.py [M] [C]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drakegpt_amd import ops

M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
C = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
dev = torch.device("cuda:0")
E4, E5 = torch.float8_e4m3fn, torch.float8_e5m2


def timeit(fn, reps=20):
    fn(); fn()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); g.replay(); e.record(); e.synchronize()
    return s.elapsed_time(e) * 1e3 / reps


gen = torch.Generator().manual_seed(0)
one = torch.ones(1, device=dev)
for name, N, K in (("qkv", 3 * C, C), ("proj", C, C), ("ffn1", 4 * C, C), ("ffn2", C, 4 * C)):
    A = torch.randn(M, K, generator=gen).bfloat16().to(dev)
    B = (torch.randn(N, K, generator=gen) * 0.05).bfloat16().to(dev)
    Aq, sa = ops.fp8_quantize(A, E4)
    Bq, sb = ops.fp8_quantize(B, E4)
    A5, s5 = ops.fp8_quantize(A, E5)
    fl = 2.0 * M * N * K
    t_b = timeit(lambda: ops.gemm_nt(A, B, torch.bfloat16))
    t_8 = timeit(lambda: ops.gemm_nt(Aq, Bq, torch.bfloat16, scale_a=sa, scale_b=sb))
    t_5 = timeit(lambda: ops.gemm_nt(A5, Bq, torch.bfloat16, scale_a=s5, scale_b=sb))
    t_q = timeit(lambda: ops.fp8_quantize(A, E4))
    st = ops.new_rng_state(1, dev, 0)
    parts2 = torch.zeros(2 * ops.FP8_AMAX_PARTS, device=dev)
    ops.fp8_quantize(A, E4, amax=parts2[:ops.FP8_AMAX_PARTS]); parts2[ops.FP8_AMAX_PARTS:].copy_(parts2[:ops.FP8_AMAX_PARTS])
    t_d = timeit(lambda: ops.fp8_quantize_delayed(A, E4, parts2, st))
    print(f"{name:5s} M={M} N={N} K={K}: bf16 {t_b:7.1f} us ({fl / t_b / 1e6:6.0f} TF/s)  e4m3 {t_8:7.1f} us ({fl / t_8 / 1e6:6.0f} TF/s)  "
          f"e5m2 {t_5:7.1f} us  quantize A ({A.numel() * 2 / 1e6:.0f} MB) jit {t_q:6.1f} us, delayed {t_d:6.1f} us", flush=True)
