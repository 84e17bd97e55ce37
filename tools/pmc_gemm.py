#!/usr/bin/env python3
"""run a handful of launches of each GEMM shape for PMC collection (rocprofv3 --pmc ...)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drakegpt_amd import ops
dev = torch.device("cuda:0")
bf = torch.bfloat16
g = torch.Generator().manual_seed(0)
def rnd(*s, dtype=bf): return torch.randn(*s, generator=g).to(dtype).to(dev)
M = 16384
A, B = rnd(M, 384), rnd(1152, 384)
out = torch.empty(M, 1152, dtype=bf, device=dev)
for _ in range(5): ops.gemm_nt(A, B, bf, out=out)
A2, B2 = rnd(4096, 4096), rnd(4096, 4096)
out2 = torch.empty(4096, 4096, dtype=bf, device=dev)
for _ in range(3): ops.gemm_nt(A2, B2, bf, out=out2)
Y, X = rnd(M, 1536), rnd(M, 384)
part = torch.empty(8, 1536, 384, device=dev)
for _ in range(5): ops.gemm_tn(Y, X, part, 1536 * 384, 8, 1536, 384)
torch.cuda.synchronize()
