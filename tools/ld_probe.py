import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drakegpt_amd import ops
from tools.kbench import timeit
dev = torch.device("cuda:0"); bf = torch.bfloat16
g = torch.Generator().manual_seed(0)
M = 16384
def rnd(*s, dtype=bf): return torch.randn(*s, generator=g).to(dtype).to(dev)
for (N, K) in ((384, 1536), (1536, 384), (1152, 384), (384, 384)):
    for padA, padC in ((0, 0), (64, 0), (0, 64), (64, 64), (8, 8)):
        Afull = rnd(M, K + padA); A = Afull[:, :K]
        B = rnd(N, K)
        Cfull = torch.empty(M, N + padC, dtype=bf, device=dev); C = Cfull[:, :N]
        t = timeit(lambda: ops.gemm_nt(A, B, bf, out=C))
        print(f"N={N} K={K} lda={K+padA} ldc={N+padC}: {t*1e6:.1f} us {2*M*N*K/t/1e12:.0f} TF/s")
