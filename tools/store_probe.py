import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drakegpt_amd import ops
from tools.kbench import timeit
dev = torch.device("cuda:0"); bf = torch.bfloat16
g = torch.Generator().manual_seed(0)
def rnd(*s, dtype=bf): return torch.randn(*s, generator=g).to(dtype).to(dev)
M = 16384
for N, K in ((1152, 384), (1536, 384), (384, 1536)):
    A, B = rnd(M, K), rnd(N, K)
    for od in (bf, torch.float32):
        out = torch.empty(M, N, dtype=od, device=dev)
        t = timeit(lambda: ops.gemm_nt(A, B, od, out=out))
        print(f"N={N} K={K} out={od}: {t*1e6:.1f} us  out MB={out.numel()*out.element_size()/1e6:.1f}")
# raw copy bandwidths via torch for reference
x = torch.empty(64 * 1024 * 1024, dtype=torch.float32, device=dev)
y = torch.empty_like(x)
t = timeit(lambda: y.copy_(x)); print(f"torch copy f32 256MB: {t*1e6:.1f} us {2*x.numel()*4/t/1e12:.2f} TB/s")
xb = torch.empty(16 * 1024 * 1024, dtype=torch.float32, device=dev); yb = torch.empty(16 * 1024 * 1024, dtype=bf, device=dev)
t = timeit(lambda: ops.cast(xb, bf, out=yb)); print(f"dg_cast f32->bf16 64MB->32MB: {t*1e6:.1f} us {(xb.numel()*6)/t/1e12:.2f} TB/s")
t = timeit(lambda: yb.copy_(xb)); print(f"torch cast f32->bf16: {t*1e6:.1f} us {(xb.numel()*6)/t/1e12:.2f} TB/s")
t = timeit(lambda: y.zero_()); print(f"torch zero 256MB: {t*1e6:.1f} us {x.numel()*4/t/1e12:.2f} TB/s")
