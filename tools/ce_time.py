"""time dg_cross_entropy on bf16 logits in place:  python tools/ce_time.py [M=8192] [V=50257]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drakegpt_amd import ops
M = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
V = int(sys.argv[2]) if len(sys.argv) > 2 else 50257
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
ld = (V + 7) // 8 * 8
src = torch.zeros((M, ld), dtype=torch.bfloat16)
src[:, :V] = (torch.randn(M, V, generator=g) * 2.0).bfloat16()
src = src.to(dev)
tg = torch.randint(0, V, (M,), generator=g).to(dev)
buf = src.clone()
rows = ops.cross_entropy(buf[:, :V], tg, V, dlogits=buf, grad_scale=1.0 / M)
x = src[:, :V].double()
lse = torch.logsumexp(x, 1)
ref_rows = lse - x.gather(1, tg[:, None])[:, 0]
ref_g = (torch.softmax(x, 1) - torch.nn.functional.one_hot(tg, V)) / M
print("loss rows rel err %.3e; gradient rel err %.3e (bf16 rounding)" % (((rows.double() - ref_rows).norm() / ref_rows.norm()).item(),
      ((buf[:, :V].double() - ref_g).norm() / ref_g.norm()).item()))
def body(with_ce):
    buf.copy_(src)
    if with_ce:
        ops.cross_entropy(buf[:, :V], tg, V, dlogits=buf, grad_scale=1.0 / M)
def timeit(with_ce):
    body(with_ce); torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    with torch.cuda.graph(gr):
        for _ in range(5): body(with_ce)
    gr.replay(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record(); gr.replay(); e.record(); e.synchronize()
    return s.elapsed_time(e) * 1e3 / 5
for _ in range(3):
    c = timeit(False)
    print(f"M={M} V={V}: cross entropy in place {timeit(True) - c:.0f} us (copy {c:.0f} us)")
