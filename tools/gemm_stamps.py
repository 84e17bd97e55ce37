import os, sys, ctypes, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from drakegpt_amd import ops, _lib
dev = torch.device("cuda:0"); bf = torch.bfloat16
g = torch.Generator().manual_seed(0)
def rnd(*s, dtype=bf): return torch.randn(*s, generator=g).to(dtype).to(dev)
M, N, K = 16384, int(sys.argv[1]) if len(sys.argv) > 1 else 1152, int(sys.argv[2]) if len(sys.argv) > 2 else 384
A, B = rnd(M, K), rnd(N, K)
out = torch.empty(M, N, dtype=bf, device=dev)
for _ in range(3): ops.gemm_nt(A, B, bf, out=out)
buf = torch.zeros(256 * 64, dtype=torch.int64, device=dev)
_lib.lib.dg_debug_set_stamp_buffer.argtypes = [ctypes.c_void_p]
_lib.lib.dg_debug_set_stamp_buffer(buf.data_ptr())
ops.gemm_nt(A, B, bf, out=out)
torch.cuda.synchronize()
_lib.lib.dg_debug_set_stamp_buffer(None)
st = buf.cpu().view(256, 64)
nk = K // 64
for wg in (0, 1, 100, 255):
    t = st[wg]; t = t[t > 0]
    d = (t[1:] - t[:-1]).tolist()
    print(f"wg {wg}: n={len(t)} total={int(t[-1]-t[0])} cycles; deltas: {d}")
import statistics
tot = [int(st[w][st[w] > 0][-1] - st[w][st[w] > 0][0]) for w in range(256)]
print("total cycles per WG: min/med/max", min(tot), statistics.median(tot), max(tot))
starts = [int(st[w][0]) for w in range(256)]; ends=[int(st[w][st[w] > 0][-1]) for w in range(256)]
print("span all WGs (cycles):", max(ends) - min(starts), " start skew:", max(starts) - min(starts))
