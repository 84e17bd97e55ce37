import os, sys, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from drakegpt_amd import ops
from oracle import rng_ref
dev = torch.device("cuda:0")
B, T, NH, H, p = 2, 256, 3, 64, 0.2
seed, step, site = 77, 3, 4
g = torch.Generator().manual_seed(1)
qkv = torch.randn(B * T, 3 * NH * H, generator=g).bfloat16().to(dev)
rng = ops.new_rng_state(seed, dev, step)
out, lse = ops.attn_fwd(qkv, B, T, NH, H, H ** -0.5, p, rng, site, keep=True)
torch.cuda.synchronize()
kb = out.dg_keep.cpu().numpy().view(np.uint64)
keep = rng_ref.keep_mask(seed, step, site, p, B * NH * T * T).reshape(B * NH, T, T)
nblk = T // 32
ntri = nblk * (nblk + 1) // 2
def krow(r, hh): return (r & 3) + 8 * (r >> 2) + 4 * hh
bad = {}
tot = 0
for bh in range(B * NH):
    for qb in range(nblk):
        for kt in range(qb + 1):
            rec = kb[(bh * ntri + qb * (qb + 1) // 2 + kt) * 16:][:16]
            for r in range(16):
                m = int(rec[r])
                for lane in range(64):
                    c, hh = lane & 31, lane >> 5
                    qi, kj = qb * 32 + c, kt * 32 + krow(r, hh)
                    if kj > qi: continue
                    tot += 1
                    if ((m >> lane) & 1) != int(keep[bh, qi, kj]):
                        bad[(bh, qb, kt)] = bad.get((bh, qb, kt), 0) + 1
print("checked", tot, "bad tiles", len(bad))
for k in sorted(bad)[:40]: print(k, bad[k])
# where inside a tile?
bh, qb, kt = 0, 3, 0
rec = kb[(bh * ntri + qb * (qb + 1) // 2 + kt) * 16:][:16]
pos = []
for r in range(16):
    for lane in range(64):
        c, hh = lane & 31, lane >> 5
        qi, kj = qb * 32 + c, kt * 32 + krow(r, hh)
        if ((int(rec[r]) >> lane) & 1) != int(keep[bh, qi, kj]): pos.append((r, lane))
print("tile (0,3,0) mismatches (r, lane):", pos)
import collections
print("by r:", collections.Counter(r for r, _ in pos))
