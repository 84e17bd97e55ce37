#!/usr/bin/env python3
"""One-GPU rehearsal of the data-parallel step over RCCL: a world-size-1 "nccl" process group, the engine forced onto its
multi-rank path (backward graph -> all-reduce of the flat gradient on the RCCL stream -> optimizer graph), compared with
the single-graph path step by step.  With one rank the all-reduce is the identity, so the losses must match exactly.

    python tools/dp_rccl_smoke.py          (MASTER_ADDR / MASTER_PORT default to 127.0.0.1:29533)
"""
import os, sys
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drakegpt_amd as D
from drakegpt_amd.config import PRESETS
from drakegpt_amd.engine import TrainEngine

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
cfg = PRESETS["scaled"]
V, C, T, NH, L, B = 80, cfg["embedding_dim"], cfg["context_length"], cfg["num_heads"], cfg["num_layers"], 16
corpus = torch.randint(0, V, (200_000,), generator=torch.Generator().manual_seed(1))
offs = torch.randint(200_000 - T, (6, B), generator=torch.Generator().manual_seed(2)).to(dev)
losses = []
for dp in (False, True):
    torch.manual_seed(42)
    m = D.TransformerLM(V, C, T, NH, L, cfg["dropout"], precision="bf16").to(dev)
    eng = TrainEngine(m, B, T, lr=1e-3, seed=7, rank=0, world_size=1, process_group=dist.group.WORLD if dp else None)
    if dp:
        eng.force_dp_path = True          # two graphs with the all-reduce between them, as on N > 1 ranks
    eng.set_corpus(corpus)
    ls = []
    for i in range(6):
        eng.set_offsets(offs[i])
        ls.append(eng.step().item())
    dist.barrier(device_ids=[0])
    torch.cuda.synchronize()
    losses.append(ls)
    print("dp path" if dp else "single graph", [round(x, 6) for x in ls])
assert losses[0] == losses[1], "the RCCL path changed the result"
dist.destroy_process_group()
print("RCCL data-parallel path ok (world size 1: identical losses)")
