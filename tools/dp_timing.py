#!/usr/bin/env python3
"""Where a multi-rank step spends its time, rehearsed on ONE GPU: a world-size-1 "nccl" group with the engine forced onto its
data-parallel paths at the bench configuration (TransformerLM_scaled, B = 64).  With one rank the all-reduce moves nothing, so
this measures what the data-parallel FORM costs -- the graph seams, the RCCL call overhead, the less well filled per-group dW
launches -- not the exchange itself (that needs the 8-GPU node).  engine.debug_timing: HIP events around backward graph(s) /
exchange / optimizer graph.

    python tools/dp_timing.py [--steps 30]
"""
import argparse, os, sys, time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drakegpt_amd as D
from drakegpt_amd.config import PRESETS
from drakegpt_amd.engine import TrainEngine

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=30)
args = ap.parse_args()
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29534")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
cfg = PRESETS["scaled"]
V, C, T, NH, L, B = 80, cfg["embedding_dim"], cfg["context_length"], cfg["num_heads"], cfg["num_layers"], cfg["batch_size"]
corpus = torch.randint(0, V, (1_000_000,), generator=torch.Generator().manual_seed(42))
n = args.steps + 5
offs = torch.randint(1_000_000 - T, (n, B), generator=torch.Generator().manual_seed(2)).to(dev)
for dp, buckets in ((False, None), (True, 1), (True, 3)):
    torch.manual_seed(42)
    m = D.TransformerLM(V, C, T, NH, L, cfg["dropout"], precision="bf16").to(dev)
    eng = TrainEngine(m, B, T, lr=cfg["base_lr"], betas=cfg["betas"], seed=42, rank=0, world_size=1,
                      process_group=dist.group.WORLD if dp else None, dp_buckets=buckets)
    eng.force_dp_path = dp
    eng.set_corpus(corpus)
    eng.stage_offsets(offs)
    for _ in range(5):
        eng.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        eng.step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / args.steps
    split = ""
    if dp:
        eng.stage_offsets(offs)
        eng.debug_timing = True
        acc = {}
        for _ in range(5):
            eng.step()
            for k, v in eng.last_timing.items():
                acc[k] = acc.get(k, 0.0) + v / 5
        split = "  split (ms): " + ", ".join(f"{k} {v:.3f}" for k, v in acc.items())
    eng.check_status()
    print((f"forced data-parallel path, {buckets} bucket(s)" if dp else "single graph") + f": {ms:.3f} ms/step" + split, flush=True)
dist.destroy_process_group()
