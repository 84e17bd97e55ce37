#!/usr/bin/env python3
"""evaluate_loss (ref: src/train.py:61-75) at the scaled config: seconds per call with the captured and the eager path."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drakegpt_amd as D
from drakegpt_amd import train
from drakegpt_amd.config import PRESETS
from drakegpt_amd.engine import TrainEngine
cfg = PRESETS["scaled"]
dev = torch.device("cuda:0")
V, C, T, NH, L, B = 80, cfg["embedding_dim"], cfg["context_length"], cfg["num_heads"], cfg["num_layers"], cfg["batch_size"]
m = D.TransformerLM(V, C, T, NH, L, cfg["dropout"], precision="bf16").to(dev)
data = torch.randint(0, V, (1_000_000,), generator=torch.Generator().manual_seed(1)).to(dev)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for graph in (True, False):
    eng = TrainEngine(m, B, T, lr=1e-3, use_graph=graph)
    m.eval()
    train.evaluate_loss(data, data, m, 3, T, B, dev, engine=eng, generator=torch.Generator().manual_seed(0))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = train.evaluate_loss(data, data, m, iters, T, B, dev, engine=eng, generator=torch.Generator().manual_seed(0))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"graph={graph}: evaluate_loss({iters} iters x 2 splits) {dt:.3f} s = {dt / (2 * iters) * 1e3:.3f} ms per batch; train {out['train'].item():.4f} val {out['val'].item():.4f}")
# the loop as it was before eval_losses: one .item() per batch
from drakegpt_amd import ops
from drakegpt_amd.preprocessing import draw_offsets
gen = torch.Generator().manual_seed(0)
torch.cuda.synchronize()
t0 = time.perf_counter()
for it in range(2 * iters):
    ix = draw_offsets(len(data), T, B, gen).to(dev)
    x, y = ops.batch_gather(data, ix, T)
    eng.eval_loss(x, y).item()
dt = time.perf_counter() - t0
print(f"per-batch .item() loop: {dt:.3f} s = {dt / (2 * iters) * 1e3:.3f} ms per batch")
