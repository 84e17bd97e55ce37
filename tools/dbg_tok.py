import os, sys, gc, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drakegpt_amd as D
from drakegpt_amd.engine import TrainEngine
dev = torch.device("cuda:0")
fix = torch.load("tests/golden/traj5_TransformerLM.pt", weights_only=True)
def run(graph, variant):
    m = D.TransformerLM(80, 32, 8, 4, 3, 0.1)
    m.load_state_dict(fix["init"])
    m = m.to(dev).train()
    e = TrainEngine(m, 32, 8, lr=1e-3, betas=(0.9, 0.95), seed=77, use_graph=graph)
    e.keep_logits = True
    out = []
    for step in range(3):
        e.set_batch(fix["x"][step].to(dev), fix["y"][step].to(dev))
        loss = e.step().item()
        out.append({k: v.detach().clone().cpu() for k, v in e.named_grads().items()})
        if variant == 1: _ = e.last_logits.double().cpu()
        if variant == 2: _ = e.last_logits.cpu()
        if variant == 3: _ = e.last_logits.clone(); torch.cuda.synchronize()
        if variant == 4: _ = e.last_logits.double(); torch.cuda.synchronize()
        if variant == 5: _ = torch.zeros(256, 80, dtype=torch.float64, device=dev); torch.cuda.synchronize()
        if variant == 6: _ = e.loss.double(); torch.cuda.synchronize()
        if variant == 7: _ = e.gflat.double(); torch.cuda.synchronize()
    return out
first = run(True, int(sys.argv[1]) if len(sys.argv) > 1 else 1)
ref = run(False, 0)
for variant in range(1):
    g = first
    res = []
    for step in range(3):
        bad = {k: int((g[step][k] - ref[step][k]).abs().gt(1e-2).sum()) for k in g[step] if not ((g[step][k] - ref[step][k]).abs().max().item() <= 1e-3)}
        if bad: res.append((step, bad))
    print("variant", variant, res)
