import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import drakegpt_amd as D
from oracle import drake_ref as R, rng_ref
dev = torch.device("cuda:0")
cfg = R.SCALED
V, B, T = 80, 2, 256
torch.manual_seed(42)
m = D.TransformerLM(V, cfg["embedding_dim"], T, cfg["num_heads"], cfg["num_layers"], cfg["dropout"]).to(dev).train()
sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
m.seed_dropout(99)
g = torch.Generator().manual_seed(3)
x = torch.randint(0, V, (B, T), generator=g); y = torch.randint(0, V, (B, T), generator=g)
logits, loss = m(x.to(dev), y.to(dev)); loss.backward()
masks = rng_ref.transformer_masks(99, 0, cfg["dropout"], B, T, cfg["embedding_dim"], cfg["num_heads"], cfg["num_layers"])
torch.set_num_threads(8)
lo, ls, grads = R.loss_and_grads("TransformerLM", sd, x, y, p=cfg["dropout"], training=True, masks=masks)
def rel(a, b): a, b = a.double().cpu(), b.double().cpu(); return ((a - b).norm() / b.norm()).item()
for k, p_ in m.named_parameters():
    if p_.grad is None or not k.startswith("blocks.5."): continue
    if "heads" in k and not k.startswith("blocks.5.sa_head.heads.0"): continue
    print(k, f"{rel(p_.grad, grads[k]):.2e}")
k = "blocks.5.ffwd.net.0.weight"
d = (dict(m.named_parameters())[k].grad.cpu().double() - grads[k].double())
rows = d.norm(dim=1) / grads[k].double().norm(dim=1)
print("W1 rows with rel err > 1e-3:", int((rows > 1e-3).sum()), "of", rows.numel(), "max", rows.max().item(), "median", rows.median().item())
k = "blocks.5.ffwd.net.2.weight"
d = (dict(m.named_parameters())[k].grad.cpu().double() - grads[k].double())
cols = d.norm(dim=0) / grads[k].double().norm(dim=0)
print("W2 cols with rel err > 1e-3:", int((cols > 1e-3).sum()), "of", cols.numel(), "max", cols.max().item(), "median", cols.median().item())
