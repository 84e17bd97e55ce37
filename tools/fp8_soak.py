"""Does precision "fp8" TRAIN at a GPT-2 shape?  The same model, seed, corpus and window offsets stepped in bf16 and in fp8; the loss
of both every few steps.

    python tools/fp8_soak.py gpt2_medium 8 300 [every=25] [lr=3e-4]

Corpus (synthetic, there is no tokenizer or dataset on the box): 4 000 fixed "phrases" of 6-24 random token ids drawn Zipf-like
from the whole vocabulary, concatenated in random order -- inside a phrase the next token is a function of the context, at a phrase
boundary it is not, so the loss has a floor well above zero and a long way to fall from ln V.
"""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def corpus(V, n_tokens, seed=7):
    g = torch.Generator().manual_seed(seed)
    rank = torch.arange(1, V + 1, dtype=torch.float64)
    prob = (1.0 / rank) / (1.0 / rank).sum()
    perm = torch.randperm(V, generator=g)
    phrases = []
    for _ in range(4000):
        n = int(torch.randint(6, 25, (1,), generator=g))
        phrases.append(perm[torch.multinomial(prob, n, replacement=True, generator=g)])
    out, total = [], 0
    order = torch.randint(0, len(phrases), (n_tokens // 6,), generator=g).tolist()
    for i in order:
        out.append(phrases[i])
        total += phrases[i].numel()
        if total >= n_tokens:
            break
    return torch.cat(out)[:n_tokens]


def main():
    import drakegpt_amd as D
    from drakegpt_amd.config import DRAKE_VOCAB_SIZE, PRESETS
    from drakegpt_amd.engine import TrainEngine
    name, B, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    every = int(sys.argv[4]) if len(sys.argv) > 4 else 25
    lr = float(sys.argv[5]) if len(sys.argv) > 5 else 3e-4
    dev = torch.device("cuda:0")
    cfg = dict(PRESETS[name])
    V, T = cfg.get("vocab_size", DRAKE_VOCAB_SIZE), cfg["context_length"]
    data = corpus(V, 4_000_000)
    gen = torch.Generator().manual_seed(11)
    offs = torch.stack([torch.randint(data.numel() - T - 1, (B,), generator=gen) for _ in range(steps)]).to(dev)
    curves = {}
    for prec in ("bf16", "fp8"):
        torch.manual_seed(42)
        model = D.TransformerLM(V, cfg["embedding_dim"], T, cfg["num_heads"], cfg["num_layers"], cfg["dropout"], precision=prec).to(dev)
        eng = TrainEngine(model, B, T, lr=lr, betas=cfg["betas"], seed=42)
        eng.set_corpus(data)
        eng.stage_offsets(offs)
        acc, pts = torch.zeros((), device=dev), []
        t0 = time.perf_counter()
        for i in range(steps):
            acc += eng.step()
            if (i + 1) % every == 0:
                pts.append(round((acc / every).item(), 4))
                acc.zero_()
                print(f"{prec} step {i + 1}: mean loss of the last {every} steps {pts[-1]:.4f}  ({time.perf_counter() - t0:.0f} s)", flush=True)
        eng.check_status()
        curves[prec] = pts
        del eng, model
        torch.cuda.empty_cache()
    print(f"{name} B={B} T={T} V={V} lr={lr} dropout={cfg['dropout']}: mean loss per {every} steps")
    print(" step   bf16     fp8     fp8/bf16")
    for k, (a, b) in enumerate(zip(curves["bf16"], curves["fp8"])):
        print(f"{(k + 1) * every:5d}  {a:7.4f}  {b:7.4f}  {b / a:6.4f}")
    print(json.dumps({"config": name, "B": B, "steps": steps, "every": every, "lr": lr, **curves}))


if __name__ == "__main__":
    main()
