"""Pin the oracle to the reference and write tests/golden/ -- TEST INFRASTRUCTURE.

Run ONLY in the build container (``/root/reference`` is not on the GPU box):

    python oracle/make_golden.py

1. imports the real reference modules from /root/reference/src (model.py, model_component.py,
   preprocessing.py need nothing but torch), loads the six shipped checkpoints with
   ``weights_only=True`` and asserts that ``oracle/drake_ref.py`` is BIT-IDENTICAL to the
   reference for: eval forward (logits, loss), every parameter gradient, train-mode dropout under
   the same seed, sampled tokens, default init under seed 42, get_batch, a 5-step AdamW
   trajectory and the CyclicLR sequence;
2. writes the reference's outputs as small fixtures (inputs + expected outputs only; the
   checkpoints are re-saved as plain tensor dicts -- data, no code).
"""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
REF = "/root/reference"
sys.path.insert(0, os.path.join(REF, "src"))

import model as ref_model                      # noqa: E402  (the reference)
import preprocessing as ref_prep               # noqa: E402
from oracle import drake_ref as R              # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
V = 80


def ref_build(name, cfg, vocab=V):
    """What src/train.py:31-58 passes to each constructor."""
    kw = {
        "BigramLM": dict(vocab_size=vocab),
        "SingleHeadAttentionLM": dict(vocab_size=vocab, embedding_dim=cfg["embedding_dim"],
                                      context_length=cfg["context_length"], head_size=cfg["head_size"]),
        "MultiHeadAttentionLM": dict(vocab_size=vocab, embedding_dim=cfg["embedding_dim"],
                                     context_length=cfg["context_length"], head_size=cfg["head_size"],
                                     num_heads=cfg["num_heads"]),
        "BlocksLM": dict(vocab_size=vocab, embedding_dim=cfg["embedding_dim"],
                         context_length=cfg["context_length"], num_heads=cfg["num_heads"],
                         num_layers=cfg["num_layers"]),
        "ResidualBlocksLM": dict(vocab_size=vocab, embedding_dim=cfg["embedding_dim"],
                                 context_length=cfg["context_length"], num_heads=cfg["num_heads"],
                                 num_layers=cfg["num_layers"]),
        "TransformerLM": dict(vocab_size=vocab, embedding_dim=cfg["embedding_dim"],
                              context_length=cfg["context_length"], num_heads=cfg["num_heads"],
                              num_layers=cfg["num_layers"], dropout=cfg["dropout"]),
    }[name]
    return getattr(ref_model, name)(**kw)


def make_corpus_text(n_lines: int = 420, seed: int = 7) -> str:
    """a small synthetic lyric-shaped text (own word list; no dataset is available offline): mixed case, digits, punctuation,
    a few non-ASCII characters and blank lines, so that the char vocabulary is not trivially a-z"""
    import random
    rnd = random.Random(seed)
    words = ("river stone lantern morning window ocean paper thunder quiet ember harbor signal velvet copper winter echo "
             "meadow circuit marble feather garden mirror shadow anchor violet pepper candle bridge engine orchard "
             "we you they never always again over under until because maybe slowly tonight tomorrow 7 24 1999").split()
    tails = [",", ".", "", "!", "?", " --", ";", "...", " (yeah)", " é", " ñ"]
    out = []
    for i in range(n_lines):
        n = rnd.randint(3, 9)
        line = " ".join(rnd.choice(words) for _ in range(n))
        line = line.capitalize() if rnd.random() < 0.8 else line.upper()
        out.append(line + rnd.choice(tails))
        if i % 17 == 16:
            out.append("")
            out.append(f"[Verse {i // 17 + 1}]")
    return "\n".join(out) + "\n"


def same(a, b, what):
    assert a.shape == b.shape, (what, a.shape, b.shape)
    assert torch.equal(a, b), f"oracle != reference for {what}: max|d|={(a - b).abs().max().item():.3e}"


def main():
    os.makedirs(os.path.join(OUT, "checkpoints"), exist_ok=True)
    torch.set_num_threads(1)          # fixed reduction order for the bit-equality checks
    summary = {}

    # ---------------------------------------------------------------- shipped checkpoints
    gen = {}
    for name in R.MODEL_NAMES:
        sd = torch.load(os.path.join(REF, "model", f"{name}.pt"), map_location="cpu", weights_only=True)
        sd = {k: v.clone() for k, v in sd.items()}
        m = ref_build(name, R.TINY)
        missing = m.load_state_dict(sd)
        assert not missing.missing_keys and not missing.unexpected_keys
        torch.save(sd, os.path.join(OUT, "checkpoints", f"{name}.pt"))

        # (1) sampled tokens: eval(), seed 42, generate(zeros(1,1), 100)  (ref: src/train.py:174-177)
        m.eval()
        torch.manual_seed(42)
        toks_ref = m.generate(torch.zeros((1, 1), dtype=torch.long), max_new_tokens=100)
        torch.manual_seed(42)
        toks_orc = R.lm_generate(name, sd, torch.zeros((1, 1), dtype=torch.long), 100)
        assert torch.equal(toks_ref, toks_orc), name
        gen[name] = toks_ref[0].tolist()

        # (2) eval fwd + bwd on a seeded (4,8) batch
        g = torch.Generator().manual_seed(0)
        x = torch.randint(0, V, (4, 8), generator=g)
        y = torch.randint(0, V, (4, 8), generator=g)
        m.zero_grad()
        logits, loss = m(x, y)
        loss.backward()
        lo, ls, grads = R.loss_and_grads(name, sd, x, y)
        same(logits.detach(), lo, f"{name} logits")
        same(loss.detach(), ls, f"{name} loss")
        fix = {"x": x, "y": y, "logits": logits.detach().clone(), "loss": loss.detach().clone()}
        for k, p in m.named_parameters():
            if p.grad is None:
                assert k.startswith("ln_f."), k
                assert k not in grads
                continue
            same(p.grad, grads[k], f"{name} grad {k}")
            fix["grad." + k] = p.grad.detach().clone()
        logits3, none = m(x)
        assert none is None and logits3.shape == (4, 8, V)
        same(logits3.detach(), R.lm_forward(name, sd, x)[0], f"{name} logits3")
        torch.save(fix, os.path.join(OUT, f"fwdbwd_{name}.pt"))
        summary[name] = {"loss": float(loss), "n_params": sum(p.numel() for p in m.parameters())}

    with open(os.path.join(OUT, "generate.json"), "w") as f:
        json.dump({"seed": 42, "start": [[0]], "max_new_tokens": 100, "tokens": gen}, f)

    # ---------------------------------------------------------------- train-mode dropout, same seed
    sd = torch.load(os.path.join(OUT, "checkpoints", "TransformerLM.pt"), weights_only=True)
    m = ref_build("TransformerLM", R.TINY)
    m.load_state_dict(sd)
    m.train()
    g = torch.Generator().manual_seed(1)
    x = torch.randint(0, V, (32, 8), generator=g)
    y = torch.randint(0, V, (32, 8), generator=g)
    torch.manual_seed(7)
    logits, loss = m(x, y)
    torch.manual_seed(7)
    lo, ls = R.lm_forward("TransformerLM", sd, x, y, p=R.TINY["dropout"], training=True)
    same(logits.detach(), lo, "train-mode logits")
    same(loss.detach(), ls, "train-mode loss")

    # ---------------------------------------------------------------- default init + reduced shapes
    small = dict(R.TINY, context_length=32, embedding_dim=64, head_size=64, num_heads=4, num_layers=2, dropout=0.0)
    for name in R.MODEL_NAMES:
        for cfg_name, cfg in (("tiny", R.TINY), ("small", small)):
            torch.manual_seed(42)
            m = ref_build(name, cfg)
            sd_o = R.init_state_dict(name, V, cfg, seed=42)
            sd_r = m.state_dict()
            assert list(sd_o.keys()) == list(sd_r.keys()), (name, cfg_name)
            for k in sd_r:
                same(sd_r[k], sd_o[k], f"init {name}/{cfg_name}/{k}")
    # scaled config: init equality too (10.8M params)
    torch.manual_seed(42)
    m = ref_build("TransformerLM", R.SCALED)
    sd_o = R.init_state_dict("TransformerLM", V, R.SCALED, seed=42)
    for k, v in m.state_dict().items():
        same(v, sd_o[k], f"init scaled/{k}")
    summary["scaled_n_params"] = sum(p.numel() for p in m.parameters())

    # reduced-shape TransformerLM (B=2, T in {1,5,32}, C=64, NH=4 -> H=16) fwd/bwd fixtures
    torch.manual_seed(42)
    m = ref_build("TransformerLM", small)
    m.eval()
    sd_small = {k: v.clone() for k, v in m.state_dict().items()}
    fix = {}
    for T in (1, 5, 32):
        g = torch.Generator().manual_seed(100 + T)
        x = torch.randint(0, V, (2, T), generator=g)
        y = torch.randint(0, V, (2, T), generator=g)
        m.zero_grad()
        logits, loss = m(x, y)
        loss.backward()
        lo, ls, grads = R.loss_and_grads("TransformerLM", sd_small, x, y)
        same(logits.detach(), lo, f"small T={T} logits")
        same(loss.detach(), ls, f"small T={T} loss")
        fix[f"T{T}.x"], fix[f"T{T}.y"] = x, y
        fix[f"T{T}.logits"], fix[f"T{T}.loss"] = logits.detach().clone(), loss.detach().clone()
        for k, p in m.named_parameters():
            if p.grad is not None:
                same(p.grad, grads[k], f"small T={T} grad {k}")
                fix[f"T{T}.grad.{k}"] = p.grad.detach().clone()
    torch.save(fix, os.path.join(OUT, "small_TransformerLM.pt"))
    summary["small_cfg"] = {k: v for k, v in small.items() if k != "betas"}

    # ---------------------------------------------------------------- get_batch
    data = torch.randint(0, V, (5000,), generator=torch.Generator().manual_seed(42))
    torch.manual_seed(3)
    xr, yr = ref_prep.get_batch(data, 8, 32, "cpu")
    torch.manual_seed(3)
    xo, yo = R.get_batch(data, 8, 32)
    assert torch.equal(xr, xo) and torch.equal(yr, yo)

    # ---------------------------------------------------------------- 5-step AdamW trajectory, p = 0
    traj = {}
    for name in ("TransformerLM", "BigramLM"):
        cfg = dict(R.TINY, dropout=0.0)
        torch.manual_seed(42)
        m = ref_build(name, cfg)
        sd_t = {k: v.clone() for k, v in m.state_dict().items()}
        opt = torch.optim.AdamW(m.parameters(), lr=cfg["base_lr"], betas=cfg["betas"])   # train.py:121
        o_opt = R.AdamWState(R.trainable_keys(name, sd_t), cfg["base_lr"], cfg["betas"])
        m.train()
        torch.manual_seed(5)
        losses = []
        batches = []
        for it in range(5):
            x, y = ref_prep.get_batch(data, cfg["context_length"], cfg["batch_size"], "cpu")
            batches.append((x, y))
            logits, loss = m(x, y)
            opt.zero_grad()
            loss.backward()
            opt.step()
            lo = R.train_step(name, sd_t, o_opt, x, y, p=0.0, training=True)
            assert lo == float(loss), (name, it, lo, float(loss))
            losses.append(float(loss))
        maxd = 0.0
        for k, v in m.state_dict().items():
            d = (v - sd_t[k]).abs().max().item()
            maxd = max(maxd, d)
        # torch's default AdamW is the foreach implementation; ours is per-tensor: allow 1e-7 abs
        assert maxd <= 1e-7, (name, maxd)
        traj[name] = {"losses": losses, "max_param_diff_vs_oracle": maxd}
        torch.save({"init": R.init_state_dict(name, V, cfg, 42), "final": {k: v.clone() for k, v in m.state_dict().items()},
                    "x": torch.stack([b[0] for b in batches]), "y": torch.stack([b[1] for b in batches]),
                    "losses": torch.tensor(losses, dtype=torch.float64)},
                   os.path.join(OUT, f"traj5_{name}.pt"))
    summary["traj5"] = traj

    # ---------------------------------------------------------------- CyclicLR sequence
    lin = torch.nn.Linear(1, 1)
    opt = torch.optim.AdamW(lin.parameters(), lr=R.TINY["base_lr"], betas=R.TINY["betas"])
    sch = torch.optim.lr_scheduler.CyclicLR(opt, base_lr=R.TINY["base_lr"], max_lr=R.TINY["max_lr"],
                                            step_size_up=5, mode="triangular", cycle_momentum=False)
    seq = [opt.param_groups[0]["lr"]]
    for _ in range(20):
        opt.step()
        sch.step()
        seq.append(opt.param_groups[0]["lr"])
    for i, lr in enumerate(seq):
        assert abs(R.cyclic_lr(i, R.TINY["base_lr"], R.TINY["max_lr"]) - lr) < 1e-15, (i, lr)
    summary["cyclic_lr"] = seq

    # ---------------------------------------------------------------- data side: get_train_val_data (src/preprocessing.py:48-86)
    import contextlib
    import io
    import tempfile
    text_path = os.path.join(OUT, "corpus_fixture.txt")
    if not os.path.exists(text_path):
        with open(text_path, "w", encoding="utf-8") as f:
            f.write(make_corpus_text())
    with open(text_path, "r", encoding="utf-8") as f:
        text = f.read()
    with tempfile.TemporaryDirectory() as td:
        tp, vp = os.path.join(td, "train_data.pt"), os.path.join(td, "val_data.pt")
        with contextlib.redirect_stdout(io.StringIO()):
            ref_prep.get_train_val_data(text_path, tp, vp)           # the reference writes the two .pt files
        rt = torch.load(tp, weights_only=True)
        rv = torch.load(vp, weights_only=True)
    ot, ov, ovocab, vocab = R.encode_corpus(text)
    same(ot, rt, "train_data.pt")
    same(ov, rv, "val_data.pt")
    enc, dec, rvocab = ref_prep.get_mapper(text)
    assert rvocab == ovocab and dec(rt[:200].tolist()) == text[:200] and enc(text[:50]) == ot[:50].tolist()
    assert rt.dtype == torch.int64 and rt.dim() == 1
    torch.save({"train": rt.to(torch.int16), "val": rv.to(torch.int16), "vocab_size": rvocab, "vocab": "".join(vocab)},
               os.path.join(OUT, "corpus_fixture.pt"))
    summary["corpus_fixture"] = {"chars": len(text), "vocab_size": rvocab, "n_train": len(rt), "n_val": len(rv)}

    with open(os.path.join(OUT, "summary.json"), "w") as f:
        json.dump(summary, f, indent=1)
    print("oracle == reference (bit-exact) on all checks; fixtures written to", OUT)
    print(json.dumps({k: summary[k] for k in ("TransformerLM", "scaled_n_params")}, indent=1))


if __name__ == "__main__":
    main()
