"""Model-level parity on the GPU: the drop-in nn.Module path against the committed golden
fixtures (outputs of the real reference) and against the CPU oracle on seeded inputs.

Tolerances (stated, per BASELINE.json north_star):
  fp32 mode : logits/loss/grads within 1e-4 relative (L2) of the reference -- the target is 1e-3;
              sampled tokens bit-exact.
  bf16 mode : ||delta|| / ||ref|| <= 3e-2 for logits, loss within 2e-2 abs.
"""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

V = 80
KW = {
    "BigramLM": dict(vocab_size=V),
    "SingleHeadAttentionLM": dict(vocab_size=V, embedding_dim=32, context_length=8, head_size=32),
    "MultiHeadAttentionLM": dict(vocab_size=V, embedding_dim=32, context_length=8, head_size=32, num_heads=4),
    "BlocksLM": dict(vocab_size=V, embedding_dim=32, context_length=8, num_heads=4, num_layers=3),
    "ResidualBlocksLM": dict(vocab_size=V, embedding_dim=32, context_length=8, num_heads=4, num_layers=3),
    "TransformerLM": dict(vocab_size=V, embedding_dim=32, context_length=8, num_heads=4, num_layers=3, dropout=0.1),
}
NAMES = list(KW)


def rel(a, b):
    """||a-b|| / ||b||, with an absolute floor of 1e-7 per element so that exactly-zero reference
    gradients (T = 1: the softmax over one key has no gradient) compare against fp32 round-off."""
    a, b = a.double().cpu(), b.double().cpu()
    floor = 1e-7 * b.numel() ** 0.5
    return ((a - b).norm() / b.norm().clamp_min(floor / 1e-4)).item() if b.norm() < floor else ((a - b).norm() / b.norm()).item()


def build(name, dev, golden_dir, precision="fp32"):
    import drakegpt_amd as D
    m = D.MODEL_CLASSES[name](**KW[name], precision=precision)
    sd = torch.load(os.path.join(golden_dir, "checkpoints", f"{name}.pt"), weights_only=True)
    m.load_state_dict(sd)
    return m.to(dev)


@pytest.mark.parametrize("name", NAMES)
def test_checkpoint_forward_backward_fp32(dev, golden_dir, name):
    fix = torch.load(os.path.join(golden_dir, f"fwdbwd_{name}.pt"), weights_only=True)
    m = build(name, dev, golden_dir).eval()
    x, y = fix["x"].to(dev), fix["y"].to(dev)
    logits, loss = m(x, y)
    assert logits.shape == (32, V)
    loss.backward()
    assert rel(logits, fix["logits"]) < 1e-4, rel(logits, fix["logits"])
    assert abs(loss.item() - fix["loss"].item()) < 1e-4 * abs(fix["loss"].item())
    for k, p in m.named_parameters():
        if k.startswith("ln_f."):
            assert p.grad is None            # the reference never applies ln_f
            continue
        assert rel(p.grad, fix["grad." + k]) < 1e-4, (k, rel(p.grad, fix["grad." + k]))
    logits3, none = m(x)
    assert none is None and logits3.shape == (4, 8, V)
    assert rel(logits3.reshape(32, V), fix["logits"]) < 1e-4


@pytest.mark.parametrize("name", NAMES)
def test_generate_tokens_bit_exact(dev, golden_dir, name):
    gold = json.load(open(os.path.join(golden_dir, "generate.json")))
    m = build(name, dev, golden_dir).eval()
    torch.manual_seed(gold["seed"])
    out = m.generate(torch.zeros((1, 1), dtype=torch.long, device=dev), max_new_tokens=gold["max_new_tokens"])
    assert out.shape == (1, 101)
    assert out[0].tolist() == gold["tokens"][name]


@pytest.mark.parametrize("T", [1, 5, 32])
def test_small_transformer_T_le_ctx(dev, golden_dir, T):
    import drakegpt_amd as D
    fix = torch.load(os.path.join(golden_dir, "small_TransformerLM.pt"), weights_only=True)
    torch.manual_seed(42)          # same default init, drawn in the reference's construction order
    m = D.TransformerLM(V, 64, 32, 4, 2, 0.0).to(dev).eval()
    x, y = fix[f"T{T}.x"].to(dev), fix[f"T{T}.y"].to(dev)
    logits, loss = m(x, y)
    loss.backward()
    assert rel(logits, fix[f"T{T}.logits"]) < 1e-4
    assert abs(loss.item() - fix[f"T{T}.loss"].item()) < 1e-4
    for k, p in m.named_parameters():
        if p.grad is not None:
            assert rel(p.grad, fix[f"T{T}.grad.{k}"]) < 2e-4, (k, rel(p.grad, fix[f"T{T}.grad.{k}"]))


@pytest.mark.parametrize("precision,tol", [("fp32", 1e-4), ("bf16", 4e-2)])
def test_train_mode_dropout_matches_oracle_with_same_masks(dev, golden_dir, precision, tol):
    """The kernels' dropout masks are a stateless hash; the oracle is handed the very same masks."""
    from oracle import drake_ref as R
    from oracle import rng_ref
    sd = torch.load(os.path.join(golden_dir, "checkpoints", "TransformerLM.pt"), weights_only=True)
    m = build("TransformerLM", dev, golden_dir, precision).train()
    seed = 2024
    m.seed_dropout(seed)
    g = torch.Generator().manual_seed(1)
    B, T = 32, 8
    x = torch.randint(0, V, (B, T), generator=g)
    y = torch.randint(0, V, (B, T), generator=g)
    for step in range(2):              # the device-side step counter advances per forward
        m.zero_grad()
        logits, loss = m(x.to(dev), y.to(dev))
        loss.backward()
        masks = rng_ref.transformer_masks(seed, step, 0.1, B, T, 32, 4, 3)
        lo, ls, grads = R.loss_and_grads("TransformerLM", sd, x, y, p=0.1, training=True, masks=masks)
        assert rel(logits, lo) < tol, (step, rel(logits, lo))
        assert abs(loss.item() - ls.item()) < max(tol, 1e-4) * abs(ls.item())
        names = [k for k, p in m.named_parameters() if p.grad is not None]
        flat = torch.cat([dict(m.named_parameters())[k].grad.reshape(-1).cpu() for k in names])
        flat_ref = torch.cat([grads[k].reshape(-1) for k in names])
        assert rel(flat, flat_ref) < 2 * tol, (step, rel(flat, flat_ref))
        if precision == "fp32":
            for k, p in m.named_parameters():
                if p.grad is not None:
                    assert rel(p.grad, grads[k]) < 3e-4, (step, k, rel(p.grad, grads[k]))
        else:
            # per tensor the bf16 path is held against the reference computed WITH the kernels' bf16 roundings (oracle
            # bf16=True): small q/k gradients carry 10 %+ of bf16 round-off relative to fp32, which a bound against the fp32
            # arithmetic could only cover by being loose enough to pass a wrong term.  Measured: logits 2.0e-3,
            # worst tensor 6.4e-2 .. 0.21 depending on the mask realisation (query / key weights of head size 8: 256-term sums
            # of bf16 products); the whole-vector bound above (8e-2) is what is tight at this size
            lo2, ls2, g2 = R.loss_and_grads("TransformerLM", sd, x, y, p=0.1, training=True, masks=masks, bf16=True)
            per = {k: rel(p.grad, g2[k]) for k, p in m.named_parameters() if p.grad is not None}
            worst = max(per.items(), key=lambda kv: kv[1])
            if os.environ.get("DG_TEST_REPORT"):
                print(f"[parity] tiny bf16 module vs bf16-rounded oracle step {step}: logits {rel(logits, lo2):.3e} worst {worst}", flush=True)
            assert rel(logits, lo2) < 6e-3 and worst[1] < 0.25, (step, rel(logits, lo2), sorted(per.items(), key=lambda kv: -kv[1])[:6])


def test_bf16_logits_close(dev, golden_dir):
    fix = torch.load(os.path.join(golden_dir, "fwdbwd_TransformerLM.pt"), weights_only=True)
    m = build("TransformerLM", dev, golden_dir, "bf16").eval()
    logits, loss = m(fix["x"].to(dev), fix["y"].to(dev))
    assert rel(logits, fix["logits"]) < 3e-2
    assert abs(loss.item() - fix["loss"].item()) < 2e-2 * abs(fix["loss"].item())


def test_components_standalone(dev):
    """Each building block is callable on its own with the reference's positional signature."""
    import drakegpt_amd as D
    from oracle import drake_ref as R
    torch.manual_seed(0)
    C, T, NH, B = 64, 16, 4, 3
    x = torch.randn(B, T, C)
    mods = {
        "Head": D.Head(16, C, T), "Head2": D.Head2(16, C, T, 0.0),
        "MultiHeadAttention": D.MultiHeadAttention(NH, 16, C, T), "MultiHeadAttention2": D.MultiHeadAttention2(NH, 16, C, T),
        "MultiHeadAttention3": D.MultiHeadAttention3(NH, 16, C, T, 0.0),
        "FeedForward": D.FeedForward(C), "FeedForward2": D.FeedForward2(C), "FeedForward3": D.FeedForward3(C, 0.0),
        "Block": D.Block(C, T, NH), "ResidualBlock": D.ResidualBlock(C, NH, T), "ResidualBlock2": D.ResidualBlock2(C, NH, T, 0.0),
    }
    for name, mod in mods.items():
        sd = {k: v.detach().clone() for k, v in mod.state_dict().items()}
        if name.startswith("Head"):
            ref = R.head_forward(sd, "", x)
        elif name.startswith("Multi"):
            ref = R.mha_forward(sd, "", x)
        elif name.startswith("Feed"):
            ref = R.ffn_forward(sd, "", x)
        else:
            ref = R.block_forward(sd, "", name, x)
        xg = x.clone().to(dev).requires_grad_(True)
        out = mod.to(dev)(xg)
        assert out.shape == ref.shape, name
        assert rel(out, ref) < 1e-4, (name, rel(out, ref))
        out.sum().backward()
        assert xg.grad is not None and torch.isfinite(xg.grad).all()


def test_cpu_input_fails_loudly():
    import drakegpt_amd as D
    m = D.BigramLM(V)
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros((1, 4), dtype=torch.long))


def test_kv_cached_generate_equals_uncached(dev):
    """scaled-like shape, ctx 64: cached decoding == the reference algorithm (full forward per token), token for
    token, including the hand-over to the uncached path when the window starts to slide."""
    import drakegpt_amd as D
    torch.manual_seed(0)
    m = D.TransformerLM(V, 64, 24, 4, 2, 0.1).to(dev).eval()
    start = torch.zeros((2, 3), dtype=torch.long, device=dev)
    torch.manual_seed(11)
    a = m.generate(start, 40, use_cache=False)
    torch.manual_seed(11)
    b = m.generate(start, 40, use_cache=True)
    assert a.shape == (2, 43) and torch.equal(a, b)
    # logits of the cached step equal the uncached forward's last row bit for bit (fp32 mode)
    ws, w_lm = m._decode_weights()
    caches = [torch.zeros((2, 24, 3 * 64), device=dev) for _ in m.blocks]
    seq = a[:, :10].contiguous()
    for t in range(10):
        lg = m._decode_step(seq[:, t:t + 1].contiguous(), t, caches, ws, w_lm)
    full, _ = m(seq)
    assert torch.equal(lg, full[:, -1, :])


def test_kv_cache_prefill_is_one_pass_and_matches_uncached(dev):
    """a 17-token prompt: the cache is filled by ONE pass through the training-forward kernels (not 17 decode steps) and the
    continuation equals the reference algorithm's (full forward per token), in fp32 token for token and logits bit for bit"""
    import drakegpt_amd as D
    from drakegpt_amd import ops
    torch.manual_seed(1)
    m = D.TransformerLM(V, 64, 48, 4, 2, 0.1).to(dev).eval()
    g = torch.Generator().manual_seed(4)
    prompt = torch.randint(0, V, (3, 17), generator=g).to(dev)
    calls = {"decode": 0}
    real = ops.attn_decode

    def counting(*a, **k):
        calls["decode"] += 1
        return real(*a, **k)
    ops.attn_decode = counting
    try:
        torch.manual_seed(5)
        b = m.generate(prompt, 9, use_cache=True)
    finally:
        ops.attn_decode = real
    assert calls["decode"] == 8 * 2                            # 8 decode steps x 2 layers; the 17 prompt positions took none
    torch.manual_seed(5)
    a = m.generate(prompt, 9, use_cache=False)
    assert torch.equal(a, b) and a.shape == (3, 26)
    ws, w_lm = m._decode_weights()
    caches = [torch.zeros((3, 48, 3 * 64), device=dev) for _ in m.blocks]
    lg = m._prefill(prompt, caches, ws, w_lm)
    full, _ = m(prompt)
    assert torch.equal(lg, full[:, -1, :])


def test_flat_adamw_state_dict_round_trip(dev):
    """the flat-buffer AdamW keeps m, v and the step count outside torch's per-parameter `state`; state_dict() /
    load_state_dict() export and import them in torch.optim.AdamW's format, so a resumed run continues bit for bit (bias
    correction and moments included) and a torch.optim.AdamW checkpoint loads (ref: src/train.py:121)."""
    import copy
    import drakegpt_amd as D
    from drakegpt_amd.optim import AdamW
    V, C, T = 80, 32, 8
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(1)
    xs = [torch.randint(0, V, (4, T), generator=g).to(dev) for _ in range(6)]
    ys = [torch.randint(0, V, (4, T), generator=g).to(dev) for _ in range(6)]

    def run(m, opt, lo, hi):
        for i in range(lo, hi):
            _, loss = m(xs[i], ys[i])
            opt.zero_grad()
            loss.backward()
            opt.step()

    m1 = D.ResidualBlocksLM(V, C, T, 4, 2).to(dev)
    init = copy.deepcopy(m1.state_dict())
    o1 = AdamW(m1.parameters(), lr=1e-2, betas=(0.9, 0.95))
    run(m1, o1, 0, 3)
    ck_m, ck_o = copy.deepcopy(m1.state_dict()), o1.state_dict()
    n_state = len(ck_o["state"])
    assert n_state == sum(1 for p in m1.parameters() if p.grad is not None) and n_state > 0
    assert all(float(s["step"]) == 3.0 and s["exp_avg"].abs().sum() > 0 for s in ck_o["state"].values())
    run(m1, o1, 3, 6)
    # resume in a fresh model / optimizer
    m2 = D.ResidualBlocksLM(V, C, T, 4, 2).to(dev)
    m2.load_state_dict(ck_m)
    o2 = AdamW(m2.parameters(), lr=1e-2, betas=(0.9, 0.95))
    o2.load_state_dict(ck_o)
    run(m2, o2, 3, 6)
    for (k, a), (_, b) in zip(m1.state_dict().items(), m2.state_dict().items()):
        assert torch.equal(a, b), k
    # without the optimizer state the run differs (the test would otherwise pass vacuously)
    m3 = D.ResidualBlocksLM(V, C, T, 4, 2).to(dev)
    m3.load_state_dict(ck_m)
    o3 = AdamW(m3.parameters(), lr=1e-2, betas=(0.9, 0.95))
    run(m3, o3, 3, 6)
    assert any(not torch.equal(a, b) for a, b in zip(m1.state_dict().values(), m3.state_dict().values()))
    # torch.optim.AdamW's checkpoint loads too: same trajectory from the same start as torch's own continuation (foreach AdamW
    # differs from the fused kernel by rounding only)
    m4 = D.ResidualBlocksLM(V, C, T, 4, 2).to(dev)
    m4.load_state_dict(init)
    ot = torch.optim.AdamW(m4.parameters(), lr=1e-2, betas=(0.9, 0.95))
    run(m4, ot, 0, 3)
    m5 = D.ResidualBlocksLM(V, C, T, 4, 2).to(dev)
    m5.load_state_dict(m4.state_dict())
    o5 = AdamW(m5.parameters(), lr=1e-2, betas=(0.9, 0.95))
    o5.load_state_dict(ot.state_dict())
    run(m4, ot, 3, 6)
    run(m5, o5, 3, 6)
    for (k, a), (_, b) in zip(m4.state_dict().items(), m5.state_dict().items()):
        assert (a.float() - b.float()).abs().max().item() < 2e-5, k
