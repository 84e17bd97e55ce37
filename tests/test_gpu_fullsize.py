"""Full-size and odd-shape checks: the scaled configuration against the CPU oracle, vocabularies
that are not a multiple of the MFMA granule, ragged sequence lengths, and size-independent
properties at BASELINE.json's shapes (causality, linearity, gradient-sum identities)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def test_scaled_config_fp32_step_matches_oracle(dev):
    """TransformerLM_scaled (C 384, T 256, 6 heads, 6 layers), fp32 mode, dropout on with shared masks."""
    import drakegpt_amd as D
    from oracle import drake_ref as R
    from oracle import rng_ref
    cfg = R.SCALED
    V, B, T = 80, 2, 256
    torch.manual_seed(42)
    m = D.TransformerLM(V, cfg["embedding_dim"], T, cfg["num_heads"], cfg["num_layers"], cfg["dropout"]).to(dev).train()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    m.seed_dropout(99)
    g = torch.Generator().manual_seed(3)
    x = torch.randint(0, V, (B, T), generator=g)
    y = torch.randint(0, V, (B, T), generator=g)
    logits, loss = m(x.to(dev), y.to(dev))
    loss.backward()
    masks = rng_ref.transformer_masks(99, 0, cfg["dropout"], B, T, cfg["embedding_dim"], cfg["num_heads"], cfg["num_layers"])
    torch.set_num_threads(8)
    lo, ls, grads = R.loss_and_grads("TransformerLM", sd, x, y, p=cfg["dropout"], training=True, masks=masks)
    assert rel(logits, lo) < 1e-4 and abs(loss.item() - ls.item()) < 1e-4
    # A ReLU pre-activation within fp32 round-off of zero can land on the other side of the kink on the GPU (786k hidden
    # activations per layer: it happens about once per step): that flips one (token, unit) mask bit, changes that unit's row of
    # W1 / b1 by percents and reaches everything BELOW it at the 1e-3 level.  Which units flip depends on the dropout
    # realisation (with seed 99 and the round-2 mask stream: one unit of the top block, 8.7 % in its W1 row, 2.3e-3 in the
    # tensor, 4e-4 .. 1e-3 below).  So: what lies above every ReLU is exact; the top block's W1 is exact row by row except for
    # at most a few flipped units; every tensor within 5e-3 and the whole gradient vector within 2e-3.
    errs = {k: rel(p.grad, grads[k]) for k, p in m.named_parameters() if p.grad is not None}
    if __import__("os").environ.get("DG_TEST_REPORT"):
        byl = {}
        for k, e in errs.items():
            l = k.split(".")[1] if k.startswith("blocks.") else k
            byl[l] = max(byl.get(l, 0.0), e)
        print("[parity] scaled fp32 module step: logits", rel(logits, lo), "worst per layer", {k: f"{v:.1e}" for k, v in byl.items()}, flush=True)
    assert max(errs.values()) < 5e-3, max(errs.items(), key=lambda kv: kv[1])
    top = cfg["num_layers"] - 1
    for k in ("lm_head.weight", "lm_head.bias", f"blocks.{top}.ffwd.net.2.weight", f"blocks.{top}.ffwd.net.2.bias"):
        assert errs[k] < 1e-5, (k, errs[k])
    w1 = f"blocks.{top}.ffwd.net.0.weight"
    d = dict(m.named_parameters())[w1].grad.cpu().double() - grads[w1].double()
    rows = d.norm(dim=1) / grads[w1].double().norm(dim=1)
    assert int((rows > 1e-3).sum()) <= 4 and rows.median().item() < 1e-5, (int((rows > 1e-3).sum()), rows.median().item())
    flat = torch.cat([p.grad.reshape(-1).cpu() for k, p in m.named_parameters() if p.grad is not None])
    flat_ref = torch.cat([grads[k].reshape(-1) for k, p in m.named_parameters() if p.grad is not None])
    assert rel(flat, flat_ref) < 2e-3


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
def test_odd_vocab_and_ragged_T_engine_vs_module(dev, precision):
    """V = 83 (not a multiple of 8: padded contraction in the lm_head dX), T = 40 < context 64,
    head size 64 (MFMA attention with a ragged last tile in bf16): engine gradients == autograd-path gradients."""
    import drakegpt_amd as D
    from drakegpt_amd.engine import TrainEngine
    V, C, Tctx, NH, L, B, T = 83, 128, 64, 2, 2, 4, 40
    torch.manual_seed(7)
    m = D.TransformerLM(V, C, Tctx, NH, L, 0.0, precision=precision).to(dev).train()
    g = torch.Generator().manual_seed(5)
    x = torch.randint(0, V, (B, T), generator=g).to(dev)
    y = torch.randint(0, V, (B, T), generator=g).to(dev)
    logits, loss = m(x, y)
    assert logits.shape == (B * T, V)
    loss.backward()
    ref = {k: p.grad.clone() for k, p in m.named_parameters() if p.grad is not None}
    m.zero_grad(set_to_none=True)
    eng = TrainEngine(m, B, T, lr=0.0, weight_decay=0.0, use_graph=False)
    eng.set_batch(x, y)
    l2 = eng.step()
    # measured (round 2): the two paths run the same kernels on the same operands -- every tensor bit-identical except the
    # atomically accumulated token table (1e-8) and, in bf16, lm_head.bias (1.6e-3: the engine's cross entropy hands dlogits
    # over in bf16 and the bias gradient is their column sum; the module path sums the fp32 dlogits)
    tol = 1e-6 if precision == "fp32" else 4e-3
    assert abs(l2.item() - loss.item()) < 1e-5 * abs(loss.item())
    views = dict(wq=eng.grad_view("0.wqkv"), w1=eng.grad_view("1.w1"), lm=eng.grad_view("lm.w"), lmb=eng.grad_view("lm.b"),
                 tok=eng.grad_view("tok"), pos=eng.grad_view("pos"), ln=eng.grad_view("0.ln1w"))
    H = C // NH
    if __import__("os").environ.get("DG_TEST_REPORT"):
        print(f"[parity] odd-vocab engine vs module {precision}: " + ", ".join(f"{k}={rel(v, r):.2e}" for k, v, r in (
            ("wq", views["wq"][:H], ref["blocks.0.sa_head.heads.0.query.weight"]), ("w1", views["w1"], ref["blocks.1.ffwd.net.0.weight"]),
            ("lm", views["lm"], ref["lm_head.weight"]), ("lmb", views["lmb"], ref["lm_head.bias"]), ("tok", views["tok"], ref["token_embedding_table.weight"]),
            ("pos", views["pos"], ref["position_embedding_table.weight"]), ("ln", views["ln"], ref["blocks.0.ln1.weight"]))), flush=True)
    assert rel(views["wq"][:H], ref["blocks.0.sa_head.heads.0.query.weight"]) < tol
    assert rel(views["w1"], ref["blocks.1.ffwd.net.0.weight"]) < tol
    assert rel(views["lm"], ref["lm_head.weight"]) < tol and rel(views["lmb"], ref["lm_head.bias"]) < tol
    assert rel(views["tok"], ref["token_embedding_table.weight"]) < tol
    assert rel(views["pos"], ref["position_embedding_table.weight"]) < tol and torch.all(views["pos"][T:] == 0)
    assert rel(views["ln"], ref["blocks.0.ln1.weight"]) < tol


def test_attention_causality_and_dropout_determinism_T1024(dev):
    """GPT-2 shape (T 1024, head 64): changing the future never changes the past; same (seed, step) =>
    same output, next step => different mask."""
    from drakegpt_amd import ops
    B, T, NH, H = 2, 1024, 3, 64
    g = torch.Generator().manual_seed(0)
    qkv = torch.randn(B * T, 3 * NH * H, generator=g).bfloat16().to(dev)
    out, lse = ops.attn_fwd(qkv, B, T, NH, H, H ** -0.5, 0.0, None, 0)
    qkv2 = qkv.clone().view(B, T, -1)
    qkv2[:, 700:] = torch.randn(B, T - 700, 3 * NH * H, generator=g).bfloat16().to(dev)
    out2, _ = ops.attn_fwd(qkv2.view(B * T, -1), B, T, NH, H, H ** -0.5, 0.0, None, 0)
    o1, o2 = out.view(B, T, -1), out2.view(B, T, -1)
    assert torch.equal(o1[:, :700], o2[:, :700]) and not torch.equal(o1[:, 700:], o2[:, 700:])
    assert torch.isfinite(out).all() and torch.isfinite(lse).all()
    rng = ops.new_rng_state(5, dev, 0)
    a, _ = ops.attn_fwd(qkv, B, T, NH, H, H ** -0.5, 0.1, rng, 3)
    b, _ = ops.attn_fwd(qkv, B, T, NH, H, H ** -0.5, 0.1, rng, 3)
    ops.state_advance(rng)
    c, _ = ops.attn_fwd(qkv, B, T, NH, H, H ** -0.5, 0.1, rng, 3)
    assert torch.equal(a, b) and not torch.equal(a, c)
    # row 0 attends to itself only: out[0] == v[0] when nothing is dropped
    v0 = qkv.view(B, T, 3, NH, H)[:, 0, 2]
    assert torch.equal(o1[:, 0].view(B, NH, H), v0)


def test_gemm_linearity_and_ce_gradient_identity_full_size(dev):
    from drakegpt_amd import ops
    M, N, K = 16384, 1152, 384
    g = torch.Generator().manual_seed(1)
    A1 = torch.randn(M, K, generator=g).bfloat16().to(dev)
    A2 = torch.randn(M, K, generator=g).bfloat16().to(dev)
    W = torch.randn(N, K, generator=g).bfloat16().to(dev)
    # exact-in-fp32 operands (bf16 sums of two bf16 are not exact, so test with A2 = 2*A1 pattern and negation)
    y1 = ops.gemm_nt(A1, W, torch.float32)
    y2 = ops.gemm_nt((A1.float() * 2).bfloat16(), W, torch.float32)
    y3 = ops.gemm_nt((-A1.float()).bfloat16(), W, torch.float32)
    assert torch.equal(y2, 2 * y1)                       # exact scaling by a power of two
    assert rel(y3, -y1) < 1e-6                           # the MFMA's internal summation is not sign-symmetric bit for bit
    # dW of a column-permuted dY is the row-permuted dW (TN GEMM, 8 splits)
    perm = torch.randperm(N, generator=g).to(dev)
    dY = torch.randn(M, N, generator=g).bfloat16().to(dev)
    p1, p2 = torch.empty(8, N, K, device=dev), torch.empty(8, N, K, device=dev)
    ops.gemm_tn(dY, A2, p1, N * K, 8, N, K)
    ops.gemm_tn(dY[:, perm].contiguous(), A2, p2, N * K, 8, N, K)
    assert torch.equal(p1.sum(0)[perm], p2.sum(0))
    # cross entropy: each gradient row sums to zero, loss rows are non-negative
    V = 50257
    logits = torch.randn(512, V, generator=g).to(dev) * 2
    tgt = torch.randint(0, V, (512,), generator=g).to(dev)
    dl = torch.empty(512, 50264, device=dev)
    rows = ops.cross_entropy(logits, tgt, V, dlogits=dl, grad_scale=1.0)
    assert rows.min().item() >= 0 and dl[:, :V].sum(1).abs().max().item() < 1e-4


def test_gpt2_small_shape_engine_steps(dev):
    """BASELINE.json configs[2]: GPT-2-small shape (V 50257, C 768, T 1024, 12 heads, 12 layers), bf16, B = 2.  Two captured
    engine steps: loss finite and near ln V at init, falling after one AdamW step on the same batch; the engine's gradient equals
    the autograd (module) path's on the same (seed, step) dropout masks; size-independent properties of the gradient.
    (The oracle at this size needs ~20 s / step on the host and 40 GB of autograd state: the engine is tied to the oracle at
    the scaled configuration in test_gpu_engine_oracle.py and to the module path here.)  ref: src/model.py:558-609."""
    import math
    import drakegpt_amd as D
    from drakegpt_amd.config import GPT2_SMALL as cfg
    from drakegpt_amd.engine import TrainEngine
    V, C, T, NH, L, p, B = cfg["vocab_size"], cfg["embedding_dim"], cfg["context_length"], cfg["num_heads"], cfg["num_layers"], cfg["dropout"], 2
    torch.manual_seed(42)
    m = D.TransformerLM(V, C, T, NH, L, p, precision="bf16").to(dev).train()
    assert sum(q.numel() for q in m.parameters()) == 163059793
    g = torch.Generator().manual_seed(11)
    x = torch.randint(0, V, (B, T), generator=g).to(dev)
    y = torch.randint(0, V, (B, T), generator=g).to(dev)
    seed = 31337
    m.seed_dropout(seed)
    logits, loss = m(x, y)
    assert logits.shape == (B * T, V)
    loss.backward()
    ref = {k: q.grad.detach().clone() for k, q in m.named_parameters() if q.grad is not None}
    l_mod = loss.item()
    del logits, loss
    m.zero_grad(set_to_none=True)
    torch.cuda.empty_cache()
    # (a) fp32 logits, as the module path: the two run the same kernels on the same operands
    eng32 = TrainEngine(m, B, T, lr=0.0, weight_decay=0.0, seed=seed, use_graph=False, logits="fp32", grad_stream="fp32")
    assert not eng32.bf16_logits and eng32.stream_dtype == torch.float32
    eng32.set_batch(x, y)
    l32 = eng32.step().item()
    g32 = {k: v.detach().clone() for k, v in eng32.named_grads().items()}
    assert abs(l32 - l_mod) < 1e-5 * l_mod
    f32 = torch.cat([g32[k].reshape(-1).double() for k in ref])
    f_ref = torch.cat([ref[k].reshape(-1).double() for k in ref])
    e32 = ((f32 - f_ref).norm() / f_ref.norm()).item()
    w32 = max(((k, rel(g32[k], ref[k])) for k in ref), key=lambda kv: kv[1])
    assert e32 < 1e-5 and w32[1] < 5e-4, (e32, w32)        # measured 4.2e-7 / 3.3e-5 (lm_head.bias: column sums of bf16 dlogits)
    del eng32, g32, f32
    torch.cuda.empty_cache()
    # (b) the default at this vocabulary: bf16 logits overwritten in place by their gradient
    eng = TrainEngine(m, B, T, lr=cfg["base_lr"], betas=cfg["betas"], seed=seed, use_graph=True)
    assert eng.grouped_dw and eng.onehot is None          # V = 50257: the token-table gradient keeps the atomic scatter-add
    assert eng.bf16_logits and eng.stream_dtype == torch.bfloat16      # the defaults at this shape
    eng.set_batch(x, y)
    l0 = eng.step().item()
    torch.cuda.synchronize()
    assert math.isfinite(l0) and abs(l0 - l_mod) < 1e-3 * l_mod
    assert math.log(V) - 0.5 < l0 < math.log(V) + 3.0, l0
    grads = eng.named_grads()
    assert set(grads) == set(ref)
    flat = torch.cat([grads[k].reshape(-1).double() for k in ref])
    flat_ref = torch.cat([ref[k].reshape(-1).double() for k in ref])
    e_flat = ((flat - flat_ref).norm() / flat_ref.norm()).item()
    worst = max(((k, rel(grads[k], ref[k])) for k in ref), key=lambda kv: kv[1])
    if __import__("os").environ.get("DG_TEST_REPORT"):
        print(f"[parity] gpt2-small engine vs module: flat {e_flat:.3e}, worst tensor {worst}", flush=True)
    assert e_flat < 1e-2 and worst[1] < 3e-2, (e_flat, worst)        # bf16 logits: every gradient sees the 2^-9 rounding of the scores
    # properties: softmax-minus-one-hot rows sum to zero => so does the lm_head bias gradient; token rows that do not occur in the
    # batch get exactly zero; position rows are all used (T = context length)
    gb = grads["lm_head.bias"].double()
    assert abs(gb.sum().item()) < 1e-3 * gb.abs().sum().item()
    gt = grads["token_embedding_table.weight"]
    unseen = torch.ones(V, dtype=torch.bool, device=dev)
    unseen[x.reshape(-1)] = False
    assert torch.all(gt[unseen] == 0) and torch.all(gt[~unseen].abs().sum(1) > 0)
    assert torch.all(grads["position_embedding_table.weight"].abs().sum(1) > 0)
    assert torch.equal(m.ln_f.weight, torch.ones_like(m.ln_f.weight))
    l1 = eng.step().item()                                 # same batch again, after one AdamW step: a graph replay
    assert math.isfinite(l1) and l1 < l0, (l0, l1)
    assert eng.step_count() == 2
