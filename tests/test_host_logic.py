"""CPU: host-side logic of the drop-in surface -- constructors, state_dict layout, default-init
order, the parameter-count estimate, tokenizer, schedule, loud failure without a GPU."""
import os
import re

import pytest
import torch

import drakegpt_amd as D
from drakegpt_amd import config, preprocessing, train
from oracle import drake_ref as R

V = 80
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _kw(name, cfg):
    return train.build_model.__wrapped__(name, cfg) if hasattr(train.build_model, "__wrapped__") else None


@pytest.mark.parametrize("name", list(D.MODEL_CLASSES))
def test_reference_checkpoints_load_unchanged(golden_dir, name):
    sd = torch.load(os.path.join(golden_dir, "checkpoints", f"{name}.pt"), weights_only=True)
    m, cfg, _ = train.build_model(name, False, config.PARAMS, config.SCALE_PARAMS, V, "cpu")
    res = m.load_state_dict(sd)
    assert not res.missing_keys and not res.unexpected_keys
    mine = m.state_dict()
    assert list(mine.keys()) == list(sd.keys())
    assert all(mine[k].shape == sd[k].shape and mine[k].dtype == sd[k].dtype for k in sd)


@pytest.mark.parametrize("name", list(D.MODEL_CLASSES))
def test_default_init_equals_reference_construction_order(name):
    """torch.manual_seed(42) + construction gives the reference's weights (oracle init is pinned bit-exact)."""
    torch.manual_seed(42)
    m, _, _ = train.build_model(name, False, config.PARAMS, config.SCALE_PARAMS, V, "cpu")
    ref = R.init_state_dict(name, V, R.TINY, seed=42)
    for k, v in m.state_dict().items():
        assert torch.equal(v, ref[k]), k


def test_parameter_counts_and_estimate():
    actual = {"BigramLM": 6400, "SingleHeadAttentionLM": 8528, "MultiHeadAttentionLM": 8528, "BlocksLM": 17840,
              "ResidualBlocksLM": 42896, "TransformerLM": 43344}
    for name, n in actual.items():
        m, _, _ = train.build_model(name, False, config.PARAMS, config.SCALE_PARAMS, V, "cpu")
        assert sum(p.numel() for p in m.parameters()) == n
    # the reference's own (inexact) estimate, README.md:30-35 / SURVEY 0.11
    assert D.model_params(config.PARAMS, "TransformerLM", V) == 45584
    assert D.model_params(config.SCALE_PARAMS, "TransformerLM", V) == 11223632
    C, L, T = 384, 6, 256
    m, _, _ = train.build_model("TransformerLM", True, config.PARAMS, config.SCALE_PARAMS, V, "cpu")
    assert sum(p.numel() for p in m.parameters()) == L * (12 * C * C + 10 * C) + V * C + T * C + 2 * C + (C * V + V) == 10800464


def test_constructor_signatures_are_positional_like_the_reference():
    D.Head(8, 32, 8); D.Head2(8, 32, 8, 0.1)
    D.MultiHeadAttention(4, 8, 32, 8); D.MultiHeadAttention2(4, 8, 32, 8); D.MultiHeadAttention3(4, 8, 32, 8, 0.1)
    D.FeedForward(32); D.FeedForward2(32); D.FeedForward3(32, 0.1)
    b = D.Block(32, 8, 4)                   # (embedding_dim, context_length, num_heads)
    r = D.ResidualBlock(32, 4, 8)           # (embedding_dim, num_heads, context_length)  -- swapped order
    r2 = D.ResidualBlock2(32, 4, 8, 0.1)
    assert len(b.sa_head.heads) == 4 and len(r.sa_head.heads) == 4 and len(r2.sa_head.heads) == 4
    m = D.MultiHeadAttentionLM(V, 32, 8, 32, 4)
    assert m.sa_head.heads[0].head_size == 8        # head_size // num_heads (src/model.py:264)
    t = D.TransformerLM(V, 32, 8, 4, 3, 0.1)
    assert t.context_length == 8 and hasattr(t, "ln_f")
    assert [blk.layer_index for blk in t.blocks] == [0, 1, 2]
    f8 = D.TransformerLM(V, 32, 8, 4, 3, 0.1, precision="fp8")      # round 2: bf16 activations + fp8 operands for the block Linears
    assert f8.fp8 and f8.act_dtype == torch.bfloat16 and f8.blocks[0].sa_head.run_mode == "fp8" and not t.fp8
    with pytest.raises(ValueError):
        D.TransformerLM(V, 32, 8, 4, 3, 0.1, precision="fp16")


def test_no_cpu_fallback_and_no_oracle_import_in_product():
    m = D.TransformerLM(V, 32, 8, 4, 1, 0.0)
    with pytest.raises(RuntimeError, match="GPU"):
        m(torch.zeros((2, 8), dtype=torch.long))
    with pytest.raises(RuntimeError, match="GPU"):
        preprocessing.get_batch(torch.zeros(100, dtype=torch.long), 8, 4, "cpu")
    pkg = os.path.join(ROOT, "drakegpt_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), fn
            assert "/root/reference" not in src, fn


def test_tokenizer_and_split():
    enc, dec, n = preprocessing.get_mapper("hello world")
    assert n == 8 and dec(enc("hello world")) == "hello world" and enc("d") == [1]
    data, decode, vs = preprocessing.encode_text("abcabcabca")
    assert data.dtype == torch.int64 and vs == 3
    tr, va = preprocessing.split_train_val(data)
    assert len(tr) == 9 and len(va) == 1


def test_offsets_match_reference_get_batch_draws():
    data = torch.randint(0, V, (5000,), generator=torch.Generator().manual_seed(42))
    torch.manual_seed(3)
    ix = preprocessing.draw_offsets(len(data), 8, 32)
    torch.manual_seed(3)
    x, y = R.get_batch(data, 8, 32)
    assert torch.equal(torch.stack([data[i:i + 8] for i in ix]), x)
    assert torch.equal(torch.stack([data[i + 1:i + 9] for i in ix]), y)


def test_schedule_and_paths_and_presets():
    assert [round(train.cyclic_lr(i, 1e-3, 5e-3), 10) for i in range(7)] == [0.001, 0.0018, 0.0026, 0.0034, 0.0042, 0.005, 0.0042]
    assert train.get_model_path("m", "TransformerLM", True).endswith("TransformerLM_scaled.pt")
    assert train.get_model_path("m", "BigramLM", False).endswith("BigramLM.pt")
    assert config.SCALE_PARAMS["embedding_dim"] == 384 and config.PARAMS["context_length"] == 8
    assert config.TRAIN == {"iters": 10000, "eval_iters": 200, "eval_interval": 500}


def test_dropout_hash_reference_properties():
    from oracle import rng_ref
    m = rng_ref.keep_mask(1, 0, 0, 0.2, 200000)
    assert abs(m.mean() - 0.8) < 0.005
    assert not (m == rng_ref.keep_mask(1, 1, 0, 0.2, 200000)).all()      # step re-keys the stream
    assert not (m == rng_ref.keep_mask(1, 0, 1, 0.2, 200000)).all()      # site re-keys the stream
    assert (m == rng_ref.keep_mask(1, 0, 0, 0.2, 200000)).all()
    assert rng_ref.threshold(0.0) == 0 and rng_ref.threshold(0.5) == 2 ** 15 and rng_ref.threshold(0.2) == 13107
    # the lean element hash behaves like independent Bernoulli draws: lag correlations ~ 1/sqrt(n),
    # row sums (256 consecutive elements = one attention row) binomially dispersed
    import numpy as np
    k = rng_ref.keep_mask(7, 3, 5, 0.2, 256 * 8192).astype(np.float64)
    for lag in (1, 2, 3, 8, 256, 257, 384):
        assert abs(np.corrcoef(k[:-lag], k[lag:])[0, 1]) < 4e-3, lag
    rows = k.reshape(-1, 256).sum(1)
    assert 0.93 < rows.var() / (256 * 0.2 * 0.8) < 1.07
    # two elements share one 32-bit hash (low / high 16 bits): no correlation inside a pair, columns dispersed too
    assert abs(np.corrcoef(k[0::2], k[1::2])[0, 1]) < 4e-3
    # (per key the ratio of 256 column means scatters by ~ sqrt(2 / 255) = 9 %: the bound is on the mean over several keys)
    ratios = []
    for seed, step, site in ((7, 3, 5), (1, 0, 0), (99, 500, 2), (12345, 77777, 13), (3, 9, 27), (8, 1, 6)):
        kk = k if (seed, step, site) == (7, 3, 5) else rng_ref.keep_mask(seed, step, site, 0.2, 256 * 8192).astype(np.float64)
        ratios.append(kk.reshape(-1, 256).mean(0).var() / (0.2 * 0.8 / 8192))
    assert all(0.7 < r < 1.35 for r in ratios) and 0.9 < sum(ratios) / len(ratios) < 1.1, ratios


# ------------------------------------------------------------------------------------------------ data side
def test_get_train_val_data_writes_the_reference_files(golden_dir, tmp_path, capsys):
    """tests/golden/corpus_fixture.pt holds what the REFERENCE's get_train_val_data wrote for corpus_fixture.txt
    (oracle/make_golden.py ran it, ref: src/preprocessing.py:48-86): same tokens, same split, int64, bare tensors."""
    fix = torch.load(os.path.join(golden_dir, "corpus_fixture.pt"), weights_only=True)
    src = os.path.join(golden_dir, "corpus_fixture.txt")
    tp, vp = str(tmp_path / "train_data.pt"), str(tmp_path / "val_data.pt")
    tr, va, vocab = preprocessing.get_train_val_data(src, tp, vp)
    out = capsys.readouterr().out
    assert f"Vocab size of the text: {fix['vocab_size']}" in out and "Input (decoded):" in out
    lt, lv = preprocessing.load_train_val_data(tp, vp)
    for got, mem, want in ((lt, tr, fix["train"]), (lv, va, fix["val"])):
        assert got.dtype == torch.int64 and got.dim() == 1
        assert torch.equal(got, want.to(torch.int64)) and torch.equal(mem, got)
    assert vocab == fix["vocab_size"] == len(fix["vocab"])
    # a saved split is a tensor of its own, not a view dragging the whole corpus along
    assert os.path.getsize(vp) < 8 * len(lv) + 4096
    # the oracle's restatement agrees, and the mapper round-trips the text
    text = open(src, encoding="utf-8").read()
    ot, ov, ovocab, chars = R.encode_corpus(text)
    assert torch.equal(ot, lt) and torch.equal(ov, lv) and ovocab == vocab and "".join(chars) == fix["vocab"]
    enc, dec, _ = preprocessing.get_mapper(text)
    assert dec(lt[:300].tolist()) == text[:300] and enc(text[-40:]) == lv[-40:].tolist()


def test_load_train_val_data_rejects_other_formats(tmp_path):
    good, bad = str(tmp_path / "a.pt"), str(tmp_path / "b.pt")
    torch.save(torch.arange(10), good)
    torch.save(torch.arange(10, dtype=torch.int32), bad)
    with pytest.raises(ValueError, match="int64"):
        preprocessing.load_train_val_data(good, bad)
    torch.save({"data": torch.arange(10)}, bad)
    with pytest.raises(ValueError, match="1-D int64"):
        preprocessing.load_train_val_data(bad, good)


@pytest.mark.parametrize("iters,interval,world", [(10, 5, 1), (13, 5, 1), (7, 3, 2), (4, 10, 1)])
def test_staged_offsets_equal_the_reference_draw_order(iters, interval, world):
    """engine_loop stages the window offsets of a whole evaluation interval at once; the draws must be the ones the
    reference's per-step loop makes, with evaluate_loss's draws from the same global generator in between
    (ref: src/train.py:141-172, src/preprocessing.py:43), also when iters is not a multiple of the interval."""
    n_train, T, B, eval_iters = 5000, 8, 4, 3

    class FakeEngine:
        def __init__(self, rank):
            self.rank, self.seen, self.block, self.at = rank, [], None, 0

        def stage_offsets(self, block):             # what TrainEngine.stage_offsets promises: step k takes row k of the block
            assert block.dim() == 2 and block.shape[1] == B
            assert block.min() >= 0 and block.max() + T + 1 <= n_train
            self.block, self.at = block.clone(), 0

        def step(self):
            self.seen.append(self.block[self.at])
            self.at += 1

        def check_status(self):                      # engine_loop's end-of-run check of the dW hand-over error word
            self.checked = True

    def eval_draws(gen, sink):
        for _ in range(2 * eval_iters):                               # train, then val: evaluate_loss's draws
            sink.append(preprocessing.draw_offsets(n_train, T, B, gen))

    for rank in range(world):
        gen = torch.Generator().manual_seed(42)
        eng, ev = FakeEngine(rank), []
        train.engine_loop(eng, n_train, T, B, rank, world, iters, interval, lambda it: eval_draws(gen, ev), "cpu", generator=gen)
        # the reference order, one draw per step
        gen2 = torch.Generator().manual_seed(42)
        want, ev2 = [], []
        for it in range(iters):
            ix = torch.randint(n_train - T, (B * world,), generator=gen2)
            want.append(ix[rank * B:(rank + 1) * B])
            if (it + 1) % interval == 0:
                eval_draws(gen2, ev2)
        assert len(eng.seen) == iters and all(torch.equal(a, b) for a, b in zip(eng.seen, want)) and eng.checked
        assert len(ev) == len(ev2) and all(torch.equal(a, b) for a, b in zip(ev, ev2))
        assert torch.equal(torch.randint(100, (4,), generator=gen), torch.randint(100, (4,), generator=gen2))


def test_check_ids_raises_like_torch():
    from drakegpt_amd import ops
    ops.check_ids(torch.tensor([[0, 79]]), 80, "idx")
    for bad in ([[0, 80]], [[-1, 3]]):
        with pytest.raises(IndexError, match="out of range"):
            ops.check_ids(torch.tensor(bad), 80, "idx")
    ops.check_ids(torch.zeros((0,), dtype=torch.long), 80, "idx")
