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
    with pytest.raises(ValueError):
        D.TransformerLM(V, 32, 8, 4, 3, 0.1, precision="fp8")


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
    assert rng_ref.threshold(0.0) == 0 and rng_ref.threshold(0.5) == 2 ** 31
    # the lean element hash behaves like independent Bernoulli draws: lag correlations ~ 1/sqrt(n),
    # row sums (256 consecutive elements = one attention row) binomially dispersed
    import numpy as np
    k = rng_ref.keep_mask(7, 3, 5, 0.2, 256 * 8192).astype(np.float64)
    for lag in (1, 2, 3, 8, 256, 257, 384):
        assert abs(np.corrcoef(k[:-lag], k[lag:])[0, 1]) < 4e-3, lag
    rows = k.reshape(-1, 256).sum(1)
    assert 0.93 < rows.var() / (256 * 0.2 * 0.8) < 1.07
