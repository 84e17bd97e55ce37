"""CPU: the oracle restatement against the committed golden fixtures (outputs of the real
reference, written by oracle/make_golden.py where /root/reference is mounted)."""
import json
import os

import pytest
import torch

from oracle import drake_ref as R

V = 80


@pytest.fixture(autouse=True)
def _one_thread():
    n = torch.get_num_threads()
    torch.set_num_threads(1)      # the fixtures were produced with a fixed reduction order
    yield
    torch.set_num_threads(n)


@pytest.mark.parametrize("name", R.MODEL_NAMES)
def test_forward_backward_bit_exact(golden_dir, name):
    sd = torch.load(os.path.join(golden_dir, "checkpoints", f"{name}.pt"), weights_only=True)
    fix = torch.load(os.path.join(golden_dir, f"fwdbwd_{name}.pt"), weights_only=True)
    logits, loss, grads = R.loss_and_grads(name, sd, fix["x"], fix["y"])
    assert torch.equal(logits, fix["logits"]) and torch.equal(loss, fix["loss"])
    for k, g in grads.items():
        assert torch.equal(g, fix["grad." + k]), k
    assert not any(k.startswith("ln_f.") for k in grads)
    assert R.lm_forward(name, sd, fix["x"])[0].shape == (4, 8, V)


@pytest.mark.parametrize("name", R.MODEL_NAMES)
def test_generate_tokens(golden_dir, name):
    gold = json.load(open(os.path.join(golden_dir, "generate.json")))
    sd = torch.load(os.path.join(golden_dir, "checkpoints", f"{name}.pt"), weights_only=True)
    torch.manual_seed(gold["seed"])
    out = R.lm_generate(name, sd, torch.zeros((1, 1), dtype=torch.long), 30)
    assert out[0].tolist() == gold["tokens"][name][:31]


def test_known_answers_from_survey(golden_dir):
    """the first 21 sampled tokens and the loss on the seeded (4,8) batch."""
    gold = json.load(open(os.path.join(golden_dir, "generate.json")))
    assert gold["tokens"]["TransformerLM"][:21] == [0, 24, 15, 32, 15, 28, 1, 11, 1, 14, 25, 22, 14, 1, 19, 1, 12, 35, 1, 30, 18]
    fix = torch.load(os.path.join(golden_dir, "fwdbwd_TransformerLM.pt"), weights_only=True)
    assert abs(fix["loss"].item() - 11.540583610534668) < 1e-6


def test_small_transformer_and_init_order(golden_dir):
    summ = json.load(open(os.path.join(golden_dir, "summary.json")))
    cfg = dict(R.TINY, **summ["small_cfg"])
    sd = R.init_state_dict("TransformerLM", V, cfg, seed=42)
    fix = torch.load(os.path.join(golden_dir, "small_TransformerLM.pt"), weights_only=True)
    for T in (1, 5, 32):
        logits, loss, grads = R.loss_and_grads("TransformerLM", sd, fix[f"T{T}.x"], fix[f"T{T}.y"])
        assert torch.equal(logits, fix[f"T{T}.logits"])
        for k, g in grads.items():
            assert torch.equal(g, fix[f"T{T}.grad.{k}"]), (T, k)


@pytest.mark.parametrize("name", ["TransformerLM", "BigramLM"])
def test_five_step_trajectory(golden_dir, name):
    fix = torch.load(os.path.join(golden_dir, f"traj5_{name}.pt"), weights_only=True)
    sd = {k: v.clone() for k, v in fix["init"].items()}
    opt = R.AdamWState(R.trainable_keys(name, sd), R.TINY["base_lr"], R.TINY["betas"])
    for it in range(5):
        loss = R.train_step(name, sd, opt, fix["x"][it], fix["y"][it], p=0.0, training=True)
        assert loss == fix["losses"][it].item()
    for k, v in fix["final"].items():
        assert (sd[k] - v).abs().max().item() <= 1e-7, k


def test_cyclic_lr_sequence(golden_dir):
    seq = json.load(open(os.path.join(golden_dir, "summary.json")))["cyclic_lr"]
    assert len(seq) == 21
    for i, lr in enumerate(seq):
        assert abs(R.cyclic_lr(i, 1e-3, 5e-3) - lr) < 1e-15
    assert abs(seq[5] - 5e-3) < 1e-12 and abs(seq[10] - 1e-3) < 1e-12


def test_explicit_mask_dropout_matches_functional_dropout():
    """drake_ref's explicit-mask path is the same arithmetic as nn.Dropout: x * keep / (1-p)"""
    x = torch.randn(4, 5)
    keep = (torch.rand(4, 5) > 0.3).float()
    assert torch.equal(R._drop(x, 0.3, True, keep), x * keep * (1.0 / 0.7))
