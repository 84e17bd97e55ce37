"""Hyper-parameter presets. ``PARAMS`` / ``SCALE_PARAMS`` / ``TRAIN`` carry the reference's values
(src/config.py:14-44); ``GPT2_SMALL`` / ``GPT2_MEDIUM`` are the larger shapes BASELINE.json names.
Unlike the reference's train loop, which keeps drawing batches and learning rates from ``PARAMS``
even under ``--scale`` (src/train.py:121-126,143), this package uses the selected
preset everywhere -- the README's stated intent."""

PARAMS = {
    "context_length": 8, "batch_size": 32, "base_lr": 1e-3, "max_lr": 5e-3, "betas": (0.9, 0.95),
    "embedding_dim": 32, "head_size": 32, "num_heads": 4, "num_layers": 3, "dropout": 0.1,
}
SCALE_PARAMS = {
    "context_length": 256, "batch_size": 64, "base_lr": 3e-4, "max_lr": 6e-4, "betas": (0.9, 0.95),
    "embedding_dim": 384, "head_size": 64, "num_heads": 6, "num_layers": 6, "dropout": 0.2,
}
GPT2_SMALL = {
    "context_length": 1024, "batch_size": 16, "base_lr": 3e-4, "max_lr": 6e-4, "betas": (0.9, 0.95),
    "embedding_dim": 768, "head_size": 64, "num_heads": 12, "num_layers": 12, "dropout": 0.1, "vocab_size": 50257,
}
GPT2_MEDIUM = {
    "context_length": 1024, "batch_size": 8, "base_lr": 3e-4, "max_lr": 6e-4, "betas": (0.9, 0.95),
    "embedding_dim": 1024, "head_size": 64, "num_heads": 16, "num_layers": 24, "dropout": 0.1, "vocab_size": 50257,
}
TRAIN = {"iters": 10000, "eval_iters": 200, "eval_interval": 500}
DRAKE_VOCAB_SIZE = 80     # model/*.pt: token_embedding_table.weight has 80 rows

PRESETS = {"tiny": PARAMS, "scaled": SCALE_PARAMS, "gpt2_small": GPT2_SMALL, "gpt2_medium": GPT2_MEDIUM}
