"""Training harness -- the reference's src/train.py on the HIP path.

    python -m drakegpt_amd.train --model TransformerLM --scale --data path/to/text.txt [--iters N]
    python -m torch.distributed.run --nproc-per-node 8 -m drakegpt_amd.train --model TransformerLM --scale ...

Data parallel semantics (weak scaling, as BASELINE.json configs[3] "global batch = 8 x local"): every rank trains the preset's
batch_size rows, the global batch is batch_size * world_size drawn by ONE seeded CPU generator (every rank draws all of it and
keeps its rows), the gradient is the mean over the global batch, and the learning rate is NOT rescaled -- the reference has a
single fixed batch of batch_size rows, so a world_size > 1 run is a different (larger-batch) optimisation problem by design.

Kept from the reference (src/train.py): build_model's per-model constructor arguments (:31-57),
evaluate_loss (eval mode, mean of eval_iters batch losses on train and val, :61-75), get_model_path
naming (:77-83), AdamW(lr=base_lr, betas) (:121), CyclicLR(base_lr, max_lr, step_size_up=5,
triangular) stepped once per evaluation (:122-126,162), seed 42 (:86), the step order
forward -> zero_grad -> backward -> step (:146-151), the final 100-token sample (:174-178) and the
state_dict checkpoint (:181-183).  Differences, on purpose: the selected preset
is used everywhere (batch shape and learning rates too), flags are real booleans, wandb is replaced
by JSON lines on stdout, and without --data a synthetic uniform char corpus stands in for the
Kaggle download.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import time
from typing import Optional

import torch

from . import dist as ddist
from .config import DRAKE_VOCAB_SIZE, PARAMS, PRESETS, SCALE_PARAMS, TRAIN
from .model import MODEL_CLASSES, model_params
from .preprocessing import draw_offsets, encode_text, get_mapper, load_train_val_data, split_train_val


def build_model(model_name: str, scale: bool, params: dict, scale_params: dict, vocab_size: int, device, precision: str = "fp32"):
    """ref: src/train.py:16-59 -- returns (model, model_config, params)."""
    if scale:
        params = scale_params
    if model_name not in MODEL_CLASSES:
        raise KeyError(f"unknown model {model_name!r}; choose from {list(MODEL_CLASSES)}")
    C, T = params["embedding_dim"], params["context_length"]
    cfg = {
        "BigramLM": dict(vocab_size=vocab_size),
        "SingleHeadAttentionLM": dict(vocab_size=vocab_size, embedding_dim=C, context_length=T, head_size=params["head_size"]),
        "MultiHeadAttentionLM": dict(vocab_size=vocab_size, embedding_dim=C, context_length=T, head_size=params["head_size"],
                                     num_heads=params["num_heads"]),
        "BlocksLM": dict(vocab_size=vocab_size, embedding_dim=C, context_length=T, num_heads=params["num_heads"],
                         num_layers=params["num_layers"]),
        "ResidualBlocksLM": dict(vocab_size=vocab_size, embedding_dim=C, context_length=T, num_heads=params["num_heads"],
                                 num_layers=params["num_layers"]),
        "TransformerLM": dict(vocab_size=vocab_size, embedding_dim=C, context_length=T, num_heads=params["num_heads"],
                              num_layers=params["num_layers"], dropout=params["dropout"]),
    }[model_name]
    model = MODEL_CLASSES[model_name](**cfg, precision=precision).to(device)
    return model, cfg, params


def get_model_path(dir, model_name: str, scale: bool) -> str:
    """ref: src/train.py:77-83"""
    return os.path.join(dir, f"{model_name}_scaled.pt" if scale else f"{model_name}.pt")


def cyclic_lr(step_count: int, base_lr: float, max_lr: float, step_size_up: int = 5) -> float:
    """torch CyclicLR(mode='triangular', cycle_momentum=False) after `step_count` scheduler steps"""
    total = 2.0 * step_size_up
    cycle = math.floor(1 + step_count / total)
    x = 1.0 + step_count / total - cycle
    ratio = step_size_up / total
    scale = x / ratio if x <= ratio else (x - 1) / (ratio - 1)
    return base_lr + (max_lr - base_lr) * scale


@torch.no_grad()
def evaluate_loss(train_data, val_data, model, eval_iters, context_length, batch_size, device, engine=None, generator=None):
    """ref: src/train.py:61-75.  `model` must be in eval mode; batches are drawn as get_batch does."""
    from . import ops
    out = {}
    for name, data in (("train", train_data), ("val", val_data)):
        if engine is not None and data.is_cuda:
            # same draws in the same order as the loop below, staged once; the engine replays a captured forward per batch
            offs = torch.stack([draw_offsets(len(data), context_length, batch_size, generator) for _ in range(eval_iters)])
            out[name] = engine.eval_losses(data, offs.to(device)).mean().cpu()
            continue
        losses = torch.zeros(eval_iters)
        for it in range(eval_iters):
            ix = draw_offsets(len(data), context_length, batch_size, generator).to(device)
            x, y = ops.batch_gather(data, ix, context_length)
            loss = engine.eval_loss(x, y) if engine is not None else model(x, y)[1]
            losses[it] = loss.item()
        out[name] = losses.mean()
    return out


def engine_loop(engine, n_train: int, T: int, B: int, rank: int, world: int, iters: int, eval_interval: int, on_eval, device,
                generator: Optional[torch.Generator] = None) -> None:
    """The training iterations of ref: src/train.py:141-172 on the engine path.  The reference draws one randint(len(data) - T,
    (B,)) per step from the global CPU generator and, every eval_interval steps, 2 * eval_iters more inside evaluate_loss --
    the SAME generator.  Here the offsets of all steps up to the next evaluation are drawn in one go (same draws, same order:
    a stage never crosses an evaluation) and staged in HBM once (TrainEngine.stage_offsets): the captured step finds its own row
    through the device-side step counter, so a step is one graph launch and nothing else.
    `on_eval(it)` runs after step `it` when (it + 1) % eval_interval == 0."""
    for it in range(iters):
        if it % eval_interval == 0:
            n = min(eval_interval, iters - it)
            engine.stage_offsets(torch.stack([ddist.shard_rows(draw_offsets(n_train, T, B * world, generator), rank, world)
                                              for _ in range(n)]))
        engine.step()
        if (it + 1) % eval_interval == 0:
            on_eval(it)
    engine.check_status()         # end of the run: a timed-out dW hand-over must not end in a saved checkpoint


def main(argv=None):
    ap = argparse.ArgumentParser(description="Train a DrakeGPT language model on MI355X")
    ap.add_argument("--model", default="TransformerLM", choices=list(MODEL_CLASSES))
    ap.add_argument("--scale", action="store_true", help="use SCALE_PARAMS (ref: --scale True)")
    ap.add_argument("--preset", default=None, choices=list(PRESETS), help="overrides --scale.  Under torch.distributed.run the "
                    "preset's batch_size is PER RANK (global batch = batch_size * world_size, learning rate unchanged)")
    ap.add_argument("--no-save", action="store_true")
    ap.add_argument("--data", default=None, help="UTF-8 text file (the reference's data/input.txt): tokenised and split 90/10 on "
                    "the fly, or -- with --train-data/--val-data -- only the source of the char mapper (ref: src/train.py:94-100)")
    ap.add_argument("--train-data", default=None, help="train_data.pt written by drakegpt_amd.preprocessing.get_train_val_data "
                    "(or the reference's src/preprocessing.py): 1-D int64 token tensor")
    ap.add_argument("--val-data", default=None, help="val_data.pt, same format")
    ap.add_argument("--iters", type=int, default=TRAIN["iters"])
    ap.add_argument("--eval-interval", type=int, default=TRAIN["eval_interval"])
    ap.add_argument("--eval-iters", type=int, default=TRAIN["eval_iters"])
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "fp8"])
    ap.add_argument("--model-dir", default="model")
    ap.add_argument("--sample", type=int, default=100)
    args = ap.parse_args(argv)

    torch.manual_seed(42)
    if not torch.cuda.is_available():
        raise SystemExit("drakegpt_amd.train needs an MI355X (no CPU path)")
    rank, local_rank, world = ddist.env_world()
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    pg = ddist.init("nccl", device)

    if bool(args.train_data) != bool(args.val_data):
        raise SystemExit("--train-data and --val-data go together")
    decode = lambda ids: " ".join(str(i) for i in ids)      # noqa: E731
    if args.train_data:
        # the reference's flow (src/train.py:94-100): token streams from the .pt files, the mapper from the text
        train_data, val_data = load_train_val_data(args.train_data, args.val_data)
        if args.data:
            with open(args.data, "r", encoding="utf-8") as f:
                _, decode, vocab_size = get_mapper(f.read())
        else:
            vocab_size = int(max(train_data.max(), val_data.max())) + 1
    elif args.data:
        with open(args.data, "r", encoding="utf-8") as f:
            text = f.read()
        data, decode, vocab_size = encode_text(text)
        train_data, val_data = split_train_val(data)
    else:
        vocab_size = DRAKE_VOCAB_SIZE
        data = torch.randint(0, vocab_size, (1_000_000,), generator=torch.Generator().manual_seed(42))
        train_data, val_data = split_train_val(data)
    train_dev, val_dev = train_data.to(device), val_data.to(device)

    params = PRESETS[args.preset] if args.preset else (SCALE_PARAMS if args.scale else PARAMS)
    vocab_size = params.get("vocab_size", vocab_size)
    model, model_config, params = build_model(args.model, False, params, params, vocab_size, device, args.precision)
    if rank == 0:
        print(f"Selected {args.model} model for training. Model has {model_params(params, args.model, vocab_size)} parameters "
              f"(reference estimate; actual {sum(p.numel() for p in model.parameters())}).")
    B, T = params["batch_size"], params["context_length"]
    base_lr, max_lr = params["base_lr"], params["max_lr"]

    engine = None
    if args.model == "TransformerLM":
        from .engine import TrainEngine
        engine = TrainEngine(model, B, T, lr=base_lr, betas=params["betas"], seed=42, rank=rank, world_size=world, process_group=pg)
        engine.set_corpus(train_dev)
    else:
        # the five earlier-stage models train through the autograd path; their flat-buffer AdamW all-reduces the gradient
        from .optim import AdamW
        optimizer = AdamW(model.parameters(), lr=base_lr, betas=params["betas"], process_group=pg, world_size=world)

    model.train()
    sched = {"steps": 0}
    t0 = time.perf_counter()

    def on_eval(it):
        model.eval()
        if engine is not None:
            # the evaluation synchronises the host for its losses anyway: the one place inside the loop where reading the
            # grouped dW GEMM's sticky error word costs nothing.  Raises (-> non-zero exit) if a hand-over ever timed out:
            # every weight gradient since then is suspect, and the losses would still look plausible.
            engine.check_status()
        losses = evaluate_loss(train_dev, val_dev, model, args.eval_iters, T, B, device, engine=engine)
        sched["steps"] += 1
        lr = cyclic_lr(sched["steps"], base_lr, max_lr)
        if engine is not None:
            engine.set_lr(lr)
        else:
            for g in optimizer.param_groups:
                g["lr"] = lr
        if rank == 0:
            el = time.perf_counter() - t0
            print(json.dumps({"step": it + 1, "train_loss": float(losses["train"]), "val_loss": float(losses["val"]), "lr": lr,
                              "tokens_per_s": (it + 1) * B * T * world / el}), flush=True)
        model.train()

    if engine is not None:
        engine_loop(engine, len(train_data), T, B, rank, world, args.iters, args.eval_interval, on_eval, device)
    else:
        from . import ops
        for it in range(args.iters):
            ix = ddist.shard_rows(draw_offsets(len(train_data), T, B * world, None), rank, world).to(device, non_blocking=True)
            x, y = ops.batch_gather(train_dev, ix, T)
            logits, loss = model(x, y)
            optimizer.zero_grad()
            loss.backward()
            optimizer.step()
            if (it + 1) % args.eval_interval == 0:
                on_eval(it)

    model.eval()
    if rank == 0:
        idx = torch.zeros((1, 1), dtype=torch.long, device=device)
        print(decode(model.generate(idx, max_new_tokens=args.sample)[0].tolist()))
        if not args.no_save:
            os.makedirs(args.model_dir, exist_ok=True)
            path = get_model_path(args.model_dir, args.model, args.scale)
            torch.save({k: v.detach().cpu().clone() for k, v in model.state_dict().items()}, path)
            print(f"saved {path}")
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
