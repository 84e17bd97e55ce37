#!/usr/bin/env python3
"""Headline benchmark: training tokens/s of TransformerLM_scaled (fwd + bwd + AdamW) on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config scaled] [--batch B] [--precision bf16]

N > 1 is launched by the driver as `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`
(one process per GPU, RCCL): weak scaling, every rank trains B_local rows of the same global batch and
the flat gradient is all-reduced once per step.  Rank 0 prints ONE JSON line.

Workload (BASELINE.json configs[1]): TransformerLM_scaled = V 80, C 384, T 256, 6 heads, 6 blocks,
dropout 0.2 (src/config.py:27-38), batch 64 per GPU, synthetic uniform char corpus of 1e6 tokens
(seed 42), random-init weights (torch default initialisers, seed 42), bf16 MFMA operands with
fp32 accumulation / master weights.  Inputs (corpus, window offsets) are resident in HBM when the
timed region starts.

Beside the headline line's fields the JSON carries (N = 1 only, timed AFTER the headline region, each with its own engine):
`extra_configs` -- the same model at dropout 0 and at the largest batch (B = 256), and the GPT-2-small shape (BASELINE.json
configs[2]: V 50257, C 768, T 1024, 12 x 12, B = 8), each as tokens/s + fraction of the bf16 MFMA peak; `hbm_kernels` -- the
HBM-bound kernels of the step (LayerNorm fwd / bwd, cross entropy, AdamW, embedding fwd / bwd, batch gather) as achieved GB/s
= algorithmic bytes / launch duration (HIP events, back to back), SURVEY.md section 8d.

roofline: the MFMA-bound kernel symbol that takes the most time in the step; `achieved` is its
algorithmic FLOP per launch / its average launch duration, both taken live with HIP events on the
launch stream in an eager (un-captured) replay of the same step.  cpu_baseline: the CPU oracle
(oracle/drake_ref.py, bit-identical to the reference) timed on this host on a bounded sample.
"""
from __future__ import annotations

import argparse
import json
import os

# multi-process GPU work on this pool needs dmabuf IPC (the host driver has no legacy IPC: RCCL fails with
# hipIpcGetMemHandle: invalid argument otherwise); the driver's environment exports it, this is the belt to its braces
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_BF16_TFLOPS = 2500.0      # MI355X dense bf16 MFMA peak (guides/MI355X_MICROARCH.md: ~2.5 PF)
PEAK_FP8_TFLOPS = 5000.0       # dense block-scaled fp8 MFMA peak (same guide: ~5 PF)
PEAK_F32_TFLOPS = 157.3        # fp32 MFMA = vector peak
PEAK_HBM_GBS = 8000.0


def flops_per_token(cfg, V):
    """algorithmic FLOP of one training step per token, causal attention counted at the work
    actually required: F_step = 3 * [L (24 C^2 + 2 (T+1) C) + 2 C V]"""
    C, T, L = cfg["embedding_dim"], cfg["context_length"], cfg["num_layers"]
    return 3 * (L * (24 * C * C + 2 * (T + 1) * C) + 2 * C * V)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="scaled")
    ap.add_argument("--batch", type=int, default=None, help="rows per GPU (default: the preset's batch_size)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32", "fp8"])
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra_configs leg (p = 0, B = 256, GPT-2-small)")
    ap.add_argument("--dropout", type=float, default=None, help="override the preset's dropout")
    ap.add_argument("--dp-buckets", type=int, default=None, help="N > 1: layer groups whose gradient ranges are all-reduced while the "
                    "next group's backward runs (default: by gradient size, see TrainEngine._choose_buckets)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N > 1 must be launched through torch.distributed.run (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; there is no CPU path")
    # rehearsal hook for a 1-GPU box: DG_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and moves the gradient
    # over gloo (RCCL refuses two ranks on one device); the driver's multi-GPU runs use one GPU per rank + RCCL
    share = os.environ.get("DG_BENCH_SHARE_GPU") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    pg = None
    # under torch.distributed.run the process group exists at every N, N = 1 included: the 1-GPU rehearsal of the driver's SCALE
    # entry then covers the RCCL bring-up (communicator, barrier, MAX all-reduce of the time, teardown) it will meet at N = 8
    launched = world > 1 or ("TORCHELASTIC_RUN_ID" in os.environ and "MASTER_PORT" in os.environ)
    if launched:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
        pg = dist.group.WORLD if world > 1 else None

    import drakegpt_amd as D
    from drakegpt_amd.config import DRAKE_VOCAB_SIZE, PRESETS
    from drakegpt_amd.engine import TrainEngine

    cfg = dict(PRESETS[args.config])
    if args.dropout is not None:
        cfg["dropout"] = args.dropout
    V = cfg.get("vocab_size", DRAKE_VOCAB_SIZE)
    B = args.batch or cfg["batch_size"]
    T = cfg["context_length"]
    torch.manual_seed(42)
    model = D.TransformerLM(V, cfg["embedding_dim"], T, cfg["num_heads"], cfg["num_layers"], cfg["dropout"],
                            precision=args.precision).to(dev)
    n_params = sum(p.numel() for p in model.parameters())
    eng = TrainEngine(model, B, T, lr=cfg["base_lr"], betas=cfg["betas"], seed=42, rank=rank, world_size=world,
                      process_group=pg, use_graph=not args.no_graph, dp_buckets=args.dp_buckets)
    dp_buckets = eng.dp_buckets
    n_corpus = 1_000_000 if V <= 256 else 10_000_000
    corpus = torch.randint(0, V, (n_corpus,), generator=torch.Generator().manual_seed(42))
    eng.set_corpus(corpus)
    # window offsets: the global batch is drawn by the host CPU generator exactly as get_batch does
    # (src/preprocessing.py:43); rank r takes rows [r*B, (r+1)*B).  Staged in HBM before timing.
    gen = torch.Generator().manual_seed(42)
    total = args.warmup + args.steps
    offs = torch.stack([torch.randint(n_corpus - T, (B * world,), generator=gen)[rank * B:(rank + 1) * B] for _ in range(total)])
    offs = offs.to(dev)

    def barrier():
        if launched:
            import torch.distributed as dist
            if share:
                dist.barrier()
            else:
                dist.barrier(device_ids=[local_rank])
        torch.cuda.synchronize(dev)

    if rank == 0:
        log(f"warm-up ({args.warmup} steps, graph capture on the first)")
    eng.stage_offsets(offs)           # one block for all steps: a step then is a graph launch (the step reads its own row)
    for i in range(args.warmup):
        eng.step()
    barrier()
    if rank == 0:
        log(f"timing {args.steps} steps")
    t0 = time.perf_counter()
    for i in range(args.warmup, total):
        eng.step()
    barrier()
    dt = time.perf_counter() - t0
    if launched:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    final_loss = eng.loss.item()
    eng.check_status()          # raises (non-zero exit) if a dW hand-over of any timed step ran out: such a number must not be reported
    tokens = args.steps * B * T * world
    tok_s = tokens / dt
    fpt = flops_per_token(cfg, V)
    peak = {"bf16": PEAK_BF16_TFLOPS, "fp8": PEAK_FP8_TFLOPS, "fp32": PEAK_F32_TFLOPS}[args.precision]

    if rank == 0:
        log(f"{tok_s:.0f} tokens/s, {1e3 * dt / args.steps:.3f} ms/step, loss {final_loss:.4f}")
    roofline = hbm_kernels = None
    if rank == 0 and not args.no_kernel_timing:
        roofline, hbm_kernels = kernel_roofline(eng, offs[0], peak)
    extra = None
    if rank == 0 and world == 1 and not args.no_extra and args.config == "scaled" and args.precision == "bf16":
        del eng, model
        torch.cuda.empty_cache()
        extra = {}
        # scaled_fp32: the exact-fp32 MFMA parity mode -- the mode that meets the north star's "logits within 1e-3 rel" (2e-7 measured)
        for name, cname, b, p_drop, st, wu, prec in (("scaled_dropout0", "scaled", B, 0.0, 30, 5, "bf16"), ("scaled_B256", "scaled", 256, None, 20, 5, "bf16"),
                                                     ("scaled_fp32", "scaled", B, None, 8, 3, "fp32"),
                                                     ("gpt2_small_B8", "gpt2_small", 8, None, 8, 3, "bf16"),
                                                     ("gpt2_small_B16", "gpt2_small", 16, None, 5, 2, "bf16"),
                                                     ("gpt2_medium_B8_bf16", "gpt2_medium", 8, None, 5, 2, "bf16"),
                                                     ("gpt2_medium_B8_fp8", "gpt2_medium", 8, None, 5, 2, "fp8"),
                                                     # BASELINE configs[4] names no batch: 288 GB hold far more than 8 sequences, and the fp8
                                                     # path pays more the larger the GEMMs (same box: 1.21x at B = 8, 1.25x at 16, 1.29x at 32)
                                                     ("gpt2_medium_B32_bf16", "gpt2_medium", 32, None, 3, 2, "bf16"),
                                                     ("gpt2_medium_B32_fp8", "gpt2_medium", 32, None, 3, 2, "fp8")):
            try:
                extra[name] = run_extra(cname, b, p_drop, st, wu, dev, prec)
                log(f"extra {name}: {extra[name]['value']:.0f} tokens/s, {extra[name]['ms_per_step']:.3f} ms/step")
            except Exception as e:                     # an extra line must never take the headline line down with it
                extra[name] = {"error": f"{type(e).__name__}: {e}"}
                log(f"extra {name} failed: {extra[name]['error']}")
            torch.cuda.empty_cache()
    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = cpu_baseline_leg(args.config, cfg, V)

    if rank == 0:
        out = {
            "metric": "training tokens/sec (fwd+bwd+AdamW)", "value": tok_s, "unit": "tokens/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"bf16": "bf16", "fp8": "fp8", "fp32": "f32"}[args.precision], "data": "synthetic",
            "config": {"workload": f"TransformerLM_{args.config}: V={V} C={cfg['embedding_dim']} T={T} heads={cfg['num_heads']} "
                                   f"layers={cfg['num_layers']} dropout={cfg['dropout']}; fwd+bwd+AdamW; hipGraph={'off' if args.no_graph else 'on'}",
                       "batch_per_gpu": B, "global_batch": B * world, "seq_len": T, "parallelism": f"dp{world}", "dp_gradient_buckets": dp_buckets,
                       "params": n_params},
            "model_flops_per_token": fpt,
            "achieved_tflops_per_gpu": tok_s * fpt / 1e12 / world,
            "mfma_peak_frac_whole_step": tok_s * fpt / 1e12 / world / peak,
            "final_loss": final_loss,
            "roofline": roofline,
            "cpu_baseline": cpu_baseline,
            "hbm_kernels": hbm_kernels,
            "extra_configs": extra,
        }
        print(json.dumps(out), flush=True)
    if launched:
        import torch.distributed as dist
        barrier()                      # rank 0 did the kernel-timing leg: leave together
        dist.destroy_process_group()


def run_extra(config_name, B, dropout, steps, warmup, dev, precision="bf16"):
    """one more single-GPU configuration, timed like the headline region (offsets resident, graph replay, synchronize on
    both sides); synthetic corpus, random-init weights (seed 42).  precision "fp8": the residual blocks' Linears (forward and
    dX) on e4m3 / e5m2 operands, everything else as "bf16"; its MFMA fraction is quoted against the fp8 peak."""
    import drakegpt_amd as D
    from drakegpt_amd.config import DRAKE_VOCAB_SIZE, PRESETS
    from drakegpt_amd.engine import TrainEngine
    cfg = dict(PRESETS[config_name])
    if dropout is not None:
        cfg["dropout"] = dropout
    V = cfg.get("vocab_size", DRAKE_VOCAB_SIZE)
    T = cfg["context_length"]
    torch.manual_seed(42)
    model = D.TransformerLM(V, cfg["embedding_dim"], T, cfg["num_heads"], cfg["num_layers"], cfg["dropout"], precision=precision).to(dev)
    eng = TrainEngine(model, B, T, lr=cfg["base_lr"], betas=cfg["betas"], seed=42)
    n_corpus = 1_000_000 if V <= 256 else 10_000_000
    eng.set_corpus(torch.randint(0, V, (n_corpus,), generator=torch.Generator().manual_seed(42)))
    gen = torch.Generator().manual_seed(42)
    offs = torch.stack([torch.randint(n_corpus - T, (B,), generator=gen) for _ in range(warmup + steps)]).to(dev)
    eng.stage_offsets(offs)
    for i in range(warmup):
        eng.step()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for i in range(warmup, warmup + steps):
        eng.step()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    tok_s = steps * B * T / dt
    fpt = flops_per_token(cfg, V)
    peak = {"bf16": PEAK_BF16_TFLOPS, "fp8": PEAK_FP8_TFLOPS, "fp32": PEAK_F32_TFLOPS}[precision]
    loss = eng.loss.item()
    eng.check_status()
    del eng, model
    return {"value": tok_s, "unit": "tokens/s", "ms_per_step": 1e3 * dt / steps, "steps": steps, "warmup": warmup,
            "workload": f"TransformerLM_{config_name}: V={V} C={cfg['embedding_dim']} T={T} heads={cfg['num_heads']} layers={cfg['num_layers']} "
                        f"dropout={cfg['dropout']} batch={B}; fwd+bwd+AdamW; {precision}; hipGraph=on",
            "dtype": precision, "model_flops_per_token": fpt, "achieved_tflops": tok_s * fpt / 1e12,
            "peak_tflops": peak, "mfma_peak_frac_whole_step": tok_s * fpt / 1e12 / peak, "final_loss": loss}


def kernel_roofline(eng, offsets, peak_tflops):
    """Live roofline of the dominant MFMA kernel.  One eager step records every GEMM launch of the step
    (operands kept alive); each recorded launch is then replayed back to back between two HIP events on
    the launch stream (torch's current stream = the stream the C ABI enqueues on), which is how the
    launches sit inside the captured graph.  achieved = algorithmic FLOP of the symbol's launches /
    their summed duration; the rocprofv3 summary of the same command is under profiles/."""
    from drakegpt_amd import ops

    calls = []            # (symbol, flops, closure)
    real_nt, real_tn, real_tng = ops.gemm_nt, ops.gemm_tn, ops.gemm_tn_grouped
    ncu = torch.cuda.get_device_properties(0).multi_processor_count

    def nt(A, Bm, out_dtype, **kw):
        M = A.shape[0]
        K = kw.get("K") or A.shape[1]
        N = kw.get("N") or Bm.shape[0]
        r = real_nt(A, Bm, out_dtype, **kw)
        to = "bf16" if out_dtype == torch.bfloat16 else "float"
        f8 = A.dtype in (torch.float8_e4m3fn, torch.float8_e5m2)
        if f8 or (A.dtype == torch.bfloat16 and K % 64 == 0 and K >= 128):        # dispatch rule of dg_gemm_nt
            pf = kw.get("sign_bits") is not None                           # mask-bit prefetch variant
            nj = 6 if N % 192 == 0 else 4                                  # 128 x 192 tiles when they divide N
            has = lambda k: kw.get(k) is not None and kw.get(k) is not False
            drop = kw.get("dropout_p", 0.0) > 0.0 and kw.get("rng_state") is not None
            opts = [has("bias"), has("relu"), has("relu_mask"), drop, has("residual"), has("sign_bits"), has("sign_bits_out")]
            if not any(opts): epi = 1
            elif to == "bf16" and opts == [True, True, False, False, False, False, True]: epi = 2
            elif opts == [True, False, False, True, True, False, False]: epi = 3
            elif to == "bf16" and opts == [False, False, False, False, False, True, False]: epi = 6 if has("colsum_part") else 4
            elif opts == [True, False, False, False, False, False, False]: epi = 5
            elif opts == [True, False, False, False, True, False, False]: epi = 7
            else: epi = 0
            f8code = 0 if not f8 else (2 if A.dtype == torch.float8_e5m2 else 1)
            sym = f"gemm_nt_ws_kernel<{to},{'true' if pf else 'false'},{nj},{epi},{f8code}>"      # dispatch rule of dg_gemm_nt
        else:
            sym = f"gemm_nt_kernel<{'bf16' if A.dtype == torch.bfloat16 else 'float'},{to}>"
        kw2 = dict(kw)
        kw2["out"] = r
        calls.append((sym, 2.0 * M * N * K, lambda: real_nt(A, Bm, out_dtype, **kw2)))
        return r

    def tn(A, Bm, out_part, split_stride, n_splits, P, Q, ldo=None):
        real_tn(A, Bm, out_part, split_stride, n_splits, P, Q, ldo)
        tiles = ((P + 127) // 128) * ((Q + 127) // 128)
        if A.dtype == torch.bfloat16:                                          # dispatch rule of dg_gemm_tn
            sym = "gemm_tn_ws_kernel" if (A.shape[0] % 64 == 0 and tiles * n_splits <= ncu) else "gemm_tn_bf16_kernel"
        else:
            sym = "gemm_tn_f32_kernel"
        calls.append((sym, 2.0 * A.shape[0] * P * Q, lambda: real_tn(A, Bm, out_part, split_stride, n_splits, P, Q, ldo)))

    def tng(problems, workspace=None):
        problems = list(problems)
        real_tng(problems, workspace)
        fl = sum(2.0 * pr[0].shape[0] * pr[3] * pr[4] for pr in problems)
        f8 = "true" if len(problems[0]) == 7 else "false"           # fp8 operands (precision fp8): (A, B, out, P, Q, scale_a, scale_b)
        sym = "gemm_tn_grouped_kernel" if os.environ.get("DG_TN_TILE") == "128" else f"gemm_tn_grouped256_kernel<{1 if os.environ.get('DG_TN_WAVETILE') == '1' else 0},{f8}>"
        calls.append((sym, fl, lambda: real_tng(problems, workspace)))

    # the row-local chain of a residual block as one launch (round 3: the forward default at C = 384): its GEMM pieces' FLOP
    real_chain = ops.block_chain_fwd

    def chain(mode, M, Cd, **kw):
        r = real_chain(mode, M, Cd, **kw)
        per = {0: 24.0, 1: 18.0, 2: 6.0, 3: 2.0, 4: 8.0}[mode]          # 2 M C (C + 4C + 4C + 3C) for the whole chain
        calls.append((f"block_chain_fwd_kernel<{mode}>", per * M * Cd * Cd, lambda: real_chain(mode, M, Cd, **kw)))
        return r

    # HBM-bound kernels: algorithmic bytes per launch (what the kernel must read + write once; SURVEY 8d conventions)
    hbm = []              # (symbol, bytes, closure)
    esz = lambda t: t.element_size()
    hbm_names = ("layernorm_fwd", "layernorm_bwd_fused", "layernorm_bwd", "cross_entropy", "adamw_step", "embed_fwd", "embed_bwd",
                 "batch_gather", "batch_embed_fwd", "cross_entropy_fused")
    real_hbm = {n: getattr(ops, n) for n in hbm_names}

    def wrap(name, nbytes):
        real = real_hbm[name]

        def f(*a, **kw):
            r = real(*a, **kw)
            hbm.append((name, float(nbytes(r, *a, **kw)), lambda: real(*a, **kw)))
            return r
        return f
    wrapped = {
        # x fp32 in, y out (+ mean, rstd)
        "layernorm_fwd": wrap("layernorm_fwd", lambda r, x, g, b, od, *a, **k: x.numel() * (4 + esz(r[0])) + 8 * r[1].numel()),
        # dy, x, dresid in; dx (+ g) out
        "layernorm_bwd_fused": wrap("layernorm_bwd_fused", lambda r, dy, x, *a, **k: x.numel() * (esz(dy) + 4 + (esz(a[3]) if a[3] is not None else 0) + esz(r[0]) + esz(r[1]))),
        "layernorm_bwd": wrap("layernorm_bwd", lambda r, dy, x, g, mean, rstd, dresid, *a, **k: x.numel() * (esz(dy) + 4 + (4 if dresid is not None else 0) + 4)),
        # logits in, dlogits out
        "cross_entropy": wrap("cross_entropy", lambda r, logits, tg, V, dlogits=None, **k: logits.shape[0] * V * 4 + (dlogits.numel() * esz(dlogits) if dlogits is not None else 0)),
        # the loss head in one launch: fp32 logits in, dlogits (padded width) out; partials and shares are noise
        "cross_entropy_fused": wrap("cross_entropy_fused", lambda r, logits, tg, V, dlogits, *a, **k: logits.shape[0] * V * 4 + dlogits.numel() * esz(dlogits)),
        # p, g, m, v read; p, m, v (+ bf16 shadow) written
        "adamw_step": wrap("adamw_step", lambda r, p, g, m, v, hy, st, gs=1.0, shadow_bf16=None, n=None, advance=False: (n or p.numel()) * (28 + (2 if shadow_bf16 is not None else 0))),
        "embed_fwd": wrap("embed_fwd", lambda r, idx, tok, pos, out=None, onehot=None: r.numel() * 4 + idx.numel() * 8 + (onehot.numel() * 2 if onehot is not None else 0)),
        "embed_bwd": wrap("embed_bwd", lambda r, idx, dx, dtok, dpos, V=None: dx.numel() * esz(dx) + (dpos.numel() * 4 if dpos is not None else 0) + (dtok.numel() * 4 if dtok is not None else 0)),
        "batch_gather": wrap("batch_gather", lambda r, corpus, offsets, T, x=None, y=None: offsets.numel() * T * 32),
        # ids gathered (x and x + 1 from the corpus), ids / targets and the fp32 stream written
        "batch_embed_fwd": wrap("batch_embed_fwd", lambda r, corpus, offs, st, ctl, x, y, tok, pos, onehot=None: r.numel() * 4 + x.numel() * 32 + (onehot.numel() * 2 if onehot is not None else 0)),
    }
    eng.set_offsets(offsets)
    ops.gemm_nt, ops.gemm_tn, ops.gemm_tn_grouped, ops.block_chain_fwd = nt, tn, tng, chain
    for n_, f_ in wrapped.items():
        setattr(ops, n_, f_)
    try:
        eng._prog_fwd_bwd()
        eng._prog_update()
        torch.cuda.synchronize()
    finally:
        ops.gemm_nt, ops.gemm_tn, ops.gemm_tn_grouped, ops.block_chain_fwd = real_nt, real_tn, real_tng, real_chain
        for n_, f_ in real_hbm.items():
            setattr(ops, n_, f_)
    snap = (eng.flat.clone(), eng.m_.clone(), eng.v_.clone())       # the AdamW replays below must not train the model away
    hagg = {}
    side = torch.cuda.Stream()
    for sym, nb, fn in hbm:
        # ten launches inside one captured graph: these kernels take 8-50 us, less than an eager ctypes launch + allocation
        # costs on the host, and inside the training step they run from a graph too
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn(); fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(10):
                fn()
        gr.replay()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        gr.replay()
        e.record()
        e.synchronize()
        a = hagg.setdefault(sym, [0.0, 0.0, 0])
        a[0] += nb
        a[1] += s.elapsed_time(e) * 1e-3 / 10
        a[2] += 1
        del gr
    eng.flat.copy_(snap[0]); eng.m_.copy_(snap[1]); eng.v_.copy_(snap[2])
    eng.refresh_shadows()
    hbm_kernels = {k: {"achieved_GBs": v[0] / v[1] / 1e9, "frac_of_8TBs": v[0] / v[1] / 1e9 / PEAK_HBM_GBS, "avg_us": 1e6 * v[1] / v[2],
                       "launches_per_step": v[2], "algorithmic_bytes_per_launch": v[0] / v[2]} for k, v in hagg.items()}
    reps = 10
    agg = {}
    for sym, fl, fn in calls:
        fn(); fn()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(reps):
            fn()
        e.record()
        e.synchronize()
        a = agg.setdefault(sym, [0.0, 0.0, 0])
        a[0] += fl
        a[1] += s.elapsed_time(e) * 1e-3 / reps
        a[2] += 1
    sym, (fl, sec, n) = max(agg.items(), key=lambda kv: kv[1][1])
    achieved = fl / sec / 1e12
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")      # PMC pass (tools/collect_traffic.py), per launch
    if os.path.exists(tpath):
        t = json.load(open(tpath)).get("kernels", {}).get(sym)
        if t:
            traffic = t["hbm_bytes_per_launch"]
    return {"bound": "mfma", "kernel": sym, "achieved": achieved, "peak": peak_tflops, "unit": "TFLOP/s",
            "frac": achieved / peak_tflops, "traffic": traffic, "launches_per_step": n,
            "avg_launch_us": 1e6 * sec / n, "flops_per_launch": fl / n,
            "all_gemm_symbols": {k: {"tflops": v[0] / v[1] / 1e12, "avg_us": 1e6 * v[1] / v[2], "launches_per_step": v[2]}
                                 for k, v in agg.items()}}, hbm_kernels


def usable_cores() -> int:
    """CPU share of this process: min(affinity mask, cgroup quota, 16 -- the GPU box's share per GPU)"""
    n = os.cpu_count() or 1
    try:
        n = min(n, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return max(1, min(n, 16))


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def cpu_baseline_leg(config_name, cfg, V):
    """the CPU oracle (== the reference's arithmetic) on this host: bounded sample of the workload"""
    from oracle import drake_ref as R
    ncores = usable_cores()
    torch.set_num_threads(ncores)
    log(f"cpu baseline on {ncores} threads")
    B = cfg["batch_size"]                  # the configuration's own batch ("configs 1-2 in full")
    T = cfg["context_length"]
    ocfg = dict(cfg)
    sd = R.init_state_dict("TransformerLM", V, ocfg, seed=42)
    opt = R.AdamWState(R.trainable_keys("TransformerLM", sd), cfg["base_lr"], cfg["betas"])
    data = torch.randint(0, V, (100_000,), generator=torch.Generator().manual_seed(42))
    gen = torch.Generator().manual_seed(42)
    x, y = R.get_batch(data, T, B, gen)
    R.train_step("TransformerLM", sd, opt, x, y, p=cfg["dropout"], training=True)       # untimed warm-up
    steps = 0
    t0 = time.perf_counter()
    while True:
        x, y = R.get_batch(data, T, B, gen)
        R.train_step("TransformerLM", sd, opt, x, y, p=cfg["dropout"], training=True)
        steps += 1
        el = time.perf_counter() - t0
        log(f"cpu baseline: {steps} steps, {el:.1f} s")
        if (el > 12.0 and steps >= 2) or steps >= 200:
            break
    return {"value": steps * B * T / el, "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{steps} fwd+bwd+AdamW steps of TransformerLM_{config_name} at batch {B} x {T} tokens, fp32 torch CPU ops "
                      f"(oracle/drake_ref.py, bit-identical to the reference), {el:.1f} s"}


if __name__ == "__main__":
    main()
