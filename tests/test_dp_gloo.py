"""CPU, world_size 2, gloo: the data-parallel logic -- row sharding + SUM all-reduce of the flat
gradient + 1/world scaling -- reproduces the single-process gradient of the concatenated batch.
The per-rank gradients come from the CPU oracle here (no GPU in this container); on the GPU the
same helpers move the engine's flat gradient buffer over RCCL."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from drakegpt_amd import dist as ddist
from oracle import drake_ref as R

V = 80


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, golden_dir, bucket, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    torch.set_num_threads(1)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        sd = torch.load(os.path.join(golden_dir, "checkpoints", "TransformerLM.pt"), weights_only=True)
        g = torch.Generator().manual_seed(11)
        x = torch.randint(0, V, (8, 8), generator=g)
        y = torch.randint(0, V, (8, 8), generator=g)
        xs, ys = ddist.shard_rows(x, rank, world), ddist.shard_rows(y, rank, world)
        assert xs.shape[0] == 4 and torch.equal(xs, x[rank * 4:(rank + 1) * 4])
        _, loss, grads = R.loss_and_grads("TransformerLM", sd, xs, ys)
        keys = R.trainable_keys("TransformerLM", sd)          # ln_f excluded identically on every rank
        flat = ddist.flatten([grads[k] for k in keys])
        ddist.allreduce_sum_(flat, bucket_elems=bucket)
        flat *= 1.0 / world                                   # what dg_adamw_step's grad_scale does
        gl = ddist.mean_loss(loss)
        if rank == 0:
            _, loss_full, grads_full = R.loss_and_grads("TransformerLM", sd, x, y)
            ref = ddist.flatten([grads_full[k] for k in keys])
            ret["grad_err"] = ((flat - ref).norm() / ref.norm()).item()
            ret["loss_err"] = abs(gl.item() - loss_full.item())
            ret["n"] = flat.numel()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("bucket", [0, 10000])
def test_dp2_gradient_equals_full_batch(golden_dir, bucket):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = _free_port()
    mp.spawn(_worker, args=(2, port, golden_dir, bucket, ret), nprocs=2, join=True)
    assert ret["n"] == 43344 - 64          # every parameter except ln_f.weight / ln_f.bias
    assert ret["grad_err"] < 1e-6, ret["grad_err"]
    assert ret["loss_err"] < 1e-5          # fp32 loss ~11.5: one ulp is 9.5e-7


def test_shard_rows_rejects_uneven_split():
    with pytest.raises(ValueError):
        ddist.shard_rows(torch.zeros(7, 2), 0, 2)


def test_env_world_defaults(monkeypatch):
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        monkeypatch.delenv(k, raising=False)
    assert ddist.env_world() == (0, 0, 1)
    assert ddist.init() is None
