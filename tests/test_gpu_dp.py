"""Data-parallel engine path on real hardware: two processes share cuda:0 (gloo moves the flat
gradient; on a multi-GPU node the same code runs over RCCL with one GPU per rank).  The 2-rank
result must equal the single-process result on the concatenated batch."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
V = 80


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run(rank, world, port, golden_dir, ret, precision="fp32", buckets=None, key=None):
    import torch.distributed as dist
    import drakegpt_amd as D
    from drakegpt_amd import dist as ddist
    from drakegpt_amd.engine import TrainEngine
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    dev = torch.device("cuda:0")
    pg = None
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        pg = dist.group.WORLD
    try:
        fix = torch.load(os.path.join(golden_dir, "traj5_TransformerLM.pt"), weights_only=True)
        m = D.TransformerLM(V, 32, 8, 4, 3, 0.0, precision=precision)
        m.load_state_dict(fix["init"])
        m = m.to(dev)
        Bg = 32
        eng = TrainEngine(m, Bg // world, 8, lr=1e-3, betas=(0.9, 0.95), rank=rank, world_size=world, process_group=pg,
                          dp_buckets=buckets)
        if buckets and world > 1:
            assert eng.dp_buckets == buckets and len(eng._dp_plan()) == buckets
        losses = []
        for it in range(3):
            x = ddist.shard_rows(fix["x"][it], rank, world)
            y = ddist.shard_rows(fix["y"][it], rank, world)
            eng.set_batch(x.to(dev), y.to(dev))
            losses.append(ddist.mean_loss(eng.step().clone(), pg).item())
        eng.check_status()
        if rank == 0:
            ret[key if key is not None else world] = (losses, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_two_rank_engine_equals_single_process(dev, golden_dir):
    mgr = mp.Manager()
    ret = mgr.dict()
    ctx = mp.get_context("spawn")
    for world in (1, 2):
        port = _free_port()
        procs = [ctx.Process(target=_run, args=(r, world, port, golden_dir, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
    l1, sd1 = ret[1]
    l2, sd2 = ret[2]
    fix = torch.load(os.path.join(golden_dir, "traj5_TransformerLM.pt"), weights_only=True)
    for a, b, c in zip(l1, l2, fix["losses"].tolist()):
        assert abs(a - b) < 2e-5 * abs(a) and abs(a - c) < 2e-4 * abs(c), (l1, l2)
    for k in sd1:
        assert (sd1[k] - sd2[k]).abs().max().item() < 2e-6, k


def test_two_rank_bucketed_overlap_equals_single_exchange(dev, golden_dir):
    """bf16 (the grouped dW GEMM), three layer groups: each group's range of the flat gradient is all-reduced asynchronously
    while the next group's backward graph runs.  Same result as one all-reduce of the whole gradient and as the
    single-process step on the concatenated batch, up to the fp32 summation order of the differently cut dW contractions."""
    mgr = mp.Manager()
    ret = mgr.dict()
    ctx = mp.get_context("spawn")
    for key, world, buckets in (("one", 1, None), ("flat", 2, 1), ("bucketed", 2, 3)):
        port = _free_port()
        procs = [ctx.Process(target=_run, args=(r, world, port, golden_dir, ret, "bf16", buckets, key)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
    (l1, sd1), (l2, sd2), (l3, sd3) = ret["one"], ret["flat"], ret["bucketed"]
    for a, b, c in zip(l1, l2, l3):
        assert abs(a - b) < 2e-4 * abs(a) and abs(b - c) < 2e-5 * abs(b), (l1, l2, l3)
    for k in sd2:
        assert (sd2[k] - sd3[k]).abs().max().item() < 2e-5, k
        assert (sd1[k] - sd2[k]).abs().max().item() < 5e-4, k


def _run_module(rank, world, port, golden_dir, ret):
    """the autograd (nn.Module) path with the flat-buffer AdamW: an earlier-stage model, data parallel"""
    import torch.distributed as dist
    import drakegpt_amd as D
    from drakegpt_amd import dist as ddist
    from drakegpt_amd.optim import AdamW
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    dev = torch.device("cuda:0")
    pg = None
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
        pg = dist.group.WORLD
    try:
        fix = torch.load(os.path.join(golden_dir, "traj5_TransformerLM.pt"), weights_only=True)
        torch.manual_seed(3)
        m = D.ResidualBlocksLM(V, 32, 8, 4, 2).to(dev).train()
        opt = AdamW(m.parameters(), lr=1e-3, betas=(0.9, 0.95), process_group=pg, world_size=world)
        losses = []
        for it in range(3):
            x = ddist.shard_rows(fix["x"][it], rank, world).to(dev)
            y = ddist.shard_rows(fix["y"][it], rank, world).to(dev)
            _, loss = m(x, y)
            opt.zero_grad()
            loss.backward()
            opt.step()
            losses.append(ddist.mean_loss(loss.detach(), pg).item())
        if rank == 0:
            ret[world] = (losses, {k: v.detach().cpu().clone() for k, v in m.state_dict().items()})
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_two_rank_module_path_equals_single_process(dev, golden_dir):
    mgr = mp.Manager()
    ret = mgr.dict()
    ctx = mp.get_context("spawn")
    for world in (1, 2):
        port = _free_port()
        procs = [ctx.Process(target=_run_module, args=(r, world, port, golden_dir, ret)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
    (l1, sd1), (l2, sd2) = ret[1], ret[2]
    for a, b in zip(l1, l2):
        assert abs(a - b) < 2e-5 * abs(a), (l1, l2)
    for k in sd1:
        assert (sd1[k] - sd2[k]).abs().max().item() < 2e-6, k
