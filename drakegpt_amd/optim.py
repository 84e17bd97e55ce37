"""AdamW on the HIP kernel for the drop-in (autograd) path -- the reference's
``torch.optim.AdamW(model.parameters(), lr=..., betas=...)`` (src/train.py:121) with the same defaults
(eps 1e-8, weight_decay 1e-2, amsgrad off); parameters whose ``.grad`` is None are skipped entirely
(``ln_f``).  The graph-captured TrainEngine has its own flat-buffer optimizer step.

MI355X-first: on the first step the parameters that have a gradient are moved into ONE flat fp32 buffer (their
``.data`` become views of it, so ``state_dict()`` keeps working), with flat ``m`` / ``v`` / gradient buffers beside it:
a step is one fused copy of the gradients, one ``dg_adamw_step`` launch (which also moves the step counter on) instead of two
launches per parameter tensor (168 for the tiny TransformerLM).  With a process group the flat gradient is
all-reduced (SUM) first and the kernel applies 1 / world: data-parallel training for all six models."""
from __future__ import annotations

from typing import Optional

import torch

from . import ops

_ALIGN = 64          # floats: every tensor starts on a 256-byte boundary of the flat buffer


class _Flat:
    def __init__(self, params, device):
        self.key = tuple(id(p) for p in params)
        self.offsets, n = [], 0
        for p in params:
            self.offsets.append(n)
            n += (p.numel() + _ALIGN - 1) // _ALIGN * _ALIGN
        self.n = n
        self.flat = torch.zeros(n, dtype=torch.float32, device=device)
        self.g = torch.zeros(n, dtype=torch.float32, device=device)
        self.m = torch.zeros(n, dtype=torch.float32, device=device)
        self.v = torch.zeros(n, dtype=torch.float32, device=device)
        self.t = ops.new_rng_state(0, device, 0)                 # word 2 = number of steps taken
        self.hyper = torch.zeros(5, dtype=torch.float32, device=device)
        self.hyper_host = None
        self.gviews = [self.g[o:o + p.numel()].view(p.shape) for o, p in zip(self.offsets, params)]

    def view(self, buf, i, p):
        return buf[self.offsets[i]:self.offsets[i] + p.numel()].view(p.shape)


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, *, process_group=None, world_size: int = 1):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.process_group, self.world_size = process_group, int(world_size)
        self._flat = {}

    def _adopt(self, gi: int, params) -> _Flat:
        old = self._flat.get(gi)
        dev = params[0].device
        fl = _Flat(params, dev)
        carry = {}
        if old is not None:                                      # the set of trained parameters changed: keep their moments
            carry = {pid: i for i, pid in enumerate(old.key)}
            fl.t.copy_(old.t)
        for i, p in enumerate(params):
            dst = fl.view(fl.flat, i, p)
            dst.copy_(p.data)
            p.data = dst
            if id(p) in carry:
                j = carry[id(p)]
                fl.view(fl.m, i, p).copy_(old.m[old.offsets[j]:old.offsets[j] + p.numel()].view(p.shape))
                fl.view(fl.v, i, p).copy_(old.v[old.offsets[j]:old.offsets[j] + p.numel()].view(p.shape))
        self._flat[gi] = fl
        return fl

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for gi, group in enumerate(self.param_groups):
            params = [p for p in group["params"] if p.grad is not None]
            if not params:
                continue
            if any(not p.is_cuda for p in params):
                raise RuntimeError("drakegpt_amd.optim.AdamW updates GPU parameters only (no CPU path)")
            fl = self._flat.get(gi)
            if fl is None or fl.key != tuple(id(p) for p in params) or any(p.data.data_ptr() != fl.flat.data_ptr() + 4 * o
                                                                           for p, o in zip(params, fl.offsets)):
                fl = self._adopt(gi, params)
            hy = (group["lr"], group["betas"][0], group["betas"][1], group["eps"], group["weight_decay"])
            if hy != fl.hyper_host:
                fl.hyper.copy_(torch.tensor(hy, dtype=torch.float32))
                fl.hyper_host = hy
            torch._foreach_copy_(fl.gviews, [p.grad for p in params])           # one fused launch
            scale = 1.0
            if self.world_size > 1:
                import torch.distributed as dist
                dist.all_reduce(fl.g, op=dist.ReduceOp.SUM, group=self.process_group)
                scale = 1.0 / self.world_size
            ops.adamw_step(fl.flat, fl.g, fl.m, fl.v, fl.hyper, fl.t, grad_scale=scale, advance=True)
        return loss

    # ---- checkpointing: the moments and the step count live in the flat buffers, not in torch's per-parameter `state`; export /
    # import them in torch.optim.AdamW's own format (state[i] = {"step", "exp_avg", "exp_avg_sq"}) so that a resumed run keeps
    # its bias correction and moments, and a state_dict written by torch.optim.AdamW loads here (ref: src/train.py:121)
    def state_dict(self):
        sd = super().state_dict()                   # param_groups with indices; `state` is empty (nothing lives there)
        state, base = {}, 0
        for gi, group in enumerate(self.param_groups):
            fl = self._flat.get(gi)
            if fl is not None:
                idx = {id(p): base + j for j, p in enumerate(group["params"])}
                t = float(fl.t[2].item())
                by_id = {id(p): p for p in group["params"]}
                for i, pid in enumerate(fl.key):
                    p = by_id.get(pid)
                    if p is None:
                        continue
                    state[idx[pid]] = {"step": torch.tensor(t), "exp_avg": fl.view(fl.m, i, p).detach().clone(),
                                       "exp_avg_sq": fl.view(fl.v, i, p).detach().clone()}
            base += len(group["params"])
        sd["state"] = state
        return sd

    @torch.no_grad()
    def load_state_dict(self, state_dict):
        state = state_dict.get("state", {})
        super().load_state_dict({"state": {}, "param_groups": state_dict["param_groups"]})
        base = 0
        for gi, group in enumerate(self.param_groups):
            have = [(j, p) for j, p in enumerate(group["params"]) if (base + j) in state]
            if have:
                params = [p for _, p in have]
                if any(not p.is_cuda for p in params):
                    raise RuntimeError("drakegpt_amd.optim.AdamW updates GPU parameters only (no CPU path)")
                self._flat.pop(gi, None)
                fl = self._adopt(gi, params)
                steps = set()
                for i, (j, p) in enumerate(have):
                    st = state[base + j]
                    fl.view(fl.m, i, p).copy_(st["exp_avg"].to(p.device, torch.float32))
                    fl.view(fl.v, i, p).copy_(st["exp_avg_sq"].to(p.device, torch.float32))
                    steps.add(int(float(st["step"])))
                if len(steps) != 1:
                    raise ValueError("drakegpt_amd.optim.AdamW keeps ONE step count per parameter group; the state holds " + str(sorted(steps)))
                fl.t.copy_(ops.new_rng_state(0, params[0].device, steps.pop()))
            base += len(group["params"])
