"""AdamW on the HIP kernel for the drop-in (autograd) path -- the reference's
``torch.optim.AdamW(model.parameters(), lr=..., betas=...)`` (src/train.py:121) with the same defaults
(eps 1e-8, weight_decay 1e-2, amsgrad off); parameters whose ``.grad`` is None are skipped entirely
(``ln_f``).  The graph-captured TrainEngine has its own flat-buffer optimizer step."""
from __future__ import annotations

import torch

from . import ops


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            hyper = None
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not p.is_cuda:
                    raise RuntimeError("drakegpt_amd.optim.AdamW updates GPU parameters only (no CPU path)")
                st = self.state[p]
                if not st:
                    st["m"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["v"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["t"] = ops.new_rng_state(0, p.device, 0)      # word 2 = number of steps taken
                if hyper is None or hyper.device != p.device:
                    hyper = torch.tensor([group["lr"], group["betas"][0], group["betas"][1], group["eps"], group["weight_decay"]],
                                         dtype=torch.float32, device=p.device)
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                if p.is_contiguous() and p.data_ptr() % 16 == 0 and g.data_ptr() % 16 == 0:
                    ops.adamw_step(p.data.view(-1), g.view(-1), st["m"].view(-1), st["v"].view(-1), hyper, st["t"])
                else:       # odd views (rare): update a contiguous copy
                    tmp = p.data.contiguous()
                    ops.adamw_step(tmp.view(-1), g.view(-1).clone(), st["m"].view(-1), st["v"].view(-1), hyper, st["t"])
                    p.data.copy_(tmp)
                ops.state_advance(st["t"])
        return loss
