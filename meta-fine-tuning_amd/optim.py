"""Outer-loop optimiser on the HIP path: torch.optim.Adam semantics (train.py:28: Adam(model.parameters()), lr 1e-3,
betas (0.9, 0.999), eps 1e-8, no weight decay by default) with the update done by mft_adam_step (one fused
streaming launch per parameter tensor; parameters and moments never leave HBM)."""
import torch

from . import ops


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        touched = False
        for group in self.param_groups:
            b1, b2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                g = p.grad.contiguous()
                ops.adam_step(p.data, g, st["exp_avg"], st["exp_avg_sq"], st["step"], lr=group["lr"], beta1=b1, beta2=b2,
                              eps=group["eps"], weight_decay=group["weight_decay"])
                p._version  # parameters were updated through raw pointers: bump autograd's counter below
                p.data.add_(0)              # no-op write that increments the version counter (pack caches key on it)
                touched = True
        return loss
