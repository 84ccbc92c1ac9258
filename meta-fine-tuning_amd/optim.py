"""Outer-loop optimiser on the HIP path: torch.optim.Adam semantics (train.py:28: Adam(model.parameters()), lr 1e-3,
betas (0.9, 0.999), eps 1e-8, no weight decay by default) with the update of ALL parameter tensors done by one
multi-tensor launch (mft_adam_multi; parameters and moments never leave HBM)."""
import numpy as np
import torch

from . import ops
from .graph_step import bump_versions


CHUNK = 16384          # elements per workgroup of the multi-tensor launch (multiple of 4: chunks stay 16-byte aligned)


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.grad_scale = 1.0        # every gradient is multiplied by this as the update reads it (train.AllReduceAdam: 1 / world size
                                     # when .grad holds the all-reduced SUM -- the mean without a division launch)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            b1, b2 = group["betas"]
            ps = [p for p in group["params"] if p.grad is not None]
            if not ps:
                continue
            rows = []
            step = None
            for p in ps:
                st = self.state[p]
                if not st:
                    st["step"] = 0
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["step"] += 1
                if not p.grad.is_contiguous():
                    p.grad = p.grad.contiguous()
                if step is None:
                    step = st["step"]
                if st["step"] != step or not p.is_cuda:
                    step = -1                      # mixed step counts: fall back to per-tensor launches
                n = p.numel()
                pp, gp, mp, vp = p.data_ptr(), p.grad.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr()
                for off in range(0, n, CHUNK):
                    rows.append((pp + 4 * off, gp + 4 * off, mp + 4 * off, vp + 4 * off, min(CHUNK, n - off)))
            if step is not None and step > 0:
                # all tensors of the group in ONE launch (104 tensors for GnnNet): a device table of (p, g, m, v, n) chunks.  The
                # table is rebuilt (one small pageable host-to-device copy, which drains the stream) only when an address in it
                # changed: under the graphed episode loop parameters, static gradients and moments never move.
                key = (ps[0].device, tuple(rows))
                tables = self.__dict__.setdefault("_tables", {})
                cached = tables.get(id(group))
                if cached is None or cached[0] != key:
                    cached = tables[id(group)] = (key, torch.from_numpy(np.asarray(rows, dtype=np.int64)).to(ps[0].device))
                table = cached[1]
                rc = ops._lib.lib().mft_adam_multi(ops._p(table), len(rows), step, group["lr"], b1, b2, group["eps"],
                                                   group["weight_decay"], float(self.grad_scale), ops._stream())
                ops._lib.check(rc, "mft_adam_multi")
            else:
                for p in ps:
                    st = self.state[p]
                    if self.grad_scale != 1.0:
                        raise RuntimeError("grad_scale needs the multi-tensor launch (all parameters on the GPU, equal step counts)")
                    ops.adam_step(p.data, p.grad, st["exp_avg"], st["exp_avg_sq"], st["step"], lr=group["lr"], beta1=b1,
                                  beta2=b2, eps=group["eps"], weight_decay=group["weight_decay"])
            # parameters were updated through raw pointers: bump autograd's version counters ON THE PARAMETERS THEMSELVES (the
            # weight-pack caches of autograd_ops key on ``p._version``) -- host-side, no launch
            bump_versions(ps)
        return loss
