"""Per-step tail on the GPU: one fused kernel for the L1 regulariser's gradient and the
Adam(+L2) update over ONE flat fp32 parameter buffer -- the same flat gradient buffer the data-parallel
all-reduce uses (dp.py).

Reference semantics reproduced (utils/core_utils.py:216-219,242-247 + utils/utils.py:144-146,249-257):
every micro-batch adds `lambda_reg * |W|_1` to its loss un-divided, so after `gc` micro-batches autograd has added
`gc * lambda_reg * sign(W)` to the gradient; then `torch.optim.Adam(lr, weight_decay=reg)` steps.  Here the model's
loss carries NO L1 term: `FlatAdam.step(l1_micro_batches=gc)` adds `gc * lambda_reg * sign(W)` inside the kernel.
`l1_value()` returns lambda_reg * sum|W| as a device scalar for logging (no host sync).
"""
from __future__ import annotations

import torch

from ._lib import check, lib, ptr, stream_ptr


class FlatAdam:
    def __init__(self, model: torch.nn.Module, lr=2e-4, weight_decay=0.0, betas=(0.9, 0.999), eps=1e-8,
                 lambda_l1=0.0, l1_modules=None):
        """l1_modules: None = the L1 term covers every parameter (`l1_reg_all`, utils/utils.py:249-257); a list of
        sub-modules = only their parameters (`l1_reg_modules`, utils/utils.py:259-268: model.fc_omic and model.mm)."""
        self.params = [p for p in model.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev = self.params[0].device
        from .dp import flat_layout
        offs, n = flat_layout(self.params)            # every tensor on a 16-byte boundary, zero padding between
        self.n = n
        self.flat_w = torch.zeros(n, dtype=torch.float32, device=dev)
        # gradient bucket: [n gradients | 2 control words].  The control words ride along in the one all-reduce of a
        # data-parallel step (utils/core_utils.py: "did the window's last bag run" and "bags kept in the window").
        self.bucket = torch.zeros(n + 2, dtype=torch.float32, device=dev)
        self.flat_g = self.bucket[:n]
        self.tail = self.bucket[n:]
        self.l1_mask = None
        if l1_modules is not None:
            chosen = {id(p) for m in l1_modules for p in m.parameters()}
            self.l1_mask = torch.zeros(n, dtype=torch.float32, device=dev)
            for p, off in zip(self.params, offs):
                if id(p) in chosen:
                    self.l1_mask[off:off + p.numel()] = 1.0
        self.m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, off in zip(self.params, offs):          # parameters (and grads) become views of the flat buffers
                k = p.numel()
                self.flat_w[off:off + k].copy_(p.data.reshape(-1))
                p.data = self.flat_w[off:off + k].view_as(p)
                p.grad = self.flat_g[off:off + k].view_as(p)
        self.lr, self.wd, self.betas, self.eps, self.lambda_l1 = lr, weight_decay, betas, eps, lambda_l1
        self.t = 0
        self._partials = torch.empty(512, dtype=torch.float32, device=dev)

    # the flat gradient buffer doubles as the all-reduce bucket (dp.FlatGradBuffer interface)
    @property
    def flat(self):
        return self.flat_g

    def all_reduce(self, group=None):
        import torch.distributed as dist
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM, group=group)

    def zero_grad(self, set_to_none: bool = False):
        self.bucket.zero_()

    def zero(self):
        self.bucket.zero_()

    def step(self, l1_micro_batches: int = 0):
        self.t += 1
        check(lib().mmf_adam_l1_step(ptr(self.flat_w), ptr(self.flat_g), ptr(self.m), ptr(self.v), self.n,
                                     self.lr, self.betas[0], self.betas[1], self.eps, self.wd,
                                     self.lambda_l1 * l1_micro_batches, ptr(self.l1_mask), self.t, stream_ptr()),
              "mmf_adam_l1_step")

    def l1_value(self):
        """lambda_l1 * sum_W |W| (what `reg_fn(model) * lambda_reg` evaluates to), device scalar."""
        out = torch.empty((), dtype=torch.float32, device=self.flat_w.device)
        w = self.flat_w if self.l1_mask is None else self.flat_w * self.l1_mask
        check(lib().mmf_abs_sum(ptr(w), self.n, ptr(self._partials), ptr(out), stream_ptr()), "mmf_abs_sum")
        return out * self.lambda_l1
