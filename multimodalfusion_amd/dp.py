"""One-bag-per-GPU data parallelism: one process per GPU (torch.distributed, backend "nccl" == RCCL over
xGMI on ROCm), ONE all-reduce (SUM) of a flat fp32 gradient buffer per optimizer step.

Semantics == the reference's gradient accumulation `--gc G` (utils/core_utils.py:242-247): rank r contributes
grad(loss_r / G) + lambda * sign(W) -- the reference adds the L1 term un-divided on every micro-batch, so the
SUM (not the mean) over ranks reproduces `gc = world_size` exactly -- then every rank applies the identical
optimizer step.  No parameter broadcast is needed after step 0 because all ranks start from the same
weights (same seed, main.py:47) and apply identical updates; `broadcast_parameters` is there for safety.

The attention-MIL path has no intra-bag exchange step, so there is no other collective on the data path.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


def flat_layout(params, align: int = 4):
    """Offsets (in elements) of `params` inside ONE flat fp32 buffer, every tensor starting on a 16-byte boundary
    (the kernels take parameters and write gradients with 16-byte vector accesses; a 1-element tensor such as
    attention_c.bias would otherwise misalign everything behind it), and the buffer's total length.  The padding
    elements are zero and stay zero (zero weight, zero gradient => zero Adam update).  Shared by optim.FlatAdam,
    FlatGradBuffer and pipeline.BagsInFlight, whose buffers are added to / copied into each other."""
    offs, off = [], 0
    for p in params:
        offs.append(off)
        off += (p.numel() + align - 1) // align * align
    return offs, off


class FlatGradBuffer:
    """Makes every parameter's .grad a view into one contiguous buffer, so a step needs one collective."""

    def __init__(self, model: torch.nn.Module):
        self.params = [p for p in model.parameters() if p.requires_grad]
        offs, n = flat_layout(self.params)
        dev = self.params[0].device
        # [n gradients | 2 control words]: the control words travel in the same collective (utils/core_utils.py)
        self.bucket = torch.zeros(n + 2, dtype=torch.float32, device=dev)
        self.flat = self.bucket[:n]
        self.tail = self.bucket[n:]
        for p, off in zip(self.params, offs):
            p.grad = self.flat[off:off + p.numel()].view_as(p)

    def zero(self):
        self.bucket.zero_()

    def all_reduce(self, group=None):
        """One SUM all-reduce (RCCL on GPUs; gloo in the CPU tests).  A bucket of 1.6-34 MB: with 7 direct
        xGMI links per GPU this is latency- to per-link-bandwidth-bound, so it is never split."""
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(self.bucket, op=dist.ReduceOp.SUM, group=group)


def broadcast_parameters(model: torch.nn.Module, src: int = 0, group=None):
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        for p in model.parameters():
            dist.broadcast(p.data, src=src, group=group)


def dp_micro_step(loss, loss_reg, world_size: int):
    """Backward of one bag on one rank: d(loss / G + loss_reg), G = world_size (x local gc if any)."""
    total = loss / world_size
    if loss_reg is not None and not (isinstance(loss_reg, (int, float)) and loss_reg == 0):
        total = total + loss_reg
    total.backward()
    return total
