"""Several bags in flight on one GPU.

The reference trains with batch_size = 1 and gradient accumulation (`--gc`, utils/core_utils.py:242-247): the bags of
one accumulation window are independent until the optimizer step.  One bag's kernels do not fill an MI355X all the
time -- the row-parallel GEMMs run 224 workgroups on 256 CUs, every kernel has a tail, the small kernels are latency
-- so the window's bags are issued round-robin on a few HIP streams and the hardware overlaps them
(measured at 50k x 1024 fp32: 1090 -> 1205 bags/s with two streams; 100k bf16: 2024 -> 2607).

Each stream owns a gradient slot (one flat fp32 buffer): a bag's gradients are written to ITS stream's slot, so two
backward passes never touch the same memory; `reduce()` joins the streams and sums the slots (and all-reduces over
ranks when torch.distributed is initialised).  Results are bit-identical to running the same bags one after the other
into per-bag buffers (tests/test_gpu_pipeline.py).
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class BagsInFlight:
    def __init__(self, model: torch.nn.Module, n_streams: int = 2, device=None):
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.device = self.params[0].device if device is None else torch.device(device)
        self.n = max(1, int(n_streams))
        # each stream is measured to run beside the ones chosen before it (streams.stream_beside): two pool streams that
        # share a hardware queue would put the bags back in line, one after the other
        from .streams import stream_beside
        self.streams = []
        for _ in range(self.n):
            self.streams.append(stream_beside(self.streams, self.device))
        from .dp import flat_layout
        # the layout of optim.FlatAdam / dp.FlatGradBuffer (every tensor on a 16-byte boundary): slots are summed into
        # those buffers, and the kernels of the one-call step store float4s into the slot views
        self._offs, numel = flat_layout(self.params)
        self._rows = torch.zeros((self.n, numel), dtype=torch.float32, device=self.device)
        self.slots = [self._rows[k] for k in range(self.n)]
        self._count = 0
        self._views = None
        self._used = [False] * self.n
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(cur)

    def run(self, loss_fn_of_model_call, accumulate: bool = True, inputs=()):
        """Issue one bag: `loss_fn_of_model_call()` must run the forward and return the scalar loss; its gradients are
        added to (accumulate=True) or written over (False) the slot of the stream the bag runs on.  Returns the loss
        (a tensor living on that stream; read it after `join()`).

        `inputs`: the device tensors the closure reads (the bag and its labels).  They were produced on the caller's
        current stream -- an H2D copy, a bf16 narrowing, or feed.DevicePrefetcher's hand-over, which orders them
        behind the CURRENT stream only -- so the side stream first waits for everything issued there so far, and the
        caching allocator is told that the side stream uses them (otherwise a freed bag could be handed to the next
        bag's copy while this bag's backward, dW1 = du^T x, is still reading it)."""
        i = self._count % self.n
        self._count += 1
        st = self.streams[i]
        st.wait_stream(torch.cuda.current_stream(self.device))
        for t in inputs:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(st)
        from . import ops
        prev = ops.set_concurrent(self.n > 1)       # tile choice: leave CUs to the other bags' kernels
        try:
            with torch.cuda.stream(st):
                loss = loss_fn_of_model_call()
                grads = torch.autograd.grad(loss, self.params)
        finally:
            ops.set_concurrent(prev)
        with torch.cuda.stream(st):
            dst = [self.slots[i][off:off + p.numel()] for p, off in zip(self.params, self._offs)]
            src = [g.reshape(-1) for g in grads]
            if accumulate and self._used[i]:
                torch._foreach_add_(dst, src)                 # one multi-tensor launch
            else:
                torch._foreach_copy_(dst, src)
            self._used[i] = True
        return loss

    def run_fused(self, model, bag, label, c, alpha, loss_scale=1.0, accumulate: bool = True, inputs=()):
        """Issue one bag through the model's one-call step (model.nll_step: forward + nll_surv + backward in one C-ABI
        call): the kernels write the gradients straight into the views of this stream's slot -- no autograd graph, no
        concatenation launch.  Same stream ordering as run().  Returns nll_step's tuple (tensors on that stream)."""
        i = self._count % self.n
        self._count += 1
        st = self.streams[i]
        st.wait_stream(torch.cuda.current_stream(self.device))
        for t in tuple(inputs) + (bag, label, c):
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(st)
        if self._views is None:
            self._views = []
            for k in range(self.n):
                self._views.append([self.slots[k][off:off + p.numel()].view_as(p)
                                    for p, off in zip(self.params, self._offs)])
        from . import ops
        prev = ops.set_concurrent(self.n > 1)       # tile choice: leave CUs to the other bags' kernels
        try:
            with torch.cuda.stream(st):
                out = model.nll_step(bag, label, c, alpha=alpha, loss_scale=loss_scale, grad_out=self._views[i],
                                     accumulate=accumulate and self._used[i])
                self._used[i] = True
        finally:
            ops.set_concurrent(prev)
        return out

    def all_reduce_slot(self, group=None):
        """Multi-GPU benchmark helper: all-reduce the slot of the bag issued last, on its stream (the collective of
        one bag overlaps with the other stream's kernels)."""
        i = (self._count - 1) % self.n
        if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            with torch.cuda.stream(self.streams[i]):
                dist.all_reduce(self.slots[i], op=dist.ReduceOp.SUM, group=group)

    def join(self):
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            cur.wait_stream(s)

    def reduce(self, group=None, all_reduce: bool = True) -> torch.Tensor:
        """End of an accumulation window: join, sum the slots (fixed order), all-reduce over ranks; returns the flat
        gradient (views of it can be handed to the optimizer, see assign_grads)."""
        self.join()
        used = [i for i in range(self.n) if self._used[i]]
        total = self.slots[used[0]] if used else self.slots[0].zero_()
        for i in used[1:]:
            total.add_(self.slots[i])
        if all_reduce and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
            dist.all_reduce(total, op=dist.ReduceOp.SUM, group=group)
        self._used = [False] * self.n
        for s in self.streams:                       # the next window's bags must see the reduced / reset slots
            s.wait_stream(torch.cuda.current_stream(self.device))
        return total

    def release(self):
        """Call after the optimizer step that consumed `reduce()`'s result: the next window's bags (on the side
        streams) must see the updated weights."""
        cur = torch.cuda.current_stream(self.device)
        for s in self.streams:
            s.wait_stream(cur)

    def assign_grads(self, flat: torch.Tensor):
        for p, off in zip(self.params, self._offs):
            p.grad = flat[off:off + p.numel()].view_as(p)
