"""hipGraph capture of a launch-bound step (small bags are host-bound: ~0.4 ms of Python + autograd + ctypes per
step against 0.1-0.3 ms of kernels).

`GraphedStep(fn)` captures `fn()` -- typically zero-grads + model(**static_inputs) + loss + backward, all on
fixed-shape static tensors -- into one hipGraph and replays it.  The C ABI launches on the capture stream, allocates
nothing and never synchronises, so capture needs nothing special from the kernels except the dropout seed: a by-value
seed would be frozen into the captured kernel arguments and every replay would draw the same mask.  Every dropout-
bearing entry point therefore takes an optional device-resident word that the kernels add to their keys
(mmf_amil_desc::seed_dev and the `seed_dev` arguments, include/mmf_amil.h); the first node of the graph bumps that
word, so replay r uses effective seed = host seed + seed0 + r * BUMP (uint32 wrap), reproducibly.  The word is an
argument of each call, not library state: ops.py passes the DeviceSeed that is current while the step is captured
and each autograd node hands the same word to its backward, so graphed and eager steps, or two graphed models, can
live in one process.
"""
from __future__ import annotations

import torch

from . import ops

SEED_BUMP = 0x6B43A9B5


def _as_i32(v: int) -> int:
    v &= 0xFFFFFFFF
    return v - (1 << 32) if v >= (1 << 31) else v


class DeviceSeed:
    """A device word that the kernels launched inside `with seed:` add to their dropout keys."""

    def __init__(self, value: int = 0, device=None):
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.word = torch.full((1,), _as_i32(value), dtype=torch.int32, device=dev)
        self._prev = []

    def bump(self):
        self.word.add_(_as_i32(SEED_BUMP))       # int32 add wraps like uint32

    def value(self) -> int:
        return int(self.word.item()) & 0xFFFFFFFF

    def __enter__(self):
        self._prev.append(ops.set_device_seed(self.word))
        return self

    def __exit__(self, *a):
        ops.set_device_seed(self._prev.pop())


class GraphedStep:
    def __init__(self, fn, warmup: int = 3, seed0: int = 0):
        self.seed = DeviceSeed(seed0)
        # the graph's own tick words (mmf_amil_desc::sync): replays of this graph are ordered among themselves, but may
        # overlap other streams' calls, so it shares them with nobody
        self.sync = torch.zeros(ops.SYNC_WORDS, dtype=torch.int32, device=self.seed.word.device)
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        prev_sync = ops.set_sync_override(self.sync)
        try:
            self._build(fn, warmup, s)
        finally:
            ops.set_sync_override(prev_sync)

    def _build(self, fn, warmup, s):
        with self.seed:                          # the word is passed to the launches made in here, nowhere else
            with torch.cuda.stream(s):           # warm-up off the default stream (allocator pools, LDS attributes)
                for _ in range(warmup):
                    self.seed.bump()
                    fn()
            torch.cuda.current_stream().wait_stream(s)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.seed.bump()                 # first node: fresh dropout masks on every replay
                self.out = fn()

    def __call__(self):
        self.graph.replay()
        return self.out

    def close(self):
        """Nothing to unregister (the seed word is a per-call argument); kept for callers of the earlier interface."""
