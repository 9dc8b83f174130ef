"""Bag feed: host -> HBM staging of the next bag overlapped with the current bag's kernels.

The reference copies every tensor of a batch synchronously at the top of each iteration
(utils/core_utils.py:194-198: `.to(device)` on pageable memory).  At ~1 ms of GPU work per 50k bag the 205 MB copy
(~3.3 ms over PCIe Gen5 x16) dominates, so the feed keeps `depth` bags in flight: pinned host staging buffers, copies
issued with non_blocking=True on a dedicated HIP stream, an event per bag that the compute stream waits on.

`DevicePrefetcher(loader)` wraps any iterable that yields the reference's batch tuple
(radio_features: dict, path_features, genomic_features, label, event_time, c) -- e.g. a DataLoader built with
collate_MIL_survival (utils/utils.py:35-46) -- and yields the same tuple with the tensors already on the GPU.
"""
from __future__ import annotations

from collections import deque

import torch


def _pin(t: torch.Tensor) -> torch.Tensor:
    if not torch.is_tensor(t) or t.is_cuda:
        return t
    return t if t.is_pinned() else t.pin_memory()


class DevicePrefetcher:
    def __init__(self, loader, device=None, depth: int = 2):
        self.loader = loader
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.depth = max(1, int(depth))
        self.stream = torch.cuda.Stream(self.device)

    def __len__(self):
        return len(self.loader)

    def _stage(self, batch):
        radio, path, genomic, label, event_time, c = batch
        with torch.cuda.stream(self.stream):
            move = lambda t: _pin(t).to(self.device, non_blocking=True) if torch.is_tensor(t) else t
            out = ({k: move(v) for k, v in radio.items()}, move(path),
                   move(genomic.float() if torch.is_tensor(genomic) else genomic), move(label), event_time, move(c))
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return out, ev

    def __iter__(self):
        it = iter(self.loader)
        q = deque()
        try:
            for _ in range(self.depth):
                q.append(self._stage(next(it)))
        except StopIteration:
            pass
        while q:
            out, ev = q.popleft()
            cur = torch.cuda.current_stream(self.device)
            cur.wait_event(ev)                       # compute waits for THIS bag only; later copies keep flowing
            for t in list(out[0].values()) + [out[1], out[2], out[3], out[5]]:
                if torch.is_tensor(t) and t.is_cuda:
                    t.record_stream(cur)             # allocator: do not reuse before the compute stream is done
            try:
                q.append(self._stage(next(it)))
            except StopIteration:
                pass
            yield out
