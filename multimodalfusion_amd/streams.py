"""Side streams that really run beside another stream.

HIP multiplexes a process's streams onto a handful of hardware queues; two streams that land on the same queue execute
in order, whatever the program says.  Which pool stream shares a queue with which depends on how many streams the process
made before (measured, tools/r4_mm_queue.py: with the 7th pool stream as the side stream the multimodal step takes 1.07
ms, with any of its neighbours 0.89 -- and the two-bags-in-flight rate falls back to one bag at a time the same way).
`stream_beside` therefore PROBES: ~1 ms of streaming kernels on the reference stream, a tiny kernel on the candidate that
waits for their start only; the candidate is kept when its kernel finishes while the reference stream is still busy."""
from __future__ import annotations

import torch

def _runs_beside(ref: torch.cuda.Stream, cand: torch.cuda.Stream, device) -> bool:
    # 1 GiB of whatever the allocator hands out (back in its cache when the probe returns): one pass over it is ~0.4 ms of
    # HBM traffic with no library behind it
    a = torch.empty(1 << 28, device=device)
    t = torch.zeros(64, device=device)
    e0, e1, ec = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    torch.cuda.synchronize(device)
    with torch.cuda.stream(ref):
        t.add_(1.0)
        e0.record(ref)
        a.add_(1.0)
        a.add_(1.0)
        a.add_(1.0)
        e1.record(ref)
    with torch.cuda.stream(cand):
        cand.wait_event(e0)
        t.add_(1.0)
        ec.record(cand)
    torch.cuda.synchronize(device)
    return e0.elapsed_time(ec) < 0.5 * e0.elapsed_time(e1)


def stream_beside(others, device=None, tries: int = 12) -> torch.cuda.Stream:
    """A new stream whose kernels run concurrently with those of every stream in `others` (torch.cuda.Stream objects).
    While a stream is being captured nothing can be measured: the next pool stream is returned as it is (a captured
    graph's branches are placed by the graph launch, not by these streams)."""
    device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if torch.cuda.is_current_stream_capturing():
        return torch.cuda.Stream(device)
    last = None
    for _ in range(tries):
        last = torch.cuda.Stream(device)
        try:
            if all(_runs_beside(o, last, device) for o in others):
                return last
        except torch.cuda.OutOfMemoryError:  # no room for the probe's operand: an unmeasured stream is still a correct one
            torch.cuda.synchronize(device)
            return last
    return last                              # nothing passed (one hardware queue?): still correct, just not concurrent
