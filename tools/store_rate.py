"""Calibration: chip-wide store rate of a plain fill (torch's vectorised fill kernel) for buffers of the sizes the
epilogues write (h: 51 MB, TN slabs: 66 MB) and a large one.  Compare with the ~2 TB/s at which the GEMM epilogues drain."""
import torch

def rate(nbytes, reps=50):
    x = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda")
    for _ in range(5):
        x.fill_(1.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        x.fill_(1.0)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / reps
    return us, nbytes / us / 1e6

for mb in (51, 66, 205, 1024):
    us, tbs = rate(mb * 1000 * 1000)
    print(f"fill {mb:5d} MB: {us:8.1f} us  {tbs:.2f} TB/s")
