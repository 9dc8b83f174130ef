"""Diagnostic: K-lin alone through the C ABI for K = 32 .. 1024 (1 .. 32 chunks) at 50k rows: the intercept of time over
chunks is prologue + epilogue, the slope the cost of a chunk.  Run with MMF_LIB_PATH set to a diag build to compare."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalfusion_amd import _lib
l = _lib.lib()
dev = "cuda"
M, N = 50000, 256
for K in (32, 64, 128, 256, 512, 1024):
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.03; b = torch.zeros(N, device=dev)
    y = torch.empty(M, N, device=dev)
    segs = (C.c_void_p * 1)(x.data_ptr())
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def run():
        return l.mmf_linear_forward(segs, 1, K, M, C.c_void_p(W.data_ptr()), C.c_void_p(b.data_ptr()), N, 1, C.c_float(0.25), 7, 0, None,
                                    C.c_void_p(y.data_ptr()), None, 0, None, 0, st)
    for _ in range(20): assert run() == 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 100
    e0.record()
    for _ in range(reps): run()
    e1.record(); torch.cuda.synchronize()
    print(f"K={K:5d} chunks={K // 32:3d}: {e0.elapsed_time(e1) / reps * 1e3:8.1f} us")
