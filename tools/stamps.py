"""Diagnostic (needs a -DMMF_STAMPS build): per-phase cycles of the GEMM main loop, per chunk and wave.
   MMF_EXTRA_FLAGS=-DMMF_STAMPS python -m multimodalfusion_amd.build --force && python tools/stamps.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalfusion_amd import _lib
import bench
dev = torch.device("cuda", 0)
model = bench.build_model(dev, False)
x = torch.randn(int(os.environ.get("N", 50000)), 1024, device=dev)
step = bench.make_step(model, x, dev)
for _ in range(3): step()
torch.cuda.synchronize()
buf = (C.c_uint64 * 8)()
l = _lib.lib()
for which, name in ((0, "fwd TU (linear_nt + gate_fwd)"), (1, "bwd TU (bwd_dh + tn)")):
    l.mmf_debug_stamps(which, buf)       # clear
for _ in range(5): step()
torch.cuda.synchronize()
for which, name in ((0, "fwd TU (linear_nt + gate_fwd)"), (1, "bwd TU (bwd_dh + tn)")):
    l.mmf_debug_stamps(which, buf)
    load, mfma, store, bar, n = [int(v) for v in buf[:5]]
    n = max(n, 1)
    print(f"{name}: per wave-chunk cycles: load-issue {load/n:.0f}  mfma {mfma/n:.0f}  wait+store {store/n:.0f}  barrier {bar/n:.0f}  (wave-chunks {n})")
