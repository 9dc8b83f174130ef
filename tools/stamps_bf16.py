"""Diagnostic (stamps build: python tools/diag_build.py stamps): per-kernel phase cycles of the bf16 path.
    MMF_LIB_PATH=multimodalfusion_amd/_diag/libmmf_stamps.so python tools/stamps_bf16.py [N]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multimodalfusion_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
dev = torch.device("cuda", 0)
model = bench.build_model(dev, False)
x = torch.randn(N, 1024, device=dev).to(torch.bfloat16)
step = bench.make_step(model, x, dev)
for _ in range(3): step()
torch.cuda.synchronize()
buf = (C.c_uint64 * 32)()
_lib.lib().mmf_debug_stamps(2, buf)
R = 10
for _ in range(R): step()
torch.cuda.synchronize()
_lib.lib().mmf_debug_stamps(2, buf)
for k, name in enumerate(["linear", "gate", "dh", "tn"]):
    v = [int(t) for t in buf[8 * k:8 * k + 8]]
    w = max(v[7], 1)
    tot = v[0] + v[1] + v[2]
    mhz = tot / max(v[6], 1) * 100.0            # shader cycles per 100 MHz tick
    tot = v[0] + v[1] + v[2] + v[3]
    mhz = tot / max(v[6], 1) * 100.0
    print(f"{name:7s} waves/launch {v[7] // R:6d}  per wave shader cycles: slot0 {v[0] / w:9.0f}  slot1 {v[1] / w:9.0f}  "
          f"slot2 {v[2] / w:9.0f}  slot3 {v[3] / w:9.0f}   wave life {v[6] / w / 100.0:7.1f} us  (clock {mhz:6.0f} MHz)")
