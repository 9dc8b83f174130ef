"""Diagnostic (stamps build): phase cycles of the bf16 fused forward, training step or forward-only (no h / a / b stores).
    MMF_LIB_PATH=multimodalfusion_amd/_diag/libmmf_stamps.so python tools/stamps_f2.py [N] [infer]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multimodalfusion_amd import _lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
infer = len(sys.argv) > 2 and sys.argv[2] == "infer"
dev = torch.device("cuda", 0)
model = bench.build_model(dev, False)
x = torch.randn(N, 1024, device=dev).to(torch.bfloat16)
if infer:
    model.eval()
    def step():
        with torch.no_grad():
            model(path_features=x, attention_only=True)
else:
    step = bench.make_step(model, x, dev)
for _ in range(3): step()
torch.cuda.synchronize()
buf = (C.c_uint64 * 32)()
_lib.lib().mmf_debug_stamps(2, buf)
R = 10
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(R): step()
t1.record()
torch.cuda.synchronize()
_lib.lib().mmf_debug_stamps(2, buf)
v = [int(t) for t in buf[0:8]]
w = max(v[7], 1)
tot = v[0] + v[1] + v[2] + v[3]
mhz = tot / max(v[6], 1) * 100.0
print(f"{'infer' if infer else 'train'} step {t0.elapsed_time(t1) / R * 1e3:7.1f} us  waves/launch {v[7] // R:6d}  loop {v[0] / w:8.0f}  epi1 {v[1] / w:8.0f}  "
      f"gate {v[2] / w:8.0f}  pool {v[3] / w:8.0f}  wave life {v[6] / w / 100.0:7.1f} us  (clock {mhz:6.0f} MHz)")
