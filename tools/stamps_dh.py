"""Diagnostic (stamps build: python tools/diag_build.py stamps): phases of the fp32 K-dh and K-lin workgroups.
    MMF_LIB_PATH=multimodalfusion_amd/_diag/libmmf_stamps.so python tools/stamps_dh.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from multimodalfusion_amd import _lib
dev = torch.device("cuda", 0)
model = bench.build_model(dev, False)
x = torch.randn(int(sys.argv[1]) if len(sys.argv) > 1 else 50000, 1024, device=dev)
step = bench.make_step(model, x, dev)
for _ in range(3): step()
torch.cuda.synchronize()
buf = (C.c_uint64 * 8)()
l = _lib.lib()
l.mmf_debug_stamps(0, buf); l.mmf_debug_stamps(1, buf)
for _ in range(10): step()
torch.cuda.synchronize()
l.mmf_debug_stamps(1, buf)
v = [int(t) for t in buf[:8]]; w = max(v[7], 1)
if v[2] or v[3]:
    print(f"tn_kernel workgroup life (cycles): plain tiles {v[0]/max(v[2],1):9.0f} ({v[2]//10} per launch)   gate tiles {v[1]/max(v[3],1):9.0f} ({v[3]//10} per launch)")
print(f"bwd_dh   per wave cycles: prologue {v[4]/w:8.0f}  main loop {v[5]/w:8.0f}  epilogue {v[6]/w:8.0f}  (waves/launch {v[7]//10})")
l.mmf_debug_stamps(0, buf)
v = [int(t) for t in buf[:8]]; w = max(v[7], 1)
print(f"linear_nt per wave cycles: entry->loaders {v[4]/w:6.0f}  first stage {v[2]/max(v[3],1):7.0f}  main loop (incl. first stage) {v[5]/w:8.0f}  epilogue {v[6]/w:8.0f}  (waves/launch {v[7]//10})")
