"""Diagnostic (-DMMF_STAMPS build): phase cycles of linear_nt alone.  N=50000 M/K as the path head."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from multimodalfusion_amd import _lib
l = _lib.lib()
N = int(os.environ.get("N", 50000))
x = torch.randn(N, 1024, device="cuda"); W = torch.randn(256, 1024, device="cuda") * 0.03; b = torch.zeros(256, device="cuda")
y = torch.empty(N, 256, device="cuda")
segs = (C.c_void_p * 1)(x.data_ptr())
def run():
    rc = l.mmf_linear_forward(segs, 1, 1024, N, W.data_ptr(), b.data_ptr(), 256, 1, 0.25, 7, 0, None, y.data_ptr(), None, 0, None, 0, None)
    assert rc == 0
for _ in range(3): run()
torch.cuda.synchronize()
buf = (C.c_uint64 * 8)(); l.mmf_debug_stamps(0, buf)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): run()
e1.record(); torch.cuda.synchronize()
l.mmf_debug_stamps(0, buf)
load, mfma, store, bar, n, tot, epi, waves = [int(v) for v in buf[:8]]
n = max(n, 1); waves = max(waves, 1)
print(f"linear_nt {e0.elapsed_time(e1)/10*1e3:.1f} us/launch; per wave-chunk: compute+hooks {mfma/n:.0f}  barrier {bar/n:.0f} cycles;"
      f" per wave: mainloop {tot/waves:.0f}  epilogue {epi/waves:.0f} cycles (waves {waves//10} per launch)")
