"""Diagnostic (-DMMF_STAMPS -DMMF_STAMPS_LIGHT build, MMF_LIB_PATH=.../libmmf_stamps.so): where a wave of the split-operand
projection (unit 0) and of the TN kernel's plain tiles (unit 1) spends a chunk -- work (fragment reads, MFMAs, staging
pieces) vs waiting at the chunk barrier -- for the two waves of a SIMD separately.  s_memtime ticks are 10 ns."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MMF_GEMM", "1")
import torch
from multimodalfusion_amd import _lib
from bench import build_model, make_step
l = _lib.lib()
dev = torch.device("cuda", 0)
model = build_model(dev, False)
x = torch.randn(int(os.environ.get("N", 50000)), 1024, device=dev)
step = make_step(model, x, dev, None, 1)
for _ in range(3): step()
torch.cuda.synchronize()
buf = (C.c_uint64 * 8)()
for unit in (0, 1): l.mmf_debug_stamps(unit, buf)
R = 10
for _ in range(R): step()
torch.cuda.synchronize()
for unit, name, waves_lo in ((0, "linear_nt_split (224 tiles)", 224 * 4), (1, "tn_split plain tiles", 168 * 4)):
    l.mmf_debug_stamps(unit, buf)
    v = [int(t) for t in buf[:8]]
    chunks_per_wave = v[4] / R / (waves_lo / 4)        # [4] counts chunks once per workgroup
    for grp, tag in ((0, "waves 0-3"), (2, "waves 4-7")):
        work = v[grp] / R / waves_lo / max(chunks_per_wave, 1)
        bar = v[grp + 1] / R / waves_lo / max(chunks_per_wave, 1)
        print(f"{name:28s} {tag}: per chunk work {work*10:.0f} ns, barrier wait {bar*10:.0f} ns  (chunks per wave {chunks_per_wave:.1f})")
