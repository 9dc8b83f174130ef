#!/usr/bin/env python3
"""Kernel resource usage (VGPRs, scratch bytes per lane, LDS) of every kernel in the --keep-temps ISA under
multimodalfusion_amd/_build:  python tools/kres.py [substring ...]   (only kernels whose name holds a substring)"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
keys = sys.argv[1:]
for f in sorted(glob.glob(os.path.join(ROOT, "multimodalfusion_amd", "_build", "*-hip-amdgcn-amd-amdhsa-gfx950.s"))):
    s = open(f).read()
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        name, body = m.group(1), m.group(2)
        if keys and not any(k in name for k in keys):
            continue
        g = lambda k: (re.search(r"\.amdhsa_" + k + r" (\d+)", body) or [None, "?"])[1]
        try:
            dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
        except Exception:
            dem = name
        dem = dem.replace("mmf::", "").replace("void ", "")[:110]
        print(f"{dem:110s} vgpr {g('next_free_vgpr'):>4} scratch {g('private_segment_fixed_size'):>5} lds {g('group_segment_fixed_size'):>6}")
