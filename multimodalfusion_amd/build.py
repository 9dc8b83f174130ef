"""Builds libmmf_amil.so (HIP kernels + C ABI) for gfx950, in-tree, with plain hipcc.

    python -m multimodalfusion_amd.build [--force] [--keep-temps]

hipcc cross-compiles without a GPU; the built .so travels to the GPU box with the tree.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmmf_amil.so")
OBJ = os.path.join(HERE, "_build")   # objects and -save-temps output (git- and gpurun-ignored)
SOURCES = ["mmf_api.hip", "mmf_amil_fwd.hip", "mmf_amil_bwd.hip", "mmf_amil_bf16.hip", "mmf_amil_bf16_fwd2.hip", "mmf_amil_bf16_dh2.hip", "mmf_small.hip", "mmf_mlp.hip", "mmf_maxnet.hip"]  # missing files are skipped
HEADERS = ["mmf_common.h", "mmf_gemm_core.h", "mmf_gemm_split.h", "mmf_gemm_dma.h", "mmf_kernels.h", "mmf_small.h", "mmf_mlp.h", "mmf_bf16.h",
           os.path.join("..", "..", "include", "mmf_amil.h")]
# Per-file flags.  mmf_amil_bf16_fwd2.hip: no SLP vectorisation, i.e. no packed-fp32 VALU instructions (v_pk_fma_f32 ...).  With
# them the kernel (two 4-wave workgroups per CU, a wave's vector work beside its SIMD partner's MFMA stream) returned wrong
# score partials in lanes 16-31 of the low register of a packed pair, a few tiles per launch, never with one workgroup per
# CU (tools/f2_debug.py, tools/f2_debug2.py).  Packed fp32 is no gain beside MFMAs anyway (MI355X_MICROARCH.md).
FILE_FLAGS = {"mmf_amil_bf16_fwd2.hip": ["-fno-slp-vectorize"], "mmf_amil_bf16_dh2.hip": ["-fno-slp-vectorize"]}
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wall", "-Wno-unused-function",
         "-Wno-unused-variable", "-Wno-unused-result"] + os.environ.get("MMF_EXTRA_FLAGS", "").split()


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, keep_temps: bool = False, verbose: bool = True) -> str:
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    hdrs = [os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    jobs = []
    os.makedirs(OBJ, exist_ok=True)
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + hdrs):
            cmd = [HIPCC] + FLAGS + FILE_FLAGS.get(s, []) + ["-c", src, "-o", obj]
            if keep_temps:
                cmd += ["-save-temps=obj", "-Rpass-analysis=kernel-resource-usage"]
            jobs.append(cmd)

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=OBJ)
        return cmd, r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for cmd, r in ex.map(run, jobs):
                if verbose and (r.returncode != 0 or keep_temps):
                    sys.stderr.write(r.stderr)
                if r.returncode != 0:
                    raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + r.stderr[-4000:])
    if keep_temps:
        # static checks of the hand-synchronised kernels in the ISA just written (tools/isa_check.py): no instruction may touch
        # the destination of a hand-issued load before its hand-placed wait, the waits must be covered by younger VM
        # operations, no scratch beside a hand-counted queue, no packed-fp32 VALU in the two units built without SLP
        tools = os.path.join(os.path.dirname(HERE), "tools")
        if os.path.exists(os.path.join(tools, "isa_check.py")):
            sys.path.insert(0, tools)
            import isa_check
            units = [os.path.join(OBJ, u + "-hip-amdgcn-amd-amdhsa-gfx950.s") for u in ("mmf_amil_bf16_fwd2", "mmf_amil_bf16_dh2")]
            bad = []
            for u in units:
                if os.path.exists(u):
                    bad += isa_check.check_file(u, verbose=verbose)
            if bad:
                raise RuntimeError("ISA check failed:\n  " + "\n  ".join(bad))
    if jobs or force or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed: " + r.stderr[-4000:])
    return LIB


if __name__ == "__main__":
    path = build(force="--force" in sys.argv, keep_temps="--keep-temps" in sys.argv)
    print(path)
