#!/usr/bin/env python3
"""Diagnostic (timing-only, WRONG results) builds of the library, to split a kernel's time into its phases:

    python tools/diag_build.py            # -> multimodalfusion_amd/_diag/libmmf_{noload,nomfma}.so
    MMF_LIB_PATH=multimodalfusion_amd/_diag/libmmf_noload.so python bench.py --no-cpu-baseline ...

tune:   the product library with the launchers' MMF_* environment overrides compiled in (csrc/mmf_common.h: tune_int);
        the shipped library reads no environment variable -- every sweep script under tools/ loads this one
noload: the main loops stage nothing after the first chunk (MFMA + LDS reads + epilogue only)
nomfma: the main loops issue no MFMA (global loads + LDS writes + barriers + epilogue only)
nofrag: noload + the fp32 core reads its LDS fragments once per chunk only (MFMA + barriers + epilogue)
nogload: the fp32 main loops write stale registers to LDS and issue no global loads after the first chunk
epi1: row-major epilogues write only their first row block
nobar / nosched: nofrag + no barrier between chunks / + no sched_barrier around the MFMA blocks
nostore / noepi: the projection kernel's epilogue without its global stores / no epilogue at all
s_*: the same for the split-operand core (mmf_gemm_split.h): no MFMA / no staging at all / no global loads / LDS writes of
     unsplit data / no staging and no barrier
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from multimodalfusion_amd import build as B   # noqa: E402

# round 4: the fused bf16 forward built WITH the SLP vectoriser (the shipped library builds it without: build.py FILE_FLAGS),
# alone and with one suspect removed at a time -- tools/f2_slp_hazard.sh runs the determinism stress on each
#   f2slp      SLP on: 316 v_pk_*_f32 in the kernel (the round-3 corruption)
#   f2slp_wz   + s_waitcnt 0 behind every memory instruction (is it a counter the compiler mis-tracks?)
#   f2slp_nopk + the packed-fp32 instructions switched off in the backend (SLP's reordering without v_pk_*)
#   f2cond / f2cond_slp   the weight refill of the main loop as a conditional definition of the asm-loaded registers (what
#              the branch-free loop of the same round-3 commit replaced), without / with SLP
NO_FILE_FLAGS = {"f2slp", "f2slp_wz", "f2slp_nopk", "f2cond_slp"}
VARIANTS = {"f2cond": ["-DMMF_F2_COND_LOAD"], "f2cond_slp": ["-DMMF_F2_COND_LOAD"],
            "f2slp": [], "f2slp_wz": ["-mllvm", "-amdgpu-waitcnt-forcezero=1"],
            "f2slp_nopk": ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"],
            "tune": ["-DMMF_TUNE"],      # the product kernels + the MMF_* environment overrides of the launch plans (tools/README.md)
            "noload": ["-DMMF_DIAG_NOLOAD"], "nomfma": ["-DMMF_DIAG_NOMFMA"],
            "stamps": ["-DMMF_STAMPS", "-DMMF_STAMPS_LIGHT"],
            "mxnofence": ["-DMMF_STAMPS", "-DMMF_STAMPS_LIGHT", "-DMMF_MX_NOFENCE"],   # omic step: what the barriers' fences cost (WRONG results)
            "f2noact": ["-DMMF_F2_GATE_NOACT"], "f2nown": ["-DMMF_F2_GATE_NOWN"], "f2nomm": ["-DMMF_F2_GATE_NOMM"],
            "f2prio0": ["-DMMF_F2_PRIO=0"], "f2prio3": ["-DMMF_F2_PRIO=3"],
            "gprio0": ["-DMMF_GEMM_PRIO=0"],
            "f2nogate": ["-DMMF_F2_GATE_NOACT", "-DMMF_F2_GATE_NOWN", "-DMMF_F2_GATE_NOMM"],
            "f2base": ["-DMMF_F2_STAGE_BASE=16384"], "f2nop": ["-DMMF_DMA_M0NOP"],
            "f2noslp": ["-fno-slp-vectorize"],
            "f2nostore": ["-DMMF_F2_NOSTORE"], "f2nohstore": ["-DMMF_F2_NOHSTORE"], "f2nostores": ["-DMMF_F2_NOSTORE", "-DMMF_F2_NOHSTORE"],
            "f2r1": ["-DMMF_F2_REV_STORE_BRANCH"], "f2r2": ["-DMMF_F2_REV_GPAR"], "f2r3": ["-DMMF_F2_REV_SCHED"],
            "f2dbg": ["-DMMF_F2_DEBUG"],     # MMF_F2_DEBUG_MASK leaves phases of the bf16 fused forward out (mmf_amil_bf16_fwd2.hip)
            "nostore": ["-DMMF_DIAG_NOSTORE"], "noepi": ["-DMMF_DIAG_NOEPI"],
            "nofrag": ["-DMMF_DIAG_NOLOAD", "-DMMF_DIAG_NOFRAG"],
            "epi1": ["-DMMF_DIAG_EPI1"], "nogload": ["-DMMF_DIAG_NOGLOAD"],
            "nobar": ["-DMMF_DIAG_NOLOAD", "-DMMF_DIAG_NOFRAG", "-DMMF_DIAG_NOBAR"],
            "s_nomfma": ["-DMMF_SDIAG_NOMFMA"], "s_nostage": ["-DMMF_SDIAG_NOSTAGE"], "s_nogload": ["-DMMF_SDIAG_NOGLOAD"],
            "s_nosplit": ["-DMMF_SDIAG_NOSPLIT"], "s_noldsw": ["-DMMF_SDIAG_NOLDSW"], "s_nobar2": ["-DMMF_SDIAG_NOBAR"], "s_nofrag": ["-DMMF_SDIAG_NOFRAG"],
            "s_nofragbar": ["-DMMF_SDIAG_NOFRAG", "-DMMF_SDIAG_NOBAR"], "s_nobar": ["-DMMF_SDIAG_NOSTAGE", "-DMMF_SDIAG_NOBAR"],
            "nosched": ["-DMMF_DIAG_NOLOAD", "-DMMF_DIAG_NOFRAG", "-DMMF_DIAG_NOBAR", "-DMMF_DIAG_NOSCHED"]}


def main():
    out_dir = os.path.join(B.HERE, "_diag")
    os.makedirs(out_dir, exist_ok=True)
    for name, flags in VARIANTS.items():
        if len(sys.argv) > 1 and name not in sys.argv[1:]:
            continue
        objdir = os.path.join(B.OBJ, "diag_" + name)
        os.makedirs(objdir, exist_ok=True)
        jobs = []
        objs = []
        if name != "tune" and "-DMMF_TUNE" not in flags:
            flags = flags + ["-DMMF_TUNE"]        # every diagnostic library honours the overrides too
        for s in B.SOURCES:
            obj = os.path.join(objdir, s.replace(".hip", ".o"))
            objs.append(obj)
            ff = [] if name in NO_FILE_FLAGS else B.FILE_FLAGS.get(s, [])
            jobs.append([B.HIPCC] + B.FLAGS + ff + flags + ["-c", os.path.join(B.CSRC, s), "-o", obj])
        with ThreadPoolExecutor(max_workers=4) as ex:
            for r in ex.map(lambda c: subprocess.run(c, capture_output=True, text=True), jobs):
                if r.returncode != 0:
                    raise SystemExit(r.stderr[-3000:])
        lib = os.path.join(out_dir, f"libmmf_{name}.so")
        subprocess.check_call([B.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
        print(lib)


if __name__ == "__main__":
    main()
