"""The ctypes mirrors in multimodalfusion_amd/_lib.py must lay their fields out exactly as include/mmf_amil.h does: a
mismatch would not fail, it would hand the kernels the wrong pointers.  gcc compiles the header as plain C (the boundary is
a C ABI) and prints sizeof / offsetof of every field; ctypes must agree, and the header's ABI version must be the binding's."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mmf_amil.h")


def _c_layout(tmp_path, structs):
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "mmf_amil.h"', "int main(void) {"]
    for cname, fields in structs.items():
        lines.append(f'  printf("{cname} sizeof %zu\\n", sizeof({cname}));')
        for f in fields:
            lines.append(f'  printf("{cname} {f} %zu\\n", offsetof({cname}, {f}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.dirname(HEADER), str(src), "-o", str(exe)])
    out = {}
    for line in subprocess.check_output([str(exe)], text=True).splitlines():
        s, f, v = line.split()
        out[(s, f)] = int(v)
    return out


def test_ctypes_mirrors_match_the_c_header(tmp_path):
    from multimodalfusion_amd import _lib
    mirrors = {"mmf_amil_desc": _lib.AmilDesc, "mmf_amil_grads": _lib.AmilGrads, "mmf_surv_head": _lib.SurvHead,
               "mmf_nll_target": _lib.NllTarget, "mmf_xreduce_io": _lib.XReduceIO, "mmf_maxnet_desc": _lib.MaxnetDesc,
               "mmf_maxnet_grads": _lib.MaxnetGrads}
    structs = {c: [n for n, _ in m._fields_] for c, m in mirrors.items()}
    got = _c_layout(tmp_path, structs)
    import ctypes as C
    for cname, m in mirrors.items():
        assert got[(cname, "sizeof")] == C.sizeof(m), cname
        for n, _ in m._fields_:
            assert got[(cname, n)] == getattr(m, n).offset, (cname, n)


def test_header_is_plain_c_and_declares_what_the_binding_loads():
    from multimodalfusion_amd import _lib
    text = open(HEADER).read()
    declared = set(re.findall(r"\b(mmf_[a-z0-9_]+)\s*\(", text))
    missing = [s for s in _lib.SYMBOLS if s not in declared]
    assert not missing, missing
    assert "MMF_GEMM_F32" in text and "MMF_GEMM_BF16X3" in text
