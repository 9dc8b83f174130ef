#!/usr/bin/env python3
"""Static look at a kernel's loops in the -save-temps ISA (python -m multimodalfusion_amd.build --keep-temps --force):
for every backward branch whose body holds MFMAs: instruction, MFMA, scalar-branch and barrier counts.

    python tools/isa_loops.py mmf_amil_bwd tn_kernelINS_4TileILi256
"""
import re
import sys

def main(unit, key):
    s = open(f"multimodalfusion_amd/_build/{unit}-hip-amdgcn-amd-amdhsa-gfx950.s").read()
    for name in [n for n in re.findall(r"^\s*\.amdhsa_kernel (\S+)", s, re.M) if key in n]:
        i = s.index("\n" + name + ":")
        j = s.find(".Lfunc_end", i)
        lines, labels = [], {}
        for l in s[i:j].split("\n"):
            l = l.split(";")[0].strip()
            m = re.match(r"^(\.LBB\w+):", l)
            if m:
                labels[m.group(1)] = len(lines)
            elif l and not l.startswith((".", "//")) and not l.endswith(":"):
                lines.append(l)
        print(name[7:90], len(lines), "instructions")
        for k, l in enumerate(lines):
            if l.startswith(("s_cbranch", "s_branch")):
                d = labels.get(l.split()[-1])
                if d is not None and d < k:
                    body = lines[d:k]
                    nm = sum("v_mfma" in x for x in body)
                    if nm:
                        nb = sum(x.startswith(("s_cbranch", "s_branch")) for x in body)
                        print(f"   loop {d}..{k}: {k - d} instructions, {nm} mfma, {nb} branches, "
                              f"{sum(x.startswith('s_barrier') for x in body)} barriers, "
                              f"{sum(x.startswith('scratch_') for x in body)} scratch ops")

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
