#!/usr/bin/env python3
"""Static checks of the hand-synchronised kernels in the -save-temps ISA (run by `python -m multimodalfusion_amd.build
--keep-temps`, which fails on a violation; or stand-alone on any .s file):

    python tools/isa_check.py [file.s ...]      default: the bf16 fused forward / K-dh units under multimodalfusion_amd/_build

The bf16 fused forward (csrc/mmf_amil_bf16_fwd2.hip) issues its weight-fragment loads from inline asm and counts their
completion by hand (s_waitcnt vmcnt(4) from inline asm): the compiler believes the destination registers are defined at
the asm statement, the data lands later.  That is only correct if

  1. no instruction between such a load and the next hand-placed vmcnt wait reads, copies or overwrites a destination
     register of the load (a v_mov placed there by the register allocator, a v_pk_* pair that overlaps, a spill);
  2. at least as many vector-memory operations are issued behind the last hand-issued load as the wait's immediate admits
     (vmcnt(4) proves the fragments have landed only if >= 4 younger operations are in the queue);
  3. the kernel has no scratch (a spill or a reload inside the loop is a VM operation of the compiler's own that the hand
     count does not know about, and a fragment register parked in scratch would be stored before its data has landed).

It also reports packed-fp32 VALU instructions (v_pk_*_f32) per kernel: the two hand-scheduled bf16 kernels are built
without SLP vectorisation (build.py: FILE_FLAGS) and must contain none.
"""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEFAULT = [os.path.join(ROOT, "multimodalfusion_amd", "_build", u + "-hip-amdgcn-amd-amdhsa-gfx950.s")
           for u in ("mmf_amil_bf16_fwd2", "mmf_amil_bf16_dh2")]
NO_PACKED = ("amil_fwd_fused2_bf16_kernel", "dh2_bf16_kernel")       # kernels that must hold no v_pk_*_f32
NO_SCRATCH = ("amil_fwd_fused2_bf16_kernel",)                          # kernels whose hand-counted queue forbids scratch

REG = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def vregs(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(1) is not None:
            out.add(int(m.group(1)))
        else:
            out.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return out


def kernels(s):
    """name -> list of (line_no, instruction text, inside_inline_asm)"""
    out = {}
    names = re.findall(r"^\s*\.amdhsa_kernel (\S+)", s, re.M)
    lines = s.split("\n")
    starts = {n: i for i, l in enumerate(lines) for n in names if l.startswith(n + ":")}
    for n, i in starts.items():
        body, in_asm = [], False
        for k in range(i + 1, len(lines)):
            l = lines[k]
            if l.startswith(".Lfunc_end"):
                break
            if "#ASMSTART" in l:
                in_asm = True
                continue
            if "#ASMEND" in l:
                in_asm = False
                continue
            t = l.split(";")[0].strip()
            if not t or t.startswith(".") or t.endswith(":") or t.startswith("//"):
                continue
            body.append((k + 1, t, in_asm))
        out[n] = body
    return out


def is_vm(t):
    return t.startswith(("buffer_load", "buffer_store", "global_load", "global_store", "flat_load", "flat_store",
                         "scratch_load", "scratch_store", "buffer_atomic", "global_atomic"))


def check_kernel(name, body, scratch_bytes):
    """-> list of violation strings"""
    bad = []
    short = name[:70]
    if any(k in name for k in NO_PACKED):
        n = sum(1 for _, t, _ in body if re.match(r"v_pk_\w+_f32", t))
        if n:
            bad.append(f"{short}: {n} packed-fp32 VALU instructions (v_pk_*_f32) in a kernel built to hold none")
    if any(k in name for k in NO_SCRATCH):
        ns = sum(1 for _, t, _ in body if t.startswith("scratch_"))
        if ns or (scratch_bytes or 0) > 0:
            bad.append(f"{short}: scratch in a kernel with a hand-counted vmcnt queue ({ns} scratch instructions, {scratch_bytes} B/lane)")
    # hand-issued loads: buffer loads with a VGPR destination inside an inline-asm block
    i = 0
    n_loads = n_waits = 0
    while i < len(body):
        ln, t, in_asm = body[i]
        if in_asm and t.startswith("buffer_load") and " lds" not in t:
            # a run of hand-issued loads, then everything up to the next hand-placed vmcnt wait
            dst = {}
            j = i
            younger = 0
            wait_at = None
            while j < len(body):
                ln2, t2, asm2 = body[j]
                if asm2 and t2.startswith("s_waitcnt") and "vmcnt(" in t2:
                    wait_at = j
                    break
                if asm2 and t2.startswith("buffer_load") and " lds" not in t2:
                    d = vregs(t2.split(",")[0])
                    touched = d & set(dst)
                    if touched:
                        bad.append(f"{short}: line {ln2}: hand-issued load overwrites v{sorted(touched)} of the load at line {dst[min(touched)]} before any wait")
                    for r in d:
                        dst[r] = ln2
                    srcs = vregs(",".join(t2.split(",")[1:]))
                    hit = srcs & set(dst)
                    if hit:
                        bad.append(f"{short}: line {ln2}: address of a hand-issued load reads v{sorted(hit)} still in flight")
                    younger = 0
                    n_loads += 1
                else:
                    if is_vm(t2):
                        younger += 1
                    if t2.startswith(("s_cbranch", "s_branch", "s_endpgm", "s_setpc")) and not asm2:
                        # the path from a hand-issued load to its wait must be straight-line code
                        bad.append(f"{short}: line {ln2}: branch between a hand-issued load (line {ln}) and its vmcnt wait")
                        break
                    hit = vregs(t2) & set(dst)
                    if hit:
                        bad.append(f"{short}: line {ln2}: `{t2}` touches v{sorted(hit)[:4]}... while the hand-issued load of line {dst[min(hit)]} is in flight")
                j += 1
            if wait_at is None:
                bad.append(f"{short}: hand-issued load at line {ln} is never followed by a hand-placed vmcnt wait")
                break
            n_waits += 1
            imm = int(re.search(r"vmcnt\((\d+)\)", body[wait_at][1]).group(1))
            if younger < imm:
                bad.append(f"{short}: line {body[wait_at][0]}: vmcnt({imm}) behind only {younger} younger VM operations: the fragments may still be in flight")
            i = wait_at + 1
            continue
        i += 1
    return bad, n_loads, n_waits


def scratch_sizes(s):
    out = {}
    for m in re.finditer(r"\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel", s, re.S):
        q = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", m.group(2))
        out[m.group(1)] = int(q.group(1)) if q else None
    return out


def check_file(path, verbose=True):
    s = open(path).read()
    sc = scratch_sizes(s)
    bad_all = []
    for name, body in kernels(s).items():
        bad, nl, nw = check_kernel(name, body, sc.get(name))
        npk = sum(1 for _, t, _ in body if re.match(r"v_pk_\w+_f32", t))
        if verbose:
            print(f"{os.path.basename(path)[:28]} {name[7:64]}: {len(body)} instructions, {nl} hand-issued loads in {nw} "
                  f"wait groups, {npk} v_pk_*_f32, scratch {sc.get(name)} B/lane -> {'OK' if not bad else str(len(bad)) + ' VIOLATIONS'}")
        bad_all += bad
    return bad_all


def main(paths):
    bad = []
    for p in paths:
        if not os.path.exists(p):
            print("missing:", p, "(build with --keep-temps first)")
            return 2
        bad += check_file(p)
    for b in bad:
        print("VIOLATION:", b)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] or DEFAULT))
