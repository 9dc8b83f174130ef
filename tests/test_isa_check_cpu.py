"""CPU: tools/isa_check.py (the static guard `build.py --keep-temps` runs on the hand-synchronised bf16 kernels) must catch what
it is there to catch.  Synthetic ISA: a kernel whose hand-issued load is followed by a register copy of its destination before
the hand-placed wait, one whose wait is not covered by enough younger VM operations, one with scratch and packed fp32, and a
clean one."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_check  # noqa: E402


def _kernel(name, body, scratch=0):
    return f"""
\t.text
{name}:
{body}
\ts_endpgm
.Lfunc_end_{name}:
\t.amdhsa_kernel {name}
\t\t.amdhsa_private_segment_fixed_size {scratch}
\t.end_amdhsa_kernel
"""


LOAD = "\t;;#ASMSTART\n\tbuffer_load_dwordx4 v[10:13], v1, s[4:7], s8 offen offset:0\n\t;;#ASMEND\n"
DMA = "\tbuffer_load_dwordx4 v20, s[12:15], s9 offen lds\n"
WAIT4 = "\t;;#ASMSTART\n\ts_waitcnt vmcnt(4)\n\t;;#ASMEND\n"


def _check(tmp_path, text):
    p = tmp_path / "k.s"
    p.write_text(text)
    return isa_check.check_file(str(p), verbose=False)


def test_clean_kernel_passes(tmp_path):
    body = LOAD + DMA * 4 + "\tv_mfma_f32_32x32x16_bf16 v[30:45], v[50:53], v[54:57], v[30:45]\n" + WAIT4 + "\tv_mov_b32_e32 v2, v10\n"
    assert _check(tmp_path, _kernel("_ZN3mmf27amil_fwd_fused2_bf16_kernelILb1EEEv", body)) == []


def test_copy_of_an_unanswered_load_is_flagged(tmp_path):
    body = LOAD + DMA * 4 + "\tv_mov_b32_e32 v99, v11\n" + WAIT4
    bad = _check(tmp_path, _kernel("_ZN3mmf27amil_fwd_fused2_bf16_kernelILb1EEEv", body))
    assert len(bad) == 1 and "in flight" in bad[0] and "v_mov_b32_e32 v99, v11" in bad[0]


def test_uncovered_wait_branch_scratch_and_packed_ops_are_flagged(tmp_path):
    body = LOAD + DMA * 2 + WAIT4                       # vmcnt(4) behind only two younger VM operations
    bad = _check(tmp_path, _kernel("_ZN3mmf27amil_fwd_fused2_bf16_kernelILb1EEEv", body))
    assert any("younger VM operations" in b for b in bad)
    body = LOAD + "\ts_cbranch_scc1 .LBB0_3\n" + DMA * 4 + WAIT4
    bad = _check(tmp_path, _kernel("_ZN3mmf27amil_fwd_fused2_bf16_kernelILb1EEEv", body))
    assert any("branch between" in b for b in bad)
    body = "\tscratch_store_dword off, v3, s0\n\tv_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[2:3]\n"
    bad = _check(tmp_path, _kernel("_ZN3mmf27amil_fwd_fused2_bf16_kernelILb0EEEv", body, scratch=16))
    assert any("scratch" in b for b in bad) and any("packed-fp32" in b for b in bad)
    # other kernels may hold packed fp32 and scratch: the rules are for the two hand-scheduled units
    assert _check(tmp_path, _kernel("_ZN3mmf13reduce_kernelENS_12ReduceParamsE", body, scratch=16)) == []
