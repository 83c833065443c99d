"""CPU-side audit of the emitted gfx950 ISA of gemm.hip (no GPU needed: hipcc cross-compiles).

Several GEMM kernels issue register-destination loads from inline asm so that hipcc does not track them (a tracked load
makes it drain vmcnt to 0 where the value is first used, which would stall the LDS-DMA pipeline): the bias pair of
gemm_px_kernel, the residual pieces of gemm_x3r_kernel, the residual rows of gemm_fp8p_kernel<RESID_F32>; conv3x3_c64_kernel
issues its LDS fragment reads the same way (counted lgkmcnt).  Their
correctness rests on NO instruction touching a destination register before the hand-counted `s_waitcnt vmcnt` that covers
the load -- which the source enforces by naming the registers as read-write operands of that wait statement, and which this
test checks on the code the compiler actually emitted (tools/isa_async_reg_check.py: every path from each load, around
backward branches too)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from tools.isa_async_reg_check import audit  # noqa: E402

SRC = os.path.join(ROOT, "hamer_yolo_amd", "csrc", "gemm.hip")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def gemm_isa(tmp_path_factory):
    if not os.path.exists(HIPCC):
        pytest.skip("hipcc not installed")
    out = str(tmp_path_factory.mktemp("isa") / "gemm.s")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S", "--cuda-device-only", SRC, "-o", out],
                   check=True, capture_output=True, timeout=900)
    return open(out).read()


@pytest.mark.parametrize("kernel,min_loads,kind", [
    ("gemm_px_kernelI4TF16Li0ELb1E", 2, "vm"), ("gemm_px_kernelI4TF16Li1ELb1E", 2, "vm"), ("gemm_px_kernelI5TBf16Li0ELb1E", 2, "vm"),
    ("gemm_px_kernelI5TBf16Li1ELb1E", 2, "vm"), ("gemm_px_kernelI4TF16Li0ELb0E", 2, "vm"), ("gemm_px_kernelI4TF16Li1ELb0E", 2, "vm"),
    ("gemm_x3r_kernelI4TF16E", 32, "vm"), ("gemm_x3r_kernelI5TBf16E", 32, "vm"),
    ("gemm_fp8p_kernelILi2E", 16, "vm"),
    # conv3x3_c64_kernel: the 72 fragment reads of a tile are issued from asm (ds_read_b128) and released by counted lgkmcnt waits
    ("conv3x3_c64_kernelI4TF16Li1E", 72, "lds"), ("conv3x3_c64_kernelI5TBf16Li1E", 72, "lds"),
    ("conv3x3_s2c32_kernelI4TF16Li1E", 36, "lds"), ("conv3x3_s2c32_kernelI5TBf16Li1E", 36, "lds"),
    ("conv3x3_s2c64_kernelI4TF16Li1E", 72, "lds"),
    ("conv1x1_wreg_kernelI4TF16Li8ELi4ELi1E", 32, "lds"), ("conv1x1_wreg_kernelI4TF16Li4ELi2ELi1E", 16, "lds"), ("conv1x1_wreg_kernelI5TBf16Li8ELi2ELi1E", 16, "lds"),
])
def test_asm_loaded_registers_are_fenced(gemm_isa, kernel, min_loads, kind):
    ok, report, n = audit(gemm_isa, kernel, kind)
    assert n >= min_loads, f"{kernel}: expected at least {min_loads} asm-issued register loads, found {n}\n{report}"
    assert ok, report


def test_audit_tool_catches_an_unfenced_use():
    """The checker itself: a destination read with no hand-written wait on the path is a finding; behind one it is not; a
    use reached only around a loop's back edge is still found."""
    bad = """
_Z3badv:
\t;;#ASMSTART
\tglobal_load_dword v5, v1, s[2:3]
\t;;#ASMEND
\tv_add_f32 v6, v5, v5
\ts_endpgm
"""
    good = bad.replace("\tv_add_f32", "\t;;#ASMSTART\n\ts_waitcnt vmcnt(0)\n\t;;#ASMEND\n\tv_add_f32")
    loop = """
_Z4loopv:
.LBB0_1:
\tv_mov_b32 v9, v5
\t;;#ASMSTART
\tglobal_load_dword v5, v1, s[2:3]
\t;;#ASMEND
\ts_cbranch_scc1 .LBB0_1
\t;;#ASMSTART
\ts_waitcnt vmcnt(0)
\t;;#ASMEND
\tv_add_f32 v6, v5, v5
\ts_endpgm
"""
    assert not audit(bad, "bad")[0]
    assert audit(good, "bad")[0]
    assert not audit(loop, "loop")[0]


def test_asm_issued_stores_carry_their_wait_states(gemm_isa):
    """The direct convolution kernels issue their 16-byte output stores from inline asm (uniform SGPR base + lane offset).  Two
    hazards hipcc's recogniser cannot handle for an instruction it does not see: (1) a store of more than 64 bits followed by a
    write of its data registers needs wait states -- every such store must be followed by `s_nop 1` inside the same asm
    statement (without it conv1x1_wreg_kernel wrote wrong first dwords in lanes 12-15); (2) a VALU write of an SGPR (v_readfirstlane)
    must not sit within 5 instructions in front of a memory instruction that reads it as its base."""
    import re
    n = 0
    for m in re.finditer(r"^(_ZN\w*(conv3x3_c64|conv3x3_s2c32|conv3x3_s2c64|conv1x1_wreg)_kernel\w*):(.*?)\.amdhsa_kernel", gemm_isa, re.S | re.M):
        lines = [l.strip() for l in m.group(3).splitlines() if l.strip() and not l.strip().startswith((";", "."))]
        for i, l in enumerate(lines):
            mm = re.match(r"global_store_dwordx4 v\d+, v\[\d+:\d+\], s\[(\d+):(\d+)\]", l)
            if not mm:
                continue
            n += 1
            assert lines[i + 1].startswith("s_nop 1"), (m.group(1), l, lines[i + 1])
            base = {int(mm.group(1)), int(mm.group(2))}
            for back in lines[max(0, i - 5):i]:
                d = re.match(r"v_read(?:first)?lane\w* s(\d+)", back)
                assert not (d and int(d.group(1)) in base), (m.group(1), back, l)
    assert n >= 40, n


def test_direct_convolution_kernels_do_not_spill(gemm_isa):
    """Their hand-counted `s_waitcnt vmcnt(N)` (N = the stores of the previous epilogue) assume that the copies and those stores
    are the ONLY vector-memory instructions of the tile loop: a register spill (scratch_store / scratch_load count in vmcnt too)
    would silently break the count.  They sit at 199-250 VGPRs, so this is a compile-time property worth pinning."""
    import re
    seen = 0
    for m in re.finditer(r"^(_ZN\w*(conv3x3_c64|conv3x3_s2c32|conv3x3_s2c64|conv1x1_wreg|conv3x3_direct)_kernel\w*):(.*?)\.amdhsa_kernel", gemm_isa, re.S | re.M):
        seen += 1
        assert "scratch_" not in m.group(3), m.group(1)
        priv = re.search(re.escape(m.group(1)) + r"\.private_seg_size, (\d+)", gemm_isa)
        assert priv and int(priv.group(1)) == 0, m.group(1)
    assert seen >= 16, seen
