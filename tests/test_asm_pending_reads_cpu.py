"""The fused HTSAT kernels request LDS reads with inline asm and wait for them later with counted `s_waitcnt lgkmcnt(N)`: the compiler does not
know that the destination registers are not valid yet, so a spill, a copy or a reuse of one of them between the request and the wait is a
silent wrong result (it happened in round 6 in a build for 256 registers: DESIGN 8.6.9).  `tools/probe/check_pending_lds_regs.py` replays the
gfx950 listings of the kernel sources and must find no instruction that touches a register with a read pending."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


# (file, kernels it must see at least): every translation unit whose kernels wait for inline-asm LDS reads with counted lgkmcnt
@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
@pytest.mark.parametrize("name,min_kernels", [("htsat_fused", 20), ("gemm", 30), ("attention", 5), ("attention_fwd", 4), ("attention_bwd_fused", 3),
                                              ("attention_bwd_fused8", 2), ("htsat", 10)])
def test_no_instruction_touches_a_register_with_an_lds_read_pending(tmp_path, name, min_kernels):
    listing = tmp_path / f"{name}.s"
    src = os.path.join(ROOT, "adt_str_amd", "csrc", f"{name}.hip")
    # the flags of adt_str_amd/csrc/Makefile (register allocation depends on them)
    out = subprocess.run([HIPCC, "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=fast", "-S", "--cuda-device-only", "-o", str(listing), src],
                         cwd=str(tmp_path), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    chk = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "probe", "check_pending_lds_regs.py"), str(listing)], capture_output=True, text=True)
    assert chk.returncode == 0, chk.stdout[-4000:]
    assert chk.stdout.count(" 0 instruction(s)") >= min_kernels, chk.stdout[-2000:]          # the file's kernels were seen
    shutil.rmtree(tmp_path, ignore_errors=True)
