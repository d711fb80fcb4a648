"""The hand-pipelined forward kernel issues its vector loads by inline asm and places the s_waitcnt itself; the
compiler does not know those registers have loads in flight.  This test compiles the kernel to ISA and checks, in
program order, that no instruction reads a register before an s_waitcnt covers its load (tools/check_inflight_regs.py)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_no_read_of_in_flight_registers(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_inflight_regs as chk
    from xpt_mde_2021_amd.csrc import build as xpt_build
    asm = tmp_path / "fused.s"
    cmd = [HIPCC, *xpt_build.CODEGEN_FLAGS, "--cuda-device-only", "-S", "-o", str(asm),
           os.path.join(ROOT, "xpt_mde_2021_amd", "csrc", "xpt_fused.hip")]
    subprocess.run(cmd, check=True, capture_output=True, timeout=600)
    assert chk.main(str(asm), "fused_fwd_kernelILb0ELb1E") == 0
