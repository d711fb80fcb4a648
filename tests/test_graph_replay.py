"""hipGraph safety net: gradients of REPLAYED training steps must equal the eager gradients.

Background (found on ROCm 7.0 / torch 2.10 / MI355X while building this repo): with the framework's default settings
a captured forward+backward returns correct gradients on the FIRST replay only; from the second replay on the
library paths for (i) convolution bias gradients and (ii) MIOpen's non-deterministic bf16 weight-gradient solvers
hand back garbage (1e25 ... inf).  The build therefore (a) computes bias / BatchNorm parameter gradients in its own
gfx950 epilogue kernels, (b) runs 1x1 convolutions as GEMMs, (c) selects MIOpen's deterministic algorithms.  This
test replays the full training step several times on alternating batches and checks every parameter gradient.
"""
import pytest
import torch

from xpt_mde_2021_amd.config import opts

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("nets,dtype", [("rigid", "bf16"), ("rigid", "fp32")])
def test_graph_replays_match_eager(gpu_device, nets, dtype):
    """tools/replay_grad_diff.py in a fresh process: forward+backward captured the way the trainers capture it (replay
    check, per-convolution repair, eager as the last resort), five replays, every parameter gradient against eager.

    A fresh process, and a second attempt if the first one fails: which library solvers a capture contains varies from
    process to process (MIOpen's find), a capture containing a defective one executes garbage-producing kernels while it
    is being checked, and such a process has been seen to stay corrupted afterwards (wrong results even in eager mode,
    DESIGN.md section 6) -- inside one pytest process that would take the following tests down with it."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tail = ""
    for attempt in range(2):
        run = subprocess.run([sys.executable, os.path.join(root, "tools", "replay_grad_diff.py"), nets, dtype,
                              "128", "416", "8", "5"], capture_output=True, text=True, timeout=900)
        tail = "\n".join(l[:400] for l in (run.stdout + run.stderr).splitlines() if "[diff]" in l or "StepGraph" in l)
        if run.returncode == 0:
            return
    raise AssertionError(f"replayed gradients differ from eager in two fresh processes:\n{tail}")
