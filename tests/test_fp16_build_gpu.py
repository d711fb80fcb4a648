"""BASELINE configs[4] names "fp16 convs + fp32 loss accumulation".  libxpt_hip_f16.so is the same source tree compiled with
IEEE-half activations (csrc/xpt_common.h, XPT_HALF_F16): same kernels, same C ABI, `dtype == 1` means half instead of
bfloat16; fp32 masters, losses and optimizer as always; a static loss scale seeds the backward pass (config.LOSS_SCALE_FP16).
One 16-bit format per process, so everything here runs in child processes with XPT_HALF=fp16."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu

# every kernel family against its fp32 torch reference / the oracle, unchanged bars (half has three more mantissa bits than bf16)
KERNEL_SUITES = ["tests/test_conv_igemm_gpu.py", "tests/test_cell_tail_gpu.py", "tests/test_grad_sink.py", "tests/test_hip_parity.py",
                 "tests/test_adam_parity.py"]


def _child(args, timeout, **env):
    run = subprocess.run([sys.executable, *args], cwd=ROOT, capture_output=True, text=True, timeout=timeout,
                         env=dict(os.environ, XPT_HALF="fp16", **env))
    return run


def test_kernel_parity_suites_pass_on_the_half_precision_build(gpu_device):
    run = _child(["-m", "pytest", *KERNEL_SUITES, "-m", "gpu", "-q", "-p", "no:cacheprovider"], 1500)
    tail = (run.stdout + run.stderr)[-3000:]
    assert run.returncode == 0, tail
    assert " passed" in run.stdout and "failed" not in run.stdout.splitlines()[-1], tail


def _losses(mode, steps, **env):
    run = _child([os.path.join("tools", "determinism_train.py"), mode, "noaug", str(steps)], 900, XPT_DET_DTYPE="fp16", **env)
    assert run.returncode == 0, (run.stdout + run.stderr)[-2500:]
    losses = [l for l in run.stdout.splitlines() if l.startswith("LOSSES")][0].split()[3:]
    psum = [l for l in run.stdout.splitlines() if l.startswith("PARAMSUM")][0]
    return [float(v) for v in losses], psum, run.stdout


def test_fp16_training_step_is_captured_equals_eager_and_learns(gpu_device):
    """Ten full training steps of the bench configuration with IEEE-half convolutions: the captured step (own kernels, no
    memset node, the one-pass march with the loss scale in its gradient hint) and the eager step give the SAME losses and
    final weights; the loss falls as it does in bf16; without the loss scale the first update already differs (per-pixel
    gradients of ~1e-7 flush to zero in half), which is what the scale is for."""
    graph, gsum, out = _losses("graph", 10)
    assert "CAPTURED True own-kernels" in out and "'memset': 0" in out, out[-1500:]
    eager, esum, _ = _losses("eager", 10)
    assert graph == eager and gsum == esum, f"graph {graph}\neager {eager}"
    assert all(v == v and v < 10 for v in graph) and graph[-1] < 0.5 * graph[0], graph
    unscaled, _, _ = _losses("graph", 3, XPT_LOSS_SCALE_FP16="1")
    assert unscaled[0] == graph[0] and unscaled[1] != graph[1], (unscaled, graph[:3])


def test_mixed_shape_stereo_steps_c5_in_fp16(gpu_device):
    """configs[4] as written: the mixed-shape stereo + mono step of tests/test_configs_gpu.py on the half-precision build."""
    run = _child(["-m", "pytest", "tests/test_configs_gpu.py::test_mixed_shape_stereo_steps_c5", "-m", "gpu", "-x", "-q", "-p",
                  "no:cacheprovider"], 900)
    assert run.returncode == 0, (run.stdout + run.stderr)[-3000:]
