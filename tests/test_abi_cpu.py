"""CPU-side checks of the C ABI: the library loads, exports every symbol include/xpt_hip.h declares,
and rejects bad arguments before touching the GPU (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "xpt_hip.h")


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    from xpt_mde_2021_amd.hip import lib as xl
    if not os.path.isfile(xl.LIB_PATH):
        ge.build()
    return xl.load()


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(xpt_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound(lib):
    from xpt_mde_2021_amd.hip import lib as xl
    syms = declared_symbols()
    assert len(syms) >= 16
    for name in syms:
        assert hasattr(lib, name), f"{name} declared in xpt_hip.h but not exported"
        assert name in xl.SIGNATURES, f"{name} has no ctypes signature"
    assert set(xl.SIGNATURES) == set(syms)


def test_the_half_precision_build_exports_the_same_abi(lib):
    """libxpt_hip_f16.so (the same sources with IEEE-half activations, BASELINE configs[4]) sits next to libxpt_hip.so, exports
    every declared symbol and reports its format; the default library reports bfloat16."""
    from xpt_mde_2021_amd.hip import lib as xl
    assert lib.xpt_half_format() == 0
    assert os.path.isfile(xl.LIB_PATH_F16), "csrc/build.py builds both libraries"
    f16 = ctypes.CDLL(xl.LIB_PATH_F16)
    for name in declared_symbols():
        assert hasattr(f16, name), f"{name} missing from libxpt_hip_f16.so"
    f16.xpt_half_format.restype = ctypes.c_int
    assert f16.xpt_half_format() == 1


def test_version_and_arch(lib):
    assert lib.xpt_abi_version() >= 1
    assert lib.xpt_build_arch() == b"gfx950"


def test_bad_arguments_are_rejected_without_gpu(lib):
    null = None
    one = ctypes.c_void_p(16)   # never dereferenced: argument checks fail first
    assert lib.xpt_pose_rvec2matr_fwd(null, one, 4, null) == -1
    assert lib.xpt_pose_rvec2matr_fwd(one, one, 0, null) == -2
    assert lib.xpt_warp_fwd(one, one, one, one, null, 1, 1, 4, 4, 1.0, null) == -1
    assert lib.xpt_warp_fwd(one, one, one, one, one, 1, 0, 4, 4, 1.0, null) == -2
    assert lib.xpt_warp_fwd(one, one, one, one, one, 1, 1, 4, 4, 0.0, null) == -2
    assert lib.xpt_resize_down_fwd(one, one, 1, 6, 8, 3, 4, null) == -2      # 6 % 4 != 0
    assert lib.xpt_bilinear_fwd(one, one, null, one, 1, 1, 4, 4, 3, 4, null) == -3
    assert lib.xpt_photo_fwd(7, one, one, one, null, null, 0, 1, 1, 4, 4, null) == -3
    assert lib.xpt_photo_fwd(0, one, one, null, null, null, 0, 1, 1, 4, 4, null) == -1
    assert lib.xpt_photo_fwd(0, one, one, null, one, one, 0, 1, 1, 4, 4, null) == -4
    assert lib.xpt_photo_bwd(2, one, one, one, one, one, one, 10 ** 9, 1, 1, 4, 4, null) == -3   # both grads given
    assert lib.xpt_smooth_fwd(one, one, one, one, 100, 1, 1, 8, 4.0, 0, null) == -2           # h < 2
    assert lib.xpt_warp_bwd(one, one, one, one, one, one, one, one, 1, 2, 4, 64, 64, 1.0, null) == -4
    assert lib.xpt_warp_bwd_workspace_floats(2, 4, 64, 64) == 2 * 4 * 16 * 12
    assert lib.xpt_photo_workspace_floats(2, 4, 8, 8) == 2 * 4 * 64 * 9


def test_round2_entry_points_reject_bad_arguments_without_gpu(lib):
    """The multi-scale / fused entry points added in round 2: argument checks come before any launch."""
    null = None
    one = ctypes.c_void_p(16)
    four = (ctypes.c_void_p * 4)(16, 16, 16, 16)
    hw = (ctypes.c_int * 4)(8, 8, 8, 8)
    n64 = (ctypes.c_longlong * 4)(8, 8, 8, 8)
    # smoothness, all scales: at most 4 scales, h and w >= 2, workspace large enough
    assert lib.xpt_smooth_ms_fwd(5, four, four, one, one, 10 ** 6, 1, hw, hw, 4.0, 0, null) == -2
    assert lib.xpt_smooth_ms_fwd(2, four, four, one, one, 1, 1, hw, hw, 4.0, 0, null) == -4
    assert lib.xpt_smooth_ms_fwd(2, four, four, null, one, 10 ** 6, 1, hw, hw, 4.0, 0, null) == -1
    assert lib.xpt_smooth_ms_bwd(2, four, four, null, four, 1, hw, hw, 4.0, 0, null) == -1
    # depth activation, all scales
    assert lib.xpt_depth_head_ms_fwd(0, four, four, four, n64, null) == -2
    assert lib.xpt_depth_head_ms_bwd(2, four, null, four, four, n64, null) == -1
    # image pyramids: at most 10 jobs, frames inside the snippet, even factors that divide the image
    jobs = (ctypes.c_int * 2)
    outs = (ctypes.c_void_p * 2)(16, 16)
    assert lib.xpt_image_pyramids(one, 1, 5, 8, 8, 11, jobs(0, 4), jobs(4, 1), jobs(1, 1), outs, null) == -2
    assert lib.xpt_image_pyramids(one, 1, 5, 8, 8, 2, jobs(0, 4), jobs(4, 2), jobs(1, 1), outs, null) == -2      # frame 5 of 5
    assert lib.xpt_image_pyramids(one, 1, 5, 8, 8, 2, jobs(0, 4), jobs(4, 1), jobs(3, 1), outs, null) == -2      # odd factor
    assert lib.xpt_image_pyramids(null, 1, 5, 8, 8, 2, jobs(0, 4), jobs(4, 1), jobs(1, 1), outs, null) == -1
    # total-loss merge: at most 64 terms and types
    assert lib.xpt_merge_total_fwd(65, one, one, one, one, one, 8, 3, null) == -2
    assert lib.xpt_merge_total_fwd(2, null, one, one, one, one, 8, 3, null) == -1
    assert lib.xpt_merge_total_bwd(2, one, one, one, 0, null) == -2
    # exact 2x up-sampling / global average pooling: dtype 0 or 1
    assert lib.xpt_upsample2x_fwd(one, one, 1, 4, 4, 2, null) == -3
    assert lib.xpt_upsample2x_bwd(one, 0, one, 1, 4, 4, 0, null) == -2        # pixel pitch < 1
    assert lib.xpt_global_avgpool_fwd(one, one, 1, 4, 8, 5, null) == -3
    assert lib.xpt_global_avgpool_bwd(null, one, 1, 4, 8, 0, null) == -1
    # fused pointwise backward: a data gradient needs the weight; partial buffers must hold every split
    args = (one, null, null, one, one)
    assert lib.xpt_conv1x1_bn_bwd_fused(*args, null, one, one, one, 1e-3, one, one, 10 ** 9, one, 10 ** 9, 64, 8, 8, 8, 0, 0,
                                        8, null) == -1
    assert lib.xpt_conv1x1_bn_bwd_fused(*args, one, one, one, one, 1e-3, one, one, 1, one, 10 ** 9, 64, 8, 8, 8, 0, 0, 8,
                                        null) == -4
    assert lib.xpt_conv1x1_bn_bwd_fused(*args, one, one, one, one, 1e-3, one, one, 10 ** 9, one, 10 ** 9, 64, 8, 8, 4, 0, 0,
                                        8, null) == -2                          # gradient row pitch < channels


def test_ops_refuse_cpu_tensors(lib):
    import torch
    from xpt_mde_2021_amd.hip import ops
    from xpt_mde_2021_amd.hip.lib import XptHipError
    with pytest.raises(XptHipError):
        ops.pose_rvec2matr(torch.zeros(2, 4, 6))
    with pytest.raises(XptHipError):
        ops.photometric("L1", torch.zeros(1, 1, 4, 4, 3), torch.zeros(1, 4, 4, 3))
