"""ctypes binding of libxpt_hip.so (C ABI: include/xpt_hip.h).

The library is built in-tree by ``__graft_entry__.build()`` / ``csrc/build.py`` with
``hipcc --offload-arch=gfx950``.  There is deliberately NO fallback: if the shared object is
missing or does not export a declared symbol, importing the ops raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# (XPT_HIP_LIB: another build of the same library, e.g. the previous commit's, for same-box A/B measurements; lab use only)
LIB_PATH = os.environ.get("XPT_HIP_LIB") or os.path.join(os.path.dirname(_HERE), "libxpt_hip.so")
LIB_PATH_F16 = os.environ.get("XPT_HIP_LIB_F16") or os.path.join(os.path.dirname(_HERE), "libxpt_hip_f16.so")

# ---- the 16-bit activation format of this PROCESS: "bf16" (libxpt_hip.so) or "fp16" (libxpt_hip_f16.so, the same sources built
# with IEEE-half activations: BASELINE configs[4]).  Chosen before the library is first loaded -- XPT_HALF in the environment,
# set_half_format(), or opts.CONV_DTYPE = "fp16" at model construction --; one format per process (packed weights, shadow
# copies and captured graphs all hold 16-bit data of that format).
_half_format = os.environ.get("XPT_HALF", "bf16")


def set_half_format(fmt):
    global _half_format
    if fmt not in ("bf16", "fp16"):
        raise XptHipError(f"unknown 16-bit format {fmt!r} (bf16 | fp16)")
    if _lib is not None and fmt != _half_format:
        raise XptHipError(f"the {_half_format} build of the library is already loaded in this process: {fmt} needs its own process "
                          f"(XPT_HALF={fmt} in the environment, or set_half_format() before the first op)")
    _half_format = fmt


def half_format():
    return _half_format


_half_dtype = None


def half():
    """torch dtype of the 16-bit activations / packed weights of this process."""
    global _half_dtype
    if _half_dtype is None or _half_dtype[0] != _half_format:
        import torch
        _half_dtype = (_half_format, torch.float16 if _half_format == "fp16" else torch.bfloat16)
    return _half_dtype[1]

XPT_PHOTO_L1, XPT_PHOTO_L2, XPT_PHOTO_SSIM = 0, 1, 2
PHOTO_METHODS = {"L1": XPT_PHOTO_L1, "L2": XPT_PHOTO_L2, "SSIM": XPT_PHOTO_SSIM}

_ERRORS = {-1: "XPT_ERR_NULL (required pointer is NULL)", -2: "XPT_ERR_SHAPE (bad dimension)",
           -3: "XPT_ERR_ARG (bad enum/flag)", -4: "XPT_ERR_WORKSPACE (workspace too small)",
           -5: "XPT_ERR_LAUNCH (hipGetLastError != hipSuccess)"}

_p = ctypes.c_void_p
_i = ctypes.c_int
_f = ctypes.c_float
_z = ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/xpt_hip.h declares
SIGNATURES = {
    "xpt_abi_version": (_i, []),
    "xpt_build_arch": (ctypes.c_char_p, []),
    "xpt_pose_rvec2matr_fwd": (_i, [_p, _p, _i, _p]),
    "xpt_pose_rvec2matr_bwd": (_i, [_p, _p, _p, _i, _p]),
    "xpt_resize_down_fwd": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "xpt_image_pyramids": (_i, [_p, _i, _i, _i, _i, _i, _p, _p, _p, _p, _p]),
    "xpt_warp_fwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _f, _p]),
    "xpt_warp_bwd_workspace_floats": (_z, [_i, _i, _i, _i]),
    "xpt_warp_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _z, _i, _i, _i, _i, _f, _p]),
    "xpt_bilinear_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "xpt_bilinear_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "xpt_photo_workspace_floats": (_z, [_i, _i, _i, _i]),
    "xpt_photo_fwd": (_i, [_i, _p, _p, _p, _p, _p, _z, _i, _i, _i, _i, _p]),
    "xpt_photo_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _z, _i, _i, _i, _i, _p]),
    "xpt_photo_fused_tune": (_i, [_i, _i, _i]),
    "xpt_photo_fused_variant": (_i, [_i]),
    "xpt_photo_fused_workspace_floats": (_z, [_i, _i, _i, _i]),
    "xpt_photo_fused_fwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _i, _i, _i, _i, _f, _p]),
    "xpt_photo_fused_bwd": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _i, _i, _i, _i, _f, _p]),
    "xpt_smooth_workspace_floats": (_z, [_i, _i, _i]),
    "xpt_smooth_fwd": (_i, [_p, _p, _p, _p, _z, _i, _i, _i, _f, _i, _p]),
    "xpt_smooth_bwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _i, _p]),
    "xpt_merge_total_fwd": (_i, [_i, _p, _p, _p, _p, _p, _i, _i, _p]),
    "xpt_merge_total_bwd": (_i, [_i, _p, _p, _p, _i, _p]),
    "xpt_smooth_ms_fwd": (_i, [_i, _p, _p, _p, _p, _z, _i, _p, _p, _f, _i, _p]),
    "xpt_smooth_ms_bwd": (_i, [_i, _p, _p, _p, _p, _i, _p, _p, _f, _i, _p]),
    "xpt_adam_step": (_i, [_p, _p, _p, _p, ctypes.c_longlong, _p, _f, _f, _f, _f, _f, _i, _p, _p]),
    "xpt_sgd_step": (_i, [_p, _p, ctypes.c_longlong, _f, _f, _i, _p, _p]),
    "xpt_dwconv_fwd": (_i, [_p, _p, _p] + [_i] * 12 + [_p]),
    "xpt_dwconv_bwd_data": (_i, [_p, _p, _p, _p] + [_i] * 12 + [_p]),
    "xpt_dwconv_tune": (_i, [_i]),
    "xpt_dwconv_bwd_weight_workspace_floats": (_z, [_i] * 5),
    "xpt_dwconv_bwd_weight": (_i, [_p, _p, _p, _p, _z] + [_i] * 12 + [_p]),
    "xpt_conv1x1_bwd_weight_workspace_floats": (_z, [ctypes.c_longlong, _i, _i]),
    "xpt_conv1x1_bwd_weight_tune": (_i, [_i, _i, _i, _i]),
    "xpt_conv1x1_bwd_weight_counters": (_i, [ctypes.c_longlong, _i, _i]),
    "xpt_conv1x1_bwd_weight": (_i, [_p, _p, _p, _p, _z, _p, _i, ctypes.c_longlong, _i, _i, ctypes.c_longlong,
                                    ctypes.c_longlong, _p]),
    "xpt_reduce_job_bytes": (_i, []),
    "xpt_reduce_partials": (_i, [_p, _p, _i, _p]),
    "xpt_half_format": (_i, []),
    "xpt_affine_act_bwd_blocks": (_i, [ctypes.c_longlong, _i]),
    "xpt_affine_act_bwd_partials": (_i, [_p, _p, _p, ctypes.c_longlong, _p, _p, _p, _p, _f, _p, _p, _z, ctypes.c_longlong,
                                         _i, _f, _i, _i, _p]),
    "xpt_dwconv_bwd_weight_chunks": (_i, [_i] * 6),
    "xpt_set_xcd_affinity": (_i, [_i]),
    "xpt_dwconv_bwd_weight_partials": (_i, [_p, _p, _p, _z] + [_i] * 12 + [_p]),
    "xpt_dwconv_bwd_both": (_i, [_p, _p, _p, _p, _p, _z] + [_i] * 12 + [_p]),
    "xpt_dwconv_multi_fwd": (_i, [_p, _p, _p, _p, _p, _p] + [_i] * 10 + [_p]),
    "xpt_dwconv_multi_bwd": (_i, [_p, _p, _i, _p, _p, _p, _p, _p, _p, _p] + [_i] * 10 + [_p]),
    "xpt_conv1x1_bwd_weight_defer_cap": (_i, [_i]),
    "xpt_conv1x1_bwd_weight_splits": (_i, [ctypes.c_longlong, _i, _i]),
    "xpt_conv1x1_bn_bwd_partials": (_i, [_p, _p, _p, _p, _p, _p, _f, _p, _p, _z, _p, _z, ctypes.c_longlong, _i, _i,
                                         ctypes.c_longlong, ctypes.c_longlong, _p]),
    "xpt_conv1x1_bn_bwd_partials_sum": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _z, _p, _z, ctypes.c_longlong, _i, _i,
                                             ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, _p]),
    "xpt_conv1x1_bwd_fused": (_i, [_p, _p, _p, _p, _p, _z, ctypes.c_longlong, _i, _i, ctypes.c_longlong, ctypes.c_longlong, _p]),
    "xpt_conv1x1_bn_bwd_fused": (_i, [_p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _z, _p, _z, ctypes.c_longlong, _i, _i,
                                      ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, ctypes.c_longlong, _p]),
    "xpt_conv1x1_bn_multi_bwd_fused": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _z, _z, ctypes.c_longlong,
                                            _i, _i, ctypes.c_longlong, _p]),
    "xpt_conv1x1_bn_multi_bwd_fused_fan": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _z, _z,
                                                ctypes.c_longlong, _i, _i, ctypes.c_longlong, _p]),
    "xpt_conv1x1_bn_multi_bwd_partials": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _z, _z, ctypes.c_longlong, _i, _i,
                                               ctypes.c_longlong, _p]),
    "xpt_conv1x1_bwd_weight_partials": (_i, [_p, _p, _p, _z, ctypes.c_longlong, _i, _i, ctypes.c_longlong,
                                             ctypes.c_longlong, _p]),
    "xpt_pwconv_bn_fwd": (_i, [_p, _p, _p, _p, _p, _p, _f, _p, _p, _p, ctypes.c_longlong, _i, _i, ctypes.c_longlong, _p]),
    "xpt_multi_copy": (_i, [_p, _p, _p, _i, _p]),
    "xpt_concat_channels": (_i, [_p, _p, _p, _i, _p, ctypes.c_longlong, _i, _p]),
    "xpt_avgpool3_same": (_i, [_p, ctypes.c_longlong, _p, _i, _i, _i, _i, _f, _i, _i, _p]),
    "xpt_pwconv_bn_multi_fwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, ctypes.c_longlong, _i, _i, ctypes.c_longlong, _p]),
    "xpt_pwconv_bn_multi_fwd_sib": (_i, [_i, _p, _p, _p, _p, _p, _p, _f, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p,
                                         ctypes.c_longlong, _i, _i, ctypes.c_longlong, ctypes.c_longlong, _p]),
    "xpt_corr_cost_channels": (_i, [_i, _i]),
    "xpt_corr_cost_fwd": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "xpt_corr_cost_bwd": (_i, [_p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "xpt_headconv_tune": (_i, [_i, _i]),
    "xpt_pwconv_tune": (_i, [_i, _i]),
    "xpt_depth_head_fwd": (_i, [_p, _p, _p, ctypes.c_longlong, _p]),
    "xpt_depth_head_bwd": (_i, [_p, _p, _p, _p, ctypes.c_longlong, _p]),
    "xpt_global_avgpool_fwd": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "xpt_global_avgpool_bwd": (_i, [_p, _p, _i, _i, _i, _i, _p]),
    "xpt_upsample2x_fwd": (_i, [_p, _p, ctypes.c_longlong, _i, _i, _i, _p]),
    "xpt_upsample2x_bwd": (_i, [_p, ctypes.c_longlong, _p, ctypes.c_longlong, _i, _i, _i, _p]),
    "xpt_upsample2x_bwd_add": (_i, [_p, ctypes.c_longlong, _p, _p, ctypes.c_longlong, _i, _i, _i, _p]),
    "xpt_depth_head_ms_fwd": (_i, [_i, _p, _p, _p, _p, _p]),
    "xpt_depth_head_ms_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p]),
    "xpt_sum_rows": (_i, [_p, _p, _i, _p, ctypes.c_longlong, _i, _i, _p]),
    "xpt_photo_fused_ms_fwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _z, _i, _i, _p, _p, _p, _p]),   # losses is one [2 n, B] buffer
    "xpt_photo_fused_ms_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _i, _i, _p, _p, _p, _p]),
    "xpt_graph_node_census": (_i, [_p, _p]),
    "xpt_sepconv_bn_multi_fwd": (_i, [_i] + [_p] * 23 + [_f, _i, _i, _i, _i, _i, _p]),
    "xpt_photo_march_tune": (_i, [_i, _i, _i]),
    "xpt_photo_march_plan": (_i, [_i, _i, _i, _i, _i]),
    "xpt_photo_march_ms_fwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _z, _i, _i, _p, _p, _p, _p]),
    "xpt_photo_march_ms_bwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _i, _i, _p, _p, _p, _p]),
    "xpt_photo_march_ms_fwdbwd": (_i, [_i, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _p, _z, _i, _i, _p, _p, _p, _p]),
    "xpt_augment_pin": (_i, [_p]),
    "xpt_augment": (_i, [_p, _p, _p, _p, _p, _p, _i, _p, _p, _i, _p, _p, _p, _p, _i, _p, _p, _p, _p, _i, _p, _p, _i, _i,
                         _f, _f, _f, _f, _p]),
    "xpt_stem_input": (_i, [_p, ctypes.c_longlong, _p, _i, _i, _i, _p]),
    "xpt_pose_metric": (_i, [_p, _p, _p, _i, _i, _p]),
    "xpt_depth_metric": (_i, [_p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _f, _f, _p]),
    "xpt_adjust_gather": (_i, [_p, ctypes.c_longlong, _p, _i, _i, _i, _i, _i, _p]),
    "xpt_adjust_scatter": (_i, [_p, ctypes.c_longlong, _p, ctypes.c_longlong, _p, _i, _i, _i, _i, _i, _p]),
    "xpt_pool_pair_fwd": (_i, [_p, ctypes.c_longlong, _p, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "xpt_pool_pair_bwd": (_i, [_p, ctypes.c_longlong, _p, ctypes.c_longlong, _p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "xpt_pool_pair_bwd2": (_i, [_p, ctypes.c_longlong, _p, ctypes.c_longlong, _p, ctypes.c_longlong, _p, _p, _i, _i, _i, _i, _i, _i,
                                _i, _i, _i, _p]),
    "xpt_cell_tail_fwd": (_i, [_i, _p, _p, _p, _p, _p, _p, ctypes.c_longlong, _i, _i, _i, _i, _i, _p]),
    "xpt_cell_tail_bwd": (_i, [_i, _p, _p, _p, ctypes.c_longlong, _p, _i, _i, _p, _p, _p, _p, _p, _i, _i, _i, _i, _i, _p]),
    "xpt_crc32c": (ctypes.c_uint32, [_p, _z]),
    "xpt_tfrecord_index": (ctypes.c_longlong, [_p, _z, _i, _p, _p, _p, ctypes.c_longlong]),
    "xpt_tfrecord_decode": (_i, [_p, _z, ctypes.c_uint32, _i, _i, _p, _p, _p]),
    "xpt_conv_pack_job_bytes": (_i, []),
    "xpt_conv_pack_weights": (_i, [_p, _i, ctypes.c_longlong, _p]),
    "xpt_conv2d_tune": (_i, [_i]),
    "xpt_conv2d_fwd": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, ctypes.c_longlong, _i, _i, _i, _i, _i, _i, _i, _i,
                            ctypes.c_longlong, _i, _f, _p]),
    "xpt_conv2d_bwd_data": (_i, [_p, _p, _p, _i, _i, _i, _i, ctypes.c_longlong, _i, _i, _i, _i, _i, _i, _i, _i,
                                 ctypes.c_longlong, _i, _p]),
    "xpt_conv2d_splitk_tune": (_i, [_i, _i, _i, _i]),
    "xpt_conv2d_splitk_workspace_floats": (_z, [ctypes.c_longlong, _i, _i, _i, _i]),
    "xpt_conv2d_fwd_splitk": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, ctypes.c_longlong, _i, _i, _i, _i, _i, _i, _i,
                                   ctypes.c_longlong, _i, _f, _p, _z, _p]),
    "xpt_conv2d_bwd_data_splitk": (_i, [_p, _p, _p, _i, _i, _i, _i, ctypes.c_longlong, _i, _i, _i, _i, _i, _i, _i,
                                        ctypes.c_longlong, _i, _p, _z, _p]),
    "xpt_conv2d_stream_tune": (_i, [_i, _i, _i, _i]),
    "xpt_conv2d_stream_serves": (_i, [_i] * 9),
    "xpt_conv2d_fwd_stream": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, ctypes.c_longlong, _i, _i, _i, _i, _i, ctypes.c_longlong,
                                   _i, _f, _p]),
    "xpt_conv2d_fwd_stream_k5s2": (_i, [_p, _p, _p, _p, _i, _i, _i, _i, ctypes.c_longlong, _i, _i, _i, _i, _i, ctypes.c_longlong,
                                        _f, _p]),
    "xpt_conv2d_bwd_data_stream": (_i, [_p, _p, _p, _i, _i, _i, _i, ctypes.c_longlong, _i, _i, _i, _i, _i, ctypes.c_longlong,
                                        _i, _p]),
    "xpt_conv2d_bwd_weight_tune": (_i, [_i, _i]),
    "xpt_conv2d_bwd_weight_splits": (_i, [_i] * 8),
    "xpt_conv2d_bwd_weight_partials": (_i, [_p, _p, _p, _z, _i, _i, _i, _i, _i, ctypes.c_longlong, _i, ctypes.c_longlong,
                                            _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "xpt_headconv_bwd_blocks": (_i, [_i, _i, _i, _i]),
    "xpt_headconv_fwd": (_i, [_p, ctypes.c_longlong, _p, _p, _p, _i, _i, _i, _i, _p]),
    "xpt_headconv_bwd": (_i, [_p, ctypes.c_longlong, _p, _p, _p, _p, _z, _i, _i, _i, _i, _p]),
    "xpt_headconv_bwd_add": (_i, [_p, ctypes.c_longlong, _p, _p, _p, ctypes.c_longlong, _p, _p, _z, _i, _i, _i, _i, _p]),
    "xpt_restack_bf16": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "xpt_affine_act_fwd": (_i, [_p, _p, _p, _p, _p, _f, _p, _p, ctypes.c_longlong, _i, _f, _i, _i, _p]),
    "xpt_affine_act_bwd_workspace_floats": (_z, [ctypes.c_longlong, _i]),
    "xpt_affine_act_bwd": (_i, [_p, _p, _p, ctypes.c_longlong, _p, _p, _p, _p, _f, _p, _p, _p, _p, _z, ctypes.c_longlong,
                                _i, _f, _i, _i, _p]),
}



class ConvPackJob(ctypes.Structure):
    """Mirror of xpt_conv_pack_job (include/xpt_hip.h)."""
    _fields_ = [("src", ctypes.c_void_p), ("fwd", ctypes.c_void_p), ("bwd", ctypes.c_void_p),
                ("sn", ctypes.c_longlong), ("sc", ctypes.c_longlong), ("sh", ctypes.c_longlong), ("sw", ctypes.c_longlong),
                ("N", ctypes.c_int), ("T", ctypes.c_int), ("KW", ctypes.c_int), ("C", ctypes.c_int), ("Cp", ctypes.c_int),
                ("Np", ctypes.c_int), ("first_block", ctypes.c_longlong)]


class ReduceJob(ctypes.Structure):
    """Mirror of xpt_reduce_job (include/xpt_hip.h)."""
    _fields_ = [("dst", ctypes.c_void_p), ("n", ctypes.c_longlong), ("nseg", ctypes.c_int),
                ("split_waves", ctypes.c_int), ("src", ctypes.c_void_p * 4), ("stride", ctypes.c_longlong * 4),
                ("nsplit", ctypes.c_int * 4)]


_lib = None


class XptHipError(RuntimeError):
    pass


def load():
    """dlopen libxpt_hip.so and bind every declared entry point (raises if anything is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    path = LIB_PATH_F16 if _half_format == "fp16" else LIB_PATH
    if not os.path.isfile(path):
        raise XptHipError(f"HIP extension not built: {path} is missing. Run `python -c 'import __graft_entry__ as g; "
                          f"g.build()'` (hipcc --offload-arch=gfx950). There is no CPU fallback for these ops.")
    # PyTorch-ROCm bundles its own libamdhip64: it must be the HIP runtime already resident when our
    # library is dlopen'ed, otherwise two runtimes coexist and launches on torch's streams fail.
    import torch  # noqa: F401
    lib = ctypes.CDLL(path)
    for name, (restype, argtypes) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise XptHipError(f"{path} does not export {name}; rebuild the extension") from e
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.xpt_half_format() != (1 if _half_format == "fp16" else 0):
        raise XptHipError(f"{path} was not built for {_half_format} activations; rebuild the extension (csrc/build.py)")
    variant = os.environ.get("XPT_FUSED_FWD_PIPE")
    if variant is not None:                      # A/B switch of the fused forward's row loop (default: the library's)
        lib.xpt_photo_fused_variant(int(variant))
    _lib = lib
    return lib


def check(code, what):
    if code != 0:
        raise XptHipError(f"{what} failed: {_ERRORS.get(code, code)}")
