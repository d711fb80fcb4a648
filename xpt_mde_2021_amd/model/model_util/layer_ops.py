"""Convolution factory with the reference's defaults (model/model_util/layer_ops.py:5-50):
Conv2D(padding="same") with TF SAME semantics, LeakyReLU(0.1), TruncatedNormal(0.025) kernels, zero bias.

Tensors are NCHW-indexed torch tensors (stored channels_last for MIOpen when opts.CHANNELS_LAST); the
convolutions themselves run in MIOpen / rocBLAS (MFMA) through torch.nn.functional.conv2d.
"""
import math

import torch
import torch.nn as nn
from ...hip.lib import half as _half      # torch dtype of the 16-bit activations (bf16 | fp16 build of the library)

import torch.nn.functional as F

from ...hip import conv as _conv
from ...hip import ops as _ops


def same_pad(n, k, s, d=1):
    """TF "SAME": total = max((ceil(n/s) - 1) * s + k_eff - n, 0), k_eff = (k - 1) d + 1; before = total // 2, after = the rest."""
    total = max((math.ceil(n / s) - 1) * s + (k - 1) * d + 1 - n, 0)
    return total // 2, total - total // 2


def make_activation(name, param=None):
    if name in (None, "linear"):
        return nn.Identity()
    if name == "leaky_relu":
        return nn.LeakyReLU(0.1 if param is None else param)
    if name == "relu":
        return nn.ReLU()
    raise ValueError(f"unknown activation {name}")


# Weight gradients of convolutions whose output map has at most this many pixels (OH x OW) are evaluated as ONE fp32
# GEMM on the unfolded input instead of a library weight-gradient solver (see _ConvFp32WeightGrad): PWC-Net's 2x6 and
# 4x12 levels, PoseNet's 4x13 / 2x7 maps.  (Larger maps pass tests/test_graph_replay.py with the library solvers, and
# unfolding them costs tens of MB per layer.)
SMALL_MAP_AREA = int(__import__("os").environ.get("XPT_DEBUG_SMALL_MAP_AREA", "64"))     # A/B: 0 = library solvers everywhere


# Convolutions that went through the library (MIOpen) on the GPU since process start.  A training step that contains any is
# not captured into a hipGraph (train_val._StepGraph): MIOpen solvers that zero a workspace with a memset node return garbage
# from the second replay on with this runtime; the bf16 rigid step has none.
LIBRARY_CONV_CALLS = [0]


def note_library_conv(tensor):
    if tensor.is_cuda:
        LIBRARY_CONV_CALLS[0] += 1


class _ConvFp32WeightGrad(torch.autograd.Function):
    """Library convolution (MIOpen) whose WEIGHT gradient is always evaluated in fp32.

    Forward and data gradient run in the activation dtype (bf16 under autocast).  MIOpen's bf16 weight-gradient
    solvers accumulate in an fp32 workspace and cast; replayed from a hipGraph on ROCm 7.0 / torch 2.10 that path
    returns garbage from the second replay on (tests/test_graph_replay.py), and its deterministic replacements are
    ~25x slower naive kernels.  The fp32 weight-gradient solvers are correct under replay for the larger maps, so x
    and dy are up-cast for that one call (the gradient is wanted in fp32 for the flat Adam buffers anyway).  For SMALL
    maps (PoseNet's and PWC-Net's 2x6 ... 8x24 levels) the fp32 solvers show the same defect (found with the 196 -> 196
    3x3 convolution of PWC-Net's level 6 on [32, 196, 2, 6]): there dW = dy^T . unfold(x) is one rocBLAS GEMM."""

    @staticmethod
    def forward(ctx, x, weight, stride, padding, compute_dtype, dilation=1, safe_dgrad=False):
        w = _low_precision_weight(weight, compute_dtype)
        xc = x.to(compute_dtype)
        with torch.autocast(device_type=x.device.type, enabled=False):
            note_library_conv(xc)
            y = F.conv2d(xc, w, None, stride, padding, dilation)
        ctx.save_for_backward(xc, weight)
        ctx.cfg = (stride, padding, compute_dtype, dilation, safe_dgrad)
        return y

    @staticmethod
    def backward(ctx, dy):
        xc, weight = ctx.saved_tensors
        stride, padding, compute_dtype, dilation, safe_dgrad = ctx.cfg
        s2, p2, d2 = [stride, stride], list(padding), [dilation, dilation]
        dx = dw = None
        small = dy.shape[2] * dy.shape[3] <= SMALL_MAP_AREA
        with torch.autocast(device_type=dy.device.type, enabled=False):
            # small maps: ALWAYS the forward-convolution form (the library's data-gradient solvers for these shapes came
            # back with garbage from the second replay of a captured step on, intermittently -- which solver MIOpen's
            # find picks varies from run to run; first seen on PWC-Net, then on PoseNet's 2x7 level in fp32 mode)
            if ctx.needs_input_grad[0] and small:
                dx = flipped_conv_data_grad(dy.to(compute_dtype), _low_precision_weight(weight, compute_dtype),
                                            xc.shape, stride, padding, dilation)
            elif ctx.needs_input_grad[0]:
                dx = torch.ops.aten.convolution_backward(dy.to(compute_dtype), xc,
                                                         _low_precision_weight(weight, compute_dtype), None, s2, p2,
                                                         d2, False, [0, 0], 1, [True, False, False])[0]
            if ctx.needs_input_grad[1]:
                if small or getattr(weight, "xpt_safe_wgrad", False):     # the latter: set by the trainer's replay check
                    dw = unfolded_weight_grad(dy, xc, weight.shape, stride, padding, dilation)
                else:
                    # (the two up-casts stay separate launches: one multi-tensor _foreach_copy_ measured 0.16 ms/step slower)
                    dw = torch.ops.aten.convolution_backward(dy.float(), xc.float(), weight.float(), None, s2, p2, d2,
                                                             False, [0, 0], 1, [False, True, False])[1]
        return dx, dw, None, None, None, None, None


def flipped_conv_data_grad(dy, weight, x_shape, stride, padding, dilation):
    """Data gradient of conv2d as a FORWARD convolution: dx = conv2d(zero-stuffed dy, flip(W)^T, padding d (k - 1) - p,
    dilation d).  Used for small maps of the flow net, where the library's bf16 data-gradient solvers (like its
    weight-gradient solvers) return garbage from the second hipGraph replay on: first seen on the 96 -> 96 3x3
    convolutions of PWC-Net's 4x8 level (tools/replay_grad_diff.py); forward solvers replay correctly."""
    cout, cin, kh, kw = weight.shape
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    B, _, OH, OW = dy.shape
    H, W = x_shape[2], x_shape[3]
    if stride > 1:
        up = dy.new_zeros((B, cout, (OH - 1) * stride + 1, (OW - 1) * stride + 1))
        up[:, :, ::stride, ::stride] = dy
        dy = up
    wt = weight.flip(2, 3).permute(1, 0, 2, 3).contiguous(memory_format=torch.channels_last)
    fh, fw = dilation * (kh - 1) - ph, dilation * (kw - 1) - pw
    # rows / columns of x beyond the last window (stride > 1, or explicit extra padding) receive no gradient
    eh = H - (dy.shape[2] + 2 * fh - dilation * (kh - 1))
    ew = W - (dy.shape[3] + 2 * fw - dilation * (kw - 1))
    if fh < 0 or fw < 0 or eh or ew:
        dy = F.pad(dy, (max(fw, 0), max(fw, 0) + ew, max(fh, 0), max(fh, 0) + eh))
        if fh < 0 or fw < 0:                    # padding larger than the dilated kernel reach: crop instead of pad
            dy = dy[:, :, -min(fh, 0):dy.shape[2] + min(fh, 0), -min(fw, 0):dy.shape[3] + min(fw, 0)]
        fh = fw = 0
    return F.conv2d(dy.contiguous(memory_format=torch.channels_last), wt, None, 1, (fh, fw), dilation)


def unfolded_weight_grad(dy, x, weight_shape, stride, padding, dilation):
    """dW[co, ci, kh, kw] = sum_p dy[p, co] * patches(x)[p, (ci, kh, kw)] in fp32: the patches are strided VIEWS of the
    padded input (Tensor.unfold), gathered by one copy launch, then one GEMM -- F.unfold would launch one im2col kernel
    per batch element."""
    cout, cin, kh, kw = weight_shape
    ph, pw = (padding, padding) if isinstance(padding, int) else padding
    xp = F.pad(x.float(), (pw, pw, ph, ph)) if (ph or pw) else x.float()
    B, _, OH, OW = dy.shape
    win = xp.unfold(2, (kh - 1) * dilation + 1, stride).unfold(3, (kw - 1) * dilation + 1, stride)
    win = win[:, :, :OH, :OW, ::dilation, ::dilation]                            # [B, cin, OH, OW, kh, kw]
    cols = win.permute(0, 2, 3, 1, 4, 5).reshape(B * OH * OW, cin * kh * kw)     # the one copy
    dyr = dy.float().permute(0, 2, 3, 1).reshape(B * OH * OW, cout)              # a view for channels_last dy
    return torch.mm(dyr.t(), cols).view(cout, cin, kh, kw)


def _low_precision_weight(weight, dtype):
    """bf16 view of the weight: the shadow copy the fused Adam kernel keeps current (no cast launch) when there is one."""
    shadow = getattr(weight, "shadow_bf16", None)
    if shadow is not None and dtype == _half():
        return shadow
    return weight.to(dtype)


def conv2d_library(x, weight, stride, padding, dilation=1, safe_dgrad=False):
    """Dense (groups = 1) convolution without bias through MIOpen; see _ConvFp32WeightGrad for the backward."""
    if x.is_cuda:
        dtype = torch.get_autocast_dtype("cuda") if torch.is_autocast_enabled() else x.dtype
        return _ConvFp32WeightGrad.apply(x, weight, int(stride), (int(padding[0]), int(padding[1])), dtype, int(dilation),
                                         bool(safe_dgrad))
    return F.conv2d(x, weight, None, stride, padding, dilation)


class Conv2DSame(nn.Module):
    """keras.layers.Conv2D(filters, k, strides, padding="same", activation=...) on NCHW tensors."""

    def __init__(self, in_channels, filters, kernel_size=3, strides=1, activation="leaky_relu", activation_param=0.1,
                 kernel_initializer="truncated_normal", kernel_initializer_param=0.025, use_bias=True, groups=1,
                 dilation_rate=1):
        super().__init__()
        self.k, self.s, self.d = int(kernel_size), int(strides), int(dilation_rate)
        self.replay_safe_dgrad = False         # set by nets whose small-map data gradients need the forward-conv path
        self.conv = nn.Conv2d(in_channels, filters, self.k, self.s, padding=0, dilation=self.d, bias=use_bias,
                              groups=groups)
        self.act = make_activation(activation, activation_param)
        # negative-side slope of the activation for the fused epilogue (1 = linear, 0 = ReLU)
        self.slope = {None: 1.0, "linear": 1.0, "relu": 0.0,
                      "leaky_relu": 0.1 if activation_param is None else float(activation_param)}.get(activation)
        if kernel_initializer == "truncated_normal":
            std = kernel_initializer_param
            nn.init.trunc_normal_(self.conv.weight, mean=0.0, std=std, a=-2 * std, b=2 * std)
        elif kernel_initializer == "he_normal":
            nn.init.kaiming_normal_(self.conv.weight, mode="fan_in", nonlinearity="relu")
        else:
            nn.init.xavier_uniform_(self.conv.weight)
        if use_bias:
            nn.init.zeros_(self.conv.bias)

    def forward(self, x, upsample=False):
        """`upsample`: the input is consumed through UpSampling2D(2, "nearest") (depth_net.py:76-84); on the matrix-core
        path the up-sampled tensor is never materialised."""
        if _conv.usable(x, self.conv, self.slope):
            # gfx950 implicit-GEMM kernels (hip/conv.py): TF-SAME padding, bias, activation (and the up-sampling) fused;
            # no library call, so nothing in a captured training step depends on MIOpen's solvers or workspaces
            cp = _conv.round_up(self.conv.in_channels, 8)
            if x.dtype != _half():
                x = x.to(_half())
            if x.shape[1] < cp:
                x = F.pad(x, (0, 0, 0, 0, 0, cp - x.shape[1]))
            return _conv.conv2d_same(x, self.conv.weight, self.conv.bias, self.s, self.slope, upsample)
        if upsample:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
        ph = same_pad(x.shape[2], self.k, self.s, self.d)
        pw = same_pad(x.shape[3], self.k, self.s, self.d)
        # on the GPU the bias add + activation (and the bias gradient) run in one gfx950 epilogue kernel
        fused = x.is_cuda and self.conv.bias is not None and self.slope is not None
        bias = None if fused else self.conv.bias
        if ph[0] != ph[1] or pw[0] != pw[1]:   # stride 2 on an even extent: TF pads (k-2)//2 before, (k-1)//2 after
            x = F.pad(x, (pw[0], pw[1], ph[0], ph[1]))
            ph, pw = (0, 0), (0, 0)
        if fused and self.conv.groups == 1:
            y = conv2d_library(x, self.conv.weight, self.s, (ph[0], pw[0]), self.d, self.replay_safe_dgrad)
        else:
            note_library_conv(x)
            y = F.conv2d(x, self.conv.weight, bias, self.s, (ph[0], pw[0]), self.d, self.conv.groups)
        if fused:
            return _ops.bias_act(y, self.conv.bias, self.slope)
        return self.act(y)


class CustomConv2D:
    """layer_ops.py:5-36: a factory holding default Conv2D arguments; `conv(in_channels, filters, ...)` builds a
    layer (torch needs the input channel count that Keras infers lazily)."""

    def __init__(self, kernel_size=3, strides=1, activation="relu", activation_param=None,
                 kernel_initializer="glorot_uniform", kernel_initializer_param=None):
        self.kernel_size = kernel_size
        self.strides = strides
        self.activation = activation
        self.activation_param = activation_param
        self.kernel_initializer = kernel_initializer
        self.kernel_initializer_param = kernel_initializer_param

    def __call__(self, in_channels, filters, kernel_size=None, strides=None, activation=None, name="", dilation_rate=1):
        return Conv2DSame(in_channels, filters,
                          self.kernel_size if kernel_size is None else kernel_size,
                          self.strides if strides is None else strides,
                          self.activation if activation is None else activation, self.activation_param,
                          self.kernel_initializer, self.kernel_initializer_param, dilation_rate=dilation_rate)


def resize_image(src, dst_height, dst_width):
    """layer_ops.py:43-50: TF2 bilinear resize (half-pixel centres) of NCHW `src`; identity if sizes match."""
    if src.shape[2] == dst_height and src.shape[3] == dst_width:
        return src
    return F.interpolate(src, size=(dst_height, dst_width), mode="bilinear", align_corners=False, antialias=False)


def resize_like(src, ref):
    """layer_ops.py:39-41."""
    return resize_image(src, ref.shape[2], ref.shape[3])
