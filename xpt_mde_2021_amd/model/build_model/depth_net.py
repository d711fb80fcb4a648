"""DepthNetPretrained: multi-scale encoder + the reference's U-Net style decoder
(model/build_model/depth_net.py:76-92 upsample / get_scaled_depth, :101-109 NoResize skip block, :137-167 decode).
"""
import torch
import torch.nn as nn
from ...hip.lib import half as _half      # torch dtype of the 16-bit activations (bf16 | fp16 build of the library)

import torch.nn.functional as F

from ...hip import conv as _conv
from ...hip import ops as _ops
from ..model_util import layer_ops as lo
from ...utils.util_class import WrongInputException
from .pretrained_nets import PretrainedModel


_BATCHED_HEADS = __import__("os").environ.get("XPT_DEBUG_PER_SCALE_LOSS", "0") != "1"     # A/B: one activation launch per scale


class _ChannelsLastOne(torch.autograd.Function):
    """[B,1,H,W] contiguous -> the same memory with channels_last strides (a view); the gradient passes through as it
    comes (as_strided's own backward would materialise it with a zero fill and a copy)."""

    @staticmethod
    def forward(ctx, x):
        return x.as_strided(x.shape, (x.shape[2] * x.shape[3], 1, x.shape[3], 1))

    @staticmethod
    def backward(ctx, g):
        return g


_FUSED_CONCAT = __import__("os").environ.get("XPT_DEBUG_TORCH_CAT", "0") != "1"     # A/B: torch.cat + cached zero pad
_FUSED_FAN_IN = __import__("os").environ.get("XPT_DEBUG_ATEN_FAN_IN", "0") != "1"    # A/B: autograd's add launches for the decoder's fan-ins


class UpconvWithSkip(nn.Module):
    """upconv_with_skip_connection of DepthNetNoResize (depth_net.py:101-109):
    nearest 2x -> conv3x3 -> concat([., skip][, up-sampled previous prediction]) -> conv3x3."""

    def __init__(self, conv2d, cin, skip_channels, out_channels, upsample_interp="nearest"):
        super().__init__()
        self.interp = upsample_interp
        self.conv1 = conv2d(cin, out_channels, 3)
        self.conv2 = conv2d(out_channels + skip_channels, out_channels, 3)

    def forward(self, bef_layer, skips):
        if self.interp == "nearest":
            up = self.conv1(bef_layer, upsample=True)                             # UpSampling2D(2, "nearest") fused into the conv
        else:
            up = self.conv1(F.interpolate(bef_layer, scale_factor=2, mode="bilinear", align_corners=False))
        parts = [up] + [s.to(up.dtype) for s in skips]
        if up.is_cuda and up.dtype == _half() and len(parts) <= 4 and _FUSED_CONCAT:
            # the concatenation and its zero pad channels in one launch (torch.cat's batched copy ran at 0.5 - 0.8 TB/s here)
            return self.conv2(_ops.concat_channels(parts))
        if up.is_cuda:
            # a one-channel map is "contiguous" in both layouts; give it channels_last strides explicitly, or torch.cat
            # sees mixed layouts, answers in NCHW and the convolution below pays a full re-layout copy of the concatenation
            parts = [_ChannelsLastOne.apply(p) if p.shape[1] == 1 and p.is_contiguous() else p for p in parts]
        total = sum(p.shape[1] for p in parts)
        if up.is_cuda and up.dtype == _half() and total % 8:
            # the matrix-core convolution reads 8-channel groups: the concatenation is built with its zero pad channels
            parts.append(self._zeros(up, -total % 8))
        return self.conv2(torch.cat(parts, dim=1))

    def _zeros(self, like, channels):
        key = (like.shape[0], channels, like.shape[2], like.shape[3], like.device)
        cache = self.__dict__.setdefault("_zero_cache", {})
        if key not in cache:                       # allocated by the first eager step, reused (also by captured steps)
            cache[key] = torch.zeros((like.shape[0], channels, like.shape[2], like.shape[3]), dtype=like.dtype,
                                     device=like.device).contiguous(memory_format=torch.channels_last)
        return cache[key]


class ScaledDepthHead(nn.Module):
    """get_scaled_depth (depth_net.py:87-92): 3x3 conv to one LINEAR channel, the depth activation on it, and the
    bilinearly up-sampled raw prediction that feeds the next decoder level.  Runs in fp32 (16..128 -> 1 channels:
    negligible cost) so that the depth handed to the warp kernels is not quantised to bf16."""

    def __init__(self, conv2d, cin, pred_depth):
        super().__init__()
        self.conv = conv2d(cin, 1, 3, activation="linear")
        self.predict_depth = pred_depth

    def forward(self, src, dst_height, dst_width, activate=True, split=False):
        """activate=False: depth is None -- the caller applies the activation to all scales at once
        (predict_depth.with_disparity_multi), the decoder itself only consuming the raw prediction.
        split=True (a feature map that ALSO feeds the next decoder level): a fourth result, `src` as an alias for that other
        consumer -- head and next level become one autograd node and their gradients are added inside the head's backward
        launch; likewise the raw prediction's two consumers (activation, up-sampling) below."""
        src_alias = src
        with torch.autocast(device_type=src.device.type, enabled=False):
            fused_fan_in = _FUSED_FAN_IN and split and torch.is_grad_enabled() and src.requires_grad
            if _conv.head_usable(src, self.conv.conv) and self.conv.slope == 1.0:
                if fused_fan_in:
                    conv, src_alias = _conv.head_conv_split(src, self.conv.conv.weight, self.conv.conv.bias)
                else:
                    conv = _conv.head_conv(src, self.conv.conv.weight, self.conv.conv.bias)     # bf16 features in, fp32 prediction out
            else:
                conv = self.conv(src.float())
            if not activate:
                depth, self.last_disp = None, None
            elif hasattr(self.predict_depth, "with_disparity"):
                depth, self.last_disp = self.predict_depth.with_disparity(conv)
            else:
                depth, self.last_disp = self.predict_depth(conv), None
            if (_BATCHED_HEADS and conv.is_cuda and conv.dtype == torch.float32 and conv.is_contiguous()
                    and src.dtype == _half() and (dst_height, dst_width) == (2 * conv.shape[2], 2 * conv.shape[3])):
                # the exact 2x resize and the cast to the decoder's dtype in one launch (one more for the backward)
                if _FUSED_FAN_IN and not activate and torch.is_grad_enabled() and conv.requires_grad:
                    conv, conv_up = _ops.upsample2x_split(conv, _half())
                else:
                    conv_up = _ops.upsample2x(conv, _half())
            else:
                conv_up = lo.resize_image(conv, dst_height, dst_width)
        if split:
            return depth, conv_up, conv, src_alias
        return depth, conv_up, conv


class DepthNetPretrained(nn.Module):
    """depth_net.py:112-167.  forward(image5d [B,S,H,W,3]) -> {"depth_ms": [d0 (1/1), d1 (1/2), d2 (1/4), d3 (1/8)]
    each [B,h,w,1] fp32, "debug_out": [p0, up0, p3, up3]}.  The target frame is the LAST frame (:131)."""

    def __init__(self, total_shape, global_batch, conv2d, pred_depth, upsample_iterp, net_name, use_pt_weight,
                 high_res):
        super().__init__()
        self.total_shape = tuple(total_shape)
        self.high_res = high_res
        self.encoder = PretrainedModel(net_name, use_pt_weight).encoder()
        self.cut_backward = False      # see forward()
        self.cuts = []                 # [(encoder outputs, their detached stand-ins)] of the forward calls of this step
        # taps inside a cell with structurally-zero filters arrive WITH them (no gather launch): the convolution that reads
        # such a skip has zero columns at those input channels -- zero by construction, like the encoder's (pretrained_nets)
        layout = self.encoder.tap_layout() if hasattr(self.encoder, "tap_layout") else [(c, None) for c in self.encoder.TAP_CHANNELS]
        self.physical_taps = any(sel is not None for _, sel in layout)
        (c1, _), (c2, sel2), (c3, _), (c4, _), (c5, _) = layout
        self.up4 = UpconvWithSkip(conv2d, c5, c4, 256, upsample_iterp)            # 1/16
        self.up3 = UpconvWithSkip(conv2d, 256, c3, 128, upsample_iterp)           # 1/8
        self.depth3 = ScaledDepthHead(conv2d, 128, pred_depth)
        self.up2 = UpconvWithSkip(conv2d, 128, c2 + 1, 64, upsample_iterp)        # 1/4
        self.depth2 = ScaledDepthHead(conv2d, 64, pred_depth)
        self.up1 = UpconvWithSkip(conv2d, 64, c1 + 1, 32, upsample_iterp)         # 1/2
        self.depth1 = ScaledDepthHead(conv2d, 32, pred_depth)
        self.up0 = UpconvWithSkip(conv2d, 32, 1, 16, upsample_iterp)              # 1/1: the only skip is p1 up-sampled
        # (hip/conv.py, XPT_WGRAD_DEFER: the decoder's weight gradients as one side-stream branch, issued when up4.conv1 --
        #  the decoder's first layer, last in the backward pass -- has sent its data gradient on to the encoder)
        for up in (self.up4, self.up3, self.up2, self.up1, self.up0):
            for layer in (up.conv1, up.conv2):
                layer.conv.weight.defer_wgrad = True
        self.up4.conv1.conv.weight.flush_wgrads = True
        self.depth0 = ScaledDepthHead(conv2d, 16, pred_depth)
        if any(sel is not None for (_, sel) in (layout[0], layout[2], layout[3], layout[4])):
            raise WrongInputException("structurally-zero channels are handled for the 1/4 tap only")
        self.skip2_zero = None
        if sel2 is not None:                                # input channels of up2.conv2: [64 upconv | c2 skip | 1 prediction]
            keep = torch.zeros(c2, dtype=torch.bool)
            keep[sel2] = True
            self.skip2_zero = 64 + torch.nonzero(~keep).flatten()
            self.apply_structural_zeros()

    def structural_pads(self):
        """[(tensor, out_sel, in_sel)] of the DECODER's own tensors with structurally-zero entries (the encoder lists its own):
        up2.conv2 reads the 1/4 tap with its zero channels, so its logical input channels are all but `skip2_zero`."""
        if self.skip2_zero is None:
            return []
        w = self.up2.conv2.conv.weight
        keep = torch.ones(w.shape[1], dtype=torch.bool)
        keep[self.skip2_zero] = False
        return [(w, None, torch.nonzero(keep).flatten())]

    def apply_structural_zeros(self):
        if self.skip2_zero is not None:
            with torch.no_grad():
                w = self.up2.conv2.conv.weight
                w[:, self.skip2_zero.to(w.device)] = 0

    def forward(self, image5d):
        target = image5d[:, -1].permute(0, 3, 1, 2)                               # [B,3,H,W] view of the NHWC frame
        height, width = target.shape[2:]
        taps = self.encoder(target, physical_taps=True) if self.physical_taps else self.encoder(target)
        if self.cut_backward and torch.is_grad_enabled() and all(t.requires_grad for t in taps):
            # the backward pass is cut between decoder and encoder (train_val.ModelTrainerDistrib: the gradients of
            # everything behind the encoder are all-reduced while the encoder's backward still runs): the decoder reads
            # detached leaves, whose gradients the trainer feeds into the encoder's graph in a second backward call
            leaves = [t.detach().requires_grad_(True) for t in taps]
            self.cuts.append((taps, leaves))
            taps = leaves
        conv1, conv2, conv3, conv4, conv5 = taps
        outputs = self.decode(conv1, conv2, conv3, conv4, conv5, height, width)
        return outputs

    def decode(self, conv1, conv2, conv3, conv4, conv5, height, width):
        heads = (self.depth0, self.depth1, self.depth2, self.depth3)
        # the depth activation of all four scales runs as one launch after the decoder (only the raw predictions feed
        # the next decoder level)
        batched = _BATCHED_HEADS and conv5.is_cuda and hasattr(self.depth0.predict_depth, "with_disparity_multi")
        upconv4 = self.up4(conv5, [conv4])
        upconv3 = self.up3(upconv4, [conv3])
        # (split=True: the level's feature map comes back as an alias for the next level -- its two gradients meet inside the
        #  prediction head's backward launch instead of in an aten add)
        depth3, dpconv2_up, dpconv3, upconv3 = self.depth3(upconv3, height // 4, width // 4, not batched, split=True)
        upconv2 = self.up2(upconv3, [conv2, dpconv2_up])
        depth2, dpconv1_up, dpconv2, upconv2 = self.depth2(upconv2, height // 2, width // 2, not batched, split=True)
        upconv1 = self.up1(upconv2, [conv1, dpconv1_up])
        depth1, dpconv0_up, dpconv1, upconv1 = self.depth1(upconv1, height, width, not batched, split=True)
        upconv0 = self.up0(upconv1, [dpconv0_up])
        depth0, _, dpconv0 = self.depth0(upconv0, height, width, not batched)
        if batched:
            with torch.autocast(device_type=conv5.device.type, enabled=False):
                (depth0, depth1, depth2, depth3), disps = self.depth0.predict_depth.with_disparity_multi(
                    [dpconv0, dpconv1, dpconv2, dpconv3])
            for h, d in zip(heads, disps):
                h.last_disp = d

        def nhwc1(x):      # [B,1,h,w] -> [B,h,w,1]: same memory, the reference's axis order
            return x.contiguous().reshape(x.shape[0], x.shape[2], x.shape[3], 1)

        out = {"depth_ms": [nhwc1(depth0), nhwc1(depth1), nhwc1(depth2), nhwc1(depth3)],
               "debug_out": [dpconv0, upconv0, dpconv3, upconv3]}
        disps = [h.last_disp for h in heads]
        for h in heads:        # hand the tensors over: a module attribute would keep this step's autograd graph (and its
            h.last_disp = None  # AccumulateGrad nodes with their stream) alive into the next step / a hipGraph capture
        if all(d is not None for d in disps):      # model_wrappers.py:48-49 would compute it with four more elementwise ops per scale
            out["disp_ms"] = [nhwc1(d) for d in disps]
        return out
