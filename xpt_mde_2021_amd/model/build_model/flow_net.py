"""PWCNet (reference: model/build_model/flow_net.py:10-196) as a torch module on MI355X.

Structure exactly as the reference builds it: SEPARATE encoders for the target ("_l") and the source ("_r") frames
(`pwc_encode`, :66-85: six levels of three 3x3 convolutions, 16/32/64/96/128/196 channels, stride 2 first), the
target features tiled numsrc times (:87-99), per level p = 6..2 a correlation cost volume between the target features
and the source features warped by the up-sampled flow of level p+1 (`upconv_flow`, :112-129; flow scales 0.625, 1.25,
2.5, 5.0), a DenseNet-style estimator (`predict_flow`, :131-152: 128-128-96-64-32 with every output concatenated to
its input, a linear 2-channel flow head, two Conv2DTranspose(2, 4, 2, "same") up-samplers for the flow and the last
feature), and the dilated context network on level 2 (:154-163).  Output: {"flow_ms": [flow2, flow3, flow4, flow5]}
as [batch, numsrc, H/2^p, W/2^p, 2].

The cost volume is the gfx950 kernel pair of csrc/xpt_corr.hip (tfa.layers.CorrelationCost semantics); the feature warp
follows tfa.image.dense_image_warp (bilinear, border-clamped, flow channel 0 = rows) through the library's grid sampler;
convolutions go through the same Conv2DSame factory as PoseNet / the decoder (MIOpen, fp32 weight gradients).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from ...hip import ops as _ops
from ...utils.util_class import WrongInputException
from ..model_util import layer_ops as lo


def correlation_cost(cl, cr, max_disp, stride2):
    """NCHW-indexed feature maps -> [B, D*D, H, W] (flow_net.py:181-196)."""
    if cl.is_cuda:
        return _ops.correlation_cost(cl, cr, max_disp, stride2)
    rad = max_disp // stride2
    pad = rad * stride2
    B, C, H, W = cl.shape
    rp = F.pad(cr, (pad, pad, pad, pad))
    out = [(cl * rp[:, :, pad + (ty - rad) * stride2:pad + (ty - rad) * stride2 + H,
                   pad + (tx - rad) * stride2:pad + (tx - rad) * stride2 + W]).mean(dim=1)
           for ty in range(2 * rad + 1) for tx in range(2 * rad + 1)]
    return torch.stack(out, dim=1)


def dense_image_warp(image, flow):
    """tfa.image.dense_image_warp on NCHW-indexed tensors: out[b,:,y,x] = bilinear(image[b], (y - flow[b,0,y,x],
    x - flow[b,1,y,x])), queries clamped to the border (interpolate_bilinear: floor in [0, size-2], fraction in [0,1])."""
    B, C, H, W = image.shape
    flow = flow.float()
    ys = torch.arange(H, dtype=torch.float32, device=image.device).view(1, H, 1)
    xs = torch.arange(W, dtype=torch.float32, device=image.device).view(1, 1, W)
    qy, qx = ys - flow[:, 0], xs - flow[:, 1]
    grid = torch.stack([qx * (2.0 / max(W - 1, 1)) - 1.0, qy * (2.0 / max(H - 1, 1)) - 1.0], dim=-1)
    out = F.grid_sample(image.float(), grid, mode="bilinear", padding_mode="border", align_corners=True)
    return out.to(image.dtype)


_LIBRARY_UPCONV = __import__("os").environ.get("XPT_DEBUG_LIBRARY_UPCONV", "0") == "1"      # A/B switch


class _UpConvFp32(torch.autograd.Function):
    """conv_transpose2d(x, W, stride 2, padding 1) in fp32 with a replay-safe weight gradient: the transposed
    convolution is the adjoint of z = conv2d(u, W, stride 2, padding 1), so dW is that convolution's weight gradient
    with (dz, u) = (x, dy) -- one GEMM on the unfolded dy (layer_ops.unfolded_weight_grad)."""

    @staticmethod
    def forward(ctx, x, weight):
        ctx.save_for_backward(x, weight)
        lo.note_library_conv(x)
        return F.conv_transpose2d(x, weight, None, 2, 1)

    @staticmethod
    def backward(ctx, dy):
        x, weight = ctx.saved_tensors
        dx = F.conv2d(dy, weight, None, 2, 1) if ctx.needs_input_grad[0] else None
        dw = lo.unfolded_weight_grad(x, dy, weight.shape, 2, 1, 1) if ctx.needs_input_grad[1] else None
        return dx, dw


class _UpConv(nn.Module):
    """layers.Conv2DTranspose(filters, kernel_size=4, strides=2, padding="same") (flow_net.py:145-148): with k = 4, s = 2
    TF's SAME padding is symmetric (1, 1), i.e. exactly conv_transpose2d(padding=1); Keras defaults: bias, linear,
    glorot_uniform.  Evaluated in fp32 without the library's bias path: bf16 weight gradients and library bias
    gradients do not survive hipGraph replay on this stack, and the maps have 2 / 32 channels."""

    def __init__(self, cin, filters):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(cin, filters, 4, 4))
        self.bias = nn.Parameter(torch.zeros(filters))
        nn.init.xavier_uniform_(self.weight)

    def forward(self, x):
        with torch.autocast(device_type=x.device.type, enabled=False):
            y = F.conv_transpose2d(x.float(), self.weight, None, 2, 1) if _LIBRARY_UPCONV else \
                _UpConvFp32.apply(x.float(), self.weight)
            if y.is_cuda:
                return _ops.bias_act(y.contiguous(memory_format=torch.channels_last), self.bias, 1.0)
            return y + self.bias.view(1, -1, 1, 1)


class _FlowEstimator(nn.Module):
    """predict_flow (flow_net.py:131-152)."""

    def __init__(self, conv2d, cin, up):
        super().__init__()
        self.convs = nn.ModuleList()
        for filters in (128, 128, 96, 64):
            self.convs.append(conv2d(cin, filters))
            cin += filters
        self.last = conv2d(cin, 32)
        self.out = conv2d(32, 2, activation="linear")
        self.up = up
        if up:
            self.up_flow = _UpConv(2, 2)
            self.up_feat = _UpConv(32, 2)

    def forward(self, inputs):
        x = torch.cat(inputs, dim=1)
        for conv in self.convs:
            x = torch.cat([x, conv(x)], dim=1)
        c = self.last(x)
        flow = self.out(c)
        if self.up:
            return flow, self.up_flow(flow), self.up_feat(c)
        return flow, c


class PWCNet(nn.Module):
    CHANNELS = (16, 32, 64, 96, 128, 196)
    FLOW_SCALE = {5: 0.625, 4: 1.25, 3: 2.5, 2: 5.0}

    def __init__(self, total_shape, global_batch, conv2d):
        super().__init__()
        self.total_shape = tuple(total_shape)
        self.max_displacement = 128                            # flow_net.py:16
        channel = self.total_shape[-1]
        self.enc_l = self._encoder(conv2d, channel)
        self.enc_r = self._encoder(conv2d, channel)
        self.flow6 = _FlowEstimator(conv2d, self.corr_channels(6), up=True)
        self.flow5 = _FlowEstimator(conv2d, self.corr_channels(5) + self.CHANNELS[4] + 4, up=True)
        self.flow4 = _FlowEstimator(conv2d, self.corr_channels(4) + self.CHANNELS[3] + 4, up=True)
        self.flow3 = _FlowEstimator(conv2d, self.corr_channels(3) + self.CHANNELS[2] + 4, up=True)
        self.flow2 = _FlowEstimator(conv2d, self.corr_channels(2) + self.CHANNELS[1] + 4, up=False)
        self.context = nn.ModuleList([conv2d(32, 128, 3, dilation_rate=1), conv2d(128, 128, 3, dilation_rate=2),
                                      conv2d(128, 128, 3, dilation_rate=4), conv2d(128, 96, 3, dilation_rate=8),
                                      conv2d(96, 64, 3, dilation_rate=16), conv2d(64, 32, 3, dilation_rate=1),
                                      conv2d(32, 2, 3, activation="linear")])
        for m in self.modules():            # small pyramid levels: data gradients through forward solvers (layer_ops)
            if isinstance(m, lo.Conv2DSame):
                m.replay_safe_dgrad = True

    def _encoder(self, conv2d, cin):
        levels = nn.ModuleList()
        for filters in self.CHANNELS:
            levels.append(nn.Sequential(conv2d(cin, filters, 3, strides=2), conv2d(filters, filters, 3),
                                        conv2d(filters, filters, 3)))
            cin = filters
        return levels

    def corr_params(self, p):
        md = self.max_displacement // 2 ** p
        return md, max(md // 4, 1)

    def corr_channels(self, p):
        md, stride2 = self.corr_params(p)
        return (2 * (md // stride2) + 1) ** 2

    @staticmethod
    def _encode(levels, x):
        feats = []
        for level in levels:
            x = level(x)
            feats.append(x)
        return feats

    def correlation(self, cl, cr, p):
        md, stride2 = self.corr_params(p)
        return correlation_cost(cl, cr, md, stride2)

    def upconv_flow(self, p, estimator, cp_l, cp_r, up_flowq, up_featq):
        cp_r_warp = dense_image_warp(cp_r, up_flowq * self.FLOW_SCALE[p])
        corrp = self.correlation(cp_l, cp_r_warp, p)
        return estimator([corrp, cp_l, up_flowq.to(corrp.dtype), up_featq.to(corrp.dtype)])

    def forward(self, image5d):
        batch, snippet, height, width, channel = image5d.shape
        if height % 64 or width % 64:
            # six stride-2 levels and four x2 transposed convolutions must meet again: the reference graph fails to
            # build (concat shape mismatch) for other sizes as well
            raise WrongInputException(f"PWCNet needs height and width divisible by 64, got {height}x{width}")
        numsrc = snippet - 1
        frames = image5d.permute(0, 1, 4, 2, 3)                                # [B, S, C, H, W]
        target = frames[:, -1]
        sources = frames[:, :-1].reshape(batch * numsrc, channel, height, width)
        if image5d.is_cuda:
            target = target.contiguous(memory_format=torch.channels_last)
            sources = sources.contiguous(memory_format=torch.channels_last)
        feats_l = self._encode(self.enc_l, target)
        feats_r = self._encode(self.enc_r, sources)
        # repeat_features (:87-99): every target feature numsrc times, batch-major
        feats_l = [f.unsqueeze(1).expand(-1, numsrc, -1, -1, -1).reshape(batch * numsrc, *f.shape[1:]) for f in feats_l]
        c2l, c3l, c4l, c5l, c6l = feats_l[1:]
        c2r, c3r, c4r, c5r, c6r = feats_r[1:]

        corr6 = self.correlation(c6l, c6r, 6)
        flow6, up_flow6, up_feat6 = self.flow6([corr6])
        flow5, up_flow5, up_feat5 = self.upconv_flow(5, self.flow5, c5l, c5r, up_flow6, up_feat6)
        flow4, up_flow4, up_feat4 = self.upconv_flow(4, self.flow4, c4l, c4r, up_flow5, up_feat5)
        flow3, up_flow3, up_feat3 = self.upconv_flow(3, self.flow3, c3l, c3r, up_flow4, up_feat4)
        flow2, flow_feat2 = self.upconv_flow(2, self.flow2, c2l, c2r, up_flow3, up_feat3)

        c = flow_feat2
        for conv in self.context:
            c = conv(c)
        flow2 = c + flow2
        # reshape_batch_back (:101-110): [B*N, 2, h, w] -> [B, N, h, w, 2]
        flow_ms = [f.float().permute(0, 2, 3, 1).reshape(batch, numsrc, f.shape[2], f.shape[3], 2)
                   for f in (flow2, flow3, flow4, flow5)]
        return {"flow_ms": flow_ms}
