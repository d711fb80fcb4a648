"""oracle (test infrastructure): the FlowNet branch -- PWC-Net's correlation cost volume, dense_image_warp, the
flow-warped targets and the flow-aided losses.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this package.

Follows, on PyTorch-CPU tensors:
  model/build_model/flow_net.py:126-196 (upconv_flow / predict_flow / context_network / correlation),
  model/synthesize/flow_warping.py:11-71 (FlowWarpMultiScale), model/synthesize/bilinear_interp.py:166-202,
  model/loss_and_metric/losses.py:235-279 (CombinedLossMultiScale), :497-533 (FlowWarpLossMultiScale, L2Regularizer),
  utils/util_funcs.py:178-190 (multi_scale_like_flow).

Third-party arithmetic that is not under /root/reference (tensorflow-addons==0.12.1, requirements.txt:42), restated
from its published definition:
  * tfa.layers.CorrelationCost: for every pixel and every displacement (dy, dx) on the stride_2 grid within
    max_displacement, the mean over kernel_size^2 * channels of the products of the two (zero-padded) maps; output
    channel = (dy index) * D + (dx index).  The reference's own test (flow_net.py:204-222) pins the channel count
    ((2 * (md // stride_2) + 1)^2) only: **parity of the values is unpinned** beyond this restatement.
  * tfa.image.dense_image_warp: out[b, y, x] = bilinear sample of image[b] at (y - flow[b,y,x,0], x - flow[b,y,x,1]);
    interpolate_bilinear clamps floor to [0, size - 2] and the fraction to [0, 1] (border replication).  Pinned by the
    reference's test_warp_simple (flow_net.py:225-262): equal to FlowBilinearInterpolation and to the hand-interpolated
    image in the interior for a constant flow.
"""
import torch

from . import ref_loss
from .ref_synthesize import bilinear_interpolation, flow_to_pixel_coordinates, tf_resize_bilinear


def correlation_cost(left, right, max_displacement, stride_2):
    """left, right [B,H,W,C] -> [B,H,W,D*D]; kernel_size 1, stride_1 1, pad = max_displacement."""
    B, H, W, C = left.shape
    rad = max_displacement // stride_2
    D = 2 * rad + 1
    pad = rad * stride_2
    rp = torch.zeros((B, H + 2 * pad, W + 2 * pad, C), dtype=right.dtype)
    rp[:, pad:pad + H, pad:pad + W] = right
    out = []
    for ty in range(D):
        for tx in range(D):
            dy, dx = (ty - rad) * stride_2, (tx - rad) * stride_2
            shifted = rp[:, pad + dy:pad + dy + H, pad + dx:pad + dx + W]
            out.append((left * shifted).sum(dim=-1) / C)
    return torch.stack(out, dim=-1)


def dense_image_warp(image, flow):
    """image [B,H,W,C], flow [B,H,W,2] = (dy, dx) -> [B,H,W,C]."""
    B, H, W, C = image.shape
    ys, xs = torch.meshgrid(torch.arange(H, dtype=flow.dtype), torch.arange(W, dtype=flow.dtype), indexing="ij")
    qy, qx = ys.unsqueeze(0) - flow[..., 0], xs.unsqueeze(0) - flow[..., 1]

    def split(q, size):
        fl = torch.clamp(torch.floor(q), 0, size - 2)
        return fl.long(), torch.clamp(q - fl, 0.0, 1.0)

    y0, ay = split(qy, H)
    x0, ax = split(qx, W)
    flat = image.reshape(B, H * W, C)

    def gather(yy, xx):
        idx = (yy * W + xx).reshape(B, H * W, 1).expand(B, H * W, C)
        return torch.gather(flat, 1, idx).reshape(B, H, W, C)

    tl, tr, bl, br = gather(y0, x0), gather(y0, x0 + 1), gather(y0 + 1, x0), gather(y0 + 1, x0 + 1)
    ax, ay = ax.unsqueeze(-1), ay.unsqueeze(-1)
    top = tl + ax * (tr - tl)
    bot = bl + ax * (br - bl)
    return top + ay * (bot - top)


def flow_bilinear_interpolation(image, flow):
    """bilinear_interp.py:166-180: image [B*N,H,W,C], flow [B*N,H,W,2(u,v)] -> [B*N,H,W,C] (zero outside)."""
    coords = flow_to_pixel_coordinates(flow.unsqueeze(1))
    return bilinear_interpolation(image.unsqueeze(1), coords).squeeze(1)


def multi_scale_like_flow(image, flow_ms):
    """util_funcs.py:178-190."""
    return [tf_resize_bilinear(image, flow.shape[2:4]) for flow in flow_ms]


def flow_warp_multi_scale(source_image, flow_ms):
    """flow_warping.py:17-71: sources resized to every flow scale, sampled at grid - flow."""
    B, N, H, W, _ = source_image.shape
    out = []
    for flow in flow_ms:
        h, w = flow.shape[2:4]
        src = tf_resize_bilinear(source_image.reshape(B * N, H, W, 3), (h, w)).reshape(B, N, h, w, 3)
        out.append(bilinear_interpolation(src, flow_to_pixel_coordinates(flow)))
    return out


def flow_warp_loss_multi_scale(method, warped_target_ms, flow_target_ms, scale_weights):
    """losses.py:497-519."""
    fn = ref_loss.PHOTOMETRIC[method]
    return ref_loss.merge_multi_scale_losses([fn(w, t) for w, t in zip(warped_target_ms, flow_target_ms)], scale_weights)


def l2_regularizer(weights, batch):
    """losses.py:522-533: sum_w tf.nn.l2_loss(w) = sum(w^2) / 2, tiled to [batch]."""
    loss = sum((w.double() ** 2).sum() / 2 for w in weights)
    return loss.to(weights[0].dtype).repeat(batch)


def combined_loss_multi_scale(method, synth_target_ms, warped_target_ms, original_target, scale_weights):
    """losses.py:235-279: static (depth + pose) per-pixel loss kept only where it is below the optical-flow loss."""
    fn = ref_loss.PHOTOMETRIC[method]
    Ho, Wo = original_target.shape[1:3]
    flow_loss = fn(ref_loss.resize_bilinear_5d(warped_target_ms[0], (Ho, Wo)), original_target, False)
    losses = []
    for synt in synth_target_ms:
        static = fn(ref_loss.resize_bilinear_5d(synt, (Ho, Wo)), original_target, False)
        static = static * (static < flow_loss).to(static.dtype)
        losses.append(static.mean(dim=(1, 2, 3, 4)))
    return ref_loss.merge_multi_scale_losses(losses, scale_weights)
