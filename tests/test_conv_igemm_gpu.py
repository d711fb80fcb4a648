"""The matrix-core convolutions (csrc/xpt_conv.hip, xpt_conv_wgrad.hip) against a plain fp32 PyTorch reference of the
same op on the same bf16-rounded operands: forward (+ bias + LeakyReLU, TF-SAME padding, nearest-2x input), data
gradient (+ 2x2 fold, stride-2 residue classes) and weight / bias gradients, over every layer shape of PoseNetImproved
(model/build_model/pose_net.py:57-91) and of the depth decoder (model/build_model/depth_net.py:101-109, 137-167).

Tolerances: products of bf16 operands are exact in fp32, both sides accumulate in fp32 -> the forward differs only by
the final bf16 rounding of y (2^-8 relative) and the summation order; gradients see one more bf16 rounding (g)."""
import math

import pytest
import torch

from xpt_mde_2021_amd.hip.lib import half as _half_dtype

HALF = _half_dtype()      # 16-bit activation dtype of this process: bf16, or fp16 under XPT_HALF=fp16 (tests/test_fp16_build_gpu.py)
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def same_pad(n, k, s):
    total = max((math.ceil(n / s) - 1) * s + k - n, 0)
    return total // 2, total - total // 2


def reference(x, w, b, stride, slope, upsample, valid=False):
    """fp32 CPU-style reference: F.pad (TF SAME, asymmetric) + conv2d + bias + leaky_relu."""
    if upsample:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    k = w.shape[-1]
    if not valid:
        (pt, pb), (pl, pr) = same_pad(x.shape[2], k, stride), same_pad(x.shape[3], k, stride)
        x = F.pad(x, (pl, pr, pt, pb))
    y = F.conv2d(x, w, b, stride)
    return F.leaky_relu(y, slope) if slope != 1.0 else y


# (cin, cout, k, stride, H, W, upsample): PoseNet c0..c7 + 1x1 head, decoder up4..up0 first / second convolutions
SHAPES = [
    (15, 32, 5, 2, 128, 416, False), (32, 32, 5, 2, 64, 208, False), (32, 64, 3, 2, 32, 104, False),
    (64, 128, 3, 2, 16, 52, False), (128, 256, 3, 2, 8, 26, False), (256, 256, 3, 2, 4, 13, False),
    (256, 256, 3, 1, 2, 7, False), (256, 24, 1, 1, 2, 7, False),
    (1056, 256, 3, 1, 4, 13, True), (432, 256, 3, 1, 8, 26, False), (256, 128, 3, 1, 8, 26, True),
    (216, 128, 3, 1, 16, 52, False), (128, 64, 3, 1, 16, 52, True), (87, 64, 3, 1, 32, 104, False),
    (64, 32, 3, 1, 32, 104, True), (65, 32, 3, 1, 64, 208, False), (32, 16, 3, 1, 64, 208, True),
    (17, 16, 3, 1, 128, 416, False),
]


@pytest.fixture
def conv_plan():
    """Forces one kernel instantiation for forward + data gradient (xpt_conv2d_tune), restores the automatic choice."""
    from xpt_mde_2021_amd.hip import lib as _lib
    lib = _lib.load()

    def set_plan(plan):
        # 7002: as plan 0, with the specialised persistent weight-gradient kernel (conv_wgrad_fast_kernel) on EVERY 3 x 3 stride-1
        # layer of at most 32 output channels (the product uses it from 512 tiles on); every other plan: from 512 tiles (none here)
        _lib.check(lib.xpt_conv2d_bwd_weight_tune(-1000, 2 if plan == 7002 else 1), "wgrad tune")
        if plan == 7002:
            plan = 0
        if plan == 0:             # the product's automatic choice among ALL kernel families
            _lib.check(lib.xpt_conv2d_tune(0), "tune")
            _lib.check(lib.xpt_conv2d_splitk_tune(1, 0, 1024, 8192), "splitk tune")
            _lib.check(lib.xpt_conv2d_stream_tune(1, 512, 3, 80), "stream tune")
        elif plan >= 6000:        # 6000 + w: the persistent weight-stationary kernels (csrc/xpt_conv_stream.hip), w workgroups per CU,
            _lib.check(lib.xpt_conv2d_tune(0), "tune")                      # on EVERY 3 x 3 stride-1 layer they can hold
            _lib.check(lib.xpt_conv2d_splitk_tune(0, 0, 0, 0), "splitk tune")
            # (6100 + w: the generic kernel for every shape; 6000 + w: the specialised instantiations where they exist)
            _lib.check(lib.xpt_conv2d_stream_tune(2 if plan >= 6100 else 1, 1, plan % 100, 150), "stream tune")
        elif plan >= 5000:        # 5000 + n: the split-K kernels (csrc/xpt_conv_splitk.hip) with n slices on EVERY stride-1 layer
            _lib.check(lib.xpt_conv2d_tune(0), "tune")
            _lib.check(lib.xpt_conv2d_stream_tune(0, 0, 0, 0), "stream tune")
            # (5100 + n: four waves per workgroup instead of eight on the 128-channel tile)
            _lib.check(lib.xpt_conv2d_splitk_tune(4 if plan >= 5100 else 8, plan % 100, 1, 1 << 30), "splitk tune")
        else:                     # the kernels of csrc/xpt_conv.hip, one instantiation forced
            _lib.check(lib.xpt_conv2d_splitk_tune(0, 0, 0, 0), "splitk tune")
            _lib.check(lib.xpt_conv2d_stream_tune(0, 0, 0, 0), "stream tune")
            _lib.check(lib.xpt_conv2d_tune(plan), "tune")

    yield set_plan
    lib.xpt_conv2d_bwd_weight_tune(-1000, 1)
    lib.xpt_conv2d_tune(0)
    lib.xpt_conv2d_splitk_tune(8, 0, 1024, 8192)
    lib.xpt_conv2d_stream_tune(1, 512, 3, 80)


# 0: automatic choice; 901 / 902: the LDS-staged kernel (32 / 64 output channels per workgroup) forced on EVERY layer
# shape (ragged tiles, residue classes, quad fold, 8-channel inputs); 110: the direct 32 x 32 kernel forced likewise;
# 911 / 912: the halo-tile kernel (32 / 64 output channels per workgroup) on every stride-1 layer, forward and data gradient
@pytest.mark.parametrize("cin,cout,k,stride,H,W,ups", SHAPES)
# 5001 / 5002 / 5008 / 5016: the split-K tile kernels (1 / 2 / 8 / 16 slices of the reduction axis) on every stride-1 layer
# 6001 / 6003 / 6102: the persistent weight-stationary kernels, 1 / 3 workgroups per CU (6102: generic kernel only, 2 per CU), on
# every 3 x 3 stride-1 layer that fits
@pytest.mark.parametrize("plan", [0, 901, 902, 110, 911, 912, 5001, 5002, 5008, 5016, 5104, 6001, 6003, 6102, 7002])
@pytest.mark.parametrize("batch", [2])
def test_conv_fwd_bwd_matches_fp32_reference(gpu_device, conv_plan, cin, cout, k, stride, H, W, ups, batch, plan):
    from xpt_mde_2021_amd.hip import conv as xc
    dev = gpu_device
    conv_plan(plan)
    g = torch.Generator().manual_seed(cin * 131 + cout * 7 + k)
    cp = xc.round_up(cin, 8)
    x = torch.randn(batch, cin, H, W, generator=g).to(HALF)
    w = (torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).to(HALF).float()
    b = 0.1 * torch.randn(cout, generator=g)
    slope = 0.1 if cout != 24 else 1.0
    # reference on the CPU in fp32 (operands already rounded to bf16)
    xr = x.float().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True)
    yr = reference(xr, wr, br, stride, slope, ups)
    gy = torch.randn(yr.shape, generator=g).to(HALF)
    (yr * gy.float()).sum().backward()
    # device
    xd = F.pad(x, (0, 0, 0, 0, 0, cp - cin)).to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = w.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True)
    yd = xc.conv2d_same(xd, wd, bd, stride, slope, ups)
    assert yd.shape == yr.shape and yd.dtype == HALF
    (yd.float() * gy.to(dev).float()).sum().backward()
    torch.cuda.synchronize()

    def close(a, ref, rtol, what):
        a, ref = a.detach().float().cpu(), ref.detach().float()
        scale = ref.abs().max().item() + 1e-12
        err = (a - ref).abs().max().item() / scale
        assert err < rtol, f"{what}: max error {err:.3e} of the largest magnitude (allowed {rtol})"

    close(yd, yr, 6e-3, "forward")                    # bf16 output rounding (2^-8) + accumulation order
    close(xd.grad[:, :cin], xr.grad, 1.5e-2, "data gradient")
    assert float(xd.grad[:, cin:].abs().max()) == 0.0 if cp > cin else True
    close(wd.grad, wr.grad, 1.5e-2, "weight gradient")
    close(bd.grad, br.grad, 1.5e-2, "bias gradient")


def test_conv_valid_stem_and_channel_slice_input(gpu_device):
    """keras padding="valid" stride-2 stem (3 -> 32 channels on a 130x418 image) and an input that is a channel slice of a
    wider tensor (consumed through its pixel pitch, no copy)."""
    from xpt_mde_2021_amd.hip import conv as xc
    dev = gpu_device
    g = torch.Generator().manual_seed(5)
    x = torch.randn(2, 3, 66, 98, generator=g).to(HALF)
    w = (0.2 * torch.randn(32, 3, 3, 3, generator=g)).to(HALF).float()
    yr = reference(x.float(), w, None, 2, 1.0, False, valid=True)
    xd = F.pad(x, (0, 0, 0, 0, 0, 5)).to(dev).contiguous(memory_format=torch.channels_last)
    yd = xc.conv2d_same(xd, w.to(dev), None, 2, 1.0, valid=True)
    assert yd.shape == yr.shape
    assert (yd.float().cpu() - yr).abs().max().item() < 6e-3 * yr.abs().max().item()
    # slice input: channels 8..39 of a 48-channel tensor
    wide = torch.randn(2, 48, 12, 20, generator=g).to(HALF)
    w2 = (0.1 * torch.randn(16, 32, 3, 3, generator=g)).to(HALF).float()
    yr2 = reference(wide[:, 8:40].float(), w2, None, 1, 0.1, False)
    wd = wide.to(dev).contiguous(memory_format=torch.channels_last)
    yd2 = xc.conv2d_same(wd[:, 8:40], w2.to(dev), None, 1, 0.1)
    assert (yd2.float().cpu() - yr2).abs().max().item() < 6e-3 * yr2.abs().max().item()


def test_packer_tracks_weight_updates(gpu_device):
    """The packed bf16 operands follow the fp32 master after packer.pack() (what every model forward launches first)."""
    from xpt_mde_2021_amd.hip import conv as xc
    dev = gpu_device
    w = torch.randn(16, 8, 3, 3, device=dev)
    x = torch.randn(1, 8, 6, 10, device=dev).to(HALF).contiguous(memory_format=torch.channels_last)
    y0 = xc.conv2d_same(x, w, None, 1, 1.0).float()
    w.mul_(2.0)
    xc.packer.pack()
    y1 = xc.conv2d_same(x, w, None, 1, 1.0).float()
    torch.cuda.synchronize()
    assert torch.allclose(y1, 2 * y0, rtol=2e-2, atol=1e-3)


@pytest.mark.parametrize("C,H,W", [(16, 32, 104), (32, 16, 52), (64, 8, 26), (128, 5, 13)])
def test_head_conv_fwd_bwd(gpu_device, C, H, W):
    """Decoder prediction head (depth_net.py:87-92): Conv2D(1, 3, "same", linear) in fp32 arithmetic on bf16 features."""
    from xpt_mde_2021_amd.hip import conv as xc
    dev = gpu_device
    g = torch.Generator().manual_seed(C)
    x = torch.randn(3, C, H, W, generator=g).to(HALF)
    w = 0.1 * torch.randn(1, C, 3, 3, generator=g)
    b = torch.tensor([0.3])
    xr, wr, br = x.float().requires_grad_(True), w.clone().requires_grad_(True), b.clone().requires_grad_(True)
    yr = F.conv2d(xr, wr, br, 1, 1)
    gy = torch.randn(yr.shape, generator=g)
    (yr * gy).sum().backward()
    xd = x.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    wd = w.to(dev).contiguous(memory_format=torch.channels_last).requires_grad_(True)
    bd = b.to(dev).requires_grad_(True)
    yd = xc.head_conv(xd, wd, bd)
    assert yd.dtype == torch.float32 and yd.shape == yr.shape
    (yd * gy.to(dev)).sum().backward()
    torch.cuda.synchronize()
    assert (yd.cpu() - yr).abs().max().item() < 1e-4 * max(yr.abs().max().item(), 1.0)          # fp32 both sides
    assert (xd.grad.float().cpu() - xr.grad).abs().max().item() < 6e-3 * xr.grad.abs().max().item()   # bf16 dx
    assert (wd.grad.cpu() - wr.grad).abs().max().item() < 1e-4 * wr.grad.abs().max().item()
    assert abs(bd.grad.item() - br.grad.item()) < 1e-4 * max(abs(br.grad.item()), 1.0)


@pytest.mark.parametrize("B,H,W", [(2, 128, 416), (1, 37, 53), (3, 64, 192)])
def test_stem_input_matches_the_reference_preprocessing(gpu_device, B, H, W):
    """xpt_stem_input == PretrainedModel's preprocessing (pretrained_nets.py:36-43: image / 127.5 - 1, TF2 bilinear resize
    to (H+2, W+2)) followed by the bf16 cast and the 3 -> 8 channel padding the stem convolution reads; the frame is read in
    place out of a [B, S, H, W, 3] snippet tensor."""
    from xpt_mde_2021_amd.hip import conv as xc
    g = torch.Generator().manual_seed(H)
    image5d = (torch.rand(B, 5, H, W, 3, generator=g) * 255.0).to(gpu_device)
    image = image5d[:, -1].permute(0, 3, 1, 2)                      # [B,3,H,W] view of the NHWC target frame
    with torch.autocast("cuda", dtype=HALF):
        assert xc.stem_input_usable(image)
        got = xc.stem_input(image)
    assert not xc.stem_input_usable(image)                          # outside bf16 autocast: the tensor-op path
    x = image / 127.5 - 1.0
    want = F.interpolate(x, size=(H + 2, W + 2), mode="bilinear", align_corners=False, antialias=False).to(HALF)
    assert got.shape == (B, 8, H + 2, W + 2) and got.dtype == HALF
    assert got.is_contiguous(memory_format=torch.channels_last)
    assert float(got[:, 3:].abs().max()) == 0.0
    diff = (got[:, :3].float() - want.float()).abs()
    # same fp32 expression, then one bf16 rounding: a last-bit difference of the fp32 value can flip that rounding
    # (half has three more mantissa bits: the ulp at 1 is 2^-10 there, and proportionally more values sit next to a rounding boundary)
    ulp, flips = (2 ** -7, 1e-3) if HALF == torch.bfloat16 else (2 ** -10, 5e-3)
    assert (diff > 0).float().mean().item() < flips and diff.max().item() <= ulp


@pytest.mark.parametrize("channels_last", [True, False])
def test_deferred_weight_gradient_lands_unpermuted_whatever_the_weight_layout(gpu_device, channels_last):
    """FlatParameters over k x k kernels stored channels_last (the default) or plain NCHW: the gradient that ends up in the
    flat buffer must be dW in the PARAMETER's own element order either way.  (A plain-NCHW kernel gets no deferred-gradient
    sink: the dense convolution's partials are laid out [cout][kh][kw][cin] and would land permuted without any error.)"""
    from xpt_mde_2021_amd.hip import conv as xc, ops
    from xpt_mde_2021_amd.model.model_util.optimizers import FlatParameters
    dev = gpu_device
    g = torch.Generator().manual_seed(5)
    cin, cout, k, H, W = 16, 32, 3, 16, 24
    conv = torch.nn.Conv2d(cin, cout, k, bias=True)
    with torch.no_grad():
        conv.weight.copy_((torch.randn(cout, cin, k, k, generator=g) / math.sqrt(cin * k * k)).to(HALF).float())
    conv = conv.to(dev)
    if channels_last:
        conv = conv.to(memory_format=torch.channels_last)
    flat = FlatParameters([conv.weight, conv.bias])
    assert hasattr(conv.weight, "flat_grad") == channels_last
    x = torch.randn(2, cin, H, W, generator=g).to(HALF)
    gy = torch.randn(2, cout, H, W, generator=g).to(HALF)
    xr = x.float()
    wr = conv.weight.detach().float().cpu().contiguous().requires_grad_(True)
    (reference(xr, wr, conv.bias.detach().float().cpu(), 1, 0.1, False) * gy.float()).sum().backward()
    xd = x.to(dev).contiguous(memory_format=torch.channels_last)
    with torch.autocast("cuda", dtype=HALF):
        yd = xc.conv2d_same(xd, conv.weight, conv.bias, 1, 0.1, False)
    (yd.float() * gy.to(dev).float()).sum().backward()
    ops.grad_sink.flush()
    flat.gather_grads()
    torch.cuda.synchronize()
    got = flat.grad_views[0].detach().float().cpu()                 # the parameter-shaped view of the flat gradient buffer
    scale = wr.grad.abs().max().item()
    assert (got - wr.grad).abs().max().item() < 1.5e-2 * scale


@pytest.mark.parametrize("chans", [(16, 1), (32, 32, 1), (64, 22, 1), (8,), (24, 8, 16, 3)])
def test_concat_channels_matches_torch_cat_with_zero_pad(gpu_device, chans):
    """xpt_concat_channels (the decoder's concat([upconv, skip, up-sampled prediction]) + the pad to 8-channel groups,
    depth_net.py:104-107): values equal torch.cat + zeros bit for bit, gradients are the channel slices; inputs include a
    channel slice of a wider tensor (row pitch) and a one-channel map in NCHW-contiguous layout."""
    from xpt_mde_2021_amd.hip import ops
    dev = gpu_device
    g = torch.Generator().manual_seed(sum(chans))
    B, H, W = 2, 6, 10
    parts = []
    for i, c in enumerate(chans):
        if i == 1 and c > 1:         # a channel slice with a row pitch
            wide = torch.randn(B, c + 8, H, W, generator=g).to(dev, HALF).contiguous(memory_format=torch.channels_last)
            t = wide[:, 8:]
        elif c == 1:
            t = torch.randn(B, 1, H, W, generator=g).to(dev, HALF)
        else:
            t = torch.randn(B, c, H, W, generator=g).to(dev, HALF).contiguous(memory_format=torch.channels_last)
        parts.append(t.detach().requires_grad_(True))
    out = ops.concat_channels(parts)
    total = sum(chans)
    ct = -(-total // 8) * 8
    assert out.shape == (B, ct, H, W) and out.is_contiguous(memory_format=torch.channels_last)
    ref = torch.cat([p.detach() for p in parts] + ([torch.zeros(B, ct - total, H, W, device=dev, dtype=HALF)] if ct > total else []), dim=1)
    assert torch.equal(out.detach(), ref)
    gy = torch.randn(B, ct, H, W, generator=g).to(dev, HALF).contiguous(memory_format=torch.channels_last)
    out.backward(gy)
    off = 0
    for p, c in zip(parts, chans):
        assert torch.equal(p.grad, gy[:, off:off + c])
        off += c
