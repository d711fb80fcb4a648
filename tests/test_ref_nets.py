"""PoseNetImproved and the depth decoder against the INDEPENDENT restatement oracle/ref_nets.py (plain F.pad + conv2d,
NHWC / HWIO conventions of the reference, no code shared with the product): a padding, channel-order or concat-order
error in the product modules shows up here.  Reference: model/build_model/pose_net.py:44-91,
model/build_model/depth_net.py:76-109, 137-167, model/model_util/layer_ops.py:5-50."""
import pytest
import torch

from oracle import ref_nets as rn
from xpt_mde_2021_amd.config import opts
from xpt_mde_2021_amd.utils import synthetic_data as sd


def build_model(height, width, batch, high_res, dtype):
    from xpt_mde_2021_amd.model.build_model.model_factory import ModelFactory
    saved = (opts.CONV_DTYPE, opts.HIGH_RES)
    opts.CONV_DTYPE, opts.HIGH_RES = dtype, high_res
    try:
        feats = sd.make_features(batch, height, width, 5, 7)
        torch.manual_seed(11)
        model = ModelFactory(sd.tfr_config_for(feats), global_batch=batch, net_names=opts.RIGID_NET, high_res=high_res).get_model()
    finally:
        opts.CONV_DTYPE, opts.HIGH_RES = saved
    return model, feats


def hwio(conv):
    """torch [cout, cin, kh, kw] -> Keras [kh, kw, cin, cout] (+ bias), detached fp32 CPU copies."""
    return conv.weight.detach().float().cpu().permute(2, 3, 1, 0).contiguous(), conv.bias.detach().float().cpu()


def pose_params(posenet):
    return [hwio(layer.conv) for layer in posenet.convs] + [hwio(posenet.head.conv)]


def decoder_params(depthnet):
    """The decoder's weights as the restatement takes them.  The 1/4 tap reaches the decoder with the structurally-zero
    channels of its encoder cell (depth_net.DepthNetPretrained): the convolution that reads it has two more input columns
    than the reference's -- the restatement gets the logical ones."""
    P = {}
    for lvl in (4, 3, 2, 1, 0):
        up = getattr(depthnet, f"up{lvl}")
        P[f"dp_up{lvl}_conv1"], P[f"dp_up{lvl}_conv2"] = hwio(up.conv1.conv), hwio(up.conv2.conv)
    if getattr(depthnet, "skip2_zero", None) is not None:
        w, bias = P["dp_up2_conv2"]                             # [kh, kw, cin, cout]
        keep = torch.ones(w.shape[2], dtype=torch.bool)
        keep[depthnet.skip2_zero] = False
        P["dp_up2_conv2"] = (w[:, :, keep].contiguous(), bias)
    for lvl in (3, 2, 1, 0):
        P[f"dp_depth{lvl}_conv"] = hwio(getattr(depthnet, f"depth{lvl}").conv.conv)
    return P


def random_taps(depthnet, batch, height, width, seed=3):
    g = torch.Generator().manual_seed(seed)
    return [torch.randn(batch, c, height >> (i + 1), width >> (i + 1), generator=g) * 0.5
            for i, c in enumerate(depthnet.encoder.TAP_CHANNELS)]


def physical_taps(depthnet, taps):
    """Logical taps -> what the encoder hands the decoder (zero channels where a cell carries structurally-zero filters)."""
    out = []
    for t, (ch, sel) in zip(taps, depthnet.encoder.tap_layout()):
        if sel is None:
            out.append(t)
        else:
            full = torch.zeros((t.shape[0], ch) + tuple(t.shape[2:]), dtype=t.dtype)
            full[:, sel] = t
            out.append(full)
    return out


def test_structural_pins():
    """Parameter counts implied by the reference's layer lists; the product's modules must have exactly these."""
    assert rn.pose_net_parameter_count() == 2_201_592 and rn.pose_net_parameter_count(high_res=True) == 8_107_512
    model, _ = build_model(64, 192, 1, False, "fp32")
    assert sum(p.numel() for p in model.models["posenet"].parameters()) == 2_201_592
    model_hr, _ = build_model(64, 192, 1, True, "fp32")
    assert sum(p.numel() for p in model_hr.models["posenet"].parameters()) == 8_107_512


@pytest.mark.parametrize("high_res", [False, True])
def test_cpu_modules_match_restatement(high_res):
    """Product modules on the CPU (library convolutions) vs the restatement, fp32, 64x192."""
    H, W, B = 64, 192, 2
    model, feats = build_model(H, W, B, high_res, "fp32")
    posenet, depthnet = model.models["posenet"], model.models["depthnet"]
    with torch.no_grad():
        pose = posenet(feats["image5d"])["pose"]
        ref = rn.pose_net_improved(feats["image5d"], pose_params(posenet), high_res)
        assert pose.shape == ref.shape == (B, 4, 6)
        assert (pose - ref).abs().max().item() < 1e-5 * max(ref.abs().max().item(), 1e-3) + 1e-7
        taps = random_taps(depthnet, B, H, W)
        out = depthnet.decode(*physical_taps(depthnet, taps), H, W)
        ref = rn.depth_decoder([t.permute(0, 2, 3, 1) for t in taps], decoder_params(depthnet), H, W)
        for d, r in zip(out["depth_ms"], ref["depth_ms"]):
            assert d.shape == r.shape
            assert (d - r).abs().max().item() < 1e-4 * r.abs().max().item()
        for d, r in zip(out["debug_out"], ref["debug_out"]):
            assert (d.permute(0, 2, 3, 1) - r).abs().max().item() < 1e-4 * max(r.abs().max().item(), 1e-3)


@pytest.mark.gpu
@pytest.mark.parametrize("height,width,batch,high_res", [(128, 416, 4, False), (256, 832, 2, True)])
@pytest.mark.parametrize("dtype", ["fp32", "bf16"])
def test_gpu_modules_match_restatement(gpu_device, height, width, batch, high_res, dtype):
    """The modules as they run on the MI355X (bf16: the matrix-core kernels of hip/conv.py) vs the CPU restatement.
    bf16 tolerance: 8-11 layers of bf16 activations (2^-8 each) on random weights."""
    model, feats = build_model(height, width, batch, high_res, dtype)
    model.to(gpu_device)
    posenet, depthnet = model.models["posenet"], model.models["depthnet"]
    tol = 1e-4 if dtype == "fp32" else 4e-2
    with torch.no_grad():
        image = feats["image5d"].to(gpu_device)
        pose = model._run(posenet, image)["pose"].float().cpu()
        ref = rn.pose_net_improved(feats["image5d"], pose_params(posenet), high_res)
        assert (pose - ref).abs().max().item() < tol * max(ref.abs().max().item(), 1e-3)
        taps = random_taps(depthnet, batch, height, width)
        from xpt_mde_2021_amd.hip import conv as xc
        xc.packer.pack()

        def decode(*t):
            return depthnet.decode(*t, height, width)
        dev_taps = [t.to(gpu_device).contiguous(memory_format=torch.channels_last) for t in physical_taps(depthnet, taps)]
        if dtype == "bf16":
            with torch.autocast("cuda", dtype=torch.bfloat16):
                out = decode(*[t.to(torch.bfloat16) for t in dev_taps])
            ref_taps = [t.to(torch.bfloat16).float().permute(0, 2, 3, 1) for t in taps]
        else:
            out = decode(*dev_taps)
            ref_taps = [t.permute(0, 2, 3, 1) for t in taps]
        ref = rn.depth_decoder(ref_taps, decoder_params(depthnet), height, width)
        for d, r in zip(out["depth_ms"], ref["depth_ms"]):
            d = d.float().cpu()
            assert d.shape == r.shape
            assert (d - r).abs().max().item() < tol * r.abs().max().item(), (d - r).abs().max().item()
