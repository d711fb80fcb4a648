"""the min-over-sources loss family -- MonoDepth2LossMultiScale, MoALossMultiScale,
MD2CombLossMultiScale (model/loss_and_metric/losses.py:198-232, 282-321, 324-374) -- on the HIP per-pixel kernels
(reduce=False) against the fp64 oracle restatement (oracle/ref_loss.py), values and gradients w.r.t. every synthesized
view; plus hand cases of the restatement itself on the CPU."""
import numpy as np
import pytest
import torch

from oracle import ref_loss
from xpt_mde_2021_amd.utils import synthetic_data as sd

SCALE_WEIGHTS = np.array([1.0, 0.5, 0.25, 0.125])


def make_views(B, N, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    target = sd.smooth_noise((B, H, W, 3), g)
    ms = []
    for s in (1, 2, 4, 8):
        views = torch.stack([sd.smooth_noise((B, H // s, W // s, 3), g) for _ in range(N)], dim=1)
        views[:, :, : max(H // s // 8, 1)] = 0.0            # an out-of-view band (all-zero synthesis) per view
        ms.append(views)
    return target, ms


def test_md2_hand_case_cpu():
    """Two views, one exact and one far off: the minimum picks the exact one -> L1 loss 0; swapping the order changes
    nothing; with both views off by constants 0.2 / 0.6 the minimum is 0.2 everywhere."""
    target = torch.rand(1, 4, 6, 3, dtype=torch.float64)
    exact, off = target.unsqueeze(1), (target + 0.5).unsqueeze(1)
    for views in (torch.cat([exact, off], 1), torch.cat([off, exact], 1)):
        loss = ref_loss.monodepth2_loss_multi_scale("L1", [views], target, np.array([1.0]))
        assert float(loss.abs().max()) < 1e-12
    views = torch.cat([(target + 0.2).unsqueeze(1), (target - 0.6).unsqueeze(1)], 1)
    loss = ref_loss.monodepth2_loss_multi_scale("L1", [views], target, np.array([1.0]))
    assert abs(float(loss) - 0.2) < 1e-12


def test_md2comb_hand_case_cpu():
    """static = 0.3 in view 0 and 0.05 in view 1; flow loss 0.1 in both: view 0 exceeds 2 x 0.1 and is pushed to 1000.3,
    view 1 (0.05 < 0.2) survives -> min = 0.05 everywhere -> sum / count = 0.05.  With the flow loss at 0.01 both views
    are rejected: every element >= 1000, nothing is kept, the reference divides 0 by 0 (NaN) -- restated as is."""
    target = torch.rand(1, 4, 6, 3, dtype=torch.float64) * 0.2 + 0.3
    synth = torch.cat([(target + 0.3).unsqueeze(1), (target - 0.05).unsqueeze(1)], 1)
    warped = torch.cat([(target + 0.1).unsqueeze(1), (target + 0.1).unsqueeze(1)], 1)
    loss = ref_loss.md2comb_loss_multi_scale("L1", [synth], [warped], target, np.array([1.0]))
    assert abs(float(loss) - 0.05) < 1e-12
    warped2 = torch.cat([(target + 0.01).unsqueeze(1)] * 2, 1)
    loss2 = ref_loss.md2comb_loss_multi_scale("L1", [synth], [warped2], target, np.array([1.0]))
    assert bool(torch.isnan(loss2).all())


@pytest.mark.gpu
@pytest.mark.parametrize("method", ["L1", "SSIM"])
@pytest.mark.parametrize("kind", ["md2", "moa", "md2comb"])
def test_min_source_losses_match_oracle(gpu_device, method, kind):
    from xpt_mde_2021_amd.model.loss_and_metric import losses as pl
    B, N, H, W = 2, 4, 32, 104
    target, synth_ms = make_views(B, N, H, W, 5)
    _, other_ms = make_views(B, 1 if kind == "moa" else N, H, W, 9)
    dev = gpu_device

    def run(device, dtype, product):
        tgt = target.to(device=device, dtype=dtype)
        syn = [v.to(device=device, dtype=dtype).requires_grad_(True) for v in synth_ms]
        oth = [v.to(device=device, dtype=dtype).requires_grad_(True) for v in other_ms]
        if product:
            augm = {"synth_target_ms": syn, "target": tgt, "stereo_synth_ms": oth, "warped_target_ms": oth}
            cls = {"md2": pl.MonoDepth2LossMultiScale, "moa": pl.MoALossMultiScale, "md2comb": pl.MD2CombLossMultiScale}[kind]
            loss = cls(method, SCALE_WEIGHTS)(None, None, augm)
        elif kind == "md2":
            loss = ref_loss.monodepth2_loss_multi_scale(method, syn, tgt, SCALE_WEIGHTS)
        elif kind == "moa":
            loss = ref_loss.moa_loss_multi_scale(method, syn, oth, tgt, SCALE_WEIGHTS)
        else:
            loss = ref_loss.md2comb_loss_multi_scale(method, syn, oth, tgt, SCALE_WEIGHTS)
        loss = loss.reshape(-1)
        weights = torch.tensor([1.0, 0.7], device=device, dtype=dtype)
        (loss * weights).sum().backward()
        grads = [v.grad for v in syn] + ([v.grad for v in oth] if kind == "moa" else [])
        return loss.detach(), grads

    loss_o, grads_o = run("cpu", torch.float64, False)
    loss_d, grads_d = run(dev, torch.float32, True)
    torch.cuda.synchronize()
    assert torch.allclose(loss_d.cpu().double(), loss_o, rtol=1e-4, atol=1e-6), (loss_d, loss_o)
    for gd, go in zip(grads_d, grads_o):
        gd, scale = gd.cpu().double(), go.abs().max().item() + 1e-30
        # the arg-min over views / the < comparisons are decided in fp32 on the device: at near-ties the gradient moves from
        # one view to another, so a small fraction of elements may differ; everything else must agree to 1e-4
        bad = ((gd - go).abs() > 1e-4 * scale).double().mean().item()
        assert bad < 2e-3, bad
