"""Closed-form pins for the half of the oracle the reference holds no vectors for (loss VALUES): cases small enough to
evaluate by hand from the formulas in model/loss_and_metric/loss_util.py:52-96 (SSIM, 3x3 SAME average pooling whose
divisor excludes the padding) and model/loss_and_metric/losses.py:409-440 (edge-aware smoothness), run through the
oracle (CPU) and through the HIP kernels (GPU)."""
import math

import pytest
import torch

from oracle import ref_loss

C1, C2 = 0.01 ** 2, 0.03 ** 2


def ssim_loss_from_moments(mx, my, exx, eyy, exy):
    sx, sy, sxy = exx - mx * mx, eyy - my * my, exy - mx * my
    ssim = (2 * mx * my + C1) * (2 * sxy + C2) / ((mx * mx + my * my + C1) * (sx + sy + C2))
    return min(max((1 - ssim) / 2, 0.0), 1.0)


def constant_case(a, b, h=6, w=7, n=2):
    target = torch.full((1, h, w, 3), a, dtype=torch.float64)
    synth = torch.full((1, n, h, w, 3), b, dtype=torch.float64)
    return synth, target, (1 - (2 * a * b + C1) / (a * a + b * b + C1)) / 2


def stripe_case(p, q, b, h=5, w=6):
    """target: column parity stripes (p on even columns, q on odd ones); synth: constant b."""
    cols = torch.tensor([p if c % 2 == 0 else q for c in range(w)], dtype=torch.float64)
    target = cols.view(1, 1, w, 1).expand(1, h, w, 3).contiguous()
    synth = torch.full((1, 1, h, w, 3), b, dtype=torch.float64)
    expect = torch.zeros(h, w, dtype=torch.float64)
    for r in range(h):
        for c in range(w):
            vals = [float(cols[cc]) for rr in range(max(r - 1, 0), min(r + 2, h)) for cc in range(max(c - 1, 0), min(c + 2, w))]
            mx = sum(vals) / len(vals)                       # divisor = taps inside the image: 4 corner, 6 edge, 9 interior
            exx = sum(v * v for v in vals) / len(vals)
            expect[r, c] = ssim_loss_from_moments(mx, b, exx, b * b, mx * b)
    return synth, target, expect


def test_ssim_constant_images_oracle():
    for a, b in [(0.5, 0.2), (0.8, -0.3), (0.1, 0.1)]:
        synth, target, expect = constant_case(a, b)
        got = ref_loss.photometric_loss_ssim(synth, target, False)
        assert torch.allclose(got, torch.full_like(got, min(max(expect, 0.0), 1.0)), atol=1e-12)
        assert abs(float(ref_loss.photometric_loss_ssim(synth, target)) - min(max(expect, 0.0), 1.0)) < 1e-12


def test_ssim_stripes_hand_moments_oracle():
    p, q, b = 0.9, 0.3, 0.5
    synth, target, expect = stripe_case(p, q, b)
    got = ref_loss.photometric_loss_ssim(synth, target, False)[0, 0, :, :, 0]
    assert torch.allclose(got, expect, atol=1e-12)
    # the three window kinds written out: corner (0,0) = {p,q,p,q} / 4, left edge (2,0) = {p,q} x 3 / 6,
    # interior even column (2,2) = 3 p + 6 q over 9
    corner = ssim_loss_from_moments((p + q) / 2, b, (p * p + q * q) / 2, b * b, (p + q) / 2 * b)
    interior = ssim_loss_from_moments((p + 2 * q) / 3, b, (p * p + 2 * q * q) / 3, b * b, (p + 2 * q) / 3 * b)
    assert abs(float(got[0, 0]) - corner) < 1e-12 and abs(float(got[2, 0]) - corner) < 1e-12     # same moments, divisor 4 vs 6
    assert abs(float(got[2, 2]) - interior) < 1e-12
    assert float(got[2, 2]) != float(got[2, 1])


def ramp_case(g, k, h=5, w=9):
    cols = torch.arange(w, dtype=torch.float64).view(1, 1, w, 1)
    disp = (g * cols).expand(1, h, w, 1).contiguous() + 0.5
    image = (k * cols).expand(1, h, w, 3).contiguous()
    return disp, image, 0.5 * abs(g) * math.exp(-4.0 * abs(k))


def test_smoothness_linear_ramps_oracle():
    for g, k in [(0.02, 0.05), (-0.01, 0.0), (0.03, -0.2)]:
        disp, image, expect = ramp_case(g, k)
        assert abs(float(ref_loss.smootheness_loss(disp, image)) - expect) < 1e-12
    const = torch.full((1, 4, 6, 1), 0.3, dtype=torch.float64)
    assert float(ref_loss.smootheness_loss(const, torch.rand(1, 4, 6, 3, dtype=torch.float64))) == 0.0


def test_l1_half_black_oracle():
    """a synthesized view that is black (invalid) on its left half contributes |y - x| only on the right half."""
    target = torch.full((1, 4, 8, 3), 0.6, dtype=torch.float64)
    synth = torch.full((1, 1, 4, 8, 3), 0.2, dtype=torch.float64)
    synth[:, :, :, :4] = 0.0
    assert abs(float(ref_loss.photometric_loss_l1(synth, target)) - 0.4 * 0.5) < 1e-12


@pytest.mark.gpu
def test_closed_forms_on_hip_kernels(gpu_device):
    from xpt_mde_2021_amd.hip import ops
    dev = gpu_device
    for a, b in [(0.5, 0.2), (0.8, -0.3)]:
        synth, target, expect = constant_case(a, b)
        m = ops.photometric("SSIM", synth.float().to(dev), target.float().to(dev), reduce=False)
        assert (m - expect).abs().max().item() < 2e-5          # fp32: sigma = E[x^2] - mu^2 cancels to ~1e-8, c2 = 9e-4
    synth, target, expect = stripe_case(0.9, 0.3, 0.5)
    m = ops.photometric("SSIM", synth.float().to(dev), target.float().to(dev), reduce=False)[0, 0, :, :, 0]
    assert (m.cpu().double() - expect).abs().max().item() < 2e-5
    for g, k in [(0.02, 0.05), (0.03, -0.2)]:
        disp, image, expect = ramp_case(g, k)
        s = ops.smoothness(disp.float().to(dev), image.float().to(dev), 4.0)
        assert abs(float(s) - expect) < 1e-6
    target = torch.full((1, 4, 8, 3), 0.6)
    synth = torch.full((1, 1, 4, 8, 3), 0.2)
    synth[:, :, :, :4] = 0.0
    assert abs(float(ops.photometric("L1", synth.to(dev), target.to(dev))) - 0.2) < 1e-6
