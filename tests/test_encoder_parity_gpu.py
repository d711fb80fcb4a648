"""NASNet-A Mobile encoder: the gfx950 path (wide cells: multi-branch depthwise launches, fused pointwise + BatchNorm,
the fused cell tail of csrc/xpt_celltail.hip) against the SAME module evaluated on the CPU with plain PyTorch ops
(the layer-by-layer composition of tensorflow.keras.applications.nasnet that the reference instantiates,
model/build_model/pretrained_nets.py:11-44), in fp32: the five decoder taps and parameter / input gradients."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused_tail", [True, False])
def test_encoder_taps_and_gradients_match_cpu_composition(gpu_device, fused_tail, monkeypatch):
    from xpt_mde_2021_amd.model.build_model import pretrained_nets as pn
    monkeypatch.setattr(pn, "_CELL_TAIL", fused_tail)
    torch.manual_seed(3)
    cpu = pn.NASNetMobileEncoder().eval()
    # non-trivial BatchNorm statistics (the frozen inference-mode affine must not be the identity)
    with torch.no_grad():
        for m in cpu.modules():
            if isinstance(m, pn.FrozenBatchNorm):
                m.running_mean.normal_(0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0, 0.1)
    dev = copy.deepcopy(cpu).to(gpu_device)
    image = torch.rand(2, 3, 64, 192) * 255.0
    xc = image.clone().requires_grad_(True)
    xd = image.to(gpu_device).requires_grad_(True)
    taps_c = cpu(xc)
    taps_d = dev(xd)
    assert [t.shape for t in taps_c] == [t.shape for t in taps_d]
    gens = [torch.randn(t.shape, generator=torch.Generator().manual_seed(i)) for i, t in enumerate(taps_c)]
    sum((t * g).sum() for t, g in zip(taps_c, gens)).backward()
    sum((t * g.to(gpu_device)).sum() for t, g in zip(taps_d, gens)).backward()
    torch.cuda.synchronize()
    for k, (a, b) in enumerate(zip(taps_d, taps_c)):
        scale = b.abs().max().item() + 1e-12
        assert (a.detach().cpu() - b.detach()).abs().max().item() <= 2e-4 * scale, f"tap {k}"
    # gradients: 190 convolution layers deep -- fp32 summation order differences accumulate; 1e-3 of the largest magnitude
    pc, pd = dict(cpu.named_parameters()), dict(dev.named_parameters())
    checked = 0
    for name, p in pc.items():
        if p.grad is None:
            continue
        scale = p.grad.abs().max().item() + 1e-12
        err = (pd[name].grad.cpu() - p.grad).abs().max().item()
        assert err <= 2e-3 * scale, (name, err, scale)
        checked += 1
    assert checked > 500
    scale = xc.grad.abs().max().item()
    assert (xd.grad.cpu() - xc.grad).abs().max().item() <= 2e-3 * scale
