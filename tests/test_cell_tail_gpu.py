"""The fused NASNet cell tail (csrc/xpt_celltail.hip: pools + adds + concat + the consumers' relu in one launch, and the
gradient fan-in + relu mask + pool adjoints + concat split in one backward launch) against the same function composed
from PyTorch ops in fp32 -- the keras layers it replaces: AveragePooling2D(3, 1, 'same') (divisor without the padding),
add, concatenate, Activation('relu') of tensorflow.keras.applications.nasnet's _normal_a_cell / _reduction_a_cell
(the backbone the reference instantiates at model/build_model/pretrained_nets.py:11-44)."""
import pytest
import torch

from xpt_mde_2021_amd.hip.lib import half as _half_dtype

HALF = _half_dtype()      # 16-bit activation dtype of this process: bf16, or fp16 under XPT_HALF=fp16 (tests/test_fp16_build_gpu.py)
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

NORMAL = (((0, 0, 1.0),), ((1, 0, 1.0),), ((2, 0, 1.0),), ((3, 1, 1.0), (0, 0, 1.0)), ((0, 1, 2.0),), ((4, 0, 1.0),))
REDUCTION = (((0, 0, 1.0),), ((1, 0, 1.0),), ((0, 0, 1.0), (2, 1, 1.0)), ((3, 0, 1.0),))


def reference(spec, inputs):
    slices = []
    for terms in spec:
        acc = 0
        for i, pooled, scale in terms:
            t = inputs[i]
            if pooled:
                t = F.avg_pool2d(t, 3, 1, 1, count_include_pad=False)
            acc = acc + scale * t
        slices.append(acc)
    return torch.relu(torch.cat(slices, dim=1))


@pytest.mark.parametrize("spec,n_in", [(NORMAL, 5), (REDUCTION, 4)])
@pytest.mark.parametrize("dtype,F_,H,W", [(torch.float32, 12, 5, 7), (HALF, 44, 16, 52), (HALF, 88, 8, 26),
                                          (HALF, 22, 9, 13), (HALF, 11, 6, 5), (torch.float32, 6, 1, 3)])
def test_cell_tail_matches_composed_ops(gpu_device, spec, n_in, dtype, F_, H, W):
    from xpt_mde_2021_amd.hip import ops
    dev = gpu_device
    g = torch.Generator().manual_seed(F_ * 100 + H)
    B = 2
    host = [torch.randn(B, F_, H, W, generator=g).to(dtype) for _ in range(n_in)]
    # input 1 is a channel slice of a wider tensor (what torch.cat's backward / a multi-layer launch hands out)
    wide = torch.randn(B, 2 * F_, H, W, generator=g).to(dtype)
    host[1] = wide[:, F_:]
    ref_in = [t.float().clone().requires_grad_(True) for t in host]
    ref = reference(spec, ref_in)
    # three consumers with different gradients, one of them unused (None gradient)
    w0 = torch.randn(ref.shape, generator=g).to(dtype)
    w1 = torch.randn(ref.shape, generator=g).to(dtype)
    (ref * w0.float() + ref * w1.float()).sum().backward()

    wide_d = wide.to(dev).contiguous(memory_format=torch.channels_last)
    dev_in = [t.to(dev).contiguous(memory_format=torch.channels_last) for t in host]
    dev_in[1] = wide_d[:, F_:]
    dev_in = [t.detach().requires_grad_(True) for t in dev_in]
    outs = ops.cell_tail(spec, dev_in, 3)
    assert len(outs) == 3 and all(o.data_ptr() == outs[0].data_ptr() for o in outs)
    ((outs[0].float() * w0.to(dev).float()).sum() + (outs[2].float() * w1.to(dev).float()).sum()).backward()
    torch.cuda.synchronize()

    tol = 1e-5 if dtype == torch.float32 else 1.2e-2          # bf16: one rounding of the result (2^-8) and of each gradient
    scale = ref.abs().max().item() + 1e-12
    assert (outs[0].float().cpu() - ref.detach()).abs().max().item() <= tol * scale
    # the relu mask must agree wherever the reference is not within rounding of zero
    for i in range(n_in):
        gr = ref_in[i].grad
        gd = dev_in[i].grad.float().cpu()
        gs = gr.abs().max().item() + 1e-12
        # elements whose pre-activation is within rounding of 0 may flip their mask in bf16: compare away from them
        err = (gd - gr).abs()
        if dtype == torch.float32:
            assert err.max().item() <= 1e-5 * gs, f"input {i}"
        else:
            assert (err > 3e-2 * gs).float().mean().item() < 2e-3, f"input {i}: too many gradient elements off"
            assert err.median().item() <= 1e-2 * gs


def test_cell_tail_without_grad_and_single_alias(gpu_device):
    from xpt_mde_2021_amd.hip import ops
    dev = gpu_device
    x = [torch.randn(1, 8, 4, 4, device=dev) for _ in range(4)]
    with torch.no_grad():
        out, = ops.cell_tail(REDUCTION, x, 1)
    assert torch.allclose(out, reference(REDUCTION, x), atol=1e-6)
    assert out.is_contiguous(memory_format=torch.channels_last)


@pytest.mark.parametrize("dtype,C,H,W", [(torch.float32, 6, 6, 10), (torch.float32, 5, 7, 9), (HALF, 32, 16, 52),
                                         (HALF, 44, 9, 14), (HALF, 11, 4, 6)])
def test_adjust_gather_matches_pad_crop_stride(gpu_device, dtype, C, H, W):
    """keras _adjust_block, spatial mode: p[::2, ::2] and ZeroPadding2D(((0,1),(0,1))) -> Cropping2D(((1,0),(1,0))) -> [::2, ::2]."""
    from xpt_mde_2021_amd.hip import ops
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(2, C, H, W, generator=g).to(dtype)
    xr = x.float().requires_grad_(True)
    r1 = xr[:, :, ::2, ::2]
    r2 = F.pad(xr, (0, 1, 0, 1))[:, :, 1:, 1:][:, :, ::2, ::2]
    w1, w2 = torch.randn(r1.shape, generator=g).to(dtype), torch.randn(r2.shape, generator=g).to(dtype)
    ((r1 * w1.float()).sum() + (r2 * w2.float()).sum()).backward()
    xd = x.to(gpu_device).contiguous(memory_format=torch.channels_last).detach().requires_grad_(True)
    d1, d2 = ops.adjust_gather(xd)
    assert d1.shape == r1.shape and d2.shape == r2.shape
    assert torch.equal(d1.float().cpu(), r1.detach()) and torch.equal(d2.float().cpu(), r2.detach())
    ((d1.float() * w1.to(gpu_device).float()).sum() + (d2.float() * w2.to(gpu_device).float()).sum()).backward()
    assert torch.equal(xd.grad.float().cpu(), xr.grad.to(dtype).float())
    # one half unused: its gradient is None
    xd2 = x.to(gpu_device).contiguous(memory_format=torch.channels_last).detach().requires_grad_(True)
    e1, _ = ops.adjust_gather(xd2)
    (e1.float() * w1.to(gpu_device).float()).sum().backward()
    only1 = torch.zeros_like(xr.grad)
    only1[:, :, ::2, ::2] = w1.float()
    assert torch.equal(xd2.grad.float().cpu(), only1)


@pytest.mark.parametrize("dtype,C,H,W", [(torch.float32, 6, 8, 12), (torch.float32, 5, 7, 9), (HALF, 32, 16, 52),
                                         (HALF, 22, 9, 13), (HALF, 88, 8, 26)])
def test_pool_pair_matches_padded_library_pools(gpu_device, dtype, C, H, W):
    from xpt_mde_2021_amd.hip import ops
    from xpt_mde_2021_amd.model.build_model.pretrained_nets import correct_pad
    g = torch.Generator().manual_seed(C * 3 + W)
    x = torch.randn(2, C, H, W, generator=g).to(dtype)
    pads = correct_pad(H, W, 3)
    (pt, pb), (pl, pr) = pads
    xr = x.float().requires_grad_(True)
    xp = F.pad(xr, (pl, pr, pt, pb))
    mr, ar = F.max_pool2d(xp, 3, 2), F.avg_pool2d(xp, 3, 2)
    wm, wa = torch.randn(mr.shape, generator=g).to(dtype), torch.randn(ar.shape, generator=g).to(dtype)
    ((mr * wm.float()).sum() + (ar * wa.float()).sum()).backward()
    xd = x.to(gpu_device).contiguous(memory_format=torch.channels_last).detach().requires_grad_(True)
    md, ad = ops.pool_pair(xd, pads)
    assert md.shape == mr.shape and ad.shape == ar.shape
    assert torch.equal(md.float().cpu(), mr.detach())                          # a maximum is one of the inputs: exact
    tol = 1e-6 if dtype == torch.float32 else 8e-3
    assert (ad.float().cpu() - ar.detach()).abs().max().item() <= tol * (ar.abs().max().item() + 1e-12)
    ((md.float() * wm.to(gpu_device).float()).sum() + (ad.float() * wa.to(gpu_device).float()).sum()).backward()
    scale = xr.grad.abs().max().item()
    assert (xd.grad.float().cpu() - xr.grad).abs().max().item() <= (1e-5 if dtype == torch.float32 else 1e-2) * scale


@pytest.mark.parametrize("dtype,C,H,W", [(torch.float32, 6, 8, 12), (HALF, 24, 9, 13), (HALF, 88, 8, 26)])
def test_pool_pair_with_two_consumers_of_the_maximum(gpu_device, dtype, C, H, W):
    """pool_pair(split_mp=True): the max-pooled tensor as two aliases whose gradients the backward adds on load
    (xpt_pool_pair_bwd2) == one tensor with both gradients added beforehand; one consumer silent: the other alone."""
    from xpt_mde_2021_amd.hip import ops
    from xpt_mde_2021_amd.model.build_model.pretrained_nets import correct_pad
    g = torch.Generator().manual_seed(C + H)
    x = torch.randn(2, C, H, W, generator=g).to(dtype)
    pads = correct_pad(H, W, 3)
    xa = x.to(gpu_device).contiguous(memory_format=torch.channels_last).detach().requires_grad_(True)
    xb = xa.detach().clone().requires_grad_(True)
    m1, m2, ap = ops.pool_pair(xa, pads, split_mp=True)
    mr, ar = ops.pool_pair(xb, pads)
    assert torch.equal(m1, mr) and torch.equal(m2, mr) and torch.equal(ap, ar)
    w1, w2, wa = (torch.randn(mr.shape, generator=g).to(gpu_device, dtype) for _ in range(3))
    torch.autograd.backward([m1, m2, ap], [w1, w2, wa])
    torch.autograd.backward([mr, ar], [(w1.float() + w2.float()).to(dtype), wa])
    scale = float(xb.grad.abs().max())
    assert float((xa.grad.float() - xb.grad.float()).abs().max()) <= (1e-6 if dtype == torch.float32 else 2 ** -7) * scale
    xc = xa.detach().clone().requires_grad_(True)
    xd = xa.detach().clone().requires_grad_(True)
    n1, n2, _ = ops.pool_pair(xc, pads, split_mp=True)
    n2.backward(w2)                                          # only the second alias receives a gradient
    q, _ = ops.pool_pair(xd, pads)
    q.backward(w2)
    assert torch.equal(xc.grad, xd.grad)
