"""xpt_adam_step (csrc/xpt_optim.hip, row a14) against the Keras-Adam restatement oracle/ref_adam.py over five steps:
parameters, both moments, the bf16 shadow copy the convolutions read, the in-kernel zero_grad and grad_scale; and the
optimizer's host (CPU tensor) branch against the same restatement.  Reference: model/model_util/optimizers.py:7-13."""
import numpy as np
import pytest
import torch

from xpt_mde_2021_amd.hip.lib import half as _half_dtype

HALF = _half_dtype()      # 16-bit activation dtype of this process: bf16, or fp16 under XPT_HALF=fp16 (tests/test_fp16_build_gpu.py)

from oracle.ref_adam import KerasAdamRef


def make_problem(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    w0 = torch.randn(n, generator=g)
    grads = [torch.randn(n, generator=g) * (10.0 ** float(e)) for e in (-3, 0, -6, 2, -1)]
    grads[2][::7] = 0.0                       # exactly-zero gradients: update = lr_t * m / (sqrt(v) + 1e-7) stays finite
    return w0, grads


def test_keras_adam_first_step_closed_form():
    """t = 1: m = 0.1 g, v = 0.001 g^2, lr_1 = lr sqrt(0.001) / 0.1  =>  step = lr * g / (|g| + eps / sqrt(0.001))."""
    ref = KerasAdamRef(1e-3)
    gval = np.array([2.0, -0.5, 1e-9, 0.0])
    out = ref.apply_gradients(np.zeros(4), gval)
    expect = -1e-3 * gval / (np.abs(gval) + 1e-7 / np.sqrt(1e-3))
    assert np.allclose(out, expect, rtol=2e-5, atol=0)        # hyper-parameters are float32 values (1 - beta_2 = 9.99987e-4)


def test_host_branch_matches_keras_adam():
    """KerasAdam.apply_gradients on CPU tensors (the branch the gloo data-parallel test exercises)."""
    from xpt_mde_2021_amd.model.model_util.optimizers import KerasAdam
    w0, grads = make_problem(1000)
    p = torch.nn.Parameter(w0.clone())
    opt = KerasAdam(1e-4)
    flat = opt.bind([p])
    ref, w = KerasAdamRef(1e-4), w0.double().numpy()
    for g in grads:
        flat.grad[:p.numel()].copy_(g)
        opt.apply_gradients()
        w = ref.apply_gradients(w, g.double().numpy())
        assert np.allclose(flat.data[:p.numel()].double().numpy(), w, rtol=0, atol=2e-7 * np.abs(w).max())
        assert float(flat.grad.abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("n", [4096 + 3, 1 << 20])
@pytest.mark.parametrize("grad_scale", [1.0, 0.125])
def test_adam_kernel_matches_keras_adam(gpu_device, n, grad_scale):
    from xpt_mde_2021_amd.hip import lib as _lib
    lib = _lib.load()
    w0, grads = make_problem(n, seed=n)
    dev = gpu_device
    p, m, v = w0.clone().to(dev), torch.zeros(n, device=dev), torch.zeros(n, device=dev)
    gbuf = torch.zeros(n, device=dev)
    shadow = torch.zeros(n, dtype=HALF, device=dev)
    step = torch.zeros(1, device=dev)
    ref, w = KerasAdamRef(1e-4), w0.double().numpy()
    m_err = np.zeros(n)
    for g in grads:
        gbuf.copy_(g.to(dev) / grad_scale)                      # the kernel multiplies by grad_scale (1 / world size in DP)
        step += 1
        _lib.check(lib.xpt_adam_step(p.data_ptr(), gbuf.data_ptr(), m.data_ptr(), v.data_ptr(), n, step.data_ptr(), 1e-4,
                                     0.9, 0.999, 1e-7, grad_scale, 1, shadow.data_ptr(),
                                     torch.cuda.current_stream().cuda_stream), "xpt_adam_step")
        torch.cuda.synchronize()
        m_prev = np.zeros(n) if ref.m is None else ref.m.copy()
        w = ref.apply_gradients(w, g.double().numpy())
        got = p.double().cpu().numpy()
        # fp32 arithmetic: the update lr_t * m / (sqrt(v) + eps) is <= ~lr * 3.2 per step and carries ~1e-6 relative error
        assert np.abs(got - w).max() < 2e-7 * np.abs(w).max() + 1e-9, np.abs(got - w).max()
        # m = b1 m + (1 - b1) g in fp32: each of the two products and the sum round once (2^-24 each); with cancellation
        # the error is relative to the operands, not to the result
        # (and the fp32 state carries the error of the earlier steps, shrunk by b1 per step)
        m_err = 0.9 * m_err + 2e-7 * (0.9 * np.abs(m_prev) + 0.1 * np.abs(g.double().numpy())) + 1e-30
        assert (np.abs(m.double().cpu().numpy() - ref.m) <= m_err).all()
        assert np.allclose(v.double().cpu().numpy(), ref.v, rtol=4e-6, atol=1e-30)
        assert float(gbuf.abs().max()) == 0.0                   # zero_grad happened in the same pass
        assert torch.equal(shadow, p.to(HALF))        # bf16 shadow = round-to-nearest-even of the updated weight


def test_sgd_constant_host_branch():
    """optimizer_factory("sgd_constant") = tf.optimizers.SGD(lr) (reference optimizers.py:10-11): w <- w - lr g, momentum 0."""
    from xpt_mde_2021_amd.model.model_util.optimizers import KerasSGD, optimizer_factory
    w0, grads = make_problem(1000, seed=3)
    p = torch.nn.Parameter(w0.clone())
    opt = optimizer_factory("sgd_constant", 1e-2)
    assert isinstance(opt, KerasSGD)
    flat = opt.bind([p])
    w = w0.double()
    for g in grads:
        flat.grad[:p.numel()].copy_(g)
        opt.apply_gradients()
        w = w - 1e-2 * g.double()
        assert torch.allclose(flat.data[:p.numel()].double(), w, rtol=0, atol=3e-7 * float(w.abs().max()))
        assert float(flat.grad.abs().max()) == 0.0


@pytest.mark.gpu
@pytest.mark.parametrize("grad_scale", [1.0, 0.125])
def test_sgd_kernel(gpu_device, grad_scale):
    from xpt_mde_2021_amd.hip import lib as _lib
    lib = _lib.load()
    n = 4096 + 3
    w0, grads = make_problem(n, seed=9)
    p, gbuf = w0.clone().to(gpu_device), torch.zeros(n, device=gpu_device)
    shadow = torch.zeros(n, dtype=HALF, device=gpu_device)
    w = w0.clone()
    for g in grads:
        gbuf.copy_(g.to(gpu_device) / grad_scale)
        _lib.check(lib.xpt_sgd_step(p.data_ptr(), gbuf.data_ptr(), n, 1e-2, grad_scale, 1, shadow.data_ptr(),
                                    torch.cuda.current_stream().cuda_stream), "xpt_sgd_step")
        torch.cuda.synchronize()
        w = w - 1e-2 * ((g / grad_scale) * grad_scale)
        assert torch.allclose(p.cpu(), w, rtol=0, atol=3e-7 * float(w.abs().max()))
        assert float(gbuf.abs().max()) == 0.0
        assert torch.equal(shadow.cpu(), p.cpu().to(HALF))
