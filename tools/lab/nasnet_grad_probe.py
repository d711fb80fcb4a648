"""Who is the outlier in the tap-4 input gradient (tests/test_ref_nasnet.py): the restatement in fp64, the restatement in
fp32, the product module on the CPU, or the product module on the GPU?  All on the same weights and image."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import ref_nasnet
from xpt_mde_2021_amd.model.build_model import pretrained_nets as pn

weights = ref_nasnet.random_weights(7)
g = torch.Generator().manual_seed(5)
image = (torch.rand((2, 64, 192, 3), generator=g) * 2 - 1)
p = torch.randn((2, 2, 6, 1056), generator=torch.Generator().manual_seed(24))
grads = {}
for name, dt in (("restatement fp64", torch.float64), ("restatement fp32", torch.float32)):
    x = image.to(dt).clone().requires_grad_(True)
    out = ref_nasnet.forward({k: v.to(dt) for k, v in weights.items()}, x)
    (out[4] * p.to(dt)).sum().backward()
    grads[name] = x.grad.double()
for name, dev in (("module cpu fp32", "cpu"), ("module gpu fp32", "cuda:0")):
    if dev != "cpu" and not torch.cuda.is_available():
        continue
    net = pn.NASNetMobileEncoder().float().eval()
    pn.load_keras_weights(net, {k: v.numpy() for k, v in weights.items()})
    net = net.to(dev)
    x = image.permute(0, 3, 1, 2).contiguous().to(dev).requires_grad_(True)
    out = net(x)
    (out[4].permute(0, 2, 3, 1) * p.to(dev)).sum().backward()
    grads[name] = x.grad.permute(0, 2, 3, 1).double().cpu()
names = list(grads)
scale = float(grads[names[0]].abs().max())
print("torch", torch.__version__, "threads", torch.get_num_threads(), "mkldnn", torch.backends.mkldnn.is_available())
for i, a in enumerate(names):
    for b in names[i + 1:]:
        d = (grads[a] - grads[b]).abs()
        print(f"{a:18s} vs {b:18s}: max {float(d.max()) / scale:.2e}, share > 2e-3: {float((d > 2e-3 * scale).double().mean()):.2e}")
