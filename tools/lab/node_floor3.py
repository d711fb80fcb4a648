"""One suspect launch inside a 320-node chain of tiny kernels: does it change the per-node time of the WHOLE graph?"""
import sys, os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dev = torch.device("cuda:0")
n = 320
x = torch.ones(4096, device=dev)


def timed(fn, reps=20):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / n


def chain_with(extra, every=None):
    def fn():
        for i in range(n):
            x.mul_(1.0001)
            if extra is not None and (i == 100 if every is None else i % every == 0):
                extra()
    return fn


A = torch.randn(256, 64, device=dev, dtype=torch.bfloat16)
Bm = torch.randn(64, 64, device=dev, dtype=torch.bfloat16)
Af = torch.randn(256, 64, device=dev)
Bf = torch.randn(64, 64, device=dev)
t1 = torch.randn(2, 8, 16, 16, device=dev)
img = torch.randn(2, 3, 32, 32, device=dev).contiguous(memory_format=torch.channels_last)
wconv = torch.randn(8, 3, 3, 3, device=dev)
lst_a = [torch.ones(64, device=dev) for _ in range(4)]
lst_b = [torch.ones(64, device=dev) for _ in range(4)]
suspects = {
    "nothing": None,
    "torch.mm bf16 (hipBLASLt/rocBLAS)": lambda: torch.mm(A, Bm),
    "torch.mm fp32": lambda: torch.mm(Af, Bf),
    "torch.cat": lambda: torch.cat([t1, t1], dim=1),
    "clamp": lambda: t1.clamp(min=0.0),
    "sum (reduction)": lambda: t1.sum(),
    "_foreach_copy_": lambda: torch._foreach_copy_(lst_a, lst_b),
    "F.conv2d (MIOpen)": lambda: torch.nn.functional.conv2d(img, wconv, padding=1),
}
with torch.no_grad():
    for name, extra in suspects.items():
        print(f"{name:36s}: {timed(chain_with(extra)):.2f} us/node", flush=True)
    print(f"{'torch.mm bf16 every 16th node':36s}: {timed(chain_with(suspects['torch.mm bf16 (hipBLASLt/rocBLAS)'], 16)):.2f} us/node (+{n // 16} GEMMs)", flush=True)

from xpt_mde_2021_amd.hip import ops, conv as hconv
src = torch.rand(2, 1, 32, 64, 3, device=dev); depth = torch.rand(2, 32, 64, 1, device=dev) + 1.0
T = torch.eye(4, device=dev).repeat(2, 1, 1, 1); K = torch.tensor([[40., 0, 32], [0, 40., 16], [0, 0, 1]], device=dev).repeat(2, 1, 1)
tgt = torch.rand(2, 32, 64, 3, device=dev)
own = {
    "own: photo_fused (dynamic LDS)": lambda: ops.photo_fused(src, depth, T, K, tgt, 1),
    "own: depthwise conv": lambda: ops.depthwise_conv2d(torch.randn(2, 16, 8, 8, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last), torch.randn(16, 1, 3, 3, device=dev), 1, (1, 1, 1, 1)),
}
with torch.no_grad():
    for name, extra in own.items():
        try:
            print(f"{name:36s}: {timed(chain_with(extra)):.2f} us/node", flush=True)
        except Exception as e:       # noqa: BLE001
            print(f"{name:36s}: failed: {type(e).__name__}: {e}", flush=True)
