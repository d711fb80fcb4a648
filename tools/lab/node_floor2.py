"""Which ingredient of the captured training step lifts the per-node time from the 1.6 us of a chain of tiny kernels to
the 4.7 us the step's tiny kernels show?  Adds one suspect at a time to a 320-node chain."""
import ctypes
import sys, os
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
dev = torch.device("cuda:0")
n = 320
x = torch.ones(4096, device=dev)
y = torch.ones(4096, device=dev)
big = torch.ones(1 << 20, device=dev)
big2 = torch.ones(1 << 20, device=dev)


def timed(fn, reps=20, keep=False):
    fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph(keep_graph=True) if keep else torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        fn()
    if keep:
        g.instantiate()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps / n


def plain():
    for _ in range(n):
        x.mul_(1.0001)


def with_memcpy():
    for i in range(n):
        x.mul_(1.0001)
        if i == 100:
            y.copy_(x)                      # contiguous same-dtype D2D: hipMemcpyAsync -> a memcpy node


def with_many_memcpy():
    for i in range(n):
        if i % 2:
            y.copy_(x)
        else:
            x.mul_(1.0001)


def with_alloc():
    t = x
    for i in range(n):
        t = t * 1.0001                      # allocation from the capture pool per node


print(f"plain chain:                 {timed(plain):.2f} us/node", flush=True)
print(f"plain chain, keep_graph:     {timed(plain, keep=True):.2f} us/node", flush=True)
print(f"one memcpy node in it:       {timed(with_memcpy):.2f} us/node", flush=True)
print(f"every other node a memcpy:   {timed(with_many_memcpy):.2f} us/node", flush=True)
print(f"allocating chain:            {timed(with_alloc):.2f} us/node", flush=True)

from xpt_mde_2021_amd.hip import ops, lib as _lib
lib = _lib.load()
a = torch.randn(2, 44, 8, 26, device=dev, dtype=torch.bfloat16).contiguous(memory_format=torch.channels_last)
g_ = torch.ones(44, device=dev); b_ = torch.zeros(44, device=dev); m_ = torch.zeros(44, device=dev); v_ = torch.ones(44, device=dev)


def own_kernel():
    for _ in range(n):
        ops.batchnorm_inference(a, g_, b_, m_, v_, 1e-3)


def own_mixed():
    for i in range(n):
        if i % 2:
            ops.batchnorm_inference(a, g_, b_, m_, v_, 1e-3)
        else:
            x.mul_(1.0001)


with torch.no_grad():
    print(f"own tiny kernel (affine_act): {timed(own_kernel):.2f} us/node", flush=True)
    print(f"own / aten alternating:       {timed(own_mixed):.2f} us/node", flush=True)

# a rand node (philox state update: torch registers a generator with the graph) and a host-side stream sync pattern
gen_x = torch.empty(4096, device=dev)


def with_rand():
    for i in range(n):
        x.mul_(1.0001)
        if i == 100:
            gen_x.uniform_()


print(f"one uniform_() node in it:   {timed(with_rand):.2f} us/node", flush=True)


def bigger():
    for i in range(n):
        torch.mul(big, 1.0001, out=big2)


print(f"4 MiB read + 4 MiB written:  {timed(bigger):.2f} us/node", flush=True)
