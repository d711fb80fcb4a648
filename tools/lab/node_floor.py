"""What sets the per-node time of a serial hipGraph chain of tiny kernels?  (a) kernel variety (instruction fetch of a
kernel that is not in the instruction caches), (b) the memory each node touches (address translation / cold data)."""
import torch

dev = torch.device("cuda:0")
n = 320


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


x = torch.ones(4096, device=dev)
xi = torch.ones(4096, device=dev, dtype=torch.int32)
xh = torch.ones(4096, device=dev, dtype=torch.bfloat16)
xd = torch.ones(4096, device=dev, dtype=torch.float64)
ops = [lambda: x.mul_(1.0001), lambda: x.add_(1e-6), lambda: x.abs_(), lambda: x.clamp_(0.5, 2.0), lambda: x.sqrt_(),
       lambda: x.sigmoid_(), lambda: x.tanh_(), lambda: x.exp_(), lambda: xi.add_(1), lambda: xi.bitwise_and_(255),
       lambda: xh.mul_(1.0), lambda: xh.add_(0.0), lambda: xh.abs_(), lambda: xd.mul_(1.0), lambda: xd.add_(0.0),
       lambda: x.neg_(), lambda: x.floor_(), lambda: x.fill_(1.0), lambda: xi.fill_(1), lambda: xh.fill_(1.0),
       lambda: x.sin_(), lambda: x.cos_(), lambda: x.erf_(), lambda: x.log1p_(), lambda: xh.sigmoid_(), lambda: xh.tanh_(),
       lambda: xd.abs_(), lambda: xd.sqrt_(), lambda: xi.mul_(1), lambda: xi.neg_(), lambda: x.reciprocal_(), lambda: x.relu_()]
for k in (1, 2, 4, 8, 16, 32):
    def fn():
        for i in range(n):
            ops[i % k]()
    print(f"variety {k:2d} distinct kernels: {timed(fn):.2f} us/node", flush=True)

# (b) one kernel, every node on its own tensor; tensors spread over a large pool
for numel, count in ((4096, 320), (1 << 16, 320), (1 << 20, 320)):
    pool = [torch.ones(numel, device=dev) for _ in range(count)]
    def fn():
        for i in range(n):
            pool[i % count].mul_(1.0001)
    print(f"own tensor per node, {numel * 4 / 1024:.0f} KiB each: {timed(fn):.2f} us/node", flush=True)
    def fn2():
        for i in range(n):
            pool[0].mul_(1.0001)
    print(f"  same tensor every node: {timed(fn2):.2f} us/node", flush=True)
    del pool
# (c) dependent chain through different tensors: y_{i+1} = y_i * c (out=) -- the producer's lines are dirty in another XCD's L2
for numel in (4096, 1 << 16, 1 << 20):
    pool = [torch.ones(numel, device=dev) for _ in range(n + 1)]
    def fn():
        for i in range(n):
            torch.mul(pool[i], 1.0001, out=pool[i + 1])
    print(f"producer -> consumer chain, {numel * 4 / 1024:.0f} KiB: {timed(fn):.2f} us/node", flush=True)
    del pool
