"""Cost of a tiny kernel that follows kernels with a real footprint: [one kernel moving S MiB] + [k tiny kernels], repeated;
the slope over k is the per-node cost of the tiny ones in that environment."""
import torch

dev = torch.device("cuda:0")
x = torch.ones(4096, device=dev)
groups = 40


def timed(fn, reps=10):
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
    return e0.elapsed_time(e1) * 1e3 / reps


for mib in (0.25, 4, 32, 128):
    numel = int(mib * (1 << 20) / 4)
    pool = [torch.ones(numel, device=dev) for _ in range(groups + 1)]
    res = []
    for k in (0, 4, 8):
        def fn():
            for gi in range(groups):
                torch.mul(pool[gi], 1.0001, out=pool[gi + 1])
                for _ in range(k):
                    x.mul_(1.0001)
        res.append(timed(fn) / groups)
    print(f"big kernel {mib:6.2f} MiB in + out: group of 1 big = {res[0]:7.2f} us; +4 tiny = {res[1]:7.2f}; +8 tiny = {res[2]:7.2f}"
          f"  -> {(res[2] - res[0]) / 8:.2f} us per tiny kernel", flush=True)
    # tiny kernels that READ a slice of what the big kernel just wrote (cold lines from another XCD's L2 / HBM)
    res2 = []
    for k in (0, 4, 8):
        def fn2():
            for gi in range(groups):
                torch.mul(pool[gi], 1.0001, out=pool[gi + 1])
                for j in range(k):
                    torch.mul(pool[gi + 1][j * 4096:(j + 1) * 4096], 1.0001, out=x)
        res2.append(timed(fn2) / groups)
    print(f"    tiny kernels reading the big kernel's output: {(res2[2] - res2[0]) / 8:.2f} us per tiny kernel", flush=True)
    del pool
