"""Do two independent kernel chains inside ONE captured hipGraph overlap?  Chains of small launches (the regime of the
batch-8 step: ~5 us of per-node floor) captured serially on one stream vs forked onto two streams (one fork, one join)."""
import sys
import torch

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")


def chain(x, n):
    for _ in range(n):
        x.mul_(1.0001)


def timed(graph, reps=20):
    graph.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


for numel in (1 << 12, 1 << 18, 1 << 22):
    a = torch.ones(numel, device=dev)
    b = torch.ones(numel, device=dev)
    side = torch.cuda.Stream()
    chain(a, 3); chain(b, 3)
    torch.cuda.synchronize()
    g1 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g1):
        chain(a, n)
        chain(b, n)
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2):
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            chain(b, n)
        chain(a, n)
        main.wait_stream(side)
    g3 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g3):
        chain(a, n)
    t1, t2, t3 = timed(g1), timed(g2), timed(g3)
    print(f"numel {numel:8d}: 2 chains serial {t1:8.1f} us ({t1 / (2 * n):.2f}/node), forked {t2:8.1f} us, one chain alone {t3:8.1f} us", flush=True)
