"""The same 320-node chain of tiny kernels, timed by HIP events; run it under rocprofv3 --kernel-trace to compare the
per-kernel durations the profiler reports with the per-node time the events give."""
import torch
dev = torch.device("cuda:0")
n = 320
x = torch.ones(4096, device=dev)
def fn():
    for _ in range(n):
        x.mul_(1.0001)
fn(); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    fn()
g.replay(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    g.replay()
e1.record(); torch.cuda.synchronize()
print(f"events: {e0.elapsed_time(e1) * 1e3 / 20 / n:.2f} us/node", flush=True)
