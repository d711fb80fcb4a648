"""GradSink two-pass finishing (more than 256 splits) under hipGraph replay: producer kernel -> flush, captured once."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xpt_mde_2021_amd.hip import ops
dev = torch.device("cuda:0")
sink = ops.GradSink()
cases = [(4608, 768), (2448, 768), (864, 768), (12000, 491), (18432, 341), (300, 64), (1000, 200)]
g = torch.Generator().manual_seed(0)
srcs = [torch.randn(ns * n, generator=g).to(dev) for n, ns in cases]
work = [torch.empty_like(s) for s in srcs]
dsts = [torch.full((n,), float("nan"), device=dev) for n, _ in cases]
scale = torch.ones(1, device=dev)
def step():
    for w, s in zip(work, srcs):
        torch.mul(s, scale, out=w)                       # the "producer": partials = src * scale
    for (n, ns), w, d in zip(cases, work, dsts):
        sink.add(d, w, 0, n, ns, n)
    sink.flush()
side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    step(); step()
torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    step()
for it in range(4):
    scale.fill_(float(it + 1))
    for d in dsts: d.fill_(float("nan"))
    gr.replay(); torch.cuda.synchronize()
    errs = []
    for (n, ns), s, d in zip(cases, srcs, dsts):
        e = (s.double().view(ns, n).sum(0) * (it + 1))
        errs.append(float((d.double() - e).abs().max() / e.abs().max()))
    print(f"[sinkprobe] replay {it}: " + " ".join(f"{e:.1e}" for e in errs), flush=True)
