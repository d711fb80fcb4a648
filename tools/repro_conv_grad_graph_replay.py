"""Stand-alone reproducer (no code of this repo in the network) of the ROCm 7.0 / torch 2.10 defect described in
DESIGN.md section 6: gradients of convolution biases / MIOpen bf16 weight gradients are garbage from the SECOND replay
of a captured hipGraph.  Usage: python tools/repro_conv_grad_graph_replay.py <variant>  (variant: plain | sepbias | det |
nomiopen, combinable with '-')."""
import sys, torch
import torch.nn as nn, torch.nn.functional as F
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
from xpt_mde_2021_amd.utils import synthetic_data as sd
variant = sys.argv[1]
if "det" in variant: torch.backends.cudnn.deterministic = True
if "nomiopen" in variant: torch.backends.cudnn.enabled = False
class Conv(nn.Module):
    def __init__(s, ci, co, k, st, sepbias):
        super().__init__(); s.c = nn.Conv2d(ci, co, k, st, k // 2, bias=not sepbias); s.b = nn.Parameter(torch.zeros(co)) if sepbias else None
    def forward(s, x):
        y = s.c(x)
        if s.b is not None: y = y + s.b.view(1, -1, 1, 1).to(y.dtype)
        return F.leaky_relu(y, 0.1)
torch.manual_seed(0)
spec = [(15, 32, 5, 2), (32, 32, 5, 2), (32, 64, 3, 2), (64, 128, 3, 2), (128, 256, 3, 2), (256, 256, 3, 2), (256, 256, 3, 1), (256, 256, 3, 1), (256, 24, 1, 1)]
net = nn.Sequential(*[Conv(*s, sepbias=("sepbias" in variant)) for s in spec]).cuda().to(memory_format=torch.channels_last)
named = list(net.named_parameters())
xs = [torch.randn(8, 15, 128, 416, device="cuda").contiguous(memory_format=torch.channels_last) for _ in range(2)]
static_x = xs[0].clone()
def step():
    for _, p in named: p.grad = None
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=("bf16" in variant)):
        out = net(static_x)
    (out.float() ** 2).mean().backward()
ref = []
for x in xs:
    static_x.copy_(x); step(); torch.cuda.synchronize(); ref.append({n: p.grad.clone() for n, p in named})
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3): step()
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g): step()
bad = []
for it in range(4):
    static_x.copy_(xs[it % 2]); g.replay(); torch.cuda.synchronize()
    for n, p in named:
        r = ref[it % 2][n]
        err = float(torch.nan_to_num((p.grad - r).abs().float(), nan=1e38, posinf=1e38).max() / r.abs().max())
        if err > 2e-2: bad.append((it, n, "%.2e" % err))
print(variant, "BAD:", bad if bad else "none")
