import torch
x = torch.rand(8, 53248, device="cuda")
ref = torch.sort(x, dim=1).values
s = torch.cuda.Stream(); s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): y = torch.sort(x, dim=1).values
torch.cuda.current_stream().wait_stream(s); torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    y = torch.sort(x, dim=1).values
    m = y.gather(1, torch.full((8, 1), 1000, device="cuda"))
for it in range(4):
    x.copy_(torch.rand(8, 53248, device="cuda")); ref = torch.sort(x, dim=1).values
    g.replay(); torch.cuda.synchronize()
    print("[sortprobe] replay", it, "equal:", bool(torch.equal(y, ref)))
