"""Does a captured torch.rand keep drawing fresh numbers on every replay -- also after an EAGER torch.rand between replays?"""
import torch
dev = torch.device("cuda:0")
torch.manual_seed(5)
x = torch.rand(8, device=dev)          # warm-up
torch.cuda.synchronize()
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    u = torch.rand(8, device=dev)
def replay():
    g.replay(); torch.cuda.synchronize(); return tuple(round(v, 6) for v in u[:3].tolist())
print("replays after capture:      ", len({replay() for _ in range(6)}), "distinct of 6")
e = torch.rand(8, device=dev); torch.cuda.synchronize()
print("replays after an eager rand:", len({replay() for _ in range(6)}), "distinct of 6")
buf = torch.zeros(9, device=dev); buf[1:].uniform_(); torch.cuda.synchronize()
print("replays after uniform_():   ", len({replay() for _ in range(6)}), "distinct of 6")
