"""Does torch's own global (multi-block, semaphore-based) sum survive hipGraph replay on this stack?  One reduction per graph."""
import torch
torch.manual_seed(0)
for shape, dim in [((768, 4608), 0), ((768, 2448), 0), ((768, 864), 0), ((341, 18432), 0), ((256, 4608), 0), ((8, 256, 4, 13), (0, 2, 3)),
                   ((8, 32, 64, 208), (0, 2, 3)), ((106496, 32), 0)]:
    x = torch.randn(*shape, device="cuda")
    ref = x.double().sum(dim).float()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2): y = x.sum(dim)
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        y = x.sum(dim)
    errs = []
    for it in range(4):
        y.fill_(float("nan"))
        g.replay(); torch.cuda.synchronize()
        errs.append(float(torch.nan_to_num((y - ref).abs(), nan=1e30).max() / ref.abs().max()))
    print(f"[reduce] sum{tuple(shape)} over {dim}: replay errors " + " ".join(f"{e:.1e}" for e in errs), flush=True)
